#!/usr/bin/env python3
"""Pre-packed (MFMA-fragment-order) weights: packed_kernel configs vs the product kernels (probe, GPU only)."""
import ctypes, json, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from swiftllm_amd import _hip
lib = ctypes.CDLL(os.path.join(here, "libgemm_probe.so"))
lib.probe_packed.argtypes = [ctypes.c_int] * 2 + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]

def pack(w):      # [N, K] -> [N/32][K/16][2 (k half)][32 rows][8]
    N, K = w.shape
    return w.view(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()

def bench(fn, iters=100):
    for i in range(5): fn(i)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for i in range(iters): fn(i)
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1e3 / iters

SHAPES = {"qkv": (6144, 4096, (4, 2)), "o": (4096, 4096, (8, 4)), "down": (4096, 14336, (8,)),
          "up_gate": (28672, 4096, (1,)), "lm_head": (128256, 4096, (1,))}
CONFIGS = [(2, 2), (3, 2), (4, 2), (2, 3), (3, 3), (2, 4)]
M = 32
st = torch.cuda.current_stream().cuda_stream
for name, (N, K, splits) in SHAPES.items():
    copies = max(2, min(8, int(2e9 // (N * K * 2))))
    ws = [torch.empty(N, K, dtype=torch.bfloat16, device="cuda").normal_(0, 0.02) for _ in range(copies)]
    wp = [pack(w) for w in ws]
    x = torch.randn(M, K, device="cuda").bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    slabs = torch.empty(16 * M * N, dtype=torch.float32, device="cuda")
    ref = x.float() @ ws[99 % copies].float().t()
    for ks in splits:
        if ks == 1:
            t = bench(lambda i: _hip.call("swl_gemm_skinny", out.data_ptr(), x.data_ptr(), ws[i % copies].data_ptr(), 0, 0,
                                          M, N, K, K, N, 1, 1, st))
        else:
            t = bench(lambda i: _hip.call("swl_gemm_skinny_partial", slabs.data_ptr(), slabs.numel() * 4, x.data_ptr(),
                                          ws[i % copies].data_ptr(), M, N, K, K, ks, 1, st))
        print(json.dumps({"shape": name, "ks": ks, "cfg": "product", "us": round(t, 2), "TBps": round(N * K * 2 / t / 1e6, 2)}), flush=True)
        for d, occ in CONFIGS:
            if (K // ks) // 128 < d - 1:
                continue
            dst = out if ks == 1 else slabs
            def run(i):
                rc = lib.probe_packed(d, occ, dst.data_ptr(), x.data_ptr(), wp[i % copies].data_ptr(), M, N, K, ks, st)
                assert rc == 0, rc
            t = bench(run)
            got = out.float() if ks == 1 else slabs[:ks * M * N].view(ks, M, N).sum(0)
            err = (got - ref).abs().max().item()
            print(json.dumps({"shape": name, "ks": ks, "cfg": f"packed_d{d}_occ{occ}", "us": round(t, 2),
                              "TBps": round(N * K * 2 / t / 1e6, 2), "max_err": round(err, 5)}), flush=True)
    del ws, wp
