#!/usr/bin/env python3
"""Does touching the first tiles of W in a preceding kernel shorten the skinny GEMM? (GPU only, probe)"""
import ctypes, json, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from swiftllm_amd import _hip
lib = ctypes.CDLL(os.path.join(here, "libgemm_probe.so"))
lib.probe_prefetch.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]

def bench(fn, iters=100):
    for i in range(5): fn(i)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for i in range(iters): fn(i)
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1e3 / iters

M = 32
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for name, (N, K, silu) in {"up_gate": (28672, 4096, True), "qkv": (6144, 4096, False), "o": (4096, 4096, False)}.items():
    copies = 8
    ws = [torch.empty(N, K, dtype=torch.bfloat16, device="cuda").normal_(0, 0.02) for _ in range(copies)]
    x = torch.randn(M, K, device="cuda").bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    slabs = torch.empty(16 * M * N, dtype=torch.float32, device="cuda")
    ks = _hip.load().swl_gemm_skinny_choose_splits(N, K)
    def gemm(i):
        w = ws[i % copies]
        if silu:
            _hip.call("swl_gemm_skinny_silu_gate", out.data_ptr(), x.data_ptr(), w.data_ptr(), M, N // 2, K, K, N // 2, 1, st)
        elif ks > 1:
            _hip.call("swl_gemm_skinny_partial", slabs.data_ptr(), slabs.numel() * 4, x.data_ptr(), w.data_ptr(), M, N, K, K, ks, 1, st)
        else:
            _hip.call("swl_gemm_skinny", out.data_ptr(), x.data_ptr(), w.data_ptr(), 0, 0, M, N, K, K, N, 1, 1, st)
    res = {"shape": name, "gemm_us": round(bench(gemm), 2)}
    for bpr in (512, 1024, 2048):
        def pf(i):
            lib.probe_prefetch(ws[i % copies].data_ptr(), K * 2, N, 128, bpr, sink.data_ptr(), st)
        def both(i):
            pf(i); gemm(i)
        t_pf, t_both = bench(pf), bench(both)
        res[f"pf{bpr}_us"] = round(t_pf, 2); res[f"pf{bpr}+gemm_us"] = round(t_both, 2)
        res[f"gemm_after_pf{bpr}_us"] = round(t_both - t_pf, 2)
    print(json.dumps(res), flush=True)
