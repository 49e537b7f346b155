#!/usr/bin/env python3
"""gemm_micro.py — decode-sized projections of Llama-3-8B: hipBLASLt (F.linear) vs swl_gemm_skinny (GPU).
Weights cycle through `copies` distinct matrices so every launch streams from HBM (not L2/MALL)."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swiftllm_amd import _hip

SHAPES = {"qkv": (6144, 4096), "o": (4096, 4096), "up_gate": (28672, 4096), "down": (4096, 14336),
          "lm_head": (128256, 4096), "q": (4096, 4096), "kv": (1024, 4096)}


def bench(fn, iters):
    for i in range(8):
        fn(i)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for i in range(iters):
        fn(i)
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=32)
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--shapes", default="qkv,o,up_gate,down,lm_head")
    ap.add_argument("--tunable", action="store_true", help="let PyTorch TunableOp pick the BLAS solution")
    a = ap.parse_args()
    if a.tunable:
        import torch.cuda.tunable as tun
        tun.enable(True); tun.tuning_enable(True); tun.set_max_tuning_duration(200); tun.set_rotating_buffer_size(1024)
        tun.set_filename('/tmp/tunableop.csv')
    dtype = getattr(torch, a.dtype)
    code = _hip.dtype_code(dtype)
    for name in a.shapes.split(","):
        N, K = SHAPES[name]
        copies = max(2, min(16 if a.m <= 32 else 8, int(2e9 // (N * K * 2))))
        ws = [torch.empty(N, K, dtype=dtype, device="cuda").normal_(0, 0.02) for _ in range(copies)]
        x = torch.randn(a.m, K, device="cuda").to(dtype)
        out = torch.empty(a.m, N, dtype=dtype, device="cuda")
        res = {"shape": name, "M": a.m, "N": N, "K": K, "MB": round(N * K * 2 / 1e6, 1)}
        t = bench(lambda i: torch.nn.functional.linear(x, ws[i % copies]), a.iters)
        res["blas_us"] = round(t, 2); res["blas_TBps"] = round(N * K * 2 / t / 1e6, 2)
        wsp = torch.empty(16 * a.m * N, dtype=torch.float32, device="cuda")
        if a.m > 32:
            wps = []
            for w_ in ws:
                wp_ = torch.empty_like(w_)
                _hip.call("swl_gemm_pack_weight", wp_.data_ptr(), w_.data_ptr(), N, K, code, _hip.stream())
                wps.append(wp_)
            for ks in (0, 1, 2, 4, 8):
                if ks and K % (128 * ks):
                    continue
                def run_mid(i, ks=ks):
                    _hip.call("swl_gemm_packed_mid", out.data_ptr(), x.data_ptr(), wps[i % copies].data_ptr(), wsp.data_ptr(),
                              wsp.numel() * 4, a.m, N, K, K, N, ks, code, _hip.stream())
                t = bench(run_mid, a.iters)
                res[f"pmid_ks{ks}_us"] = round(t, 2); res[f"pmid_ks{ks}_TBps"] = round(N * K * 2 / t / 1e6, 2)
            del wps
        for ks in (0, 1, 2, 4, 8, 16):
            if a.m > 32:
                break
            if ks and (K % (128 * ks) or (ks > 1 and N > 32768)):
                continue
            def run(i, ks=ks):
                _hip.call("swl_gemm_skinny", out.data_ptr(), x.data_ptr(), ws[i % copies].data_ptr(), wsp.data_ptr(),
                          wsp.numel() * 4, a.m, N, K, K, N, ks, code, _hip.stream())
            t = bench(run, a.iters)
            res[f"swl_ks{ks}_us"] = round(t, 2); res[f"swl_ks{ks}_TBps"] = round(N * K * 2 / t / 1e6, 2)
        print(json.dumps(res), flush=True)
        del ws


if __name__ == "__main__":
    main()
