#!/usr/bin/env python3
"""prefetch_mall_probe.py — does reading a weight matrix ahead of time (swl_cache_prefetch) make the decode GEMM that
streams it faster? For each projection shape: time [GEMM] cold (cycling through copies far larger than the 256 MiB
Infinity Cache), [prefetch ; GEMM] back to back on one stream (serial: what the prefetch itself costs is reported too),
and [prefetch on a side stream || a stand-in HBM-bound kernel on the main stream ; GEMM]."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from swiftllm_amd import _hip

SHAPES = {"qkv": (6144, 4096), "o": (4096, 4096), "up_gate": (28672, 4096), "down": (4096, 14336)}


def ev_time(fn, iters):
    for i in range(4):
        fn(i)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for i in range(iters):
        fn(i)
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--wgs", type=int, default=64)
    ap.add_argument("--frac", type=float, default=1.0, help="fraction of the matrix to prefetch")
    a = ap.parse_args()
    dt = torch.bfloat16
    code = _hip.dtype_code(dt)
    st = lambda: torch.cuda.current_stream().cuda_stream     # noqa: E731
    for name, (N, K) in SHAPES.items():
        copies = max(4, int(3e9 // (N * K * 2)))
        ws = []
        for _ in range(copies):
            w = torch.empty(N, K, dtype=dt, device="cuda").normal_(0, 0.02)
            wp = torch.empty_like(w)
            _hip.call("swl_gemm_pack_weight", wp.data_ptr(), w.data_ptr(), N, K, code, st())
            ws.append(wp)
            del w
        x = torch.randn(32, K, device="cuda").to(dt)
        out = torch.empty(32, N, dtype=dt, device="cuda")
        wsp = torch.empty(16 * 32 * N, dtype=torch.float32, device="cuda")
        nbytes = int(N * K * 2 * a.frac) // 16 * 16

        def gemm(i):
            _hip.call("swl_gemm_skinny_packed", out.data_ptr(), x.data_ptr(), ws[i % copies].data_ptr(), wsp.data_ptr(),
                      wsp.numel() * 4, 32, N, K, K, N, 0, code, st())

        def pre(i):
            _hip.call("swl_cache_prefetch", ws[i % copies].data_ptr(), nbytes, a.wgs, st())

        t_gemm = ev_time(gemm, a.iters)
        t_pre = ev_time(pre, a.iters)
        t_both = ev_time(lambda i: (pre(i), gemm(i)), a.iters)
        # prefetch of copy i+1 on a side stream WHILE the GEMM of copy i runs on the main stream
        side = torch.cuda.Stream()

        def overlapped(i):
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                _hip.call("swl_cache_prefetch", ws[(i + 1) % copies].data_ptr(), nbytes, a.wgs, side.cuda_stream)
            gemm(i)
            main.wait_stream(side)
        t_ovl = ev_time(overlapped, a.iters)
        print(json.dumps(dict(shape=name, MB=round(N * K * 2 / 1e6, 1), prefetch_MB=round(nbytes / 1e6, 1), wgs=a.wgs,
                              gemm_cold_us=round(t_gemm, 2), prefetch_us=round(t_pre, 2),
                              prefetch_then_gemm_us=round(t_both, 2), gemm_after_prefetch_us=round(t_both - t_pre, 2),
                              gemm_with_next_prefetch_alongside_us=round(t_ovl, 2))), flush=True)
        del ws


if __name__ == "__main__":
    main()
