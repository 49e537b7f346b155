#!/usr/bin/env python3
"""prefill_gemm_probe.py — the four projections of a Llama-3-8B layer at prefill size (M tokens) on hipBLASLt through
F.linear: the heuristic's default solution vs the best one PyTorch TunableOp finds (hipBLASLt + rocBLAS candidates).
Answers: is the 85 % of prefill time spent in the vendor GEMM (profiles/r02b) leaving speed on the table by solution
CHOICE alone?"""
import argparse, json, os, time
import torch
import torch.nn.functional as F

SHAPES = {"qkv": (6144, 4096), "o": (4096, 4096), "up_gate": (28672, 4096), "down": (4096, 14336)}


def bench(fn, iters):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=32768)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--tune-ms", type=int, default=300)
    a = ap.parse_args()
    dt = getattr(torch, a.dtype)
    ws = {n: torch.empty(N, K, dtype=dt, device="cuda").normal_(0, 0.02) for n, (N, K) in SHAPES.items()}
    xs = {K: torch.randn(a.m, K, device="cuda").to(dt) for K in (4096, 14336)}
    base = {}
    for n, (N, K) in SHAPES.items():
        base[n] = bench(lambda: F.linear(xs[K], ws[n]), a.iters)
    import torch.cuda.tunable as tun
    tun.enable(True); tun.tuning_enable(True); tun.set_max_tuning_duration(a.tune_ms); tun.set_max_tuning_iterations(20)
    tun.set_filename("/tmp/tunableop_prefill.csv")
    tot_b = tot_t = 0.0
    for n, (N, K) in SHAPES.items():
        t0 = time.time()
        F.linear(xs[K], ws[n]); torch.cuda.synchronize()
        tune_s = time.time() - t0
        t = bench(lambda: F.linear(xs[K], ws[n]), a.iters)
        fl = 2.0 * a.m * N * K
        tot_b += base[n]; tot_t += t
        print(json.dumps({"shape": n, "M": a.m, "N": N, "K": K, "default_us": round(base[n], 1),
                          "default_TF": round(fl / base[n] / 1e6, 1), "tuned_us": round(t, 1),
                          "tuned_TF": round(fl / t / 1e6, 1), "tuning_s": round(tune_s, 1)}), flush=True)
    print(json.dumps({"layer_default_us": round(tot_b, 1), "layer_tuned_us": round(tot_t, 1),
                      "results": tun.get_results() if hasattr(tun, "get_results") else None}, default=str)[:3000], flush=True)


if __name__ == "__main__":
    main()
