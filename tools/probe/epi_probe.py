#!/usr/bin/env python3
"""Cost breakdown of the split-K completion epilogue (SWL_EPI_DEBUG stages) on o_proj / down_proj / qkv shapes."""
import json, os, sys, subprocess
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from swiftllm_amd import _hip
import importlib
L = importlib.import_module("swiftllm_amd.worker.kernels.linear")

def bench(fn, iters=100):
    for i in range(5): fn(i)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for i in range(iters): fn(i)
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1e3 / iters

M = 32
for name, (N, K) in {"o": (4096, 4096), "down": (4096, 14336), "qkv_as_resid": (6144, 4096)}.items():
    copies = 8
    ws = [torch.empty(N, K, dtype=torch.bfloat16, device="cuda").normal_(0, 0.02) for _ in range(copies)]
    x = torch.randn(M, K, device="cuda").bfloat16()
    res = torch.zeros(M, N, device="cuda").bfloat16()
    t_part = bench(lambda i: L.linear_splitk(x, ws[i % copies]))
    t_epi = bench(lambda i: L.linear_add_residual(x, ws[i % copies], res))
    print(json.dumps({"shape": name, "debug": os.environ.get("SWL_EPI_DEBUG", "0"), "partial_us": round(t_part, 2),
                      "add_residual_us": round(t_epi, 2)}), flush=True)
