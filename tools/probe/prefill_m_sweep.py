#!/usr/bin/env python3
"""prefill_m_sweep.py — how sensitive are the prompt-pass GEMMs (torch.nn.functional.linear -> hipBLASLt) to the token
count M? Llama-3-8B projection shapes, bf16; us per GEMM and the sum per layer for each M."""
import json, sys
import torch
import torch.nn.functional as F

SHAPES = {"qkv": (6144, 4096), "o": (4096, 4096), "up_gate": (28672, 4096), "down": (4096, 14336)}
if len(sys.argv) > 1 and sys.argv[1] == "dense":
    MS = list(range(256, 8193, 256)) + list(range(9216, 16385, 1024)) + [20480, 24576, 32768] + [4097, 5000, 6000, 7000]
else:
    MS = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else
                            "4096,4100,4124,4160,4224,4352,1024,1052,1056,1280,3977,4000").split(",")]
dev = "cuda"
w = {k: torch.randn(n, kk, device=dev, dtype=torch.bfloat16) * 0.02 for k, (n, kk) in SHAPES.items()}
for M in MS:
    row = {"M": M}
    tot = 0.0
    for name, (n, kk) in SHAPES.items():
        x = torch.randn(M, kk, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            F.linear(x, w[name])
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            F.linear(x, w[name])
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 100
        row[name] = round(us, 1)
        tot += us
    row["layer_us"] = round(tot, 1)
    row["tflops"] = round(2 * M * sum(n * kk for n, kk in SHAPES.values()) / tot / 1e6, 1)
    print(json.dumps(row), flush=True)
