#!/usr/bin/env python3
"""prefill_m_sweep.py — how sensitive are the projections (torch.nn.functional.linear -> hipBLASLt) to the token count
M? Llama-3-8B projection shapes, bf16; us per GEMM and the sum per layer for each M.

    python tools/probe/prefill_m_sweep.py [dense | M1,M2,...] [--cold]

--cold: every launch takes the next of enough weight copies to exceed the 256 MiB Infinity Cache several times over
(what a decode step sees: each layer's weights come from HBM) — for decode-sized M, where the product is memory-bound."""
import json, sys
import torch
import torch.nn.functional as F

SHAPES = {"qkv": (6144, 4096), "o": (4096, 4096), "up_gate": (28672, 4096), "down": (4096, 14336)}
argv = [x for x in sys.argv[1:] if x != "--cold"]
cold = "--cold" in sys.argv[1:]
if argv and argv[0] == "dense":
    MS = list(range(256, 8193, 256)) + list(range(9216, 16385, 1024)) + [20480, 24576, 32768] + [4097, 5000, 6000, 7000]
else:
    MS = [int(x) for x in (argv[0] if argv else "4096,4100,4124,4160,4224,4352,1024,1052,1056,1280,3977,4000").split(",")]
dev = "cuda"
ITERS = 24 if cold else 10
w = {}
for k, (n, kk) in SHAPES.items():
    copies = max(1, -(-(3 << 29) // (n * kk * 2))) if cold else 1       # >= 1.5 GB of weights per projection when cold
    w[k] = [torch.randn(n, kk, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(copies)]
for M in MS:
    row = {"M": M, "cold": cold}
    tot = 0.0
    for name, (n, kk) in SHAPES.items():
        x = torch.randn(M, kk, device=dev, dtype=torch.bfloat16)
        ws = w[name]
        for i in range(3):
            F.linear(x, ws[i % len(ws)])
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(ITERS):
            F.linear(x, ws[(i + 3) % len(ws)])
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / ITERS
        row[name] = round(us, 1)
        tot += us
    row["layer_us"] = round(tot, 1)
    row["tflops"] = round(2 * M * sum(n * kk for n, kk in SHAPES.values()) / tot / 1e6, 1)
    print(json.dumps(row), flush=True)
