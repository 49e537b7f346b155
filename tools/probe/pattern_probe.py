#!/usr/bin/env python3
"""HBM stream rate of the weight matrix vs contiguous bytes per row per load instruction (probe, GPU only)."""
import ctypes, json, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libgemm_probe.so"))
lib.probe_stream_pattern.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
def bench(fn, iters=50):
    for i in range(5): fn(i)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for i in range(iters): fn(i)
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1e3 / iters
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for name, (N, K) in {"up_gate": (28672, 4096), "lm_head": (128256, 4096), "down": (4096, 14336)}.items():
    copies = max(2, min(8, int(2e9 // (N * K * 2))))
    ws = [torch.empty(N, K, dtype=torch.bfloat16, device="cuda").normal_(0, 0.02) for _ in range(copies)]
    for run in (256, 512, 1024):
        def f(i):
            assert lib.probe_stream_pattern(run, ws[i % copies].data_ptr(), N, K, sink.data_ptr(), st) == 0
        t = bench(f)
        print(json.dumps({"shape": name, "run_bytes": run, "us": round(t, 2), "TBps": round(N * K * 2 / t / 1e6, 2)}), flush=True)
    del ws
