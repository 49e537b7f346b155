#!/usr/bin/env python3
"""Times the ablation variants of tools/probe/gemm_probe.hip (see its header). GPU only; not product code."""
import ctypes, json, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libgemm_probe.so"))
lib.probe_gemm.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]

def bench(fn, iters=100):
    for i in range(5): fn(i)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for i in range(iters): fn(i)
    e.record(); e.synchronize()
    return s.elapsed_time(e) * 1e3 / iters

VARIANTS = [tuple(map(int, v.split(":"))) for v in os.environ.get(
    "PROBE_VARIANTS", "0:2,1:2,10:2,10:3,11:2,11:4,12:4,12:6").split(",")]
for name, (N, K) in {"up_gate": (28672, 4096), "lm_head": (128256, 4096)}.items():
    copies = max(2, min(8, int(2e9 // (N * K * 2))))
    ws = [torch.empty(N, K, dtype=torch.bfloat16, device="cuda").normal_(0, 0.02) for _ in range(copies)]
    x = torch.randn(32, K, device="cuda").bfloat16()
    out = torch.empty(32, N, dtype=torch.bfloat16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for v, occ in VARIANTS:
        def run(i):
            rc = lib.probe_gemm(v, occ, out.data_ptr(), x.data_ptr(), ws[i % copies].data_ptr(), 32, N, K, st)
            assert rc == 0, rc
        t = bench(run)
        err = None
        if v in (0, 10, 11, 12):
            ref = (x.float() @ ws[(5 + 99) % copies].float().t())
            err = round((out.float() - ref).abs().max().item(), 4)
        print(json.dumps({"shape": name, "variant": v, "occ": occ, "us": round(t, 2), "TBps": round(N * K * 2 / t / 1e6, 2), "max_err": err}), flush=True)
    del ws
