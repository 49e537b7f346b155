#!/usr/bin/env python3
"""gemm_silu_micro.py — the up/gate projection + SiLU-gate of one Llama-3-8B decode layer (M tokens) launched
`iters` times over `copies` distinct weight matrices (so every launch streams from HBM). Used under
rocprofv3 --pmc (tools/gpu_pmc_gemm.sh) to compare HBM traffic with the algorithmic bytes."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swiftllm_amd import _hip

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=32)
ap.add_argument("--iters", type=int, default=64)
ap.add_argument("--copies", type=int, default=16)
ap.add_argument("--row-major", action="store_true", help="stream the row-major weights (default: the pre-packed copy)")
ap.add_argument("--nf", action="store_true", help="the norm-on-the-fly entry the r05 decode step runs (swl_gemm_skinny_packed_silu_gate_nf)")
a = ap.parse_args()
I, K = 14336, 4096
ws = [torch.empty(2 * I, K, dtype=torch.bfloat16, device="cuda").normal_(0, 0.02) for _ in range(a.copies)]
x = torch.randn(a.m, K, device="cuda").bfloat16()
out = torch.empty(a.m, I, dtype=torch.bfloat16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
if not a.row_major:
    packed = []
    for w in ws:
        wp = torch.empty_like(w)
        _hip.call("swl_gemm_pack_weight", wp.data_ptr(), w.data_ptr(), 2 * I, K, 1, st)
        packed.append(wp)
    torch.cuda.synchronize()
    ws = packed
fn = "swl_gemm_skinny_silu_gate" if a.row_major else ("swl_gemm_skinny_packed_silu_gate_nf" if a.nf else "swl_gemm_skinny_packed_silu_gate")
nw = (1 + 0.1 * torch.randn(K, device="cuda")).bfloat16()
def launch(i):
    if a.nf:
        _hip.call(fn, out.data_ptr(), x.data_ptr(), nw.data_ptr(), 1e-5, ws[i % a.copies].data_ptr(), a.m, I, K, K, I, 1, st)
    else:
        _hip.call(fn, out.data_ptr(), x.data_ptr(), ws[i % a.copies].data_ptr(), a.m, I, K, K, I, 1, st)
for i in range(4):
    launch(i)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); s.record()
for i in range(a.iters):
    launch(i)
e.record(); e.synchronize()
us = s.elapsed_time(e) * 1e3 / a.iters
alg = 2 * I * K * 2 + a.m * K * 2 + a.m * I * 2
print(json.dumps({"kernel": "gemm_skinny_ring_kernel<bf16, SiluGate%s%s>" % ("" if a.row_major else ", packed W", ", norm on the fly" if a.nf else ""), "M": a.m, "I": I, "K": K, "us": round(us, 2),
                  "algorithmic_bytes": alg, "TBps": round(alg / us / 1e6, 3)}))
