#!/usr/bin/env python3
"""prefill_attn_micro.py — the varlen causal prefill attention kernel alone (GPU): TFLOP/s at BASELINE shapes.
flop = 4 * sum(len^2) * D * H / 2 (causal)."""
import argparse, json, os, sys, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swiftllm_amd.worker import kernels as K

SHAPES = {"c3": (32, 8, 128, [1024] * 32), "c4": (32, 32, 128, [16384] * 4), "mid": (32, 8, 128, [4096] * 8),
          "ragged": (32, 8, 128, [100, 700, 1024, 3000, 57, 2048, 1, 513] * 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="c3")
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    H, KVH, D, lens = SHAPES[a.shape]
    dtype = getattr(torch, a.dtype)
    P = sum(lens)
    q = torch.randn(P, H, D, device="cuda").to(dtype)
    k = torch.randn(P, KVH, D, device="cuda").to(dtype)
    v = torch.randn(P, KVH, D, device="cuda").to(dtype)
    o = torch.empty_like(q)
    cu = torch.zeros(len(lens) + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(torch.tensor(lens, dtype=torch.int32), 0)
    st = types.SimpleNamespace(num_prefill_seqs=len(lens), max_prefill_len=max(lens), softmax_scale=D ** -0.5,
                               prefill_seq_start_locs_with_end=cu.cuda())
    mc = types.SimpleNamespace(num_q_heads=H, num_kv_heads=KVH, head_dim=D)
    for _ in range(3):
        K.prefill_attention(q, k, v, o, mc, None, st)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(a.iters):
        K.prefill_attention(q, k, v, o, mc, None, st)
    e.record(); e.synchronize()
    ms = s.elapsed_time(e) / a.iters
    flop = 4 * sum(n * n for n in lens) * D * H / 2
    print(json.dumps(dict(shape=a.shape, dtype=a.dtype, tokens=P, ms=round(ms, 3), TFLOPs=round(flop / ms / 1e9, 1),
                          frac_of_2500=round(flop / ms / 1e9 / 2500, 4))))


if __name__ == "__main__":
    main()
