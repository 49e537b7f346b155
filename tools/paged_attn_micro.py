#!/usr/bin/env python3
"""paged_attn_micro.py — the decode paged-attention op alone, at a BASELINE.json shape (GPU).

    python tools/paged_attn_micro.py --shape c3            # Llama-3-8B, batch 32, context ~1088
    python tools/paged_attn_micro.py --shape c2            # Llama-3-8B, batch 1, context 1024
    python tools/paged_attn_micro.py --shape c4            # Llama-2-7B dims, batch 4, context 16384
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- python tools/paged_attn_micro.py --shape c3 --iters 64

KV pools hold N(0,1) data for `--layers` layers; launches cycle through the layers so the footprint
(>= 1 GB) is far beyond the 256 MiB Infinity Cache. Timing: HIP events on the launch stream.
Prints one JSON line: algorithmic bytes, us per op (phase 1 + phase 2 when there is one), GB/s.
"""
import argparse
import json
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swiftllm_amd.worker import kernels as K                       # noqa: E402
from swiftllm_amd.worker.batch_plan import select_seq_block_size    # noqa: E402

SHAPES = {  # H, KVH, D, batch, len, layers kept resident
    "c2": (32, 8, 128, 1, 1024, 32),
    "c3": (32, 8, 128, 32, 1088, 32),
    "c4": (32, 32, 128, 4, 16384, 4),
    "c3_b128": (32, 8, 128, 128, 1088, 8),
    "c3_b64": (32, 8, 128, 64, 1088, 16),
    "c3_b256": (32, 8, 128, 256, 1088, 4),
    "long": (32, 8, 128, 1, 131072, 8),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="c3", choices=sorted(SHAPES))
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--iters", type=int, default=256)
    ap.add_argument("--sbs", type=int, default=0, help="override the split-K width")
    ap.add_argument("--slab-copies", type=int, default=1, help="with --qkv: distinct slab tensors cycled through (cold slabs)")
    ap.add_argument("--qkv", type=int, default=0, help="k-splits of fused-qkv slabs: time swl_paged_attn_decode_qkv "
                                                       "(rotary + KV store in the prologue), the variant a decode step runs")
    a = ap.parse_args()
    H, KVH, D, B, n, L = SHAPES[a.shape]
    dtype = getattr(torch, a.dtype)
    dev = "cuda"
    nblk_seq = -(-n // 16)
    nblk = B * nblk_seq
    kc = torch.randn(nblk, L, KVH, 16, D, device=dev, dtype=torch.float32).to(dtype) if nblk * L * KVH * 16 * D < 2**31 \
        else torch.empty(nblk, L, KVH, 16, D, device=dev, dtype=dtype).normal_()
    vc = torch.empty_like(kc).normal_()
    perm = torch.randperm(nblk, device=dev).to(torch.int32)         # scattered physical blocks
    bt = perm.view(B, nblk_seq).contiguous()
    lens = [n] * B
    sbs = a.sbs or select_seq_block_size(lens, KVH, torch.cuda.get_device_properties(0).multi_processor_count)
    nsb = -(-n // sbs)
    st = types.SimpleNamespace(num_decoding_seqs=B, num_prefill_seqs=0, seq_block_size=sbs, num_seq_blocks=nsb,
                               softmax_scale=D ** -0.5,
                               decoding_seq_lens=torch.tensor(lens, dtype=torch.int32, device=dev),
                               seq_ids=torch.arange(B, dtype=torch.int32, device=dev), paged_attn_scratch=None)
    if nsb > 1:
        st.paged_attn_scratch = torch.empty(B * H * nsb * (D + 1), dtype=torch.float32, device=dev)
    mc = types.SimpleNamespace(num_q_heads=H, num_kv_heads=KVH, head_dim=D, num_layers=L)
    ec = types.SimpleNamespace(block_size=16)
    q = torch.randn(B, H, D, device=dev).to(dtype)
    o = torch.empty_like(q)
    if a.qkv:
        from swiftllm_amd.worker.kernels.linear import SplitKPartials
        from swiftllm_amd.worker.kernels.paged_attn import paged_attention_from_qkv_splitk
        st.position_indices = (st.decoding_seq_lens - 1).contiguous()
        ang = torch.rand(n + 8, D // 2, device=dev) * 6.28
        st.position_cos, st.position_sin = torch.cos(ang).to(dtype), torch.sin(ang).to(dtype)
        # --slab-copies N: cycle through N distinct slab tensors (in the decode step the slabs were written by the previous
        # kernel on other XCDs and are read cold; one tensor re-read every launch sits in the reader's L2)
        slab_parts = [SplitKPartials(torch.randn(a.qkv, B, (H + 2 * KVH) * D, device=dev, dtype=torch.float32) * 0.5,
                                     a.qkv, B, (H + 2 * KVH) * D, dtype) for _ in range(max(1, a.slab_copies))]
        run = lambda layer: paged_attention_from_qkv_splitk(slab_parts[layer % len(slab_parts)], kc, vc, bt, mc, ec, st,  # noqa: E731
                                                            layer % L, o)
    else:
        run = lambda layer: K.paged_attention(q, kc, vc, bt, mc, ec, st, layer, o)   # noqa: E731
    for i in range(min(a.iters, 2 * L)):
        run(i if a.qkv else i % L)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for i in range(a.iters):
        run(i if a.qkv else i % L)
    stop.record()
    stop.synchronize()
    us = start.elapsed_time(stop) * 1e3 / a.iters
    e = dtype.itemsize
    kv = sum(lens) * 2 * KVH * D * e
    part = B * H * nsb * (D + 1) * 4 * 2 if nsb > 1 else 0
    # as bench.py prices it: KV read once per kv head + q (or the fp32 qkv slabs the prologue sums) + o + partials
    q_bytes = a.qkv * B * (H + 2 * KVH) * D * 4 if a.qkv else B * H * D * e
    alg = kv + q_bytes + B * H * D * e + part
    kernel = ("paged_attn_phase1_kernel<%s, D=%d, G=%d%s>" % (a.dtype, D, H // KVH, ", QKV: rotary + KV store in the prologue" if a.qkv else ""))
    print(json.dumps(dict(kernel=kernel, algorithmic_bytes=alg, shape=a.shape, qkv_slabs=a.qkv, slab_copies=a.slab_copies, lib=os.environ.get("SWIFTLLM_HIP_LIB", "default"), dtype=a.dtype, H=H, KVH=KVH, D=D, batch=B, len=n, seq_block_size=sbs,
                          num_seq_blocks=nsb, workgroups=B * KVH * nsb, us_per_op=round(us, 2),
                          alg_bytes=alg, kv_bytes=kv, GBps=round(alg / us / 1e3, 1),
                          frac_of_8TBps=round(alg / us / 1e3 / 8000, 4), iters=a.iters)))


if __name__ == "__main__":
    main()
