#!/usr/bin/env python3
"""kv_pool_gap_probe.py — does the distance between the K pool and the V pool matter to the decode paged-attention kernel (GPU)?
A wave requests the K tile and the V tile of the same (block, layer, kv-head) together: at the same offset in two pools whose
bases are a large power of two apart, both requests meet the same HBM channel / bank. One allocation, K pool at its start, V pool
`blocks * block_bytes + gap` behind it; Llama-3-8B geometry, batch 32, context 1088, blocks of a sequence contiguous, launches
cycle through the layers. One JSON line per (pool blocks, gap)."""
import argparse, json, os, sys, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swiftllm_amd.worker import kernels as K                       # noqa: E402
from swiftllm_amd.worker.batch_plan import select_seq_block_size    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", default="4096,4160,8192")
    ap.add_argument("--gaps", default="0,4096,65536,1052672,2101248")
    ap.add_argument("--iters", type=int, default=256)
    a = ap.parse_args()
    H, KVH, D, B, n, L = 32, 8, 128, 32, 1088, 32
    dt = torch.bfloat16
    dev = "cuda"
    nblk_seq = -(-n // 16)
    block_elems = L * KVH * 16 * D
    lens = [n] * B
    sbs = select_seq_block_size(lens, KVH, torch.cuda.get_device_properties(0).multi_processor_count)
    nsb = -(-n // sbs)
    mc = types.SimpleNamespace(num_q_heads=H, num_kv_heads=KVH, head_dim=D, num_layers=L)
    ec = types.SimpleNamespace(block_size=16)
    q = torch.randn(B, H, D, device=dev).to(dt)
    o = torch.empty_like(q)
    for nb in [int(x) for x in a.blocks.split(",")]:
        pool_elems = nb * block_elems
        max_gap = max(int(g) for g in a.gaps.split(","))
        buf = torch.empty(2 * pool_elems + max_gap // 2 + 64, device=dev, dtype=dt)
        for s0 in range(0, buf.numel(), 1 << 28):
            buf[s0:s0 + (1 << 28)].normal_()
        # a sequence's blocks contiguous, sequences spread over the pool
        starts = [(i * (nb // B)) for i in range(B)]
        bt = torch.tensor([[s + j for j in range(nblk_seq)] for s in starts], dtype=torch.int32, device=dev)
        st = types.SimpleNamespace(num_decoding_seqs=B, num_prefill_seqs=0, seq_block_size=sbs, num_seq_blocks=nsb,
                                   softmax_scale=D ** -0.5, decoding_seq_lens=torch.tensor(lens, dtype=torch.int32, device=dev),
                                   seq_ids=torch.arange(B, dtype=torch.int32, device=dev),
                                   paged_attn_scratch=torch.empty(16, dtype=torch.float32, device=dev))
        for gap in [int(g) for g in a.gaps.split(",")]:
            kc = buf[:pool_elems].view(nb, L, KVH, 16, D)
            v0 = pool_elems + gap // 2
            vc = buf[v0:v0 + pool_elems].view(nb, L, KVH, 16, D)
            for i in range(8):
                K.paged_attention(q, kc, vc, bt, mc, ec, st, i % L, o)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for i in range(a.iters):
                K.paged_attention(q, kc, vc, bt, mc, ec, st, i % L, o)
            e.record(); e.synchronize()
            us = s.elapsed_time(e) * 1e3 / a.iters
            print(json.dumps({"pool_blocks": nb, "pool_bytes": pool_elems * 2, "v_minus_k_bytes": (v0) * 2, "gap_bytes": gap,
                              "us": round(us, 2), "TBps": round(sum(lens) * 2 * KVH * D * 2 / us / 1e6, 3)}), flush=True)
        del buf, kc, vc
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
