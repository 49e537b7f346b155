#!/usr/bin/env python3
"""kv_layout_probe.py — what does the reference's pool layout [block][layer][kv-head][16][D] cost the decode attention kernel
against a layer-major one [layer][block][kv-head][16][D] (GPU)? One launch reads ONE layer: 4 KiB out of every MiB of the
block-major pool (its tiles are spread over the whole pool: one page per tile), 32 KiB runs of a layer-major one. Emulated
without touching the kernel: a pool with `num_layers = 1` IS a layer's slice of a layer-major pool — 32 such pools against one
32-layer pool, same block table (scattered / contiguous / interleaved block ids), Llama-3-8B geometry, batch 32 x 1088 tokens."""
import argparse, json, os, sys, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swiftllm_amd.worker import kernels as K                       # noqa: E402
from swiftllm_amd.worker.batch_plan import select_seq_block_size    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", default="4096,49152")
    ap.add_argument("--iters", type=int, default=256)
    a = ap.parse_args()
    H, KVH, D, B, n, L = 32, 8, 128, 32, 1088, 32
    dt, dev = torch.bfloat16, "cuda"
    nblk_seq = -(-n // 16)
    lens = [n] * B
    sbs = select_seq_block_size(lens, KVH, torch.cuda.get_device_properties(0).multi_processor_count)
    ec = types.SimpleNamespace(block_size=16)
    q = torch.randn(B, H, D, device=dev).to(dt)
    o = torch.empty_like(q)
    st = types.SimpleNamespace(num_decoding_seqs=B, num_prefill_seqs=0, seq_block_size=sbs, num_seq_blocks=-(-n // sbs),
                               softmax_scale=D ** -0.5, decoding_seq_lens=torch.tensor(lens, dtype=torch.int32, device=dev),
                               seq_ids=torch.arange(B, dtype=torch.int32, device=dev),
                               paged_attn_scratch=torch.empty(16, dtype=torch.float32, device=dev))

    def fill(t):
        flat = t.view(-1)
        for s0 in range(0, flat.numel(), 1 << 28):
            flat[s0:s0 + (1 << 28)].normal_()
        return t

    def time_us(run):
        for i in range(2 * L):
            run(i % L)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for i in range(a.iters):
            run(i % L)
        e.record(); e.synchronize()
        return s.elapsed_time(e) * 1e3 / a.iters

    for nb in [int(x) for x in a.blocks.split(",")]:
        g = torch.Generator(device="cpu").manual_seed(1)
        tables = {
            "contiguous": torch.tensor([[i * (nb // B) + j for j in range(nblk_seq)] for i in range(B)], dtype=torch.int32),
            "interleaved": torch.tensor([[j * B + i for j in range(nblk_seq)] for i in range(B)], dtype=torch.int32),
            "scattered": torch.randperm(nb, generator=g)[:B * nblk_seq].to(torch.int32).view(B, nblk_seq),
        }
        kc = fill(torch.empty(nb, L, KVH, 16, D, device=dev, dtype=dt))
        vc = fill(torch.empty(nb, L, KVH, 16, D, device=dev, dtype=dt))
        mcL = types.SimpleNamespace(num_q_heads=H, num_kv_heads=KVH, head_dim=D, num_layers=L)
        res = {}
        for name, bt in tables.items():
            btd = bt.to(dev).contiguous()
            res[("block_major", name)] = time_us(lambda layer: K.paged_attention(q, kc, vc, btd, mcL, ec, st, layer, o))
        # layer-major: the same bytes as 32 single-layer pools (views of the same allocation: [L][nb][KVH][16][D])
        kl = kc.view(L, nb, 1, KVH, 16, D)
        vl = vc.view(L, nb, 1, KVH, 16, D)
        mc1 = types.SimpleNamespace(num_q_heads=H, num_kv_heads=KVH, head_dim=D, num_layers=1)
        for name, bt in tables.items():
            btd = bt.to(dev).contiguous()
            res[("layer_major", name)] = time_us(lambda layer: K.paged_attention(q, kl[layer], vl[layer], btd, mc1, ec, st, 0, o))
        for (layout, name), us in res.items():
            print(json.dumps({"pool_blocks": nb, "pool_gb_each": round(nb * L * KVH * 16 * D * 2 / 1e9, 1), "layout": layout,
                              "block_ids": name, "us": round(us, 2), "TBps": round(sum(lens) * 2 * KVH * D * 2 / us / 1e6, 3)}), flush=True)
        del kc, vc, kl, vl
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
