"""KV-pool offsets beyond 2^31 elements (a 288 GB pool has > 2^31 elements, every
pool offset must be 64-bit): a pool of 8 256 blocks at Llama-3-8B KV dimensions (32 layers x 8 kv heads x 16 x 128:
2^19 elements = 1 MiB per block and pool, 8.6 GB per pool), the test sequences in its HIGHEST block ids, layer 31 —
element offsets up to 4.3e9. KV store (prefill + decode), paged attention (plain and slab-fed: rotary + store inside
the attention prologue) and swap out -> trample -> swap in are checked against the CPU oracle, which works on a copy of
the top slice of the pool with the block ids rebased. A kernel that formed one of these offsets in 32 bits would read
or write 2^31 elements (4 GiB) lower: the untouched-below check and the oracle comparison both catch it.

The whole-model counterpart (prefill + 128 decode steps with every block id >= 4096) is tests/test_gpu_parity_fulldepth.py.
"""
import types

import pytest
import torch

from oracle import eager_ops as ops

pytestmark = pytest.mark.gpu
NS = types.SimpleNamespace
L, KVH, D, H, BS = 32, 8, 128, 32, 16
NUM_BLOCKS = 8256
TOP = 192                    # blocks in the slice the oracle mirrors: ids NUM_BLOCKS-TOP .. NUM_BLOCKS-1
LAYER = L - 1


def _pool(dtype):
    k = torch.zeros(NUM_BLOCKS, L, KVH, BS, D, dtype=dtype, device="cuda")
    v = torch.zeros(NUM_BLOCKS, L, KVH, BS, D, dtype=dtype, device="cuda")
    assert k.numel() > 2 ** 32 and (NUM_BLOCKS - TOP) * k[0].numel() > 2 ** 31
    return k, v


def _tables(lens, seq_ids, g):
    """Block tables over the top slice, highest ids first, scattered; returns (device table with real ids, CPU table with
    ids rebased to the slice)."""
    base = NUM_BLOCKS - TOP
    perm = (torch.randperm(TOP, generator=g) + base).tolist()
    bt = torch.zeros(max(seq_ids) + 1, 80, dtype=torch.int32)
    for sid, n in zip(seq_ids, lens):
        for j in range(-(-n // BS)):
            bt[sid, j] = perm.pop()
    rebased = torch.where(bt > 0, bt - base, bt)
    return bt, rebased, base


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["float16", "bfloat16"])
def test_store_attention_and_swap_beyond_2_31_elements(dtype):
    from swiftllm_amd.worker import kernels as K
    from swiftllm_amd.worker.kernels.linear import SplitKPartials
    from swiftllm_amd.worker.kernels.paged_attn import paged_attention_from_qkv_splitk
    g = torch.Generator().manual_seed(231)
    kc, vc = _pool(dtype)
    mc = NS(num_layers=L, num_q_heads=H, num_kv_heads=KVH, head_dim=D)
    ec = NS(block_size=BS, max_blocks_per_seq=80)

    # ---- KV store: 3 prefill sequences + 2 decoding ones -------------------------------------------------------
    plens, dlens = [1000, 17, 333], [49, 1024]
    seq_ids = [4, 0, 2, 1, 3]
    bt, bt_small, base = _tables(plens + dlens, seq_ids, g)
    T = sum(plens) + len(dlens)
    k = torch.randn(T, KVH, D, generator=g).to(dtype)
    v = torch.randn(T, KVH, D, generator=g).to(dtype)
    # live content in the slice must survive the store
    kc[base:].copy_(torch.randn(TOP, L, KVH, BS, D, generator=g).to(dtype))
    vc[base:].copy_(torch.randn(TOP, L, KVH, BS, D, generator=g).to(dtype))
    ek, ev = kc[base:].cpu(), vc[base:].cpu()

    def store_state(device):
        pl = torch.tensor(plens, dtype=torch.int32, device=device)
        return NS(seq_ids=torch.tensor(seq_ids, dtype=torch.int32, device=device), num_prefill_seqs=len(plens),
                  num_prefill_tokens=sum(plens), max_prefill_len=max(plens), prefill_seq_lens=pl,
                  prefill_seq_start_locs=torch.cumsum(pl, 0, dtype=torch.int32) - pl, num_decoding_seqs=len(dlens),
                  decoding_seq_lens=torch.tensor(dlens, dtype=torch.int32, device=device))
    ops.store_kvcache(k, v, ek, ev, bt_small, mc, ec, store_state("cpu"), LAYER)
    K.store_kvcache(k.cuda(), v.cuda(), kc, vc, bt.cuda(), mc, ec, store_state("cuda"), LAYER)
    assert torch.equal(kc[base:].cpu(), ek) and torch.equal(vc[base:].cpu(), ev)
    assert not kc[:base].any() and not vc[:base].any()       # nothing landed 2^31 (or any other amount) lower

    # ---- paged attention over the same sequences, now all decoding ------------------------------------------------
    lens = plens + dlens
    q = torch.randn(len(lens), H, D, generator=g).to(dtype)
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    for sbs in (256, 2048):
        nsb = -(-max(lens) // sbs)

        def st(device):
            return NS(num_decoding_seqs=len(lens), num_prefill_seqs=0, seq_block_size=sbs, num_seq_blocks=nsb,
                      softmax_scale=D ** -0.5, decoding_seq_lens=torch.tensor(lens, dtype=torch.int32, device=device),
                      seq_ids=torch.tensor(seq_ids, dtype=torch.int32, device=device))
        want = torch.zeros_like(q)
        ops.paged_attention(q, ek, ev, bt_small, mc, ec, st("cpu"), LAYER, want)
        o = torch.zeros_like(q).cuda()
        K.paged_attention(q.cuda(), kc, vc, bt.cuda(), mc, ec, st("cuda"), LAYER, o)
        err = (o.cpu().float() - want.float()).abs().max().item()
        assert err <= tol, (sbs, err)

    # ---- slab-fed attention: rotary + KV store of the new token inside the attention prologue (64-bit store offsets) --
    ang = torch.rand(2048, D // 2, generator=g) * 6.28
    cos, sin = torch.cos(ang).to(dtype), torch.sin(ang).to(dtype)
    n_qkv = (H + 2 * KVH) * D
    slabs = torch.randn(2, len(lens), n_qkv, generator=g)           # two fp32 "split-K slabs" of a fused qkv projection
    qkv = (slabs[0] + slabs[1]).to(dtype)
    q2 = qkv[:, :H * D].reshape(len(lens), H, D).clone()
    k2 = qkv[:, H * D:(H + KVH) * D].reshape(len(lens), KVH, D).clone()
    v2 = qkv[:, (H + KVH) * D:].reshape(len(lens), KVH, D).clone()
    pos = torch.tensor([n - 1 for n in lens], dtype=torch.int32)
    ops.rotary_embedding_inplace(q2, k2, NS(position_cos=cos[pos.long()], position_sin=sin[pos.long()]))
    dec_state = NS(seq_ids=torch.tensor(seq_ids, dtype=torch.int32), num_prefill_seqs=0, num_prefill_tokens=0,
                   max_prefill_len=0, prefill_seq_lens=torch.empty(0, dtype=torch.int32),
                   prefill_seq_start_locs=torch.empty(0, dtype=torch.int32), num_decoding_seqs=len(lens),
                   decoding_seq_lens=torch.tensor(lens, dtype=torch.int32))
    ops.store_kvcache(k2, v2, ek, ev, bt_small, mc, ec, dec_state, LAYER)
    sbs, nsb = 2048, 1
    stc = NS(num_decoding_seqs=len(lens), num_prefill_seqs=0, seq_block_size=sbs, num_seq_blocks=nsb,
             softmax_scale=D ** -0.5, decoding_seq_lens=torch.tensor(lens, dtype=torch.int32),
             seq_ids=torch.tensor(seq_ids, dtype=torch.int32))
    want = torch.zeros(len(lens), H, D, dtype=dtype)
    ops.paged_attention(q2, ek, ev, bt_small, mc, ec, stc, LAYER, want)
    std = NS(num_decoding_seqs=len(lens), num_prefill_seqs=0, seq_block_size=sbs, num_seq_blocks=nsb,
             softmax_scale=D ** -0.5, decoding_seq_lens=torch.tensor(lens, dtype=torch.int32, device="cuda"),
             seq_ids=torch.tensor(seq_ids, dtype=torch.int32, device="cuda"), position_cos=cos.cuda(),
             position_sin=sin.cuda(), position_indices=pos.cuda(), paged_attn_scratch=None)
    part = SplitKPartials(slabs.cuda().contiguous().view(-1), 2, len(lens), n_qkv, dtype)
    o = torch.zeros(len(lens), H * D, dtype=dtype, device="cuda")
    paged_attention_from_qkv_splitk(part, kc, vc, bt.cuda(), mc, ec, std, LAYER, o)
    assert torch.equal(kc[base:].cpu(), ek) and torch.equal(vc[base:].cpu(), ev)     # the new tokens' K/V, bit for bit
    assert not kc[:base].any() and not vc[:base].any()
    err = (o.cpu().view(len(lens), H, D).float() - want.float()).abs().max().item()
    assert err <= tol, err

    # ---- swap: top blocks out to the host pool, trample, back in ---------------------------------------------------
    from swiftllm_amd.worker.kernels.block_swapping import swap_blocks
    n_swap = 24
    ks = torch.zeros(32, L, KVH, BS, D, dtype=dtype, pin_memory=True)
    vs = torch.zeros(32, L, KVH, BS, D, dtype=dtype, pin_memory=True)
    src = [NUM_BLOCKS - 1 - 2 * i for i in range(n_swap)]           # every other block from the very top
    dst = [(5 * i + 3) % 32 for i in range(n_swap)]
    assert len(set(dst)) == n_swap
    swap_blocks(src, dst, False, kc, vc, ks, vs)
    torch.cuda.synchronize()
    for s_, d_ in zip(src, dst):
        assert torch.equal(ks[d_], ek[s_ - base]) and torch.equal(vs[d_], ev[s_ - base])
    kc[base:].zero_()
    vc[base:].zero_()
    swap_blocks(dst, src, True, kc, vc, ks, vs)
    torch.cuda.synchronize()
    back_k, back_v = kc[base:].cpu(), vc[base:].cpu()
    for s_ in src:
        assert torch.equal(back_k[s_ - base], ek[s_ - base]) and torch.equal(back_v[s_ - base], ev[s_ - base])
    assert not kc[:base].any() and not vc[:base].any()
