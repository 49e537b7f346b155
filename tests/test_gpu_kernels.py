"""GPU parity tests: every gfx950 kernel, called through the C ABI (swiftllm_amd.worker.kernels ->
libswiftllm_hip.so), against the oracle on the same seeded inputs, against the golden vectors frozen
from the reference's Triton kernels, and through size-independent properties at full sizes.

Tolerances (written here, per ③ of the task): copies / integer work bit-exact; rotary bit-exact
(same rounding points, exact fp32 intermediates); rmsnorm / silu <= 1 ulp of the storage dtype
(fp32 reduction order; silu <= 2 ulp: exp implementation); attention <= 4e-3 abs vs the reference's fp16-score path
and <= 2e-3 vs the exact-score oracle on N(0,1) data.
"""
import types

import pytest
import torch

from oracle import eager_ops as ops
from conftest import ulp_diff_fp16

pytestmark = pytest.mark.gpu
NS = types.SimpleNamespace
DTYPES = [torch.float16, torch.bfloat16]


def K():
    from swiftllm_amd.worker import kernels
    return kernels


def gen(seed):
    return torch.Generator().manual_seed(seed)


def dev(t):
    return t.cuda() if isinstance(t, torch.Tensor) else t


# ---- rmsnorm ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tokens,hidden", [(1, 4096), (33, 4096), (7, 128), (5, 256), (3, 8192), (2, 5120), (4, 16384)])
def test_rmsnorm(dtype, tokens, hidden):
    g = gen(tokens * 7 + hidden)
    x = torch.randn(tokens, hidden, generator=g).to(dtype)
    r = torch.randn(tokens, hidden, generator=g).to(dtype)
    w = (1 + 0.1 * torch.randn(hidden, generator=g)).to(dtype)
    ex = x.clone()
    ops.rmsnorm_inplace(ex, w, 1e-5)
    gx = x.cuda()
    K().rmsnorm_inplace(gx, w.cuda(), 1e-5)
    assert ulp_diff_fp16(gx.cpu(), ex) <= 1
    ex, er = x.clone(), r.clone()
    ops.fused_add_rmsnorm_inplace(ex, er, w, 1e-5)
    gx, gr = x.cuda(), r.cuda()
    K().fused_add_rmsnorm_inplace(gx, gr, w.cuda(), 1e-5)
    assert torch.equal(gr.cpu(), er)                # the rounded residual sum is exact
    assert ulp_diff_fp16(gx.cpu(), ex) <= 1


def test_rmsnorm_golden(golden):
    g = golden("elementwise.pt")
    a = g["rmsnorm"]
    x = a["x"].cuda()
    K().rmsnorm_inplace(x, a["w"].cuda(), a["eps"])
    assert ulp_diff_fp16(x.cpu(), a["out"]) <= 1
    b = g["fused_add_rmsnorm"]
    x, r = b["x"].cuda(), b["r"].cuda()
    K().fused_add_rmsnorm_inplace(x, r, b["w"].cuda(), b["eps"])
    assert torch.equal(r.cpu(), b["out_r"]) and ulp_diff_fp16(x.cpu(), b["out_x"]) <= 1


def test_empty_batches_are_noops():
    k = K()
    for dtype in DTYPES:
        x = torch.empty(0, 4096, dtype=dtype, device="cuda")
        w = torch.ones(4096, dtype=dtype, device="cuda")
        k.rmsnorm_inplace(x, w, 1e-5)
        k.fused_add_rmsnorm_inplace(x, x.clone(), w, 1e-5)
        k.silu_and_mul_inplace(torch.empty(0, 512, dtype=dtype, device="cuda"))
    torch.cuda.synchronize()


# ---- silu ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tokens,inter", [(1, 14336), (19, 14336), (3, 256), (5, 11008)])
def test_silu_and_mul(dtype, tokens, inter):
    x = (2 * torch.randn(tokens, 2 * inter, generator=gen(inter + tokens))).to(dtype)
    ex = x.clone()
    ops.silu_and_mul_inplace(ex)
    gx = x.cuda()
    K().silu_and_mul_inplace(gx)
    gx = gx.cpu()
    assert torch.equal(gx[:, inter:], x[:, inter:])
    # silu rounded to the storage dtype may differ by 1 ulp (fp32 exp implementations differ in the
    # last bit); the storage-dtype product can turn that into 2 ulp
    assert ulp_diff_fp16(gx[:, :inter], ex[:, :inter]) <= 2


def test_silu_golden(golden):
    a = golden("elementwise.pt")["silu_and_mul"]
    x = a["x"].cuda()
    K().silu_and_mul_inplace(x)
    inter = x.shape[1] // 2
    assert ulp_diff_fp16(x.cpu()[:, :inter], a["out"][:, :inter]) <= 2


# ---- rotary -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,H,KVH,D", [(1, 32, 8, 128), (37, 32, 8, 128), (9, 4, 2, 32), (5, 8, 8, 64)])
@pytest.mark.parametrize("indexed", [False, True])
def test_rotary(dtype, T, H, KVH, D, indexed):
    g = gen(T + H + D)
    q = torch.randn(T, H, D, generator=g).to(dtype)
    k = torch.randn(T, KVH, D, generator=g).to(dtype)
    ang = torch.rand(64, D // 2, generator=g) * 6.28
    cos_tab, sin_tab = torch.cos(ang).to(dtype), torch.sin(ang).to(dtype)
    pos = torch.randint(0, 64, (T,), generator=g, dtype=torch.int32)
    eq, ek = q.clone(), k.clone()
    ops.rotary_embedding_inplace(eq, ek, NS(position_cos=cos_tab, position_sin=sin_tab, position_indices=pos))
    gq, gk = q.cuda(), k.cuda()
    if indexed:
        st = NS(position_cos=cos_tab.cuda(), position_sin=sin_tab.cuda(), position_indices=pos.cuda())
    else:
        st = NS(position_cos=cos_tab[pos.long()].cuda(), position_sin=sin_tab[pos.long()].cuda(),
                position_indices=None)
    K().rotary_embedding_inplace(gq, gk, st)
    assert torch.equal(gq.cpu(), eq) and torch.equal(gk.cpu(), ek)


def test_rotary_golden_and_strided_qkv(golden):
    a = golden("elementwise.pt")["rotary"]
    q, k = a["q"].cuda(), a["k"].cuda()
    st = NS(position_cos=a["cos"].cuda(), position_sin=a["sin"].cuda(), position_indices=None)
    K().rotary_embedding_inplace(q, k, st)
    assert torch.equal(q.cpu(), a["out_q"]) and torch.equal(k.cpu(), a["out_k"])
    # q and k as column slices of one fused qkv buffer (token pitch > heads*dim)
    T, H, KVH, D = a["q"].shape[0], a["q"].shape[1], a["k"].shape[1], a["q"].shape[2]
    qkv = torch.zeros(T, (H + 2 * KVH) * D, dtype=torch.float16, device="cuda")
    qkv[:, :H * D] = a["q"].cuda().view(T, -1)
    qkv[:, H * D:(H + KVH) * D] = a["k"].cuda().view(T, -1)
    qv = qkv[:, :H * D].view(T, H, D)
    kv = qkv[:, H * D:(H + KVH) * D].view(T, KVH, D)
    K().rotary_embedding_inplace(qv, kv, st)
    assert torch.equal(qv.cpu(), a["out_q"]) and torch.equal(kv.cpu(), a["out_k"])
    assert float(qkv[:, (H + KVH) * D:].abs().max()) == 0.0     # v columns untouched


# ---- KV store -----------------------------------------------------------------------------------------
def _store_state(seq_ids, plens, dlens, device):
    pl = torch.tensor(plens, dtype=torch.int32)
    cu = torch.zeros(len(plens) + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(pl, 0)
    t = lambda x: x.to(device)  # noqa: E731
    return NS(seq_ids=t(torch.tensor(seq_ids, dtype=torch.int32)), num_prefill_seqs=len(plens),
              num_prefill_tokens=sum(plens), max_prefill_len=max(plens) if plens else 0,
              prefill_seq_lens=t(pl), prefill_seq_start_locs=t(cu[:-1].contiguous()),
              prefill_seq_start_locs_with_end=t(cu), num_decoding_seqs=len(dlens),
              decoding_seq_lens=t(torch.tensor(dlens, dtype=torch.int32)))


def test_store_kvcache_golden(golden):
    g = golden("kvcache_blocks.pt")["store_kvcache"]
    st = _store_state(g["seq_ids"].tolist(), g["plens"], g["dlens"], "cuda")
    kc, vc = torch.zeros_like(g["k_cache"]).cuda(), torch.zeros_like(g["v_cache"]).cuda()
    K().store_kvcache(g["k"].cuda(), g["v"].cuda(), kc, vc, g["block_table"].cuda(),
                      NS(num_layers=g["L"], num_kv_heads=g["KVH"], head_dim=g["D"]),
                      NS(block_size=g["block_size"]), st, g["layer"])
    assert torch.equal(kc.cpu(), g["k_cache"]) and torch.equal(vc.cpu(), g["v_cache"])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("KVH,D,L", [(8, 128, 3), (2, 32, 2), (32, 128, 1)])
def test_store_kvcache_random(dtype, KVH, D, L):
    g = gen(KVH + D)
    bs, layer = 16, L - 1
    plens, dlens = [1, 16, 17, 100], [1, 16, 17, 33, 250]
    seq_ids = [7, 3, 0, 9, 1, 2, 4, 5, 6]
    need = [-(-n // bs) for n in plens + dlens]
    nblocks = sum(need) + 5
    perm = torch.randperm(nblocks, generator=g).tolist()
    bt = torch.zeros(12, 32, dtype=torch.int32)
    for sid, n in zip(seq_ids, need):
        for j in range(n):
            bt[sid, j] = perm.pop()
    T = sum(plens) + len(dlens)
    k = torch.randn(T, KVH, D, generator=g).to(dtype)
    v = torch.randn(T, KVH, D, generator=g).to(dtype)
    kc = torch.randn(nblocks, L, KVH, bs, D, generator=g).to(dtype)     # pre-existing content must survive
    vc = torch.randn(nblocks, L, KVH, bs, D, generator=g).to(dtype)
    mc, ec = NS(num_layers=L, num_kv_heads=KVH, head_dim=D), NS(block_size=bs)
    ekc, evc = kc.clone(), vc.clone()
    ops.store_kvcache(k, v, ekc, evc, bt, mc, ec, _store_state(seq_ids, plens, dlens, "cpu"), layer)
    gkc, gvc = kc.cuda(), vc.cuda()
    K().store_kvcache(k.cuda(), v.cuda(), gkc, gvc, bt.cuda(), mc, ec,
                      _store_state(seq_ids, plens, dlens, "cuda"), layer)
    assert torch.equal(gkc.cpu(), ekc) and torch.equal(gvc.cpu(), evc)


# ---- paged attention ----------------------------------------------------------------------------------
def _paged_case(g, H, KVH, D, L, lens, dtype, layer, extra_blocks=3, bs=16):
    seq_ids = list(range(1, 1 + len(lens)))
    nblk = sum(-(-n // bs) for n in lens) + extra_blocks
    kc = torch.randn(nblk, L, KVH, bs, D, generator=g).to(dtype)
    vc = torch.randn(nblk, L, KVH, bs, D, generator=g).to(dtype)
    perm = torch.randperm(nblk, generator=g).tolist()
    bt = torch.zeros(len(lens) + 2, max(-(-max(lens) // bs), 1) + 1, dtype=torch.int32)
    for sid, n in zip(seq_ids, lens):
        for j in range(-(-n // bs)):
            bt[sid, j] = perm.pop()
    q = torch.randn(len(lens), H, D, generator=g).to(dtype)
    return q, kc, vc, bt, seq_ids


def _paged_state(lens, seq_ids, sbs, D, device):
    return NS(num_decoding_seqs=len(lens), num_prefill_seqs=0, seq_block_size=sbs,
              num_seq_blocks=-(-max(lens) // sbs), softmax_scale=D ** -0.5,
              decoding_seq_lens=torch.tensor(lens, dtype=torch.int32, device=device),
              seq_ids=torch.tensor(seq_ids, dtype=torch.int32, device=device))


def _run_paged(q, kc, vc, bt, lens, seq_ids, sbs, H, KVH, D, L, layer):
    o = torch.zeros_like(q).cuda()
    K().paged_attention(q.cuda(), kc.cuda(), vc.cuda(), bt.cuda(),
                        NS(num_q_heads=H, num_kv_heads=KVH, head_dim=D, num_layers=L), NS(block_size=16),
                        _paged_state(lens, seq_ids, sbs, D, "cuda"), layer, o)
    return o.cpu()


@pytest.mark.parametrize("name", ["gqa4_d128", "mha_d64", "gqa2_d32", "llama3_heads"])
def test_paged_attention_golden(golden, name):
    """Against the reference's Triton kernels (fp16-score path): final output and the partials."""
    from swiftllm_amd import _hip
    g = golden("paged_attention.pt")[name]
    H, KVH, D, L, lens, sbs = g["H"], g["KVH"], g["D"], g["L"], g["lens"], g["seq_block_size"]
    o = _run_paged(g["q"], g["k_cache"], g["v_cache"], g["block_table"], lens, g["seq_ids"], sbs, H, KVH, D, L, g["layer"])
    err = (o.float() - g["out"].float()).abs().max().item()
    assert err <= 4e-3, err
    # phase 1 alone: the partials have the reference's format
    nsb = -(-max(lens) // sbs)
    if nsb > 1:
        mid_o = torch.zeros(len(lens), H, nsb, D, dtype=torch.float32, device="cuda")
        mid_lse = torch.full((len(lens), H, nsb), float("-inf"), dtype=torch.float32, device="cuda")
        st = _paged_state(lens, g["seq_ids"], sbs, D, "cuda")
        q, kc, vc, bt = g["q"].cuda(), g["k_cache"].cuda(), g["v_cache"].cuda(), g["block_table"].cuda()
        _hip.call("swl_paged_attn_phase1", 0, q.data_ptr(), kc.data_ptr(), vc.data_ptr(), bt.data_ptr(),
                  st.seq_ids.data_ptr(), st.decoding_seq_lens.data_ptr(), mid_o.data_ptr(),
                  mid_lse.data_ptr(), st.softmax_scale, len(lens), H, KVH, D, L, 16, g["layer"],
                  bt.shape[1], sbs, nsb, H * D, H * D, _hip.SWL_F16, _hip.stream())
        valid = torch.isfinite(g["mid_lse"])
        assert torch.equal(torch.isfinite(mid_lse.cpu()), valid)
        assert (mid_lse.cpu()[valid] - g["mid_lse"][valid]).abs().max().item() <= 2e-2
        assert (mid_o.cpu()[valid] - g["mid_o"][valid]).abs().max().item() <= 2e-2


@pytest.mark.parametrize("sbs", [256, 64, 2048])    # the reference's own width, a finer split, one split (8 waves)
def test_paged_attention_real_geometry_golden(golden, sbs):
    """Llama-3-8B head geometry at a configs[2]-sized context against the reference's own Triton kernels (interpreter
    golden; inputs regenerated from the seed): <= 1e-3 — the reference's fp16-score noise at this size is 2.4e-4
    (tests/test_oracle_golden.py prints it), ours computes exact scores."""
    from oracle import synth
    g = golden("paged_attention_llama3_1k.pt")
    seq_ids, kc, vc, bt, q, checksum = synth.seeded_paged_case(g["seed"], g["H"], g["KVH"], g["D"], g["L"], g["lens"])
    assert checksum == g["kv_checksum"]
    o = _run_paged(q, kc, vc, bt, g["lens"], seq_ids, sbs, g["H"], g["KVH"], g["D"], g["L"], g["layer"])
    err = (o.float() - g["out"].float()).abs().max().item()
    assert err <= 1e-3, err


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sbs", [128, 1024])    # 4-wave and 8-wave workgroups; 1024 = single split
@pytest.mark.parametrize("H,KVH,D", [(32, 8, 128), (32, 32, 128), (8, 1, 128), (8, 4, 64), (4, 2, 32), (16, 2, 64)])
def test_paged_attention_vs_oracle(dtype, sbs, H, KVH, D):
    g = gen(H * 131 + D)
    lens = [1, 15, 16, 17, 63, 64, 65, 300, 1000]
    L, layer = 2, 1
    q, kc, vc, bt, seq_ids = _paged_case(g, H, KVH, D, L, lens, dtype, layer)
    mc, ec = NS(num_q_heads=H, num_kv_heads=KVH, head_dim=D, num_layers=L), NS(block_size=16)
    eo = torch.zeros_like(q)
    ops.paged_attention(q, kc, vc, bt, mc, ec, _paged_state(lens, seq_ids, sbs, D, "cpu"), layer, eo)
    o = _run_paged(q, kc, vc, bt, lens, seq_ids, sbs, H, KVH, D, L, layer)
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2     # one rounding of an O(1) output
    err = (o.float() - eo.float()).abs().max().item()
    assert err <= tol, err


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sbs", [64, 1024])
@pytest.mark.parametrize("H,KVH,D", [(32, 8, 128), (8, 4, 64), (16, 2, 64), (32, 32, 128)])
def test_paged_attention_is_run_to_run_deterministic(dtype, sbs, H, KVH, D):
    """The same launch on the same inputs must give the same bits every time, 8-wave and 4-wave workgroups, sequences
    spread over one / some / all waves of a workgroup. (r02: the matrix-core kernel with its block maximum taken
    through ds_bpermute returned a different — self-consistent — running maximum from run to run in the 8-wave
    variant; outputs then differed in their last bit. The cross-lane step is VALU-only now.)"""
    g = gen(H * 17 + D + sbs)
    lens = [1, 15, 16, 17, 63, 64, 65, 300, 129, 1000]
    L, layer = 2, 1
    q, kc, vc, bt, seq_ids = _paged_case(g, H, KVH, D, L, lens, dtype, layer)
    mc, ec = NS(num_q_heads=H, num_kv_heads=KVH, head_dim=D, num_layers=L), NS(block_size=16)
    qd, kd, vd, btd = q.cuda(), kc.cuda(), vc.cuda(), bt.cuda()
    st = _paged_state(lens, seq_ids, sbs, D, "cuda")
    outs = []
    for _ in range(12):
        o = torch.zeros_like(qd)
        K().paged_attention(qd, kd, vd, btd, mc, ec, st, layer, o)
        outs.append(o)
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def test_paged_attention_split_invariance_and_properties():
    """Size-independent properties at Llama-3-8B head geometry and long contexts: the result must
    not depend on the split-K width (up to fp32 reassociation), must be linear in V, and must be
    exactly v when a sequence has one token."""
    g = gen(99)
    H, KVH, D, L, layer = 32, 8, 128, 1, 0
    lens = [1, 2049, 4097, 777]
    q, kc, vc, bt, seq_ids = _paged_case(g, H, KVH, D, L, lens, torch.float16, layer)
    outs = [_run_paged(q, kc, vc, bt, lens, seq_ids, sbs, H, KVH, D, L, layer) for sbs in (64, 256, 2048, 8192)]
    for o in outs[1:]:
        assert (o.float() - outs[0].float()).abs().max().item() <= 1e-3
    # one-token sequence: softmax over a single key == 1 -> output == that token's v for every q head
    blk = int(bt[seq_ids[0], 0])
    v0 = vc[blk, layer, :, 0, :].repeat_interleave(H // KVH, dim=0)
    assert torch.equal(outs[0][0], v0)
    # linearity in V: attn(q, K, 2V) == 2 attn(q, K, V). Power-of-two scaling is exact in fp32 all the
    # way to the store; only outputs in fp16's subnormal range (|x| < 6.1e-5, fixed 6e-8 grid) may
    # round differently
    o2 = _run_paged(q, kc, vc * 2, bt, lens, seq_ids, 256, H, KVH, D, L, layer)
    assert (o2.float() - 2 * outs[1].float()).abs().max().item() <= 1.2e-7
    normal = outs[1].float().abs() >= 2.0 ** -13
    assert torch.equal(o2[normal], (outs[1] * 2)[normal])


def test_paged_attention_ignores_stale_tail_and_other_layers():
    """Tokens past len in the last block, other layers and unrelated blocks must not matter."""
    g = gen(5)
    H, KVH, D, L, layer = 8, 2, 128, 3, 1
    lens = [19, 40]
    q, kc, vc, bt, seq_ids = _paged_case(g, H, KVH, D, L, lens, torch.float16, layer)
    base = _run_paged(q, kc, vc, bt, lens, seq_ids, 64, H, KVH, D, L, layer)
    kc2, vc2 = kc.clone(), vc.clone()
    for sid, n in zip(seq_ids, lens):
        blk = int(bt[sid, (n - 1) // 16])
        kc2[blk, layer, :, n % 16:, :] = 77.0 if n % 16 else kc2[blk, layer, :, n % 16:, :]
        vc2[blk, layer, :, n % 16:, :] = -55.0 if n % 16 else vc2[blk, layer, :, n % 16:, :]
    kc2[:, 0], vc2[:, 2] = 9.0, 9.0
    again = _run_paged(q, kc2, vc2, bt, lens, seq_ids, 64, H, KVH, D, L, layer)
    assert torch.equal(base, again)


# ---- prefill attention --------------------------------------------------------------------------------
def _prefill_state(lens, D, device):
    cu = torch.zeros(len(lens) + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(torch.tensor(lens, dtype=torch.int32), 0)
    return NS(num_prefill_seqs=len(lens), max_prefill_len=max(lens), softmax_scale=D ** -0.5,
              prefill_seq_start_locs_with_end=cu.to(device), num_prefill_tokens=sum(lens))


@pytest.mark.parametrize("name", ["gqa2_d64", "gqa4_d128", "mha_d32"])
def test_prefill_attention_golden(golden, name):
    g = golden("prefill_attention.pt")[name]
    o = torch.zeros_like(g["out"]).cuda()
    K().prefill_attention(g["q"].cuda(), g["k"].cuda(), g["v"].cuda(), o,
                          NS(num_q_heads=g["H"], num_kv_heads=g["KVH"], head_dim=g["D"]), None,
                          _prefill_state(g["lens"], g["D"], "cuda"))
    err = (o.cpu().float() - g["out"].float()).abs().max().item()
    assert err <= 2e-3, err


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H,KVH,D", [(32, 8, 128), (8, 8, 128), (8, 2, 64), (4, 2, 32)])
def test_prefill_attention_vs_oracle(dtype, H, KVH, D):
    g = gen(H + KVH + D)
    lens = [1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 300, 513]
    P = sum(lens)
    q = torch.randn(P, H, D, generator=g).to(dtype)
    k = torch.randn(P, KVH, D, generator=g).to(dtype)
    v = torch.randn(P, KVH, D, generator=g).to(dtype)
    mc = NS(num_q_heads=H, num_kv_heads=KVH, head_dim=D)
    eo = torch.zeros_like(q)
    ops.prefill_attention(q, k, v, eo, mc, None, _prefill_state(lens, D, "cpu"))
    o = torch.full_like(q, 7.0).cuda()
    K().prefill_attention(q.cuda(), k.cuda(), v.cuda(), o, mc, None, _prefill_state(lens, D, "cuda"))
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    err = (o.cpu().float() - eo.float()).abs().max().item()
    assert err <= tol, err


def test_prefill_attention_strided_and_causality():
    """q/k/v as column slices of a fused qkv buffer; causality: changing the FUTURE must not change
    the past; first row == v[0]."""
    g = gen(21)
    H, KVH, D, T = 8, 2, 128, 200
    qkv = torch.randn(T, (H + 2 * KVH) * D, generator=g).half().cuda()
    q = qkv[:, :H * D].view(T, H, D)
    k = qkv[:, H * D:(H + KVH) * D].view(T, KVH, D)
    v = qkv[:, (H + KVH) * D:].view(T, KVH, D)
    mc = NS(num_q_heads=H, num_kv_heads=KVH, head_dim=D)
    o1 = torch.zeros(T, H, D, dtype=torch.float16, device="cuda")
    K().prefill_attention(q, k, v, o1, mc, None, _prefill_state([T], D, "cuda"))
    eo = torch.zeros(T, H, D, dtype=torch.float16)
    ops.prefill_attention(q.cpu().contiguous(), k.cpu().contiguous(), v.cpu().contiguous(), eo, mc, None,
                          _prefill_state([T], D, "cpu"))
    assert (o1.cpu().float() - eo.float()).abs().max().item() <= 2e-3
    assert torch.equal(o1[0].cpu(), v[0].cpu().repeat_interleave(H // KVH, dim=0))
    qkv2 = qkv.clone()
    qkv2[150:, H * D:] = torch.randn(50, 2 * KVH * D, generator=g).half().cuda()   # rewrite future k, v
    o2 = torch.zeros_like(o1)
    K().prefill_attention(qkv2[:, :H * D].view(T, H, D), qkv2[:, H * D:(H + KVH) * D].view(T, KVH, D),
                          qkv2[:, (H + KVH) * D:].view(T, KVH, D), o2, mc, None, _prefill_state([T], D, "cuda"))
    assert torch.equal(o1[:150], o2[:150])


def test_prefill_equals_decode_on_last_token():
    """Cross-kernel property at Llama-3-8B geometry, 1024-token prompt: the last row of causal prefill
    attention equals paged decode attention of that row's q over the stored K/V (two independent
    kernels, MFMA vs dot2 paths)."""
    g = gen(31)
    H, KVH, D, T, L = 32, 8, 128, 1024, 1
    q = torch.randn(T, H, D, generator=g).half().cuda()
    k = torch.randn(T, KVH, D, generator=g).half().cuda()
    v = torch.randn(T, KVH, D, generator=g).half().cuda()
    mc, ec = NS(num_q_heads=H, num_kv_heads=KVH, head_dim=D, num_layers=L), NS(block_size=16)
    o = torch.zeros_like(q)
    K().prefill_attention(q, k, v, o, mc, ec, _prefill_state([T], D, "cuda"))
    nblk = T // 16
    kc = torch.zeros(nblk, L, KVH, 16, D, dtype=torch.float16, device="cuda")
    vc = torch.zeros_like(kc)
    bt = torch.arange(nblk, dtype=torch.int32, device="cuda").flip(0).view(1, -1).contiguous()
    K().store_kvcache(k, v, kc, vc, bt, mc, ec, _store_state([0], [T], [], "cuda"), 0)
    od = torch.zeros(1, H, D, dtype=torch.float16, device="cuda")
    st = _paged_state([T], [0], 256, D, "cuda")
    K().paged_attention(q[T - 1:], kc, vc, bt, mc, ec, st, 0, od)
    assert (od[0].float() - o[T - 1].float()).abs().max().item() <= 1.5e-3


# ---- block table kernels ------------------------------------------------------------------------------
def test_block_table_kernels_vs_oracle(golden):
    k = K()
    g = gen(3)
    ms, mb, nb = 10, 32, 200
    num_alloc = torch.zeros(ms, dtype=torch.int32)
    bt = torch.zeros(ms, mb, dtype=torch.int32)
    free = torch.ones(nb, dtype=torch.bool)
    d_alloc, d_bt, d_free = num_alloc.cuda(), bt.cuda(), free.cuda()
    for rnd in range(6):
        ids = torch.randperm(ms, generator=g)[:4].to(torch.int32)
        need = torch.randint(0, 5, (4,), generator=g, dtype=torch.int32)
        cand = torch.nonzero(free)[:int(need.sum())].view(-1).to(torch.int32)
        free[cand.long()] = False
        ops.set_block_table_and_num_seq_alloc_blocks(num_alloc, bt, cand, ids, need)
        k.set_block_table_and_num_seq_alloc_blocks(d_alloc, d_bt, cand.cuda(), ids.cuda(), need.cuda(),
                                                   is_block_free=d_free)
        if rnd % 2:
            ids2 = torch.randperm(ms, generator=g)[:3].to(torch.int32)
            if rnd == 3:
                e = ops.gather_allocated_blocks_and_unset(num_alloc, bt, ids2, free)
                got = k.gather_allocated_blocks_and_unset(d_alloc, d_bt, ids2.cuda(), d_free)
                assert got.cpu().tolist() == e.tolist()
            else:
                ops.unset_block_table_and_num_seq_alloc_blocks(num_alloc, bt, ids2, free)
                k.unset_block_table_and_num_seq_alloc_blocks(d_alloc, d_bt, ids2.cuda(), d_free)
        assert torch.equal(d_alloc.cpu(), num_alloc)
        assert torch.equal(d_free.cpu(), free)
        for s in range(ms):
            n = int(num_alloc[s])
            assert d_bt[s, :n].cpu().tolist() == bt[s, :n].tolist()


# ---- swap ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pinned", [True, False])
def test_swap_blocks_round_trip(pinned):
    g = gen(8)
    shape = (12, 2, 2, 16, 32)
    kc = torch.randn(shape, generator=g).half()
    vc = torch.randn(shape, generator=g).half()
    d_kc, d_vc = kc.cuda(), vc.cuda()
    ks = torch.zeros((8,) + shape[1:], dtype=torch.float16, pin_memory=pinned)
    vs = torch.zeros((8,) + shape[1:], dtype=torch.float16, pin_memory=pinned)
    src, dst = [3, 4, 5, 9, 0], [1, 2, 3, 7, 5]        # one coalescible run of 3, two singles
    K().swap_blocks(src, dst, False, d_kc, d_vc, ks, vs)
    torch.cuda.synchronize()
    eks, evs = torch.zeros_like(ks), torch.zeros_like(vs)
    ops.swap_blocks(src, dst, False, kc, vc, eks, evs)
    assert torch.equal(ks, eks) and torch.equal(vs, evs)
    # swap back in to different GPU blocks: encode -> erase -> decode round trip
    d_kc.zero_()
    d_vc.zero_()
    back = [11, 10, 6, 2, 8]
    K().swap_blocks(dst, back, True, d_kc, d_vc, ks, vs)
    torch.cuda.synchronize()
    for s, b in zip(src, back):
        assert torch.equal(d_kc[b].cpu(), kc[s]) and torch.equal(d_vc[b].cpu(), vc[s])


# ---- skinny GEMM --------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [1, 5, 32])
@pytest.mark.parametrize("N,K", [(4096, 4096), (6144, 4096), (28672, 4096), (4096, 14336), (128, 128), (256, 256), (512, 1024), (96, 384),
                                 # ring kernel (>= 8 K-tiles per chunk): every drain remainder, ragged last workgroup
                                 (160, 1152), (160, 1280), (96, 1408), (64, 1536), (32, 2048 + 128)])
def test_gemm_skinny_vs_fp32_reference(dtype, M, N, K):
    """out = round(x @ W^T) with fp32 accumulation: agreement with an fp32 matmul to one rounding of the
    storage dtype (summation order differs), for every k-split the kernel offers."""
    from swiftllm_amd import _hip
    g = gen(N + K + M)
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(dtype).cuda()
    ref = x.float() @ w.float().T
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    out = K_linear(x, w)
    assert out.shape == (M, N) and out.dtype == dtype
    assert ((out.float() - ref).abs() <= eps * ref.abs() + 1e-3 * eps * (K ** 0.5)).all()
    ws = torch.empty(16 * M * N, dtype=torch.float32, device="cuda")
    for ks in (1, 2, 4, 8, 16):
        if K % (128 * ks):
            continue
        o2 = torch.empty_like(out)
        _hip.call("swl_gemm_skinny", o2.data_ptr(), x.data_ptr(), w.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                  M, N, K, K, N, ks, _hip.dtype_code(dtype), _hip.stream())
        assert ((o2.float() - ref).abs() <= eps * ref.abs() + 1e-3 * eps * (K ** 0.5)).all(), ks
    # strided activations (the down projection reads up_gate[:, :I]) and exact agreement with itself
    wide = torch.randn(M, 2 * K, generator=g).to(dtype).cuda()
    o3 = K_linear(wide[:, :K], w)
    o4 = K_linear(wide[:, :K].contiguous(), w)
    assert torch.equal(o3, o4)


def K_linear(x, w):
    return K().linear(x, w, skinny=True)


def test_linear_falls_back_to_blas_for_prefill_sizes():
    x = torch.randn(33, 256, dtype=torch.float16, device="cuda")
    w = torch.randn(512, 256, dtype=torch.float16, device="cuda")
    assert torch.equal(K().linear(x, w, skinny=True), torch.nn.functional.linear(x, w))


# ---- split-K consumers: fused kernels must equal "reduce, then the plain kernel" bit for bit ---------
@pytest.mark.parametrize("dtype", DTYPES)
def test_splitk_fused_add_rmsnorm_equals_unfused(dtype):
    from swiftllm_amd.worker.kernels.linear import linear_splitk, SplitKPartials
    from swiftllm_amd.worker.kernels.rmsnorm import fused_add_rmsnorm_from_splitk
    g = gen(77)
    M, N, Kd = 32, 4096, 4096
    x = torch.randn(M, Kd, generator=g).to(dtype).cuda()
    w = (torch.randn(N, Kd, generator=g) * 0.02).to(dtype).cuda()
    res = torch.randn(M, N, generator=g).to(dtype).cuda()
    nw = (1 + 0.1 * torch.randn(N, generator=g)).to(dtype).cuda()
    part = linear_splitk(x, w)
    assert isinstance(part, SplitKPartials) and part.k_splits > 1
    y = part.materialize()
    assert torch.equal(y, K().linear(x, w, skinny=True))         # same bits as the self-reducing call
    r1 = res.clone()
    K().fused_add_rmsnorm_inplace(y, r1, nw, 1e-5)
    r2 = res.clone()
    y2 = fused_add_rmsnorm_from_splitk(part, r2, nw, 1e-5)
    assert torch.equal(y2, y) and torch.equal(r2, r1)


@pytest.mark.parametrize("dtype", DTYPES)
def test_splitk_rotary_store_equals_unfused(dtype):
    from swiftllm_amd.worker.kernels.linear import linear_splitk, SplitKPartials
    from swiftllm_amd.worker.kernels.rotary_emb import (rotary_embedding_and_store_kvcache_decode,
                                                        rotary_embedding_and_store_kvcache_decode_from_splitk)
    g = gen(78)
    H, KVH, D, L, layer, nd, hid = 32, 8, 128, 2, 1, 7, 4096
    n = (H + 2 * KVH) * D
    x = torch.randn(nd, hid, generator=g).to(dtype).cuda()
    wqkv = (torch.randn(n, hid, generator=g) * 0.02).to(dtype).cuda()
    lens = [1, 16, 17, 33, 100, 64, 5]
    seq_ids = [3, 0, 6, 1, 2, 5, 4]
    need = [-(-v // 16) for v in lens]
    perm = torch.randperm(sum(need) + 2, generator=g).tolist()
    bt = torch.zeros(8, 16, dtype=torch.int32)
    for sid, c in zip(seq_ids, need):
        for j in range(c):
            bt[sid, j] = perm.pop()
    bt = bt.cuda()
    ang = torch.rand(128, D // 2, generator=g) * 6.28
    st = NS(num_prefill_seqs=0, num_decoding_seqs=nd, position_cos=torch.cos(ang).to(dtype).cuda(),
            position_sin=torch.sin(ang).to(dtype).cuda(),
            position_indices=torch.tensor([v - 1 for v in lens], dtype=torch.int32, device="cuda"),
            seq_ids=torch.tensor(seq_ids, dtype=torch.int32, device="cuda"),
            decoding_seq_lens=torch.tensor(lens, dtype=torch.int32, device="cuda"))
    mc, ec = NS(num_q_heads=H, num_kv_heads=KVH, head_dim=D, num_layers=L), NS(block_size=16)
    shape = (sum(need) + 2, L, KVH, 16, D)
    part = linear_splitk(x, wqkv)
    assert isinstance(part, SplitKPartials)
    qkv = part.materialize()
    q1 = qkv[:, :H * D].view(nd, H, D)
    k1 = qkv[:, H * D:(H + KVH) * D].view(nd, KVH, D)
    v1 = qkv[:, (H + KVH) * D:].view(nd, KVH, D)
    kc1, vc1 = torch.zeros(shape, dtype=dtype, device="cuda"), torch.zeros(shape, dtype=dtype, device="cuda")
    rotary_embedding_and_store_kvcache_decode(q1, k1, v1, kc1, vc1, bt, mc, ec, st, layer)
    kc2, vc2 = torch.zeros_like(kc1), torch.zeros_like(vc1)
    q2, k2, v2 = rotary_embedding_and_store_kvcache_decode_from_splitk(part, kc2, vc2, bt, mc, ec, st, layer)
    assert torch.equal(q2, q1) and torch.equal(k2, k1) and torch.equal(v2, v1)
    assert torch.equal(kc2, kc1) and torch.equal(vc2, vc1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,I,Kd", [(32, 14336, 4096), (7, 11008, 4096), (1, 256, 128), (32, 96, 384),
                                    (9, 96, 1152), (32, 160, 1280)])
def test_gemm_silu_gate_fusion_equals_two_ops(dtype, M, I, Kd):
    """linear_silu_gate == skinny linear (one k-split) followed by silu_and_mul, bit for bit."""
    from swiftllm_amd import _hip
    from swiftllm_amd.worker.kernels.linear import linear_silu_gate
    g = gen(M + I)
    x = torch.randn(M, Kd, generator=g).to(dtype).cuda()
    w = (torch.randn(2 * I, Kd, generator=g) * 0.03).to(dtype).cuda()
    fused = linear_silu_gate(x, w)
    assert fused is not None and fused.shape == (M, I)
    two = torch.empty(M, 2 * I, dtype=dtype, device="cuda")
    _hip.call("swl_gemm_skinny", two.data_ptr(), x.data_ptr(), w.data_ptr(), 0, 0, M, 2 * I, Kd, Kd, 2 * I, 1,
              _hip.dtype_code(dtype), _hip.stream())
    K().silu_and_mul_inplace(two)
    assert torch.equal(fused, two[:, :I])
    # and against the oracle's arithmetic on fp32-accumulated projections (tolerance: one rounding of T)
    ref = (x.float() @ w.float().T).to(dtype).cpu()
    ops.silu_and_mul_inplace(ref)
    # three chained roundings (up, silu(gate), product) on projections that may each be 1 ulp off
    eps = 2.0 ** -7 if dtype == torch.float16 else 2.0 ** -4
    assert ((fused.cpu().float() - ref[:, :I].float()).abs() <= eps * ref[:, :I].float().abs() + 2e-3).all()


# ---- greedy sampling -----------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,n", [(1, 8), (3, 512), (32, 128256), (7, 32000), (5, 1000)])
def test_argmax_rows_matches_torch_with_lowest_index_ties(dtype, rows, n):
    from swiftllm_amd.worker.kernels.sampling import argmax_rows
    g = gen(rows * 31 + n)
    x = torch.randn(rows, n, generator=g).to(dtype)
    # plant exact ties for the maximum, far apart (different splits, waves and lanes)
    for r in range(rows):
        top = x[r].max() + 1
        for j in {(r * 37) % n, n - 1 - (r * 11) % n, n // 2}:
            x[r, j] = top
    want = torch.tensor([int((x[r] == x[r].max()).nonzero()[0]) for r in range(rows)])
    got = argmax_rows(x.cuda())
    assert got.dtype == torch.int64 and torch.equal(got.cpu(), want)
    wide = torch.randn(rows, n + 64, generator=g).to(dtype).cuda()          # strided rows (a column slice)
    assert torch.equal(argmax_rows(wide[:, :n]), torch.argmax(wide[:, :n].float(), dim=1))
    neg = torch.full((2, 64), float("-inf"), dtype=dtype).cuda()
    assert argmax_rows(neg).tolist() == [0, 0]


# ---- paged attention with rotary + KV store in its prologue ------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sbs", [64, 1024])      # several splits (only the last owns the new token) / one split
@pytest.mark.parametrize("H,KVH,D,hid", [(32, 8, 128, 4096), (8, 8, 128, 512), (8, 1, 128, 1024), (8, 4, 64, 256),
                                         (4, 2, 32, 128)])
def test_paged_attention_from_qkv_slabs_equals_three_kernels(dtype, sbs, H, KVH, D, hid):
    """qkv slabs -> [rotary + KV store + paged attention] in one launch == split-K rotary/store kernel followed by
    the paged-attention kernel: outputs and both pools bit for bit (k_splits > 1 and the single-slab case)."""
    from swiftllm_amd.worker.kernels.linear import linear_splitk, SplitKPartials
    from swiftllm_amd.worker.kernels.paged_attn import paged_attention_from_qkv_splitk
    from swiftllm_amd.worker.kernels.rotary_emb import rotary_embedding_and_store_kvcache_decode_from_splitk
    g = gen(H * 3 + D + hid + sbs)
    L, layer = 2, 1
    lens = [1, 15, 16, 17, 63, 64, 65, 300, 129]
    nd = len(lens)
    _, kc, vc, bt, seq_ids = _paged_case(g, H, KVH, D, L, lens, dtype, layer)
    n = (H + 2 * KVH) * D
    x = torch.randn(nd, hid, generator=g).to(dtype).cuda()
    wqkv = (torch.randn(n, hid, generator=g) * (hid ** -0.5)).to(dtype).cuda()
    ang = torch.rand(512, D // 2, generator=g) * 6.28
    st = _paged_state(lens, seq_ids, sbs, D, "cuda")
    st.position_cos, st.position_sin = torch.cos(ang).to(dtype).cuda(), torch.sin(ang).to(dtype).cuda()
    st.position_indices = torch.tensor([v - 1 for v in lens], dtype=torch.int32, device="cuda")
    mc, ec = NS(num_q_heads=H, num_kv_heads=KVH, head_dim=D, num_layers=L), NS(block_size=16)
    part = linear_splitk(x, wqkv, always=True)
    assert isinstance(part, SplitKPartials)
    kc1, vc1, btc = kc.cuda(), vc.cuda(), bt.cuda()
    q1, _, _ = rotary_embedding_and_store_kvcache_decode_from_splitk(part, kc1, vc1, btc, mc, ec, st, layer)
    o1 = torch.zeros(nd, H, D, dtype=dtype, device="cuda")
    K().paged_attention(q1, kc1, vc1, btc, mc, ec, st, layer, o1)
    kc2, vc2 = kc.cuda(), vc.cuda()
    o2 = torch.zeros(nd, H * D, dtype=dtype, device="cuda")
    paged_attention_from_qkv_splitk(part, kc2, vc2, btc, mc, ec, st, layer, o2)
    assert torch.equal(o2.view(nd, H, D), o1)
    assert torch.equal(kc2, kc1) and torch.equal(vc2, vc1)
    assert not torch.equal(kc2.cpu(), kc)                      # the new token really went into the pool


# ---- pre-packed weights ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [1, 7, 32])
@pytest.mark.parametrize("N,K", [(4096, 4096), (6144, 4096), (28672, 4096), (4096, 14336), (128, 128), (96, 384),
                                 (160, 1152), (64, 1536), (32, 2048 + 128)])
def test_packed_weight_gemms_equal_row_major_bits(dtype, M, N, K):
    """W repacked in MFMA-fragment order: swl_gemm_skinny_packed / _partial / _silu_gate give the bits of their
    row-major twins for every k-split (same MFMA order), through the operator API too."""
    from swiftllm_amd import _hip
    from swiftllm_amd.worker.kernels import linear as _  # noqa: F401
    import importlib
    L = importlib.import_module("swiftllm_amd.worker.kernels.linear")
    g = gen(N + K + M)
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(dtype).cuda()
    plain = L.linear(x, w, skinny=True)
    ws = torch.empty(16 * M * N, dtype=torch.float32, device="cuda")
    code = _hip.dtype_code(dtype)
    wp = torch.empty_like(w)
    _hip.call("swl_gemm_pack_weight", wp.data_ptr(), w.data_ptr(), N, K, code, _hip.stream())
    lib_ = _hip.load()
    same_default_split = lib_.swl_gemm_skinny_packed_choose_splits(N, K) == lib_.swl_gemm_skinny_choose_splits(N, K)
    tol = (2.0 ** -9 if dtype == torch.float16 else 2.0 ** -6) * max(1.0, float(plain.float().abs().max()))
    for ks in (0, 1, 2, 4, 8, 16):
        if ks and K % (128 * ks):
            continue
        a, b = torch.empty_like(plain), torch.empty_like(plain)
        _hip.call("swl_gemm_skinny", a.data_ptr(), x.data_ptr(), w.data_ptr(), ws.data_ptr(), ws.numel() * 4, M, N, K,
                  K, N, ks, code, _hip.stream())
        _hip.call("swl_gemm_skinny_packed", b.data_ptr(), x.data_ptr(), wp.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                  M, N, K, K, N, ks, code, _hip.stream())
        if ks == 0 and not same_default_split:
            # (the packed path may split K into chunks that differ by one tile where the row-major kernels cannot:
            # another summation order of the same products — test_packed_gemm_uneven_k_splits)
            assert (a.float() - b.float()).abs().max().item() <= tol, ks
        else:
            assert torch.equal(a, b), ks
    # operator API: attaching the packed copy switches linear / linear_splitk / linear_silu_gate over, same bits
    part0 = L.linear_splitk(x, w, always=True)
    s0 = part0.materialize() if isinstance(part0, L.SplitKPartials) else part0
    gate0 = L.linear_silu_gate(x, w) if N % 64 == 0 else None
    L.pack_weight(w)
    part1 = L.linear_splitk(x, w, always=True)
    s1 = part1.materialize() if isinstance(part1, L.SplitKPartials) else part1
    if same_default_split:
        assert torch.equal(L.linear(x, w, skinny=True), plain)
        assert torch.equal(s1, s0)
    else:
        assert (L.linear(x, w, skinny=True).float() - plain.float()).abs().max().item() <= tol
        assert (s1.float() - s0.float()).abs().max().item() <= tol
    if gate0 is not None:
        assert torch.equal(L.linear_silu_gate(x, w), gate0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [33, 64, 48, 7])       # (65..128 tokens: csrc/gemm_wide.hip since r04; the MT = 4 form is gone, r05)
@pytest.mark.parametrize("N,K", [(4096, 4096), (6144, 4096), (28672, 4096), (4096, 14336), (32, 128), (160, 256),
                                 (96, 384), (64, 1152), (448, 1024)])
def test_gemm_packed_mid_vs_fp32_reference(dtype, M, N, K):
    """Medium-batch GEMM on packed weights (2 / 4 token blocks per weight fragment): one rounding of an
    fp32-accumulated product for the library's split choice and forced ones; ragged workgroups, odd tile counts; more than 64
    tokens are refused (SWL_ERR_UNSUPPORTED)."""
    from swiftllm_amd import _hip
    g = gen(N + K + M)
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(dtype).cuda()
    ref = x.float() @ w.float().T
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    code = _hip.dtype_code(dtype)
    wp = torch.empty_like(w)
    _hip.call("swl_gemm_pack_weight", wp.data_ptr(), w.data_ptr(), N, K, code, _hip.stream())
    ws = torch.empty(16 * M * N, dtype=torch.float32, device="cuda")
    for ks in (0, 1, 2, 4, 8):
        if ks and K % (128 * ks):
            continue
        out = torch.full((M, N), float("nan"), dtype=dtype, device="cuda")
        _hip.call("swl_gemm_packed_mid", out.data_ptr(), x.data_ptr(), wp.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                  M, N, K, K, N, ks, code, _hip.stream())
        assert ((out.float() - ref).abs() <= eps * ref.abs() + 1e-3 * eps * (K ** 0.5)).all(), ks


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,I,Kd", [(48, 14336, 4096), (64, 11008, 4096), (33, 256, 128), (40, 96, 384),
                                    (64, 160, 1280), (7, 96, 1152)])
def test_gemm_packed_mid_silu_gate_equals_two_ops(dtype, M, I, Kd):
    """Medium-batch SiLU-gate GEMM == medium-batch GEMM (one k-split) followed by silu_and_mul, bit for bit."""
    from swiftllm_amd import _hip
    g = gen(M + I)
    x = torch.randn(M, Kd, generator=g).to(dtype).cuda()
    w = (torch.randn(2 * I, Kd, generator=g) * 0.03).to(dtype).cuda()
    code = _hip.dtype_code(dtype)
    wp = torch.empty_like(w)
    _hip.call("swl_gemm_pack_weight", wp.data_ptr(), w.data_ptr(), 2 * I, Kd, code, _hip.stream())
    fused = torch.empty(M, I, dtype=dtype, device="cuda")
    _hip.call("swl_gemm_packed_mid_silu_gate", fused.data_ptr(), x.data_ptr(), wp.data_ptr(), M, I, Kd, Kd, I, code,
              _hip.stream())
    two = torch.empty(M, 2 * I, dtype=dtype, device="cuda")
    _hip.call("swl_gemm_packed_mid", two.data_ptr(), x.data_ptr(), wp.data_ptr(), 0, 0, M, 2 * I, Kd, Kd, 2 * I, 1, code,
              _hip.stream())
    K().silu_and_mul_inplace(two)
    assert torch.equal(fused, two[:, :I])


# ---- large decode batches: 64 < M <= 256 tokens on packed weights (csrc/gemm_wide.hip) -------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [65, 100, 128, 129, 192, 193, 256, 40])
@pytest.mark.parametrize("N,K", [(4096, 4096), (6144, 4096), (4096, 14336), (32, 64), (160, 256), (96, 384), (448, 1024),
                                 (2080, 832)])
def test_gemm_packed_wide_vs_fp32_reference(dtype, M, N, K):
    """Large-batch GEMM on packed weights (up to 8 token blocks per weight fragment, x^T shared through LDS): one rounding
    of an fp32-accumulated product for the library's plan and for forced workgroup widths / K splits; ragged workgroups,
    token counts that are not multiples of 32, K-chunks shorter than the weight ring. At one K split it carries the bits
    of the medium-batch kernel (same MFMA order)."""
    from swiftllm_amd import _hip
    g = gen(N + K + M)
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(dtype).cuda()
    ref = x.float() @ w.float().T
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    code = _hip.dtype_code(dtype)
    wp = torch.empty_like(w)
    _hip.call("swl_gemm_pack_weight", wp.data_ptr(), w.data_ptr(), N, K, code, _hip.stream())
    ws = torch.empty(16 * M * N, dtype=torch.float32, device="cuda")
    mid = None
    if M <= 64 and K % 128 == 0:
        mid = torch.empty(M, N, dtype=dtype, device="cuda")
        _hip.call("swl_gemm_packed_mid", mid.data_ptr(), x.data_ptr(), wp.data_ptr(), 0, 0, M, N, K, K, N, 1, code,
                  _hip.stream())
    by_split = {}
    for nwv in (0, 4, 8):
        for ks in (0, 1, 2, 4, 8):
            if ks and K % (64 * ks):
                continue
            out = torch.full((M, N), float("nan"), dtype=dtype, device="cuda")
            _hip.call("swl_gemm_packed_wide", out.data_ptr(), x.data_ptr(), wp.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                      M, N, K, K, N, nwv, ks, code, _hip.stream())
            assert ((out.float() - ref).abs() <= eps * ref.abs() + 1e-3 * eps * (K ** 0.5)).all(), (nwv, ks)
            if ks == 1 and mid is not None:
                assert torch.equal(out, mid), nwv
            if ks:
                # the register tiling is not allowed to show: 4-wave groups (token split where the plan picks it, r06d)
                # and 8-wave groups (never split) sum every output in the same K order
                assert torch.equal(by_split.setdefault(ks, out), out), (nwv, ks)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [65, 128, 160, 192, 200, 256])
def test_gemm_packed_wide_up_gate_shape(dtype, M):
    """The widest projection of a decode layer (Llama-3-8B up/gate: 28672 x 4096) through the plain wide kernel, one
    M per token-block count 3..8."""
    from swiftllm_amd import _hip
    N, K = 28672, 4096
    g = gen(M)
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * 0.03).to(dtype).cuda()
    ref = x.float() @ w.float().T
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    code = _hip.dtype_code(dtype)
    wp = torch.empty_like(w)
    _hip.call("swl_gemm_pack_weight", wp.data_ptr(), w.data_ptr(), N, K, code, _hip.stream())
    for nwv in (0, 8):
        out = torch.full((M, N), float("nan"), dtype=dtype, device="cuda")
        _hip.call("swl_gemm_packed_wide", out.data_ptr(), x.data_ptr(), wp.data_ptr(), 0, 0, M, N, K, K, N, nwv, 1, code,
                  _hip.stream())
        assert ((out.float() - ref).abs() <= eps * ref.abs() + 1e-3 * eps * (K ** 0.5)).all(), nwv


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,I,Kd", [(256, 14336, 4096), (128, 14336, 4096), (192, 11008, 4096), (65, 256, 128), (100, 96, 384),
                                    (200, 160, 1280), (129, 96, 1152), (40, 64, 64)])
def test_gemm_packed_wide_silu_gate_equals_two_ops(dtype, M, I, Kd):
    """Large-batch SiLU-gate GEMM == large-batch GEMM (one k-split) followed by silu_and_mul, bit for bit, for both
    workgroup widths (two exchange rounds of MT/2 token blocks through the x buffers)."""
    from swiftllm_amd import _hip
    g = gen(M + I)
    x = torch.randn(M, Kd, generator=g).to(dtype).cuda()
    w = (torch.randn(2 * I, Kd, generator=g) * 0.03).to(dtype).cuda()
    code = _hip.dtype_code(dtype)
    wp = torch.empty_like(w)
    _hip.call("swl_gemm_pack_weight", wp.data_ptr(), w.data_ptr(), 2 * I, Kd, code, _hip.stream())
    two = torch.empty(M, 2 * I, dtype=dtype, device="cuda")
    _hip.call("swl_gemm_packed_wide", two.data_ptr(), x.data_ptr(), wp.data_ptr(), 0, 0, M, 2 * I, Kd, Kd, 2 * I, 0, 1, code,
              _hip.stream())
    K().silu_and_mul_inplace(two)
    for nwv in (0, 4, 8):
        fused = torch.full((M, I), float("nan"), dtype=dtype, device="cuda")
        _hip.call("swl_gemm_packed_wide_silu_gate", fused.data_ptr(), x.data_ptr(), wp.data_ptr(), M, I, Kd, Kd, I, nwv, code,
                  _hip.stream())
        assert torch.equal(fused, two[:, :I]), nwv


# ---- deferred RMSNorm (decode fast path) ---------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,hidden", [(32, 4096), (1, 4096), (7, 1024), (5, 8192)])
def test_add_scale_from_splitk(dtype, M, hidden):
    """The element-wise split-K consumer: residual gets the bits fused_add_rmsnorm_from_splitk gives it, x_scaled is
    exactly round(residual * w), and the sums of squares add up to the row's (fp64 reference, fp32 summation order)."""
    from swiftllm_amd.worker.kernels.linear import linear_splitk, SplitKPartials
    from swiftllm_amd.worker.kernels.rmsnorm import add_scale_from_splitk, fused_add_rmsnorm_from_splitk
    g = gen(M + hidden)
    x = torch.randn(M, 4096, generator=g).to(dtype).cuda()
    w = (torch.randn(hidden, 4096, generator=g) * 0.02).to(dtype).cuda()
    res = torch.randn(M, hidden, generator=g).to(dtype).cuda()
    nw = (1 + 0.1 * torch.randn(hidden, generator=g)).to(dtype).cuda()
    part = linear_splitk(x, w, always=True)
    assert isinstance(part, SplitKPartials)
    r_exact = res.clone()
    fused_add_rmsnorm_from_splitk(part, r_exact, nw, 1e-5)
    r_def = res.clone()
    pend = add_scale_from_splitk(part, r_def, nw, 1e-5, unsafe_float16_ok=True)
    assert torch.equal(r_def, r_exact)
    assert torch.equal(pend.x, (r_def.float() * nw.float()).to(dtype))
    assert pend.ssq.shape == (hidden // 1024, M)
    want = r_def.double().pow(2).view(M, hidden // 1024, 1024).sum(-1).T
    assert torch.allclose(pend.ssq.double(), want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [1, 32])
def test_deferred_rmsnorm_ffn_matches_exact_path(dtype, M):
    """o_proj slabs -> [add + scale] -> SiLU-gate GEMM with the 1/rms in its epilogue, against the exact-rounding path
    (fused_add_rmsnorm, then the same GEMM) and against an fp64 evaluation of the FFN input/gate: the deferred path
    moves ONE rounding (un-normalised activations are rounded instead of normalised ones), so the two agree to a few
    ulps of the storage dtype and are equally far from the fp64 value."""
    from swiftllm_amd.worker.kernels.linear import linear_splitk, linear_silu_gate, pack_weight
    from swiftllm_amd.worker.kernels.rmsnorm import add_scale_from_splitk, fused_add_rmsnorm_from_splitk
    g = gen(M * 3 + 1)
    h, inter = 4096, 14336
    a = torch.randn(M, h, generator=g).to(dtype).cuda()
    wo = (torch.randn(h, h, generator=g) * 0.02).to(dtype).cuda()
    res = (3 * torch.randn(M, h, generator=g)).to(dtype).cuda()
    nw = (1 + 0.1 * torch.randn(h, generator=g)).to(dtype).cuda()
    wug = (torch.randn(2 * inter, h, generator=g) * 0.02).to(dtype).cuda()
    pack_weight(wug)
    part = linear_splitk(a, wo)
    r1 = res.clone()
    xn = fused_add_rmsnorm_from_splitk(part, r1, nw, 1e-5)
    exact = linear_silu_gate(xn, wug)
    r2 = res.clone()
    pend = add_scale_from_splitk(part, r2, nw, 1e-5, unsafe_float16_ok=True)
    got = linear_silu_gate(pend.x, wug, row_scale=pend)
    assert torch.equal(r1, r2)
    # fp64 value of up * silu(gate) on the un-rounded normalised activations
    rd = r1.double()
    xd = rd / torch.sqrt(rd.pow(2).mean(-1, keepdim=True) + 1e-5) * nw.double()
    ug = xd @ wug.double().T
    ref = ug[:, :inter] * (ug[:, inter:] / (1 + torch.exp(-ug[:, inter:])))
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    scale = ref.abs().amax(dim=1, keepdim=True).clamp(min=1e-3)
    err_exact = ((exact.double() - ref).abs() / scale).max().item()
    err_def = ((got.double() - ref).abs() / scale).max().item()
    print(f"\n[deferred rmsnorm, FFN, {dtype}, M={M}] max |err| / row max: exact path {err_exact:.3e}, deferred {err_def:.3e} (eps {eps:.1e})")
    assert err_def <= 3 * eps and err_def <= 1.5 * err_exact + eps
    assert ((got.double() - exact.double()).abs() / scale).max().item() <= 3 * eps


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sbs", [64, 1024])
def test_deferred_rmsnorm_attention_matches_exact_path(dtype, sbs):
    """down_proj slabs -> [add + scale] -> qkv slabs -> slab-fed attention with the 1/rms applied in its prologue,
    against fused_add_rmsnorm -> the same projection -> the same attention: rotated k / v written to the pools within
    one rounding step of the exact path's, attention output within the storage dtype's resolution."""
    from swiftllm_amd.worker.kernels.linear import linear_splitk, SplitKPartials
    from swiftllm_amd.worker.kernels.paged_attn import paged_attention_from_qkv_splitk
    from swiftllm_amd.worker.kernels.rmsnorm import add_scale_from_splitk, fused_add_rmsnorm_from_splitk
    H, KVH, D, hid = 32, 8, 128, 4096
    g = gen(sbs + 17)
    L, layer = 2, 1
    lens = [1, 15, 16, 17, 63, 64, 65, 300, 129]
    nd = len(lens)
    _, kc, vc, bt, seq_ids = _paged_case(g, H, KVH, D, L, lens, dtype, layer)
    n = (H + 2 * KVH) * D
    a = torch.randn(nd, 14336, generator=g).to(dtype).cuda()
    wdown = (torch.randn(hid, 14336, generator=g) * 0.01).to(dtype).cuda()
    res = (2 * torch.randn(nd, hid, generator=g)).to(dtype).cuda()
    nw = (1 + 0.1 * torch.randn(hid, generator=g)).to(dtype).cuda()
    wqkv = (torch.randn(n, hid, generator=g) * (hid ** -0.5)).to(dtype).cuda()
    ang = torch.rand(512, D // 2, generator=g) * 6.28
    st = _paged_state(lens, seq_ids, sbs, D, "cuda")
    st.position_cos, st.position_sin = torch.cos(ang).to(dtype).cuda(), torch.sin(ang).to(dtype).cuda()
    st.position_indices = torch.tensor([v - 1 for v in lens], dtype=torch.int32, device="cuda")
    mc, ec = NS(num_q_heads=H, num_kv_heads=KVH, head_dim=D, num_layers=L), NS(block_size=16)
    btc = bt.cuda()
    down = linear_splitk(a, wdown)
    # (the slabs live in the shared split-K workspace, which the qkv projections below reuse: keep a private copy)
    down = SplitKPartials(down.slabs[:down.k_splits * nd * hid].clone(), down.k_splits, nd, hid, down.dtype)
    # exact path
    r1 = res.clone()
    xn = fused_add_rmsnorm_from_splitk(down, r1, nw, 1e-5)
    kc1, vc1 = kc.cuda(), vc.cuda()
    o1 = torch.zeros(nd, H * D, dtype=dtype, device="cuda")
    paged_attention_from_qkv_splitk(linear_splitk(xn, wqkv, always=True), kc1, vc1, btc, mc, ec, st, layer, o1)
    # deferred path
    r2 = res.clone()
    pend = add_scale_from_splitk(down, r2, nw, 1e-5, unsafe_float16_ok=True)
    kc2, vc2 = kc.cuda(), vc.cuda()
    o2 = torch.zeros(nd, H * D, dtype=dtype, device="cuda")
    paged_attention_from_qkv_splitk(linear_splitk(pend.x, wqkv, always=True), kc2, vc2, btc, mc, ec, st, layer, o2,
                                    row_scale=pend)
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    assert torch.equal(r1, r2)
    dk = (kc2.float() - kc1.float()).abs().max().item()
    dv = (vc2.float() - vc1.float()).abs().max().item()
    do = (o2.float() - o1.float()).abs().max().item()
    print(f"\n[deferred rmsnorm, attention, {dtype}, sbs={sbs}] max |dk| {dk:.3e} |dv| {dv:.3e} |do| {do:.3e}")
    kmax = kc1.float().abs().max().item()
    assert dk <= 4 * eps * kmax and dv <= 4 * eps * kmax
    assert do <= 4 * eps * max(1.0, o1.float().abs().max().item())
    assert not torch.equal(kc2.cpu(), kc)


# ---- very small decode batches: the projection consumes the previous projection's slabs itself (gemm_tiny.hip) ----
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [1, 2, 3, 4])
@pytest.mark.parametrize("hid,N,inter", [(4096, 6144, 14336), (512, 768, 1024), (1024, 256, 384)])
def test_tiny_batch_projections_equal_consumer_plus_gemm(dtype, M, hid, N, inter):
    """linear_splitk_from_splitk / linear_silu_gate_from_splitk (one launch each) against the two-launch paths they replace:
    add_scale_from_splitk + linear_splitk, add_scale_from_splitk + linear_silu_gate(row_scale). Residual and qkv slabs
    bit for bit (same slab order, same MFMA order); the sums of squares are grouped by K-chunk instead of by 1024
    columns, so they — and the SiLU-gate output through its 1/rms — may differ by fp32 summation order: a last-bit
    difference in rstd can flip the rounding of the up and of the gate projection (one ulp each), silu and the product
    carry them on: <= 4 ulp of the storage dtype on the output."""
    import importlib
    L = importlib.import_module("swiftllm_amd.worker.kernels.linear")
    R = importlib.import_module("swiftllm_amd.worker.kernels.rmsnorm")
    g = gen(M * 7 + hid + N)
    eps = 1e-5
    wq = (torch.randn(N, hid, generator=g) * hid ** -0.5).to(dtype).cuda()
    wug = (torch.randn(2 * inter, hid, generator=g) * hid ** -0.5).to(dtype).cuda()
    L.pack_weight(wq); L.pack_weight(wug)
    norm_w = (1 + 0.1 * torch.randn(hid, generator=g)).to(dtype).cuda()
    res0 = torch.randn(M, hid, generator=g).to(dtype).cuda()
    ks_in = 8
    slabs = (torch.randn(ks_in, M, hid, generator=g) * 0.3).float().cuda().contiguous()
    part = L.SplitKPartials(slabs.view(-1), ks_in, M, hid, dtype)
    if not L.tiny_from_splitk_ok(part, wq):
        pytest.skip("shape outside the tiny-batch kernel's limits")
    # two-launch reference
    r_ref = res0.clone()
    pend = R.add_scale_from_splitk(part, r_ref, norm_w, eps, unsafe_float16_ok=True) if R.deferred_norm_ok(M, hid) else None
    if pend is None:
        pytest.skip("deferred norm needs hidden % 1024 == 0 for the two-launch twin")
    q_ref = L.linear_splitk(pend.x, wq, always=True)
    q_ref_sum = q_ref.slabs[: q_ref.k_splits * M * N].view(q_ref.k_splits, M, N).clone()
    act_ref = L.linear_silu_gate(pend.x, wug, row_scale=pend)
    # one launch each
    r_new = torch.empty_like(res0)
    q_new, ssq = L.linear_splitk_from_splitk(part, res0, r_new, norm_w, wq)
    assert q_new.k_splits == q_ref.k_splits
    assert torch.equal(r_new, r_ref)
    assert torch.equal(q_new.slabs[: q_new.k_splits * M * N].view(q_new.k_splits, M, N), q_ref_sum)
    tot_ref, tot_new = pend.ssq.sum(0), ssq.sum(0)
    assert ((tot_ref - tot_new).abs() <= 2e-6 * tot_ref).all()
    r_new2 = torch.empty_like(res0)
    act_new = L.linear_silu_gate_from_splitk(part, res0, r_new2, norm_w, eps, wug)
    assert torch.equal(r_new2, r_ref)
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    err = (act_new.float() - act_ref.float()).abs()
    assert (err <= 4 * ulp * act_ref.float().abs() + 1e-5).all(), err.max().item()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [1, 2, 4])
def test_tiny_batch_o_proj_merges_attention_partials(dtype, M):
    """linear_splitk_from_attn_partials (o_proj that merges the flash-decoding partials itself) against phase 2 +
    linear_splitk: the merged rows may differ from phase 2's by the fp32 order of the weight sum (one ulp of the storage
    dtype, rarely), so the projections agree to a few ulp of their own scale and almost everywhere exactly."""
    import importlib
    from swiftllm_amd import _hip
    L = importlib.import_module("swiftllm_amd.worker.kernels.linear")
    H, D, nsb, sbs = 32, 128, 6, 64
    hid = H * D
    g = gen(M * 13 + 5)
    lens = [300, 70, 64, 381][:M]
    w = (torch.randn(hid, hid, generator=g) * hid ** -0.5).to(dtype).cuda()
    L.pack_weight(w)
    mid_o = torch.randn(M, H, nsb, D, generator=g).float()
    mid_lse = (torch.randn(M, H, nsb, generator=g) * 3).float()
    scratch = torch.cat([mid_o.view(-1), mid_lse.view(-1)]).cuda()
    seq_lens = torch.tensor(lens, dtype=torch.int32, device="cuda")
    assert L.attn_partials_ok(M, H, D, w)
    # reference: the phase-2 kernel, then the ordinary split-K projection
    o = torch.empty(M, H, D, dtype=dtype, device="cuda")
    mo = scratch[: M * H * nsb * D]
    ml = scratch[M * H * nsb * D:]
    _hip.call("swl_paged_attn_phase2", o.data_ptr(), mo.data_ptr(), ml.data_ptr(), seq_lens.data_ptr(), M, H, D, sbs, nsb,
              H * D, _hip.dtype_code(dtype), _hip.stream())
    ref = L.linear_splitk(o.view(M, hid), w)
    ref_out = ref.materialize().float() if isinstance(ref, L.SplitKPartials) else ref.float()
    new = L.linear_splitk_from_attn_partials(scratch, seq_lens, M, H, D, sbs, nsb, w, dtype)
    new_out = new.materialize().float()
    # also against an fp64 merge + product
    n_used = [-(-n // sbs) for n in lens]
    x64 = torch.zeros(M, hid, dtype=torch.float64)
    for m_ in range(M):
        for h in range(H):
            lw = torch.exp2((mid_lse[m_, h, :n_used[m_]] - mid_lse[m_, h, :n_used[m_]].max()).double())
            x64[m_, h * D:(h + 1) * D] = (lw[:, None] * mid_o[m_, h, :n_used[m_]].double()).sum(0) / lw.sum()
    out64 = x64.to(dtype).double() @ w.double().cpu().t()
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    scale = max(1.0, float(out64.abs().max()))
    assert (new_out.double().cpu() - out64).abs().max().item() <= 4 * eps * scale
    assert (new_out - ref_out).abs().max().item() <= 4 * eps * scale
    assert (new_out == ref_out).float().mean().item() > 0.9


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [1, 4, 32])
@pytest.mark.parametrize("N,K", [(4096, 11008), (4096, 9856), (2048, 6784), (4096, 4096), (256, 11008)])
def test_packed_gemm_uneven_k_splits(dtype, M, N, K):
    """K with no power-of-two split of whole 128-column tiles that fills the chip (Llama-2-7B down_proj: 11008 = 86
    tiles): the packed kernels split it into chunks that differ by one tile. Reduced output and partial slabs against
    an fp64 product; the chooser entry agrees with what the kernel does (slab count); shapes with an even split keep
    their old split count and bits."""
    import importlib
    from swiftllm_amd import _hip
    L = importlib.import_module("swiftllm_amd.worker.kernels.linear")
    lib = _hip.load()
    g = gen(N + K + M)
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).cuda()
    L.pack_weight(w)
    ref = x.double().cpu() @ w.double().cpu().t()
    ks = lib.swl_gemm_skinny_packed_choose_splits(N, K)
    ks_even = lib.swl_gemm_skinny_choose_splits(N, K)
    assert ks >= ks_even
    if (N, K) == (4096, 11008):
        assert ks == 8 and ks_even == 2
    if K % (128 * ks) == 0 and ks == ks_even:
        pass    # unchanged shape
    out = L.linear(x, w, skinny=True)
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    scale = max(1.0, float(ref.abs().max()))
    assert (out.double().cpu() - ref).abs().max().item() <= 2 * eps * scale
    part = L.linear_splitk(x, w, always=True)
    assert isinstance(part, L.SplitKPartials) and part.k_splits == ks
    slabs = part.slabs[: ks * M * N].view(ks, M, N)
    assert (slabs.sum(0).double().cpu() - ref).abs().max().item() <= 3e-5 * scale
    assert torch.equal(part.materialize(), out)
    # every split did its share: no slab is all zeros unless its chunk of x is
    assert all(float(slabs[i].abs().max()) > 0 for i in range(ks))
