"""GPU parity tests of the row-owned decode projections (csrc/gemm_rows.hip) and of "norm on the fly" (the NF mode of
csrc/gemm_skinny.hip's packed ring kernel), r05.

What they replace: o_proj / down_proj as split-K GEMMs + the residual-add consumer launch (reference
swiftllm/worker/kernels/linear.py:3-12 + rmsnorm.py:67-89 at layers/transformer_layer.py:117-128,46). The bar: the
residual stream keeps the BITS of the split-K + consumer path wherever the summation order allows it to be checked
against it (it rounds the same fp32-accumulated product once), x_scaled built on the fly is bit-identical to the
consumer's, the sums of squares agree to fp32 summation order, and end to end the engine with rows_decode on agrees with
the engine with it off to the storage dtype's rounding, greedy ids identical except on near-ties."""
import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu


def gen(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 7, 16, 17, 32])
@pytest.mark.parametrize("N,K", [(4096, 4096), (4096, 14336), (64, 1024), (96, 3072), (32, 5120), (160, 8192), (128, 2048)])
def test_rows_add_is_one_rounding_of_the_product_plus_residual(dtype, M, N, K):
    """residual += round(x . W^T): against an fp64 product (one rounding of the product, one of the sum: <= 1 ulp of each),
    straight-line schedules (K = 2048 / 4096 / 8192 / 14336) and the run-time ring (3 and 5 chunks), one and two token
    blocks, tiles that are the upper and the lower half of a packed 32-row fragment, rows >= M untouched."""
    from swiftllm_amd import _hip
    from swiftllm_amd.worker.kernels.linear import pack_weight, linear_rows_add, rows_add_ok
    g = gen(M + N + K)
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * 0.03).to(dtype).cuda()
    pack_weight(w)
    res = torch.randn(M + 2, N, generator=g).to(dtype).cuda()       # two guard rows behind the batch
    want_guard = res[M:].clone()
    r = res[:M]
    assert rows_add_ok(x, w, r) and _hip.load().swl_gemm_rows_supported(M, N, K) == 1
    prod = x.double() @ w.double().T
    before = r.clone()
    linear_rows_add(x, w, r)
    torch.cuda.synchronize()
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    # the product is accumulated in fp32 (<= ~1e-6 relative of the row's magnitude), rounded to T, added, rounded again
    p_t = prod.to(dtype).double()
    want = (p_t + before.double())
    tol = eps * want.abs() + eps * prod.abs() + 1e-3 * eps * (K ** 0.5)
    assert ((r.double() - want).abs() <= tol).all()
    assert torch.equal(res[M:], want_guard)


@pytest.mark.parametrize("M", [1, 8, 32])
@pytest.mark.parametrize("N,K", [(4096, 4096), (4096, 14336)])
def test_rows_add_carries_the_bits_of_splitk_plus_consumer(M, N, K):
    """Same residual bits as linear_splitk + add_scale_from_splitk at the Llama-3-8B shapes (measured: identical; the two
    sum the same fp32 partial products in K order, eight runs added in order)."""
    from swiftllm_amd.worker.kernels.linear import pack_weight, linear_rows_add, linear_splitk, SplitKPartials
    from swiftllm_amd.worker.kernels.rmsnorm import add_scale_from_splitk
    dtype = torch.bfloat16
    g = gen(M + K)
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * 0.02).to(dtype).cuda()
    pack_weight(w)
    res = torch.randn(M, N, generator=g).to(dtype).cuda()
    nw = (1 + 0.1 * torch.randn(N, generator=g)).to(dtype).cuda()
    part = linear_splitk(x, w)
    assert isinstance(part, SplitKPartials)
    r_old = res.clone()
    add_scale_from_splitk(part, r_old, nw, 1e-5)
    r_new = res.clone()
    linear_rows_add(x, w, r_new)
    diff = (r_old != r_new).float().mean().item()
    # not a contract (another MFMA shape sums another fp32 order) — but at these shapes the rounded bits coincide
    assert diff <= 1e-3, diff
    assert ((r_old.float() - r_new.float()).abs() <= 2.0 ** -7 * r_old.float().abs() + 1e-6).all()


@pytest.mark.parametrize("M", [1, 3, 8, 16, 32])
def test_norm_on_the_fly_projections(M):
    """linear_splitk_nf / linear_silu_gate_nf on the raw residual rows against add-scale-then-project: the staged
    activations are round(r * w) in both, so the qkv slabs are BIT-identical; the sums of squares (one per K-chunk here,
    one per 1024 columns there) agree to fp32 summation order; the SiLU-gate output differs by the fp32 rounding of 1/rms."""
    from swiftllm_amd.worker.kernels.linear import (pack_weight, linear_splitk, linear_splitk_nf, linear_silu_gate,
                                                    linear_silu_gate_nf, nf_ok)
    from swiftllm_amd.worker.kernels.rmsnorm import RowScalePending
    dtype = torch.bfloat16
    h, inter, nqkv = 4096, 14336, 6144
    g = gen(M + 23)
    r = (2 * torch.randn(M, h, generator=g)).to(dtype).cuda()
    nw = (1 + 0.1 * torch.randn(h, generator=g)).to(dtype).cuda()
    wqkv = (torch.randn(nqkv, h, generator=g) * 0.02).to(dtype).cuda()
    wug = (torch.randn(2 * inter, h, generator=g) * 0.02).to(dtype).cuda()
    pack_weight(wqkv)
    pack_weight(wug)
    assert nf_ok(r, wqkv, nw) and nf_ok(r, wug, nw)
    xs = (r.float() * nw.float()).to(dtype)
    old = linear_splitk(xs, wqkv, always=True)
    old_slabs = old.slabs[: old.k_splits * M * nqkv].clone()
    new, pend = linear_splitk_nf(r, nw, wqkv, 1e-5)
    assert new.k_splits == old.k_splits and pend.ssq.shape == (new.k_splits, M) and pend.hidden == h
    assert torch.equal(new.slabs[: new.k_splits * M * nqkv], old_slabs)
    want = r.double().pow(2).sum(1)
    assert torch.allclose(pend.ssq.double().sum(0), want, rtol=1e-5)
    per_chunk = r.double().pow(2).view(M, new.k_splits, h // new.k_splits).sum(2).T
    assert torch.allclose(pend.ssq.double(), per_chunk, rtol=1e-5)
    ssq8 = r.float().pow(2).view(M, h // 1024, 1024).sum(2).t().contiguous()
    a_old = linear_silu_gate(xs, wug, row_scale=RowScalePending(xs, ssq8, h // 1024, 1e-5))
    a_new = linear_silu_gate_nf(r, nw, 1e-5, wug)
    scale = a_old.float().abs().amax(dim=1, keepdim=True).clamp(min=1e-3)
    assert ((a_old.float() - a_new.float()).abs() / scale).max().item() <= 2.0 ** -7
    assert (a_old != a_new).float().mean().item() <= 1e-2


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [1, 8, 9, 32])
def test_exact_norm_on_the_fly_is_rmsnorm_then_projection(dtype, M):
    """r06c, the reference's rounding points on the row-owned path (float16 included): swl_gemm_rows_add_ssq leaves the bits of
    swl_gemm_rows_add plus per-tile sums of squares of the updated rows (vs fp64: fp32 summation order); the consuming
    projections (_nx) finish the 1/rms and stage round(r * rstd * w) themselves — against the product's own rmsnorm kernel
    followed by the same GEMMs: identical up to the summation order of the sums of squares (a last-bit difference of rstd can
    move an activation by one ulp: a small fraction of the elements, never more than an ulp of the row's scale)."""
    from swiftllm_amd.worker.kernels.linear import (pack_weight, linear_rows_add, linear_silu_gate, linear_splitk,
                                                    linear_silu_gate_nx, linear_splitk_nx, nx_ok)
    from swiftllm_amd.worker.kernels.rmsnorm import rmsnorm_inplace
    H, I, NQKV = 4096, 14336, 6144
    g = gen(40 + M)
    mk = lambda n, k: pack_weight_ret((torch.randn(n, k, generator=g) * 0.02).to(dtype).cuda())
    wo, wug, wqkv = mk(H, H), mk(2 * I, H), mk(NQKV, H)
    nw = (1.0 + 0.1 * torch.randn(H, generator=g)).to(dtype).cuda()
    attn = torch.randn(M, H, generator=g).to(dtype).cuda()
    r0 = torch.randn(M, H, generator=g).to(dtype).cuda()
    eps = 1e-5
    ra, rb = r0.clone(), r0.clone()
    linear_rows_add(attn, wo, ra)
    ssq = linear_rows_add(attn, wo, rb, with_ssq=True)
    torch.cuda.synchronize()
    assert torch.equal(ra, rb)
    want_ssq = (rb.double() ** 2).view(M, H // 16, 16).sum(-1)
    assert ((ssq.double() - want_ssq).abs() <= 1e-5 * want_ssq.abs() + 1e-12).all()
    assert nx_ok(rb, wug, nw) and nx_ok(rb, wqkv, nw)
    xn = rb.clone()
    rmsnorm_inplace(xn, nw, eps)
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    # SiLU-gate
    want = linear_silu_gate(xn, wug)
    got = linear_silu_gate_nx(rb, nw, eps, wug, ssq)
    torch.cuda.synchronize()
    scale = want.float().abs().amax(dim=1, keepdim=True).clamp(min=1e-3)
    assert ((got.float() - want.float()).abs() <= 2 * ulp * scale).all()
    assert (got != want).float().mean().item() <= 2e-2
    # fused qkv slabs
    sk = linear_splitk(xn, wqkv, always=True)
    want_q = sk.slabs[: sk.k_splits * M * NQKV].clone().view(sk.k_splits, M, NQKV).sum(0)
    nx = linear_splitk_nx(rb, nw, wqkv, eps, ssq)
    assert nx is not None and nx.k_splits == sk.k_splits
    got_q = nx.slabs[: nx.k_splits * M * NQKV].view(nx.k_splits, M, NQKV).sum(0)
    torch.cuda.synchronize()
    qs = want_q.abs().amax(dim=1, keepdim=True).clamp(min=1e-3)
    assert ((got_q - want_q).abs() <= 2 * ulp * qs).all()


def pack_weight_ret(w):
    from swiftllm_amd.worker.kernels.linear import pack_weight
    pack_weight(w)
    return w


def _engine_config(path, **kw):
    from swiftllm_amd import EngineConfig
    base = dict(model_path=path, use_dummy=False, block_size=16, gpu_mem_utilization=0.9, num_cpu_blocks=8,
                max_seqs_in_block_table=64, max_blocks_per_seq=32, max_batch_size=32, max_tokens_in_batch=2048)
    base.update(kw)
    return EngineConfig(**base)


@pytest.mark.parametrize("batch,dtype", [(b, "bfloat16") for b in (1, 2, 8, 9, 16, 17, 32)] +
                         [(b, "float16") for b in (1, 9, 17, 32)])
def test_rows_decode_path_equals_the_consumer_path(tmp_path, batch, dtype, monkeypatch):
    """The rows_decode switch at Llama-3-8B layer geometry (3 layers): the engine with row-owned o_proj / down_proj + norm on
    the fly against the same engine with the split-K + consumer launches (and the <= 2-sequence tiny path), teacher-forced over
    6 decode steps, with hipGraph replay and with eager launches: logits within the storage dtype's rounding (the arithmetic
    differences are the fp32 summation order of the sums of squares and of the row-owned products), greedy ids identical
    except on near-ties. Batches on both sides of the x-through-LDS switch of csrc/gemm_rows.hip (8 | 9), on both sides of
    ROWS_DOWN_MAX_M (16 | 17) and at ROWS_O_MAX_M (32)."""
    cfg = synth.make_config(num_hidden_layers=3, hidden_size=4096, num_attention_heads=32, num_key_value_heads=8,
                            intermediate_size=14336, vocab_size=4096, max_position_embeddings=2048, rope_theta=500000.0)
    sd = synth.make_state_dict(cfg, seed=19, dtype=getattr(torch, dtype))
    g = torch.Generator().manual_seed(5)
    lens = ([300, 17, 1, 64, 129, 40, 33, 250] * 4)[:batch]
    prompts = [torch.randint(0, cfg["vocab_size"], (n,), generator=g).tolist() for n in lens]
    synth.write_model_dir(str(tmp_path), cfg, sd)
    del sd
    from swiftllm_amd import LlamaModel

    def run(opts, forced=None):
        model = LlamaModel(_engine_config(str(tmp_path), dtype=dtype, **opts))
        model.load_weights()
        model.init_kvcache_and_swap(32 * 24)
        model.post_layer.logits_tap = []
        seq_ids = list(range(batch))
        toks = [model.forward(prompts, seq_ids, [])]
        logits = []
        cur = list(lens)
        for step in range(6):
            cur = [n + 1 for n in cur]
            feed = forced[step] if forced is not None else toks[-1]
            toks.append(model.forward([[t] for t in feed], seq_ids, list(cur)))
            logits.append(model.post_layer.logits_tap[-1].float().cpu())
        del model
        torch.cuda.empty_cache()
        return toks, logits

    from swiftllm_amd import _hip
    calls, real_call = [], _hip.call

    def spy(name, *args):
        calls.append(name)
        return real_call(name, *args)
    monkeypatch.setattr(_hip, "call", spy)
    ref_toks, ref_logits = run(dict(tuning=dict(rows_decode=False)))
    assert not any(c.startswith("swl_gemm_rows_add") for c in calls)
    calls.clear()
    eps = 2.0 ** -7 if dtype == "bfloat16" else 2.0 ** -10     # (float16: the exact norm on the fly, r06c)
    for opts in (dict(), dict(use_hip_graph=False)):
        toks, logits = run(opts, forced=ref_toks)
        # the path under test really ran: bfloat16 the deferred norm on the fly, float16 the exact one
        assert ("swl_gemm_rows_add_ssq" if dtype == "float16" else "swl_gemm_rows_add") in calls, sorted(set(calls))
        assert ("swl_gemm_skinny_packed_silu_gate_nx" if dtype == "float16" else "swl_gemm_skinny_packed_silu_gate_nf") in calls
        assert ("swl_gemm_skinny_packed_partial_nx" in calls) == (dtype == "float16" and batch <= 16), sorted(set(calls))
        for step, (a, b) in enumerate(zip(logits, ref_logits)):
            scale = b.abs().amax(dim=1, keepdim=True).clamp(min=1.0)
            assert ((a - b).abs() <= 4 * eps * scale).all(), (opts, step, ((a - b).abs() / scale).max().item())
        for step, (x, y) in enumerate(zip(toks, ref_toks)):
            for seq, (tx, ty) in enumerate(zip(x, y)):
                if tx != ty:
                    top2 = ref_logits[step - 1][seq].topk(2).values if step else None
                    assert top2 is not None and float(top2[0] - top2[1]) <= 8 * eps * float(top2[0].abs().clamp(min=1.0))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("H,KVH,D", [(32, 8, 128), (4, 1, 64), (8, 8, 128)])
@pytest.mark.parametrize("n_dec", [0, 3])
def test_fused_prefill_rotary_store_equals_the_two_kernels(dtype, H, KVH, D, n_dec):
    """rotary_embedding_and_store_kvcache_prefill (one pass over k: csrc/kvcache.hip rotary_store_prefill_kernel, plus the
    fused decode launch for riding decodes) against rotary_embedding_inplace + store_kvcache on the same inputs: q, k and
    both pools bit for bit — ragged prompts (1 token, block boundaries, a partial last block), untouched pool blocks stay
    untouched."""
    import types
    from swiftllm_amd.worker.kernels.rotary_emb import rotary_embedding_inplace, rotary_embedding_and_store_kvcache_prefill
    from swiftllm_amd.worker.kernels.kvcache_mgmt import store_kvcache
    g = gen(H + D + n_dec)
    plens = [1, 16, 17, 40, 32]
    dlens = [5, 33, 16][:n_dec]                       # lengths INCLUDING the new token
    L, bs, layer, max_blocks = 2, 16, 1, 8
    n_seqs = len(plens) + n_dec
    T = sum(plens) + n_dec
    need = [-(-n // bs) for n in plens + dlens]
    num_blocks = sum(need) + 3
    perm = torch.randperm(num_blocks, generator=g).tolist()
    bt = torch.zeros(n_seqs + 2, max_blocks, dtype=torch.int32)
    seq_ids = list(range(1, n_seqs + 1))              # (row 0 of the table unused)
    it = iter(perm)
    for sid, nb in zip(seq_ids, need):
        for j in range(nb):
            bt[sid, j] = next(it)
    pos = torch.cat([torch.arange(n) for n in plens] + [torch.tensor([n - 1]) for n in dlens]).to(torch.int32)
    table_rows = 64
    cos = torch.randn(table_rows, D // 2, generator=g).to(dtype).cuda()
    sin = torch.randn(table_rows, D // 2, generator=g).to(dtype).cuda()
    q0 = torch.randn(T, H, D, generator=g).to(dtype).cuda()
    k0 = torch.randn(T, KVH, D, generator=g).to(dtype).cuda()
    v0 = torch.randn(T, KVH, D, generator=g).to(dtype).cuda()
    pool0 = torch.randn(num_blocks, L, KVH, bs, D, generator=g).to(dtype).cuda()
    pl = torch.tensor(plens, dtype=torch.int32)
    st = types.SimpleNamespace(
        seq_ids=torch.tensor(seq_ids, dtype=torch.int32).cuda(), num_prefill_seqs=len(plens), num_prefill_tokens=sum(plens),
        prefill_seq_start_locs=(torch.cumsum(pl, 0, dtype=torch.int32) - pl).cuda(), prefill_seq_lens=pl.cuda(),
        max_prefill_len=max(plens), num_decoding_seqs=n_dec,
        decoding_seq_lens=torch.tensor(dlens, dtype=torch.int32).cuda(), position_cos=cos, position_sin=sin,
        position_indices=pos.cuda())
    mc = types.SimpleNamespace(num_layers=L, num_kv_heads=KVH, head_dim=D)
    ec = types.SimpleNamespace(block_size=bs)
    btd = bt.cuda()
    q1, k1, v1, kc1, vc1 = q0.clone(), k0.clone(), v0.clone(), pool0.clone(), pool0.flip(0).clone()
    rotary_embedding_inplace(q1, k1, st)
    store_kvcache(k1, v1, kc1, vc1, btd, mc, ec, st, layer)
    q2, k2, v2, kc2, vc2 = q0.clone(), k0.clone(), v0.clone(), pool0.clone(), pool0.flip(0).clone()
    rotary_embedding_and_store_kvcache_prefill(q2, k2, v2, kc2, vc2, btd, mc, ec, st, layer)
    torch.cuda.synchronize()
    assert torch.equal(q1, q2) and torch.equal(k1, k2) and torch.equal(v1, v2)
    assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2)
    assert not torch.equal(kc2, pool0)                # (something was stored)
