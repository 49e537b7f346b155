"""oracle/eager_ops.py — CPU restatement of the reference's operator layer (TEST INFRASTRUCTURE).

This file is the checker, never the product: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it. Nothing under swiftllm_amd/ does.

Each function restates, in plain PyTorch on CPU tensors, what one operator of
swiftllm/worker/kernels/ computes, following the reference's rounding points (the file:line of the
code each function follows is cited in its docstring). The storage dtype is whatever the input
tensors carry (float16 = the reference's precision; bfloat16 = the MI355X headline precision with the
same rounding points).

Pinning: the reference ships no tests or golden vectors, so this oracle is pinned
against the reference ITSELF: oracle/gen_golden.py imports /root/reference, executes its Triton
kernels through Triton's CPU interpreter and freezes their outputs under tests/golden/;
tests/test_oracle_golden.py holds this file to those vectors.
"""
import math

import torch

LOG2E = 1.442695040888963   # the constant the reference multiplies softmax_scale by (paged_attn.py:195)


# ---- linear ---------------------------------------------------------------------------------------------
def linear(a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """kernels/linear.py:3-12 — F.linear, fp32 accumulation inside the BLAS, one rounding."""
    return torch.nn.functional.linear(a.float(), w.float()).to(a.dtype)


# ---- rmsnorm --------------------------------------------------------------------------------------------
def rmsnorm_inplace(x: torch.Tensor, weight: torch.Tensor, eps: float):
    """kernels/rmsnorm.py:5-24 — fp32: x * (1/sqrt(mean(x^2)+eps)) * w, rounded once."""
    xf = x.float()
    var = (xf * xf).sum(dim=-1, keepdim=True) / x.shape[-1]
    rstd = 1.0 / torch.sqrt(var + eps)
    x.copy_((xf * rstd * weight.float()).to(x.dtype))


def fused_add_rmsnorm_inplace(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float):
    """kernels/rmsnorm.py:39-65 — the add happens in the storage dtype and is stored to `residual`
    (:54-57); the norm is then taken over the ROUNDED sum in fp32."""
    s = (x.float() + residual.float()).to(x.dtype)      # == x + r evaluated in the storage dtype
    residual.copy_(s)
    sf = s.float()
    var = (sf * sf).sum(dim=-1, keepdim=True) / x.shape[-1]
    rstd = 1.0 / torch.sqrt(var + eps)
    x.copy_((sf * rstd * weight.float()).to(x.dtype))


# ---- rotary ---------------------------------------------------------------------------------------------
def _round(t: torch.Tensor, dtype) -> torch.Tensor:
    return t.to(dtype).float()


def rotary_embedding_inplace(q: torch.Tensor, k: torch.Tensor, infer_state):
    """kernels/rotary_emb.py:26-42 — rotate-half; q0*cos - q1*sin and q0*sin + q1*cos with EVERY
    product and sum rounded to the storage dtype (the Triton kernel computes on fp16 tensors)."""
    cos, sin = infer_state.position_cos, infer_state.position_sin
    idx = getattr(infer_state, "position_indices", None)
    if idx is not None:
        cos, sin = cos[idx.long()], sin[idx.long()]
    dt = q.dtype
    half = q.shape[-1] // 2
    c = cos.float()[:, None, :]
    s = sin.float()[:, None, :]
    for x in (q, k):
        x0, x1 = x[..., :half].float(), x[..., half:].float()
        r0 = _round(_round(x0 * c, dt) - _round(x1 * s, dt), dt)
        r1 = _round(_round(x0 * s, dt) + _round(x1 * c, dt), dt)
        x[..., :half] = r0.to(dt)
        x[..., half:] = r1.to(dt)


# ---- KV store -------------------------------------------------------------------------------------------
def store_kvcache(k, v, k_cache, v_cache, block_table, model_config, engine_config, infer_state,
                  cur_layer: int):
    """kernels/kvcache_mgmt.py:36-48 (prefill tiles), :68-79 (decode token) and the eager restatement
    in the reference's own comments (:124-132). Bit-exact copies.
    pool layout [num_blocks, L, KVH, block_size, D]."""
    bs = engine_config.block_size
    seq_ids = infer_state.seq_ids.tolist()
    starts = infer_state.prefill_seq_start_locs.tolist()
    plens = infer_state.prefill_seq_lens.tolist()
    for i in range(infer_state.num_prefill_seqs):
        sid, st, ln = seq_ids[i], starts[i], plens[i]
        for j in range((ln + bs - 1) // bs):
            n = min(bs, ln - j * bs)
            blk = int(block_table[sid, j])
            rows = slice(st + j * bs, st + j * bs + n)
            k_cache[blk, cur_layer, :, :n, :] = k[rows].transpose(0, 1)
            v_cache[blk, cur_layer, :, :n, :] = v[rows].transpose(0, 1)
    p = infer_state.num_prefill_tokens
    dlens = infer_state.decoding_seq_lens.tolist()
    for i in range(infer_state.num_decoding_seqs):
        sid = seq_ids[infer_state.num_prefill_seqs + i]
        pos = dlens[i] - 1
        blk = int(block_table[sid, pos // bs])
        k_cache[blk, cur_layer, :, pos % bs, :] = k[p + i]
        v_cache[blk, cur_layer, :, pos % bs, :] = v[p + i]


# ---- silu * mul -----------------------------------------------------------------------------------------
def silu_and_mul_inplace(x: torch.Tensor):
    """kernels/silu_and_mul.py:16-23 — gate -> fp32, g/(1+exp(-g)), rounded to the storage dtype;
    then up*gate in the storage dtype. up = first half, gate = second half (weight.py:133)."""
    inter = x.shape[1] // 2
    g = x[:, inter:].float()
    act = (g / (1.0 + torch.exp(-g))).to(x.dtype)
    x[:, :inter] = (x[:, :inter].float() * act.float()).to(x.dtype)


# ---- paged attention (decode) ---------------------------------------------------------------------------
def _gather_kv(cache, block_table, seq_id: int, length: int, layer: int, bs: int):
    nblk = (length + bs - 1) // bs
    blocks = block_table[seq_id, :nblk].long()
    t = cache[blocks, layer]                        # [nblk, KVH, bs, D]
    return t.permute(1, 0, 2, 3).reshape(t.shape[1], nblk * bs, t.shape[3])[:, :length]


def paged_attention_phase1(q, k_cache, v_cache, block_table, model_config, engine_config,
                           infer_state, cur_layer: int, score_dtype: str = "fp32"):
    """kernels/paged_attn.py:45-108 — per (sequence, q-head, seq-block) online softmax walking the
    16-token blocks in order; returns (mid_o [Bd,H,nsb,D] fp32 NORMALISED, mid_lse [Bd,H,nsb] fp32 in
    the base-2 domain), entries of seq-blocks past a sequence's end are left at 0 / -inf.
    score_dtype "fp32": exact scores (what the reference's commented eager code computes,
    :224-259); "ref": the Triton kernel's rounding — q*k products and their sum in the storage dtype,
    scale*log2e rounded to fp16 (:17, :72-73)."""
    bs = engine_config.block_size
    H, KVH, D = model_config.num_q_heads, model_config.num_kv_heads, model_config.head_dim
    G = H // KVH
    nd, nsb, sbs = infer_state.num_decoding_seqs, infer_state.num_seq_blocks, infer_state.seq_block_size
    mid_o = torch.zeros(nd, H, nsb, D, dtype=torch.float32)
    mid_lse = torch.full((nd, H, nsb), float("-inf"), dtype=torch.float32)
    seq_ids = infer_state.seq_ids[infer_state.num_prefill_seqs:].tolist()
    lens = infer_state.decoding_seq_lens.tolist()
    scale = infer_state.softmax_scale * LOG2E
    if score_dtype == "ref":
        scale = float(torch.tensor(scale, dtype=torch.float16))
    for b in range(nd):
        K = _gather_kv(k_cache, block_table, seq_ids[b], lens[b], cur_layer, bs)    # [KVH, len, D]
        V = _gather_kv(v_cache, block_table, seq_ids[b], lens[b], cur_layer, bs)
        K = K.repeat_interleave(G, dim=0)
        V = V.repeat_interleave(G, dim=0).float()
        if score_dtype == "ref":
            prod = (q[b][:, None, :].float() * K.float()).to(q.dtype)      # fp16 products
            score = prod.sum(dim=-1, dtype=q.dtype)                        # fp16 reduction
            score = (score.float() * scale).to(q.dtype).float()
        else:
            score = torch.einsum("hd,hld->hl", q[b].float(), K.float()) * scale
        for sb in range((lens[b] + sbs - 1) // sbs):
            lo, hi = sb * sbs, min(lens[b], (sb + 1) * sbs)
            m = torch.full((H,), -1e20)
            se = torch.zeros(H)
            acc = torch.zeros(H, D)
            for t0 in range(lo, hi, bs):
                t1 = min(hi, t0 + bs)
                s_blk = score[:, t0:t1]
                m_new = torch.maximum(m, s_blk.max(dim=1).values)
                p = torch.exp2(s_blk - m_new[:, None])
                alpha = torch.exp2(m - m_new)
                acc = acc * alpha[:, None] + torch.einsum("hl,hld->hd", p, V[:, t0:t1])
                se = se * alpha + p.sum(dim=1)
                m = m_new
            mid_o[b, :, sb] = acc / se[:, None]
            mid_lse[b, :, sb] = torch.log2(se) + m
    return mid_o, mid_lse


def paged_attention_phase2(mid_o, mid_lse, infer_state, o: torch.Tensor):
    """kernels/paged_attn.py:128-149 — LSE-weighted merge of the partials, rounded to o's dtype."""
    sbs = infer_state.seq_block_size
    lens = infer_state.decoding_seq_lens.tolist()
    for b in range(infer_state.num_decoding_seqs):
        n = (lens[b] + sbs - 1) // sbs
        lse = mid_lse[b, :, :n]
        m = lse.max(dim=1).values
        w = torch.exp2(lse - m[:, None])
        out = (w[:, :, None] * mid_o[b, :, :n]).sum(dim=1) / w.sum(dim=1)[:, None]
        o[b] = out.to(o.dtype).reshape(o[b].shape)


def paged_attention(q, k_cache, v_cache, block_table, model_config, engine_config, infer_state,
                    cur_layer: int, o: torch.Tensor, score_dtype: str = "fp32"):
    """kernels/paged_attn.py:152-222 — phase 1 + phase 2."""
    if infer_state.num_decoding_seqs == 0:
        return
    mid_o, mid_lse = paged_attention_phase1(q, k_cache, v_cache, block_table, model_config,
                                            engine_config, infer_state, cur_layer, score_dtype)
    paged_attention_phase2(mid_o, mid_lse, infer_state, o)


def paged_attention_dense(q, k_cache, v_cache, block_table, model_config, engine_config, infer_state,
                          cur_layer: int, o: torch.Tensor):
    """The value paged_attention() computes — softmax(q.K^T * scale) V over each sequence's cached tokens, fp32, one
    rounding — evaluated as ONE dense softmax per sequence instead of the reference kernel's walk over 16-token blocks and
    sequence blocks (kernels/paged_attn.py:45-149). Same result up to fp32 reassociation (tests/test_oracle_golden.py holds
    it to paged_attention); ~20x faster on a CPU because nothing loops in Python per block. Used where the oracle is
    TIMED (bench.py's cpu_baseline leg): a baseline should measure the host's arithmetic, not the interpreter."""
    bs = engine_config.block_size
    G = model_config.num_q_heads // model_config.num_kv_heads
    seq_ids = infer_state.seq_ids[infer_state.num_prefill_seqs:].tolist()
    lens = infer_state.decoding_seq_lens.tolist()
    for b in range(infer_state.num_decoding_seqs):
        K = _gather_kv(k_cache, block_table, seq_ids[b], lens[b], cur_layer, bs).float()    # [KVH, len, D]
        V = _gather_kv(v_cache, block_table, seq_ids[b], lens[b], cur_layer, bs).float()
        qh = q[b].float().view(K.shape[0], G, -1)                                           # [KVH, G, D]
        p = torch.softmax(torch.einsum("kgd,kld->kgl", qh, K) * infer_state.softmax_scale, dim=-1)
        o[b] = torch.einsum("kgl,kld->kgd", p, V).reshape(o[b].shape).to(o.dtype)


# ---- prefill attention ----------------------------------------------------------------------------------
def prefill_attention(q, k, v, o, model_config, engine_config, infer_state):
    """kernels/prefill_attn.py:45-100 (and the vllm_flash_attn call it stands in for,
    transformer_layer.py:83-96) — per sequence causal softmax(QK^T*scale)V with GQA; scores and
    accumulation in fp32; P is rounded to the storage dtype before the PV product (:71), the result
    rounded once at the store. (Exact softmax instead of the tiled online form: the same value up to
    fp32 reassociation.)"""
    H, KVH = model_config.num_q_heads, model_config.num_kv_heads
    G = H // KVH
    starts = infer_state.prefill_seq_start_locs_with_end.tolist()
    ov = o.view(o.shape[0], H, -1) if o.dim() == 2 else o
    for i in range(infer_state.num_prefill_seqs):
        s, e = starts[i], starts[i + 1]
        n = e - s
        if n == 0:
            continue
        ki = k[s:e].float().transpose(0, 1).repeat_interleave(G, dim=0)      # [H, n, D]
        vi = v[s:e].float().transpose(0, 1).repeat_interleave(G, dim=0)
        # query rows are independent: long prompts go in blocks of rows so that the [H, rows, keys] score tensor stays
        # small (16k tokens x 8 heads would be 8.6 GB per temporary); each row still sees one exact softmax over its keys
        rows = n if n <= 4096 else 1024
        for r0 in range(0, n, rows):
            r1 = min(n, r0 + rows)
            qi = q[s + r0:s + r1].float().transpose(0, 1)                       # [H, rows, D]
            score = torch.matmul(qi, ki[:, :r1].transpose(1, 2)) * (infer_state.softmax_scale * LOG2E)
            mask = torch.ones(r1 - r0, r1, dtype=torch.bool).tril(diagonal=r0)
            score = torch.where(mask, score, torch.tensor(-1e20))
            m = score.max(dim=-1, keepdim=True).values
            p = torch.exp2(score - m)
            l = p.sum(dim=-1, keepdim=True)
            out = torch.matmul(p.to(q.dtype).float(), vi[:, :r1]) / l
            ov[s + r0:s + r1] = out.transpose(0, 1).to(o.dtype)


# ---- block table ----------------------------------------------------------------------------------------
def set_block_table_and_num_seq_alloc_blocks(num_seq_allocated_blocks, block_table,
                                             candidate_blocks, seq_ids, block_needed):
    """kernels/block_mgmt.py:33-40 (docstring spec) / :5-24."""
    off = 0
    for i, s in enumerate(seq_ids.tolist()):
        n = int(block_needed[i])
        have = int(num_seq_allocated_blocks[s])
        block_table[s, have:have + n] = candidate_blocks[off:off + n].to(block_table.dtype)
        num_seq_allocated_blocks[s] = have + n
        off += n


def unset_block_table_and_num_seq_alloc_blocks(num_seq_allocated_blocks, block_table, seq_ids,
                                               is_block_free):
    """kernels/block_mgmt.py:72-75 / :49-64."""
    for s in seq_ids.tolist():
        n = int(num_seq_allocated_blocks[s])
        is_block_free[block_table[s, :n].long()] = True
        num_seq_allocated_blocks[s] = 0


def gather_allocated_blocks_and_unset(num_seq_allocated_blocks, block_table, seq_ids, is_block_free):
    """kernels/block_mgmt.py:112-114 / :83-104."""
    out = []
    for s in seq_ids.tolist():
        n = int(num_seq_allocated_blocks[s])
        ids = block_table[s, :n]
        out.append(ids.clone())
        is_block_free[ids.long()] = True
        num_seq_allocated_blocks[s] = 0
    if not out:
        return torch.empty((0,), dtype=torch.int32)
    return torch.cat(out).to(torch.int32)


# ---- swap -----------------------------------------------------------------------------------------------
def swap_blocks(source_block_ids, target_block_ids, is_swap_in, k_cache, v_cache, k_swap, v_swap):
    """csrc/src/block_swapping.cpp:22-85 — block-wise copies (the run-length coalescing there is a
    transport optimisation with no effect on the result)."""
    for s, t in zip(source_block_ids, target_block_ids):
        if is_swap_in:
            k_cache[t] = k_swap[s]
            v_cache[t] = v_swap[s]
        else:
            k_swap[t] = k_cache[s]
            v_swap[t] = v_cache[s]


# ---- rope tables ----------------------------------------------------------------------------------------
def rope_tables(model_config, dtype: torch.dtype):
    """worker/model.py:177-225 — cos/sin [positions, D/2]; scalar scaling = linear interpolation,
    dict scaling = the reference's own low/high frequency split (not the HuggingFace formula)."""
    base, dim = model_config.rope_theta, model_config.head_dim
    scaling = model_config.rope_scaling
    f32 = torch.float32
    if isinstance(scaling, dict):
        factor = scaling.get("factor", 4.0)
        low = scaling.get("low_freq_factor", 1.0)
        high = scaling.get("high_freq_factor", 1.0)
        orig = scaling.get("original_max_position_embeddings", model_config.max_position_embeddings)
        t = torch.arange(int(orig * factor) + 128, dtype=f32)
        split = int((dim // 2) * low / (low + high))
        inv_low = 1.0 / (base ** (torch.arange(0, split * 2, 2, dtype=f32) / dim))
        inv_high = 1.0 / (base ** (torch.arange(split * 2, dim, 2, dtype=f32) / dim))
        freqs = torch.cat([torch.outer(t / low, inv_low), torch.outer(t / high, inv_high)], dim=-1)
    else:
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=f32) / dim))
        t = torch.arange(model_config.max_position_embeddings * scaling + 128, dtype=f32) / scaling
        freqs = torch.outer(t, inv_freq)
    return torch.cos(freqs).to(dtype), torch.sin(freqs).to(dtype)


def select_seq_block_size(decoding_seq_lens, num_kv_heads: int) -> int:
    """worker/model.py:305-324 — the reference's split-K width heuristic, verbatim arithmetic."""
    size = 2048
    total = sum(decoding_seq_lens)
    longest = max(decoding_seq_lens) if decoding_seq_lens else 0
    while num_kv_heads * (total / size) < 1024 and size // 2 >= 64 and longest / (size // 2) <= 128:
        size //= 2
    return size
