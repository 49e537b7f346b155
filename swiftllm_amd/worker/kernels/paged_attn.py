"""Decode-stage paged attention operator. Reference: swiftllm/worker/kernels/paged_attn.py:152-222."""
import torch

from swiftllm_amd import _hip
from ._layout import token_stride


def paged_attention(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                    block_table: torch.Tensor, model_config, engine_config, infer_state,
                    cur_layer: int, o: torch.Tensor):
    """Flash-decoding over the paged KV pool for the decoding sequences of this batch.

    q, o: [num_decoding_seqs, num_q_heads, head_dim] (o may be given as [seqs, hidden]).
    `infer_state.seq_block_size / num_seq_blocks` pick the split-K width exactly as in the
    reference; the fp32 partials use the reference's shapes and format. A preallocated scratch
    (`infer_state.paged_attn_scratch`) is used when present, else one is taken from torch's caching
    allocator as the reference does (paged_attn.py:170-180).
    """
    _hip.require_gpu_tensor(q, "q")
    nd = infer_state.num_decoding_seqs
    if nd == 0:
        return
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and block_table.is_contiguous()
    assert infer_state.seq_block_size % engine_config.block_size == 0
    assert q.dtype == k_cache.dtype == v_cache.dtype == o.dtype
    if o.dim() == 2:
        o = o.view(o.shape[0], model_config.num_q_heads, model_config.head_dim)
    nsb = infer_state.num_seq_blocks
    scratch = None
    if nsb > 1:
        need = _hip.scratch_bytes(nd, model_config.num_q_heads, model_config.head_dim, nsb)
        scratch = getattr(infer_state, "paged_attn_scratch", None)
        if scratch is None or scratch.numel() * scratch.element_size() < need:
            scratch = torch.empty(need // 4, dtype=torch.float32, device=q.device)
    _hip.call("swl_paged_attn_decode", _hip.ptr(o), _hip.ptr(q), _hip.ptr(k_cache),
              _hip.ptr(v_cache), _hip.ptr(block_table),
              _hip.ptr(infer_state.seq_ids[infer_state.num_prefill_seqs:]),
              _hip.ptr(infer_state.decoding_seq_lens), _hip.ptr(scratch), infer_state.softmax_scale,
              nd, model_config.num_q_heads, model_config.num_kv_heads, model_config.head_dim,
              model_config.num_layers, engine_config.block_size, cur_layer, block_table.shape[1],
              infer_state.seq_block_size, nsb, token_stride(q, "q"), token_stride(o, "o"),
              _hip.dtype_code(q.dtype), _hip.stream())


def paged_attention_from_qkv_splitk(partials, k_cache: torch.Tensor, v_cache: torch.Tensor,
                                    block_table: torch.Tensor, model_config, engine_config, infer_state,
                                    cur_layer: int, o: torch.Tensor, row_scale=None, merge: bool = True):
    """Pure-decode batches: rotary + KV-store + paged attention in one launch, fed by the split-K partial slabs
    of the fused qkv projection (kernels/linear.py: SplitKPartials). Same bits as
    rotary_embedding_and_store_kvcache_decode_from_splitk followed by paged_attention; q/k/v never exist as
    tensors (the rotated k and the v go to the pools, q stays in the kernel)."""
    nd = infer_state.num_decoding_seqs
    if nd == 0:
        return
    h, kvh, d = model_config.num_q_heads, model_config.num_kv_heads, model_config.head_dim
    assert infer_state.num_prefill_seqs == 0 and infer_state.position_indices is not None
    nsb = infer_state.num_seq_blocks
    assert o is not None or (not merge and nsb > 1), "no output tensor: only for split sequences with merge=False"
    assert partials.shape == (nd, (h + 2 * kvh) * d) and partials.dtype == k_cache.dtype
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and block_table.is_contiguous()
    o_stride, o_dtype = h * d, k_cache.dtype
    if o is not None:
        assert o.dtype == k_cache.dtype
        if o.dim() == 2:
            o = o.view(o.shape[0], h, d)
        o_stride = token_stride(o, "o")
    scratch = None
    if nsb > 1:
        need = _hip.scratch_bytes(nd, h, d, nsb)
        scratch = getattr(infer_state, "paged_attn_scratch", None)
        if scratch is None or scratch.numel() * scratch.element_size() < need:
            scratch = torch.empty(need // 4, dtype=torch.float32, device=k_cache.device)
    if row_scale is not None:
        # the qkv projection ran on activations whose RMSNorm scale is pending (kernels/rmsnorm.py: RowScalePending):
        # the prologue applies 1/rms to the fp32 slab sums before rounding them
        assert partials.k_splits in (1, 2, 4) and row_scale.ssq.shape == (row_scale.parts, nd)
        # merge=False (split sequences only): stop after phase 1 and hand the partials to a consumer that merges them
        # itself (kernels/linear.py: linear_splitk_from_attn_partials); returns the scratch holding them
        entry = "swl_paged_attn_decode_qkv_rs" if merge or nsb == 1 else "swl_paged_attn_decode_qkv_rs_partials"
        _hip.call(entry, _hip.ptr(o), _hip.ptr(partials.slabs), partials.k_splits,
                  _hip.ptr(row_scale.ssq), row_scale.parts, row_scale.hidden, row_scale.eps,
                  _hip.ptr(infer_state.position_cos), _hip.ptr(infer_state.position_sin),
                  _hip.ptr(infer_state.position_indices), _hip.ptr(k_cache), _hip.ptr(v_cache), _hip.ptr(block_table),
                  _hip.ptr(infer_state.seq_ids), _hip.ptr(infer_state.decoding_seq_lens), _hip.ptr(scratch),
                  infer_state.softmax_scale, nd, h, kvh, d, model_config.num_layers, engine_config.block_size,
                  cur_layer, block_table.shape[1], infer_state.seq_block_size, nsb, o_stride,
                  _hip.dtype_code(o_dtype), _hip.stream())
        return scratch if not merge and nsb > 1 else None
    assert merge, "phase-1-only attention needs the deferred-norm entry point"
    _hip.call("swl_paged_attn_decode_qkv", _hip.ptr(o), _hip.ptr(partials.slabs), partials.k_splits,
              _hip.ptr(infer_state.position_cos), _hip.ptr(infer_state.position_sin),
              _hip.ptr(infer_state.position_indices), _hip.ptr(k_cache), _hip.ptr(v_cache), _hip.ptr(block_table),
              _hip.ptr(infer_state.seq_ids), _hip.ptr(infer_state.decoding_seq_lens), _hip.ptr(scratch),
              infer_state.softmax_scale, nd, h, kvh, d, model_config.num_layers, engine_config.block_size,
              cur_layer, block_table.shape[1], infer_state.seq_block_size, nsb, token_stride(o, "o"),
              _hip.dtype_code(o.dtype), _hip.stream())
