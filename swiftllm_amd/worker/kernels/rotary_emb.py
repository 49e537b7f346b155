"""Rotary embedding operator. Reference: swiftllm/worker/kernels/rotary_emb.py:44-58."""
import torch

from swiftllm_amd import _hip
from ._layout import token_stride


def rotary_embedding_inplace(q: torch.Tensor, k: torch.Tensor, infer_state):
    """Rotate-half RoPE on q[T, H, D] and k[T, KVH, D], in place.

    `infer_state.position_cos/sin` are either the per-token rows the reference gathers
    (model.py:350-351; `position_indices is None`) or the model's whole rope cache together with
    `infer_state.position_indices` (int32 [T]) — the kernel then does the row lookup itself and the
    two gather launches + [T, D/2] temporaries disappear.
    """
    _hip.require_gpu_tensor(q, "q")
    num_tokens = q.shape[0]
    if num_tokens == 0:
        return
    cos, sin = infer_state.position_cos, infer_state.position_sin
    pos_idx = getattr(infer_state, "position_indices", None)
    assert cos.is_contiguous() and sin.is_contiguous() and cos.dtype == q.dtype == k.dtype
    assert cos.shape[-1] * 2 == q.shape[2] == k.shape[2]
    if pos_idx is None:
        assert cos.shape[0] == num_tokens
    else:
        assert pos_idx.dtype == torch.int32 and pos_idx.is_contiguous() and pos_idx.numel() == num_tokens
    _hip.call("swl_rotary", _hip.ptr(q), _hip.ptr(k), _hip.ptr(cos), _hip.ptr(sin),
              _hip.ptr(pos_idx), num_tokens, q.shape[1], k.shape[1], q.shape[2],
              token_stride(q, "q"), token_stride(k, "k"), _hip.dtype_code(q.dtype), _hip.stream())


def rotary_embedding_and_store_kvcache_decode(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                                              k_cache: torch.Tensor, v_cache: torch.Tensor,
                                              block_table: torch.Tensor, model_config,
                                              engine_config, infer_state, cur_layer: int):
    """Pure-decode batches only: rotary on q,k and the KV store of the rotated k and of v in ONE
    launch (= rotary_embedding_inplace + store_kvcache of the reference's
    transformer_layer.py:62-77, same results). Needs `infer_state.position_indices`."""
    _hip.require_gpu_tensor(q, "q")
    assert infer_state.num_prefill_seqs == 0 and infer_state.position_indices is not None
    nd = infer_state.num_decoding_seqs
    if nd == 0:
        return
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and block_table.is_contiguous()
    _hip.call("swl_rotary_store_kv_decode", _hip.ptr(q), _hip.ptr(k), _hip.ptr(v),
              _hip.ptr(infer_state.position_cos), _hip.ptr(infer_state.position_sin),
              _hip.ptr(infer_state.position_indices), _hip.ptr(k_cache), _hip.ptr(v_cache),
              _hip.ptr(block_table), _hip.ptr(infer_state.seq_ids),
              _hip.ptr(infer_state.decoding_seq_lens), nd, q.shape[1], k.shape[1], q.shape[2],
              cur_layer, model_config.num_layers, engine_config.block_size, block_table.shape[1],
              token_stride(q, "q"), token_stride(k, "k"), token_stride(v, "v"),
              _hip.dtype_code(q.dtype), _hip.stream())


def rotary_embedding_and_store_kvcache_prefill(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                                               k_cache: torch.Tensor, v_cache: torch.Tensor,
                                               block_table: torch.Tensor, model_config,
                                               engine_config, infer_state, cur_layer: int):
    """Batches with prefill sequences: rotary on q, k and the KV store in ONE pass over k for the prompt tokens
    (csrc/kvcache.hip: rotary_store_prefill_kernel) — and, when decoding sequences ride along (SARATHI), the fused decode
    launch for their tokens. = rotary_embedding_inplace + store_kvcache of the reference's transformer_layer.py:62-77,
    bit-identical. Needs `infer_state.position_indices`."""
    _hip.require_gpu_tensor(q, "q")
    st = infer_state
    assert st.num_prefill_seqs > 0 and st.position_indices is not None
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and block_table.is_contiguous()
    qs, ks, vs = token_stride(q, "q"), token_stride(k, "k"), token_stride(v, "v")
    code, stream = _hip.dtype_code(q.dtype), _hip.stream()
    _hip.call("swl_rotary_store_kv_prefill", _hip.ptr(q), _hip.ptr(k), _hip.ptr(v), _hip.ptr(st.position_cos),
              _hip.ptr(st.position_sin), _hip.ptr(st.position_indices), _hip.ptr(k_cache), _hip.ptr(v_cache),
              _hip.ptr(block_table), _hip.ptr(st.seq_ids), _hip.ptr(st.prefill_seq_start_locs),
              _hip.ptr(st.prefill_seq_lens), st.num_prefill_seqs, st.max_prefill_len, cur_layer, model_config.num_layers,
              q.shape[1], k.shape[1], engine_config.block_size, q.shape[2], block_table.shape[1], qs, ks, vs, code, stream)
    nd = st.num_decoding_seqs
    if nd > 0:
        p = st.num_prefill_tokens
        _hip.call("swl_rotary_store_kv_decode", _hip.ptr(q[p:]), _hip.ptr(k[p:]), _hip.ptr(v[p:]),
                  _hip.ptr(st.position_cos), _hip.ptr(st.position_sin), _hip.ptr(st.position_indices[p:]),
                  _hip.ptr(k_cache), _hip.ptr(v_cache), _hip.ptr(block_table), _hip.ptr(st.seq_ids[st.num_prefill_seqs:]),
                  _hip.ptr(st.decoding_seq_lens), nd, q.shape[1], k.shape[1], q.shape[2], cur_layer,
                  model_config.num_layers, engine_config.block_size, block_table.shape[1], qs, ks, vs, code, stream)


def rotary_embedding_and_store_kvcache_decode_from_splitk(partials, k_cache: torch.Tensor,
                                                          v_cache: torch.Tensor, block_table: torch.Tensor,
                                                          model_config, engine_config, infer_state,
                                                          cur_layer: int):
    """rotary_embedding_and_store_kvcache_decode fed by the split-K partial slabs of the fused qkv
    projection: sums + rounds them, rotates q and k, stores k/v into the paged pools, and returns the
    (rotated) q, k and v as [Bd, heads, head_dim] views of one fresh [Bd, (H+2*KVH)*D] buffer."""
    assert infer_state.num_prefill_seqs == 0 and infer_state.position_indices is not None
    nd = infer_state.num_decoding_seqs
    h, kvh, d = model_config.num_q_heads, model_config.num_kv_heads, model_config.head_dim
    m, n = partials.shape
    assert m == nd and n == (h + 2 * kvh) * d
    qkv = torch.empty((nd, n), dtype=partials.dtype, device=k_cache.device)
    q = qkv[:, :h * d].view(nd, h, d)
    k = qkv[:, h * d:(h + kvh) * d].view(nd, kvh, d)
    v = qkv[:, (h + kvh) * d:].view(nd, kvh, d)
    _hip.call("swl_splitk_rotary_store_kv_decode", _hip.ptr(q), _hip.ptr(k), _hip.ptr(v),
              _hip.ptr(partials.slabs), partials.k_splits, _hip.ptr(infer_state.position_cos),
              _hip.ptr(infer_state.position_sin), _hip.ptr(infer_state.position_indices),
              _hip.ptr(k_cache), _hip.ptr(v_cache), _hip.ptr(block_table), _hip.ptr(infer_state.seq_ids),
              _hip.ptr(infer_state.decoding_seq_lens), nd, h, kvh, d, cur_layer, model_config.num_layers,
              engine_config.block_size, block_table.shape[1], n, n, n, _hip.dtype_code(partials.dtype),
              _hip.stream())
    return q, k, v
