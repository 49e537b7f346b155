"""KV-cache store operator. Reference: swiftllm/worker/kernels/kvcache_mgmt.py:81-122."""
import torch

from swiftllm_amd import _hip
from ._layout import token_stride


def store_kvcache(k: torch.Tensor, v: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                  block_table: torch.Tensor, model_config, engine_config, infer_state,
                  cur_layer: int):
    """Write this forward's K/V rows into the paged pools: whole prefill sequences block by block,
    and the single new token of every decoding sequence."""
    _hip.require_gpu_tensor(k, "k")
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and block_table.is_contiguous()
    assert infer_state.seq_ids.is_contiguous() and infer_state.decoding_seq_lens.is_contiguous()
    assert k.dtype == v.dtype == k_cache.dtype == v_cache.dtype
    ks, vs = token_stride(k, "k"), token_stride(v, "v")
    code, stream = _hip.dtype_code(k.dtype), _hip.stream()
    common = (cur_layer, model_config.num_layers, model_config.num_kv_heads,
              engine_config.block_size, model_config.head_dim, block_table.shape[1], ks, vs, code,
              stream)
    if infer_state.num_prefill_seqs > 0:
        _hip.call("swl_store_kv_prefill", _hip.ptr(k_cache), _hip.ptr(v_cache), _hip.ptr(k),
                  _hip.ptr(v), _hip.ptr(block_table), _hip.ptr(infer_state.seq_ids),
                  _hip.ptr(infer_state.prefill_seq_start_locs),
                  _hip.ptr(infer_state.prefill_seq_lens), infer_state.num_prefill_seqs,
                  infer_state.max_prefill_len, *common)
    if infer_state.num_decoding_seqs > 0:
        p = infer_state.num_prefill_tokens
        _hip.call("swl_store_kv_decode", _hip.ptr(k_cache), _hip.ptr(v_cache), _hip.ptr(k[p:]),
                  _hip.ptr(v[p:]), _hip.ptr(block_table),
                  _hip.ptr(infer_state.seq_ids[infer_state.num_prefill_seqs:]),
                  _hip.ptr(infer_state.decoding_seq_lens), infer_state.num_decoding_seqs, *common)
