"""Prefill (varlen causal) attention operator.

Reference: swiftllm/worker/kernels/prefill_attn.py:102-139 (`prefill_attention`), which has the
contract of the `vllm_flash_attn.flash_attn_varlen_func` call the reference's layer actually makes
(transformer_layer.py:83-96). Here it is the one and only prefill attention path.
"""
import torch

from swiftllm_amd import _hip
from ._layout import token_stride


def prefill_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, o: torch.Tensor,
                      model_config, engine_config, infer_state):
    """o[:P] = causal softmax(q k^T * scale) v per prefill sequence (GQA). q/k/v/o are
    [tokens, heads, head_dim]; only the first `num_prefill_tokens` rows are touched."""
    _hip.require_gpu_tensor(q, "q")
    if infer_state.num_prefill_seqs == 0:
        return
    cu = infer_state.prefill_seq_start_locs_with_end
    assert cu.dtype == torch.int32 and cu.is_contiguous() and cu.numel() == infer_state.num_prefill_seqs + 1
    assert q.dtype == k.dtype == v.dtype == o.dtype
    if o.dim() == 2:
        o = o.view(o.shape[0], model_config.num_q_heads, model_config.head_dim)
    _hip.call("swl_prefill_attn_varlen", _hip.ptr(o), _hip.ptr(q), _hip.ptr(k), _hip.ptr(v),
              _hip.ptr(cu), infer_state.num_prefill_seqs, infer_state.max_prefill_len,
              model_config.num_q_heads, model_config.num_kv_heads, model_config.head_dim,
              infer_state.softmax_scale, token_stride(q, "q"), token_stride(k, "k"),
              token_stride(v, "v"), token_stride(o, "o"), _hip.dtype_code(q.dtype), _hip.stream())
