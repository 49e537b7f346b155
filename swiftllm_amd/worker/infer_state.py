"""LlamaInferState — the per-forward metadata bundle handed to every operator.

Field-compatible with the reference's swiftllm/worker/infer_state.py:4-29 (same names and meaning:
this is the kernel-argument contract), plus three optional fields this implementation uses to keep
the hot path free of extra launches.
"""
import dataclasses
from typing import Optional

import torch


@dataclasses.dataclass
class LlamaInferState:
    batch_size: int
    num_tokens: int

    seq_ids: torch.Tensor   # [batch_size] int32
    softmax_scale: float    # head_dim ** -0.5

    num_prefill_seqs: int
    num_prefill_tokens: int
    prefill_seq_start_locs: torch.Tensor            # [num_prefill_seqs] int32
    prefill_seq_start_locs_with_end: torch.Tensor   # [num_prefill_seqs + 1] int32
    prefill_seq_lens: torch.Tensor                  # [num_prefill_seqs] int32
    max_prefill_len: int

    num_decoding_seqs: int
    decoding_seq_lens: torch.Tensor     # [num_decoding_seqs] int32, INCLUDING the token being decoded
    max_decoding_len: int

    seq_block_size: int     # split-K width of flash-decoding (tokens)
    num_seq_blocks: int     # ceil(max_decoding_len / seq_block_size)

    position_cos: torch.Tensor  # [num_tokens, head_dim/2] rows, or the whole rope cache (see below)
    position_sin: torch.Tensor

    ignore_kvcache: bool    # profiling run: no KV store, no paged attention

    # ---- additions --------------------------------------------------------------------------------
    # When set, position_cos/sin are the model's full rope tables and the rotary kernel looks up row
    # position_indices[t] itself (the reference gathers the rows first, model.py:350-351).
    position_indices: Optional[torch.Tensor] = None     # [num_tokens] int32
    # Row index of each sequence's last token in the activation matrix (post layer gather).
    last_token_indices: Optional[torch.Tensor] = None   # [batch_size] int32
    # Preallocated fp32 scratch for the flash-decoding partials.
    paged_attn_scratch: Optional[torch.Tensor] = None
