"""The operator layer — the drop-in boundary of the data plane.

One module per module of the reference's swiftllm/worker/kernels, exporting functions with the same
names and argument meaning; each one validates its tensors (the reference's asserts), then calls the
gfx950 kernel through the C ABI (swiftllm_amd._hip). `swap_blocks` (reference: the swiftllm_c
extension) lives in block_swapping.py.
"""
from .linear import linear
from .rmsnorm import rmsnorm_inplace, fused_add_rmsnorm_inplace
from .rotary_emb import rotary_embedding_inplace
from .kvcache_mgmt import store_kvcache
from .prefill_attn import prefill_attention
from .paged_attn import paged_attention
from .silu_and_mul import silu_and_mul_inplace
from .block_mgmt import (
    set_block_table_and_num_seq_alloc_blocks,
    unset_block_table_and_num_seq_alloc_blocks,
    gather_allocated_blocks_and_unset,
)
from .block_swapping import swap_blocks

__all__ = [
    "linear", "rmsnorm_inplace", "fused_add_rmsnorm_inplace", "rotary_embedding_inplace",
    "store_kvcache", "prefill_attention", "paged_attention", "silu_and_mul_inplace",
    "set_block_table_and_num_seq_alloc_blocks", "unset_block_table_and_num_seq_alloc_blocks",
    "gather_allocated_blocks_and_unset", "swap_blocks",
]
