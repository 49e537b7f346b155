"""Block-table maintenance operators. Reference: swiftllm/worker/kernels/block_mgmt.py:26-46,
:66-80 and :106-127 (same names, same argument order)."""
import torch

from swiftllm_amd import _hip


def _i32(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.int32 else t.to(torch.int32)


def set_block_table_and_num_seq_alloc_blocks(num_seq_allocated_blocks: torch.Tensor,
                                             block_table: torch.Tensor,
                                             candidate_blocks: torch.Tensor, seq_ids: torch.Tensor,
                                             block_needed: torch.Tensor,
                                             is_block_free: torch.Tensor = None,
                                             block_needed_excl_cumsum: torch.Tensor = None):
    """For batch entry i (sequence s = seq_ids[i]): append candidate_blocks[off_i : off_i+need_i] to
    block_table[s] and add need_i to num_seq_allocated_blocks[s] (off = exclusive cumsum of
    block_needed). With `is_block_free` given, the candidates are also marked used."""
    _hip.require_gpu_tensor(block_table, "block_table")
    batch = seq_ids.shape[0]
    if batch == 0:
        return
    block_needed = _i32(block_needed)
    if block_needed_excl_cumsum is None:
        block_needed_excl_cumsum = (torch.cumsum(block_needed, 0) - block_needed).to(torch.int32)
    candidate_blocks = _i32(candidate_blocks)
    _hip.call("swl_block_table_set", _hip.ptr(num_seq_allocated_blocks), _hip.ptr(block_table),
              _hip.ptr(candidate_blocks), _hip.ptr(_i32(seq_ids)), _hip.ptr(block_needed),
              _hip.ptr(block_needed_excl_cumsum), _hip.ptr(is_block_free), batch,
              block_table.shape[1], _hip.stream())


def unset_block_table_and_num_seq_alloc_blocks(num_seq_allocated_blocks: torch.Tensor,
                                               block_table: torch.Tensor, seq_ids: torch.Tensor,
                                               is_block_free: torch.Tensor):
    """Mark every block of the given sequences free and zero their allocated-block counts."""
    _hip.require_gpu_tensor(block_table, "block_table")
    batch = seq_ids.shape[0]
    if batch == 0:
        return
    _hip.call("swl_block_table_unset", _hip.ptr(num_seq_allocated_blocks), _hip.ptr(block_table),
              _hip.ptr(_i32(seq_ids)), _hip.ptr(is_block_free), batch, block_table.shape[1],
              _hip.stream())


def gather_allocated_blocks_and_unset(num_seq_allocated_blocks: torch.Tensor,
                                      block_table: torch.Tensor, seq_ids: torch.Tensor,
                                      is_block_free: torch.Tensor, out_excl_cumsum: torch.Tensor = None,
                                      total: int = None) -> torch.Tensor:
    """Return the block ids of the given sequences (concatenated in batch order, int32) and free
    them. `out_excl_cumsum`/`total` may be supplied by a caller that mirrors the counts on the host
    (no device sync); otherwise they are derived on the device like the reference does."""
    _hip.require_gpu_tensor(block_table, "block_table")
    if seq_ids.numel() == 0:
        return torch.empty((0,), dtype=torch.int32, device=block_table.device)
    seq_ids = _i32(seq_ids)
    if out_excl_cumsum is None:
        counts = num_seq_allocated_blocks[seq_ids.long()]
        incl = torch.cumsum(counts, 0)
        out_excl_cumsum = (incl - counts).to(torch.int32)
        total = int(incl[-1].item())
    gathered = torch.empty((total,), dtype=torch.int32, device=block_table.device)
    _hip.call("swl_block_table_gather", _hip.ptr(num_seq_allocated_blocks), _hip.ptr(block_table),
              _hip.ptr(seq_ids), _hip.ptr(is_block_free), _hip.ptr(out_excl_cumsum),
              _hip.ptr(gathered), seq_ids.shape[0], block_table.shape[1], _hip.stream())
    return gathered
