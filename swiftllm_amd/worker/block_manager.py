"""BlockManager — owner of the device-resident block table of one KV pool.

Same public surface as the reference's swiftllm/worker/block_manager.py:5-103 (constructor
arguments, `allocate_blocks_for_seqs`, `free_blocks_for_seqs`, `gather_allocated_blocks_and_free`,
`get_num_allocated_blocks`, the three device tensors and `num_free_blocks`) and the same results:
blocks are handed out lowest-id-first (reference: `torch.nonzero(is_block_free)[:n]`,
block_manager.py:50) and appended to the sequence's row of `block_table`.

What differs is where decisions are made. The reference decides on the device and reads the answer
back (an `.all()` assert, a `.item()` and a `nonzero` per forward = 3 host syncs and an
O(num_blocks) device scan, block_manager.py:70-75). Here the allocator state is mirrored on the host
(`BlockAllocatorHost`: numpy bitmap + per-sequence id lists), every decision is host arithmetic, and
the device tensors are brought up to date by ONE small pinned H2D copy plus ONE kernel
(`swl_block_table_set`, which also clears `is_block_free`) — nothing is read back, so the forward
never stalls, and a decode step that needs no new block launches nothing at all.
"""
from typing import Iterable, List, Sequence

import numpy as np
import torch

from .kernels.block_mgmt import (
    set_block_table_and_num_seq_alloc_blocks,
    unset_block_table_and_num_seq_alloc_blocks,
    gather_allocated_blocks_and_unset,
)


class BlockAllocatorHost:
    """Pure-host mirror of the allocator state (no torch, no device): unit-testable on CPU."""

    def __init__(self, device_name: str, num_blocks: int, max_seqs: int, max_blocks_per_seq: int,
                 block_size: int):
        self.device_name = device_name
        self.num_blocks = num_blocks
        self.num_free_blocks = num_blocks
        self.max_seqs = max_seqs
        self.max_blocks_per_seq = max_blocks_per_seq
        self.block_size = block_size
        self.is_free = np.ones(num_blocks, dtype=bool)
        self.seq_blocks = {}        # seq_id -> list[int] of block ids, in logical order
        self._lowest_maybe_free = 0  # every index below this is known to be in use

    def num_allocated(self, seq_id: int) -> int:
        blocks = self.seq_blocks.get(seq_id)
        return len(blocks) if blocks else 0

    def plan_allocation(self, seq_ids: Sequence[int], target_lens: Sequence[int]):
        """Decide the blocks each sequence gains so that it owns ceil(len / block_size) blocks.
        Returns (block_needed int32[B], new_block_ids int32[sum]) and commits the decision."""
        bs = self.block_size
        needed = np.empty(len(seq_ids), dtype=np.int32)
        for i, (sid, tlen) in enumerate(zip(seq_ids, target_lens)):
            if not 0 <= sid < self.max_seqs:
                raise RuntimeError(f"sequence id {sid} outside the block table (0..{self.max_seqs - 1})")
            target = -(-int(tlen) // bs)
            have = self.num_allocated(sid)
            if have > target:
                raise AssertionError(
                    f"(On {self.device_name}) Logic error: sequence {sid} already owns {have} blocks, "
                    f"more than the {target} needed for length {tlen}")
            if target > self.max_blocks_per_seq:
                raise RuntimeError(f"sequence {sid} needs {target} blocks > max_blocks_per_seq "
                                   f"{self.max_blocks_per_seq}")
            needed[i] = target - have
        total = int(needed.sum())
        if total > self.num_free_blocks:
            raise RuntimeError(
                f"No enough free blocks available on {self.device_name} ({self.num_blocks} in total, "
                f"{self.num_free_blocks} free, {total} requested)")
        if total == 0:
            return needed, np.empty(0, dtype=np.int32)
        picked = self._lowest_free(total)
        self.is_free[picked] = False
        self.num_free_blocks -= total
        self._lowest_maybe_free = int(picked[-1]) + 1
        off = 0
        for sid, n in zip(seq_ids, needed):
            if n:
                self.seq_blocks.setdefault(sid, []).extend(picked[off:off + n].tolist())
                off += n
        return needed, picked

    def _lowest_free(self, total: int) -> np.ndarray:
        """The `total` lowest free block ids (the caller has checked that many exist). Scanned in windows from the first
        index that may be free: a 288 GB pool of a small model has tens of millions of blocks, and a decode step that
        needs one block per sequence must not pay for a scan of all of them."""
        lo, found, have = self._lowest_maybe_free, [], 0
        window = max(4096, 4 * total)
        while have < total:
            hit = np.flatnonzero(self.is_free[lo:lo + window])
            if hit.size:
                found.append(hit[:total - have].astype(np.int32) + lo)
                have += found[-1].size
            lo += window
            window *= 2
        return found[0] if len(found) == 1 else np.concatenate(found)

    def release(self, seq_ids: Iterable[int]) -> List[int]:
        """Free every block of the given sequences; returns the freed ids in batch order."""
        freed: List[int] = []
        for sid in seq_ids:
            blocks = self.seq_blocks.pop(sid, None)
            if blocks:
                freed.extend(blocks)
        if freed:
            idx = np.asarray(freed, dtype=np.int64)
            self.is_free[idx] = True
            self.num_free_blocks += len(freed)
            self._lowest_maybe_free = min(self._lowest_maybe_free, int(idx.min()))
        return freed


def _to_list(x) -> list:
    if isinstance(x, torch.Tensor):
        return x.tolist()
    return list(x)


class BlockManager:
    """Block table + free list of one pool ("GPU" KV pool or "CPU" swap pool). As in the reference
    the tables of BOTH managers live in device memory (block_manager.py:25-41)."""

    def __init__(self, device_name: str, num_blocks: int, max_seqs_in_block_table: int,
                 max_blocks_per_seq: int, block_size: int, device="cuda"):
        self.device_name = device_name
        self.num_blocks = num_blocks
        self.block_size = block_size
        self.host = BlockAllocatorHost(device_name, num_blocks, max_seqs_in_block_table,
                                       max_blocks_per_seq, block_size)
        self.device = torch.device(device)
        self.num_seq_allocated_blocks = torch.zeros((max_seqs_in_block_table,), dtype=torch.int32,
                                                    device=self.device)
        self.block_table = torch.empty((max_seqs_in_block_table, max_blocks_per_seq),
                                       dtype=torch.int32, device=self.device)
        self.is_block_free = torch.ones((num_blocks,), dtype=torch.bool, device=self.device)
        self._staging = None        # pinned int32 staging buffer, grown on demand
        self._staging_done = None   # event recorded after the last copy out of it

    @property
    def num_free_blocks(self) -> int:
        return self.host.num_free_blocks

    # ---- host -> device metadata upload -------------------------------------------------------------
    def _upload(self, *arrays: np.ndarray):
        """Pack int32 arrays into the pinned staging buffer, issue one async H2D copy, return device
        views. The staging buffer is reused; an event guards it against being overwritten while the
        previous copy is still queued (normally long complete: one cheap event query)."""
        total = sum(a.size for a in arrays)
        if self._staging_done is not None:
            self._staging_done.synchronize()    # previous copy out of the staging buffer finished
        if self._staging is None or self._staging.numel() < total:
            cap = max(1024, 1 << (max(total, 1) - 1).bit_length())
            self._staging = torch.empty(cap, dtype=torch.int32, pin_memory=True)
            self._staging_np = self._staging.numpy()
        off = 0
        for a in arrays:
            self._staging_np[off:off + a.size] = a
            off += a.size
        dev = torch.empty(total, dtype=torch.int32, device=self.device)
        dev.copy_(self._staging[:total], non_blocking=True)
        if self._staging_done is None:
            self._staging_done = torch.cuda.Event()
        self._staging_done.record()
        views, off = [], 0
        for a in arrays:
            views.append(dev[off:off + a.size])
            off += a.size
        return views

    # ---- public API (reference names) -----------------------------------------------------------------
    def allocate_blocks_for_seqs(self, seq_ids, target_lens) -> torch.Tensor:
        """Make sure sequence seq_ids[i] owns ceil(target_lens[i] / block_size) blocks. Returns the
        newly allocated block ids (int32 device tensor, batch order) — used by the swap path."""
        seq_ids_l, lens_l = _to_list(seq_ids), _to_list(target_lens)
        needed, picked = self.host.plan_allocation(seq_ids_l, lens_l)
        if picked.size == 0:
            return torch.empty((0,), dtype=torch.int32, device=self.device)
        ids_np = np.asarray(seq_ids_l, dtype=np.int32)
        excl = (np.cumsum(needed) - needed).astype(np.int32)
        d_ids, d_need, d_excl, d_new = self._upload(ids_np, needed, excl, picked)
        set_block_table_and_num_seq_alloc_blocks(
            self.num_seq_allocated_blocks, self.block_table, d_new, d_ids, d_need,
            is_block_free=self.is_block_free, block_needed_excl_cumsum=d_excl)
        return d_new

    def free_blocks_for_seqs(self, seq_ids):
        """Release every block of the given sequences."""
        seq_ids_l = [s for s in _to_list(seq_ids) if self.host.num_allocated(s) > 0]
        if not seq_ids_l:
            return
        self.host.release(seq_ids_l)
        (d_ids,) = self._upload(np.asarray(seq_ids_l, dtype=np.int32))
        unset_block_table_and_num_seq_alloc_blocks(self.num_seq_allocated_blocks, self.block_table,
                                                   d_ids, self.is_block_free)

    def gather_allocated_blocks_and_free(self, seq_ids) -> torch.Tensor:
        """Block ids of the given sequences (batch order, int32 device tensor); frees them."""
        seq_ids_l = _to_list(seq_ids)
        counts = np.asarray([self.host.num_allocated(s) for s in seq_ids_l], dtype=np.int32)
        total = int(counts.sum())
        if not seq_ids_l or total == 0:
            self.host.release(seq_ids_l)
            return torch.empty((0,), dtype=torch.int32, device=self.device)
        self._last_gathered_host = self.host.release(seq_ids_l)
        excl = (np.cumsum(counts) - counts).astype(np.int32)
        d_ids, d_excl = self._upload(np.asarray(seq_ids_l, dtype=np.int32), excl)
        return gather_allocated_blocks_and_unset(self.num_seq_allocated_blocks, self.block_table,
                                                 d_ids, self.is_block_free, out_excl_cumsum=d_excl,
                                                 total=total)

    def get_num_allocated_blocks(self, seq_ids) -> torch.Tensor:
        """Blocks currently owned by each of the given sequences (int32 device tensor)."""
        counts = [self.host.num_allocated(s) for s in _to_list(seq_ids)]
        return torch.tensor(counts, dtype=torch.int32, device=self.device)

    # ---- host-side conveniences used by LlamaModel (no device round trips) ---------------------------
    def get_num_allocated_blocks_host(self, seq_ids: Sequence[int]) -> List[int]:
        return [self.host.num_allocated(s) for s in seq_ids]

    def get_block_ids_host(self, seq_id: int) -> List[int]:
        return list(self.host.seq_blocks.get(seq_id, ()))
