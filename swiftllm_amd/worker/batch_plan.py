"""Host-side batch planning: everything `LlamaModel.forward` derives from its Python-list arguments
before touching the device. Pure numpy — unit-tested on CPU against the reference's arithmetic.

Reference: swiftllm/worker/model.py:268-324 (flattening, sequence lengths, start offsets, position
indices, the seq_block_size heuristic) and post_layer.py:24-29 (last-token indices).
"""
import dataclasses
import itertools
from typing import List, Sequence

import numpy as np


def select_seq_block_size(decoding_seq_lens: Sequence[int], num_kv_heads: int,
                          num_slots: int = 256, max_splits: int = 128) -> int:
    """Split-K width (tokens) of flash-decoding, sized for MI355X.

    The reference halves 2048 until `KVH * sum(len) / width >= 1024` programs exist (model.py:305-324:
    "1024 because ~128 SMs", one 1-warp program per q-head). Our phase-1 workgroup is 8 waves serving
    a whole kv-head group, one per CU, so the right launch is ~`num_slots` (= CU count, 256)
    workgroups in whole rounds, each with as many KV blocks as possible:

        width = KVH * sum(len) / num_slots        (tokens per workgroup for one full round)

    rounded up to 64 tokens. When that width already covers the longest sequence there is a single
    sequence block per sequence — phase 1 writes the output directly and phase 2 is never launched —
    and the width is rounded up to 256 so it (and a captured hipGraph) stays valid while the
    sequences grow. The width is a free internal choice: results are invariant up to fp32
    reassociation, which tests/test_gpu_kernels.py checks.
    """
    if len(decoding_seq_lens) == 0:
        return 2048
    total = sum(-(-n // 64) * 64 for n in decoding_seq_lens)
    longest = max(decoding_seq_lens)
    size = max(64, -(-(num_kv_heads * total) // (num_slots * 64)) * 64)
    if size >= longest:
        size = -(-size // 256) * 256
    while -(-longest // size) > max_splits:
        size *= 2
    return size


@dataclasses.dataclass
class BatchPlan:
    batch_size: int
    num_tokens: int
    num_prefill_seqs: int
    num_prefill_tokens: int
    max_prefill_len: int
    num_decoding_seqs: int
    max_decoding_len: int
    seq_block_size: int
    num_seq_blocks: int
    seq_lengths_list: List[int]
    # int32 arrays
    input_ids: np.ndarray               # [T]
    seq_ids: np.ndarray                 # [B]
    seq_lengths: np.ndarray             # [B]
    prefill_seq_lens: np.ndarray        # [Bp]
    prefill_start_locs_with_end: np.ndarray  # [Bp + 1]
    decoding_seq_lens: np.ndarray       # [Bd]
    position_indices: np.ndarray        # [T]
    last_token_indices: np.ndarray      # [B]
    # sequences of the caller's batch; batch_size - num_real_seqs trailing rows are inert padding (length 0) that rounds a
    # pure-decode batch up to its hipGraph bucket (worker/model.py: _decode_batch_bucket). -1: not set = batch_size
    num_real_seqs: int = -1

    @property
    def real_seqs(self) -> int:
        return self.num_real_seqs if self.num_real_seqs >= 0 else self.batch_size

    SEGMENTS = ("input_ids", "seq_ids", "seq_lengths", "prefill_seq_lens",
                "prefill_start_locs_with_end", "decoding_seq_lens", "position_indices",
                "last_token_indices")

    def packed_layout(self):
        """(name, offset, size) of every array inside one int32 buffer; offsets are multiples of 4
        elements (16 bytes). The layout depends only on (T, B, Bp), so a replayed hipGraph of a
        pure-decode batch of size B always finds its metadata at the same addresses."""
        out, off = [], 0
        for name in self.SEGMENTS:
            n = getattr(self, name).size
            out.append((name, off, n))
            off += (n + 3) // 4 * 4
        return out, off

    def pack_into(self, buf: np.ndarray) -> int:
        layout, total = self.packed_layout()
        for name, off, n in layout:
            buf[off:off + n] = getattr(self, name)
        return total


def plan_batch(input_ids_list: Sequence[Sequence[int]], seq_ids_list: Sequence[int],
               decoding_seq_lens_list: Sequence[int], num_kv_heads: int,
               num_slots: int = 256) -> BatchPlan:
    """Prefill sequences come first in all three lists (reference model.py:268-270); a decoding
    sequence contributes exactly one token and its length INCLUDES that token."""
    batch_size = len(input_ids_list)
    num_decoding = len(decoding_seq_lens_list)
    num_prefill = batch_size - num_decoding
    if num_prefill < 0 or len(seq_ids_list) != batch_size:
        raise ValueError("inconsistent batch: need len(seq_ids) == len(input_ids) >= len(decoding_seq_lens)")
    prefill_lens = [len(ids) for ids in input_ids_list[:num_prefill]]
    flat = np.fromiter(itertools.chain.from_iterable(input_ids_list), dtype=np.int32)
    num_tokens = flat.size
    num_prefill_tokens = sum(prefill_lens)
    if num_tokens - num_prefill_tokens != num_decoding:
        raise ValueError("every decoding sequence must contribute exactly one token")
    pl = np.asarray(prefill_lens, dtype=np.int32)
    dl = np.asarray(decoding_seq_lens_list, dtype=np.int32)
    starts_with_end = np.zeros(num_prefill + 1, dtype=np.int32)
    np.cumsum(pl, out=starts_with_end[1:])
    pos = np.empty(num_tokens, dtype=np.int32)
    if num_prefill_tokens:
        # 0..len-1 for every prefill sequence: global arange minus each token's sequence start
        pos[:num_prefill_tokens] = (np.arange(num_prefill_tokens, dtype=np.int32)
                                    - np.repeat(starts_with_end[:-1], pl))
    pos[num_prefill_tokens:] = dl - 1
    last = np.concatenate((starts_with_end[1:] - 1,
                           np.arange(num_prefill_tokens, num_tokens, dtype=np.int32))).astype(np.int32)
    sbs = select_seq_block_size(decoding_seq_lens_list, num_kv_heads, num_slots)
    max_dec = max(decoding_seq_lens_list) if num_decoding else 0
    seq_lengths_list = prefill_lens + list(decoding_seq_lens_list)
    return BatchPlan(
        batch_size=batch_size, num_tokens=num_tokens, num_prefill_seqs=num_prefill,
        num_prefill_tokens=num_prefill_tokens,
        max_prefill_len=max(prefill_lens) if prefill_lens else 0,
        num_decoding_seqs=num_decoding, max_decoding_len=max_dec, seq_block_size=sbs,
        num_seq_blocks=(max_dec + sbs - 1) // sbs, seq_lengths_list=seq_lengths_list,
        input_ids=flat, seq_ids=np.asarray(seq_ids_list, dtype=np.int32),
        seq_lengths=np.asarray(seq_lengths_list, dtype=np.int32), prefill_seq_lens=pl,
        prefill_start_locs_with_end=starts_with_end, decoding_seq_lens=dl, position_indices=pos,
        last_token_indices=last)
