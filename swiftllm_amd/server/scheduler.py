"""Iteration-level FCFS scheduler over the paged KV pool, with swapping and (optionally) piggybacking.

Behaviour follows the reference's swiftllm/server/scheduler.py:33-140: strict arrival order; a prefill
batch is admitted while it fits the sequence/token/block budgets; otherwise every running request
decodes one token; when the pool cannot hold the running set the most recent requests are swapped out,
and swapped requests return (oldest first) before any new prompt is admitted.

Two deliberate differences:
  * `piggyback=True` lets the running (decoding) requests ride along with an admitted prefill batch in
    ONE forward (prefill sequences first, as LlamaModel.forward requires) — the SARATHI-style batch the
    reference's forward supports but its scheduler never emits ("If you want decoding requests to be
    piggybacked, you can do it here", scheduler.py:93-94);
  * `get_next_batch` returns three lists. The reference returns `reversed(...)` for the swapped-out
    requests — an iterator that is always truthy, so its idle engine never sleeps (engine.py:122).
"""
from collections import deque
from typing import Optional, Deque, List, Tuple

from swiftllm_amd.utils import cdiv
from .structs import Request


class RequestIdManager:
    """Hands out block-table rows [0, max_id); lowest free id first."""

    def __init__(self, max_id: int):
        self.max_id = max_id
        self._free = list(range(max_id - 1, -1, -1))

    def get_id(self) -> int:
        if not self._free:
            raise RuntimeError("No more available request ids. Please try to increase `max_seqs_in_block_table`")
        return self._free.pop()

    def free_id(self, req_id: int):
        self._free.append(req_id)

    def free_ids(self, req_ids: List[int]):
        self._free.extend(req_ids)


class Scheduler:
    def __init__(self, model_config, engine_config, num_gpu_blocks: int, piggyback: bool = False):
        self.model_config = model_config    # unused, kept for the reference's constructor signature
        self.engine_config = engine_config
        self.num_gpu_blocks = num_gpu_blocks
        self.piggyback = piggyback
        self.waiting_q: Deque[Request] = deque()
        self.running_q: List[Request] = []
        self.swapped_q: Deque[Request] = deque()
        self.request_id_manager = RequestIdManager(engine_config.max_seqs_in_block_table)
        # limits of the data plane the Engine fills in once the model exists (None = unknown, not checked):
        # rows of the rotary table (LlamaModel.forward raises past it) and the vocabulary size
        self.max_seq_len: Optional[int] = None
        self.vocab_size: Optional[int] = getattr(model_config, "vocab_size", None)

    # ---- budgets -------------------------------------------------------------------------------------
    def _blocks(self, req: Request, extra_tokens: int = 0) -> int:
        return cdiv(req.num_tokens() + extra_tokens, self.engine_config.block_size)

    def _running_blocks(self) -> int:
        # (once per decode iteration over the whole running set: spelled out, no per-request method calls)
        bs = self.engine_config.block_size
        return sum((r.prompt_len + len(r.output_token_ids) + bs - 1) // bs for r in self.running_q)

    # ---- events --------------------------------------------------------------------------------------
    def why_unservable(self, req: Request) -> Optional[str]:
        """None, or the reason this request could never run to completion under the engine's limits. Such a
        request must not enter the queues: strict FCFS would park everything behind it (a prompt over the token
        budget is never admitted) or it would swap in and out forever (a sequence larger than the pool). The
        reference has no such check and hangs (scheduler.py:60-90)."""
        ecfg = self.engine_config
        if req.prompt_len <= 0:
            return "empty prompt"
        if req.output_len <= 0:
            return "output_len must be positive"
        ids = req.prompt_token_ids
        if self.vocab_size is not None and ids is not None:
            # ids index the embedding table on the device: out-of-range values must never reach a kernel
            if not all(isinstance(t, int) and not isinstance(t, bool) and 0 <= t < self.vocab_size for t in ids):
                return f"prompt_token_ids must be integers in [0, {self.vocab_size})"
        if self.max_seq_len is not None and req.prompt_len + req.output_len > self.max_seq_len:
            return (f"prompt ({req.prompt_len}) + output_len ({req.output_len}) exceeds the model's "
                    f"{self.max_seq_len} rotary positions")
        if req.prompt_len > ecfg.max_tokens_in_batch:
            return f"prompt of {req.prompt_len} tokens exceeds max_tokens_in_batch ({ecfg.max_tokens_in_batch})"
        blocks = cdiv(req.prompt_len + req.output_len, ecfg.block_size)
        if blocks > ecfg.max_blocks_per_seq:
            return f"sequence needs {blocks} KV blocks, max_blocks_per_seq is {ecfg.max_blocks_per_seq}"
        if blocks > self.num_gpu_blocks:
            return f"sequence needs {blocks} KV blocks, the pool has {self.num_gpu_blocks}"
        return None

    def on_requests_arrival(self, requests: List[Request]):
        self.waiting_q.extend(requests)

    def get_next_batch(self) -> Tuple[List[Request], List[Request], List[Request]]:
        """(batch to forward, requests to swap in first, requests to swap out first)."""
        ecfg = self.engine_config
        if not self.swapped_q:
            admitted = self._admit_prefills()
            if admitted:
                for req in admitted:
                    req.request_id = self.request_id_manager.get_id()
                riders = list(self.running_q) if self.piggyback and self._riders_fit(admitted) else []
                self.running_q.extend(admitted)
                return admitted + riders, [], []

        # decode step for everything running; make room first if the pool is over-committed
        swapped_out: List[Request] = []
        used = self._running_blocks()
        while len(self.running_q) > ecfg.max_batch_size or used > self.num_gpu_blocks:
            victim = self.running_q.pop()           # the most recently admitted request yields
            used -= self._blocks(victim)
            swapped_out.append(victim)
        swapped_in: List[Request] = []
        if swapped_out:
            self.swapped_q.extendleft(swapped_out)  # they keep their place ahead of older swapped ones
        else:
            while self.swapped_q:
                cand = self.swapped_q[0]
                need = self._blocks(cand)
                if len(self.running_q) + 1 > ecfg.max_batch_size or used + need > self.num_gpu_blocks:
                    break
                self.running_q.append(self.swapped_q.popleft())
                used += need
                swapped_in.append(cand)
        return list(self.running_q), swapped_in, swapped_out[::-1]

    def _admit_prefills(self) -> List[Request]:
        ecfg = self.engine_config
        batch: List[Request] = []
        if not self.waiting_q:
            return batch
        blocks = self._running_blocks()
        tokens = 0
        while self.waiting_q:
            cand = self.waiting_q[0]
            need = self._blocks(cand)
            if (len(self.running_q) + len(batch) + 1 > ecfg.max_batch_size
                    or blocks + need > self.num_gpu_blocks
                    or tokens + cand.prompt_len > ecfg.max_tokens_in_batch):
                break       # strict FCFS: nothing may overtake the head of the queue
            batch.append(self.waiting_q.popleft())
            blocks += need
            tokens += cand.prompt_len
        return batch

    def _riders_fit(self, admitted: List[Request]) -> bool:
        """Decoding requests ride along only if their next token fits the token and block budgets."""
        ecfg = self.engine_config
        tokens = sum(r.prompt_len for r in admitted) + len(self.running_q)
        blocks = sum(self._blocks(r) for r in admitted) + sum(self._blocks(r, 1) for r in self.running_q)
        return tokens <= ecfg.max_tokens_in_batch and blocks <= self.num_gpu_blocks

    def on_batch_finish(self, batch: List[Request]):
        self.request_id_manager.free_ids([r.request_id for r in batch if r.is_finished()])
        self.running_q = [r for r in self.running_q if not r.is_finished()]

    def has_work(self) -> bool:
        return bool(self.waiting_q or self.running_q or self.swapped_q)
