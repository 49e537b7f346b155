"""Engine — the asyncio control loop around one LlamaModel replica.

Public surface of the reference's swiftllm/server/engine.py:15-180: `initialize()`,
`add_request_and_stream()`, `add_request_and_wait()`, `start_all_event_loops()`. Two cooperating loops:
one tokenizes arrivals in batches and hands them to the scheduler, one asks the scheduler for the next
batch, performs the swaps it orders, runs `LlamaModel.forward` in a worker thread (the event loop stays
responsive while the GPU works) and fans the tokens out to per-request queues.

The fan-out of step k (one queue put and one consumer wake-up per request: ~6 us each, ~200 us at batch 32) is not
on the critical path: it is held back until step k+1 has been LAUNCHED — LlamaModel.forward calls
`after_launch_hook` from the worker thread once its kernels are enqueued and before it blocks on the tokens — and
then runs on the event loop while the GPU works. With nothing left to launch it runs at once.
"""
import asyncio
import functools
from typing import AsyncGenerator, List, Optional, Tuple

from swiftllm_amd.engine_config import EngineConfig
from swiftllm_amd.model_config import LlamaModelConfig
from swiftllm_amd.utils import GB

from .scheduler import Scheduler
from .structs import RawRequest, Request, StepOutput
from .tokenization import TokenizationEngine


class Engine:
    def __init__(self, engine_config: EngineConfig, model=None, piggyback: bool = False):
        """`model`: an object with LlamaModel's methods (tests inject a fake); None = build the real one
        in `initialize()`."""
        self.engine_config = engine_config
        self.model = model
        self.model_config = getattr(model, "model_config", None)
        self.piggyback = piggyback
        self.initialized = False
        self.event_loop = None
        self.scheduler: Optional[Scheduler] = None
        self.tokenization_engine = None
        self.untokenized_raw_requests: List[Tuple[Request, RawRequest]] = []
        self.num_forwards = 0
        self._undelivered: List[Tuple[Request, int]] = []    # (request, token) of the last step, not yet fanned out

    async def _run_on_model_async(self, func, *args, **kwargs):
        return await self.event_loop.run_in_executor(None, functools.partial(func, *args, **kwargs))

    async def initialize(self, num_gpu_blocks: Optional[int] = None):
        self.event_loop = asyncio.get_running_loop()
        if self.model is None:
            from swiftllm_amd.worker.model import LlamaModel
            print("[Engine] Initializing model...")
            self.model = LlamaModel(self.engine_config)
            self.model_config = self.model.model_config
            print("[Engine] Loading weights...")
            self.model.load_weights()
            print("[Engine] Profiling kv blocks...")
            num_gpu_blocks = self.model.profile_num_blocks()
            block_bytes = self.engine_config.block_size * self.model_config.get_kvslot_size(self.model.dtype)
            print(f"[Engine] Number of GPU blocks: {num_gpu_blocks} ({num_gpu_blocks * block_bytes / GB:.2f} GB)")
            print(f"[Engine] Number of CPU blocks: {self.engine_config.num_cpu_blocks} "
                  f"({self.engine_config.num_cpu_blocks * block_bytes / GB:.2f} GB)")
            print("[Engine] Allocating kv cache and swap...")
            self.model.init_kvcache_and_swap(num_gpu_blocks)
        elif num_gpu_blocks is None:
            num_gpu_blocks = self.model.num_blocks
        self.scheduler = Scheduler(self.model_config, self.engine_config, num_gpu_blocks, self.piggyback)
        rope = getattr(self.model, "_cos_cached", None)
        if rope is not None:     # a request that would outgrow the rotary table is refused up front (HTTP 400),
            self.scheduler.max_seq_len = int(rope.shape[0])     # not left to raise inside forward mid-flight
        self.tokenization_engine = TokenizationEngine(self.engine_config)
        if hasattr(self.model, "after_launch_hook"):
            self.model.after_launch_hook = self._on_forward_launched
        self.initialized = True
        print("[Engine] Model initialized")

    # ---- request entry points -----------------------------------------------------------------------------
    def _enqueue(self, raw_request: RawRequest) -> Request:
        request = Request(raw_request)
        self.untokenized_raw_requests.append((request, raw_request))
        return request

    async def add_request_and_stream(self, raw_request: RawRequest) -> AsyncGenerator[StepOutput, None]:
        """Yield a StepOutput per generated token. (Ends after `output_len` deliveries, not on
        `request.is_finished()`: the request's own token list runs one step ahead of what has been fanned out.)"""
        request = self._enqueue(raw_request)
        delivered = 0
        while True:
            step_output = await request.output_q.get()
            if step_output is None:     # rejected (request.error says why)
                break
            yield step_output
            request.output_q.task_done()
            delivered += 1
            if delivered >= request.output_len:
                break

    async def add_request_and_wait(self, raw_request: RawRequest) -> Tuple[Request, List[int]]:
        """Wait for the whole generation; returns (request, output token ids)."""
        request = self._enqueue(raw_request)
        await request.finished_event.wait()
        return request, request.output_token_ids

    # ---- loops ------------------------------------------------------------------------------------------------
    async def _tokenize_raw_request_event_loop(self):
        while True:
            if not self.untokenized_raw_requests:
                await asyncio.sleep(0.002)
                continue
            pending, self.untokenized_raw_requests = self.untokenized_raw_requests, []
            texts = [(req, raw.prompt) for req, raw in pending if not req.prompt_token_ids]
            if texts:
                ids = await self.tokenization_engine.batched_tokenize([p for _, p in texts])
                for (req, _), token_ids in zip(texts, ids):
                    req.prompt_token_ids = list(token_ids)
                    req.prompt_len = len(token_ids)
            servable = []
            for req, _ in pending:
                req.error = self.scheduler.why_unservable(req)
                if req.error is None:
                    servable.append(req)
                else:           # answer at once: waiters wake up with no tokens, streams end
                    req.finished_event.set()
                    req.output_q.put_nowait(None)
            self.scheduler.on_requests_arrival(servable)
            await asyncio.sleep(0.001)

    def _deliver(self):
        """Fan the held-back tokens out (event-loop thread only; idempotent)."""
        pending, self._undelivered = self._undelivered, []
        for req, tok in pending:
            req.output_q.put_nowait(StepOutput(tok, req))
            if req.is_finished():
                req.finished_event.set()

    def _on_forward_launched(self):
        """LlamaModel.after_launch_hook: runs in the worker thread, right after the step's kernels were enqueued."""
        self.event_loop.call_soon_threadsafe(self._deliver)

    async def step(self) -> bool:
        """One scheduling iteration; False when there was nothing to do."""
        batch, swap_in, swap_out = self.scheduler.get_next_batch()
        if not batch:
            self._deliver()     # no launch to hide behind
            if not swap_in and not swap_out:
                return False
        if swap_out:
            await self._run_on_model_async(self.model.swap_out_seqs, [r.request_id for r in swap_out])
        if swap_in:
            await self._run_on_model_async(self.model.swap_in_seqs, [r.request_id for r in swap_in])
        if batch:
            # prefill sequences first (the scheduler orders them so), their whole prompt; decoding ones
            # bring their last token and their length INCLUDING it
            input_ids = [r.prompt_token_ids if r.is_prefill_stage() else [r.output_token_ids[-1]] for r in batch]
            seq_ids = [r.request_id for r in batch]
            decoding_lens = [r.num_tokens() for r in batch if not r.is_prefill_stage()]
            try:
                tokens = await self._run_on_model_async(self.model.forward, input_ids, seq_ids, decoding_lens)
            finally:
                self._deliver()     # (a data plane without the hook, or a forward that raised before launching)
            self.num_forwards += 1
            for req, tok in zip(batch, tokens):
                req.output_token_ids.append(tok)
            finished = [r.request_id for r in batch if r.is_finished()]
            if finished:
                # release KV blocks before anyone is told: a caller that sees "finished" may tear us down
                await self._run_on_model_async(self.model.free_seqs_resources, finished)
            self._undelivered = list(zip(batch, tokens))
            self.scheduler.on_batch_finish(batch)
        return True

    async def _main_event_loop(self):
        while True:
            if not await self.step():
                await asyncio.sleep(0.005)

    async def start_all_event_loops(self):
        assert self.initialized, "Engine not initialized. Please call `initialize()` before starting the event loop."
        await asyncio.gather(self._tokenize_raw_request_event_loop(), self._main_event_loop())
