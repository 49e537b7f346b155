"""Engine — the control loop around one LlamaModel replica.

Public surface of the reference's swiftllm/server/engine.py:15-180: `initialize()`,
`add_request_and_stream()`, `add_request_and_wait()`, `start_all_event_loops()`.

The reference runs scheduling on the asyncio loop and hands every `forward` to a thread pool
(engine.py:121-176): two thread wake-ups per decode step (the pool worker, then the event loop: ~170 us when
both slept through a 4 ms step) plus the per-request fan-out (one queue put and one consumer wake-up per
request, ~6 us each) between one step's tokens and the next step's launch — 450-650 us per iteration against a
4 ms decode step (`tools/engine_overhead.py`). Here the critical loop lives on ONE dedicated model thread that
never sleeps while there is work: arrivals are drained from a thread-safe inbox, the scheduler picks the batch,
`LlamaModel.forward` runs in place, and the fan-out of step k is posted to the event loop only when step k+1 has
been LAUNCHED (`LlamaModel.after_launch_hook`, called once the kernels are enqueued and before the host blocks on
the tokens) — it runs on the asyncio thread while the GPU works. The asyncio side keeps what belongs there:
tokenization of arrivals, per-request queues and events, the HTTP layer.
"""
import asyncio
import functools
import gc
import queue
import threading
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
        self.num_swapped_out = self.num_swapped_in = 0     # sequences moved to / from the host swap pool so far
        # asyncio thread -> model thread: lists of servable requests (the scheduler is touched by the model thread only)
        self._inbox: "queue.SimpleQueue[List[Request]]" = queue.SimpleQueue()
        # model thread only: (request, token, finished) of the last step, not fanned out yet
        self._undelivered: List[Tuple[Request, int, bool]] = []
        self._stop = threading.Event()
        self._model_thread: Optional[threading.Thread] = None
        # event-loop thread only: requests somebody may still be waiting on (woken with `error` set if the model thread dies)
        self._live: set = set()
        self._dead = None               # why the model thread is gone (set by _fail_live): new requests are refused with it

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
            self.model.after_launch_hook = self._post_undelivered
        self.initialized = True
        print("[Engine] Model initialized")

    # ---- request entry points -----------------------------------------------------------------------------
    def _enqueue(self, raw_request: RawRequest) -> Request:
        request = Request(raw_request)
        if self._dead is not None:      # the model thread is gone: answer at once, nobody will ever serve this
            request.error = self._dead
            request.finished_event.set()
            request.output_q.put_nowait(None)
            return request
        self._live.add(request)         # (event-loop thread; leaves in _deliver / on rejection / in _fail_live)
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

    # ---- asyncio side -------------------------------------------------------------------------------------------
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
                req.error = self._dead or self.scheduler.why_unservable(req)      # (reads the engine's limits only)
                if req.error is None:
                    servable.append(req)
                else:           # answer at once: waiters wake up with no tokens, streams end
                    self._live.discard(req)
                    req.finished_event.set()
                    req.output_q.put_nowait(None)
            if servable:
                self._inbox.put(servable)
            await asyncio.sleep(0.001)

    def _deliver(self, outputs: List[Tuple[Request, int, bool]]):
        """Fan one step's tokens out to the per-request queues and events (event-loop thread)."""
        for req, tok, finished in outputs:
            req.output_q.put_nowait(StepOutput(tok, req))
            if finished:
                self._live.discard(req)
                req.finished_event.set()

    def _fail_live(self, why: str):
        """The model thread is gone: wake every caller still waiting (event-loop thread). `add_request_and_wait` returns
        with what was generated so far and `request.error` set; streams end."""
        self._dead = why                # from now on _enqueue and the tokenize loop refuse instead of queueing
        live, self._live = list(self._live), set()
        pending, self.untokenized_raw_requests = self.untokenized_raw_requests, []
        for req in live + [r for r, _ in pending]:
            req.error = req.error or why
            req.finished_event.set()
            req.output_q.put_nowait(None)

    # ---- model thread ---------------------------------------------------------------------------------------------
    def _post_undelivered(self):
        """Hand the held-back tokens to the event loop. Called on the model thread: by LlamaModel.forward right after
        a step's kernels were enqueued (`after_launch_hook`), and by `_iterate` when nothing will be launched."""
        if not self._undelivered:
            return
        outputs, self._undelivered = self._undelivered, []
        try:
            self.event_loop.call_soon_threadsafe(self._deliver, outputs)
        except RuntimeError:        # the event loop is gone (shutdown): nobody is listening any more
            pass

    def _drain_inbox(self, wait_s: float = 0.0):
        try:
            if wait_s > 0.0:
                self.scheduler.on_requests_arrival(self._inbox.get(timeout=wait_s))
            while True:
                self.scheduler.on_requests_arrival(self._inbox.get_nowait())
        except queue.Empty:
            pass

    def _iterate(self) -> bool:
        """One scheduling iteration, start to finish, on the calling (model) thread; False when there was nothing
        to do. Reference: engine.py:121-176."""
        self._drain_inbox()
        batch, swap_in, swap_out = self.scheduler.get_next_batch()
        if not batch:
            self._post_undelivered()        # no launch to hide the fan-out behind
            if not swap_in and not swap_out:
                return False
        if swap_out:
            self.model.swap_out_seqs([r.request_id for r in swap_out])
            self.num_swapped_out += len(swap_out)
        if swap_in:
            self.model.swap_in_seqs([r.request_id for r in swap_in])
            self.num_swapped_in += len(swap_in)
        if batch:
            # prefill sequences first (the scheduler orders them so), their whole prompt; decoding ones
            # bring their last token and their length INCLUDING it
            input_ids = [r.prompt_token_ids if r.is_prefill_stage() else [r.output_token_ids[-1]] for r in batch]
            seq_ids = [r.request_id for r in batch]
            decoding_lens = [r.num_tokens() for r in batch if not r.is_prefill_stage()]
            try:
                tokens = self.model.forward(input_ids, seq_ids, decoding_lens)
            finally:
                self._post_undelivered()    # (a data plane without the hook, or a forward that raised before launching)
            self.num_forwards += 1
            outputs, finished = [], []
            for req, tok in zip(batch, tokens):
                req.output_token_ids.append(tok)
                done = req.is_finished()
                outputs.append((req, tok, done))
                if done:
                    finished.append(req.request_id)
            if finished:
                # release KV blocks before anyone is told: a caller that sees "finished" may tear us down
                self.model.free_seqs_resources(finished)
            self._undelivered = outputs
            self.scheduler.on_batch_finish(batch)
        return True

    def _model_loop(self, failed: "asyncio.Future"):
        # Garbage collection around the latency-critical section (EngineConfig tuning `pause_gc_while_serving`): a
        # generation-2 pass over a process that holds an 8 G-parameter model's tensor objects and thousands of request
        # records takes tens of ms — several decode steps. The long-lived heap is frozen out of the collector's working
        # set once, automatic collection is off while the loop runs, and the loop collects where no request waits for it:
        # the young generations whenever it goes idle after work (or every 256 busy iterations), everything every 4096.
        pause_gc = bool(getattr(self.engine_config, "pause_gc_while_serving", False))
        gc_was_enabled = gc.isenabled()
        if pause_gc:
            gc.collect()
            gc.freeze()
            gc.disable()
        busy = total = 0
        try:
            while not self._stop.is_set():
                if self._iterate():
                    busy += 1
                    total += 1
                    if pause_gc and busy >= 256:
                        gc.collect(2 if total >= 4096 else 1)
                        busy, total = 0, (0 if total >= 4096 else total)
                else:
                    if pause_gc and busy:
                        gc.collect(2 if total >= 4096 else 1)
                        busy, total = 0, (0 if total >= 4096 else total)
                    self._drain_inbox(wait_s=0.005)     # idle: block on the inbox (wakes at once on an arrival)
            self._post_undelivered()
        except BaseException as exc:     # noqa: BLE001 — surfaces in start_all_event_loops(), as in the reference
            def report(exc=exc):
                if not failed.done():
                    failed.set_exception(exc)
            try:
                self.event_loop.call_soon_threadsafe(report)
            except RuntimeError:
                pass
        finally:
            if pause_gc:
                gc.unfreeze()
                if gc_was_enabled:
                    gc.enable()

    async def step(self) -> bool:
        """One scheduling iteration (single-stepping for tests and tools; the serving loop is `_model_loop`). Its
        tokens are fanned out before it returns. False when there was nothing to do."""
        if self._model_thread is not None and self._model_thread.is_alive():
            raise RuntimeError("Engine.step() while the serving loop runs: the scheduler belongs to the model thread")

        def one():
            did = self._iterate()
            self._post_undelivered()
            return did
        did = await self._run_on_model_async(one)
        await asyncio.sleep(0)      # the posted fan-out runs before the caller continues
        return did

    async def _main_event_loop(self):
        """Owns the model thread: starts it, re-raises what it dies of, stops it when cancelled."""
        self._stop.clear()
        failed = self.event_loop.create_future()
        thread = threading.Thread(target=self._model_loop, args=(failed,), name="swiftllm-model", daemon=True)
        self._model_thread = thread
        thread.start()
        try:
            await failed
        except BaseException as exc:        # the model thread died (or we are being cancelled): nobody may wait for ever
            self._fail_live(f"engine stopped: {type(exc).__name__}: {exc}")
            raise
        finally:
            self._stop.set()
            self._inbox.put([])     # wake it if it is waiting for arrivals
            # (at most the step in flight) — joined off the event loop: HTTP and the other coroutines keep running
            try:
                await asyncio.shield(self.event_loop.run_in_executor(None, thread.join, 10.0))
            except asyncio.CancelledError:
                pass

    async def start_all_event_loops(self):
        assert self.initialized, "Engine not initialized. Please call `initialize()` before starting the event loop."
        await asyncio.gather(self._tokenize_raw_request_event_loop(), self._main_event_loop())
