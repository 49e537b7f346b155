"""Control plane on CPU: scheduler policy, engine loop (with a fake data plane), HTTP layer, router."""
import asyncio
import types

import pytest

from swiftllm_amd.engine_config import EngineConfig
from swiftllm_amd.server import Engine, RawRequest, Request, Scheduler, RequestIdManager
from swiftllm_amd.server.router import ReplicaRouter, request_cost


def _cfg(**kw):
    base = dict(model_path="", use_dummy=True, block_size=16, gpu_mem_utilization=0.9, num_cpu_blocks=64,
                max_seqs_in_block_table=16, max_blocks_per_seq=64, max_batch_size=4, max_tokens_in_batch=100)
    base.update(kw)
    return EngineConfig(**base)


def _req(prompt_len, output_len=4):
    return Request(RawRequest("", output_len, list(range(prompt_len))))


def test_request_id_manager():
    m = RequestIdManager(3)
    assert [m.get_id(), m.get_id(), m.get_id()] == [0, 1, 2]
    with pytest.raises(RuntimeError, match="max_seqs_in_block_table"):
        m.get_id()
    m.free_ids([1])
    assert m.get_id() == 1


def test_scheduler_fcfs_admission_budgets():
    s = Scheduler(None, _cfg(), num_gpu_blocks=10)
    reqs = [_req(40), _req(40), _req(40), _req(5)]
    s.on_requests_arrival(reqs)
    batch, sin, sout = s.get_next_batch()
    # 40+40 = 80 tokens fit max_tokens_in_batch=100, the third prompt does not — and the short one
    # behind it must NOT overtake (strict FCFS)
    assert batch == reqs[:2] and sin == [] and sout == []
    assert [r.request_id for r in batch] == [0, 1]
    for r in batch:
        r.output_token_ids.append(1)
    s.on_batch_finish(batch)
    batch2, _, _ = s.get_next_batch()
    assert batch2 == reqs[2:4]              # 3+3 blocks running, 3+1 more fit in 10
    assert s.get_next_batch()[0] == s.running_q     # nothing waiting: decode everyone


def test_scheduler_swaps_out_newest_and_back_in_oldest_first():
    s = Scheduler(None, _cfg(max_tokens_in_batch=1000), num_gpu_blocks=4)
    a, b = _req(31, 40), _req(31, 40)      # 2 blocks each: the pool is exactly full
    s.on_requests_arrival([a, b])
    batch, _, _ = s.get_next_batch()
    assert batch == [a, b]
    for r in (a, b):
        r.output_token_ids.append(7)
    s.on_batch_finish(batch)
    batch, sin, sout = s.get_next_batch()   # 32 tokens each: still 2 blocks
    assert (batch, sin, sout) == ([a, b], [], [])
    for r in (a, b):
        r.output_token_ids.append(7)        # 33 tokens -> 3 blocks each > 4 in total
    s.on_batch_finish(batch)
    batch, sin, sout = s.get_next_batch()
    assert batch == [a] and sout == [b] and sin == []
    assert list(s.swapped_q) == [b]
    c = _req(5)
    s.on_requests_arrival([c])              # no new prompt while something is swapped out
    a.output_token_ids.extend([7] * 38)     # a finishes
    s.on_batch_finish([a])
    batch, sin, sout = s.get_next_batch()
    assert sin == [b] and batch == [b] and sout == []
    assert list(s.waiting_q) == [c]
    assert isinstance(sout, list)           # an empty list is falsy: the idle engine can sleep


def test_scheduler_piggyback_mixes_prefill_and_decode():
    s = Scheduler(None, _cfg(max_tokens_in_batch=64), num_gpu_blocks=100, piggyback=True)
    a = _req(10)
    s.on_requests_arrival([a])
    s.get_next_batch()
    a.output_token_ids.append(3)
    s.on_batch_finish([a])
    b = _req(20)
    s.on_requests_arrival([b])
    batch, _, _ = s.get_next_batch()
    assert batch == [b, a]                  # prefill first, then the riding decode
    assert b.is_prefill_stage() and not a.is_prefill_stage()
    plain = Scheduler(None, _cfg(max_tokens_in_batch=64), num_gpu_blocks=100)
    plain.running_q.append(a)
    plain.on_requests_arrival([_req(20)])
    assert len(plain.get_next_batch()[0]) == 1     # the reference's behaviour: prefill alone


class FakeModel:
    """The four calls the engine makes, with deterministic tokens: next = (sum of inputs + position) % 97."""
    num_blocks = 8

    def __init__(self):
        self.model_config = types.SimpleNamespace()
        self.calls = []
        self.freed = []

    def forward(self, input_ids, seq_ids, decoding_lens):
        n_prefill = len(input_ids) - len(decoding_lens)
        assert all(len(x) == 1 for x in input_ids[n_prefill:])
        self.calls.append((n_prefill, len(decoding_lens)))
        lens = [len(x) for x in input_ids[:n_prefill]] + list(decoding_lens)
        return [(sum(ids) + n) % 97 for ids, n in zip(input_ids, lens)]

    def swap_in_seqs(self, ids):
        self.calls.append(("in", list(ids)))

    def swap_out_seqs(self, ids):
        self.calls.append(("out", list(ids)))

    def free_seqs_resources(self, ids):
        self.freed.extend(ids)


def _expected(prompt, n):
    out, last, length = [], None, len(prompt)
    for i in range(n):
        if i == 0:
            tok = (sum(prompt) + length) % 97
        else:
            length += 1
            tok = (last + length) % 97
        out.append(tok)
        last = tok
    return out


@pytest.mark.parametrize("piggyback", [False, True])
def test_engine_generates_streams_and_frees(piggyback):
    async def run():
        model = FakeModel()
        eng = Engine(_cfg(max_batch_size=3), model=model, piggyback=piggyback)
        await eng.initialize()
        loops = asyncio.ensure_future(eng.start_all_event_loops())
        prompts = [[5, 6, 7], [1], [9, 9, 9, 9], [2, 3]]
        waits = [asyncio.ensure_future(eng.add_request_and_wait(RawRequest("", n, p)))
                 for n, p in zip((3, 6, 7), prompts[:3])]

        async def stream(p):
            return [s.token_id async for s in eng.add_request_and_stream(RawRequest("", 4, p))]
        streamed = asyncio.ensure_future(stream(prompts[3]))
        done = await asyncio.wait_for(asyncio.gather(*waits, streamed), timeout=20)
        loops.cancel()
        return model, done
    model, done = asyncio.run(run())
    for (req, toks), p, n in zip(done[:3], [[5, 6, 7], [1], [9, 9, 9, 9]], (3, 6, 7)):
        assert toks == _expected(p, n) and req.is_finished()
    assert done[3] == _expected([2, 3], 4)
    assert len(model.freed) == 4            # every request freed exactly once
    if piggyback:
        assert any(isinstance(c[0], int) and c[0] > 0 and c[1] > 0 for c in model.calls)   # a mixed batch happened
    else:
        assert all(not (isinstance(c[0], int) and c[0] > 0 and c[1] > 0) for c in model.calls)


def test_serving_loop_keeps_the_cyclic_collector_out_of_its_busy_iterations():
    """EngineConfig tuning pause_gc_while_serving (default on): automatic collection is off inside every
    forward the loop runs and restored when the loop stops; with the switch off the loop leaves the collector alone."""
    import gc

    class GcModel(FakeModel):
        def __init__(self):
            super().__init__()
            self.gc_enabled_in_forward = []

        def forward(self, input_ids, seq_ids, decoding_lens):
            self.gc_enabled_in_forward.append(gc.isenabled())
            return super().forward(input_ids, seq_ids, decoding_lens)

    async def run(tuning):
        model = GcModel()
        eng = Engine(_cfg(tuning=tuning), model=model)
        await eng.initialize()
        loops = asyncio.ensure_future(eng.start_all_event_loops())
        await asyncio.wait_for(eng.add_request_and_wait(RawRequest("", 5, [3, 4])), timeout=20)
        loops.cancel()
        try:
            await loops
        except (asyncio.CancelledError, Exception):
            pass
        for _ in range(200):                 # the model thread restores the collector on its way out
            if eng._model_thread is None or not eng._model_thread.is_alive():
                break
            await asyncio.sleep(0.01)
        return model
    assert gc.isenabled()
    model = asyncio.run(run(None))
    assert model.gc_enabled_in_forward and not any(model.gc_enabled_in_forward)
    assert gc.isenabled() and gc.get_freeze_count() == 0
    model = asyncio.run(run(dict(pause_gc_while_serving=False)))
    assert model.gc_enabled_in_forward and all(model.gc_enabled_in_forward)


class HookedModel(FakeModel):
    """A data plane with LlamaModel's `after_launch_hook`: fires it once its "kernels are enqueued", then "waits for
    the GPU" (GIL released) before returning the tokens. Records what the clients had seen at both moments."""

    def __init__(self, seen):
        super().__init__()
        self.after_launch_hook = None
        self.seen = seen            # tokens the streaming client has received so far (appended on the event loop)
        self.log = []

    def forward(self, input_ids, seq_ids, decoding_lens):
        import time
        before = len(self.seen)
        if self.after_launch_hook is not None:
            self.after_launch_hook()
        time.sleep(0.02)            # the step "runs": the event loop has all the time it needs for the fan-out
        self.log.append((before, len(self.seen)))
        return super().forward(input_ids, seq_ids, decoding_lens)


def test_engine_fans_a_step_out_behind_the_next_launch():
    """The tokens of step k reach the client while step k+1 runs (posted by after_launch_hook), not before its launch
    and not after its end; the last token is delivered without a further launch; the model thread ends with the loops."""
    import threading
    seen = []

    async def run():
        model = HookedModel(seen)
        eng = Engine(_cfg(), model=model)
        await eng.initialize()
        assert model.after_launch_hook is not None
        loops = asyncio.ensure_future(eng.start_all_event_loops())

        async def stream():
            async for s in eng.add_request_and_stream(RawRequest("", 5, [3, 4])):
                seen.append(s.token_id)
        await asyncio.wait_for(stream(), timeout=20)
        assert any(t.name == "swiftllm-model" for t in threading.enumerate())
        loops.cancel()
        await asyncio.gather(loops, return_exceptions=True)
        return model
    model = asyncio.run(run())
    assert seen == _expected([3, 4], 5)
    # forward k (k = 0..4) was entered with k-1 tokens delivered (none for k <= 1: token k-1 is posted by ITS launch)
    # and left with k tokens delivered
    assert model.log == [(0, 0), (0, 1), (1, 2), (2, 3), (3, 4)]
    assert not any(t.name == "swiftllm-model" and t.is_alive() for t in threading.enumerate())


def test_engine_surfaces_a_failing_data_plane():
    """An exception on the model thread ends start_all_event_loops() with that exception (the reference's engine dies
    the same way, engine.py:121-176) instead of leaving the server answering nothing."""
    class Broken(FakeModel):
        def forward(self, input_ids, seq_ids, decoding_lens):
            raise RuntimeError("HIP error: the device fell over")

    async def run():
        eng = Engine(_cfg(), model=Broken())
        await eng.initialize()
        loops = asyncio.ensure_future(eng.start_all_event_loops())
        asyncio.ensure_future(eng.add_request_and_wait(RawRequest("", 3, [1, 2])))
        with pytest.raises(RuntimeError, match="the device fell over"):
            await asyncio.wait_for(loops, timeout=20)
    asyncio.run(run())


@pytest.mark.parametrize("piggyback", [False, True])
def test_engine_under_churn_delivers_every_token_in_order(piggyback):
    """40 requests of random lengths arriving in waves, streamed and awaited, on a pool of 8 blocks (swaps happen) and a
    data plane with the launch hook: every stream carries exactly its generation, in order; every request is freed once;
    swap-ins and swap-outs pair up."""
    import random
    rng = random.Random(5)
    shapes = [([rng.randrange(90) for _ in range(rng.randint(1, 30))], rng.randint(1, 12)) for _ in range(40)]

    async def run():
        model = HookedModel([])
        model.num_blocks = 8
        eng = Engine(_cfg(max_batch_size=5), model=model, piggyback=piggyback)
        await eng.initialize()
        loops = asyncio.ensure_future(eng.start_all_event_loops())

        async def stream(p, n):
            return [s.token_id async for s in eng.add_request_and_stream(RawRequest("", n, p))]

        async def wait(p, n):
            return (await eng.add_request_and_wait(RawRequest("", n, p)))[1]
        tasks = []
        for i, (p, n) in enumerate(shapes):
            tasks.append(asyncio.ensure_future((stream if i % 2 else wait)(p, n)))
            if i % 7 == 6:
                await asyncio.sleep(0.05)       # the next wave arrives while the first is being served
        got = await asyncio.wait_for(asyncio.gather(*tasks), timeout=60)
        loops.cancel()
        await asyncio.gather(loops, return_exceptions=True)
        return model, got

    import time
    real_sleep = time.sleep
    time.sleep = lambda s: real_sleep(min(s, 0.0005))       # (HookedModel's 20 ms "GPU time" would make this 10 s)
    try:
        model, got = asyncio.run(run())
    finally:
        time.sleep = real_sleep
    for (p, n), toks in zip(shapes, got):
        assert toks == _expected(p, n)
    assert len(model.freed) == len(shapes) and len(set(model.freed)) <= 16
    outs = sum(len(c[1]) for c in model.calls if c[0] == "out")
    ins = sum(len(c[1]) for c in model.calls if c[0] == "in")
    assert outs == ins and outs > 0         # the small pool did force swapping


def test_engine_idles_without_calling_the_model():
    async def run():
        model = FakeModel()
        eng = Engine(_cfg(), model=model)
        await eng.initialize()
        assert await eng.step() is False
        return model
    assert asyncio.run(run()).calls == []


def test_api_server_generate_endpoint():
    from fastapi.testclient import TestClient
    from swiftllm_amd.server.api_server import build_app

    async def boot():
        eng = Engine(_cfg(), model=FakeModel())
        await eng.initialize()
        return eng
    loop = asyncio.new_event_loop()
    eng = loop.run_until_complete(boot())
    app = build_app(eng)

    @app.on_event("startup")
    async def start_loops():
        eng.event_loop = asyncio.get_running_loop()
        asyncio.ensure_future(eng.start_all_event_loops())
    with TestClient(app) as client:
        r = client.post("/generate", json={"prompt_token_ids": [4, 5], "output_len": 3})
        assert r.status_code == 200 and r.json() == {"output_token_ids": _expected([4, 5], 3)}
        r = client.post("/generate", json={"prompt_token_ids": [8], "output_len": 4, "stream": True})
        assert [int(x) for x in r.text.split()] == _expected([8], 4)
        assert client.get("/load").json() == {"outstanding_tokens": 0}
        r = client.post("/generate", json={"prompt_token_ids": list(range(200)), "output_len": 3})
        assert r.status_code == 400 and "max_tokens_in_batch" in r.json()["error"]
        # malformed bodies are answered 400 and never reach the engine (which dies on any exception)
        for body in ({"prompt_token_ids": [1, 2]}, {"prompt_token_ids": [1, 2], "output_len": "3"},
                     {"prompt_token_ids": [1, "x"], "output_len": 3}, {"prompt_token_ids": [1, -5], "output_len": 3},
                     {"prompt_token_ids": [1, 2 ** 40], "output_len": 3}, {"prompt_token_ids": "12", "output_len": 3},
                     {"prompt": 17, "output_len": 3}, [1, 2, 3],
                     {"prompt_token_ids": [], "prompt": 5, "output_len": 1},          # (was a 500)
                     {"prompt_token_ids": [1, 2], "prompt": 5, "output_len": 1}, {"prompt_token_ids": [], "output_len": 1}):
            r = client.post("/generate", json=body)
            assert r.status_code == 400 and "error" in r.json(), body
        r = client.post("/generate", content=b"{not json", headers={"content-type": "application/json"})
        assert r.status_code == 400
        r = client.post("/generate", json={"prompt_token_ids": [4, 5], "output_len": 3})
        assert r.status_code == 200 and r.json() == {"output_token_ids": _expected([4, 5], 3)}  # still alive


def test_wait_until_ready_fails_fast_when_a_replica_process_is_dead():
    """router.wait_until_ready polls GET /load; a replica whose process already exited must fail the wait at once,
    not after the 30-minute timeout."""
    import subprocess
    import sys
    import time
    from swiftllm_amd.server.router import wait_until_ready
    dead = subprocess.Popen([sys.executable, "-c", "import sys; sys.exit(3)"])
    dead.wait()
    t0 = time.time()
    with pytest.raises(RuntimeError, match="exited with code 3"):
        asyncio.run(wait_until_ready(["http://127.0.0.1:9"], timeout_s=600, procs=[dead]))
    assert time.time() - t0 < 30


def test_replica_router_balances_by_outstanding_tokens():
    r = ReplicaRouter([f"http://x:{i}" for i in range(4)])
    picks = [r.acquire(c) for c in (100, 10, 10, 10, 10, 10)]
    assert picks[:4] == [0, 1, 2, 3]
    assert picks[4] == 1 and picks[5] == 2      # the heavy replica is avoided
    r.release(0, 100)
    assert r.acquire(1) == 0
    assert request_cost({"prompt_token_ids": [1, 2, 3], "output_len": 7}) == 10
    assert request_cost({"prompt": "a b c d", "output_len": 1}) == 5


def test_unservable_requests_are_rejected_not_queued():
    s = Scheduler(None, _cfg(max_tokens_in_batch=64, max_blocks_per_seq=8), num_gpu_blocks=6)
    assert s.why_unservable(_req(10, 4)) is None
    assert "max_tokens_in_batch" in s.why_unservable(_req(65, 1))
    assert "max_blocks_per_seq" in s.why_unservable(_req(60, 100))        # 160 tokens = 10 blocks > 8
    assert "pool" in s.why_unservable(_req(60, 45))                        # 105 tokens = 7 blocks > 6 in the pool
    assert "empty" in s.why_unservable(_req(0, 4))
    # limits of the data plane: vocabulary (ids index the embedding table on the device) and rotary positions
    # (LlamaModel.forward raises past the table: one long request must not take the replica down)
    s.vocab_size, s.max_seq_len = 50, 40
    assert s.why_unservable(_req(10, 4)) is None
    assert "rotary" in s.why_unservable(_req(30, 11))
    bad_id = _req(3, 2)
    bad_id.prompt_token_ids = [1, 50, 2]
    assert "prompt_token_ids" in s.why_unservable(bad_id)
    s.vocab_size = s.max_seq_len = None

    async def run():
        model = FakeModel()
        eng = Engine(_cfg(max_tokens_in_batch=16), model=model)
        await eng.initialize()
        loops = asyncio.ensure_future(eng.start_all_event_loops())
        bad = asyncio.ensure_future(eng.add_request_and_wait(RawRequest("", 3, list(range(17)))))
        good = asyncio.ensure_future(eng.add_request_and_wait(RawRequest("", 3, [1, 2])))

        async def stream_bad():
            return [s async for s in eng.add_request_and_stream(RawRequest("", 3, list(range(40))))]
        (breq, btoks), (greq, gtoks), streamed = await asyncio.wait_for(asyncio.gather(bad, good, stream_bad()), 20)
        loops.cancel()
        return breq, btoks, greq, gtoks, streamed
    breq, btoks, greq, gtoks, streamed = asyncio.run(run())
    assert breq.error and btoks == [] and streamed == []
    assert greq.error is None and gtoks == _expected([1, 2], 3)             # the queue behind it keeps moving


def test_router_relays_to_live_replicas_and_balances():
    """Two fake replicas (real HTTP servers on localhost) behind the router app: JSON and streamed answers are
    relayed unchanged, consecutive requests spread over the replicas, the books return to zero."""
    import socket
    import threading
    import time
    import fastapi
    import uvicorn
    from fastapi.responses import JSONResponse, StreamingResponse
    from fastapi.testclient import TestClient
    from swiftllm_amd.server.router import build_app

    def free_port():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            return s.getsockname()[1]

    served = {}
    servers = []

    def make_replica(tag):
        app = fastapi.FastAPI()

        @app.post("/generate")
        async def generate(req: fastapi.Request):
            body = await req.json()
            served[tag] = served.get(tag, 0) + 1
            n = int(body["output_len"])
            if body.get("stream"):
                async def lines():
                    for i in range(n):
                        await asyncio.sleep(0.01)
                        yield f"{tag}{i}\n"
                return StreamingResponse(lines(), media_type="text/plain")
            await asyncio.sleep(0.6)        # long enough for the next request to see this one outstanding
            return JSONResponse({"output_token_ids": [tag] * n})
        return app

    urls = []
    for tag in (7, 9):
        port = free_port()
        server = uvicorn.Server(uvicorn.Config(make_replica(tag), host="127.0.0.1", port=port, log_level="error"))
        threading.Thread(target=server.run, daemon=True).start()
        servers.append(server)
        urls.append(f"http://127.0.0.1:{port}")
    deadline = time.time() + 10
    while not all(s.started for s in servers):
        assert time.time() < deadline, "fake replicas did not start"
        time.sleep(0.05)
    try:
        router = ReplicaRouter(urls)
        with TestClient(build_app(router)) as client:
            results = []

            def post(n):
                results.append(client.post("/generate", json={"prompt_token_ids": [1, 2, 3], "output_len": n}).json())
            threads = [threading.Thread(target=post, args=(n,)) for n in (3, 4)]
            for t in threads:
                t.start()
                time.sleep(0.15)
            for t in threads:
                t.join()
            assert sorted(len(r["output_token_ids"]) for r in results) == [3, 4]
            assert served == {7: 1, 9: 1}                       # the second request avoided the busy replica
            r = client.post("/generate", json={"prompt_token_ids": [5], "output_len": 3, "stream": True})
            assert r.text.split() in (["70", "71", "72"], ["90", "91", "92"])
            assert client.get("/load").json() == {"outstanding_tokens": [0, 0]}
    finally:
        for s in servers:
            s.should_exit = True


@pytest.mark.gpu
def test_router_in_front_of_two_real_replicas_on_one_gpu(tmp_path):
    """two real `api_server` replicas (own process, own weights, own KV pool — here both on the one
    GPU of the box, each taking a slice of its memory) behind the router app. Requests carrying token ids are spread
    over both replicas, the answers are what a LlamaModel of the same weights generates offline, the books return to
    zero. (On an 8-GPU node `python -m swiftllm_amd.server.router --num-replicas 8` does the same with one GPU each.)"""
    import json
    import os
    import socket
    import subprocess
    import sys
    import time
    import urllib.request
    from concurrent.futures import ThreadPoolExecutor
    import torch
    from fastapi.testclient import TestClient
    from oracle import synth
    from swiftllm_amd.server.router import ReplicaRouter, build_app

    def free_port():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            return s.getsockname()[1]

    cfg = synth.make_config(num_hidden_layers=2, hidden_size=256, num_attention_heads=4, num_key_value_heads=2,
                            intermediate_size=512, vocab_size=512, max_position_embeddings=512)
    synth.write_model_dir(str(tmp_path), cfg, synth.make_state_dict(cfg, seed=3, dtype=torch.float16))
    ports = [free_port(), free_port()]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=os.path.dirname(os.path.dirname(__file__)))
    urls = [f"http://127.0.0.1:{p}" for p in ports]
    procs = []

    def start(p, util):
        procs.append(subprocess.Popen(
            [sys.executable, "-m", "swiftllm_amd.server.api_server", "--port", str(p), "--model-path", str(tmp_path),
             "--gpu-mem-utilization", str(util), "--num-cpu-blocks", "8", "--max-seqs-in-block-table", "64",
             "--max-blocks-per-seq", "64", "--max-batch-size", "8", "--max-tokens-in-batch", "512"], env=env,
            stdout=subprocess.DEVNULL, stderr=open(tmp_path / f"replica_{p}.err", "w")))

    def wait_ready(u):     # GET /load answers once weights are loaded and the KV pool is profiled
        deadline = time.time() + 240
        while True:
            try:
                with urllib.request.urlopen(u + "/load", timeout=2) as r:
                    if r.status == 200:
                        return
            except OSError:
                pass
            assert all(q.poll() is None for q in procs), "a replica died during start-up:\n" + "\n".join(
                open(tmp_path / f"replica_{q}.err").read()[-1500:] for q in ports)
            assert time.time() < deadline, "replicas did not come up"
            time.sleep(0.5)

    try:
        # gpu_mem_utilization bounds what the DEVICE may have in use when a replica sizes its KV pool; sharing one GPU
        # the second replica has to be given room above what the first already holds (one GPU each: not an issue)
        start(ports[0], 0.08)
        wait_ready(urls[0])
        start(ports[1], 0.2)
        wait_ready(urls[1])
        router = ReplicaRouter(urls)
        client = TestClient(build_app(router))
        g = torch.Generator().manual_seed(1)
        prompts = [torch.randint(0, cfg["vocab_size"], (n,), generator=g).tolist() for n in (5, 40, 17, 33, 8, 64)]

        def ask(ids):
            r = client.post("/generate", json={"prompt_token_ids": ids, "output_len": 12})
            assert r.status_code == 200, r.text
            return r.json()["output_token_ids"]

        with ThreadPoolExecutor(max_workers=len(prompts)) as ex:
            answers = list(ex.map(ask, prompts))
        assert all(len(a) == 12 for a in answers)
        loads = [json.loads(urllib.request.urlopen(u + "/load", timeout=5).read()) for u in urls]
        assert router.outstanding == [0, 0] and all(v.get("outstanding_tokens", 0) == 0 for v in loads)
        # the same weights offline: greedy decoding is deterministic, whichever replica served the request
        from swiftllm_amd import EngineConfig, LlamaModel
        model = LlamaModel(EngineConfig(model_path=str(tmp_path), use_dummy=False, block_size=16, gpu_mem_utilization=0.2,
                                        num_cpu_blocks=8, max_seqs_in_block_table=16, max_blocks_per_seq=16,
                                        max_batch_size=8, max_tokens_in_batch=512))
        model.load_weights()
        model.init_kvcache_and_swap(64)
        for sid, (ids, got) in enumerate(zip(prompts, answers)):
            toks = model.forward([ids], [sid], [])
            out, n = [toks[0]], len(ids)
            for _ in range(11):
                n += 1
                out.append(model.forward([[out[-1]]], [sid], [n])[0])
            assert out == got, (sid, out, got)
    finally:
        for p in procs:
            p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()


def test_waiters_wake_up_when_the_model_thread_dies():
    """a forward() that raises kills the model thread; callers of add_request_and_wait / add_request_and_stream
    must not hang — they return with `request.error` set — and start_all_event_loops re-raises the failure."""
    class Exploding(FakeModel):
        def forward(self, input_ids, seq_ids, decoding_lens):
            if decoding_lens:
                raise RuntimeError("HIP error: boom")
            return super().forward(input_ids, seq_ids, decoding_lens)

    async def scenario():
        engine = Engine(_cfg(max_batch_size=3), model=Exploding())
        await engine.initialize()
        loops = asyncio.ensure_future(engine.start_all_event_loops())
        waiter = asyncio.ensure_future(engine.add_request_and_wait(RawRequest("", 4, [1, 2, 3])))

        async def stream():
            return [o.token_id async for o in engine.add_request_and_stream(RawRequest("", 4, [4, 5]))]
        streamer = asyncio.ensure_future(stream())
        req, toks = await asyncio.wait_for(waiter, 20)
        got = await asyncio.wait_for(streamer, 20)
        assert req.error and "boom" in req.error and len(toks) < 4 and len(got) < 4
        # a request that arrives AFTER the model thread died is refused at once, not parked for ever ...
        late, late_toks = await asyncio.wait_for(engine.add_request_and_wait(RawRequest("", 4, [9, 9])), 5)
        assert late.error and "boom" in late.error and late_toks == [] and not engine._live
        assert [o async for o in engine.add_request_and_stream(RawRequest("", 2, [7]))] == []
        with pytest.raises(RuntimeError, match="boom"):
            await asyncio.wait_for(loops, 20)
        return engine
    engine = asyncio.run(scenario())
    # ... and the HTTP face answers 503 (the server's fault: a router retries elsewhere), not the client's 400
    from starlette.testclient import TestClient
    from swiftllm_amd.server.api_server import build_app
    r = TestClient(build_app(engine)).post("/generate", json=dict(prompt_token_ids=[1, 2], output_len=2))
    assert r.status_code == 503 and "boom" in r.json()["error"]
