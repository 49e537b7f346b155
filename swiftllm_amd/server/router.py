"""Request-sharded data parallelism for serving: N independent replicas on the N GPUs of one node.

    python -m swiftllm_amd.server.router --num-replicas 8 --model-path DIR [--port 8000] [engine flags]

Each replica is an `api_server` process pinned to one GPU (HIP_VISIBLE_DEVICES=i, so it sees its GPU as
device 0 like a single-GPU run) with its own weights, KV pool and scheduler. This process only routes:
a request goes to the replica with the fewest outstanding tokens and stays there for its whole life —
no KV migration, no collective, nothing on xGMI (the reference has no multi-GPU mode).
"""
import argparse
import asyncio
import os
import subprocess
import sys
from typing import List

from swiftllm_amd import dp


class ReplicaRouter:
    """Least-outstanding-tokens routing over a fixed set of replica URLs (pure bookkeeping)."""

    def __init__(self, urls: List[str]):
        self.urls = list(urls)
        self.outstanding = [0] * len(urls)

    def acquire(self, cost: int) -> int:
        i = dp.least_loaded(self.outstanding)
        self.outstanding[i] += cost
        return i

    def release(self, i: int, cost: int):
        self.outstanding[i] -= cost


def request_cost(body: dict) -> int:
    ids = body.get("prompt_token_ids")
    return int(body.get("output_len", 0)) + (len(ids) if ids else len(str(body.get("prompt", "")).split()))


def build_app(router: ReplicaRouter):
    import aiohttp
    import fastapi
    from fastapi.responses import JSONResponse, StreamingResponse
    app = fastapi.FastAPI()
    no_limit = aiohttp.ClientTimeout(total=None)    # a queued long generation may take longer than aiohttp's 300 s

    @app.post("/generate")
    async def generate(req: fastapi.Request):
        body = await req.json()
        cost = request_cost(body)
        i = router.acquire(cost)
        url = router.urls[i] + "/generate"
        if body.get("stream", False):
            async def relay():
                try:
                    async with aiohttp.ClientSession(timeout=no_limit) as s, s.post(url, json=body) as r:
                        async for chunk in r.content.iter_any():
                            yield chunk
                finally:
                    router.release(i, cost)
            return StreamingResponse(relay(), media_type="text/plain")
        try:
            async with aiohttp.ClientSession(timeout=no_limit) as s, s.post(url, json=body) as r:
                return JSONResponse(await r.json(), status_code=r.status)
        finally:
            router.release(i, cost)

    @app.get("/load")
    async def load():
        return JSONResponse({"outstanding_tokens": router.outstanding})

    return app


def spawn_replicas(num: int, base_port: int, passthrough: List[str]) -> List[subprocess.Popen]:
    procs = []
    for i in range(num):
        env = dict(os.environ, HIP_VISIBLE_DEVICES=str(i), HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "swiftllm_amd.server.api_server", "--port", str(base_port + 1 + i)] + passthrough
        cpus = dp.cpus_for_local_rank(i, num)      # NUMA-local cores of GPU i, disjoint from the other replicas'

        def pin(cpus=cpus):
            try:
                os.sched_setaffinity(0, cpus)
            except OSError:
                pass
        procs.append(subprocess.Popen(cmd, env=env, preexec_fn=pin))
    return procs


async def wait_until_ready(urls: List[str], timeout_s: float = 1800.0, procs=None) -> None:
    """Poll every replica's GET /load until it answers (weights loaded, KV pool profiled). `procs` (the replicas'
    subprocess.Popen objects, same order as `urls`): a replica whose process has exited fails the wait at once
    instead of being polled until the timeout."""
    import aiohttp
    deadline = asyncio.get_event_loop().time() + timeout_s
    async with aiohttp.ClientSession(timeout=aiohttp.ClientTimeout(total=5)) as s:
        for i, url in enumerate(urls):
            while True:
                if procs is not None and procs[i].poll() is not None:
                    raise RuntimeError(f"replica {url} exited with code {procs[i].returncode} before it came up")
                try:
                    async with s.get(url + "/load") as r:
                        if r.status == 200:
                            break
                except (aiohttp.ClientError, asyncio.TimeoutError):
                    pass
                if asyncio.get_event_loop().time() > deadline:
                    raise RuntimeError(f"replica {url} did not come up within {timeout_s:.0f} s")
                await asyncio.sleep(0.5)


def main():
    ap = argparse.ArgumentParser(description="swiftllm_amd replica router (request-sharded DP)")
    ap.add_argument("--num-replicas", type=int, default=8)
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8000)
    args, passthrough = ap.parse_known_args()
    procs = spawn_replicas(args.num_replicas, args.port, passthrough)
    router = ReplicaRouter([f"http://127.0.0.1:{args.port + 1 + i}" for i in range(args.num_replicas)])
    try:
        import uvicorn
        asyncio.run(wait_until_ready(router.urls, procs=procs))      # accept traffic only when every replica can serve it
        uvicorn.run(build_app(router), host=args.host, port=args.port, log_level="warning")
    finally:
        for p in procs:
            p.terminate()


if __name__ == "__main__":
    main()
