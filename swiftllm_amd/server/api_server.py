"""HTTP front-end of one engine replica.

    python -m swiftllm_amd.server.api_server --model-path DIR [--port 8000] [engine flags]

`POST /generate` with JSON {prompt | prompt_token_ids, output_len, stream?, decode?} — the request
format of the reference's swiftllm/server/api_server.py:16-84: non-streaming answers
{"output_token_ids": [...]} (or {"output": text} with decode), streaming sends one line per token.
`GET /load` reports outstanding tokens (used by the replica router). Any engine failure takes the
process down (reference api_server.py:114-119) so a supervisor can restart the replica.
"""
import argparse
import asyncio
import os
import traceback

import fastapi
import uvicorn
from fastapi.responses import JSONResponse, StreamingResponse

from swiftllm_amd.engine_config import EngineConfig
from .engine import Engine
from .structs import RawRequest


def _validate_body(body) -> "str | None":
    """Shape checks the engine must never have to survive (it dies on any exception, taking every in-flight
    request with it): types only — ranges (vocabulary, rotary positions, pool size) are the scheduler's
    `why_unservable`, answered with the same 400."""
    if not isinstance(body, dict):
        return "body must be a JSON object"
    n = body.get("output_len")
    if not isinstance(n, int) or isinstance(n, bool):
        return "output_len must be an integer"
    ids = body.get("prompt_token_ids")
    if ids is not None:
        if not isinstance(ids, list) or not all(isinstance(t, int) and not isinstance(t, bool) for t in ids):
            return "prompt_token_ids must be a list of integers"
        if len(ids) == 0:
            return "prompt_token_ids must not be empty"
        if any(t < 0 or t >= 2 ** 31 for t in ids):
            return "prompt_token_ids out of range"
    if not isinstance(body.get("prompt", ""), str):    # also when token ids are given: the handler still touches it
        return "prompt must be a string"
    return None


def build_app(engine: Engine) -> fastapi.FastAPI:
    app = fastapi.FastAPI()
    state = {"outstanding_tokens": 0}

    @app.post("/generate")
    async def generate(req: fastapi.Request):
        try:
            body = await req.json()
        except Exception:     # noqa: BLE001 — malformed JSON is the client's problem, not the engine's
            return JSONResponse({"error": "body must be a JSON object"}, status_code=400)
        problem = _validate_body(body)
        if problem is not None:
            return JSONResponse({"error": problem}, status_code=400)
        raw = RawRequest(body.get("prompt", ""), body["output_len"], body.get("prompt_token_ids"))
        want_text = bool(body.get("decode", False))
        cost = raw.output_len + len(raw.prompt_token_ids or raw.prompt.split())
        state["outstanding_tokens"] += cost
        if body.get("stream", False):
            async def lines():
                try:
                    async for step in engine.add_request_and_stream(raw):
                        if want_text:
                            yield await engine.tokenization_engine.decode([step.token_id]) + "\n"
                        else:
                            yield f"{step.token_id}\n"
                finally:
                    state["outstanding_tokens"] -= cost
            return StreamingResponse(lines(), media_type="text/plain")
        try:
            request, token_ids = await engine.add_request_and_wait(raw)
        finally:
            state["outstanding_tokens"] -= cost
        if request.error is not None:
            # a request the engine can never serve is the client's error (400); an engine whose model thread is gone is
            # the server's (503: a router in front retries on another replica)
            dead = getattr(engine, "_dead", None) is not None and request.error == engine._dead
            return JSONResponse({"error": request.error}, status_code=503 if dead else 400)
        if want_text:
            return JSONResponse({"output": await engine.tokenization_engine.decode(token_ids)})
        return JSONResponse({"output_token_ids": token_ids})

    @app.get("/load")
    async def load():
        return JSONResponse(state)

    return app


async def _serve(args):
    fields = {f for f in EngineConfig.__dataclass_fields__}
    engine = Engine(EngineConfig(**{k: v for k, v in vars(args).items() if k in fields}),
                    piggyback=args.piggyback)
    await engine.initialize()
    server = uvicorn.Server(uvicorn.Config(build_app(engine), host=args.host, port=args.port, log_level="warning"))

    async def guarded_loops():
        try:
            await engine.start_all_event_loops()
        except Exception:     # noqa: BLE001 — a dead engine must not leave a zombie HTTP server behind
            traceback.print_exc()
            os._exit(1)
    await asyncio.gather(server.serve(), guarded_loops())


def main():
    ap = argparse.ArgumentParser(description="swiftllm_amd API server (one replica)")
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--piggyback", action="store_true", help="let decodes ride along with prefill batches")
    EngineConfig.add_cli_args(ap)
    asyncio.run(_serve(ap.parse_args()))


if __name__ == "__main__":
    main()
