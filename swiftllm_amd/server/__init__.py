"""Control plane: request structs, FCFS scheduler (with optional prefill/decode piggybacking), asyncio
engine, HTTP server and the request-sharded replica router. Pure Python; it drives the data plane only
through LlamaModel.forward / swap_in_seqs / swap_out_seqs / free_seqs_resources, as the reference's
swiftllm/server does (engine.py:129-168)."""
from .structs import RawRequest, Request, StepOutput
from .scheduler import Scheduler, RequestIdManager
from .engine import Engine

__all__ = ["RawRequest", "Request", "StepOutput", "Scheduler", "RequestIdManager", "Engine"]
