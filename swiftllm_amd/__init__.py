"""swiftllm_amd — an MI355X (gfx950) native implementation of swiftLLM's data plane.

Public names match the reference package (swiftllm/__init__.py:1-9): `EngineConfig`,
`LlamaModel`, and — imported lazily because they pull in the serving stack — `Engine`,
`RawRequest`.
"""
from swiftllm_amd.engine_config import EngineConfig
from swiftllm_amd.model_config import LlamaModelConfig
from swiftllm_amd.worker.model import LlamaModel

__all__ = ["EngineConfig", "LlamaModelConfig", "LlamaModel", "Engine", "RawRequest"]


def __getattr__(name):
    if name in ("Engine", "RawRequest"):
        from swiftllm_amd import server
        return getattr(server, name)
    raise AttributeError(f"module 'swiftllm_amd' has no attribute {name!r}")
