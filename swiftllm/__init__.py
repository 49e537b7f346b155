"""`import swiftllm` — alias of the MI355X implementation in `swiftllm_amd`.

Scripts written against the reference package (`swiftllm.EngineConfig`, `swiftllm.LlamaModel`,
`swiftllm.Engine`, `swiftllm.RawRequest`, and submodule paths such as `swiftllm.worker.model` or
`swiftllm.worker.kernels.paged_attn`) resolve to the modules of `swiftllm_amd`: the alias adds no code of
its own, every `swiftllm.x.y` IS `swiftllm_amd.x.y` (same module object).
"""
import importlib
import importlib.abc
import importlib.util
import sys

import swiftllm_amd as _impl
from swiftllm_amd import EngineConfig, LlamaModel, LlamaModelConfig  # noqa: F401

_PREFIX, _REAL = __name__ + ".", _impl.__name__ + "."


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, module):
        self._module = module

    def create_module(self, spec):
        return self._module

    def exec_module(self, module):
        pass

    def get_code(self, fullname):
        # `python -m swiftllm.server.api_server`: runpy runs the code of the real module as __main__
        real = self._module.__name__
        return importlib.util.find_spec(real).loader.get_code(real)


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        try:
            real = importlib.import_module(_REAL + fullname[len(_PREFIX):])
        except ImportError:
            return None
        return importlib.util.spec_from_loader(fullname, _AliasLoader(real), origin=getattr(real, "__file__", None),
                                               is_package=hasattr(real, "__path__"))


sys.meta_path.insert(0, _AliasFinder())
__path__ = []   # a namespace with no files of its own: every submodule comes from the finder above


def __getattr__(name):
    return getattr(_impl, name)
