"""Shared test plumbing.

Two tiers:
  * `-m "not gpu"`: the oracle against the golden vectors frozen from the reference, the host logic,
    and the C-ABI surface (library loads, every declared symbol is exported). No compute calls.
  * `-m gpu`: parity tests proper — the HIP kernels, called through the C ABI, against the oracle on
    the same seeded inputs and against the golden fixtures. Run on an MI355X.
Only tests import `oracle`; the product never does.
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an AMD GPU (gfx950) and libswiftllm_hip.so")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name: str):
    return torch.load(os.path.join(GOLDEN_DIR, name), weights_only=False)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """Build libswiftllm_hip.so if it is missing (hipcc cross-compiles without a GPU)."""
    from swiftllm_amd import _hip
    if not _hip.is_available():
        from swiftllm_amd.csrc import build
        build.build(verbose=False)
    yield


def ulp_diff_fp16(a: torch.Tensor, b: torch.Tensor) -> int:
    """Largest distance in units-in-the-last-place between two fp16/bf16 tensors of the same dtype."""
    ai = a.contiguous().view(torch.int16).to(torch.int32)
    bi = b.contiguous().view(torch.int16).to(torch.int32)
    # map sign-magnitude to a monotonic integer line
    ai = torch.where(ai < 0, -(ai & 0x7FFF), ai)
    bi = torch.where(bi < 0, -(bi & 0x7FFF), bi)
    return int((ai - bi).abs().max()) if ai.numel() else 0
