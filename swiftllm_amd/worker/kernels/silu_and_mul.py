"""SiLU-gate operator. Reference: swiftllm/worker/kernels/silu_and_mul.py:25-34."""
import torch

from swiftllm_amd import _hip


def silu_and_mul_inplace(x: torch.Tensor):
    """x[:, :I] <- x[:, :I] * silu(x[:, I:]) for x = [tokens, 2*I] (up first, gate second)."""
    _hip.require_gpu_tensor(x, "x")
    assert x.dim() == 2 and x.is_contiguous() and x.shape[1] % 2 == 0
    _hip.call("swl_silu_mul", _hip.ptr(x), x.shape[0], x.shape[1] // 2, _hip.dtype_code(x.dtype),
              _hip.stream())
