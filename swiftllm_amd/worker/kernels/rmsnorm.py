"""RMSNorm operators. Reference: swiftllm/worker/kernels/rmsnorm.py:26-37 and :67-89."""
import torch

from swiftllm_amd import _hip


def _check_rows(x: torch.Tensor, name: str):
    _hip.require_gpu_tensor(x, name)
    assert x.dim() == 2 and x.is_contiguous(), f"{name} must be a contiguous [tokens, hidden] tensor"


def rmsnorm_inplace(input_and_output: torch.Tensor, weight: torch.Tensor, eps: float):
    """x <- x * rsqrt(mean(x^2) + eps) * weight, in place."""
    _check_rows(input_and_output, "input_and_output")
    assert weight.is_contiguous() and weight.dtype == input_and_output.dtype
    _hip.call("swl_rmsnorm", _hip.ptr(input_and_output), _hip.ptr(weight), eps,
              input_and_output.shape[0], input_and_output.shape[1],
              _hip.dtype_code(input_and_output.dtype), _hip.stream())


def fused_add_rmsnorm_inplace(input_and_output: torch.Tensor, residual_io: torch.Tensor,
                              weight: torch.Tensor, eps: float):
    """residual <- x + residual; x <- rmsnorm(residual) * weight (both in place)."""
    _check_rows(input_and_output, "input_and_output")
    _check_rows(residual_io, "residual_io")
    assert residual_io.shape == input_and_output.shape
    assert residual_io.dtype == input_and_output.dtype == weight.dtype
    assert weight.is_contiguous()
    _hip.call("swl_fused_add_rmsnorm", _hip.ptr(input_and_output), _hip.ptr(residual_io),
              _hip.ptr(weight), eps, input_and_output.shape[0], input_and_output.shape[1],
              _hip.dtype_code(input_and_output.dtype), _hip.stream())


def fused_add_rmsnorm_from_splitk(partials, residual_io: torch.Tensor, weight: torch.Tensor,
                                  eps: float) -> torch.Tensor:
    """fused_add_rmsnorm_inplace whose input x is still the split-K partial slabs of the projection
    that produced it (kernels/linear.py: SplitKPartials): residual <- round(sum slabs) + residual;
    returns x = rmsnorm(residual) * weight as a fresh [tokens, hidden] tensor. Same bits as reducing
    first, one launch fewer."""
    m, n = partials.shape
    _check_rows(residual_io, "residual_io")
    assert residual_io.shape == (m, n) and residual_io.dtype == partials.dtype == weight.dtype
    out = torch.empty((m, n), dtype=partials.dtype, device=residual_io.device)
    _hip.call("swl_splitk_fused_add_rmsnorm", _hip.ptr(out), _hip.ptr(residual_io), _hip.ptr(weight), eps,
              _hip.ptr(partials.slabs), partials.k_splits, m, n, _hip.dtype_code(partials.dtype),
              _hip.stream())
    return out
