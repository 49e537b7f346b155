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


# ---- deferred normalisation (decode fast path; csrc/rmsnorm.hip: splitk_add_scale_kernel) -----------------------------
_ADD_SCALE_CHUNK = 1024      # kAddScaleChunk: columns per workgroup = granularity of the sums of squares
_MAX_SSQ_PARTS = 8


class RowScalePending:
    """Activations whose RMSNorm scale is still pending: `x` = round(residual * norm_weight) [tokens, hidden], `ssq`
    [parts, tokens] the per-1024-column sums of squares of the residual rows. The consumer projection multiplies its
    fp32 results by 1/sqrt(sum(ssq)/hidden + eps) before rounding (linear_silu_gate / paged_attention_from_qkv_splitk)."""
    __slots__ = ("x", "ssq", "parts", "eps", "hidden")

    def __init__(self, x, ssq: torch.Tensor, parts: int, eps: float, hidden: int = 0):
        # x is None when the consumer of the pending scale never sees the scaled activations as a tensor
        # (kernels/linear.py: linear_splitk_from_splitk keeps them in LDS); `hidden` = their row length
        self.x, self.ssq, self.parts, self.eps = x, ssq, parts, eps
        self.hidden = hidden or (x.shape[1] if x is not None else 0)


def deferred_norm_ok(num_tokens: int, hidden: int, dtype: torch.dtype = torch.bfloat16) -> bool:
    """bfloat16 only: the deferred form stores round(residual * norm_weight) BEFORE the 1/rms is applied. bfloat16 has
    fp32's exponent range, so that intermediate can neither overflow nor go subnormal; in float16 a residual outlier of
    1e4 times a norm weight of 3 is already at 3e4 of 65504, and rows with rms << 1 would lose bits to subnormals —
    neither can happen on the reference's path (rmsnorm.py:59-64 rounds x * rstd * w once). float16 therefore keeps the
    reference's rounding points (fused_add_rmsnorm_from_splitk)."""
    return (dtype == torch.bfloat16 and 0 < num_tokens <= 32 and hidden % _ADD_SCALE_CHUNK == 0
            and hidden // _ADD_SCALE_CHUNK <= _MAX_SSQ_PARTS)


def add_scale_from_splitk(partials, residual_io: torch.Tensor, weight: torch.Tensor, eps: float,
                          unsafe_float16_ok: bool = False) -> RowScalePending:
    """The element-wise half of fused_add_rmsnorm on split-K slabs: residual <- round(sum slabs) + residual (same bits
    as fused_add_rmsnorm_from_splitk); returns round(residual * weight) with the 1/rms pending. Fully parallel over
    rows and columns (the row-wide reduction is what keeps fused_add_rmsnorm at one workgroup per token).
    bfloat16 only (deferred_norm_ok); `unsafe_float16_ok=True` lets kernel tests exercise the float16 instantiation on
    data they know to be in range — the product never passes it."""
    m, n = partials.shape
    _check_rows(residual_io, "residual_io")
    assert residual_io.shape == (m, n) and residual_io.dtype == partials.dtype == weight.dtype
    # shape limits AND the bfloat16-only policy (a direct float16 caller would get the overflow-prone form)
    assert deferred_norm_ok(m, n, partials.dtype) or (unsafe_float16_ok and deferred_norm_ok(m, n)), \
        "deferred RMSNorm: bfloat16, <= 32 tokens, hidden % 1024 == 0"
    parts = n // _ADD_SCALE_CHUNK
    xs = torch.empty((m, n), dtype=partials.dtype, device=residual_io.device)
    ssq = torch.empty((parts, m), dtype=torch.float32, device=residual_io.device)
    _hip.call("swl_splitk_add_scale", _hip.ptr(xs), _hip.ptr(residual_io), _hip.ptr(weight), _hip.ptr(partials.slabs),
              partials.k_splits, _hip.ptr(ssq), m, n, _hip.dtype_code(partials.dtype), _hip.stream())
    return RowScalePending(xs, ssq, parts, eps)
