"""Dense projections. Reference: swiftllm/worker/kernels/linear.py:3-12.

`F.linear` dispatches to hipBLASLt/rocBLAS on ROCm exactly as the reference's call dispatches to
cuBLAS; keeping the same torch op keeps the GEMM numerics identical to the reference run on the
same box. (A hand-written skinny weight-streaming GEMM for decode is SURVEY.md §8f rank 1.)
"""
import torch
import torch.nn.functional as F


def linear(a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """a[T, in] @ w[out, in]^T -> [T, out]."""
    return F.linear(a, w)
