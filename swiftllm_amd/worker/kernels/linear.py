"""Dense projections. Reference: swiftllm/worker/kernels/linear.py:3-12.

Default: `torch.nn.functional.linear`, which dispatches to hipBLASLt/rocBLAS on ROCm exactly as the
reference's call dispatches to cuBLAS — same GEMM kernels as the reference run on the same box.

`skinny=True` (EngineConfig.use_skinny_gemm) routes decode-sized calls (M <= 32 tokens) to the
hand-written weight-streaming MFMA kernel `swl_gemm_skinny` (csrc/gemm_skinny.hip): at that size the
projection is pure HBM streaming of the weight matrix, 77 % of all bytes a decode step moves
(SURVEY.md §8f rank 1). Larger M (prefill) stays on the BLAS, which is compute-bound territory.

`linear_splitk` is the same product stopped one step earlier: when the kernel splits K across
workgroups it returns the fp32 partial slabs (`SplitKPartials`) instead of launching the reduce, and a
fused consumer (`fused_add_rmsnorm_from_splitk`, `rotary_embedding_and_store_kvcache_decode_from_splitk`)
sums them — identical bits, one launch fewer per projection.
"""
import torch
import torch.nn.functional as F

from swiftllm_amd import _hip

_SKINNY_MAX_M = 32
_workspaces = {}    # device -> persistent fp32 split-K scratch (fixed address: hipGraph replays use it)


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    ws = _workspaces.get(device)
    if ws is None or ws.numel() * 4 < nbytes:
        # sized once for the widest split projection a LLaMA has (fused qkv / o / down: N <= 16384)
        ws = torch.empty(max(nbytes, 16 * _SKINNY_MAX_M * 16384 * 4) // 4, dtype=torch.float32, device=device)
        _workspaces[device] = ws
    return ws


def _skinny_ok(a: torch.Tensor, w: torch.Tensor) -> bool:
    return (a.is_cuda and a.dim() == 2 and 0 < a.shape[0] <= _SKINNY_MAX_M and a.dtype == w.dtype
            and a.dtype in (torch.float16, torch.bfloat16) and w.is_contiguous() and a.stride(1) == 1
            and w.shape[0] % 32 == 0 and w.shape[1] % 128 == 0 and a.stride(0) % 8 == 0)


def _row_stride(a: torch.Tensor) -> int:
    return a.stride(0) if a.shape[0] > 1 else max(a.stride(0), a.shape[1])


class SplitKPartials:
    """fp32 partial slabs [k_splits][M][N] of a projection whose K was split across workgroups.
    Lives in the shared split-K workspace: it must be consumed (or materialised) before the next
    split projection is launched on the same stream — the layer code does exactly that."""
    __slots__ = ("slabs", "k_splits", "shape", "dtype")

    def __init__(self, slabs: torch.Tensor, k_splits: int, m: int, n: int, dtype: torch.dtype):
        self.slabs, self.k_splits, self.shape, self.dtype = slabs, k_splits, (m, n), dtype

    @property
    def device(self):
        return self.slabs.device

    def materialize(self) -> torch.Tensor:
        """round(sum of slabs) as an ordinary [M, N] tensor (the stand-alone reduce kernel)."""
        m, n = self.shape
        out = torch.empty((m, n), dtype=self.dtype, device=self.slabs.device)
        _hip.call("swl_splitk_reduce", _hip.ptr(out), _hip.ptr(self.slabs), self.k_splits, m, n, n,
                  _hip.dtype_code(self.dtype), _hip.stream())
        return out


def linear(a: torch.Tensor, w: torch.Tensor, skinny: bool = False) -> torch.Tensor:
    """a[T, in] @ w[out, in]^T -> [T, out] (fp32 accumulation, one rounding)."""
    if skinny and _skinny_ok(a, w):
        m, k = a.shape
        n = w.shape[0]
        out = torch.empty((m, n), dtype=a.dtype, device=a.device)
        need = _hip.load().swl_gemm_skinny_workspace_bytes(m, n, k)
        ws = _workspace(a.device, need) if need else None
        _hip.call("swl_gemm_skinny", _hip.ptr(out), _hip.ptr(a), _hip.ptr(w), _hip.ptr(ws),
                  ws.numel() * 4 if ws is not None else 0, m, n, k, _row_stride(a), n, 0,
                  _hip.dtype_code(a.dtype), _hip.stream())
        return out
    return F.linear(a, w)


def linear_splitk(a: torch.Tensor, w: torch.Tensor):
    """Like linear(a, w, skinny=True) but returns SplitKPartials when the kernel splits K (the caller
    hands them to a fused consumer); falls through to `linear` otherwise."""
    if _skinny_ok(a, w):
        m, k = a.shape
        n = w.shape[0]
        ks = _hip.load().swl_gemm_skinny_choose_splits(n, k)
        if ks > 1:
            ws = _workspace(a.device, ks * m * n * 4)
            _hip.call("swl_gemm_skinny_partial", _hip.ptr(ws), ws.numel() * 4, _hip.ptr(a), _hip.ptr(w), m, n, k,
                      _row_stride(a), ks, _hip.dtype_code(a.dtype), _hip.stream())
            return SplitKPartials(ws, ks, m, n, a.dtype)
    return linear(a, w, skinny=True)


def linear_silu_gate(a: torch.Tensor, w_up_gate: torch.Tensor):
    """The FFN's `silu_and_mul_inplace(linear(a, up_gate_proj))[:, :I]` in one launch for decode-sized
    batches: returns [T, I], or None when the shapes do not qualify (the caller then takes the two-op
    path)."""
    if not _skinny_ok(a, w_up_gate) or w_up_gate.shape[0] % 64 != 0:
        return None
    m, k = a.shape
    inter = w_up_gate.shape[0] // 2
    out = torch.empty((m, inter), dtype=a.dtype, device=a.device)
    _hip.call("swl_gemm_skinny_silu_gate", _hip.ptr(out), _hip.ptr(a), _hip.ptr(w_up_gate), m, inter, k,
              _row_stride(a), inter, _hip.dtype_code(a.dtype), _hip.stream())
    return out
