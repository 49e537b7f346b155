"""Dense projections. Reference: swiftllm/worker/kernels/linear.py:3-12.

Default: `torch.nn.functional.linear`, which dispatches to hipBLASLt/rocBLAS on ROCm exactly as the
reference's call dispatches to cuBLAS — same GEMM kernels as the reference run on the same box.

`skinny=True` (EngineConfig.use_skinny_gemm) routes decode-sized calls (M <= 32 tokens) to the
hand-written weight-streaming MFMA kernel `swl_gemm_skinny` (csrc/gemm_skinny.hip): at that size the
projection is pure HBM streaming of the weight matrix, 77 % of all bytes a decode step moves
(SURVEY.md §8f rank 1). Larger M (prefill) stays on the BLAS, which is compute-bound territory.

Weights that carry a packed twin (`pack_weight`: a copy in MFMA-fragment order made at load time,
EngineConfig.pack_decode_weights) are streamed from that copy — same bits, 6-9 % faster (long sequential DRAM
bursts, no LDS transpose) — and additionally serve decode batches of 33..64 tokens (`swl_gemm_packed_mid`: 2 blocks of
32 tokens share every weight fragment) and, where measured faster than the library, 65..256 (`swl_gemm_packed_wide`).
For <= 32 tokens in bfloat16, o_proj / down_proj can finish their rows inside the workgroup (`linear_rows_add`) and the
projection after them normalises the raw residual rows on the fly (`linear_splitk_nf`, `linear_silu_gate_nf`).

`linear_splitk` is the same product stopped one step earlier: when the kernel splits K across
workgroups it returns the fp32 partial slabs (`SplitKPartials`) instead of launching the reduce, and a
fused consumer (`fused_add_rmsnorm_from_splitk`, `rotary_embedding_and_store_kvcache_decode_from_splitk`)
sums them — identical bits, one launch fewer per projection.
"""
import torch
import torch.nn.functional as F

from swiftllm_amd import _hip
from . import route_tune

_SKINNY_MAX_M = 32
_workspaces = {}    # device -> persistent fp32 split-K scratch (fixed address: hipGraph replays use it)


_retired = []       # outgrown workspaces stay allocated: captured hipGraphs may still replay against them


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    ws = _workspaces.get(device)
    if ws is None or ws.numel() * 4 < nbytes:
        if ws is not None:
            _retired.append(ws)
        # sized once for the widest split projection a LLaMA has (fused qkv / o / down: N <= 16384, <= 16 slabs of 32
        # tokens, or the medium-batch equivalent): 64 MiB
        ws = torch.empty(max(nbytes, 64 << 20) // 4, dtype=torch.float32, device=device)
        _workspaces[device] = ws
    return ws


def _skinny_ok(a: torch.Tensor, w: torch.Tensor) -> bool:
    return (a.is_cuda and a.dim() == 2 and 0 < a.shape[0] <= _SKINNY_MAX_M and a.dtype == w.dtype
            and a.dtype in (torch.float16, torch.bfloat16) and w.is_contiguous() and a.stride(1) == 1
            and w.shape[0] % 32 == 0 and w.shape[1] % 128 == 0 and a.stride(0) % 8 == 0)


def _row_stride(a: torch.Tensor) -> int:
    return a.stride(0) if a.shape[0] > 1 else max(a.stride(0), a.shape[1])


def pack_weight(w: torch.Tensor) -> torch.Tensor:
    """A copy of w[N, K] in MFMA-fragment order (csrc/gemm_skinny.hip, PACKED ring kernel), attached to `w` as
    `w._swl_packed` so that `linear(..., skinny=True)` and its split-K / SiLU-gate variants stream the packed copy
    for decode-sized calls. Done once per weight at load time; call again after changing `w` in place."""
    assert w.is_cuda and w.dim() == 2 and w.is_contiguous() and w.shape[0] % 32 == 0 and w.shape[1] % 128 == 0
    wp = torch.empty_like(w)
    _hip.call("swl_gemm_pack_weight", _hip.ptr(wp), _hip.ptr(w), w.shape[0], w.shape[1], _hip.dtype_code(w.dtype),
              _hip.stream())
    w._swl_packed = wp
    return wp


def packable(w) -> bool:
    return (isinstance(w, torch.Tensor) and w.is_cuda and w.dim() == 2 and w.is_contiguous()
            and w.dtype in (torch.float16, torch.bfloat16) and w.shape[0] % 32 == 0 and w.shape[1] % 128 == 0)


def _packed_of(w: torch.Tensor):
    return getattr(w, "_swl_packed", None)


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


_MID_MAX_M = 64


def _mid_ok(a: torch.Tensor, w: torch.Tensor) -> bool:
    """Medium decode batches (32 < M <= 64) on a packed weight: 2 blocks of 32 tokens share every weight fragment
    (swl_gemm_packed_mid). Measured against hipBLASLt (tools/gemm_micro.py --m 48/64): 20-30 % faster over a layer.
    (65..256 tokens: `_wide_ok`, csrc/gemm_wide.hip.)"""
    m = a.shape[0] if a.dim() == 2 else 0
    if not (_SKINNY_MAX_M < m <= _MID_MAX_M) or _packed_of(w) is None:
        return False
    return a.is_cuda and a.dtype == w.dtype and a.stride(1) == 1 and a.stride(0) % 8 == 0


_WIDE_MAX_M = 256


def _launch_wide(a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    m, k = a.shape
    n = w.shape[0]
    out = torch.empty((m, n), dtype=a.dtype, device=a.device)
    need = _hip.load().swl_gemm_packed_wide_workspace_bytes(m, n, k)
    ws = _workspace(a.device, need) if need else None
    _hip.call("swl_gemm_packed_wide", _hip.ptr(out), _hip.ptr(a), _hip.ptr(_packed_of(w)), _hip.ptr(ws),
              ws.numel() * 4 if ws is not None else 0, m, n, k, _row_stride(a), n, 0, 0, _hip.dtype_code(a.dtype),
              _hip.stream())
    return out


def _launch_wide_silu(a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    m, k = a.shape
    inter = w.shape[0] // 2
    out = torch.empty((m, inter), dtype=a.dtype, device=a.device)
    _hip.call("swl_gemm_packed_wide_silu_gate", _hip.ptr(out), _hip.ptr(a), _hip.ptr(_packed_of(w)), m, inter, k,
              _row_stride(a), inter, 0, _hip.dtype_code(a.dtype), _hip.stream())
    return out


def _wide_ok(a: torch.Tensor, w: torch.Tensor) -> bool:
    """Large decode batches (64 < M <= 256) on a packed weight: swl_gemm_packed_wide (csrc/gemm_wide.hip) — up to 8 blocks
    of 32 tokens share every weight fragment, x^T shared by the workgroup through LDS. Which projections it serves is a
    measured policy, decided once per deployment and only LOOKED UP here: kernels/route_tune.py (the r04 table for the
    shapes it was measured on; `tune_wide_routes` at load time for any other)."""
    m = a.shape[0] if a.dim() == 2 else 0
    if not (64 < m <= _WIDE_MAX_M) or _packed_of(w) is None:
        return False
    n, k = w.shape
    if not (a.is_cuda and a.dtype == w.dtype and a.stride(1) == 1 and a.stride(0) % 8 == 0 and k % 64 == 0 and n % 32 == 0
            and m * _row_stride(a) < (1 << 31)):
        return False
    return route_tune.decide(m, n, k, a.dtype, False)


def _wide_silu_ok(a: torch.Tensor, w: torch.Tensor) -> bool:
    m = a.shape[0] if a.dim() == 2 else 0
    if not (64 < m <= _WIDE_MAX_M) or _packed_of(w) is None or w.shape[0] % 64:
        return False
    if not (a.is_cuda and a.dtype == w.dtype and a.stride(1) == 1 and a.stride(0) % 8 == 0 and w.shape[1] % 64 == 0
            and m * _row_stride(a) < (1 << 31)):
        return False
    return route_tune.decide(m, w.shape[0], w.shape[1], a.dtype, True)


def tune_wide_routes(weights, up_gate_weights, dtype: torch.dtype, device, model_path: str) -> dict:
    """Decide, ONCE per deployment and before any request, which side serves every projection class of this model at every
    32-token bucket of 65..256 tokens (route_tune.prepare). `weights`: the packed projection tensors a decode step
    streams; `up_gate_weights`: those that also have the SiLU-gate form. Classes the r04 table covers cost nothing;
    others are timed here (or read from the table the first replica of this node wrote)."""
    rep = {}
    for w in weights:
        if _packed_of(w) is not None and w.shape[1] % 64 == 0 and w.shape[0] % 32 == 0:
            rep.setdefault((w.shape[0], w.shape[1], False), w)
    for w in up_gate_weights:
        if _packed_of(w) is not None and w.shape[1] % 64 == 0 and w.shape[0] % 64 == 0:
            rep.setdefault((w.shape[0], w.shape[1], True), w)

    def measure(n, k, silu, bucket):
        w = rep[(n, k, silu)]
        a = torch.randn((bucket, k), dtype=torch.float32, device=device).to(dtype)
        if silu:
            def library():
                r = F.linear(a, w)
                _hip.call("swl_silu_mul", _hip.ptr(r), bucket, n // 2, _hip.dtype_code(dtype), _hip.stream())
            ours = lambda: _launch_wide_silu(a, w)     # noqa: E731
        else:
            library = lambda: F.linear(a, w)            # noqa: E731
            ours = lambda: _launch_wide(a, w)           # noqa: E731
        return route_tune.wins(route_tune.time_us(ours), route_tune.time_us(library))
    return route_tune.prepare(rep.keys(), dtype, device, model_path, measure)


# (the r04 names, kept for tests/test_host_logic.py: the measured table itself)
_wide_wins = route_tune.table_wide_wins
_wide_silu_wins = route_tune.table_wide_silu_wins


def linear(a: torch.Tensor, w: torch.Tensor, skinny: bool = False) -> torch.Tensor:
    """a[T, in] @ w[out, in]^T -> [T, out] (fp32 accumulation, one rounding)."""
    if skinny and _wide_ok(a, w):
        return _launch_wide(a, w)
    if skinny and _mid_ok(a, w):
        m, k = a.shape
        n = w.shape[0]
        out = torch.empty((m, n), dtype=a.dtype, device=a.device)
        ks = _hip.load().swl_gemm_packed_mid_choose_splits(m, n, k)
        ws = _workspace(a.device, ks * m * n * 4 if ks > 1 else 0)
        _hip.call("swl_gemm_packed_mid", _hip.ptr(out), _hip.ptr(a), _hip.ptr(_packed_of(w)), _hip.ptr(ws),
                  ws.numel() * 4, m, n, k, _row_stride(a), n, ks, _hip.dtype_code(a.dtype), _hip.stream())
        return out
    if skinny and _skinny_ok(a, w):
        m, k = a.shape
        n = w.shape[0]
        out = torch.empty((m, n), dtype=a.dtype, device=a.device)
        need = _hip.load().swl_gemm_skinny_workspace_bytes(m, n, k)
        ws = _workspace(a.device, need) if need else None
        wp = _packed_of(w)
        _hip.call("swl_gemm_skinny_packed" if wp is not None else "swl_gemm_skinny", _hip.ptr(out), _hip.ptr(a),
                  _hip.ptr(wp if wp is not None else w), _hip.ptr(ws),
                  ws.numel() * 4 if ws is not None else 0, m, n, k, _row_stride(a), n, 0,
                  _hip.dtype_code(a.dtype), _hip.stream())
        return out
    return _blas_linear(a, w)


# hipBLASLt's cost per token is far from flat in the token count M (Llama-3-8B widths, bf16, MI355X,
# profiles/r03_prefill_gemm_m_sweep.md): 1.52 / 1.58 / 1.57 PF at M = 4096 / 8192 / 16384 and 1.54-1.56 PF on the multiples
# of 4096 above, but 1.06-1.14 PF from 4097 to 4608 (the down projection alone 708 instead of 295 us), 1.16-1.2 PF around
# 5.5k, 6.5k, 7.5k, 1.26 PF at 10240. A prompt pass (or a piggybacked batch: 4 x 1024 prompt tokens + 28 decodes = 4124
# rows ran 24 % SLOWER than the two forwards apart) must not depend on where its token count happens to fall. Above
# 4096 rows the product is therefore taken in row blocks of sizes the library is good at, each block one BLAS call
# writing its rows of the result in place: from 16384 rows on, the largest multiple of 4096 in one call (the plateau);
# below, 8192 while that many rows remain, then 4096; then the remainder (< 4096 rows, a range without cliffs). Rows are
# independent, so this is the same product up to the order of a row's K-sum, which already depended on which rows shared
# a launch.
_BLAS_ROW_BLOCKS = (8192, 4096)
_BLAS_PLATEAU_ROWS = 16384


def _blas_linear(a: torch.Tensor, w: torch.Tensor, row_blocks=None, plateau=None) -> torch.Tensor:
    blocks = _BLAS_ROW_BLOCKS if row_blocks is None else row_blocks
    if a.dim() != 2 or not blocks or a.shape[0] <= min(blocks):
        return F.linear(a, w)
    plateau = _BLAS_PLATEAU_ROWS if plateau is None else plateau
    m, unit = a.shape[0], min(blocks)
    if m >= plateau and m % unit == 0:
        return F.linear(a, w)
    out = torch.empty((m, w.shape[0]), dtype=a.dtype, device=a.device)
    wt = w.t()
    start = 0
    if m >= plateau:
        start = m // unit * unit
        torch.mm(a[:start], wt, out=out[:start])
    for size in sorted(blocks, reverse=True):
        while m - start >= size:
            torch.mm(a[start:start + size], wt, out=out[start:start + size])
            start += size
    if start < m:
        torch.mm(a[start:], wt, out=out[start:])
    return out


def linear_splitk(a: torch.Tensor, w: torch.Tensor, always: bool = False):
    """Like linear(a, w, skinny=True) but returns SplitKPartials when the kernel splits K (the caller
    hands them to a fused consumer); falls through to `linear` otherwise. `always`: also return the
    partial form (a single fp32 slab) when K is not split — for consumers that only take slabs."""
    if _wide_ok(a, w):                          # large batch: slabs for the add+norm / slab-fed attention consumers
        m, k = a.shape
        n = w.shape[0]
        ks = _hip.load().swl_gemm_packed_wide_choose_splits(m, n, k)
        if ks > 1 or (always and ks == 1):
            ws = _workspace(a.device, ks * m * n * 4)
            _hip.call("swl_gemm_packed_wide_partial", _hip.ptr(ws), ws.numel() * 4, _hip.ptr(a), _hip.ptr(_packed_of(w)),
                      m, n, k, _row_stride(a), 0, ks, _hip.dtype_code(a.dtype), _hip.stream())
            return SplitKPartials(ws, ks, m, n, a.dtype)
        return linear(a, w, skinny=True)
    if _mid_ok(a, w):                           # medium batch on a packed weight: same contract, own kernel
        m, k = a.shape
        n = w.shape[0]
        ks = _hip.load().swl_gemm_packed_mid_choose_splits(m, n, k)
        if ks > 1 or (always and ks == 1):
            ws = _workspace(a.device, ks * m * n * 4)
            _hip.call("swl_gemm_packed_mid_partial", _hip.ptr(ws), ws.numel() * 4, _hip.ptr(a), _hip.ptr(_packed_of(w)),
                      m, n, k, _row_stride(a), ks, _hip.dtype_code(a.dtype), _hip.stream())
            return SplitKPartials(ws, ks, m, n, a.dtype)
        return linear(a, w, skinny=True)
    if _skinny_ok(a, w):
        m, k = a.shape
        n = w.shape[0]
        wp = _packed_of(w)
        lib = _hip.load()
        ks = lib.swl_gemm_skinny_packed_choose_splits(n, k) if wp is not None else lib.swl_gemm_skinny_choose_splits(n, k)
        if ks > 1 or (always and ks == 1):
            ws = _workspace(a.device, ks * m * n * 4)
            _hip.call("swl_gemm_skinny_packed_partial" if wp is not None else "swl_gemm_skinny_partial", _hip.ptr(ws),
                      ws.numel() * 4, _hip.ptr(a), _hip.ptr(wp if wp is not None else w), m, n, k,
                      _row_stride(a), ks, _hip.dtype_code(a.dtype), _hip.stream())
            return SplitKPartials(ws, ks, m, n, a.dtype)
    return linear(a, w, skinny=True)


def row_scaled_silu_gate_ok(a: torch.Tensor, w_up_gate: torch.Tensor) -> bool:
    """Can linear_silu_gate take activations whose RMSNorm scale is pending (packed weight, <= 32 tokens)?"""
    return _skinny_ok(a, w_up_gate) and w_up_gate.shape[0] % 64 == 0 and _packed_of(w_up_gate) is not None


def linear_silu_gate(a: torch.Tensor, w_up_gate: torch.Tensor, row_scale=None):
    """The FFN's `silu_and_mul_inplace(linear(a, up_gate_proj))[:, :I]` in one launch for decode-sized
    batches: returns [T, I], or None when the shapes do not qualify (the caller then takes the two-op
    path). `row_scale` (kernels/rmsnorm.py: RowScalePending, `a` is its .x): the projection's fp32 results are
    multiplied by the pending 1/rms of their token before they are rounded."""
    if row_scale is not None:
        assert row_scaled_silu_gate_ok(a, w_up_gate) and row_scale.ssq.shape == (row_scale.parts, a.shape[0])
        m, k = a.shape
        inter = w_up_gate.shape[0] // 2
        out = torch.empty((m, inter), dtype=a.dtype, device=a.device)
        _hip.call("swl_gemm_skinny_packed_silu_gate_rs", _hip.ptr(out), _hip.ptr(a), _hip.ptr(_packed_of(w_up_gate)),
                  _hip.ptr(row_scale.ssq), row_scale.parts, row_scale.eps, m, inter, k, _row_stride(a), inter,
                  _hip.dtype_code(a.dtype), _hip.stream())
        return out
    if _wide_silu_ok(a, w_up_gate):        # large batch, packed weight
        return _launch_wide_silu(a, w_up_gate)
    if _mid_ok(a, w_up_gate) and w_up_gate.shape[0] % 64 == 0:   # medium batch, packed weight
        m, k = a.shape
        inter = w_up_gate.shape[0] // 2
        out = torch.empty((m, inter), dtype=a.dtype, device=a.device)
        _hip.call("swl_gemm_packed_mid_silu_gate", _hip.ptr(out), _hip.ptr(a), _hip.ptr(_packed_of(w_up_gate)), m,
                  inter, k, _row_stride(a), inter, _hip.dtype_code(a.dtype), _hip.stream())
        return out
    if not _skinny_ok(a, w_up_gate) or w_up_gate.shape[0] % 64 != 0:
        return None
    m, k = a.shape
    inter = w_up_gate.shape[0] // 2
    out = torch.empty((m, inter), dtype=a.dtype, device=a.device)
    wp = _packed_of(w_up_gate)
    _hip.call("swl_gemm_skinny_packed_silu_gate" if wp is not None else "swl_gemm_skinny_silu_gate", _hip.ptr(out),
              _hip.ptr(a), _hip.ptr(wp if wp is not None else w_up_gate), m, inter, k,
              _row_stride(a), inter, _hip.dtype_code(a.dtype), _hip.stream())
    return out


# ---- o_proj / down_proj with K split inside the workgroup and the residual add in the epilogue (csrc/gemm_rows.hip) ----
class RawResidual:
    """What a layer hands to the next one when its down projection already added itself into the residual buffer
    (linear_rows_add): there is no activation tensor and no slabs — the residual rows ARE the layer output, and the next
    projection normalises them on the fly (linear_splitk_nf; with `ssq` — the rows' per-tile sums of squares — exactly:
    linear_splitk_nx)."""
    __slots__ = ("shape", "dtype", "ssq")

    def __init__(self, residual: torch.Tensor, ssq: torch.Tensor = None):
        self.shape, self.dtype, self.ssq = tuple(residual.shape), residual.dtype, ssq


def rows_add_ok(a: torch.Tensor, w: torch.Tensor, residual: torch.Tensor) -> bool:
    """Can `linear_rows_add` run the projection + residual add in one launch (no slabs)? Packed weight, <= 32 tokens."""
    if _packed_of(w) is None or not _skinny_ok(a, w):
        return False
    m, k = a.shape
    n = w.shape[0]
    return (residual.is_contiguous() and residual.shape == (m, n) and residual.dtype == a.dtype
            and bool(_hip.load().swl_gemm_rows_supported(m, n, k)))


def linear_rows_add(a: torch.Tensor, w: torch.Tensor, residual_io: torch.Tensor, with_ssq: bool = False):
    """residual_io += round(a @ w^T), in place, one launch: the workgroup that owns 16 rows of w for all of K finishes
    them itself (csrc/gemm_rows.hip). Returns residual_io — or, `with_ssq`, the fp32 [tokens, out / 16] per-tile sums of
    squares of the updated rows (what an exact norm on the fly needs: linear_silu_gate_nx / linear_splitk_nx)."""
    assert rows_add_ok(a, w, residual_io)
    m, k = a.shape
    n = w.shape[0]
    if with_ssq:
        ssq = torch.empty((m, n // 16), dtype=torch.float32, device=a.device)
        _hip.call("swl_gemm_rows_add_ssq", _hip.ptr(residual_io), _hip.ptr(ssq), _hip.ptr(a), _hip.ptr(_packed_of(w)), m, n, k,
                  _row_stride(a), _hip.dtype_code(a.dtype), _hip.stream())
        return ssq
    _hip.call("swl_gemm_rows_add", _hip.ptr(residual_io), _hip.ptr(a), _hip.ptr(_packed_of(w)), m, n, k,
              _row_stride(a), _hip.dtype_code(a.dtype), _hip.stream())
    return residual_io


def nf_ok(r: torch.Tensor, w: torch.Tensor, norm_w: torch.Tensor) -> bool:
    """Can a projection stage round(r * norm_w) itself (norm on the fly)? Packed weight, <= 32 tokens, bfloat16 (the
    deferred norm's policy: kernels/rmsnorm.py deferred_norm_ok)."""
    return (_packed_of(w) is not None and _skinny_ok(r, w) and r.dtype == torch.bfloat16 and norm_w.dtype == r.dtype
            and norm_w.is_contiguous() and norm_w.numel() == r.shape[1])


def linear_splitk_nf(r: torch.Tensor, norm_w: torch.Tensor, w: torch.Tensor, eps: float):
    """linear_splitk(round(r * norm_w), w, always=True) with the rows' sums of squares on the side: returns (SplitKPartials,
    RowScalePending) — the pending 1/rms is applied by the consumer of the slabs (paged_attention_from_qkv_splitk)."""
    from .rmsnorm import RowScalePending
    assert nf_ok(r, w, norm_w)
    m, k = r.shape
    n = w.shape[0]
    ks = int(_hip.load().swl_gemm_skinny_packed_choose_splits(n, k))
    if k % (128 * ks):
        return None
    ws = _workspace(r.device, ks * m * n * 4)
    ssq = torch.empty((ks, m), dtype=torch.float32, device=r.device)
    _hip.call("swl_gemm_skinny_packed_partial_nf", _hip.ptr(ws), ws.numel() * 4, _hip.ptr(ssq), _hip.ptr(r), _hip.ptr(norm_w),
              _hip.ptr(_packed_of(w)), m, n, k, _row_stride(r), ks, _hip.dtype_code(r.dtype), _hip.stream())
    return SplitKPartials(ws, ks, m, n, r.dtype), RowScalePending(None, ssq, ks, eps, k)


def linear_silu_gate_nf(r: torch.Tensor, norm_w: torch.Tensor, eps: float, w_up_gate: torch.Tensor) -> torch.Tensor:
    """silu_and_mul(rmsnorm(r) * norm_w @ up_gate^T)[:, :I] from the raw residual rows, one launch: the norm weight is applied
    while the rows are staged, the 1/rms (from the kernel's own sums of squares) in fp32 before the projection's rounding."""
    assert nf_ok(r, w_up_gate, norm_w) and w_up_gate.shape[0] % 64 == 0
    m, k = r.shape
    inter = w_up_gate.shape[0] // 2
    out = torch.empty((m, inter), dtype=r.dtype, device=r.device)
    _hip.call("swl_gemm_skinny_packed_silu_gate_nf", _hip.ptr(out), _hip.ptr(r), _hip.ptr(norm_w), eps,
              _hip.ptr(_packed_of(w_up_gate)), m, inter, k, _row_stride(r), inter, _hip.dtype_code(r.dtype), _hip.stream())
    return out


def nx_ok(r: torch.Tensor, w: torch.Tensor, norm_w: torch.Tensor) -> bool:
    """Can a projection stage the EXACTLY normalised rows itself from the raw residual and its per-tile sums of squares
    (swl_gemm_skinny_packed_*_nx)? Packed weight, <= 32 tokens, hidden a multiple of 1024 (64 partials per piece), either
    16-bit dtype — the reference's rounding points, so no dtype policy applies."""
    return (_packed_of(w) is not None and _skinny_ok(r, w) and norm_w.dtype == r.dtype and norm_w.is_contiguous()
            and norm_w.numel() == r.shape[1] and r.shape[1] % 1024 == 0 and r.shape[1] >= 1024)


def linear_splitk_nx(r: torch.Tensor, norm_w: torch.Tensor, w: torch.Tensor, eps: float, ssq: torch.Tensor):
    """linear_splitk(rmsnorm(r) * norm_w, w, always=True) from the raw residual rows and their per-tile sums of squares
    (linear_rows_add(..., with_ssq=True)): SplitKPartials with nothing pending, or None when K has no even split."""
    assert nx_ok(r, w, norm_w) and ssq.dtype == torch.float32 and ssq.shape == (r.shape[0], r.shape[1] // 16) and ssq.is_contiguous()
    m, k = r.shape
    n = w.shape[0]
    ks = int(_hip.load().swl_gemm_skinny_packed_choose_splits(n, k))
    if k % (128 * ks):
        return None
    ws = _workspace(r.device, ks * m * n * 4)
    _hip.call("swl_gemm_skinny_packed_partial_nx", _hip.ptr(ws), ws.numel() * 4, _hip.ptr(r), _hip.ptr(norm_w), eps,
              _hip.ptr(ssq), ssq.shape[1], _hip.ptr(_packed_of(w)), m, n, k, _row_stride(r), ks, _hip.dtype_code(r.dtype),
              _hip.stream())
    return SplitKPartials(ws, ks, m, n, r.dtype)


def linear_silu_gate_nx(r: torch.Tensor, norm_w: torch.Tensor, eps: float, w_up_gate: torch.Tensor, ssq: torch.Tensor):
    """silu_and_mul(rmsnorm(r) * norm_w @ up_gate^T)[:, :I] from the raw residual rows and their per-tile sums of squares: one
    launch, the reference's one rounding of the normalised activation."""
    assert nx_ok(r, w_up_gate, norm_w) and w_up_gate.shape[0] % 64 == 0
    assert ssq.dtype == torch.float32 and ssq.shape == (r.shape[0], r.shape[1] // 16) and ssq.is_contiguous()
    m, k = r.shape
    inter = w_up_gate.shape[0] // 2
    out = torch.empty((m, inter), dtype=r.dtype, device=r.device)
    _hip.call("swl_gemm_skinny_packed_silu_gate_nx", _hip.ptr(out), _hip.ptr(r), _hip.ptr(norm_w), eps, _hip.ptr(ssq),
              ssq.shape[1], _hip.ptr(_packed_of(w_up_gate)), m, inter, k, _row_stride(r), inter, _hip.dtype_code(r.dtype),
              _hip.stream())
    return out


# ---- very small decode batches: the projection sums the previous projection's slabs itself (csrc/gemm_tiny.hip) ------
_TINY_MAX_M = 4             # swl_gemm_tiny_max_tokens(): what the kernels accept
_TINY_POLICY_M = 2          # what the layer uses them for: each workgroup re-reads 8 slabs x M x K-chunk through its L1
                            # (55 GB/s per CU): measured on MI355X, batch 1 gains 2-3 %, batch 2 is even, batch 4 loses 3 %
_TINY_MAX_KC = 4096         # K-chunk a workgroup keeps resident in LDS
_alt_workspaces = {}        # device -> second split-K scratch: a projection cannot write the slabs it is reading
_alt_residuals = {}         # (device, shape, dtype) -> second residual buffer (same reason; fixed address for hipGraphs)


def _alt_workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    ws = _alt_workspaces.get(device)
    if ws is None or ws.numel() * 4 < nbytes:
        if ws is not None:
            _retired.append(ws)
        ws = torch.empty(max(nbytes, 4 << 20) // 4, dtype=torch.float32, device=device)
        _alt_workspaces[device] = ws
    return ws


def alt_residual_like(residual: torch.Tensor) -> torch.Tensor:
    key = (residual.device, tuple(residual.shape), residual.dtype)
    buf = _alt_residuals.get(key)
    if buf is None:
        buf = _alt_residuals[key] = torch.empty_like(residual)
    return buf


def tiny_from_splitk_ok(partials, w: torch.Tensor, silu: bool = False) -> bool:
    """Can `w`'s projection consume `partials` (the previous projection's slabs) itself? <= 4 tokens, packed weight,
    the K-chunk of a workgroup fits LDS."""
    if not isinstance(partials, SplitKPartials) or _packed_of(w) is None:
        return False
    m, k = partials.shape
    n = w.shape[0]
    if not (0 < m <= _TINY_MAX_M and w.shape[1] == k and k % 128 == 0 and w.dtype == partials.dtype):
        return False
    if silu:
        return k <= _TINY_MAX_KC and n % 128 == 0
    ks = _hip.load().swl_gemm_skinny_choose_splits(n, k)
    return n % 32 == 0 and ks in (1, 2, 4) and k % ks == 0 and (k // ks) % 128 == 0 and k // ks <= _TINY_MAX_KC


def linear_splitk_from_splitk(partials, residual_in: torch.Tensor, residual_out: torch.Tensor, norm_w: torch.Tensor,
                              w: torch.Tensor):
    """residual_out = round(sum of `partials`) + residual_in; x = round(residual_out * norm_w) (never materialised);
    returns (SplitKPartials of x @ w^T, ssq[k_splits, M]) — the slabs and the per-K-chunk sums of squares the slab-fed
    attention kernel takes (paged_attention_from_qkv_splitk with a RowScalePending). One launch instead of
    add_scale_from_splitk + linear_splitk."""
    assert tiny_from_splitk_ok(partials, w) and residual_in.data_ptr() != residual_out.data_ptr()
    m, k = partials.shape
    n = w.shape[0]
    ks = _hip.load().swl_gemm_skinny_choose_splits(n, k)
    ws = _alt_workspace(residual_in.device, ks * m * n * 4)
    assert ws.data_ptr() != partials.slabs.data_ptr()
    ssq = torch.empty((ks, m), dtype=torch.float32, device=residual_in.device)
    _hip.call("swl_gemm_tiny_partial_from_splitk", _hip.ptr(ws), ws.numel() * 4, ks, _hip.ptr(ssq), _hip.ptr(partials.slabs),
              partials.k_splits, _hip.ptr(residual_in), _hip.ptr(residual_out), _hip.ptr(norm_w), _hip.ptr(_packed_of(w)),
              m, n, k, _hip.dtype_code(partials.dtype), _hip.stream())
    return SplitKPartials(ws, ks, m, n, partials.dtype), ssq


def attn_partials_ok(num_tokens: int, num_q_heads: int, head_dim: int, w: torch.Tensor) -> bool:
    """Can `w`'s projection (o_proj) merge the flash-decoding partials of `num_tokens` sequences itself?"""
    if _packed_of(w) is None or not (0 < num_tokens <= _TINY_MAX_M) or head_dim % 8:
        return False
    n, k = w.shape
    if k != num_q_heads * head_dim or n % 32 or k % 128:
        return False
    ks = _hip.load().swl_gemm_skinny_choose_splits(n, k)
    return ks > 1 and k % ks == 0 and (k // ks) % 128 == 0 and k // ks <= _TINY_MAX_KC


def linear_splitk_from_attn_partials(scratch: torch.Tensor, seq_lens: torch.Tensor, num_tokens: int, num_q_heads: int,
                                     head_dim: int, seq_block_size: int, num_seq_blocks: int, w: torch.Tensor,
                                     dtype: torch.dtype):
    """o_proj on the partials of a split flash-decoding (paged_attention_from_qkv_splitk(..., merge=False)): every
    workgroup merges the partials of its K-chunk of heads itself; returns the projection's SplitKPartials. One launch
    instead of the phase-2 merge + linear_splitk."""
    assert attn_partials_ok(num_tokens, num_q_heads, head_dim, w) and num_seq_blocks > 1
    n, k = w.shape
    ks = _hip.load().swl_gemm_skinny_choose_splits(n, k)
    ws = _workspace(scratch.device, ks * num_tokens * n * 4)
    _hip.call("swl_gemm_tiny_partial_from_attn", _hip.ptr(ws), ws.numel() * 4, ks, _hip.ptr(scratch), _hip.ptr(seq_lens),
              num_q_heads, head_dim, seq_block_size, num_seq_blocks, _hip.ptr(_packed_of(w)), num_tokens, n,
              _hip.dtype_code(dtype), _hip.stream())
    return SplitKPartials(ws, ks, num_tokens, n, dtype)


def linear_silu_gate_from_splitk(partials, residual_in: torch.Tensor, residual_out: torch.Tensor, norm_w: torch.Tensor,
                                 eps: float, w_up_gate: torch.Tensor) -> torch.Tensor:
    """residual_out = round(sum of `partials`) + residual_in; returns silu_and_mul(rmsnorm(residual_out) @ up_gate^T)
    [M, I] with the 1/rms applied in fp32 in the epilogue (deferred normalisation). One launch instead of
    add_scale_from_splitk + linear_silu_gate."""
    assert tiny_from_splitk_ok(partials, w_up_gate, silu=True) and residual_in.data_ptr() != residual_out.data_ptr()
    m, k = partials.shape
    inter = w_up_gate.shape[0] // 2
    out = torch.empty((m, inter), dtype=partials.dtype, device=residual_in.device)
    _hip.call("swl_gemm_tiny_silu_gate_from_splitk", _hip.ptr(out), _hip.ptr(partials.slabs), partials.k_splits,
              _hip.ptr(residual_in), _hip.ptr(residual_out), _hip.ptr(norm_w), eps, _hip.ptr(_packed_of(w_up_gate)), m,
              inter, k, inter, _hip.dtype_code(partials.dtype), _hip.stream())
    return out
