"""ctypes binding of libswiftllm_hip.so — the C-ABI boundary declared in include/swiftllm_hip.h.

There is exactly one compute backend: the hand-written gfx950 kernels behind this library. If the
library is missing or a call fails this module raises; it never falls back to PyTorch ops or to
anything under oracle/ (that directory is test infrastructure).

Tensors cross the boundary as raw device pointers (`tensor.data_ptr()`), sizes as ints, the stream
as `torch.cuda.current_stream().cuda_stream` — the same "launch on torch's current stream"
contract the reference's Triton kernels and swiftllm_c.swap_blocks have
(csrc/src/block_swapping.cpp:32).
"""
import ctypes
import os
import threading

import torch

_LIB_NAME = "libswiftllm_hip.so"
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", _LIB_NAME)

SWL_F16 = 0
SWL_BF16 = 1
ABI_VERSION = 1

_P = ctypes.c_void_p
_I32 = ctypes.c_int32
_I64 = ctypes.c_int64
_F32 = ctypes.c_float

# name -> argument ctypes, in the order of include/swiftllm_hip.h. tests/test_abi.py checks this table
# against the header (symbol set and arity) and against the built library's exports.
SIGNATURES = {
    "swl_rmsnorm": [_P, _P, _F32, _I64, _I32, _I32, _P],
    "swl_fused_add_rmsnorm": [_P, _P, _P, _F32, _I64, _I32, _I32, _P],
    "swl_rotary": [_P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I64, _I64, _I32, _P],
    "swl_store_kv_prefill": [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32,
                             _I32, _I32, _I64, _I64, _I32, _P],
    "swl_store_kv_decode": [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32,
                            _I64, _I64, _I32, _P],
    "swl_rotary_store_kv_prefill": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32,
                                    _I32, _I32, _I32, _I64, _I64, _I64, _I32, _P],
    "swl_silu_mul": [_P, _I64, _I32, _I32, _P],
    "swl_argmax": [_P, _P, _P, ctypes.c_size_t, _I64, _I32, _I64, _I32, _P],
    "swl_paged_attn_decode": [_P, _P, _P, _P, _P, _P, _P, _P, _F32, _I32, _I32, _I32, _I32, _I32,
                              _I32, _I32, _I32, _I32, _I32, _I64, _I64, _I32, _P],
    "swl_paged_attn_phase1": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _F32, _I32, _I32, _I32, _I32,
                              _I32, _I32, _I32, _I32, _I32, _I32, _I64, _I64, _I32, _P],
    "swl_paged_attn_decode_qkv": [_P, _P, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F32, _I32, _I32, _I32, _I32,
                                  _I32, _I32, _I32, _I32, _I32, _I32, _I64, _I32, _P],
    "swl_paged_attn_phase2": [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I64, _I32, _P],
    "swl_prefill_attn_varlen": [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _F32, _I64, _I64,
                                _I64, _I64, _I32, _P],
    "swl_block_table_set": [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _P],
    "swl_block_table_unset": [_P, _P, _P, _P, _I32, _I32, _P],
    "swl_block_table_gather": [_P, _P, _P, _P, _P, _P, _I32, _I32, _P],
    "swl_swap_blocks": [_P, _P, _I64, _I32, _P, _P, _P, _P, _I64, _P],
    "swl_rotary_store_kv_decode": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32,
                                   _I32, _I32, _I32, _I32, _I32, _I64, _I64, _I64, _I32, _P],
    "swl_decode_positions": [_P, _P, _I32, _P],
    "swl_gemm_skinny": [_P, _P, _P, _P, ctypes.c_size_t, _I32, _I32, _I32, _I64, _I64, _I32, _I32, _P],
    "swl_gemm_pack_weight": [_P, _P, _I32, _I32, _I32, _P],
    "swl_gemm_skinny_packed": [_P, _P, _P, _P, ctypes.c_size_t, _I32, _I32, _I32, _I64, _I64, _I32, _I32, _P],
    "swl_gemm_packed_mid_partial": [_P, ctypes.c_size_t, _P, _P, _I32, _I32, _I32, _I64, _I32, _I32, _P],
    "swl_gemm_packed_mid_silu_gate": [_P, _P, _P, _I32, _I32, _I32, _I64, _I64, _I32, _P],
    "swl_gemm_packed_mid": [_P, _P, _P, _P, ctypes.c_size_t, _I32, _I32, _I32, _I64, _I64, _I32, _I32, _P],
    "swl_gemm_packed_wide": [_P, _P, _P, _P, ctypes.c_size_t, _I32, _I32, _I32, _I64, _I64, _I32, _I32, _I32, _P],
    "swl_gemm_packed_wide_partial": [_P, ctypes.c_size_t, _P, _P, _I32, _I32, _I32, _I64, _I32, _I32, _I32, _P],
    "swl_gemm_packed_wide_silu_gate": [_P, _P, _P, _I32, _I32, _I32, _I64, _I64, _I32, _I32, _P],
    "swl_gemm_skinny_packed_partial": [_P, ctypes.c_size_t, _P, _P, _I32, _I32, _I32, _I64, _I32, _I32, _P],
    "swl_gemm_skinny_packed_silu_gate": [_P, _P, _P, _I32, _I32, _I32, _I64, _I64, _I32, _P],
    "swl_gemm_skinny_partial": [_P, ctypes.c_size_t, _P, _P, _I32, _I32, _I32, _I64, _I32, _I32, _P],
    "swl_splitk_reduce": [_P, _P, _I32, _I32, _I32, _I64, _I32, _P],
    "swl_gemm_skinny_silu_gate": [_P, _P, _P, _I32, _I32, _I32, _I64, _I64, _I32, _P],
    "swl_splitk_fused_add_rmsnorm": [_P, _P, _P, _F32, _P, _I32, _I64, _I32, _I32, _P],
    "swl_splitk_add_scale": [_P, _P, _P, _P, _I32, _P, _I64, _I32, _I32, _P],
    "swl_gemm_skinny_packed_silu_gate_rs": [_P, _P, _P, _P, _I32, _F32, _I32, _I32, _I32, _I64, _I64, _I32, _P],
    "swl_paged_attn_decode_qkv_rs": [_P, _P, _I32, _P, _I32, _I32, _F32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F32, _I32,
                                     _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I64, _I32, _P],
    "swl_splitk_rotary_store_kv_decode": [_P, _P, _P, _P, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32,
                                          _I32, _I32, _I32, _I32, _I32, _I32, _I64, _I64, _I64, _I32, _P],
    "swl_paged_attn_decode_qkv_rs_partials": [_P, _P, _I32, _P, _I32, _I32, _F32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F32,
                                              _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I64, _I32, _P],
    "swl_gemm_tiny_partial_from_attn": [_P, ctypes.c_size_t, _I32, _P, _P, _I32, _I32, _I32, _I32, _P, _I32, _I32, _I32, _P],
    "swl_gemm_tiny_partial_from_splitk": [_P, ctypes.c_size_t, _I32, _P, _P, _I32, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P],
    "swl_gemm_tiny_silu_gate_from_splitk": [_P, _P, _I32, _P, _P, _P, _F32, _P, _I32, _I32, _I32, _I64, _I32, _P],
    "swl_gemm_rows_add": [_P, _P, _P, _I32, _I32, _I32, _I64, _I32, _P],
    "swl_gemm_skinny_packed_partial_nf": [_P, ctypes.c_size_t, _P, _P, _P, _P, _I32, _I32, _I32, _I64, _I32, _I32, _P],
    "swl_gemm_skinny_packed_silu_gate_nf": [_P, _P, _P, _F32, _P, _I32, _I32, _I32, _I64, _I64, _I32, _P],
    "swl_gemm_rows_add_ssq": [_P, _P, _P, _P, _I32, _I32, _I32, _I64, _I32, _P],
    "swl_gemm_skinny_packed_silu_gate_nx": [_P, _P, _P, _F32, _P, _I32, _P, _I32, _I32, _I32, _I64, _I64, _I32, _P],
    "swl_gemm_skinny_packed_partial_nx": [_P, ctypes.c_size_t, _P, _P, _F32, _P, _I32, _P, _I32, _I32, _I32, _I64, _I32, _I32, _P],
    "swl_decode_engine_reset": [_P, ctypes.c_size_t, _P],
    "swl_decode_engine_step": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _P, _P, _I32, _I32,
                               _I32, _I32, _I32, _I32, _I32, _F32, _F32, _I32, _I32, _P],
}
# Entry points that do not follow the "int rc = f(...)" convention.
_SPECIAL = {
    "swl_abi_version": ([], _I32),
    "swl_strerror": ([_I32], ctypes.c_char_p),
    "swl_paged_attn_scratch_bytes": ([_I32, _I32, _I32, _I32], ctypes.c_size_t),
    "swl_argmax_scratch_bytes": ([_I64], ctypes.c_size_t),
    "swl_gemm_skinny_workspace_bytes": ([_I32, _I32, _I32], ctypes.c_size_t),
    "swl_gemm_skinny_choose_splits": ([_I32, _I32], _I32),
    "swl_gemm_skinny_packed_choose_splits": ([_I32, _I32], _I32),
    "swl_gemm_packed_mid_choose_splits": ([_I32, _I32, _I32], _I32),
    "swl_gemm_packed_wide_workspace_bytes": ([_I32, _I32, _I32], ctypes.c_size_t),
    "swl_gemm_packed_wide_choose_splits": ([_I32, _I32, _I32], _I32),
    "swl_gemm_tiny_max_tokens": ([], _I32),
    "swl_gemm_rows_supported": ([_I32, _I32, _I32], _I32),
    "swl_decode_engine_supported": ([_I32, _I32, _I32, _I32, _I32, _I32], _I32),
    "swl_decode_engine_slots_per_layer": ([_I32, _I32, _I32, _I32], _I32),
    "swl_decode_engine_workspace_bytes": ([_I32, _I32, _I32, _I32], ctypes.c_size_t),
}

_lock = threading.Lock()
_lib = None


class HipLibraryError(RuntimeError):
    """libswiftllm_hip.so is missing, stale, or a kernel entry point returned an error code."""


def library_path() -> str:
    return os.environ.get("SWIFTLLM_HIP_LIB", _LIB_PATH)


def load():
    """Load (once) and return the ctypes handle; raises HipLibraryError when it cannot."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not os.path.exists(path):
            raise HipLibraryError(
                f"{path} not found. Build it with `python -m swiftllm_amd.csrc.build` "
                "(hipcc, gfx950). There is no fallback backend.")
        try:
            lib = ctypes.CDLL(path)
        except OSError as e:
            raise HipLibraryError(f"cannot load {path}: {e}") from e
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = _I32
        for name, (argtypes, restype) in _SPECIAL.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = restype
        got = lib.swl_abi_version()
        if got != ABI_VERSION:
            raise HipLibraryError(f"{path} has ABI version {got}, this package needs {ABI_VERSION}; rebuild")
        _lib = lib
        return lib


def is_available() -> bool:
    """True when the library file exists (does not load it)."""
    return os.path.exists(library_path())


def call(name: str, *args) -> None:
    """Invoke `name`; map a non-zero return code to HipLibraryError (a RuntimeError)."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        msg = load().swl_strerror(rc).decode()
        raise HipLibraryError(f"{name} failed with code {rc}: {msg}")


def scratch_bytes(num_decoding_seqs: int, num_q_heads: int, head_dim: int, num_seq_blocks: int) -> int:
    return int(load().swl_paged_attn_scratch_bytes(num_decoding_seqs, num_q_heads, head_dim,
                                                   num_seq_blocks))


def dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.float16:
        return SWL_F16
    if dtype == torch.bfloat16:
        return SWL_BF16
    raise TypeError(f"libswiftllm_hip supports float16 and bfloat16, got {dtype}")


def ptr(t) -> int:
    """Device (or host) address of a tensor, or NULL for None."""
    return 0 if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream() -> int:
    """The raw hipStream_t of torch's current stream on the current device. Called once per kernel launch: through the two C
    entry points torch itself uses (0.3 us) instead of `torch.cuda.current_stream()` (a Python Stream object per call: 8.5 us,
    a quarter of the host time of an eager forward — tools/prefill_host_profile.py, r06b)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def require_gpu_tensor(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise HipLibraryError(
            f"{what} lives on {t.device}; the swiftllm_amd operators only run on a HIP device "
            "(no CPU fallback).")
