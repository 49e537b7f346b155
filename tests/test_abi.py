"""The C-ABI surface: include/swiftllm_hip.h, the ctypes table in swiftllm_amd/_hip.py and the built
library must agree; the library must load without a GPU; bad arguments come back as error codes."""
import ctypes
import os
import re

import pytest

from swiftllm_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "swiftllm_hip.h")


def _declared():
    """name -> number of parameters, parsed from the header's prototypes."""
    text = open(HEADER, encoding="utf-8").read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|const char \*)\s*\*?\s*(swl_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        params = m.group(2).strip()
        out[m.group(1)] = 0 if params in ("", "void") else len(params.split(","))
    return out


def test_header_declares_the_expected_entry_points():
    decl = _declared()
    assert len(decl) >= 19
    for name in ("swl_paged_attn_decode", "swl_prefill_attn_varlen", "swl_swap_blocks",
                 "swl_fused_add_rmsnorm", "swl_store_kv_prefill", "swl_block_table_set"):
        assert name in decl


def test_ctypes_table_matches_header():
    decl = _declared()
    table = dict(_hip.SIGNATURES)
    table.update({k: v[0] for k, v in _hip._SPECIAL.items()})
    assert set(table) == set(decl), set(table) ^ set(decl)
    for name, argtypes in table.items():
        assert len(argtypes) == decl[name], name


def test_library_loads_and_exports_every_symbol():
    lib = _hip.load()
    for name in _declared():
        assert hasattr(lib, name), f"{name} missing from {_hip.library_path()}"
    assert lib.swl_abi_version() == _hip.ABI_VERSION
    assert lib.swl_strerror(0) == b"ok"
    assert b"bad argument" in lib.swl_strerror(-1)


def test_scratch_bytes_formula():
    # mid_o [Bd, H, nsb, D] fp32 + mid_lse [Bd, H, nsb] fp32 (reference paged_attn.py:170-180)
    assert _hip.scratch_bytes(32, 32, 128, 5) == 32 * 32 * 5 * 129 * 4
    assert _hip.scratch_bytes(0, 32, 128, 5) == 0


def test_argument_validation_without_a_gpu():
    """Validation happens before any launch, so it is observable on a CPU-only box."""
    lib = _hip.load()
    # empty batches are legal and launch nothing
    assert lib.swl_rmsnorm(None, None, 1e-5, 0, 4096, _hip.SWL_F16, None) == 0
    assert lib.swl_silu_mul(None, 0, 14336, _hip.SWL_F16, None) == 0
    assert lib.swl_argmax(None, None, None, 0, 0, 128256, 128256, _hip.SWL_BF16, None) == 0
    assert lib.swl_argmax(None, None, None, 0, 4, 128257, 128264, _hip.SWL_BF16, None) == -1   # n % 8 != 0
    assert lib.swl_argmax_scratch_bytes(32) == 32 * 64 * 8
    assert lib.swl_swap_blocks(None, None, 0, 1, None, None, None, None, 1 << 20, None) == 0
    # null pointers / bad shapes
    assert lib.swl_rmsnorm(None, None, 1e-5, 4, 4096, _hip.SWL_F16, None) == -1
    assert lib.swl_rmsnorm(None, None, 1e-5, 4, 4095, _hip.SWL_F16, None) == -1
    buf = ctypes.create_string_buffer(64)
    p = ctypes.addressof(buf) // 16 * 16 + 16
    assert lib.swl_rmsnorm(p, p, 1e-5, 1, 8, 7, None) == -1     # unknown dtype code
    with pytest.raises(_hip.HipLibraryError):
        _hip.call("swl_silu_mul", None, 4, 7, _hip.SWL_F16, None)   # I % 8 != 0


def test_split_heuristics_are_host_side_and_stable():
    """K-split choices are pure host arithmetic (observable without a GPU); the values are the measured optima
    recorded in DESIGN.md section 4.3 / 5.1 for the Llama-3-8B projections."""
    lib = _hip.load()
    assert lib.swl_gemm_skinny_choose_splits(6144, 4096) == 4        # fused qkv
    assert lib.swl_gemm_skinny_choose_splits(4096, 4096) == 8        # o_proj
    assert lib.swl_gemm_skinny_choose_splits(4096, 14336) == 8       # down_proj
    assert lib.swl_gemm_skinny_choose_splits(28672, 4096) == 1       # up/gate: enough tiles already
    assert lib.swl_gemm_skinny_choose_splits(128256, 4096) == 1      # lm_head
    assert lib.swl_gemm_skinny_choose_splits(100, 4096) == 0         # N % 32 != 0: unsupported
    assert lib.swl_gemm_packed_mid_choose_splits(64, 4096, 4096) == 4     # slab traffic capped at K / (12 M)
    assert lib.swl_gemm_packed_mid_choose_splits(64, 4096, 14336) == 8
    assert lib.swl_gemm_packed_mid_choose_splits(128, 4096, 14336) == 0   # > 64 tokens: not this kernel's (gemm_wide.hip)
    assert lib.swl_gemm_packed_mid_choose_splits(64, 28672, 4096) == 1
    assert lib.swl_gemm_packed_mid_choose_splits(129, 4096, 4096) == 0    # M > 128
    # argument validation of the packed entry points (no launch)
    assert lib.swl_gemm_pack_weight(None, None, 4096, 4096, _hip.SWL_BF16, None) == -1
    assert lib.swl_gemm_skinny_packed(None, None, None, None, 0, 0, 4096, 4096, 4096, 4096, 0, _hip.SWL_BF16, None) == 0
    assert lib.swl_gemm_packed_mid(None, None, None, None, 0, 0, 4096, 4096, 4096, 4096, 0, _hip.SWL_BF16, None) == 0


def test_split_choosers_are_host_functions_with_the_documented_values():
    """swl_gemm_skinny_choose_splits / swl_gemm_skinny_packed_choose_splits / swl_gemm_tiny_max_tokens need no device: the values the Python layer (and DESIGN.md section 4.3) relies on."""
    from swiftllm_amd import _hip
    lib = _hip.load()
    even, packed = lib.swl_gemm_skinny_choose_splits, lib.swl_gemm_skinny_packed_choose_splits
    # Llama-3-8B decode projections: the packed path keeps the even splits
    for (n, k), ks in {(6144, 4096): 4, (4096, 4096): 8, (4096, 14336): 8, (28672, 4096): 1, (128256, 4096): 1}.items():
        assert even(n, k) == ks and packed(n, k) == ks, (n, k)
    # Llama-2-7B down_proj: 86 K-tiles have no power-of-two split beyond 2; the packed path splits unevenly into 8
    assert even(4096, 11008) == 2 and packed(4096, 11008) == 8
    assert even(12288, 4096) == packed(12288, 4096) == 2
    # never fewer splits than the row-major kernels, never a split shorter than 8 tiles when uneven
    for n in (32, 256, 4096, 6144):
        for kt in (8, 17, 43, 86, 112, 129):
            k = kt * 128
            e, p = even(n, k), packed(n, k)
            assert 1 <= e <= p <= 16
            if k % (128 * p):
                assert kt // p >= 8
    assert even(100, 4096) == 0 and packed(4096, 100) == 0          # N % 32, K % 128
    assert lib.swl_gemm_tiny_max_tokens() == 4
