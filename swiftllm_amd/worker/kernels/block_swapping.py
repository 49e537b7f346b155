"""swap_blocks — KV block copies between the GPU pools and the host swap pools.

Reference: the `swiftllm_c.swap_blocks` pybind11 function (csrc/src/block_swapping.cpp:22-85,
csrc/src/entrypoints.cpp:5-7), called from swiftllm/worker/model.py:372-379.
"""
import ctypes

import torch

from swiftllm_amd import _hip


def swap_blocks(source_block_ids: list, target_block_ids: list, is_swap_in: bool,
                k_cache: torch.Tensor, v_cache: torch.Tensor, k_swap: torch.Tensor,
                v_swap: torch.Tensor):
    """Copy block source_block_ids[i] -> target_block_ids[i] (K and V) on torch's current stream.
    swap-in = host swap pool -> GPU pool, swap-out = GPU pool -> host swap pool."""
    n = len(source_block_ids)
    assert n == len(target_block_ids)
    if n == 0:
        return
    _hip.require_gpu_tensor(k_cache, "k_cache")
    assert k_cache.is_contiguous() and v_cache.is_contiguous()
    assert k_swap.is_contiguous() and v_swap.is_contiguous() and not k_swap.is_cuda
    block_bytes = k_cache.numel() * k_cache.element_size() // k_cache.shape[0]
    assert block_bytes == k_swap.numel() * k_swap.element_size() // max(k_swap.shape[0], 1)
    arr = ctypes.c_int64 * n
    src = arr(*source_block_ids)
    dst = arr(*target_block_ids)
    _hip.call("swl_swap_blocks", ctypes.cast(src, ctypes.c_void_p), ctypes.cast(dst, ctypes.c_void_p),
              n, 1 if is_swap_in else 0, _hip.ptr(k_cache), _hip.ptr(v_cache), _hip.ptr(k_swap),
              _hip.ptr(v_swap), block_bytes, _hip.stream())
