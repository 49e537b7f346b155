// swap_blocks.hip — KV block swapping between the GPU pools and the host swap pools.
//
// Replaces the reference's only native function, swiftllm_c.swap_blocks
// (csrc/src/block_swapping.cpp:22-85, bound at csrc/src/entrypoints.cpp:5-7). Host code on the HIP
// runtime: consecutive (src, dst) block-id pairs are run-length coalesced and each run becomes ONE
// hipMemcpyAsync per pool on the caller's stream (the reference issues them on torch's current
// stream, block_swapping.cpp:32). With pinned swap pools (how this framework allocates them) the
// copies are truly asynchronous DMA over PCIe Gen5; with pageable memory they degrade to the
// reference's effectively-synchronous behaviour.
#include "swl_common.h"

extern "C" int swl_swap_blocks(const int64_t *src_ids, const int64_t *dst_ids,
                               int64_t num_blocks_to_swap, int32_t is_swap_in, void *k_cache,
                               void *v_cache, void *k_swap, void *v_swap, int64_t block_bytes,
                               swl_stream_t stream) {
    if (num_blocks_to_swap < 0 || block_bytes <= 0) return SWL_ERR_BAD_ARG;
    if (num_blocks_to_swap == 0) return SWL_OK;
    if (!src_ids || !dst_ids || !k_cache || !v_cache || !k_swap || !v_swap) return SWL_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    char *kc = static_cast<char *>(k_cache), *vc = static_cast<char *>(v_cache);
    char *ks = static_cast<char *>(k_swap), *vs = static_cast<char *>(v_swap);
    int64_t run_begin = 0;
    while (run_begin < num_blocks_to_swap) {
        int64_t run_end = run_begin + 1;
        while (run_end < num_blocks_to_swap && src_ids[run_end] == src_ids[run_end - 1] + 1 &&
               dst_ids[run_end] == dst_ids[run_end - 1] + 1)
            ++run_end;
        const size_t bytes = static_cast<size_t>(run_end - run_begin) * block_bytes;
        const size_t src_off = static_cast<size_t>(src_ids[run_begin]) * block_bytes;
        const size_t dst_off = static_cast<size_t>(dst_ids[run_begin]) * block_bytes;
        hipError_t e1, e2;
        if (is_swap_in) { // host swap pool -> GPU pool
            e1 = hipMemcpyAsync(kc + dst_off, ks + src_off, bytes, hipMemcpyHostToDevice, s);
            e2 = hipMemcpyAsync(vc + dst_off, vs + src_off, bytes, hipMemcpyHostToDevice, s);
        } else { // GPU pool -> host swap pool
            e1 = hipMemcpyAsync(ks + dst_off, kc + src_off, bytes, hipMemcpyDeviceToHost, s);
            e2 = hipMemcpyAsync(vs + dst_off, vc + src_off, bytes, hipMemcpyDeviceToHost, s);
        }
        if (e1 != hipSuccess || e2 != hipSuccess) return SWL_ERR_RUNTIME;
        run_begin = run_end;
    }
    return SWL_OK;
}

extern "C" int swl_abi_version(void) { return SWL_ABI_VERSION; }

extern "C" const char *swl_strerror(int code) {
    switch (code) {
    case SWL_OK: return "ok";
    case SWL_ERR_BAD_ARG: return "bad argument (null/misaligned pointer, negative size or stride)";
    case SWL_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
    case SWL_ERR_LAUNCH: return "kernel launch failed (hipGetLastError)";
    case SWL_ERR_RUNTIME: return "HIP runtime call failed";
    default: return "unknown error";
    }
}
