// prefetch.hip — pull a byte range into the memory-side caches ahead of the kernel that will stream it.
//
// A decode step is a chain of HBM-bound kernels, and every kernel spends its first and last microseconds with
// the memory pipe idle (launch, first-load latency, drain: ~20-25 % of a 125 us layer — DESIGN.md section 4.6).
// The weights of the NEXT projection depend on nothing: a few light workgroups on a second stream can read them
// while the current kernel ramps up or drains, so they sit in the 256 MiB Infinity Cache (and partly in L2) when
// their consumer starts. No result, no hand-off, no synchronisation: correctness cannot depend on this kernel.
#include "swl_common.h"

namespace swl {

// 16 bytes per lane per load, 8 loads in flight per lane; nothing is written (the xor keeps the loads alive).
__global__ __launch_bounds__(256) void cache_prefetch_kernel(const uint4 *__restrict__ src, int64_t n16) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
    unsigned acc = 0;
    int64_t i = blockIdx.x * 256ll + threadIdx.x;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) {
        const uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    asm volatile("" ::"v"(acc));
}

} // namespace swl

/* Read [ptr, ptr + bytes) with `workgroups` light workgroups (256 threads, <= 24 VGPRs: they co-reside with the
 * streaming kernels of the main stream) so the range is cache-resident for its consumer. No reference counterpart
 * (the reference has one stream of dependent kernels, model.py:236-248). ptr 16-byte aligned. */
extern "C" int swl_cache_prefetch(const void *ptr, size_t bytes, int32_t workgroups, swl_stream_t stream) {
    if (bytes == 0) return SWL_OK;
    if (!ptr || !swl::aligned16(ptr) || workgroups <= 0 || workgroups > 65535) return SWL_ERR_BAD_ARG;
    hipLaunchKernelGGL(swl::cache_prefetch_kernel, dim3(workgroups), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint4 *>(ptr), static_cast<int64_t>(bytes / 16));
    return swl::check_launch();
}
