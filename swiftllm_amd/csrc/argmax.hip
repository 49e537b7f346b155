// argmax.hip — greedy sampling: row-wise argmax of the logits, ties -> lowest index.
//
// The reference samples with `torch.argmax(logits, dim=1)` (swiftllm/worker/layers/post_layer.py:40);
// on ROCm that is a generic reduce kernel that takes ~41 us for [32, 128256] bf16 (8 MB, L2/MALL
// resident right after the lm_head GEMM). Two latency-sized launches instead: every (row, split)
// workgroup scans its slice with 16-byte loads and leaves one (value, index) candidate; the second
// kernel picks among the candidates. Comparison is on the stored values (exact, no rounding), NaNs are
// never selected (torch would return the first NaN — logits of a healthy model have none).
#include "swl_common.h"

namespace swl {

constexpr int kArgmaxSplits = 64;

__device__ __forceinline__ void argmax_merge(float &v, int &i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void argmax_partial_kernel(const T *__restrict__ x, int64_t row_stride,
                                                             int n, float *__restrict__ part_val,
                                                             int *__restrict__ part_idx) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int row = blockIdx.y;
    const int nvec = n >> 3;
    const int per = (nvec + kArgmaxSplits - 1) / kArgmaxSplits;
    const int v_begin = blockIdx.x * per;
    const int v_end = min(nvec, v_begin + per);
    const T *xr = x + row * row_stride;
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    for (int v = v_begin + threadIdx.x; v < v_end; v += 256) {
        const vec8_t<T> xv = load8(xr + v * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = to_f(xv[j]);
            if (f > best) { // ascending scan: strict '>' keeps the lowest index among equals
                best = f;
                best_i = v * 8 + j;
            }
        }
    }
#pragma unroll
    for (int mask = 1; mask < 64; mask <<= 1)
        argmax_merge(best, best_i, __shfl_xor(best, mask, 64), __shfl_xor(best_i, mask, 64));
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sv[wave] = best;
        si[wave] = best_i;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) argmax_merge(best, best_i, sv[w], si[w]);
        part_val[row * kArgmaxSplits + blockIdx.x] = best;
        part_idx[row * kArgmaxSplits + blockIdx.x] = best_i;
    }
}

__global__ __launch_bounds__(64) void argmax_final_kernel(const float *__restrict__ part_val,
                                                          const int *__restrict__ part_idx,
                                                          int64_t *__restrict__ out) {
    const int row = blockIdx.x;
    float best = part_val[row * kArgmaxSplits + threadIdx.x];
    int best_i = part_idx[row * kArgmaxSplits + threadIdx.x];
#pragma unroll
    for (int mask = 1; mask < 64; mask <<= 1)
        argmax_merge(best, best_i, __shfl_xor(best, mask, 64), __shfl_xor(best_i, mask, 64));
    // a row without any finite-comparable value (all NaN / -inf) falls back to index 0
    if (threadIdx.x == 0) out[row] = best_i == 0x7fffffff ? 0 : best_i;
}

} // namespace swl

extern "C" size_t swl_argmax_scratch_bytes(int64_t num_rows) {
    return num_rows > 0 ? static_cast<size_t>(num_rows) * swl::kArgmaxSplits * 8 : 0;
}

extern "C" int swl_argmax(int64_t *out, const void *x, void *scratch, size_t scratch_bytes, int64_t num_rows,
                          int32_t n, int64_t row_stride, int32_t dtype, swl_stream_t stream) {
    static_assert(swl::kArgmaxSplits == 64, "the final kernel is one wave wide");
    if (num_rows < 0 || n <= 0 || (n & 7) || row_stride < n || (row_stride & 7)) return SWL_ERR_BAD_ARG;
    if (num_rows == 0) return SWL_OK;
    if (!out || !x || !scratch || !swl::aligned16(x) || !swl::aligned16(scratch)) return SWL_ERR_BAD_ARG;
    if (scratch_bytes < swl_argmax_scratch_bytes(num_rows)) return SWL_ERR_BAD_ARG;
    if (num_rows > 65535) return SWL_ERR_UNSUPPORTED;
    float *pv = static_cast<float *>(scratch);
    int *pi = reinterpret_cast<int *>(pv + num_rows * swl::kArgmaxSplits);
    hipStream_t s = static_cast<hipStream_t>(stream);
    SWL_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL((swl::argmax_partial_kernel<T>), dim3(swl::kArgmaxSplits, static_cast<unsigned>(num_rows)),
                           dim3(256), 0, s, static_cast<const T *>(x), row_stride, n, pv, pi);
        hipLaunchKernelGGL(swl::argmax_final_kernel, dim3(static_cast<unsigned>(num_rows)), dim3(64), 0, s, pv, pi,
                           out);
    });
    return swl::check_launch();
}
