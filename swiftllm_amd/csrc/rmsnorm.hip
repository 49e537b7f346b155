// rmsnorm.hip — RMSNorm and fused residual-add + RMSNorm for gfx950.
//
// Replaces the reference's Triton kernels _fwd_rmsnorm (swiftllm/worker/kernels/rmsnorm.py:5-24)
// and _fwd_fused_add_rmsnorm (rmsnorm.py:39-65). Pure HBM-bandwidth kernels:
//   rmsnorm            2*T*h*e  (+h*e weight, L2 resident)
//   fused_add_rmsnorm  4*T*h*e
// Mapping: one workgroup per token row; each lane owns VPT 16-byte vectors (8 elements) that stay
// in registers between the reduction and the scale pass, so every byte is read exactly once.
// Rounding points follow the reference: the residual add is rounded to the storage dtype and
// stored (rmsnorm.py:54-57), the norm is fp32 (sum of squares, 1/sqrt, *w) with one final rounding.
#include "swl_common.h"

namespace swl {

template <int NWAVES>
__device__ __forceinline__ float block_allreduce_sum(float v, float *lds) {
    v = wave_allreduce_sum(v);
    if constexpr (NWAVES > 1) {
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) lds[wave] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < NWAVES; ++i) t += lds[i];
        v = t;
    }
    return v;
}

// NT = threads per block (multiple of 64), VPT = vectors (of 8 elements) per thread.
template <typename T, int NT, int VPT, bool FUSED_ADD>
__global__ __launch_bounds__(NT) void rmsnorm_kernel(T *__restrict__ x, T *__restrict__ residual,
                                                     const T *__restrict__ w, float eps, int hidden,
                                                     const float *__restrict__ slabs, int ks,
                                                     int64_t slab_stride) {
    __shared__ float red[NT / 64];
    const int64_t row = blockIdx.x;
    const int nvec = hidden >> 3;
    T *xr = x + row * hidden;
    T *rr = FUSED_ADD ? residual + row * hidden : nullptr;

    // the norm weight does not depend on anything computed here: fetch it with the first wave of loads
    // instead of after the reduction (one L2 round trip off the critical path of a latency-bound kernel)
    vec8_t<T> wv[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = threadIdx.x + i * NT;
        if (v < nvec) wv[i] = load8(w + v * 8);
    }
    float vals[VPT][8];
    float ssq = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = threadIdx.x + i * NT;
        if (v < nvec) {
            // x either as stored activations or as the split-K partial slabs of the GEMM that produced it
            vec8_t<T> xv = slabs ? load8_splitk<T>(slabs, ks, slab_stride, row * hidden + v * 8)
                                 : load8(xr + v * 8);
            if constexpr (FUSED_ADD) {
                vec8_t<T> rv = load8(rr + v * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] = add_t<T>(xv[j], rv[j]); // rounded to T, as stored
                store8(rr + v * 8, xv);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                vals[i][j] = to_f(xv[j]);
                ssq = fmaf(vals[i][j], vals[i][j], ssq);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) vals[i][j] = 0.f;
        }
    }
    ssq = block_allreduce_sum<NT / 64>(ssq, red);
    const float variance = ssq / static_cast<float>(hidden);
    const float rstd = 1.0f / sqrtf(variance + eps);
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int v = threadIdx.x + i * NT;
        if (v < nvec) {
            vec8_t<T> ov;
#pragma unroll
            for (int j = 0; j < 8; ++j) ov[j] = to_t<T>(vals[i][j] * rstd * to_f(wv[i][j]));
            store8(xr + v * 8, ov);
        }
    }
}

template <typename T, bool FUSED_ADD>
static int launch_rmsnorm(T *x, T *residual, const T *w, float eps, int64_t num_tokens, int hidden,
                          hipStream_t stream, const float *slabs = nullptr, int ks = 0) {
    const int64_t slab_stride = num_tokens * hidden;
    const int nvec = hidden / 8;
    const dim3 grid(static_cast<unsigned>(num_tokens));
#define SWL_RMS_CASE(NT, VPT)                                                                    \
    hipLaunchKernelGGL((rmsnorm_kernel<T, NT, VPT, FUSED_ADD>), grid, dim3(NT), 0, stream, x,    \
                       residual, w, eps, hidden, slabs, ks, slab_stride)
    if (nvec <= 64) SWL_RMS_CASE(64, 1);
    else if (nvec <= 128) SWL_RMS_CASE(128, 1);
    else if (nvec <= 256) SWL_RMS_CASE(256, 1);
    else if (nvec <= 512) SWL_RMS_CASE(256, 2);   // hidden = 4096: 2 x 16 B per lane
    else if (nvec <= 1024) SWL_RMS_CASE(256, 4);
    else if (nvec <= 2048) SWL_RMS_CASE(256, 8);
    else return SWL_ERR_UNSUPPORTED;
#undef SWL_RMS_CASE
    return check_launch();
}

} // namespace swl

extern "C" int swl_rmsnorm(void *x, const void *w, float eps, int64_t num_tokens, int32_t hidden,
                           int32_t dtype, swl_stream_t stream) {
    if (num_tokens < 0 || hidden <= 0 || (hidden & 7)) return SWL_ERR_BAD_ARG;
    if (num_tokens == 0) return SWL_OK;
    if (!x || !w || !swl::aligned16(x) || !swl::aligned16(w)) return SWL_ERR_BAD_ARG;
    if (num_tokens > 0x7fffffffLL) return SWL_ERR_UNSUPPORTED;
    SWL_DISPATCH_DTYPE(dtype, T, {
        return swl::launch_rmsnorm<T, false>(static_cast<T *>(x), nullptr,
                                             static_cast<const T *>(w), eps, num_tokens, hidden,
                                             static_cast<hipStream_t>(stream));
    });
}

extern "C" int swl_fused_add_rmsnorm(void *x, void *residual, const void *w, float eps,
                                     int64_t num_tokens, int32_t hidden, int32_t dtype,
                                     swl_stream_t stream) {
    if (num_tokens < 0 || hidden <= 0 || (hidden & 7)) return SWL_ERR_BAD_ARG;
    if (num_tokens == 0) return SWL_OK;
    if (!x || !residual || !w || !swl::aligned16(x) || !swl::aligned16(residual) ||
        !swl::aligned16(w))
        return SWL_ERR_BAD_ARG;
    if (num_tokens > 0x7fffffffLL) return SWL_ERR_UNSUPPORTED;
    SWL_DISPATCH_DTYPE(dtype, T, {
        return swl::launch_rmsnorm<T, true>(static_cast<T *>(x), static_cast<T *>(residual),
                                            static_cast<const T *>(w), eps, num_tokens, hidden,
                                            static_cast<hipStream_t>(stream));
    });
}

extern "C" int swl_splitk_fused_add_rmsnorm(void *x_out, void *residual, const void *w, float eps,
                                            const float *slabs, int32_t k_splits,
                                            int64_t num_tokens, int32_t hidden, int32_t dtype,
                                            swl_stream_t stream) {
    if (num_tokens < 0 || hidden <= 0 || (hidden & 7) || k_splits <= 0) return SWL_ERR_BAD_ARG;
    if (num_tokens == 0) return SWL_OK;
    if (!x_out || !residual || !w || !slabs || !swl::aligned16(x_out) || !swl::aligned16(residual) ||
        !swl::aligned16(w) || !swl::aligned16(slabs))
        return SWL_ERR_BAD_ARG;
    if (num_tokens > 0x7fffffffLL) return SWL_ERR_UNSUPPORTED;
    SWL_DISPATCH_DTYPE(dtype, T, {
        return swl::launch_rmsnorm<T, true>(static_cast<T *>(x_out), static_cast<T *>(residual),
                                            static_cast<const T *>(w), eps, num_tokens, hidden,
                                            static_cast<hipStream_t>(stream), slabs, k_splits);
    });
}
