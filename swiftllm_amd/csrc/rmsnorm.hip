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

// ---- deferred normalisation (decode fast path) --------------------------------------------------------------------
// fused_add_rmsnorm needs a whole row before it can write anything (the 1/rms), so as a split-K consumer it runs one
// workgroup per token: 32 workgroups on a 256-CU part, each pulling 131 KB of slabs — 4.6 us of a 130 us layer, twice
// per layer. But the 1/rms is a per-row SCALAR and commutes with the projection that follows:
//     rmsnorm(r) . W^T  =  rstd * ((r * w_norm) . W^T)
// so this kernel only does the element-wise part, fully parallel over rows AND columns — reduce the slabs, add the
// residual (same rounding as fused_add_rmsnorm: the rounded sum is stored to `residual`), write xs = round(r * w_norm)
// and the per-chunk sums of squares of r — and the consumer GEMM / attention prologue multiplies its fp32 results by
// rstd = 1/sqrt(sum(ssq)/hidden + eps) before their one rounding. Numerically this moves one rounding (the reference
// rounds the normalised activations, rmsnorm.py:57-64; here the un-normalised r * w is rounded and the scale is applied
// in fp32): a deviation of the same order as a GEMM's summation order, covered by the end-to-end parity tolerances.
constexpr int kAddScaleThreads = 128;
constexpr int kAddScaleChunk = kAddScaleThreads * 8; // columns per workgroup

template <typename T>
__global__ __launch_bounds__(kAddScaleThreads) void splitk_add_scale_kernel(
    T *__restrict__ xs, T *__restrict__ residual, const T *__restrict__ w, const float *__restrict__ slabs, int ks,
    int64_t slab_stride, float *__restrict__ ssq_out, int num_tokens, int hidden) {
    __shared__ float red[kAddScaleThreads / 64];
    const int part = blockIdx.x, row = blockIdx.y;
    const int col = part * kAddScaleChunk + threadIdx.x * 8;
    const int64_t off = static_cast<int64_t>(row) * hidden + col;
    const vec8_t<T> wv = load8(w + col);
    const vec8_t<T> rv = load8(residual + off);
    vec8_t<T> xv = load8_splitk<T>(slabs, ks, slab_stride, off);
    vec8_t<T> sv;
    float ssq = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        xv[j] = add_t<T>(xv[j], rv[j]); // rounded to T, as stored (rmsnorm.py:54-57)
        const float v = to_f(xv[j]);
        ssq = fmaf(v, v, ssq);
        sv[j] = to_t<T>(v * to_f(wv[j]));
    }
    store8(residual + off, xv);
    store8(xs + off, sv);
    ssq = block_allreduce_sum<kAddScaleThreads / 64>(ssq, red);
    if (threadIdx.x == 0) ssq_out[static_cast<int64_t>(part) * num_tokens + row] = ssq;
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

/* Deferred-normalisation split-K consumer (see splitk_add_scale_kernel): residual += round(sum_k slabs[k]);
 * x_scaled = round(residual * w); ssq_out[hidden / 1024][num_tokens] = per-1024-column sums of squares of the updated
 * residual rows. The 1/rms is applied by the consumer (swl_gemm_skinny_packed_silu_gate_rs, swl_paged_attn_decode_qkv_rs).
 * hidden % 1024 == 0. Replaces fused_add_rmsnorm (reference rmsnorm.py:67-89) on the decode fast path. */
extern "C" int swl_splitk_add_scale(void *x_scaled, void *residual, const void *w, const float *slabs, int32_t k_splits,
                                    float *ssq_out, int64_t num_tokens, int32_t hidden, int32_t dtype,
                                    swl_stream_t stream) {
    if (num_tokens < 0 || hidden <= 0 || k_splits <= 0) return SWL_ERR_BAD_ARG;
    if (num_tokens == 0) return SWL_OK;
    if (hidden % swl::kAddScaleChunk) return SWL_ERR_UNSUPPORTED;
    if (!x_scaled || !residual || !w || !slabs || !ssq_out || !swl::aligned16(x_scaled) || !swl::aligned16(residual) ||
        !swl::aligned16(w) || !swl::aligned16(slabs))
        return SWL_ERR_BAD_ARG;
    if (num_tokens > 65535) return SWL_ERR_UNSUPPORTED;
    const dim3 grid(hidden / swl::kAddScaleChunk, static_cast<unsigned>(num_tokens));
    SWL_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL((swl::splitk_add_scale_kernel<T>), grid, dim3(swl::kAddScaleThreads), 0,
                           static_cast<hipStream_t>(stream), static_cast<T *>(x_scaled), static_cast<T *>(residual),
                           static_cast<const T *>(w), slabs, k_splits, num_tokens * static_cast<int64_t>(hidden), ssq_out,
                           static_cast<int>(num_tokens), hidden);
    });
    return swl::check_launch();
}
