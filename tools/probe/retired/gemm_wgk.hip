// gemm_wgk.hip — decode projections whose K is split INSIDE the workgroup (gfx950, M <= 32 tokens, packed W).
//
// out[M, N] = x[M, K] . W[N, K]^T for the two hidden-by-hidden projections of a decode layer (reference:
// swiftllm/worker/kernels/linear.py:3-12 called from transformer_layer.py:54-56 (q/k/v) and :117 (o_proj)), with what
// follows them in the layer folded into the epilogue (transformer_layer.py:120: fused_add_rmsnorm, rmsnorm.py:67-89).
//
// Why a second split-K scheme. gemm_skinny.hip splits K ACROSS workgroups: every split writes an fp32 slab and a
// consumer kernel adds the slabs. For K = hidden that hand-off is the expensive part of the projection on MI355X
// (profiles/r02b: o_proj 9.1 us + its add+scale consumer 5.0 us for 33.5 MB of weights = 4.2 us of HBM time; the
// slab-fed attention prologue costs 3-5 us over the plain kernel). Here one workgroup owns a 32-row tile of W for ALL
// of K: its 8 waves take one eighth of K each (a contiguous 32 KiB run of the packed weight for K = 4096), add their
// accumulators through LDS in wave order — the same order and the same bits as the 8-slab sum of the cross-workgroup
// scheme — and the workgroup finishes the job itself: one rounding, residual add, next norm's element-wise half.
// No slab traffic, no counters, no second launch.
//
// The price is activation traffic: waves that share rows of W do not share x, so every workgroup reads all of
// x[32, K] from L2 (N/32 x 64 KiB x K/1024). For K = hidden = 4096 that is 32-48 MB of L2 reads next to 33-50 MB of
// HBM reads — affordable; for down_proj (K = 14336, 117 MB of x re-reads per 117 MB of W) it is not, unless the
// batch is small (rows >= M are clamped to row M-1: at M <= 8 the x traffic is a quarter or less). The host side
// (kernels/linear.py) picks accordingly.
//
// x tiles go global -> VGPR -> wave-private XOR-swizzled LDS tile -> B fragments (the staging map of
// gemm_skinny_kernel, no barriers in the K loop); W fragments go global -> VGPR -> MFMA (packed order).
#include "swl_common.h"

namespace swl {

__device__ __forceinline__ float16_t wgk_mfma(vec8_t<f16> a, vec8_t<f16> b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float16_t wgk_mfma(vec8_t<bf16> a, vec8_t<bf16> b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

constexpr int kWgkKT = 128;     // k elements per tile, as in gemm_skinny.hip (one packed tile = 8 fragments = 8 KiB)
constexpr int kWgkWaves = 8;    // K splits = waves per workgroup (2 per SIMD: 256 registers each)
constexpr int kWgkRedPitch = 36; // floats per token row of a wave's accumulator image in LDS (144 B: conflict-free b128)

enum WgkEpi {
    kWgkDirect = 0,   // out = round(rs[m] * acc)            (T or fp32)
    kWgkAddScale = 1, // residual += round(acc); xs = round(residual * norm_w); ssq_out[tile][m] = sum of squares
};

struct WgkArgs {
    void *out;
    const void *x;
    const void *wp;
    int M, N, K;
    int64_t x_stride, out_stride;
    // deferred RMSNorm of x (rmsnorm.hip: splitk_add_scale_kernel): rs[m] = 1/sqrt(sum_p ssq_in[p*ssq_stride + m] / K + eps)
    const float *ssq_in;
    int ssq_parts, ssq_stride;
    float eps;
    // kWgkAddScale
    void *residual;
    const void *norm_w;
    void *xs;
    float *ssq_out; // [N/32][32]
};

template <typename T, int NKT, int EPI, bool OUT_F32>
__global__ __launch_bounds__(kWgkWaves * 64, 1) void gemm_wgk_kernel(WgkArgs a) {
    constexpr int NW = kWgkWaves;
    // per wave: the x tile [32 tokens][128 k], 16-byte slots XOR-swizzled per row; reused for the accumulator image
    __shared__ __attribute__((aligned(16))) T xl_all[NW][32 * kWgkKT];
    typedef T vec4 __attribute__((ext_vector_type(4)));

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int M = a.M, N = a.N, K = a.K;
    const int tile = blockIdx.x;
    const int n0 = tile * 32;
    const int kc = K / NW;
    const int k_begin = wave * kc;
    const int nkt = NKT > 0 ? NKT : kc / kWgkKT;

    // ---- epilogue operands first: the oldest requests of the wave, never inside a wait of the stream ----
    // threads 0..255 -> (token er, columns n0 + 4*ec .. +3)
    const int er = threadIdx.x >> 3, ec = threadIdx.x & 7;
    const bool epi_thread = threadIdx.x < 256;
    const bool epi_store = epi_thread && er < M;
    float ssv[2] = {0.f, 0.f};
    vec4 rv = {}, nv = {};
    if (epi_thread) {
        const int m = min(er, M - 1);
        if (a.ssq_in != nullptr) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (ec + 8 * q < a.ssq_parts) ssv[q] = a.ssq_in[(ec + 8 * q) * a.ssq_stride + m];
        }
        if constexpr (EPI == kWgkAddScale) {
            rv = *reinterpret_cast<const vec4 *>(static_cast<const T *>(a.residual) + static_cast<int64_t>(m) * N + n0 + 4 * ec);
            nv = *reinterpret_cast<const vec4 *>(static_cast<const T *>(a.norm_w) + n0 + 4 * ec);
        }
    }

    // ---- the stream ----
    const int rsub = lane >> 4, chunk = lane & 15;
    const T *wsrc = static_cast<const T *>(a.wp) +
                    (static_cast<int64_t>(tile) * (K / 16) + k_begin / 16) * 512 + lane * 8;
    const T *xsrc = static_cast<const T *>(a.x) + k_begin + chunk * 8;
    int xrow_off[8], lds_wr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + rsub;
        xrow_off[i] = min(row, M - 1) * static_cast<int>(a.x_stride); // rows >= M: clamped (L1 hits), never stored
        lds_wr[i] = row * kWgkKT + ((chunk ^ (row & 15)) << 3);
    }
    const int l32 = lane & 31, hf = lane >> 5;
    T *xl = &xl_all[wave][0];

    // W fragments ride a 3-deep register ring (two 8 KiB tiles in flight while one is multiplied), x tiles a 2-deep
    // one, requested BEFORE the W tile of the same step: loads return in order, so waiting for x(t) leaves x(t+1),
    // W(t+1) and W(t+2) in flight. NKT > 0: the tile count is a compile-time constant and the whole schedule is
    // straight-line code (exact counted waits); NKT == 0: same schedule behind run-time guards.
    vec8_t<T> wr[3][8], xr[2][8];
    float16_t acc = float16_t{};
#define SWL_WGK_ISSUE_X(slot, t)                                                                               \
    { _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) xr[slot][i_] = load8(xsrc + xrow_off[i_] + (t) * kWgkKT); }
#define SWL_WGK_ISSUE_W(slot, t)                                                                               \
    { _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                                         \
          wr[slot][i_] = load8_nt(wsrc + (static_cast<int64_t>(t) * 8 + i_) * 512); }
#define SWL_WGK_PROCESS(ws, xs)                                                                                \
    {                                                                                                          \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                                       \
            *reinterpret_cast<vec8_t<T> *>(xl + lds_wr[i_]) = xr[xs][i_];                                      \
        _Pragma("unroll") for (int kk_ = 0; kk_ < kWgkKT / 16; ++kk_) {                                        \
            const int off_ = l32 * kWgkKT + (((2 * kk_ + hf) ^ (l32 & 15)) << 3);                              \
            const vec8_t<T> b_ = *reinterpret_cast<const vec8_t<T> *>(xl + off_);                              \
            acc = wgk_mfma(wr[ws][kk_], b_, acc);                                                              \
        }                                                                                                      \
    }
    SWL_WGK_ISSUE_X(0, 0);
    SWL_WGK_ISSUE_W(0, 0);
    if (1 < nkt) SWL_WGK_ISSUE_W(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int kOuter = NKT > 0 ? (NKT + 5) / 6 : 1;
    if constexpr (NKT > 0) {
#pragma unroll
        for (int t0 = 0; t0 < kOuter * 6; t0 += 6) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int t = t0 + j;
                if (t < NKT) {
                    if (t + 1 < NKT) SWL_WGK_ISSUE_X((j + 1) % 2, t + 1);
                    if (t + 2 < NKT) SWL_WGK_ISSUE_W((j + 2) % 3, t + 2);
                    // pinned: left alone, the scheduler sinks the requests between the MFMAs of the tile they follow
                    // (fewer live registers, but nothing in flight while the wave multiplies)
                    __builtin_amdgcn_sched_barrier(0);
                    SWL_WGK_PROCESS(j % 3, j % 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    } else {
        for (int t0 = 0; t0 < nkt; t0 += 6) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int t = t0 + j;
                if (t < nkt) {
                    if (t + 1 < nkt) SWL_WGK_ISSUE_X((j + 1) % 2, t + 1);
                    if (t + 2 < nkt) SWL_WGK_ISSUE_W((j + 2) % 3, t + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    SWL_WGK_PROCESS(j % 3, j % 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
#undef SWL_WGK_ISSUE_X
#undef SWL_WGK_ISSUE_W
#undef SWL_WGK_PROCESS

    // ---- in-workgroup reduction: acc[r] = out^T[n = (r&3) + 8*(r>>2) + 4*hf][m = l32] -> red[wave][m][n] ----
    // (a wave's image goes into its OWN x tile: its LDS reads above were issued before these writes, in order)
    mfma_results_ready<8>(acc); // acc is stored by DS instructions next (swl_common.h)
    float *red = reinterpret_cast<float *>(xl);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const float4_t v = {acc[4 * r4], acc[4 * r4 + 1], acc[4 * r4 + 2], acc[4 * r4 + 3]};
        *reinterpret_cast<float4_t *>(red + l32 * kWgkRedPitch + 8 * r4 + 4 * hf) = v;
    }
    __syncthreads();
    if (!epi_thread) return;
    float4_t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) // wave order = K order: the bits of the 8-slab sum of gemm_skinny.hip
        s += *reinterpret_cast<const float4_t *>(reinterpret_cast<const float *>(&xl_all[w][0]) + er * kWgkRedPitch + 4 * ec);
    float rs = 1.0f; // (x * 1.0f is exact: one code path)
    if (a.ssq_in != nullptr) {
        const float tot = group_allreduce_sum<8>(ssv[0] + ssv[1]);
        rs = 1.0f / sqrtf(tot / static_cast<float>(K) + a.eps); // rmsnorm.hip's formula
    }
    if constexpr (EPI == kWgkDirect) {
        if (epi_store) {
            if constexpr (OUT_F32) {
                *reinterpret_cast<float4_t *>(static_cast<float *>(a.out) + static_cast<int64_t>(er) * a.out_stride + n0 + 4 * ec) = s * rs;
            } else {
                vec4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = to_t<T>(s[e] * rs);
                *reinterpret_cast<vec4 *>(static_cast<T *>(a.out) + static_cast<int64_t>(er) * a.out_stride + n0 + 4 * ec) = o;
            }
        }
    } else {
        // splitk_add_scale_kernel's arithmetic (rmsnorm.hip) on this thread's 4 columns
        vec4 xn, sv;
        float ssq = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            xn[e] = add_t<T>(to_t<T>(s[e] * rs), rv[e]); // the projection is rounded, then the sum (rmsnorm.py:54-57)
            const float v = to_f(xn[e]);
            ssq = fmaf(v, v, ssq);
            sv[e] = to_t<T>(v * to_f(nv[e]));
        }
        ssq = group_allreduce_sum<8>(ssq);
        if (epi_store) {
            const int64_t off = static_cast<int64_t>(er) * N + n0 + 4 * ec;
            *reinterpret_cast<vec4 *>(static_cast<T *>(a.residual) + off) = xn;
            *reinterpret_cast<vec4 *>(static_cast<T *>(a.xs) + off) = sv;
            if (ec == 0) a.ssq_out[tile * 32 + er] = ssq;
        }
    }
}

static bool wgk_shape_ok(int M, int N, int K) {
    return M > 0 && M <= 32 && N > 0 && (N & 31) == 0 && K > 0 && K % (kWgkKT * kWgkWaves) == 0;
}

template <typename T, int EPI, bool OUT_F32>
static int launch_wgk(const WgkArgs &a, hipStream_t stream) {
    const dim3 grid(a.N / 32), block(kWgkWaves * 64);
    const int nkt = a.K / (kWgkKT * kWgkWaves);
    // K = 4096 (Llama-3-8B / Llama-2-7B hidden), 8192 (70B hidden), 14336 (Llama-3-8B FFN): straight-line schedules
    if (nkt == 4) hipLaunchKernelGGL((gemm_wgk_kernel<T, 4, EPI, OUT_F32>), grid, block, 0, stream, a);
    else if (nkt == 8) hipLaunchKernelGGL((gemm_wgk_kernel<T, 8, EPI, OUT_F32>), grid, block, 0, stream, a);
    else if (nkt == 14) hipLaunchKernelGGL((gemm_wgk_kernel<T, 14, EPI, OUT_F32>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((gemm_wgk_kernel<T, 0, EPI, OUT_F32>), grid, block, 0, stream, a);
    return check_launch();
}

} // namespace swl

extern "C" int swl_gemm_wgk_supported(int32_t M, int32_t N, int32_t K) {
    return swl::wgk_shape_ok(M, N, K) ? 1 : 0;
}

extern "C" int swl_gemm_wgk(void *out, int32_t out_fp32, const void *x, const void *w_packed, const float *row_ssq,
                            int32_t ssq_parts, int32_t ssq_stride, float eps, int32_t M, int32_t N, int32_t K,
                            int64_t x_row_stride, int64_t out_row_stride, int32_t dtype, swl_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!out || !x || !w_packed) return SWL_ERR_BAD_ARG;
    if (!swl::wgk_shape_ok(M, N, K)) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || out_row_stride < N || (x_row_stride & 7) || (out_row_stride & 3)) return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(x) || !swl::aligned16(w_packed) || !swl::aligned16(out)) return SWL_ERR_BAD_ARG;
    if (row_ssq && (ssq_parts <= 0 || ssq_parts > 16 || ssq_stride < M)) return SWL_ERR_BAD_ARG;
    swl::WgkArgs a = {};
    a.out = out; a.x = x; a.wp = w_packed;
    a.M = M; a.N = N; a.K = K;
    a.x_stride = x_row_stride; a.out_stride = out_row_stride;
    a.ssq_in = row_ssq; a.ssq_parts = ssq_parts; a.ssq_stride = ssq_stride; a.eps = eps;
    SWL_DISPATCH_DTYPE(dtype, T, {
        if (out_fp32) return swl::launch_wgk<T, swl::kWgkDirect, true>(a, static_cast<hipStream_t>(stream));
        return swl::launch_wgk<T, swl::kWgkDirect, false>(a, static_cast<hipStream_t>(stream));
    });
}

extern "C" int swl_gemm_wgk_add_scale(void *x_scaled, void *residual, float *ssq_out, const void *norm_w, const void *x,
                                      const void *w_packed, int32_t M, int32_t N, int32_t K, int64_t x_row_stride,
                                      int32_t dtype, swl_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!x_scaled || !residual || !ssq_out || !norm_w || !x || !w_packed) return SWL_ERR_BAD_ARG;
    if (!swl::wgk_shape_ok(M, N, K)) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || (x_row_stride & 7)) return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(x) || !swl::aligned16(w_packed) || !swl::aligned16(x_scaled) || !swl::aligned16(residual) ||
        !swl::aligned16(norm_w) || !swl::aligned16(ssq_out))
        return SWL_ERR_BAD_ARG;
    swl::WgkArgs a = {};
    a.x = x; a.wp = w_packed;
    a.M = M; a.N = N; a.K = K;
    a.x_stride = x_row_stride; a.out_stride = N;
    a.residual = residual; a.norm_w = norm_w; a.xs = x_scaled; a.ssq_out = ssq_out;
    SWL_DISPATCH_DTYPE(dtype, T, {
        return swl::launch_wgk<T, swl::kWgkAddScale, false>(a, static_cast<hipStream_t>(stream));
    });
}
