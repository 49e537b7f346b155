// gemm_skinny.hip — weight-streaming GEMM for decode batches (M <= 32 tokens) on gfx950.
//
// out[M, N] = x[M, K] . W[N, K]^T  — the shape of every projection of a decode step
// (reference: swiftllm/worker/kernels/linear.py:3-12 called from transformer_layer.py:54-56,117,
// 126,128 and post_layer.py:38). At M <= 32 the op is pure HBM streaming of W (77 % of all bytes a
// decode step moves); algorithmic bytes = N*K*e (+ M*K*e + M*N*e).
//
// Mapping (not a tiled compute GEMM):
//   * one wave owns 32 consecutive rows of W (output columns) over one K-chunk and issues
//     v_mfma_f32_32x32x16 with A = W rows, B = x^T: the 32x32 accumulator is out^T[n][m], so all
//     M <= 32 tokens ride along with every weight byte exactly once;
//   * loads are 64 contiguous bytes per lane per super-step (4 x dwordx4, non-temporal): lane
//     (row = l%32, half = l/32) covers k0 + half*32 .. +32, so a row contributes whole 128-byte lines;
//     the k-order inside a super-step is permuted identically for W and x (a dot product does not
//     care), which is what makes the contiguous per-lane run legal for the MFMA operand layout;
//   * PD super-steps (PD x 4 KiB of W per wave) are in flight while the oldest is consumed;
//     x comes from L2 with the same addressing (rows >= M are clamped and never stored);
//   * K is split across the KS waves of a workgroup (so even N = 4096 launches >= 1024 waves) and
//     reduced through LDS in a fixed order: deterministic, no atomics, one rounding at the store.
#include "swl_common.h"

namespace swl {

__device__ __forceinline__ float16_t mfma32x32x16(vec8_t<f16> a, vec8_t<f16> b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float16_t mfma32x32x16(vec8_t<bf16> a, vec8_t<bf16> b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

constexpr int kSS = 64;          // k elements per super-step
constexpr int kTileStride = 36;  // floats per m-row of a partial tile in LDS (32 + pad)

// KS = k-splits per workgroup, NTW = 32-row tiles per workgroup, PD = super-steps in flight.
template <typename T, int KS, int NTW, int PD>
__global__ __launch_bounds__(KS * NTW * 64) void gemm_skinny_kernel(
    T *__restrict__ out, const T *__restrict__ x, const T *__restrict__ w, int M, int N, int K,
    int64_t x_stride, int64_t out_stride) {
    constexpr int NWAVES = KS * NTW;
    __shared__ __attribute__((aligned(16))) float part[NWAVES][32 * kTileStride];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ks = wave % KS;
    const int nt = wave / KS;
    const int n0 = (blockIdx.x * NTW + nt) * 32;
    const int l32 = lane & 31;
    const int hf = lane >> 5;
    const int kc = K / KS;
    const int nss = kc / kSS;
    const bool tile_ok = n0 < N; // N % 32 == 0: a tile is either fully inside or fully outside

    float16_t acc = float16_t{};
    if (tile_ok) {
        const T *wp = w + static_cast<int64_t>(n0 + l32) * K + ks * kc + hf * 32;
        const T *xp = x + static_cast<int64_t>(min(l32, M - 1)) * x_stride + ks * kc + hf * 32;
        vec8_t<T> wv[PD][4], xv[PD][4];
        // Every tile walks its K-chunk from a different starting super-step (wrapping around): rows
        // of W are a power-of-two pitch apart, so waves marching in lockstep would all sit on the
        // same DRAM channel phase. The order of the fp32 accumulation depends only on the tile index
        // (deterministic).
        const int rot = ((blockIdx.x * NTW + nt) * 5 + ks * 3) % nss;
        auto issue = [&](int slot, int ss_linear) {
            int ss = ss_linear + rot;
            ss -= ss >= nss ? nss : 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // default cache policy on purpose: a 128-byte line of W is consumed by 8 separate
                // 16-byte loads (4 per lane of a lane pair); the non-temporal hint made every one of
                // them a fresh trip (measured 2x slower)
                wv[slot][j] = load8(wp + ss * kSS + j * 8);
                xv[slot][j] = load8(xp + ss * kSS + j * 8);
            }
        };
        auto consume = [&](int slot) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = mfma32x32x16(wv[slot][j], xv[slot][j], acc);
        };
        // Software pipeline over groups of PD super-steps. The steady-state body has NO branches:
        // hipcc then emits exact counted waits (vmcnt((PD-1)*8)) before each slot's MFMAs; with a
        // guard per slot it falls back to vmcnt(0) at the loop head and the pipeline collapses.
        const int nfull = nss / PD * PD;
        int ss = 0;
        if (nfull > 0) {
#pragma unroll
            for (int p = 0; p < PD; ++p) issue(p, p);
            for (; ss + PD < nfull; ss += PD) {
#pragma unroll
                for (int p = 0; p < PD; ++p) {
                    consume(p);
                    issue(p, ss + p + PD);
                }
            }
#pragma unroll
            for (int p = 0; p < PD; ++p) consume(p);
            ss = nfull;
        }
        for (; ss < nss; ++ss) { // fewer than PD super-steps left
            issue(0, ss);
            consume(0);
        }
    }

    // acc[r] = out^T[n = (r&3) + 8*(r>>2) + 4*hf][m = l32]; park it as part[wave][m][n]
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        float4_t v = {acc[4 * r4], acc[4 * r4 + 1], acc[4 * r4 + 2], acc[4 * r4 + 3]};
        *reinterpret_cast<float4_t *>(&part[wave][l32 * kTileStride + 8 * r4 + 4 * hf]) = v;
    }
    __syncthreads();

    // fixed-order reduction over the KS partials of each tile; 8 outputs (16 bytes) per thread
    constexpr int ITEMS = NTW * 32 * 4; // (tile, m, n8)
    for (int it = threadIdx.x; it < ITEMS; it += NWAVES * 64) {
        const int n8 = it & 3;
        const int m = (it >> 2) & 31;
        const int t = it >> 7;
        const int tn0 = (blockIdx.x * NTW + t) * 32;
        if (m >= M || tn0 >= N) continue;
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < KS; ++k2) {
            const float *src = &part[t * KS + k2][m * kTileStride + n8 * 8];
            const float4_t a = *reinterpret_cast<const float4_t *>(src);
            const float4_t b = *reinterpret_cast<const float4_t *>(src + 4);
            s[0] += a[0]; s[1] += a[1]; s[2] += a[2]; s[3] += a[3];
            s[4] += b[0]; s[5] += b[1]; s[6] += b[2]; s[7] += b[3];
        }
        vec8_t<T> ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = to_t<T>(s[e]);
        store8(out + static_cast<int64_t>(m) * out_stride + tn0 + n8 * 8, ov);
    }
}

template <typename T, int KS, int NTW, int PD>
static int launch_gemm(T *out, const T *x, const T *w, int M, int N, int K, int64_t xs, int64_t os,
                       hipStream_t stream) {
    const int tiles = N / 32;
    const dim3 grid((tiles + NTW - 1) / NTW);
    hipLaunchKernelGGL((gemm_skinny_kernel<T, KS, NTW, PD>), grid, dim3(KS * NTW * 64), 0, stream, out,
                       x, w, M, N, K, xs, os);
    return check_launch();
}

// Pick the k-split so that a launch has >= ~2048 waves (8 per CU) whenever K allows it.
template <typename T>
static int dispatch_gemm(T *out, const T *x, const T *w, int M, int N, int K, int64_t xs, int64_t os,
                         int ks_override, hipStream_t stream) {
    const int tiles = N / 32;
    int ks = 1;
    while (ks < 8 && tiles * ks < 2048 && K % (kSS * ks * 2) == 0) ks *= 2;
    if (ks_override > 0) ks = ks_override;
    if (K % (kSS * ks) != 0) return SWL_ERR_UNSUPPORTED;
    switch (ks) {
    case 1: return launch_gemm<T, 1, 8, 4>(out, x, w, M, N, K, xs, os, stream);
    case 2: return launch_gemm<T, 2, 4, 4>(out, x, w, M, N, K, xs, os, stream);
    case 4: return launch_gemm<T, 4, 2, 4>(out, x, w, M, N, K, xs, os, stream);
    case 8: return launch_gemm<T, 8, 1, 4>(out, x, w, M, N, K, xs, os, stream);
    default: return SWL_ERR_UNSUPPORTED;
    }
}

} // namespace swl

extern "C" int swl_gemm_skinny(void *out, const void *x, const void *w, int32_t M, int32_t N,
                               int32_t K, int64_t x_row_stride, int64_t out_row_stride,
                               int32_t k_splits, int32_t dtype, swl_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!out || !x || !w) return SWL_ERR_BAD_ARG;
    if (M > 32 || (N & 31) || (K & 63)) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || out_row_stride < N || (x_row_stride & 7) || (out_row_stride & 7))
        return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(out) || !swl::aligned16(x) || !swl::aligned16(w)) return SWL_ERR_BAD_ARG;
    if (k_splits != 0 && k_splits != 1 && k_splits != 2 && k_splits != 4 && k_splits != 8)
        return SWL_ERR_BAD_ARG;
    SWL_DISPATCH_DTYPE(dtype, T, {
        return swl::dispatch_gemm<T>(static_cast<T *>(out), static_cast<const T *>(x),
                                     static_cast<const T *>(w), M, N, K, x_row_stride,
                                     out_row_stride, k_splits, static_cast<hipStream_t>(stream));
    });
}
