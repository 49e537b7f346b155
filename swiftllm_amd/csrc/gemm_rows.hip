// gemm_rows.hip — the hidden-wide projections of a decode layer with K split INSIDE the workgroup and the residual add in
// the epilogue (gfx950, M <= 32 tokens, packed W).
//
// Replaces, for decode batches of <= 32 (o_proj) / <= 8 (down_proj) tokens in bfloat16, the pair
//     projection (reference swiftllm/worker/kernels/linear.py:3-12 at layers/transformer_layer.py:117 / :128)
//   + the residual-add half of fused_add_rmsnorm (kernels/rmsnorm.py:54-57 at transformer_layer.py:120 / the next layer's :46)
// which r01-r04 ran as a split-K GEMM (8 fp32 slabs, gemm_skinny.hip) followed by a consumer launch that does nothing
// but wait on memory: 9.6 + 5.0 us (o) per layer in the batch-32 trace (profiles/r04d_kernel_trace_configs2.md) for 33.6 MB
// of weights = 4.7 us of HBM time. The norm weight and the 1/rms are left to the projection that follows, which applies
// them while it stages the raw residual rows ("norm on the fly": gemm_skinny.hip, NF).
//
// Why a second in-workgroup scheme after r02's gemm_wgk (128 workgroups of 32 rows: 11.2 us, retired). What bounds such a
// kernel on this part is what ONE CU can pull through its L1: ~57 GB/s, L2 hit or HBM miss (DESIGN.md sections 4.6, 4.8). A
// workgroup that owns rows of W for all of K reads all of x[M, K]: bytes per CU = rows * K * e + M * K * e. With 32-row tiles
// only 128 CUs work and each moves 256 + 256 KiB (9.3 us); with SIXTEEN-row tiles all 256 CUs work and each moves
// 128 + 256 KiB (7 us) — and the op is one v_mfma_f32_16x16x32 per 1 KiB of W, so the tile height costs nothing:
//   * workgroup = 16 rows of W x all of K = 8 waves (2 per SIMD), wave w takes K/8 contiguous columns;
//   * W comes from the SAME packed copy the other decode kernels stream (swl_gemm_pack_weight: 32-row x 16-k fragments
//     of 1 KiB): the 16 x 32 A fragment of tile half h is lanes {16h..16h+15} and {32+16h..} of two consecutive
//     fragments — four 256-byte runs per wave-load, non-temporal; the sibling workgroup reads the other halves;
//   * x^T is the B operand, loaded straight into fragment layout (lane -> token l%16, 8 k): 64-byte runs of 16 rows,
//     L2-resident (every workgroup reads the same rows); rows >= M are clamped to row M-1 (L1 hits, never stored);
//   * EVERYTHING a wave needs is requested before its first MFMA: K = 4096 is 16 k-steps = 16 + 32 loads of 16 B per
//     lane (192 VGPRs), the whole workgroup 384 KiB in flight — the kernel is one memory round trip plus a drain at
//     the CU's own rate; longer K (down_proj: 14 chunks) refills a 4-deep ring of 4-step chunks behind exact counted waits;
//   * the 8 waves' 16 x 32 fp32 tiles meet in LDS (20 KiB), are added in wave order (= K order: deterministic), and the
//     512 threads finish one (token, column) each: residual += round(acc), the consumer launch's rounding points.
// r06b — x through a wave-private LDS tile (XLDS = true: from 9 tokens on). The fragment-shaped x loads above are 16 rows x 64 B
// per instruction: sixteen HALF cache lines per KiB delivered, and at 32 tokens x is 2/3 of what a CU pulls through its L1
// (the r05 "57 GB/s per CU, hit or miss" was measured through exactly that pattern). Here a wave requests its x chunk
// (32 tokens x 128 k) as FULL lines — lane -> (token 4i + lane/16, 16-byte piece lane%16): 4 rows x 256 contiguous bytes per
// instruction, the same register budget (8 loads per chunk at two token blocks) — and turns it into B fragments through an
// 8 KiB LDS tile of its own: ds_write_b128 at piece ^ (token & 15), ds_read_b128 of lane (r, kq) at (4j + kq) ^ r —
// conflict-free for the 8-lane write groups and for the irregular 16-lane groups of the b128 read.
// One wave's LDS operations execute in order: no barrier in the stream. Same MFMAs in the same order: the same bits.
// No slabs, no second launch, no atomics, no cross-workgroup hand-off. (r05 also built an epilogue that wrote
// round(residual * w_norm) and one sum-of-squares partial per tile, with a SiLU-gate GEMM that added the 256 partials from
// LDS-DMA'd copies: correct, +2 us on the up/gate kernel, removed — last revision that has it: ca4e9a9, numbers in
// profiles/r05a_gemm_rows_micro_o_proj_pair.jsonl.)
#include "swl_common.h"

namespace swl {

__device__ __forceinline__ float4_t rows_mfma(vec8_t<f16> a, vec8_t<f16> b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float4_t rows_mfma(vec8_t<bf16> a, vec8_t<bf16> b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

constexpr int kRowsWaves = 8;     // K splits = waves per workgroup
constexpr int kRowsTile = 16;     // rows of W per workgroup
constexpr int kRowsCh = 4;        // k-steps (of 32) per ring chunk
constexpr int kRowsRing = 4;      // chunks in flight per wave: 16 k-steps = 192 VGPRs at two token blocks
constexpr int kRowsRedPitch = 20; // floats per token row of a wave's tile image in LDS (80 B: 16-byte stores spread)

struct RowsArgs {
    const void *x;
    const void *wp;
    void *residual;
    int M, N, K;
    int64_t x_stride;
    // optional (r06c): ssq_out[M][N / 16] = sum of squares of the NEW residual values of (token, 16-column tile) — what an
    // EXACT rmsnorm of the updated rows needs before the next projection can stage them (swl_gemm_skinny_packed_*_nx)
    float *ssq_out;
};

// NCH > 0: the chunk count per wave is a compile-time constant <= kRowsRing — every load is issued up front, straight-line
// code with exact counted waits. NCH == 0: run-time chunk count, the ring refilled behind guards.
template <typename T, int NCH, int MB, bool XLDS>
__global__ __launch_bounds__(kRowsWaves * 64, 2) void gemm_rows_kernel(RowsArgs a) {
    __shared__ __attribute__((aligned(16))) float red[kRowsWaves][32 * kRowsRedPitch];
    constexpr int XCK = 32 * kRowsCh;          // k per chunk (128): one 256-byte run per token
    constexpr int NXL = XLDS ? 4 * MB : kRowsCh * MB;   // x loads per chunk and lane (XLDS: 4 token rows each)
    __shared__ __attribute__((aligned(16))) T xt[XLDS ? kRowsWaves : 1][XLDS ? 16 * MB * XCK : 8];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int M = a.M, N = a.N, K = a.K;
    const int tile16 = blockIdx.x;
    const int n0 = tile16 * kRowsTile;

    // ---- epilogue operands first: the oldest requests of the wave, no wait of the stream ever includes them ----
    const int et = threadIdx.x >> 4, en = threadIdx.x & 15; // (token, column n0 + en)
    const bool e_ok = et < M;
    T *res_p = static_cast<T *>(a.residual) + static_cast<int64_t>(e_ok ? et : 0) * N + n0 + en;
    const T rv = *res_p;

    // ---- the stream ----
    const int r = lane & 15, kq = lane >> 4;
    const int nsteps = K / 32 / kRowsWaves; // k-steps of this wave
    const int s0 = wave * nsteps;
    const int nch = NCH > 0 ? NCH : nsteps / kRowsCh;
    // packed W: fragment (tile32, kstep16) = 512 elements, lane L of it = row L%32, k = 16*kstep16 + 8*(L/32) .. +8.
    // 16x32 A fragment of step s, lane (r, kq): row 16*half + r, k = 32*s + 8*kq -> kstep16 = 2s + kq/2, L = 16*half + r + 32*(kq%2)
    const T *wsrc = static_cast<const T *>(a.wp) +
                    (static_cast<int64_t>(tile16 >> 1) * (K / 16) + 2 * s0 + (kq >> 1)) * 512 +
                    (16 * (tile16 & 1) + r + 32 * (kq & 1)) * 8;
    // fragment form: xsrc[mb] = row r + 16 mb at k = 8 kq; XLDS: xsrc[i] = row 4 i + lane/16 at piece lane%16 of the chunk
    const T *xsrc[XLDS ? NXL : MB];
    int xw_off[XLDS ? NXL : 1];                  // XLDS: where this lane's piece of load i goes in the tile (elements)
    int xr_off[XLDS ? kRowsCh : 1];              // XLDS: B fragment of k-step j of the chunk, token block 0 (elements)
    if constexpr (XLDS) {
        const int tsub = lane >> 4, pc = lane & 15;
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            const int tk = 4 * i + tsub;
            xsrc[i] = static_cast<const T *>(a.x) + static_cast<int64_t>(min(tk, M - 1)) * a.x_stride + 32 * s0 + 8 * pc;
            xw_off[i] = tk * XCK + ((pc ^ (tk & 15)) << 3);
        }
#pragma unroll
        for (int j = 0; j < kRowsCh; ++j) xr_off[j] = r * XCK + (((4 * j + kq) ^ r) << 3);
    } else {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
            xsrc[mb] = static_cast<const T *>(a.x) + static_cast<int64_t>(min(r + 16 * mb, M - 1)) * a.x_stride + 32 * s0 + 8 * kq;
    }
    T *xtw = &xt[XLDS ? wave : 0][0];

    vec8_t<T> wr[kRowsRing][kRowsCh], xr[kRowsRing][NXL];
    float4_t acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = float4_t{0.f, 0.f, 0.f, 0.f};
#define SWL_ROWS_ISSUE(slot, c)                                                                         \
    {                                                                                                   \
        if constexpr (XLDS) {   /* x first: the chunk's x is what the LDS pass waits for (loads return in order) */ \
            _Pragma("unroll") for (int i_ = 0; i_ < NXL; ++i_) xr[slot][i_] = load8(xsrc[i_] + (c) * XCK); \
        }                                                                                               \
        _Pragma("unroll") for (int j_ = 0; j_ < kRowsCh; ++j_) {                                        \
            const int s_ = (c) * kRowsCh + j_;                                                          \
            wr[slot][j_] = load8_nt(wsrc + static_cast<int64_t>(s_) * 1024);                            \
            if constexpr (!XLDS) {                                                                      \
                _Pragma("unroll") for (int mb_ = 0; mb_ < MB; ++mb_) xr[slot][j_ * MB + mb_] = load8(xsrc[mb_] + s_ * 32); \
            }                                                                                           \
        }                                                                                               \
    }
#define SWL_ROWS_PROCESS(slot)                                                                          \
    {                                                                                                   \
        if constexpr (XLDS) {                                                                           \
            _Pragma("unroll") for (int i_ = 0; i_ < NXL; ++i_)                                          \
                *reinterpret_cast<vec8_t<T> *>(xtw + xw_off[i_]) = xr[slot][i_];                        \
        }                                                                                               \
        _Pragma("unroll") for (int j_ = 0; j_ < kRowsCh; ++j_)                                          \
            _Pragma("unroll") for (int mb_ = 0; mb_ < MB; ++mb_) {                                      \
                vec8_t<T> b_;                                                                           \
                if constexpr (XLDS) b_ = *reinterpret_cast<const vec8_t<T> *>(xtw + xr_off[j_] + mb_ * 16 * XCK); \
                else b_ = xr[slot][j_ * MB + mb_];                                                      \
                acc[mb_] = rows_mfma(wr[slot][j_], b_, acc[mb_]);                                       \
            }                                                                                           \
    }
    if constexpr (NCH > 0) {
        // straight-line schedule, exact counted waits: the first min(NCH, ring) chunks are requested up front (K = 4096:
        // the whole K range of the wave), longer K (down_proj: 14 chunks) refills a slot as soon as it is multiplied
#pragma unroll
        for (int c = 0; c < (NCH < kRowsRing ? NCH : kRowsRing); ++c) SWL_ROWS_ISSUE(c, c);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            // pinned: left alone the scheduler sinks requests between the MFMAs of earlier chunks (fewer live registers,
            // and nothing in flight while the wave multiplies)
            __builtin_amdgcn_sched_barrier(0);
            SWL_ROWS_PROCESS(c % kRowsRing);
            __builtin_amdgcn_sched_barrier(0);
            if (c + kRowsRing < NCH) SWL_ROWS_ISSUE(c % kRowsRing, c + kRowsRing);
        }
    } else {
#pragma unroll
        for (int c = 0; c < kRowsRing; ++c)
            if (c < nch) SWL_ROWS_ISSUE(c, c);
        for (int c0 = 0; c0 < nch; c0 += kRowsRing) {
#pragma unroll
            for (int d = 0; d < kRowsRing; ++d) {
                if (c0 + d < nch) {
                    __builtin_amdgcn_sched_barrier(0);
                    SWL_ROWS_PROCESS(d);
                    __builtin_amdgcn_sched_barrier(0);
                    if (c0 + d + kRowsRing < nch) SWL_ROWS_ISSUE(d, c0 + d + kRowsRing);
                }
            }
        }
    }
#undef SWL_ROWS_ISSUE
#undef SWL_ROWS_PROCESS

    // ---- in-workgroup reduction: acc[mb][j] = out^T[n = 4*kq + j][token = r + 16*mb] -> red[wave][token][n] ----
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) mfma_results_tie(acc[mb]);
    mfma_results_ready<4>(acc[MB - 1]); // stored by DS instructions next (swl_common.h)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
        *reinterpret_cast<float4_t *>(&red[wave][(r + 16 * mb) * kRowsRedPitch + 4 * kq]) = acc[mb];
    __syncthreads();
    float s = 0.f;
    if (MB == 2 || et < 16) {
#pragma unroll
        for (int w = 0; w < kRowsWaves; ++w) s += red[w][et * kRowsRedPitch + en]; // wave order = K order
    }
    // splitk_add_scale_kernel's arithmetic (rmsnorm.hip) on this thread's element
    const T xn = add_t<T>(to_t<T>(s), rv); // the projection is rounded, then the sum (rmsnorm.py:54-57)
    if (e_ok) *res_p = xn;
    if (a.ssq_out != nullptr) {             // (wave-uniform) the 16 column lanes of a token: one DPP row reduction
        const float v = e_ok ? to_f(xn) : 0.f;
        const float t = group_allreduce_sum<16>(v * v);
        if (e_ok && en == 0) a.ssq_out[static_cast<int64_t>(et) * (N / kRowsTile) + tile16] = t;
    }
}

static bool rows_shape_ok(int M, int N, int K) {
    return M > 0 && M <= 32 && N > 0 && N % 32 == 0 && K > 0 && K % (32 * kRowsWaves * kRowsCh) == 0;
}

template <typename T, int MB, bool XLDS>
static int launch_rows(const RowsArgs &a, hipStream_t stream) {
    const dim3 grid(a.N / kRowsTile), block(kRowsWaves * 64);
    const int nch = a.K / (32 * kRowsWaves * kRowsCh);
    // K = 4096 (Llama-3-8B / Llama-2-7B hidden): 4 chunks, all in flight; 14336 (Llama-3-8B FFN): 14; 8192: 8
    if (nch == 4) hipLaunchKernelGGL((gemm_rows_kernel<T, 4, MB, XLDS>), grid, block, 0, stream, a);
    else if (nch == 14) hipLaunchKernelGGL((gemm_rows_kernel<T, 14, MB, XLDS>), grid, block, 0, stream, a);
    else if (nch == 8) hipLaunchKernelGGL((gemm_rows_kernel<T, 8, MB, XLDS>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((gemm_rows_kernel<T, 0, MB, XLDS>), grid, block, 0, stream, a);
    return check_launch();
}

// x through LDS pays from 9 tokens on (graph-timed on MI355X, profiles/r06b_rows_kernel_us.jsonl: o_proj 32 tokens 11.7 ->
// 10.5 us, down_proj 16 / 32 tokens 25.0 -> 23.8 / 34.4 -> 28.4; at one token the LDS hop costs o_proj 0.45 us, at 16 the
// two forms tie). A/B switch for measurements: SWL_ROWS_X=frag / lds forces one form for every M (same bits either way).
static bool rows_x_through_lds(int M) {
    static const int forced = [] { const char *e = getenv("SWL_ROWS_X"); return !e ? 0 : e[0] == 'f' ? 1 : e[0] == 'l' ? 2 : 0; }();
    return forced ? forced == 2 : M > 8;
}

} // namespace swl

extern "C" int swl_gemm_rows_supported(int32_t M, int32_t N, int32_t K) { return swl::rows_shape_ok(M, N, K) ? 1 : 0; }

static int rows_add_impl(void *residual, float *ssq_out, const void *x, const void *w_packed, int32_t M, int32_t N, int32_t K,
                         int64_t x_row_stride, int32_t dtype, swl_stream_t stream);

extern "C" int swl_gemm_rows_add(void *residual, const void *x, const void *w_packed, int32_t M, int32_t N, int32_t K,
                                 int64_t x_row_stride, int32_t dtype, swl_stream_t stream) {
    return rows_add_impl(residual, nullptr, x, w_packed, M, N, K, x_row_stride, dtype, stream);
}

/* swl_gemm_rows_add + ssq_out[M][N / 16] (fp32): per (token, 16-column tile) the sum of squares of the updated residual. */
extern "C" int swl_gemm_rows_add_ssq(void *residual, float *ssq_out, const void *x, const void *w_packed, int32_t M, int32_t N,
                                     int32_t K, int64_t x_row_stride, int32_t dtype, swl_stream_t stream) {
    if (M > 0 && (!ssq_out || (reinterpret_cast<uintptr_t>(ssq_out) & 3u))) return SWL_ERR_BAD_ARG;
    return rows_add_impl(residual, ssq_out, x, w_packed, M, N, K, x_row_stride, dtype, stream);
}

static int rows_add_impl(void *residual, float *ssq_out, const void *x, const void *w_packed, int32_t M, int32_t N, int32_t K,
                         int64_t x_row_stride, int32_t dtype, swl_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!residual || !x || !w_packed) return SWL_ERR_BAD_ARG;
    if (!swl::rows_shape_ok(M, N, K)) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || (x_row_stride & 7)) return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(x) || !swl::aligned16(w_packed) || (reinterpret_cast<uintptr_t>(residual) & 1u)) return SWL_ERR_BAD_ARG;
    swl::RowsArgs a = {};
    a.x = x; a.wp = w_packed; a.residual = residual;
    a.M = M; a.N = N; a.K = K; a.x_stride = x_row_stride; a.ssq_out = ssq_out;
    SWL_DISPATCH_DTYPE(dtype, T, {
        const hipStream_t s = static_cast<hipStream_t>(stream);
        if (swl::rows_x_through_lds(M))
            return M <= 16 ? swl::launch_rows<T, 1, true>(a, s) : swl::launch_rows<T, 2, true>(a, s);
        return M <= 16 ? swl::launch_rows<T, 1, false>(a, s) : swl::launch_rows<T, 2, false>(a, s);
    });
}
