// gemm_wide.hip — decode projections for LARGE batches (64 < M <= 256 tokens) on packed weights, gfx950.
//
// out[M, N] = x[M, K] . W[N, K]^T (reference: swiftllm/worker/kernels/linear.py:3-12, called from
// transformer_layer.py:54-56,117,126,128), for the batch sizes a 264 GB KV pool is sized for. Up to 32 tokens the
// product is pure weight streaming (gemm_skinny.hip); up to 64 two token blocks share every weight fragment
// (gemm_packed_mt_kernel). Beyond that hipBLASLt was the fallback, and its 256x256 tiles leave most of the chip idle on
// the narrow projections of a decode step (N = 4096 / 6144 at M = 256: 16-24 tiles on 256 CUs; qkv 36 us for 50 MB,
// down 48 us for 117 MB — profiles/r03_blas_m_sweep_decode_sizes.jsonl). At these M the op sits between streaming and
// a GEMM: W (N*K*e bytes) still has to come from HBM exactly once, but x (M*K*e) is re-read from L2 by every workgroup
// and the matrix cores are busy for more than half of the stream time. Structure:
//   * W stays in MFMA-fragment order (swl_gemm_pack_weight): global -> VGPR -> MFMA A operand, non-temporal, a
//     3/4-deep register ring of 64-column K-tiles (4 KiB per wave per tile, 8-12 KiB per wave in flight);
//   * a wave owns 32 rows of W for ALL M tokens: MT = ceil(M/32) <= 8 accumulators of 32x32 (128 VGPRs at MT = 8),
//     every weight fragment feeds MT MFMAs, consecutive MFMAs never share an accumulator;
//   * x^T is the B operand: the workgroup's NWV waves share one [32*MT tokens][64 k] tile in LDS (row pitch 128 B,
//     16-byte slot = chunk ^ ((row >> 1) & 7): conflict-free for the 8-lane ds_write_b128 groups and for the 16-lane
//     groups of the ds_read_b128 fragment reads), double-buffered, staged global -> VGPR -> LDS with full-line loads
//     one tile ahead, one barrier per K-tile (32 MFMAs per wave between barriers at MT = 8);
//   * NWV = 8 (512 threads, 256 rows of W per workgroup: x is re-read N/256 times) for wide projections, NWV = 4
//     (two workgroups per CU, x re-read N/128 times but twice the workgroups) for narrow ones; K is split across
//     workgroups into fp32 slabs as far as that keeps <= 256 workgroups and K-chunks of >= 16 (8 beyond 128 tokens)
//     K-tiles — gemm_wide_plan(), the measured optimum of profiles/r04d_gemm_wide_micro.jsonl. (r04's header claimed a
//     cap "slab traffic 2 * ks * M * N * 4 bytes below the weight bytes"; the plan never enforced one and should not:
//     o_proj at 256 tokens runs ks = 8 — 64 MiB of slab traffic for 32 MiB of weights — in 30.1 us, 35.7 us at the
//     ks = 4 such a cap would force: filling the chip beats the L2-resident slab round trip. ADVICE r04.)
//   * SiLU-gate mode: the first half of the waves own `up` tiles, the second half the `gate` tiles of the same
//     columns; activated gate tiles change hands through the (then idle) x buffers, MT/2 token blocks per round;
//     same rounding points as linear -> silu_and_mul_inplace (silu_and_mul.py:16-23).
//   * r06d, two variants of the same loop, each shipped where it measured faster (gemm_wide_plan):
//     TS = 2 — a wave owns 64 rows of W x half of the token blocks, every B fragment feeds two MFMAs (short K-chunks: qkv /
//     o up to 192 tokens); BQ = 2 — the B fragments of k-step kk+1 requested before the MFMAs of kk (SiLU-gate mode up to
//     192 tokens). Timing-only ablations of the loop: tools/make_gemm_wide_ablations.py.
// Bits: every output is the fp32 sum over k in ascending 16-element steps inside a K-chunk, chunks added in slab
// order — the summation order of the M <= 64 kernels at the same split count.
#include "swl_common.h"

extern "C" int swl_splitk_reduce(void *out, const float *slabs, int32_t k_splits, int32_t M, int32_t N,
                                 int64_t out_row_stride, int32_t dtype, swl_stream_t stream);

namespace swl {

__device__ __forceinline__ float16_t mfma_w(vec8_t<f16> a, vec8_t<f16> b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float16_t mfma_w(vec8_t<bf16> a, vec8_t<bf16> b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

constexpr int kWT = 64;      // k elements per tile

enum WideMode { kWideDirect = 0, kWidePartial = 1, kWideSiluGate = 2 };

// 4-wave workgroups with more than 4 token blocks run ONE wave per SIMD (launch bound 1: the whole 512-register file per
// wave — 96-128 accumulator registers, two x register sets and a 6-deep weight ring, 20 KiB per wave in flight).
//
// TS (token split, r06d): with TS = 1 every MFMA reads its own B fragment from LDS — 1 KiB per MFMA, and four waves doing
// that ask the CU's 128 B/clk LDS port for exactly as many cycles as the matrix cores are busy. TS = 2 tiles the wave's
// registers in BOTH directions: a wave owns TWO 32-row blocks of W (64 rows) and HALF of the token blocks, so each B
// fragment feeds two MFMAs (LDS reads halve) while the accumulator count stays MT; its sibling wave takes the other token
// half of the same rows (W is requested by both: the second request is an L1 / L2 hit). Same workgroup footprint (NWV * 32
// rows of W x all tokens), same K order per output: the bits do not change.
template <typename T, int MT, int NWV, int MODE, int TS>
__global__ __launch_bounds__(NWV * 64, (NWV == 4 && (MT > 4 || TS == 2 || MODE == kWideSiluGate)) ? 1 : 2) void gemm_packed_wide_kernel(
    void *__restrict__ out_, const T *__restrict__ x, const T *__restrict__ wpk, int M, int N, int K, int kc,
    int64_t x_stride, int64_t out_stride) {
    constexpr int NT = NWV * 64;
    constexpr int RB = TS;                       // 32-row blocks of W per wave
    constexpr int MTW = MT / TS;                 // token blocks per wave
    static_assert(MT % TS == 0 && (NWV / 2) % TS == 0, "token split must divide the token blocks and the wave halves");
    constexpr int XL = MT * 256 / NT;            // 16-byte chunks of the x tile each thread stages
    constexpr int RPP = NT / 8;                  // rows per staging pass (8 chunks per row)
    constexpr int kXTile = MT * 32 * kWT;
    constexpr int HW = NWV / 2;
    // x prefetch distance in steps: 2 where two register sets fit beside the accumulators, else 1; W ring depth in tiles
    constexpr bool kBig = NWV == 4 && (MT > 4 || TS == 2 || MODE == kWideSiluGate);   // one wave per SIMD
    constexpr int BQ = (NWV == 4 && MODE == kWideSiluGate && MT <= 6) ? 2 : 1;   // B-fragment register sets (SWL_W_PROCESS)
    constexpr int XD = (MT >= 8 && !kBig) ? 1 : 2;
    constexpr int kWD = TS == 2 ? (MT >= 8 ? 3 : 4) : (kBig ? 6 : 4);   // a TS = 2 slot is two fragments: 8 KiB per wave
    static_assert(MT * 256 % NT == 0, "x tile must split evenly over the workgroup");
    __shared__ __attribute__((aligned(16))) T xs[2 * kXTile];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_gate = MODE == kWideSiluGate && wave >= HW;
    // wave -> (row group, token part): TS consecutive waves share a row group; SiLU-gate mode numbers them inside each half
    const int wslot = MODE == kWideSiluGate ? wave % HW : wave;
    const int tp = wslot % TS;
    const int col0 = ((MODE == kWideSiluGate ? blockIdx.x * (HW / TS) : blockIdx.x * (NWV / TS)) + wslot / TS) * (32 * RB);
    const bool tile_ok = col0 < N;
    const int nt = tile_ok ? (col0 + (is_gate ? N : 0)) / 32 : 0;
    const int ksplit = blockIdx.y;
    const int k_begin = ksplit * kc;
    const int nkt = kc / kWT;
    const T *wsrc = wpk + (static_cast<int64_t>(nt) * (K / 16) + k_begin / 16) * 512 + lane * 8;
    const int srow = tid >> 3, chunk = tid & 7;
    // staging: thread -> (row q*RPP + srow, chunk); 32-bit row offsets from one base pointer (the host side checks
    // M * x_stride < 2^31), LDS slot of pass q = slot of pass 0 + q*RPP rows (RPP is a multiple of 16: same swizzle)
    const T *xb = x + k_begin + chunk * 8;
    int xoff[XL];
#pragma unroll
    for (int q = 0; q < XL; ++q) xoff[q] = min(q * RPP + srow, M - 1) * static_cast<int>(x_stride);
    const int xs_wr0 = srow * kWT + ((chunk ^ ((srow >> 1) & 7)) << 3);
    const int l32 = lane & 31, hf = lane >> 5;
    const int swz = (l32 >> 1) & 7;

    const int64_t wrb = static_cast<int64_t>(K / 16) * 512;   // the next 32-row block of W
    // W is streamed once: non-temporal — except with TS = 2, where the sibling wave asks for the same lines a moment later
    auto load_w = [](const T *p) { return TS == 2 ? load8(p) : load8_nt(p); };
    vec8_t<T> wr[kWD][RB * 4], xr[XD][XL];
    float16_t acc[MT];                                         // [row block][token block of this wave]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = float16_t{};

#define SWL_W_ISSUE_W(slot, tile)                                                                     \
    {                                                                                                \
        _Pragma("unroll") for (int rb_ = 0; rb_ < RB; ++rb_)                                         \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                         \
                wr[slot][rb_ * 4 + i_] = load_w(wsrc + rb_ * wrb + (static_cast<int64_t>(tile) * 4 + i_) * 512); \
    }
#define SWL_W_ISSUE_X(set, tile)                                                                      \
    {                                                                                                \
        _Pragma("unroll") for (int q_ = 0; q_ < XL; ++q_) xr[set][q_] = load8(xb + xoff[q_] + (tile) * kWT); \
    }
#define SWL_W_STAGE_X(set, buf)                                                                       \
    {                                                                                                \
        _Pragma("unroll") for (int q_ = 0; q_ < XL; ++q_)                                            \
            *reinterpret_cast<vec8_t<T> *>(xs + (buf) * kXTile + xs_wr0 + q_ * RPP * kWT) = xr[set][q_]; \
    }
// SiLU-gate form up to 192 tokens (BQ = 2): the B fragments of k-step kk+1 are requested BEFORE the MFMAs of k-step kk (two
// register sets, pinned by scheduling barriers). Left to itself the compiler issues each pair of ds_reads right in front of
// the two MFMAs that consume them (`dd w M w M` in tools/isa_mix.py --seq). Measured (profiles/r06d_gemm_wide_pipe_ab.jsonl):
// up/gate + SiLU 51.5 vs 55.2 us at 96 tokens, 54.1 vs 55.9 at 128, 71.0 vs 76.5 at 192; the partial-slab kernels (qkv, o,
// down) are 1-4 us SLOWER with it at 256 tokens and level below, and keep one set.
#define SWL_W_PROCESS(slot, buf)                                                                      \
    {                                                                                                \
        const T *xl_ = xs + (buf) * kXTile + (tp * MTW * 32 + l32) * kWT;                            \
        vec8_t<T> bq_[BQ][MTW];                                                                      \
        if constexpr (BQ == 2) {                                                                     \
            _Pragma("unroll") for (int mt_ = 0; mt_ < MTW; ++mt_)                                    \
                bq_[0][mt_] = *reinterpret_cast<const vec8_t<T> *>(xl_ + mt_ * 32 * kWT + ((hf ^ swz) << 3)); \
        }                                                                                            \
        _Pragma("unroll") for (int kk_ = 0; kk_ < kWT / 16; ++kk_) {                                 \
            const int nk_ = BQ == 2 ? kk_ + 1 : kk_;                                                 \
            if (nk_ < kWT / 16) {                                                                    \
                const int off_ = ((2 * nk_ + hf) ^ swz) << 3;                                        \
                _Pragma("unroll") for (int mt_ = 0; mt_ < MTW; ++mt_)                                \
                    bq_[nk_ % BQ][mt_] = *reinterpret_cast<const vec8_t<T> *>(xl_ + mt_ * 32 * kWT + off_); \
            }                                                                                        \
            if constexpr (BQ == 2) __builtin_amdgcn_sched_barrier(0);                                \
            _Pragma("unroll") for (int mt_ = 0; mt_ < MTW; ++mt_)                                    \
                _Pragma("unroll") for (int rb_ = 0; rb_ < RB; ++rb_)                                 \
                    acc[rb_ * MTW + mt_] = mfma_w(wr[slot][rb_ * 4 + kk_], bq_[kk_ % BQ][mt_], acc[rb_ * MTW + mt_]); \
            if constexpr (BQ == 2) __builtin_amdgcn_sched_barrier(0);                                \
        }                                                                                            \
    }
    // Tile t is multiplied in step t. Its x tile was requested in step t-2 (an L2 round trip under load is longer than
    // the 16-32 MFMAs of one step: r04 measured the one-step distance at 3.4 TB/s on the up/gate shape), sits in register
    // set t&1 and is written to LDS buffer t&1 at the end of step t-1; W tile t was requested kWD-1 steps ahead. Requests
    // return in order: x(t+2) is requested BEFORE W(t+kWD-1), so staging x(t+1) at the end of step t waits for nothing
    // younger than the loads of step t-1.
    constexpr int U = kWD % 2 == 0 ? kWD : 2 * kWD;     // unroll: ring slot and register set are compile-time
#pragma unroll
    for (int d = 0; d < kWD - 1; ++d)
        if (d < nkt) SWL_W_ISSUE_W(d, d);
    SWL_W_ISSUE_X(0, 0);
    SWL_W_STAGE_X(0, 0);
    if (XD == 2 && nkt > 1) SWL_W_ISSUE_X(1 % XD, 1);
    __syncthreads();
    int kt = 0;
    for (; kt + U + (kWD > XD + 1 ? kWD - 1 : XD) <= nkt; kt += U) {
#pragma unroll
        for (int d = 0; d < U; ++d) {
            SWL_W_ISSUE_X((d + XD) % XD, kt + d + XD);
            SWL_W_ISSUE_W((d + kWD - 1) % kWD, kt + d + kWD - 1);
            SWL_W_PROCESS(d % kWD, d & 1);
            SWL_W_STAGE_X((d + 1) % XD, (d + 1) & 1);
            __syncthreads();
        }
    }
    const int rem = nkt - kt;       // (kt is a multiple of U, hence even: set / buffer parity below is t & 1)
#pragma unroll
    for (int t = 0; t < U + (kWD > XD + 1 ? kWD - 1 : XD) - 1; ++t) {
        if (t < rem) {
            if (t + XD < rem) SWL_W_ISSUE_X((t + XD) % XD, kt + t + XD);
            if (t + kWD - 1 < rem) SWL_W_ISSUE_W((t + kWD - 1) % kWD, kt + t + kWD - 1);
            SWL_W_PROCESS(t % kWD, t & 1);
            if (t + 1 < rem) {
                SWL_W_STAGE_X((t + 1) % XD, (t + 1) & 1);
                __syncthreads();
            }
        }
    }
#undef SWL_W_ISSUE_W
#undef SWL_W_ISSUE_X
#undef SWL_W_STAGE_X
#undef SWL_W_PROCESS
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) mfma_results_tie(acc[mt]);
    mfma_results_ready<8>(acc[MT - 1]); // acc comes straight out of the K loop (swl_common.h)

    // acc[a][r], a = rb * MTW + mt: out^T[n = col0 + 32*rb + (r&3) + 8*(r>>2) + 4*hf][m = 32*(tp*MTW + mt) + l32]
    if constexpr (MODE == kWideSiluGate) {
        constexpr int MTR = MT / 2;     // token blocks per exchange round: HW * MTR tiles of 32 x 40 elements fit the x buffers
        static_assert(HW * MTR * 1280 <= 2 * kXTile, "exchange tiles must fit the x buffers");
        typedef T vec4 __attribute__((ext_vector_type(4)));
        T *xch = xs + (wave % HW) * MTR * 1280;
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            __syncthreads();            // the x tiles (round 0) / the previous round's exchange tiles are dead
            if (is_gate) {
#pragma unroll
                for (int j = 0; j < MTR; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float g = to_f(to_t<T>(acc[round * MTR + j][r]));
                        xch[j * 1280 + l32 * 40 + (r & 3) + 8 * (r >> 2) + 4 * hf] = to_t<T>(g / (1.0f + expf(-g)));
                    }
            }
            __syncthreads();
            if (!is_gate && tile_ok) {
#pragma unroll
                for (int j = 0; j < MTR; ++j) {
                    const int mt = round * MTR + j;
                    const int m = 32 * (tp * MTW + mt % MTW) + l32;
                    if (m >= M) continue;
                    T *o = static_cast<T *>(out_) + static_cast<int64_t>(m) * out_stride + col0 + 32 * (mt / MTW) + 4 * hf;
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const vec4 a = *reinterpret_cast<const vec4 *>(xch + j * 1280 + l32 * 40 + 8 * r4 + 4 * hf);
                        vec4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = mul_t<T>(to_t<T>(acc[mt][4 * r4 + e]), a[e]);
                        *reinterpret_cast<vec4 *>(o + 8 * r4) = v;
                    }
                }
            }
        }
        return;
    }
    if (!tile_ok) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = 32 * (tp * MTW + mt % MTW) + l32;
        if (m >= M) continue;
        const int n = col0 + 32 * (mt / MTW) + 4 * hf;
        if constexpr (MODE == kWidePartial) {
            float *slab = static_cast<float *>(out_) + (static_cast<int64_t>(ksplit) * M + m) * N + n;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4_t v = {acc[mt][4 * r4], acc[mt][4 * r4 + 1], acc[mt][4 * r4 + 2], acc[mt][4 * r4 + 3]};
                *reinterpret_cast<float4_t *>(slab + 8 * r4) = v;
            }
        } else {
            typedef T vec4 __attribute__((ext_vector_type(4)));
            T *o = static_cast<T *>(out_) + static_cast<int64_t>(m) * out_stride + n;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                vec4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = to_t<T>(acc[mt][4 * r4 + e]);
                *reinterpret_cast<vec4 *>(o + 8 * r4) = v;
            }
        }
    }
}

struct WidePlan {
    int nwv;    // waves per workgroup: 8 (256 rows of W) or 4 (128 rows)
    int ks;     // K splits (fp32 slabs when > 1)
    int ts;     // token split inside a row group: 1, or 2 (4-wave groups, N % 64 == 0; see the kernel's TS note)
};

// SWL_WIDE_TS=1|2 forces one register tiling of the plain / partial kernels (A/B runs); unset: the measured choice below.
static int wide_ts_forced() {
    static const int v = [] {
        const char *e = getenv("SWL_WIDE_TS");
        return e && (e[0] == '1' || e[0] == '2') ? e[0] - '0' : 0;
    }();
    return v;
}
// Measured (profiles/r06d_gemm_wide_token_split_ab.jsonl, weights cycled through 1.2 GB): with K-chunks of <= 16 K-tiles
// — the qkv and o projections — TS = 2 is 5-9 % faster up to 192 tokens (qkv 23.6 vs 25.4 us at 128, 29.6 vs 32.4 at
// 192; o 23.7 vs 26.1 at 192) and level at 256; the long chunks of down_proj (28 tiles) and the SiLU-gate form are 3-15 %
// SLOWER with it and keep TS = 1.
static int wide_ts(int M, int N, int kc, int nwv, bool silu) {
    if (nwv != 4 || (N & 63) || silu) return 1;
    const int f = wide_ts_forced();
    if (f) return f;
    return (kc <= 16 * kWT && M <= 192) ? 2 : 1;
}

// Workgroup width and K-split for (M, N, K), from the sweeps on MI355X (profiles/r04c_/r04d_gemm_wide_micro.jsonl,
// Llama-3-8B widths, bf16, M = 96..256): 4-wave workgroups everywhere (two per CU up to 128 tokens, one wave per SIMD
// above — they beat the 8-wave groups at every shape once they had the register file to themselves); K split into the
// largest power of two that keeps the launch at <= 256 workgroups and a chunk >= 16 K-tiles (<= 128 tokens) / 8 K-tiles
// long: qkv 4 chunks, o_proj 4 / 8, down 8.
static WidePlan gemm_wide_plan(int M, int N, int K, int forced_nwv, int forced_ks) {
    WidePlan p;
    const int tiles = N / 32;
    p.nwv = forced_nwv ? forced_nwv : 4;
    const int wgs = (tiles + p.nwv - 1) / p.nwv;
    const int min_chunk = (M <= 128 ? 16 : 8) * kWT;
    int ks = 1;
    while (ks < 16 && wgs * ks * 2 <= 256 && K % (kWT * ks * 2) == 0 && K / (ks * 2) >= min_chunk) ks *= 2;
    p.ks = forced_ks ? forced_ks : ks;
    p.ts = wide_ts(M, N, K / p.ks, p.nwv, false);
    return p;
}

template <typename T, int MODE>
static void launch_wide(int mt, int nwv, int ts, dim3 grid, hipStream_t s, void *out, const T *x, const T *wp, int M, int N,
                        int K, int kc, int64_t xs, int64_t os) {
#define SWL_W_LAUNCH(MT_, NWV_, TS_)                                                                                \
    hipLaunchKernelGGL((gemm_packed_wide_kernel<T, MT_, NWV_, MODE, TS_>), grid, dim3(NWV_ * 64), 0, s, out, x, wp, M, N, \
                       K, kc, xs, os)
    if (nwv == 8) {
        if (mt <= 4) SWL_W_LAUNCH(4, 8, 1);
        else if (mt <= 6) SWL_W_LAUNCH(6, 8, 1);
        else SWL_W_LAUNCH(8, 8, 1);
    } else if (ts == 2 && MODE != kWideSiluGate) {
        if constexpr (MODE != kWideSiluGate) {
            if (mt <= 4) SWL_W_LAUNCH(4, 4, 2);
            else if (mt <= 6) SWL_W_LAUNCH(6, 4, 2);
            else SWL_W_LAUNCH(8, 4, 2);
        }
    } else {
        if (mt <= 4) SWL_W_LAUNCH(4, 4, 1);
        else if (mt <= 6) SWL_W_LAUNCH(6, 4, 1);
        else SWL_W_LAUNCH(8, 4, 1);
    }
#undef SWL_W_LAUNCH
}

} // namespace swl

/* Workspace bytes swl_gemm_packed_wide needs for (M, N, K) with the library's own plan (0 = K is not split). */
extern "C" size_t swl_gemm_packed_wide_workspace_bytes(int32_t M, int32_t N, int32_t K) {
    if (M <= 0 || M > 256 || N <= 0 || K <= 0 || (N & 31) || (K & (swl::kWT - 1))) return 0;
    const swl::WidePlan p = swl::gemm_wide_plan(M, N, K, 0, 0);
    return p.ks > 1 ? static_cast<size_t>(p.ks) * M * N * sizeof(float) : 0;
}

/* out[M, N] = x . W^T for up to 256 tokens on a weight packed by swl_gemm_pack_weight. N % 32 == 0, K % 64 == 0.
 * waves_per_group: 0 = library's choice, 4 or 8; k_splits: 0 = library's choice, else a power of two <= 16 with
 * K % (64 * k_splits) == 0. workspace >= k_splits * M * N * 4 bytes when K is split
 * (swl_gemm_packed_wide_workspace_bytes for the library's plan). */
extern "C" int swl_gemm_packed_wide(void *out, const void *x, const void *w_packed, void *workspace,
                                    size_t workspace_bytes, int32_t M, int32_t N, int32_t K, int64_t x_row_stride,
                                    int64_t out_row_stride, int32_t waves_per_group, int32_t k_splits, int32_t dtype,
                                    swl_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!out || !x || !w_packed) return SWL_ERR_BAD_ARG;
    if (M > 256 || (N & 31) || (K & (swl::kWT - 1))) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || out_row_stride < N || (x_row_stride & 7) || (out_row_stride & 3)) return SWL_ERR_BAD_ARG;
    if (static_cast<int64_t>(M) * x_row_stride >= (1ll << 31)) return SWL_ERR_UNSUPPORTED;
    if (!swl::aligned16(x) || !swl::aligned16(w_packed) || (reinterpret_cast<uintptr_t>(out) & 7u) ||
        (workspace && !swl::aligned16(workspace)))
        return SWL_ERR_BAD_ARG;
    if (!(waves_per_group == 0 || waves_per_group == 4 || waves_per_group == 8)) return SWL_ERR_BAD_ARG;
    if (k_splits < 0 || k_splits > 16 || (k_splits & (k_splits - 1))) return SWL_ERR_BAD_ARG;
    const swl::WidePlan p = swl::gemm_wide_plan(M, N, K, waves_per_group, k_splits);
    if (K % (swl::kWT * p.ks) != 0) return SWL_ERR_UNSUPPORTED;
    const int mt = (M + 31) / 32;
    const int kc = K / p.ks;
    const dim3 grid((N / 32 + p.nwv - 1) / p.nwv, p.ks);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (p.ks == 1) {
        SWL_DISPATCH_DTYPE(dtype, T, {
            swl::launch_wide<T, swl::kWideDirect>(mt, p.nwv, p.ts, grid, s, out, static_cast<const T *>(x),
                                                  static_cast<const T *>(w_packed), M, N, K, kc, x_row_stride,
                                                  out_row_stride);
        });
        return swl::check_launch();
    }
    if (!workspace || workspace_bytes < static_cast<size_t>(p.ks) * M * N * sizeof(float)) return SWL_ERR_BAD_ARG;
    SWL_DISPATCH_DTYPE(dtype, T, {
        swl::launch_wide<T, swl::kWidePartial>(mt, p.nwv, p.ts, grid, s, workspace, static_cast<const T *>(x),
                                               static_cast<const T *>(w_packed), M, N, K, kc, x_row_stride, N);
    });
    if (swl::check_launch() != SWL_OK) return SWL_ERR_LAUNCH;
    return swl_splitk_reduce(out, static_cast<const float *>(workspace), p.ks, M, N, out_row_stride, dtype, stream);
}

/* The split count swl_gemm_packed_wide picks for k_splits = 0 (0 = shape unsupported). */
extern "C" int swl_gemm_packed_wide_choose_splits(int32_t M, int32_t N, int32_t K) {
    if (M <= 0 || M > 256 || N <= 0 || K <= 0 || (N & 31) || (K & (swl::kWT - 1))) return 0;
    return swl::gemm_wide_plan(M, N, K, 0, 0).ks;
}

/* Partial slabs only (slabs[k_splits][M][N] fp32, k_splits >= 1: a power of two <= 16 with K % (64 * k_splits) == 0) for
 * the split-K consumers (swl_splitk_fused_add_rmsnorm, swl_splitk_rotary_store_kv_decode): swl_gemm_packed_wide minus its
 * reduce launch, same bits. */
extern "C" int swl_gemm_packed_wide_partial(float *slabs, size_t slabs_bytes, const void *x, const void *w_packed,
                                            int32_t M, int32_t N, int32_t K, int64_t x_row_stride,
                                            int32_t waves_per_group, int32_t k_splits, int32_t dtype,
                                            swl_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!slabs || !x || !w_packed || k_splits < 1 || k_splits > 16 || (k_splits & (k_splits - 1))) return SWL_ERR_BAD_ARG;
    if (M > 256 || (N & 31) || (K & (swl::kWT - 1)) || K % (swl::kWT * k_splits)) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || (x_row_stride & 7) || !swl::aligned16(x) || !swl::aligned16(w_packed) || !swl::aligned16(slabs))
        return SWL_ERR_BAD_ARG;
    if (static_cast<int64_t>(M) * x_row_stride >= (1ll << 31)) return SWL_ERR_UNSUPPORTED;
    if (!(waves_per_group == 0 || waves_per_group == 4 || waves_per_group == 8)) return SWL_ERR_BAD_ARG;
    if (slabs_bytes < static_cast<size_t>(k_splits) * M * N * sizeof(float)) return SWL_ERR_BAD_ARG;
    const swl::WidePlan p = swl::gemm_wide_plan(M, N, K, waves_per_group, k_splits);
    const dim3 grid((N / 32 + p.nwv - 1) / p.nwv, p.ks);
    SWL_DISPATCH_DTYPE(dtype, T, {
        swl::launch_wide<T, swl::kWidePartial>((M + 31) / 32, p.nwv, p.ts, grid, static_cast<hipStream_t>(stream), slabs,
                                               static_cast<const T *>(x), static_cast<const T *>(w_packed), M, N, K,
                                               K / p.ks, x_row_stride, N);
    });
    return swl::check_launch();
}

/* out[M, I] = up * silu(gate) of x . [up ; gate]^T for up to 256 tokens on a packed weight (the large-batch twin of
 * swl_gemm_skinny_packed_silu_gate; same rounding points as linear + silu_and_mul). I % 32 == 0, K % 64 == 0.
 * waves_per_group: 0 = library's choice (4), 4 or 8. */
extern "C" int swl_gemm_packed_wide_silu_gate(void *out, const void *x, const void *w_up_gate_packed, int32_t M,
                                              int32_t I, int32_t K, int64_t x_row_stride, int64_t out_row_stride,
                                              int32_t waves_per_group, int32_t dtype, swl_stream_t stream) {
    if (M < 0 || I <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!out || !x || !w_up_gate_packed) return SWL_ERR_BAD_ARG;
    if (M > 256 || (I & 31) || (K & (swl::kWT - 1))) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || out_row_stride < I || (x_row_stride & 7) || (out_row_stride & 3)) return SWL_ERR_BAD_ARG;
    if (static_cast<int64_t>(M) * x_row_stride >= (1ll << 31)) return SWL_ERR_UNSUPPORTED;
    if (!swl::aligned16(x) || !swl::aligned16(w_up_gate_packed) || (reinterpret_cast<uintptr_t>(out) & 7u))
        return SWL_ERR_BAD_ARG;
    if (!(waves_per_group == 0 || waves_per_group == 4 || waves_per_group == 8)) return SWL_ERR_BAD_ARG;
    const int nwv = waves_per_group ? waves_per_group : 4;
    const int hw = nwv / 2;
    const int mt = (M + 31) / 32;
    const dim3 grid((I / 32 + hw - 1) / hw, 1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    SWL_DISPATCH_DTYPE(dtype, T, {
        swl::launch_wide<T, swl::kWideSiluGate>(mt, nwv, 1, grid, s, out, static_cast<const T *>(x),
                                                static_cast<const T *>(w_up_gate_packed), M, I, K, K, x_row_stride,
                                                out_row_stride);
    });
    return swl::check_launch();
}
