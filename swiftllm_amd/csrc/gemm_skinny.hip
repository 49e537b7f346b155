// gemm_skinny.hip — weight-streaming GEMMs for decode batches on gfx950.
//
// Map of this file (all variants share the tile geometry and the MFMA order, hence their bits):
//   gemm_skinny_kernel       row-major W, wave-private W and x tiles, no barriers   (short K-chunks, M <= 32)
//   gemm_skinny_ring_kernel  x tile shared by the workgroup, register ring of W tiles (M <= 32)
//        PACKED = false      ... row-major W through a wave-private LDS transpose
//        PACKED = true       ... W pre-packed in MFMA-fragment order (swl_gemm_pack_weight): global -> VGPR -> MFMA;
//                                the default decode path (EngineConfig.pack_decode_weights)
//   gemm_packed_mt_kernel    packed W, 2 blocks of 32 tokens per weight fragment (32 < M <= 64)
//   splitk_reduce_kernel, pack_weight_kernel
//
// out[M, N] = x[M, K] . W[N, K]^T  — the shape of every projection of a decode step
// (reference: swiftllm/worker/kernels/linear.py:3-12 called from transformer_layer.py:54-56,117,
// 126,128 and post_layer.py:38). At M <= 32 the op is pure HBM streaming of W (77 % of all bytes a
// decode step moves); algorithmic bytes = N*K*e (+ M*K*e + M*N*e).
//
// Structure (a streaming kernel, not a tiled compute GEMM):
//   * one wave owns 32 consecutive rows of W (= 32 output columns) over one K-chunk and issues
//     v_mfma_f32_32x32x16 with A = W rows, B = x^T: the accumulator is out^T[n][m], so all M <= 32
//     tokens ride along with every weight byte exactly once;
//   * W is fetched in FULL lines: one load instruction = 4 rows x 256 contiguous bytes (16 B per lane),
//     never "fragment-shaped" (an MFMA fragment is 32 rows a row-pitch apart: 64 separate 16-byte
//     requests per instruction — measured 3 TB/s and a thrashing L1). The tile [32 rows][128 k] is
//     transposed into the fragment layout through a wave-private, XOR-swizzled LDS tile
//     (slot = chunk ^ (row & 15): conflict-free for both the 8-lane write groups and the 16-lane
//     ds_read_b128 groups); the same for the x tile (from L2). No barriers: a wave only ever reads
//     LDS it wrote itself;
//   * the NEXT K-tile (8 KiB of W + 8 KiB of x per wave) is in flight in registers while the current
//     one is transposed and multiplied (branch-free steady state => counted vmcnt waits);
//   * long K-chunks (>= 8 tiles) run the RING variant below: the x tile is shared by the four waves of
//     the workgroup (each stages a quarter; one barrier per K-tile) and W tiles ride a 3-deep register
//     ring. Measured on MI355X (tools/probe/, bf16, M = 32): per-wave x tiles cost 10-15 % of the
//     stream rate (4.7 -> 5.4 TB/s with the x traffic removed); the ring variant recovers most of it
//     (up_gate 46.2 -> 43.0 us, lm_head 199 -> 189 us, down 22.5 -> 21.1 us). Short chunks (o_proj:
//     4 tiles per wave) are better off without barriers and keep the private-tile kernel;
//   * N/32 tiles alone do not fill 256 CUs for N = 4096..6144, so K is split across workgroups
//     (grid.y): each split writes an fp32 partial slab, a second tiny kernel adds the slabs in a fixed
//     order and rounds once. Deterministic, no atomics.
#include "swl_common.h"

namespace swl {

__device__ __forceinline__ float16_t mfma32x32x16(vec8_t<f16> a, vec8_t<f16> b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float16_t mfma32x32x16(vec8_t<bf16> a, vec8_t<bf16> b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

constexpr int kKT = 128;    // k elements per tile (256 bytes per row: two full lines)
constexpr int kGemmWaves = 4;

enum GemmMode {
    kGemmDirect = 0,   // out[M, N] in T
    kGemmPartial = 1,  // fp32 slab [ksplit][M][N]
    kGemmSiluGate = 2, // W = [up ; gate] (2*I rows): out[M, I] = up * silu(gate), the FFN's SiLU-gate fused in
};

// Optional per-launch extras of the packed ring kernel.
struct GemmExtra {
    // uneven K-splits (partial slabs): > 0 = the number of 128-column K-tiles of the whole projection; split y of
    // gridDim.y takes tiles [y * total / gridDim.y, (y + 1) * total / gridDim.y) — for K that has no power-of-two split
    // of whole tiles (Llama-2-7B's down projection: K = 11008 = 86 tiles)
    int k_tiles_total;
    // deferred RMSNorm (SiLU-gate mode): ssq_in[ssq_parts][M] partial sums of squares of the token rows behind x
    const float *ssq_in;
    float eps;
    int ssq_parts;
    // NF (norm on the fly): x is the RAW residual stream; the kernel stages round(x * norm_w) (splitk_add_scale's bits)
    // and adds up the rows' sums of squares itself — SiLU-gate mode applies the 1/rms in its epilogue, partial mode writes
    // ssq_out[ksplit][M] (the row_ssq the slab-fed attention prologue takes)
    const void *norm_w;
    float *ssq_out;
};

// acc[r] = out^T[n = n0 + (r&3) + 8*(r>>2) + 4*hf][m = l32] -> the three output modes.
// `wtiles + w * wave_pitch` is wave w's private W tile (reused as exchange space in SiLU-gate mode).
// `rs`: deferred-RMSNorm row scale of token m = lane % 32 (rmsnorm.hip: splitk_add_scale_kernel), applied in fp32
// before the projection's one rounding; 1.0f (exact no-op) everywhere else.
template <typename T, int MODE>
__device__ __forceinline__ void gemm_epilogue(const float16_t &acc, void *__restrict__ out_, T *wtiles,
                                              int wave_pitch, int wave, int lane, bool is_gate, bool tile_ok,
                                              int col0, int n0, int ksplit, int M, int N, int64_t out_stride,
                                              float rs = 1.0f, int gate_waves_off = 2) {
    const int l32 = lane & 31;
    const int hf = lane >> 5;
    if constexpr (MODE == kGemmSiluGate) {
        // Same rounding points as linear -> silu_and_mul_inplace (reference silu_and_mul.py:16-23): the
        // projection is rounded to T, silu is evaluated in fp32 and rounded to T, the product is in T.
        T *mine = wtiles + wave * wave_pitch;
        if (is_gate) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float g = to_f(to_t<T>(acc[r] * rs));
                mine[l32 * 40 + (r & 3) + 8 * (r >> 2) + 4 * hf] = to_t<T>(g / (1.0f + expf(-g)));
            }
        }
        __syncthreads();
        if (!is_gate && tile_ok && l32 < M) {
            const T *act = wtiles + (wave + gate_waves_off) * wave_pitch;   // the gate tile of this wave's columns
            typedef T vec4 __attribute__((ext_vector_type(4)));
            T *o = static_cast<T *>(out_) + static_cast<int64_t>(l32) * out_stride + col0 + 4 * hf;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const vec4 a = *reinterpret_cast<const vec4 *>(act + l32 * 40 + 8 * r4 + 4 * hf);
                vec4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = mul_t<T>(to_t<T>(acc[4 * r4 + e] * rs), a[e]);
                *reinterpret_cast<vec4 *>(o + 8 * r4) = v;
            }
        }
        return;
    }
    if (tile_ok && l32 < M) {
        if constexpr (MODE == kGemmPartial) {
            float *slab = static_cast<float *>(out_) +
                          (static_cast<int64_t>(ksplit) * M + l32) * N + n0 + 4 * hf;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4_t v = {acc[4 * r4], acc[4 * r4 + 1], acc[4 * r4 + 2], acc[4 * r4 + 3]};
                *reinterpret_cast<float4_t *>(slab + 8 * r4) = v;
            }
        } else {
            typedef T vec4 __attribute__((ext_vector_type(4)));
            T *o = static_cast<T *>(out_) + static_cast<int64_t>(l32) * out_stride + n0 + 4 * hf;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                vec4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = to_t<T>(acc[4 * r4 + e]);
                *reinterpret_cast<vec4 *>(o + 8 * r4) = v;
            }
        }
    }
}

template <typename T, int MODE>
// 2 waves per SIMD (<= 256 registers): two 4-wave workgroups per CU = 64 KiB of W in flight per CU
__global__ __launch_bounds__(kGemmWaves * 64, 2) void gemm_skinny_kernel(
    void *__restrict__ out_, const T *__restrict__ x, const T *__restrict__ w, int M, int N, int K,
    int kc, int64_t x_stride, int64_t out_stride) {
    // [wave][0 = W tile, 1 = x tile][32 rows x 128 elements, 16-byte slots XOR-swizzled per row]
    __shared__ __attribute__((aligned(16))) T lds[kGemmWaves][2][32 * kKT];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // SiLU-gate mode: N = I (output columns); waves 0,1 own two `up` tiles, waves 2,3 the `gate`
    // tiles of the same columns (rows I + ... of W), and hand their activated tile over through LDS.
    const bool is_gate = MODE == kGemmSiluGate && wave >= 2;
    const int col0 = MODE == kGemmSiluGate ? (blockIdx.x * 2 + (wave & 1)) * 32
                                           : (blockIdx.x * kGemmWaves + wave) * 32;
    const bool tile_ok = col0 < N;
    if (MODE != kGemmSiluGate && !tile_ok) return; // no barriers in these modes: a whole wave may leave
    const int n0 = tile_ok ? col0 + (is_gate ? N : 0) : 0; // row of W this wave starts at
    const int ksplit = blockIdx.y;
    const int k_begin = ksplit * kc;
    const int nkt = kc / kKT;

    // staging map: load instruction i covers rows 4i..4i+3, lane -> (row 4i + lane/16, chunk lane%16)
    const int rsub = lane >> 4;
    const int chunk = lane & 15;
    const T *wsrc = w + static_cast<int64_t>(n0 + rsub) * K + k_begin + chunk * 8;
    const T *xsrc = x + k_begin + chunk * 8;
    int xrow_off[8];     // x is M <= 32 rows: 32-bit offsets are ample
    int lds_wr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + rsub;
        xrow_off[i] = min(row, M - 1) * static_cast<int>(x_stride); // rows >= M: clamped, never stored
        lds_wr[i] = row * kKT + ((chunk ^ (row & 15)) << 3);
    }
    // fragment map: lane -> (row/column lane%32, k-half lane/32)
    const int l32 = lane & 31;
    const int hf = lane >> 5;
    T *wl = &lds[wave][0][0];
    T *xl = &lds[wave][1][0];

    vec8_t<T> wa[8], xa[8], wb[8], xb[8];
    auto issue = [&](vec8_t<T>(&wr)[8], vec8_t<T>(&xr)[8], int kt) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            wr[i] = load8_nt(wsrc + static_cast<int64_t>(4 * i) * K + kt * kKT); // full lines, read once
            xr[i] = load8(xsrc + xrow_off[i] + kt * kKT);
        }
    };
    float16_t acc = float16_t{};
    auto process = [&](const vec8_t<T>(&wr)[8], const vec8_t<T>(&xr)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            *reinterpret_cast<vec8_t<T> *>(wl + lds_wr[i]) = wr[i];
            *reinterpret_cast<vec8_t<T> *>(xl + lds_wr[i]) = xr[i];
        }
#pragma unroll
        for (int kk = 0; kk < kKT / 16; ++kk) {
            const int off = l32 * kKT + (((2 * kk + hf) ^ (l32 & 15)) << 3);
            const vec8_t<T> a = *reinterpret_cast<const vec8_t<T> *>(wl + off);
            const vec8_t<T> b = *reinterpret_cast<const vec8_t<T> *>(xl + off);
            acc = mfma32x32x16(a, b, acc);
        }
    };

    // register double buffer; the steady-state body is branch-free (exact counted waits)
    issue(wa, xa, 0);
    int kt = 0;
    for (; kt + 2 < nkt; kt += 2) {
        issue(wb, xb, kt + 1);
        process(wa, xa);
        issue(wa, xa, kt + 2);
        process(wb, xb);
    }
    if (nkt - kt == 2) {
        issue(wb, xb, kt + 1);
        process(wa, xa);
        process(wb, xb);
    } else {
        process(wa, xa);
    }

    mfma_results_ready<8>(acc); // acc comes straight out of the K loop (swl_common.h)
    gemm_epilogue<T, MODE>(acc, out_, &lds[0][0][0], 2 * 32 * kKT, wave, lane, is_gate, tile_ok, col0, n0, ksplit, M, N,
                           out_stride);
}

// ---- RING variant: x tile shared by the workgroup, kRing-deep register ring of W tiles --------------
constexpr int kRing = 3;

// PACKED: W comes pre-packed in MFMA-fragment order (swl_gemm_pack_weight: [N/32][K/16][64 lanes][8]) — the A
// operand of one v_mfma_f32_32x32x16 is ONE contiguous KiB and a wave's whole K range for its 32 rows one
// contiguous run of K*64 bytes. Loads go global -> VGPR -> MFMA: no LDS round trip for W, and the DRAM sees long
// sequential bursts instead of 32 row segments of 256 B per tile (stream probe: 5.7-5.9 TB/s with 256-B row
// segments, 6.2-6.4 TB/s with KiB runs). Same MFMA order as the row-major kernels, hence the same bits. RD = ring
// depth (3 for K-chunks of >= 8 tiles, 2 for shorter ones).
// NWV = waves per workgroup: 4, or 3 (packed W, partial slabs only) for projections whose tile count is a multiple of
// three — the fused qkv projection of Llama-3-8B is 192 tiles x 4 K-splits = 768 wave-chunks: 192 four-wave workgroups
// leave a quarter of the 256 CUs idle, 256 three-wave ones fill the chip (r02). Three waves stage the 8 row-groups of the
// x tile as 3 + 3 + 2: the ninth (dummy) group is a clamped load into four spare LDS rows — no branch in the pipeline.
// NF (packed W, SiLU-gate or partial mode): norm on the fly — x is the raw residual stream r (what gemm_rows.hip leaves
// behind: no consumer launch computed round(r * norm_w) or the sums of squares). The norm-weight chunk of a K-tile rides
// with the tile's x loads, the staging pass multiplies and rounds (the bits of splitk_add_scale_kernel) and accumulates
// sum r^2 per row in fp32 — every workgroup redundantly, it sees all of x anyway.
// NX (with NF; r06c): the EXACT norm on the fly — x is the raw residual stream r as in NF, but the rows' sums of squares are
// already known (fuse.ssq_in[M][ssq_parts]: the per-tile partials swl_gemm_rows_add_ssq left behind), so the workgroup
// first adds them up in a fixed order, and the staging pass writes round(r * rstd * w) — the reference's one rounding of the
// normalised activation (rmsnorm.py:57-64; rmsnorm.hip's expression) — in float16 as in bfloat16. Nothing is deferred: the
// epilogues see rs = 1.
template <typename T, int MODE, bool PACKED = false, int RD = kRing, int NWV = kGemmWaves, bool NF = false, bool NX = false>
// 2 waves per SIMD: the ring holds RD x 8 KiB of W per wave in registers (~220 VGPRs at RD = 3); the two-wave SiLU-gate
// form stages twice the x rows per wave and runs one wave per SIMD (two workgroups per CU)
__global__ __launch_bounds__(NWV * 64, NWV == 2 ? 1 : 2) void gemm_skinny_ring_kernel(
    void *__restrict__ out_, const T *__restrict__ x, const T *__restrict__ w, int M, int N, int K,
    int kc, int64_t x_stride, int64_t out_stride, GemmExtra fuse) {
    static_assert(NWV == kGemmWaves || (NWV == 3 && PACKED && MODE == kGemmPartial) ||
                      (NWV == 2 && PACKED && MODE == kGemmSiluGate),
                  "3 waves: packed partial only; 2 waves: packed SiLU-gate only");
    constexpr int HWV = NWV / 2;         // SiLU-gate: waves [0, HWV) own `up` tiles, [HWV, NWV) the matching `gate` tiles
    static_assert(!NF || (PACKED && MODE != kGemmDirect), "norm on the fly: packed SiLU-gate / partial only");
    static_assert(!NX || NF, "the exact norm on the fly is a mode of NF");
    constexpr int D = RD;
    constexpr int XL = (8 + NWV - 1) / NWV; // x row-groups (4 rows each) a wave stages per tile
    constexpr int XROWS = 4 * XL * NWV;      // 32, or 36 with the dummy group of the 3-wave variant
    // [0..1] the double-buffered x tile of the workgroup, [2 + wave] the wave-private W tile (row-major W only)
    __shared__ __attribute__((aligned(16))) T lds[2 + (PACKED ? 0 : kGemmWaves)][XROWS * kKT];
    __shared__ float ssq_row[NF ? XROWS : 1];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool is_gate = MODE == kGemmSiluGate && wave >= HWV;
    const int col0 = MODE == kGemmSiluGate ? (blockIdx.x * HWV + (wave % HWV)) * 32
                                           : (blockIdx.x * NWV + wave) * 32;
    const bool tile_ok = col0 < N; // barriers below: a wave without a tile still stages x and syncs
    const int n0 = tile_ok ? col0 + (is_gate ? N : 0) : 0;
    const int ksplit = blockIdx.y;
    int k_begin = ksplit * kc;
    int nkt = kc / kKT;
    if (PACKED && MODE == kGemmPartial && fuse.k_tiles_total > 0) {
        const int t0 = ksplit * fuse.k_tiles_total / static_cast<int>(gridDim.y);
        nkt = (ksplit + 1) * fuse.k_tiles_total / static_cast<int>(gridDim.y) - t0;
        k_begin = t0 * kKT;
    }

    const int rsub = lane >> 4;
    const int chunk = lane & 15;
    // row-major: lane -> (row 4i + lane/16, 16-byte chunk lane%16) of tile i; packed: lane -> its fragment slot of
    // block (n0/32, k/16), blocks of 512 elements back to back along k
    const T *wsrc = PACKED ? w + (static_cast<int64_t>(n0 / 32) * (K / 16) + k_begin / 16) * 512 + lane * 8
                           : w + static_cast<int64_t>(n0 + rsub) * K + k_begin + chunk * 8;
    int lds_wr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + rsub;
        lds_wr[i] = row * kKT + ((chunk ^ (row & 15)) << 3);
    }
    const T *xsrc[XL];
    int xs_wr[XL];
#pragma unroll
    for (int q = 0; q < XL; ++q) {
        const int row = 4 * (wave * XL + q) + rsub;
        xsrc[q] = x + static_cast<int64_t>(min(row, M - 1)) * x_stride + k_begin + chunk * 8;
        xs_wr[q] = row * kKT + ((chunk ^ (row & 15)) << 3);
    }
    const int l32 = lane & 31;
    const int hf = lane >> 5;
    T *wl = &lds[PACKED ? 0 : 2 + wave][0]; // (unused when PACKED)

    // deferred RMSNorm (SiLU-gate mode, packed W): the per-1024-column sums of squares of this lane's token row, requested
    // BEFORE the weight stream (oldest loads: no wait of the pipeline ever includes them) and summed after the K loop
    float ssv[8];
    const bool row_scaled = PACKED && MODE == kGemmSiluGate && (NF || fuse.ssq_in != nullptr);
    if (row_scaled && !NF) {
        const int m = min(lane & 31, M - 1);
#pragma unroll
        for (int p2 = 0; p2 < 8; ++p2) ssv[p2] = p2 < fuse.ssq_parts ? fuse.ssq_in[p2 * M + m] : 0.f;
    }

    // NX: the partial sums of squares are the OLDEST requests of the workgroup (they come back before any tile)
    float nx_part[NX ? 2 : 1][8];
    if constexpr (NX) {
        // 32 tokens x 8 pieces of ssq_parts / 8 consecutive partials each (ssq_parts % 8 == 0): item = token * 8 + piece
        const int per = fuse.ssq_parts >> 3;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = static_cast<int>(threadIdx.x) + it * NWV * 64;
            const int tok = min(item >> 3, M - 1), piece = item & 7;
            const float *src = fuse.ssq_in + static_cast<int64_t>(tok) * fuse.ssq_parts + piece * per;
#pragma unroll
            for (int e = 0; e < 8; ++e) nx_part[it][e] = 0.f;
            if (item < 256) {
                if (per == 32) {
                    // hidden = 4096: the piece is 128 bytes — all eight requests before the first add (a load inside the
                    // accumulation loop below is waited for on the spot: four round trips in a row instead of one)
                    float4_t u[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) u[j] = *reinterpret_cast<const float4_t *>(src + 4 * j);
#pragma unroll
                    for (int j = 0; j < 8; j += 2)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            nx_part[it][e] += u[j][e];
                            nx_part[it][4 + e] += u[j + 1][e];
                        }
                } else {
                    for (int c = 0; c < per; c += 8) {      // (per % 8 == 0 is checked by the host: 32-byte runs)
                        const float4_t u0 = *reinterpret_cast<const float4_t *>(src + c);
                        const float4_t u1 = *reinterpret_cast<const float4_t *>(src + c + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            nx_part[it][e] += u0[e];
                            nx_part[it][4 + e] += u1[e];
                        }
                    }
                }
            }
        }
    }
    vec8_t<T> wr[D][8], xr[D][XL];
    vec8_t<T> wn[NF ? D : 1];           // NF: the norm-weight chunk of the tile (same 8 columns for every row of the lane)
    float ssq_acc[NF ? XL : 1];
#pragma unroll
    for (int q = 0; q < (NF ? XL : 1); ++q) ssq_acc[q] = 0.f;
    const T *nwsrc = NF ? static_cast<const T *>(fuse.norm_w) + k_begin + chunk * 8 : nullptr;
    float16_t acc = float16_t{};
    // x loads go first: they are the ones the stage-ahead below waits for (loads return in order)
#define SWL_ISSUE(slot, tile)                                                                        \
    {                                                                                                \
        if constexpr (NF) wn[slot] = load8(nwsrc + (tile) * kKT);                                    \
        _Pragma("unroll") for (int q_ = 0; q_ < XL; ++q_) xr[slot][q_] = load8(xsrc[q_] + (tile) * kKT); \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                             \
            wr[slot][i_] = PACKED ? load8_nt(wsrc + (static_cast<int64_t>(tile) * 8 + i_) * 512)     \
                                  : load8_nt(wsrc + static_cast<int64_t>(4 * i_) * K + (tile) * kKT); \
    }
#define SWL_STAGE_X(slot, buf, tile)                                                                 \
    {                                                                                                \
        _Pragma("unroll") for (int q_ = 0; q_ < XL; ++q_) {                                          \
            vec8_t<T> xv_ = xr[slot][q_];                                                            \
            if constexpr (NX) {                                                                      \
                _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_)                                     \
                    xv_[j_] = to_t<T>(to_f(xv_[j_]) * rstd_q[q_] * to_f(wn[slot][j_]));              \
            } else if constexpr (NF) {                                                               \
                _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) {                                   \
                    const float v_ = to_f(xv_[j_]);                                                  \
                    ssq_acc[q_] = fmaf(v_, v_, ssq_acc[q_]);                                         \
                    xv_[j_] = to_t<T>(v_ * to_f(wn[slot][j_]));                                      \
                }                                                                                    \
            }                                                                                        \
            *reinterpret_cast<vec8_t<T> *>(&lds[buf][xs_wr[q_]]) = xv_;                              \
        }                                                                                            \
    }
#define SWL_PROCESS(slot, buf)                                                                       \
    {                                                                                                \
        if constexpr (!PACKED) {                                                                     \
            _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                         \
                *reinterpret_cast<vec8_t<T> *>(wl + lds_wr[i_]) = wr[slot][i_];                      \
        }                                                                                            \
        const T *xl_ = &lds[buf][0];                                                                 \
        _Pragma("unroll") for (int kk_ = 0; kk_ < kKT / 16; ++kk_) {                                 \
            const int off_ = l32 * kKT + (((2 * kk_ + hf) ^ (l32 & 15)) << 3);                       \
            vec8_t<T> a_;                                                                            \
            if constexpr (PACKED) a_ = wr[slot][kk_];                                                \
            else a_ = *reinterpret_cast<const vec8_t<T> *>(wl + off_);                               \
            const vec8_t<T> b_ = *reinterpret_cast<const vec8_t<T> *>(xl_ + off_);                   \
            acc = mfma32x32x16(a_, b_, acc);                                                         \
        }                                                                                            \
    }
    // prologue: up to D-1 tiles in flight; x of tile 0 published. (Tried in r02: unconditional requests in program
    // order, so that the first wait of every iteration is sized exactly instead of conservatively — vmcnt(17) instead
    // of vmcnt(7) in the ISA. No measurable difference on MI355X: 39.7 us either way for the up/gate projection.)
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
        if (d < nkt) SWL_ISSUE(d, d);
    float rstd_q[NX ? XL : 1];
    if constexpr (NX) {
        // pieces -> LDS (the second x buffer is idle until tile 1 is staged), 8 pieces per token in order, 1/rms per row
        float *scr = reinterpret_cast<float *>(&lds[1][0]);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = static_cast<int>(threadIdx.x) + it * NWV * 64;
            if (item < 256)
                scr[item] = ((nx_part[it][0] + nx_part[it][1]) + (nx_part[it][2] + nx_part[it][3])) +
                            ((nx_part[it][4] + nx_part[it][5]) + (nx_part[it][6] + nx_part[it][7]));
        }
        __syncthreads();
        if (threadIdx.x < 32) {
            const float *pp = scr + threadIdx.x * 8;
            const float ss = ((pp[0] + pp[1]) + (pp[2] + pp[3])) + ((pp[4] + pp[5]) + (pp[6] + pp[7]));
            ssq_row[threadIdx.x] = 1.0f / sqrtf(ss / static_cast<float>(K) + fuse.eps);     // rmsnorm.hip's formula
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < XL; ++q) rstd_q[q] = ssq_row[min(4 * (wave * XL + q) + rsub, 31)];
        __syncthreads();        // (scr is the x buffer of tile 1: nobody may still be reading it when tile 1 is staged)
    }
    SWL_STAGE_X(0, 0, 0);
    __syncthreads();
    int kt = 0;
    // steady state, branch-free (counted vmcnt waits): step d multiplies tile kt+d out of slot d, refills
    // slot (d+D-1)%D with tile kt+d+D-1 and publishes the x tile of kt+d+1 (in the other x buffer: its
    // last readers finished before the previous barrier)
    for (; kt + 2 * D - 1 <= nkt; kt += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            SWL_ISSUE((d + D - 1) % D, kt + d + D - 1);
            SWL_PROCESS(d, (kt + d) & 1);
            // (NF: staging the next tile BEFORE the MFMAs — its ~50 VALU instructions per row group then run while the
            // wave would wait for its weight tile — measured slower in the layer chain, 80.5 vs 78.0 us at batch 1:
            // profiles/r05c_rows_layer_micro_*.jsonl)
            SWL_STAGE_X((d + 1) % D, (kt + d + 1) & 1, kt + d + 1);
            __syncthreads();
        }
    }
    // drain: up to 2D-2 tiles left, the first min(D-1, rem) of them in flight in slots 0..D-2
    const int rem = nkt - kt;
#pragma unroll
    for (int t = 0; t < 2 * D - 2; ++t) {
        if (t < rem) {
            if (t + D - 1 < rem) SWL_ISSUE((t + D - 1) % D, kt + t + D - 1);
            SWL_PROCESS(t % D, (kt + t) & 1);
            if (t + 1 < rem) {
                SWL_STAGE_X((t + 1) % D, (kt + t + 1) & 1, kt + t + 1);
                __syncthreads();
            }
        }
    }
#undef SWL_ISSUE
#undef SWL_STAGE_X
#undef SWL_PROCESS
    mfma_results_ready<8>(acc); // acc comes straight out of the K loop (swl_common.h)
    if constexpr (PACKED) {
        if constexpr (NF && !NX) { // the 16 chunk lanes of a row hold its sum of squares in pieces: DPP row reduction -> LDS
#pragma unroll
            for (int q = 0; q < XL; ++q) {
                const float t = group_allreduce_sum<16>(ssq_acc[q]);
                if (chunk == 0) ssq_row[4 * (wave * XL + q) + rsub] = t;
            }
        }
        // no W tiles in LDS: the SiLU-gate exchange (32 x 40 elements per wave) reuses the x buffers once every
        // wave is done reading them
        if constexpr (MODE == kGemmSiluGate || NF) __syncthreads();
        if constexpr (NF && !NX && MODE == kGemmPartial) {
            if (blockIdx.x == 0 && wave == 0 && lane < M) fuse.ssq_out[ksplit * M + lane] = ssq_row[lane];
        }
        float rs = 1.0f;
        if constexpr (NX) {
            // (already normalised: nothing pending)
        } else if constexpr (NF && MODE == kGemmSiluGate) {
            rs = 1.0f / sqrtf(ssq_row[min(lane & 31, M - 1)] / static_cast<float>(K) + fuse.eps);
        } else if (row_scaled) {
            const float ss = ((ssv[0] + ssv[1]) + (ssv[2] + ssv[3])) + ((ssv[4] + ssv[5]) + (ssv[6] + ssv[7]));
            rs = 1.0f / sqrtf(ss / static_cast<float>(K) + fuse.eps); // rmsnorm.hip's formula
        }
        gemm_epilogue<T, MODE>(acc, out_, &lds[0][0], 32 * 40, wave, lane, is_gate, tile_ok, col0, n0, ksplit, M, N,
                               out_stride, rs, HWV);
    } else {
        gemm_epilogue<T, MODE>(acc, out_, &lds[PACKED ? 0 : 2][0], 32 * kKT, wave, lane, is_gate, tile_ok, col0, n0,
                               ksplit, M, N, out_stride);
    }
}

// out[m][n] = round(sum over splits, in split order) — 4 outputs per thread.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(T *__restrict__ out,
                                                            const float *__restrict__ slabs, int M,
                                                            int N, int KS, int64_t out_stride) {
    const int n4 = N >> 2;
    const int64_t items = static_cast<int64_t>(M) * n4;
    for (int64_t it = blockIdx.x * 256ll + threadIdx.x; it < items;
         it += static_cast<int64_t>(gridDim.x) * 256ll) {
        const int m = static_cast<int>(it / n4);
        const int c = static_cast<int>(it - static_cast<int64_t>(m) * n4);
        float4_t s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < KS; ++k) {
            const float4_t v = *reinterpret_cast<const float4_t *>(
                slabs + (static_cast<int64_t>(k) * M + m) * N + 4 * c);
            s += v;
        }
        typedef T vec4 __attribute__((ext_vector_type(4)));
        vec4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = to_t<T>(s[e]);
        *reinterpret_cast<vec4 *>(out + static_cast<int64_t>(m) * out_stride + 4 * c) = o;
    }
}

// K-splits: the smallest power of two that launches >= 768 waves (3 per CU; each keeps 8-16 KiB of W
// in flight), never leaving a split fewer than two K-tiles. Measured on MI355X at M = 32 (bf16,
// tools/gemm_micro.py): N=6144,K=4096 best at 4 splits (15.6 us), N=4096,K=4096 at 8 (13.2 us),
// N=4096,K=14336 at 8 (26.4 us), N=28672 and N=128256 unsplit (5.25 / 5.22 TB/s): more splits only add
// slab traffic and reduce work.
static int choose_k_splits(int N, int K) {
    const int tiles = N / 32;
    int ks = 1;
    while (ks < 16 && tiles * ks < 768 && K % (kKT * ks * 2) == 0 && K / (ks * 2) >= 2 * kKT) ks *= 2;
    return ks;
}

// Packed weights only: when K has no power-of-two split of whole tiles that fills the chip (the even rule above stops
// at 2 splits = 64 workgroups for N = 4096, K = 11008), the splits may differ by one tile: the smallest power of two
// that launches >= 768 waves with >= 8 tiles per split (ring kernel), as the even rule would have picked.
static int choose_k_splits_packed(int N, int K, bool *uneven) {
    const int even = choose_k_splits(N, K);
    *uneven = false;
    const int tiles = N / 32, ktiles = K / kKT;
    if (tiles * even >= 512) return even;
    int ks = even;
    while (ks < 16 && tiles * ks < 768 && ktiles / (ks * 2) >= 8) ks *= 2;
    if (ks == even) return even;
    *uneven = K % (kKT * ks) != 0;
    return ks;
}

// K-chunks of >= 8 tiles amortise the ring's barriers; shorter ones keep the barrier-free kernel.
static bool use_ring(int kc) { return kc / kKT >= 8; }

// reduce == false: stop after the partial slabs (a fused consumer sums them: swl_splitk_*)
template <typename T>
static int run_gemm(T *out, const T *x, const T *w, float *ws, size_t ws_bytes, int M, int N, int K,
                    int64_t xs, int64_t os, int ks, hipStream_t stream, bool reduce = true) {
    if (ks <= 0) ks = choose_k_splits(N, K);
    if (K % (kKT * ks) != 0) return SWL_ERR_UNSUPPORTED;
    const int tiles = N / 32;
    const dim3 grid((tiles + kGemmWaves - 1) / kGemmWaves, ks);
    const int kc = K / ks;
    const bool ring = use_ring(kc);
    if (ks == 1 && reduce) {
        if (ring)
            hipLaunchKernelGGL((gemm_skinny_ring_kernel<T, kGemmDirect>), grid, dim3(kGemmWaves * 64), 0, stream,
                               out, x, w, M, N, K, kc, xs, os, GemmExtra{});
        else
            hipLaunchKernelGGL((gemm_skinny_kernel<T, kGemmDirect>), grid, dim3(kGemmWaves * 64), 0, stream, out,
                               x, w, M, N, K, kc, xs, os);
        return check_launch();
    }
    if (!ws || ws_bytes < static_cast<size_t>(ks) * M * N * sizeof(float)) return SWL_ERR_BAD_ARG;
    if (ring)
        hipLaunchKernelGGL((gemm_skinny_ring_kernel<T, kGemmPartial>), grid, dim3(kGemmWaves * 64), 0, stream, ws,
                           x, w, M, N, K, kc, xs, static_cast<int64_t>(N), GemmExtra{});
    else
        hipLaunchKernelGGL((gemm_skinny_kernel<T, kGemmPartial>), grid, dim3(kGemmWaves * 64), 0, stream, ws, x, w,
                           M, N, K, kc, xs, static_cast<int64_t>(N));
    if (!reduce) return check_launch();
    const int64_t items = static_cast<int64_t>(M) * (N / 4);
    const unsigned rgrid = static_cast<unsigned>((items + 255) / 256);
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(rgrid), dim3(256), 0, stream, out, ws, M, N, ks,
                       os);
    return check_launch();
}

} // namespace swl

extern "C" size_t swl_gemm_skinny_workspace_bytes(int32_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    bool uneven = false;
    const int ks = swl::choose_k_splits_packed(N, K, &uneven); // (>= the row-major kernels' choice)
    return ks > 1 ? static_cast<size_t>(16) * M * N * sizeof(float) : 0; // room for any legal override
}

extern "C" int swl_gemm_skinny(void *out, const void *x, const void *w, void *workspace,
                               size_t workspace_bytes, int32_t M, int32_t N, int32_t K,
                               int64_t x_row_stride, int64_t out_row_stride, int32_t k_splits,
                               int32_t dtype, swl_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!out || !x || !w) return SWL_ERR_BAD_ARG;
    if (M > 32 || (N & 31) || (K & (swl::kKT - 1))) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || out_row_stride < N || (x_row_stride & 7) || (out_row_stride & 3))
        return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(x) || !swl::aligned16(w) || (reinterpret_cast<uintptr_t>(out) & 7u) ||
        (workspace && !swl::aligned16(workspace)))
        return SWL_ERR_BAD_ARG;
    if (k_splits < 0 || k_splits > 16 || (k_splits & (k_splits - 1))) return SWL_ERR_BAD_ARG;
    SWL_DISPATCH_DTYPE(dtype, T, {
        return swl::run_gemm<T>(static_cast<T *>(out), static_cast<const T *>(x),
                                static_cast<const T *>(w), static_cast<float *>(workspace),
                                workspace_bytes, M, N, K, x_row_stride, out_row_stride, k_splits,
                                static_cast<hipStream_t>(stream));
    });
}

extern "C" int swl_gemm_skinny_choose_splits(int32_t N, int32_t K) {
    if (N <= 0 || K <= 0 || (N & 31) || (K & (swl::kKT - 1))) return 0;
    return swl::choose_k_splits(N, K);
}

/* The split count swl_gemm_skinny_packed / _packed_partial pick with k_splits = 0: as above, except that splits may
 * differ by one K-tile when K has no power-of-two split of whole tiles that fills the chip (K = 11008). */
extern "C" int swl_gemm_skinny_packed_choose_splits(int32_t N, int32_t K) {
    if (N <= 0 || K <= 0 || (N & 31) || (K & (swl::kKT - 1))) return 0;
    bool uneven = false;
    return swl::choose_k_splits_packed(N, K, &uneven);
}

/* Partial slabs only: slabs[k_splits][M][N] fp32, k_splits = swl_gemm_skinny_choose_splits(N, K) > 1. */
extern "C" int swl_gemm_skinny_partial(float *slabs, size_t slabs_bytes, const void *x, const void *w,
                                       int32_t M, int32_t N, int32_t K, int64_t x_row_stride,
                                       int32_t k_splits, int32_t dtype, swl_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!slabs || !x || !w || k_splits < 1 || k_splits > 16 || (k_splits & (k_splits - 1)))
        return SWL_ERR_BAD_ARG;
    if (M > 32 || (N & 31) || (K & (swl::kKT - 1))) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || (x_row_stride & 7) || !swl::aligned16(x) || !swl::aligned16(w) ||
        !swl::aligned16(slabs))
        return SWL_ERR_BAD_ARG;
    SWL_DISPATCH_DTYPE(dtype, T, {
        return swl::run_gemm<T>(static_cast<T *>(nullptr), static_cast<const T *>(x),
                                static_cast<const T *>(w), slabs, slabs_bytes, M, N, K, x_row_stride, N,
                                k_splits, static_cast<hipStream_t>(stream), false);
    });
}

/* out[M, N] = round(sum_k slabs[k]) in slab order (what swl_gemm_skinny does internally). */
extern "C" int swl_splitk_reduce(void *out, const float *slabs, int32_t k_splits, int32_t M, int32_t N,
                                 int64_t out_row_stride, int32_t dtype, swl_stream_t stream) {
    if (M < 0 || N <= 0 || (N & 3) || k_splits <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!out || !slabs || out_row_stride < N || (out_row_stride & 3)) return SWL_ERR_BAD_ARG;
    const int64_t items = static_cast<int64_t>(M) * (N / 4);
    const unsigned grid = static_cast<unsigned>((items + 255) / 256);
    SWL_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL((swl::splitk_reduce_kernel<T>), dim3(grid), dim3(256), 0,
                           static_cast<hipStream_t>(stream), static_cast<T *>(out), slabs, M, N,
                           k_splits, out_row_stride);
    });
    return swl::check_launch();
}

/* FFN up/gate projection with the SiLU-gate fused into the epilogue:
 * out[M, I] = (x . W[0:I]^T) * silu(x . W[I:2I]^T), W = [up ; gate] as weight.py:133 builds it.
 * Bit-identical to swl_gemm_skinny (k_splits = 1) followed by swl_silu_mul, minus one launch and the
 * [M, 2*I] round trip (reference: transformer_layer.py:126-127). I % 32 == 0, K % 128 == 0, M <= 32. */
extern "C" int swl_gemm_skinny_silu_gate(void *out, const void *x, const void *w_up_gate, int32_t M,
                                         int32_t I, int32_t K, int64_t x_row_stride,
                                         int64_t out_row_stride, int32_t dtype, swl_stream_t stream) {
    if (M < 0 || I <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!out || !x || !w_up_gate) return SWL_ERR_BAD_ARG;
    if (M > 32 || (I & 31) || (K & (swl::kKT - 1))) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || out_row_stride < I || (x_row_stride & 7) || (out_row_stride & 3))
        return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(x) || !swl::aligned16(w_up_gate) || (reinterpret_cast<uintptr_t>(out) & 7u))
        return SWL_ERR_BAD_ARG;
    const dim3 grid((I / 32 + 1) / 2, 1);
    SWL_DISPATCH_DTYPE(dtype, T, {
        if (swl::use_ring(K))
            hipLaunchKernelGGL((swl::gemm_skinny_ring_kernel<T, swl::kGemmSiluGate>), grid,
                               dim3(swl::kGemmWaves * 64), 0, static_cast<hipStream_t>(stream), out,
                               static_cast<const T *>(x), static_cast<const T *>(w_up_gate), M, I, K, K,
                               x_row_stride, out_row_stride, swl::GemmExtra{});
        else
            hipLaunchKernelGGL((swl::gemm_skinny_kernel<T, swl::kGemmSiluGate>), grid,
                               dim3(swl::kGemmWaves * 64), 0, static_cast<hipStream_t>(stream), out,
                               static_cast<const T *>(x), static_cast<const T *>(w_up_gate), M, I, K, K,
                               x_row_stride, out_row_stride);
    });
    return swl::check_launch();
}

// ---- pre-packed weights (PACKED ring kernel) ------------------------------------------------------------------
namespace swl {

// dst[N/32][K/16][64][8]: lane l of block (nt, k16) holds src[nt*32 + l%32][k16*16 + 8*(l/32) .. +8]
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_kernel(T *__restrict__ dst, const T *__restrict__ src, int N, int K) {
    const int64_t blocks = static_cast<int64_t>(N / 32) * (K / 16);
    const int lane = threadIdx.x & 63;
    for (int64_t b = blockIdx.x * 4ll + (threadIdx.x >> 6); b < blocks; b += gridDim.x * 4ll) {
        const int64_t nt = b / (K / 16);
        const int k16 = static_cast<int>(b - nt * (K / 16));
        const T *s = src + (nt * 32 + (lane & 31)) * K + k16 * 16 + 8 * (lane >> 5);
        store8(dst + (b * 64 + lane) * 8, load8(s));
    }
}

// Partial slabs on a packed weight whose tile count divides by three and fills the chip better in three-wave workgroups
// (see gemm_skinny_ring_kernel's NWV): same bits, other launch geometry.
static bool prefer_three_waves(int N, int ks) {
    const int tiles = N / 32;
    if (tiles % 3 != 0) return false;
    const int wg4 = (tiles + kGemmWaves - 1) / kGemmWaves * ks, wg3 = tiles / 3 * ks;
    return wg4 < 256 && wg3 <= 256;
}

template <typename T>
static void launch_packed_partial3(hipStream_t stream, void *out, const T *x, const T *wp, int M, int N, int K, int kc,
                                   int ks, int64_t xs) {
    const dim3 grid(N / 32 / 3, ks);
    if (use_ring(kc))
        hipLaunchKernelGGL((gemm_skinny_ring_kernel<T, kGemmPartial, true, 3, 3>), grid, dim3(3 * 64), 0,
                           stream, out, x, wp, M, N, K, kc, xs, static_cast<int64_t>(N), GemmExtra{});
    else
        hipLaunchKernelGGL((gemm_skinny_ring_kernel<T, kGemmPartial, true, 2, 3>), grid, dim3(3 * 64), 0,
                           stream, out, x, wp, M, N, K, kc, xs, static_cast<int64_t>(N), GemmExtra{});
}

template <typename T, int MODE>
static void launch_packed(dim3 grid, hipStream_t stream, void *out, const T *x, const T *wp, int M, int N, int K, int kc,
                          int64_t xs, int64_t os, const GemmExtra &fuse = GemmExtra{}) {
    if (use_ring(kc))
        hipLaunchKernelGGL((gemm_skinny_ring_kernel<T, MODE, true, 3>), grid, dim3(kGemmWaves * 64), 0,
                           stream, out, x, wp, M, N, K, kc, xs, os, fuse);
    else
        hipLaunchKernelGGL((gemm_skinny_ring_kernel<T, MODE, true, 2>), grid, dim3(kGemmWaves * 64), 0,
                           stream, out, x, wp, M, N, K, kc, xs, os, fuse);
}

template <typename T>
static int run_gemm_packed(T *out, const T *x, const T *wp, float *ws, size_t ws_bytes, int M, int N, int K, int64_t xs,
                           int64_t os, int ks, hipStream_t stream, bool reduce) {
    bool uneven = false;
    if (ks <= 0) ks = choose_k_splits_packed(N, K, &uneven);
    else uneven = K % (kKT * ks) != 0;
    if (uneven && (K / kKT) / ks < 8) return SWL_ERR_UNSUPPORTED; // (uneven splits run the ring kernel only)
    const dim3 grid((N / 32 + kGemmWaves - 1) / kGemmWaves, ks);
    const int kc = uneven ? (K / kKT / ks) * kKT : K / ks;  // (uneven: the shorter chunk — selects the ring depth)
    if (ks == 1 && reduce) {
        launch_packed<T, kGemmDirect>(grid, stream, out, x, wp, M, N, K, kc, xs, os);
        return check_launch();
    }
    if (!ws || ws_bytes < static_cast<size_t>(ks) * M * N * sizeof(float)) return SWL_ERR_BAD_ARG;
    GemmExtra fuse{};
    fuse.k_tiles_total = uneven ? K / kKT : 0;
    if (!uneven && prefer_three_waves(N, ks)) launch_packed_partial3<T>(stream, ws, x, wp, M, N, K, kc, ks, xs);
    else launch_packed<T, kGemmPartial>(grid, stream, ws, x, wp, M, N, K, kc, xs, static_cast<int64_t>(N), fuse);
    if (!reduce) return check_launch();
    const int64_t items = static_cast<int64_t>(M) * (N / 4);
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(static_cast<unsigned>((items + 255) / 256)), dim3(256), 0,
                       stream, out, ws, M, N, ks, os);
    return check_launch();
}

} // namespace swl

/* dst <- W repacked in MFMA-fragment order ([N/32][K/16][64 lanes][8 elements], same size as W) for the
 * swl_gemm_skinny_packed* entry points. Done once per weight at load time. N % 32 == 0, K % 16 == 0. */
extern "C" int swl_gemm_pack_weight(void *dst, const void *src, int32_t N, int32_t K, int32_t dtype,
                                    swl_stream_t stream) {
    if (N <= 0 || K <= 0 || !dst || !src || dst == src) return SWL_ERR_BAD_ARG;
    if ((N & 31) || (K & 15)) return SWL_ERR_UNSUPPORTED;
    if (!swl::aligned16(dst) || !swl::aligned16(src)) return SWL_ERR_BAD_ARG;
    const int64_t blocks = static_cast<int64_t>(N / 32) * (K / 16);
    const unsigned grid = static_cast<unsigned>(blocks / 4 + 1 < 65536 ? blocks / 4 + 1 : 65536);
    SWL_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL((swl::pack_weight_kernel<T>), dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<T *>(dst), static_cast<const T *>(src), N, K);
    });
    return swl::check_launch();
}

/* swl_gemm_skinny / swl_gemm_skinny_partial / swl_gemm_skinny_silu_gate on a weight packed by
 * swl_gemm_pack_weight: same arguments, same results bit for bit; K % 128 == 0, N (I) % 32 == 0, M <= 32. */
extern "C" int swl_gemm_skinny_packed(void *out, const void *x, const void *w_packed, void *workspace,
                                      size_t workspace_bytes, int32_t M, int32_t N, int32_t K, int64_t x_row_stride,
                                      int64_t out_row_stride, int32_t k_splits, int32_t dtype, swl_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!out || !x || !w_packed) return SWL_ERR_BAD_ARG;
    if (M > 32 || (N & 31) || (K & (swl::kKT - 1))) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || out_row_stride < N || (x_row_stride & 7) || (out_row_stride & 3)) return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(x) || !swl::aligned16(w_packed) || (reinterpret_cast<uintptr_t>(out) & 7u) ||
        (workspace && !swl::aligned16(workspace)))
        return SWL_ERR_BAD_ARG;
    if (k_splits < 0 || k_splits > 16 || (k_splits & (k_splits - 1))) return SWL_ERR_BAD_ARG;
    SWL_DISPATCH_DTYPE(dtype, T, {
        return swl::run_gemm_packed<T>(static_cast<T *>(out), static_cast<const T *>(x),
                                       static_cast<const T *>(w_packed), static_cast<float *>(workspace),
                                       workspace_bytes, M, N, K, x_row_stride, out_row_stride, k_splits,
                                       static_cast<hipStream_t>(stream), true);
    });
}

extern "C" int swl_gemm_skinny_packed_partial(float *slabs, size_t slabs_bytes, const void *x, const void *w_packed,
                                              int32_t M, int32_t N, int32_t K, int64_t x_row_stride,
                                              int32_t k_splits, int32_t dtype, swl_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!slabs || !x || !w_packed || k_splits < 1 || k_splits > 16 || (k_splits & (k_splits - 1)))
        return SWL_ERR_BAD_ARG;
    if (M > 32 || (N & 31) || (K & (swl::kKT - 1))) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || (x_row_stride & 7) || !swl::aligned16(x) || !swl::aligned16(w_packed) ||
        !swl::aligned16(slabs))
        return SWL_ERR_BAD_ARG;
    SWL_DISPATCH_DTYPE(dtype, T, {
        return swl::run_gemm_packed<T>(static_cast<T *>(nullptr), static_cast<const T *>(x),
                                       static_cast<const T *>(w_packed), slabs, slabs_bytes, M, N, K, x_row_stride, N,
                                       k_splits, static_cast<hipStream_t>(stream), false);
    });
}

extern "C" int swl_gemm_skinny_packed_silu_gate(void *out, const void *x, const void *w_up_gate_packed, int32_t M,
                                                int32_t I, int32_t K, int64_t x_row_stride, int64_t out_row_stride,
                                                int32_t dtype, swl_stream_t stream) {
    if (M < 0 || I <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!out || !x || !w_up_gate_packed) return SWL_ERR_BAD_ARG;
    if (M > 32 || (I & 31) || (K & (swl::kKT - 1))) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || out_row_stride < I || (x_row_stride & 7) || (out_row_stride & 3)) return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(x) || !swl::aligned16(w_up_gate_packed) || (reinterpret_cast<uintptr_t>(out) & 7u))
        return SWL_ERR_BAD_ARG;
    const dim3 grid((I / 32 + 1) / 2, 1);
    SWL_DISPATCH_DTYPE(dtype, T, {
        swl::launch_packed<T, swl::kGemmSiluGate>(grid, static_cast<hipStream_t>(stream), out, static_cast<const T *>(x),
                                                  static_cast<const T *>(w_up_gate_packed), M, I, K, K, x_row_stride,
                                                  out_row_stride);
    });
    return swl::check_launch();
}

/* swl_gemm_skinny_packed_silu_gate on an input whose RMSNorm scale is still pending (deferred normalisation,
 * swl_splitk_add_scale): x = round(residual * w_norm), row_ssq[ssq_parts][M] the per-1024-column sums of squares of the
 * residual rows; out[m, :] = up * silu(gate) of rstd[m] * (x . [up ; gate]^T), rstd = 1/sqrt(sum_p row_ssq[p][m] / K + eps),
 * the scale applied in fp32 before the projection's rounding. ssq_parts <= 8. Replaces fused_add_rmsnorm + linear +
 * silu_and_mul of transformer_layer.py:120-127 on the decode fast path. */
extern "C" int swl_gemm_skinny_packed_silu_gate_rs(void *out, const void *x, const void *w_up_gate_packed,
                                                   const float *row_ssq, int32_t ssq_parts, float eps, int32_t M,
                                                   int32_t I, int32_t K, int64_t x_row_stride, int64_t out_row_stride,
                                                   int32_t dtype, swl_stream_t stream) {
    if (M < 0 || I <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!out || !x || !w_up_gate_packed || !row_ssq || ssq_parts <= 0) return SWL_ERR_BAD_ARG;
    if (M > 32 || (I & 31) || (K & (swl::kKT - 1)) || ssq_parts > 8) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || out_row_stride < I || (x_row_stride & 7) || (out_row_stride & 3)) return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(x) || !swl::aligned16(w_up_gate_packed) || (reinterpret_cast<uintptr_t>(out) & 7u))
        return SWL_ERR_BAD_ARG;
    swl::GemmExtra f{};
    f.ssq_in = row_ssq;
    f.ssq_parts = ssq_parts;
    f.eps = eps;
    const dim3 grid((I / 32 + 1) / 2, 1);
    SWL_DISPATCH_DTYPE(dtype, T, {
        swl::launch_packed<T, swl::kGemmSiluGate>(grid, static_cast<hipStream_t>(stream), out, static_cast<const T *>(x),
                                                  static_cast<const T *>(w_up_gate_packed), M, I, K, K, x_row_stride,
                                                  out_row_stride, f);
    });
    return swl::check_launch();
}

/* ---- norm on the fly: the projections that consume the RAW residual stream swl_gemm_rows_add leaves behind -------------
 * x = the residual rows r[M, K] (no consumer launch computed round(r * norm_w) or the sums of squares); the kernel stages
 * round(r * norm_w) itself (swl_splitk_add_scale's bits) and adds up sum r^2 per row in fp32 while it does.
 *   swl_gemm_skinny_packed_partial_nf: swl_gemm_skinny_packed_partial (even splits) + ssq_out[k_splits][M], the row_ssq
 *     swl_paged_attn_decode_qkv_rs takes with ssq_parts = k_splits. The fused qkv projection, transformer_layer.py:46-56.
 *   swl_gemm_skinny_packed_silu_gate_nf: swl_gemm_skinny_packed_silu_gate_rs with rstd from its own sums.
 *     transformer_layer.py:120-127. */
extern "C" int swl_gemm_skinny_packed_partial_nf(float *slabs, size_t slabs_bytes, float *ssq_out, const void *x,
                                                 const void *norm_w, const void *w_packed, int32_t M, int32_t N, int32_t K,
                                                 int64_t x_row_stride, int32_t k_splits, int32_t dtype,
                                                 swl_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!slabs || !ssq_out || !x || !norm_w || !w_packed || k_splits < 1 || k_splits > 16 || (k_splits & (k_splits - 1)))
        return SWL_ERR_BAD_ARG;
    if (M > 32 || (N & 31) || K % (swl::kKT * k_splits)) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || (x_row_stride & 7) || !swl::aligned16(x) || !swl::aligned16(w_packed) ||
        !swl::aligned16(norm_w) || !swl::aligned16(slabs))
        return SWL_ERR_BAD_ARG;
    if (slabs_bytes < static_cast<size_t>(k_splits) * M * N * sizeof(float)) return SWL_ERR_BAD_ARG;
    swl::GemmExtra f{};
    f.norm_w = norm_w;
    f.ssq_out = ssq_out;
    const int kc = K / k_splits;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t n64 = N;
#define SWL_NF_PARTIAL(RD_, NWV_)                                                                                        \
    hipLaunchKernelGGL((swl::gemm_skinny_ring_kernel<T, swl::kGemmPartial, true, RD_, NWV_, true>), grid,          \
                       dim3(NWV_ * 64), 0, s, static_cast<void *>(slabs), static_cast<const T *>(x),                     \
                       static_cast<const T *>(w_packed), M, N, K, kc, x_row_stride, n64, f)
    SWL_DISPATCH_DTYPE(dtype, T, {
        // (three-wave workgroups stage 3 row groups per wave: with the norm weights and the fp32 staging temporaries a
        // 3-deep weight ring no longer fits 256 registers — 11-30 spilled; they run the 2-deep ring. A/B on MI355X: the
        // spilling deep ring and 4-wave workgroups with the deep ring — 192 instead of 256 of them for Llama-3-8B — are
        // within noise of it in the layer chain: profiles/r05c_rows_layer_micro_nf_rd3 / _nf_4w.jsonl.)
        if (swl::prefer_three_waves(N, k_splits)) {
            const dim3 grid(N / 32 / 3, k_splits);
            SWL_NF_PARTIAL(2, 3);
        } else {
            const dim3 grid((N / 32 + swl::kGemmWaves - 1) / swl::kGemmWaves, k_splits);
            if (swl::use_ring(kc)) SWL_NF_PARTIAL(3, 4);
            else SWL_NF_PARTIAL(2, 4);
        }
    });
#undef SWL_NF_PARTIAL
    return swl::check_launch();
}

extern "C" int swl_gemm_skinny_packed_silu_gate_nf(void *out, const void *x, const void *norm_w, float eps,
                                                   const void *w_up_gate_packed, int32_t M, int32_t I, int32_t K,
                                                   int64_t x_row_stride, int64_t out_row_stride, int32_t dtype,
                                                   swl_stream_t stream) {
    if (M < 0 || I <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!out || !x || !norm_w || !w_up_gate_packed) return SWL_ERR_BAD_ARG;
    if (M > 32 || (I & 31) || (K & (swl::kKT - 1))) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || out_row_stride < I || (x_row_stride & 7) || (out_row_stride & 3)) return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(x) || !swl::aligned16(w_up_gate_packed) || !swl::aligned16(norm_w) ||
        (reinterpret_cast<uintptr_t>(out) & 7u))
        return SWL_ERR_BAD_ARG;
    swl::GemmExtra f{};
    f.norm_w = norm_w;
    f.eps = eps;
    const dim3 grid((I / 32 + 1) / 2, 1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // r06b: up to 8 tokens, workgroups of ONE up tile + ONE gate tile (448 workgroups for Llama-3-8B instead of 224: every CU
    // streams; one wave per SIMD, two workgroups per CU) — same K order per tile, hence the same bits. Each wave then stages
    // twice the x rows, which costs more than the idle CUs from 9 tokens on (tools/gpu_silu_waves_ab.sh, four interleaved
    // rounds, profiles/r06b_silu_gate_two_wave_groups_ab.jsonl: 39.2 vs 39.8 us at 1 token, 39.2 vs 40.0 at 8, 41.2 vs 40.3
    // at 32). A/B switch: SWL_SILU_WAVES=2 / 4 forces one form for every M.
    static const int forced = [] { const char *e = getenv("SWL_SILU_WAVES"); return !e ? 0 : e[0] == '2' ? 2 : e[0] == '4' ? 4 : 0; }();
    const bool two_waves = forced ? forced == 2 : M <= 8;
    if (two_waves && swl::use_ring(K)) {
        SWL_DISPATCH_DTYPE(dtype, T, {
            hipLaunchKernelGGL((swl::gemm_skinny_ring_kernel<T, swl::kGemmSiluGate, true, 3, 2, true>), dim3(I / 32, 1),
                               dim3(128), 0, s, out, static_cast<const T *>(x), static_cast<const T *>(w_up_gate_packed), M,
                               I, K, K, x_row_stride, out_row_stride, f);
        });
        return swl::check_launch();
    }
    SWL_DISPATCH_DTYPE(dtype, T, {
        if (swl::use_ring(K))
            hipLaunchKernelGGL((swl::gemm_skinny_ring_kernel<T, swl::kGemmSiluGate, true, 3, swl::kGemmWaves, true>),
                               grid, dim3(swl::kGemmWaves * 64), 0, s, out, static_cast<const T *>(x),
                               static_cast<const T *>(w_up_gate_packed), M, I, K, K, x_row_stride, out_row_stride, f);
        else
            hipLaunchKernelGGL((swl::gemm_skinny_ring_kernel<T, swl::kGemmSiluGate, true, 2, swl::kGemmWaves, true>),
                               grid, dim3(swl::kGemmWaves * 64), 0, s, out, static_cast<const T *>(x),
                               static_cast<const T *>(w_up_gate_packed), M, I, K, K, x_row_stride, out_row_stride, f);
    });
    return swl::check_launch();
}

/* ---- the EXACT norm on the fly (r06c): the reference's rounding points on the row-owned decode path, float16 included --------
 * x = the raw residual rows r[M, K] that swl_gemm_rows_add_ssq left behind, ssq_in[M][ssq_parts] its per-tile sums of squares
 * (ssq_parts = K / 16 for a row-owned projection; % 64 == 0). The workgroup adds the partials in a fixed order and stages
 * round(r * rstd * norm_w), rstd = 1/sqrt(sum / K + eps) — rmsnorm.hip's expression, reference rmsnorm.py:57-64 — so the
 * products are those of rmsnorm + projection up to the summation order of the sums of squares. */
extern "C" int swl_gemm_skinny_packed_silu_gate_nx(void *out, const void *x, const void *norm_w, float eps, const float *ssq_in,
                                                   int32_t ssq_parts, const void *w_up_gate_packed, int32_t M, int32_t I,
                                                   int32_t K, int64_t x_row_stride, int64_t out_row_stride, int32_t dtype,
                                                   swl_stream_t stream) {
    if (M < 0 || I <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!out || !x || !norm_w || !w_up_gate_packed || !ssq_in) return SWL_ERR_BAD_ARG;
    if (M > 32 || (I & 31) || (K & (swl::kKT - 1)) || ssq_parts <= 0 || (ssq_parts & 63) || !swl::use_ring(K)) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || out_row_stride < I || (x_row_stride & 7) || (out_row_stride & 3)) return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(x) || !swl::aligned16(w_up_gate_packed) || !swl::aligned16(norm_w) || !swl::aligned16(ssq_in) ||
        (reinterpret_cast<uintptr_t>(out) & 7u))
        return SWL_ERR_BAD_ARG;
    swl::GemmExtra f{};
    f.norm_w = norm_w;
    f.eps = eps;
    f.ssq_in = ssq_in;
    f.ssq_parts = ssq_parts;
    hipStream_t s = static_cast<hipStream_t>(stream);
    SWL_DISPATCH_DTYPE(dtype, T, {
        if (M <= 8)     // (two-wave workgroups: see swl_gemm_skinny_packed_silu_gate_nf)
            hipLaunchKernelGGL((swl::gemm_skinny_ring_kernel<T, swl::kGemmSiluGate, true, 3, 2, true, true>), dim3(I / 32, 1),
                               dim3(128), 0, s, out, static_cast<const T *>(x), static_cast<const T *>(w_up_gate_packed), M,
                               I, K, K, x_row_stride, out_row_stride, f);
        else
            hipLaunchKernelGGL((swl::gemm_skinny_ring_kernel<T, swl::kGemmSiluGate, true, 3, swl::kGemmWaves, true, true>),
                               dim3((I / 32 + 1) / 2, 1), dim3(swl::kGemmWaves * 64), 0, s, out, static_cast<const T *>(x),
                               static_cast<const T *>(w_up_gate_packed), M, I, K, K, x_row_stride, out_row_stride, f);
    });
    return swl::check_launch();
}

/* slabs[k_splits][M][N] of round(r * rstd * norm_w) . W^T (nothing pending: swl_paged_attn_decode_qkv takes them as they are).
 * Even K splits of whole 128-column tiles; N / 32 % 3 == 0 runs three-wave workgroups as swl_gemm_skinny_packed_partial_nf. */
extern "C" int swl_gemm_skinny_packed_partial_nx(float *slabs, size_t slabs_bytes, const void *x, const void *norm_w, float eps,
                                                 const float *ssq_in, int32_t ssq_parts, const void *w_packed, int32_t M,
                                                 int32_t N, int32_t K, int64_t x_row_stride, int32_t k_splits, int32_t dtype,
                                                 swl_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!slabs || !x || !norm_w || !w_packed || !ssq_in || k_splits < 1 || k_splits > 16) return SWL_ERR_BAD_ARG;
    if (M > 32 || (N & 31) || (K & (swl::kKT - 1)) || K % (swl::kKT * k_splits) || ssq_parts <= 0 || (ssq_parts & 63))
        return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || (x_row_stride & 7)) return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(x) || !swl::aligned16(w_packed) || !swl::aligned16(norm_w) || !swl::aligned16(slabs) || !swl::aligned16(ssq_in))
        return SWL_ERR_BAD_ARG;
    if (slabs_bytes < static_cast<size_t>(k_splits) * M * N * sizeof(float)) return SWL_ERR_BAD_ARG;
    swl::GemmExtra f{};
    f.norm_w = norm_w;
    f.eps = eps;
    f.ssq_in = ssq_in;
    f.ssq_parts = ssq_parts;
    const int kc = K / k_splits;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t n64 = N;
#define SWL_NX_PARTIAL(RD_, NWV_)                                                                                        \
    hipLaunchKernelGGL((swl::gemm_skinny_ring_kernel<T, swl::kGemmPartial, true, RD_, NWV_, true, true>), grid,    \
                       dim3(NWV_ * 64), 0, s, static_cast<void *>(slabs), static_cast<const T *>(x),                     \
                       static_cast<const T *>(w_packed), M, N, K, kc, x_row_stride, n64, f)
    SWL_DISPATCH_DTYPE(dtype, T, {
        if (swl::prefer_three_waves(N, k_splits)) {
            const dim3 grid(N / 32 / 3, k_splits);
            SWL_NX_PARTIAL(2, 3);
        } else {
            const dim3 grid((N / 32 + swl::kGemmWaves - 1) / swl::kGemmWaves, k_splits);
            if (swl::use_ring(kc)) SWL_NX_PARTIAL(3, 4);
            else SWL_NX_PARTIAL(2, 4);
        }
    });
#undef SWL_NX_PARTIAL
    return swl::check_launch();
}

// ---- medium batches on packed weights: 32 < M <= 64 ------------------------------------------------------------
// With W arriving in fragment order there is no LDS traffic for W at all, so MT = 2 or 4 blocks of 32 tokens can
// share every W fragment at the price of MT x-fragment reads per k-step (1 LDS read per MFMA; the row-major attempt
// paid 1 + MT reads plus the W round trip and was LDS bound: profiles/r01e_gemm_mtile_experiment.jsonl). hipBLASLt
// streams these shapes at 1.3-4.7 TB/s (profiles/r01e_hipblaslt_m48_256.jsonl).
namespace swl {

// MODE as in the M <= 32 kernels: kGemmDirect, kGemmPartial (fp32 slabs) or kGemmSiluGate (W = [up ; gate], N = I:
// waves 0,1 own `up` tiles, waves 2,3 the `gate` tiles of the same columns; the activated gate tiles change hands
// through the x buffers after the last K-tile).
template <typename T, int MT, int MODE>
__global__ __launch_bounds__(kGemmWaves * 64, 2) void gemm_packed_mt_kernel(
    void *__restrict__ out_, const T *__restrict__ x, const T *__restrict__ wpk, int M, int N, int K, int kc,
    int64_t x_stride, int64_t out_stride) {
    constexpr bool PARTIAL = MODE == kGemmPartial;
    constexpr int D = kRing;
    constexpr int XL = MT * 8 / kGemmWaves; // x row-groups (4 rows each) a wave stages per tile
    constexpr int kXTile = MT * 32 * kKT;
    __shared__ __attribute__((aligned(16))) T xs[2 * kXTile]; // double-buffered [32*MT][128] x tile (<= 64 KiB)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool is_gate = MODE == kGemmSiluGate && wave >= 2;
    const int col0 = MODE == kGemmSiluGate ? (blockIdx.x * 2 + (wave & 1)) * 32
                                           : (blockIdx.x * kGemmWaves + wave) * 32;
    const bool tile_ok = col0 < N;
    const int nt = tile_ok ? (col0 + (is_gate ? N : 0)) / 32 : 0;
    const int ksplit = blockIdx.y;
    const int k_begin = ksplit * kc;
    const int nkt = kc / kKT;
    const int rsub = lane >> 4, chunk = lane & 15;
    const T *wsrc = wpk + (static_cast<int64_t>(nt) * (K / 16) + k_begin / 16) * 512 + lane * 8;
    const T *xsrc[XL];
    int xs_wr[XL];
#pragma unroll
    for (int q = 0; q < XL; ++q) {
        const int row = 4 * (wave * XL + q) + rsub;
        xsrc[q] = x + static_cast<int64_t>(min(row, M - 1)) * x_stride + k_begin + chunk * 8;
        xs_wr[q] = row * kKT + ((chunk ^ (row & 15)) << 3);
    }
    const int l32 = lane & 31, hf = lane >> 5;

    vec8_t<T> wr[D][8], xr[XL];
    float16_t acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = float16_t{};
    // x one tile ahead, requested BEFORE that step's W request (loads return in order: staging it at the end of the
    // step waits for nothing younger than itself); W D-1 tiles ahead
#define SWL_MT_ISSUE_W(slot, tile)                                                                    \
    {                                                                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                             \
            wr[slot][i_] = load8_nt(wsrc + (static_cast<int64_t>(tile) * 8 + i_) * 512);              \
    }
#define SWL_MT_ISSUE_X(tile)                                                                          \
    {                                                                                                \
        _Pragma("unroll") for (int q_ = 0; q_ < XL; ++q_) xr[q_] = load8(xsrc[q_] + (tile) * kKT);   \
    }
#define SWL_MT_STAGE_X(buf)                                                                           \
    {                                                                                                \
        _Pragma("unroll") for (int q_ = 0; q_ < XL; ++q_)                                            \
            *reinterpret_cast<vec8_t<T> *>(xs + (buf) * kXTile + xs_wr[q_]) = xr[q_];                \
    }
#define SWL_MT_PROCESS(slot, buf)                                                                     \
    {                                                                                                \
        const T *xl_ = xs + (buf) * kXTile;                                                          \
        _Pragma("unroll") for (int kk_ = 0; kk_ < kKT / 16; ++kk_) {                                 \
            const int off_ = l32 * kKT + (((2 * kk_ + hf) ^ (l32 & 15)) << 3);                       \
            _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_) {                                   \
                const vec8_t<T> b_ = *reinterpret_cast<const vec8_t<T> *>(xl_ + mt_ * 32 * kKT + off_); \
                acc[mt_] = mfma32x32x16(wr[slot][kk_], b_, acc[mt_]);                                \
            }                                                                                        \
        }                                                                                            \
    }
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
        if (d < nkt) SWL_MT_ISSUE_W(d, d);
    SWL_MT_ISSUE_X(0);
    SWL_MT_STAGE_X(0);
    __syncthreads();
    int kt = 0;
    for (; kt + 2 * D - 1 <= nkt; kt += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            SWL_MT_ISSUE_X(kt + d + 1);
            SWL_MT_ISSUE_W((d + D - 1) % D, kt + d + D - 1);
            SWL_MT_PROCESS(d, (kt + d) & 1);
            SWL_MT_STAGE_X((kt + d + 1) & 1);
            __syncthreads();
        }
    }
    const int rem = nkt - kt;
#pragma unroll
    for (int t = 0; t < 2 * D - 2; ++t) {
        if (t < rem) {
            if (t + 1 < rem) SWL_MT_ISSUE_X(kt + t + 1);
            if (t + D - 1 < rem) SWL_MT_ISSUE_W((t + D - 1) % D, kt + t + D - 1);
            SWL_MT_PROCESS(t % D, (kt + t) & 1);
            if (t + 1 < rem) {
                SWL_MT_STAGE_X((kt + t + 1) & 1);
                __syncthreads();
            }
        }
    }
#undef SWL_MT_ISSUE_W
#undef SWL_MT_ISSUE_X
#undef SWL_MT_STAGE_X
#undef SWL_MT_PROCESS
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) mfma_results_tie(acc[mt]);
    mfma_results_ready<8>(acc[MT - 1]); // acc comes straight out of the K loop (swl_common.h)
    if constexpr (MODE == kGemmSiluGate) {
        // same rounding points as linear -> silu_and_mul_inplace: projection rounded to T, silu in fp32 rounded to
        // T, product in T. Exchange tile of (gate wave w, token block mt): 32 x 40 elements in the x buffers.
        __syncthreads(); // every wave is done reading the x tiles
        T *xch = xs + ((wave & 1) * MT) * 1280;
        if (is_gate) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float g = to_f(to_t<T>(acc[mt][r]));
                    xch[mt * 1280 + l32 * 40 + (r & 3) + 8 * (r >> 2) + 4 * hf] = to_t<T>(g / (1.0f + expf(-g)));
                }
        }
        __syncthreads();
        if (!is_gate && tile_ok) {
            typedef T vec4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = 32 * mt + l32;
                if (m >= M) continue;
                T *o = static_cast<T *>(out_) + static_cast<int64_t>(m) * out_stride + col0 + 4 * hf;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const vec4 a = *reinterpret_cast<const vec4 *>(xch + mt * 1280 + l32 * 40 + 8 * r4 + 4 * hf);
                    vec4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = mul_t<T>(to_t<T>(acc[mt][4 * r4 + e]), a[e]);
                    *reinterpret_cast<vec4 *>(o + 8 * r4) = v;
                }
            }
        }
        return;
    }
    if (!tile_ok) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = 32 * mt + l32;
        if (m >= M) continue;
        const int n = col0 + 4 * hf;
        if constexpr (PARTIAL) {
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

// Splits as for M <= 32 (>= 768 waves), capped so the slab traffic (2 * ks * M * N * 4 B) stays under a third of
// the weight bytes: ks <= K / (12 * M) (measured optimum at M = 48..128: o_proj 4, down_proj 8).
// Two blocks of 32 tokens per weight fragment: 33..64 tokens. (r01-r04 also built MT = 4 for 65..128 tokens: csrc/gemm_wide.hip
// serves those since r04 and the routing never reached it — it was the only kernel of the library that spilled. Removed in r05.)
constexpr int kMidMaxM = 64;

static int choose_k_splits_mt(int M, int N, int K) {
    const int tiles = N / 32;
    const int cap = K / (12 * M);
    int ks = 1;
    while (ks < 16 && ks * 2 <= cap && tiles * ks < 768 && K % (kKT * ks * 2) == 0 && K / (ks * 2) >= 2 * kKT) ks *= 2;
    return ks;
}

// reduce == false: stop after the partial slabs (ks >= 1) for a fused consumer
template <typename T>
static int run_gemm_packed_mt(T *out, const T *x, const T *wp, float *ws, size_t ws_bytes, int M, int N, int K,
                              int64_t xs, int64_t os, int ks, hipStream_t stream, bool reduce = true) {
    if (ks <= 0) ks = choose_k_splits_mt(M, N, K);
    if (K % (kKT * ks) != 0) return SWL_ERR_UNSUPPORTED;
    const dim3 grid((N / 32 + kGemmWaves - 1) / kGemmWaves, ks), block(kGemmWaves * 64);
    const int kc = K / ks;
    if (ks == 1 && reduce) {
        hipLaunchKernelGGL((gemm_packed_mt_kernel<T, 2, kGemmDirect>), grid, block, 0, stream, out, x, wp, M, N, K, kc, xs, os);
        return check_launch();
    }
    if (!ws || ws_bytes < static_cast<size_t>(ks) * M * N * sizeof(float)) return SWL_ERR_BAD_ARG;
    const int64_t n64 = N;
    hipLaunchKernelGGL((gemm_packed_mt_kernel<T, 2, kGemmPartial>), grid, block, 0, stream, ws, x, wp, M, N, K, kc, xs, n64);
    if (!reduce) return check_launch();
    const int64_t items = static_cast<int64_t>(M) * (N / 4);
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(static_cast<unsigned>((items + 255) / 256)), dim3(256), 0,
                       stream, out, ws, M, N, ks, os);
    return check_launch();
}

} // namespace swl

/* out[M, N] = x . W^T for 32 < M <= 64 tokens (any M <= 64 is valid) on a weight packed by swl_gemm_pack_weight.
 * N % 32 == 0, K % 128 == 0. workspace >= k_splits * M * N * 4 bytes when K is split (k_splits = 0: library's
 * choice; 16 * M * N * 4 bytes cover any). */
extern "C" int swl_gemm_packed_mid(void *out, const void *x, const void *w_packed, void *workspace,
                                   size_t workspace_bytes, int32_t M, int32_t N, int32_t K, int64_t x_row_stride,
                                   int64_t out_row_stride, int32_t k_splits, int32_t dtype, swl_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!out || !x || !w_packed) return SWL_ERR_BAD_ARG;
    if (M > swl::kMidMaxM || (N & 31) || (K & (swl::kKT - 1))) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || out_row_stride < N || (x_row_stride & 7) || (out_row_stride & 3)) return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(x) || !swl::aligned16(w_packed) || (reinterpret_cast<uintptr_t>(out) & 7u) ||
        (workspace && !swl::aligned16(workspace)))
        return SWL_ERR_BAD_ARG;
    if (k_splits < 0 || k_splits > 16 || (k_splits & (k_splits - 1))) return SWL_ERR_BAD_ARG;
    SWL_DISPATCH_DTYPE(dtype, T, {
        return swl::run_gemm_packed_mt<T>(static_cast<T *>(out), static_cast<const T *>(x),
                                          static_cast<const T *>(w_packed), static_cast<float *>(workspace),
                                          workspace_bytes, M, N, K, x_row_stride, out_row_stride, k_splits,
                                          static_cast<hipStream_t>(stream));
    });
}

/* The split count swl_gemm_packed_mid picks for k_splits = 0 (0 = shape unsupported). */
extern "C" int swl_gemm_packed_mid_choose_splits(int32_t M, int32_t N, int32_t K) {
    if (M <= 0 || M > swl::kMidMaxM || N <= 0 || K <= 0 || (N & 31) || (K & (swl::kKT - 1))) return 0;
    return swl::choose_k_splits_mt(M, N, K);
}

/* Partial slabs only (slabs[k_splits][M][N] fp32, k_splits >= 1) for the split-K consumers (swl_splitk_*,
 * swl_paged_attn_decode_qkv). */
extern "C" int swl_gemm_packed_mid_partial(float *slabs, size_t slabs_bytes, const void *x, const void *w_packed,
                                           int32_t M, int32_t N, int32_t K, int64_t x_row_stride, int32_t k_splits,
                                           int32_t dtype, swl_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!slabs || !x || !w_packed || k_splits < 1 || k_splits > 16 || (k_splits & (k_splits - 1)))
        return SWL_ERR_BAD_ARG;
    if (M > swl::kMidMaxM || (N & 31) || (K & (swl::kKT - 1))) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || (x_row_stride & 7) || !swl::aligned16(x) || !swl::aligned16(w_packed) ||
        !swl::aligned16(slabs))
        return SWL_ERR_BAD_ARG;
    SWL_DISPATCH_DTYPE(dtype, T, {
        return swl::run_gemm_packed_mt<T>(static_cast<T *>(nullptr), static_cast<const T *>(x),
                                          static_cast<const T *>(w_packed), slabs, slabs_bytes, M, N, K, x_row_stride, N,
                                          k_splits, static_cast<hipStream_t>(stream), false);
    });
}

/* out[M, I] = up * silu(gate) of x . [up ; gate]^T for 32 < M <= 64 tokens on a packed weight (the medium-batch
 * twin of swl_gemm_skinny_packed_silu_gate; same rounding points as linear + silu_and_mul). I % 32 == 0. */
extern "C" int swl_gemm_packed_mid_silu_gate(void *out, const void *x, const void *w_up_gate_packed, int32_t M,
                                             int32_t I, int32_t K, int64_t x_row_stride, int64_t out_row_stride,
                                             int32_t dtype, swl_stream_t stream) {
    if (M < 0 || I <= 0 || K <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!out || !x || !w_up_gate_packed) return SWL_ERR_BAD_ARG;
    if (M > swl::kMidMaxM || (I & 31) || (K & (swl::kKT - 1))) return SWL_ERR_UNSUPPORTED;
    if (x_row_stride < K || out_row_stride < I || (x_row_stride & 7) || (out_row_stride & 3)) return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(x) || !swl::aligned16(w_up_gate_packed) || (reinterpret_cast<uintptr_t>(out) & 7u))
        return SWL_ERR_BAD_ARG;
    const dim3 grid((I / 32 + 1) / 2, 1), block(swl::kGemmWaves * 64);
    hipStream_t s = static_cast<hipStream_t>(stream);
    SWL_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL((swl::gemm_packed_mt_kernel<T, 2, swl::kGemmSiluGate>), grid, block, 0, s, out,
                           static_cast<const T *>(x), static_cast<const T *>(w_up_gate_packed), M, I, K, K,
                           x_row_stride, out_row_stride);
    });
    return swl::check_launch();
}
