// gemm_tiny.hip — decode projections for very small batches (M <= 4 tokens) that build their own input from what the
// PREVIOUS kernel left behind (gfx950, packed W).
//
// At batch 1 a decode layer is eight launches of which three move next to nothing and still cost a kernel boundary plus
// a dependent round trip each: the two split-K consumers (swl_splitk_add_scale, reference rmsnorm.py:67-89 at
// transformer_layer.py:46,120: 4.8 us apiece on FOUR workgroups) and the flash-decoding merge (paged_attn.py:108-150:
// 4.6 us) — 14 % of the 99 us layer (profiles/r02i). With M <= 4 the activation row is so small that every workgroup of
// the NEXT projection can afford to rebuild it while its first weight tiles are in flight, keep it in LDS for the whole
// K loop (no x traffic, no barrier in the loop) and carry the sums of squares to where the deferred 1/rms is applied:
//   qkv':     down slabs (layer L-1) + residual -> qkv slabs + row_ssq + residual'      (transformer_layer.py:46-56)
//   o_proj':  flash-decoding partials            -> o_proj slabs                         (transformer_layer.py:117)
//   up/gate': o_proj slabs + residual'          -> up * silu(gate)   + residual         (transformer_layer.py:120-127)
// From slabs: sum the 8 slabs of the K-chunk, add the residual (workgroups with blockIdx.x == 0 also store it — into a
// SECOND buffer: the others are still reading the old one), apply the norm weight: swl_splitk_add_scale's arithmetic, then
// the MFMA order of swl_gemm_skinny_packed_partial / _silu_gate_rs — slabs and residual come out bit-identical; only the
// sums of squares are grouped by K-chunk instead of by 1024 columns (another fp32 order of the same numbers).
// From attention partials: the LSE-weighted merge of phase 2 (same weights, same order of the weighted sum; the weight
// SUM in a different fp32 order), rounded to the storage dtype as phase 2 stores it.
// What bounds the gain: every workgroup pulls the rows it rebuilds through its own L1 at ~55 GB/s
// (up/gate': 8 slabs x M x 16 KB): batch 1 gains 2 %, batch 2 is even, batch 4 loses — the layer uses this up to M = 2.
#include "swl_common.h"

namespace swl {

__device__ __forceinline__ float16_t tiny_mfma(vec8_t<f16> a, vec8_t<f16> b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float16_t tiny_mfma(vec8_t<bf16> a, vec8_t<bf16> b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

constexpr int kTinyKT = 128;      // k elements per packed tile (gemm_skinny.hip)
constexpr int kTinyWaves = 4;
constexpr int kTinyMaxM = 4;
constexpr int kTinyMaxKc = 4096;  // K-chunk resident in LDS: 4 rows x 4096 x 2 B = 32 KiB
constexpr int kTinyPitch = kTinyMaxKc + 8;

enum TinyMode { kTinyPartial = 0, kTinySiluGate = 1 };
enum TinySrc { kTinyFromSlabs = 0, kTinyFromAttnPartials = 1 };

struct TinyArgs {
    void *out;            // Partial: fp32 slabs [ks][M][N]; SiluGate: T [M][I]
    const void *wp;
    int M, N, K, kc;      // N = output columns (I in SiluGate mode)
    int64_t out_stride;
    const float *slabs_in; // [ks_in][M][K]
    int ks_in;
    const void *residual_in;
    void *residual_out;
    const void *norm_w;
    float eps;
    float *ssq_out;       // Partial: [gridDim.y][M]
    // kTinyFromAttnPartials: x[seq][head * D + d] = LSE-weighted merge of the flash-decoding partials (paged_attn.hip phase 2)
    const float *mid_o, *mid_lse;
    const int *seq_lens;
    int H, D, seq_block_size, num_seq_blocks;
};

template <typename T, int MODE, int SRC = kTinyFromSlabs>
__global__ __launch_bounds__(kTinyWaves * 64, 2) void gemm_tiny_kernel(TinyArgs a) {
    __shared__ __attribute__((aligned(16))) T xres[kTinyMaxM][kTinyPitch];
    __shared__ __attribute__((aligned(16))) T zero16[8];
    __shared__ float red[kTinyWaves][kTinyMaxM];
    __shared__ float rs_s[kTinyMaxM];
    typedef T vec4 __attribute__((ext_vector_type(4)));

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int M = a.M, N = a.N, K = a.K, kc = a.kc;
    const bool is_gate = MODE == kTinySiluGate && wave >= 2;
    const int col0 = MODE == kTinySiluGate ? (blockIdx.x * 2 + (wave & 1)) * 32 : (blockIdx.x * kTinyWaves + wave) * 32;
    const bool tile_ok = col0 < N;
    const int n0 = tile_ok ? col0 + (is_gate ? N : 0) : 0;
    const int ksplit = blockIdx.y;
    const int k_begin = ksplit * kc;
    const int nkt = kc / kTinyKT;
    const int l32 = lane & 31, hf = lane >> 5;

    // ---- the weight stream starts first: two tiles per wave on their way before anything else ----
    const T *wsrc = static_cast<const T *>(a.wp) + (static_cast<int64_t>(n0 / 32) * (K / 16) + k_begin / 16) * 512 + lane * 8;
    constexpr int D = 3;
    vec8_t<T> wr[D][8];
#define SWL_TINY_ISSUE(slot, t)                                                                          \
    { _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                                   \
          wr[slot][i_] = load8_nt(wsrc + (static_cast<int64_t>(t) * 8 + i_) * 512); }
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
        if (d < nkt) SWL_TINY_ISSUE(d, d);

    if (threadIdx.x < 8) zero16[threadIdx.x] = to_t<T>(0.f);
    float ssq[kTinyMaxM] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (SRC == kTinyFromAttnPartials) {
        // ---- this K-chunk of the attention output, merged from the flash-decoding partials (reference paged_attn.py:
        // 108-150; paged_attn_phase2_kernel's arithmetic up to the fp32 order of the weight sum): one item = 8 columns
        // of one (sequence, head); the log-sum-exps and the partial rows of up to 16 splits are requested together ----
        const int per_row = kc >> 3;
        for (int it = threadIdx.x; it < M * per_row; it += kTinyWaves * 64) {
            const int row = it / per_row;
            const int c8 = it - row * per_row;
            const int col = k_begin + 8 * c8;
            const int head = col / a.D, d = col - head * a.D;
            const int n = (a.seq_lens[row] + a.seq_block_size - 1) / a.seq_block_size;
            const int64_t base = (static_cast<int64_t>(row) * a.H + head) * a.num_seq_blocks;
            float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            float lsum = 0.f;
            if (n <= 16) {
                // the common case in ONE round trip: the partial rows do not depend on the log-sum-exps, only their
                // weights do — everything is requested before the first use
                float lse[16];
                float4_t va[16], vb[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int sidx = min(u, n - 1);
                    lse[u] = a.mid_lse[base + sidx];
                    const float *src = a.mid_o + (base + sidx) * a.D + d;
                    va[u] = *reinterpret_cast<const float4_t *>(src);
                    vb[u] = *reinterpret_cast<const float4_t *>(src + 4);
                }
                float mx = kNegBig;
#pragma unroll
                for (int u = 0; u < 16; ++u) mx = fmaxf(mx, lse[u]);
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const float wgt = u < n ? fast_exp2(lse[u] - mx) : 0.f;
                    lsum += wgt;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc8[e] = fmaf(wgt, va[u][e], acc8[e]);
                        acc8[4 + e] = fmaf(wgt, vb[u][e], acc8[4 + e]);
                    }
                }
            } else {
                float mx = kNegBig;
                for (int s0 = 0; s0 < n; s0 += 16) {
                    float t16[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) t16[u] = a.mid_lse[base + min(s0 + u, n - 1)];
#pragma unroll
                    for (int u = 0; u < 16; ++u) mx = fmaxf(mx, t16[u]);
                }
                for (int s0 = 0; s0 < n; s0 += 8) {
                    float lse[8];
                    float4_t va[8], vb[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int sidx = min(s0 + u, n - 1);
                        lse[u] = a.mid_lse[base + sidx];
                        const float *src = a.mid_o + (base + sidx) * a.D + d;
                        va[u] = *reinterpret_cast<const float4_t *>(src);
                        vb[u] = *reinterpret_cast<const float4_t *>(src + 4);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float wgt = s0 + u < n ? fast_exp2(lse[u] - mx) : 0.f;
                        lsum += wgt;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc8[e] = fmaf(wgt, va[u][e], acc8[e]);
                            acc8[4 + e] = fmaf(wgt, vb[u][e], acc8[4 + e]);
                        }
                    }
                }
            }
            vec8_t<T> sv;
#pragma unroll
            for (int j = 0; j < 8; ++j) sv[j] = to_t<T>(acc8[j] / lsum);
            *reinterpret_cast<vec8_t<T> *>(&xres[row][8 * c8]) = sv;
        }
    } else {
    // ---- rebuild this K-chunk of the activations: swl_splitk_add_scale's arithmetic (rmsnorm.hip) ----
    {
        const int per_row = kc >> 3; // items (8 columns) per row
        const int64_t slab_stride = static_cast<int64_t>(M) * K;
        const bool store_res = blockIdx.x == 0;
        // two items (8 columns each) per thread and round trip: all their loads are requested before the first add
        const int items = M * per_row;
        for (int it0 = threadIdx.x; it0 < items; it0 += 2 * kTinyWaves * 64) {
            int row[2], c8[2];
            int64_t off[2];
            vec8_t<T> wv[2], rv[2], xv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int it = min(it0 + u * kTinyWaves * 64, items - 1); // (a clamped duplicate is never stored)
                row[u] = it / per_row;
                c8[u] = it - row[u] * per_row;
                off[u] = static_cast<int64_t>(row[u]) * K + k_begin + 8 * c8[u];
                wv[u] = load8(static_cast<const T *>(a.norm_w) + k_begin + 8 * c8[u]);
                rv[u] = load8(static_cast<const T *>(a.residual_in) + off[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) xv[u] = load8_splitk<T>(a.slabs_in, a.ks_in, slab_stride, off[u]);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (it0 + u * kTinyWaves * 64 >= items) continue;
                vec8_t<T> sv;
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xv[u][j] = add_t<T>(xv[u][j], rv[u][j]); // rounded to T, as stored (rmsnorm.py:54-57)
                    const float v = to_f(xv[u][j]);
                    s = fmaf(v, v, s);
                    sv[j] = to_t<T>(v * to_f(wv[u][j]));
                }
                *reinterpret_cast<vec8_t<T> *>(&xres[row[u]][8 * c8[u]]) = sv;
                if (store_res) store8(static_cast<T *>(a.residual_out) + off[u], xv[u]);
#pragma unroll
                for (int r = 0; r < kTinyMaxM; ++r)
                    if (r == row[u]) ssq[r] += s;
            }
        }
#pragma unroll
        for (int r = 0; r < kTinyMaxM; ++r) {
            const float tot = wave_allreduce_sum(ssq[r]);
            if (lane == 0) red[wave][r] = tot;
        }
    }
    } // SRC
    __syncthreads();
    if (SRC == kTinyFromSlabs && threadIdx.x < kTinyMaxM) {
        const float tot = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if constexpr (MODE == kTinySiluGate) rs_s[threadIdx.x] = 1.0f / sqrtf(tot / static_cast<float>(K) + a.eps);
        else if (blockIdx.x == 0 && static_cast<int>(threadIdx.x) < M) a.ssq_out[ksplit * M + threadIdx.x] = tot;
    }

    // ---- K loop: W fragments from the ring, x fragments from the resident rows (lanes of tokens >= M read zeros) ----
    const T *xb = l32 < M ? &xres[l32][8 * hf] : &zero16[0];
    const int xstep = l32 < M ? 1 : 0;
    float16_t acc = float16_t{};
#define SWL_TINY_PROCESS(slot, t)                                                                        \
    { _Pragma("unroll") for (int kk_ = 0; kk_ < kTinyKT / 16; ++kk_) {                                   \
          const vec8_t<T> b_ = *reinterpret_cast<const vec8_t<T> *>(xb + xstep * ((t) * kTinyKT + 16 * kk_)); \
          acc = tiny_mfma(wr[slot][kk_], b_, acc);                                                       \
      } }
    int kt = 0;
    for (; kt + 2 * D - 1 <= nkt; kt += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            SWL_TINY_ISSUE((d + D - 1) % D, kt + d + D - 1);
            SWL_TINY_PROCESS(d, kt + d);
        }
    }
    const int rem = nkt - kt;
#pragma unroll
    for (int t = 0; t < 2 * D - 2; ++t) {
        if (t < rem) {
            if (t + D - 1 < rem) SWL_TINY_ISSUE((t + D - 1) % D, kt + t + D - 1);
            SWL_TINY_PROCESS(t % D, kt + t);
        }
    }
#undef SWL_TINY_ISSUE
#undef SWL_TINY_PROCESS
    mfma_results_ready<8>(acc);

    // ---- epilogue: acc[r] = out^T[n = n0 + (r&3) + 8*(r>>2) + 4*hf][m = l32] ----
    if constexpr (MODE == kTinyPartial) {
        if (tile_ok && l32 < M) {
            float *slab = static_cast<float *>(a.out) + (static_cast<int64_t>(ksplit) * M + l32) * N + n0 + 4 * hf;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<float4_t *>(slab + 8 * r4) =
                    float4_t{acc[4 * r4], acc[4 * r4 + 1], acc[4 * r4 + 2], acc[4 * r4 + 3]};
        }
    } else {
        // gemm_skinny.hip's SiLU-gate epilogue: projection rounded to T (after the deferred 1/rms, in fp32), silu in
        // fp32 rounded to T, product in T (reference silu_and_mul.py:16-23)
        __syncthreads(); // rs_s written; every wave is done reading xres, whose first bytes become the exchange tiles
        const float rs = rs_s[min(l32, M - 1)];
        T *xch = &xres[0][0];
        T *mine = xch + wave * (32 * 40);
        if (is_gate) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float g = to_f(to_t<T>(acc[r] * rs));
                mine[l32 * 40 + (r & 3) + 8 * (r >> 2) + 4 * hf] = to_t<T>(g / (1.0f + expf(-g)));
            }
        }
        __syncthreads();
        if (!is_gate && tile_ok && l32 < M) {
            const T *act = xch + (wave + 2) * (32 * 40);
            T *o = static_cast<T *>(a.out) + static_cast<int64_t>(l32) * a.out_stride + col0 + 4 * hf;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const vec4 g4 = *reinterpret_cast<const vec4 *>(act + l32 * 40 + 8 * r4 + 4 * hf);
                vec4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = mul_t<T>(to_t<T>(acc[4 * r4 + e] * rs), g4[e]);
                *reinterpret_cast<vec4 *>(o + 8 * r4) = v;
            }
        }
    }
}

static bool tiny_common_ok(int M, int N, int K, int kc) {
    return M > 0 && M <= kTinyMaxM && N > 0 && (N & 31) == 0 && K > 0 && kc > 0 && (kc % kTinyKT) == 0 &&
           kc <= kTinyMaxKc && K % kc == 0;
}

} // namespace swl

extern "C" int swl_gemm_tiny_max_tokens(void) { return swl::kTinyMaxM; }

extern "C" int swl_gemm_tiny_partial_from_splitk(float *slabs_out, size_t slabs_out_bytes, int32_t k_splits_out,
                                                 float *ssq_out, const float *slabs_in, int32_t k_splits_in,
                                                 const void *residual_in, void *residual_out, const void *norm_w,
                                                 const void *w_packed, int32_t M, int32_t N, int32_t K, int32_t dtype,
                                                 swl_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0 || k_splits_out <= 0 || k_splits_in <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!slabs_out || !ssq_out || !slabs_in || !residual_in || !residual_out || !norm_w || !w_packed ||
        residual_in == residual_out || slabs_in == slabs_out)
        return SWL_ERR_BAD_ARG;
    if (K % k_splits_out != 0 || !swl::tiny_common_ok(M, N, K, K / k_splits_out)) return SWL_ERR_UNSUPPORTED;
    if (slabs_out_bytes < static_cast<size_t>(k_splits_out) * M * N * sizeof(float)) return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(slabs_out) || !swl::aligned16(slabs_in) || !swl::aligned16(residual_in) ||
        !swl::aligned16(residual_out) || !swl::aligned16(norm_w) || !swl::aligned16(w_packed))
        return SWL_ERR_BAD_ARG;
    swl::TinyArgs a = {};
    a.out = slabs_out; a.wp = w_packed;
    a.M = M; a.N = N; a.K = K; a.kc = K / k_splits_out; a.out_stride = N;
    a.slabs_in = slabs_in; a.ks_in = k_splits_in;
    a.residual_in = residual_in; a.residual_out = residual_out; a.norm_w = norm_w;
    a.ssq_out = ssq_out;
    const dim3 grid((N / 32 + swl::kTinyWaves - 1) / swl::kTinyWaves, k_splits_out);
    SWL_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL((swl::gemm_tiny_kernel<T, swl::kTinyPartial>), grid, dim3(swl::kTinyWaves * 64), 0,
                           static_cast<hipStream_t>(stream), a);
    });
    return swl::check_launch();
}

extern "C" int swl_gemm_tiny_partial_from_attn(float *slabs_out, size_t slabs_out_bytes, int32_t k_splits_out,
                                               const float *attn_scratch, const int32_t *seq_lens,
                                               int32_t num_q_heads, int32_t head_dim, int32_t seq_block_size,
                                               int32_t num_seq_blocks, const void *w_packed, int32_t M, int32_t N,
                                               int32_t dtype, swl_stream_t stream) {
    if (M < 0 || N <= 0 || k_splits_out <= 0 || num_q_heads <= 0 || head_dim <= 0 || seq_block_size <= 0 ||
        num_seq_blocks <= 0)
        return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!slabs_out || !attn_scratch || !seq_lens || !w_packed) return SWL_ERR_BAD_ARG;
    const int K = num_q_heads * head_dim;
    if ((head_dim & 7) || K % k_splits_out != 0 || !swl::tiny_common_ok(M, N, K, K / k_splits_out))
        return SWL_ERR_UNSUPPORTED;
    if (slabs_out_bytes < static_cast<size_t>(k_splits_out) * M * N * sizeof(float)) return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(slabs_out) || !swl::aligned16(attn_scratch) || !swl::aligned16(w_packed)) return SWL_ERR_BAD_ARG;
    swl::TinyArgs a = {};
    a.out = slabs_out; a.wp = w_packed;
    a.M = M; a.N = N; a.K = K; a.kc = K / k_splits_out; a.out_stride = N;
    a.mid_o = attn_scratch;
    a.mid_lse = attn_scratch + static_cast<size_t>(M) * num_q_heads * num_seq_blocks * head_dim;
    a.seq_lens = seq_lens;
    a.H = num_q_heads; a.D = head_dim; a.seq_block_size = seq_block_size; a.num_seq_blocks = num_seq_blocks;
    const dim3 grid((N / 32 + swl::kTinyWaves - 1) / swl::kTinyWaves, k_splits_out);
    SWL_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL((swl::gemm_tiny_kernel<T, swl::kTinyPartial, swl::kTinyFromAttnPartials>), grid,
                           dim3(swl::kTinyWaves * 64), 0, static_cast<hipStream_t>(stream), a);
    });
    return swl::check_launch();
}

extern "C" int swl_gemm_tiny_silu_gate_from_splitk(void *out, const float *slabs_in, int32_t k_splits_in,
                                                   const void *residual_in, void *residual_out, const void *norm_w,
                                                   float eps, const void *w_up_gate_packed, int32_t M, int32_t I,
                                                   int32_t K, int64_t out_row_stride, int32_t dtype,
                                                   swl_stream_t stream) {
    if (M < 0 || I <= 0 || K <= 0 || k_splits_in <= 0) return SWL_ERR_BAD_ARG;
    if (M == 0) return SWL_OK;
    if (!out || !slabs_in || !residual_in || !residual_out || !norm_w || !w_up_gate_packed ||
        residual_in == residual_out)
        return SWL_ERR_BAD_ARG;
    if ((I & 63) || !swl::tiny_common_ok(M, I, K, K)) return SWL_ERR_UNSUPPORTED;
    if (out_row_stride < I || (out_row_stride & 3)) return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(slabs_in) || !swl::aligned16(residual_in) || !swl::aligned16(residual_out) ||
        !swl::aligned16(norm_w) || !swl::aligned16(w_up_gate_packed) || (reinterpret_cast<uintptr_t>(out) & 7u))
        return SWL_ERR_BAD_ARG;
    swl::TinyArgs a = {};
    a.out = out; a.wp = w_up_gate_packed;
    a.M = M; a.N = I; a.K = K; a.kc = K; a.out_stride = out_row_stride;
    a.slabs_in = slabs_in; a.ks_in = k_splits_in;
    a.residual_in = residual_in; a.residual_out = residual_out; a.norm_w = norm_w; a.eps = eps;
    const dim3 grid(I / 64, 1);
    SWL_DISPATCH_DTYPE(dtype, T, {
        hipLaunchKernelGGL((swl::gemm_tiny_kernel<T, swl::kTinySiluGate>), grid, dim3(swl::kTinyWaves * 64), 0,
                           static_cast<hipStream_t>(stream), a);
    });
    return swl::check_launch();
}
