// paged_attn.hip — decode-stage paged attention ("paged attention v2" / flash-decoding) for gfx950.
//
// Replaces _fwd_paged_attention_phase1 (swiftllm/worker/kernels/paged_attn.py:9-108) and
// _fwd_paged_attention_phase2 (paged_attn.py:111-149). The north-star kernel: HBM-bound,
// algorithmic bytes per call = sum_i len_i * 2 * KVH * D * e  (every KV byte ONCE per kv-head)
// + q/o (2*Bd*H*D*e) + partials.
//
// Design (MI355X-first, not the reference's one-warp-per-q-head mapping):
//   * grid (seq-block, kv-head, sequence); a workgroup = 4 or 8 waves serves ALL G = H/KVH q-heads of
//     its kv-head, so a KV tile is fetched once instead of G times (the reference leans on L2 for
//     that). The host sizes the sequence block so a launch is ~one 8-wave workgroup per CU in whole
//     rounds (batch_plan.select_seq_block_size); with one sequence block the output is written
//     directly and phase 2 is not launched at all;
//   * the 16-token x D tile of one (block, layer, kv-head) is 4 KiB contiguous in the pool; a wave
//     reads it with fully coalesced 16-byte-per-lane loads (lane -> token = i*TPI + lane/LPT,
//     8-element chunk = lane%LPT), 1 KiB per instruction, non-temporal (read once);
//   * waves stride over the 16-token blocks of the sequence block; the next block's K and V (8 KiB
//     per wave) are in flight while the current one is consumed (register double buffering);
//   * q.k partial dots with v_dot2c_f32_{f16,bf16} (fp32 accumulate), reduced across the LPT lanes
//     of a token with DPP adds — no LDS, no ds_bpermute in the main loop;
//   * every LPT-lane row keeps its own online-softmax state (max, sum, 8-wide slice of the
//     accumulator for each of the G heads) so nothing crosses rows until the end; rows are merged
//     once per workgroup with wave shuffles, waves once through 8 KiB of LDS;
//   * block-table entries are scalar loads (the wave index is made provably uniform);
//   * scores are kept UNscaled; scale*log2(e) is folded into the exp2 argument with one fma.
// Numerics: fp32 scores/softmax/accumulation (the reference rounds scores to fp16,
// paged_attn.py:72-73; ours is closer to the exact value). Partials use the reference's format:
// mid_o = acc/sum (normalised), mid_lse = log2(sum) + max in the scaled base-2 domain
// (paged_attn.py:106-108), so phase 1 can be compared with the reference's phase 1 directly.
#include "swl_common.h"
#include "attend_block.h"

namespace swl {

template <int V>
struct IntTag {
    static constexpr int value = V;
};

struct PagedAttnParams {
    void *o_direct;
    const void *q;
    const void *k_cache;
    const void *v_cache;
    const int *block_table;
    const int *seq_ids;
    const int *seq_lens;
    float *mid_o;
    float *mid_lse;
    float scale_log2e;
    int H, KVH, L, layer, max_blocks_per_seq, seq_block_size, num_seq_blocks;
    int64_t q_tok_stride, o_tok_stride;
    // QKV variant: q/k/v of the new token are still the split-K partial slabs of the fused qkv projection
    const float *qkv_slabs; // [ks][Bd][(H + 2*KVH) * D] fp32
    int ks;
    const void *cos_t, *sin_t; // rope tables [positions][D/2]
    const int *pos_idx;        // table row per sequence (NULL: seq_len - 1)
    // deferred RMSNorm (rmsnorm.hip, splitk_add_scale_kernel): the qkv projection ran on un-normalised activations;
    // its fp32 slab sums are multiplied by rstd[seq] = 1/sqrt(sum_p row_ssq[p][seq] / hidden + eps) before their rounding
    const float *row_ssq; // [ssq_parts][Bd], NULL = the slabs are final
    int ssq_parts, hidden;
    float eps;
};

// ---- matrix-core variant of attend_block (G >= 2) --------------------------------------------------------------------
// With G query heads per kv head the VALU version above does G x (dot products + 16-lane reductions + 8-wide FMAs) per
// 16-byte K/V fragment: measured on MI355X the arithmetic costs 22-28 % of the kernel at G = 4 (batch 32 x 1k context:
// 32.0 us, 25.0 us with the arithmetic compiled out; G = 1: 2 % — profiles/r02f_paged_attn_nomath.md). Here the G heads
// become the N dimension of 16 x 16 MFMA tiles (columns >= G are zero padding) and a block's 16 tokens the M / K one:
//   S[token][head]  = K_blk . Q^T   D/32 x v_mfma_f32_16x16x32  (A = K rows from LDS, B = Q^T in registers all kernel long)
//   O^T[d][head]   += V_blk^T . P^T D/16 x v_mfma_f32_16x16x16  (A = V^T via ds_read_b64_tr_b16, B = P^T = the S registers)
// The K/V registers arrive in the coalesced layout of the ring (lane -> token row, 16-byte chunk); one wave-private LDS
// tile turns them into A fragments (in-order LDS pipeline of one wave: no barrier). In the 16 x 16 C layout lane
// (q = l/16, h = l%16) holds tokens 4q..4q+3 of head h: the scores a lane gets from QK^T are exactly the B fragment PV
// needs from it, the online-softmax state is ONE (m, l) pair per lane, and O^T costs D/16 x 4 registers for ANY G
// (the VALU version: 8 G). P is fed as hi + lo 16-bit halves (two MFMAs): the product keeps fp32-level accuracy
// instead of the storage dtype's, so the numerics stay those of the VALU version (and of the reference's fp32 p,
// paged_attn.py:74-79) at 16 more MFMA issues per block.
typedef short short4_t __attribute__((ext_vector_type(4)));
template <typename T>
struct Vec4 {
    typedef T type __attribute__((ext_vector_type(4)));
};

__device__ __forceinline__ float4_t mfma16x32(vec8_t<f16> a, vec8_t<f16> b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float4_t mfma16x32(vec8_t<bf16> a, vec8_t<bf16> b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float4_t mfma16x16(short4_t a, typename Vec4<f16>::type b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(typename Vec4<f16>::type, a), b, c, 0, 0, 0);
}
__device__ __forceinline__ float4_t mfma16x16(short4_t a, typename Vec4<bf16>::type b, float4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4_t, b), c, 0, 0, 0);
}
// LDS transpose read (gfx950): the 16 lanes of a group each give the address of 4 consecutive 16-bit elements (lanes
// 4r..4r+3 = the four quarters of row r); lane i receives column i of that 4 x 16 block: {row0[i], .., row3[i]}.
template <typename T>
__device__ __forceinline__ short4_t lds_tr16_b64(const T *p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3))) *)(p));
}

// All-reduce over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48), VALU only:
// v_permlane16_swap(a, a) -> {rows 0,0,2,2 | rows 1,1,3,3}, v_permlane32_swap(b, b) -> {lo, lo | hi, hi}.
__device__ __forceinline__ float rows_allreduce_max(float v) {
    const auto r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r1[0]), __uint_as_float(r1[1]));
    const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r2[0]), __uint_as_float(r2[1]));
}
__device__ __forceinline__ float rows_allreduce_sum(float v) {
    const auto r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r1[0]) + __uint_as_float(r1[1]);
    const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
}

template <typename T, int D>
struct MfmaTile {
    static constexpr int KRS = D + 8;    // K row pitch (elements): 16 rows -> 16 distinct 16-byte slots for ds_read_b128
    static constexpr int VRS = D + 16;   // V row pitch: 8 rows x 32 B tile the 64 banks exactly for the b64 transpose read
    static constexpr int ELEMS = 16 * VRS;
    static constexpr int QS = D / 32;    // QK^T MFMAs per block
    static constexpr int OS = D / 16;    // PV MFMA pairs per block
};

template <typename T, int D, int G>
__device__ __forceinline__ void attend_block_mfma(const vec8_t<T> (&qb)[MfmaTile<T, D>::QS],
                                                  const vec8_t<T> (&Kv)[DecodeTile<T, D, G>::NI],
                                                  const vec8_t<T> (&Vv)[DecodeTile<T, D, G>::NI], float &m, float &l,
                                                  float4_t (&acc)[MfmaTile<T, D>::OS], T *stage, float c, int tok0,
                                                  int row, int chunk, int lane, int len, bool partial) {
    using Tile = DecodeTile<T, D, G>;
    using MT = MfmaTile<T, D>;
    constexpr int NI = Tile::NI;
    const int q = lane >> 4, i16 = lane & 15;
    // K: ring layout -> row-major tile -> A fragments (row = token i16, k = d)
#pragma unroll
    for (int i = 0; i < NI; ++i)
        *reinterpret_cast<vec8_t<T> *>(stage + (i * Tile::TPI + row) * MT::KRS + chunk * 8) = Kv[i];
    float4_t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MT::QS; ++j) {
        const vec8_t<T> kf = *reinterpret_cast<const vec8_t<T> *>(stage + i16 * MT::KRS + 32 * j + 8 * q);
        s = mfma16x32(kf, qb[j], s);
    }
    mfma_results_ready<4>(s); // the scores are read by VALU next, behind a branch (swl_common.h)
    // V goes into the same tile once the K fragments are out (same wave: LDS executes in order)
#pragma unroll
    for (int i = 0; i < NI; ++i)
        *reinterpret_cast<vec8_t<T> *>(stage + (i * Tile::TPI + row) * MT::VRS + chunk * 8) = Vv[i];
    // s[r] = score of token tok0 + 4q + r for head i16
    if (partial) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (tok0 + 4 * q + r >= len) s[r] = kNegBig;
    }
    // block maximum of head i16 over its 16 tokens = over the four lanes q = 0..3: v_permlane16_swap / v_permlane32_swap
    // (VALU only; 1.2x cheaper than two ds_bpermute round trips through the LDS pipe this loop keeps busy)
    float mb = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
    mb = rows_allreduce_max(mb);
    const float m_new = fmaxf(m, mb);
    const float alpha = fast_exp2((m - m_new) * c); // difference first (see attend_block)
    const float mc = m_new * c;
    float pf[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) pf[r] = fast_exp2(fmaf(s[r], c, -mc));
    if (partial) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (tok0 + 4 * q + r >= len) pf[r] = 0.f;
    }
    l = fmaf(l, alpha, (pf[0] + pf[1]) + (pf[2] + pf[3]));
    m = m_new;
    typename Vec4<T>::type ph, pl;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        ph[r] = to_t<T>(pf[r]);
        pl[r] = to_t<T>(pf[r] - to_f(ph[r]));
    }
    // rescale only when some head of this wave raised its maximum (wave-uniform branch; alpha == 1 is the common case
    // after the first blocks of a sequence)
    if (!__all(alpha == 1.0f)) {
#pragma unroll
        for (int mm = 0; mm < MT::OS; ++mm) acc[mm] *= alpha;
    }
#pragma unroll
    for (int mm = 0; mm < MT::OS; ++mm) {
        const short4_t vf = lds_tr16_b64(stage + (4 * q + (i16 >> 2)) * MT::VRS + 16 * mm + 4 * (i16 & 3));
        acc[mm] = mfma16x16(vf, ph, acc[mm]);
        acc[mm] = mfma16x16(vf, pl, acc[mm]);
    }
}

// Ring depth of the K/V register pipeline: a wave keeps kPaDepth 16-token blocks (K + V = 8 KiB, 32 VGPRs each)
// resident — the one it attends plus kPaDepth-1 in flight. Little's law on this part: one CU needs ~31 GB/s
// (8 TB/s / 256) against ~2 us of loaded HBM latency = ~64 KB in flight; 8 waves x 1 block in flight (depth 2) is
// exactly that with nothing to spare, depth 3 doubles it. (r02 A/B builds of deeper rings and of an L2 look-ahead touch
// lost or tied — profiles/r02g_paged_attn_mfma.md; the knobs are gone, the numbers stay there.)
constexpr int kPaMfmaDepth = 2;   // matrix-core path (G >= 2): measured best on MI355X (profiles/r02g_paged_attn_mfma.md)

// NW = waves per workgroup: 4 for short sequence blocks (latency-bound launches that want many
// small workgroups), 8 for long ones (one workgroup per CU, every wave streams many KV blocks and the
// per-workgroup prologue/merge is amortised; 8 waves x 16 KiB of K/V in flight per CU).
//
// QKV = true: the rotary + KV-store step of the layer (rotary_emb.py + kvcache_mgmt.py:50-79) runs in this
// kernel's prologue instead of a launch of its own. The workgroup sums the fused-qkv slabs of its G query
// heads and its kv head, rotates q and k (same arithmetic and rounding as rotary.hip), keeps the rounded q in
// LDS for its waves, and — in the split that owns the last position — writes the new k/v into the pool and
// patches them into the registers of the wave that attends that block (the pool read raced with the write).
template <typename T, int D, int G, int NW, bool QKV = false>
// block_table / seq_lens / seq_ids are passed a second time as __restrict__ kernel arguments: the QKV variant
// stores into the pools before its main loop, and without the no-alias guarantee the compiler must assume those
// stores clobber the block table — it then fetches bt[b] with a VECTOR load and waits vmcnt(0) for it, draining
// the whole K/V ring on every refill (seen in the ISA; the plain variant has no stores and got s_load all along).
__global__ __launch_bounds__(NW * 64) void paged_attn_phase1_kernel(PagedAttnParams p,
                                                                    const int *__restrict__ block_table,
                                                                    const int *__restrict__ seq_lens_r,
                                                                    const int *__restrict__ seq_ids_r) {
    using Tile = DecodeTile<T, D, G>;
    constexpr int LPT = Tile::LPT, TPI = Tile::TPI, NI = Tile::NI;
    constexpr int NT = NW * 64;
    __shared__ float sm_ml[NW][G][2];
    __shared__ float sm_acc[NW][G][D];
    __shared__ __attribute__((aligned(16))) T sm_q[QKV ? G * D : 8];
    __shared__ __attribute__((aligned(16))) T sm_kv[QKV ? 2 * D : 8];
    constexpr bool MF = G >= 2;          // matrix-core attend_block (see attend_block_mfma)
    using MT = MfmaTile<T, D>;
    __shared__ __attribute__((aligned(16))) T sm_stage[MF ? NW : 1][MF ? MT::ELEMS : 8];

    const int split = blockIdx.x;
    const int kvh = blockIdx.y;
    const int seq = blockIdx.z;
    const int len = seq_lens_r[seq];
    const int tok_begin = split * p.seq_block_size;
    if (tok_begin >= len) return; // uniform for the workgroup, before any barrier
    const int tok_end = min(len, tok_begin + p.seq_block_size);
    const int blk_end = (tok_end + kBlk - 1) / kBlk;
    const int seq_id = seq_ids_r[seq];
    const int *__restrict__ bt = block_table + static_cast<int64_t>(seq_id) * p.max_blocks_per_seq;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunk = lane % LPT;
    const int row = lane / LPT;
    const float c = p.scale_log2e;

    const T *kc = static_cast<const T *>(p.k_cache);
    const T *vc = static_cast<const T *>(p.v_cache);
    const int64_t tile_elems = static_cast<int64_t>(kBlk) * D;
    const int64_t layer_head = static_cast<int64_t>(p.layer) * p.KVH + kvh;
    const int64_t blk_pitch = static_cast<int64_t>(p.L) * p.KVH;

    vec8_t<T> qv[MF ? 1 : G];
    vec8_t<T> qb[MT::QS];               // MF: Q^T B fragments, lane (q, h) -> Q[head h][32 j + 8 q ..], zero for h >= G
    const int mq = lane >> 4, mh = lane & 15;
    if constexpr (!QKV) {
        if constexpr (MF) {
            const T *qp = static_cast<const T *>(p.q) + seq * p.q_tok_stride +
                          (static_cast<int64_t>(kvh) * G + min(mh, G - 1)) * D + 8 * mq;
#pragma unroll
            for (int j = 0; j < MT::QS; ++j) {
                qb[j] = load8(qp + 32 * j);
                if (mh >= G) qb[j] = vec8_t<T>{};
            }
        } else {
            const T *qp = static_cast<const T *>(p.q) + seq * p.q_tok_stride +
                          static_cast<int64_t>(kvh) * G * D + chunk * 8;
#pragma unroll
            for (int g = 0; g < G; ++g) qv[g] = load8(qp + g * D);
        }
    }
    const int pos = len - 1;          // the token being decoded
    const int last_blk = pos / kBlk;

    float m[MF ? 1 : G], l[MF ? 1 : G], acc[MF ? 1 : G][8];
    float4_t acc4[MT::OS];              // MF: O^T[d = 16 mm + 4 q + r][head h]

    // ring slots: what the 256-VGPR budget of a 2-waves-per-SIMD kernel holds without spilling: 4 for the VALU path
    // (G = 1), kPaMfmaDepth for the matrix-core path (D/16 x 4 accumulator registers whatever G is)
    constexpr int ND = G == 1 ? 4 : kPaMfmaDepth;
    vec8_t<T> Kr[ND][NI], Vr[ND][NI];
    auto load_phys = [&](int64_t phys, vec8_t<T>(&Kd)[NI], vec8_t<T>(&Vd)[NI]) {
        const int64_t base = (phys * blk_pitch + layer_head) * tile_elems + lane * 8;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            Kd[i] = load8_nt(kc + base + i * 512);
            Vd[i] = load8_nt(vc + base + i * 512);
        }
    };
    auto load_block = [&](int b, vec8_t<T>(&Kd)[NI], vec8_t<T>(&Vd)[NI]) {
        load_phys(bt[b], Kd, Vd); // scalar load: b is wave-uniform
    };
    auto attend = [&](int b, vec8_t<T>(&Kd)[NI], vec8_t<T>(&Vd)[NI]) {
        const int tok0 = b * kBlk;
        if constexpr (MF)
            attend_block_mfma<T, D, G>(qb, Kd, Vd, m[0], l[0], acc4, &sm_stage[wave][0], c, tok0, row, chunk, lane, len,
                                       tok0 + kBlk > len);
        else
            attend_block<T, D, G>(qv, Kd, Vd, m, l, acc, c, tok0, row, len, tok0 + kBlk > len);
    };
    // Drain steps only (a wave's last block is always attended there): in the QKV variant the block that holds the
    // token being decoded takes that token's k/v from the prologue's LDS copy — its pool read raced with the store.
    // Kept out of the steady loop: the per-lane patch costs registers the ring needs.
    auto attend_tail = [&](int b, vec8_t<T>(&Kd)[NI], vec8_t<T>(&Vd)[NI]) {
        if constexpr (QKV) {
            const int tok0 = b * kBlk;
            if (b == last_blk) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    if (tok0 + i * TPI + row == pos) {
                        Kd[i] = *reinterpret_cast<const vec8_t<T> *>(&sm_kv[chunk * 8]);
                        Vd[i] = *reinterpret_cast<const vec8_t<T> *>(&sm_kv[D + chunk * 8]);
                    }
            }
        }
        attend(b, Kd, Vd);
    };
    // the first ND blocks of this wave go into slots 0..ND-1 (QKV: issued from inside the prologue)
    auto prefetch_kv = [&](int b0, auto lo_tag, auto hi_tag) {
#pragma unroll
        for (int d = decltype(lo_tag)::value; d < decltype(hi_tag)::value; ++d)
            if (b0 + d * NW < blk_end) load_block(b0 + d * NW, Kr[d], Vr[d]);
    };
    constexpr int NDP = ND < 2 ? ND : 2;    // QKV: slots requested while the slab loads of the prologue are pending

    int b = tok_begin / kBlk + wave;
    if constexpr (!QKV) {
        prefetch_kv(b, IntTag<0>{}, IntTag<ND>{});
    } else {
        // Prologue order matters (loads return in order within a wave): a thread first REQUESTS the slabs of its
        // item, then its wave requests two KV blocks (16 KiB, the stream is running), and only then the item is
        // finished — the slab data never queues behind 16 KiB of KV, and no wave delays its KV requests.
        constexpr int kRot = D / 16;       // rotation items per head
        const bool owner = tok_end == len; // the split that attends (and stores) the new token
        const int n_items = G * kRot + (owner ? kRot + D / 8 : 0); // <= one per thread
        const int64_t qkv_row = static_cast<int64_t>(p.H + 2 * p.KVH) * D;
        const int64_t slab_stride = static_cast<int64_t>(gridDim.z) * qkv_row;
        const int64_t row_off = seq * qkv_row;
        const int item = threadIdx.x;
        const bool has = item < n_items;
        const bool is_rot = item < (G + 1) * kRot;
        const bool is_q = item < G * kRot;
        const int c = is_rot ? item % kRot : item - (G + 1) * kRot;
        const int head = is_q ? kvh * G + item / kRot : (is_rot ? p.H + kvh : p.H + p.KVH + kvh);
        const int64_t off = row_off + static_cast<int64_t>(head) * D + c * 8;
        auto finish = [&](vec8_t<T> x0, vec8_t<T> x1, const vec8_t<T> &cv, const vec8_t<T> &sv) {
            if (is_rot) {
                rotate8<T>(x0, x1, cv, sv);
                T *dst = is_q ? &sm_q[(item / kRot) * D] : &sm_kv[0];
                *reinterpret_cast<vec8_t<T> *>(dst + c * 8) = x0;
                *reinterpret_cast<vec8_t<T> *>(dst + D / 2 + c * 8) = x1;
                if (!is_q) {
                    T *pool = const_cast<T *>(kc) + (static_cast<int64_t>(bt[last_blk]) * blk_pitch + layer_head) *
                                                        tile_elems + (pos % kBlk) * D;
                    store8(pool + c * 8, x0);
                    store8(pool + D / 2 + c * 8, x1);
                }
            } else {
                *reinterpret_cast<vec8_t<T> *>(&sm_kv[D + c * 8]) = x0;
                T *pool = const_cast<T *>(vc) + (static_cast<int64_t>(bt[last_blk]) * blk_pitch + layer_head) *
                                                    tile_elems + (pos % kBlk) * D;
                store8(pool + c * 8, x0);
            }
        };
        auto prologue = [&](auto ks_tag) {
            constexpr int KS = decltype(ks_tag)::value;
            float4_t a0[KS], b0[KS], a1[KS], b1[KS];
            vec8_t<T> cv = {}, sv = {};
            float ssv[8];
            const bool row_scaled = p.row_ssq != nullptr; // (uniform; same memory round trip as the slabs)
            if (row_scaled) {
#pragma unroll
                for (int q2 = 0; q2 < 8; ++q2)
                    ssv[q2] = q2 < p.ssq_parts ? p.row_ssq[q2 * static_cast<int>(gridDim.z) + seq] : 0.f;
            }
            if (has) {
                const float *s0 = p.qkv_slabs + off;
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    a0[k] = *reinterpret_cast<const float4_t *>(s0 + k * slab_stride);
                    b0[k] = *reinterpret_cast<const float4_t *>(s0 + k * slab_stride + 4);
                }
                if (is_rot) {
                    const int64_t trow = p.pos_idx ? p.pos_idx[seq] : pos;
                    cv = load8(static_cast<const T *>(p.cos_t) + trow * (D / 2) + c * 8);
                    sv = load8(static_cast<const T *>(p.sin_t) + trow * (D / 2) + c * 8);
#pragma unroll
                    for (int k = 0; k < KS; ++k) {
                        a1[k] = *reinterpret_cast<const float4_t *>(s0 + D / 2 + k * slab_stride);
                        b1[k] = *reinterpret_cast<const float4_t *>(s0 + D / 2 + k * slab_stride + 4);
                    }
                }
            }
            prefetch_kv(b, IntTag<0>{}, IntTag<NDP>{});
            if (has) {
                // slab order, one rounding: the bits of load8_splitk / the stand-alone reduce kernel
                float4_t sa0 = {0.f, 0.f, 0.f, 0.f}, sb0 = sa0, sa1 = sa0, sb1 = sa0;
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    sa0 += a0[k];
                    sb0 += b0[k];
                }
                if (is_rot) {
#pragma unroll
                    for (int k = 0; k < KS; ++k) {
                        sa1 += a1[k];
                        sb1 += b1[k];
                    }
                }
                float rs = 1.0f; // (x * 1.0f is exact: one code path)
                if (row_scaled) {
                    const float ss = ((ssv[0] + ssv[1]) + (ssv[2] + ssv[3])) + ((ssv[4] + ssv[5]) + (ssv[6] + ssv[7]));
                    rs = 1.0f / sqrtf(ss / static_cast<float>(p.hidden) + p.eps);
                }
                vec8_t<T> x0, x1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x0[e] = static_cast<T>(sa0[e] * rs);
                    x0[4 + e] = static_cast<T>(sb0[e] * rs);
                    x1[e] = static_cast<T>(sa1[e] * rs);
                    x1[4 + e] = static_cast<T>(sb1[e] * rs);
                }
                finish(x0, x1, cv, sv);
            }
        };
        if (p.ks == 4) prologue(IntTag<4>{});
        else if (p.ks == 2) prologue(IntTag<2>{});
        else if (p.ks == 1) prologue(IntTag<1>{});
        else { // many slabs: too many registers to hold them raw — KV first, then the summing loads
            prefetch_kv(b, IntTag<0>{}, IntTag<NDP>{});
            if (has) {
                vec8_t<T> cv = {}, sv = {}, x1 = {};
                vec8_t<T> x0 = load8_splitk<T>(p.qkv_slabs, p.ks, slab_stride, off);
                if (is_rot) {
                    const int64_t trow = p.pos_idx ? p.pos_idx[seq] : pos;
                    cv = load8(static_cast<const T *>(p.cos_t) + trow * (D / 2) + c * 8);
                    sv = load8(static_cast<const T *>(p.sin_t) + trow * (D / 2) + c * 8);
                    x1 = load8_splitk<T>(p.qkv_slabs, p.ks, slab_stride, off + D / 2);
                }
                finish(x0, x1, cv, sv);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // do not hoist the next requests above the slab sums (register pressure)
        prefetch_kv(b, IntTag<NDP>{}, IntTag<ND>{}); // the remaining slots: the slab registers are free again
        __syncthreads();
        if constexpr (MF) {
#pragma unroll
            for (int j = 0; j < MT::QS; ++j) {
                qb[j] = *reinterpret_cast<const vec8_t<T> *>(&sm_q[min(mh, G - 1) * D + 32 * j + 8 * mq]);
                if (mh >= G) qb[j] = vec8_t<T>{};
            }
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) qv[g] = *reinterpret_cast<const vec8_t<T> *>(&sm_q[g * D + chunk * 8]);
        }
    }
#pragma unroll
    for (int g = 0; g < (MF ? 1 : G); ++g) {
        m[g] = kNegBig;
        l[g] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[g][j] = 0.f;
    }
#pragma unroll
    for (int mm = 0; mm < MT::OS; ++mm) acc4[mm] = float4_t{0.f, 0.f, 0.f, 0.f};
    // steady state: every refill is unconditional, so the waits between slots are exact counted vmcnt waits (a
    // conditional load in the body makes the compiler wait for one slot more than needed); slot d attends block
    // b + d*NW and is refilled with block b + (d+ND)*NW
    if (b + (2 * ND - 1) * NW < blk_end) {
        // The first ND requests above are conditional (short sequence blocks), so on entry the compiler cannot know how
        // many loads are outstanding and would size EVERY wait of the loop for the fewest — i.e. wait for the newest
        // slot before touching the oldest (seen in the ISA of the r01 kernel as well: its prefetch never overlapped
        // its own compute). One full wait here — all ND slots were requested long ago, this is the pipeline fill —
        // hands the loop an exact state; from then on every wait in it is a counted vmcnt.
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0), expcnt/lgkmcnt untouched
      do {
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            // scheduling fences: left alone, the compiler computes all ND blocks back to back behind ONE vmcnt(0)
            // and sinks every refill to the end of the iteration — nothing in flight while it computes (seen in
            // the ISA: 34 us instead of 27 at batch 32 x 1k). Pinned, the wait before slot d+1 is a counted one.
            // (the block-table entry of the refill is requested BEFORE the block is attended: behind the fence its
            // scalar-load round trip would sit between every attend and the refill it feeds)
            const int64_t phys_next = bt[b + (d + ND) * NW];
            __builtin_amdgcn_sched_barrier(0);
            attend(b + d * NW, Kr[d], Vr[d]);
            __builtin_amdgcn_sched_barrier(0);
            load_phys(phys_next, Kr[d], Vr[d]);
            __builtin_amdgcn_sched_barrier(0);
        }
        b += ND * NW;
      } while (b + (2 * ND - 1) * NW < blk_end);
    }
    // drain: at most 2*ND-1 blocks left, the first ND of them already in their slots
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        if (b + d * NW < blk_end) {
            attend_tail(b + d * NW, Kr[d], Vr[d]);
            __builtin_amdgcn_sched_barrier(0);
            if (b + (d + ND) * NW < blk_end) load_block(b + (d + ND) * NW, Kr[d], Vr[d]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int d = 0; d < ND - 1; ++d)
        if (b + (d + ND) * NW < blk_end) attend_tail(b + (d + ND) * NW, Kr[d], Vr[d]);

    if constexpr (MF) {
        // O^T of the last block is stored by DS instructions below (swl_common.h)
#pragma unroll
        for (int mm = 0; mm < MT::OS; ++mm) mfma_results_tie(acc4[mm]);
        mfma_results_ready<4>(acc4[MT::OS - 1]);
        // the four lanes (q = 0..3) of a head share m and each hold the row sum of their own tokens; O^T is complete
        const float lt = rows_allreduce_sum(l[0]);
        if (mh < G) {
            if (mq == 0) {
                sm_ml[wave][mh][0] = m[0];
                sm_ml[wave][mh][1] = lt;
            }
#pragma unroll
            for (int mm = 0; mm < MT::OS; ++mm)
#pragma unroll
                for (int r = 0; r < 4; ++r) sm_acc[wave][mh][16 * mm + 4 * mq + r] = acc4[mm][r];
        }
    } else {
    // ---- merge the TPI rows of this wave (each row holds tokens == row mod TPI) ----------------
#pragma unroll
    for (int mask = LPT; mask < 64; mask <<= 1) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float m2 = __shfl_xor(m[g], mask, 64);
            const float l2 = __shfl_xor(l[g], mask, 64);
            const float M = fmaxf(m[g], m2);
            const float w1 = fast_exp2((m[g] - M) * c);
            const float w2 = fast_exp2((m2 - M) * c);
            l[g] = l[g] * w1 + l2 * w2;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a2 = __shfl_xor(acc[g][j], mask, 64);
                acc[g][j] = acc[g][j] * w1 + a2 * w2;
            }
            m[g] = M;
        }
    }
    if (row == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (chunk == 0) {
                sm_ml[wave][g][0] = m[g];
                sm_ml[wave][g][1] = l[g];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) sm_acc[wave][g][chunk * 8 + j] = acc[g][j];
        }
    }
    } // !MF
    __syncthreads();

    // ---- merge the NW waves and write the partial (or the final output when there is one split) -
    const int nsb = p.num_seq_blocks;
    for (int oidx = threadIdx.x; oidx < G * D; oidx += NT) {
        const int g = oidx / D;
        const int d = oidx % D;
        float M = sm_ml[0][g][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) M = fmaxf(M, sm_ml[w][g][0]);
        float Lsum = 0.f, A = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float wgt = fast_exp2((sm_ml[w][g][0] - M) * c);
            Lsum = fmaf(sm_ml[w][g][1], wgt, Lsum);
            A = fmaf(sm_acc[w][g][d], wgt, A);
        }
        const float out = A / Lsum;
        const int head = kvh * G + g;
        if (nsb == 1) {
            static_cast<T *>(p.o_direct)[seq * p.o_tok_stride + static_cast<int64_t>(head) * D + d] =
                to_t<T>(out);
        } else {
            const int64_t part = (static_cast<int64_t>(seq) * p.H + head) * nsb + split;
            p.mid_o[part * D + d] = out;
            if (d == 0) p.mid_lse[part] = fast_log2(Lsum) + M * c;
        }
    }
}

// grid (H, Bd), one wave per (sequence, q-head): LSE-weighted merge of the partials (reference paged_attn.py:
// 108-150). Latency is all there is to this kernel (n <= a few dozen partials of 512 B each), so nothing in it may
// serialise on memory: the n log-sum-exps arrive in ONE load (lane s holds partial s), the weights are computed once
// per lane and handed out with v_readlane, and the partial outputs are requested eight at a time before the first
// FMA — two round trips in all. (r01: one dependent load pair per partial, 10.9 us at 16 partials = more than the
// attention itself at batch 1; profiles/r02d.) The weighted sum keeps the partial order.
template <typename T, int D>
__global__ __launch_bounds__(64) void paged_attn_phase2_kernel(
    T *__restrict__ o, const float *__restrict__ mid_o, const float *__restrict__ mid_lse,
    const int *__restrict__ seq_lens, int H, int seq_block_size, int num_seq_blocks,
    int64_t o_tok_stride) {
    const int head = blockIdx.x;
    const int seq = blockIdx.y;
    const int lane = threadIdx.x;
    const int len = seq_lens[seq];
    if (len <= 0) return; // an inert row of a padded decode batch: phase 1 wrote no partials for it
    const int n = (len + seq_block_size - 1) / seq_block_size;
    const int64_t base = (static_cast<int64_t>(seq) * H + head) * num_seq_blocks;
    constexpr int VD = (D + 63) / 64;
    constexpr int U = 8;

    // first requests: the log-sum-exps of up to 64 partials and the first U partial outputs
    float lse = lane < n ? mid_lse[base + lane] : kNegBig;
    float v[U][VD];
    auto request = [&](int s0) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < VD; ++q) {
                const int d = lane + q * 64;
                v[u][q] = (s0 + u < n && d < D) ? mid_o[(base + s0 + u) * D + d] : 0.f;
            }
    };
    request(0);
    float M = lse;
    for (int s = lane + 64; s < n; s += 64) M = fmaxf(M, mid_lse[base + s]); // (> 64 partials: 128k-token sequences)
    M = wave_allreduce_max(M);

    float acc[VD];
#pragma unroll
    for (int q = 0; q < VD; ++q) acc[q] = 0.f;
    float Lsum = 0.f;
    for (int c0 = 0; c0 < n; c0 += 64) {
        if (c0 > 0) lse = c0 + lane < n ? mid_lse[base + c0 + lane] : kNegBig;
        const float w_lane = c0 + lane < n ? fast_exp2(lse - M) : 0.f;
        Lsum += w_lane;
        const int lim = min(64, n - c0);
        for (int s0 = 0; s0 < lim; s0 += U) {
            if (c0 + s0 > 0) request(c0 + s0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                // lanes beyond the last partial hold w = 0 and v = 0: no guard needed
                const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w_lane),
                                                                                    (s0 + u) & 63));
#pragma unroll
                for (int q = 0; q < VD; ++q) acc[q] = fmaf(w, v[u][q], acc[q]);
            }
        }
    }
    Lsum = wave_allreduce_sum(Lsum);
#pragma unroll
    for (int q = 0; q < VD; ++q) {
        const int d = lane + q * 64;
        if (d < D) o[seq * o_tok_stride + static_cast<int64_t>(head) * D + d] = to_t<T>(acc[q] / Lsum);
    }
}

template <typename T, int D, int G, bool QKV>
static int launch_phase1(const PagedAttnParams &p, int Bd, hipStream_t stream) {
    const dim3 grid(p.num_seq_blocks, p.KVH, Bd);
    // >= 32 KV blocks per sequence block: 8-wave workgroups (>= 4 blocks per wave); else 4 waves.
    // (the VALU variant of G = 8 would need > 256 registers per lane; the matrix-core one, G >= 2, does not.)
    if (p.seq_block_size >= 32 * kBlk)
        hipLaunchKernelGGL((paged_attn_phase1_kernel<T, D, G, 8, QKV>), grid, dim3(512), 0, stream, p, p.block_table,
                           p.seq_lens, p.seq_ids);
    else
        hipLaunchKernelGGL((paged_attn_phase1_kernel<T, D, G, 4, QKV>), grid, dim3(256), 0, stream, p, p.block_table,
                           p.seq_lens, p.seq_ids);
    return check_launch();
}

template <typename T, int D, bool QKV>
static int dispatch_phase1_g(const PagedAttnParams &p, int Bd, int G, hipStream_t stream) {
    switch (G) {
    case 1: return launch_phase1<T, D, 1, QKV>(p, Bd, stream);
    case 2: return launch_phase1<T, D, 2, QKV>(p, Bd, stream);
    case 4: return launch_phase1<T, D, 4, QKV>(p, Bd, stream);
    case 8: return launch_phase1<T, D, 8, QKV>(p, Bd, stream);
    default: return SWL_ERR_UNSUPPORTED;
    }
}

template <typename T, bool QKV = false>
static int dispatch_phase1(const PagedAttnParams &p, int Bd, int D, int G, hipStream_t stream) {
    switch (D) {
    case 32: return dispatch_phase1_g<T, 32, QKV>(p, Bd, G, stream);
    case 64: return dispatch_phase1_g<T, 64, QKV>(p, Bd, G, stream);
    case 128: return dispatch_phase1_g<T, 128, QKV>(p, Bd, G, stream);
    default: return SWL_ERR_UNSUPPORTED;
    }
}

template <typename T>
static int dispatch_phase2(T *o, const float *mid_o, const float *mid_lse, const int *seq_lens,
                           int Bd, int H, int D, int sbs, int nsb, int64_t o_tok_stride,
                           hipStream_t stream) {
    const dim3 grid(H, Bd);
#define SWL_P2(DD)                                                                               \
    hipLaunchKernelGGL((paged_attn_phase2_kernel<T, DD>), grid, dim3(64), 0, stream, o, mid_o,   \
                       mid_lse, seq_lens, H, sbs, nsb, o_tok_stride)
    switch (D) {
    case 32: SWL_P2(32); break;
    case 64: SWL_P2(64); break;
    case 128: SWL_P2(128); break;
    default: return SWL_ERR_UNSUPPORTED;
    }
#undef SWL_P2
    return check_launch();
}

} // namespace swl

extern "C" size_t swl_paged_attn_scratch_bytes(int32_t num_decoding_seqs, int32_t num_q_heads,
                                               int32_t head_dim, int32_t num_seq_blocks) {
    if (num_decoding_seqs <= 0 || num_q_heads <= 0 || head_dim <= 0 || num_seq_blocks <= 0) return 0;
    const size_t parts = static_cast<size_t>(num_decoding_seqs) * num_q_heads * num_seq_blocks;
    return parts * (static_cast<size_t>(head_dim) + 1) * sizeof(float);
}

extern "C" int swl_paged_attn_phase1(void *o_direct, const void *q, const void *k_cache,
                                     const void *v_cache, const int32_t *block_table,
                                     const int32_t *seq_ids, const int32_t *seq_lens, float *mid_o,
                                     float *mid_lse, float softmax_scale, int32_t num_decoding_seqs,
                                     int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
                                     int32_t num_layers, int32_t block_size, int32_t cur_layer,
                                     int32_t max_blocks_per_seq, int32_t seq_block_size,
                                     int32_t num_seq_blocks, int64_t q_tok_stride,
                                     int64_t o_tok_stride, int32_t dtype, swl_stream_t stream) {
    if (num_decoding_seqs < 0) return SWL_ERR_BAD_ARG;
    if (num_decoding_seqs == 0 || num_seq_blocks == 0) return SWL_OK;
    if (!q || !k_cache || !v_cache || !block_table || !seq_ids || !seq_lens) return SWL_ERR_BAD_ARG;
    if (num_seq_blocks < 0 || num_q_heads <= 0 || num_kv_heads <= 0 ||
        num_q_heads % num_kv_heads != 0 || num_layers <= 0 || cur_layer < 0 ||
        cur_layer >= num_layers || max_blocks_per_seq <= 0)
        return SWL_ERR_BAD_ARG;
    if (block_size != swl::kBlk) return SWL_ERR_UNSUPPORTED;
    if (seq_block_size <= 0 || seq_block_size % block_size != 0) return SWL_ERR_BAD_ARG;
    if (num_seq_blocks == 1 ? !o_direct : (!mid_o || !mid_lse)) return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(q) || !swl::aligned16(k_cache) || !swl::aligned16(v_cache) ||
        (q_tok_stride & 7))
        return SWL_ERR_BAD_ARG;
    if (num_decoding_seqs > 65535 || num_kv_heads > 65535) return SWL_ERR_UNSUPPORTED;
    swl::PagedAttnParams p{};
    p.o_direct = o_direct;
    p.q = q;
    p.k_cache = k_cache;
    p.v_cache = v_cache;
    p.block_table = block_table;
    p.seq_ids = seq_ids;
    p.seq_lens = seq_lens;
    p.mid_o = mid_o;
    p.mid_lse = mid_lse;
    p.scale_log2e = softmax_scale * 1.44269504088896340736f;
    p.H = num_q_heads;
    p.KVH = num_kv_heads;
    p.L = num_layers;
    p.layer = cur_layer;
    p.max_blocks_per_seq = max_blocks_per_seq;
    p.seq_block_size = seq_block_size;
    p.num_seq_blocks = num_seq_blocks;
    p.q_tok_stride = q_tok_stride;
    p.o_tok_stride = o_tok_stride;
    const int G = num_q_heads / num_kv_heads;
    SWL_DISPATCH_DTYPE(dtype, T, {
        return swl::dispatch_phase1<T>(p, num_decoding_seqs, head_dim, G,
                                       static_cast<hipStream_t>(stream));
    });
}

extern "C" int swl_paged_attn_phase2(void *o, const float *mid_o, const float *mid_lse,
                                     const int32_t *seq_lens, int32_t num_decoding_seqs,
                                     int32_t num_q_heads, int32_t head_dim, int32_t seq_block_size,
                                     int32_t num_seq_blocks, int64_t o_tok_stride, int32_t dtype,
                                     swl_stream_t stream) {
    if (num_decoding_seqs < 0) return SWL_ERR_BAD_ARG;
    if (num_decoding_seqs == 0 || num_seq_blocks == 0) return SWL_OK;
    if (!o || !mid_o || !mid_lse || !seq_lens || num_q_heads <= 0 || seq_block_size <= 0 ||
        num_seq_blocks < 0)
        return SWL_ERR_BAD_ARG;
    if (num_decoding_seqs > 65535) return SWL_ERR_UNSUPPORTED;
    SWL_DISPATCH_DTYPE(dtype, T, {
        return swl::dispatch_phase2<T>(static_cast<T *>(o), mid_o, mid_lse, seq_lens,
                                       num_decoding_seqs, num_q_heads, head_dim, seq_block_size,
                                       num_seq_blocks, o_tok_stride,
                                       static_cast<hipStream_t>(stream));
    });
}

extern "C" int swl_paged_attn_decode(void *o, const void *q, const void *k_cache,
                                     const void *v_cache, const int32_t *block_table,
                                     const int32_t *seq_ids, const int32_t *seq_lens, void *scratch,
                                     float softmax_scale, int32_t num_decoding_seqs,
                                     int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
                                     int32_t num_layers, int32_t block_size, int32_t cur_layer,
                                     int32_t max_blocks_per_seq, int32_t seq_block_size,
                                     int32_t num_seq_blocks, int64_t q_tok_stride,
                                     int64_t o_tok_stride, int32_t dtype, swl_stream_t stream) {
    if (num_decoding_seqs < 0) return SWL_ERR_BAD_ARG;
    if (num_decoding_seqs == 0 || num_seq_blocks == 0) return SWL_OK;
    if (!o) return SWL_ERR_BAD_ARG;
    float *mid_o = nullptr, *mid_lse = nullptr;
    if (num_seq_blocks > 1) {
        if (!scratch || !swl::aligned16(scratch)) return SWL_ERR_BAD_ARG;
        mid_o = static_cast<float *>(scratch);
        mid_lse = mid_o + static_cast<size_t>(num_decoding_seqs) * num_q_heads * num_seq_blocks *
                              head_dim;
    }
    int rc = swl_paged_attn_phase1(o, q, k_cache, v_cache, block_table, seq_ids, seq_lens, mid_o,
                                   mid_lse, softmax_scale, num_decoding_seqs, num_q_heads,
                                   num_kv_heads, head_dim, num_layers, block_size, cur_layer,
                                   max_blocks_per_seq, seq_block_size, num_seq_blocks, q_tok_stride,
                                   o_tok_stride, dtype, stream);
    if (rc != SWL_OK || num_seq_blocks == 1) return rc;
    return swl_paged_attn_phase2(o, mid_o, mid_lse, seq_lens, num_decoding_seqs, num_q_heads,
                                 head_dim, seq_block_size, num_seq_blocks, o_tok_stride, dtype,
                                 stream);
}

/* Decode attention fed by the split-K slabs of the fused qkv projection: rotary + KV-store of the new token run
 * in the attention kernel's prologue (no launch of their own), the new k/v are attended from registers. */
static int paged_attn_decode_qkv_impl(void *o, const float *qkv_slabs, int32_t k_splits, const void *cos_table,
                                      const void *sin_table, const int32_t *pos_idx, void *k_cache, void *v_cache,
                                      const int32_t *block_table, const int32_t *seq_ids, const int32_t *seq_lens,
                                      void *scratch, float softmax_scale, int32_t num_decoding_seqs,
                                      int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim, int32_t num_layers,
                                      int32_t block_size, int32_t cur_layer, int32_t max_blocks_per_seq,
                                      int32_t seq_block_size, int32_t num_seq_blocks, int64_t o_tok_stride, int32_t dtype,
                                      swl_stream_t stream, const float *row_ssq, int32_t ssq_parts, int32_t hidden,
                                      float eps, bool merge = true);

extern "C" int swl_paged_attn_decode_qkv(void *o, const float *qkv_slabs, int32_t k_splits, const void *cos_table,
                                         const void *sin_table, const int32_t *pos_idx, void *k_cache,
                                         void *v_cache, const int32_t *block_table, const int32_t *seq_ids,
                                         const int32_t *seq_lens, void *scratch, float softmax_scale,
                                         int32_t num_decoding_seqs, int32_t num_q_heads, int32_t num_kv_heads,
                                         int32_t head_dim, int32_t num_layers, int32_t block_size,
                                         int32_t cur_layer, int32_t max_blocks_per_seq, int32_t seq_block_size,
                                         int32_t num_seq_blocks, int64_t o_tok_stride, int32_t dtype,
                                         swl_stream_t stream) {
    return paged_attn_decode_qkv_impl(o, qkv_slabs, k_splits, cos_table, sin_table, pos_idx, k_cache, v_cache, block_table,
                                      seq_ids, seq_lens, scratch, softmax_scale, num_decoding_seqs, num_q_heads,
                                      num_kv_heads, head_dim, num_layers, block_size, cur_layer, max_blocks_per_seq,
                                      seq_block_size, num_seq_blocks, o_tok_stride, dtype, stream, nullptr, 0, 0, 0.f);
}

/* swl_paged_attn_decode_qkv on the slabs of a qkv projection whose input had its RMSNorm scale deferred
 * (swl_splitk_add_scale): row_ssq[ssq_parts][num_decoding_seqs] are the per-1024-column sums of squares of the residual
 * rows, `hidden` their length; the slab sums are multiplied by 1/sqrt(sum/hidden + eps) in fp32 before they are rounded,
 * rotated and stored. k_splits in {1, 2, 4}, ssq_parts <= 8. */
extern "C" int swl_paged_attn_decode_qkv_rs(void *o, const float *qkv_slabs, int32_t k_splits, const float *row_ssq,
                                            int32_t ssq_parts, int32_t hidden, float eps, const void *cos_table,
                                            const void *sin_table, const int32_t *pos_idx, void *k_cache,
                                            void *v_cache, const int32_t *block_table, const int32_t *seq_ids,
                                            const int32_t *seq_lens, void *scratch, float softmax_scale,
                                            int32_t num_decoding_seqs, int32_t num_q_heads, int32_t num_kv_heads,
                                            int32_t head_dim, int32_t num_layers, int32_t block_size,
                                            int32_t cur_layer, int32_t max_blocks_per_seq, int32_t seq_block_size,
                                            int32_t num_seq_blocks, int64_t o_tok_stride, int32_t dtype,
                                            swl_stream_t stream) {
    if (num_decoding_seqs == 0 || num_seq_blocks == 0) return num_decoding_seqs < 0 ? SWL_ERR_BAD_ARG : SWL_OK;
    if (!row_ssq || ssq_parts <= 0 || hidden <= 0) return SWL_ERR_BAD_ARG;
    if (ssq_parts > 8 || !(k_splits == 1 || k_splits == 2 || k_splits == 4)) return SWL_ERR_UNSUPPORTED;
    return paged_attn_decode_qkv_impl(o, qkv_slabs, k_splits, cos_table, sin_table, pos_idx, k_cache, v_cache, block_table,
                                      seq_ids, seq_lens, scratch, softmax_scale, num_decoding_seqs, num_q_heads,
                                      num_kv_heads, head_dim, num_layers, block_size, cur_layer, max_blocks_per_seq,
                                      seq_block_size, num_seq_blocks, o_tok_stride, dtype, stream, row_ssq, ssq_parts,
                                      hidden, eps);
}

static int paged_attn_decode_qkv_impl(void *o, const float *qkv_slabs, int32_t k_splits, const void *cos_table,
                                         const void *sin_table, const int32_t *pos_idx, void *k_cache,
                                         void *v_cache, const int32_t *block_table, const int32_t *seq_ids,
                                         const int32_t *seq_lens, void *scratch, float softmax_scale,
                                         int32_t num_decoding_seqs, int32_t num_q_heads, int32_t num_kv_heads,
                                         int32_t head_dim, int32_t num_layers, int32_t block_size,
                                         int32_t cur_layer, int32_t max_blocks_per_seq, int32_t seq_block_size,
                                         int32_t num_seq_blocks, int64_t o_tok_stride, int32_t dtype,
                                         swl_stream_t stream, const float *row_ssq, int32_t ssq_parts, int32_t hidden,
                                         float eps, bool merge) {
    if (num_decoding_seqs < 0) return SWL_ERR_BAD_ARG;
    if (num_decoding_seqs == 0 || num_seq_blocks == 0) return SWL_OK;
    if ((!o && (merge || num_seq_blocks == 1)) || !qkv_slabs || !cos_table || !sin_table || !k_cache || !v_cache || !block_table || !seq_ids ||
        !seq_lens || k_splits <= 0)
        return SWL_ERR_BAD_ARG;
    if (num_seq_blocks < 0 || num_q_heads <= 0 || num_kv_heads <= 0 || num_q_heads % num_kv_heads != 0 ||
        num_layers <= 0 || cur_layer < 0 || cur_layer >= num_layers || max_blocks_per_seq <= 0)
        return SWL_ERR_BAD_ARG;
    if (block_size != swl::kBlk) return SWL_ERR_UNSUPPORTED;
    if (seq_block_size <= 0 || seq_block_size % block_size != 0) return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(qkv_slabs) || !swl::aligned16(k_cache) || !swl::aligned16(v_cache) ||
        !swl::aligned16(cos_table) || !swl::aligned16(sin_table))
        return SWL_ERR_BAD_ARG;
    if (num_decoding_seqs > 65535 || num_kv_heads > 65535) return SWL_ERR_UNSUPPORTED;
    float *mid_o = nullptr, *mid_lse = nullptr;
    if (num_seq_blocks > 1) {
        if (!scratch || !swl::aligned16(scratch)) return SWL_ERR_BAD_ARG;
        mid_o = static_cast<float *>(scratch);
        mid_lse = mid_o + static_cast<size_t>(num_decoding_seqs) * num_q_heads * num_seq_blocks * head_dim;
    }
    swl::PagedAttnParams p{};
    p.o_direct = o;
    p.k_cache = k_cache;
    p.v_cache = v_cache;
    p.block_table = block_table;
    p.seq_ids = seq_ids;
    p.seq_lens = seq_lens;
    p.mid_o = mid_o;
    p.mid_lse = mid_lse;
    p.scale_log2e = softmax_scale * 1.44269504088896340736f;
    p.H = num_q_heads;
    p.KVH = num_kv_heads;
    p.L = num_layers;
    p.layer = cur_layer;
    p.max_blocks_per_seq = max_blocks_per_seq;
    p.seq_block_size = seq_block_size;
    p.num_seq_blocks = num_seq_blocks;
    p.o_tok_stride = o_tok_stride;
    p.qkv_slabs = qkv_slabs;
    p.ks = k_splits;
    p.cos_t = cos_table;
    p.sin_t = sin_table;
    p.pos_idx = pos_idx;
    p.row_ssq = row_ssq;
    p.ssq_parts = ssq_parts;
    p.hidden = hidden;
    p.eps = eps;
    const int G = num_q_heads / num_kv_heads;
    int rc;
    SWL_DISPATCH_DTYPE(dtype, T, {
        rc = swl::dispatch_phase1<T, true>(p, num_decoding_seqs, head_dim, G, static_cast<hipStream_t>(stream));
    });
    if (rc != SWL_OK || num_seq_blocks == 1 || !merge) return rc;
    return swl_paged_attn_phase2(o, mid_o, mid_lse, seq_lens, num_decoding_seqs, num_q_heads, head_dim,
                                 seq_block_size, num_seq_blocks, o_tok_stride, dtype, stream);
}

/* swl_paged_attn_decode_qkv_rs stopped after phase 1 when the sequences are split (num_seq_blocks > 1): the partials stay
 * in `scratch` (mid_o fp32 [Bd][H][nsb][D] followed by mid_lse fp32 [Bd][H][nsb], the reference's format,
 * paged_attn.py:106-108) for a consumer that merges them itself (swl_gemm_tiny_partial_from_attn). With one split per
 * sequence it is swl_paged_attn_decode_qkv_rs: the output goes to `o`. */
extern "C" int swl_paged_attn_decode_qkv_rs_partials(void *o, const float *qkv_slabs, int32_t k_splits,
                                                     const float *row_ssq, int32_t ssq_parts, int32_t hidden, float eps,
                                                     const void *cos_table, const void *sin_table,
                                                     const int32_t *pos_idx, void *k_cache, void *v_cache,
                                                     const int32_t *block_table, const int32_t *seq_ids,
                                                     const int32_t *seq_lens, void *scratch, float softmax_scale,
                                                     int32_t num_decoding_seqs, int32_t num_q_heads,
                                                     int32_t num_kv_heads, int32_t head_dim, int32_t num_layers,
                                                     int32_t block_size, int32_t cur_layer, int32_t max_blocks_per_seq,
                                                     int32_t seq_block_size, int32_t num_seq_blocks,
                                                     int64_t o_tok_stride, int32_t dtype, swl_stream_t stream) {
    if (num_decoding_seqs == 0 || num_seq_blocks == 0) return num_decoding_seqs < 0 ? SWL_ERR_BAD_ARG : SWL_OK;
    if (!row_ssq || ssq_parts <= 0 || hidden <= 0) return SWL_ERR_BAD_ARG;
    if (ssq_parts > 8 || !(k_splits == 1 || k_splits == 2 || k_splits == 4)) return SWL_ERR_UNSUPPORTED;
    return paged_attn_decode_qkv_impl(o, qkv_slabs, k_splits, cos_table, sin_table, pos_idx, k_cache, v_cache, block_table,
                                      seq_ids, seq_lens, scratch, softmax_scale, num_decoding_seqs, num_q_heads,
                                      num_kv_heads, head_dim, num_layers, block_size, cur_layer, max_blocks_per_seq,
                                      seq_block_size, num_seq_blocks, o_tok_stride, dtype, stream, row_ssq, ssq_parts,
                                      hidden, eps, false);
}
