// prefill_attn.hip — varlen causal flash attention (GQA) on the CDNA4 matrix cores (gfx950).
//
// Replaces the prefill attention of the reference's hot path: the third-party call
// vllm_flash_attn.flash_attn_varlen_func (swiftllm/worker/layers/transformer_layer.py:83-96) and its
// in-repo Triton equivalent _fwd_prefill_attention (swiftllm/worker/kernels/prefill_attn.py:9-100).
// Attention runs over the FRESH q/k/v projections; the paged pool is not read (no chunked prefill
// in the reference). MFMA-bound: 4 * sum_s(len_s^2) * D * H / 2 flop per layer.
//
// Structure (wave64 / 32x32x16 MFMA native, not a warp-tiled CUDA port):
//   * workgroup = 4 waves = one 128-row Q block of one (sequence, q-head); wave w owns 32 q-rows;
//     K/V are walked in 64-key tiles staged through LDS (K rows padded to D+8, V rows to D+32
//     elements: both fragment reads below are bank-conflict free);
//   * S^T = K.Q^T ("swapped" product): A = K rows (ds_read_b128), B = Q^T held in registers for the
//     whole kernel. In the 32x32 C layout lane (l%32) then owns ONE q-row, so the online-softmax
//     max/sum are in-lane reductions plus a single lane^32 exchange;
//   * O^T = V^T.P^T: A = V^T fragments fetched with the gfx950 LDS transpose read
//     (ds_read_b64_tr_b16) from the row-major V tile, B = P^T taken straight from the S^T
//     accumulator registers (the k-order of the product is permuted identically on both operands,
//     so no cross-lane shuffle is needed); the per-row rescale factor is lane-local for O^T too;
//   * the next K/V tile is fetched from HBM/L2 into registers while the current one is consumed
//     and written to LDS after the barrier (issue-early / write-late staging);
//   * 1-D grid decoded XCD-aware: all Q blocks and the G q-heads of one (sequence, kv-head) land on
//     one XCD so their shared K/V stays in that XCD's L2; long (late) Q blocks are issued first.
// Numerics as the reference: fp32 scores * (scale*log2e), exp2, P rounded to the storage dtype for
// the PV product, fp32 accumulators, one rounding at the store (prefill_attn.py:62-71,100).
#include "swl_common.h"

namespace swl {

typedef short short4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float16_t mfma32(vec8_t<f16> a, vec8_t<f16> b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float16_t mfma32(vec8_t<bf16> a, vec8_t<bf16> b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// LDS transpose read: the 16 lanes of a group each supply the address of 4 consecutive 16-bit
// elements (one quarter of a 16-element row; lanes 4r..4r+3 = row r); lane i receives column i of
// the resulting 4x16 block, i.e. {row0[i], row1[i], row2[i], row3[i]}.
template <typename T>
__device__ __forceinline__ short4_t lds_tr_read(const T *p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (short4_t __attribute__((address_space(3))) *)(p));
}

// v_max3_f32 as one opaque instruction: fmaxf() on MFMA results makes hipcc canonicalise every input first
// (v_max_f32 x, x, x) — three instructions where one does.
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

struct PrefillParams {
    void *o;
    const void *q;
    const void *k;
    const void *v;
    const int *cu_seqlens;
    int num_seqs, H, KVH, num_q_blocks;
    float scale_log2e;
    int64_t q_tok_stride, k_tok_stride, v_tok_stride, o_tok_stride;
};

constexpr int kBQ = 256; // q rows per workgroup (4 waves x 64)
constexpr int kBK = 32;  // keys per LDS tile = keys per softmax step
constexpr int kRB = 2;   // 32-row blocks per wave

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// One wave per SIMD with the whole register file, 64 q rows per wave (two 32-row blocks): every K / V^T fragment read from
// LDS feeds TWO MFMAs, the matrix pipe belongs to one in-order stream, and the softmax of tile h rides in the gaps of tile
// h+1's QK^T MFMAs (row maxima of h+1 in the gaps of tile h's PV MFMAs) — interleaved by hand, held in place by data flow
// (see pipe_step). Per 32-key tile and wave: 32 MFMA, 24 LDS fragment reads, ~230 VALU.
//     A:  S_r(h+1) = K(h+1).Q_r^T  [2 x 8 MFMA]  ||  P_r(h) = exp2(S_r(h)*c - m_r), row sums, P -> 16 bit   [VALU]
//         (O_r *= alpha_r(h) when some row's maximum moved: wave-uniform branch per row block)
//     B:  O_r += V(h)^T.P_r(h)^T   [2 x 8 MFMA]  ||  row maxima of S_r(h+1) -> m_r, alpha_r(h+1)            [VALU]
// K and V tiles are double-buffered in LDS with ONE barrier per tile: in iteration h K(h+2) and V(h+1) — requested an
// iteration earlier — are committed to the buffers whose readers (QK(h), PV(h-1)) finished before the barrier, and
// K(h+3), V(h+2) are requested. Staging goes through buffer loads: rows past the end of the sequence are out of range
// and read as zeros. The last two tiles of a wave are masked (the diagonals of its two row blocks).
template <typename T, int D>
__global__ __launch_bounds__(256, 1) void prefill_attn_kernel(PrefillParams p) {
    constexpr int KRS = D + 8;   // K row pitch (elements): 16 consecutive rows hit 16 distinct 16-B slots
    constexpr int VRS = D + 32;  // V row pitch: 4 rows x two 16-col halves tile the 64 banks exactly
    constexpr int KSTEPS = D / 16;
    constexpr int DT = D / 32;
    constexpr int CPR = D / 8;           // 16-byte chunks per row
    constexpr int RPP = 256 / CPR;       // rows staged per pass
    constexpr int NPASS = (kBK + RPP - 1) / RPP;     // passes per tile (head_dim 32: one pass, half the threads idle)
    constexpr bool PART = kBK % RPP != 0;             // ... then threads whose row lies past the tile stage nothing
    __shared__ __attribute__((aligned(16))) T Ks[2][kBK * KRS];
    __shared__ __attribute__((aligned(16))) T Vs[2][kBK * VRS];

    // ---- XCD-aware decode of the 1-D grid ------------------------------------------------------
    const int G = p.H / p.KVH;
    const int per_unit = G * p.num_q_blocks;
    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int j0 = id >> 3;
    const int unit = xcd + 8 * (j0 / per_unit);
    if (unit >= p.num_seqs * p.KVH) return;
    const int inner = j0 % per_unit;
    const int g = inner % G;
    const int qb = p.num_q_blocks - 1 - inner / G; // longest rows first
    const int seq = unit / p.KVH;
    const int kvh = unit % p.KVH;
    const int head = kvh * G + g;

    const int start = p.cu_seqlens[seq];
    const int len = p.cu_seqlens[seq + 1] - start;
    const int q0 = qb * kBQ;
    if (q0 >= len) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31;
    const int hf = lane >> 5;
    const int q0w = q0 + wave * 32 * kRB; // first q row of this wave
    const float c = p.scale_log2e;
    int qrow[kRB];                         // this lane's q rows (sequence-local), one per row block
#pragma unroll
    for (int r = 0; r < kRB; ++r) qrow[r] = q0w + 32 * r + l32;

    // ---- Q^T B-fragments: lane holds Q[qrow][kk*16 + hf*8 .. +8] ---------------------------------
    vec8_t<T> qf[kRB][KSTEPS];
#pragma unroll
    for (int r = 0; r < kRB; ++r) {
        const bool ok = qrow[r] < len;
        const T *qp = static_cast<const T *>(p.q) + (static_cast<int64_t>(start) + (ok ? qrow[r] : 0)) * p.q_tok_stride +
                      static_cast<int64_t>(head) * D + hf * 8;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            vec8_t<T> t = load8(qp + kk * 16);
            if (!ok) t = vec8_t<T>{};
            qf[r][kk] = t;
        }
    }

    // ---- staging: thread -> (row srow + pass*RPP, chunk sc) of a tile. The K / V rows of this (sequence, kv head) are
    // one buffer each, `len` rows long: byte offset of (row, chunk) = row * stride + chunk * 16; rows >= len are out of
    // range and load as zeros ----------------------------------------------------------------------------------------
    const int srow = tid / CPR;
    const int sc = tid % CPR;
    const T *kbase = static_cast<const T *>(p.k) + static_cast<int64_t>(start) * p.k_tok_stride + static_cast<int64_t>(kvh) * D;
    const T *vbase = static_cast<const T *>(p.v) + static_cast<int64_t>(start) * p.v_tok_stride + static_cast<int64_t>(kvh) * D;
    const unsigned k_row_bytes = static_cast<unsigned>(p.k_tok_stride * sizeof(T));
    const unsigned v_row_bytes = static_cast<unsigned>(p.v_tok_stride * sizeof(T));
    const unsigned k_bytes = static_cast<unsigned>(len - 1) * k_row_bytes + D * sizeof(T);
    const unsigned v_bytes = static_cast<unsigned>(len - 1) * v_row_bytes + D * sizeof(T);
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(kbase), 0, k_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(vbase), 0, v_bytes, 0x00020000);
    unsigned k_off = srow * k_row_bytes + sc * 16;   // of the next K tile to request (pass 0)
    unsigned v_off = srow * v_row_bytes + sc * 16;
    auto fetch = [&](vec8_t<T>(&dst)[NPASS], const __amdgpu_buffer_rsrc_t &rsrc, unsigned &off, unsigned row_bytes) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
            if (!PART || srow + ps * RPP < kBK)
                dst[ps] = __builtin_bit_cast(vec8_t<T>, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + ps * RPP * row_bytes, 0, 0));
        off += kBK * row_bytes;
    };
    auto commit = [&](T *buf, int pitch, const vec8_t<T>(&src)[NPASS]) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
            if (!PART || srow + ps * RPP < kBK)
                *reinterpret_cast<vec8_t<T> *>(&buf[(srow + ps * RPP) * pitch + sc * 8]) = src[ps];
    };

    const int kv_end = min(len, q0 + kBQ);
    const int ntiles = (kv_end + kBK - 1) / kBK;          // tiles the workgroup walks
    // tiles this wave attends: those that start at or below the diagonal of its SECOND row block; the last two are masked
    // (tile my_tiles-2 is the diagonal of row block 0, tile my_tiles-1 lies wholly above block 0 and is block 1's
    // diagonal). A wave whose rows all lie past the end of the sequence attends nothing.
    const int my_tiles = q0w < len ? min(ntiles, q0w / kBK + kRB) : 0;

    // per-lane LDS offsets of the fragment reads
    const int k_frag_off = l32 * KRS + hf * 8;                            // + kk*16
    const int i16 = lane & 15;
    const int v_frag_off = (4 * hf + (i16 >> 2)) * VRS + 16 * ((lane >> 4) & 1) + 4 * (i16 & 3);

    float16_t ot[kRB][DT];
#pragma unroll
    for (int r = 0; r < kRB; ++r)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) ot[r][dt] = float16_t{};
    float m_run[kRB], l_run[kRB], alpha_cur[kRB];
#pragma unroll
    for (int r = 0; r < kRB; ++r) {
        m_run[r] = kNegBig;
        l_run[r] = 0.f;
        alpha_cur[r] = 1.0f;
    }

    // causal mask (keys beyond len are > every valid q row as well)
    auto mask = [&](float16_t &st, int key0, int row) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            if (key > row) st[r] = kNegBig;
        }
    };
    // row maxima of one score block: five max3 of three scores + the sixteenth
    auto block_max = [&](const float16_t &st) -> float {
        float mx[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) mx[q] = max3(st[3 * q], st[3 * q + 1], st[3 * q + 2]);
        float m4 = max3(max3(mx[0], mx[1], mx[2]), max3(mx[3], mx[4], st[15]), kNegBig);
        return fmaxf(m4, __shfl_xor(m4, 32, 64));
    };
    auto new_max = [&](int r, float m4) {   // -> alpha_cur[r], m_run[r]
        const float m_new = fmaxf(m_run[r], m4 * c);
        alpha_cur[r] = fast_exp2(m_run[r] - m_new);
        m_run[r] = m_new;
    };
    auto rescale = [&](int r) {
        // O *= alpha only when some row of this block raised its maximum (after the first tiles that is rare)
        if (!__all(alpha_cur[r] == 1.0f)) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int e = 0; e < 16; ++e) ot[r][dt][e] *= alpha_cur[r];
        }
    };

    // ---- prologue: K(0), V(0), K(1) into LDS; K(2), V(1) on their way; S(0) and its row maxima ---------------------
    vec8_t<T> kst[NPASS], vst[NPASS];
    {
        vec8_t<T> k1[NPASS];
        fetch(kst, k_rsrc, k_off, k_row_bytes);
        fetch(vst, v_rsrc, v_off, v_row_bytes);
        fetch(k1, k_rsrc, k_off, k_row_bytes);      // (past the end of the sequence: zeros, never read)
        commit(Ks[0], KRS, kst);
        commit(Vs[0], VRS, vst);
        commit(Ks[1], KRS, k1);
    }
    fetch(kst, k_rsrc, k_off, k_row_bytes);         // K(2)
    fetch(vst, v_rsrc, v_off, v_row_bytes);         // V(1)
    __syncthreads();
    // the live score blocks swap roles every tile: even tiles sit in sA, odd ones in sB (no register copies)
    float16_t sA[kRB], sB[kRB];
    if (my_tiles > 0) {
#pragma unroll
        for (int r = 0; r < kRB; ++r) sA[r] = float16_t{};
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            const vec8_t<T> kf = *reinterpret_cast<const vec8_t<T> *>(&Ks[0][k_frag_off + kk * 16]);
#pragma unroll
            for (int r = 0; r < kRB; ++r) sA[r] = mfma32(kf, qf[r][kk], sA[r]);
        }
#pragma unroll
        for (int r = 0; r < kRB; ++r) mfma_results_tie(sA[r]);
        mfma_results_ready<8>(sA[kRB - 1]);
        if (my_tiles <= kRB) {
#pragma unroll
            for (int r = 0; r < kRB; ++r) mask(sA[r], 0, qrow[r]);
        }
#pragma unroll
        for (int r = 0; r < kRB; ++r) new_max(r, block_max(sA[r]));
    }
    // P = exp2(S*c - m), row sums, 16-bit P^T B-fragments of one row block (k-step ks: registers 8*ks .. 8*ks+7)
    auto expo = [&](int r, const float16_t &st, vec8_t<T>(&pb)[2]) {
        const float m_new = m_run[r];
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float pv_ = fast_exp2(fmaf(st[e], c, -m_new));
            if (e & 1) ps1 += pv_;
            else ps0 += pv_;
            pb[e >> 3][e & 7] = to_t<T>(pv_);
        }
        l_run[r] = fmaf(l_run[r], alpha_cur[r], ps0 + ps1);
    };
    // the wave's last tile: nothing left to overlap with
    auto last_step = [&](float16_t(&cur)[kRB], const T *v_cur) {
        vec8_t<T> pb[kRB][2];
#pragma unroll
        for (int r = 0; r < kRB; ++r) {
            expo(r, cur[r], pb[r]);
            rescale(r);
        }
#pragma unroll
        for (int i = 0; i < 2 * DT; ++i) {
            const T *vp2 = &v_cur[v_frag_off + 16 * (i / DT) * VRS + (i % DT) * 32];
            const short4_t lo = lds_tr_read(vp2);
            const short4_t hi = lds_tr_read(vp2 + 8 * VRS);
            typedef short short8_t __attribute__((ext_vector_type(8)));
            const short8_t both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int r = 0; r < kRB; ++r)
                ot[r][i % DT] = mfma32(__builtin_bit_cast(vec8_t<T>, both), pb[r][i / DT], ot[r][i % DT]);
        }
    };
    // one tile with a successor: `cur` holds S(h) of both row blocks (masked, maxima taken), `nxt` receives S(h+1). The two
    // pipes are interleaved BY HAND, two MFMAs + their share of the VALU work per step; what holds the order is data flow:
    // an empty volatile asm after the MFMAs of step k takes the next K fragment, the scores of step k and the running sums
    // as in/out operands, so step k's VALU work and the MFMAs of step k+1 can start only after it, and asm k+1 only after
    // both (sched_group_barrier and sched_barrier(0) did not hold it: profiles/r03_prefill_attn_pipeline_probe.md).
    auto pipe_step = [&](int h, float16_t(&cur)[kRB], float16_t(&nxt)[kRB], const T *k_next, const T *v_cur,
                         bool mask_next) {
        vec8_t<T> pb[kRB][2];
        // ---- A: S(h+1) = K(h+1).Q^T on the matrix pipe, P(h) = exp2(S(h)*c - m) on the VALU ---------------------------
        {
            float mn[kRB], ps0[kRB], ps1[kRB];
#pragma unroll
            for (int r = 0; r < kRB; ++r) {
                mn[r] = m_run[r];
                ps0[r] = ps1[r] = 0.f;
                nxt[r] = float16_t{};
            }
            constexpr int EPS = 16 / KSTEPS; // scores of each row block exponentiated per step (2 at D = 128)
            vec8_t<T> kf[3];
            kf[0] = *reinterpret_cast<const vec8_t<T> *>(&k_next[k_frag_off]);
            if (KSTEPS > 1) kf[1] = *reinterpret_cast<const vec8_t<T> *>(&k_next[k_frag_off + 16]);
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                if (kk + 2 < KSTEPS)
                    kf[(kk + 2) % 3] = *reinterpret_cast<const vec8_t<T> *>(&k_next[k_frag_off + (kk + 2) * 16]);
#pragma unroll
                for (int r = 0; r < kRB; ++r) nxt[r] = mfma32(kf[kk % 3], qf[r][kk], nxt[r]);
                float x[kRB][EPS];
#pragma unroll
                for (int r = 0; r < kRB; ++r)
#pragma unroll
                    for (int e = 0; e < EPS; ++e) x[r][e] = cur[r][kk * EPS + e];
                vec8_t<T> &kn = kf[(kk + 1 < KSTEPS ? kk + 1 : kk) % 3]; // the fragment step kk+1 takes (last step: any)
                asm volatile("" : "+v"(kn), "+v"(x[0][0]), "+v"(x[1][0]), "+v"(ps0[0]), "+v"(ps0[1]));
#pragma unroll
                for (int r = 0; r < kRB; ++r)
#pragma unroll
                    for (int e = 0; e < EPS; ++e) {
                        const int idx = kk * EPS + e;
                        const float pv_ = fast_exp2(fmaf(x[r][e], c, -mn[r]));
                        if (idx & 1) ps1[r] += pv_;
                        else ps0[r] += pv_;
                        pb[r][idx >> 3][idx & 7] = to_t<T>(pv_);
                    }
            }
#pragma unroll
            for (int r = 0; r < kRB; ++r) l_run[r] = fmaf(l_run[r], alpha_cur[r], ps0[r] + ps1[r]);
        }
#pragma unroll
        for (int r = 0; r < kRB; ++r) rescale(r);
#pragma unroll
        for (int r = 0; r < kRB; ++r) mfma_results_tie(nxt[r]);
        mfma_results_ready<8>(nxt[kRB - 1]);
        if (mask_next) {
#pragma unroll
            for (int r = 0; r < kRB; ++r) mask(nxt[r], (h + 1) * kBK, qrow[r]);
        }
        // ---- B: O += V(h)^T.P(h)^T on the matrix pipe, row maxima of S(h+1) on the VALU --------------------------------
        {
            typedef short short8_t __attribute__((ext_vector_type(8)));
            constexpr int NS = 2 * DT; // fragment steps: (ks, dt); two MFMAs each
            short4_t lo[3], hi[3];
            auto vread = [&](int i, int slot) {
                const T *vp2 = &v_cur[v_frag_off + 16 * (i / DT) * VRS + (i % DT) * 32];
                lo[slot] = lds_tr_read(vp2);
                hi[slot] = lds_tr_read(vp2 + 8 * VRS);
            };
            vread(0, 0);
            if (NS > 1) vread(1, 1);
            float m4[kRB];
#pragma unroll
            for (int r = 0; r < kRB; ++r) m4[r] = 0.f;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                if (i + 2 < NS) vread(i + 2, (i + 2) % 3);
                const short8_t both = {lo[i % 3][0], lo[i % 3][1], lo[i % 3][2], lo[i % 3][3],
                                       hi[i % 3][0], hi[i % 3][1], hi[i % 3][2], hi[i % 3][3]};
#pragma unroll
                for (int r = 0; r < kRB; ++r)
                    ot[r][i % DT] = mfma32(__builtin_bit_cast(vec8_t<T>, both), pb[r][i / DT], ot[r][i % DT]);
                // the row maxima of S(h+1): block r in step r (NS >= 2 for every head_dim)
                if (i < kRB) m4[i] = block_max(nxt[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < kRB; ++r) new_max(r, m4[r]);
        }
    };
    // every iteration, attended or not: barrier, commit what was requested an iteration ago, request the next tiles
    auto stage = [&](int h) {
        __syncthreads(); // K(h+1) and V(h) are in LDS; every wave is done with iteration h-1's reads
        commit(Ks[h & 1], KRS, kst);            // K(h+2)   (tiles past the end: zeros nobody reads)
        commit(Vs[(h + 1) & 1], VRS, vst);      // V(h+1)
        fetch(kst, k_rsrc, k_off, k_row_bytes); // K(h+3)
        fetch(vst, v_rsrc, v_off, v_row_bytes); // V(h+2)
    };
    // A wave walks its tiles in three stretches, each straight-line or a loop of its own so that O and the score blocks
    // stay in the same registers from one iteration to the next: the pipelined steps (unrolled by two: the score blocks
    // swap roles; the last kRB of them mask their successors), the wave's last tile, then the tiles above its diagonals
    // (staging and barriers only — the other waves of the workgroup still attend them).
    int h = 0;
    for (; h + 2 < my_tiles; h += 2) {
        stage(h);
        pipe_step(h, sA, sB, Ks[(h + 1) & 1], Vs[h & 1], h + 1 + kRB >= my_tiles);
        stage(h + 1);
        pipe_step(h + 1, sB, sA, Ks[h & 1], Vs[(h + 1) & 1], h + 2 + kRB >= my_tiles);
    }
    if (h + 1 < my_tiles) {         // my_tiles even: one pipelined step left, then the last tile (in sB)
        stage(h);
        pipe_step(h, sA, sB, Ks[(h + 1) & 1], Vs[h & 1], true);
        ++h;
        stage(h);
        last_step(sB, Vs[h & 1]);
        ++h;
    } else if (h < my_tiles) {      // my_tiles odd: the last tile is next (in sA)
        stage(h);
        last_step(sA, Vs[h & 1]);
        ++h;
    }
    for (; h < ntiles; ++h) stage(h);

    // ---- epilogue: O[qrow][d] = O^T[d][qrow] / l ---------------------------------------------------
#pragma unroll
    for (int r = 0; r < kRB; ++r)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) mfma_results_tie(ot[r][dt]);
    mfma_results_ready<8>(ot[kRB - 1][DT - 1]);
#pragma unroll
    for (int r = 0; r < kRB; ++r) {
        const float l_tot = l_run[r] + __shfl_xor(l_run[r], 32, 64);
        const float inv = 1.0f / l_tot;
        if (qrow[r] < len) {
            T *op = static_cast<T *>(p.o) + (static_cast<int64_t>(start) + qrow[r]) * p.o_tok_stride +
                    static_cast<int64_t>(head) * D + 4 * hf;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    typedef T vec4 __attribute__((ext_vector_type(4)));
                    vec4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = to_t<T>(ot[r][dt][4 * r4 + e] * inv);
                    *reinterpret_cast<vec4 *>(op + dt * 32 + 8 * r4) = ov;
                }
        }
    }
}

} // namespace swl

extern "C" int swl_prefill_attn_varlen(void *o, const void *q, const void *k, const void *v,
                                       const int32_t *cu_seqlens, int32_t num_prefill_seqs,
                                       int32_t max_prefill_len, int32_t num_q_heads,
                                       int32_t num_kv_heads, int32_t head_dim, float softmax_scale,
                                       int64_t q_tok_stride, int64_t k_tok_stride,
                                       int64_t v_tok_stride, int64_t o_tok_stride, int32_t dtype,
                                       swl_stream_t stream) {
    if (num_prefill_seqs < 0 || max_prefill_len < 0) return SWL_ERR_BAD_ARG;
    if (num_prefill_seqs == 0 || max_prefill_len == 0) return SWL_OK;
    if (!o || !q || !k || !v || !cu_seqlens) return SWL_ERR_BAD_ARG;
    if (num_q_heads <= 0 || num_kv_heads <= 0 || num_q_heads % num_kv_heads != 0)
        return SWL_ERR_BAD_ARG;
    if (!(head_dim == 32 || head_dim == 64 || head_dim == 128)) return SWL_ERR_UNSUPPORTED;
    if ((q_tok_stride & 7) || (k_tok_stride & 7) || (v_tok_stride & 7) || (o_tok_stride & 3))
        return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(q) || !swl::aligned16(k) || !swl::aligned16(v) ||
        (reinterpret_cast<uintptr_t>(o) & 7u))
        return SWL_ERR_BAD_ARG;
    swl::PrefillParams p;
    p.o = o;
    p.q = q;
    p.k = k;
    p.v = v;
    p.cu_seqlens = cu_seqlens;
    p.num_seqs = num_prefill_seqs;
    p.H = num_q_heads;
    p.KVH = num_kv_heads;
    p.num_q_blocks = (max_prefill_len + swl::kBQ - 1) / swl::kBQ;
    p.scale_log2e = softmax_scale * 1.44269504088896340736f;
    p.q_tok_stride = q_tok_stride;
    p.k_tok_stride = k_tok_stride;
    p.v_tok_stride = v_tok_stride;
    p.o_tok_stride = o_tok_stride;
    const int64_t units = static_cast<int64_t>(num_prefill_seqs) * num_kv_heads;
    const int64_t units_padded = (units + 7) / 8 * 8;
    const int64_t nblocks = units_padded * (num_q_heads / num_kv_heads) * p.num_q_blocks;
    if (nblocks > 0x7fffffffLL) return SWL_ERR_UNSUPPORTED;
    const dim3 grid(static_cast<unsigned>(nblocks));
    hipStream_t s = static_cast<hipStream_t>(stream);
    SWL_DISPATCH_DTYPE(dtype, T, {
        if (head_dim == 128)
            hipLaunchKernelGGL((swl::prefill_attn_kernel<T, 128>), grid, dim3(256), 0, s, p);
        else if (head_dim == 64)
            hipLaunchKernelGGL((swl::prefill_attn_kernel<T, 64>), grid, dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL((swl::prefill_attn_kernel<T, 32>), grid, dim3(256), 0, s, p);
    });
    return swl::check_launch();
}
