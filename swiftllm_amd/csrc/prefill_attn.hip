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
//   * epilogue: the wave's 32 x D output tile goes through LDS (the dead K/V tiles) and is stored as whole 256-byte rows,
//     16 bytes per lane (r04; r01-r03 stored 8-byte pieces of 32 rows a token pitch apart per instruction);
//   * 1-D grid decoded XCD-aware: all Q blocks and the G q-heads of one (sequence, kv-head) land on
//     one XCD so their shared K/V stays in that XCD's L2; long (late) Q blocks are issued first.
// Numerics as the reference: fp32 scores * (scale*log2e), exp2, P rounded to the storage dtype for
// the PV product, fp32 accumulators, one rounding at the store (prefill_attn.py:62-71,100).
#include <stdlib.h>

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

struct PrefillParams {
    void *o;
    const void *q;
    const void *k;
    const void *v;
    const int *cu_seqlens;
    int num_seqs, H, KVH, num_q_blocks;
    int hpw;    // LDS-DMA kernel: q-heads per workgroup (4, 2 or 1); num_q_blocks then counts blocks of (4 / hpw) * 32 rows
    float scale_log2e;
    int64_t q_tok_stride, k_tok_stride, v_tok_stride, o_tok_stride;
};

constexpr float kLazyMax = 4.0f; // see the softmax step of the kernels (p <= 16)
constexpr int kBQ = 128; // q rows per workgroup
constexpr int kBK = 64;  // keys per LDS tile

template <typename T, int D>
__global__ __launch_bounds__(256, 2) void prefill_attn_kernel(PrefillParams p) {
    constexpr int KRS = D + 8;   // K row pitch (elements): 16 consecutive rows hit 16 distinct 16-B slots
    constexpr int VRS = D + 32;  // V row pitch: 4 rows x two 16-col halves tile the 64 banks exactly
    constexpr int KSTEPS = D / 16;
    constexpr int DT = D / 32;
    constexpr int CPR = D / 8;           // 16-byte chunks per row
    constexpr int RPP = 256 / CPR;       // rows staged per pass
    constexpr int NPASS = kBK / RPP;     // passes per tile
    __shared__ __attribute__((aligned(16))) T smem[kBK * KRS + kBK * VRS];   // K tile, V tile; the O tiles of the epilogue
    T *const Ks = smem;
    T *const Vs = smem + kBK * KRS;

    // ---- XCD-aware decode of the 1-D grid ------------------------------------------------------
    const int G = p.H / p.KVH;
    const int per_unit = G * p.num_q_blocks;
    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int j = id >> 3;
    const int unit = xcd + 8 * (j / per_unit);
    if (unit >= p.num_seqs * p.KVH) return;
    const int inner = j % per_unit;
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
    const int q0w = q0 + wave * 32; // first q row of this wave
    const int qrow = q0w + l32;     // this lane's q row (sequence-local)
    const float c = p.scale_log2e;

    const T *qg = static_cast<const T *>(p.q);
    const T *kg = static_cast<const T *>(p.k) + static_cast<int64_t>(kvh) * D;
    const T *vg = static_cast<const T *>(p.v) + static_cast<int64_t>(kvh) * D;

    // ---- Q^T B-fragments: lane holds Q[qrow][kk*16 + hf*8 .. +8] ---------------------------------
    vec8_t<T> qf[KSTEPS];
    {
        const bool ok = qrow < len;
        const T *qp = qg + (static_cast<int64_t>(start) + (ok ? qrow : 0)) * p.q_tok_stride +
                      static_cast<int64_t>(head) * D + hf * 8;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            vec8_t<T> t = load8(qp + kk * 16);
            if (!ok) t = vec8_t<T>{};
            qf[kk] = t;
        }
    }

    float16_t ot[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) ot[dt] = float16_t{};
    float m_run = kNegBig;
    float l_run = 0.f;

    // ---- staging: thread -> (row srow + pass*RPP, chunk sc) ---------------------------------------
    const int srow = tid / CPR;
    const int sc = tid % CPR;
    vec8_t<T> kst[NPASS], vst[NPASS];
    // Row pointers of this thread's staging passes, advanced by one tile per fetch (tiles are fetched in order). r01
    // recomputed `(start + key) * token_stride` per load: 92 VALU instructions per tile, a third of them quarter-rate
    // 32-bit multiplies of the 64-bit product — ~700 cycles beside the 1024 cycles of the tile's 32 MFMAs, the "staging
    // cost" that neither fewer barriers, more occupancy nor a longer fetch distance could touch (profiles/r02n).
    const T *kp[NPASS], *vp[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int64_t tok = static_cast<int64_t>(start) + srow + ps * RPP;
        kp[ps] = kg + tok * p.k_tok_stride + sc * 8;
        vp[ps] = vg + tok * p.v_tok_stride + sc * 8;
    }
    const int64_t kstep = static_cast<int64_t>(kBK) * p.k_tok_stride, vstep = static_cast<int64_t>(kBK) * p.v_tok_stride;
    auto fetch_tile = [&](int key0) {
        if (key0 + kBK <= len) { // (workgroup-uniform) every row of the tile exists: plain loads
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                kst[ps] = load8(kp[ps]);
                vst[ps] = load8(vp[ps]);
            }
        } else {                  // the sequence ends inside this tile: rows past it are zeros, their pointers unused
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                vec8_t<T> kt = vec8_t<T>{}, vt = vec8_t<T>{};
                if (key0 + srow + ps * RPP < len) {
                    kt = load8(kp[ps]);
                    vt = load8(vp[ps]);
                }
                kst[ps] = kt;
                vst[ps] = vt;
            }
        }
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            kp[ps] += kstep;
            vp[ps] += vstep;
        }
    };
    auto commit_tile = [&]() {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = srow + ps * RPP;
            *reinterpret_cast<vec8_t<T> *>(&Ks[r * KRS + sc * 8]) = kst[ps];
            *reinterpret_cast<vec8_t<T> *>(&Vs[r * VRS + sc * 8]) = vst[ps];
        }
    };

    const int kv_end = min(len, q0 + kBQ);
    const int ntiles = (kv_end + kBK - 1) / kBK;

    // per-lane LDS offsets of the fragment reads
    const int k_frag_off = l32 * KRS + hf * 8;                            // + t*32*KRS + kk*16
    const int i16 = lane & 15;
    const int v_frag_off = (4 * hf + (i16 >> 2)) * VRS + 16 * ((lane >> 4) & 1) + 4 * (i16 & 3);

    fetch_tile(0);
    for (int tile = 0; tile < ntiles; ++tile) {
        const int key0 = tile * kBK;
        __syncthreads(); // everyone finished reading the previous tile
        commit_tile();
        __syncthreads();
        if (tile + 1 < ntiles) fetch_tile(key0 + kBK); // in flight during the MFMAs below

        if (key0 > q0w + 31) continue; // whole tile above this wave's diagonal

        // ---- S^T = K . Q^T  (two 32-key sub-tiles) ------------------------------------------------
        float16_t st[2];
        __builtin_amdgcn_s_setprio(1); // favour the wave that is feeding the matrix pipe (two waves per SIMD)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            st[t] = float16_t{};
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const vec8_t<T> kf =
                    *reinterpret_cast<const vec8_t<T> *>(&Ks[k_frag_off + t * 32 * KRS + kk * 16]);
                st[t] = mfma32(kf, qf[kk], st[t]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        // S^T is read by VALU next, behind the diagonal-tile branch (swl_common.h)
        mfma_results_tie(st[0]);
        mfma_results_ready<8>(st[1]);
        // causal mask on the diagonal tiles (keys beyond len are > every valid q row as well)
        if (key0 + kBK - 1 > q0w) {
            const int dmask = qrow - key0 - 4 * hf;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // key > qrow with key = key0 + 32 t + (r & 3) + 8 (r >> 2) + 4 hf: a compile-time constant against ONE
                    // per-lane value (one compare + one select per element; r01-r06a rebuilt the key first: three)
                    if ((r & 3) + 8 * (r >> 2) + t * 32 > dmask) st[t][r] = kNegBig;
                }
        }
        // ---- online softmax: this lane owns q row l32, keys split with lane^32 ---------------------
        float mx = st[0][0];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // lazy running maximum (r06b): a row moves its maximum only when the tile raised it by more than kLazyMax (base-2
        // exponent units), so p stays <= 2^kLazyMax instead of <= 1 — 16-bit floats round p to the same RELATIVE precision at
        // either scale, l and O accumulate in fp32 — and alpha is exactly 1 for that row: the 64-multiply rescale of O below
        // then fires on the first tile or two of a row block instead of on most tiles (any of a wave's 32 rows setting a
        // record). Decided per ROW, so a row's arithmetic still depends on its own scores only (the causality test holds
        // bit for bit).
        const float m_cand = fmaxf(m_run, mx * c);
        const float m_new = m_cand - m_run > kLazyMax ? m_cand : m_run;
        const float alpha = fast_exp2(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = fast_exp2(fmaf(st[t][r], c, -m_new));
                st[t][r] = pv;
                psum += pv;
            }
        l_run = fmaf(l_run, alpha, psum);
        // rescale O only when some row of this wave actually raised its max (alpha == 1 otherwise:
        // after the first tiles of a sequence that is the common case) — a wave-uniform branch that
        // removes 16*DT multiplies per lane per tile
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[dt][r] *= alpha;
        }

        // ---- O^T += V^T . P^T ------------------------------------------------------------------------
        __builtin_amdgcn_s_setprio(1);
        // k-step (t, ks): this lane's 8 k-slots are keys t*32 + 16*ks + 4*hf + {0..3} and + 8 + {0..3}
        // == accumulator registers 8*ks .. 8*ks+7 of st[t] — identical order on both operands.
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                vec8_t<T> pb;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) pb[jj] = to_t<T>(st[t][8 * ks + jj]);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const T *vp = &Vs[v_frag_off + (t * 32 + 16 * ks) * VRS + dt * 32];
                    const short4_t lo = lds_tr_read(vp);
                    const short4_t hi = lds_tr_read(vp + 8 * VRS);
                    typedef short short8_t __attribute__((ext_vector_type(8)));
                    const short8_t both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const vec8_t<T> vf = __builtin_bit_cast(vec8_t<T>, both);
                    ot[dt] = mfma32(vf, pb, ot[dt]);
                }
            }
        __builtin_amdgcn_s_setprio(0);
    }

    // ---- epilogue: O[qrow][d] = O^T[d][qrow] / l ---------------------------------------------------
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) mfma_results_tie(ot[dt]);
    mfma_results_ready<8>(ot[DT - 1]);
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    // O[qrow][d] = O^T[d][qrow] / l, written as WHOLE ROWS: a lane holds 4 consecutive d of ONE row per register quad, so a
    // direct store instruction touches 32 rows a token pitch apart with 2 x 8 bytes each — 16 such instructions per lane,
    // the store-issue-bound tail the in-box guide prices at ~4 us per workgroup (T21). The wave's 32 x D tile goes through
    // LDS instead (the K/V tiles are dead: one barrier, the waves leave the loop together) and comes back row-major,
    // 16 bytes per lane, 16 lanes = one 256-byte row: 8 full-line stores per lane at D = 128.
    constexpr int ORS = D + 8;                       // O row pitch in LDS (elements)
    static_assert(4 * 32 * ORS <= kBK * KRS + kBK * VRS, "the four waves' O tiles must fit the K/V tiles' LDS");
    __syncthreads();                                 // every wave is done with the last K/V tile
    T *ow = smem + wave * 32 * ORS;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            typedef T vec4 __attribute__((ext_vector_type(4)));
            vec4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = to_t<T>(ot[dt][4 * r4 + e] * inv);
            *reinterpret_cast<vec4 *>(ow + l32 * ORS + dt * 32 + 8 * r4 + 4 * hf) = ov;
        }
    // (wave-private tile: LDS operations of one wave complete in order, no barrier needed)
    constexpr int RPI = 64 / CPR;                    // rows per store instruction
    T *obase = static_cast<T *>(p.o) + static_cast<int64_t>(start) * p.o_tok_stride + static_cast<int64_t>(head) * D;
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
        const int row = i * RPI + lane / CPR, ch = lane % CPR;
        const vec8_t<T> v = *reinterpret_cast<const vec8_t<T> *>(ow + row * ORS + ch * 8);
        if (q0w + row < len)
            store8(obase + static_cast<int64_t>(q0w + row) * p.o_tok_stride + ch * 8, v);
    }
}


// ---- r06: LDS-DMA staged variant (head_dim 128) -----------------------------------------------------------------------------
// Same arithmetic, same fragment order, same bits as prefill_attn_kernel above; what changes is how K/V tiles reach LDS and
// how many instructions a wave issues per tile. The r02-r05 counters (profiles/r04r_prefill_attn_pmc.md: 9.0 VALU per MFMA,
// matrix pipe 31-45 % busy) said the loop is short of ISSUE SLOTS, and the ISA showed where they go beside the irreducible
// softmax: the register-staged K/V (8 global loads + 8 ds_write_b128 + 28 moves + pointer arithmetic per tile) and two
// barriers per tile. Here:
//   * K/V tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPRs, no ds_write):
//     8 DMA instructions per wave per tile, addressed SGPR base + a per-lane 32-bit offset that is the SAME for every tile
//     (the base advances by one scalar add per tile) — no vector address arithmetic in the loop at all;
//   * LDS-DMA writes lane-linear images (no row padding possible), so the bank conflicts of the fragment reads are removed
//     by swizzling on the SOURCE side (the guide's rule 21): the K image holds chunk c of row r at slot c ^ (r & 15)
//     (ds_read_b128 of 16 rows: 16 distinct slots), the V image holds 64-byte window w of row r at window w ^ (r & 3)
//     (ds_read_b64_tr_b16 of 4 rows x 64 B: 4 distinct windows); the fragment reads apply the same XOR;
//   * two LDS buffers, ONE barrier per tile: [wait own DMA of tile t] [barrier] [issue DMA of tile t+1] [compute tile t];
//     the DMA of the next tile flies under the 32 MFMAs of this one. The DMA and its counted wait are inline asm: hipcc
//     would put a vmcnt(0) in front of the first LDS read after a DMA it knows about (cdna_hip_programming.md section 5).
template <typename T>
__global__ __launch_bounds__(256, 2) void prefill_attn_dma_kernel(PrefillParams p) {
    constexpr int D = 128;
    constexpr int KSTEPS = D / 16, DT = D / 32;
    constexpr int ROWB = D * 2;                 // bytes per K/V row image (256)
    constexpr int TILEB = kBK * ROWB;           // 16 KiB per K or V tile image
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TILEB];      // [buffer][K, V]; the O tiles of the epilogue

    // r06b — the four waves of a workgroup are `hpw` q-heads of the kv-head x 4/hpw row blocks of 32 (GQA: two heads x 64
    // rows; MHA: one head x 128 rows as before; four heads x 32 rows is the other legal value). Heads of one kv-head read the
    // same K/V tiles and have the same causal extent: fewer waves idle through the last tiles of their workgroup (128-row
    // blocks: 2 of every 8i + 8 wave-tile slots) and the per-tile barrier joins waves that did more nearly the same work.
    const int G = p.H / p.KVH;
    const int hpw = p.hpw, hgroups = G / hpw;
    const int rows_wg = (4 / hpw) * 32;
    const int per_unit = hgroups * p.num_q_blocks;
    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int j = id >> 3;
    const int unit = xcd + 8 * (j / per_unit);
    if (unit >= p.num_seqs * p.KVH) return;
    const int inner = j % per_unit;
    const int qb = p.num_q_blocks - 1 - inner / hgroups; // longest rows first
    const int seq = unit / p.KVH;
    const int kvh = unit % p.KVH;

    const int start = p.cu_seqlens[seq];
    const int len = p.cu_seqlens[seq + 1] - start;
    const int q0 = qb * rows_wg;
    if (q0 >= len) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = kvh * G + (inner % hgroups) * hpw + wave % hpw;
    const int l32 = lane & 31;
    const int hf = lane >> 5;
    const int q0w = q0 + (wave / hpw) * 32;
    const int qrow = q0w + l32;
    const float c = p.scale_log2e;

    const T *qg = static_cast<const T *>(p.q);
    vec8_t<T> qf[KSTEPS];

    float16_t ot[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) ot[dt] = float16_t{};
    float m_run = kNegBig;
    float l_run = 0.f;

    // ---- DMA addressing: this wave stages rows 16 wave + 4 jj + lane/16 (jj = 0..3) of every tile --------------------
    const int drow = (lane >> 4);                 // + 16 wave + 4 jj
    const int dslot = lane & 15;
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(
        (__attribute__((address_space(3))) char *)smem));
    const int64_t kstride_b = p.k_tok_stride * 2, vstride_b = p.v_tok_stride * 2;
    unsigned koff[4], voff[4];                    // byte offsets from the tile's base address (row 0 of the tile)
    auto dma_offsets = [&](int rows_valid) {      // rows_valid: rows of the tile that exist (64 except in the last tile)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int r = 16 * wave + 4 * jj + drow;
            const int rs = min(r, rows_valid - 1);      // rows past the end: a copy of the last row (masked: key > every q row)
            koff[jj] = static_cast<unsigned>(rs * kstride_b) + static_cast<unsigned>((dslot ^ (r & 15)) * 16);
            voff[jj] = static_cast<unsigned>(rs * vstride_b) +
                       static_cast<unsigned>(((((dslot >> 2) ^ (r & 3)) << 2) | (dslot & 3)) * 16);
        }
    };
    dma_offsets(kBK);
    const char *kbase = reinterpret_cast<const char *>(static_cast<const T *>(p.k) + static_cast<int64_t>(kvh) * D) +
                        static_cast<int64_t>(start) * kstride_b;
    const char *vbase = reinterpret_cast<const char *>(static_cast<const T *>(p.v) + static_cast<int64_t>(kvh) * D) +
                        static_cast<int64_t>(start) * vstride_b;
    auto issue_tile = [&](int tile, int buf) {
        const char *kb = kbase + static_cast<int64_t>(tile) * kBK * kstride_b;     // (wave-uniform: SGPR pair)
        const char *vb = vbase + static_cast<int64_t>(tile) * kBK * vstride_b;
        const unsigned ldsk = lds0 + static_cast<unsigned>(buf) * 2 * TILEB + static_cast<unsigned>(wave) * 4096;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(koff[jj]), "s"(kb), "s"(ldsk + jj * 1024) : "memory");
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff[jj]), "s"(vb), "s"(ldsk + TILEB + jj * 1024) : "memory");
        }
    };

    const int kv_end = min(len, q0 + rows_wg);
    const int ntiles = (kv_end + kBK - 1) / kBK;
    const int last_rows = kv_end - (ntiles - 1) * kBK > 0 ? min(kBK, len - (ntiles - 1) * kBK) : kBK;

    // ---- fragment read offsets (bytes, within a buffer) ------------------------------------------------------------------
    // K: row l32 (+ 32 t), chunk 2 kk + hf at slot (2 kk + hf) ^ (row & 15)
    unsigned kfo[KSTEPS];
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) kfo[kk] = static_cast<unsigned>(l32 * ROWB + (((2 * kk + hf) ^ (l32 & 15)) * 16));
    // V (transpose read): row 4 hf + i16/4 (+ 8, + 16 ks, + 32 t), 8 bytes at column 32 dt + 16 ((lane >> 4) & 1) + 4 (i16 & 3)
    // = window dt, byte 32 ((lane >> 4) & 1) + 8 (i16 & 3) inside it; window stored at dt ^ (row & 3), row & 3 = (i16 >> 2) & 3
    const int i16 = lane & 15;
    unsigned vfo[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
        vfo[dt] = static_cast<unsigned>(TILEB + (4 * hf + (i16 >> 2)) * ROWB + ((dt ^ ((i16 >> 2) & 3)) * 64) +
                                        32 * ((lane >> 4) & 1) + 8 * (i16 & 3));

    if (ntiles == 1 && last_rows < kBK) dma_offsets(last_rows);
    issue_tile(0, 0);
    // ---- Q^T fragments (r06b): full-line loads through LDS instead of fragment-shaped ones ----------------------------------
    // The B fragment of lane (row l32, half hf) is 16 bytes of a 256-byte row: loaded directly, one instruction gathers 32
    // rows x 32 B (quarter lines: the slowest load shape on this part, tools/probe/l2_fill_probe.hip). Here a wave requests
    // its 32 rows as 8 x (4 rows x 256 B), parks them in its 8 KiB of the SECOND K/V buffer (nothing is staged there before
    // the first barrier of the loop) at piece ^ (row & 15), and reads the fragments back with ds_read_b128 — conflict-free
    // for the 8-lane write groups and the 16-lane read groups. Same bits in the same registers.
    {
        T *qs = reinterpret_cast<T *>(smem + 2 * TILEB) + wave * 32 * D;
        const int rsub = lane >> 4, pc = lane & 15;
        vec8_t<T> t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 4 * i + rsub;
            const bool ok = q0w + row < len;
            t[i] = load8(qg + (static_cast<int64_t>(start) + (ok ? q0w + row : 0)) * p.q_tok_stride +
                         static_cast<int64_t>(head) * D + pc * 8);
            if (!ok) t[i] = vec8_t<T>{};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 4 * i + rsub;
            *reinterpret_cast<vec8_t<T> *>(qs + row * D + ((pc ^ (row & 15)) << 3)) = t[i];
        }
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk)
            qf[kk] = *reinterpret_cast<const vec8_t<T> *>(qs + l32 * D + (((2 * kk + hf) ^ (l32 & 15)) << 3));
        // the reads have returned before any wave's DMA of tile 1 (issued behind the loop's first barrier) can land here
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    for (int tile = 0; tile < ntiles; ++tile) {
        const int key0 = tile * kBK;
        const int buf = tile & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of tile `tile` have landed
        __builtin_amdgcn_s_barrier();                        // ... everyone's; and everyone is done reading the other buffer
        if (tile + 1 < ntiles) {
            if (tile + 2 == ntiles && last_rows < kBK) dma_offsets(last_rows);
            issue_tile(tile + 1, buf ^ 1);                   // flies under the MFMAs below
        }
        if (key0 > q0w + 31) continue; // whole tile above this wave's diagonal
        const char *bufp = smem + buf * 2 * TILEB;

        // (no s_setprio around the MFMA blocks here: with two free-running workgroups per CU the per-block priority flips
        // of the r01-r06a kernels cost 1.4-1.7 % — profiles/r06b_prefill_attn_setprio_ab.jsonl; the guide's T5 says as much)
        float16_t st[2];
        
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            st[t] = float16_t{};
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const vec8_t<T> kf = *reinterpret_cast<const vec8_t<T> *>(bufp + kfo[kk] + t * 32 * ROWB);
                st[t] = mfma32(kf, qf[kk], st[t]);
            }
        }
        
        mfma_results_tie(st[0]);
        mfma_results_ready<8>(st[1]);
        if (key0 + kBK - 1 > q0w) {
            const int dmask = qrow - key0 - 4 * hf;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // key > qrow with key = key0 + 32 t + (r & 3) + 8 (r >> 2) + 4 hf: a compile-time constant against ONE
                    // per-lane value (one compare + one select per element; r01-r06a rebuilt the key first: three)
                    if ((r & 3) + 8 * (r >> 2) + t * 32 > dmask) st[t][r] = kNegBig;
                }
        }
        float mx = st[0][0];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // lazy running maximum (r06b): a row moves its maximum only when the tile raised it by more than kLazyMax (base-2
        // exponent units), so p stays <= 2^kLazyMax instead of <= 1 — 16-bit floats round p to the same RELATIVE precision at
        // either scale, l and O accumulate in fp32 — and alpha is exactly 1 for that row: the 64-multiply rescale of O below
        // then fires on the first tile or two of a row block instead of on most tiles (any of a wave's 32 rows setting a
        // record). Decided per ROW, so a row's arithmetic still depends on its own scores only (the causality test holds
        // bit for bit).
        const float m_cand = fmaxf(m_run, mx * c);
        const float m_new = m_cand - m_run > kLazyMax ? m_cand : m_run;
        const float alpha = fast_exp2(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = fast_exp2(fmaf(st[t][r], c, -m_new));
                st[t][r] = pv;
                psum += pv;
            }
        l_run = fmaf(l_run, alpha, psum);
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[dt][r] *= alpha;
        }

        
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                vec8_t<T> pb;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) pb[jj] = to_t<T>(st[t][8 * ks + jj]);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const T *vp = reinterpret_cast<const T *>(bufp + vfo[dt] + (t * 32 + 16 * ks) * ROWB);
                    const short4_t lo = lds_tr_read(vp);
                    const short4_t hi = lds_tr_read(vp + 8 * D);
                    typedef short short8_t __attribute__((ext_vector_type(8)));
                    const short8_t both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const vec8_t<T> vf = __builtin_bit_cast(vec8_t<T>, both);
                    ot[dt] = mfma32(vf, pb, ot[dt]);
                }
            }
        
    }

    // ---- epilogue: as prefill_attn_kernel (whole-row stores through LDS) ------------------------------------------------------
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) mfma_results_tie(ot[dt]);
    mfma_results_ready<8>(ot[DT - 1]);
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    constexpr int ORS = D + 8;
    static_assert(4 * 32 * ORS * 2 <= 4 * TILEB, "the four waves' O tiles must fit the K/V buffers");
    __syncthreads();                                 // every wave is done with the last K/V tile (no DMA is in flight)
    T *ow = reinterpret_cast<T *>(smem) + wave * 32 * ORS;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            typedef T vec4 __attribute__((ext_vector_type(4)));
            vec4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = to_t<T>(ot[dt][4 * r4 + e] * inv);
            *reinterpret_cast<vec4 *>(ow + l32 * ORS + dt * 32 + 8 * r4 + 4 * hf) = ov;
        }
    constexpr int CPR = D / 8, RPI = 64 / CPR;
    T *obase = static_cast<T *>(p.o) + static_cast<int64_t>(start) * p.o_tok_stride + static_cast<int64_t>(head) * D;
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
        const int row = i * RPI + lane / CPR, ch = lane % CPR;
        const vec8_t<T> v = *reinterpret_cast<const vec8_t<T> *>(ow + row * ORS + ch * 8);
        if (q0w + row < len)
            store8(obase + static_cast<int64_t>(q0w + row) * p.o_tok_stride + ch * 8, v);
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
    if ((q_tok_stride & 7) || (k_tok_stride & 7) || (v_tok_stride & 7) || (o_tok_stride & 7))
        return SWL_ERR_BAD_ARG;
    if (!swl::aligned16(q) || !swl::aligned16(k) || !swl::aligned16(v) || !swl::aligned16(o))    // (o: 16-byte row stores)
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
    p.hpw = 1;
    const int64_t units = static_cast<int64_t>(num_prefill_seqs) * num_kv_heads;
    const int64_t units_padded = (units + 7) / 8 * 8;
    const int G = num_q_heads / num_kv_heads;
    int64_t nblocks = units_padded * G * p.num_q_blocks;
    if (nblocks > 0x7fffffffLL) return SWL_ERR_UNSUPPORTED;
    dim3 grid(static_cast<unsigned>(nblocks));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // A/B switch for measurements: SWL_PREFILL_ATTN=v1 runs the register-staged kernel at head_dim 128 as well
    static const bool use_v1 = [] { const char *e = getenv("SWL_PREFILL_ATTN"); return e && e[0] == 'v' && e[1] == '1'; }();
    const bool fits32 = static_cast<int64_t>(swl::kBK) * (k_tok_stride > v_tok_stride ? k_tok_stride : v_tok_stride) * 2 + 4096 < (1ll << 32);
    SWL_DISPATCH_DTYPE(dtype, T, {
        if (head_dim == 128 && !use_v1 && fits32) {
            // heads of a kv-head side by side in a workgroup (see the kernel); A/B: SWL_PREFILL_HPW=1 keeps 128-row blocks
            static const int forced_hpw = [] { const char *e = getenv("SWL_PREFILL_HPW"); return e ? atoi(e) : 0; }();
            // measured (tools/gpu_prefill_hpw_ab.sh, profiles/r06b_prefill_attn_heads_per_group_ab.jsonl; GQA 32 / 8): two heads
            // x 64 rows +1.3 % at 32 x 1024, +4.3 % on ragged lengths, +0.6 % at 8 x 4096 over one head x 128 rows; four heads
            // x 32 rows +1.0 / +2.5 / +0.5 %
            p.hpw = G % 2 == 0 ? 2 : 1;
            if (forced_hpw == 1 || forced_hpw == 2 || forced_hpw == 4) p.hpw = G % forced_hpw == 0 ? forced_hpw : p.hpw;
            const int rows_wg = (4 / p.hpw) * 32;
            p.num_q_blocks = (max_prefill_len + rows_wg - 1) / rows_wg;
            nblocks = units_padded * (G / p.hpw) * p.num_q_blocks;
            if (nblocks > 0x7fffffffLL) return SWL_ERR_UNSUPPORTED;
            grid = dim3(static_cast<unsigned>(nblocks));
            hipLaunchKernelGGL((swl::prefill_attn_dma_kernel<T>), grid, dim3(256), 0, s, p);
        }
        else if (head_dim == 128)
            hipLaunchKernelGGL((swl::prefill_attn_kernel<T, 128>), grid, dim3(256), 0, s, p);
        else if (head_dim == 64)
            hipLaunchKernelGGL((swl::prefill_attn_kernel<T, 64>), grid, dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL((swl::prefill_attn_kernel<T, 32>), grid, dim3(256), 0, s, p);
    });
    return swl::check_launch();
}
