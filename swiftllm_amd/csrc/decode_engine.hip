// decode_engine.hip — ONE persistent launch for the transformer stack of a pure-decode step of ONE sequence (gfx950).
//
// Replaces, for a batch of one decoding sequence, the layer loop of LlamaModel._forward (swiftllm/worker/model.py:228-249)
// over LlamaTransformerLayer.forward (swiftllm/worker/layers/transformer_layer.py:31-130): embedding row, and per layer
// fused add + RMSNorm (rmsnorm.py:39-89), q/k/v projections (linear.py:3-12), rotary (rotary_emb.py:7-58), KV store
// (kvcache_mgmt.py:50-79), paged attention phase 1 + 2 (paged_attn.py:9-149), o projection, fused add + RMSNorm,
// up/gate projection, SiLU-gate (silu_and_mul.py:5-34), down projection. Rounding points are the reference's: residual
// add rounded and stored in the storage dtype, norm in fp32 with ONE rounding of x * rstd * w, every projection output
// rounded once from its fp32 sum, rotary in storage-dtype arithmetic, silu in fp32 -> storage dtype, then a storage-dtype
// product — so float16 runs here with exactly the reference's rounding points (no deferred norm).
//
// Why a persistent kernel at batch 1: a decode layer of one sequence streams 436 MB of weights (Llama-3-8B) through six
// launches whose ramps, tails and ~1.5 us boundaries leave the weight stream idle a fifth of the time (DESIGN.md 4.8:
// 94 us per layer against 67 us of stream). Here the stream never stops at an operator boundary:
//   * grid = 256 workgroups = one per CU, 4 waves each: wave 0 is the LOADER, waves 1-3 are CONSUMERS;
//   * the loader walks ONE contiguous per-CU weight stream (all layers, all projections, packed at load time in the
//     order it is consumed: swiftllm_amd/worker/weight.py pack_engine_weights) with LDS-DMA (global_load_lds_dwordx4,
//     non-temporal, 1 KiB per instruction, no VGPRs) into a ring of kRing x 16 KiB slots, three to four slots in flight
//     behind counted vmcnt waits; it depends on nothing but ring space, so it runs AHEAD across every dependency edge
//     (the in-box guide's "prefetch-credit": ~5 us of stream are on chip when the consumers come out of a hand-off);
//   * a slot = 8 rows of W x 1024 k as 16 lane-linear 1 KiB pieces (lane -> row l/8, 8 k at 64 p + 8 (l%8)); a consumer
//     wave owns whole slots (slot t -> consumer t % 3), reads W and the matching x chunk with ds_read_b128, v_dot2c into
//     fp32, reduces its 8-lane groups with DPP and leaves 8 partial sums in LDS; the op's outputs are the sums over the
//     K-chunks in K order (deterministic; no atomics);
//   * every CU owns N/256 rows of every projection, so each operator boundary is an all-to-all of a few KB. Hand-offs are
//     the guide's R2 granules: 8-byte {tag, 2 x 16-bit (or 1 x fp32) payload} written by ONE agent-scope (sc1) store,
//     swept by the consumer waves with agent-scope loads until every tag matches the (step, layer) epoch — no flags, no
//     fences, placement-independent. Every spin in the kernel is bounded: on a timeout the wave records an error code,
//     raises the workgroup's abort flag and exits; the other workgroups time out the same way or see the error word.
//     The kernel never uses s_barrier (the loader does not take part in the consumers' phases): consumers meet on an LDS
//     counter.
//   * attention of one sequence: CU (kv-head h, split s) of the KVH x S grid (S = 256 / KVH) attends the s-th chunk of
//     the context with the VALU block of paged_attn.hip (attend_block.h), the CU whose chunk holds the new position also
//     stores the rotated k / v; partials (normalised o + base-2 LSE, the reference's format) travel as fp32 granules to
//     256 mergers (one per 16 columns of the attention output), whose results are the next all-to-all.
// Algorithmic bytes per step: the projection weights of all layers (L x 436 MB for Llama-3-8B) + the KV of the context.
#include "attend_block.h"
#include "swl_common.h"

namespace swl {
namespace eng {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) u32 gu32;

constexpr int kCUs = 256;           // workgroups = CUs of one MI355X; the packed stream is laid out for exactly this
constexpr int kSlotBytes = 16384;   // one ring slot = 16 LDS-DMA instructions of 1 KiB
constexpr int kRing = 7;            // ring depth (112 KiB of the 160 KiB LDS)
constexpr int kConsumers = 3;
constexpr int kKChunk = 1024;       // k per slot
constexpr int kXCap = 16384;        // elements of the activation buffer (>= ffn_inter_dim, >= 2 * hidden)
constexpr int kPartSlots = 64;      // slots of the longest operator per CU (up/gate: 14 row groups x 4 k-chunks = 56)
constexpr int kD = 128;             // head dim the attention section is built for
constexpr int kMaxG = 4;            // q heads per kv head (LDS scratch is sized for it)
constexpr int kMaxSplits = 64;      // S = 256 / KVH <= 64
constexpr int kStateWords = 16;     // u64 words in front of the granule regions: [0] step counter, [1] error code
constexpr long long kTimeoutTicks = 5000000; // 50 ms of the 100 MHz wall clock: a hand-off takes microseconds

// error codes (the word is sticky until swl_decode_engine_reset)
constexpr u32 kErrRingWait = 1, kErrReadyWait = 2, kErrBarrier = 3, kErrGather = 4;

struct Params {
    const void *w_stream;   // [L][256][slots_per_layer][8192] packed projection weights
    const void *norms;      // [L][2][hidden]: attention norm, FFN norm
    const void *wte;        // [vocab][hidden]
    void *k_cache, *v_cache;
    const int *block_table, *input_ids, *seq_ids, *seq_lens;
    const void *cos_t, *sin_t;
    void *resid_out;        // [hidden]: the residual stream after the last layer (input of the final norm)
    u64 *ws;                // state words + granule regions (swl_decode_engine_workspace_bytes)
    long long *err_out;     // host-visible copy of the error word (may be NULL)
    u64 *dbg;               // optional: phase timestamps of 7 CUs, [7][L][16]
    int L, hidden, H, KVH, ffn, max_blocks_per_seq;
    int flags;              // reserved for A/B switches of experiments (0)
    float eps, scale_log2e;
};

// Per-layer slot layout of one CU's stream (same on every CU): qkv | o | up,gate | down.
struct Layout {
    int kj_h, kj_f;                 // k-chunks of a hidden-wide / ffn-wide input
    int r_qkv, r_o, r_ug, r_dn;     // rows per CU (r_ug = up rows + gate rows)
    int n_qkv, n_o, n_ug, n_dn;     // slots
    int s_o, s_ug, s_dn, spl;       // first slot of each op within a layer, slots per layer
    int S;                          // context splits per kv head
    // granule regions (u64 offsets into ws)
    int g_r0, g_qkv, g_part, g_oattn, g_r1, g_act, g_end;
};

__host__ __device__ inline bool make_layout(int hidden, int H, int KVH, int ffn, Layout &y) {
    const int D = kD;
    const int qkv_rows = (H + 2 * KVH) * D;
    if (hidden <= 0 || H <= 0 || KVH <= 0 || ffn <= 0 || H % KVH || H * D != hidden) return false;
    const int G = H / KVH;
    if (G > kMaxG || (G & (G - 1))) return false;
    if (kCUs % KVH || kCUs / KVH > kMaxSplits || kCUs % H) return false;
    if (hidden % kKChunk || ffn % kKChunk) return false;
    if (qkv_rows % (8 * kCUs) || hidden % (8 * kCUs) || ffn % (8 * kCUs)) return false;
    if (ffn > kXCap || 2 * hidden > kXCap) return false;
    if ((hidden / kCUs) > 16 || D % (hidden / kCUs)) return false;   // (merger scratch: <= 16 columns + the LSE per split)
    y.kj_h = hidden / kKChunk;
    y.kj_f = ffn / kKChunk;
    y.r_qkv = qkv_rows / kCUs;
    y.r_o = hidden / kCUs;
    y.r_ug = 2 * (ffn / kCUs);
    y.r_dn = hidden / kCUs;
    y.n_qkv = y.r_qkv / 8 * y.kj_h;
    y.n_o = y.r_o / 8 * y.kj_h;
    y.n_ug = y.r_ug / 8 * y.kj_h;
    y.n_dn = y.r_dn / 8 * y.kj_f;
    if (y.n_qkv > kPartSlots || y.n_o > kPartSlots || y.n_ug > kPartSlots || y.n_dn > kPartSlots) return false;
    y.s_o = y.n_qkv;
    y.s_ug = y.s_o + y.n_o;
    y.s_dn = y.s_ug + y.n_ug;
    y.spl = y.s_dn + y.n_dn;
    y.S = kCUs / KVH;
    y.g_r0 = kStateWords;
    y.g_qkv = y.g_r0 + hidden / 2;
    y.g_part = y.g_qkv + qkv_rows / 2;
    y.g_oattn = y.g_part + H * y.S * (D + 1);
    y.g_r1 = y.g_oattn + hidden / 2;
    y.g_act = y.g_r1 + hidden / 2;
    y.g_end = y.g_act + ffn / 2;
    return true;
}

// ---- LDS -----------------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) Lds {
    char ring[kRing][kSlotBytes];
    unsigned short xbuf[kXCap];         // activations of the running operator (storage dtype bits)
    float part[kPartSlots][8];          // per-slot partial sums of the running operator
    // attention scratch: rotated q, new k / v, per-wave online-softmax states, merger inputs
    unsigned short att_q[kMaxG * kD];
    unsigned short att_kv[2 * kD];
    float att_ml[kConsumers][kMaxG][2];
    float att_acc[kConsumers][kMaxG][kD];
    float mg[kMaxSplits * 17];
    unsigned short rmine[32];           // this CU's rows of the residual stream (<= 16 used)
    u32 ready;                          // slots landed (written by the loader only)
    u32 consumed[kConsumers];           // slots finished per consumer wave (written by that wave only)
    u32 bar;                            // consumer barrier arrivals
    u32 abort_flag;
};
static_assert(sizeof(Lds) <= 163840, "LDS budget of one CU");

template <typename T>
__device__ __forceinline__ T bits_to_t(unsigned short b) { return __builtin_bit_cast(T, b); }
template <typename T>
__device__ __forceinline__ unsigned short t_to_bits(T v) { return __builtin_bit_cast(unsigned short, v); }

__device__ __forceinline__ u32 lds_ld(const u32 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_st(u32 *p, u32 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void compiler_fence() { asm volatile("" ::: "memory"); }

// LDS accesses of the LOADER wave go through inline asm: hipcc knows its LDS-DMA writes are pending and would put a
// vmcnt(0) in front of any LDS access of the same wave it can see (the guide's "second __shared__ object" trap) — the
// whole ring would drain at every poll.
__device__ __forceinline__ u32 lds_addr(const void *p) {
    return static_cast<u32>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const void *)p));
}
__device__ __forceinline__ u32 lds_ld_asm(u32 addr) {
    u32 v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_st_asm(u32 addr, u32 v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

struct Ctx {
    Lds *s;
    gu64 *ws;
    long long *err_out;
    long long t0;
    int lane, cu;
};

// Bounded spinning: `n` counts the polls of one wait; every 64th poll looks at the workgroup's abort flag and the clock, every
// 2048th at the global error word (one hot address for 1024 waves: asked rarely on purpose).
__device__ __forceinline__ bool spin_expired(Ctx &c, u32 &n, u32 code) {
    ++n;
    if ((n & 63u) != 0) return false;
    bool bad = lds_ld(&c.s->abort_flag) != 0;
    if (!bad && wall_clock64() - c.t0 > kTimeoutTicks) {
        bad = true;
        if (c.lane == 0) {
            u64 expect = 0;
            __hip_atomic_compare_exchange_strong(c.ws + 1, &expect, static_cast<u64>(code) | (static_cast<u64>(c.cu) << 8),
                                                 __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!bad && (n & 2047u) == 0)
        bad = __hip_atomic_load(c.ws + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    if (bad) {
        lds_st(&c.s->abort_flag, 1u);
        if (c.lane == 0 && c.err_out) {
            const u64 e = __hip_atomic_load(c.ws + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *c.err_out = static_cast<long long>(e ? e : code);
        }
    }
    return bad;
}

// ---- the loader wave -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void loader_main(const Params &p, const Layout &y, Ctx &c) {
    Lds &s = *c.s;
    const int lane = c.lane;
    const long long layer_stride = static_cast<long long>(kCUs) * y.spl * kSlotBytes;
    const char *base = static_cast<const char *>(p.w_stream) + static_cast<long long>(c.cu) * y.spl * kSlotBytes + lane * 16;
    const u32 a_ready = lds_addr(&s.ready);
    const u32 a_cons = lds_addr(&s.consumed[0]);
    // loop invariants in registers: a scratch / kernarg VECTOR load inside the loop would put a vmcnt(0) in front of itself and
    // drain the ring's in-flight fills every slot (seen: the stream at half rate)
    const int spl = y.spl;
    const int total = p.L * spl;
    const long long layer_skip = layer_stride - static_cast<long long>(spl) * kSlotBytes;
    int ring_pos = 0;
    int q = 0;             // slot within the layer
    int published = 0;     // value of s.ready (slots landed, monotone)
    const char *src = base;
    for (int t = 0; t < total; ++t) {
        if (t >= kRing) {   // the slot this one overwrites (t - kRing) must have been consumed
            const int u = t - kRing;
            const u32 need = static_cast<u32>(u / kConsumers + 1);
            const u32 addr = a_cons + 4u * static_cast<u32>(u % kConsumers);
            if (lds_ld_asm(addr) < need) {
                // ring full (the consumers are at a hand-off): nothing to issue, so hand them everything in flight
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (published < t) {
                    published = t;
                    lds_st_asm(a_ready, static_cast<u32>(t));
                }
                u32 n = 0;
                while (lds_ld_asm(addr) < need) {
                    if (spin_expired(c, n, kErrRingWait)) return;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
        }
        char *dst = &s.ring[ring_pos][0];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i * 1024),
                                             (__attribute__((address_space(3))) void *)(dst + i * 1024), 16, 0, 2 /* nt */);
        // counted wait: everything but the three newest slots has landed
        asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        if (t - 2 > published) {
            published = t - 2;
            lds_st_asm(a_ready, static_cast<u32>(published));
        }
        ring_pos = ring_pos + 1 == kRing ? 0 : ring_pos + 1;
        src += kSlotBytes;
        if (++q == spl) {
            q = 0;
            src += layer_skip;
        }
    }
    asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    if (total - 2 > published) lds_st_asm(a_ready, static_cast<u32>(total - 2));
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    if (total - 1 > published) lds_st_asm(a_ready, static_cast<u32>(total - 1));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_st_asm(a_ready, static_cast<u32>(total));
}

// ---- consumer-side primitives ----------------------------------------------------------------------------------------
// The three consumer waves meet here. LDS operations of one wave execute in order, so everything a wave wrote to LDS
// before its arrival is visible to whoever sees the arrival.
__device__ __forceinline__ bool cbar(Ctx &c, u32 &gen) {
    compiler_fence();
    gen += kConsumers;
    if (c.lane == 0) __hip_atomic_fetch_add(&c.s->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    u32 n = 0;
    while (lds_ld(&c.s->bar) < gen) {
        if (spin_expired(c, n, kErrBarrier)) return false;
    }
    compiler_fence();
    return true;
}

__device__ __forceinline__ void put_granule(gu64 *g, int idx, u32 tag, u32 data) {
    __hip_atomic_store(g + idx, (static_cast<u64>(tag) << 32) | data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wave sweeps n <= 64 * NV granules — granule i of the sweep lives at g[index(i)] — until every tag matches, re-reading
// only what has not arrived yet, then hands payload i to store(i, payload). All loads of a pass are in flight together.
template <int NV, typename IndexFn, typename StoreFn>
__device__ __forceinline__ bool sweep(Ctx &c, const gu64 *g, int n, u32 tag, IndexFn index, StoreFn store) {
    u64 pending = 0;
#pragma unroll
    for (int k = 0; k < NV; ++k)
        if (k * 64 + c.lane < n) pending |= 1ull << k;
    u32 spins = 0;
    for (;;) {
        u64 x[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if (pending & (1ull << k)) x[k] = __hip_atomic_load(g + index(k * 64 + c.lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if ((pending & (1ull << k)) && static_cast<u32>(x[k] >> 32) == tag) {
                store(k * 64 + c.lane, static_cast<u32>(x[k]));     // (its readers wait behind a consumer barrier)
                pending &= ~(1ull << k);
            }
        if (!__any(pending != 0)) return true;
        if (spin_expired(c, spins, kErrGather)) return false;
        __builtin_amdgcn_s_sleep(2);
    }
}

// Contiguous run of n <= 1024 granules -> dst[0, n)
__device__ __forceinline__ bool sweep_chunk(Ctx &c, const gu64 *g, int n, u32 tag, u32 *dst) {
    return sweep<16>(c, g, n, tag, [](int i) { return i; }, [dst](int i, u32 v) { dst[i] = v; });
}

// All-gather of n granules (n % 64 == 0) into dst: the three consumer waves take contiguous thirds (in 64-granule units), each
// requests its whole share at once (NV >= the loads per lane of one wave's share).
template <int NV>
__device__ __forceinline__ bool gather(Ctx &c, int cw, const gu64 *g, int n, u32 tag, u32 *dst) {
    const int per = ((n / 64 + kConsumers - 1) / kConsumers) * 64;
    const int begin = cw * per;
    const int count = min(per, n - begin);
    if (count <= 0) return true;
    u32 *d = dst + begin;
    return sweep<NV>(c, g + begin, count, tag, [](int i) { return i; }, [d](int i, u32 v) { d[i] = v; });
}

// The projections' inner loop: this wave's slots of the operator whose first global slot index is `s0`.
// x: the operator's input in LDS (storage dtype); KJ: k-chunks per row group; slot q of the op -> (group q / KJ, chunk q % KJ).
template <typename T>
__device__ __forceinline__ bool gemv_slots(Ctx &c, int cw, int s0, int nslots, int KJ, const unsigned short *x) {
    Lds &s = *c.s;
    const int lane = c.lane;
    int t = s0 + ((cw - s0 % kConsumers) + kConsumers) % kConsumers;
    int done = t / kConsumers;          // slots this wave has finished before t (it owns t' = cw, cw + 3, ...)
    for (; t < s0 + nslots; t += kConsumers) {
        u32 n = 0;
        while (lds_ld(&s.ready) < static_cast<u32>(t + 1)) {
            if (spin_expired(c, n, kErrReadyWait)) return false;
        }
        compiler_fence();
        const int q = t - s0;
        const int j = q % KJ;
        const char *slot = &s.ring[t % kRing][0] + lane * 16;
        const unsigned short *xb = x + j * kKChunk + (lane & 7) * 8;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int pc = 0; pc < 16; pc += 4) {
            const vec8_t<T> w0 = *reinterpret_cast<const vec8_t<T> *>(slot + (pc + 0) * 1024);
            const vec8_t<T> w1 = *reinterpret_cast<const vec8_t<T> *>(slot + (pc + 1) * 1024);
            const vec8_t<T> w2 = *reinterpret_cast<const vec8_t<T> *>(slot + (pc + 2) * 1024);
            const vec8_t<T> w3 = *reinterpret_cast<const vec8_t<T> *>(slot + (pc + 3) * 1024);
            const vec8_t<T> x0 = *reinterpret_cast<const vec8_t<T> *>(xb + (pc + 0) * 64);
            const vec8_t<T> x1 = *reinterpret_cast<const vec8_t<T> *>(xb + (pc + 1) * 64);
            const vec8_t<T> x2 = *reinterpret_cast<const vec8_t<T> *>(xb + (pc + 2) * 64);
            const vec8_t<T> x3 = *reinterpret_cast<const vec8_t<T> *>(xb + (pc + 3) * 64);
            a0 = dot8<T>(w0, x0, a0);
            a1 = dot8<T>(w1, x1, a1);
            a2 = dot8<T>(w2, x2, a2);
            a3 = dot8<T>(w3, x3, a3);
        }
        float acc = (a0 + a1) + (a2 + a3);
        acc = group_allreduce_sum<8>(acc);
        if ((lane & 7) == 0) s.part[q][lane >> 3] = acc;
        compiler_fence();       // (the dot products above consumed every ds_read of the slot)
        ++done;
        if (lane == 0) lds_st(&s.consumed[cw], static_cast<u32>(done));
    }
    return true;
}

// Sum of an output row's partials over the k-chunks, in k order.
__device__ __forceinline__ float row_sum(const Lds &s, int row, int KJ) {
    const int g = row >> 3, r = row & 7;
    float a = 0.f;
    for (int j = 0; j < KJ; ++j) a += s.part[g * KJ + j][r];
    return a;
}

// r (storage dtype, n elements at xr) -> x = round(r * rstd * w) at xo; every wave computes the same sum of squares
// (same order on every CU: the normalised vector is bit-identical chip-wide). The norm weights of this thread's chunks
// were requested before the gather that produced r (norm_prefetch): no global round trip on the critical path.
constexpr int kNormIters = 6;       // hidden <= 8192: chunks of 8 elements, 192 consumer threads
template <typename T>
__device__ __forceinline__ void norm_prefetch(vec8_t<T> (&wv)[kNormIters], int ct, const T *w, int n) {
#pragma unroll
    for (int it = 0; it < kNormIters; ++it) {
        const int i = ct * 8 + it * kConsumers * 64 * 8;
        if (i < n) wv[it] = load8(w + i);
    }
}
template <typename T>
__device__ __forceinline__ void norm_from_lds(Ctx &c, int ct, const unsigned short *xr, unsigned short *xo,
                                              const vec8_t<T> (&wv)[kNormIters], int n, float eps) {
    float ss = 0.f;
    for (int i = c.lane * 8; i < n; i += 64 * 8) {
        const vec8_t<T> v = *reinterpret_cast<const vec8_t<T> *>(xr + i);
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = fmaf(to_f(v[e]), to_f(v[e]), ss);
    }
    ss = wave_allreduce_sum(ss);
    const float rstd = 1.0f / sqrtf(ss / static_cast<float>(n) + eps);
#pragma unroll
    for (int it = 0; it < kNormIters; ++it) {
        const int i = ct * 8 + it * kConsumers * 64 * 8;
        if (i < n) {
            const vec8_t<T> v = *reinterpret_cast<const vec8_t<T> *>(xr + i);
            vec8_t<T> o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = to_t<T>(to_f(v[e]) * rstd * to_f(wv[it][e]));
            *reinterpret_cast<vec8_t<T> *>(xo + i) = o;
        }
    }
}

template <typename T>
__device__ __forceinline__ u32 pack2(T a, T b) {
    return static_cast<u32>(t_to_bits(a)) | (static_cast<u32>(t_to_bits(b)) << 16);
}

// ---- the consumer waves: one decode step -------------------------------------------------------------------------------
constexpr int kStamps = 16;         // debug stamps per layer and traced CU
constexpr int kTraceStride = 37;    // traced CUs: 0, 37, 74, ... (7 of them, on 7 different XCDs)
constexpr int kTraced = 7;

template <typename T, int G>
__device__ __forceinline__ void consumer_main(const Params &p, const Layout &y, Ctx &c, int cw) {
    constexpr int D = kD;
    using Tile = DecodeTile<T, D, G>;
    constexpr int LPT = Tile::LPT, TPI = Tile::TPI, NI = Tile::NI;
    Lds &s = *c.s;
    const int lane = c.lane;
    const int ct = cw * 64 + lane;                  // thread index among the consumers
    const int cu = c.cu;
    gu64 *ws = c.ws;
    u32 gen = 0;
    const int hidden = p.hidden, ffn = p.ffn, H = p.H, KVH = p.KVH, S = y.S;

    const u32 step = static_cast<u32>(__hip_atomic_load(ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const u32 tag0 = step * static_cast<u32>(p.L + 1);

    const int len = p.seq_lens[0];
    const int pos = len - 1;
    const int64_t seq_id = p.seq_ids[0];
    const int *bt = p.block_table + seq_id * p.max_blocks_per_seq;
    // attention role of this CU
    const int kvh = cu / S, sp = cu % S;
    const bool att_active = cu < KVH * S;
    const int chunk_tok = ((len + S - 1) / S + kBlk - 1) / kBlk * kBlk;
    const int tok_begin = sp * chunk_tok;
    const int tok_end = min(len, tok_begin + chunk_tok);
    const bool has = att_active && tok_begin < len;
    const bool owner = has && tok_end == len;
    const int blk_end = (tok_end + kBlk - 1) / kBlk;
    const int b_first = tok_begin / kBlk + cw;      // this wave's first KV block
    const bool has_blk = has && b_first < blk_end;
    // merger role
    const int dpc = hidden / kCUs;                  // attention-output columns per CU
    const int cph = D / dpc;                        // CUs per q head
    const int mh = cu / cph, md0 = (cu % cph) * dpc;

    unsigned short *xa = &s.xbuf[0];                // residual stream / activations of the wide operator
    unsigned short *xb = &s.xbuf[hidden];           // normalised activations / attention output
    u32 *xa32 = reinterpret_cast<u32 *>(xa);
    u32 *xb32 = reinterpret_cast<u32 *>(xb);
    const T *kc = static_cast<const T *>(p.k_cache);
    const T *vc = static_cast<const T *>(p.v_cache);
    const int chunk = lane % LPT, row = lane / LPT;

    // per-step constants of the attention section, fetched once: the rope row of the position, this wave's first block id
    vec8_t<T> cosv = {}, sinv = {};
    int64_t phys_first = 0, phys_pos = 0;
    if (has) {
        cosv = load8(static_cast<const T *>(p.cos_t) + static_cast<int64_t>(pos) * (D / 2) + (chunk & 7) * 8);
        sinv = load8(static_cast<const T *>(p.sin_t) + static_cast<int64_t>(pos) * (D / 2) + (chunk & 7) * 8);
        if (has_blk) phys_first = bt[b_first];
        if (owner) phys_pos = bt[pos / kBlk];
    }
    const int64_t blk_pitch = static_cast<int64_t>(p.L) * KVH;

    const int trace_slot = (p.dbg != nullptr && ct == 0 && cu % kTraceStride == 0 && cu / kTraceStride < kTraced)
                               ? cu / kTraceStride : -1;
    u64 *dbg = trace_slot >= 0 ? p.dbg + static_cast<int64_t>(trace_slot) * p.L * kStamps : nullptr;
#define SWL_STAMP(k) do { if (dbg) dbg[layer * kStamps + (k)] = wall_clock64(); } while (0)

    int slot0 = 0;          // global slot index of the running layer's first slot
    for (int layer = 0; layer < p.L; ++layer, slot0 += y.spl) {
        const u32 tag = tag0 + static_cast<u32>(layer) + 1u;
        const T *norm_w = static_cast<const T *>(p.norms) + static_cast<int64_t>(layer) * 2 * hidden;
        const int64_t layer_head = static_cast<int64_t>(layer) * KVH + kvh;
        SWL_STAMP(0);

        // ---- P0: the layer input r -> xa; x = rmsnorm(r) * w_attn -> xb ------------------------------------------
        vec8_t<T> wn[kNormIters];
        norm_prefetch<T>(wn, ct, norm_w, hidden);
        if (layer == 0) {
            const T *row_p = static_cast<const T *>(p.wte) + static_cast<int64_t>(p.input_ids[0]) * hidden;
            for (int i = ct * 8; i < hidden; i += kConsumers * 64 * 8)
                *reinterpret_cast<vec8_t<T> *>(xa + i) = load8(row_p + i);
        } else {
            if (!gather<22>(c, cw, ws + y.g_r0, hidden / 2, tag - 1u, xa32)) return;   // published by the previous layer
        }
        if (!cbar(c, gen)) return;
        SWL_STAMP(1);
        norm_from_lds<T>(c, ct, xa, xb, wn, hidden, p.eps);
        if (ct < y.r_o) s.rmine[ct] = xa[cu * y.r_o + ct];
        // the first KV block of this wave does not depend on this layer's q: request it now, consume it in P2
        vec8_t<T> Kp[NI], Vp[NI];
        if (has_blk) {
            const int64_t base = (phys_first * blk_pitch + layer_head) * (kBlk * D) + lane * 8;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                Kp[i] = load8_nt(kc + base + i * 512);
                Vp[i] = load8_nt(vc + base + i * 512);
            }
        }
        if (!cbar(c, gen)) return;
        SWL_STAMP(2);

        // ---- P1: q/k/v rows of this CU ---------------------------------------------------------------------------
        if (!gemv_slots<T>(c, cw, slot0, y.n_qkv, y.kj_h, xb)) return;
        if (!cbar(c, gen)) return;
        SWL_STAMP(3);
        if (ct < y.r_qkv / 2) {
            const T a = to_t<T>(row_sum(s, 2 * ct, y.kj_h)), b = to_t<T>(row_sum(s, 2 * ct + 1, y.kj_h));
            put_granule(ws + y.g_qkv, cu * (y.r_qkv / 2) + ct, tag, pack2<T>(a, b));
        }
        SWL_STAMP(4);

        // ---- P2: attention -----------------------------------------------------------------------------------------
        if (has) {
            // q heads kvh*G .. +G (contiguous), the kv head's new k and v
            bool ok = true;
            if (cw == 0) ok = sweep_chunk(c, ws + y.g_qkv + (kvh * G * D) / 2, G * D / 2, tag, reinterpret_cast<u32 *>(s.att_q));
            else if (cw == 1) ok = sweep_chunk(c, ws + y.g_qkv + ((H + kvh) * D) / 2, D / 2, tag, reinterpret_cast<u32 *>(s.att_kv));
            else ok = sweep_chunk(c, ws + y.g_qkv + ((H + KVH + kvh) * D) / 2, D / 2, tag, reinterpret_cast<u32 *>(s.att_kv + D));
            if (!ok) return;
        }
        if (!cbar(c, gen)) return;
        SWL_STAMP(5);
        {
            float m[G], l[G], acc[G][8];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                m[g] = kNegBig;
                l[g] = 0.f;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) acc[g][jj] = 0.f;
            }
            if (has) {
                // rotate-half rotary in registers, storage-dtype arithmetic (rotary_emb.py:26-42): a lane holds elements
                // [8 chunk, +8) of a head; its partner half sits 8 chunks away; lanes of the upper half produce x1'
                const bool hi = chunk >= 8;
                auto rotated = [&](const unsigned short *head) {
                    const vec8_t<T> own = *reinterpret_cast<const vec8_t<T> *>(head + chunk * 8);
                    const vec8_t<T> oth = *reinterpret_cast<const vec8_t<T> *>(head + (chunk ^ 8) * 8);
                    vec8_t<T> r;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        r[e] = hi ? add_t<T>(mul_t<T>(oth[e], sinv[e]), mul_t<T>(own[e], cosv[e]))
                                  : sub_t<T>(mul_t<T>(own[e], cosv[e]), mul_t<T>(oth[e], sinv[e]));
                    return r;
                };
                vec8_t<T> qv[G];
#pragma unroll
                for (int g = 0; g < G; ++g) qv[g] = rotated(&s.att_q[g * D]);
                const vec8_t<T> knew = rotated(&s.att_kv[0]);
                const vec8_t<T> vnew = *reinterpret_cast<const vec8_t<T> *>(&s.att_kv[D + chunk * 8]);
                if (owner && cw == 0 && row == 0) {
                    // the new token's rotated k and its v go to the pool (kvcache_mgmt.py:50-79)
                    const int64_t off = (phys_pos * blk_pitch + layer_head) * (kBlk * D) + (pos % kBlk) * D + chunk * 8;
                    store8(const_cast<T *>(kc) + off, knew);
                    store8(const_cast<T *>(vc) + off, vnew);
                }
                for (int b = b_first; b < blk_end; b += kConsumers) {
                    vec8_t<T> Kv[NI], Vv[NI];
                    if (b == b_first) {
#pragma unroll
                        for (int i = 0; i < NI; ++i) {
                            Kv[i] = Kp[i];
                            Vv[i] = Vp[i];
                        }
                    } else {
                        const int64_t base = (static_cast<int64_t>(bt[b]) * blk_pitch + layer_head) * (kBlk * D) + lane * 8;
#pragma unroll
                        for (int i = 0; i < NI; ++i) {
                            Kv[i] = load8_nt(kc + base + i * 512);
                            Vv[i] = load8_nt(vc + base + i * 512);
                        }
                    }
                    const int tok0 = b * kBlk;
                    if (b == pos / kBlk) {  // the pool read of the new token raced with its store: take it from registers
#pragma unroll
                        for (int i = 0; i < NI; ++i)
                            if (tok0 + i * TPI + row == pos) {
                                Kv[i] = knew;
                                Vv[i] = vnew;
                            }
                    }
                    attend_block<T, D, G>(qv, Kv, Vv, m, l, acc, p.scale_log2e, tok0, row, len, tok0 + kBlk > len);
                }
                // merge the TPI rows of the wave
#pragma unroll
                for (int mask = LPT; mask < 64; mask <<= 1) {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const float m2 = __shfl_xor(m[g], mask, 64);
                        const float l2 = __shfl_xor(l[g], mask, 64);
                        const float M = fmaxf(m[g], m2);
                        const float w1 = fast_exp2((m[g] - M) * p.scale_log2e);
                        const float w2 = fast_exp2((m2 - M) * p.scale_log2e);
                        l[g] = l[g] * w1 + l2 * w2;
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) {
                            const float a2 = __shfl_xor(acc[g][jj], mask, 64);
                            acc[g][jj] = acc[g][jj] * w1 + a2 * w2;
                        }
                        m[g] = M;
                    }
                }
                if (row == 0) {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        if (chunk == 0) {
                            s.att_ml[cw][g][0] = m[g];
                            s.att_ml[cw][g][1] = l[g];
                        }
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) s.att_acc[cw][g][chunk * 8 + jj] = acc[g][jj];
                    }
                }
            }
        }
        if (!cbar(c, gen)) return;
        SWL_STAMP(6);
        if (att_active) {
            // merge the three waves; publish the partial of (head, split): normalised o + base-2 log-sum-exp
            for (int oidx = ct; oidx < G * (D + 1); oidx += kConsumers * 64) {
                const int g = oidx / (D + 1), d = oidx % (D + 1);
                float out = d == D ? kNegBig : 0.f;
                if (has) {
                    float M = s.att_ml[0][g][0];
#pragma unroll
                    for (int w = 1; w < kConsumers; ++w) M = fmaxf(M, s.att_ml[w][g][0]);
                    float Lsum = 0.f, A = 0.f;
#pragma unroll
                    for (int w = 0; w < kConsumers; ++w) {
                        const float wgt = fast_exp2((s.att_ml[w][g][0] - M) * p.scale_log2e);
                        Lsum = fmaf(s.att_ml[w][g][1], wgt, Lsum);
                        if (d < D) A = fmaf(s.att_acc[w][g][d], wgt, A);
                    }
                    out = d == D ? fast_log2(Lsum) + M * p.scale_log2e : A / Lsum;
                }
                put_granule(ws + y.g_part, ((kvh * G + g) * S + sp) * (D + 1) + d, tag, __float_as_uint(out));
            }
        }
        SWL_STAMP(7);
        // mergers: head mh, columns md0 .. md0 + dpc of the attention output
        if (cw == 0) {
            const int per = dpc + 1;
            const gu64 *pg = ws + y.g_part + mh * S * (D + 1);
            float *mg = s.mg;
            const int dpc_ = dpc, md0_ = md0;
            if (!sweep<(kMaxSplits * 17 + 63) / 64>(
                    c, pg, S * per, tag,
                    [per, dpc_, md0_](int i) { const int e = i % per; return (i / per) * (kD + 1) + (e < dpc_ ? md0_ + e : kD); },
                    [mg](int i, u32 v) { mg[i] = __uint_as_float(v); }))
                return;
        }
        if (!cbar(c, gen)) return;
        SWL_STAMP(8);
        if (cw == 0) {
            // LSE-weighted sum over the S splits (paged_attn.py:128-149), one wave: lane -> (split l % 32 [+ 32 ...], half l / 32
            // of the dpc columns); maxima and sums cross the 32 lanes of a half with DPP + one shuffle
            const int per = dpc + 1, hd = dpc / 2;
            const int sl = lane & 31, hf = lane >> 5;
            float M = kNegBig;
            for (int sp2 = sl; sp2 < S; sp2 += 32) M = fmaxf(M, s.mg[sp2 * per + dpc]);
            M = fmaxf(M, dpp_mov<kDppQuadXor1>(M));
            M = fmaxf(M, dpp_mov<kDppQuadXor2>(M));
            M = fmaxf(M, dpp_mov<kDppRowHalfMirror>(M));
            M = fmaxf(M, dpp_mov<kDppRowMirror>(M));
            M = fmaxf(M, __shfl_xor(M, 16, 64));
            float W = 0.f, o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = 0.f;
            for (int sp2 = sl; sp2 < S; sp2 += 32) {
                const float wgt = fast_exp2(s.mg[sp2 * per + dpc] - M);
                W += wgt;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (e < hd) o[e] = fmaf(wgt, s.mg[sp2 * per + hf * hd + e], o[e]);
            }
            W = group_allreduce_sum<16>(W);
            W += __shfl_xor(W, 16, 64);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = group_allreduce_sum<16>(o[e]);
                o[e] += __shfl_xor(o[e], 16, 64);
            }
            if (sl == 0) {
#pragma unroll
                for (int e = 0; e < 8; e += 2)
                    if (e < hd)
                        put_granule(ws + y.g_oattn, (mh * D + md0 + hf * hd + e) / 2, tag,
                                    pack2<T>(to_t<T>(o[e] / W), to_t<T>(o[e + 1] / W)));
            }
        }

        // ---- P3: o projection + residual add -----------------------------------------------------------------------
        if (!gather<22>(c, cw, ws + y.g_oattn, hidden / 2, tag, xb32)) return;
        if (!cbar(c, gen)) return;
        SWL_STAMP(9);
        if (!gemv_slots<T>(c, cw, slot0 + y.s_o, y.n_o, y.kj_h, xb)) return;
        if (!cbar(c, gen)) return;
        if (ct < y.r_o / 2) {
            const T a = add_t<T>(bits_to_t<T>(s.rmine[2 * ct]), to_t<T>(row_sum(s, 2 * ct, y.kj_h)));
            const T b = add_t<T>(bits_to_t<T>(s.rmine[2 * ct + 1]), to_t<T>(row_sum(s, 2 * ct + 1, y.kj_h)));
            put_granule(ws + y.g_r1, cu * (y.r_o / 2) + ct, tag, pack2<T>(a, b));
        }
        SWL_STAMP(10);

        // ---- P4: FFN norm, up/gate projection, SiLU-gate -----------------------------------------------------------
        norm_prefetch<T>(wn, ct, norm_w + hidden, hidden);
        if (!gather<22>(c, cw, ws + y.g_r1, hidden / 2, tag, xa32)) return;
        if (!cbar(c, gen)) return;
        SWL_STAMP(11);
        norm_from_lds<T>(c, ct, xa, xb, wn, hidden, p.eps);
        if (ct < y.r_dn) s.rmine[ct] = xa[cu * y.r_dn + ct];
        if (!cbar(c, gen)) return;
        SWL_STAMP(12);
        if (!gemv_slots<T>(c, cw, slot0 + y.s_ug, y.n_ug, y.kj_h, xb)) return;
        if (!cbar(c, gen)) return;
        {
            const int half = y.r_ug / 2;            // up rows, then as many gate rows
            if (ct < half / 2) {
                T act[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const T up = to_t<T>(row_sum(s, 2 * ct + e, y.kj_h));
                    const float gt = to_f(to_t<T>(row_sum(s, half + 2 * ct + e, y.kj_h)));
                    act[e] = mul_t<T>(up, to_t<T>(gt / (1.0f + expf(-gt))));
                }
                put_granule(ws + y.g_act, cu * (half / 2) + ct, tag, pack2<T>(act[0], act[1]));
            }
        }
        SWL_STAMP(13);

        // ---- P5: down projection + residual add --------------------------------------------------------------------
        if (!gather<44>(c, cw, ws + y.g_act, ffn / 2, tag, xa32)) return;
        if (!cbar(c, gen)) return;
        SWL_STAMP(14);
        if (!gemv_slots<T>(c, cw, slot0 + y.s_dn, y.n_dn, y.kj_f, xa)) return;
        if (!cbar(c, gen)) return;
        if (ct < y.r_dn / 2) {
            const T a = add_t<T>(bits_to_t<T>(s.rmine[2 * ct]), to_t<T>(row_sum(s, 2 * ct, y.kj_f)));
            const T b = add_t<T>(bits_to_t<T>(s.rmine[2 * ct + 1]), to_t<T>(row_sum(s, 2 * ct + 1, y.kj_f)));
            if (layer + 1 < p.L) {
                put_granule(ws + y.g_r0, cu * (y.r_dn / 2) + ct, tag, pack2<T>(a, b));
            } else {
                T *ro = static_cast<T *>(p.resid_out) + cu * y.r_dn + 2 * ct;
                ro[0] = a;
                ro[1] = b;
            }
        }
        SWL_STAMP(15);
        // (no barrier here: part[] / rmine[] are next written behind the first consumer barrier of the next layer, which
        // the one wave that reads them above — wave 0 — reaches only after it is done with them)
    }
#undef SWL_STAMP
    // the step is done for this CU; CU 0 advances the epoch (every CU read it before publishing anything CU 0 needed)
    if (cu == 0 && ct == 0) __hip_atomic_store(ws, static_cast<u64>(step + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T, int G>
__global__ __launch_bounds__(256, 1) void decode_engine_kernel(Params p, Layout y) {
    __shared__ Lds s;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Ctx c;
    c.s = &s;
    c.ws = (gu64 *)p.ws;
    c.err_out = p.err_out;
    c.t0 = wall_clock64();
    c.lane = lane;
    c.cu = blockIdx.x;
    if (threadIdx.x == 0) {
        s.ready = 0;
        s.consumed[0] = s.consumed[1] = s.consumed[2] = 0;
        s.bar = 0;
        s.abort_flag = 0;
    }
    __syncthreads();    // the only workgroup barrier: before the roles split
    const u64 poisoned = __hip_atomic_load(c.ws + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.err_out) *p.err_out = static_cast<long long>(poisoned);
    if (poisoned) return;       // a previous step failed: nothing runs until swl_decode_engine_reset
    if (wave == 0) loader_main(p, y, c);
    else consumer_main<T, G>(p, y, c, wave - 1);
}

} // namespace eng
} // namespace swl

using swl::eng::Layout;
using swl::eng::Params;

extern "C" int swl_decode_engine_supported(int32_t hidden, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
                                           int32_t ffn_inter_dim, int32_t num_cus) {
    Layout y;
    return (head_dim == swl::eng::kD && num_cus == swl::eng::kCUs &&
            swl::eng::make_layout(hidden, num_q_heads, num_kv_heads, ffn_inter_dim, y)) ? 1 : 0;
}

extern "C" int swl_decode_engine_slots_per_layer(int32_t hidden, int32_t num_q_heads, int32_t num_kv_heads,
                                                 int32_t ffn_inter_dim) {
    Layout y;
    return swl::eng::make_layout(hidden, num_q_heads, num_kv_heads, ffn_inter_dim, y) ? y.spl : 0;
}

extern "C" size_t swl_decode_engine_workspace_bytes(int32_t hidden, int32_t num_q_heads, int32_t num_kv_heads,
                                                    int32_t ffn_inter_dim) {
    Layout y;
    if (!swl::eng::make_layout(hidden, num_q_heads, num_kv_heads, ffn_inter_dim, y)) return 0;
    return static_cast<size_t>(y.g_end) * 8;
}

extern "C" int swl_decode_engine_reset(void *workspace, size_t workspace_bytes, swl_stream_t stream) {
    if (!workspace || workspace_bytes < swl::eng::kStateWords * 8 || !swl::aligned16(workspace)) return SWL_ERR_BAD_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(workspace, 0, workspace_bytes, st) != hipSuccess) return SWL_ERR_RUNTIME;
    const unsigned long long one = 1;   // step counter: tags are never 0
    if (hipMemcpyAsync(workspace, &one, sizeof(one), hipMemcpyHostToDevice, st) != hipSuccess) return SWL_ERR_RUNTIME;
    if (hipStreamSynchronize(st) != hipSuccess) return SWL_ERR_RUNTIME;  // `one` lives on this stack frame
    return SWL_OK;
}

extern "C" int swl_decode_engine_step(void *resid_out, const void *w_stream, const void *norms, const void *wte,
                                      void *k_cache, void *v_cache, const int32_t *block_table, const int32_t *input_ids,
                                      const int32_t *seq_ids, const int32_t *seq_lens, const void *cos_table,
                                      const void *sin_table, void *workspace, size_t workspace_bytes, int64_t *err_out,
                                      uint64_t *debug_stamps, int32_t num_layers, int32_t hidden, int32_t num_q_heads,
                                      int32_t num_kv_heads, int32_t head_dim, int32_t ffn_inter_dim,
                                      int32_t max_blocks_per_seq, float eps, float softmax_scale, int32_t flags,
                                      int32_t dtype, swl_stream_t stream) {
    Layout y;
    if (num_layers <= 0 || head_dim != swl::eng::kD ||
        !swl::eng::make_layout(hidden, num_q_heads, num_kv_heads, ffn_inter_dim, y))
        return SWL_ERR_UNSUPPORTED;
    if (!resid_out || !w_stream || !norms || !wte || !k_cache || !v_cache || !block_table || !input_ids || !seq_ids ||
        !seq_lens || !cos_table || !sin_table || !workspace || max_blocks_per_seq <= 0)
        return SWL_ERR_BAD_ARG;
    if (workspace_bytes < static_cast<size_t>(y.g_end) * 8 || !swl::aligned16(workspace) || !swl::aligned16(w_stream) ||
        !swl::aligned16(norms) || !swl::aligned16(wte) || !swl::aligned16(k_cache) || !swl::aligned16(v_cache))
        return SWL_ERR_BAD_ARG;
    Params p;
    p.w_stream = w_stream;
    p.norms = norms;
    p.wte = wte;
    p.k_cache = k_cache;
    p.v_cache = v_cache;
    p.block_table = block_table;
    p.input_ids = input_ids;
    p.seq_ids = seq_ids;
    p.seq_lens = seq_lens;
    p.cos_t = cos_table;
    p.sin_t = sin_table;
    p.resid_out = resid_out;
    p.ws = static_cast<swl::eng::u64 *>(workspace);
    p.err_out = reinterpret_cast<long long *>(err_out);
    p.dbg = reinterpret_cast<swl::eng::u64 *>(debug_stamps);
    p.L = num_layers;
    p.hidden = hidden;
    p.H = num_q_heads;
    p.KVH = num_kv_heads;
    p.ffn = ffn_inter_dim;
    p.max_blocks_per_seq = max_blocks_per_seq;
    p.flags = flags;
    p.eps = eps;
    p.scale_log2e = softmax_scale * 1.4426950408889634f;
    const int G = num_q_heads / num_kv_heads;
    hipStream_t st = static_cast<hipStream_t>(stream);
    SWL_DISPATCH_DTYPE(dtype, T, {
        switch (G) {
        case 1: hipLaunchKernelGGL((swl::eng::decode_engine_kernel<T, 1>), dim3(swl::eng::kCUs), dim3(256), 0, st, p, y); break;
        case 2: hipLaunchKernelGGL((swl::eng::decode_engine_kernel<T, 2>), dim3(swl::eng::kCUs), dim3(256), 0, st, p, y); break;
        case 4: hipLaunchKernelGGL((swl::eng::decode_engine_kernel<T, 4>), dim3(swl::eng::kCUs), dim3(256), 0, st, p, y); break;
        default: return SWL_ERR_UNSUPPORTED;
        }
    });
    return swl::check_launch();
}
