// swl_common.h — device-side helpers shared by the gfx950 kernels of libswiftllm_hip.so.
// CDNA4 only: wave = 64 lanes, 16-byte (8 x 16-bit) vector accesses everywhere, DPP row reductions.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/swiftllm_hip.h"

namespace swl {

using f16 = _Float16;
using bf16 = __bf16;

constexpr int kWave = 64;
// Finite stand-in for -inf in online-softmax running maxima: (-inf) - (-inf) would be NaN.
constexpr float kNegBig = -1e30f;

// 8 x 16-bit elements = one 16-byte global/LDS access per lane.
template <typename T>
struct Vec8 {
    typedef T type __attribute__((ext_vector_type(8)));
};
template <typename T>
using vec8_t = typename Vec8<T>::type;
template <typename T>
struct Vec2 {
    typedef T type __attribute__((ext_vector_type(2)));
};
template <typename T>
using vec2_t = typename Vec2<T>::type;

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

template <typename T>
__device__ __forceinline__ vec8_t<T> load8(const T *p) {
    return *reinterpret_cast<const vec8_t<T> *>(p);
}
template <typename T>
__device__ __forceinline__ void store8(T *p, vec8_t<T> v) {
    *reinterpret_cast<vec8_t<T> *>(p) = v;
}
// Streaming (read-once) 16-byte load: bypasses nothing semantically, just marks the line
// non-temporal so KV-pool / weight streams do not evict reusable lines.
template <typename T>
__device__ __forceinline__ vec8_t<T> load8_nt(const T *p) {
    return __builtin_nontemporal_load(reinterpret_cast<const vec8_t<T> *>(p));
}

// Round-to-nearest-even conversion float -> T (v_cvt_f16_f32 / v_cvt_pk_bf16_f32 on gfx950).
template <typename T>
__device__ __forceinline__ T to_t(float x) {
    return static_cast<T>(x);
}
template <typename T>
__device__ __forceinline__ float to_f(T x) {
    return static_cast<float>(x);
}

// Arithmetic "in T": one rounding to T per operation, never contracted to fma.
// (fp16*fp16 and bf16*bf16 are exact in fp32, so rounding the fp32 product is the correctly
// rounded T product; sums are rounded once from the fp32 sum.)
template <typename T>
__device__ __forceinline__ T mul_t(T a, T b) {
    float r = __fmul_rn(to_f(a), to_f(b));
    return to_t<T>(r);
}
template <typename T>
__device__ __forceinline__ T add_t(T a, T b) {
    float r = __fadd_rn(to_f(a), to_f(b));
    return to_t<T>(r);
}
template <typename T>
__device__ __forceinline__ T sub_t(T a, T b) {
    float r = __fsub_rn(to_f(a), to_f(b));
    return to_t<T>(r);
}

// 2-element dot product with fp32 accumulate: v_dot2c_f32_f16 / v_dot2c_f32_bf16.
__device__ __forceinline__ float dot2(vec2_t<f16> a, vec2_t<f16> b, float c) {
    return __builtin_amdgcn_fdot2(a, b, c, false);
}
__device__ __forceinline__ float dot2(vec2_t<bf16> a, vec2_t<bf16> b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false);
}
template <typename T>
__device__ __forceinline__ float dot8(vec8_t<T> a, vec8_t<T> b, float c) {
    c = dot2(vec2_t<T>{a[0], a[1]}, vec2_t<T>{b[0], b[1]}, c);
    c = dot2(vec2_t<T>{a[2], a[3]}, vec2_t<T>{b[2], b[3]}, c);
    c = dot2(vec2_t<T>{a[4], a[5]}, vec2_t<T>{b[4], b[5]}, c);
    c = dot2(vec2_t<T>{a[6], a[7]}, vec2_t<T>{b[6], b[7]}, c);
    return c;
}

// ---- DPP cross-lane helpers (no LDS traffic) ----------------------------------------------------
constexpr int kDppQuadXor1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;       // quad_perm:[2,3,0,1]
constexpr int kDppRowHalfMirror = 0x141; // lane j <-> 7-j inside each 8 lanes
constexpr int kDppRowMirror = 0x140;     // lane j <-> 15-j inside each 16 lanes

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(
        __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// All-reduce (sum) over aligned groups of N lanes, N in {1,2,4,8,16}: every lane ends with the total.
template <int N>
__device__ __forceinline__ float group_allreduce_sum(float v) {
    if constexpr (N >= 2) v += dpp_mov<kDppQuadXor1>(v);
    if constexpr (N >= 4) v += dpp_mov<kDppQuadXor2>(v);
    if constexpr (N >= 8) v += dpp_mov<kDppRowHalfMirror>(v);
    if constexpr (N >= 16) v += dpp_mov<kDppRowMirror>(v);
    return v;
}

// Full-wave (64-lane) all-reduce: DPP inside rows of 16, then two cross-row exchanges.
__device__ __forceinline__ float wave_allreduce_sum(float v) {
    v = group_allreduce_sum<16>(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float wave_allreduce_max(float v) {
    v = fmaxf(v, dpp_mov<kDppQuadXor1>(v));
    v = fmaxf(v, dpp_mov<kDppQuadXor2>(v));
    v = fmaxf(v, dpp_mov<kDppRowHalfMirror>(v));
    v = fmaxf(v, dpp_mov<kDppRowMirror>(v));
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

// Split-K consumers: 8 consecutive outputs of a skinny-GEMM whose K was split across workgroups into `ks`
// fp32 partial slabs (slab k at slabs + k*slab_stride); summed in slab order, rounded once to T — exactly
// what the stand-alone reduce kernel of gemm_skinny.hip produces.
// Device-coherent 16-byte accesses (relaxed atomics at agent scope = sc1 loads/stores on gfx950): the store
// is written through to the level all XCDs share and the load never trusts a line of the local L2, so data
// can be handed from one workgroup to another INSIDE a kernel without whole-cache writeback/invalidate
// fences (a __threadfence() per workgroup costs a full L2 writeback: measured 4.6 -> 7.7 ms per decode step).
__device__ __forceinline__ void store_f4_coherent(float *p, const float4_t &v) {
    typedef unsigned long long u64;
    union { float f[4]; u64 u[2]; } b;
    b.f[0] = v[0]; b.f[1] = v[1]; b.f[2] = v[2]; b.f[3] = v[3];
    __hip_atomic_store(reinterpret_cast<u64 *>(p), b.u[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<u64 *>(p) + 1, b.u[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4_t load_f4_coherent(const float *p) {
    typedef unsigned long long u64;
    union { float f[4]; u64 u[2]; } b;
    b.u[0] = __hip_atomic_load(reinterpret_cast<const u64 *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b.u[1] = __hip_atomic_load(reinterpret_cast<const u64 *>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return float4_t{b.f[0], b.f[1], b.f[2], b.f[3]};
}

// KS known at compile time: all KS loads are issued before the first add (one memory round trip instead
// of a load -> add chain per slab); the adds keep the slab order, so the bits do not depend on KS being
// static or not.
template <typename T, int KS, bool COHERENT = false>
__device__ __forceinline__ vec8_t<T> load8_splitk_static(const float *slabs, int64_t slab_stride,
                                                         int64_t elem_off) {
    float4_t a[KS], b[KS];
    const float *p = slabs + elem_off;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        if constexpr (COHERENT) {
            a[k] = load_f4_coherent(p + k * slab_stride);
            b[k] = load_f4_coherent(p + k * slab_stride + 4);
        } else {
            a[k] = *reinterpret_cast<const float4_t *>(p + k * slab_stride);
            b[k] = *reinterpret_cast<const float4_t *>(p + k * slab_stride + 4);
        }
    }
    float4_t sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        sa += a[k];
        sb += b[k];
    }
    vec8_t<T> r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r[e] = static_cast<T>(sa[e]);
        r[4 + e] = static_cast<T>(sb[e]);
    }
    return r;
}

// Slabs written by other workgroups of the SAME kernel (split-K completion in gemm_skinny.hip).
template <typename T>
__device__ __forceinline__ vec8_t<T> load8_splitk_coherent(const float *slabs, int ks, int64_t slab_stride,
                                                           int64_t elem_off) {
    switch (ks) {
    case 1: return load8_splitk_static<T, 1, true>(slabs, slab_stride, elem_off);
    case 2: return load8_splitk_static<T, 2, true>(slabs, slab_stride, elem_off);
    case 4: return load8_splitk_static<T, 4, true>(slabs, slab_stride, elem_off);
    case 8: return load8_splitk_static<T, 8, true>(slabs, slab_stride, elem_off);
    default: return load8_splitk_static<T, 16, true>(slabs, slab_stride, elem_off);
    }
}

template <typename T>
__device__ __forceinline__ vec8_t<T> load8_splitk(const float *slabs, int ks, int64_t slab_stride,
                                                  int64_t elem_off) {
    switch (ks) { // uniform branch; the k-split counts the skinny GEMM actually uses
    case 1: return load8_splitk_static<T, 1>(slabs, slab_stride, elem_off);
    case 2: return load8_splitk_static<T, 2>(slabs, slab_stride, elem_off);
    case 4: return load8_splitk_static<T, 4>(slabs, slab_stride, elem_off);
    case 8: return load8_splitk_static<T, 8>(slabs, slab_stride, elem_off);
    case 16: return load8_splitk_static<T, 16>(slabs, slab_stride, elem_off);
    default: break;
    }
    float4_t a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    const float *p = slabs + elem_off;
    for (int k = 0; k < ks; ++k, p += slab_stride) {
        a += *reinterpret_cast<const float4_t *>(p);
        b += *reinterpret_cast<const float4_t *>(p + 4);
    }
    vec8_t<T> r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r[e] = static_cast<T>(a[e]);
        r[4 + e] = static_cast<T>(b[e]);
    }
    return r;
}

// Rotate-half rotary embedding on 8 (first-half, second-half) pairs. Rounding points follow the
// reference (rotary_emb.py:34-42): every product and every sum is rounded to the storage dtype.
template <typename T>
__device__ __forceinline__ void rotate8(vec8_t<T> &x0, vec8_t<T> &x1, const vec8_t<T> &c,
                                        const vec8_t<T> &s) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const T a = x0[j], b = x1[j];
        x0[j] = sub_t<T>(mul_t<T>(a, c[j]), mul_t<T>(b, s[j]));
        x1[j] = add_t<T>(mul_t<T>(a, s[j]), mul_t<T>(b, c[j]));
    }
}

// Wait states between an MFMA and the first NON-MFMA instruction that reads its result. The matrix pipe writes its
// destination registers some passes after issue; the padding a VALU / DS / VMEM reader needs is software's job (CDNA3/4
// ISA, "manually inserted wait states"). hipcc's hazard recognizer inserts it along fall-through code but, on ROCm 7.2
// for gfx950, NOT on the taken side of a branch that sits between the MFMA and the reader: seen in paged_attn.hip (r02),
// where the uniform `if (partial)` left `v_mfma ... ; s_cbranch ; v_max_f32 <result>` on the common path and the block
// maximum came out different from run to run (profiles/r02g_paged_attn_mfma.md). The softmax stayed self-consistent
// (any reference maximum is a valid one), so every parity test passed; only the last bit of some outputs moved.
// mfma_results_ready(acc) is tied to the accumulator as an in/out operand, so no compiler pass can move the MFMAs
// below it or the readers above it (a free-standing `asm volatile("s_nop")` between sched_barriers WAS moved above
// the MFMA chain). Several accumulators: mfma_results_tie() each, then mfma_results_ready() on the last one written.
// PASSES (4 cycles each): 4 for 16x16x32 / 16x16x16, 8 for 32x32x16 — the wait is passes + 3, rounded up generously.
template <typename V>
__device__ __forceinline__ void mfma_results_tie(V &acc) {
    asm volatile("" : "+v"(acc));
}
template <int PASSES, typename V>
__device__ __forceinline__ void mfma_results_ready(V &acc) {
    if constexpr (PASSES <= 4) asm volatile("s_nop 9" : "+v"(acc));            // 10 wait states
    else asm volatile("s_nop 15" : "+v"(acc));                                 // 16 wait states
}

// exp2 on the hardware transcendental unit (v_exp_f32 IS 2^x).
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

inline int check_launch() { return hipGetLastError() == hipSuccess ? SWL_OK : SWL_ERR_LAUNCH; }

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

} // namespace swl

// Dispatch a templated launcher on the runtime dtype code.
#define SWL_DISPATCH_DTYPE(dtype, T, ...)            \
    do {                                             \
        if ((dtype) == SWL_F16) {                    \
            using T = swl::f16;                      \
            __VA_ARGS__                              \
        } else if ((dtype) == SWL_BF16) {            \
            using T = swl::bf16;                     \
            __VA_ARGS__                              \
        } else {                                     \
            return SWL_ERR_BAD_ARG;                  \
        }                                            \
    } while (0)
