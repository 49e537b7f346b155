// gemm_probe.hip — NOT product code. Ablations of the skinny-GEMM inner loop to find what bounds it.
// Built by tools/probe/run_gemm_probe.sh into tools/probe/libgemm_probe.so; results are wrong on purpose
// for every variant but 0.
//   0 baseline (= product kGemmDirect)          1 no per-tile x traffic (x tile staged once)
//   2 W fed to the MFMA from registers (no W LDS round trip), x as baseline
//   3 = 1 + 2                                   4 stream only: W loads + xor, no LDS, no MFMA
//   5 stream only, 3 tiles in flight            6 = 1 with 3 W tiles in flight
#include "../../swiftllm_amd/csrc/swl_common.h"
namespace swl {
__device__ __forceinline__ float16_t mfma(vec8_t<bf16> a, vec8_t<bf16> b, float16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
constexpr int kKT = 128;
constexpr int kW = 4;
typedef bf16 T;

template <int V, int OCC>
__global__ __launch_bounds__(kW * 64, OCC) void probe_kernel(T *__restrict__ out, const T *__restrict__ x,
                                                             const T *__restrict__ w, int M, int N, int K,
                                                             int64_t x_stride) {
    __shared__ __attribute__((aligned(16))) T lds[kW][2][32 * kKT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = (blockIdx.x * kW + wave) * 32;
    if (n0 >= N) return;
    const int nkt = K / kKT;
    const int rsub = lane >> 4, chunk = lane & 15;
    const T *wsrc = w + static_cast<int64_t>(n0 + rsub) * K + chunk * 8;
    const T *xsrc = x + chunk * 8;
    int xrow_off[8], lds_wr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + rsub;
        xrow_off[i] = min(row, M - 1) * static_cast<int>(x_stride);
        lds_wr[i] = row * kKT + ((chunk ^ (row & 15)) << 3);
    }
    const int l32 = lane & 31, hf = lane >> 5;
    T *wl = &lds[wave][0][0];
    T *xl = &lds[wave][1][0];
    constexpr bool kXPerTile = (V == 0 || V == 2);
    constexpr bool kWLds = (V == 0 || V == 1 || V == 6);
    constexpr bool kStreamOnly = (V == 4 || V == 5);

    vec8_t<T> wa[8], xa[8], wb[8], xb[8], wc[8];
    auto issue = [&](vec8_t<T>(&wr)[8], vec8_t<T>(&xr)[8], int kt) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            wr[i] = load8_nt(wsrc + static_cast<int64_t>(4 * i) * K + kt * kKT);
            if constexpr (kXPerTile) xr[i] = load8(xsrc + xrow_off[i] + kt * kKT);
        }
    };
    float16_t acc = float16_t{};
    vec8_t<T> sink = {};
    auto process = [&](const vec8_t<T>(&wr)[8], const vec8_t<T>(&xr)[8]) {
        if constexpr (kStreamOnly) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                u4 a = *reinterpret_cast<const u4 *>(&wr[i]);
                u4 s = *reinterpret_cast<u4 *>(&sink);
                s ^= a;
                sink = *reinterpret_cast<vec8_t<T> *>(&s);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (kWLds) *reinterpret_cast<vec8_t<T> *>(wl + lds_wr[i]) = wr[i];
            if constexpr (kXPerTile) *reinterpret_cast<vec8_t<T> *>(xl + lds_wr[i]) = xr[i];
        }
#pragma unroll
        for (int kk = 0; kk < kKT / 16; ++kk) {
            const int off = l32 * kKT + (((2 * kk + hf) ^ (l32 & 15)) << 3);
            vec8_t<T> a;
            if constexpr (kWLds) a = *reinterpret_cast<const vec8_t<T> *>(wl + off);
            else a = wr[kk];
            const vec8_t<T> b = *reinterpret_cast<const vec8_t<T> *>(xl + off);
            acc = mfma(a, b, acc);
        }
    };
    if constexpr (!kXPerTile && !kStreamOnly) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<vec8_t<T> *>(xl + lds_wr[i]) = load8(xsrc + xrow_off[i]);
    }
    if constexpr (V == 5 || V == 6) {
        issue(wa, xa, 0);
        issue(wb, xb, 1);
        int kt = 0;
        for (; kt + 5 <= nkt; kt += 3) {
            issue(wc, xa, kt + 2); process(wa, xa);
            issue(wa, xa, kt + 3); process(wb, xa);
            issue(wb, xa, kt + 4); process(wc, xa);
        }
        // tail (nkt = 32: kt ends at 30): tiles kt, kt+1 are in wa, wb
        process(wa, xa);
        process(wb, xa);
    } else {
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
    }
    if constexpr (kStreamOnly) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += to_f(sink[e]);
    }
    if (l32 < M) {
        typedef T vec4 __attribute__((ext_vector_type(4)));
        T *o = out + static_cast<int64_t>(l32) * N + n0 + 4 * hf;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            vec4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = to_t<T>(acc[4 * r4 + e]);
            *reinterpret_cast<vec4 *>(o + 8 * r4) = v;
        }
    }
}

// ---- candidate: x tile shared by the NW waves of the workgroup (one barrier per K-tile) ---------------
template <int NW, int OCC>
__global__ __launch_bounds__(NW * 64, OCC) void shared_x_kernel(T *__restrict__ out, const T *__restrict__ x,
                                                                const T *__restrict__ w, int M, int N, int K,
                                                                int64_t x_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *xs = reinterpret_cast<T *>(smem);                       // [2][32 * kKT]
    T *wl = xs + 2 * 32 * kKT + (threadIdx.x >> 6) * 32 * kKT; // wave-private W tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col0 = (blockIdx.x * NW + wave) * 32;
    const bool tile_ok = col0 < N;
    const int n0 = tile_ok ? col0 : 0;
    const int nkt = K / kKT;
    const int rsub = lane >> 4, chunk = lane & 15;
    const T *wsrc = w + static_cast<int64_t>(n0 + rsub) * K + chunk * 8;
    constexpr int XL = 8 / NW;      // x row-groups (of 4 rows) each wave stages per tile
    int lds_wr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + rsub;
        lds_wr[i] = row * kKT + ((chunk ^ (row & 15)) << 3);
    }
    const T *xsrc[XL];
    int xs_wr[XL];
#pragma unroll
    for (int j = 0; j < XL; ++j) {
        const int row = 4 * (wave * XL + j) + rsub;
        xsrc[j] = x + static_cast<int64_t>(min(row, M - 1)) * x_stride + chunk * 8;
        xs_wr[j] = row * kKT + ((chunk ^ (row & 15)) << 3);
    }
    const int l32 = lane & 31, hf = lane >> 5;

    vec8_t<T> wa[8], wb[8], xa[XL], xb[XL];
    auto issue = [&](vec8_t<T>(&wr)[8], vec8_t<T>(&xr)[XL], int kt) {
#pragma unroll
        for (int j = 0; j < XL; ++j) xr[j] = load8(xsrc[j] + kt * kKT);   // first: they return first
#pragma unroll
        for (int i = 0; i < 8; ++i) wr[i] = load8_nt(wsrc + static_cast<int64_t>(4 * i) * K + kt * kKT);
    };
    auto stage_x = [&](const vec8_t<T>(&xr)[XL], int buf) {
#pragma unroll
        for (int j = 0; j < XL; ++j) *reinterpret_cast<vec8_t<T> *>(xs + buf * 32 * kKT + xs_wr[j]) = xr[j];
    };
    float16_t acc = float16_t{};
    auto process = [&](const vec8_t<T>(&wr)[8], int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<vec8_t<T> *>(wl + lds_wr[i]) = wr[i];
        const T *xl = xs + buf * 32 * kKT;
#pragma unroll
        for (int kk = 0; kk < kKT / 16; ++kk) {
            const int off = l32 * kKT + (((2 * kk + hf) ^ (l32 & 15)) << 3);
            const vec8_t<T> a = *reinterpret_cast<const vec8_t<T> *>(wl + off);
            const vec8_t<T> b = *reinterpret_cast<const vec8_t<T> *>(xl + off);
            acc = mfma(a, b, acc);
        }
    };
    // prologue: tile 0 in flight, its x quarter staged and published
    issue(wa, xa, 0);
    stage_x(xa, 0);
    __syncthreads();
    int kt = 0;
    for (; kt + 2 < nkt; kt += 2) {
        issue(wb, xb, kt + 1);
        process(wa, 0);
        stage_x(xb, 1);
        __syncthreads();
        issue(wa, xa, kt + 2);
        process(wb, 1);
        stage_x(xa, 0);
        __syncthreads();
    }
    if (nkt - kt == 2) {
        issue(wb, xb, kt + 1);
        process(wa, 0);
        stage_x(xb, 1);
        __syncthreads();
        process(wb, 1);
    } else {
        process(wa, 0);
    }
    if (tile_ok && l32 < M) {
        typedef T vec4 __attribute__((ext_vector_type(4)));
        T *o = out + static_cast<int64_t>(l32) * N + n0 + 4 * hf;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            vec4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = to_t<T>(acc[4 * r4 + e]);
            *reinterpret_cast<vec4 *>(o + 8 * r4) = v;
        }
    }
}

// ---- candidate 2: shared x tile + D-deep register ring of W tiles + split-K (grid.y) -----------------
template <int NW, int D, int OCC>
__global__ __launch_bounds__(NW * 64, OCC) void ring_kernel(void *__restrict__ out_, const T *__restrict__ x,
                                                            const T *__restrict__ w, int M, int N, int K, int kc,
                                                            int64_t x_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *xs = reinterpret_cast<T *>(smem);                       // [2][32 * kKT]
    T *wl = xs + 2 * 32 * kKT + (threadIdx.x >> 6) * 32 * kKT; // wave-private W tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col0 = (blockIdx.x * NW + wave) * 32;
    const bool tile_ok = col0 < N;
    const int n0 = tile_ok ? col0 : 0;
    const int ksplit = blockIdx.y;
    const int k_begin = ksplit * kc;
    const int nkt = kc / kKT;
    const int rsub = lane >> 4, chunk = lane & 15;
    const T *wsrc = w + static_cast<int64_t>(n0 + rsub) * K + k_begin + chunk * 8;
    constexpr int XL = 8 / NW;
    int lds_wr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + rsub;
        lds_wr[i] = row * kKT + ((chunk ^ (row & 15)) << 3);
    }
    const T *xsrc[XL];
    int xs_wr[XL];
#pragma unroll
    for (int j = 0; j < XL; ++j) {
        const int row = 4 * (wave * XL + j) + rsub;
        xsrc[j] = x + static_cast<int64_t>(min(row, M - 1)) * x_stride + k_begin + chunk * 8;
        xs_wr[j] = row * kKT + ((chunk ^ (row & 15)) << 3);
    }
    const int l32 = lane & 31, hf = lane >> 5;

    vec8_t<T> wr[D][8], xr[D][XL];
    float16_t acc = float16_t{};
#define ISSUE(slot, kt)                                                                               \
    {                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < XL; ++j) xr[slot][j] = load8(xsrc[j] + (kt) * kKT);     \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                 \
            wr[slot][i] = load8_nt(wsrc + static_cast<int64_t>(4 * i) * K + (kt) * kKT);              \
    }
#define STAGE_X(slot, buf)                                                                            \
    {                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < XL; ++j)                                                \
            *reinterpret_cast<vec8_t<T> *>(xs + (buf) * 32 * kKT + xs_wr[j]) = xr[slot][j];           \
    }
#define PROCESS(slot, buf)                                                                            \
    {                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                 \
            *reinterpret_cast<vec8_t<T> *>(wl + lds_wr[i]) = wr[slot][i];                             \
        const T *xl = xs + (buf) * 32 * kKT;                                                          \
        _Pragma("unroll") for (int kk = 0; kk < kKT / 16; ++kk) {                                     \
            const int off = l32 * kKT + (((2 * kk + hf) ^ (l32 & 15)) << 3);                          \
            const vec8_t<T> a = *reinterpret_cast<const vec8_t<T> *>(wl + off);                       \
            const vec8_t<T> b = *reinterpret_cast<const vec8_t<T> *>(xl + off);                       \
            acc = mfma(a, b, acc);                                                                    \
        }                                                                                             \
    }
    // prologue: D-1 tiles in flight (nkt >= D - 1 is guaranteed by the host), x of tile 0 published
#pragma unroll
    for (int d = 0; d < D - 1; ++d) ISSUE(d, d);
    STAGE_X(0, 0);
    __syncthreads();
    int kt = 0;
    // steady state, branch-free: step d processes tile kt+d from slot d, refills slot (d+D-1)%D
    for (; kt + 2 * D - 1 <= nkt; kt += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            ISSUE((d + D - 1) % D, kt + d + D - 1);
            PROCESS(d, (kt + d) & 1);
            STAGE_X((d + 1) % D, (kt + d + 1) & 1);
            __syncthreads();
        }
    }
    // drain: between D-1 and 2D-2 tiles left, D-1 of them in flight (slots 0..D-2 hold kt..kt+D-2)
    const int rem = nkt - kt;
#pragma unroll
    for (int j = 0; j < 2 * D - 2; ++j) {
        if (j < rem) {
            if (j + D - 1 < rem) ISSUE((j + D - 1) % D, kt + j + D - 1);
            PROCESS(j % D, (kt + j) & 1);
            if (j + 1 < rem) {
                STAGE_X((j + 1) % D, (kt + j + 1) & 1);
                __syncthreads();
            }
        }
    }
#undef ISSUE
#undef STAGE_X
#undef PROCESS
    if (tile_ok && l32 < M) {
        if (gridDim.y > 1) {
            float *slab = static_cast<float *>(out_) + (static_cast<int64_t>(ksplit) * M + l32) * N + n0 + 4 * hf;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4_t v = {acc[4 * r4], acc[4 * r4 + 1], acc[4 * r4 + 2], acc[4 * r4 + 3]};
                *reinterpret_cast<float4_t *>(slab + 8 * r4) = v;
            }
        } else {
            typedef T vec4 __attribute__((ext_vector_type(4)));
            T *o = static_cast<T *>(out_) + static_cast<int64_t>(l32) * N + n0 + 4 * hf;
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
} // namespace swl

extern "C" int probe_gemm(int variant, int occ, void *out, const void *x, const void *w, int M, int N, int K,
                          void *stream) {
    using namespace swl;
    const dim3 grid((N / 32 + kW - 1) / kW), block(kW * 64);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define CASE(V, O)                                                                                   \
    if (variant == V && occ == O) {                                                                  \
        hipLaunchKernelGGL((probe_kernel<V, O>), grid, block, 0, s, static_cast<T *>(out),           \
                           static_cast<const T *>(x), static_cast<const T *>(w), M, N, K, (int64_t)K); \
        return hipGetLastError() == hipSuccess ? 0 : -1;                                             \
    }
    CASE(0, 2) CASE(1, 2) CASE(2, 2) CASE(3, 2) CASE(4, 2) CASE(5, 2) CASE(6, 2)
    CASE(1, 3) CASE(3, 3) CASE(4, 3) CASE(4, 4) CASE(3, 4) CASE(5, 1)
#undef CASE
#define SCASE(V, NW, O)                                                                               \
    if (variant == V && occ == O) {                                                                   \
        const size_t lds = (2 + NW) * 32 * kKT * sizeof(T);                                           \
        auto kern = shared_x_kernel<NW, O>;                                                           \
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kern, dim3((N / 32 + NW - 1) / NW), dim3(NW * 64), lds, s, static_cast<T *>(out), \
                           static_cast<const T *>(x), static_cast<const T *>(w), M, N, K, (int64_t)K); \
        return hipGetLastError() == hipSuccess ? 0 : -1;                                              \
    }
    SCASE(10, 4, 2) SCASE(10, 4, 3) SCASE(11, 8, 1) SCASE(11, 8, 2) SCASE(12, 2, 4) SCASE(12, 2, 6) SCASE(11, 8, 4)
#undef SCASE
    return -3;
}

// variant = 100 + 10*D + (NW == 8); out = T[M][N] when ks == 1, else fp32 slabs [ks][M][N]
extern "C" int probe_ring(int nw, int depth, int occ, void *out, const void *x, const void *w, int M, int N,
                          int K, int ks, void *stream) {
    using namespace swl;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int kc = K / ks;
    if (kc / kKT < depth - 1) return -3;
#define RCASE(NW, D, O)                                                                               \
    if (nw == NW && depth == D && occ == O) {                                                         \
        const size_t lds = (2 + NW) * 32 * kKT * sizeof(T);                                           \
        auto kern = ring_kernel<NW, D, O>;                                                            \
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kern, dim3((N / 32 + NW - 1) / NW, ks), dim3(NW * 64), lds, s, out,        \
                           static_cast<const T *>(x), static_cast<const T *>(w), M, N, K, kc, (int64_t)K); \
        return hipGetLastError() == hipSuccess ? 0 : -1;                                              \
    }
    RCASE(4, 2, 3) RCASE(4, 3, 2) RCASE(4, 3, 3) RCASE(4, 4, 2) RCASE(4, 5, 2) RCASE(4, 6, 1) RCASE(4, 8, 1)
    RCASE(8, 3, 2) RCASE(8, 2, 3)
#undef RCASE
    return -3;
}

// ---- prefetch probe: touch the first `bytes_per_row` of every row of W (what the GEMM's first tiles read) ----
__global__ __launch_bounds__(256) void prefetch_rows_kernel(const char *w, long long row_pitch, int rows_per_wg,
                                                            int bytes_per_row, int *sink) {
    // WG j touches the rows GEMM workgroup j will stream first (consecutive WGs -> consecutive XCDs, as there)
    const int per_row = bytes_per_row / 16;
    const int items = rows_per_wg * per_row;
    unsigned acc = 0;
    for (int it = threadIdx.x; it < items; it += 256) {
        const int r = it / per_row, c = it % per_row;
        const uint4 v = *reinterpret_cast<const uint4 *>(w + (static_cast<long long>(blockIdx.x) * rows_per_wg + r) * row_pitch + c * 16);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = 1; // never true in practice: keeps the loads alive
}

extern "C" int probe_prefetch(const void *w, long long row_pitch, int rows, int rows_per_wg, int bytes_per_row,
                              void *sink, void *stream) {
    hipLaunchKernelGGL(prefetch_rows_kernel, dim3(rows / rows_per_wg), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const char *>(w), row_pitch, rows_per_wg, bytes_per_row, static_cast<int *>(sink));
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- access-pattern probe: stream W with `RUN` contiguous bytes per row per instruction ------------------------
// A wave owns 32 rows; per step it issues 16 x 16-byte-per-lane loads = 16 KiB covering [32 rows][512 B]
// (RUN = 256: as two K-tiles of the product kernel, 4 rows x 256 B per instruction; RUN = 512: 2 rows x 512 B;
// RUN = 1024: 1 row x 1 KiB per instruction, 16 rows per step and two steps per 32 rows).
template <int RUN>
__global__ __launch_bounds__(256, 2) void stream_pattern_kernel(const swl::bf16 *__restrict__ w, int N, int K, int *sink) {
    using namespace swl;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = (blockIdx.x * 4 + wave) * 32;
    if (n0 >= N) return;
    constexpr int LPR = RUN / 16;          // lanes per row
    constexpr int RPI = 64 / LPR;          // rows per instruction
    const int r = lane / LPR, c = lane % LPR;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 acc = {0, 0, 0, 0};
    const char *base = reinterpret_cast<const char *>(w);
    const long long row_bytes = static_cast<long long>(K) * 2;
    // step = 16 instructions = 16 KiB: rows [rb, rb + 16*RPI) x RUN bytes at byte offset kb
    const int rows_per_step = 16 * RPI;                 // 64 (256), 32 (512), 16 (1024)
    const int steps_per_colblock = 32 / (rows_per_step > 32 ? 32 : rows_per_step); // 1, 1, 2
    const int col_bytes = rows_per_step > 32 ? RUN * 2 : RUN; // RUN=256: two k-chunks of 256 B per step
    for (long long kb = 0; kb < row_bytes; kb += col_bytes) {
        for (int s = 0; s < steps_per_colblock; ++s) {
            u4 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int row, off;
                if (RUN == 256) { row = (i & 7) * 4 + r; off = (i >> 3) * 256 + c * 16; }
                else { row = s * rows_per_step + i * RPI + r; off = c * 16; }
                v[i] = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(base + (n0 + row) * row_bytes + kb + off));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc ^= v[i];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}

extern "C" int probe_stream_pattern(int run, const void *w, int N, int K, void *sink, void *stream) {
    const dim3 grid(N / 128), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const swl::bf16 *wp = static_cast<const swl::bf16 *>(w);
    if (run == 256) hipLaunchKernelGGL(stream_pattern_kernel<256>, grid, block, 0, s, wp, N, K, static_cast<int *>(sink));
    else if (run == 512) hipLaunchKernelGGL(stream_pattern_kernel<512>, grid, block, 0, s, wp, N, K, static_cast<int *>(sink));
    else if (run == 1024) hipLaunchKernelGGL(stream_pattern_kernel<1024>, grid, block, 0, s, wp, N, K, static_cast<int *>(sink));
    else return -3;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- candidate 3: PRE-PACKED weights ----------------------------------------------------------------------------
// W is repacked once (at load time) into MFMA-fragment order: [N/32][K/16][64 lanes][8 elements] — the A operand of
// v_mfma_f32_32x32x16 for 32 rows x 16 k is ONE contiguous KiB, a wave's whole K range for its 32 rows one
// contiguous run (K*64 bytes). Loads go global -> VGPR -> MFMA: no LDS round trip for W, perfect DRAM locality.
// x tile shared by the workgroup through LDS as in the ring kernel. D-deep register ring of 8 KiB tiles.
template <int D, int OCC>
__global__ __launch_bounds__(256, OCC) void packed_kernel(void *__restrict__ out_, const swl::bf16 *__restrict__ x,
                                                          const swl::bf16 *__restrict__ wpk, int M, int N, int K,
                                                          int kc, int64_t x_stride) {
    using namespace swl;
    typedef bf16 T;
    constexpr int XL = 2;
    __shared__ __attribute__((aligned(16))) T xs[2][32 * kKT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col0 = (blockIdx.x * 4 + wave) * 32;
    const bool tile_ok = col0 < N;
    const int nt = tile_ok ? col0 / 32 : 0;
    const int ksplit = blockIdx.y;
    const int k_begin = ksplit * kc;
    const int nkt = kc / kKT;
    const int rsub = lane >> 4, chunk = lane & 15;
    // packed source: block (nt, k16) at ((nt * K/16) + k16) * 512 elements, lane l at + l*8
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
    vec8_t<T> wr[D][8], xr[D][XL];
    float16_t acc = float16_t{};
#define PK_ISSUE(slot, tile)                                                                          \
    {                                                                                                \
        _Pragma("unroll") for (int q_ = 0; q_ < XL; ++q_) xr[slot][q_] = load8(xsrc[q_] + (tile) * kKT); \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                             \
            wr[slot][i_] = load8_nt(wsrc + (static_cast<int64_t>(tile) * 8 + i_) * 512);             \
    }
#define PK_STAGE_X(slot, buf)                                                                         \
    {                                                                                                \
        _Pragma("unroll") for (int q_ = 0; q_ < XL; ++q_)                                            \
            *reinterpret_cast<vec8_t<T> *>(&xs[buf][xs_wr[q_]]) = xr[slot][q_];                      \
    }
#define PK_PROCESS(slot, buf)                                                                         \
    {                                                                                                \
        _Pragma("unroll") for (int kk_ = 0; kk_ < kKT / 16; ++kk_) {                                 \
            const int off_ = l32 * kKT + (((2 * kk_ + hf) ^ (l32 & 15)) << 3);                       \
            const vec8_t<T> b_ = *reinterpret_cast<const vec8_t<T> *>(&xs[buf][off_]);               \
            acc = mfma(wr[slot][kk_], b_, acc);                                                      \
        }                                                                                            \
    }
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
        if (d < nkt) PK_ISSUE(d, d);
    PK_STAGE_X(0, 0);
    __syncthreads();
    int kt = 0;
    for (; kt + 2 * D - 1 <= nkt; kt += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            PK_ISSUE((d + D - 1) % D, kt + d + D - 1);
            PK_PROCESS(d, (kt + d) & 1);
            PK_STAGE_X((d + 1) % D, (kt + d + 1) & 1);
            __syncthreads();
        }
    }
    const int rem = nkt - kt;
#pragma unroll
    for (int t = 0; t < 2 * D - 2; ++t) {
        if (t < rem) {
            if (t + D - 1 < rem) PK_ISSUE((t + D - 1) % D, kt + t + D - 1);
            PK_PROCESS(t % D, (kt + t) & 1);
            if (t + 1 < rem) {
                PK_STAGE_X((t + 1) % D, (kt + t + 1) & 1);
                __syncthreads();
            }
        }
    }
#undef PK_ISSUE
#undef PK_STAGE_X
#undef PK_PROCESS
    if (tile_ok && l32 < M) {
        const int n0 = col0;
        if (gridDim.y > 1) {
            float *slab = static_cast<float *>(out_) + (static_cast<int64_t>(ksplit) * M + l32) * N + n0 + 4 * hf;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4_t v = {acc[4 * r4], acc[4 * r4 + 1], acc[4 * r4 + 2], acc[4 * r4 + 3]};
                *reinterpret_cast<float4_t *>(slab + 8 * r4) = v;
            }
        } else {
            typedef T vec4 __attribute__((ext_vector_type(4)));
            T *o = static_cast<T *>(out_) + static_cast<int64_t>(l32) * N + n0 + 4 * hf;
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

extern "C" int probe_packed(int depth, int occ, void *out, const void *x, const void *wpk, int M, int N, int K, int ks,
                            void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int kc = K / ks;
    const dim3 grid((N / 32 + 3) / 4, ks);
#define PCASE(DD, O)                                                                                  \
    if (depth == DD && occ == O) {                                                                    \
        hipLaunchKernelGGL((packed_kernel<DD, O>), grid, dim3(256), 0, s, out,                        \
                           static_cast<const swl::bf16 *>(x), static_cast<const swl::bf16 *>(wpk), M, N, K, kc, \
                           (int64_t)K);                                                               \
        return hipGetLastError() == hipSuccess ? 0 : -1;                                              \
    }
    PCASE(2, 2) PCASE(3, 2) PCASE(4, 2) PCASE(2, 3) PCASE(3, 3) PCASE(2, 4) PCASE(5, 2) PCASE(6, 1)
#undef PCASE
    return -3;
}
