// l2_fill_probe.hip (r06b) — how many bytes per second can ONE CU pull through its L1 when the data is L2-resident, and does
// the answer depend on the SHAPE of the load instructions? DESIGN.md sections 4.6 / 4.8 carry "~55-57 GB/s per CU, L2 hit or
// HBM miss" from r02 / r05 — measured with 32 KiB in flight per CU (r02: Little's law at HBM latency) and with fragment-shaped
// x loads (r05: 16 rows x 64 B per instruction). This probe separates the cases:
//   region: shared  — every workgroup reads the same S bytes (x of a decode projection: all CUs read the same rows)
//           private — workgroup b reads its own 64 KiB (no two CUs want the same line; 16 MiB in all: fits the eight L2s)
//   form:   coal    — lane -> 16 B, 64 lanes contiguous (1 KiB per instruction = 8 full lines)
//           rows4   — lane -> (row lane/16, 16 B piece lane%16): 4 rows x 256 B, rows PITCH bytes apart (full lines)
//           frag    — lane -> (row lane%16, 16 B piece lane/16): 16 rows x 64 B (half lines; the r05 x loads)
//           dma     — coal through LDS-DMA (global_load_lds_dwordx4), no VGPRs
//   stream: 0 / 1   — half of the waves stream a private non-resident region non-temporally next to it (the W stream)
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O3 tools/probe/l2_fill_probe.hip -o /tmp/l2_fill_probe && /tmp/l2_fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__constant__ int kPitchDev;           // bytes between rows of the "x" region
static int g_pitch = 8192;            // (a 4096-wide bf16 activation; swept in main)
constexpr int kU = 8;                 // loads in flight per lane

enum { COAL = 0, ROWS4 = 1, FRAG = 2, DMA = 3 };

template <int FORM>
__global__ __launch_bounds__(512, 2) void fill_kernel(const char *__restrict__ resident, size_t region_bytes, int private_regions,
                                                      const char *__restrict__ cold, size_t cold_bytes_per_wg, int stream_waves,
                                                      int reps, unsigned long long *sink) {
    __shared__ __attribute__((aligned(16))) char lds[8][kU * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const size_t kPitch = kPitchDev;
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = wall_clock64();        // 100 MHz
    const bool streamer = wave >= nwaves - stream_waves;
    if (streamer) {
        // the W stream: this wave's share of the workgroup's cold region, 1 KiB per instruction, non-temporal, kU in flight
        const int sw = wave - (nwaves - stream_waves);
        const size_t per_wave = cold_bytes_per_wg / stream_waves;
        const char *p = cold + static_cast<size_t>(blockIdx.x) * cold_bytes_per_wg + sw * per_wave + lane * 16;
        for (size_t off = 0; off + kU * 1024 <= per_wave; off += kU * 1024) {
            u32x4 v[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + off + u * 1024));
#pragma unroll
            for (int u = 0; u < kU; ++u) acc ^= v[u];
        }
    } else {
        const int rw = nwaves - stream_waves;        // reader waves
        const char *base = resident + (private_regions ? static_cast<size_t>(blockIdx.x) * region_bytes : 0);
        // a reader wave walks its 1/rw of the region per pass in batches of kU KiB
        const size_t per_wave = region_bytes / rw;
        const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char *)&lds[wave][0]));
        for (int rep = 0; rep < reps; ++rep) {
            for (size_t off = 0; off + kU * 1024 <= per_wave; off += kU * 1024) {
                const size_t o = wave * per_wave + off;         // byte offset of this batch in a [rows][kPitch] image
                if constexpr (FORM == DMA) {
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        unsigned keep;
                        const char *b = base + o + u * 1024;
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep) : "v"(static_cast<unsigned>(lane * 16)), "s"(b), "s"(lds0 + u * 1024) : "memory");
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    acc[0] ^= *reinterpret_cast<const unsigned *>(&lds[wave][lane * 4]);
                } else {
                    u32x4 v[kU];
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        size_t a;
                        if constexpr (FORM == COAL) a = o + u * 1024 + lane * 16;
                        else if constexpr (FORM == ROWS4) {
                            // batch = kU instructions x (4 rows x 256 B): the same bytes as kU KiB, as a [rows][256 B] window
                            const size_t blk = (o / 1024 + u);                 // which 4-row block of 256-byte columns
                            const size_t rows_total = 128;                     // rows in the image
                            const size_t rb = (blk * 4) % rows_total, cb = (blk * 4) / rows_total;
                            a = (rb + (lane >> 4)) * kPitch + cb * 256 + (lane & 15) * 16;
                        } else {
                            const size_t blk = (o / 1024 + u);                 // which 16-row block of 64-byte columns
                            const size_t rows_total = 128;
                            const size_t rb = (blk * 16) % rows_total, cb = (blk * 16) / rows_total;
                            a = (rb + (lane & 15)) * kPitch + cb * 64 + (lane >> 4) * 16;
                        }
                        v[u] = *reinterpret_cast<const u32x4 *>(base + a);
                    }
#pragma unroll
                    for (int u = 0; u < kU; ++u) acc ^= v[u];
                }
            }
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[2] = 1;
    const unsigned long long dt = wall_clock64() - t0;   // longest wave of each role, over all launches since the last reset
    if (lane == 0) atomicMax(&sink[streamer ? 1 : 0], dt);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int FORM>
static double run(const char *res, size_t region, int priv, const char *cold, size_t cold_per_wg, int stream_waves, int reps,
                  unsigned long long *sink, int threads, double *reader_us, double *stream_us) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((fill_kernel<FORM>), dim3(256), dim3(threads), 0, 0, res, region, priv, cold, cold_per_wg, stream_waves, reps, sink);
    CK(hipMemsetAsync(sink, 0, 24, 0));
    CK(hipEventRecord(a));
    const int iters = 10;
    for (int i = 0; i < iters; ++i)     // (the cold region is 4x what a launch reads: rotate, so no launch finds its bytes in the 256 MiB MALL)
        hipLaunchKernelGGL((fill_kernel<FORM>), dim3(256), dim3(threads), 0, 0, res, region, priv, cold + (i & 3) * 256 * cold_per_wg,
                           cold_per_wg, stream_waves, reps, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long h[3];
    CK(hipMemcpy(h, sink, 24, hipMemcpyDeviceToHost));
    *reader_us = h[0] / 100.0;
    *stream_us = h[1] / 100.0;
    return ms * 1e3 / iters;
}

int main() {
    const size_t shared_bytes = 1 << 20, private_bytes = 64 << 10;
    const size_t cold_per_wg = 1 << 20;                   // 256 MiB of "weights" per launch
    char *res, *cold; unsigned long long *sink;
    CK(hipMalloc(&res, 256 * private_bytes > shared_bytes ? 256 * private_bytes : shared_bytes));
    CK(hipMalloc(&cold, 256 * cold_per_wg * 4));
    CK(hipMalloc(&sink, 24));
    CK(hipMemset(res, 1, 256 * private_bytes));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(kPitchDev), &g_pitch, sizeof(int)));
    CK(hipMemset(cold, 2, 256 * cold_per_wg * 4));
    const char *names[4] = {"coal", "rows4", "frag", "dma"};
    for (int threads : {256, 512})
        for (int priv = 0; priv < 2; ++priv)
            for (int stream = 0; stream < 2; ++stream)
                for (int form = 0; form < 4; ++form) {
                    if (priv && (form == ROWS4 || form == FRAG)) continue;      // (row images only for the shared x region)
                    const int nw = threads / 64, sw = stream ? nw / 2 : 0;
                    const size_t region = priv ? private_bytes : shared_bytes;
                    const int reps = priv ? 64 : 4;                              // 4 MiB per workgroup either way
                    double us, rus = 0, sus = 0;
                    const char *c = cold;
                    if (form == COAL) us = run<COAL>(res, region, priv, c, cold_per_wg, sw, reps, sink, threads, &rus, &sus);
                    else if (form == ROWS4) us = run<ROWS4>(res, region, priv, c, cold_per_wg, sw, reps, sink, threads, &rus, &sus);
                    else if (form == FRAG) us = run<FRAG>(res, region, priv, c, cold_per_wg, sw, reps, sink, threads, &rus, &sus);
                    else us = run<DMA>(res, region, priv, c, cold_per_wg, sw, reps, sink, threads, &rus, &sus);
                    const double bytes = static_cast<double>(region) * reps;
                    printf("{\"threads\": %d, \"reader_waves\": %d, \"region\": \"%s\", \"form\": \"%s\", \"w_stream_waves\": %d, "
                           "\"launch_us\": %.2f, \"reader_us\": %.2f, \"stream_us\": %.2f, \"resident_GBps_per_CU\": %.1f, "
                           "\"cold_TBps_chip\": %.2f}\n",
                           threads, nw - sw, priv ? "private" : "shared", names[form], sw, us, rus, sus, bytes / rus / 1e3,
                           sw ? 256.0 * cold_per_wg / sus / 1e6 : 0.0);
                    fflush(stdout);
                }
    // the row-shaped forms against the row pitch (power-of-two pitches put the rows of one instruction into one L2 channel)
    for (int pitch : {8192, 8192 + 128, 8192 + 256, 8192 + 512, 8192 + 1024, 8192 + 2048, 28672, 28672 + 256, 28672 + 512})
        for (int threads : {256, 512})
            for (int stream = 0; stream < 2; ++stream)
                for (int form : {ROWS4, FRAG}) {
                    CK(hipMemcpyToSymbol(HIP_SYMBOL(kPitchDev), &pitch, sizeof(int)));
                    const int nw = threads / 64, sw = stream ? nw / 2 : 0;
                    double rus = 0, sus = 0, us;
                    if (form == ROWS4) us = run<ROWS4>(res, shared_bytes, 0, cold, cold_per_wg, sw, 4, sink, threads, &rus, &sus);
                    else us = run<FRAG>(res, shared_bytes, 0, cold, cold_per_wg, sw, 4, sink, threads, &rus, &sus);
                    printf("{\"sweep\": \"pitch\", \"pitch\": %d, \"threads\": %d, \"reader_waves\": %d, \"form\": \"%s\", \"w_stream_waves\": %d, "
                           "\"launch_us\": %.2f, \"reader_us\": %.2f, \"resident_GBps_per_CU\": %.1f, \"cold_TBps_chip\": %.2f}\n",
                           pitch, threads, nw - sw, names[form], sw, us, rus, 4.0 * shared_bytes / rus / 1e3,
                           sw ? 256.0 * cold_per_wg / sus / 1e6 : 0.0);
                    fflush(stdout);
                }
    return 0;
}
