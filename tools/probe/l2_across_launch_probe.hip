// l2_across_launch_probe.hip (r06b) — does a line a workgroup pulled into its XCD's L2 survive the kernel boundary? If it does, a
// kernel's finishing workgroups could touch the first weight tiles of the NEXT kernel's workgroups of the same XCD (block b runs
// on XCD b % 8) and take the HBM latency out of the next kernel's ramp (~2.3 us per launch of which the
// first-load latency is the larger part).
//   toucher<<<256>>>  : workgroup b loads 64 KiB at region[(b + shift) % 256] (shift 0: the XCD that will read it; 1: a
//                       neighbouring XCD; the data then sits in the Infinity Cache at best)
//   reader<<<256>>>   : the NEXT launch in the stream; workgroup b stamps the 100 MHz clock around loading region[b] (16 x 16 B
//                       per lane in flight) and reports the duration
// Regions rotate through 1 GiB so that nothing is left in the 256 MiB Infinity Cache from the previous repetition.
// Build + run: hipcc --offload-arch=gfx950 -O3 tools/probe/l2_across_launch_probe.hip -o /tmp/l2x && /tmp/l2x
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kRegion = 64 << 10;

__global__ __launch_bounds__(256) void toucher(const char *base, int shift, unsigned *sink) {
    const char *p = base + static_cast<size_t>((blockIdx.x + shift) % gridDim.x) * kRegion + threadIdx.x * 16;
    u32x4 a = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i) a ^= *reinterpret_cast<const u32x4 *>(p + i * 4096);
    if ((a[0] ^ a[1] ^ a[2] ^ a[3]) == 0x12345u) sink[0] = 1;
}

__global__ __launch_bounds__(256) void reader(const char *base, unsigned long long *dur, unsigned *sink, int nt) {
    const char *p = base + static_cast<size_t>(blockIdx.x) * kRegion + threadIdx.x * 16;
    const unsigned long long t0 = wall_clock64();
    u32x4 a = {0, 0, 0, 0};
    u32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        v[i] = nt ? __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + i * 4096)) : *reinterpret_cast<const u32x4 *>(p + i * 4096);
#pragma unroll
    for (int i = 0; i < 16; ++i) a ^= v[i];
    if ((a[0] ^ a[1] ^ a[2] ^ a[3]) == 0x12345u) sink[0] = 1;
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) dur[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main() {
    const size_t slab = 256 * static_cast<size_t>(kRegion);      // 16 MiB per repetition
    const int slabs = 64;                                          // 1 GiB
    char *buf; unsigned *sink; unsigned long long *dur;
    CK(hipMalloc(&buf, slab * slabs)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&dur, 256 * 8));
    CK(hipMemset(buf, 1, slab * slabs));
    std::vector<unsigned long long> h(256);
    const char *names[4] = {"cold (no toucher)", "touched by the same block id (same XCD)", "touched by block id + 1 (another XCD)",
                            "touched by the same block id, reader loads non-temporal"};
    for (int mode = 0; mode < 4; ++mode) {
        std::vector<double> med;
        for (int rep = 0; rep < 24; ++rep) {
            const char *r = buf + static_cast<size_t>((rep * 4 + mode) % slabs) * slab;
            if (mode) hipLaunchKernelGGL(toucher, dim3(256), dim3(256), 0, 0, r, mode == 2 ? 1 : 0, sink);
            hipLaunchKernelGGL(reader, dim3(256), dim3(256), 0, 0, r, dur, sink, mode == 3 ? 1 : 0);
            CK(hipMemcpy(h.data(), dur, 256 * 8, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            if (rep >= 4) med.push_back(h[128] / 100.0);
        }
        std::sort(med.begin(), med.end());
        printf("{\"mode\": \"%s\", \"reader_workgroup_us_median_of_medians\": %.2f, \"min\": %.2f, \"max\": %.2f, \"bytes\": %d}\n", names[mode],
               med[med.size() / 2], med.front(), med.back(), kRegion);
    }
    return 0;
}
