// Shared boundary lines vs instruction alignment: variable, 8-byte-aligned segments (392 B units).
// mode 0: lane 0 at segment start (groups on the global 16-B grid)     [what the kernel did first]
// mode 1: same bytes, loop anchored on 128-B boundaries (first iteration masked)
// mode 2: line-owner ranges: wave s writes [ceilL(lo_s), ceilL(hi_s)) with L = 128 B -> no line is shared
// mode 3: line-owner ranges with L = 64 B
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void fill_segs(char* out, const int64_t* offb, int64_t nseg) {
    const u32x4 v = {0x3F800000u, 0u, 0x3F800000u, 0u};
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * 4 + wave;
    if (s >= nseg) return;
    int64_t lo = offb[s], hi = offb[s + 1];                 // bytes, multiples of 8
    if (MODE >= 2) {
        const int64_t L = MODE == 2 ? 128 : 64;
        lo = (lo + L - 1) & ~(L - 1);
        hi = (hi + L - 1) & ~(L - 1);
        if (s == nseg - 1) hi = offb[nseg] & ~15ll;
    }
    int64_t g0 = (lo + 15) >> 4, g1 = hi >> 4;             // interior 16-B groups
    if (MODE == 1) {
        const int64_t ga = g0 & ~7ll;
        const int skip = (int)(g0 - ga);
        char* seg = out + ga * 16;
        const int n = (int)(g1 - ga);
        for (int g = lane < skip ? lane + 64 : lane; g < n; g += 64) *(u32x4*)(seg + (uint32_t)g * 16u) = v;
    } else {
        char* seg = out + g0 * 16;
        const int n = (int)(g1 - g0);
        for (int g = lane; g < n; g += 64) *(u32x4*)(seg + (uint32_t)g * 16u) = v;
    }
    if (MODE < 2) {                                         // 8-byte head / tail pieces
        if (lane == 0 && (lo & 15)) *(uint64_t*)(out + lo) = 1;
        if (lane == 1 && (hi & 15)) *(uint64_t*)(out + (hi & ~15ll)) = 1;
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int64_t nseg = 65536;
    const int64_t cap = (int64_t)3e9;
    char* a; CK(hipMalloc(&a, cap)); CK(hipMemset(a, 0, cap));
    int64_t* d_off; CK(hipMalloc(&d_off, 8 * (nseg + 1)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    srand(1);
    std::vector<int64_t> off(nseg + 1);
    off[0] = 0;
    for (int64_t s = 0; s < nseg; ++s) {
        double u = 0; for (int k = 0; k < 12; ++k) u += rand() / (double)RAND_MAX; u -= 6;
        int64_t persp = (int64_t)(74 + 15 * u); if (persp < 10) persp = 10; if (persp > 98) persp = 98;
        off[s + 1] = off[s] + persp * 392;
    }
    CK(hipMemcpy(d_off, off.data(), 8 * (nseg + 1), hipMemcpyHostToDevice));
    const double bytes = (double)off[nseg];
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
        float sum = 0;
        for (int r = 0; r < 12; ++r) {
            CK(hipEventRecord(e0));
            const int grid = (int)(nseg / 4);
            if (mode == 0) hipLaunchKernelGGL(fill_segs<0>, dim3(grid), dim3(256), 0, 0, a, d_off, nseg);
            if (mode == 1) hipLaunchKernelGGL(fill_segs<1>, dim3(grid), dim3(256), 0, 0, a, d_off, nseg);
            if (mode == 2) hipLaunchKernelGGL(fill_segs<2>, dim3(grid), dim3(256), 0, 0, a, d_off, nseg);
            if (mode == 3) hipLaunchKernelGGL(fill_segs<3>, dim3(grid), dim3(256), 0, 0, a, d_off, nseg);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) sum += ms;
        }
        printf("mode %d : %.3f ms  %.0f GB/s  (%.2f GB)\n", mode, sum / 10, bytes / (sum / 10) / 1e6, bytes / 1e9);
    }
    return 0;
}
