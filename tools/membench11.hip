// "Moving front" with lattice granularity: G workgroups of T threads; workgroup b streams segments
// b, b+G, b+2G, ... with all its threads cooperating on one segment at a time.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define V4 {0x3F800000u, 0u, 0x3F800000u, 0u}
template <int T>
__global__ __launch_bounds__(T) void front(char* out, const int64_t* offb, int64_t nseg, int sync) {
    const u32x4 v = V4;
    for (int64_t s = blockIdx.x; s < nseg; s += gridDim.x) {
        const int64_t lo = (offb[s] + 127) & ~127ll, hi = (offb[s + 1] + 127) & ~127ll;
        char* seg = out + lo;
        const int n = (int)((hi - lo) >> 4);
        for (int g = threadIdx.x; g < n; g += T) *(u32x4*)(seg + (uint32_t)g * 16u) = v;
        if (sync) __syncthreads();
    }
}
__global__ __launch_bounds__(256) void segs(char* out, const int64_t* offb, int64_t nseg) {
    const u32x4 v = V4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int64_t s = (int64_t)blockIdx.x * 4 + wave; s < nseg; s += (int64_t)gridDim.x * 4) {
        const int64_t lo = (offb[s] + 127) & ~127ll, hi = (offb[s + 1] + 127) & ~127ll;
        char* seg = out + lo;
        const int n = (int)((hi - lo) >> 4);
        for (int g = lane; g < n; g += 64) *(u32x4*)(seg + (uint32_t)g * 16u) = v;
    }
}
__global__ __launch_bounds__(256) void flat(u32x4* out, int64_t n16) {
    const u32x4 v = V4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) out[i] = v;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
hipEvent_t e0, e1; double g_bytes;
template <typename F> int timeit(const char* name, F launch) {
    float sum = 0, best = 1e30f;
    for (int r = 0; r < 10; ++r) {
        CK(hipEventRecord(e0)); launch(); CK(hipGetLastError()); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    printf("%-44s %.3f ms  %5.0f GB/s (best %5.0f)\n", name, sum / 8, g_bytes / (sum / 8) / 1e6, g_bytes / best / 1e6);
    return 0;
}
int main() {
    const int64_t nseg = 65536;
    std::vector<int64_t> off(nseg + 1); off[0] = 0; srand(1);
    for (int64_t s = 0; s < nseg; ++s) {
        double u = 0; for (int k = 0; k < 12; ++k) u += rand() / (double)RAND_MAX; u -= 6;
        int64_t persp = (int64_t)(74 + 15 * u); if (persp < 10) persp = 10; if (persp > 98) persp = 98;
        off[s + 1] = off[s] + persp * 392;
    }
    const int64_t total = off[nseg]; g_bytes = (double)total;
    char* a; CK(hipMalloc(&a, total + (1 << 20))); CK(hipMemset(a, 0, total));
    int64_t* d_off; CK(hipMalloc(&d_off, 8 * (nseg + 1))); CK(hipMemcpy(d_off, off.data(), 8 * (nseg + 1), hipMemcpyHostToDevice));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        timeit("hipMemsetAsync", [&] { (void)hipMemsetAsync(a, 1, total, 0); });
        timeit("flat G=256", [&] { hipLaunchKernelGGL(flat, dim3(256), dim3(256), 0, 0, (u32x4*)a, total / 16); });
        timeit("segs one per wave G=16384 (current)", [&] { hipLaunchKernelGGL(segs, dim3(16384), dim3(256), 0, 0, a, d_off, nseg); });
        for (int G : {256, 512, 1024}) {
            char nm[64];
            for (int sync : {0, 1}) {
                snprintf(nm, 64, "front T=256  G=%d sync=%d", G, sync); timeit(nm, [&] { hipLaunchKernelGGL(front<256>, dim3(G), dim3(256), 0, 0, a, d_off, nseg, sync); });
                snprintf(nm, 64, "front T=512  G=%d sync=%d", G, sync); timeit(nm, [&] { hipLaunchKernelGGL(front<512>, dim3(G), dim3(512), 0, 0, a, d_off, nseg, sync); });
                snprintf(nm, 64, "front T=1024 G=%d sync=%d", G, sync); timeit(nm, [&] { hipLaunchKernelGGL(front<1024>, dim3(G), dim3(1024), 0, 0, a, d_off, nseg, sync); });
            }
        }
    }
    return 0;
}
