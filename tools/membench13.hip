// Cache-policy bits on the streaming stores (variable segments, one wave per segment, line-owner ranges).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void segs(char* out, const int64_t* offb, int64_t nseg) {
    const u32x4 v = {0x3F800000u, 0u, 0x3F800000u, 0u};
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * 4 + wave;
    if (s >= nseg) return;
    const int64_t lo = (offb[s] + 127) & ~127ll, hi = (offb[s + 1] + 127) & ~127ll;
    char* seg = out + lo;
    const int n = (int)((hi - lo) >> 4);
    for (int g = lane; g < n; g += 64) {
        u32x4* p = (u32x4*)(seg + (uint32_t)g * 16u);
        if (MODE == 0) *p = v;
        else if (MODE == 1) __builtin_nontemporal_store(v, p);
        else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
        else if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
        else if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int64_t nseg = 65536;
    std::vector<int64_t> off(nseg + 1); off[0] = 0; srand(1);
    for (int64_t s = 0; s < nseg; ++s) {
        double u = 0; for (int k = 0; k < 12; ++k) u += rand() / (double)RAND_MAX; u -= 6;
        int64_t persp = (int64_t)(74 + 15 * u); if (persp < 10) persp = 10; if (persp > 98) persp = 98;
        off[s + 1] = off[s] + persp * 392;
    }
    const int64_t total = off[nseg];
    char* a; CK(hipMalloc(&a, total + (1 << 20))); CK(hipMemset(a, 0, total));
    int64_t* d_off; CK(hipMalloc(&d_off, 8 * (nseg + 1))); CK(hipMemcpy(d_off, off.data(), 8 * (nseg + 1), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* nm[] = {"plain", "nt", "sc1", "sc0 sc1", "sc0"};
    for (int rep = 0; rep < 3; ++rep)
        for (int m = 0; m < 5; ++m) {
            float sum = 0;
            for (int r = 0; r < 10; ++r) {
                CK(hipEventRecord(e0));
                if (m == 0) hipLaunchKernelGGL(segs<0>, dim3(16384), dim3(256), 0, 0, a, d_off, nseg);
                if (m == 1) hipLaunchKernelGGL(segs<1>, dim3(16384), dim3(256), 0, 0, a, d_off, nseg);
                if (m == 2) hipLaunchKernelGGL(segs<2>, dim3(16384), dim3(256), 0, 0, a, d_off, nseg);
                if (m == 3) hipLaunchKernelGGL(segs<3>, dim3(16384), dim3(256), 0, 0, a, d_off, nseg);
                if (m == 4) hipLaunchKernelGGL(segs<4>, dim3(16384), dim3(256), 0, 0, a, d_off, nseg);
                CK(hipGetLastError()); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) sum += ms;
            }
            printf("%-8s %.3f ms  %.0f GB/s\n", nm[m], sum / 8, total / (sum / 8) / 1e6);
        }
    return 0;
}
