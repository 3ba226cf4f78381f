// Occupancy sweep for the line-owner store stream (variable 392-B-unit segments, one wave per segment):
// dynamic LDS padding limits resident workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ char dyn[];
template <int WAVES>
__global__ void fill_segs(char* out, const int64_t* offb, int64_t nseg, int unroll) {
    const u32x4 v = {0x3F800000u, 0u, 0x3F800000u, 0u};
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * WAVES + wave;
    if (s >= nseg) return;
    if (threadIdx.x == 0 && offb[0] == 12345) dyn[0] = 1;
    int64_t lo = (offb[s] + 127) & ~127ll, hi = (offb[s + 1] + 127) & ~127ll;
    char* seg = out + lo;
    const int n = (int)((hi - lo) >> 4);
    for (int g = lane; g < n; g += 64) *(u32x4*)(seg + (uint32_t)g * 16u) = v;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int64_t nseg = 65536;
    char* a; CK(hipMalloc(&a, (int64_t)3e9)); CK(hipMemset(a, 0, (int64_t)3e9));
    int64_t* d_off; CK(hipMalloc(&d_off, 8 * (nseg + 1)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    srand(1);
    std::vector<int64_t> off(nseg + 1); off[0] = 0;
    for (int64_t s = 0; s < nseg; ++s) {
        double u = 0; for (int k = 0; k < 12; ++k) u += rand() / (double)RAND_MAX; u -= 6;
        int64_t persp = (int64_t)(74 + 15 * u); if (persp < 10) persp = 10; if (persp > 98) persp = 98;
        off[s + 1] = off[s] + persp * 392;
    }
    CK(hipMemcpy(d_off, off.data(), 8 * (nseg + 1), hipMemcpyHostToDevice));
    const double bytes = (double)off[nseg];
    CK(hipFuncSetAttribute((const void*)fill_segs<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)fill_segs<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)fill_segs<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int waves : {1, 4, 16}) {
        for (int lds_kb : {1, 20, 40, 80, 160}) {          // -> min(limit by waves, 160/lds_kb) WGs per CU
            float sum = 0;
            for (int r = 0; r < 10; ++r) {
                CK(hipEventRecord(e0));
                const int grid = (int)((nseg + waves - 1) / waves);
                if (waves == 1) hipLaunchKernelGGL(fill_segs<1>, dim3(grid), dim3(64), lds_kb * 1024 - 64, 0, a, d_off, nseg, 1);
                if (waves == 4) hipLaunchKernelGGL(fill_segs<4>, dim3(grid), dim3(256), lds_kb * 1024 - 64, 0, a, d_off, nseg, 1);
                if (waves == 16) hipLaunchKernelGGL(fill_segs<16>, dim3(grid), dim3(1024), lds_kb * 1024 - 64, 0, a, d_off, nseg, 1);
                CK(hipGetLastError());
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 2) sum += ms;
            }
            int wg_per_cu = 160 / lds_kb; int by_waves = 32 / waves; if (wg_per_cu > by_waves) wg_per_cu = by_waves;
            printf("waves/WG %2d  lds %3d KB  -> ~%2d waves/CU : %.3f ms  %.0f GB/s\n", waves, lds_kb, wg_per_cu * waves, sum / 8,
                   bytes / (sum / 8) / 1e6);
        }
    }
    return 0;
}
