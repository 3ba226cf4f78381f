// Which part of "one 1024-thread workgroup per segment" matters: T, or resident workgroups per CU?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ char dyn[];
template <int T>
__global__ __launch_bounds__(T) void fill_coop(char* out, const int64_t* offb, int64_t nseg) {
    const u32x4 v = {0x3F800000u, 0u, 0x3F800000u, 0u};
    const int64_t s = blockIdx.x;
    if (threadIdx.x == 0 && offb[0] == 12345) dyn[0] = 1;
    const int64_t lo = (offb[s] + 127) & ~127ll, hi = (offb[s + 1] + 127) & ~127ll;
    char* seg = out + lo;
    const int n = (int)((hi - lo) >> 4);
    for (int g = threadIdx.x; g < n; g += T) *(u32x4*)(seg + (uint32_t)g * 16u) = v;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int T> int run(char* a, int64_t* d_off, int64_t nseg, double bytes, hipEvent_t e0, hipEvent_t e1) {
    CK(hipFuncSetAttribute((const void*)fill_coop<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int lds_kb : {1, 20, 40, 80, 160}) {
        float sum = 0;
        for (int r = 0; r < 10; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(fill_coop<T>, dim3((unsigned)nseg), dim3(T), lds_kb * 1024 - 64, 0, a, d_off, nseg);
            CK(hipGetLastError());
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) sum += ms;
        }
        int by_lds = 160 / lds_kb, by_waves = 2048 / T, wg = by_lds < by_waves ? by_lds : by_waves; if (wg > 8 && T >= 256) wg = 8;
        printf("T %4d  lds %3d KB -> ~%2d WG/CU (%2d waves) : %.3f ms  %.0f GB/s\n", T, lds_kb, wg, wg * T / 64, sum / 8, bytes / (sum / 8) / 1e6);
    }
    return 0;
}
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
    if (run<256>(a, d_off, nseg, bytes, e0, e1)) return 1;
    if (run<512>(a, d_off, nseg, bytes, e0, e1)) return 1;
    if (run<1024>(a, d_off, nseg, bytes, e0, e1)) return 1;
    return 0;
}
