// Cooperative streams: a workgroup of T threads writes its lattices' segments one after another, all
// waves on the same segment (thread t writes 16-byte groups t, t+T, ...).  Streams in flight =
// resident workgroups, waves in flight stay at 32/CU.  PER = segments per workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int T>
__global__ __launch_bounds__(T) void fill_coop(char* out, const int64_t* offb, int64_t nseg, int per, int sync) {
    const u32x4 v = {0x3F800000u, 0u, 0x3F800000u, 0u};
    for (int k = 0; k < per; ++k) {
        const int64_t s = (int64_t)blockIdx.x * per + k;
        if (s >= nseg) return;
        const int64_t lo = (offb[s] + 127) & ~127ll, hi = (offb[s + 1] + 127) & ~127ll;
        char* seg = out + lo;
        const int n = (int)((hi - lo) >> 4);
        for (int g = threadIdx.x; g < n; g += T) *(u32x4*)(seg + (uint32_t)g * 16u) = v;
        if (sync) __syncthreads();
    }
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
    for (int T : {64, 128, 256, 512, 1024})
        for (int per : {1, 4, 16})
            for (int sync : {0, 1}) {
                float sum = 0;
                for (int r = 0; r < 10; ++r) {
                    CK(hipEventRecord(e0));
                    const int grid = (int)((nseg + per - 1) / per);
                    if (T == 64) hipLaunchKernelGGL(fill_coop<64>, dim3(grid), dim3(T), 0, 0, a, d_off, nseg, per, sync);
                    if (T == 128) hipLaunchKernelGGL(fill_coop<128>, dim3(grid), dim3(T), 0, 0, a, d_off, nseg, per, sync);
                    if (T == 256) hipLaunchKernelGGL(fill_coop<256>, dim3(grid), dim3(T), 0, 0, a, d_off, nseg, per, sync);
                    if (T == 512) hipLaunchKernelGGL(fill_coop<512>, dim3(grid), dim3(T), 0, 0, a, d_off, nseg, per, sync);
                    if (T == 1024) hipLaunchKernelGGL(fill_coop<1024>, dim3(grid), dim3(T), 0, 0, a, d_off, nseg, per, sync);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r >= 2) sum += ms;
                }
                printf("T %4d per %2d sync %d : %.3f ms  %.0f GB/s\n", T, per, sync, sum / 8, bytes / (sum / 8) / 1e6);
            }
    return 0;
}
