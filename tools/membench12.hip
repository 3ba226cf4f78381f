// Same-box comparison: pure aligned 32 KB chunks vs windowed segment walk (with / without the narrow
// boundary stores) vs one wave per segment.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define V4 {0x3F800000u, 0u, 0x3F800000u, 0u}
__global__ __launch_bounds__(256) void chunks(char* out, int64_t bytes, int chunk) {
    const u32x4 v = V4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * 4, w = (int64_t)blockIdx.x * 4 + wave;
    for (int64_t c = w; c < bytes / chunk; c += nwaves) {
        char* p = out + c * chunk + lane * 16;
        for (int o = 0; o < chunk; o += 1024) *(u32x4*)(p + o) = v;
    }
}
template <int NARROW>
__global__ __launch_bounds__(256) void windows(char* out, const int64_t* offb, const int32_t* win_first, int64_t nseg,
                                               int64_t total, int win) {
    const u32x4 v = V4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t nwin = (total + win - 1) / win;
    for (int64_t w = (int64_t)blockIdx.x * 4 + wave; w < nwin; w += (int64_t)gridDim.x * 4) {
        const int64_t wb = w * win, we = wb + win < total ? wb + win : total;
        int64_t s = win_first[w], lo = offb[s];
        while (s < nseg && lo < we) {
            const int64_t hi = offb[s + 1];
            const int64_t a = lo > wb ? lo : wb, b = hi < we ? hi : we;
            const int64_t g0 = NARROW == 2 ? (a >> 4) : ((a + 15) >> 4), g1 = b >> 4;   // NARROW 2: the straddling group is
            char* seg = out + g0 * 16;                                                  // written whole by the later part
            const int n = (int)(g1 - g0);
            for (int g = lane; g < n; g += 64) *(u32x4*)(seg + (uint32_t)g * 16u) = v;
            if (NARROW == 1) {
                if (lane == 0 && (a & 15)) *(uint64_t*)(out + a) = 1;
                if (lane == 1 && (b & 15)) *(uint64_t*)(out + (b & ~15ll)) = 1;
            }
            lo = hi; ++s;
        }
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
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
hipEvent_t e0, e1; double g_bytes;
template <typename F> int timeit(const char* name, F launch) {
    float sum = 0, best = 1e30f;
    for (int r = 0; r < 10; ++r) {
        CK(hipEventRecord(e0)); launch(); CK(hipGetLastError()); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    printf("%-56s %.3f ms  %5.0f GB/s (best %5.0f)\n", name, sum / 8, g_bytes / (sum / 8) / 1e6, g_bytes / best / 1e6);
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
    const int64_t total = off[nseg] & ~32767ll; g_bytes = (double)total;
    char* a; CK(hipMalloc(&a, total + (1 << 20))); CK(hipMemset(a, 0, total));
    int64_t* d_off; CK(hipMalloc(&d_off, 8 * (nseg + 1))); CK(hipMemcpy(d_off, off.data(), 8 * (nseg + 1), hipMemcpyHostToDevice));
    int32_t* d_win; CK(hipMalloc(&d_win, 4 * 1000000));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int win = 32768;
    const int64_t nwin = (total + win - 1) / win;
    std::vector<int32_t> wf(nwin);
    { int64_t s = 0; for (int64_t w = 0; w < nwin; ++w) { while (off[s + 1] <= w * win) ++s; wf[w] = (int32_t)s; } }
    CK(hipMemcpy(d_win, wf.data(), 4 * nwin, hipMemcpyHostToDevice));
    printf("windows: %ld, per wave at G=2048: %.2f\n", (long)nwin, nwin / 8192.0);
    for (int rep = 0; rep < 3; ++rep) {
        timeit("hipMemsetAsync", [&] { (void)hipMemsetAsync(a, 1, total, 0); });
        timeit("segs one per wave G=16384 (current shape)", [&] { hipLaunchKernelGGL(segs, dim3(16384), dim3(256), 0, 0, a, d_off, nseg); });
        for (int G : {1024, 2048}) {
            char nm[96];
            snprintf(nm, 96, "pure 32K chunks persistent G=%d", G);
            timeit(nm, [&] { hipLaunchKernelGGL(chunks, dim3(G), dim3(256), 0, 0, a, total, win); });
            snprintf(nm, 96, "32K windows, segment walk, no boundary stores G=%d", G);
            timeit(nm, [&] { hipLaunchKernelGGL(windows<0>, dim3(G), dim3(256), 0, 0, a, d_off, d_win, nseg, total, win); });
            snprintf(nm, 96, "32K windows, segment walk, 8-B boundary stores G=%d", G);
            timeit(nm, [&] { hipLaunchKernelGGL(windows<1>, dim3(G), dim3(256), 0, 0, a, d_off, d_win, nseg, total, win); });
            snprintf(nm, 96, "32K windows, segment walk, whole straddling groups G=%d", G);
            timeit(nm, [&] { hipLaunchKernelGGL(windows<2>, dim3(G), dim3(256), 0, 0, a, d_off, d_win, nseg, total, win); });
        }
    }
    return 0;
}
