// Fixed aligned windows vs lattice-sized segments.  Same variable 392-B-unit segment table as membench3.
// mode 0: one wave per segment, line-owner ranges (what k_persp_write does now)
// mode 1: one wave per WIN-byte window of the stream; inside the window the wave walks the segments that
//         overlap it (scalar loads of the offsets), writes whole 16-byte groups of each part and the
//         8-byte pieces at segment boundaries with narrow stores
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void seg_owner(char* out, const int64_t* offb, int64_t nseg) {
    const u32x4 v = {0x3F800000u, 0u, 0x3F800000u, 0u};
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * 4 + wave;
    if (s >= nseg) return;
    const int64_t lo = (offb[s] + 127) & ~127ll, hi = (offb[s + 1] + 127) & ~127ll;
    char* seg = out + lo;
    const int n = (int)((hi - lo) >> 4);
    for (int g = lane; g < n; g += 64) *(u32x4*)(seg + (uint32_t)g * 16u) = v;
}

__global__ __launch_bounds__(256) void windows(char* out, const int64_t* offb, const int32_t* win_first, int64_t nseg,
                                               int64_t total, int win) {
    const u32x4 v = {0x3F800000u, 0u, 0x3F800000u, 0u};
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;
    const int64_t wb = w * win;
    if (wb >= total) return;
    const int64_t we = wb + win < total ? wb + win : total;
    int64_t s = win_first[w];
    int64_t lo = offb[s];
    while (s < nseg && lo < we) {
        const int64_t hi = offb[s + 1];
        const int64_t a = lo > wb ? lo : wb, b = hi < we ? hi : we;      // this segment's part of the window
        const int64_t g0 = (a + 15) >> 4, g1 = b >> 4;
        char* seg = out + g0 * 16;
        const int n = (int)(g1 - g0);
        for (int g = lane; g < n; g += 64) *(u32x4*)(seg + (uint32_t)g * 16u) = v;
        if (lane == 0 && (a & 15)) *(uint64_t*)(out + a) = 1;           // 8-byte pieces at the segment's ends
        if (lane == 1 && (b & 15)) *(uint64_t*)(out + (b & ~15ll)) = 1;
        lo = hi;
        ++s;
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int64_t nseg = 65536;
    char* a; CK(hipMalloc(&a, (int64_t)3e9)); CK(hipMemset(a, 0, (int64_t)3e9));
    int64_t* d_off; CK(hipMalloc(&d_off, 8 * (nseg + 1)));
    int32_t* d_win; CK(hipMalloc(&d_win, 4 * 1000000));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    srand(1);
    std::vector<int64_t> off(nseg + 1); off[0] = 0;
    for (int64_t s = 0; s < nseg; ++s) {
        double u = 0; for (int k = 0; k < 12; ++k) u += rand() / (double)RAND_MAX; u -= 6;
        int64_t persp = (int64_t)(74 + 15 * u); if (persp < 10) persp = 10; if (persp > 98) persp = 98;
        off[s + 1] = off[s] + persp * 392;
    }
    CK(hipMemcpy(d_off, off.data(), 8 * (nseg + 1), hipMemcpyHostToDevice));
    const int64_t total = off[nseg];
    for (int rep = 0; rep < 2; ++rep) {
        float sum = 0;
        for (int r = 0; r < 10; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(seg_owner, dim3((unsigned)(nseg / 4)), dim3(256), 0, 0, a, d_off, nseg);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) sum += ms;
        }
        printf("one wave per segment (line-owner)      : %.3f ms  %.0f GB/s\n", sum / 8, total / (sum / 8) / 1e6);
        for (int win : {8192, 16384, 32768, 65536, 131072}) {
            const int64_t nwin = (total + win - 1) / win;
            std::vector<int32_t> wf(nwin);
            int64_t s = 0;
            for (int64_t w = 0; w < nwin; ++w) { while (off[s + 1] <= w * win) ++s; wf[w] = (int32_t)s; }
            CK(hipMemcpy(d_win, wf.data(), 4 * nwin, hipMemcpyHostToDevice));
            sum = 0;
            for (int r = 0; r < 10; ++r) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(windows, dim3((unsigned)((nwin + 3) / 4)), dim3(256), 0, 0, a, d_off, d_win, nseg, total, win);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) sum += ms;
            }
            printf("one wave per %3d KB window             : %.3f ms  %.0f GB/s\n", win / 1024, sum / 8, total / (sum / 8) / 1e6);
        }
    }
    return 0;
}
