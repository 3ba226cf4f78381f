// Balanced persistent schemes on the variable segment table (one process).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define V4 {0x3F800000u, 0u, 0x3F800000u, 0u}

__device__ __forceinline__ void fill_range(char* out, int64_t lo, int64_t hi, int lane) {   // [lo,hi) bytes, 8-B aligned
    const u32x4 v = V4;
    const int64_t g0 = (lo + 15) >> 4, g1 = hi >> 4;
    char* seg = out + g0 * 16;
    const int n = (int)(g1 - g0);
    for (int g = lane; g < n; g += 64) *(u32x4*)(seg + (uint32_t)g * 16u) = v;
    if (lane == 0 && (lo & 15)) *(uint64_t*)(out + lo) = 1;
    if (lane == 1 && (hi & 15) && g1 >= g0) *(uint64_t*)(out + (hi & ~15ll)) = 1;
}
// F1/F2: one wave per segment, line-owner ranges; grid = nseg/4 (non-persistent) or persistent static
__global__ __launch_bounds__(256) void segs(char* out, const int64_t* offb, int64_t nseg) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int64_t s = (int64_t)blockIdx.x * 4 + wave; s < nseg; s += (int64_t)gridDim.x * 4) {
        const int64_t lo = (offb[s] + 127) & ~127ll, hi = (offb[s + 1] + 127) & ~127ll;
        fill_range(out, lo, hi, lane);
    }
}
// F3: persistent, the workgroup dequeues CH consecutive groups of 4 segments from one of 8 counters
__global__ __launch_bounds__(256) void segs_dyn(char* out, const int64_t* offb, int64_t nseg, unsigned long long* ctr) {
    __shared__ long long base_s;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int shard = blockIdx.x & 7;
    const int64_t ngroups = (nseg + 3) / 4;
    for (;;) {
        if (threadIdx.x == 0) base_s = (long long)atomicAdd(&ctr[shard * 16], 1ull);
        __syncthreads();
        const int64_t grp = base_s * 8 + shard;              // groups interleaved over the shards
        __syncthreads();
        if (grp >= ngroups) return;
        const int64_t s = grp * 4 + wave;
        if (s < nseg) {
            const int64_t lo = (offb[s] + 127) & ~127ll, hi = (offb[s + 1] + 127) & ~127ll;
            fill_range(out, lo, hi, lane);
        }
    }
}
// F4/F5: persistent static windows (blocked-cyclic), segments inside a window walked via win_first
__global__ __launch_bounds__(256) void windows(char* out, const int64_t* offb, const int32_t* win_first, int64_t nseg,
                                               int64_t total, int win) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t nwin = (total + win - 1) / win;
    for (int64_t w = (int64_t)blockIdx.x * 4 + wave; w < nwin; w += (int64_t)gridDim.x * 4) {
        const int64_t wb = w * win, we = wb + win < total ? wb + win : total;
        int64_t s = win_first[w], lo = offb[s];
        while (s < nseg && lo < we) {
            const int64_t hi = offb[s + 1];
            fill_range(out, lo > wb ? lo : wb, hi < we ? hi : we, lane);
            lo = hi; ++s;
        }
    }
}
// F6: persistent, each wave dequeues windows from one of 8 counters
__global__ __launch_bounds__(256) void windows_dyn(char* out, const int64_t* offb, const int32_t* win_first, int64_t nseg,
                                                   int64_t total, int win, unsigned long long* ctr) {
    const int lane = threadIdx.x & 63;
    const int shard = blockIdx.x & 7;
    const int64_t nwin = (total + win - 1) / win;
    for (;;) {
        long long t = 0;
        if (lane == 0) t = (long long)atomicAdd(&ctr[shard * 16], 1ull);
        t = __builtin_amdgcn_readfirstlane((int)t);
        const int64_t w = (int64_t)t * 8 + shard;
        if (w >= nwin) return;
        const int64_t wb = w * win, we = wb + win < total ? wb + win : total;
        int64_t s = win_first[w], lo = offb[s];
        while (s < nseg && lo < we) {
            const int64_t hi = offb[s + 1];
            fill_range(out, lo > wb ? lo : wb, hi < we ? hi : we, lane);
            lo = hi; ++s;
        }
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
hipEvent_t e0, e1; double g_bytes; unsigned long long* d_ctr;
template <typename F> int timeit(const char* name, F launch, bool zero = false) {
    float sum = 0, best = 1e30f;
    for (int r = 0; r < 10; ++r) {
        if (zero) CK(hipMemsetAsync(d_ctr, 0, 8 * 128, 0));
        CK(hipEventRecord(e0)); launch(); CK(hipGetLastError()); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    printf("%-52s %.3f ms  %5.0f GB/s (best %5.0f)\n", name, sum / 8, g_bytes / (sum / 8) / 1e6, g_bytes / best / 1e6);
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
    int32_t* d_win; CK(hipMalloc(&d_win, 4 * 1000000)); CK(hipMalloc(&d_ctr, 8 * 128));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        timeit("hipMemsetAsync", [&] { (void)hipMemsetAsync(a, 1, total, 0); });
        timeit("F1 segs one per wave (current shape)", [&] { hipLaunchKernelGGL(segs, dim3(16384), dim3(256), 0, 0, a, d_off, nseg); });
        for (int G : {1024, 2048}) {
            char nm[80];
            snprintf(nm, 80, "F2 segs persistent static G=%d", G);
            timeit(nm, [&] { hipLaunchKernelGGL(segs, dim3(G), dim3(256), 0, 0, a, d_off, nseg); });
            snprintf(nm, 80, "F3 segs persistent dynamic (4 per dequeue) G=%d", G);
            timeit(nm, [&] { hipLaunchKernelGGL(segs_dyn, dim3(G), dim3(256), 0, 0, a, d_off, nseg, d_ctr); }, true);
            for (int win : {8192, 16384, 32768}) {
                const int64_t nwin = (total + win - 1) / win;
                std::vector<int32_t> wf(nwin);
                int64_t s = 0;
                for (int64_t w = 0; w < nwin; ++w) { while (off[s + 1] <= w * win) ++s; wf[w] = (int32_t)s; }
                CK(hipMemcpy(d_win, wf.data(), 4 * nwin, hipMemcpyHostToDevice));
                snprintf(nm, 80, "F4 windows %2dK persistent static G=%d", win / 1024, G);
                timeit(nm, [&] { hipLaunchKernelGGL(windows, dim3(G), dim3(256), 0, 0, a, d_off, d_win, nseg, total, win); });
                snprintf(nm, 80, "F6 windows %2dK persistent dynamic G=%d", win / 1024, G);
                timeit(nm, [&] { hipLaunchKernelGGL(windows_dyn, dim3(G), dim3(256), 0, 0, a, d_off, d_win, nseg, total, win, d_ctr); }, true);
            }
        }
    }
    return 0;
}
