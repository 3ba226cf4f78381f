// Does the store SHAPE still matter once the kernel looks like the real one (round-2 stack write: a
// setup phase per unit of work, then an expansion loop fed from LDS: ds_read2_b32 + shift + 4x(bfe,and)
// + global_store_dwordx4, no vector loads)?  Same process, same buffer:
//   segs   one wave per variable ~29 KB segment (the production shape), S setup instructions first
//   win    persistent 256-thread workgroups over aligned WIN-byte windows: per window every wave runs S
//          setup instructions, the window's bits go to LDS, barrier, each wave expands a quarter
//   winw   persistent single waves over aligned WIN-byte windows (no barrier)
//   fill   the same windows with constant data and no LDS (ceiling of the shape)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t setup_work(uint32_t x, int S) {       // S dependent VALU instructions
    for (int i = 0; i < S; ++i) x = x * 1664525u + 1013904223u;
    return x;
}
__device__ __forceinline__ u32x4 expand4(uint32_t wb) {
    u32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = (uint32_t)(((int32_t)(wb << (31 - k))) >> 31) & 0x3F800000u;
    return r;
}
__device__ __forceinline__ void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// one wave per segment; bits of the segment (1 per f32 element) in the wave's LDS
__global__ __launch_bounds__(256) void segs(char* out, const int64_t* offb, int64_t nseg, int S) {
    __shared__ uint32_t bits[4][320];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * 4 + wave;
    if (s >= nseg) return;
    const int64_t lo = offb[s], hi = offb[s + 1];                          // multiples of 128 bytes
    uint32_t x = setup_work((uint32_t)s * 64 + lane, S);
    const int ndw = (int)((hi - lo) / 128) + 2;                            // one bit per element
    for (int i = lane; i < ndw; i += 64) bits[wave][i] = x + i;
    lds_sync();
    const int n_groups = (int)((hi - lo) >> 4);
    char* seg = out + lo;
    const uint32_t rel0 = lane * 4, ph = rel0 & 31;
    const uint32_t* bp = bits[wave] + (rel0 >> 5);
    for (int g = lane; g < n_groups; g += 64) {
        const uint32_t wb = (uint32_t)(((((uint64_t)bp[1]) << 32) | bp[0]) >> ph);
        *(u32x4*)(seg + (uint32_t)g * 16u) = expand4(wb);
        bp += 8;
    }
}

// persistent workgroups over aligned windows of WIN bytes (4 waves share a window)
template <int WIN>
__global__ __launch_bounds__(256) void win(char* out, int64_t nwin, int S) {
    constexpr int DW = WIN / 128;                                          // bits of the window, in dwords
    __shared__ uint32_t bits[DW + 8];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int64_t w = blockIdx.x; w < nwin; w += gridDim.x) {
        uint32_t x = setup_work((uint32_t)w * 256 + threadIdx.x, S);
        for (int i = threadIdx.x; i < DW + 2; i += 256) bits[i] = x + i;
        __syncthreads();
        char* seg = out + w * WIN + wave * (WIN / 4);
        const uint32_t rel0 = wave * (WIN / 16) + lane * 4, ph = rel0 & 31;
        const uint32_t* bp = bits + (rel0 >> 5);
#pragma unroll 2
        for (int g = lane; g < WIN / 64; g += 64) {
            const uint32_t wb = (uint32_t)(((((uint64_t)bp[1]) << 32) | bp[0]) >> ph);
            *(u32x4*)(seg + (uint32_t)g * 16u) = expand4(wb);
            bp += 8;
        }
        __syncthreads();
    }
}

// persistent single waves over aligned windows of WIN bytes
template <int WIN>
__global__ __launch_bounds__(256) void winw(char* out, int64_t nwin, int S) {
    constexpr int DW = WIN / 128;
    __shared__ uint32_t bits[4][DW + 8];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int64_t w = (int64_t)blockIdx.x * 4 + wave; w < nwin; w += (int64_t)gridDim.x * 4) {
        uint32_t x = setup_work((uint32_t)w * 64 + lane, S);
        for (int i = lane; i < DW + 2; i += 64) bits[wave][i] = x + i;
        lds_sync();
        char* seg = out + w * WIN;
        const uint32_t rel0 = lane * 4, ph = rel0 & 31;
        const uint32_t* bp = bits[wave] + (rel0 >> 5);
        for (int g = lane; g < WIN / 16; g += 64) {
            const uint32_t wb = (uint32_t)(((((uint64_t)bp[1]) << 32) | bp[0]) >> ph);
            *(u32x4*)(seg + (uint32_t)g * 16u) = expand4(wb);
            bp += 8;
        }
        lds_sync();
    }
}

template <int WIN>
__global__ __launch_bounds__(256) void fill(char* out, int64_t nwin) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32x4 v = {0x3F800000u, 0u, 0x3F800000u, 0u};
    for (int64_t w = blockIdx.x; w < nwin; w += gridDim.x) {
        char* seg = out + w * WIN + wave * (WIN / 4);
        for (int g = lane; g < WIN / 64; g += 64) *(u32x4*)(seg + (uint32_t)g * 16u) = v;
    }
}

hipEvent_t e0, e1;
double g_bytes;
template <typename F> int timeit(const char* name, F launch) {
    float sum = 0, best = 1e30f;
    for (int r = 0; r < 8; ++r) {
        CK(hipEventRecord(e0)); launch(); CK(hipGetLastError()); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    printf("%-52s %.3f ms  %5.0f GB/s (best %5.0f)\n", name, sum / 6, g_bytes / (sum / 6) * 1e-6, g_bytes / best * 1e-6);
    fflush(stdout);
    return 0;
}

int main() {
    const int64_t nseg = 65536;
    std::vector<int64_t> offb(nseg + 1);
    srand(7);
    offb[0] = 0;
    for (int64_t s = 0; s < nseg; ++s) {                                   // ~73 perspectives x 392 B, varying 50..96
        const int n = 50 + rand() % 47;
        offb[s + 1] = offb[s] + (((int64_t)n * 392 + 127) & ~127ll);
    }
    const int64_t total = (offb[nseg] + 65535) & ~65535ll;
    g_bytes = (double)total;
    char* out; int64_t* d_off;
    CK(hipMalloc(&out, total)); CK(hipMalloc(&d_off, (nseg + 1) * 8));
    CK(hipMemcpy(d_off, offb.data(), (nseg + 1) * 8, hipMemcpyHostToDevice));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("total %.3f GB in %ld segments\n", total * 1e-9, (long)nseg);
    // burst (one launch between two events, host sync in between) vs sustained (200 launches back to back)
    auto sustained = [&](const char* name, auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 200; ++i) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-52s %.3f ms  %5.0f GB/s   (200 launches back to back)\n", name, ms / 200, g_bytes / (ms / 200) * 1e-6);
        fflush(stdout);
    };
    sustained("SUSTAINED hipMemsetAsync", [&] { (void)hipMemsetAsync(out, 0, total, 0); });
    sustained("SUSTAINED fill 32 KB windows, G=2048", [&] { hipLaunchKernelGGL(fill<32768>, dim3(2048), dim3(256), 0, 0, out, total / 32768); });
    sustained("SUSTAINED winw 32 KB windows (wave), G=2048, S=0", [&] { hipLaunchKernelGGL(winw<32768>, dim3(2048), dim3(256), 0, 0, out, total / 32768, 0); });
    sustained("SUSTAINED segs: one wave per segment, S=0", [&] { hipLaunchKernelGGL(segs, dim3(nseg / 4), dim3(256), 0, 0, out, d_off, nseg, 0); });
    sustained("SUSTAINED segs: one wave per segment, S=300", [&] { hipLaunchKernelGGL(segs, dim3(nseg / 4), dim3(256), 0, 0, out, d_off, nseg, 300); });
    sustained("SUSTAINED segs: one wave per segment, S=1000", [&] { hipLaunchKernelGGL(segs, dim3(nseg / 4), dim3(256), 0, 0, out, d_off, nseg, 1000); });
    for (int rep = 0; rep < 1; ++rep) {
        timeit("hipMemsetAsync", [&] { (void)hipMemsetAsync(out, 0, total, 0); });
        for (int S : {0, 300, 1000}) {
            char name[96];
            snprintf(name, sizeof name, "segs: one wave per segment, S=%d", S);
            timeit(name, [&] { hipLaunchKernelGGL(segs, dim3(nseg / 4), dim3(256), 0, 0, out, d_off, nseg, S); });
        }
        for (int G : {1024, 2048, 4096}) {
            char name[96];
            snprintf(name, sizeof name, "fill  32 KB windows, G=%d", G);
            timeit(name, [&] { hipLaunchKernelGGL(fill<32768>, dim3(G), dim3(256), 0, 0, out, total / 32768); });
            for (int S : {0, 300, 1000}) {
                snprintf(name, sizeof name, "win   32 KB windows (WG), G=%d, S=%d", G, S);
                timeit(name, [&] { hipLaunchKernelGGL(win<32768>, dim3(G), dim3(256), 0, 0, out, total / 32768, S); });
            }
            snprintf(name, sizeof name, "win   64 KB windows (WG), G=%d, S=300", G);
            timeit(name, [&] { hipLaunchKernelGGL(win<65536>, dim3(G), dim3(256), 0, 0, out, total / 65536, 300); });
            snprintf(name, sizeof name, "win   16 KB windows (WG), G=%d, S=300", G);
            timeit(name, [&] { hipLaunchKernelGGL(win<16384>, dim3(G), dim3(256), 0, 0, out, total / 16384, 300); });
            for (int S : {0, 300}) {
                snprintf(name, sizeof name, "winw  32 KB windows (wave), G=%d, S=%d", G, S);
                timeit(name, [&] { hipLaunchKernelGGL(winw<32768>, dim3(G), dim3(256), 0, 0, out, total / 32768, S); });
                snprintf(name, sizeof name, "winw   8 KB windows (wave), G=%d, S=%d", G, S);
                timeit(name, [&] { hipLaunchKernelGGL(winw<8192>, dim3(G), dim3(256), 0, 0, out, total / 8192, S); });
            }
        }
    }
    return 0;
}
