// Which property of the perspective-write store stream costs bandwidth?  Progressive variants of a
// per-wave segment fill.  Build: hipcc -O3 --offload-arch=gfx950 tools/membench2.hip -o tools/membench2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// one wave per segment; seg start/len (in 16-byte groups) from an offsets table (scalar loads);
// flags: 1 = dependent scalar load chain first (prologue), 2 = LDS traffic + barrier per segment
template <int FLAGS>
__global__ __launch_bounds__(256) void fill_segs(u32x4* out, const int64_t* offs16, int64_t nseg, const uint64_t* junk) {
    __shared__ uint32_t lds[4][128];
    const u32x4 v = {0x3F800000u, 0u, 0x3F800000u, 0u};
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int64_t s = (int64_t)blockIdx.x * 4 + wave; s < nseg; s += (int64_t)gridDim.x * 4) {
        uint64_t j = 0;
        if (FLAGS & 1) { j = junk[s]; j = junk[(j & 1023) + s / 2]; }
        const int64_t lo = offs16[s], hi = offs16[s + 1] + (int64_t)(j >> 63);
        if (FLAGS & 2) { lds[wave][lane] = (uint32_t)lo; lds[wave][lane + 64] = (uint32_t)hi; __builtin_amdgcn_wave_barrier(); }
        char* seg = (char*)(out + lo);
        const int n = (int)(hi - lo);
        for (int g = lane; g < n; g += 64) {
            u32x4 vv = v;
            if (FLAGS & 2) vv.x = lds[wave][(g * 7) & 127];
            *(u32x4*)(seg + (uint32_t)g * 16u) = vv;
        }
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const int64_t nseg = 65536;
    const int64_t cap16 = (int64_t)3e9 / 16;
    u32x4* a; CK(hipMalloc(&a, cap16 * 16)); CK(hipMemset(a, 0, cap16 * 16));
    uint64_t* junk; CK(hipMalloc(&junk, 8 * (nseg + 2048))); CK(hipMemset(junk, 0, 8 * (nseg + 2048)));
    int64_t* d_off; CK(hipMalloc(&d_off, 8 * (nseg + 1)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    srand(1);
    for (int shape = 0; shape < 4; ++shape) {
        // shape 0: fixed 29 KB aligned to 64 B; 1: fixed 29000 B; 2: variable ~N(29KB, 6KB) aligned 64 B; 3: variable, 16-B aligned only
        std::vector<int64_t> off(nseg + 1);
        off[0] = 0;
        for (int64_t s = 0; s < nseg; ++s) {
            int64_t len16;
            if (shape == 0) len16 = 29 * 64;                     // 29 KiB
            else if (shape == 1) len16 = 29 * 64 + 1;            // 16-B aligned only
            else {
                double u = 0; for (int k = 0; k < 12; ++k) u += rand() / (double)RAND_MAX; u -= 6;     // ~N(0,1)
                int64_t persp = (int64_t)(74 + 15 * u); if (persp < 10) persp = 10; if (persp > 98) persp = 98;
                len16 = shape == 2 ? persp * 24 + (persp * 24) % 4 * 0 : (persp * 98) / 4;            // 392 B per perspective
                if (shape == 2) len16 = (len16 + 3) & ~3ll;
            }
            off[s + 1] = off[s] + len16;
        }
        if (off[nseg] > cap16) { printf("too big\n"); return 1; }
        CK(hipMemcpy(d_off, off.data(), 8 * (nseg + 1), hipMemcpyHostToDevice));
        const double bytes = 16.0 * off[nseg];
        for (int persistent = 0; persistent < 2; ++persistent) {
            for (int flags = 0; flags < 4; ++flags) {
                const int grid = persistent ? 2048 : (int)(nseg / 4);
                float sum = 0;
                for (int r = 0; r < 12; ++r) {
                    CK(hipEventRecord(e0));
                    if (flags == 0) hipLaunchKernelGGL(fill_segs<0>, dim3(grid), dim3(256), 0, 0, a, d_off, nseg, junk);
                    if (flags == 1) hipLaunchKernelGGL(fill_segs<1>, dim3(grid), dim3(256), 0, 0, a, d_off, nseg, junk);
                    if (flags == 2) hipLaunchKernelGGL(fill_segs<2>, dim3(grid), dim3(256), 0, 0, a, d_off, nseg, junk);
                    if (flags == 3) hipLaunchKernelGGL(fill_segs<3>, dim3(grid), dim3(256), 0, 0, a, d_off, nseg, junk);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r >= 2) sum += ms;
                }
                printf("shape %d %-10s flags %d : %.3f ms  %.0f GB/s  (%.2f GB)\n", shape, persistent ? "persistent" : "1seg/wave", flags,
                       sum / 10, bytes / (sum / 10) / 1e6, bytes / 1e9);
            }
        }
    }
    return 0;
}
