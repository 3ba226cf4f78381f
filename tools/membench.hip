// Store-bandwidth ceiling on this GPU for the access shapes the perspective write uses.
// Build: hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o tools/membench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// (1) flat grid-stride 16 B/lane stores
__global__ __launch_bounds__(256) void fill_flat(u32x4* out, int64_t n16) {
    const u32x4 v = {0x3F800000u, 0u, 0x3F800000u, 0u};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) out[i] = v;
}
// (2) one wave owns a contiguous segment of seg16 16-byte groups (the perspective-write shape)
__global__ __launch_bounds__(256) void fill_seg(u32x4* out, int64_t nseg, int seg16) {
    const u32x4 v = {0x3F800000u, 0u, 0x3F800000u, 0u};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t s = (int64_t)blockIdx.x * 4 + wave; s < nseg; s += (int64_t)gridDim.x * 4) {
        u32x4* p = out + s * seg16;
        for (int g = lane; g < seg16; g += 64) p[g] = v;
    }
}
// (3) copy, 16 B/lane
__global__ __launch_bounds__(256) void copy_flat(const u32x4* in, u32x4* out, int64_t n16) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) out[i] = in[i];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int64_t bytes = (argc > 1 ? atoll(argv[1]) : 2000) * 1000000ll;
    const int64_t n16 = bytes / 16;
    u32x4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        for (int variant = 0; variant < 4; ++variant) {
            float best = 1e30f, sum = 0;
            for (int r = 0; r < reps + 2; ++r) {
                CK(hipEventRecord(e0));
                if (variant == 0) hipLaunchKernelGGL(fill_flat, dim3(grid), dim3(256), 0, 0, a, n16);
                if (variant == 1) hipLaunchKernelGGL(fill_seg, dim3(grid), dim3(256), 0, 0, a, n16 / 1280, 1280);   // 20 KB segments
                if (variant == 2) hipLaunchKernelGGL(copy_flat, dim3(grid), dim3(256), 0, 0, a, b, n16);
                if (variant == 3) CK(hipMemsetAsync(a, 1, bytes, 0));
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 2) { sum += ms; if (ms < best) best = ms; }
            }
            const char* nm[] = {"fill_flat", "fill_seg20KB", "copy_flat(r+w)", "hipMemsetAsync"};
            const double moved = variant == 2 ? 2.0 * bytes : (double)bytes;
            printf("grid %5d %-16s avg %.3f ms  %.0f GB/s (best %.0f)\n", grid, nm[variant], sum / reps,
                   moved / (sum / reps) / 1e6, moved / best / 1e6);
            if (variant == 3 && grid != 1024) break;
        }
    }
    return 0;
}
