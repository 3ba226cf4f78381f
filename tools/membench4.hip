// Streaming-store design space: G workgroups x 4 waves, each wave writes contiguous chunks of C bytes
// (blocked-cyclic over the buffer), U independent 16-byte stores per lane per loop trip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void fill_chunks(char* out, int64_t bytes, int chunk) {
    const u32x4 v = {0x3F800000u, 0u, 0x3F800000u, 0u};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * 4, w = (int64_t)blockIdx.x * 4 + wave;
    const int64_t nchunks = bytes / chunk;
    for (int64_t c = w; c < nchunks; c += nwaves) {
        char* p = out + c * chunk + lane * 16;
        for (int o = 0; o < chunk; o += 1024 * U) {
#pragma unroll
            for (int u = 0; u < U; ++u) *(u32x4*)(p + o + u * 1024) = v;
        }
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int64_t bytes = 2ll << 30;
    char* a; CK(hipMalloc(&a, bytes)); CK(hipMemset(a, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-8s", "G\\C");
    const int chunks[] = {4096, 16384, 32768, 65536, 262144, 1048576};
    for (int c : chunks) printf("%9dK", c / 1024);
    printf("   (GB/s, U=1 | U=4)\n");
    for (int G : {256, 512, 1024, 2048, 4096}) {
        for (int U : {1, 4}) {
            printf("%-5d U%d", G, U);
            for (int c : chunks) {
                float sum = 0;
                for (int r = 0; r < 8; ++r) {
                    CK(hipEventRecord(e0));
                    if (U == 1) hipLaunchKernelGGL(fill_chunks<1>, dim3(G), dim3(256), 0, 0, a, bytes, c);
                    else hipLaunchKernelGGL(fill_chunks<4>, dim3(G), dim3(256), 0, 0, a, bytes, c);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r >= 2) sum += ms;
                }
                printf("%10.0f", bytes / (sum / 6) / 1e6);
            }
            printf("\n");
        }
    }
    float sum = 0;
    for (int r = 0; r < 8; ++r) {
        CK(hipEventRecord(e0)); CK(hipMemsetAsync(a, 1, bytes, 0)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) sum += ms;
    }
    printf("hipMemsetAsync %.0f GB/s\n", bytes / (sum / 6) / 1e6);
    return 0;
}
