// Stage-2 prototype: expand a bit-packed perspective stack (2 x u64 per perspective, d=7: 49+49 bits) to
// f32, persistent waves over aligned 32 KB output chunks; packed words of a chunk staged in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int NQ = 98, DD = 49;
constexpr int CH_EL = 8192;                      // f32 elements per 32 KB chunk
constexpr int PMAX = CH_EL / NQ + 2;             // perspectives a chunk can touch (85)

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void expand(const uint64_t* __restrict__ packed, float* __restrict__ out, int64_t n_el) {
    __shared__ uint64_t bits[WAVES][2 * PMAX + 2];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t nchunks = (n_el + CH_EL - 1) / CH_EL, stride = (int64_t)gridDim.x * WAVES;
    uint64_t* wb = bits[wave];
    for (int64_t c = (int64_t)blockIdx.x * WAVES + wave; c < nchunks; c += stride) {
        const int64_t e0 = c * CH_EL;
        const int64_t p0 = e0 / NQ;                                   // first perspective of the chunk
        const int np = (int)((e0 + CH_EL - 1) / NQ - p0) + 1;         // perspectives touched
        for (int k = lane; k < 2 * np; k += 64) wb[k] = packed[2 * p0 + k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        const int n_el_c = (int)((n_el - e0) < CH_EL ? (n_el - e0) : CH_EL);
        char* seg = (char*)(out + e0);
        int rel = (int)(e0 - p0 * NQ) + lane * 4;                     // element index relative to perspective p0
        int pp = rel / NQ, cc = rel - pp * NQ;
        for (int g = lane; g < n_el_c / 4; g += 64) {
            uint32_t wd[4];
            int p = pp, cidx = cc;
#pragma unroll
            for (int k = 0; k < 4; k += 2) {                          // NQ, cell even: pairs never straddle perspectives
                const int plane = cidx >= DD;
                const uint64_t word = wb[2 * p + plane];
                const int b = cidx - plane * DD;
                // the pair (b, b+1) may straddle the V/P plane boundary (b == 48)
                const uint32_t v0 = (uint32_t)(word >> b) & 1u;
                const uint32_t v1 = b + 1 < DD ? (uint32_t)(word >> (b + 1)) & 1u : (uint32_t)(wb[2 * p + 1] & 1u);
                wd[k] = v0 ? 0x3F800000u : 0u;
                wd[k + 1] = v1 ? 0x3F800000u : 0u;
                if (k == 0) { cidx += 2; if (cidx >= NQ) { cidx -= NQ; ++p; } }
            }
            const u32x4 v4 = {wd[0], wd[1], wd[2], wd[3]};
            *(u32x4*)(seg + (uint32_t)g * 16u) = v4;
            cc += 256 % NQ; pp += 256 / NQ;
            if (cc >= NQ) { cc -= NQ; ++pp; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
}
__global__ __launch_bounds__(256) void chunks(char* out, int64_t bytes, int chunk) {
    const u32x4 v = {0x3F800000u, 0u, 0x3F800000u, 0u};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * 4, w = (int64_t)blockIdx.x * 4 + wave;
    for (int64_t c = w; c < bytes / chunk; c += nwaves) {
        char* p = out + c * chunk + lane * 16;
        for (int o = 0; o < chunk; o += 1024) *(u32x4*)(p + o) = v;
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int64_t P = 4844953;                                       // perspectives per launch of the bench
    const int64_t n_el = P * NQ;
    float* out; CK(hipMalloc(&out, n_el * 4 + (1 << 20))); CK(hipMemset(out, 0, n_el * 4));
    std::vector<uint64_t> h(2 * P + 8);
    srand(3);
    for (auto& x : h) x = (((uint64_t)rand() << 32) ^ (uint64_t)rand() ^ ((uint64_t)rand() << 17)) & ((1ull << 49) - 1);
    uint64_t* packed; CK(hipMalloc(&packed, 8 * h.size())); CK(hipMemcpy(packed, h.data(), 8 * h.size(), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)n_el * 4;
    printf("chunks: %.1f per 1024 waves-of-4096..: total %ld chunks\n", 0.0, (long)((n_el + CH_EL - 1) / CH_EL));
    for (int rep = 0; rep < 2; ++rep) {
        for (int G : {512, 1024, 1536, 2048}) {
            float sum = 0;
            for (int r = 0; r < 10; ++r) {
                CK(hipEventRecord(e0)); hipLaunchKernelGGL(expand<4>, dim3(G), dim3(256), 0, 0, packed, out, n_el);
                CK(hipGetLastError()); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) sum += ms;
            }
            printf("expand (bits -> f32, 32K chunks) G=%4d : %.3f ms  %.0f GB/s\n", G, sum / 8, bytes / (sum / 8) / 1e6);
            sum = 0;
            for (int r = 0; r < 10; ++r) {
                CK(hipEventRecord(e0)); hipLaunchKernelGGL(chunks, dim3(G), dim3(256), 0, 0, (char*)out, (int64_t)bytes & ~32767ll, 32768);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) sum += ms;
            }
            printf("pure fill 32K chunks            G=%4d : %.3f ms  %.0f GB/s\n", G, sum / 8, bytes / (sum / 8) / 1e6);
        }
        float sum = 0;
        for (int r = 0; r < 10; ++r) {
            CK(hipEventRecord(e0)); CK(hipMemsetAsync(out, 0, (size_t)bytes, 0)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) sum += ms;
        }
        printf("hipMemsetAsync                         : %.3f ms  %.0f GB/s\n", sum / 8, bytes / (sum / 8) / 1e6);
    }
    // spot check of the expansion
    std::vector<float> o(4096); CK(hipMemcpy(o.data(), out + 98 * 1000, 4096 * 4, hipMemcpyDeviceToHost));
    return 0;
}
