// Stage-2 prototype B: dense bit stream -> f32.  out[e] = bit e of the stream.  One wave instruction
// (64 lanes x 4 elements) consumes 256 bits = 4 u64 words; lane l needs nibble (l & 15) of word (l >> 4).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// persistent waves over aligned 32 KB output chunks (8192 elements = 128 words of bits per chunk)
__global__ __launch_bounds__(256) void expand_chunks(const uint64_t* __restrict__ bits, float* __restrict__ out, int64_t n_el) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t nchunks = (n_el + 8191) / 8192, stride = (int64_t)gridDim.x * 4;
    const int sh = 4 * (lane & 15);
    for (int64_t c = (int64_t)blockIdx.x * 4 + wave; c < nchunks; c += stride) {
        const uint64_t* wsrc = bits + c * 128 + (lane >> 4);          // word of this lane in iteration 0
        char* seg = (char*)(out + c * 8192) + lane * 16;
        const int iters = (int)(((n_el - c * 8192) < 8192 ? (n_el - c * 8192) : 8192) / 256);
        uint64_t w = wsrc[0];
        for (int i = 0; i < iters; ++i) {
            const uint64_t wn = i + 1 < iters ? wsrc[4 * (i + 1)] : 0;   // prefetch next
            const uint32_t nib = (uint32_t)(w >> sh) & 15u;
            const u32x4 v4 = {(nib & 1u) ? 0x3F800000u : 0u, (nib & 2u) ? 0x3F800000u : 0u,
                              (nib & 4u) ? 0x3F800000u : 0u, (nib & 8u) ? 0x3F800000u : 0u};
            *(u32x4*)(seg + i * 1024) = v4;
            w = wn;
        }
    }
}
// same, but the chunk's 128 words are fetched up-front (2 per lane, coalesced) into LDS, the next chunk's
// words are already in flight while this chunk is expanded: no global-load dependency in the store loop
__global__ __launch_bounds__(256) void expand_chunks_lds(const uint64_t* __restrict__ bits, float* __restrict__ out, int64_t n_el) {
    __shared__ uint64_t buf[4][2][128];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t nchunks = n_el / 8192, stride = (int64_t)gridDim.x * 4;
    const int sh = 4 * (lane & 15);
    int64_t c = (int64_t)blockIdx.x * 4 + wave;
    if (c >= nchunks) return;
    uint64_t a0 = bits[c * 128 + lane], a1 = bits[c * 128 + 64 + lane];
    int ping = 0;
    for (; c < nchunks; c += stride) {
        uint64_t* b = buf[wave][ping];
        b[lane] = a0; b[64 + lane] = a1;
        const int64_t cn = c + stride;
        if (cn < nchunks) { a0 = bits[cn * 128 + lane]; a1 = bits[cn * 128 + 64 + lane]; }    // next chunk in flight
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        char* seg = (char*)(out + c * 8192) + lane * 16;
        const uint64_t* wsrc = b + (lane >> 4);
#pragma unroll 4
        for (int i = 0; i < 32; ++i) {
            const uint32_t nib = (uint32_t)(wsrc[4 * i] >> sh) & 15u;
            const u32x4 v4 = {(nib & 1u) ? 0x3F800000u : 0u, (nib & 2u) ? 0x3F800000u : 0u,
                              (nib & 4u) ? 0x3F800000u : 0u, (nib & 8u) ? 0x3F800000u : 0u};
            *(u32x4*)(seg + i * 1024) = v4;
        }
        ping ^= 1;
    }
}
// as expand_chunks_lds, but B groups are computed into registers first and then stored back-to-back
template <int B>
__global__ __launch_bounds__(256) void expand_chunks_burst(const uint64_t* __restrict__ bits, float* __restrict__ out, int64_t n_el) {
    __shared__ uint64_t buf[4][2][128];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t nchunks = n_el / 8192, stride = (int64_t)gridDim.x * 4;
    const int sh = 4 * (lane & 15);
    int64_t c = (int64_t)blockIdx.x * 4 + wave;
    if (c >= nchunks) return;
    uint64_t a0 = bits[c * 128 + lane], a1 = bits[c * 128 + 64 + lane];
    int ping = 0;
    for (; c < nchunks; c += stride) {
        uint64_t* b = buf[wave][ping];
        b[lane] = a0; b[64 + lane] = a1;
        const int64_t cn = c + stride;
        if (cn < nchunks) { a0 = bits[cn * 128 + lane]; a1 = bits[cn * 128 + 64 + lane]; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        char* seg = (char*)(out + c * 8192) + lane * 16;
        const uint64_t* wsrc = b + (lane >> 4);
        for (int i0 = 0; i0 < 32; i0 += B) {
            u32x4 v[B];
#pragma unroll
            for (int j = 0; j < B; ++j) {
                const uint32_t nib = (uint32_t)(wsrc[4 * (i0 + j)] >> sh) & 15u;
                v[j] = u32x4{(nib & 1u) ? 0x3F800000u : 0u, (nib & 2u) ? 0x3F800000u : 0u,
                             (nib & 4u) ? 0x3F800000u : 0u, (nib & 8u) ? 0x3F800000u : 0u};
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < B; ++j) *(u32x4*)(seg + (i0 + j) * 1024) = v[j];
            __builtin_amdgcn_sched_barrier(0);
        }
        ping ^= 1;
    }
}
// flat single front: G workgroups, wave w of the grid writes KiB number i*nwaves + w
__global__ __launch_bounds__(256) void expand_flat(const uint64_t* __restrict__ bits, float* __restrict__ out, int64_t n_el) {
    const int lane = threadIdx.x & 63;
    const int sh = 4 * (lane & 15);
    const int64_t nwaves = (int64_t)gridDim.x * 4, w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nk = n_el / 256;
    for (int64_t k = w0; k < nk; k += nwaves) {
        const uint64_t w = bits[k * 4 + (lane >> 4)];
        const uint32_t nib = (uint32_t)(w >> sh) & 15u;
        const u32x4 v4 = {(nib & 1u) ? 0x3F800000u : 0u, (nib & 2u) ? 0x3F800000u : 0u,
                          (nib & 4u) ? 0x3F800000u : 0u, (nib & 8u) ? 0x3F800000u : 0u};
        *(u32x4*)((char*)out + k * 1024 + lane * 16) = v4;
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
hipEvent_t e0, e1; double g_bytes;
template <typename F> int timeit(const char* name, F launch) {
    float sum = 0;
    for (int r = 0; r < 10; ++r) {
        CK(hipEventRecord(e0)); launch(); CK(hipGetLastError()); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) sum += ms;
    }
    printf("%-44s %.3f ms  %5.0f GB/s\n", name, sum / 8, g_bytes / (sum / 8) / 1e6);
    return 0;
}
int main() {
    const int64_t n_el = (4844953ll * 98) & ~255ll;
    g_bytes = (double)n_el * 4;
    float* out; CK(hipMalloc(&out, n_el * 4 + (1 << 20)));
    std::vector<uint64_t> h(n_el / 64 + 1024);
    srand(3); for (auto& x : h) x = ((uint64_t)rand() << 40) ^ ((uint64_t)rand() << 20) ^ (uint64_t)rand();
    uint64_t* bits; CK(hipMalloc(&bits, 8 * h.size())); CK(hipMemcpy(bits, h.data(), 8 * h.size(), hipMemcpyHostToDevice));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        timeit("hipMemsetAsync", [&] { (void)hipMemsetAsync(out, 0, (size_t)g_bytes, 0); });
        for (int G : {256, 512, 1024, 2048}) {
            char nm[64];
            snprintf(nm, 64, "expand bitstream, 32K chunks   G=%d", G); timeit(nm, [&] { hipLaunchKernelGGL(expand_chunks, dim3(G), dim3(256), 0, 0, bits, out, n_el); });
            snprintf(nm, 64, "expand bitstream, 32K chunks+LDS G=%d", G); timeit(nm, [&] { hipLaunchKernelGGL(expand_chunks_lds, dim3(G), dim3(256), 0, 0, bits, out, n_el & ~8191ll); });
            snprintf(nm, 64, "expand bitstream, burst x8     G=%d", G); timeit(nm, [&] { hipLaunchKernelGGL(expand_chunks_burst<8>, dim3(G), dim3(256), 0, 0, bits, out, n_el & ~8191ll); });
            snprintf(nm, 64, "expand bitstream, burst x32    G=%d", G); timeit(nm, [&] { hipLaunchKernelGGL(expand_chunks_burst<32>, dim3(G), dim3(256), 0, 0, bits, out, n_el & ~8191ll); });
            snprintf(nm, 64, "expand bitstream, flat         G=%d", G); timeit(nm, [&] { hipLaunchKernelGGL(expand_flat, dim3(G), dim3(256), 0, 0, bits, out, n_el); });
            snprintf(nm, 64, "pure fill 32K chunks           G=%d", G); timeit(nm, [&] { hipLaunchKernelGGL(chunks, dim3(G), dim3(256), 0, 0, (char*)out, (int64_t)g_bytes & ~32767ll, 32768); });
        }
    }
    // correctness spot check
    hipLaunchKernelGGL(expand_chunks_lds, dim3(1024), dim3(256), 0, 0, bits, out, n_el & ~8191ll);
    std::vector<float> o(4096); CK(hipMemcpy(o.data(), out, 4096 * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int e = 0; e < 4096; ++e) bad += (o[e] != (float)((h[e >> 6] >> (e & 63)) & 1));
    printf("spot check mismatches: %d\n", bad);
    return 0;
}
