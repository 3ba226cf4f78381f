// Which STORE ORDER reaches the fill rate of hipMemset / torch's fill kernel (6.9 TB/s on any buffer), and which falls to
// 5.2-5.6 on the buffers the stack write is slow on?  Pure stores of constant data, one process, several buffers.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/fill_shapes.hip -o tools/fill_shapes && tools/fill_shapes
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// A: one block per BLOCK_BYTES, blocks in address order, never re-used (the shape of an elementwise fill)
template <int PER_THREAD>
__global__ __launch_bounds__(256) void k_blocks(char* out, int64_t bytes, uint32_t word) {
    const u32x4 v = {word, 0u, word, 0u};
    const int64_t base = (int64_t)blockIdx.x * 256 * 16 * PER_THREAD + threadIdx.x * 16;
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        const int64_t o = base + (int64_t)k * 256 * 16;
        if (o + 16 <= bytes) *reinterpret_cast<u32x4*>(out + o) = v;
    }
}
// B: persistent grid-stride version of A: G workgroups, workgroup g takes blocks g, g + G, ...
template <int PER_THREAD>
__global__ __launch_bounds__(256) void k_stride(char* out, int64_t bytes, uint32_t word) {
    const u32x4 v = {word, 0u, word, 0u};
    const int64_t blk = 256 * 16 * PER_THREAD;
    for (int64_t b = blockIdx.x; b * blk < bytes; b += gridDim.x) {
        const int64_t base = b * blk + threadIdx.x * 16;
#pragma unroll
        for (int k = 0; k < PER_THREAD; ++k) {
            const int64_t o = base + (int64_t)k * 256 * 16;
            if (o + 16 <= bytes) *reinterpret_cast<u32x4*>(out + o) = v;
        }
    }
}
// C: G workgroups, each streams ITS OWN contiguous 1/G of the buffer (the shape of the stack write)
template <int PER_THREAD>
__global__ __launch_bounds__(256) void k_slices(char* out, int64_t bytes, uint32_t word) {
    const u32x4 v = {word, 0u, word, 0u};
    const int64_t slice = (bytes / gridDim.x) & ~(int64_t)4095;
    char* p = out + blockIdx.x * slice;
    const int64_t blk = 256 * 16 * PER_THREAD;
    for (int64_t b = 0; b + blk <= slice; b += blk) {
#pragma unroll
        for (int k = 0; k < PER_THREAD; ++k) *reinterpret_cast<u32x4*>(p + b + (int64_t)k * 256 * 16 + threadIdx.x * 16) = v;
    }
}

// D: expand a bit string (1 bit per f32 element, 1/32 of the output's size: it stays in the last-level cache) into the
// output, ONE 16-byte store per lane, one wave = 1 KiB, WAVES waves per block, blocks in address order
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_expand(const uint32_t* __restrict__ bits, char* out, int64_t bytes) {
    const int64_t chunk = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);       // 1 KiB = 256 elements = 8 dwords of bits
    const int lane = threadIdx.x & 63;
    const int64_t o = chunk * 1024 + lane * 16;
    if (o + 16 > bytes) return;
    const uint32_t w = bits[chunk * 8 + (lane >> 3)] >> ((lane & 7) * 4);
    u32x4 v;
    v.x = (w & 1u) ? 0x3F800000u : 0u; v.y = (w & 2u) ? 0x3F800000u : 0u; v.z = (w & 4u) ? 0x3F800000u : 0u; v.w = (w & 8u) ? 0x3F800000u : 0u;
    *reinterpret_cast<u32x4*>(out + o) = v;
}

int main() {
    const int64_t bytes = (int64_t)1878 << 20;
    std::vector<char*> bufs;
    for (int i = 0; i < 6; ++i) { char* p; CK(hipMalloc(&p, bytes + (i << 22))); bufs.push_back(p); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto rate = [&](auto f) {
        float a = 0;
        for (int r = 0; r < 6; ++r) { CK(hipEventRecord(e0)); f(); CK(hipGetLastError()); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) a += ms; }
        return bytes / (a / 5) / 1e6;
    };
    {
        uint32_t* bits; CK(hipMalloc(&bits, bytes / 32 + 4096));
        std::vector<uint32_t> hb(bytes / 32 / 4 + 1024);
        uint64_t x = 88172645463325252ull;
        for (auto& w : hb) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = (uint32_t)x & (uint32_t)(x >> 32) & (uint32_t)(x >> 16); }   // ~1/8 ones
        CK(hipMemcpy(bits, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        printf("expand of a bit string (%.0f MB -> %.2f GB), one 16-byte store per lane:\n", bytes / 32 / 1e6, bytes / 1e9);
        for (size_t i = 0; i < bufs.size(); ++i) {
            char* p = bufs[i];
            printf("  buffer %zu: 1 wave per block %5.0f   4 waves %5.0f   8 waves %5.0f   16 waves %5.0f GB/s\n", i,
                   rate([&] { hipLaunchKernelGGL(k_expand<1>, dim3((unsigned)(bytes / 1024)), dim3(64), 0, 0, (const uint32_t*)bits, p, bytes); }),
                   rate([&] { hipLaunchKernelGGL(k_expand<4>, dim3((unsigned)(bytes / 4096)), dim3(256), 0, 0, (const uint32_t*)bits, p, bytes); }),
                   rate([&] { hipLaunchKernelGGL(k_expand<8>, dim3((unsigned)(bytes / 8192)), dim3(512), 0, 0, (const uint32_t*)bits, p, bytes); }),
                   rate([&] { hipLaunchKernelGGL(k_expand<16>, dim3((unsigned)(bytes / 16384)), dim3(1024), 0, 0, (const uint32_t*)bits, p, bytes); }));
            fflush(stdout);
        }
    }
    for (uint32_t word : {0x3F800000u}) {
        printf("data word %08x\n", word);
        for (size_t i = 0; i < bufs.size(); ++i) {
            char* p = bufs[i];
            printf("  buffer %zu: hipMemset %5.0f |", i, rate([&] { (void)hipMemsetAsync(p, (int)(word & 0xff), bytes, 0); }));
            printf(" blocks x1 %5.0f x4 %5.0f x16 %5.0f |", rate([&] { hipLaunchKernelGGL(k_blocks<1>, dim3((unsigned)(bytes / 4096)), dim3(256), 0, 0, p, bytes, word); }),
                   rate([&] { hipLaunchKernelGGL(k_blocks<4>, dim3((unsigned)(bytes / 16384)), dim3(256), 0, 0, p, bytes, word); }),
                   rate([&] { hipLaunchKernelGGL(k_blocks<16>, dim3((unsigned)(bytes / 65536)), dim3(256), 0, 0, p, bytes, word); }));
            printf(" grid-stride 256 wg %5.0f 2048 wg %5.0f 8192 wg %5.0f |", rate([&] { hipLaunchKernelGGL(k_stride<4>, dim3(256), dim3(256), 0, 0, p, bytes, word); }),
                   rate([&] { hipLaunchKernelGGL(k_stride<4>, dim3(2048), dim3(256), 0, 0, p, bytes, word); }),
                   rate([&] { hipLaunchKernelGGL(k_stride<4>, dim3(8192), dim3(256), 0, 0, p, bytes, word); }));
            printf(" own slices 256 wg %5.0f 1024 wg %5.0f 2048 wg %5.0f 8192 wg %5.0f\n", rate([&] { hipLaunchKernelGGL(k_slices<4>, dim3(256), dim3(256), 0, 0, p, bytes, word); }),
                   rate([&] { hipLaunchKernelGGL(k_slices<4>, dim3(1024), dim3(256), 0, 0, p, bytes, word); }),
                   rate([&] { hipLaunchKernelGGL(k_slices<4>, dim3(2048), dim3(256), 0, 0, p, bytes, word); }),
                   rate([&] { hipLaunchKernelGGL(k_slices<4>, dim3(8192), dim3(256), 0, 0, p, bytes, word); }));
            fflush(stdout);
        }
    }
    return 0;
}
