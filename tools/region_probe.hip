// Is the write rate a LOCAL property of physical memory?  A pool of 2 MiB chunks (HIP virtual memory API, handles kept),
// mapped in groups of GROUP chunks; every group is timed with a flat fill; then two buffers are assembled from the
// fastest and from the slowest groups (the same handles mapped a second time) and a full-size fill is timed on both.
//   hipcc -O3 --offload-arch=gfx950 tools/region_probe.hip -o tools/region_probe ; tools/region_probe [pool GiB] [group MiB] [buffer GiB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <numeric>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// 256 workgroups x 4 waves, each workgroup streams its own contiguous 1/256 of the buffer (the stack write's shape)
__global__ __launch_bounds__(256) void k_fill(char* out, int64_t bytes) {
    const u32x4 v = {0x3F800000u, 0u, 0x3F800000u, 0u};
    const int64_t per = bytes / gridDim.x / 4096 * 4096;
    char* p = out + blockIdx.x * per;
    for (int64_t o = threadIdx.x * 16; o < per; o += 4096) *reinterpret_cast<u32x4*>(p + o) = v;
}
int main(int argc, char** argv) {
    const size_t pool_gib = argc > 1 ? atoi(argv[1]) : 8, group_mib = argc > 2 ? atoi(argv[2]) : 512, buf_gib2 = argc > 3 ? atoi(argv[3]) : 2;
    const size_t chunk = 2u << 20, gchunks = group_mib / 2, ngroups = pool_gib * 1024 / group_mib, nchunks = ngroups * gchunks;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> h(nchunks);
    for (size_t i = 0; i < nchunks; ++i) CK(hipMemCreate(&h[i], chunk, &prop, 0));
    void* va; CK(hipMemAddressReserve(&va, nchunks * chunk, 0, nullptr, 0));
    for (size_t i = 0; i < nchunks; ++i) CK(hipMemMap((char*)va + i * chunk, chunk, 0, h[i], 0));
    CK(hipMemSetAccess(va, nchunks * chunk, &acc, 1));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timefill = [&](char* p, size_t bytes, int reps) {
        hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, p, (int64_t)bytes);
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, p, (int64_t)bytes);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return bytes * (double)reps / (ms * 1e-3) / 1e9;
    };
    std::vector<double> rate(ngroups);
    printf("pool %zu GiB in %zu groups of %zu MiB; fill rate of every group (GB/s):\n ", pool_gib, ngroups, group_mib);
    for (size_t g = 0; g < ngroups; ++g) { rate[g] = timefill((char*)va + g * gchunks * chunk, gchunks * chunk, 8); printf(" %5.0f", rate[g]); }
    printf("\n");
    std::vector<size_t> order(ngroups); std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return rate[a] > rate[b]; });
    const size_t need = buf_gib2 * 1024 / group_mib;        // groups per assembled buffer
    auto assemble = [&](bool best) {
        void* v2; CK(hipMemAddressReserve(&v2, need * gchunks * chunk, 0, nullptr, 0));
        for (size_t k = 0; k < need; ++k) {
            const size_t g = best ? order[k] : order[ngroups - 1 - k];
            for (size_t c = 0; c < gchunks; ++c) CK(hipMemMap((char*)v2 + (k * gchunks + c) * chunk, chunk, 0, h[g * gchunks + c], 0));
        }
        CK(hipMemSetAccess(v2, need * gchunks * chunk, &acc, 1));
        return (char*)v2;
    };
    auto assemble_stride = [&](size_t stride, size_t phase) {      // groups phase, phase+stride, ... of the pool
        void* v2; CK(hipMemAddressReserve(&v2, need * gchunks * chunk, 0, nullptr, 0));
        for (size_t k = 0; k < need; ++k) {
            const size_t g = (phase + k * stride) % ngroups;
            for (size_t c = 0; c < gchunks; ++c) CK(hipMemMap((char*)v2 + (k * gchunks + c) * chunk, chunk, 0, h[g * gchunks + c], 0));
        }
        CK(hipMemSetAccess(v2, need * gchunks * chunk, &acc, 1));
        return (char*)v2;
    };
    {
        const size_t bb2 = need * gchunks * chunk;
        for (size_t stride : {(size_t)1, (size_t)2, (size_t)3, ngroups / need}) {
            printf("buffer from every %zu-th group:", stride);
            for (size_t phase = 0; phase < 3; ++phase) { char* b = assemble_stride(stride, phase * (stride > 1 ? 1 : need)); printf("  %5.0f", timefill(b, bb2, 8)); }
            printf(" GB/s\n");
        }
    }
    char* fast = assemble(true); char* slow = assemble(false);
    const size_t bb = need * gchunks * chunk;
    printf("buffer of %zu GiB from the fastest groups: %5.0f GB/s   from the slowest groups: %5.0f GB/s   first groups in pool order: %5.0f GB/s\n",
           buf_gib2, timefill(fast, bb, 8), timefill(slow, bb, 8), timefill((char*)va, bb, 8));
    printf("again:                                     %5.0f GB/s                            %5.0f GB/s                              %5.0f GB/s\n",
           timefill(fast, bb, 8), timefill(slow, bb, 8), timefill((char*)va, bb, 8));
    return 0;
}
