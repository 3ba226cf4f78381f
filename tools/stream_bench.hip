// Stand-alone harness for the stack-write kernel (producer / storer stream, stream_write.hpp) against the round-2
// kernel (one wave per lattice, tools/lattice_write_r02.hpp) on synthetic syndromes: same inputs, outputs compared
// dword for dword, launches timed with HIP events on SIX output buffers (the rate depends on where a buffer lies in
// HBM), a configuration sweep on the fastest and the slowest of them, and the per-role wait statistics of the
// stream kernel (STATS instantiation).  Build here, run on the GPU box:
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Iinclude -Itoric-rl-decoder_amd/csrc -Itools tools/stream_bench.hip -o tools/stream_bench
//   tools/stream_bench [d=3..11] [lattices=65536] [q = probability of a defect per check]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <vector>

#include "stream_write.hpp"   // (the kernel is called with its 257-entry table: equal shares (lg = 8, bias 0, no slot counters), as when this harness was written)
#include "lattice_write_r02.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int D>
__global__ __launch_bounds__(256) void k_counts(const uint64_t* __restrict__ vp, int32_t* __restrict__ counts, int64_t N,
                                                int64_t* __restrict__ part256) {
    using L = tq::Lat<D>;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int cnt = 0;
    if (e < N) {
        typename L::B v, p;
        for (int k = 0; k < L::W; ++k) { v.w[k] = vp[(int64_t)k * N + e]; p.w[k] = vp[((int64_t)L::W + k) * N + e]; }
        cnt = L::persp_count(v, p);
        counts[e] = cnt;
    }
    tq::block_count_partial(cnt, part256);
}

__global__ void k_diff(const uint32_t* a, const uint32_t* b, int64_t n, unsigned long long* bad) {
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(bad, c);
}

// One virtual range backed by physical chunks of `chunk` bytes mapped in a SHUFFLED order (HIP virtual memory API):
// consecutive chunks of the buffer lie in unrelated places of the HBM.
static float* alloc_shuffled(size_t bytes, size_t chunk, unsigned seed) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t n = (bytes + chunk - 1) / chunk;
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, n * chunk, 0, nullptr, 0));
    std::vector<size_t> perm(n);
    for (size_t i = 0; i < n; ++i) perm[i] = i;
    std::mt19937 rng(seed);
    if (seed) std::shuffle(perm.begin(), perm.end(), rng);
    for (size_t i = 0; i < n; ++i) {
        hipMemGenericAllocationHandle_t hnd;
        CK(hipMemCreate(&hnd, chunk, &prop, 0));
        CK(hipMemMap((char*)va + perm[i] * chunk, chunk, 0, hnd, 0));
        CK(hipMemRelease(hnd));
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, n * chunk, &acc, 1));
    return (float*)va;
}

// pure store pattern: G workgroups x NS waves; windows of WIN bytes dealt round-robin over the workgroups, inside a
// workgroup over its waves; a wave streams its window with 16 B per lane stores
template <int NS>
__global__ __launch_bounds__(64 * NS) void k_front(char* __restrict__ out, int64_t bytes, int win) {
    const tq::u32x4 v = {0x3F800000u, 0u, 0x3F800000u, 0u};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t nwin = bytes / win;
    for (int64_t i = wave;; i += NS) {
        const int64_t w = blockIdx.x + i * (int64_t)gridDim.x;
        if (w >= nwin) break;
        char* p = out + w * win + lane * 16;
        for (int o = 0; o < win; o += 1024) *reinterpret_cast<tq::u32x4*>(p + o) = v;
    }
}

struct Timer {
    hipEvent_t e0, e1;
    Timer() { CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); }
    template <class F> float run(F f) {
        CK(hipEventRecord(e0)); f(); CK(hipGetLastError()); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
    }
};

template <int D, int NS, int NP, int CPW>
void stats_run(const uint64_t* vp, int64_t N, const int64_t* off, float* out, int32_t* pos, int64_t cap, int* err, const int32_t* split) {
    constexpr int WV = NS + 1 + NP, G = 256;
    unsigned long long* st;
    CK(hipMalloc(&st, sizeof(unsigned long long) * G * WV * 4));
    CK(hipMemset(st, 0, sizeof(unsigned long long) * G * WV * 4));
    hipLaunchKernelGGL((tq::k_persp_stream<D, float, NS, NP, CPW, 14, 12, true>), dim3(G), dim3(64 * WV), 0, 0, vp, N, off, out, pos, cap,
                       err, (int64_t)0, N, split, 8, 0, (unsigned int*)nullptr, st);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)G * WV * 4);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    printf("per-role statistics, NS=%d NP=%d CPW=%d (cycles = s_memtime ticks; a trip = 4 KiB stored, an item of a producer = one lattice):\n", NS, NP, CPW);
    auto agg = [&](int w0, int w1, const char* role, const char* a_name, const char* b_name) {
        double tot = 0, a = 0, b = 0, items = 0, mx = 0; int n = 0;
        for (int g = 0; g < G; ++g) for (int w = w0; w < w1; ++w) {
            const unsigned long long* o = &h[((size_t)g * WV + w) * 4];
            if (!o[0]) continue;
            tot += o[0]; a += o[1]; b += o[2]; items += o[3]; mx = std::max(mx, (double)o[0]); ++n;
        }
        if (n) printf("  %-10s waves %5d  alive %9.0f cyc (max %9.0f)  waiting for %s %5.1f %%  %s %5.1f %%  items/wave %7.1f  busy cyc/item %7.0f\n",
                      role, n, tot / n, mx, a_name, 100 * a / tot, b_name, 100 * b / tot, items / n, (tot - a - b) / std::max(1.0, items));
    };
    agg(0, NS, "storer", "production", "-");
    agg(NS, NS + 1, "positions", "production", "-");
    agg(NS + 1, WV, "producer", "ring room", "commit turn");
    CK(hipFree(st));
}

template <int D>
int run(int64_t N, double q) {
    using L = tq::Lat<D>;
    constexpr int W = L::W, NQ = L::NQ;
    std::mt19937_64 rng(7);
    std::vector<uint64_t> h((size_t)2 * W * N, 0);
    std::bernoulli_distribution bit(q);
    for (int64_t e = 0; e < N; ++e)
        for (int pl = 0; pl < 2; ++pl)
            for (int b = 0; b < L::DD; ++b)
                if (bit(rng)) h[((size_t)pl * W + b / 64) * N + e] |= 1ull << (b & 63);
    uint64_t* vp; CK(hipMalloc(&vp, h.size() * 8)); CK(hipMemcpy(vp, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    int32_t* counts; CK(hipMalloc(&counts, 4 * N + 64));
    int64_t* part; CK(hipMalloc(&part, 8 * ((N + 255) / 256)));
    int64_t* off; CK(hipMalloc(&off, 8 * (N + 2)));
    int32_t *split, *split2; CK(hipMalloc(&split, 4 * 258)); CK(hipMalloc(&split2, 4 * 258));
    int* err; CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    hipLaunchKernelGGL(k_counts<D>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, vp, counts, N, part);
    hipLaunchKernelGGL(tq::k_scan_final, dim3((unsigned)((N + tq::SCAN_CHUNK - 1) / tq::SCAN_CHUNK)), dim3(256), 0, 0, counts,
                       (const int64_t*)part, off, (int32_t*)nullptr, N, split, 8);
    hipLaunchKernelGGL(tq::k_split, dim3((256 + 1 + 3) / 4), dim3(256), 0, 0, (const int64_t*)off, (int64_t)0, N, split2, 8);
    CK(hipDeviceSynchronize());
    int32_t hs[257], hs2[257];
    CK(hipMemcpy(hs, split, 4 * 257, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs2, split2, 4 * 257, hipMemcpyDeviceToHost));
    printf("cut points: scan by-product %s k_split\n", memcmp(hs, hs2, sizeof(hs)) ? "DIFFERS FROM" : "==");
    {   // k_split on lattice sub-ranges against the host's own search
        std::vector<int64_t> ho((size_t)N + 1);
        CK(hipMemcpy(ho.data(), off, 8 * (N + 1), hipMemcpyDeviceToHost));
        const int64_t ranges[][2] = {{0, 64}, {64, 192}, {64, 191}, {1, 129}, {128, 256}, {0, 63}, {0, 65}, {4096, 8192}, {16384, 32768}, {5, N}};
        for (auto& r : ranges) {
            if (r[1] > N) continue;
            CK(hipMemset(split2, 0xff, 4 * 258));
            hipLaunchKernelGGL(tq::k_split, dim3((256 + 1 + 3) / 4), dim3(256), 0, 0, (const int64_t*)off, r[0], r[1], split2, 8);
            CK(hipMemcpy(hs2, split2, 4 * 257, hipMemcpyDeviceToHost));
            int nbad = 0, first_bad = -1;
            for (int k = 0; k <= 256; ++k) {
                const int64_t target = ho[r[0]] + (((ho[r[1]] - ho[r[0]]) * k) >> 8);
                int64_t e = r[0];
                while (ho[e] < target) ++e;
                if (hs2[k] != e) { if (!nbad) first_bad = k; ++nbad; }
            }
            printf("k_split [%lld, %lld): %d of 257 cut points wrong%s\n", (long long)r[0], (long long)r[1], nbad, nbad ? "" : " (ok)");
            if (nbad) printf("    first wrong k=%d: got %d\n", first_bad, hs2[first_bad]);
        }
    }
    {   // the stack of lattice sub-ranges: stream kernel (cut points from k_split) against the one-wave-per-lattice kernel
        std::vector<int64_t> ho((size_t)N + 1);
        CK(hipMemcpy(ho.data(), off, 8 * (N + 1), hipMemcpyDeviceToHost));
        const int64_t ranges[][2] = {{0, 64}, {64, 192}, {64, 193}, {1, 129}, {4096, 8192}};
        unsigned long long* badr; CK(hipMalloc(&badr, 8));
        for (auto& r : ranges) {
            if (r[1] > N) continue;
            const int64_t Pr = ho[r[1]] - ho[r[0]];
            float *a, *b; int32_t *pa, *pb;
            CK(hipMalloc(&a, (size_t)Pr * NQ * 4 + 4096)); CK(hipMalloc(&b, (size_t)Pr * NQ * 4 + 4096));
            CK(hipMalloc(&pa, (size_t)Pr * 12 + 4096)); CK(hipMalloc(&pb, (size_t)Pr * 12 + 4096));
            CK(hipMemset(a, 0x11, (size_t)Pr * NQ * 4)); CK(hipMemset(b, 0x77, (size_t)Pr * NQ * 4));
            CK(hipMemset(pa, 0x11, (size_t)Pr * 12)); CK(hipMemset(pb, 0x77, (size_t)Pr * 12)); CK(hipMemset(badr, 0, 8));
            constexpr int WV = 16;
            unsigned long long* st; CK(hipMalloc(&st, 8 * 256 * WV * 4)); CK(hipMemset(st, 0, 8 * 256 * WV * 4));
            hipLaunchKernelGGL((tq::k_persp_write<D, float, 64>), dim3((unsigned)(r[1] - r[0])), dim3(64), 0, 0, vp, N, off, a, pa, Pr, err, r[0], r[1]);
            hipLaunchKernelGGL(tq::k_split, dim3((256 + 1 + 3) / 4), dim3(256), 0, 0, (const int64_t*)off, r[0], r[1], split2, 8);
            hipLaunchKernelGGL((tq::k_persp_stream<D, float, 4, 11, 8, 14, 12, true>), dim3(256), dim3(1024), 0, 0, vp, N, off, b, pb, Pr, err, r[0], r[1], (const int32_t*)split2, 8, 0, (unsigned int*)nullptr, st);
            hipLaunchKernelGGL(k_diff, dim3(256), dim3(256), 0, 0, (const uint32_t*)a, (const uint32_t*)b, Pr * NQ, badr);
            hipLaunchKernelGGL(k_diff, dim3(256), dim3(256), 0, 0, (const uint32_t*)pa, (const uint32_t*)pb, Pr * 3, badr);
            unsigned long long nb; CK(hipMemcpy(&nb, badr, 8, hipMemcpyDeviceToHost));
            std::vector<unsigned long long> hst((size_t)256 * WV * 4);
            CK(hipMemcpy(hst.data(), st, hst.size() * 8, hipMemcpyDeviceToHost));
            int alive = 0, lastwg = -1;
            for (int g = 0; g < 256; ++g) { bool any = false; for (int w = 0; w < WV; ++w) any |= hst[((size_t)g * WV + w) * 4] != 0; if (any) { ++alive; lastwg = g; } }
            printf("range [%lld, %lld): %lld perspectives, stream vs lattice kernel: %llu differing dwords; workgroups that reported: %d (last %d)\n",
                   (long long)r[0], (long long)r[1], (long long)Pr, nb, alive, lastwg);
            // the same with the product instantiation (no statistics)
            CK(hipMemset(b, 0x77, (size_t)Pr * NQ * 4)); CK(hipMemset(pb, 0x77, (size_t)Pr * 12)); CK(hipMemset(badr, 0, 8));
            hipLaunchKernelGGL((tq::k_persp_stream<D, float, 4, 11, 8, 14, 12>), dim3(256), dim3(1024), 0, 0, vp, N, off, b, pb, Pr, err, r[0], r[1], (const int32_t*)nullptr, 8, 0, (unsigned int*)nullptr, (unsigned long long*)nullptr);
            hipLaunchKernelGGL(k_diff, dim3(256), dim3(256), 0, 0, (const uint32_t*)a, (const uint32_t*)b, Pr * NQ, badr);
            hipLaunchKernelGGL(k_diff, dim3(256), dim3(256), 0, 0, (const uint32_t*)pa, (const uint32_t*)pb, Pr * 3, badr);
            CK(hipMemcpy(&nb, badr, 8, hipMemcpyDeviceToHost));
            printf("      product instantiation, cut points found in the kernel: %llu differing dwords\n", nb);
            CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(pa)); CK(hipFree(pb)); CK(hipFree(st));
        }
    }
    int64_t P; CK(hipMemcpy(&P, off + N, 8, hipMemcpyDeviceToHost));
    const double bytes = (double)P * (NQ * 4 + 12) + (double)N * NQ;
    printf("d=%d  lattices %lld  perspectives %lld (%.1f per lattice)  algorithmic bytes %.3f GB\n", D, (long long)N, (long long)P, (double)P / N, bytes / 1e9);
    float* ref; int32_t *pref, *p2;
    CK(hipMalloc(&ref, (size_t)P * NQ * 4 + 4096)); CK(hipMalloc(&pref, (size_t)P * 12 + 4096)); CK(hipMalloc(&p2, (size_t)P * 12 + 4096));
    hipLaunchKernelGGL((tq::k_persp_write<D, float, 64>), dim3((unsigned)N), dim3(64), 0, 0, vp, N, off, ref, pref, P, err, (int64_t)0, N);
    unsigned long long* bad; CK(hipMalloc(&bad, 8));
    Timer t;
    auto timeit = [&](auto k) { float a = 0; for (int r = 0; r < 6; ++r) { float x = t.run(k); if (r) a += x; } return bytes / (a / 5) / 1e6; };
    constexpr int NSP = D <= 5 ? 2 : 4, NPP = D <= 5 ? 13 : (D >= 13 ? 7 : 11);       // the library's configuration (toricenv.hip)
    std::vector<float*> bufs; std::vector<double> rate;
    const char* kind[8] = {"hipMalloc", "hipMalloc", "contiguous flag", "VMM 2 MiB in order", "VMM 2 MiB shuffled", "VMM 2 MiB shuffled", "VMM 32 MiB shuffled", "VMM 256 KiB? shuffled"};
    for (int b = 0; b < 8; ++b) {
        float* x;
        const size_t sz = (size_t)P * NQ * 4 + 4096 + (size_t)b * (3u << 20);
        if (b < 2) CK(hipMalloc(&x, sz));
        else if (b == 2) CK(hipExtMallocWithFlags((void**)&x, sz, hipDeviceMallocContiguous));
        else if (b == 3) x = alloc_shuffled(sz, 2u << 20, 0);
        else if (b == 4) x = alloc_shuffled(sz, 2u << 20, 11);
        else if (b == 5) x = alloc_shuffled(sz, 2u << 20, 12);
        else if (b == 6) x = alloc_shuffled(sz, 32u << 20, 13);
        else x = alloc_shuffled(sz, 256u << 10, 14);
        bufs.push_back(x);
    }
    printf("output buffers of different physical make-up, the same launches on each (GB/s of algorithmic bytes; memset: stack bytes only):\n");
    for (int b = 0; b < 8; ++b) {
        float* ob = bufs[b];
        auto ks = [&] { hipLaunchKernelGGL((tq::k_persp_stream<D, float, NSP, NPP, 8, 14, 12>), dim3(256), dim3(64 * (NSP + 1 + NPP)), 0, 0, vp, N, off, ob, p2, P, err, (int64_t)0, N, split, 8, 0, (unsigned int*)nullptr, (unsigned long long*)nullptr); };
        auto kl = [&] { hipLaunchKernelGGL((tq::k_persp_write<D, float, 64>), dim3((unsigned)N), dim3(64), 0, 0, vp, N, off, ob, p2, P, err, (int64_t)0, N); };
        auto km = [&] { (void)hipMemsetAsync(ob, 1, (size_t)P * NQ * 4, 0); };
        CK(hipMemset(ob, 0x77, (size_t)P * NQ * 4)); CK(hipMemset(p2, 0x77, (size_t)P * 12)); CK(hipMemset(bad, 0, 8));
        ks(); CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_diff, dim3(2048), dim3(256), 0, 0, (const uint32_t*)ref, (const uint32_t*)ob, (int64_t)P * NQ, bad);
        hipLaunchKernelGGL(k_diff, dim3(256), dim3(256), 0, 0, (const uint32_t*)pref, (const uint32_t*)p2, (int64_t)P * 3, bad);
        unsigned long long nb; CK(hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost));
        int e; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        const double rs = timeit(ks), rl = timeit(kl), rm = timeit(km) * ((double)P * NQ * 4) / bytes;
        rate.push_back(rs);
        printf("  buffer %d %-22s: stream %6.0f   one wave per lattice %6.0f   hipMemsetAsync %6.0f   (stream vs lattice: %llu differing dwords, latch %d)\n",
               b, kind[b], rs, rl, rm, nb, e);
    }
    {   // pure store patterns on every buffer: windows of WIN bytes round-robin over 256 workgroups x NS waves
        printf("pure store pattern (constant data), GB/s of stack bytes: windows round-robin over 256 workgroups x NS waves\n");
        const int64_t sb = (int64_t)P * NQ * 4 / (1 << 20) * (1 << 20);
        for (int b = 0; b < 7; ++b) {
            char* ob = (char*)bufs[b];
            printf("  buffer %d %-20s:", b, kind[b]);
            for (int win : {1024, 4096, 8192, 32768}) {
                auto k4 = [&] { hipLaunchKernelGGL(k_front<4>, dim3(256), dim3(256), 0, 0, ob, sb, win); };
                auto k16 = [&] { hipLaunchKernelGGL(k_front<16>, dim3(256), dim3(1024), 0, 0, ob, sb, win); };
                float a = 0, c = 0;
                for (int r = 0; r < 6; ++r) { float x = t.run(k4), y = t.run(k16); if (r) { a += x; c += y; } }
                printf("  %2dK: NS=4 %5.0f NS=16 %5.0f", win / 1024, sb / (a / 5) / 1e6, sb / (c / 5) / 1e6);
            }
            printf("\n");
        }
    }
    {   // does the rate on a physically contiguous buffer depend on the spacing of the 256 streams (= range size / 256)?
        float* ob = bufs[2];
        std::vector<int64_t> ho((size_t)N + 1);
        CK(hipMemcpy(ho.data(), off, 8 * (N + 1), hipMemcpyDeviceToHost));
        printf("contiguous buffer, lattice sub-ranges [0, f*N): stream spacing vs rate\n");
        for (double f : {1.0, 0.97, 0.94, 0.9, 0.85, 0.8, 0.75, 0.7, 0.6, 0.5}) {
            const int64_t e1 = (int64_t)(N * f);
            const int64_t Pr = ho[e1];
            const double by = (double)Pr * (NQ * 4 + 12) + (double)e1 * NQ;
            auto ks = [&] { hipLaunchKernelGGL((tq::k_persp_stream<D, float, NSP, NPP, 8, 14, 12>), dim3(256), dim3(64 * (NSP + 1 + NPP)), 0, 0, vp, N, off, ob, p2, Pr, err, (int64_t)0, e1, (const int32_t*)nullptr, 8, 0, (unsigned int*)nullptr, (unsigned long long*)nullptr); };
            float a = 0; for (int r = 0; r < 6; ++r) { float x = t.run(ks); if (r) a += x; }
            printf("    f=%.2f  spacing %8.3f MB  %6.0f GB/s\n", f, (double)Pr * NQ * 4 / 256 / 1e6, by / (a / 5) / 1e6);
        }
    }
    const int fast = (int)(std::max_element(rate.begin(), rate.end()) - rate.begin()), slow = (int)(std::min_element(rate.begin(), rate.end()) - rate.begin());
    for (int which : {fast, slow}) {
        float* ob = bufs[which];
        printf("storer / producer waves and window size, on buffer %d (%s):\n", which, which == fast ? "fastest" : "slowest");
#define SW(NS, NP, CPW) { auto k = [&] { hipLaunchKernelGGL((tq::k_persp_stream<D, float, NS, NP, CPW, 14, 12>), dim3(256), dim3(64 * (NS + 1 + NP)), 0, 0, vp, N, off, ob, p2, P, err, (int64_t)0, N, split, 8, 0, (unsigned int*)nullptr, (unsigned long long*)nullptr); }; \
        printf("    NS=%d NP=%2d CPW=%2d  %6.0f GB/s\n", NS, NP, CPW, timeit(k)); }
        SW(4, 11, 8) SW(2, 13, 8) SW(4, 7, 8) SW(4, 3, 8) SW(6, 9, 8) SW(8, 7, 8) SW(4, 11, 4) SW(4, 11, 32) SW(2, 5, 8)
        auto kl = [&] { hipLaunchKernelGGL((tq::k_persp_write<D, float, 64>), dim3((unsigned)N), dim3(64), 0, 0, vp, N, off, ob, p2, P, err, (int64_t)0, N); };
        printf("    one wave per lattice %6.0f GB/s\n", timeit(kl));
    }
    stats_run<D, NSP, NPP, 8>(vp, N, off, bufs[fast], p2, P, err, split);
    return 0;
}

int main(int argc, char** argv) {
    const int d = argc > 1 ? atoi(argv[1]) : 7;
    const int64_t N = argc > 2 ? atoll(argv[2]) : 65536;
    const double q = argc > 3 ? atof(argv[3]) : (d == 7 ? 0.29 : 0.31);
    if (d == 3) return run<3>(N, q);
    if (d == 5) return run<5>(N, q);
    if (d == 7) return run<7>(N, q);
    if (d == 9) return run<9>(N, q);
    if (d == 11) return run<11>(N, q);
    printf("d must be 3, 5, 7, 9 or 11\n");
    return 1;
}
