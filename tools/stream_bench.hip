// Stand-alone harness for the two stack-write kernels (one wave per lattice / producer-storer stream) on synthetic
// syndromes: same inputs, outputs compared, launches timed with HIP events, and the per-role wait statistics of
// the stream kernel (STATS instantiation).  Build on the GPU box:
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Iinclude -Itoric-rl-decoder_amd/csrc tools/stream_bench.hip -o tools/stream_bench
//   tools/stream_bench [d=7|9] [lattices=65536] [q=0.29]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <vector>

#include "stream_write.hpp"

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

struct Timer {
    hipEvent_t e0, e1;
    Timer() { CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); }
    template <class F> float run(F f) {
        CK(hipEventRecord(e0)); f(); CK(hipGetLastError()); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
    }
};

template <int D, int NS, int NP, int CPW>
void stats_run(const char* name, const uint64_t* vp, int64_t N, const int64_t* off, float* out, int32_t* pos, int64_t cap, int* err,
               const int32_t* split) {
    constexpr int WV = NS + 1 + NP, G = 256;
    unsigned long long* st;
    CK(hipMalloc(&st, sizeof(unsigned long long) * G * WV * 4));
    CK(hipMemset(st, 0, sizeof(unsigned long long) * G * WV * 4));
    hipLaunchKernelGGL((tq::k_persp_stream<D, float, NS, NP, CPW, 14, 12, true>), dim3(G), dim3(64 * WV), 0, 0, vp, N, off, out, pos, cap,
                       err, (int64_t)0, N, split, st);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)G * WV * 4);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    auto agg = [&](int w0, int w1, const char* role) {
        double tot = 0, a = 0, b = 0, items = 0, mx = 0; int n = 0;
        for (int g = 0; g < G; ++g) for (int w = w0; w < w1; ++w) {
            const unsigned long long* o = &h[((size_t)g * WV + w) * 4];
            if (!o[0]) continue;
            tot += o[0]; a += o[1]; b += o[2]; items += o[3]; mx = std::max(mx, (double)o[0]); ++n;
        }
        if (n) printf("  %-22s %-10s waves %5d  alive %9.0f cyc (max %9.0f)  wait A %5.1f %%  wait B %5.1f %%  items/wave %7.1f  cyc/item (busy) %7.0f\n",
                      name, role, n, tot / n, mx, 100 * a / tot, 100 * b / tot, items / n, (tot - a - b) / std::max(1.0, items));
    };
    agg(0, NS, "storer");
    agg(NS, NS + 1, "positions");
    agg(NS + 1, WV, "producer");
    CK(hipFree(st));
}

template <int D, int NS, int NP, int K>
void win_stats_run(const char* name, int G, const uint64_t* vp, int64_t N, const int64_t* off, float* out, int32_t* pos, int64_t cap,
                   int* err, const int32_t* widx, const int32_t* pidx) {
    constexpr int WV = NS + 1 + NP;
    unsigned long long* st;
    CK(hipMalloc(&st, sizeof(unsigned long long) * G * WV * 4));
    CK(hipMemset(st, 0, sizeof(unsigned long long) * G * WV * 4));
    hipLaunchKernelGGL((tq::k_persp_windows<D, float, NS, NP, K, true>), dim3(G), dim3(64 * WV), 0, 0, vp, N, off, out, pos, cap, err,
                       (int64_t)0, N, widx, pidx, st);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)G * WV * 4);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    auto agg = [&](int w0, int w1, const char* role) {
        double tot = 0, a = 0, items = 0, mx = 0; int n = 0;
        for (int g = 0; g < G; ++g) for (int w = w0; w < w1; ++w) {
            const unsigned long long* o = &h[((size_t)g * WV + w) * 4];
            if (!o[0]) continue;
            tot += o[0]; a += o[1]; items += o[3]; mx = std::max(mx, (double)o[0]); ++n;
        }
        if (n) printf("  %-26s %-10s waves %5d  alive %9.0f cyc (max %9.0f)  waiting %5.1f %%  items/wave %7.1f  cyc/item (busy) %7.0f\n",
                      name, role, n, tot / n, mx, 100 * a / tot, items / n, (tot - a) / std::max(1.0, items));
    };
    agg(0, NS, "storer");
    agg(NS, NS + 1, "positions");
    agg(NS + 1, WV, "producer");
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
    int32_t* split; CK(hipMalloc(&split, 4 * 258));
    int32_t *widx, *pidx;
    CK(hipMalloc(&widx, 4 * (((size_t)N * NQ * NQ >> tq::WIN_LOG) + 2))); CK(hipMalloc(&pidx, 4 * (((size_t)N * NQ * 3 >> tq::PWIN_LOG) + 2)));
    int* err; CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    hipLaunchKernelGGL(k_counts<D>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, vp, counts, N, part);
    hipLaunchKernelGGL(tq::k_scan_final, dim3((unsigned)((N + tq::SCAN_CHUNK - 1) / tq::SCAN_CHUNK)), dim3(256), 0, 0, counts,
                       (const int64_t*)part, off, (int32_t*)nullptr, N, split, 8, widx, pidx, NQ);
    CK(hipDeviceSynchronize());
    int64_t P; CK(hipMemcpy(&P, off + N, 8, hipMemcpyDeviceToHost));
    const double bytes = (double)P * (NQ * 4 + 12) + (double)N * NQ;
    printf("d=%d  lattices %lld  perspectives %lld (%.1f per lattice)  algorithmic bytes %.3f GB\n", D, (long long)N, (long long)P, (double)P / N, bytes / 1e9);
    float *o1, *o2; int32_t *p1, *p2;
    CK(hipMalloc(&o1, (size_t)P * NQ * 4 + 4096)); CK(hipMalloc(&o2, (size_t)P * NQ * 4 + 4096));
    CK(hipMalloc(&p1, (size_t)P * 12 + 4096)); CK(hipMalloc(&p2, (size_t)P * 12 + 4096));
    CK(hipMemset(o1, 0xff, (size_t)P * NQ * 4)); CK(hipMemset(o2, 0x77, (size_t)P * NQ * 4));
    CK(hipMemset(p1, 0xff, (size_t)P * 12)); CK(hipMemset(p2, 0x77, (size_t)P * 12));
    auto lattice = [&] { hipLaunchKernelGGL((tq::k_persp_write<D, float, 64>), dim3((unsigned)N), dim3(64), 0, 0, vp, N, off, o1, p1, P, err, (int64_t)0, N); };
#define STREAMK(NS, NP, CPW) [&] { hipLaunchKernelGGL((tq::k_persp_stream<D, float, NS, NP, CPW, 14, 12>), dim3(256), dim3(64 * (NS + 1 + NP)), 0, 0, vp, N, off, o2, p2, P, err, (int64_t)0, N, split, (unsigned long long*)nullptr); }
#define WINK(G, NS, NP, K) [&] { hipLaunchKernelGGL((tq::k_persp_windows<D, float, NS, NP, K>), dim3(G), dim3(64 * (NS + 1 + NP)), 0, 0, vp, N, off, o2, p2, P, err, (int64_t)0, N, widx, pidx, (unsigned long long*)nullptr); }
    auto s0 = STREAMK(4, 11, 8);
    auto s1 = WINK(256, 4, 11, 32);
    auto s2 = WINK(256, 8, 7, 32);
#define WINJ(G, NS, NP, K, J) [&] { hipLaunchKernelGGL((tq::k_persp_windows<D, float, NS, NP, K, false, J>), dim3(G), dim3(64 * (NS + 1 + NP)), 0, 0, vp, N, off, o2, p2, P, err, (int64_t)0, N, widx, pidx, (unsigned long long*)nullptr); }
#define WINM(G, NS, NP, K, M) [&] { hipLaunchKernelGGL((tq::k_persp_windows<D, float, NS, NP, K, false, 1, M>), dim3(G), dim3(64 * (NS + 1 + NP)), 0, 0, vp, N, off, o2, p2, P, err, (int64_t)0, N, widx, pidx, (unsigned long long*)nullptr); }
    auto s3 = WINM(256, 4, 11, 32, 1);
    auto s4 = WINM(256, 4, 11, 32, 2);
    auto s7 = WINM(256, 4, 11, 32, 3);
    auto s8 = WINM(256, 8, 7, 32, 1);
    auto s9 = WINM(512, 4, 3, 16, 1);
    auto s10 = WINM(1024, 4, 3, 16, 1);
    auto s5 = WINK(512, 4, 11, 32);
    auto s6 = WINK(256, 6, 9, 32);
    Timer t;
    lattice(); s1(); CK(hipDeviceSynchronize());
    unsigned long long* bad; CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(k_diff, dim3(2048), dim3(256), 0, 0, (const uint32_t*)o1, (const uint32_t*)o2, (int64_t)P * NQ, bad);
    hipLaunchKernelGGL(k_diff, dim3(256), dim3(256), 0, 0, (const uint32_t*)p1, (const uint32_t*)p2, (int64_t)P * 3, bad);
    unsigned long long nbad; CK(hipMemcpy(&nbad, bad, 8, hipMemcpyDeviceToHost));
    int e; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
    printf("stream vs lattice: %llu differing dwords, error latch %d\n", nbad, e);
    struct V { const char* name; std::vector<float> ms; };
    std::vector<V> vs = {{"lattice (1 wave/lattice)", {}}, {"stream NS=4 NP=11 CPW=8", {}}, {"windows G=256 NS=4 NP=11", {}},
                         {"windows G=256 NS=8 NP=7", {}}, {"win 256/4: storers only, const", {}}, {"win 256/4: storers only, LDS", {}},
                         {"hipMemsetAsync (stack bytes)", {}}, {"windows G=512 NS=4 NP=11", {}}, {"windows G=256 NS=6 NP=9", {}},
                         {"win 256/4: no positions", {}}, {"win 256/8: storers only, const", {}}, {"win 512/4: storers only, const", {}},
                         {"win 1024/4: storers only, const", {}}};
    for (int rep = 0; rep < 12; ++rep) {
        vs[0].ms.push_back(t.run(lattice));
        vs[1].ms.push_back(t.run(s0));
        vs[2].ms.push_back(t.run(s1));
        vs[3].ms.push_back(t.run(s2));
        vs[4].ms.push_back(t.run(s3));
        vs[5].ms.push_back(t.run(s4));
        vs[6].ms.push_back(t.run([&] { (void)hipMemsetAsync(o2, 1, (size_t)P * NQ * 4, 0); }));
        vs[7].ms.push_back(t.run(s5));
        vs[8].ms.push_back(t.run(s6));
        vs[9].ms.push_back(t.run(s7));
        vs[10].ms.push_back(t.run(s8));
        vs[11].ms.push_back(t.run(s9));
        vs[12].ms.push_back(t.run(s10));
    }
    for (auto& v : vs) {
        std::vector<float> m(v.ms.begin() + 2, v.ms.end());
        double sum = 0; for (float x : m) sum += x;
        const double avg = sum / m.size(), best = *std::min_element(m.begin(), m.end());
        printf("%-30s avg %7.1f us  %6.0f GB/s   best %7.1f us  %6.0f GB/s\n", v.name, 1e3 * avg, bytes / avg / 1e6, 1e3 * best, bytes / best / 1e6);
    }
    // H1: does the rate of the stream kernel depend on WHICH buffer it writes (physical placement)?  Then the
    // configuration sweep on the fastest and on the slowest of six buffers.
    {
        std::vector<float*> bufs; std::vector<double> rate;
        for (int b = 0; b < 6; ++b) { float* x; CK(hipMalloc(&x, (size_t)P * NQ * 4 + 4096 + (size_t)b * (3u << 20))); CK(hipMemset(x, 0, (size_t)P * NQ * 4)); bufs.push_back(x); }
        auto timeit = [&](auto k) { float a = 0; for (int r = 0; r < 6; ++r) { float x = t.run(k); if (r) a += x; } return bytes / (a / 5) / 1e6; };
        for (int b = 0; b < 6; ++b) {
            float* ob = bufs[b];
            auto ks = [&] { hipLaunchKernelGGL((tq::k_persp_stream<D, float, 4, 11, 8, 14, 12>), dim3(256), dim3(1024), 0, 0, vp, N, off, ob, p2, P, err, (int64_t)0, N, split, (unsigned long long*)nullptr); };
            auto kl = [&] { hipLaunchKernelGGL((tq::k_persp_write<D, float, 64>), dim3((unsigned)N), dim3(64), 0, 0, vp, N, off, ob, p2, P, err, (int64_t)0, N); };
            auto km = [&] { (void)hipMemsetAsync(ob, 1, (size_t)P * NQ * 4, 0); };
            const double rs = timeit(ks), rl = timeit(kl), rm = timeit(km);
            rate.push_back(rs);
            printf("  buffer %d at %p: stream %6.0f GB/s   lattice %6.0f GB/s   memset %6.0f GB/s\n", b, (void*)ob, rs, rl, rm * ((double)P * NQ * 4) / bytes);
        }
        const int fast = (int)(std::max_element(rate.begin(), rate.end()) - rate.begin()), slow = (int)(std::min_element(rate.begin(), rate.end()) - rate.begin());
        for (int which : {fast, slow}) {
            float* ob = bufs[which];
            printf("  sweep on buffer %d (%s):\n", which, which == fast ? "fastest" : "slowest");
#define SW(NS, NP, CPW) { auto k = [&] { hipLaunchKernelGGL((tq::k_persp_stream<D, float, NS, NP, CPW, 14, 12>), dim3(256), dim3(64 * (NS + 1 + NP)), 0, 0, vp, N, off, ob, p2, P, err, (int64_t)0, N, split, (unsigned long long*)nullptr); }; \
            printf("    NS=%d NP=%2d CPW=%2d  %6.0f GB/s\n", NS, NP, CPW, timeit(k)); }
            SW(4, 11, 8) SW(2, 13, 8) SW(4, 7, 8) SW(4, 5, 8) SW(4, 3, 8) SW(6, 9, 8) SW(8, 7, 8) SW(3, 6, 8) SW(4, 11, 4) SW(4, 11, 16) SW(4, 11, 32) SW(2, 5, 8) SW(2, 5, 32)
            auto kl = [&] { hipLaunchKernelGGL((tq::k_persp_write<D, float, 64>), dim3((unsigned)N), dim3(64), 0, 0, vp, N, off, ob, p2, P, err, (int64_t)0, N); };
            printf("    lattice            %6.0f GB/s\n", timeit(kl));
        }
        for (auto x : bufs) CK(hipFree(x));
    }
    // correctness of every windows variant against the lattice kernel
    int vi = 0;
    for (auto* f : {(void*)0}) { (void)f; }
    auto verify = [&](const char* name, auto launch) {
        CK(hipMemset(o2, 0x77, (size_t)P * NQ * 4)); CK(hipMemset(p2, 0x77, (size_t)P * 12)); CK(hipMemset(bad, 0, 8));
        launch(); CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_diff, dim3(2048), dim3(256), 0, 0, (const uint32_t*)o1, (const uint32_t*)o2, (int64_t)P * NQ, bad);
        hipLaunchKernelGGL(k_diff, dim3(256), dim3(256), 0, 0, (const uint32_t*)p1, (const uint32_t*)p2, (int64_t)P * 3, bad);
        unsigned long long nb; CK(hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost));
        int ee; CK(hipMemcpy(&ee, err, 4, hipMemcpyDeviceToHost));
        printf("  verify %-28s %llu differing dwords, latch %d\n", name, nb, ee);
        ++vi;
    };
    verify("stream", s0); verify("win 256/8/7", s2);  verify("win 512/4/11", s5); verify("win 256/6/9", s6);
    stats_run<D, 4, 11, 8>("stream NS=4 NP=11 CPW=8", vp, N, off, o2, p2, P, err, split);
    win_stats_run<D, 4, 11, 32>("windows G=256 NS=4 NP=11", 256, vp, N, off, o2, p2, P, err, widx, pidx);
    win_stats_run<D, 8, 7, 32>("windows G=256 NS=8 NP=7", 256, vp, N, off, o2, p2, P, err, widx, pidx);

    return 0;
}

int main(int argc, char** argv) {
    const int d = argc > 1 ? atoi(argv[1]) : 7;
    const int64_t N = argc > 2 ? atoll(argv[2]) : 65536;
    const double q = argc > 3 ? atof(argv[3]) : (d == 7 ? 0.29 : 0.31);
    if (d == 7) return run<7>(N, q);
    if (d == 9) return run<9>(N, q);
    printf("d must be 7 or 9\n");
    return 1;
}
