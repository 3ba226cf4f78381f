// Perspective stack write, windowed producer / storer form (gfx950 / CDNA4).
//
// The stack (P,2,d,d) is cut into aligned WINDOWS of WE = 8192 elements (32 KiB of f32) and the windows are dealt
// round-robin to G persistent workgroups (window w -> workgroup w mod G), so that at any moment the chip writes one
// compact front of a few thousand consecutive windows -- the store shape that runs at memset speed on this part
// (profiles/r01_membench*.txt: aligned 32 KiB chunks per wave 6.3-6.7 TB/s; one variable-length segment per wave 5.8).
// Inside a workgroup the two jobs are done by different waves, so a wave that stores never computes:
//   * NP producer waves: producer p builds the windows i = p, p+NP, ... of its workgroup as bit strings in a ring of
//     K slots in LDS (bit x of slot = element w*WE + x).  The lattices under a window come from the index the scan
//     leaves behind (widx[w] = lattice that holds the window's first element); a lattice that straddles a window
//     border is built by both neighbours, each emitting only its own perspectives (lattice.hpp, PStream: rotated
//     planes by ballot, table of row-rolled planes, one lane per hit, ds_or_b32).  Guard words on both sides of a
//     slot take the halves of the perspectives that hang over the border.
//   * NS storer waves: storer s takes the windows i = s, s+NS, ...:  ds_read_b32 -> shift -> bit->element expansion
//     -> global_store_dwordx4, 4 KiB per trip, and hands the slot back zeroed.
//   * one wave writes the positions (P,3) in windows of PWE = 8192 dwords, building the hit lists it needs itself.
// Hand-off per slot: ready[slot] / freed[slot] use counters (plain LDS words, polled with s_sleep).  No chain between
// lattices, no ownership rule for cache lines: a window is whole 128-byte lines and has exactly one writer.
// Every poll loop is bounded; a wave that gives up raises `abort` for its workgroup and latches ERR_INTERNAL.
#pragma once
#include "kernels.hpp"

namespace tq {

constexpr int ERR_INTERNAL = 32;
constexpr int WIN_SPIN_LIMIT = 1 << 21;

__device__ __forceinline__ uint32_t lds_peek(const uint32_t& w) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
// Hand-off words live in LDS, which one workgroup's waves see coherently, and a wave's LDS operations execute in
// issue order: publishing needs no memory fence, only the COMPILER must keep the order (a workgroup-scope release
// would also drain vmcnt, i.e. stall a storer on its own global stores).
__device__ __forceinline__ void lds_after_peek() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
__device__ __forceinline__ void lds_publish(uint32_t& w, uint32_t v, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (lane == 0) __hip_atomic_store(&w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint64_t readlane64(uint64_t x, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

// Window index of a lattice range for offsets that did not come with the scan (sub-ranges, foreign offsets):
// widx[w] = the lattice that holds element w*WE of the range's stack, pidx[w] = the lattice that holds dword w*PWE of
// its positions.  One thread per lattice: lattice e owns the windows that START inside its elements.
__global__ __launch_bounds__(256) void k_window_index(const int64_t* __restrict__ offsets, int64_t e_begin, int64_t e_end, int nq,
                                                      int32_t* __restrict__ widx, int32_t* __restrict__ pidx) {
    const int64_t e = e_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= e_end) return;
    const int64_t off0 = offsets[e_begin], a = offsets[e] - off0, b = offsets[e + 1] - off0;
    if (b == a) return;
    for (int64_t w = (a * nq + (1 << WIN_LOG) - 1) >> WIN_LOG; w < ((b * nq + (1 << WIN_LOG) - 1) >> WIN_LOG); ++w) widx[w] = (int32_t)e;
    for (int64_t w = (a * 3 + (1 << PWIN_LOG) - 1) >> PWIN_LOG; w < ((b * 3 + (1 << PWIN_LOG) - 1) >> PWIN_LOG); ++w) pidx[w] = (int32_t)e;
}

template <int D>
struct WinTables {                                             // private to one producer wave
    static constexpr int NQP = (Lat<D>::NQ + 7) & ~7;
    uint64_t rr[4][D][Lat<D>::W];                              // V, P, rot V, rot P rolled by every row amount
    uint64_t low[D][Lat<D>::W];                                // lowcols(k)
    uint32_t hpos[NQP];                                        // k-th hit -> layer | row << 8 | col << 16
};

template <int D, int NP, int K>
struct WinLds {
    static constexpr int GUARD = PStream<D>::ND + 1;           // dwords on either side of a slot: a perspective hangs over by < NQ bits
    static constexpr int SLOT = (1 << WIN_LOG) / 32 + 2 * GUARD;
    static constexpr int PBUF = (1 << PWIN_LOG) / 3 + 4;
    __attribute__((aligned(16))) uint32_t bits[K][SLOT];       // ring of window bit strings
    uint32_t posbuf[PBUF];                                     // packed positions of the perspectives of one positions window
    WinTables<D> tab[NP];
    uint32_t ready[K];                                         // slot s: windows produced into it so far
    uint32_t freed[K];                                         // slot s: windows stored from it (and zeroed) so far
    uint32_t abort;
};

// OR the NQ-bit string of one perspective into a slot at bit position `pos`
template <int D, class OrFn>
__device__ __forceinline__ void emit_at(uint32_t pos, const typename Lat<D>::B& ov, const typename Lat<D>::B& op, OrFn&& orfn) {
    using S = PStream<D>;
    const uint32_t base = pos >> 5;
    const int sh = (int)(pos & 31);
    uint32_t prev = 0;
#pragma unroll
    for (int j = 0; j <= S::ND; ++j) {
        const uint32_t cur = j < S::ND ? S::string_dword(ov, op, j) : 0u;
        const uint32_t val = (cur << sh) | ((prev >> 1) >> (31 - sh));
        if (j < S::ND || val) orfn(base + j, val);
        prev = cur;
    }
}

// STATS (diagnostic builds only, tools/stream_bench.hip): every wave leaves {cycles alive, cycles waiting, 0, items}.
// MODE (diagnostic builds only): 1 = storers alone, constant data, no LDS; 2 = storers alone, LDS read + hand-back;
// 3 = everything but the positions; 0 = the product.
template <int D, typename OutT, int NS, int NP, int K, bool STATS = false, int J = 1, int MODE = 0>
__global__ __launch_bounds__(64 * (NS + 1 + NP)) void k_persp_windows(const uint64_t* __restrict__ vp, int64_t N,
                                                                   const int64_t* __restrict__ offsets, OutT* __restrict__ out,
                                                                   int32_t* __restrict__ pos, int64_t capacity,
                                                                   int* __restrict__ err, int64_t e_begin, int64_t e_end,
                                                                   const int32_t* __restrict__ widx, const int32_t* __restrict__ pidx,
                                                                   unsigned long long* __restrict__ stats = nullptr) {
    using L = Lat<D>;
    using PS = PStream<D>;
    using Enc = OutEnc<OutT>;
    using B = typename L::B;
    using LDS = WinLds<D, NP, K>;
    constexpr int DD = L::DD, NQ = L::NQ, W = L::W;
    constexpr int VEC = 16 / (int)sizeof(OutT);              // elements per 16-byte lane store
    constexpr int EPC = 64 * VEC;                            // elements per chunk (one wave store instruction = 1 KiB)
    constexpr int LPD = 32 / VEC;                            // lanes that share one ring dword
    constexpr int WE = 1 << WIN_LOG, PWE = 1 << PWIN_LOG;
    constexpr int CPWIN = WE / EPC;                          // chunks per window: 32 (f32), 16 (16-bit), 8 (u8)
    constexpr int U = CPWIN < 4 ? CPWIN : 4;                 // chunks per trip
    constexpr int GUARD = LDS::GUARD;
    __shared__ LDS S;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int G = gridDim.x;
    unsigned long long t_begin = 0, t_a = 0, n_items = 0;
    if (STATS) t_begin = __builtin_readcyclecounter();
    auto stats_out = [&]() {
        if (STATS && lane == 0) {
            unsigned long long* o = stats + ((size_t)blockIdx.x * (NS + 1 + NP) + wave) * 4;
            o[0] = __builtin_readcyclecounter() - t_begin; o[1] = t_a; o[2] = 0; o[3] = n_items;
        }
    };

    // ---- extent of the stack (wave-uniform, scalar)
    const int64_t off0 = offsets[e_begin];
    int64_t p_all = offsets[e_end] - off0;                   // perspectives of the whole stack
    int64_t e_stop = e_end;                                  // lattices from e_stop on are not written
    if (p_all > capacity) {                                  // stack does not fit: only the lattices that fit whole are written
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(err, ERR_CAPACITY);
        int64_t lo = e_begin, hi = e_end;                    // largest e with offsets[e] - off0 <= capacity
        while (lo < hi) {
            const int64_t mid = (lo + hi + 1) >> 1;
            if (offsets[mid] - off0 <= capacity) lo = mid; else hi = mid - 1;
        }
        e_stop = lo;
        p_all = offsets[e_stop] - off0;
    }
    const int64_t E_all = p_all * NQ, X_all = p_all * 3;     // elements of the stack, dwords of the positions
    const int64_t nwin = (E_all + WE - 1) >> WIN_LOG, npwin = pos ? (X_all + PWE - 1) >> PWIN_LOG : 0;
    const int64_t my_nwin = nwin > blockIdx.x ? (nwin - blockIdx.x + G - 1) / G : 0;       // windows blockIdx.x, +G, ...
    const int64_t my_npwin = npwin > blockIdx.x ? (npwin - blockIdx.x + G - 1) / G : 0;
    if (my_nwin == 0 && my_npwin == 0) return;               // uniform over the workgroup

    // ---- ring and hand-off words
    for (uint32_t i = threadIdx.x; i < (uint32_t)(K * LDS::SLOT); i += blockDim.x) (&S.bits[0][0])[i] = 0u;
    if (threadIdx.x < K) { S.ready[threadIdx.x] = 0u; S.freed[threadIdx.x] = 0u; }
    if (threadIdx.x == 0) S.abort = 0u;
    __syncthreads();

    auto give_up = [&]() {
        if (lane == 0) { __hip_atomic_store(&S.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); atomicOr(err, ERR_INTERNAL); }
    };
    // bounded wait until the word reaches `want`; false when the workgroup gave up
    auto wait_for = [&](const uint32_t& word, uint32_t want) {
        if (lds_peek(word) == want) { lds_after_peek(); return true; }
        unsigned long long t0 = 0;
        if (STATS) t0 = __builtin_readcyclecounter();
        for (int spin = 0; spin < WIN_SPIN_LIMIT; ++spin) {
            __builtin_amdgcn_s_sleep(4);
            if (lds_peek(word) == want) { lds_after_peek(); if (STATS) t_a += __builtin_readcyclecounter() - t0; return true; }
            if (lds_peek(S.abort)) return false;
        }
        give_up();
        return false;
    };

    if (wave < NS) {
        // =========================================================== stack storer
        __builtin_amdgcn_s_setprio(3);                       // store issue goes before the producers' arithmetic
        const uint32_t lane_el = (uint32_t)lane * VEC;
        const int sh = (int)(lane_el & 31u);
        const uint32_t lane_dw = lane_el >> 5;
        const bool zero_lane = (lane % LPD) == 0;
        // A storer works on J windows at a time, trip by trip in turn, so that its stores in flight go to J different
        // 32 KiB regions (J = 1: one window after the other).  Windows i = (r*NS + wave)*J + j.
        for (int64_t ib = (int64_t)wave * J; ib < my_nwin; ib += (int64_t)NS * J) {
            int slot[J]; uint32_t use[J], nel[J]; int64_t E0[J]; bool live[J];
            bool all_full = true;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int64_t i = ib + j;
                live[j] = i < my_nwin;
                slot[j] = (int)(i % K); use[j] = (uint32_t)(i / K);
                E0[j] = (blockIdx.x + i * G) << WIN_LOG;
                nel[j] = live[j] ? (uint32_t)(E_all - E0[j] < WE ? E_all - E0[j] : WE) : 0u;
                all_full = all_full && nel[j] == WE;
                if (live[j]) { if (MODE != 1 && MODE != 2) { if (!wait_for(S.ready[slot[j]], use[j] + 1u)) return; } if (STATS) ++n_items; }
            }
            if (all_full) {
#pragma unroll 2
                for (int t = 0; t < CPWIN / U; ++t) {        // U x 1 KiB per window per trip
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        uint32_t* __restrict__ sb = &S.bits[slot[j]][GUARD];
                        char* __restrict__ obase = reinterpret_cast<char*>(out + E0[j]);
                        uint32_t wv[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) wv[u] = MODE == 1 ? 0x5a5a5a5au : sb[(t * U + u) * (EPC / 32) + lane_dw];
                        if (zero_lane && MODE != 1) {
#pragma unroll
                            for (int u = 0; u < U; ++u) sb[(t * U + u) * (EPC / 32) + lane_dw] = 0u;
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u)
                            *reinterpret_cast<u32x4*>(obase + ((size_t)(t * U + u) * EPC + lane_el) * sizeof(OutT)) = expand_bits<OutT>(wv[u] >> sh);
                    }
                }
            } else {                                         // the group holds the last window of the stack
                for (int j = 0; j < J; ++j) {
                    if (!live[j]) continue;
                    uint32_t* __restrict__ sb = &S.bits[slot[j]][GUARD];
                    char* __restrict__ obase = reinterpret_cast<char*>(out + E0[j]);
                    for (int c = 0; c < CPWIN; ++c) {
                        const uint32_t el = (uint32_t)c * EPC + lane_el;
                        const uint32_t wv = sb[c * (EPC / 32) + lane_dw];
                        if (zero_lane) sb[c * (EPC / 32) + lane_dw] = 0u;
                        const u32x4 val = expand_bits<OutT>(wv >> sh);
                        if (el + VEC <= nel[j]) {
                            *reinterpret_cast<u32x4*>(obase + (size_t)el * sizeof(OutT)) = val;
                        } else if (el < nel[j]) {            // the stack ends inside this lane's 16 bytes
                            const int m = (int)(nel[j] - el);
                            if (Enc::BITS == 32) {
                                for (int q = 0; q < m; ++q) reinterpret_cast<uint32_t*>(obase)[el + q] = val[q];
                            } else if (Enc::BITS == 16) {
                                for (int q = 0; q < m; ++q) reinterpret_cast<uint16_t*>(obase)[el + q] = (uint16_t)(val[q >> 1] >> (16 * (q & 1)));
                            } else {
                                for (int q = 0; q < m; ++q) reinterpret_cast<uint8_t*>(obase)[el + q] = (uint8_t)(val[q >> 2] >> (8 * (q & 3)));
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < J; ++j) {
                if (!live[j]) continue;
                if (lane < 2 * GUARD) S.bits[slot[j]][lane < GUARD ? lane : WE / 32 + lane] = 0u;     // the guard words on both sides
                lds_publish(S.freed[slot[j]], use[j] + 1u, lane);
            }
        }
        stats_out();
        return;
    }

    if (MODE == 1 || MODE == 2 || (MODE == 3 && wave == NS)) return;
    if (wave == NS) {
        // =========================================================== positions: hit lists and stores by one wave
        for (int64_t i = 0; i < my_npwin; ++i) {
            const int64_t w = blockIdx.x + i * G;
            const int64_t X0 = w << PWIN_LOG;
            const uint32_t nx = (uint32_t)(X_all - X0 < PWE ? X_all - X0 : PWE);           // dwords of this window
            const int64_t q_first = X0 / 3;
            const uint32_t r0 = (uint32_t)(X0 - 3 * q_first);
            const uint32_t nq_win = (r0 + nx + 2u) / 3u;                                    // perspectives it touches
            if (STATS) ++n_items;
            for (int64_t Lb = 0;; Lb += 64) {
                const int64_t e_l = (int64_t)pidx[w] + Lb + lane;
                const bool in = e_l < e_stop;
                uint64_t vv[W], pp[W];
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    vv[k] = in ? vp[(int64_t)k * N + e_l] : 0ull;
                    pp[k] = in ? vp[((int64_t)W + k) * N + e_l] : 0ull;
                }
                const int64_t oo = in ? offsets[e_l] - off0 - q_first : (int64_t)0x3fffffffffffffffll;   // first perspective, window-relative
                const uint64_t inmask = __ballot(in && oo < (int64_t)nq_win);
                const int cnt = __popcll(inmask);
                for (int j = 0; j < cnt; ++j) {
                    B v, pl, e0, e1;
#pragma unroll
                    for (int k = 0; k < W; ++k) { v.w[k] = readlane64(vv[k], j); pl.w[k] = readlane64(pp[k], j); }
                    L::hit_masks(v, pl, e0, e1);
                    const int n0 = e0.popc();
                    const int64_t q0 = (int64_t)readlane64((uint64_t)oo, j);
                    for (int c = lane; c < NQ; c += 64) {
                        const int l = c >= DD, bit = c - l * DD;
                        if (l ? e1.get(bit) : e0.get(bit)) {
                            const int row = bit / D, col = bit - row * D;
                            const int64_t q = q0 + (l ? n0 + e1.rank(bit) : e0.rank(bit));
                            if (q >= 0 && q < (int64_t)nq_win) S.posbuf[q] = (uint32_t)l | ((uint32_t)row << 8) | ((uint32_t)col << 16);
                        }
                    }
                }
                if (cnt < 64) break;
            }
            wave_lds_sync();
            int32_t* __restrict__ pbase = pos + X0;
            for (uint32_t x = 4u * (uint32_t)lane; x < nx; x += 256u) {
                int o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t t = x + j + r0, q = t / 3u;
                    o[j] = (int)((S.posbuf[q < (uint32_t)LDS::PBUF ? q : 0u] >> (8u * (t - 3u * q))) & 255u);
                }
                if (x + 4u <= nx) *reinterpret_cast<int4*>(pbase + x) = make_int4(o[0], o[1], o[2], o[3]);
                else for (uint32_t j = 0; x + j < nx; ++j) pbase[x + j] = o[j];
            }
            wave_lds_sync();                                 // posbuf is rewritten by the next window
        }
        stats_out();
        return;
    }

    // =============================================================== producer
    const int p = wave - NS - 1;
    WinTables<D>& T = S.tab[p];
    if (lane < D) {                                          // column masks: the same for every lattice
        const B m = L::lowcols(lane);
#pragma unroll
        for (int w = 0; w < W; ++w) T.low[lane][w] = m.w[w];
    }
    for (int64_t i = p; i < my_nwin; i += NP) {
        const int64_t w = blockIdx.x + i * G;
        const int slot = (int)(i % K);
        const uint32_t use = (uint32_t)(i / K);
        const int64_t E0 = w << WIN_LOG;
        const int64_t E1 = E0 + WE < E_all ? E0 + WE : E_all;
        const int64_t e_first = widx[w];
        if (!wait_for(S.freed[slot], use)) return;        // the slot's previous window has been stored and zeroed
        if (STATS) ++n_items;
        uint32_t* __restrict__ sb = &S.bits[slot][0];
        for (int64_t Lb = 0;; Lb += 64) {
            // planes and offsets of the next 64 lattices in one round of vector loads (a window needs two or three)
            const int64_t e_l = e_first + Lb + lane;
            const bool in = e_l < e_stop;
            uint64_t vv[W], pp[W];
#pragma unroll
            for (int k = 0; k < W; ++k) {
                vv[k] = in ? vp[(int64_t)k * N + e_l] : 0ull;
                pp[k] = in ? vp[((int64_t)W + k) * N + e_l] : 0ull;
            }
            // first element of the lattice relative to the window (negative: the lattice began in an earlier window)
            const int64_t so = in ? (offsets[e_l] - off0) * NQ - E0 : (int64_t)0x3fffffffffffffffll;
            const uint64_t inmask = __ballot(in && so < E1 - E0);            // offsets are monotone: a prefix of the lanes
            const int cnt = __popcll(inmask);
            for (int j = 0; j < cnt; ++j) {
                B v, pl, e0, e1;
#pragma unroll
                for (int k = 0; k < W; ++k) { v.w[k] = readlane64(vv[k], j); pl.w[k] = readlane64(pp[k], j); }
                L::hit_masks(v, pl, e0, e1);
                const int n0 = e0.popc();
                const int n = n0 + e1.popc();
                if (n == 0) continue;
                const int64_t s0 = (int64_t)readlane64((uint64_t)so, j);
                // its perspectives that overlap the window: [k_lo, k_hi)
                const int k_lo = s0 < 0 ? (int)((-s0) / NQ) : 0;
                int k_hi = (int)((E1 - E0 - s0 + NQ - 1) / NQ);
                k_hi = k_hi < n ? k_hi : n;
                if (k_lo >= k_hi) continue;
                // ---- tables: rotated planes (ballot), row-rolled planes, hit list
                B rv, rp;
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    const int o = 64 * k + lane;
                    const bool inb = o < DD;
                    const int oc = inb ? o : 0;
                    rv.w[k] = __ballot(inb && v.get(PS::rot_src_v(oc)));
                    rp.w[k] = __ballot(inb && pl.get(PS::rot_src_p(oc)));
                }
                if (lane < 4 * D) {
                    const int sel = lane / D, k = lane - sel * D;
                    B src;
#pragma unroll
                    for (int ww = 0; ww < W; ++ww) src.w[ww] = sel == 0 ? v.w[ww] : (sel == 1 ? pl.w[ww] : (sel == 2 ? rv.w[ww] : rp.w[ww]));
                    const B r = (src.shl(k * D) | src.shr(DD - k * D)) & L::full();
#pragma unroll
                    for (int ww = 0; ww < W; ++ww) T.rr[sel][k][ww] = r.w[ww];
                }
                for (int c = lane; c < NQ; c += 64) {
                    const int l = c >= DD, bit = c - l * DD;
                    if (l ? e1.get(bit) : e0.get(bit)) {
                        const int row = bit / D, col = bit - row * D;
                        T.hpos[l ? n0 + e1.rank(bit) : e0.rank(bit)] = (uint32_t)l | ((uint32_t)row << 8) | ((uint32_t)col << 16);
                    }
                }
                wave_lds_sync();
                // ---- one lane per hit: its perspective as two bit-planes, OR-ed into the slot
                const int32_t bit_base = (int32_t)s0 + GUARD * 32;                          // slot bit of the lattice's element 0
                for (int k = k_lo + lane; k < k_hi; k += 64) {
                    const uint32_t hp = T.hpos[k];
                    const int layer = (int)(hp & 255u), ii = (int)((hp >> 8) & 255u), jj = (int)(hp >> 16);
                    int rs, cs;
                    PS::hit_shifts(layer, ii, jj, rs, cs);
                    B a, c, low;
#pragma unroll
                    for (int ww = 0; ww < W; ++ww) { a.w[ww] = T.rr[2 * layer][rs][ww]; c.w[ww] = T.rr[2 * layer + 1][rs][ww]; low.w[ww] = T.low[cs][ww]; }
                    const B ov = PS::roll_cols_masked(a, cs, low), op = PS::roll_cols_masked(c, cs, low);
                    emit_at<D>((uint32_t)(bit_base + k * NQ), ov, op, [&](uint32_t idx, uint32_t val) { atomicOr(&sb[idx], val); });
                }
                wave_lds_sync();                             // T is rewritten by the next lattice
            }
            if (cnt < 64) break;
        }
        lds_publish(S.ready[slot], use + 1u, lane);
    }
    stats_out();
}

}  // namespace tq
