// Perspective stack write, producer / storer form (gfx950 / CDNA4).
//
// One persistent workgroup per CU owns a CONTIGUOUS range of the output (lattices [e_lo, e_hi), cut by perspective
// count, k_scan_final / find_cut: one fixed share per workgroup, larger for the workgroups on the faster XCDs).  Inside the workgroup
// the two jobs of the stack write are done by different waves:
//   * NP producer waves build lattice bitstreams (lattice.hpp, PStream: rotated planes by ballot, table of
//     row-rolled planes, one lane per hit, ds_or_b32) -- not into a per-wave buffer but into ONE ring in LDS
//     that is the workgroup's output range as a bit string (bit x = element org + x of the stack);
//   * NS storer waves do nothing but  ds_read_b32 -> shift -> bit->element expansion -> global_store_dwordx4
//     along that range, in aligned windows of CPW KiB, and hand the ring words back zeroed;
//   * NPW positions waves write the positions (P,3) from a second ring (one packed dword per perspective).
// Hand-off: `pq[p]` (producer p: how far it is -- the first perspective of its lattice that is not in the rings yet,
// published per lattice and after every pass of 64 hits inside a lattice; its earlier lattices are complete; the
// minimum over the producers is the produced PREFIX of the range, no producer ever waits for another),
// `cons[s]` (low-water mark of storer s), `pcons[w]` (of positions wave w): plain LDS words, polled with s_sleep.
// Every poll loop is bounded; a wave that gives up raises `abort` for its workgroup and latches ERR_INTERNAL, so the
// grid always drains.
//
// Output lines: the stack is cut into 128-byte lines and a workgroup stores the lines whose FIRST element lies
// in its range, whole.  The trailing elements of its last line belong to the first lattice(s) of the next
// range: its producers simply go on for the few perspectives that line needs (`need_extra`).
#pragma once
#include "kernels.hpp"

namespace tq {

constexpr int ERR_INTERNAL = 32;
constexpr int STREAM_SLOT_WORDS = 4;                           // slot counters of one launch of k_persp_stream: large, small, done (+ 1 pad)
constexpr int STREAM_SPIN_LIMIT = 1 << 21;

// Hand-off words live in LDS, which one workgroup's waves see coherently, and a wave's LDS operations execute in
// issue order: publishing needs no memory fence, only the COMPILER must keep the order (a workgroup-scope release
// would also drain vmcnt, i.e. stall a storer on its own global stores); reading needs only the s_waitcnt that the
// use of the value implies.
__device__ __forceinline__ uint32_t lds_peek(const uint32_t& w) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
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

// OR the NQ-bit string of one perspective into the ring at bit position `pos` (the caller's orfn reduces the dword
// index mod the ring)
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

template <int D>
struct LatTables {                                             // of ONE lattice
    static constexpr int NQP = (Lat<D>::NQ + 7) & ~7;
    uint64_t rr[4][D][Lat<D>::W];                              // V, P, rot V, rot P rolled by every row amount
    uint32_t hpos[NQP];                                        // k-th hit -> layer | row << 8 | col << 16
};
// private to one producer wave.  QS = 1: one lattice at a time, its hits done in passes of 64 lanes (a lattice of 73 hits
// pays for two passes).  QS > 1 (small lattices): the hits of CONSECUTIVE lattices are queued and done 64 at a time whichever
// lattice they belong to -- a d=3 lattice has 16 hits, a d=7 lattice 73 -- so up to QS lattices' tables are alive at once.
template <int D, int QS>
struct ProdTables {
    static constexpr int QN = QS == 1 ? 1 : (Lat<D>::NQ + 63 <= 128 ? 128 : 256);   // queue entries: < 64 waiting + one lattice's, power of two
    LatTables<D> t[QS];
    uint64_t low[D][Lat<D>::W];                                // lowcols(k): the same for every lattice
    uint32_t qent[QN];                                         // queued hit: number of its lattice (16 bits; slot = number % QS) << 16 | index of the hit in its lattice
    uint32_t qq[QN];                                           //             its perspective (range-relative)
};

template <int D, int NS, int NP, int RB_LOG, int RP_LOG, int NPW = 1, int QS = 1>
struct StreamLds {
    __attribute__((aligned(16))) uint32_t bits[1u << RB_LOG];  // the output range as a bit string, ring
    uint32_t posr[1u << RP_LOG];                               // packed position of perspective q at [q & mask]
    ProdTables<D, QS> tab[NP];
    uint32_t pq[NP];                                           // producer p: first perspective (range-relative) of the lattice it is
                                                               // working on; everything of ITS lattices below that is in the rings;
                                                               // 0xFFFFFFFF = it has no lattice left
    uint32_t cons[NS];                                         // storer s: first element (from org) it has not taken yet
    uint32_t pcons[NPW];                                       // positions wave w: first perspective whose position it has not written yet
    uint32_t abort;
};

// A cut point of a lattice range into G = 1 << LG parts of equal perspective count: the first lattice e in
// [e_begin, e_end] with offsets[e] - offsets[e_begin] >= (total * k) >> LG.  One wavefront, 64-ary search (three
// rounds of vector loads for 65 536 lattices); the result is wave-uniform.  k_scan_final writes the same numbers for
// the whole batch as a by-product; this serves lattice sub-ranges and offsets that did not come from the scan.
__device__ __forceinline__ int64_t find_cut(const int64_t* __restrict__ offsets, int64_t e_begin, int64_t e_end, int k, int LG, int lane) {
    const int64_t off0 = offsets[e_begin], total = offsets[e_end] - off0;
    const int64_t target = off0 + (int64_t)(((uint64_t)total * (uint64_t)k) >> LG);
    int64_t lo = e_begin, hi = e_end;                         // answer in [lo, hi]; offsets[hi] >= target always
    while (lo < hi) {
        const int64_t span = hi - lo;
        const int64_t stepw = (span + 62) / 63;               // probes lo + i*stepw, i = 0..63: lane 63 reaches hi (63*stepw >= span)
        int64_t e = lo + (int64_t)lane * stepw;
        e = e < hi ? e : hi;
        const bool ge = offsets[e] >= target;
        const uint64_t m = __ballot(ge);                      // never empty: lane 63 probes hi
        if (!m) { lo = hi; break; }
        const int f = (int)__ffsll((long long)m) - 1;
        int64_t ef = lo + (int64_t)f * stepw;
        ef = ef < hi ? ef : hi;
        const int64_t new_lo = f == 0 ? lo : (lo + (int64_t)(f - 1) * stepw + 1);
        if (ef == lo) { hi = lo; break; }
        lo = new_lo < ef ? new_lo : ef;
        hi = ef;
    }
    return lo;
}
// all cut points of a range as a table (tools/stream_bench.hip checks it against the scan's by-product)
__global__ __launch_bounds__(256) void k_split(const int64_t* __restrict__ offsets, int64_t e_begin, int64_t e_end,
                                               int32_t* __restrict__ split, int LG) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k > (1 << LG)) return;
    const int64_t e = find_cut(offsets, e_begin, e_end, k, LG, lane);
    if (lane == 0) split[k] = (int32_t)e;
}

// STATS (diagnostic builds only, tools/stream_tune.hip): every wave leaves {cycles alive, cycles waiting, begin << 32 | end on
// the 100 MHz clock, items} in stats[(block * waves + wave) * 4 ..]; waiting = storers: for production, producers: for ring room.
// split / lg / bias / slots: see "this workgroup's range" below (slots: STREAM_SLOT_WORDS zeroed words no other launch in flight uses)
// NPW: positions waves (chunks of 1 KiB dealt round-robin among them)
// QS: lattices whose tables a producer keeps alive (1 = one lattice at a time; > 1 = hit queue across lattices, d <= 7)
template <int D, typename OutT, int NS, int NP, int CPW, int RB_LOG, int RP_LOG, bool STATS = false, int NPW = 1, int QS = 1>
__global__ __launch_bounds__(64 * (NS + NPW + NP)) void k_persp_stream(const uint64_t* __restrict__ vp, int64_t N,
                                                                  const int64_t* __restrict__ offsets, OutT* __restrict__ out,
                                                                  int32_t* __restrict__ pos, int64_t capacity,
                                                                  int* __restrict__ err, int64_t e_begin, int64_t e_end,
                                                                  const int32_t* __restrict__ split, int lg, int bias,
                                                                  unsigned int* __restrict__ slots,
                                                                  unsigned long long* __restrict__ stats = nullptr) {
    using L = Lat<D>;
    using PS = PStream<D>;
    unsigned long long t_begin = 0, t_a = 0, t_rt = 0, n_items = 0, t_first = 0;
    if (STATS) { t_begin = __builtin_readcyclecounter(); t_rt = __builtin_amdgcn_s_memrealtime(); }
    auto stats_out = [&](int wv, int ln) {
        if (STATS && ln == 0) {                              // o[2]: the wave's life on the constant 100 MHz clock, begin << 32 | end
            unsigned long long* o = stats + ((size_t)blockIdx.x * (NS + NPW + NP) + wv) * 4;
            o[0] = __builtin_readcyclecounter() - t_begin; o[1] = t_a;
            o[2] = (t_rt << 32) | (__builtin_amdgcn_s_memrealtime() & 0xFFFFFFFFull); o[3] = (n_items & 0xFFFFull) | (t_first << 16);   // (+ storers: when the first trip began; producers: three stamps of the start; 10 ns units)
        }
    };
    using Enc = OutEnc<OutT>;
    using B = typename L::B;
    constexpr int DD = L::DD, NQ = L::NQ, W = L::W;
    constexpr int VEC = 16 / (int)sizeof(OutT);              // elements per 16-byte lane store
    constexpr int EPC = 64 * VEC;                            // elements per chunk (one wave store instruction = 1 KiB)
    constexpr int LE = 128 / (int)sizeof(OutT);              // elements per 128-byte line
    constexpr int LPD = 32 / VEC;                            // lanes that share one ring dword
    constexpr uint32_t RING_BITS = 32u << RB_LOG, BMASK = (1u << RB_LOG) - 1u;
    constexpr uint32_t RP = 1u << RP_LOG, PMASK = RP - 1u;
    // WHOLE: a lattice's whole stack (2d^2 hits of 2d^2 bits each at most) fits into the ring beside what the storers may lag
    // behind: the producer asks for room once per lattice.  Otherwise (d >= 19) it asks pass by pass (64 hits) and publishes its
    // progress after every pass -- the consumers must be able to take the first passes of a lattice for the last ones to find
    // room.  Per-pass publishing costs a producer ~14 % (an LDS drain and a publish per pass: 5290 against 4640 cycles per d=7
    // lattice, profiles/r04_stream_tune_ab_passes.txt), so it is used only where it is needed.
    constexpr bool WHOLE = (uint32_t)NQ * NQ + 2u * NS * CPW * EPC + 4096u < RING_BITS;
    static_assert(64u * (uint32_t)NQ + 2u * NS * CPW * EPC + 4096u < RING_BITS, "bit ring too small for this lattice size");
    static_assert((uint32_t)NQ + 512u < RP, "position ring too small");
    static_assert(QS == 1 || (uint32_t)NQ + 63u <= (uint32_t)ProdTables<D, QS>::QN, "hit queue too small");
    static_assert(QS == 1 || (uint32_t)NQ * NQ + 2u * NS * CPW * EPC + 4096u < RING_BITS, "the hit queue needs lattices that fit the ring whole");
    __shared__ StreamLds<D, NS, NP, RB_LOG, RP_LOG, NPW, QS> S;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;

    // ---- this workgroup's range (wave-uniform).  The stack is cut into 1 << lg FINE parts of equal perspective count, RR of
    // them per workgroup on average (a power of two): cut points from the scan's table `split` (k_scan_final), or --
    // split == nullptr: a lattice sub-range, or offsets that did not come with the scan -- found here by waves 0 and 1.
    // The shares are NOT equal: the CUs of the odd XCDs of an MI355X store this stream ~20 % slower than those of the even
    // ones -- with equal shares the even XCDs' workgroups end at 0.80 of the launch, on every box and buffer measured
    // (profiles/r04_workgroup_end_times.txt).  So every pair of shares (2 RR fine parts) is cut into a LARGE slot of
    // RR + bias and a SMALL one of RR - bias, and a workgroup takes the next free large slot if it runs on an even XCD
    // (HW_REG_XCC_ID), the next free small one otherwise: two counters (and a third that tells the last workgroup to
    // zero them again), two atomics per workgroup.  Workgroups go to the
    // XCDs round-robin, but from where the dispatcher happens to stand (other streams' kernels move it:
    // tools/xcc_id_probe.hip), so blockIdx.x says nothing about the XCD; and whatever the dispatcher does, gridDim.x
    // workgroups take gridDim.x different slots -- if one kind runs out the other kind is taken.
    // Equal shares (bias = 0, slot = blockIdx.x) where the launch is not bound by the stores: small stacks, and the host
    // passes bias = 0 for d <= 5, whose launches are bound by the producers (d=5: 108 -> 121 us with 5 / 32,
    // profiles/r04_xcd_bias_sweep.txt).
    const int RR = (1 << lg) / (int)gridDim.x;
    // a small stack is not bound by the stores: equal shares, and no counters (their atomic's round trip across the XCDs is
    // ~3 us at the start of a launch: nothing beside 280 us, a fifth of a 15 us launch)
    if (!slots || bias >= RR || (offsets[e_end] - offsets[e_begin]) * (int64_t)(NQ * sizeof(OutT)) < (int64_t)(64 << 20)) bias = 0;
    const bool take = bias > 0;
    __shared__ int slot_s[2];
    if (threadIdx.x == 0) {
        int large = !(blockIdx.x & 1), idx = (int)(blockIdx.x >> 1);
        if (take) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            large = !(xcc & 1u);
            const unsigned half = gridDim.x >> 1;
            unsigned t = atomicAdd(&slots[large], 1u);
            if (t >= half) { large ^= 1; t = atomicAdd(&slots[large], 1u); }
            idx = t < half ? (int)t : -1;                     // (-1: the counters were not zero when the launch began)
            // (one counter per kind, 128 workgroups on each: spreading them over eight counters per kind was measured and
            // changes nothing -- the ~3 us this adds to a launch's start are the round trip of ONE device-scope atomic)
        }
        slot_s[0] = large; slot_s[1] = idx;
    }
    // (the rings and hand-off words meanwhile: they do not depend on the range)
    for (uint32_t i = threadIdx.x; i < (1u << RB_LOG) / 4; i += blockDim.x) reinterpret_cast<uint4*>(S.bits)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (threadIdx.x == 0) S.abort = 0u;
    if (threadIdx.x < NP) S.pq[threadIdx.x] = 0u;
    if (threadIdx.x < NPW) S.pcons[threadIdx.x] = 0u;
    if (threadIdx.x < NS) S.cons[threadIdx.x] = 0u;          // (a lower bound of "the first element storer s has not taken yet")
    __syncthreads();
    if (threadIdx.x == 64 * NS && take) {                    // (a positions wave: thread 0's wave stores the range's first window)
        // the last workgroup to have taken its slot leaves the counters zero for the next launch that uses them
        // (also a replay of this very launch from a captured graph): nobody else touches them any more.  Off the
        // critical path: only this thread's wave waits for the answer.
        __threadfence();
        if (atomicAdd(&slots[2], 1u) == gridDim.x - 1u) { slots[0] = 0u; slots[1] = 0u; __threadfence(); slots[2] = 0u; }
    }
    if (slot_s[1] < 0) {
        if (threadIdx.x == 0) atomicOr(err, ERR_INTERNAL);
        return;
    }
    const int f_lo = slot_s[1] * 2 * RR + (slot_s[0] ? 0 : RR + bias);
    const int f_hi = f_lo + (slot_s[0] ? RR + bias : RR - bias);
    int64_t e_lo, e_hi;
    if (split) {
        e_lo = split[f_lo]; e_hi = split[f_hi];
    } else {
        __shared__ int64_t cut[2];
        if (wave < 2) {
            const int64_t e = find_cut(offsets, e_begin, e_end, wave ? f_hi : f_lo, lg, lane);
            if (lane == 0) cut[wave] = e;
        }
        __syncthreads();
        e_lo = cut[0]; e_hi = cut[1];
    }
    e_lo = e_lo < e_begin ? e_begin : (e_lo > e_end ? e_end : e_lo);       // whatever the table holds, stay inside the range
    e_hi = e_hi < e_lo ? e_lo : (e_hi > e_end ? e_end : e_hi);
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
        e_lo = e_lo < e_stop ? e_lo : e_stop;
        e_hi = e_hi < e_stop ? e_hi : e_stop;
    }
    const int64_t Q0 = offsets[e_lo] - off0, Q1 = offsets[e_hi] - off0;        // perspective range [Q0, Q1)
    // The offsets come from the caller: whatever they hold, nothing is stored outside [0, p_all) perspectives.  Offsets
    // that are not monotone over this workgroup's cut points are refused here; offsets that do not match the lattices'
    // own hit counts are refused by the producer that meets the first such lattice (below).
    if (Q0 < 0 || Q1 < Q0 || Q1 > p_all) {
        if (threadIdx.x == 0) atomicOr(err, ERR_INTERNAL);
        return;
    }
    if (p_all == 0 && e_stop == e_end) {                     // "the stack is empty": true only if no lattice of the range has a hit
        for (int64_t e = e_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < e_end; e += (int64_t)gridDim.x * blockDim.x) {
            B v, pl;
#pragma unroll
            for (int k = 0; k < W; ++k) { v.w[k] = vp[(int64_t)k * N + e]; pl.w[k] = vp[((int64_t)W + k) * N + e]; }
            if (L::persp_count(v, pl) != 0) atomicOr(err, ERR_INTERNAL);
        }
        return;
    }
    const bool last = Q1 >= p_all;                           // no perspective behind this range
    const int64_t S0 = Q0 * NQ, S1 = Q1 * NQ;                // element range
    const int64_t org = S0 / LE * LE;                        // ring bit x <-> stack element org + x
    const uint32_t head = (uint32_t)(S0 - org);
    const uint32_t a0 = head ? (uint32_t)LE : 0u;            // first element (from org) this workgroup stores
    int64_t A1 = (S1 + LE - 1) / LE * LE;                    // the line that holds the end of the range is stored whole ...
    if (last || A1 > p_all * NQ) A1 = p_all * NQ;            // ... unless the stack ends inside it
    const uint32_t a1 = A1 > org ? (uint32_t)(A1 - org) : 0u;                  // one past the last
    // positions: dwords, 3 per perspective, lines of 32
    const int64_t porg = Q0 * 3 / 32 * 32;
    const uint32_t phead = (uint32_t)(Q0 * 3 - porg);
    const uint32_t pa0 = phead ? 32u : 0u;
    int64_t PA1 = (Q1 * 3 + 31) / 32 * 32;
    if (last || PA1 > p_all * 3) PA1 = p_all * 3;
    const uint32_t pa1 = PA1 > porg ? (uint32_t)(PA1 - porg) : 0u;
    // perspectives behind Q1 that the last stack line / positions line of this range needs
    int64_t need_extra = 0;
    if (!last) {
        const int64_t ne_s = (A1 - S1 + NQ - 1) / NQ, ne_p = (PA1 - Q1 * 3 + 2) / 3;
        need_extra = ne_s > ne_p ? ne_s : ne_p;
        if (!pos) need_extra = ne_s;
    }
    const bool has_stack = a1 > a0, has_pos = pos != nullptr && pa1 > pa0;
    if (!has_stack && !has_pos) return;                      // uniform over the workgroup

    // bounded wait: until pred(), false when the workgroup gave up
    auto give_up = [&]() {
        if (lane == 0) { __hip_atomic_store(&S.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); atomicOr(err, ERR_INTERNAL); }
    };

    // Perspectives [0, produced()) of the range are complete in the rings: lattices are dealt to the producers round-robin
    // and every producer works through its own in order, so every lattice that starts below the smallest `pq` is done.
    // One LDS read by NP lanes, the minimum on the scalar unit; wave-uniform.  0xFFFFFFFF = everything.
    auto produced = [&]() -> uint32_t {
        uint32_t v = 0xFFFFFFFFu;
        if (lane < NP) v = __hip_atomic_load(&S.pq[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t m = 0xFFFFFFFFu;
#pragma unroll
        for (int l = 0; l < NP; ++l) { const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)v, l); m = x < m ? x : m; }
        return m;
    };
    // stream bit position (from org) up to which the stack is produced, saturating
    auto produced_bits = [&](uint32_t pr) -> uint32_t { return pr == 0xFFFFFFFFu ? 0xFFFFFFFFu : head + pr * (uint32_t)NQ; };

    if (wave < NS) {
        // =========================================================== stack storer
        if (!has_stack) return;
        __builtin_amdgcn_s_setprio(3);                       // store issue goes before the producers' arithmetic
        const int s = wave;
        constexpr int U = CPW < 4 ? CPW : 4;                 // chunks per trip: one LDS round trip and one hand-back per U KiB
        static_assert(CPW % U == 0, "window must be a whole number of trips");
        const uint32_t nchunks = (a1 - a0 + EPC - 1) / EPC;
        const uint32_t lane_el = (uint32_t)lane * VEC;
        const int sh = (int)((a0 + lane_el) & 31u);          // a0 is a multiple of LE, EPC of 32: loop-invariant
        const bool zero_lane = (lane % LPD) == 0;
        char* __restrict__ obase = reinterpret_cast<char*>(out + org);
        uint32_t prod_c = 0;                                 // cached produced()
        bool first = true;
        for (uint32_t w = (uint32_t)s; w * CPW < nchunks; w += NS) {
            for (uint32_t cb = w * CPW; cb < (w + 1) * CPW && cb < nchunks; cb += U) {
                const uint32_t el0 = a0 + cb * EPC;
                uint32_t end = el0 + U * EPC;
                end = end < a1 ? end : a1;
                if (produced_bits(prod_c) < end) {           // wait until the trip's last element is produced
                    bool ok = false;
                    unsigned long long t0 = 0;
                    if (STATS) t0 = __builtin_readcyclecounter();
                    for (int spin = 0; spin < STREAM_SPIN_LIMIT; ++spin) {
                        prod_c = produced();
                        if (produced_bits(prod_c) >= end) { ok = true; break; }
                        if (lds_peek(S.abort)) return;
                        __builtin_amdgcn_s_sleep(4);
                    }
                    if (!ok) { give_up(); return; }
                    lds_after_peek();
                    if (STATS) t_a += __builtin_readcyclecounter() - t0;
                }
                if (STATS) { if (!n_items) t_first = __builtin_amdgcn_s_memrealtime() - t_rt; ++n_items; }
                if (first) {
                    first = false;
                    if (s == 0 && a0 && lane < (int)(a0 / 32u)) S.bits[lane] = 0u;   // ring words in front of a0 (stored by the previous range)
                }
                uint32_t wv[U], idx[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    idx[u] = ((el0 + (uint32_t)u * EPC + lane_el) >> 5) & BMASK;
                    wv[u] = S.bits[idx[u]];
                }
                if (zero_lane) {                             // hand the words back zeroed
#pragma unroll
                    for (int u = 0; u < U; ++u) S.bits[idx[u]] = 0u;
                }
                if (el0 + U * EPC <= a1) {                   // whole trip inside the range: U x 1 KiB
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        *reinterpret_cast<u32x4*>(obase + (size_t)(el0 + (uint32_t)u * EPC + lane_el) * sizeof(OutT)) = expand_bits<OutT>(wv[u] >> sh);
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint32_t el = el0 + (uint32_t)u * EPC + lane_el;
                        const u32x4 val = expand_bits<OutT>(wv[u] >> sh);
                        if (el + VEC <= a1) {
                            *reinterpret_cast<u32x4*>(obase + (size_t)el * sizeof(OutT)) = val;
                        } else if (el < a1) {                // the stack ends inside this lane's 16 bytes (last range only)
                            const int nel = (int)(a1 - el);
                            if (Enc::BITS == 32) {
                                for (int j = 0; j < nel; ++j) reinterpret_cast<uint32_t*>(obase)[el + j] = val[j];
                            } else if (Enc::BITS == 16) {
                                for (int j = 0; j < nel; ++j) reinterpret_cast<uint16_t*>(obase)[el + j] = (uint16_t)(val[j >> 1] >> (16 * (j & 1)));
                            } else {
                                for (int j = 0; j < nel; ++j) reinterpret_cast<uint8_t*>(obase)[el + j] = (uint8_t)(val[j >> 2] >> (8 * (j & 3)));
                            }
                        }
                    }
                }
                // first element this storer has not taken yet: the next trip of this window, or the next window of its own
                uint32_t nb = cb + U;
                if (nb % CPW == 0) nb += (uint32_t)(NS - 1) * CPW;
                lds_publish(S.cons[s], nb < nchunks ? a0 + nb * EPC : 0xFFFFFFFFu, lane);
            }
        }
        lds_publish(S.cons[s], 0xFFFFFFFFu, lane);
        stats_out(wave, lane);
        return;
    }

    if (wave < NS + NPW) {
        // =========================================================== positions storer (1 KiB chunks, round-robin over NPW waves)
        const int pw = wave - NS;
        if (!has_pos) return;
        int32_t* __restrict__ pbase = pos + porg;
        const uint32_t nchunks = (pa1 - pa0 + 255u) / 256u;
        for (uint32_t c = (uint32_t)pw; c < nchunks; c += NPW) {
            const uint32_t x0 = pa0 + c * 256u;
            const uint32_t x_end = x0 + 256u < pa1 ? x0 + 256u : pa1;
            const uint32_t need = (x_end - phead + 2u) / 3u;
            bool ok = false;
            unsigned long long t0 = 0;
            if (STATS) { t0 = __builtin_readcyclecounter(); ++n_items; }
            for (int spin = 0; spin < STREAM_SPIN_LIMIT; ++spin) {
                if (produced() >= need) { ok = true; break; }
                if (lds_peek(S.abort)) return;
                __builtin_amdgcn_s_sleep(NPW > 1 ? 8 : 16);
            }
            if (!ok) { give_up(); return; }
            lds_after_peek();
            if (STATS) t_a += __builtin_readcyclecounter() - t0;
            const uint32_t x = x0 + 4u * (uint32_t)lane;
            if (x < x_end) {
                int o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t t = x + j - phead, q = t / 3u;
                    o[j] = (int)((S.posr[q & PMASK] >> (8u * (t - 3u * q))) & 255u);
                }
                if (x + 4u <= x_end) *reinterpret_cast<int4*>(pbase + x) = make_int4(o[0], o[1], o[2], o[3]);
                else for (uint32_t j = 0; x + j < x_end; ++j) pbase[x + j] = o[j];
            }
            // low-water mark: the first perspective of this wave's NEXT chunk (everything below it, of this wave's, is written)
            const uint32_t nc = c + NPW;
            lds_publish(S.pcons[pw], nc < nchunks ? (pa0 + nc * 256u - phead) / 3u : 0xFFFFFFFFu, lane);
        }
        lds_publish(S.pcons[pw], 0xFFFFFFFFu, lane);
        stats_out(wave, lane);
        return;
    }

    // =============================================================== producer
    const int p = wave - NS - NPW;
    if (STATS) t_first = (__builtin_amdgcn_s_memrealtime() - t_rt) & 0xFFFFull;       // (producers: when the range was known, 10 ns units ...)
    ProdTables<D, QS>& PT = S.tab[p];
    if (lane < D) {                                          // column masks: the same for every lattice
        const B m = L::lowcols(lane);
#pragma unroll
        for (int w = 0; w < W; ++w) PT.low[lane][w] = m.w[w];
    }
    const int64_t QT = Q1 + need_extra;                      // lattices are produced while they start in front of QT
    uint32_t lw_c = a0, pc_c = 0u;                           // cached low-water marks of the storers
    // wait until the rings have room for the stream bits below `bits_end` and the positions below `q_end`; wave-uniform;
    // false = the workgroup gave up (the caller returns)
    auto room_now = [&](uint32_t bits_end, uint32_t q_end) __attribute__((always_inline)) -> bool {       // by the cached marks
        return (lw_c == 0xFFFFFFFFu || bits_end + 64u <= lw_c + RING_BITS) && (!has_pos || pc_c == 0xFFFFFFFFu || q_end <= pc_c + RP);
    };
    auto wait_room = [&](uint32_t bits_end, uint32_t q_end) __attribute__((always_inline)) -> bool {
        auto fits = [&]() { return room_now(bits_end, q_end); };
        if (fits()) return true;
        unsigned long long t0 = 0;
        if (STATS) t0 = __builtin_readcyclecounter();
        for (int spin = 0; spin < STREAM_SPIN_LIMIT; ++spin) {
            uint32_t c = 0xFFFFFFFFu;
            if (has_stack && lane < NS) c = __hip_atomic_load(&S.cons[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int o = 1; o < NS; o <<= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)c, o, 64); c = t < c ? t : c; }
            lw_c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
            if (has_pos) {
                uint32_t pc = 0xFFFFFFFFu;
                if (lane < NPW) pc = __hip_atomic_load(&S.pcons[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                for (int o = 1; o < NPW; o <<= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)pc, o, 64); pc = t < pc ? t : pc; }
                pc_c = (uint32_t)__builtin_amdgcn_readfirstlane((int)pc);
            }
            if (fits()) {
                lds_after_peek();
                if (STATS) t_a += __builtin_readcyclecounter() - t0;
                return true;
            }
            if (lds_peek(S.abort)) return false;
            __builtin_amdgcn_s_sleep(8);
        }
        give_up();
        return false;
    };
    // tables of one lattice: rotated planes (ballot), row-rolled planes, hit list (+ its positions, + its queue entries)
    auto build = [&](LatTables<D>& T, const B& v, const B& pl, const B& e0, const B& e1, int n0, uint32_t q0, uint32_t seq, uint32_t qtail) __attribute__((always_inline)) {
        B rv, rp;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const int o = 64 * k + lane;
            const bool inb = o < DD;
            const int oc = inb ? o : 0;
            rv.w[k] = __ballot(inb && v.get(PS::rot_src_v(oc)));
            rp.w[k] = __ballot(inb && pl.get(PS::rot_src_p(oc)));
        }
        for (int t = lane; t < 4 * D; t += 64) {             // one lane per (plane, row amount): 4 d entries (more than 64 from d = 17 on)
            const int sel = t / D, k = t - sel * D;
            B src;
#pragma unroll
            for (int w = 0; w < W; ++w) src.w[w] = sel == 0 ? v.w[w] : (sel == 1 ? pl.w[w] : (sel == 2 ? rv.w[w] : rp.w[w]));
            const B r = (src.shl(k * D) | src.shr(DD - k * D)) & L::full();
#pragma unroll
            for (int w = 0; w < W; ++w) T.rr[sel][k][w] = r.w[w];
        }
        for (int c = lane; c < NQ; c += 64) {
            const int l = c >= DD, bit = c - l * DD;
            if (l ? e1.get(bit) : e0.get(bit)) {
                const int row = bit / D, col = bit - row * D;
                const int k = l ? n0 + e1.rank(bit) : e0.rank(bit);
                const uint32_t hp = (uint32_t)l | ((uint32_t)row << 8) | ((uint32_t)col << 16);
                T.hpos[k] = hp;
                if (has_pos) S.posr[(q0 + (uint32_t)k) & PMASK] = hp;
                if (QS > 1) {
                    constexpr uint32_t QMASK = (uint32_t)ProdTables<D, QS>::QN - 1u;
                    PT.qent[(qtail + (uint32_t)k) & QMASK] = (seq << 16) | (uint32_t)k;
                    PT.qq[(qtail + (uint32_t)k) & QMASK] = q0 + (uint32_t)k;
                }
            }
        }
    };
    // one hit: its perspective as two bit-planes (two row-rolled planes of its lattice's table, two masked column rolls),
    // OR-ed into the ring at perspective q
    auto emit_hit = [&](const LatTables<D>& T, int k, uint32_t q) __attribute__((always_inline)) {
        const uint32_t hp = T.hpos[k];
        const int layer = (int)(hp & 255u), i = (int)((hp >> 8) & 255u), jj = (int)(hp >> 16);
        int rs, cs;
        PS::hit_shifts(layer, i, jj, rs, cs);                // (moving this to the hit-list stage, per qubit lane, was measured: no gain)
        B a, c, low;
#pragma unroll
        for (int w = 0; w < W; ++w) { a.w[w] = T.rr[2 * layer][rs][w]; c.w[w] = T.rr[2 * layer + 1][rs][w]; low.w[w] = PT.low[cs][w]; }
        const B ov = PS::roll_cols_masked(a, cs, low), op = PS::roll_cols_masked(c, cs, low);
        if (has_stack)
            emit_at<D>(head + q * (uint32_t)NQ, ov, op, [&](uint32_t idx, uint32_t val) { atomicOr(&S.bits[idx & BMASK], val); });
    };
    // ---- the hit queue (QS > 1): wave-uniform state, as little of it as possible (the kernel sits at the SGPR limit)
    uint32_t qh = 0, qn = 0;                                 // head of the queue, hits waiting in it
    uint32_t seq_new = 0, seq_head = 0;                      // lattices queued so far / the lattice of the hit at the head
    // the first cnt (<= 64) waiting hits, one lane each; then say how far this wave is: everything of ITS lattices below the
    // perspective after the last hit done is in the rings (its next hit is that very perspective or lies further on)
    auto emit_pass = [&](uint32_t cnt) __attribute__((always_inline)) {
        constexpr uint32_t QMASK = (uint32_t)ProdTables<D, QS>::QN - 1u;
        uint32_t q = 0;
        const uint32_t e_next = PT.qent[(qh + cnt) & QMASK]; // the hit that will be at the head afterwards (if any)
        if ((uint32_t)lane < cnt) {
            const uint32_t at = (qh + (uint32_t)lane) & QMASK;
            const uint32_t e = PT.qent[at];
            q = PT.qq[at];
            emit_hit(PT.t[QS > 1 ? ((e >> 16) % (uint32_t)QS) : 0], (int)(e & 0xFFFFu), q);
        }
        const uint32_t done = (uint32_t)__builtin_amdgcn_readlane((int)q, (int)cnt - 1) + 1u;
        lds_publish(S.pq[p], done, lane);
        qh += cnt; qn -= cnt;
        seq_head = qn ? ((uint32_t)__builtin_amdgcn_readfirstlane((int)e_next) >> 16) : seq_new;
    };
    auto flush = [&]() __attribute__((always_inline)) { while (qn) emit_pass(qn < 64u ? qn : 64u); };

    for (int64_t Lb = 0;; Lb += 64 * NP) {
        // the planes and offsets of this wave's next 64 lattices in one round of vector loads
        const int64_t e_l = e_lo + Lb + (int64_t)lane * NP + p;
        const bool in = e_l < e_stop;
        uint64_t vv[W], pp[W];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            vv[k] = in ? vp[(int64_t)k * N + e_l] : 0ull;
            pp[k] = in ? vp[((int64_t)W + k) * N + e_l] : 0ull;
        }
        const int64_t oo = in ? offsets[e_l] - off0 : (int64_t)0x7fffffffffffffffll;
        const int64_t oo1 = in ? offsets[e_l + 1] - off0 : (int64_t)0x7fffffffffffffffll;
        const uint64_t inmask = __ballot(in && oo < QT);     // offsets are monotone: a prefix of the lanes
        if (!inmask) break;
        const int cnt = __popcll(inmask);
        for (int j = 0; j < cnt; ++j) {
            B v, pl, e0, e1;
#pragma unroll
            for (int k = 0; k < W; ++k) { v.w[k] = readlane64(vv[k], j); pl.w[k] = readlane64(pp[k], j); }
            L::hit_masks(v, pl, e0, e1);
            const int n0 = e0.popc();
            const int n = n0 + e1.popc();
            // the offsets must be the scan of THESE lattices' hit counts; a table that is not (stale, shifted, from another
            // batch) is refused at the first lattice that disagrees, before anything of it reaches the rings
            if (readlane64((uint64_t)oo1, j) - readlane64((uint64_t)oo, j) != (uint64_t)n) { give_up(); return; }
            if (n == 0) continue;
            const uint32_t q0 = (uint32_t)((int64_t)readlane64((uint64_t)oo, j) - Q0);
            const uint32_t bit0 = head + q0 * (uint32_t)NQ;
            if (STATS) { if (!n_items) t_first |= ((__builtin_amdgcn_s_memrealtime() - t_rt) & 0xFFFFull) << 16; ++n_items; }   // (... when its first lattice was loaded ...)
            if (QS > 1) {
                // ---- hit queue: a table slot (the oldest lattices give theirs back as their hits get done; if all are taken
                // the waiting hits are done now, in a short pass), ring room for the whole lattice -- this wave WAITS only
                // with an empty queue: hits it holds back keep the consumers from the very room it would wait for
                if (((seq_new - seq_head) & 0xFFFFu) >= (uint32_t)QS) flush();   // (16-bit lattice numbers in the queue entries)
                if (!room_now(bit0 + (uint32_t)n * NQ, q0 + (uint32_t)n)) {
                    flush();
                    if (!wait_room(bit0 + (uint32_t)n * NQ, q0 + (uint32_t)n)) return;
                }
                if (qn == 0) lds_publish(S.pq[p], q0, lane);    // nothing older waits: this lattice is how far the wave is
                const uint32_t slot = seq_new % (uint32_t)QS;
                if (qn == 0) seq_head = seq_new;
                build(PT.t[slot], v, pl, e0, e1, n0, q0, seq_new, qh + qn);
                seq_new = (seq_new + 1u) & 0xFFFFu;
                qn += (uint32_t)n;
                wave_lds_sync();
                while (qn >= 64u) emit_pass(64u);
                wave_lds_sync();                             // a slot given back is rewritten by a later lattice
                continue;
            }
            // ---- one lattice at a time.  This wave's earlier lattices are in the rings (its LDS operations execute in issue
            // order): say so
            lds_publish(S.pq[p], q0, lane);
            // room in the rings: everything below the storers' low-water mark has been handed back.  The marks are cached:
            // while the storers keep up the ring is nearly empty and one look lasts for dozens of lattices.  Positions: the
            // whole lattice (n <= 2d^2 < ring - 512); bits: the whole lattice, or (d >= 19) its first pass of 64 hits.
            const uint32_t pass1 = (uint32_t)((WHOLE || n < 64) ? n : 64);
            if (!wait_room(bit0 + pass1 * (uint32_t)NQ, q0 + (uint32_t)n)) return;
            LatTables<D>& T = PT.t[0];
            build(T, v, pl, e0, e1, n0, q0, 0u, 0u);
            wave_lds_sync();
            // one lane per hit, 64 hits per pass.  d >= 19 (!WHOLE): after every pass the wave says how far the lattice is (the
            // consumers may take it) and asks for the next pass's room -- 2d^2 hits x 2d^2 bits are more than the ring holds
            for (int kb = 0; kb < n; kb += 64) {
                if (!WHOLE && kb) {
                    lds_publish(S.pq[p], q0 + (uint32_t)kb, lane);
                    const uint32_t upto = (uint32_t)(n < kb + 64 ? n : kb + 64);
                    if (!wait_room(bit0 + upto * (uint32_t)NQ, q0 + (uint32_t)n)) return;
                }
                const int k = kb + lane;
                if (k >= n) continue;
                emit_hit(T, k, q0 + (uint32_t)k);
            }
            wave_lds_sync();                                 // T is rewritten by the next lattice
            if (STATS && n_items == 1 && !(t_first >> 32)) t_first |= ((__builtin_amdgcn_s_memrealtime() - t_rt) & 0xFFFFull) << 32;   // (... and in the ring)
        }
        if (cnt < 64) break;
    }
    if (QS > 1) flush();
    lds_publish(S.pq[p], 0xFFFFFFFFu, lane);                 // no lattice left: everything of this wave is in the rings
    stats_out(wave, lane);
}

}  // namespace tq
