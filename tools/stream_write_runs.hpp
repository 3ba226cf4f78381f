// HARNESS ONLY (tools/stream_tune.hip): an EXPERIMENT, measured and not adopted -- the stack write in which a workgroup
// writes several runs, the ones behind its main run taken by ticket (dynamic load balancing).  Same bytes as the library's
// kernel; no faster on slow buffers and 3-12 % slower on fast ones (the runs cost ~2.7 % by themselves, and many short
// streams next to each other are slower than 256 long ones): profiles/r04_runs_by_ticket_ab.txt.  What it was built
// for -- the workgroups of odd XCDs store ~20 % slower than those of even XCDs -- is now met by unequal fixed shares
// (csrc/stream_write.hpp).  Not part of the library.
//
// Perspective stack write, producer / storer form (gfx950 / CDNA4).
//
// One persistent workgroup per CU writes CONTIGUOUS pieces of the output ("runs": lattices [e_lo, e_hi) between two cut
// points of the scan, k_scan_final / find_cut).  Inside the workgroup the two jobs of the stack write are done by
// different waves:
//   * NP producer waves build lattice bitstreams (lattice.hpp, PStream: rotated planes by ballot, table of
//     row-rolled planes, one lane per hit, ds_or_b32) -- not into a per-wave buffer but into ONE ring in LDS
//     that is the workgroup's output as a bit string (the workgroup's STREAM: its runs one after the other);
//   * NS storer waves do nothing but  ds_read_b32 -> shift -> bit->element expansion -> global_store_dwordx4
//     along a run, in aligned windows of CPW KiB, and hand the ring words back zeroed;
//   * NPW positions waves write the positions (P,3) from a second ring (one packed dword per perspective).
// Hand-off: `pq[p]` (producer p: how far it is -- the stream position of the lattice it is working on, published per
// lattice (d >= 19: after every pass of 64 hits); its earlier lattices are complete; the minimum over the producers is
// the produced PREFIX of the stream, no producer ever waits for another), `cons[s]` (low-water mark of storer s),
// `pcons[w]` (of positions wave w): plain LDS words, polled with s_sleep.  Every poll loop is bounded; a wave that gives
// up raises `abort` for its workgroup and latches ERR_INTERNAL, so the grid always drains.
//
// Which runs a workgroup writes.  The CUs of an MI355X do not store at one rate: with equal shares the workgroups of one
// launch end between 0.78 and 1.0 of its duration, XCD by XCD and buffer by buffer (profiles/r04_workgroup_end_times.txt).
// So only a MAIN run (5/8 of an equal share) is fixed per workgroup; the rest of the stack is cut into ever smaller runs
// that the workgroups TAKE as they get there (one global ticket counter).  All waves of a workgroup go through the same
// sequence of runs on their own; the rings never drain between two runs -- the producers are in the next run while the
// storers finish the last one.  Small stacks, lattice sub-ranges and offsets without the scan's table: one run per
// workgroup, no tickets.
//
// Output lines: the stack is cut into 128-byte lines and a run stores the lines whose FIRST element lies in it, whole.
// The trailing elements of its last line belong to the first lattice(s) behind the run: its producers simply go on for
// the few perspectives that line needs (`need_extra`); what they produce outside the run's own lines is not put into
// the ring at all.
#pragma once
#include "kernels.hpp"

namespace tqr {
using namespace tq;

constexpr int ERR_INTERNAL = 32;
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
struct LatTables {                                             // of ONE lattice, private to one producer wave
    static constexpr int NQP = (Lat<D>::NQ + 7) & ~7;
    uint64_t rr[4][D][Lat<D>::W];                              // V, P, rot V, rot P rolled by every row amount
    uint32_t hpos[NQP];                                        // k-th hit -> layer | row << 8 | col << 16
    uint64_t low[D][Lat<D>::W];                                // lowcols(k): the same for every lattice
};

constexpr int STREAM_SEQ_MAX = 128;                            // runs one workgroup can take in one launch

template <int D, int NS, int NP, int RB_LOG, int RP_LOG, int NPW = 1>
struct StreamLds {
    __attribute__((aligned(16))) uint32_t bits[1u << RB_LOG];  // the workgroup's stream as a bit string, ring
    uint32_t posr[1u << RP_LOG];                               // packed position of perspective y of the stream at [y & mask]
    LatTables<D> tab[NP];
    uint32_t pq[NP];                                           // producer p: stream position (bits) of the lattice it is working on;
                                                               // everything of ITS lattices below that is in the rings;
                                                               // 0xFFFFFFFF = it has no lattice left
    uint32_t cons[NS];                                         // storer s: stream position (bits) below which it has taken everything of its own
    uint32_t pcons[NPW];                                       // positions wave w: the same in perspectives of the stream
    uint32_t seq[STREAM_SEQ_MAX];                              // j-th run of this workgroup: 0 = not asked for yet, 1 = being asked for,
                                                               // ticket + 2, 0xFFFFFFFF = there is none
    int64_t cut[2];                                            // one run per workgroup without the scan's table: its cut points
    uint32_t abort;
};

// How the stack is handed out.  The scan's table cuts it into G = gridDim.x * R fine parts of equal perspective count; in
// units of u = R / 32 fine parts a workgroup's main run is MAIN u long, and the tickets that follow are worth SZ0 u
// (the first TK0 * gridDim.x of them), SZ1 u, ...: MAIN + sum(SZ * TK) = 32.
template <int SCHED> struct StreamSched;
template <> struct StreamSched<0> { static constexpr uint32_t MAIN = 20, SZ0 = 4, TK0 = 1, SZ1 = 2, TK1 = 2, SZ2 = 1, TK2 = 4; };
template <> struct StreamSched<1> { static constexpr uint32_t MAIN = 24, SZ0 = 4, TK0 = 1, SZ1 = 2, TK1 = 1, SZ2 = 1, TK2 = 2; };
template <> struct StreamSched<2> { static constexpr uint32_t MAIN = 16, SZ0 = 8, TK0 = 1, SZ1 = 2, TK1 = 2, SZ2 = 1, TK2 = 4; };
template <> struct StreamSched<3> { static constexpr uint32_t MAIN = 26, SZ0 = 2, TK0 = 1, SZ1 = 1, TK1 = 2, SZ2 = 1, TK2 = 2; };
// (diagnostic) 4: the pieces of schedule 0, but every workgroup takes those of its OWN share, in order: what the runs cost by themselves
template <> struct StreamSched<4> { static constexpr uint32_t MAIN = 20, SZ0 = 4, TK0 = 1, SZ1 = 2, TK1 = 2, SZ2 = 1, TK2 = 4; };
constexpr int STREAM_DYN_LG = 13;                              // fine parts of the scan's table: 8192 = 32 per workgroup of 256
constexpr uint64_t STREAM_DYN_MIN_BYTES = 256ull << 20;        // smaller stacks: one run per workgroup

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
// NPW: positions waves (chunks of 1 KiB dealt round-robin among them)
// split / lg: the scan's table of (1 << lg) + 1 cut points, or nullptr (every workgroup finds the two cut points of its one run)
// ticket / ticket_clear: the counter the runs behind the main runs are taken from (nullptr: one run per workgroup) and the
// counter of a LATER launch, zeroed here (the host rotates through a few: a launch that gives up leaves its counter dirty)
template <int D, typename OutT, int NS, int NP, int CPW, int RB_LOG, int RP_LOG, bool STATS = false, int NPW = 1, int SCHED = 0>
__global__ __launch_bounds__(64 * (NS + NPW + NP)) void k_persp_stream(const uint64_t* __restrict__ vp, int64_t N,
                                                                  const int64_t* __restrict__ offsets, OutT* __restrict__ out,
                                                                  int32_t* __restrict__ pos, int64_t capacity,
                                                                  int* __restrict__ err, int64_t e_begin, int64_t e_end,
                                                                  const int32_t* __restrict__ split, int lg,
                                                                  unsigned int* __restrict__ ticket, unsigned int* __restrict__ ticket_clear,
                                                                  unsigned long long* __restrict__ stats = nullptr) {
    using L = Lat<D>;
    using PS = PStream<D>;
    using SC = StreamSched<SCHED>;
    static_assert(SC::MAIN + SC::SZ0 * SC::TK0 + SC::SZ1 * SC::TK1 + SC::SZ2 * SC::TK2 == 32u, "the schedule must cover the stack");
    unsigned long long t_begin = 0, t_a = 0, t_rt = 0, n_items = 0;
    if (STATS) { t_begin = __builtin_readcyclecounter(); t_rt = __builtin_amdgcn_s_memrealtime(); }
    auto stats_out = [&](int wv, int ln) {
        if (STATS && ln == 0) {
            unsigned long long* o = stats + ((size_t)blockIdx.x * (NS + NPW + NP) + wv) * 4;
            o[0] = __builtin_readcyclecounter() - t_begin; o[1] = t_a;
            o[2] = (t_rt << 32) | (__builtin_amdgcn_s_memrealtime() & 0xFFFFFFFFull); o[3] = n_items;
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
    constexpr uint32_t WINBITS = (uint32_t)CPW * EPC;        // a window in elements = stream bits
    constexpr uint32_t XPAD = 12u;                           // perspectives behind a run its producers may start (need_extra <= 11)
    // between two runs of the stream: bits nobody stores (the extra perspectives' places, rounding to whole windows, one
    // window more so that a storer's last trip never reaches into the next run's bits)
    constexpr uint32_t GAP_BITS = XPAD * NQ + 2u * WINBITS, GAP_Q = XPAD + NQ;
    // WHOLE: a lattice's whole stack (2d^2 hits of 2d^2 bits each at most) fits into the ring beside what the storers may lag
    // behind: the producer asks for room once per lattice.  Otherwise (d >= 19) it asks pass by pass (64 hits) and publishes its
    // progress after every pass -- the consumers must be able to take the first passes of a lattice for the last ones to find
    // room.  Per-pass publishing costs a producer ~14 % (profiles/r04_stream_tune_ab_passes.txt), so it is used only where needed.
    constexpr bool WHOLE = (uint32_t)NQ * NQ + 2u * NS * WINBITS + GAP_BITS + 4096u < RING_BITS;
    static_assert(64u * (uint32_t)NQ + 2u * NS * WINBITS + GAP_BITS + 4096u < RING_BITS, "bit ring too small for this lattice size");
    static_assert((uint32_t)NQ + GAP_Q + 512u < RP, "position ring too small");
    __shared__ StreamLds<D, NS, NP, RB_LOG, RP_LOG, NPW> S;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;

    // ---- the whole stack (wave-uniform)
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
    if (blockIdx.x == 0 && threadIdx.x == 0 && ticket_clear) *ticket_clear = 0u;
    if (p_all == 0 && e_stop == e_end) {                     // "the stack is empty": true only if no lattice of the range has a hit
        for (int64_t e = e_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < e_end; e += (int64_t)gridDim.x * blockDim.x) {
            B v, pl;
#pragma unroll
            for (int k = 0; k < W; ++k) { v.w[k] = vp[(int64_t)k * N + e]; pl.w[k] = vp[((int64_t)W + k) * N + e]; }
            if (L::persp_count(v, pl) != 0) atomicOr(err, ERR_INTERNAL);
        }
        return;
    }
    // fine parts per workgroup share; runs are taken by ticket only from the scan's own table, and only for stacks
    // worth it (a ticket costs an atomic and a handful of dependent scalar loads per wave)
    const uint32_t RR = split ? (1u << lg) / gridDim.x : 1u;
    const bool dyn = ticket != nullptr && split != nullptr && RR >= 32u && (RR & 31u) == 0u &&
                     (uint64_t)p_all * (uint64_t)(NQ * sizeof(OutT)) >= STREAM_DYN_MIN_BYTES;

    // ---- rings and hand-off words
    for (uint32_t i = threadIdx.x; i < (1u << RB_LOG) / 4; i += blockDim.x) reinterpret_cast<uint4*>(S.bits)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (threadIdx.x == 0) S.abort = 0u;
    if (threadIdx.x < NP) S.pq[threadIdx.x] = 0u;
    if (threadIdx.x < NPW) S.pcons[threadIdx.x] = 0u;
    if (threadIdx.x < NS) S.cons[threadIdx.x] = 0u;
    if (threadIdx.x < STREAM_SEQ_MAX) S.seq[threadIdx.x] = 0u;
    if (!split && wave < 2) {                                // no table: the two cut points of this workgroup's one run (gridDim.x a power of two)
        const int64_t e = find_cut(offsets, e_begin, e_end, (int)blockIdx.x + wave, 31 - __clz((int)gridDim.x), lane);
        if (lane == 0) S.cut[wave] = e;
    }
    __syncthreads();

    auto give_up = [&]() {
        if (lane == 0) { __hip_atomic_store(&S.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); atomicOr(err, ERR_INTERNAL); }
    };

    // ---- the runs of this workgroup, one after the other; every wave walks through them by itself (wave-uniform state)
    int64_t r_elo = 0, r_Q0 = 0, r_QT = 0, r_org = 0, r_porg = 0;
    uint32_t r_head = 0, r_a0 = 0, r_a1 = 0, r_phead = 0, r_pa0 = 0, r_pa1 = 0;
    uint32_t r_X = 0, r_Y = 0, r_wb = 0, r_cb = 0;           // where the run lies in the stream: bits, perspectives, windows, position chunks
    bool r_stack = false, r_pos = false;
    uint32_t n_X = 0, n_Y = 0, n_wb = 0, n_cb = 0;           // the same of the run behind it
    // the j-th run: 1 = opened, 0 = there is none, -1 = the workgroup gave up
    auto open_run = [&](uint32_t j) __attribute__((always_inline)) -> int {
        int64_t e_lo, e_hi;
        if (j == 0) {
            if (!split) { e_lo = S.cut[0]; e_hi = S.cut[1]; }
            else {
                const uint32_t per = dyn ? SC::MAIN * (RR >> 5) : RR;
                if (SCHED == 4 && dyn) { e_lo = split[blockIdx.x * RR]; e_hi = split[blockIdx.x * RR + per]; }
                else { e_lo = split[blockIdx.x * per]; e_hi = split[(blockIdx.x + 1u) * per]; }
            }
        } else {
            if (!dyn || j >= (uint32_t)STREAM_SEQ_MAX || n_X > 0xC0000000u) return 0;
            if (SCHED == 4) {
                if (j > 7u) return 0;
                const uint32_t u = RR >> 5;
                const uint32_t o = j == 1u ? 20u : (j == 2u ? 24u : (j == 3u ? 26u : 24u + j)), len = j == 1u ? 4u : (j < 4u ? 2u : 1u);
                e_lo = split[blockIdx.x * RR + o * u]; e_hi = split[blockIdx.x * RR + (o + len) * u];
            } else {
            uint32_t v = 0;
            if (lane == 0) {
                v = __hip_atomic_load(&S.seq[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (v == 0u) {
                    v = atomicCAS(&S.seq[j], 0u, 1u);
                    if (v == 0u) {                           // this wave asks for the workgroup
                        const uint32_t t = atomicAdd(ticket, 1u);
                        const uint32_t NT = (SC::TK0 + SC::TK1 + SC::TK2) * gridDim.x;
                        v = t < NT ? t + 2u : 0xFFFFFFFFu;
                        __hip_atomic_store(&S.seq[j], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                for (int spin = 0; v == 1u && spin < STREAM_SPIN_LIMIT; ++spin) {
                    __builtin_amdgcn_s_sleep(2);
                    v = __hip_atomic_load(&S.seq[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
            if (v == 1u) { give_up(); return -1; }
            if (v == 0xFFFFFFFFu) return 0;
            uint32_t t = v - 2u;
            const uint32_t u = RR >> 5, G = gridDim.x;
            uint32_t f0 = SC::MAIN * u * G, len;
            if (t < SC::TK0 * G) { len = SC::SZ0 * u; f0 += t * len; }
            else {
                f0 += SC::TK0 * G * SC::SZ0 * u; t -= SC::TK0 * G;
                if (t < SC::TK1 * G) { len = SC::SZ1 * u; f0 += t * len; }
                else { f0 += SC::TK1 * G * SC::SZ1 * u; t -= SC::TK1 * G; len = SC::SZ2 * u; f0 += t * len; }
            }
            e_lo = split[f0]; e_hi = split[f0 + len];
            }
        }
        e_lo = e_lo < e_begin ? e_begin : (e_lo > e_stop ? e_stop : e_lo);     // whatever the table holds, stay inside the range
        e_hi = e_hi < e_lo ? e_lo : (e_hi > e_stop ? e_stop : e_hi);
        const int64_t Q0 = offsets[e_lo] - off0, Q1 = offsets[e_hi] - off0;    // perspective range [Q0, Q1)
        // The offsets come from the caller: whatever they hold, nothing is stored outside [0, p_all) perspectives.  Offsets
        // that are not monotone over the cut points are refused here; offsets that do not match the lattices' own hit
        // counts are refused by the producer that meets the first such lattice (below).
        if (Q0 < 0 || Q1 < Q0 || Q1 > p_all || Q1 - Q0 > 0x7FFFFFFF / NQ) { give_up(); return -1; }
        const bool last = Q1 >= p_all;                       // no perspective behind this run
        const int64_t S0 = Q0 * NQ, S1 = Q1 * NQ;            // element range
        const int64_t org = S0 / LE * LE;                    // stream bit X + x <-> stack element org + x
        const uint32_t head = (uint32_t)(S0 - org);
        const uint32_t a0 = head ? (uint32_t)LE : 0u;        // first element (from org) this run stores
        int64_t A1 = (S1 + LE - 1) / LE * LE;                // the line that holds the end of the run is stored whole ...
        if (last || A1 > p_all * NQ) A1 = p_all * NQ;        // ... unless the stack ends inside it
        const uint32_t a1 = A1 > org ? (uint32_t)(A1 - org) : 0u;              // one past the last
        // positions: dwords, 3 per perspective, lines of 32
        const int64_t porg = Q0 * 3 / 32 * 32;
        const uint32_t phead = (uint32_t)(Q0 * 3 - porg);
        const uint32_t pa0 = phead ? 32u : 0u;
        int64_t PA1 = (Q1 * 3 + 31) / 32 * 32;
        if (last || PA1 > p_all * 3) PA1 = p_all * 3;
        const uint32_t pa1 = PA1 > porg ? (uint32_t)(PA1 - porg) : 0u;
        // perspectives behind Q1 that the last stack line / positions line of this run needs
        int64_t need_extra = 0;
        if (!last) {
            const int64_t ne_s = (A1 - S1 + NQ - 1) / NQ, ne_p = (PA1 - Q1 * 3 + 2) / 3;
            need_extra = ne_s > ne_p ? ne_s : ne_p;
            if (!pos) need_extra = ne_s;
        }
        r_elo = e_lo; r_Q0 = Q0; r_QT = Q1 + need_extra; r_org = org; r_porg = porg;
        r_head = head; r_a0 = a0; r_a1 = a1; r_phead = phead; r_pa0 = pa0; r_pa1 = pa1;
        r_stack = a1 > a0; r_pos = pos != nullptr && pa1 > pa0;
        r_X = n_X; r_Y = n_Y; r_wb = n_wb; r_cb = n_cb;
        const uint32_t nq = (uint32_t)(Q1 - Q0);
        n_X = (r_X + head + nq * (uint32_t)NQ + GAP_BITS) / WINBITS * WINBITS;
        n_Y = r_Y + nq + GAP_Q;
        n_wb = r_wb + (r_stack ? ((a1 - a0 + (uint32_t)EPC - 1u) / (uint32_t)EPC + (uint32_t)CPW - 1u) / (uint32_t)CPW : 0u);
        n_cb = r_cb + (r_pos ? (pa1 - pa0 + 255u) / 256u : 0u);
        return 1;
    };

    // The stream is complete in the rings below produced(): lattices are dealt to the producers round-robin and every
    // producer works through its own in stream order, so every lattice that starts below the smallest `pq` is done.
    // One LDS read by NP lanes, the minimum on the scalar unit; wave-uniform.  0xFFFFFFFF = everything.
    auto produced = [&]() -> uint32_t {
        uint32_t v = 0xFFFFFFFFu;
        if (lane < NP) v = __hip_atomic_load(&S.pq[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t m = 0xFFFFFFFFu;
#pragma unroll
        for (int l = 0; l < NP; ++l) { const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)v, l); m = x < m ? x : m; }
        return m;
    };

    if (wave < NS) {
        // =========================================================== stack storer
        __builtin_amdgcn_s_setprio(3);                       // store issue goes before the producers' arithmetic
        const int s = wave;
        constexpr int U = CPW < 4 ? CPW : 4;                 // chunks per trip: one LDS round trip and one hand-back per U KiB
        static_assert(CPW % U == 0, "window must be a whole number of trips");
        const uint32_t lane_el = (uint32_t)lane * VEC;
        const bool zero_lane = (lane % LPD) == 0;
        uint32_t prod_c = 0;                                 // cached produced()
        for (uint32_t j = 0;; ++j) {
            const int got = open_run(j);
            if (got < 0) return;
            if (got == 0) break;
            if (r_stack) {
                const uint32_t nchunks = (r_a1 - r_a0 + EPC - 1) / EPC;
                const uint32_t nwin = (nchunks + CPW - 1) / CPW;
                const int sh = (int)((r_a0 + lane_el) & 31u);    // X and a0 are multiples of 32, EPC too: invariant over the run
                char* __restrict__ obase = reinterpret_cast<char*>(out + r_org);
                // windows are dealt round-robin along the STREAM: this storer's first one in this run
                for (uint32_t w = ((uint32_t)s + (uint32_t)NS - r_wb % (uint32_t)NS) % (uint32_t)NS; w < nwin; w += NS) {
                    for (uint32_t cb = w * CPW; cb < (w + 1) * CPW && cb < nchunks; cb += U) {
                        const uint32_t el0 = r_a0 + cb * EPC;
                        uint32_t end = el0 + U * EPC;
                        end = end < r_a1 ? end : r_a1;
                        if (prod_c < r_X + end) {            // wait until the trip's last element is produced
                            bool ok = false;
                            unsigned long long t0 = 0;
                            if (STATS) t0 = __builtin_readcyclecounter();
                            for (int spin = 0; spin < STREAM_SPIN_LIMIT; ++spin) {
                                prod_c = produced();
                                if (prod_c >= r_X + end) { ok = true; break; }
                                if (lds_peek(S.abort)) return;
                                __builtin_amdgcn_s_sleep(4);
                            }
                            if (!ok) { give_up(); return; }
                            lds_after_peek();
                            if (STATS) t_a += __builtin_readcyclecounter() - t0;
                        }
                        if (STATS) ++n_items;
                        uint32_t wv[U], idx[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            idx[u] = ((r_X + el0 + (uint32_t)u * EPC + lane_el) >> 5) & BMASK;
                            wv[u] = S.bits[idx[u]];
                        }
                        if (zero_lane) {                     // hand the words back zeroed
#pragma unroll
                            for (int u = 0; u < U; ++u) S.bits[idx[u]] = 0u;
                        }
                        if (el0 + U * EPC <= r_a1) {         // whole trip inside the run: U x 1 KiB
#pragma unroll
                            for (int u = 0; u < U; ++u)
                                *reinterpret_cast<u32x4*>(obase + (size_t)(el0 + (uint32_t)u * EPC + lane_el) * sizeof(OutT)) = expand_bits<OutT>(wv[u] >> sh);
                        } else {
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                const uint32_t el = el0 + (uint32_t)u * EPC + lane_el;
                                const u32x4 val = expand_bits<OutT>(wv[u] >> sh);
                                if (el + VEC <= r_a1) {
                                    *reinterpret_cast<u32x4*>(obase + (size_t)el * sizeof(OutT)) = val;
                                } else if (el < r_a1) {      // the stack ends inside this lane's 16 bytes (last run only)
                                    const int nel = (int)(r_a1 - el);
                                    if (Enc::BITS == 32) {
                                        for (int jj = 0; jj < nel; ++jj) reinterpret_cast<uint32_t*>(obase)[el + jj] = val[jj];
                                    } else if (Enc::BITS == 16) {
                                        for (int jj = 0; jj < nel; ++jj) reinterpret_cast<uint16_t*>(obase)[el + jj] = (uint16_t)(val[jj >> 1] >> (16 * (jj & 1)));
                                    } else {
                                        for (int jj = 0; jj < nel; ++jj) reinterpret_cast<uint8_t*>(obase)[el + jj] = (uint8_t)(val[jj >> 2] >> (8 * (jj & 3)));
                                    }
                                }
                            }
                        }
                        // what this storer has not taken yet begins at: the next trip of this window, its next window of
                        // this run, or (at the earliest) the run behind
                        uint32_t nb = cb + U;
                        if (nb % CPW == 0) nb += (uint32_t)(NS - 1) * CPW;
                        lds_publish(S.cons[s], nb < nchunks ? r_X + r_a0 + nb * EPC : n_X, lane);
                    }
                }
            }
            lds_publish(S.cons[s], n_X, lane);               // (a run in which this storer had nothing to do)
        }
        lds_publish(S.cons[s], 0xFFFFFFFFu, lane);
        stats_out(wave, lane);
        return;
    }

    if (wave < NS + NPW) {
        // =========================================================== positions storer (1 KiB chunks, round-robin over NPW waves)
        const int pw = wave - NS;
        for (uint32_t j = 0;; ++j) {
            const int got = open_run(j);
            if (got < 0) return;
            if (got == 0) break;
            if (r_pos) {
                int32_t* __restrict__ pbase = pos + r_porg;
                const uint32_t nchunks = (r_pa1 - r_pa0 + 255u) / 256u;
                for (uint32_t c = ((uint32_t)pw + (uint32_t)NPW - r_cb % (uint32_t)NPW) % (uint32_t)NPW; c < nchunks; c += NPW) {
                    const uint32_t x0 = r_pa0 + c * 256u;
                    const uint32_t x_end = x0 + 256u < r_pa1 ? x0 + 256u : r_pa1;
                    // perspectives [0, need) of the run: in the rings once the stream is produced up to where `need` would start
                    const uint32_t need = (x_end - r_phead + 2u) / 3u;
                    const uint32_t need_x = r_X + r_head + need * (uint32_t)NQ;
                    bool ok = false;
                    unsigned long long t0 = 0;
                    if (STATS) { t0 = __builtin_readcyclecounter(); ++n_items; }
                    for (int spin = 0; spin < STREAM_SPIN_LIMIT; ++spin) {
                        if (produced() >= need_x) { ok = true; break; }
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
                        for (int jj = 0; jj < 4; ++jj) {
                            const uint32_t t = x + jj - r_phead, q = t / 3u;
                            o[jj] = (int)((S.posr[(r_Y + q) & PMASK] >> (8u * (t - 3u * q))) & 255u);
                        }
                        if (x + 4u <= x_end) *reinterpret_cast<int4*>(pbase + x) = make_int4(o[0], o[1], o[2], o[3]);
                        else for (uint32_t jj = 0; x + jj < x_end; ++jj) pbase[x + jj] = o[jj];
                    }
                    // low-water mark: the first perspective of this wave's NEXT chunk (everything below it, of this wave's, is written)
                    const uint32_t nc = c + NPW;
                    lds_publish(S.pcons[pw], nc < nchunks ? r_Y + (r_pa0 + nc * 256u - r_phead) / 3u : n_Y, lane);
                }
            }
            lds_publish(S.pcons[pw], n_Y, lane);
        }
        lds_publish(S.pcons[pw], 0xFFFFFFFFu, lane);
        stats_out(wave, lane);
        return;
    }

    // =============================================================== producer
    const int p = wave - NS - NPW;
    LatTables<D>& T = S.tab[p];
    if (lane < D) {                                          // column masks: the same for every lattice
        const B m = L::lowcols(lane);
#pragma unroll
        for (int w = 0; w < W; ++w) T.low[lane][w] = m.w[w];
    }
    const bool use_pos = pos != nullptr;
    uint32_t lw_c = 0u, pc_c = 0u;                           // cached low-water marks of the storers
    // wait until the rings have room for the stream bits below `bits_end` and the positions below `q_end`; wave-uniform;
    // false = the workgroup gave up (the caller returns)
    auto wait_room = [&](uint32_t bits_end, uint32_t q_end) __attribute__((always_inline)) -> bool {
        auto fits = [&]() { return (lw_c == 0xFFFFFFFFu || bits_end + 64u <= lw_c + RING_BITS) && (!use_pos || pc_c == 0xFFFFFFFFu || q_end <= pc_c + RP); };
        if (fits()) return true;
        unsigned long long t0 = 0;
        if (STATS) t0 = __builtin_readcyclecounter();
        for (int spin = 0; spin < STREAM_SPIN_LIMIT; ++spin) {
            uint32_t c = 0xFFFFFFFFu;
            if (lane < NS) c = __hip_atomic_load(&S.cons[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int o = 1; o < NS; o <<= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)c, o, 64); c = t < c ? t : c; }
            lw_c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
            if (use_pos) {
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

    for (uint32_t j = 0;; ++j) {
        const int got = open_run(j);
        if (got < 0) return;
        if (got == 0) break;
        // ring words this run may set: those of its own lines [a0, a1) -- the head of the first perspective (stored by the run
        // in front) and the extra perspectives' bits behind the last line never reach the ring
        const uint32_t w_lo = (r_X + r_a0) >> 5, w_n = r_stack ? ((r_X + r_a1 + 31u) >> 5) - w_lo : 0u;
        for (int64_t Lb = 0;; Lb += 64 * NP) {
            // the planes and offsets of this wave's next 64 lattices in one round of vector loads
            const int64_t e_l = r_elo + Lb + (int64_t)lane * NP + p;
            const bool in = e_l < e_stop;
            uint64_t vv[W], pp[W];
#pragma unroll
            for (int k = 0; k < W; ++k) {
                vv[k] = in ? vp[(int64_t)k * N + e_l] : 0ull;
                pp[k] = in ? vp[((int64_t)W + k) * N + e_l] : 0ull;
            }
            const int64_t oo = in ? offsets[e_l] - off0 : (int64_t)0x7fffffffffffffffll;
            const int64_t oo1 = in ? offsets[e_l + 1] - off0 : (int64_t)0x7fffffffffffffffll;
            const uint64_t inmask = __ballot(in && oo < r_QT);   // offsets are monotone: a prefix of the lanes
            if (!inmask) break;
            const int cnt = __popcll(inmask);
            for (int jl = 0; jl < cnt; ++jl) {
                B v, pl, e0, e1;
#pragma unroll
                for (int k = 0; k < W; ++k) { v.w[k] = readlane64(vv[k], jl); pl.w[k] = readlane64(pp[k], jl); }
                L::hit_masks(v, pl, e0, e1);
                const int n0 = e0.popc();
                const int n = n0 + e1.popc();
                // the offsets must be the scan of THESE lattices' hit counts; a table that is not (stale, shifted, from another
                // batch) is refused at the first lattice that disagrees, before anything of it reaches the rings
                if (readlane64((uint64_t)oo1, jl) - readlane64((uint64_t)oo, jl) != (uint64_t)n) { give_up(); return; }
                if (n == 0) continue;
                const int64_t q0l = (int64_t)readlane64((uint64_t)oo, jl) - r_Q0;
                if (q0l < 0 || q0l > 0x7FFFFFFF / NQ) { give_up(); return; }      // (offsets that are not monotone inside the run)
                const uint32_t q0 = (uint32_t)q0l;
                const uint32_t bit0 = r_X + r_head + q0 * (uint32_t)NQ, y0 = r_Y + q0;
                if (STATS) ++n_items;
                // this wave's earlier lattices are in the rings (its LDS operations execute in issue order): say so
                lds_publish(S.pq[p], bit0, lane);
                // room in the rings: everything below the storers' low-water mark has been handed back.  The marks are cached:
                // while the storers keep up the ring is nearly empty and one look lasts for dozens of lattices.  Positions: the
                // whole lattice (n <= 2d^2 < ring - 512); bits: the whole lattice, or (d >= 19) its first pass of 64 hits.
                const uint32_t pass1 = (uint32_t)((WHOLE || n < 64) ? n : 64);
                if (!wait_room(bit0 + pass1 * (uint32_t)NQ, y0 + (uint32_t)n)) return;
                // tables of the lattice: rotated planes (ballot), row-rolled planes, hit list (+ its positions)
                {
                    B rv, rp;
#pragma unroll
                    for (int k = 0; k < W; ++k) {
                        const int o = 64 * k + lane;
                        const bool inb = o < DD;
                        const int oc = inb ? o : 0;
                        rv.w[k] = __ballot(inb && v.get(PS::rot_src_v(oc)));
                        rp.w[k] = __ballot(inb && pl.get(PS::rot_src_p(oc)));
                    }
                    for (int t = lane; t < 4 * D; t += 64) {  // one lane per (plane, row amount): 4 d entries (more than 64 from d = 17 on)
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
                            if (use_pos) S.posr[(y0 + (uint32_t)k) & PMASK] = hp;
                        }
                    }
                }
                wave_lds_sync();
                // one lane per hit, 64 hits per pass.  d >= 19 (!WHOLE): after every pass the wave says how far the lattice is (the
                // consumers may take it) and asks for the next pass's room -- 2d^2 hits x 2d^2 bits are more than the ring holds
                for (int kb = 0; kb < n; kb += 64) {
                    if (!WHOLE && kb) {
                        lds_publish(S.pq[p], bit0 + (uint32_t)kb * (uint32_t)NQ, lane);
                        const uint32_t upto = (uint32_t)(n < kb + 64 ? n : kb + 64);
                        if (!wait_room(bit0 + upto * (uint32_t)NQ, y0 + (uint32_t)n)) return;
                    }
                    const int k = kb + lane;
                    if (k >= n) continue;
                    // the hit's perspective as two bit-planes: two row-rolled planes of the table, two masked column rolls
                    const uint32_t hp = T.hpos[k];
                    const int layer = (int)(hp & 255u), i = (int)((hp >> 8) & 255u), jj = (int)(hp >> 16);
                    int rs, cs;
                    PS::hit_shifts(layer, i, jj, rs, cs);    // (moving this to the hit-list stage, per qubit lane, was measured: no gain)
                    B a, c, low;
#pragma unroll
                    for (int w = 0; w < W; ++w) { a.w[w] = T.rr[2 * layer][rs][w]; c.w[w] = T.rr[2 * layer + 1][rs][w]; low.w[w] = T.low[cs][w]; }
                    const B ov = PS::roll_cols_masked(a, cs, low), op = PS::roll_cols_masked(c, cs, low);
                    tqr::emit_at<D>(bit0 + (uint32_t)k * (uint32_t)NQ, ov, op,
                               [&](uint32_t idx, uint32_t val) { if (idx - w_lo < w_n) atomicOr(&S.bits[idx & BMASK], val); });
                }
                wave_lds_sync();                             // T is rewritten by the next lattice
            }
            if (cnt < 64) break;
        }
        lds_publish(S.pq[p], n_X, lane);                     // everything of this wave up to the run behind is in the rings
    }
    lds_publish(S.pq[p], 0xFFFFFFFFu, lane);                 // no lattice left
    stats_out(wave, lane);
}

}  // namespace tqr
