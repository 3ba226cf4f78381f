// HIP kernels of the batched toric-code environment (gfx950 / CDNA4).
//
// Data layout in HBM (per handle, N lattices, W = ceil(d*d/64)):
//   planes  u64[6][W][N]   bit-planes X0 X1 Z0 Z1 V P, structure-of-arrays over lattices, so the
//                          thread-per-lattice kernels load/store 8 B per lane fully coalesced
//   prev    u64[2][W][N]   V,P before the last tq_step (for tq_transition_write)
//   counts  i32[N], offsets i64[N+1], episodes/steps u32[N], p_roof f64[N]
//
// Two kernel shapes:
//   * env dynamics (reset / step / transition / fused actor step): ONE THREAD PER LATTICE.
//     In bit-plane form a whole-lattice syndrome is ~20 shifts/xors, so there is nothing to
//     share between lanes; 65 536 lattices = 1 024 wavefronts.  (Exception: the few resets inside
//     the fused step are served by the whole wave, one qubit per lane, planes built by __ballot.)
//   * perspective stack write (the HBM-bound kernel): stream_write.hpp -- one persistent workgroup per CU,
//     producer waves build lattice bitstreams into an LDS ring with bit-plane operations (periodic shifts =
//     plane rolls, the layer-1 rotation = the same rolls on the once-rotated planes; one lane per hit), storer
//     waves expand the ring: every lane stores 16 B, one wave instruction = 1 KiB of the stack.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "lattice.hpp"

namespace tq {

enum { PL_X0 = 0, PL_X1 = 1, PL_Z0 = 2, PL_Z1 = 3, PL_V = 4, PL_P = 5 };
enum { ERR_ACTION = 1, ERR_CAPACITY = 2, ERR_RESET_DUP = 4, ERR_RESET_ROUNDS = 8, ERR_INDEX = 16 };

template <int W>
__device__ __forceinline__ Bits<W> load_plane(const uint64_t* __restrict__ planes, int plane, int64_t N, int64_t e) {
    Bits<W> b;
#pragma unroll
    for (int k = 0; k < W; ++k) b.w[k] = planes[((int64_t)plane * W + k) * N + e];
    return b;
}
template <int W>
__device__ __forceinline__ void store_plane(uint64_t* __restrict__ planes, int plane, int64_t N, int64_t e, const Bits<W>& b) {
#pragma unroll
    for (int k = 0; k < W; ++k) planes[((int64_t)plane * W + k) * N + e] = b.w[k];
}
template <int D>
__device__ __forceinline__ typename Lat<D>::State load_state(const uint64_t* __restrict__ planes, int64_t N, int64_t e) {
    constexpr int W = Lat<D>::W;
    typename Lat<D>::State s;
    s.x[0] = load_plane<W>(planes, PL_X0, N, e);
    s.x[1] = load_plane<W>(planes, PL_X1, N, e);
    s.z[0] = load_plane<W>(planes, PL_Z0, N, e);
    s.z[1] = load_plane<W>(planes, PL_Z1, N, e);
    s.v = load_plane<W>(planes, PL_V, N, e);
    s.p = load_plane<W>(planes, PL_P, N, e);
    return s;
}
template <int D>
__device__ __forceinline__ void store_state(uint64_t* __restrict__ planes, int64_t N, int64_t e, const typename Lat<D>::State& s) {
    constexpr int W = Lat<D>::W;
    store_plane<W>(planes, PL_X0, N, e, s.x[0]);
    store_plane<W>(planes, PL_X1, N, e, s.x[1]);
    store_plane<W>(planes, PL_Z0, N, e, s.z[0]);
    store_plane<W>(planes, PL_Z1, N, e, s.z[1]);
    store_plane<W>(planes, PL_V, N, e, s.v);
    store_plane<W>(planes, PL_P, N, e, s.p);
}

// Position of the k-th set bit (k < popc) of the concatenated hit mask [E0 | E1] as a flat
// qubit index layer*DD + row*D + col -- the k-th entry of the reference's positions list.
template <int D>
__device__ __forceinline__ int kth_hit(const typename Lat<D>::B& e0, const typename Lat<D>::B& e1, int k) {
    constexpr int W = Lat<D>::W;
    constexpr int DD = Lat<D>::DD;
    int base = 0;
    uint64_t word = 0;
    bool found = false;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const uint64_t wv = l ? e1.w[j] : e0.w[j];
            const int c = popc64(wv);
            const bool here = !found && k < c;
            word = here ? wv : word;
            base = here ? l * DD + 64 * j : base;
            k = (found || here) ? k : k - c;
            found = found || here;
        }
    }
    // position of the k-th set bit of `word` (k < popc(word)): binary search on popcounts, six steps,
    // no data-dependent loop (a wave pays for its slowest lane)
    uint32_t w32 = (uint32_t)word;
    int posn = 0;
    {
        const int c = __popc(w32);
        const bool up = k >= c;
        k = up ? k - c : k; posn = up ? 32 : 0; w32 = up ? (uint32_t)(word >> 32) : w32;
    }
#pragma unroll
    for (int s = 16; s >= 1; s >>= 1) {
        const int c = __popc(w32 & ((1u << s) - 1u));
        const bool up = k >= c;
        k = up ? k - c : k; posn += up ? s : 0; w32 = up ? (w32 >> s) : w32;
    }
    return base + posn;
}

struct PerrSchedule {
    int strategy;          // TQ_PERR_*
    double p_default, p_start, p_final, p_delta;
};

// First pass of the exclusive scan, folded into the kernels that produce the hit counts: every
// 256-thread block of an all-lattice kernel leaves the sum of its 256 counts in part256[blockIdx.x].
// All 256 threads must call it (threads past N pass 0).
constexpr int PART_BLOCK = 256;
__device__ __forceinline__ void block_count_partial(int my_count, int64_t* __restrict__ part256) {
    __shared__ int ws_[4];
    int s = my_count;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) ws_[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part256[blockIdx.x] = (int64_t)ws_[0] + ws_[1] + ws_[2] + ws_[3];
}

// ------------------------------------------------------------------ reset
// env.reset(p_error) per lattice (EnvSet.py:19-36).  idx == nullptr: all lattices.
// Indexed mode: an index outside [0,N) latches ERR_INDEX; an index listed twice latches
// ERR_RESET_DUP (mark[e] holds the epoch of the last indexed reset that touched lattice e, so the
// second thread of a pair sees its own epoch) and only the first thread resets the lattice.  A
// lattice whose syndrome is still empty after MAX_RESET_ROUNDS rounds (p_error ~ 0) latches
// ERR_RESET_ROUNDS.
template <int D>
__global__ __launch_bounds__(256) void k_reset(uint64_t* __restrict__ planes, uint32_t* __restrict__ episodes,
                                               uint32_t* __restrict__ steps, int32_t* __restrict__ counts,
                                               const int32_t* __restrict__ idx, int n_idx,
                                               const double* __restrict__ p_err, double p_default,
                                               uint64_t seed, int64_t first_env, int64_t N,
                                               int64_t* __restrict__ part256, uint32_t* __restrict__ mark,
                                               uint32_t epoch, int min_err, int* __restrict__ err) {
    using L = Lat<D>;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t m = idx ? n_idx : N;
    int64_t e = -1;
    if (t < m) e = idx ? idx[t] : t;
    if (idx && t < m) {
        if (e < 0 || e >= N) { atomicOr(err, ERR_INDEX); e = -1; }
        else if (atomicExch(&mark[e], epoch) == epoch) { atomicOr(err, ERR_RESET_DUP); e = -1; }
    }
    int cnt = 0;
    if (e >= 0 && e < N) {
        typename L::State s;
        const uint32_t ep = episodes[e];
        if (min_err > 0) reset_lattice_n<D>(s, seed, (uint32_t)(first_env + e), ep, min_err);   // config "min_qubit_errors"
        else reset_lattice<D>(s, seed, (uint32_t)(first_env + e), ep, p_err ? p_err[t] : p_default);
        if (!(s.v.any() || s.p.any())) atomicOr(err, ERR_RESET_ROUNDS);
        store_state<D>(planes, N, e, s);
        episodes[e] = ep + 1;
        steps[e] = 0;
        cnt = L::persp_count(s.v, s.p);
        counts[e] = cnt;
    }
    if (part256) block_count_partial(cnt, part256);          // all-lattice mode only (uniform branch)
}

// validated decode of action = [layer,row,col,op]
template <int D>
__device__ __forceinline__ bool action_ok(int layer, int row, int col, int op) {
    return ((unsigned)layer < 2u) & ((unsigned)row < (unsigned)D) & ((unsigned)col < (unsigned)D) &
           ((unsigned)(op - 1) < 3u);
}
// op == 0 is "no action" (what tq_select_action emits for a lattice without defects): the step is
// counted, nothing changes, and no error is latched.
__device__ __forceinline__ bool action_noop(int op) { return op == 0; }

// ------------------------------------------------------------------ step (EnvSet.step)
template <int D>
__global__ __launch_bounds__(256) void k_step(uint64_t* __restrict__ planes, uint64_t* __restrict__ prev,
                                              const int32_t* __restrict__ actions, float* __restrict__ rewards,
                                              uint8_t* __restrict__ terminals, uint32_t* __restrict__ steps,
                                              int32_t* __restrict__ counts, float terminal_reward, int64_t N,
                                              int* __restrict__ err, int64_t* __restrict__ part256) {
    using L = Lat<D>;
    constexpr int W = L::W;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int cnt = 0;
    if (e < N) {
        typename L::State s = load_state<D>(planes, N, e);
        const int4 a = reinterpret_cast<const int4*>(actions)[e];
        store_plane<W>(prev, 0, N, e, s.v);
        store_plane<W>(prev, 1, N, e, s.p);
        const int before = s.v.popc() + s.p.popc();
        if (action_ok<D>(a.x, a.y, a.z, a.w)) L::apply(s, a.x, a.y, a.z, a.w);
        else if (!action_noop(a.w)) atomicOr(err, ERR_ACTION);
        L::syndrome(s);
        const int after = s.v.popc() + s.p.popc();
        store_state<D>(planes, N, e, s);
        if (rewards) rewards[e] = after == 0 ? terminal_reward : (float)(before - after);
        if (terminals) terminals[e] = after == 0;
        steps[e] += 1;
        cnt = L::persp_count(s.v, s.p);
        counts[e] = cnt;
    }
    block_count_partial(cnt, part256);
}

// ------------------------------------------------------------------ packed transition block
struct BlockView {     // SoA sections of a packed transition block (see include/toricenv.h)
    uint64_t* pv; uint64_t* pp; uint64_t* nv; uint64_t* np;
    uint32_t* action; float* reward; float* priority; uint8_t* terminal;
    int64_t cap;
};
__host__ __device__ inline int64_t align8(int64_t x) { return (x + 7) & ~(int64_t)7; }
__host__ __device__ inline BlockView block_view(void* base, int W, int64_t cap) {
    BlockView b;
    char* p = (char*)base;
    b.cap = cap;
    b.pv = (uint64_t*)p; p += 8 * (int64_t)W * cap;
    b.pp = (uint64_t*)p; p += 8 * (int64_t)W * cap;
    b.nv = (uint64_t*)p; p += 8 * (int64_t)W * cap;
    b.np = (uint64_t*)p; p += 8 * (int64_t)W * cap;
    b.action = (uint32_t*)p; p += align8(4 * cap);
    b.reward = (float*)p; p += align8(4 * cap);
    b.priority = (float*)p; p += align8(4 * cap);
    b.terminal = (uint8_t*)p;
    return b;
}
__host__ __device__ inline int64_t block_bytes(int W, int64_t cap) {
    return 4 * 8 * (int64_t)W * cap + 3 * align8(4 * cap) + align8(cap);
}

// The four checks of the acted qubit in ITS OWN centred frame -- v[gs,gs], v[gs+1,gs], p[gs,gs], p[gs,gs-1], for
// either layer (centred-frame property) -- so the perspective of the post-step syndrome is the
// perspective of the pre-step syndrome with these bits flipped: Z component -> the two vertices, X component
// -> the two plaquettes.  (perspective() is linear over GF(2).)
template <int D>
__device__ __forceinline__ void centred_flip(int op, typename Lat<D>::B& dv, typename Lat<D>::B& dp) {
    using L = Lat<D>;
    constexpr int GS = L::GS;
    dv = L::B::zero(); dp = L::B::zero();
    const int fx = (op == 1) | (op == 2), fz = (op >> 1) & 1;
    dv.flip(GS * D + GS, fz); dv.flip((GS + 1) * D + GS, fz);
    dp.flip(GS * D + GS, fx); dp.flip(GS * D + GS - 1, fx);
}

template <int D>
__device__ __forceinline__ void write_transition(const BlockView& b, int64_t slot, const typename Lat<D>::B& v0,
                                                 const typename Lat<D>::B& p0, const typename Lat<D>::B& v1,
                                                 const typename Lat<D>::B& p1, int layer, int row, int col, int op,
                                                 float reward, int terminal, bool stepped = false) {
    using L = Lat<D>;
    constexpr int W = L::W;
    typename L::B a, c;
    L::perspective(v0, p0, layer, row, col, a, c);
#pragma unroll
    for (int k = 0; k < W; ++k) { b.pv[(int64_t)k * b.cap + slot] = a.w[k]; b.pp[(int64_t)k * b.cap + slot] = c.w[k]; }
    if (stepped) {                                            // (v1,p1) = (v0,p0) after `op` on this very qubit
        typename L::B dv, dp;
        centred_flip<D>(op, dv, dp);
        a = a ^ dv; c = c ^ dp;
    } else {
        L::perspective(v1, p1, layer, row, col, a, c);
    }
#pragma unroll
    for (int k = 0; k < W; ++k) { b.nv[(int64_t)k * b.cap + slot] = a.w[k]; b.np[(int64_t)k * b.cap + slot] = c.w[k]; }
    // action rewritten to the centred frame (util_actor.py:256,261)
    b.action[slot] = (uint32_t)layer | ((uint32_t)L::GS << 8) | ((uint32_t)L::GS << 16) | ((uint32_t)op << 24);
    b.reward[slot] = reward;
    b.terminal[slot] = (uint8_t)terminal;
}

// A slot without a transition (no-op or rejected action): action word 0 (op = 0 marks the slot
// invalid for tq_transition_unpack / wire.decode), everything else zero -- never stale data.
template <int D>
__device__ __forceinline__ void write_empty_slot(const BlockView& b, int64_t slot) {
    constexpr int W = Lat<D>::W;
#pragma unroll
    for (int k = 0; k < W; ++k) {
        b.pv[(int64_t)k * b.cap + slot] = 0; b.pp[(int64_t)k * b.cap + slot] = 0;
        b.nv[(int64_t)k * b.cap + slot] = 0; b.np[(int64_t)k * b.cap + slot] = 0;
    }
    b.action[slot] = 0u;
    b.reward[slot] = 0.f;
    b.terminal[slot] = 0;
}

// generateTransitionParallel for the last tq_step, into a packed block
template <int D>
__global__ __launch_bounds__(256) void k_transition(const uint64_t* __restrict__ planes, const uint64_t* __restrict__ prev,
                                                    const int32_t* __restrict__ actions, BlockView b, int64_t slot_base,
                                                    int64_t N, int* __restrict__ err) {
    using L = Lat<D>;
    constexpr int W = L::W;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const int4 a = reinterpret_cast<const int4*>(actions)[e];
    if (!action_ok<D>(a.x, a.y, a.z, a.w)) {
        if (!action_noop(a.w)) atomicOr(err, ERR_ACTION);
        write_empty_slot<D>(b, slot_base + e);
        return;
    }
    const auto v0 = load_plane<W>(prev, 0, N, e), p0 = load_plane<W>(prev, 1, N, e);
    const auto v1 = load_plane<W>(planes, PL_V, N, e), p1 = load_plane<W>(planes, PL_P, N, e);
    write_transition<D>(b, slot_base + e, v0, p0, v1, p1, a.x, a.y, a.z, a.w, 0.f, 0);
}

// env.reset for ONE lattice by a whole wavefront: lane c draws qubit c (same Philox counters as
// reset_lattice, so bit-identical) and __ballot assembles the 64-bit plane words directly.  All 64
// lanes must be active; env/episode/p are wave-uniform; the result is the same in every lane.
template <int D>
__device__ __forceinline__ void reset_lattice_wave(typename Lat<D>::State& s, uint64_t seed, uint32_t env,
                                                   uint32_t episode, double p, int lane) {
    using L = Lat<D>;
    for (int r = 0; r < MAX_RESET_ROUNDS; ++r) {
#pragma unroll
        for (int l = 0; l < 2; ++l) {
#pragma unroll
            for (int k = 0; k < L::W; ++k) {
                const int c = 64 * k + lane;
                const U4 w = draw(seed, env, episode, (uint32_t)r, DOMAIN_ERR, (uint32_t)(l * L::DD + c));
                const bool err = (c < L::DD) && (u01(w.x) < p);
                const uint32_t pauli = 1 + mulhi32(w.y, 3);
                s.x[l].w[k] = __ballot(err && ((pauli == 1) | (pauli == 2)));
                s.z[l].w[k] = __ballot(err && (pauli >> 1));
            }
        }
        L::syndrome(s);
        if (s.v.any() || s.p.any()) return;
    }
}

// ------------------------------------------------------------------ fused actor step
// Actor_mp.py:116-183 after the policy: step -> transition -> reset(terminal | too many steps)
// -> perspective counts.  actions == nullptr: pure exploration (eps = 1) drawn in-kernel.
// The lattices are read from `planes_in` and written to `planes` (the handle's two plane buffers take turns): every
// plane of every lattice is stored, so the output buffer is complete, and the input buffer stays what the stack
// write of this step reads.
template <int D>
__global__ __launch_bounds__(256) void k_actor_step(const uint64_t* __restrict__ planes_in, uint64_t* __restrict__ planes,
                                                    uint32_t* __restrict__ episodes,
                                                    uint32_t* __restrict__ steps, int32_t* __restrict__ counts,
                                                    double* __restrict__ p_roof, const int32_t* __restrict__ actions,
                                                    int32_t* __restrict__ actions_out, float* __restrict__ rewards,
                                                    uint8_t* __restrict__ terminals, BlockView blk, int has_block,
                                                    int64_t slot_base, PerrSchedule sched, float terminal_reward,
                                                    int max_steps, int min_err, uint64_t seed, int64_t first_env, int64_t N,
                                                    int* __restrict__ err, int64_t* __restrict__ part256) {
    using L = Lat<D>;
    // every lane stays alive to the end (the reset below is wave-cooperative); lanes past N work on a
    // clamped copy of the last lattice and store nothing
    const int64_t e_raw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = e_raw < N;
    const int64_t e = valid ? e_raw : N - 1;
    const int lane = threadIdx.x & 63;
    typename L::State s = load_state<D>(planes_in, N, e);   // read from one buffer, written to the other (below): a stack write
    uint32_t ep = episodes[e], st = steps[e];               // of the pre-step lattices may run beside this kernel
    const uint32_t env = (uint32_t)(first_env + e);
    int layer, row, col, op;
    bool ok;
    if (actions) {
        const int4 a = reinterpret_cast<const int4*>(actions)[e];
        layer = a.x; row = a.y; col = a.z; op = a.w;
        ok = action_ok<D>(layer, row, col, op);
        if (!ok && !action_noop(op) && valid) atomicOr(err, ERR_ACTION);
    } else {
        // non-greedy branch of _selectActionBatch_prime (numba/util_actor.py:97-98)
        typename L::B e0, e1;
        L::hit_masks(s.v, s.p, e0, e1);
        const int n = e0.popc() + e1.popc();
        ok = n > 0;
        layer = row = col = op = 0;
        if (ok) {
            const U4 w = draw(seed, env, ep, st, DOMAIN_SEL, 0);
            const int h = kth_hit<D>(e0, e1, (int)mulhi32(w.y, (uint32_t)n));
            layer = h >= L::DD;
            const int rem = h - layer * L::DD;
            row = rem / D; col = rem - row * D;
            op = 1 + (int)mulhi32(w.z, 3);
        }
    }
    if (actions_out && valid) reinterpret_cast<int4*>(actions_out)[e] = make_int4(layer, row, col, op);
    const typename L::B v0 = s.v, p0 = s.p;
    const int before = v0.popc() + p0.popc();
    if (ok) L::apply(s, layer, row, col, op);
    L::syndrome(s);
    const int after = s.v.popc() + s.p.popc();
    const int terminal = after == 0;
    const float reward = terminal ? terminal_reward : (float)(before - after);
    st += 1;
    if (rewards && valid) rewards[e] = reward;
    if (terminals && valid) terminals[e] = (uint8_t)terminal;
    if (has_block && valid) {                                // every slot is written, every step: no stale records
        if (ok) write_transition<D>(blk, slot_base + e, v0, p0, s.v, s.p, layer, row, col, op, reward, terminal, true);
        else write_empty_slot<D>(blk, slot_base + e);
    }
    // reset policy of the caller (Actor_mp.py:171-183).  Few lanes of a wave reset in a given step, so
    // the lanes that do are served one after another by the whole wave (98 Philox draws in two passes
    // instead of a 98-iteration loop in one lane while 63 wait).
    const bool need_reset = valid && (terminal || st > (uint32_t)max_steps);
    double p = sched.p_default;
    if (need_reset) {
        if (sched.strategy != 0) {
            double roof = p_roof[e] + sched.p_delta;
            roof = roof < sched.p_final ? roof : sched.p_final;
            p_roof[e] = roof;
            p = roof;
            if (sched.strategy == 2) {
                const U4 w = draw(seed, env, ep, 0, DOMAIN_PERR, 0);
                const double span = roof - sched.p_start;
                const double t = span * u01(w.x);
                p = sched.p_start + t;
            }
        }
    }
    uint64_t pending = __ballot(need_reset);
    while (pending) {                                        // wave-uniform
        const int src = (int)__ffsll((long long)pending) - 1;
        pending &= pending - 1;
        typename L::State fresh;
        if (min_err > 0) {                                   // fixed-n sampler: sequential by nature, done by the lane itself
            if (lane == src) reset_lattice_n<D>(fresh, seed, env, ep, min_err);
        } else {
            reset_lattice_wave<D>(fresh, seed, (uint32_t)__shfl((int)env, src, 64), (uint32_t)__shfl((int)ep, src, 64),
                                  __shfl(p, src, 64), lane);
        }
        if (lane == src) {
            if (!(fresh.v.any() || fresh.p.any())) atomicOr(err, ERR_RESET_ROUNDS);
            s = fresh; ep += 1; st = 0;
        }
    }
    const int cnt = valid ? L::persp_count(s.v, s.p) : 0;
    if (valid) {
        store_state<D>(planes, N, e, s);
        episodes[e] = ep;
        steps[e] = st;
        counts[e] = cnt;
    }
    block_count_partial(cnt, part256);
}

// ------------------------------------------------------------------ u8 views
// syndrome planes -> u8[n,2,d,d]; one thread per output byte, coalesced byte stores.
// idx == nullptr: rows 0..n-1 of the handle, else rows idx[0..n).
template <int D>
__global__ __launch_bounds__(256) void k_get_state(const uint64_t* __restrict__ planes, int64_t N,
                                                   const int32_t* __restrict__ idx, int64_t n,
                                                   uint8_t* __restrict__ out) {
    using L = Lat<D>;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * L::NQ) return;
    const int64_t row = t / L::NQ;
    const int c = (int)(t - row * L::NQ);
    const int64_t e = idx ? idx[row] : row;
    if (e < 0 || e >= N) { out[t] = 0; return; }
    const int plane = c < L::DD ? PL_V : PL_P;
    const int bit = c < L::DD ? c : c - L::DD;
    out[t] = (uint8_t)((planes[((int64_t)plane * L::W + (bit >> 6)) * N + e] >> (bit & 63)) & 1);
}
template <int D>
__global__ __launch_bounds__(256) void k_get_qubits(const uint64_t* __restrict__ planes, int64_t N,
                                                    uint8_t* __restrict__ out) {
    using L = Lat<D>;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * L::NQ) return;
    const int64_t e = t / L::NQ;
    const int c = (int)(t - e * L::NQ);
    const int layer = c >= L::DD;
    const int bit = c - layer * L::DD;
    const int x = (int)((planes[((int64_t)(PL_X0 + layer) * L::W + (bit >> 6)) * N + e] >> (bit & 63)) & 1);
    const int z = (int)((planes[((int64_t)(PL_Z0 + layer) * L::W + (bit >> 6)) * N + e] >> (bit & 63)) & 1);
    out[t] = (uint8_t)(z ? (x ? 2 : 3) : x);
}
// u8 Pauli codes -> planes (+ syndrome, counts); thread per lattice.
template <int D>
__global__ __launch_bounds__(256) void k_set_qubits(uint64_t* __restrict__ planes, int32_t* __restrict__ counts,
                                                    const uint8_t* __restrict__ q, int64_t N,
                                                    int64_t* __restrict__ part256) {
    using L = Lat<D>;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int cnt = 0;
    if (e < N) {
        typename L::State s;
#pragma unroll
        for (int l = 0; l < 2; ++l) {
#pragma unroll
            for (int k = 0; k < L::W; ++k) {
                uint64_t ax = 0, az = 0;
                const int nb = (L::DD - 64 * k) < 64 ? (L::DD - 64 * k) : 64;
                for (int bit = 0; bit < nb; ++bit) {
                    const int code = q[e * L::NQ + l * L::DD + 64 * k + bit];
                    ax |= (uint64_t)((code == 1) | (code == 2)) << bit;
                    az |= (uint64_t)((code >> 1) & 1) << bit;
                }
                s.x[l].w[k] = ax; s.z[l].w[k] = az;
            }
        }
        L::syndrome(s);
        store_state<D>(planes, N, e, s);
        cnt = L::persp_count(s.v, s.p);
        counts[e] = cnt;
    }
    block_count_partial(cnt, part256);
}
// u8[n,2,d,d] syndromes -> V/P planes u64[2][W][n] (+ counts); for states outside a handle.
template <int D>
__global__ __launch_bounds__(256) void k_pack_states(const uint8_t* __restrict__ st, uint64_t* __restrict__ vp,
                                                     int32_t* __restrict__ counts, int64_t n) {
    using L = Lat<D>;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    typename L::B b[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
#pragma unroll
        for (int k = 0; k < L::W; ++k) {
            uint64_t a = 0;
            const int nb = (L::DD - 64 * k) < 64 ? (L::DD - 64 * k) : 64;
            for (int bit = 0; bit < nb; ++bit) a |= (uint64_t)(st[e * L::NQ + l * L::DD + 64 * k + bit] != 0) << bit;
            b[l].w[k] = a;
            vp[((int64_t)l * L::W + k) * n + e] = a;
        }
    }
    if (counts) counts[e] = L::persp_count(b[0], b[1]);
}
template <int D>
__global__ __launch_bounds__(256) void k_flags(const uint64_t* __restrict__ planes, int64_t N,
                                               uint8_t* __restrict__ ground, uint8_t* __restrict__ terminal) {
    using L = Lat<D>;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const typename L::State s = load_state<D>(planes, N, e);
    if (ground) ground[e] = (uint8_t)L::ground_state(s);
    if (terminal) terminal[e] = (uint8_t)!(s.v.any() || s.p.any());
}

// packed block slots -> u8 grids etc.; thread per output byte of one grid pair
template <int D>
__global__ __launch_bounds__(256) void k_block_unpack(BlockView b, int64_t first, int64_t count,
                                                      uint8_t* __restrict__ persp, uint8_t* __restrict__ next_persp,
                                                      int32_t* __restrict__ actions, float* __restrict__ rewards,
                                                      uint8_t* __restrict__ terminals, float* __restrict__ priorities) {
    using L = Lat<D>;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count * L::NQ) return;
    const int64_t r = t / L::NQ;
    const int c = (int)(t - r * L::NQ);
    const int64_t slot = first + r;
    const int bit = c < L::DD ? c : c - L::DD;
    const int64_t wi = (int64_t)(bit >> 6) * b.cap + slot;
    if (persp) persp[t] = (uint8_t)(((c < L::DD ? b.pv : b.pp)[wi] >> (bit & 63)) & 1);
    if (next_persp) next_persp[t] = (uint8_t)(((c < L::DD ? b.nv : b.np)[wi] >> (bit & 63)) & 1);
    if (c == 0) {
        if (actions) {
            const uint32_t a = b.action[slot];
            reinterpret_cast<int4*>(actions)[r] = make_int4(a & 255, (a >> 8) & 255, (a >> 16) & 255, a >> 24);
        }
        if (rewards) rewards[r] = b.reward[slot];
        if (terminals) terminals[r] = b.terminal[slot];
        if (priorities) priorities[r] = b.priority[slot];
    }
}

// computePrioritiesParallel (util_actor.py:268-287) over a packed block that holds T steps of n
// lattices in slot order t*n + e:  priority = | R + discount * max_a Q[t+1][e][a] - Q[t][e][op-1] |,
// evaluated in f64 like the reference (its buffers are f64 arrays) and rounded once to f32.
// q = f32[T+1][n][3] (the q_values selectActionBatch returned at each step, plus the step after the
// last: local_buffer_Q and its np.roll, Actor_mp.py:146-150); q == nullptr reads as all zeros
// (pure exploration).  Slots without a transition (op = 0) get priority 0.
__global__ __launch_bounds__(256) void k_block_priorities(BlockView b, int64_t n, int64_t T, const float* __restrict__ q,
                                                          double discount) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n * T) return;
    const uint32_t a = b.action[s];
    const int op = (int)(a >> 24);
    float pr = 0.f;
    if (op >= 1 && op <= 3) {
        double qn = 0.0, qv = 0.0;
        if (q) {
            const float* nx = q + 3 * (s + n);                // row (t+1, e)
            qn = (double)fmaxf(fmaxf(nx[0], nx[1]), nx[2]);
            qv = (double)q[3 * s + (op - 1)];
        }
        const double m = discount * qn;
        const double td = ((double)b.reward[s] + m) - qv;
        pr = (float)fabs(td);
    }
    b.priority[s] = pr;
}

// ------------------------------------------------------------------ exclusive scan of counts
// Two-level scan.  Level 1: one sum per 256 counts (part256), left behind by the kernel that produced
// the counts (block_count_partial) or, when lattices were reset by index, recomputed by
// k_scan_partials.  Level 2 (k_scan_final): a 256-thread workgroup owns SCAN_CHUNK = 2048 counts
// (8 per thread, two int4 loads), adds the partials before its chunk (<= N/256 values, one strided
// wave reduction) and writes the offsets.
constexpr int SCAN_CHUNK = 2048;

__device__ __forceinline__ int64_t wave_sum64(int64_t x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

__device__ __forceinline__ void scan_load8(const int32_t* __restrict__ counts, int64_t N, int64_t i0, int (&c)[8]) {
    if (i0 + 8 <= N) {
        const int4 a = *reinterpret_cast<const int4*>(counts + i0);
        const int4 b = *reinterpret_cast<const int4*>(counts + i0 + 4);
        c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = i0 + k < N ? counts[i0 + k] : 0;
    }
}

__global__ __launch_bounds__(256) void k_scan_partials(const int32_t* __restrict__ counts, int64_t* __restrict__ part256,
                                                       int64_t N) {
    const int64_t e = (int64_t)blockIdx.x * PART_BLOCK + threadIdx.x;
    block_count_partial(e < N ? counts[e] : 0, part256);
}

// `split` (may be NULL): cut points of the batch into G = 1 << LG parts of equal perspective count for the
// stack write (stream_write.hpp): split[k] = first lattice e with offsets[e] >= (P * k) >> LG, k = 0..G.
// A by-product of the scan: every thread knows the offsets around its eight lattices, the workgroup sums all
// level-1 partials for P, and the thread whose interval holds a cut point writes it.
__global__ __launch_bounds__(256) void k_scan_final(const int32_t* __restrict__ counts, const int64_t* __restrict__ partial,
                                                    int64_t* __restrict__ offsets, int32_t* __restrict__ counts_out,
                                                    int64_t N, int32_t* __restrict__ split, int LG) {
    __shared__ int64_t ws[4];
    __shared__ int64_t base_s, total_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_CHUNK + tid * 8;
    int c[8];
    scan_load8(counts, N, i0, c);
    int64_t mine = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) mine += c[k];
    int64_t inc = mine;                                       // inclusive scan of thread sums in the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) ws[wave] = inc;
    if (wave == 0) {                                          // sum of the 256-count partials before this chunk
        int64_t b = 0;
        for (int j = lane; j < (int)blockIdx.x * (SCAN_CHUNK / PART_BLOCK); j += 64) b += partial[j];
        b = wave_sum64(b);
        if (lane == 0) base_s = b;
    }
    if (wave == 1 && split) {                                 // P = sum of all partials
        int64_t b = 0;
        const int nparts = (int)((N + PART_BLOCK - 1) / PART_BLOCK);
        for (int j = lane; j < nparts; j += 64) b += partial[j];
        b = wave_sum64(b);
        if (lane == 0) total_s = b;
    }
    __syncthreads();
    int64_t run = base_s + inc - mine;
    for (int w = 0; w < wave; ++w) run += ws[w];
    int64_t o[9];
    o[0] = run;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k + 1] = o[k] + c[k];       // counts past N were loaded as 0
    if (i0 + 8 <= N) {
        longlong2* dst = reinterpret_cast<longlong2*>(offsets + i0);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = make_longlong2(o[2 * k], o[2 * k + 1]);
        if (counts_out) {
            *reinterpret_cast<int4*>(counts_out + i0) = make_int4(c[0], c[1], c[2], c[3]);
            *reinterpret_cast<int4*>(counts_out + i0 + 4) = make_int4(c[4], c[5], c[6], c[7]);
        }
        if (i0 + 8 == N) offsets[N] = o[8];
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (i0 + k < N) {
                offsets[i0 + k] = o[k];
                if (counts_out) counts_out[i0 + k] = c[k];
                if (i0 + k + 1 == N) offsets[N] = o[k + 1];
            }
        }
    }
    if (split) {
        const int64_t total = total_s;
        const int G = 1 << LG;
        if (total == 0) {                                     // empty stack: any valid table will do
            if (blockIdx.x == 0) for (int k = tid; k <= G; k += 256) split[k] = 0;
        } else {
            // kfloor(x) = the largest k with T_k = (total * k) >> LG <= x: float estimate, exact fix-up
            auto T = [&](int64_t k) { return (int64_t)(((uint64_t)total * (uint64_t)k) >> LG); };
            auto kfloor = [&](int64_t x) {
                int64_t k = (int64_t)((double)(x + 1) * (double)G / (double)total);
                k = k < 0 ? 0 : (k > G ? G : k);
                while (k < G && T(k + 1) <= x) ++k;
                while (k > 0 && T(k) > x) --k;
                return k;
            };
            // cut points with T_k = 0 (k = 0, and k < G / total when the stack has fewer perspectives than parts) lie in
            // no thread's interval (o[0], o[8]]: lattice 0 is their answer.  Every entry of the table is written by
            // every scan -- nothing of an earlier, larger stack survives in it.
            if (blockIdx.x == 0) {
                const int64_t kz = kfloor(0);
                for (int64_t k = tid; k <= kz; k += 256) split[k] = 0;
            }
            if (o[8] > o[0]) {
                const int64_t kA = kfloor(o[0]) + 1, kB = kfloor(o[8]);
                for (int64_t k = kA; k <= kB; ++k) {          // cut points inside (o[0], o[8]]: usually none, rarely one
                    const int64_t t = T(k);
                    int j = 0;
#pragma unroll
                    for (int q = 1; q < 8; ++q) j += o[q] < t;        // smallest j with o[j+1] >= t
                    split[k] = (int32_t)(i0 + j + 1);
                }
            }
        }
    }
}

// ------------------------------------------------------------------ perspective stack write
// Element encodings of the stack: how the bit b in {0,1} of a syndrome cell is stored, and how the
// elements of one 16-byte lane store are packed into four dwords.
struct bf16_t { unsigned short u; };
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct OutEnc;
template <> struct OutEnc<float> { static constexpr int BITS = 32; static constexpr uint32_t ONE = 0x3F800000u; };
template <> struct OutEnc<__half> { static constexpr int BITS = 16; static constexpr uint32_t ONE = 0x3C00u; };
template <> struct OutEnc<bf16_t> { static constexpr int BITS = 16; static constexpr uint32_t ONE = 0x3F80u; };
template <> struct OutEnc<uint8_t> { static constexpr int BITS = 8; static constexpr uint32_t ONE = 1u; };

__device__ __forceinline__ void wave_lds_sync() {
    // LDS ops of one wave execute in order; this only stops the compiler from moving LDS
    // reads of other lanes' data above the writes (and drains lgkmcnt).
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// (hit, cell) -> source cell table, u16[NQ][NQ] (NQ = 450 at d = 15): filled once per (device, size); used by
// k_states_transition (the stack write needs no table).
template <int D>
__global__ void k_build_lut(uint16_t* __restrict__ lut) {
    using L = Lat<D>;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L::NQ * L::NQ) return;
    const int hit = t / L::NQ, cell = t - hit * L::NQ;
    const int layer = hit >= L::DD, hrem = hit - layer * L::DD, i = hrem / D, j = hrem - i * D;
    const int c = cell >= L::DD, crem = cell - c * L::DD, r = crem / D, s = crem - r * D;
    lut[t] = (uint16_t)L::persp_src(layer, i, j, c, r, s);
}

// `VEC` stream bits -> the four dwords of one 16-byte lane store
template <typename OutT>
__device__ __forceinline__ u32x4 expand_bits(uint32_t wb) {
    using Enc = OutEnc<OutT>;
    u32x4 r;
    if (Enc::BITS == 32) {                                   // 4 bits -> 4 floats: sign-extended bit & 1.0f
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = (uint32_t)(((int32_t)(wb << (31 - k))) >> 31) & Enc::ONE;
    } else if (Enc::BITS == 16) {                            // 8 bits -> 4 x (2 halves): spread two bits, 24-bit multiply by ONE
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t t2 = (wb >> (2 * k)) & 3u;
            r[k] = __umul24((t2 | (t2 << 15)) & 0x00010001u, Enc::ONE);
        }
    } else {                                                 // 16 bits -> 4 x (4 bytes): bit i of a nibble to byte i
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = __umul24((wb >> (4 * k)) & 15u, 0x00204081u) & 0x01010101u;
    }
    return r;
}

// generateTransitionParallel on explicit u8 grids: one thread per output byte, the (hit, cell)
// -> source-cell table does shift_state + rotate_state in one lookup; loads and stores coalesced.
template <int D>
__global__ __launch_bounds__(256) void k_states_transition(const uint8_t* __restrict__ st, const uint8_t* __restrict__ nst,
                                                           const int32_t* __restrict__ actions, uint8_t* __restrict__ persp,
                                                           uint8_t* __restrict__ next_persp, int32_t* __restrict__ actions_out,
                                                           const uint16_t* __restrict__ lut, int64_t n, int* __restrict__ err) {
    using L = Lat<D>;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * L::NQ) return;
    const int64_t e = t / L::NQ;
    const int c = (int)(t - e * L::NQ);
    const int4 a = reinterpret_cast<const int4*>(actions)[e];
    if (!action_ok<D>(a.x, a.y, a.z, a.w)) { if (c == 0) atomicOr(err, ERR_ACTION); return; }
    const int src = lut[(a.x * L::DD + a.y * D + a.z) * L::NQ + c];
    if (persp) persp[t] = st[e * L::NQ + src];
    if (next_persp) next_persp[t] = nst[e * L::NQ + src];
    if (c == 0 && actions_out) reinterpret_cast<int4*>(actions_out)[e] = make_int4(a.x, L::GS, L::GS, a.w);
}

// ------------------------------------------------------------------ epsilon-greedy selection
// _selectActionBatch_prime (numba/util_actor.py:69-107): one wavefront per lattice.
template <int D>
__global__ __launch_bounds__(256) void k_select(const float* __restrict__ q, const int64_t* __restrict__ offsets,
                                                const int32_t* __restrict__ pos, const double* __restrict__ eps,
                                                const uint32_t* __restrict__ episodes, const uint32_t* __restrict__ steps,
                                                uint32_t c1, uint32_t c2, uint32_t domain,
                                                int32_t* __restrict__ actions, float* __restrict__ qv, uint64_t seed,
                                                int64_t first_env, int64_t N) {
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (e >= N) return;
    const int64_t lo = offsets[e];
    const int n = (int)(offsets[e + 1] - lo);
    if (n == 0) {
        if (lane < 4) actions[4 * e + lane] = 0;
        if (qv && lane < 3) qv[3 * e + lane] = 0.f;
        return;
    }
    // handle-bound: counters = the lattice's (episode, step); stateless: the caller's call counter
    const U4 w = draw(seed, (uint32_t)(first_env + e), episodes ? episodes[e] : c1, steps ? steps[e] : c2, domain, 0);
    const bool greedy = q != nullptr && (1.0 - eps[e]) > u01(w.x);
    int pidx, a;
    if (greedy) {
        float best = -__builtin_inff();
        int bk = 0x7fffffff;
        for (int k = lane; k < 3 * n; k += 64) {
            const float x = q[3 * lo + k];
            if (x > best) { best = x; bk = k; }               // strict >: first maximum within the lane
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int ok = __shfl_xor(bk, o, 64);
            if (ob > best || (ob == best && ok < bk)) { best = ob; bk = ok; }
        }
        bk = bk == 0x7fffffff ? 0 : bk;
        pidx = bk / 3; a = bk - 3 * pidx;
    } else {
        pidx = (int)mulhi32(w.y, (uint32_t)n);
        a = (int)mulhi32(w.z, 3);
    }
    if (lane < 3) {
        actions[4 * e + lane] = pos[3 * (lo + pidx) + lane];
        if (qv) qv[3 * e + lane] = q ? q[3 * (lo + pidx) + lane] : 0.f;
    }
    if (lane == 3) actions[4 * e + 3] = a + 1;
}

// The learner's target max (util_learner.py:48-111, predictMaxOptimized): per state the maximum of
// its (n_i, 3) Q-slice.  The reference pads every slice with zero rows up to the longest one before
// the argmax (:98-100), so a state with fewer perspectives than the longest gets max(max_q, 0);
// states without perspectives (terminal) give 0 (:74-76,108).  One wavefront per state.
__global__ __launch_bounds__(256) void k_segment_max(const float* __restrict__ q, const int64_t* __restrict__ offsets,
                                                     const int32_t* __restrict__ largest, float* __restrict__ out, int64_t n) {
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (e >= n) return;
    const int64_t lo = offsets[e];
    const int cnt = (int)(offsets[e + 1] - lo);
    float best = -__builtin_inff();
    for (int k = lane; k < 3 * cnt; k += 64) best = fmaxf(best, q[3 * lo + k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o, 64));
    if (cnt == 0) best = 0.f;
    else if (largest && cnt < *largest) best = fmaxf(best, 0.f);
    if (lane == 0) out[e] = best;
}

}  // namespace tq
