// TEST INFRASTRUCTURE (oracle/): a host twin of the hot path of include/toricenv.h -- the SAME C-ABI entry points
// (same names, arguments and error codes), for `device = -1`, on HOST memory, single- or multi-threaded (OpenMP).
// "Host-backend twins (device=-1) run the same API on CPU for the baseline"; section 8(d): the
// CPU baseline "additionally the C++ host backend of libtoricenv on 1 core and on all cores".
//
// It is NOT part of the product and the product never loads it: toric-rl-decoder_amd/ has exactly one compute path,
// libtoricenv.so (HIP).  Only tests/ and bench.py's cpu_baseline leg use this file (through oracle/host_twin.py).
//
// What it is made of: the product's own host+device header csrc/lattice.hpp (bit-plane lattice algebra, Philox
// contract, reset samplers, the perspective closed forms) plus a scalar restatement of the kernels around it, each
// citing the device code it follows.  So tests of the twin against the numpy / C oracle check the product's HEADER on
// a machine without a GPU, and the twin's throughput is the same-ABI, same-algorithm CPU number.
//
// Entry points present: tq_version, tq_last_error, tq_create, tq_destroy, tq_set_params, tq_set_min_qubit_errors,
// tq_set_perror_schedule, tq_num_envs, tq_size, tq_reset_all, tq_get_state, tq_get_qubits, tq_get_counters,
// tq_persp_count, tq_persp_write (f32 / u8), tq_transition_block_bytes, tq_actor_step, tq_check.  `stream` arguments are
// ignored.  Everything else of the ABI is absent here.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <vector>

#include "toricenv.h"
#include "lattice.hpp"

using namespace tq;

namespace {
thread_local char g_err[256] = "";
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
// device error latch bits (csrc/kernels.hpp:28)
enum { ERR_ACTION = 1, ERR_CAPACITY = 2, ERR_RESET_ROUNDS = 8 };
enum { PL_X0 = 0, PL_X1 = 1, PL_Z0 = 2, PL_Z1 = 3, PL_V = 4, PL_P = 5 };
int64_t align8(int64_t x) { return (x + 7) & ~(int64_t)7; }
}  // namespace

struct tq_env {
    int n, d, w;
    uint64_t seed;
    int64_t first_env;
    double p_default, terminal_reward;
    int max_steps, min_err;
    int strategy;
    double p_start, p_final, p_delta;
    std::vector<uint64_t> planes;      // [6][W][N], as in HBM (csrc/kernels.hpp:3-7)
    std::vector<uint32_t> episodes, steps;
    std::vector<int32_t> counts;
    std::vector<double> p_roof;
    int err;
};

namespace {
template <int D>
typename Lat<D>::State load_state(const tq_env* h, int64_t e) {
    constexpr int W = Lat<D>::W;
    typename Lat<D>::State s;
    const int64_t N = h->n;
    auto pl = [&](int plane, int k) { return h->planes[((int64_t)plane * W + k) * N + e]; };
    for (int k = 0; k < W; ++k) {
        s.x[0].w[k] = pl(PL_X0, k); s.x[1].w[k] = pl(PL_X1, k); s.z[0].w[k] = pl(PL_Z0, k); s.z[1].w[k] = pl(PL_Z1, k);
        s.v.w[k] = pl(PL_V, k); s.p.w[k] = pl(PL_P, k);
    }
    return s;
}
template <int D>
void store_state(tq_env* h, int64_t e, const typename Lat<D>::State& s) {
    constexpr int W = Lat<D>::W;
    const int64_t N = h->n;
    auto pl = [&](int plane, int k) -> uint64_t& { return h->planes[((int64_t)plane * W + k) * N + e]; };
    for (int k = 0; k < W; ++k) {
        pl(PL_X0, k) = s.x[0].w[k]; pl(PL_X1, k) = s.x[1].w[k]; pl(PL_Z0, k) = s.z[0].w[k]; pl(PL_Z1, k) = s.z[1].w[k];
        pl(PL_V, k) = s.v.w[k]; pl(PL_P, k) = s.p.w[k];
    }
}

// k-th set bit of [E0 | E1] as layer*DD + row*D + col (csrc/kernels.hpp:67-103; plain loop here)
template <int D>
int kth_hit(const typename Lat<D>::B& e0, const typename Lat<D>::B& e1, int k) {
    constexpr int DD = Lat<D>::DD;
    for (int l = 0; l < 2; ++l)
        for (int c = 0; c < DD; ++c)
            if ((l ? e1.get(c) : e0.get(c)) && k-- == 0) return l * DD + c;
    return -1;
}

// tq_reset_all: k_reset, all-lattice mode (csrc/kernels.hpp:131-162)
template <int D>
int reset_all(tq_env* h, const double* p_err) {
    using L = Lat<D>;
    int err = 0;
#pragma omp parallel for schedule(static) reduction(| : err)
    for (int64_t e = 0; e < h->n; ++e) {
        typename L::State s;
        const uint32_t ep = h->episodes[e];
        if (h->min_err > 0) reset_lattice_n<D>(s, h->seed, (uint32_t)(h->first_env + e), ep, h->min_err);
        else reset_lattice<D>(s, h->seed, (uint32_t)(h->first_env + e), ep, p_err ? p_err[e] : h->p_default);
        if (!(s.v.any() || s.p.any())) err |= ERR_RESET_ROUNDS;
        store_state<D>(h, e, s);
        h->episodes[e] = ep + 1;
        h->steps[e] = 0;
        h->counts[e] = L::persp_count(s.v, s.p);
    }
    h->err |= err;
    return TQ_OK;
}

struct Block {     // SoA sections of a packed transition block (include/toricenv.h; csrc/kernels.hpp:206-225)
    uint64_t *pv, *pp, *nv, *np;
    uint32_t* action; float* reward; float* priority; uint8_t* terminal;
    int64_t cap;
};
Block block_view(void* base, int W, int64_t cap) {
    Block b;
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

// tq_actor_step: k_actor_step (csrc/kernels.hpp:333-433), one lattice after the other
template <int D>
int actor_step(tq_env* h, const int32_t* actions, int32_t* actions_out, float* rewards, uint8_t* terminals, void* block,
               int64_t block_cap, int64_t slot_base) {
    using L = Lat<D>;
    constexpr int W = L::W, GS = L::GS, DD = L::DD;
    const Block blk = block ? block_view(block, W, block_cap) : Block{};
    int err = 0;
#pragma omp parallel for schedule(static) reduction(| : err)
    for (int64_t e = 0; e < h->n; ++e) {
        typename L::State s = load_state<D>(h, e);
        uint32_t ep = h->episodes[e], st = h->steps[e];
        const uint32_t env = (uint32_t)(h->first_env + e);
        int layer, row, col, op;
        bool ok;
        if (actions) {
            layer = actions[4 * e]; row = actions[4 * e + 1]; col = actions[4 * e + 2]; op = actions[4 * e + 3];
            ok = ((unsigned)layer < 2u) & ((unsigned)row < (unsigned)D) & ((unsigned)col < (unsigned)D) & ((unsigned)(op - 1) < 3u);
            if (!ok && op != 0) err |= ERR_ACTION;
        } else {                                               // non-greedy branch of _selectActionBatch_prime (numba/util_actor.py:97-98)
            typename L::B e0, e1;
            L::hit_masks(s.v, s.p, e0, e1);
            const int n = e0.popc() + e1.popc();
            ok = n > 0;
            layer = row = col = op = 0;
            if (ok) {
                const U4 w = draw(h->seed, env, ep, st, DOMAIN_SEL, 0);
                const int hh = kth_hit<D>(e0, e1, (int)mulhi32(w.y, (uint32_t)n));
                layer = hh >= DD;
                const int rem = hh - layer * DD;
                row = rem / D; col = rem - row * D;
                op = 1 + (int)mulhi32(w.z, 3);
            }
        }
        if (actions_out) { actions_out[4 * e] = layer; actions_out[4 * e + 1] = row; actions_out[4 * e + 2] = col; actions_out[4 * e + 3] = op; }
        const typename L::B v0 = s.v, p0 = s.p;
        const int before = v0.popc() + p0.popc();
        if (ok) L::apply(s, layer, row, col, op);
        L::syndrome(s);
        const int after = s.v.popc() + s.p.popc();
        const int terminal = after == 0;
        const float reward = terminal ? (float)h->terminal_reward : (float)(before - after);
        st += 1;
        if (rewards) rewards[e] = reward;
        if (terminals) terminals[e] = (uint8_t)terminal;
        if (block) {                                           // every slot is written, every step (csrc/kernels.hpp:244-283)
            const int64_t slot = slot_base + e;
            typename L::B a, c;
            if (ok) {
                L::perspective(v0, p0, layer, row, col, a, c);
                for (int k = 0; k < W; ++k) { blk.pv[(int64_t)k * blk.cap + slot] = a.w[k]; blk.pp[(int64_t)k * blk.cap + slot] = c.w[k]; }
                L::perspective(s.v, s.p, layer, row, col, a, c);
                for (int k = 0; k < W; ++k) { blk.nv[(int64_t)k * blk.cap + slot] = a.w[k]; blk.np[(int64_t)k * blk.cap + slot] = c.w[k]; }
                blk.action[slot] = (uint32_t)layer | ((uint32_t)GS << 8) | ((uint32_t)GS << 16) | ((uint32_t)op << 24);
                blk.reward[slot] = reward;
                blk.terminal[slot] = (uint8_t)terminal;
            } else {
                for (int k = 0; k < W; ++k) {
                    blk.pv[(int64_t)k * blk.cap + slot] = 0; blk.pp[(int64_t)k * blk.cap + slot] = 0;
                    blk.nv[(int64_t)k * blk.cap + slot] = 0; blk.np[(int64_t)k * blk.cap + slot] = 0;
                }
                blk.action[slot] = 0u; blk.reward[slot] = 0.f; blk.terminal[slot] = 0;
            }
        }
        // reset policy of the caller (Actor_mp.py:171-183)
        if (terminal || st > (uint32_t)h->max_steps) {
            double p = h->p_default;
            if (h->strategy != 0) {
                double roof = h->p_roof[e] + h->p_delta;
                roof = roof < h->p_final ? roof : h->p_final;
                h->p_roof[e] = roof;
                p = roof;
                if (h->strategy == 2) {
                    const U4 w = draw(h->seed, env, ep, 0, DOMAIN_PERR, 0);
                    const double span = roof - h->p_start;
                    const double t = span * u01(w.x);
                    p = h->p_start + t;
                }
            }
            typename L::State fresh;
            if (h->min_err > 0) reset_lattice_n<D>(fresh, h->seed, env, ep, h->min_err);
            else reset_lattice<D>(fresh, h->seed, env, ep, p);
            if (!(fresh.v.any() || fresh.p.any())) err |= ERR_RESET_ROUNDS;
            s = fresh; ep += 1; st = 0;
        }
        store_state<D>(h, e, s);
        h->episodes[e] = ep;
        h->steps[e] = st;
        h->counts[e] = L::persp_count(s.v, s.p);
    }
    h->err |= err;
    return TQ_OK;
}

// tq_persp_write: the env-major stack, hits in argwhere order, all layer-0 hits before the layer-1 hits
// (numba/util.py:53-59,67-74; what k_persp_stream writes)
template <int D, typename OutT>
int persp_write(tq_env* h, const int64_t* offsets, OutT* out, int32_t* positions, int64_t capacity) {
    using L = Lat<D>;
    constexpr int DD = L::DD, NQ = L::NQ;
    const int64_t N = h->n;
    int64_t e_stop = N;
    if (offsets[N] > capacity) {                              // the lattices that fit whole are written (toricenv.h)
        h->err |= ERR_CAPACITY;
        e_stop = 0;
        while (e_stop < N && offsets[e_stop + 1] <= capacity) ++e_stop;
    }
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t e = 0; e < e_stop; ++e) {
        const typename L::State s = load_state<D>(h, e);
        typename L::B e0, e1;
        L::hit_masks(s.v, s.p, e0, e1);
        int64_t q = offsets[e];
        for (int l = 0; l < 2; ++l)
            for (int c = 0; c < DD; ++c) {
                if (!(l ? e1.get(c) : e0.get(c))) continue;
                const int i = c / D, j = c - i * D;
                typename L::B ov, op;
                L::perspective(s.v, s.p, l, i, j, ov, op);
                OutT* o = out + q * NQ;
                for (int b = 0; b < DD; ++b) { o[b] = (OutT)ov.get(b); o[DD + b] = (OutT)op.get(b); }
                if (positions) { positions[3 * q] = l; positions[3 * q + 1] = i; positions[3 * q + 2] = j; }
                ++q;
            }
    }
    return TQ_OK;
}

template <int D>
void get_state(const tq_env* h, uint8_t* out, bool qubits) {
    using L = Lat<D>;
    constexpr int DD = L::DD, NQ = L::NQ;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < h->n; ++e) {
        const typename L::State s = load_state<D>(h, e);
        uint8_t* o = out + e * NQ;
        for (int c = 0; c < DD; ++c) {
            if (qubits) { o[c] = (uint8_t)L::code(s, 0, c); o[DD + c] = (uint8_t)L::code(s, 1, c); }
            else { o[c] = (uint8_t)s.v.get(c); o[DD + c] = (uint8_t)s.p.get(c); }
        }
    }
}

#define DISPATCH_D(d, CALL)          \
    switch (d) {                     \
        case 3: CALL(3); break;      \
        case 5: CALL(5); break;      \
        case 7: CALL(7); break;      \
        case 9: CALL(9); break;      \
        case 11: CALL(11); break;    \
        case 13: CALL(13); break;    \
        case 15: CALL(15); break;    \
        case 17: CALL(17); break;    \
        case 19: CALL(19); break;    \
        case 21: CALL(21); break;    \
        default: return fail(TQ_E_INVALID, "unsupported lattice size d=%d (odd 3..21)", d); \
    }
}  // namespace

extern "C" {

int tq_version(void) { return TQ_VERSION; }
const char* tq_last_error(void) { return g_err; }

int tq_create(tq_env** out, int n_envs, int d, int device, uint64_t seed, int64_t first_env_id) {
    if (!out) return fail(TQ_E_INVALID, "out is NULL");
    *out = nullptr;
    if (device != -1) return fail(TQ_E_INVALID, "this is the host twin of the ABI: device must be -1 (got %d)", device);
    if (n_envs <= 0) return fail(TQ_E_INVALID, "n_envs must be > 0 (got %d)", n_envs);
    if (d < 3 || d > 21 || !(d & 1)) return fail(TQ_E_INVALID, "unsupported lattice size d=%d (odd 3..21)", d);
    if (first_env_id < 0 || first_env_id + n_envs > 0xFFFFFFFFll) return fail(TQ_E_INVALID, "global env ids must fit in 32 bits");
    tq_env* h = new (std::nothrow) tq_env();
    if (!h) return fail(TQ_E_HIP, "out of host memory");
    h->n = n_envs; h->d = d; h->w = (d * d + 63) / 64;
    h->seed = seed; h->first_env = first_env_id;
    h->p_default = 0.1; h->terminal_reward = 100.0; h->max_steps = 75; h->min_err = 0;
    h->strategy = 0; h->p_start = h->p_final = 0.1; h->p_delta = 0.0;
    h->planes.assign((size_t)6 * h->w * n_envs, 0);
    h->episodes.assign(n_envs, 0); h->steps.assign(n_envs, 0); h->counts.assign(n_envs, 0);
    h->p_roof.assign(n_envs, 0.0);
    h->err = 0;
    *out = h;
    return TQ_OK;
}
int tq_destroy(tq_env* h) { delete h; return TQ_OK; }
int tq_num_envs(const tq_env* h) { return h ? h->n : 0; }
int tq_size(const tq_env* h) { return h ? h->d : 0; }

int tq_set_params(tq_env* h, double p_error_default, double terminal_reward, int max_steps_per_episode) {
    if (!h) return fail(TQ_E_INVALID, "NULL handle");
    const bool p_unused = h->min_err > 0;
    if (!((p_error_default > 0.0 || (p_unused && p_error_default == 0.0)) && p_error_default <= 1.0))
        return fail(TQ_E_INVALID, "p_error must be in (0,1] (0 is accepted only with min_qubit_errors > 0)");
    if (max_steps_per_episode < 1) return fail(TQ_E_INVALID, "max_steps_per_episode must be >= 1");
    h->p_default = p_error_default; h->terminal_reward = terminal_reward; h->max_steps = max_steps_per_episode;
    return TQ_OK;
}
int tq_set_min_qubit_errors(tq_env* h, int n_errors) {
    if (!h) return fail(TQ_E_INVALID, "NULL handle");
    if (n_errors < 0 || n_errors > 2 * h->d * h->d) return fail(TQ_E_INVALID, "min_qubit_errors must be in [0, 2*d*d]");
    if (n_errors == 0 && !(h->p_default > 0.0))
        return fail(TQ_E_INVALID, "min_qubit_errors = 0 selects the depolarizing sampler, which needs p_error in (0,1]");
    h->min_err = n_errors;
    return TQ_OK;
}
int tq_set_perror_schedule(tq_env* h, int strategy, double p_start, double p_final, double p_delta) {
    if (!h) return fail(TQ_E_INVALID, "NULL handle");
    if (strategy < TQ_PERR_FIXED || strategy > TQ_PERR_RANDOM) return fail(TQ_E_INVALID, "unknown p_error strategy %d", strategy);
    if (!(p_start > 0.0 && p_start <= 1.0 && p_final > 0.0 && p_final <= 1.0 && p_delta >= 0.0))
        return fail(TQ_E_INVALID, "p_error schedule needs 0 < p_start, p_final <= 1 and p_delta >= 0");
    h->strategy = strategy; h->p_start = p_start; h->p_final = p_final; h->p_delta = p_delta;
    for (auto& r : h->p_roof) r = p_start;
    return TQ_OK;
}

int tq_reset_all(tq_env* h, const double* p_err, void*) {
    if (!h) return fail(TQ_E_INVALID, "NULL handle");
#define CALL(D) return reset_all<D>(h, p_err)
    DISPATCH_D(h->d, CALL)
#undef CALL
    return TQ_OK;
}
int tq_get_state(tq_env* h, uint8_t* out, void*) {
    if (!h || !out) return fail(TQ_E_INVALID, "NULL handle / out");
#define CALL(D) get_state<D>(h, out, false)
    DISPATCH_D(h->d, CALL)
#undef CALL
    return TQ_OK;
}
int tq_get_qubits(tq_env* h, uint8_t* out, void*) {
    if (!h || !out) return fail(TQ_E_INVALID, "NULL handle / out");
#define CALL(D) get_state<D>(h, out, true)
    DISPATCH_D(h->d, CALL)
#undef CALL
    return TQ_OK;
}
int tq_get_counters(tq_env* h, uint32_t* episodes, uint32_t* steps, void*) {
    if (!h) return fail(TQ_E_INVALID, "NULL handle");
    if (episodes) memcpy(episodes, h->episodes.data(), 4 * (size_t)h->n);
    if (steps) memcpy(steps, h->steps.data(), 4 * (size_t)h->n);
    return TQ_OK;
}

int tq_persp_count(tq_env* h, int32_t* counts, int64_t* offsets, void*) {
    if (!h || !offsets) return fail(TQ_E_INVALID, "NULL handle / offsets");
    int64_t run = 0;
    for (int64_t e = 0; e < h->n; ++e) {
        offsets[e] = run;
        run += h->counts[e];
        if (counts) counts[e] = h->counts[e];
    }
    offsets[h->n] = run;
    return TQ_OK;
}
int tq_persp_write(tq_env* h, const int64_t* offsets, void* out, int32_t* positions, int64_t capacity, int dtype, void*) {
    if (!h || !offsets || !out) return fail(TQ_E_INVALID, "NULL handle / offsets / out");
    if (capacity < 0) return fail(TQ_E_INVALID, "negative capacity");
    if (dtype == TQ_F32) {
#define CALL(D) return persp_write<D, float>(h, offsets, (float*)out, positions, capacity)
        DISPATCH_D(h->d, CALL)
#undef CALL
    } else if (dtype == TQ_U8) {
#define CALL(D) return persp_write<D, uint8_t>(h, offsets, (uint8_t*)out, positions, capacity)
        DISPATCH_D(h->d, CALL)
#undef CALL
    }
    return fail(TQ_E_INVALID, "the host twin writes f32 and u8 stacks only (dtype %d)", dtype);
}

int64_t tq_transition_block_bytes(int d, int64_t cap) {
    if (d < 3 || d > 21 || !(d & 1) || cap <= 0) return -1;
    const int W = (d * d + 63) / 64;
    return 4 * 8 * (int64_t)W * cap + 3 * align8(4 * cap) + align8(cap);
}

int tq_actor_step(tq_env* h, const int32_t* actions, int32_t* actions_out, float* rewards, uint8_t* terminals, void* block,
                  int64_t block_cap, int64_t slot_base, void*) {
    if (!h) return fail(TQ_E_INVALID, "NULL handle");
    if (block && (slot_base < 0 || slot_base + h->n > block_cap)) return fail(TQ_E_INVALID, "slots outside the block");
#define CALL(D) return actor_step<D>(h, actions, actions_out, rewards, terminals, block, block_cap, slot_base)
    DISPATCH_D(h->d, CALL)
#undef CALL
    return TQ_OK;
}

int tq_check(tq_env* h, void*) {
    if (!h) return fail(TQ_E_INVALID, "NULL handle");
    const int e = h->err;
    h->err = 0;
    if (e & ERR_ACTION) return fail(TQ_E_ACTION, "an action outside the lattice / op not in 1..3 was seen");
    if (e & ERR_CAPACITY) return fail(TQ_E_CAPACITY, "perspective stack capacity exceeded");
    if (e & ERR_RESET_ROUNDS) return fail(TQ_E_RESET, "a reset hit the round limit without producing a defect");
    return TQ_OK;
}

}  // extern "C"
