// Host-side instantiation of the product's bit-plane header (csrc/lattice.hpp) so its
// algebra can be unit-tested against the oracle on a machine without a GPU.
// TEST ONLY: built by tests/test_lattice_host.py into a temp dir with g++; it is not a
// backend of the product (the product's only compute path is the HIP library).
#include <stdint.h>
#include <string.h>

#include "lattice.hpp"

using namespace tq;

template <int D>
static void pack_qubits(const uint8_t* q, typename Lat<D>::State& s) {
    using L = Lat<D>;
    for (int l = 0; l < 2; ++l) {
        s.x[l] = L::B::zero(); s.z[l] = L::B::zero();
        for (int c = 0; c < L::DD; ++c) {
            int code = q[l * L::DD + c];
            s.x[l].flip(c, (code == 1) | (code == 2));
            s.z[l].flip(c, (code >> 1) & 1);
        }
    }
}
template <int D>
static void pack_state(const uint8_t* st, typename Lat<D>::B& v, typename Lat<D>::B& p) {
    using L = Lat<D>;
    v = L::B::zero(); p = L::B::zero();
    for (int c = 0; c < L::DD; ++c) { v.flip(c, st[c] != 0); p.flip(c, st[L::DD + c] != 0); }
}
template <int D>
static void unpack_state(const typename Lat<D>::B& v, const typename Lat<D>::B& p, uint8_t* st) {
    using L = Lat<D>;
    for (int c = 0; c < L::DD; ++c) { st[c] = (uint8_t)v.get(c); st[L::DD + c] = (uint8_t)p.get(c); }
}
template <int D>
static void unpack_qubits(const typename Lat<D>::State& s, uint8_t* q) {
    using L = Lat<D>;
    for (int l = 0; l < 2; ++l)
        for (int c = 0; c < L::DD; ++c) q[l * L::DD + c] = (uint8_t)L::code(s, l, c);
}

template <int D>
static void t_syndrome(int n, const uint8_t* q, uint8_t* st) {
    using L = Lat<D>;
    for (int e = 0; e < n; ++e) {
        typename L::State s;
        pack_qubits<D>(q + (size_t)e * L::NQ, s);
        L::syndrome(s);
        unpack_state<D>(s.v, s.p, st + (size_t)e * L::NQ);
    }
}
template <int D>
static void t_counts(int n, const uint8_t* st, int32_t* counts, uint8_t* masks) {
    using L = Lat<D>;
    for (int e = 0; e < n; ++e) {
        typename L::B v, p, e0, e1;
        pack_state<D>(st + (size_t)e * L::NQ, v, p);
        counts[e] = L::persp_count(v, p);
        L::hit_masks(v, p, e0, e1);
        unpack_state<D>(e0, e1, masks + (size_t)e * L::NQ);
    }
}
template <int D>
static void t_lut(int32_t* lut) {
    using L = Lat<D>;
    for (int layer = 0; layer < 2; ++layer)
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j)
                for (int c = 0; c < 2; ++c)
                    for (int r = 0; r < D; ++r)
                        for (int s = 0; s < D; ++s)
                            lut[((size_t)(layer * L::DD + i * D + j)) * L::NQ + c * L::DD + r * D + s] =
                                L::persp_src(layer, i, j, c, r, s);
}
template <int D>
static void t_perspective(int n, const uint8_t* st, const int32_t* act, uint8_t* out) {
    using L = Lat<D>;
    for (int e = 0; e < n; ++e) {
        typename L::B v, p, ov, op;
        pack_state<D>(st + (size_t)e * L::NQ, v, p);
        L::perspective(v, p, act[4 * e], act[4 * e + 1], act[4 * e + 2], ov, op);
        unpack_state<D>(ov, op, out + (size_t)e * L::NQ);
    }
}
template <int D>
static void t_reset(int n, uint64_t seed, int64_t first_env, const uint32_t* episodes, const double* p,
                    uint8_t* q, uint8_t* st, int32_t* rounds) {
    using L = Lat<D>;
    for (int e = 0; e < n; ++e) {
        typename L::State s;
        rounds[e] = reset_lattice<D>(s, seed, (uint32_t)(first_env + e), episodes[e], p[e]);
        unpack_qubits<D>(s, q + (size_t)e * L::NQ);
        unpack_state<D>(s.v, s.p, st + (size_t)e * L::NQ);
    }
}
template <int D>
static void t_reset_n(int n, uint64_t seed, int64_t first_env, const uint32_t* episodes, int n_err,
                      uint8_t* q, uint8_t* st) {
    using L = Lat<D>;
    for (int e = 0; e < n; ++e) {
        typename L::State s;
        reset_lattice_n<D>(s, seed, (uint32_t)(first_env + e), episodes[e], n_err);
        unpack_qubits<D>(s, q + (size_t)e * L::NQ);
        unpack_state<D>(s.v, s.p, st + (size_t)e * L::NQ);
    }
}
template <int D>
static void t_step(int n, uint8_t* q, const int32_t* act, uint8_t* st, int32_t* ground) {
    using L = Lat<D>;
    for (int e = 0; e < n; ++e) {
        typename L::State s;
        pack_qubits<D>(q + (size_t)e * L::NQ, s);
        L::apply(s, act[4 * e], act[4 * e + 1], act[4 * e + 2], act[4 * e + 3]);
        L::syndrome(s);
        unpack_qubits<D>(s, q + (size_t)e * L::NQ);
        unpack_state<D>(s.v, s.p, st + (size_t)e * L::NQ);
        ground[e] = L::ground_state(s);
    }
}

#define DISPATCH(d, CALL)                  \
    switch (d) {                           \
        case 3: CALL(3); break;            \
        case 5: CALL(5); break;            \
        case 7: CALL(7); break;            \
        case 9: CALL(9); break;            \
        case 11: CALL(11); break;          \
        case 13: CALL(13); break;          \
        case 15: CALL(15); break;          \
        case 17: CALL(17); break;          \
        case 19: CALL(19); break;          \
        case 21: CALL(21); break;          \
        default: return -1;                \
    }                                      \
    return 0;

// The perspective stack of each lattice through the bitstream algebra the stack-write kernel uses
// (PStream): rotated planes, row-rolled table, masked column rolls, emit into the stream, window
// reads back -- every lane's work done serially here.  out: u8 (P,2,d,d) at offsets[e]*NQ.
template <int D>
static void t_stream_stack(int n, const uint8_t* st, const int64_t* offsets, uint8_t* out) {
    using L = Lat<D>;
    using S = PStream<D>;
    using B = typename L::B;
    static uint32_t bits[S::MAX_DW];
    for (int e = 0; e < n; ++e) {
        B v, p, e0, e1, rv, rp;
        pack_state<D>(st + (size_t)e * L::NQ, v, p);
        L::hit_masks(v, p, e0, e1);
        const int n0 = e0.popc(), cnt = n0 + e1.popc();
        S::rotate_planes(v, p, rv, rp);
        B rr[4][D], low[D];
        for (int k = 0; k < D; ++k) {
            rr[0][k] = L::roll_rows(v, k); rr[1][k] = L::roll_rows(p, k);
            rr[2][k] = L::roll_rows(rv, k); rr[3][k] = L::roll_rows(rp, k);
            low[k] = L::lowcols(k);
        }
        const int nd = (cnt * L::NQ + 31) / 32 + 2;
        for (int i = 0; i < nd; ++i) bits[i] = 0;
        int pidx = 0;
        for (int l = 0; l < 2; ++l)
            for (int c = 0; c < L::DD; ++c) {
                if (!(l ? e1.get(c) : e0.get(c))) continue;
                int rs, cs;
                S::hit_shifts(l, c / D, c % D, rs, cs);
                const B ov = S::roll_cols_masked(rr[2 * l][rs], cs, low[cs]);
                const B op = S::roll_cols_masked(rr[2 * l + 1][rs], cs, low[cs]);
                S::emit(pidx, ov, op, [&](int idx, uint32_t val) { bits[idx] |= val; });
                ++pidx;
            }
        uint8_t* o = out + (size_t)offsets[e] * L::NQ;
        for (uint32_t rel = 0; rel < (uint32_t)(cnt * L::NQ); ++rel) o[rel] = (uint8_t)(S::window(bits, rel) & 1u);
    }
}

extern "C" {
int shim_stream_stack(int d, int n, const uint8_t* st, const int64_t* offsets, uint8_t* out) {
#define C_(D) t_stream_stack<D>(n, st, offsets, out)
    DISPATCH(d, C_)
#undef C_
}
int shim_syndrome(int d, int n, const uint8_t* q, uint8_t* st) {
#define C_(D) t_syndrome<D>(n, q, st)
    DISPATCH(d, C_)
#undef C_
}
int shim_counts(int d, int n, const uint8_t* st, int32_t* counts, uint8_t* masks) {
#define C_(D) t_counts<D>(n, st, counts, masks)
    DISPATCH(d, C_)
#undef C_
}
int shim_lut(int d, int32_t* lut) {
#define C_(D) t_lut<D>(lut)
    DISPATCH(d, C_)
#undef C_
}
int shim_perspective(int d, int n, const uint8_t* st, const int32_t* act, uint8_t* out) {
#define C_(D) t_perspective<D>(n, st, act, out)
    DISPATCH(d, C_)
#undef C_
}
int shim_reset(int d, int n, uint64_t seed, int64_t first_env, const uint32_t* episodes, const double* p,
               uint8_t* q, uint8_t* st, int32_t* rounds) {
#define C_(D) t_reset<D>(n, seed, first_env, episodes, p, q, st, rounds)
    DISPATCH(d, C_)
#undef C_
}
int shim_reset_n(int d, int n, uint64_t seed, int64_t first_env, const uint32_t* episodes, int n_err, uint8_t* q,
                 uint8_t* st) {
#define C_(D) t_reset_n<D>(n, seed, first_env, episodes, n_err, q, st)
    DISPATCH(d, C_)
#undef C_
}
int shim_step(int d, int n, uint8_t* q, const int32_t* act, uint8_t* st, int32_t* ground) {
#define C_(D) t_step<D>(n, q, act, st, ground)
    DISPATCH(d, C_)
#undef C_
}
void shim_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
    U4 r = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
}
