// Bit-plane algebra of one toric-code lattice (d x d, two qubit layers).
//
// A d*d grid is one bitset of DD = d*d bits (bit r*d+c = cell (r,c)), W = ceil(DD/64)
// 64-bit words.  A lattice is six such planes:
//   x[0], x[1]   X-component of the Pauli on layer 0/1 qubits   (code 1 or 2)
//   z[0], z[1]   Z-component                                    (code 2 or 3)
//   v, p         vertex / plaquette syndrome  (state[0], state[1] of the reference)
// Pauli codes I=0 X=1 Y=2 Z=3 (reference docs/toric_model.md:11); the Pauli product
// mod phase is XOR of codes = XOR of (x,z) pairs (docs/toric_model.md:15).
//
// Geometry (reference src/util.py:68-69,77-78):
//   v[i,j] = z0[i,j] ^ z0[i-1,j] ^ z1[i,j] ^ z1[i,j-1]
//   p[i,j] = x0[i,j] ^ x0[i,j+1] ^ x1[i,j] ^ x1[i+1,j]
// Defect-adjacent qubits (reference src/numba/util.py:48-52,62-66):
//   E0[i,j] = v[i,j] | v[i+1,j] | p[i,j] | p[i,j-1]
//   E1[i,j] = v[i,j] | v[i,j+1] | p[i,j] | p[i-1,j]
//
// Everything here is integer bit arithmetic, usable from host and device code.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TQ_HD __host__ __device__ __forceinline__
#else
#define TQ_HD inline
#endif

namespace tq {

TQ_HD int popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}

template <int W>
struct Bits {
    uint64_t w[W];

    TQ_HD static Bits zero() { Bits b; for (int k = 0; k < W; ++k) b.w[k] = 0; return b; }
    TQ_HD Bits operator|(const Bits& o) const { Bits r; for (int k = 0; k < W; ++k) r.w[k] = w[k] | o.w[k]; return r; }
    TQ_HD Bits operator&(const Bits& o) const { Bits r; for (int k = 0; k < W; ++k) r.w[k] = w[k] & o.w[k]; return r; }
    TQ_HD Bits operator^(const Bits& o) const { Bits r; for (int k = 0; k < W; ++k) r.w[k] = w[k] ^ o.w[k]; return r; }
    TQ_HD bool any() const { uint64_t a = 0; for (int k = 0; k < W; ++k) a |= w[k]; return a != 0; }
    TQ_HD int popc() const { int n = 0; for (int k = 0; k < W; ++k) n += popc64(w[k]); return n; }
    TQ_HD int get(int i) const {
        if (W == 1) return (int)((w[0] >> i) & 1);
        // Every word is shifted and masked with "is this the word?" -- NOT a chain of selects between the words: LLVM turns
        // select(c, w[k], w[0]) into a load through a selected POINTER before the struct is split into registers, which
        // pins the whole bitset in memory (scratch, or 48-64 bytes of LDS per thread: 37-49 KB per workgroup at d >= 13).
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < W; ++k) acc |= (w[k] >> (i & 63)) & (uint64_t)((i >> 6) == k);
        return (int)acc;
    }
    TQ_HD void flip(int i, int on) {
        for (int k = 0; k < W; ++k) w[k] ^= (uint64_t)(on & ((i >> 6) == k)) << (i & 63);
    }
    // number of set bits strictly below position i
    TQ_HD int rank(int i) const {
        int n = 0;
        for (int k = 0; k < W; ++k) {
            int lim = i - 64 * k;                       // bits of word k below i
            uint64_t m = lim >= 64 ? ~0ull : (lim <= 0 ? 0ull : ((1ull << lim) - 1));
            n += popc64(w[k] & m);
        }
        return n;
    }
    // logical shifts by 0 <= s < 64*W.  Word indices stay compile-time (select chains) so a
    // runtime shift never turns the word array into scratch memory on the GPU.
    TQ_HD Bits shl(int s) const {
        Bits r;
        const int ws = s >> 6, bs = s & 63;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            uint64_t lo = 0, lo2 = 0;
#pragma unroll
            for (int j = 0; j <= k; ++j) {
                lo = (ws == k - j) ? w[j] : lo;
                lo2 = (ws == k - j - 1) ? w[j] : lo2;
            }
            r.w[k] = (lo << bs) | ((lo2 >> 1) >> (63 - bs));
        }
        return r;
    }
    // shifts by 0 <= s < 64 (no word moves)
    TQ_HD Bits shl_small(int s) const {
        Bits r;
#pragma unroll
        for (int k = 0; k < W; ++k) r.w[k] = (w[k] << s) | (k > 0 ? (w[k > 0 ? k - 1 : 0] >> 1) >> (63 - s) : 0ull);
        return r;
    }
    TQ_HD Bits shr_small(int s) const {
        Bits r;
#pragma unroll
        for (int k = 0; k < W; ++k) r.w[k] = (w[k] >> s) | (k < W - 1 ? (w[k < W - 1 ? k + 1 : k] << 1) << (63 - s) : 0ull);
        return r;
    }
    // dword j of the bitset (j compile-time after unrolling); 0 beyond the last word
    TQ_HD uint32_t dword(int j) const { return (j >= 0 && j < 2 * W) ? (uint32_t)(w[(j >= 0 && j < 2 * W) ? j / 2 : 0] >> (32 * (j & 1))) : 0u; }
    TQ_HD Bits shr(int s) const {
        Bits r;
        const int ws = s >> 6, bs = s & 63;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            uint64_t hi = 0, hi2 = 0;
#pragma unroll
            for (int j = k; j < W; ++j) {
                hi = (ws == j - k) ? w[j] : hi;
                hi2 = (ws == j - k - 1) ? w[j] : hi2;
            }
            r.w[k] = (hi >> bs) | ((hi2 << 1) << (63 - bs));
        }
        return r;
    }
};

template <int D>
struct Lat {
    static constexpr int DD = D * D;
    static constexpr int NQ = 2 * DD;
    static constexpr int W = (DD + 63) / 64;
    static constexpr int GS = D / 2;                 // grid_shift = int(d/2) (Actor_mp.py:59)
    using B = Bits<W>;

    // mask with the DD valid bits set
    TQ_HD static B full() {
        B m;
        for (int k = 0; k < W; ++k) {
            int rem = DD - 64 * k;
            m.w[k] = rem >= 64 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1));
        }
        return m;
    }
    // mask whose every row holds the D-bit pattern `rowpat` (bit c of rowpat = column c);
    // word index and shift amounts are compile-time after unrolling.
    TQ_HD static B rows_of(uint64_t rowpat) {
        B m;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            uint64_t acc = 0;
#pragma unroll
            for (int r = 0; r < D; ++r) {
                const int sh = r * D - 64 * k;             // position of row r inside word k
                if (sh >= 0 && sh < 64) acc |= rowpat << sh;
                else if (sh < 0 && sh > -D) acc |= rowpat >> (-sh);
            }
            m.w[k] = acc;
        }
        return m & full();
    }
    TQ_HD static B colmask(int c) { return rows_of(1ull << c); }
    // mask of the columns [0, k)
    TQ_HD static B lowcols(int k) { return rows_of((1ull << k) - 1); }

    // out[r, c] = in[(r - k) mod D, c]   (np.roll(in, +k, axis=0)), 0 <= k < D
    TQ_HD static B roll_rows(const B& a, int k) {
        if (k == 0) return a;
        return (a.shl(k * D) | a.shr(DD - k * D)) & full();
    }
    // out[r, c] = in[r, (c - k) mod D]   (np.roll(in, +k, axis=1)), 0 <= k < D
    TQ_HD static B roll_cols(const B& a, int k) {
        if (k == 0) return a;
        B low = lowcols(k);                                    // destination columns [0,k) come from the wrap
        B hi = full() ^ low;
        return (a.shl(k) & hi) | (a.shr(D - k) & low);
    }

    struct State {        // one lattice
        B x[2], z[2], v, p;
    };

    // createSyndromOpt (absent upstream; geometry as above)
    TQ_HD static void syndrome(State& s) {
        s.v = s.z[0] ^ roll_rows(s.z[0], 1) ^ s.z[1] ^ roll_cols(s.z[1], 1);
        s.p = s.x[0] ^ roll_cols(s.x[0], D - 1) ^ s.x[1] ^ roll_rows(s.x[1], D - 1);
    }
    // defect-adjacent qubit masks E0 / E1 (numba/util.py:48-52,62-66)
    TQ_HD static void hit_masks(const B& v, const B& p, B& e0, B& e1) {
        e0 = v | roll_rows(v, D - 1) | p | roll_cols(p, 1);
        e1 = v | roll_cols(v, D - 1) | p | roll_rows(p, 1);
    }
    TQ_HD static int persp_count(const B& v, const B& p) {
        B e0, e1;
        hit_masks(v, p, e0, e1);
        return e0.popc() + e1.popc();
    }
    // env.step's qubit update: q[layer,row,col] ^= op  (op in 1..3)
    TQ_HD static void apply(State& s, int layer, int row, int col, int op) {
        const int i = row * D + col;
        const int fx = (op == 1) | (op == 2), fz = (op >> 1) & 1;
        s.x[0].flip(i, fx & (layer == 0));
        s.x[1].flip(i, fx & (layer == 1));
        s.z[0].flip(i, fz & (layer == 0));
        s.z[1].flip(i, fz & (layer == 1));
    }
    TQ_HD static int code(const State& s, int layer, int i) {
        const int x = layer ? s.x[1].get(i) : s.x[0].get(i);
        const int z = layer ? s.z[1].get(i) : s.z[0].get(i);
        return z ? (x ? 2 : 3) : x;                      // (x,z): (1,0)=X=1 (1,1)=Y=2 (0,1)=Z=3
    }
    // evalGroundState (theory): all four layer parities even
    TQ_HD static int ground_state(const State& s) {
        return !((s.x[0].popc() | s.x[1].popc() | s.z[0].popc() | s.z[1].popc()) & 1);
    }

    // Source cell (flat index into (2,d,d)) of output cell (c,r,s) of the perspective
    // centred on hit (layer,i,j)  -- closed forms of SURVEY 8(a) A5/A6:
    //   layer 0: P0[c,r,s] = state[c,(r+i-gs)%d,(s+j-gs)%d]
    //   layer 1: P1[1,r,s] = state[1,(s+i-gs)%d,(d-1-r+j-gs)%d]
    //            P1[0,r,s] = state[0,(s+i-gs)%d,((d-r)%d+j-gs)%d]
    TQ_HD static int persp_src(int layer, int i, int j, int c, int r, int s) {
        // every "% D" below has an argument in [0, 2D): a conditional subtraction (no division on the device)
        int a = i - GS + D, b = j - GS + D;
        a = a >= D ? a - D : a; b = b >= D ? b - D : b;
        int row, col;
        if (layer == 0) { row = r; col = s; }
        else { row = s; col = c ? (D - 1 - r) : (r ? D - r : 0); }
        row += a; col += b;
        row = row >= D ? row - D : row; col = col >= D ? col - D : col;
        return c * DD + row * D + col;
    }

    // The perspective of (v,p) centred on qubit (layer,i,j), as bit-planes
    // (shift_state + rotate_state of util.py:87-102 in one pass).
    TQ_HD static void perspective(const B& v, const B& p, int layer, int i, int j, B& ov, B& op) {
        const int a = (GS - i + D) % D, b = (GS - j + D) % D;     // np.roll amounts
        B rv = roll_cols(roll_rows(v, a), b);
        B rp = roll_cols(roll_rows(p, a), b);
        if (layer == 0) { ov = rv; op = rp; return; }
#pragma unroll
        for (int k = 0; k < W; ++k) {
            uint64_t av = 0, ap = 0;
            const int nb = (DD - 64 * k) < 64 ? (DD - 64 * k) : 64;
            for (int bit = 0; bit < nb; ++bit) {
                const int o = 64 * k + bit, r = o / D, s = o - r * D;
                const int sp = s * D + (D - 1 - r);               // rot_p[r,s] = p[s,d-1-r]
                const int sv = s * D + (r ? D - r : 0);           // rot_v[r,s] = v[s,(d-r)%d]
                ap |= (uint64_t)rp.get(sp) << bit;
                av |= (uint64_t)rv.get(sv) << bit;
            }
            ov.w[k] = av; op.w[k] = ap;
        }
    }
};

// ------------------------------------------------------------------ perspective bitstream
// The perspective stack of ONE lattice as a bit string: bit  pidx*NQ + cell  = element `cell` of
// the lattice's pidx-th perspective (hits in the reference's argwhere order).  The stack-write
// kernel builds this string in LDS with bit-parallel plane operations and then only expands bits
// into elements; no (hit, cell) -> source table is needed:
//   * layer-0 hit (i,j): np.roll of (V,P) by (gs-i, gs-j)                        (shift_state)
//   * layer-1 hit (i,j): rotate_state(shift_state(.)) = np.roll of the ONCE-rotated planes
//     (RV,RP) = rotate_state(V,P) by rows +(j-gs), cols +(gs-i):
//       P1[c,r,s] = state[c,(s+i-gs)%d, rotcol(r)+j-gs] = R[c,(r-(j-gs))%d,(s+(i-gs))%d]
//     so every hit costs two row-rolled planes (shared by all hits with the same row amount, kept
//     in a small table) and two column rolls.
template <int D>
struct PStream {
    using L = Lat<D>;
    using B = typename L::B;
    static constexpr int NQ = L::NQ, DD = L::DD, W = L::W;
    static constexpr int ND = (NQ + 31) / 32;                    // dwords of one perspective's NQ-bit string
    static constexpr int MAX_DW = (NQ * NQ + 31) / 32 + 2;       // all NQ perspectives; +2: the 64-bit window read stays inside

    // rotate_state (util.py:87-94) on bit-planes: source bit of output bit o = (r,s)
    TQ_HD static int rot_src_p(int o) { const int r = o / D, s = o - r * D; return s * D + (D - 1 - r); }      // rot_p[r,s] = p[s,d-1-r]
    TQ_HD static int rot_src_v(int o) { const int r = o / D, s = o - r * D; return s * D + (r ? D - r : 0); }  // rot_v[r,s] = v[s,(d-r)%d]
    TQ_HD static void rotate_planes(const B& v, const B& p, B& rv, B& rp) {
        rv = B::zero(); rp = B::zero();
        for (int o = 0; o < DD; ++o) { rv.flip(o, v.get(rot_src_v(o))); rp.flip(o, p.get(rot_src_p(o))); }
    }
    // np.roll(a, +k, axis=1) with the low-column mask lowcols(k) supplied (table lookup in the kernel)
    TQ_HD static B roll_cols_masked(const B& a, int k, const B& low) {
        const B hi = L::full() ^ low;
        return (a.shl_small(k) & hi) | (a.shr_small(D - k) & low);
    }
    // np.roll amounts (rows, cols) that turn the planes -- (V,P) for layer 0, (RV,RP) for layer 1 --
    // into the perspective of hit (layer,i,j)
    TQ_HD static void hit_shifts(int layer, int i, int j, int& rs, int& cs) {
        const int ci = (L::GS - i + D) % D;
        rs = layer ? (j - L::GS + D) % D : ci;
        cs = layer ? ci : (L::GS - j + D) % D;
    }
    // dword j of the NQ-bit string  ov | (op << DD)  of one perspective (bits of ov/op above DD are 0)
    TQ_HD static uint32_t string_dword(const B& ov, const B& op, int j) {
        constexpr int q = DD / 32, r = DD % 32;
        uint32_t t = ov.dword(j) | (op.dword(j - q) << r);
        if (r) t |= (op.dword(j - q - 1) >> 1) >> (31 - r);
        return t;
    }
    // OR the string of perspective `pidx` into the lattice's bitstream: orfn(dword index, value)
    template <class OrFn>
    TQ_HD static void emit(int pidx, const B& ov, const B& op, OrFn&& orfn) {
        const uint32_t pos = (uint32_t)pidx * NQ;
        const int base = (int)(pos >> 5), sh = (int)(pos & 31);
        uint32_t prev = 0;
#pragma unroll
        for (int j = 0; j <= ND; ++j) {
            const uint32_t cur = j < ND ? string_dword(ov, op, j) : 0u;
            orfn(base + j, (cur << sh) | ((prev >> 1) >> (31 - sh)));
            prev = cur;
        }
    }
    // the 32 stream bits starting at bit `rel`
    TQ_HD static uint32_t window(const uint32_t* bits, uint32_t rel) {
        const uint32_t idx = rel >> 5, ph = rel & 31;
        return (uint32_t)(((((uint64_t)bits[idx + 1]) << 32) | bits[idx]) >> ph);
    }
};

// ------------------------------------------------------------------ Philox4x32-10
struct U4 { uint32_t x, y, z, w; };

TQ_HD uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

TQ_HD U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

enum : uint32_t { DOMAIN_ERR = 0, DOMAIN_SEL = 1, DOMAIN_PERR = 2, DOMAIN_SEL_CALL = 3, DOMAIN_NERR = 4 };
constexpr int MAX_RESET_ROUNDS = 4096;

TQ_HD U4 draw(uint64_t seed, uint32_t env, uint32_t episode, uint32_t round, uint32_t domain, uint32_t index) {
    return philox4x32_10(env, episode, round, (domain << 24) | index, (uint32_t)seed, (uint32_t)(seed >> 32));
}
TQ_HD double u01(uint32_t w) { return (double)w * (1.0 / 4294967296.0); }

// env.reset(p_error): depolarizing draw rounds until the syndrome is non-empty
// (results/small_p_error_test.py:22-31,109-120).  RNG contract: DESIGN.md section 4.
template <int D>
TQ_HD int reset_lattice(typename Lat<D>::State& s, uint64_t seed, uint32_t env, uint32_t episode, double p) {
    using L = Lat<D>;
    int r = 0;
    for (; r < MAX_RESET_ROUNDS; ++r) {
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            typename L::B x, z;
#pragma unroll
            for (int k = 0; k < L::W; ++k) {
                uint64_t ax = 0, az = 0;
                const int nb = (L::DD - 64 * k) < 64 ? (L::DD - 64 * k) : 64;
                for (int bit = 0; bit < nb; ++bit) {
                    const U4 w = draw(seed, env, episode, (uint32_t)r, DOMAIN_ERR,
                                      (uint32_t)(l * L::DD + 64 * k + bit));
                    const int err = u01(w.x) < p;
                    const uint32_t pauli = 1 + mulhi32(w.y, 3);
                    ax |= (uint64_t)(err & ((pauli == 1) | (pauli == 2))) << bit;
                    az |= (uint64_t)(err & (int)(pauli >> 1)) << bit;
                }
                x.w[k] = ax; z.w[k] = az;
            }
            s.x[l] = x; s.z[l] = z;
        }
        L::syndrome(s);
        if (s.v.any() || s.p.any()) return r + 1;
    }
    return r;
}

// env.reset with config "min_qubit_errors" = n > 0: exactly n errors on uniformly chosen distinct
// qubits, Pauli uniform (the fixed-n sampler, results/small_p_error_test.py:34-40; p_error is not
// used), redrawn until the syndrome is non-empty.  Selection sampling (Knuth's algorithm S) over the
// qubits in index order: qubit c is taken iff floor(w0 * (NQ - c) / 2^32) < n - taken -- one Philox
// draw per qubit, no permutation array.
template <int D>
TQ_HD int reset_lattice_n(typename Lat<D>::State& s, uint64_t seed, uint32_t env, uint32_t episode, int n_err) {
    using L = Lat<D>;
    int r = 0;
    for (; r < MAX_RESET_ROUNDS; ++r) {
        int taken = 0;
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            typename L::B x, z;
#pragma unroll
            for (int k = 0; k < L::W; ++k) {
                uint64_t ax = 0, az = 0;
                const int nb = (L::DD - 64 * k) < 64 ? (L::DD - 64 * k) : 64;
                for (int bit = 0; bit < nb; ++bit) {
                    const int c = l * L::DD + 64 * k + bit;
                    const U4 w = draw(seed, env, episode, (uint32_t)r, DOMAIN_NERR, (uint32_t)c);
                    const int err = (int)mulhi32(w.x, (uint32_t)(L::NQ - c)) < n_err - taken;
                    taken += err;
                    const uint32_t pauli = 1 + mulhi32(w.y, 3);
                    ax |= (uint64_t)(err & ((pauli == 1) | (pauli == 2))) << bit;
                    az |= (uint64_t)(err & (int)(pauli >> 1)) << bit;
                }
                x.w[k] = ax; z.w[k] = az;
            }
            s.x[l] = x; s.z[l] = z;
        }
        L::syndrome(s);
        if (s.v.any() || s.p.any()) return r + 1;
    }
    return r;
}

}  // namespace tq
