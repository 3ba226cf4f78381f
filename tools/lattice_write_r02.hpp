// The round-1/2 stack-write kernel: ONE WAVEFRONT PER LATTICE (set-up and stores in the same wave).  Replaced in
// the library by the producer / storer form (toric-rl-decoder_amd/csrc/stream_write.hpp); kept here, unchanged, only
// as the A/B reference of tools/stream_bench.hip (bit-exact against it; profiles/r03_stack_write_ab.txt).
#pragma once
#include "kernels.hpp"

namespace tq {

// Per-wave LDS of the stack write: the lattice's perspective bitstream (lattice.hpp, PStream) and the
// small tables it is built from.
template <int D>
struct PerspLds {
    using S = PStream<D>;
    static constexpr int BITS_DW = (S::MAX_DW + 3) & ~3;
    static constexpr int NQP = (Lat<D>::NQ + 7) & ~7;
    __attribute__((aligned(16))) uint32_t bits[BITS_DW];       // bit pidx*NQ + cell = element `cell` of perspective pidx
    uint64_t rr[4][D][Lat<D>::W];                              // V, P, rot V, rot P rolled by every row amount
    uint64_t low[D][Lat<D>::W];                                // lowcols(k): destination columns [0,k) of a column roll
    uint32_t hpos[NQP];                                        // k-th hit -> its position, packed layer | row << 8 | col << 16
};

// vp = V/P planes: V word k of lattice e at vp[(0*W+k)*N+e], P at vp[(1*W+k)*N+e].
//
// One wavefront per lattice (the hardware dispatcher balances the variable-size lattices).
//   1. wave-uniform part (scalar unit): plane words, offset, hit masks, counts.
//   2. bit-parallel construction of the lattice's perspective bitstream in LDS (PStream, lattice.hpp):
//      the rotated planes by __ballot, a table of row-rolled planes (one lane per entry), then one
//      lane per hit: two masked column rolls and five-to-nine ds_or_b32.  No lookup table, no
//      global vector load anywhere in this kernel (vector loads share the in-order vmcnt with the
//      stores that follow; the per-lattice inputs arrive through scalar loads).
//   3. expansion: a lane's 16-byte store needs VEC consecutive stream bits, and since one wave
//      instruction advances the stream by 64*VEC bits (a multiple of 32) the lane's bit phase is
//      loop-invariant: one ds_read2_b32, one 64-bit shift, the bit->element expansion, one
//      global_store_dwordx4 per KiB written.
// Ownership rule for the output: the stack is cut into 128-byte lines of the address space and a
// line is written -- whole -- by the wave of the lattice that contains the line's FIRST element.
// Two waves (usually on different XCDs, whose L2s are not coherent) therefore never write parts
// of one line; measured +6.5 % over byte-exact segment ownership (tools/membench3.hip).
//   * lines entirely inside the lattice's segment: the fast loop;
//   * the last owned line, when the segment ends inside it: lanes 0..31 store one dword each; its
//     trailing elements belong to the following lattice(s) and are resolved from their bit-planes.
// The leading elements of a segment that sit in a line begun by an earlier lattice are written by
// that lattice's wave, by the same rule.
template <int D, typename OutT>
__device__ __forceinline__ void persp_lattice(int64_t e, const uint64_t* __restrict__ vp, int64_t N,
                                              const int64_t* __restrict__ offsets, OutT* __restrict__ out,
                                              int32_t* __restrict__ pos, int64_t capacity, PerspLds<D>& t,
                                              int* __restrict__ err, int lane, int64_t e_begin, int64_t e_end) {
    using L = Lat<D>;
    using S = PStream<D>;
    using Enc = OutEnc<OutT>;
    constexpr int DD = L::DD, NQ = L::NQ, W = L::W;
    constexpr int VEC = 16 / (int)sizeof(OutT);              // elements per 16-byte lane store
    constexpr int EPW = 32 / Enc::BITS;                      // elements per dword
    constexpr int LE = 128 / (int)sizeof(OutT);              // elements per 128-byte line
    typename L::B v, p, e0, e1;
#pragma unroll
    for (int k = 0; k < W; ++k) { v.w[k] = vp[(int64_t)k * N + e]; p.w[k] = vp[((int64_t)W + k) * N + e]; }
    L::hit_masks(v, p, e0, e1);
    const int n0 = e0.popc();
    const int n = n0 + e1.popc();
    if (n == 0) return;
    // the stack written is that of the lattices [e_begin, e_end): perspective 0 of `out` is the first of e_begin
    const int64_t off0 = offsets[e_begin];
    const int64_t off = offsets[e] - off0;
    // What the mixed last line needs from outside this lattice -- the planes of the lattice behind it and the
    // length of the stack -- is fetched here, in the same scalar-load round trip as the lattice's own data,
    // not when the wave is about to finish.
    const int64_t p_total = offsets[e_end] - off0;
    const bool has_next = e + 1 < e_end;
    typename L::B v_next, p_next;
#pragma unroll
    for (int k = 0; k < W; ++k) {
        v_next.w[k] = has_next ? vp[(int64_t)k * N + e + 1] : 0ull;
        p_next.w[k] = has_next ? vp[((int64_t)W + k) * N + e + 1] : 0ull;
    }
    if (off + n > capacity) { if (lane == 0) atomicOr(err, ERR_CAPACITY); return; }

    // ---- tables: rotated planes (ballot), row-rolled planes, column masks, hit list; zeroed stream
    typename L::B rv, rp;
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const int o = 64 * k + lane;
        const bool in = o < DD;
        const int oc = in ? o : 0;
        rv.w[k] = __ballot(in && v.get(S::rot_src_v(oc)));
        rp.w[k] = __ballot(in && p.get(S::rot_src_p(oc)));
    }
    if (lane < 4 * D) {
        const int sel = lane / D, k = lane - sel * D;
        typename L::B src;
#pragma unroll
        for (int w = 0; w < W; ++w) src.w[w] = sel == 0 ? v.w[w] : (sel == 1 ? p.w[w] : (sel == 2 ? rv.w[w] : rp.w[w]));
        const typename L::B r = (src.shl(k * D) | src.shr(DD - k * D)) & L::full();      // roll_rows, branch-free (k = 0: shr(DD) = 0)
#pragma unroll
        for (int w = 0; w < W; ++w) t.rr[sel][k][w] = r.w[w];
    }
    if (lane < D) {
        const typename L::B m = L::lowcols(lane);
#pragma unroll
        for (int w = 0; w < W; ++w) t.low[lane][w] = m.w[w];
    }
    for (int c = lane; c < NQ; c += 64) {
        const int l = c >= DD, bit = c - l * DD;
        if (l ? e1.get(bit) : e0.get(bit)) {
            const int row = bit / D, col = bit - row * D;
            t.hpos[l ? n0 + e1.rank(bit) : e0.rank(bit)] = (uint32_t)l | ((uint32_t)row << 8) | ((uint32_t)col << 16);
        }
    }
    {
        const int nd4 = ((n * NQ + 31) / 32 + 2 + 3) / 4;     // <= BITS_DW / 4
        uint4* b4 = reinterpret_cast<uint4*>(t.bits);
        for (int i = lane; i < nd4; i += 64) b4[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    wave_lds_sync();

    // ---- one lane per hit: its perspective as two bit-planes, OR-ed into the stream
    for (int k = lane; k < n; k += 64) {
        const uint32_t hp = t.hpos[k];
        const int layer = (int)(hp & 255u), i = (int)((hp >> 8) & 255u), j = (int)(hp >> 16);
        int rs, cs;
        S::hit_shifts(layer, i, j, rs, cs);
        typename L::B a, c, low;
#pragma unroll
        for (int w = 0; w < W; ++w) { a.w[w] = t.rr[2 * layer][rs][w]; c.w[w] = t.rr[2 * layer + 1][rs][w]; low.w[w] = t.low[cs][w]; }
        const typename L::B ov = S::roll_cols_masked(a, cs, low), op = S::roll_cols_masked(c, cs, low);
        S::emit(k, ov, op, [&](int idx, uint32_t val) { atomicOr(&t.bits[idx], val); });
    }
    wave_lds_sync();

    // positions (P,3) i32: (layer,row,col) of each hit.  Same ownership rule on its own 128-byte
    // lines (32 dwords): whole lines inside the lattice's [plo, phi) by 16-byte stores here, the
    // mixed last line further down together with the stack's.
    const int64_t plo = off * 3, phi = plo + 3 * n;
    const int64_t PA = (plo + 31) / 32 * 32, PF = phi / 32 * 32;
    if (pos && PF > PA) {
        const int n_g = (int)((PF - PA) / 4);
        int4* __restrict__ pseg = reinterpret_cast<int4*>(pos + PA);
        for (int g = lane; g < n_g; g += 64) {
            const int k0 = (int)(PA - plo) + 4 * g;
            int o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + j, hidx = k / 3;
                o[j] = (int)((t.hpos[hidx] >> (8 * (k - 3 * hidx))) & 255u);
            }
            pseg[g] = make_int4(o[0], o[1], o[2], o[3]);
        }
    }

    const int total = n * NQ;
    const int64_t lo = off * NQ, hi = lo + total;            // the lattice's segment, in elements
    const int64_t A = (lo + LE - 1) / LE * LE;               // first line start >= lo
    const int64_t F = hi / LE * LE;                          // last line start <= hi

    // ---- mixed last lines: the stack line [F, F+LE) on lanes 0..31 and the positions line
    // [PF, PF+32) on lanes 32..63, each present when the lattice's range ends inside a line that
    // starts in it.  Their trailing elements belong to the lattices that follow: one wave-uniform
    // walk over those lattices serves both.
    // This is done BEFORE the main loop so that the wave's last instructions are its stores: it retires as soon
    // as they are issued and its slot goes to the next lattice.
    const bool s_mixed = F >= lo && F < hi;
    const bool p_mixed = pos != nullptr && PF >= plo && PF < phi;
    if (s_mixed || p_mixed) {
        // all positions below are relative to the start of the mixed line (stack: elements, positions: dwords)
        const int64_t p_cap = p_total < capacity ? p_total : capacity;   // perspectives that may be written
        const int64_t s_room = p_cap * NQ - F, p_room = p_cap * 3 - PF;  // what lies inside the stack from the line start on
        const int s_lim = s_room < LE ? (int)s_room : LE, p_lim = p_room < 32 ? (int)p_room : 32;
        const int s_own = s_mixed ? (int)(hi - F) : LE;                  // this lattice's part of the line
        const int p_own = p_mixed ? (int)(phi - PF) : 32;
        const bool s_lane = s_mixed && lane < 32, p_lane = p_mixed && lane >= 32;
        const int xr0 = lane * EPW;                          // stack: this lane's dword = elements xr0 .. xr0+EPW-1 of the line
        const int yr = lane - 32;                            // positions: this lane's dword of the line
        uint32_t word = 0;
        int pval = 0;
        if (s_lane) {
#pragma unroll
            for (int j = 0; j < EPW; ++j) {                  // own elements
                if (xr0 + j < s_own) {
                    const uint32_t b = S::window(t.bits, (uint32_t)(F - lo) + (uint32_t)(xr0 + j)) & 1u;
                    word |= ((0u - b) & Enc::ONE) << (j * Enc::BITS);
                }
            }
        }
        if (p_lane && yr < p_own) {
            const int k = (int)(PF - plo) + yr, hidx = k / 3;
            pval = (int)((t.hpos[hidx] >> (8 * (k - 3 * hidx))) & 255u);
        }
        int64_t e2 = e + 1;
        int spos = s_own, ppos = p_own;
        while (e2 < e_end && ((s_mixed && spos < s_lim) || (p_mixed && ppos < p_lim))) {
            typename L::B v2 = v_next, p2 = p_next, f0, f1;
            if (e2 != e + 1) {                               // beyond the prefetched neighbour (tiny or empty lattices only)
#pragma unroll
                for (int k = 0; k < W; ++k) { v2.w[k] = vp[(int64_t)k * N + e2]; p2.w[k] = vp[((int64_t)W + k) * N + e2]; }
            }
            L::hit_masks(v2, p2, f0, f1);
            const int n2 = f0.popc() + f1.popc();
            const int send = spos + n2 * NQ, pend = ppos + 3 * n2;
#pragma unroll
            for (int j = 0; j < EPW; ++j) {
                // one k-th-hit search serves both halves of the wave: stack lanes ask for the hit of their
                // element's perspective, position lanes (first pass only) for the hit of their dword
                const int xr = xr0 + j;
                const bool s_act = s_lane && xr >= spos && xr < send;
                const bool p_act = j == 0 && p_lane && yr >= ppos && yr < pend;
                if (s_act || p_act) {
                    const int rel = s_act ? xr - spos : yr - ppos;                // < LE / < 32
                    const int kq = s_act ? rel / NQ : rel / 3;
                    const int h = kth_hit<D>(f0, f1, kq);
                    const int hl = h >= DD, hrem = h - hl * DD, hi_ = hrem / D, hj = hrem - hi_ * D;
                    if (s_act) {
                        const int cell = rel - kq * NQ;
                        const int cc = cell >= DD, crem = cell - cc * DD, cr = crem / D, cs_ = crem - cr * D;
                        const int src = L::persp_src(hl, hi_, hj, cc, cr, cs_);
                        const uint32_t b = (uint32_t)(src >= DD ? p2.get(src - DD) : v2.get(src));
                        word |= ((0u - b) & Enc::ONE) << (j * Enc::BITS);
                    } else {
                        const int comp = rel - 3 * kq;
                        pval = comp == 0 ? hl : (comp == 1 ? hi_ : hj);
                    }
                }
            }
            spos = send;
            ppos = pend;
            ++e2;
        }
        if (s_lane) {
            if (xr0 + EPW <= s_lim) {
                reinterpret_cast<uint32_t*>(out)[(F + xr0) / EPW] = word;
            } else {
#pragma unroll
                for (int j = 0; j < EPW; ++j)                // the stack ends inside this dword
                    if (xr0 + j < s_lim) {
                        if (Enc::BITS == 16) reinterpret_cast<uint16_t*>(out)[F + xr0 + j] = (uint16_t)(word >> (16 * j));
                        else if (Enc::BITS == 8) reinterpret_cast<uint8_t*>(out)[F + xr0 + j] = (uint8_t)(word >> (8 * j));
                    }
            }
        }
        if (p_lane && yr < p_lim) pos[PF + yr] = pval;
    }

    // ---- whole lines inside the segment: [A, F)
    if (F > A) {
        // Lane `lane` writes 16-byte groups lane, lane+64, ... of [A, F); group g holds the stream bits
        // rel0 + g*VEC ... : the dword index advances by 2*VEC per iteration, the bit phase never changes.
        const int n_groups = (int)((F - A) / VEC);
        char* __restrict__ seg = reinterpret_cast<char*>(out + A);            // wave-uniform, 128-byte aligned
        const uint32_t rel0 = (uint32_t)(A - lo) + (uint32_t)lane * VEC;
        const uint32_t ph = rel0 & 31u;
        const uint32_t* __restrict__ bp = t.bits + (rel0 >> 5);
        uint32_t w0 = 0, w1 = 0;
        if (lane < n_groups) { w0 = bp[0]; w1 = bp[1]; }
        for (int gi = lane; gi < n_groups; gi += 64) {
            const uint32_t wb = (uint32_t)(((((uint64_t)w1) << 32) | w0) >> ph);
            bp += 2 * VEC;
            if (gi + 64 < n_groups) { w0 = bp[0]; w1 = bp[1]; }              // next window is in flight while this one is stored
            *reinterpret_cast<u32x4*>(seg + (uint32_t)gi * 16u) = expand_bits<OutT>(wb);
        }
    }
}

// SGPRs capped at 80: up to 80 a CU admits 8 of these workgroups (32 waves); the d >= 9 instantiations would
// otherwise take 87-98 and lose one or two (MI355X_MICROARCH.md, residency), and the bandwidth of this kernel
// follows the number of waves that are storing.
template <int D, typename OutT, int THREADS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_num_sgpr(80))) void k_persp_write(const uint64_t* __restrict__ vp, int64_t N,
                                                         const int64_t* __restrict__ offsets, OutT* __restrict__ out,
                                                         int32_t* __restrict__ pos, int64_t capacity, int* __restrict__ err,
                                                         int64_t e_begin, int64_t e_end) {
    constexpr int WAVES = THREADS / 64;
    __shared__ PerspLds<D> tables[WAVES];
    // the wave index is made provably uniform so that the lattice id, its plane words and its
    // offset live in SGPRs (scalar loads) and the hit masks are computed on the scalar unit
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t e = e_begin + (int64_t)blockIdx.x * WAVES + wave;
    if (e < e_end) persp_lattice<D, OutT>(e, vp, N, offsets, out, pos, capacity, tables[wave], err, lane, e_begin, e_end);
}

}  // namespace tq
