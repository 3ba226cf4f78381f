/*
 * CPU oracle (plain C) for the toric-code env hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product never links or calls it.
 *
 * It restates, function by function, the same algorithm as oracle/toric_oracle.py
 * (see that file's header for the reference citations and the Philox RNG contract):
 *   perspective / rotate / shift / transition / selection:
 *       src/util.py:46-150, src/numba/util.py:8-76, src/util_actor.py:223-264,
 *       src/numba/util_actor.py:11-107                (parity pinned by tests/golden)
 *   single-lattice env (gym_ToricCode, absent upstream):
 *       results/small_p_error_test.py:22-31,109-120, src/util.py:68-69,77-78,
 *       docs/toric_model.md:11,15, src/evaluation.py:97,175   (PARITY UNPINNED for
 *       reset()/RNG stream: the reference holds no recorded outputs and never seeds)
 *
 * The perspective stack is built the way the reference builds it: per lattice, per
 * hit, roll the whole (2,d,d) state, rotate it for layer-1 hits, append.  That makes
 * tor_actor_steps() a fair "port" CPU baseline of EnvSet.step +
 * generatePerspectiveBatch + generateTransitionParallel (Actor_mp.py:104-185).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TOR_DOMAIN_ERR 0u
#define TOR_DOMAIN_SEL 1u
#define TOR_DOMAIN_PERR 2u
#define TOR_MAX_RESET_ROUNDS 4096
#define TOR_MAX_D 21
#define TOR_MAX_CELLS (2 * TOR_MAX_D * TOR_MAX_D)

/* ------------------------------------------------------------------ Philox */
void tor_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline void draw(uint64_t seed, uint32_t env, uint32_t episode, uint32_t round,
                        uint32_t domain, uint32_t index, uint32_t out[4])
{
    uint32_t ctr[4] = { env, episode, round, (domain << 24) | index };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    tor_philox4x32(ctr, key, out);
}

static inline double u01(uint32_t w) { return (double)w * (1.0 / 4294967296.0); }
static inline uint32_t mulhi(uint32_t w, uint32_t n) { return (uint32_t)(((uint64_t)w * n) >> 32); }

/* ------------------------------------------------------------- lattice algebra */
/* createSyndromOpt: vertex = z0[i,j]^z0[i-1,j]^z1[i,j]^z1[i,j-1];
 *                   plaq   = x0[i,j]^x0[i,j+1]^x1[i,j]^x1[i+1,j]     (util.py:68-69,77-78) */
int tor_syndrome(const uint8_t *q, uint8_t *s, int d)
{
    const int dd = d * d;
    int defects = 0;
    for (int i = 0; i < d; ++i) {
        for (int j = 0; j < d; ++j) {
            int im = (i + d - 1) % d, ip = (i + 1) % d, jm = (j + d - 1) % d, jp = (j + 1) % d;
#define ZC(v) (((v) >> 1) & 1)
#define XC(v) ((((v) ^ ((v) >> 1))) & 1)
            uint8_t v = ZC(q[i * d + j]) ^ ZC(q[im * d + j]) ^ ZC(q[dd + i * d + j]) ^ ZC(q[dd + i * d + jm]);
            uint8_t p = XC(q[i * d + j]) ^ XC(q[i * d + jp]) ^ XC(q[dd + i * d + j]) ^ XC(q[dd + ip * d + j]);
            s[i * d + j] = v;
            s[dd + i * d + j] = p;
            defects += v + p;
        }
    }
    return defects;
}

/* evalGroundState (theory): even X- and Z-parity in both layers. */
int tor_eval_ground_state(const uint8_t *q, int d)
{
    const int dd = d * d;
    int bad = 0;
    for (int l = 0; l < 2; ++l) {
        int zp = 0, xp = 0;
        for (int c = 0; c < dd; ++c) { zp ^= ZC(q[l * dd + c]); xp ^= XC(q[l * dd + c]); }
        bad |= zp | xp;
    }
    return !bad;
}

/* env.reset(p_error): redraw until >= 1 defect (small_p_error_test.py:22-31,109-120). */
int tor_reset_one(uint64_t seed, uint32_t env, uint32_t episode, double p, int d,
                  uint8_t *q, uint8_t *s)
{
    const int nq = 2 * d * d;
    int r;
    for (r = 0; r < TOR_MAX_RESET_ROUNDS; ++r) {
        for (int c = 0; c < nq; ++c) {
            uint32_t w[4];
            draw(seed, env, episode, (uint32_t)r, TOR_DOMAIN_ERR, (uint32_t)c, w);
            q[c] = (u01(w[0]) < p) ? (uint8_t)(1 + mulhi(w[1], 3)) : 0;
        }
        if (tor_syndrome(q, s, d) > 0) return r + 1;
    }
    return r;
}

/* idx == NULL: all n lattices; p == NULL: p_default.  episodes/steps updated in place. */
void tor_reset_batch(uint64_t seed, int64_t first_env, int n, int d, const int32_t *idx, int nidx,
                     const double *p, double p_default, uint8_t *qubits, uint8_t *state,
                     uint32_t *episodes, uint32_t *steps)
{
    const int nq = 2 * d * d;
    const int m = idx ? nidx : n;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < m; ++k) {
        int e = idx ? idx[k] : k;
        if (e < 0 || e >= n) continue;
        tor_reset_one(seed, (uint32_t)(first_env + e), episodes[e], p ? p[k] : p_default, d,
                      qubits + (size_t)e * nq, state + (size_t)e * nq);
        episodes[e] += 1;
        steps[e] = 0;
    }
}

/* env.step over a batch (EnvSet.py:38-47): q ^= op, resyndrome, reward = delta defects
 * or terminal_reward when cleared (evaluation.py:97,175). */
void tor_step_batch(int n, int d, const int32_t *actions, double terminal_reward, uint8_t *qubits,
                    uint8_t *state, float *rewards, uint8_t *terminals, uint32_t *steps)
{
    const int nq = 2 * d * d, dd = d * d;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < n; ++e) {
        uint8_t *q = qubits + (size_t)e * nq, *s = state + (size_t)e * nq;
        const int32_t *a = actions + 4 * (size_t)e;
        int before = 0;
        for (int c = 0; c < nq; ++c) before += s[c];
        q[a[0] * dd + a[1] * d + a[2]] ^= (uint8_t)a[3];
        int after = tor_syndrome(q, s, d);
        terminals[e] = after == 0;
        rewards[e] = after == 0 ? (float)terminal_reward : (float)(before - after);
        steps[e] += 1;
    }
}

/* ------------------------------------------------- reference-shaped lattice ops */
/* np.roll(state, sh, axis=1) then np.roll(.., sw, axis=2) on a (2,d,d) grid. */
static void roll_state(const uint8_t *in, uint8_t *out, int d, int sh, int sw)
{
    const int dd = d * d;
    sh = ((sh % d) + d) % d;
    sw = ((sw % d) + d) % d;
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < d; ++r)
            for (int s = 0; s < d; ++s)
                out[c * dd + ((r + sh) % d) * d + (s + sw) % d] = in[c * dd + r * d + s];
}

/* rotate_state (util.py:87-94): plaquette rot90; vertex rot90 then roll +1 on axis 0. */
void tor_rotate_state(const uint8_t *in, uint8_t *out, int d)
{
    const int dd = d * d;
    for (int r = 0; r < d; ++r)
        for (int s = 0; s < d; ++s) {
            out[dd + r * d + s] = in[dd + s * d + (d - 1 - r)];      /* rot90: out[r,s] = in[s, d-1-r] */
            out[((r + 1) % d) * d + s] = in[s * d + (d - 1 - r)];     /* then roll +1 along axis 0 */
        }
}

static inline int hit0(const uint8_t *s, int d, int i, int j)
{
    const int dd = d * d;
    return s[i * d + j] | s[((i + 1) % d) * d + j] | s[dd + i * d + j] | s[dd + i * d + (j + d - 1) % d];
}
static inline int hit1(const uint8_t *s, int d, int i, int j)
{
    const int dd = d * d;
    return s[i * d + j] | s[i * d + (j + 1) % d] | s[dd + i * d + j] | s[dd + ((i + d - 1) % d) * d + j];
}

/* number of defect-adjacent qubits of one lattice (numba/util.py:48-53,62-67) */
int tor_persp_count_one(const uint8_t *s, int d)
{
    int n = 0;
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) n += (hit0(s, d, i, j) != 0) + (hit1(s, d, i, j) != 0);
    return n;
}

void tor_persp_count(int n, int d, const uint8_t *state, int32_t *counts, int64_t *offsets)
{
    const int nq = 2 * d * d;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < n; ++e) counts[e] = tor_persp_count_one(state + (size_t)e * nq, d);
    offsets[0] = 0;
    for (int e = 0; e < n; ++e) offsets[e + 1] = offsets[e] + counts[e];
}

/* generatePerspectiveOptimized for one lattice; out_u8/out_f32: either may be NULL. */
int tor_persp_write_one(const uint8_t *s, int d, uint8_t *out_u8, float *out_f32, int32_t *pos)
{
    const int nq = 2 * d * d, gs = d / 2;
    uint8_t rolled[TOR_MAX_CELLS], rot[TOR_MAX_CELLS];
    int n = 0;
    for (int layer = 0; layer < 2; ++layer)
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                if (!(layer ? hit1(s, d, i, j) : hit0(s, d, i, j))) continue;
                roll_state(s, rolled, d, gs - i, gs - j);
                const uint8_t *src = rolled;
                if (layer) { tor_rotate_state(rolled, rot, d); src = rot; }
                if (out_u8) memcpy(out_u8 + (size_t)n * nq, src, (size_t)nq);
                if (out_f32) for (int c = 0; c < nq; ++c) out_f32[(size_t)n * nq + c] = (float)src[c];
                if (pos) { pos[3 * n] = layer; pos[3 * n + 1] = i; pos[3 * n + 2] = j; }
                ++n;
            }
    return n;
}

/* generatePerspectiveBatch + concatenate (numba/util_actor.py:33-39,56-67) */
void tor_persp_write(int n, int d, const uint8_t *state, const int64_t *offsets, uint8_t *out_u8,
                     float *out_f32, int32_t *pos)
{
    const int nq = 2 * d * d;
#pragma omp parallel for schedule(dynamic, 64)
    for (int e = 0; e < n; ++e) {
        size_t o = (size_t)offsets[e];
        tor_persp_write_one(state + (size_t)e * nq, d, out_u8 ? out_u8 + o * nq : NULL,
                            out_f32 ? out_f32 + o * nq : NULL, pos ? pos + 3 * o : NULL);
    }
}

/* generateTransitionParallel (util_actor.py:223-264): shift both states to centre the
 * acted qubit, rotate both for layer 1, rewrite position to (layer, gs, gs). */
void tor_transition(int n, int d, const int32_t *actions, const uint8_t *state, const uint8_t *next_state,
                    uint8_t *persp, int32_t *act_out, uint8_t *next_persp)
{
    const int nq = 2 * d * d, gs = d / 2;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < n; ++e) {
        const int32_t *a = actions + 4 * (size_t)e;
        uint8_t t0[TOR_MAX_CELLS], t1[TOR_MAX_CELLS];
        roll_state(state + (size_t)e * nq, t0, d, gs - a[1], gs - a[2]);
        roll_state(next_state + (size_t)e * nq, t1, d, gs - a[1], gs - a[2]);
        if (a[0] == 1) {
            tor_rotate_state(t0, persp + (size_t)e * nq, d);
            tor_rotate_state(t1, next_persp + (size_t)e * nq, d);
        } else {
            memcpy(persp + (size_t)e * nq, t0, (size_t)nq);
            memcpy(next_persp + (size_t)e * nq, t1, (size_t)nq);
        }
        act_out[4 * e] = a[0]; act_out[4 * e + 1] = gs; act_out[4 * e + 2] = gs; act_out[4 * e + 3] = a[3];
    }
}

/* _selectActionBatch_prime (numba/util_actor.py:69-107) with Philox draws. */
void tor_select_action(int n, const float *q_table, const int64_t *offsets, const int32_t *positions,
                       const double *eps, uint64_t seed, int64_t first_env, const uint32_t *episodes,
                       const uint32_t *steps, int32_t *actions, float *qv)
{
#pragma omp parallel for schedule(static)
    for (int e = 0; e < n; ++e) {
        int64_t lo = offsets[e], hi = offsets[e + 1];
        int32_t *a = actions + 4 * (size_t)e;
        float *qo = qv + 3 * (size_t)e;
        if (hi == lo) { a[0] = a[1] = a[2] = a[3] = 0; qo[0] = qo[1] = qo[2] = 0.f; continue; }
        uint32_t w[4];
        draw(seed, (uint32_t)(first_env + e), episodes[e], steps[e], TOR_DOMAIN_SEL, 0, w);
        int64_t p; int op;
        if ((1.0 - eps[e]) > u01(w[0])) {
            int64_t best = 0; float bv = q_table[3 * lo];
            for (int64_t k = 1; k < 3 * (hi - lo); ++k)
                if (q_table[3 * lo + k] > bv) { bv = q_table[3 * lo + k]; best = k; }
            p = best / 3; op = (int)(best % 3);
        } else {
            p = mulhi(w[1], (uint32_t)(hi - lo)); op = (int)mulhi(w[2], 3);
        }
        a[0] = positions[3 * (lo + p)]; a[1] = positions[3 * (lo + p) + 1]; a[2] = positions[3 * (lo + p) + 2];
        a[3] = op + 1;
        qo[0] = q_table[3 * (lo + p)]; qo[1] = q_table[3 * (lo + p) + 1]; qo[2] = q_table[3 * (lo + p) + 2];
    }
}

double tor_perror_draw(uint64_t seed, uint32_t env, uint32_t episode, double p_start, double p_roof)
{
    uint32_t w[4];
    draw(seed, env, episode, 0, TOR_DOMAIN_PERR, 0, w);
    double span = p_roof - p_start;
    double t = span * u01(w[0]);
    return p_start + t;
}

/* ---------------------------------------------------------------- actor loop */
/* One pass of the hot path over the batch, `n_steps` times, in the call order of
 * Actor_mp.py:104-185: perspectives (f32 stack, as fed to the NN) -> epsilon-greedy
 * select (q_table = zeros: the NN is out of scope) -> step -> transition -> reset of
 * terminal / timed-out lattices at p_reset.  Scratch is allocated once up front.
 * Returns the total number of perspectives produced; checksum[0] accumulates a
 * digest of the outputs so the work cannot be optimised away and runs can be
 * compared with the HIP path: sum over steps of (sum(rewards) + 3*#terminal + P). */
int64_t tor_actor_steps(uint64_t seed, int64_t first_env, int n, int d, int n_steps, double eps_all,
                        double p_reset, double terminal_reward, int max_steps_per_episode,
                        uint8_t *qubits, uint8_t *state, uint32_t *episodes, uint32_t *steps,
                        double *checksum)
{
    const int nq = 2 * d * d;
    int32_t *counts = malloc(sizeof(int32_t) * (size_t)n);
    int64_t *offsets = malloc(sizeof(int64_t) * ((size_t)n + 1));
    size_t cap = (size_t)n * nq;                       /* worst case: every qubit is a hit */
    float *persp = malloc(sizeof(float) * cap * nq);
    int32_t *pos = malloc(sizeof(int32_t) * 3 * cap);
    float *qtab = calloc(3 * cap, sizeof(float));
    double *eps = malloc(sizeof(double) * (size_t)n);
    int32_t *actions = malloc(sizeof(int32_t) * 4 * (size_t)n);
    float *qv = malloc(sizeof(float) * 3 * (size_t)n);
    uint8_t *prev = malloc((size_t)n * nq);
    uint8_t *tp = malloc((size_t)n * nq), *tnp = malloc((size_t)n * nq);
    int32_t *tact = malloc(sizeof(int32_t) * 4 * (size_t)n);
    float *rewards = malloc(sizeof(float) * (size_t)n);
    uint8_t *terminals = malloc((size_t)n);
    int32_t *ridx = malloc(sizeof(int32_t) * (size_t)n);
    int64_t total_p = 0;
    double cs = 0.0;
    for (int e = 0; e < n; ++e) eps[e] = eps_all;
    for (int t = 0; t < n_steps; ++t) {
        tor_persp_count(n, d, state, counts, offsets);
        tor_persp_write(n, d, state, offsets, NULL, persp, pos);
        total_p += offsets[n];
        tor_select_action(n, qtab, offsets, pos, eps, seed, first_env, episodes, steps, actions, qv);
        memcpy(prev, state, (size_t)n * nq);
        tor_step_batch(n, d, actions, terminal_reward, qubits, state, rewards, terminals, steps);
        tor_transition(n, d, actions, prev, state, tp, tact, tnp);
        int nr = 0;
        for (int e = 0; e < n; ++e) {
            cs += rewards[e] + 3.0 * terminals[e];
            if (terminals[e] || steps[e] > (uint32_t)max_steps_per_episode) ridx[nr++] = e;
        }
        cs += (double)offsets[n];
        if (nr) tor_reset_batch(seed, first_env, n, d, ridx, nr, NULL, p_reset, qubits, state, episodes, steps);
    }
    if (checksum) *checksum = cs;
    free(counts); free(offsets); free(persp); free(pos); free(qtab); free(eps); free(actions); free(qv);
    free(prev); free(tp); free(tnp); free(tact); free(rewards); free(terminals); free(ridx);
    return total_p;
}

int tor_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void tor_set_threads(int t)
{
#ifdef _OPENMP
    omp_set_num_threads(t);
#else
    (void)t;
#endif
}
