"""CPU oracle for the toric-code env hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``toric-rl-decoder_amd``) never does:
it talks to the HIP library through the C-ABI and fails loudly without it.

What it restates (all citations are relative to the upstream reference tree):

* perspective / rotate / shift / transition / action selection -- the part of the
  path whose source IS in the reference: ``src/util.py:46-150``,
  ``src/numba/util.py:8-76``, ``src/util_actor.py:223-264``,
  ``src/numba/util_actor.py:11-107``, ``src/EnvSet.py:4-51``,
  ``src/Actor_mp.py:104-185``.  Parity of these functions is PINNED: they are
  checked against the imported reference functions by ``tests/golden/make_golden.py``
  (authoring container only) and the resulting vectors are committed under
  ``tests/golden/``.
* the single-lattice env (``gym_ToricCode``: reset / step / createSyndromOpt /
  isTerminalState / evalGroundState) -- an un-vendored git submodule
  (``.gitmodules:1-3`` -> github.com/Lindeby/gym_ToricCode, pinned commit unknown,
  absent from the tree).  Its algorithm is restated from the reference's own call
  sites and in-repo copies: sampler ``results/small_p_error_test.py:22-31``,
  reset-until-non-terminal ``:109-120``, adjacency tests ``src/util.py:68-69,77-78``,
  Pauli encoding / rule table ``docs/toric_model.md:11,15``, reward / terminal
  ``src/evaluation.py:97,110,175``.  The reference holds no recorded outputs for
  this half and never seeds its RNG, so for reset()/RNG stream: PARITY UNPINNED
  (bit-exactness is defined against this oracle under the Philox contract below).

Two forms are kept side by side and tested against each other:

* ``*_ref`` functions: the reference's algorithmic shape (per-lattice python loop,
  ``np.roll`` / ``np.rot90``, int64, fresh allocations).  These are what gets
  compared with the imported reference and timed as the "reference-shaped" numpy
  CPU baseline.
* batch closed forms (index arithmetic over the whole batch) used to check the
  HIP kernels at thousands of lattices in seconds.

RNG contract (shared bit-for-bit with the HIP kernels and oracle/toric_oracle.c)
-------------------------------------------------------------------------------
Philox4x32-10, key = (seed & 0xffffffff, seed >> 32), counter =
(env_id, episode, round_or_step, domain << 24 | index):

* DOMAIN_ERR  (0): reset round ``r`` of episode ``e`` draws, for qubit index
  ``q = layer*d*d + row*d + col``, words (w0, w1, _, _) from counter
  (env, e, r, q):  ``u = w0 * 2**-32`` (exact in f64), error iff ``u < p_error``,
  pauli = ``1 + ((w1 * 3) >> 32)``; rounds repeat until the syndrome is non-empty
  (at most MAX_RESET_ROUNDS rounds).
* DOMAIN_SEL  (1): epsilon-greedy draw of step ``t`` of episode ``e``: counter
  (env, e, t, 1 << 24): greedy iff ``(1 - eps) > w0 * 2**-32``; otherwise
  perspective ``(w1 * n) >> 32`` and op ``1 + ((w2 * 3) >> 32)``.
* DOMAIN_PERR (2): the caller's 'random' p_error strategy (``Actor_mp.py:176-180``)
  for the reset that STARTS episode ``e``: counter (env, e, 0, 2 << 24):
  ``p = start + (roof - start) * (w0 * 2**-32)`` evaluated in f64 without FMA.
"""
import numpy as np

DOMAIN_ERR = 0
DOMAIN_SEL = 1
DOMAIN_PERR = 2
DOMAIN_SEL_CALL = 3
DOMAIN_NERR = 4
MAX_RESET_ROUNDS = 4096
TERMINAL_REWARD = 100.0  # evaluation.py:175, Learner_mp.py:151 (clamp +-100)

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


# --------------------------------------------------------------------------- RNG
def philox4x32(c0, c1, c2, c3, k0, k1, rounds=10):
    """Philox4x32-R (Salmon et al., SC'11) vectorised over numpy arrays.

    All arguments broadcast; returns four uint32 arrays.
    """
    c0, c1, c2, c3 = np.broadcast_arrays(*[np.asarray(x, dtype=np.uint64) & _MASK
                                           for x in (c0, c1, c2, c3)])
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(rounds):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> _S32, p0 & _MASK
        hi1, lo1 = p1 >> _S32, p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0)
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32),
            c2.astype(np.uint32), c3.astype(np.uint32))


def _key(seed):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return seed & 0xFFFFFFFF, seed >> 32


def _mulhi(w, n):
    """(w * n) >> 32 for uint32 words and small n (array or scalar)."""
    return ((w.astype(np.uint64) * np.asarray(n, dtype=np.uint64)) >> _S32).astype(np.int64)


def _u01(w):
    """uint32 word -> f64 in [0,1) exactly (w * 2**-32)."""
    return w.astype(np.float64) * (1.0 / 4294967296.0)


# ---------------------------------------------------------------- lattice algebra
def syndrome(qubits):
    """createSyndromOpt restated (absent upstream; geometry from util.py:68-69,77-78).

    qubits: (..., 2, d, d) Pauli codes I=0 X=1 Y=2 Z=3 (docs/toric_model.md:11).
    Returns (..., 2, d, d) uint8, [0]=vertex matrix, [1]=plaquette matrix
    (util.py:63-64).
      vertex[i,j]    = z0[i,j] ^ z0[i-1,j] ^ z1[i,j] ^ z1[i,j-1]
      plaquette[i,j] = x0[i,j] ^ x0[i,j+1] ^ x1[i,j] ^ x1[i+1,j]
    """
    q = np.asarray(qubits).astype(np.uint8)
    z = (q >> 1) & 1              # Y or Z
    x = (q ^ (q >> 1)) & 1        # X or Y
    z0, z1 = z[..., 0, :, :], z[..., 1, :, :]
    x0, x1 = x[..., 0, :, :], x[..., 1, :, :]
    vertex = z0 ^ np.roll(z0, 1, axis=-2) ^ z1 ^ np.roll(z1, 1, axis=-1)
    plaq = x0 ^ np.roll(x0, -1, axis=-1) ^ x1 ^ np.roll(x1, -1, axis=-2)
    return np.stack((vertex, plaq), axis=-3).astype(np.uint8)


def is_terminal(state):
    """isTerminalState: no excitation left (small_p_error_test.py:116)."""
    s = np.asarray(state)
    return ~s.reshape(s.shape[:-3] + (-1,)).any(axis=-1)


def eval_ground_state(qubits):
    """evalGroundState restated from theory (no in-repo source).

    With an empty syndrome and odd d the state is in the ground state iff the
    parity of X-components and of Z-components is even in layer 0 and in layer 1
    (logical operators along the two non-contractible cycles).
    """
    q = np.asarray(qubits).astype(np.uint8)
    z = (q >> 1) & 1
    x = (q ^ (q >> 1)) & 1
    zp = z.reshape(z.shape[:-2] + (-1,)).sum(axis=-1) & 1   # (..., 2)
    xp = x.reshape(x.shape[:-2] + (-1,)).sum(axis=-1) & 1
    return ~((zp | xp).any(axis=-1))


def sample_errors(seed, env_ids, episodes, rounds, p_errors, d):
    """One depolarizing draw round for a batch (small_p_error_test.py:22-31).

    u < p -> error (u > p and u == p leave the qubit untouched for p < 1/3, :24-31);
    Pauli uniform in {1,2,3} (:28).  Returns (n, 2, d, d) uint8 Pauli codes.
    """
    env_ids = np.asarray(env_ids, dtype=np.uint64).reshape(-1, 1)
    episodes = np.asarray(episodes, dtype=np.uint64).reshape(-1, 1)
    rounds = np.asarray(rounds, dtype=np.uint64).reshape(-1, 1)
    p = np.asarray(p_errors, dtype=np.float64).reshape(-1, 1)
    nq = 2 * d * d
    q = np.arange(nq, dtype=np.uint64).reshape(1, -1) | np.uint64(DOMAIN_ERR << 24)
    k0, k1 = _key(seed)
    w0, w1, _, _ = philox4x32(env_ids, episodes, rounds, q, k0, k1)
    err = _u01(w0) < p
    pauli = 1 + _mulhi(w1, 3)
    return np.where(err, pauli, 0).astype(np.uint8).reshape(-1, 2, d, d)


def sample_n_errors(seed, env_ids, episodes, rounds, n_err, d):
    """One draw round of the fixed-n sampler (config "min_qubit_errors" = n > 0): exactly n errors on
    uniformly chosen distinct qubits, Pauli uniform -- generateNRandomErrors,
    results/small_p_error_test.py:34-40 (there: n Paulis shuffled over the 2*d*d positions).
    Selection sampling in qubit index order with one Philox draw per qubit (DOMAIN_NERR): qubit c
    is taken iff (w0 * (NQ - c)) >> 32 < n - taken.  PARITY UNPINNED like the rest of the env half."""
    env_ids = np.asarray(env_ids, dtype=np.uint64).reshape(-1, 1)
    episodes = np.asarray(episodes, dtype=np.uint64).reshape(-1, 1)
    rounds = np.asarray(rounds, dtype=np.uint64).reshape(-1, 1)
    nq = 2 * d * d
    q = np.arange(nq, dtype=np.uint64).reshape(1, -1) | np.uint64(DOMAIN_NERR << 24)
    k0, k1 = _key(seed)
    w0, w1, _, _ = philox4x32(env_ids, episodes, rounds, q, k0, k1)
    pauli = 1 + _mulhi(w1, 3)
    out = np.zeros(w0.shape, np.uint8)
    taken = np.zeros(w0.shape[0], np.int64)
    for c in range(nq):
        err = _mulhi(w0[:, c], nq - c) < (int(n_err) - taken)
        taken += err
        out[:, c] = np.where(err, pauli[:, c], 0)
    return out.reshape(-1, 2, d, d)


def reset_lattices(seed, env_ids, episodes, p_errors, d, min_errors=0):
    """env.reset(p_error) for a batch: redraw until >= 1 defect.

    Evidence for reset-until-non-terminal: small_p_error_test.py:109-120 and
    tests/test_select_action.py:17-26 (would crash on an empty syndrome).
    Returns qubits (n,2,d,d) u8, state (n,2,d,d) u8.
    """
    env_ids = np.asarray(env_ids, dtype=np.int64).reshape(-1)
    n = env_ids.shape[0]
    episodes = np.broadcast_to(np.asarray(episodes, dtype=np.int64), (n,))
    p = np.broadcast_to(np.asarray(p_errors, dtype=np.float64), (n,))
    qubits = np.zeros((n, 2, d, d), np.uint8)
    state = np.zeros((n, 2, d, d), np.uint8)
    todo = np.arange(n)
    for r in range(MAX_RESET_ROUNDS):
        if todo.size == 0:
            break
        if min_errors > 0:
            qb = sample_n_errors(seed, env_ids[todo], episodes[todo], r, min_errors, d)
        else:
            qb = sample_errors(seed, env_ids[todo], episodes[todo], r, p[todo], d)
        st = syndrome(qb)
        qubits[todo] = qb
        state[todo] = st
        todo = todo[is_terminal(st)]
    return qubits, state


def step_lattices(qubits, state, actions, terminal_reward=TERMINAL_REWARD):
    """env.step for a batch (EnvSet.py:38-47 loop, one action per lattice).

    action = [layer,row,col,op]; Pauli product mod phase = XOR of codes
    (docs/toric_model.md:15).  reward = sum(state) - sum(next_state), or
    ``terminal_reward`` when next_state is empty (evaluation.py:97,175).
    Returns new qubits, next_state, rewards f64, terminals bool.
    """
    qubits = np.array(qubits, dtype=np.uint8, copy=True)
    a = np.asarray(actions).astype(np.int64).reshape(-1, 4)
    n = qubits.shape[0]
    idx = np.arange(n)
    qubits[idx, a[:, 0], a[:, 1], a[:, 2]] ^= a[:, 3].astype(np.uint8)
    nxt = syndrome(qubits)
    before = np.asarray(state).reshape(n, -1).sum(axis=1).astype(np.int64)
    after = nxt.reshape(n, -1).sum(axis=1).astype(np.int64)
    term = after == 0
    rew = np.where(term, float(terminal_reward), (before - after).astype(np.float64))
    return qubits, nxt, rew, term


# ------------------------------------------------- reference-shaped perspective ops
def rotate_state_ref(state):
    """util.py:87-94 / numba/util.py:17-25."""
    vertex_matrix = state[0, :, :]
    plaquette_matrix = state[1, :, :]
    rot_p = np.rot90(plaquette_matrix)
    rot_v = np.roll(np.rot90(vertex_matrix), 1, axis=0)
    return np.stack((rot_v, rot_p), axis=0)


def shift_state_ref(row, col, previous_state, state, grid_shift):
    """util.py:97-102 / numba/util.py:8-14."""
    pp = np.roll(np.roll(previous_state, grid_shift - row, axis=1), grid_shift - col, axis=2)
    p = np.roll(np.roll(state, grid_shift - row, axis=1), grid_shift - col, axis=2)
    return pp, p


def generate_perspective_ref(grid_shift, toric_size, state):
    """util.py:46-85 (authoritative loop form) restated; returns (list, list)."""
    d = toric_size
    v, p = state[0], state[1]
    persp, pos = [], []
    for i in range(d):
        for j in range(d):
            if v[i, j] == 1 or v[(i + 1) % d, j] == 1 or p[i, j] == 1 or p[i, (j - 1) % d] == 1:
                ns = np.roll(np.roll(state, grid_shift - i, axis=1), grid_shift - j, axis=2)
                persp.append(ns)
                pos.append((0, i, j))
    for i in range(d):
        for j in range(d):
            if v[i, j] == 1 or v[i, (j + 1) % d] == 1 or p[i, j] == 1 or p[(i - 1) % d, j] == 1:
                ns = np.roll(np.roll(state, grid_shift - i, axis=1), grid_shift - j, axis=2)
                persp.append(rotate_state_ref(ns))
                pos.append((1, i, j))
    return persp, pos


def generate_perspective_batch_ref(grid_shift, toric_size, states):
    """numba/util_actor.py:56-67 + flatten :33-39: env-major concatenation.

    Returns perspectives (P,2,d,d) int64, positions (P,3) int64, counts (N,) int64.
    """
    pers, poss, counts = [], [], []
    for s in states:
        a, b = generate_perspective_ref(grid_shift, toric_size, s)
        pers.extend(a)
        poss.extend(b)
        counts.append(len(a))
    d = toric_size
    P = len(pers)
    out = np.asarray(pers, dtype=np.int64).reshape(P, 2, d, d)
    return out, np.asarray(poss, dtype=np.int64).reshape(P, 3), np.asarray(counts, dtype=np.int64)


def generate_transition_ref(action, reward, state, next_state, terminal, grid_shift):
    """util_actor.py:223-264 per-lattice loop; returns dict of arrays
    (perspective, position, op, reward, next_perspective, terminal)."""
    n = next_state.shape[0]
    d = next_state.shape[-1]
    per = np.empty((n, 2, d, d), np.int64)
    nper = np.empty((n, 2, d, d), np.int64)
    position = np.empty((n, 3), np.int64)
    op = np.empty(n, np.int64)
    for i in range(n):
        qm, row, col, o = (int(x) for x in action[i])
        pp, p = shift_state_ref(row, col, state[i], next_state[i], grid_shift)
        if qm == 1:
            pp = rotate_state_ref(pp)
            p = rotate_state_ref(p)
        per[i], nper[i] = pp, p
        position[i] = (qm, grid_shift, grid_shift)
        op[i] = o
    return dict(perspective=per, position=position, op=op,
                reward=np.asarray(reward, np.float64).copy(),
                next_perspective=nper, terminal=np.asarray(terminal, bool).copy())


# ----------------------------------------------------------- batch closed forms
def hit_masks(states):
    """Defect-adjacent qubit masks (numba/util.py:48-52, :62-66).

    E0[i,j] = v[i,j] | v[i+1,j] | p[i,j] | p[i,j-1]
    E1[i,j] = v[i,j] | v[i,j+1] | p[i,j] | p[i-1,j]
    Returns (N,2,d,d) bool.
    """
    s = np.asarray(states) != 0
    v, p = s[..., 0, :, :], s[..., 1, :, :]
    e0 = v | np.roll(v, -1, axis=-2) | p | np.roll(p, 1, axis=-1)
    e1 = v | np.roll(v, -1, axis=-1) | p | np.roll(p, 1, axis=-2)
    return np.stack((e0, e1), axis=-3)


def perspective_source_index(d):
    """LUT src[layer, i, j, c*d*d + r*d + s] -> flat index into state (2,d,d).

    layer 0:  P0[c,r,s] = state[c,(r+i-gs)%d,(s+j-gs)%d]                 (numba/util.py:56-57)
    layer 1:  P1[1,r,s] = state[1,(s+i-gs)%d,(d-1-r+j-gs)%d]             (rot90)
              P1[0,r,s] = state[0,(s+i-gs)%d,((d-r)%d+j-gs)%d]           (rot90 then roll +1)
    """
    gs = d // 2
    i = np.arange(d).reshape(d, 1, 1, 1, 1)
    j = np.arange(d).reshape(1, d, 1, 1, 1)
    c = np.arange(2).reshape(1, 1, 2, 1, 1)
    r = np.arange(d).reshape(1, 1, 1, d, 1)
    s = np.arange(d).reshape(1, 1, 1, 1, d)
    src0 = c * d * d + ((r + i - gs) % d) * d + ((s + j - gs) % d)
    col1 = np.where(c == 1, (d - 1 - r + j - gs) % d, ((d - r) % d + j - gs) % d)
    src1 = c * d * d + ((s + i - gs) % d) * d + col1
    src0 = np.broadcast_to(src0, (d, d, 2, d, d))
    src1 = np.broadcast_to(src1, (d, d, 2, d, d))
    return np.stack((src0, src1), axis=0).reshape(2, d, d, 2 * d * d).astype(np.int64)


def generate_perspective_batch(states, dtype=np.uint8):
    """Closed-form batch equivalent of generate_perspective_batch_ref.

    Returns perspectives (P,2,d,d) ``dtype``, positions (P,3) int32, counts (N,)
    int32, offsets (N+1,) int64 (exclusive scan of counts).
    """
    s = np.asarray(states)
    n, _, d, _ = s.shape
    m = hit_masks(s).reshape(n, -1)
    env, hit = np.nonzero(m)                 # env-major, then layer-major, row-major
    counts = m.sum(axis=1).astype(np.int32)
    offsets = np.zeros(n + 1, np.int64)
    np.cumsum(counts, out=offsets[1:])
    lut = perspective_source_index(d).reshape(2 * d * d, 2 * d * d)
    flat = s.reshape(n, -1)
    persp = flat[env[:, None], lut[hit]].astype(dtype).reshape(-1, 2, d, d)
    layer, rem = np.divmod(hit, d * d)
    row, col = np.divmod(rem, d)
    pos = np.stack((layer, row, col), axis=1).astype(np.int32)
    return persp, pos, counts, offsets


def generate_transition_batch(actions, states, next_states, dtype=np.uint8):
    """Closed form of util_actor.py:223-264: perspective of the acted qubit applied
    to ``state`` and to ``next_state``; position rewritten to (layer, gs, gs)."""
    s = np.asarray(states)
    ns = np.asarray(next_states)
    n, _, d, _ = s.shape
    a = np.asarray(actions).astype(np.int64).reshape(n, 4)
    lut = perspective_source_index(d)
    src = lut[a[:, 0], a[:, 1], a[:, 2]]                      # (n, 2dd)
    idx = np.arange(n)[:, None]
    per = s.reshape(n, -1)[idx, src].astype(dtype).reshape(n, 2, d, d)
    nper = ns.reshape(n, -1)[idx, src].astype(dtype).reshape(n, 2, d, d)
    gs = d // 2
    act = np.stack((a[:, 0], np.full(n, gs), np.full(n, gs), a[:, 3]), axis=1).astype(np.int32)
    return per, act, nper


def select_action_batch(q_table, offsets, positions, eps, seed, env_ids, episodes, steps, domain=DOMAIN_SEL):
    """_selectActionBatch_prime (numba/util_actor.py:69-107) with Philox draws.

    ``domain=DOMAIN_SEL_CALL`` is the stateless form (selectActionBatch over an explicit state
    array, numba/util_actor.py:11-53): env_ids = state index, episodes / steps = low / high
    32 bits of the caller's call counter.

    greedy iff (1-eps) > U (:49-50); greedy = first (p,a) attaining the max in
    row-major order (:93-95); else p ~ randint(n), a ~ randint(3) (:97-98);
    action = [pos[p], a+1], q = q_table row p (:100-104).
    Returns actions (N,4) int32, q_values (N,3) f32, chosen perspective index (N,).
    """
    offsets = np.asarray(offsets, np.int64)
    n = offsets.shape[0] - 1
    k0, k1 = _key(seed)
    w0, w1, w2, _ = philox4x32(np.asarray(env_ids, np.uint64), np.asarray(episodes, np.uint64),
                               np.asarray(steps, np.uint64), np.uint64(domain << 24), k0, k1)
    w0, w1, w2 = (np.broadcast_to(w, (n,)) for w in (w0, w1, w2))
    eps = np.broadcast_to(np.asarray(eps, np.float64), (n,))
    greedy = (1.0 - eps) > _u01(w0)
    cnt = offsets[1:] - offsets[:-1]
    actions = np.zeros((n, 4), np.int32)
    qv = np.zeros((n, 3), np.float32)
    chosen = np.zeros(n, np.int64)
    q_table = np.asarray(q_table, np.float32).reshape(-1, 3)
    rp = _mulhi(w1, cnt)
    ra = _mulhi(w2, 3)
    for i in range(n):
        lo, hi = offsets[i], offsets[i + 1]
        if hi == lo:                       # empty syndrome: no legal action
            actions[i] = (0, 0, 0, 0)
            continue
        if greedy[i]:
            flat = int(np.argmax(q_table[lo:hi].reshape(-1)))
            p, a = divmod(flat, 3)
        else:
            p, a = int(rp[i]), int(ra[i])
        chosen[i] = lo + p
        qv[i] = q_table[lo + p]
        actions[i, :3] = positions[lo + p]
        actions[i, 3] = a + 1
    return actions, qv, chosen


def compute_priorities(A, R, Q, Qns, discount):
    """computePrioritiesParallel (util_actor.py:268-287): |R + discount * max_a Qns - Q[a]| with the
    reference's dtypes (its local buffers are f64 arrays, Actor_mp.py:65-70) -> f64 (N,T)."""
    A, R = np.asarray(A), np.asarray(R, np.float64)
    Q, Qns = np.asarray(Q, np.float64), np.asarray(Qns, np.float64)
    qns_max = np.amax(Qns, axis=2)
    actions = A[:, :, -1].astype(np.int64) - 1
    row = np.arange(actions.shape[-1])
    qv = np.array([Q[env, row, actions[env]] for env in range(len(Q))])
    return np.absolute(R + discount * qns_max - qv)


def predict_max(q_fn, states):
    """predictMaxOptimized (util_learner.py:48-111): the max Q-value of every state.  Quirks kept:
    a state without perspectives (terminal) contributes one all-zero dummy perspective (:74-76) and
    its output is forced to 0 (:108); every Q-slice is padded with zero rows up to the longest
    slice before the argmax (:98-100), so a shorter slice yields max(max_q, 0).
    ``q_fn(perspectives (P,2,d,d) float32) -> (P,3) float32``.  -> float32 (n,)."""
    states = np.asarray(states)
    n, d = states.shape[0], states.shape[-1]
    per, _, cnt, off = generate_perspective_batch(states.astype(np.uint8))
    chunks, lengths = [], []
    for i in range(n):
        p = per[off[i]:off[i + 1]] if cnt[i] else np.zeros((1, 2, d, d), per.dtype)
        chunks.append(p)
        lengths.append(p.shape[0])
    q = np.asarray(q_fn(np.concatenate(chunks).astype(np.float32)), np.float32)
    largest = max(lengths)
    out = np.zeros(n, np.float32)
    lo = 0
    for i, m in enumerate(lengths):
        padded = np.concatenate((q[lo:lo + m], np.zeros((largest - m, 3), np.float32)), axis=0)
        out[i] = 0.0 if cnt[i] == 0 else padded.reshape(-1).max()
        lo += m
    return out


def perror_schedule(seed, env_ids, episodes, p_start, p_roof, strategy):
    """Actor_mp.py:176-180: 'random' -> U(start, roof) else roof (linear)."""
    p_roof = np.asarray(p_roof, np.float64)
    if strategy != 'random':
        return p_roof.copy()
    k0, k1 = _key(seed)
    w0, _, _, _ = philox4x32(np.asarray(env_ids, np.uint64), np.asarray(episodes, np.uint64),
                             np.uint64(0), np.uint64(DOMAIN_PERR << 24), k0, k1)
    span = p_roof - float(p_start)
    return float(p_start) + span * _u01(w0)


# ------------------------------------------------------------------ env objects
class OracleEnvSet:
    """EnvSet (EnvSet.py:4-51) over the restated env, numpy only.

    Same surface: size, no_envs, resetAll, resetTerminalEnvs, step.  RNG is the
    Philox contract above keyed by (seed, first_env_id + local index, episode).
    """

    def __init__(self, size, no_envs, p_error=0.1, seed=0, first_env_id=0,
                 terminal_reward=TERMINAL_REWARD, min_qubit_errors=0):
        self.min_qubit_errors = int(min_qubit_errors)
        self.size = int(size)
        self.no_envs = int(no_envs)
        self.p_error = float(p_error)
        self.seed = int(seed)
        self.env_ids = np.arange(first_env_id, first_env_id + no_envs, dtype=np.int64)
        self.terminal_reward = float(terminal_reward)
        d = self.size
        self.qubits = np.zeros((no_envs, 2, d, d), np.uint8)
        self.states = np.zeros((no_envs, 2, d, d), np.uint8)
        self.episodes = np.zeros(no_envs, np.int64)     # episodes started so far
        self.steps = np.zeros(no_envs, np.int64)        # steps taken in the current episode

    def _reset(self, idx, p_errors):
        idx = np.asarray(idx, np.int64).reshape(-1)
        if p_errors is None:
            p = np.full(idx.shape[0], self.p_error)
        else:
            p = np.asarray(p_errors, np.float64).reshape(-1)
        q, s = reset_lattices(self.seed, self.env_ids[idx], self.episodes[idx], p, self.size, self.min_qubit_errors)
        self.qubits[idx] = q
        self.states[idx] = s
        self.episodes[idx] += 1
        self.steps[idx] = 0
        return s

    def resetAll(self, p_errors=None):
        self._reset(np.arange(self.no_envs), p_errors)
        return self.states.astype(np.int64)

    def resetTerminalEnvs(self, idx, p_errors=None):
        return self._reset(idx, p_errors).astype(np.float64)   # EnvSet.py:20 float64

    def step(self, actions):
        self.qubits, nxt, rew, term = step_lattices(self.qubits, self.states, actions,
                                                    self.terminal_reward)
        self.states = nxt
        self.steps += 1
        return nxt.astype(np.int64), rew, term, {}


def run_actor_steps_ref(env, n_steps, eps=1.0, q_fn=None, max_steps_per_episode=75,
                        p_error=None, record=False):
    """The call order of Actor_mp.py:104-185 with the reference-shaped per-lattice
    loops (CPU baseline form).  ``q_fn(perspectives) -> (P,3)`` stands in for the NN;
    None means zeros (with eps=1 the Q-values never influence the action)."""
    d, gs = env.size, env.size // 2
    state = env.states.astype(np.int64)
    log = []
    for _ in range(n_steps):
        persp, pos, counts = generate_perspective_batch_ref(gs, d, state)
        offsets = np.zeros(env.no_envs + 1, np.int64)
        np.cumsum(counts, out=offsets[1:])
        q = np.zeros((persp.shape[0], 3), np.float32) if q_fn is None else q_fn(persp)
        actions, qv, _ = select_action_batch(q, offsets, pos, eps, env.seed, env.env_ids,
                                             env.episodes, env.steps)
        next_state, reward, terminal, _ = env.step(actions)
        tr = generate_transition_ref(actions, reward, state, next_state, terminal, gs)
        done = terminal | (env.steps > max_steps_per_episode)
        if done.any():
            idx = np.nonzero(done)[0]
            next_state[idx] = env.resetTerminalEnvs(idx, None if p_error is None else
                                                    np.full(idx.shape[0], p_error))
        if record:
            log.append(dict(actions=actions, q=qv, transition=tr, reward=reward,
                            terminal=terminal, done=done, state=next_state.copy()))
        state = next_state
    return log
