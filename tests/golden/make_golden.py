#!/usr/bin/env python3
"""Generate the committed golden vectors from the IMPORTED reference helpers.

Runs only in the authoring container (needs /root/reference); the GPU box and the
test-suite only read the resulting ``*.npz`` files (plain uint8/int/f32 arrays,
loaded with allow_pickle=False).

What is imported from the reference (numpy half of the path, importable here):
  src/util.py          generatePerspective, generatePerspectiveOptimized,
                       rotate_state, shift_state
  src/util_actor.py    generateTransitionParallel, selectActionParallel_prime,
                       computePrioritiesParallel
  src/util_learner.py  predictMaxOptimized (driven with an integer-weight linear model, so
                       its fp32 Q-values are exact and the expected maxima are bit-exact)
  src/numba/util.py        generatePerspectiveOptimized, rotate_state, shift_state
  src/numba/util_actor.py  generatePerspectiveBatch (+ the np.concatenate / float32 cast of :33-39),
                           _selectActionBatch_prime (greedy branch, forced ties)
                       -- the variant PRODUCTION imports (src/Actor_mp.py:13).
Shims: the numpy-1 aliases np.int/np.bool/np.float that numpy 2 removed (src/util.py:10 uses
np.int), and -- numba is absent here and stays absent -- an identity stand-in for the two names
``src/numba/*`` takes from it: ``njit = jit = identity decorator`` and ``numba.typed.List = list``
(SURVEY.md 8(c) records this import).  The decorated functions are plain Python/numpy, so the
stand-in runs exactly the reference's source text, un-jitted; outputs go to ``numba_d*.npz``.

The gym_ToricCode env (reset/step/syndrome) is not in the reference tree, so no
vector for it can be generated from the reference: env_kat_d*.npz below holds
DERIVED known answers (single-Pauli syndromes per the adjacency rule of
util.py:68-69,77-78 and the facts SURVEY.md 8(c) records about fixture rows
0/2/7/13/19 of output_speed_test/transitions_0.npy: Z on the centred qubit flips
vertex (gs,gs) and (gs+1,gs) in both layers' perspectives).  That file itself is
a pickled object array; numpy.load(allow_pickle=False) refuses it and it is not
unpickled here.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("TORIC_REFERENCE", "/root/reference")

np.int = int      # numpy-1 aliases used by the reference (src/util.py:10)
np.bool = bool
np.float = float
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)


def _install_numba_stand_in():
    """`from numba import njit, jit` / `from numba.typed import List` with numba absent: decorators that return
    the function unchanged (bare and called forms) and the builtin list."""
    import types

    def identity(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda f: f
    nb = types.ModuleType("numba")
    nb.njit = nb.jit = identity
    typed = types.ModuleType("numba.typed")
    typed.List = list
    nb.typed = typed
    assert "numba" not in sys.modules
    sys.modules["numba"] = nb
    sys.modules["numba.typed"] = typed


_install_numba_stand_in()

import torch                     # noqa: E402
import src.util as RU            # noqa: E402  (reference)
import src.util_actor as RA      # noqa: E402  (reference)
import src.util_learner as RL    # noqa: E402  (reference)
import src.numba.util as NU      # noqa: E402  (reference, the variant production imports)
import src.numba.util_actor as NA  # noqa: E402  (reference)
from oracle import toric_oracle as O  # noqa: E402


def random_states(rng, d, k):
    """Mix of realistic syndromes (from the oracle sampler at several p) and
    adversarial binary grids (empty, full, single cell, dense random)."""
    out = []
    for p in (0.02, 0.1, 0.15, 0.3):
        _, st = O.reset_lattices(int(rng.integers(1 << 30)), np.arange(k), 0, p, d)
        out.append(st)
    dense = (rng.random((k, 2, d, d)) < 0.5).astype(np.uint8)
    out.append(dense)
    special = np.zeros((4 + 2 * d * d, 2, d, d), np.uint8)
    special[1] = 1
    special[2, 0, 0, 0] = 1
    special[3, 1, d - 1, d - 1] = 1
    for c in range(2 * d * d):
        special[4 + c].reshape(-1)[c] = 1
    out.append(special)
    return np.concatenate(out, axis=0)


def ref_perspectives(d, states):
    gs = int(d / 2)
    pers, poss, counts = [], [], []
    for s in states:
        s64 = s.astype(np.int64)
        a, b = RU.generatePerspectiveOptimized(gs, d, s64)
        loop = RU.generatePerspective(gs, d, s64)          # authoritative loop form
        assert len(loop) == len(a)
        for x, y, z in zip(loop, a, b):
            assert np.array_equal(x.perspective, y) and tuple(int(t) for t in x.position) == tuple(int(t) for t in z)
        pers.extend(a)
        poss.extend(b)
        counts.append(len(a))
    P = len(pers)
    return (np.asarray(pers, np.int64).reshape(P, 2, d, d),
            np.asarray(poss, np.int64).reshape(P, 3), np.asarray(counts, np.int64))


class IntLinearQ(torch.nn.Module):
    """Q(perspective) = flatten(perspective) @ w with small integer weights: every product and sum
    is an exact fp32 integer on any device, so the reference's maxima are reproducible bit-for-bit."""

    def __init__(self, w):
        super().__init__()
        self.w = torch.nn.Parameter(torch.as_tensor(w, dtype=torch.float32), requires_grad=False)

    def forward(self, x):
        return x.flatten(1).float() @ self.w


def learner_and_priority_vectors(d, rng):
    """predictMaxOptimized (util_learner.py:48-111) and computePrioritiesParallel
    (util_actor.py:268-287) run from the reference, inputs and outputs frozen."""
    gs = int(d / 2)
    n = 48
    _, st = O.reset_lattices(int(rng.integers(1 << 30)), np.arange(n), 0, 0.08, d)
    st[::7] = 0                                              # terminal states (dummy perspective, output 0)
    st[1] = 0
    st[1, 0, 0, 0] = st[1, 0, 1, 0] = 1                      # a short slice next to long ones: zero-padding quirk
    w = rng.integers(-2, 3, (2 * d * d, 3)).astype(np.float32)
    w[:, :] -= 1.0                                           # mostly negative Q: padding with zero rows matters
    model = IntLinearQ(w)
    batch_state = [torch.from_numpy(x.astype(np.float32)) for x in st]
    out = RL.predictMaxOptimized(model, batch_state, gs, d, 'cpu').numpy().astype(np.float32)
    # priorities: the shapes and dtypes of the actor's local buffers (Actor_mp.py:65-70,146-150)
    N, T = 24, 5
    A = np.stack((rng.integers(0, 2, (N, T)), rng.integers(0, d, (N, T)), rng.integers(0, d, (N, T)),
                  rng.integers(1, 4, (N, T))), axis=2).astype(np.int64)
    Q = np.empty((N, T + 1), dtype=(np.float64, 3))
    Q[:] = (rng.standard_normal((N, T + 1, 3)) * 40).astype(np.float32)        # f32 network outputs in an f64 buffer
    R = rng.integers(-4, 5, (N, T + 1)).astype(np.float64)
    R[rng.random((N, T + 1)) < 0.15] = 100.0
    pr = RA.computePrioritiesParallel(A, R[:, :-1], Q[:, :-1], np.roll(Q, -1, axis=1)[:, :-1], 0.95)
    assert pr.dtype == np.float64 and np.array_equal(pr, O.compute_priorities(A, R[:, :-1], Q[:, :-1],
                                                                               np.roll(Q, -1, axis=1)[:, :-1], 0.95))
    return dict(pm_states=st.astype(np.uint8), pm_w=w, pm_out=out, pr_A=A.astype(np.uint8), pr_R=R[:, :-1],
                pr_Q=Q.astype(np.float32), pr_discount=np.float64(0.95), pr_out=pr)


def numba_variant_vectors(d, states, rp, rpos, rcnt, rng):
    """The numba-source variant (src/numba/util.py:28-76, src/numba/util_actor.py:33-39,56-107) on the same states:
    asserted equal to the numpy variant and to both oracle forms, outputs frozen.  The batch + concatenate of
    :33-38 cannot take a state without defects (np.concatenate of a (0,) with (n,3) position lists raises), so it
    runs on the non-empty states, in order; `nonempty` is frozen with it."""
    gs = int(d / 2)
    n = states.shape[0]
    offs = np.zeros(n + 1, np.int64)
    np.cumsum(rcnt, out=offs[1:])
    # --- generatePerspectiveOptimized, state by state (empty states included: two empty lists)
    for i, s in enumerate(states):
        per, pos = NU.generatePerspectiveOptimized(gs, d, s.astype(np.int64))
        assert len(per) == len(pos) == rcnt[i]
        if len(per):
            assert np.array_equal(np.asarray(per), rp[offs[i]:offs[i + 1]])
            assert np.array_equal(np.asarray(pos, np.int64), rpos[offs[i]:offs[i + 1]])
    # --- rotate_state / shift_state of the variant
    ar = np.arange(2 * d * d).reshape(2, d, d)
    assert np.array_equal(NU.rotate_state(ar), RU.rotate_state(ar))
    na, nb_ = NU.shift_state(1, d - 1, ar, ar[::-1].copy(), gs)
    ra, rb = RU.shift_state(1, d - 1, ar, ar[::-1].copy(), gs)
    assert np.array_equal(na, ra) and np.array_equal(nb_, rb)
    # --- generatePerspectiveBatch + the flatten of selectActionBatch (:33-39)
    nz = rcnt > 0
    sub = states[nz].astype(np.int64)
    perspectives, positions, splice_idx = NA.generatePerspectiveBatch(gs, d, sub)
    splice_idx = np.cumsum(splice_idx)
    positions = np.concatenate(positions)
    perspectives = np.concatenate(perspectives)
    as_tensor = torch.from_numpy(perspectives).type('torch.Tensor')            # :39, what the network is fed
    assert as_tensor.dtype == torch.float32 and np.array_equal(as_tensor.numpy(), perspectives.astype(np.float32))
    assert np.array_equal(perspectives, rp) and np.array_equal(positions, rpos)    # empty states contribute nothing
    assert np.array_equal(splice_idx, offs[1:][nz])
    bp, bpos, bcnt, boff = O.generate_perspective_batch(states[nz])
    assert np.array_equal(bp, perspectives) and np.array_equal(bpos, positions) and np.array_equal(boff[1:], splice_idx)
    op_, opos, ocnt = O.generate_perspective_batch_ref(gs, d, sub)
    assert np.array_equal(op_, perspectives) and np.array_equal(opos, positions)
    # --- _selectActionBatch_prime, greedy everywhere, ties forced: inside a slice, across ops of one row, and
    # a slice whose every entry is equal (first (p, a) in row-major order must win, :93-95)
    q = rng.standard_normal((perspectives.shape[0], 3)).astype(np.float32)
    first = np.concatenate(([0], splice_idx[:-1]))
    for k, (lo, hi) in enumerate(zip(first, splice_idx)):
        if k % 3 == 0 and hi - lo >= 2:
            m = q[lo:hi].max() + 1.0
            rows = rng.choice(hi - lo, size=2, replace=False)
            q[lo + rows[0], rng.integers(3)] = m
            q[lo + rows[1], rng.integers(3)] = m
        if k % 7 == 1:
            q[lo + rng.integers(hi - lo), :] = q[lo:hi].max() + 2.0
        if k % 11 == 2:
            q[lo:hi] = 0.25
    acts, qv = NA._selectActionBatch_prime(q, splice_idx, positions, np.ones(len(splice_idx), bool))
    assert acts.dtype == np.float64 and qv.dtype == np.float64
    oact, oqv, _ = O.select_action_batch(q, np.concatenate(([0], splice_idx)), positions, 0.0, 1, np.arange(len(splice_idx)), 1, 0)
    assert np.array_equal(acts.astype(np.int64), oact) and np.array_equal(qv.astype(np.float32), oqv)
    # the numpy variant's selection (util_actor.py:189-221) on the same table agrees as well
    ract, rqv = RA.selectActionParallel_prime([q[a:b] for a, b in zip(first, splice_idx)],
                                              [positions[a:b] for a, b in zip(first, splice_idx)], np.ones(len(splice_idx), bool))
    assert np.array_equal(ract, acts.astype(np.int64)) and np.array_equal(rqv.astype(np.float64), qv)
    return dict(nonempty=nz, perspectives=perspectives.astype(np.uint8), positions=positions.astype(np.uint8),
                splice_idx=splice_idx.astype(np.int64), sel_q=q, sel_actions=acts.astype(np.uint8), sel_qv=qv.astype(np.float32))


def main():
    rng = np.random.default_rng(20200318)
    report = []
    for d in (3, 5, 7, 9, 11, 13, 15, 17, 19, 21):        # new sizes last: the draws of the earlier files are unchanged
        gs = int(d / 2)
        k = {3: 40, 5: 30, 7: 24, 9: 16, 11: 8, 13: 6, 15: 5, 17: 3, 19: 2, 21: 2}[d]
        states = random_states(rng, d, k)
        n = states.shape[0]

        # --- perspectives: reference vs oracle (both forms), then freeze
        rp, rpos, rcnt = ref_perspectives(d, states)
        op, opos, ocnt = O.generate_perspective_batch_ref(gs, d, states.astype(np.int64))
        bp, bpos, bcnt, boff = O.generate_perspective_batch(states)
        assert np.array_equal(rp, op) and np.array_equal(rpos, opos) and np.array_equal(rcnt, ocnt)
        assert np.array_equal(rp, bp) and np.array_equal(rpos, bpos) and np.array_equal(rcnt, bcnt)

        # --- rotate / shift KATs
        ar = np.arange(2 * d * d).reshape(2, d, d)
        rot = RU.rotate_state(ar)
        assert np.array_equal(rot, O.rotate_state_ref(ar))
        sh_a, sh_b = RU.shift_state(1, d - 1, ar, ar[::-1].copy(), gs)
        oa, ob = O.shift_state_ref(1, d - 1, ar, ar[::-1].copy(), gs)
        assert np.array_equal(sh_a, oa) and np.array_equal(sh_b, ob)

        # --- transitions: random (state,next_state) pairs + random legal actions
        nxt = states[rng.permutation(n)]
        actions = np.stack((rng.integers(0, 2, n), rng.integers(0, d, n),
                            rng.integers(0, d, n), rng.integers(1, 4, n)), axis=1).astype(np.int64)
        reward = rng.integers(-4, 5, n).astype(np.float64)
        terminal = rng.random(n) < 0.2
        trans_type = np.dtype([('perspective', (np.int64, (2, d, d))),
                               ('action', RU.action_type), ('reward', np.float64),
                               ('next_perspective', (np.int64, (2, d, d))),
                               ('terminal', np.bool_)])   # Actor_mp.py:52-56
        rt = RA.generateTransitionParallel(actions, reward, states.astype(np.int64),
                                           nxt.astype(np.int64), terminal, gs, trans_type)
        ot = O.generate_transition_ref(actions, reward, states.astype(np.int64),
                                       nxt.astype(np.int64), terminal, gs)
        bper, bact, bnper = O.generate_transition_batch(actions, states, nxt)
        assert np.array_equal(rt['perspective'], ot['perspective'])
        assert np.array_equal(rt['next_perspective'], ot['next_perspective'])
        assert np.array_equal(rt['action']['position'], ot['position'])
        assert np.array_equal(rt['action']['op'], ot['op'])
        assert np.array_equal(rt['perspective'], bper) and np.array_equal(rt['next_perspective'], bnper)
        assert np.array_equal(rt['action']['position'], bact[:, :3]) and np.array_equal(rt['action']['op'], bact[:, 3])
        assert np.array_equal(rt['reward'], reward) and np.array_equal(rt['terminal'], terminal)

        # --- greedy selection (reference branch without RNG): util_actor.py:189-221
        q = rng.standard_normal((rp.shape[0], 3)).astype(np.float32)
        q[rng.integers(0, q.shape[0], 8)] = q.max()           # force ties
        nz = rcnt > 0                                           # reference cannot select on an empty slice
        offs = np.zeros(n + 1, np.int64)
        np.cumsum(rcnt, out=offs[1:])
        qs = [q[offs[i]:offs[i + 1]] for i in range(n) if nz[i]]
        ps = [rpos[offs[i]:offs[i + 1]] for i in range(n) if nz[i]]
        ract, rqv = RA.selectActionParallel_prime(qs, ps, np.ones(len(qs), bool))
        oact, oqv, _ = O.select_action_batch(q, offs, rpos, 0.0, 1, np.arange(n), 1, 0)
        assert np.array_equal(ract, oact[nz]) and np.array_equal(rqv.astype(np.float32), oqv[nz])

        np.savez_compressed(os.path.join(HERE, f"numba_d{d}.npz"),
                            **numba_variant_vectors(d, states, rp, rpos, rcnt, np.random.default_rng(9000 + d)))

        np.savez_compressed(os.path.join(HERE, f"learner_d{d}.npz"),
                            **learner_and_priority_vectors(d, np.random.default_rng(7000 + d)))

        np.savez_compressed(
            os.path.join(HERE, f"reference_d{d}.npz"),
            states=states.astype(np.uint8), perspectives=rp.astype(np.uint8),
            positions=rpos.astype(np.uint8), counts=rcnt.astype(np.int32),
            rotate_in=ar.astype(np.int16), rotate_out=rot.astype(np.int16),
            shift_prev=sh_a.astype(np.int16), shift_next=sh_b.astype(np.int16),
            t_next_states=nxt.astype(np.uint8), t_actions=actions.astype(np.uint8),
            t_reward=reward, t_terminal=terminal,
            t_perspective=rt['perspective'].astype(np.uint8),
            t_next_perspective=rt['next_perspective'].astype(np.uint8),
            t_position=rt['action']['position'].astype(np.uint8), t_op=rt['action']['op'].astype(np.uint8),
            sel_q=q, sel_nonempty=nz, sel_actions=ract.astype(np.uint8), sel_qv=rqv.astype(np.float32))
        report.append((d, n, int(rp.shape[0])))

    # --- roll KAT of tests/roll_numba.py:36-56 (np.roll on arange(162).reshape(2,9,9))
    b = np.arange(162).reshape(2, 9, 9)
    np.savez_compressed(os.path.join(HERE, "roll_kat.npz"), b=b.astype(np.int16),
                        roll_axis1_p1=np.roll(b, 1, axis=1).astype(np.int16),
                        roll_axis2_m1=np.roll(b, -1, axis=2).astype(np.int16))

    # --- hand-made d=5 qubit matrix of tests/plotSyndroms.py:10-20 (input only upstream)
    s = np.zeros((2, 5, 5), np.uint8)
    s[0, 2, 2] = s[0, 3, 2] = 3
    s[1, 1, 0] = s[1, 1, 1] = 3
    np.savez_compressed(os.path.join(HERE, "plot_syndroms_d5.npz"), qubits=s)

    for d, n, P in report:
        print(f"d={d}: {n} states, {P} perspectives frozen; reference (numpy variant) == reference (numba-source variant) "
              f"== oracle_ref == oracle_batch")


if __name__ == "__main__":
    main()
