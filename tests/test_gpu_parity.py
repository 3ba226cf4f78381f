"""Parity of the HIP path (through the C-ABI) with the oracle, bit-exact, on a real MI355X.

Run with ``pytest -m gpu``.  Every comparison is integer/byte exact; the only floats are
rewards (small integers) and Q-values that are copied, never computed.
"""
import os

import numpy as np
import pytest
import torch

from oracle import toric_oracle as O

pytestmark = pytest.mark.gpu

SIZES = (3, 5, 7, 9, 11, 13, 15, 17, 19, 21)
P_OF = {3: 0.1, 5: 0.1, 7: 0.1, 9: 0.15, 11: 0.1, 13: 0.1, 15: 0.08, 17: 0.08, 19: 0.07, 21: 0.06}


@pytest.fixture(scope="module")
def T():
    import toric_rl_decoder_amd as T
    assert torch.cuda.is_available(), "these tests need the GPU"
    assert os.path.exists(T.LIB_PATH), "libtoricenv.so must be built (no fallback path exists)"
    T.load()
    return T


def make_pair(T, d, n, p=None, seed=1234, first=0, numpy_io=True, **kw):
    p = P_OF[d] if p is None else p
    env = T.make("toric-code-v0", {"size": d, "min_qubit_errors": 0, "p_error": p})
    gpu = T.EnvSet(env, n, seed=seed, first_env_id=first, numpy_io=numpy_io, **kw)
    ora = O.OracleEnvSet(d, n, p, seed=seed, first_env_id=first)
    return gpu, ora


def random_actions_from(pos, off, rng):
    n = off.shape[0] - 1
    cnt = off[1:] - off[:-1]
    pick = off[:-1] + (rng.random(n) * cnt).astype(np.int64)
    a = np.zeros((n, 4), np.int64)
    a[:, :3] = pos[pick]
    a[:, 3] = rng.integers(1, 4, n)
    return a


# ------------------------------------------------------------------ EnvSet surface, every size
@pytest.mark.parametrize("d", SIZES)
def test_envset_surface_parity(T, d):
    n = 500                                     # not a multiple of the 256-thread workgroup
    gpu, ora = make_pair(T, d, n, first=37)
    s = gpu.resetAll()
    os_ = ora.resetAll()
    assert s.dtype == np.int64 and np.array_equal(s, os_)
    assert np.array_equal(gpu.getQubits(), ora.qubits)
    rng = np.random.default_rng(d)
    for t in range(16):
        per, pos, cnt = gpu.generatePerspective()
        bp, bpos, bcnt, boff = O.generate_perspective_batch(ora.states)
        assert per.dtype == np.float32 and np.array_equal(per, bp.astype(np.float32))
        assert np.array_equal(pos, bpos) and np.array_equal(cnt, bcnt)
        q = rng.standard_normal((per.shape[0], 3)).astype(np.float32)
        q[rng.integers(0, q.shape[0], 16)] = q.max()                       # ties -> first maximum
        eps = rng.random(n)
        act, qv = gpu.selectAction(q, eps)
        oact, oqv, _ = O.select_action_batch(q, boff, bpos, eps, ora.seed, ora.env_ids, ora.episodes, ora.steps)
        assert np.array_equal(act, oact) and np.array_equal(qv, oqv)
        prev = ora.states.copy()
        ns, rew, term, info = gpu.step(act)
        ons, orew, oterm, _ = ora.step(oact)
        assert np.array_equal(ns, ons) and np.array_equal(rew, orew) and np.array_equal(term, oterm)
        assert rew.dtype == np.float64 and term.dtype == bool and info == {}
        tr = gpu.generateTransition(act)
        tper, tact, tnper = O.generate_transition_batch(oact, prev, ora.states)
        assert np.array_equal(tr["perspective"], tper) and np.array_equal(tr["next_perspective"], tnper)
        assert np.array_equal(tr["action"], tact)
        assert np.array_equal(gpu.evalGroundState(), O.eval_ground_state(ora.qubits))
        assert np.array_equal(gpu.isTerminal(), oterm)
        done = term | (ora.steps > 6)
        idx = np.nonzero(done)[0]
        if idx.size:
            p_new = rng.uniform(0.05, 0.2, idx.size)
            rs = gpu.resetTerminalEnvs(idx, p_new)
            ors = ora.resetTerminalEnvs(idx, p_new)
            assert rs.dtype == np.float64 and np.array_equal(rs, ors)
        ep, st = gpu.getCounters()
        assert np.array_equal(ep, ora.episodes) and np.array_equal(st, ora.steps)
    gpu.close()


# ------------------------------------------------------------------ BASELINE configs[1]
def test_config2_4096_envs_d5_bit_exact(T):
    """4096 envs, d=5, p=0.10, seed 1234: reset + 64 random-action steps; qubits, syndrome, reward,
    terminal, perspective stack, positions, counts compared every step."""
    d, n = 5, 4096
    gpu, ora = make_pair(T, d, n, p=0.10, seed=1234)
    assert np.array_equal(gpu.resetAll(), ora.resetAll())
    rng = np.random.default_rng(1234)
    for t in range(64):
        per, pos, cnt = gpu.generatePerspective(dtype=torch.uint8)
        bp, bpos, bcnt, boff = O.generate_perspective_batch(ora.states)
        assert np.array_equal(per, bp) and np.array_equal(pos, bpos) and np.array_equal(cnt, bcnt)
        act = random_actions_from(bpos, boff, rng)
        ns, rew, term, _ = gpu.step(act)
        ons, orew, oterm, _ = ora.step(act)
        assert np.array_equal(ns, ons) and np.array_equal(rew, orew) and np.array_equal(term, oterm)
        assert np.array_equal(gpu.getQubits(), ora.qubits)
        idx = np.nonzero(term | (ora.steps > 75))[0]
        if idx.size:
            assert np.array_equal(gpu.resetTerminalEnvs(idx), ora.resetTerminalEnvs(idx))
    gpu.close()


# ------------------------------------------------------------------ reference golden vectors on the GPU
@pytest.mark.parametrize("d", SIZES)
def test_golden_reference_vectors(T, golden_dir, d):
    g = np.load(os.path.join(golden_dir, f"reference_d{d}.npz"), allow_pickle=False)
    per, pos, cnt = T.generatePerspectiveBatch(d // 2, d, g["states"], dtype=torch.float32)
    assert np.array_equal(per.cpu().numpy(), g["perspectives"].astype(np.float32))
    assert np.array_equal(pos.cpu().numpy(), g["positions"]) and np.array_equal(cnt.cpu().numpy(), g["counts"])
    # transitions of the fixture through the stateless drop-in (util_actor.py:223-264 signature)
    act = np.concatenate((g["t_actions"][:, :3], g["t_op"][:, None]), axis=1).astype(np.int64)
    rec = T.generateTransitionParallel(act, g["t_reward"], g["states"], g["t_next_states"], g["t_terminal"], d // 2)
    assert np.array_equal(rec["perspective"], g["t_perspective"])
    assert np.array_equal(rec["next_perspective"], g["t_next_perspective"])
    assert np.array_equal(rec["action"]["position"], g["t_position"]) and np.array_equal(rec["action"]["op"], g["t_op"])
    assert np.array_equal(rec["reward"], g["t_reward"]) and np.array_equal(rec["terminal"], g["t_terminal"])
    # the numba-source variant production imports (src/numba/util_actor.py:33-39,56-107), frozen from the reference's
    # own source: batch + concatenate + float32 cast, then the greedy selection with forced ties
    nb = np.load(os.path.join(golden_dir, f"numba_d{d}.npz"), allow_pickle=False)
    st = g["states"][nb["nonempty"]]
    per, pos, cnt, off = T.generatePerspectiveBatch(d // 2, d, st, dtype=torch.float32, return_offsets=True)
    assert per.dtype == torch.float32 and np.array_equal(per.cpu().numpy(), nb["perspectives"].astype(np.float32))
    assert np.array_equal(pos.cpu().numpy(), nb["positions"])
    assert np.array_equal(off.cpu().numpy()[1:], nb["splice_idx"])
    acts, qv = T._selectActionBatch_prime(nb["sel_q"], nb["splice_idx"], nb["positions"], np.ones(st.shape[0], bool))
    assert acts.dtype == np.float64 and qv.dtype == np.float64
    assert np.array_equal(acts, nb["sel_actions"].astype(np.float64)) and np.array_equal(qv, nb["sel_qv"].astype(np.float64))
    # the same through the handle-bound kernel (EnvSet.selectAction), eps = 0
    env = T.make("toric-code-v0", {"size": d, "p_error": 0.1})
    hs = T.EnvSet(env, st.shape[0], seed=1, numpy_io=True)
    hs.resetAll()
    a2, q2 = hs.selectAction(nb["sel_q"], 0.0, positions=pos, offsets=off)
    assert np.array_equal(a2, nb["sel_actions"]) and np.array_equal(q2, nb["sel_qv"])
    hs.close()


@pytest.mark.parametrize("dtype", (torch.float16, torch.bfloat16, torch.uint8))
def test_output_dtypes(T, dtype):
    d, n = 7, 300
    gpu, ora = make_pair(T, d, n, numpy_io=False)
    gpu.resetAll()
    ora.resetAll()
    per, pos, cnt = gpu.generatePerspective(dtype=dtype)
    bp, bpos, bcnt, _ = O.generate_perspective_batch(ora.states)
    assert per.dtype == dtype
    assert np.array_equal(per.float().cpu().numpy(), bp.astype(np.float32))
    assert np.array_equal(pos.cpu().numpy(), bpos)


# ------------------------------------------------------------------ edge cases
def test_empty_full_and_single_lattice(T):
    d = 5
    env = T.make("toric-code-v0", {"size": d, "p_error": 0.1})
    gpu = T.EnvSet(env, 3, seed=1)
    q = np.zeros((3, 2, d, d), np.uint8)
    q[1, 0] = 3                      # Z everywhere on layer 0: every vertex touched twice -> empty
    q[2, 0, 1, 1] = 2                # one Y: 2 vertex + 2 plaquette defects
    gpu.setQubits(q)
    s = gpu.getStates()
    assert np.array_equal(s, O.syndrome(q)) and s[0].sum() == 0 and s[1].sum() == 0 and s[2].sum() == 4
    per, pos, cnt = gpu.generatePerspective()
    bp, bpos, bcnt, _ = O.generate_perspective_batch(O.syndrome(q))
    assert cnt.tolist() == bcnt.tolist() and cnt[0] == 0 and cnt[1] == 0
    assert np.array_equal(per, bp.astype(np.float32)) and np.array_equal(pos, bpos)
    act, qv = gpu.selectAction(np.zeros((per.shape[0], 3), np.float32), np.zeros(3))
    assert act[0].tolist() == [0, 0, 0, 0] and act[1].tolist() == [0, 0, 0, 0]   # empty syndrome: no legal action
    assert np.array_equal(gpu.isTerminal(), [True, True, False])
    assert np.array_equal(gpu.evalGroundState(), O.eval_ground_state(q))
    gpu.close()
    one = T.EnvSet(env, 1, seed=9)
    o1 = O.OracleEnvSet(d, 1, 0.1, seed=9)
    assert np.array_equal(one.resetAll(), o1.resetAll())
    per, pos, cnt = one.generatePerspective()
    bp, bpos, _, _ = O.generate_perspective_batch(o1.states)
    assert np.array_equal(per, bp.astype(np.float32)) and np.array_equal(pos, bpos)
    one.close()


def test_dense_syndrome_maximum_size(T):
    """Every qubit a hit: the maximum stack (2*d*d perspectives per lattice)."""
    d, n = 7, 64
    rng = np.random.default_rng(3)
    st = (rng.random((n, 2, d, d)) < 0.6).astype(np.uint8)
    st[0] = 1
    per, pos, cnt = T.generatePerspectiveBatch(d // 2, d, st, dtype=torch.uint8)
    bp, bpos, bcnt, _ = O.generate_perspective_batch(st)
    assert int(cnt[0]) == 2 * d * d
    assert np.array_equal(per.cpu().numpy(), bp) and np.array_equal(pos.cpu().numpy(), bpos)
    assert np.array_equal(cnt.cpu().numpy(), bcnt)


@pytest.mark.parametrize("d,dtype", [(19, torch.float32), (21, torch.uint8), (21, torch.bfloat16), (17, torch.float32)])
def test_lattice_whose_stack_is_larger_than_the_ring(T, d, dtype):
    """Every qubit a hit at d >= 19: 2d^2 perspectives of 2d^2 bits are more than the 64 KB bit ring of a workgroup holds
    (d=21: 778 k bits against 524 k).  The producers publish their progress after every pass of 64 hits and ask for
    room pass by pass, so such a lattice streams through the ring; mixed with sparse and empty lattices."""
    rng = np.random.default_rng(d)
    n = 300
    st = (rng.random((n, 2, d, d)) < 0.02).astype(np.uint8)
    st[::7] = 1                                                # dense: all 2 d^2 qubits are hits
    st[3::11] = 0
    st[5::13] = (rng.random((len(range(5, n, 13)), 2, d, d)) < 0.6).astype(np.uint8)
    per, pos, cnt = T.generatePerspectiveBatch(d // 2, d, st, dtype=dtype)
    bp, bpos, bcnt, _ = O.generate_perspective_batch(st)
    assert int(cnt[0]) == 2 * d * d                            # (d=21: 882 x 882 = 778 k bits; d=19: 521 k + the storers' lag)
    assert np.array_equal(cnt.cpu().numpy(), bcnt) and np.array_equal(pos.cpu().numpy(), bpos)
    assert np.array_equal(per.float().cpu().numpy(), bp.astype(np.float32))


@pytest.mark.parametrize("d", SIZES)
@pytest.mark.parametrize("dtype", (torch.float32, torch.float16, torch.uint8))
def test_line_ownership_with_empty_and_tiny_lattices(T, d, dtype):
    """The write kernel assigns whole 128-byte lines to the lattice that owns the line's first
    element; lines then hold elements of several lattices when these are empty or tiny.  Mix of
    empty syndromes, single defect pairs and dense grids, every size and element width."""
    rng = np.random.default_rng(1000 + d)
    n = 777
    st = np.zeros((n, 2, d, d), np.uint8)
    kind = rng.integers(0, 4, n)
    for e in range(n):
        if kind[e] == 1:                                   # one defect pair (a handful of hits)
            q = np.zeros((2, d, d), np.uint8)
            q[rng.integers(0, 2), rng.integers(0, d), rng.integers(0, d)] = rng.integers(1, 4)
            st[e] = O.syndrome(q)
        elif kind[e] == 2:
            st[e] = rng.random((2, d, d)) < 0.15
        elif kind[e] == 3:
            st[e] = rng.random((2, d, d)) < 0.7
    st[:5] = 0                                             # a run of empty lattices at the start ...
    st[-3:] = 0                                            # ... and at the end
    per, pos, cnt = T.generatePerspectiveBatch(d // 2, d, st, dtype=dtype)
    bp, bpos, bcnt, _ = O.generate_perspective_batch(st)
    assert np.array_equal(cnt.cpu().numpy(), bcnt) and np.array_equal(pos.cpu().numpy(), bpos)
    assert np.array_equal(per.float().cpu().numpy(), bp.astype(np.float32))
    # exact-capacity buffer with a canary behind it: nothing may be written past the stack
    P = int(bp.shape[0])
    nq = 2 * d * d
    env = T.make("toric-code-v0", {"size": d})
    gpu = T.EnvSet(env, n, numpy_io=False)
    q = np.zeros((n, 2, d, d), np.uint8)
    gpu.setQubits(q)                                       # counts come from the handle: all empty
    c0, off0 = gpu.perspectiveCounts()
    assert int(off0[-1].item()) == 0
    gpu.close()
    flat = torch.full((P * nq + 256,), 7, dtype=dtype, device=per.device)
    from toric_rl_decoder_amd import _lib
    import ctypes as C
    L = _lib.load()
    dev_st = torch.as_tensor(st, device=per.device)
    offs = torch.empty(n + 1, dtype=torch.int64, device=per.device)
    _lib.check(L.tq_states_persp_count(d, n, C.c_void_p(dev_st.data_ptr()), None, C.c_void_p(offs.data_ptr()), None))
    code = {torch.float32: 0, torch.float16: 1, torch.uint8: 3}[dtype]
    pflat = torch.full((P * 3 + 64,), -5, dtype=torch.int32, device=per.device)
    _lib.check(L.tq_states_persp_write(d, n, C.c_void_p(dev_st.data_ptr()), C.c_void_p(offs.data_ptr()),
                                       C.c_void_p(flat.data_ptr()), C.c_void_p(pflat.data_ptr()), P, code, None))
    torch.cuda.synchronize()
    assert np.array_equal(flat[:P * nq].float().cpu().numpy(), bp.astype(np.float32).reshape(-1))
    assert bool((flat[P * nq:] == 7).all())
    assert np.array_equal(pflat[:P * 3].cpu().numpy(), bpos.reshape(-1)) and bool((pflat[P * 3:] == -5).all())
    # a stack without positions
    flat.fill_(7)
    _lib.check(L.tq_states_persp_write(d, n, C.c_void_p(dev_st.data_ptr()), C.c_void_p(offs.data_ptr()),
                                       C.c_void_p(flat.data_ptr()), None, P, code, None))
    torch.cuda.synchronize()
    assert np.array_equal(flat[:P * nq].float().cpu().numpy(), bp.astype(np.float32).reshape(-1))
    assert bool((flat[P * nq:] == 7).all())


def test_bad_actions_and_capacity_are_reported(T):
    d, n = 5, 100
    gpu, ora = make_pair(T, d, n)
    gpu.resetAll()
    bad = np.zeros((n, 4), np.int64)
    bad[:, 3] = 1
    bad[7] = (0, d, 0, 1)                                  # row out of range
    with pytest.raises(ValueError):
        gpu.step(bad)
    bad[7] = (0, 0, 0, 4)                                  # op outside 1..3
    with pytest.raises(ValueError):
        gpu.step(bad)
    with pytest.raises(ValueError):
        gpu.step(np.zeros((n - 1, 4), np.int64))
    with pytest.raises(ValueError):
        gpu.resetTerminalEnvs([0, 0])
    with pytest.raises(ValueError):
        gpu.resetTerminalEnvs([n])
    st = np.zeros((4, 2, d, d), np.uint8)
    with pytest.raises(ValueError):                        # stateless drop-in reports bad actions too
        T.generateTransitionParallel(np.array([[0, 0, 0, 1], [2, 0, 0, 1], [0, 0, 0, 1], [0, 0, 0, 1]]), np.zeros(4), st, st,
                                     np.zeros(4, bool), d // 2)
    T.generateTransitionParallel(np.array([[0, 0, 0, 1]] * 4), np.zeros(4), st, st, np.zeros(4, bool), d // 2)
    cnt, off = gpu.perspectiveCounts()
    P = int(off[-1].item())
    small = torch.zeros((P - 1, 2, d, d), dtype=torch.float32, device=gpu.device)
    gpu.writePerspectives(small)
    with pytest.raises(T.ToricEnvError):
        gpu.check()
    gpu.check()                                            # latch is cleared by the read
    gpu.close()


# ------------------------------------------------------------------ fused actor step
@pytest.mark.parametrize("d,strategy", [(3, "random"), (5, "linear"), (7, "fixed"), (9, "random"), (13, "linear"), (15, "random"), (17, "fixed"), (21, "random")])
def test_fused_actor_step_matches_oracle_loop(T, d, strategy):
    """tq_actor_step (step -> transition -> scheduled auto-reset -> counts) against the same loop
    spelled out with oracle calls in the order of Actor_mp.py:104-185."""
    n, T_steps, max_steps = 700, 40, 9
    p0 = P_OF[d]
    gpu, ora = make_pair(T, d, n, p=p0, seed=77, first=1000, numpy_io=False, max_steps_per_episode=max_steps)
    p_start, p_final, p_delta = 0.05, 0.2, 0.03
    gpu.set_perror_schedule(strategy, p_start, p_final, p_delta)
    roof = np.full(n, p_start)
    gpu.resetAll()
    ora.resetAll()
    blk = gpu.newTransitionBlock(steps=T_steps)
    log = []
    for t in range(T_steps):
        cnt, off = gpu.perspectiveCounts()
        bp, bpos, bcnt, boff = O.generate_perspective_batch(ora.states)
        assert np.array_equal(cnt.cpu().numpy(), bcnt) and np.array_equal(off.cpu().numpy(), boff)
        oact, _, _ = O.select_action_batch(np.zeros((bp.shape[0], 3), np.float32), boff, bpos, 1.0, ora.seed,
                                           ora.env_ids, ora.episodes, ora.steps)
        act, rew, term = gpu.actorStep(None, block=blk, slot=t)
        prev = ora.states.copy()
        ons, orew, oterm, _ = ora.step(oact)
        assert np.array_equal(act.cpu().numpy(), oact)
        assert np.array_equal(rew.cpu().numpy(), orew.astype(np.float32))
        assert np.array_equal(term.cpu().numpy().astype(bool), oterm)
        tper, tact, tnper = O.generate_transition_batch(oact, prev, ora.states)
        log.append((tper, tact, tnper, orew.astype(np.float32), oterm))
        idx = np.nonzero(oterm | (ora.steps > max_steps))[0]
        if idx.size:
            if strategy == "fixed":
                p_new = np.full(idx.size, p0)
            else:
                roof[idx] = np.minimum(p_final, roof[idx] + p_delta)
                p_new = O.perror_schedule(ora.seed, ora.env_ids[idx], ora.episodes[idx], p_start, roof[idx], strategy)
            ora.resetTerminalEnvs(idx, p_new)
        assert np.array_equal(gpu.getStates().cpu().numpy(), ora.states)
        assert np.array_equal(gpu.getQubits().cpu().numpy(), ora.qubits)
    ep, st = gpu.getCounters()
    assert np.array_equal(ep.cpu().numpy(), ora.episodes) and np.array_equal(st.cpu().numpy(), ora.steps)
    assert ora.episodes.max() >= 3
    for t in (0, T_steps // 2, T_steps - 1):
        u = blk.unpack(first=t * n, count=n)
        tper, tact, tnper, orew, oterm = log[t]
        assert np.array_equal(u["perspective"].cpu().numpy(), tper)
        assert np.array_equal(u["next_perspective"].cpu().numpy(), tnper)
        assert np.array_equal(u["action"].cpu().numpy(), tact)
        assert np.array_equal(u["reward"].cpu().numpy(), orew)
        assert np.array_equal(u["terminal"].cpu().numpy().astype(bool), oterm)
    rec = T.to_structured(blk.unpack(first=0, count=n), d)
    assert rec.dtype == T.transition_dtype(d) and np.array_equal(rec["reward"], log[0][3].astype(np.float64))
    gpu.close()


def test_fused_step_with_given_actions_equals_unfused(T):
    d, n = 7, 1000
    a, _ = make_pair(T, d, n, seed=5, numpy_io=False, max_steps_per_episode=1000)
    b, _ = make_pair(T, d, n, seed=5, numpy_io=False, max_steps_per_episode=1000)
    a.resetAll()
    b.resetAll()
    for t in range(5):
        per, pos, cnt = a.generatePerspective(dtype=torch.uint8)
        b.perspectiveCounts()
        act, _ = a.selectAction(None, 1.0)
        act = act.clone()
        s1, r1, t1, _ = a.step(act)
        r1, t1 = r1.clone(), t1.clone()
        _, r2, t2 = b.actorStep(act)
        assert torch.equal(r1, r2) and torch.equal(t1, t2)
        live = ~t1.bool()
        assert torch.equal(a.getStates()[live], b.getStates()[live])
        idx = torch.nonzero(t1).flatten().int()
        if idx.numel():
            a.resetTerminalEnvs(idx)
        assert torch.equal(a.getStates(), b.getStates()) and torch.equal(a.getQubits(), b.getQubits())
    a.close()
    b.close()


def test_sharding_is_partition_invariant(T):
    """Global env ids key the RNG: two shards of 128 lattices == one handle of 256."""
    d = 7
    whole, _ = make_pair(T, d, 256, seed=99, first=0, numpy_io=False)
    lo, _ = make_pair(T, d, 128, seed=99, first=0, numpy_io=False)
    hi, _ = make_pair(T, d, 128, seed=99, first=128, numpy_io=False)
    for e in (whole, lo, hi):
        e.resetAll()
    for t in range(12):
        aw, rw, tw = (x.clone() for x in whole.actorStep(None))
        al, rl, tl = (x.clone() for x in lo.actorStep(None))
        ah, rh, th = (x.clone() for x in hi.actorStep(None))
        assert torch.equal(aw, torch.cat((al, ah))) and torch.equal(rw, torch.cat((rl, rh)))
        assert torch.equal(whole.getStates(), torch.cat((lo.getStates().clone(), hi.getStates().clone())))
    for e in (whole, lo, hi):
        e.close()


# ------------------------------------------------------------------ single-env facade
def test_toric_env_facade(T):
    d = 5
    env = T.make("toric-code-v0", {"size": d, "min_qubit_errors": 0, "p_error": 0.1}, seed=3)
    ora = O.OracleEnvSet(d, 1, 0.1, seed=3)
    s = env.reset()
    assert np.array_equal(s, ora.resetAll()[0]) and not env.isTerminalState(s)
    assert np.array_equal(env.qubit_matrix, ora.qubits[0]) and np.array_equal(env.state, s)
    per, pos, _, _ = O.generate_perspective_batch(ora.states)
    a = [int(pos[0][0]), int(pos[0][1]), int(pos[0][2]), 2]
    ns, r, t, _ = env.step(a)
    ons, orew, oterm, _ = ora.step(np.array([a]))
    assert np.array_equal(ns, ons[0]) and r == orew[0] and t == bool(oterm[0])
    s2 = env.reset(p_error=0.2)
    assert np.array_equal(s2, ora.resetTerminalEnvs([0], [0.2])[0])
    qm = np.zeros((2, d, d), np.int64)
    qm[0, 2, :] = 1
    assert np.array_equal(env.createSyndromOpt(qm), O.syndrome(qm)) and env.isTerminalState(env.createSyndromOpt(qm))
    env.qubit_matrix = qm                                   # small_p_error_test.py:119-120 pattern
    assert env.isTerminalState(env.state) and env.evalGroundState() is False
    env.qubit_matrix = np.zeros((2, d, d), np.int64)
    assert env.evalGroundState() is True


# ------------------------------------------------------------------ BASELINE full sizes
@pytest.mark.parametrize("d,p,n", [(7, 0.10, 65536), (9, 0.15, 65536)])
def test_full_size_configs(T, d, p, n):
    """configs[2] / configs[3]: exact counts/offsets/positions AND the exact stack against the oracle
    for all 65 536 lattices, plus size-independent properties on the whole stack:
    every perspective is a permutation of its lattice's syndrome (equal defect count) and has a
    defect on one of the four centre checks (centred-frame property)."""
    gpu, _ = make_pair(T, d, n, p=p, seed=2020, numpy_io=False)
    gpu.resetAll()
    for _ in range(3):
        gpu.actorStep(None)
    states = gpu.getStates().clone()
    st_np = states.cpu().numpy()
    assert np.array_equal(O.syndrome(gpu.getQubits().cpu().numpy()), st_np)
    per, pos, cnt = gpu.generatePerspective(dtype=torch.float32)
    hm = O.hit_masks(st_np).reshape(n, -1)
    ocnt = hm.sum(1).astype(np.int32)
    assert np.array_equal(cnt.cpu().numpy(), ocnt)
    off = np.zeros(n + 1, np.int64)
    np.cumsum(ocnt, out=off[1:])
    assert per.shape[0] == off[-1]
    # positions: exact for all lattices
    env_idx, hit = np.nonzero(hm)
    layer, rem = np.divmod(hit, d * d)
    assert np.array_equal(pos.cpu().numpy(), np.stack((layer, rem // d, rem % d), 1))
    # properties over the full stack (on the device)
    defects = states.reshape(n, -1).sum(1).to(torch.float32)
    owner = torch.repeat_interleave(torch.arange(n, device=per.device), cnt.long())
    assert torch.equal(per.reshape(per.shape[0], -1).sum(1), defects[owner])
    gs = d // 2
    centre = per[:, 0, gs, gs] + per[:, 0, gs + 1, gs] + per[:, 1, gs, gs] + per[:, 1, gs, gs - 1]
    assert bool((centre > 0).all())
    assert bool(((per == 0) | (per == 1)).all())
    # exact compare of the WHOLE stack, all 65 536 lattices, against the C oracle (OpenMP, seconds):
    # the oracle's u8 stack is uploaded and compared on the device in chunks
    from oracle.c_oracle import CEnvBatch
    ce = CEnvBatch(d, n, p, seed=2020)
    cper, cpos, ccnt, coff = ce.perspectives(states=st_np, dtype=np.uint8)
    assert np.array_equal(ccnt, ocnt) and np.array_equal(coff, off) and np.array_equal(cpos, pos.cpu().numpy())
    assert cper.shape[0] == per.shape[0]
    step = 1 << 20
    for i in range(0, cper.shape[0], step):
        want = torch.as_tensor(cper[i:i + step], device=per.device)
        assert torch.equal(per[i:i + step], want.to(torch.float32)), f"stack differs in perspectives [{i}, {i + step})"
    # and the numpy batch oracle on a strided subset (ties the C oracle to the golden-pinned numpy one here too)
    sel = np.unique(np.concatenate((np.arange(0, n, 97), [n - 1])))
    bp, _, _, _ = O.generate_perspective_batch(st_np[sel])
    rows = np.concatenate([np.arange(off[e], off[e + 1]) for e in sel])
    assert np.array_equal(cper[rows], bp)
    gpu.close()


@pytest.mark.parametrize("d,n,steps", [(3, 3000, 400), (5, 4096, 300), (7, 2048, 200), (9, 1024, 160)])
def test_long_run_matches_c_oracle_actor_loop(T, d, n, steps):
    """Soak: hundreds of fused exploration steps (selection, step, auto-reset at 75 steps or when solved;
    d=3 needs several redraw rounds per reset) against the C oracle's actor loop; the whole lattice state
    and every counter must match at the end, and the perspective totals along the way."""
    from oracle.c_oracle import CEnvBatch
    p = P_OF[d]
    env = T.make("toric-code-v0", {"size": d, "p_error": p})
    gpu = T.EnvSet(env, n, seed=606, first_env_id=17, numpy_io=False)
    ce = CEnvBatch(d, n, p, seed=606, first_env_id=17)
    gpu.resetAll()
    ce.reset()
    p_gpu = torch.zeros((), dtype=torch.int64, device=gpu.device)
    for t in range(steps):
        cnt, off = gpu.perspectiveCounts()
        p_gpu += off[-1]
        gpu.actorStep(None, want_actions=False)
    P, _ = ce.actor_steps(steps)
    assert int(p_gpu.item()) == P
    assert np.array_equal(gpu.getQubits().cpu().numpy(), ce.qubits)
    assert np.array_equal(gpu.getStates().cpu().numpy(), ce.states)
    ep, st = gpu.getCounters()
    assert np.array_equal(ep.cpu().numpy().astype(np.uint32), ce.episodes) and np.array_equal(st.cpu().numpy().astype(np.uint32), ce.steps)
    assert ce.episodes.min() >= 2 and ce.episodes.max() >= 3          # every lattice was reset at least once
    gpu.check()
    gpu.close()


# ------------------------------------------------------------------ BASELINE configs[0] on the HIP path
def test_config0_single_env_d3_100_random_steps(T):
    """configs[0] (1 env, d=3, p_error=0.1, 100 random-action steps through the EnvSet surface, numpy
    in / numpy out) on the HIP path, every array of every step against the oracle."""
    d = 3
    gpu, ora = make_pair(T, d, 1, p=0.1, seed=11)
    assert np.array_equal(gpu.resetAll(), ora.resetAll())
    rng = np.random.default_rng(0)
    episodes = 0
    for t in range(100):
        per, pos, cnt = gpu.generatePerspective()
        bp, bpos, bcnt, boff = O.generate_perspective_batch(ora.states)
        assert np.array_equal(per, bp.astype(np.float32)) and np.array_equal(pos, bpos) and np.array_equal(cnt, bcnt)
        a = random_actions_from(bpos, boff, rng)                  # eps = 1 branch of _selectActionBatch_prime
        ns, r, term, _ = gpu.step(a)
        ons, orr, oterm, _ = ora.step(a)
        assert np.array_equal(ns, ons) and np.array_equal(r, orr) and np.array_equal(term, oterm)
        assert np.array_equal(gpu.getQubits(), ora.qubits)
        if term[0] or ora.steps[0] > 75:
            assert np.array_equal(gpu.resetTerminalEnvs([0]), ora.resetTerminalEnvs([0]))
            episodes += 1
    assert episodes >= 1
    gpu.close()


# ------------------------------------------------------------------ ABI hygiene
def test_reset_idx_is_validated_on_the_device(T):
    """Duplicate or out-of-range indices are detected by the kernel itself (device-tensor path, no
    host-side check): the lattice is reset once, the rest are untouched, and tq_check reports it."""
    d, n = 5, 300
    gpu, ora = make_pair(T, d, n, seed=5, numpy_io=False)
    gpu.resetAll()
    ora.resetAll()
    before = gpu.getStates().clone()
    idx = torch.tensor([4, 9, 4, 17], dtype=torch.int32, device=gpu.device)
    gpu.resetTerminalEnvs(idx)
    with pytest.raises(ValueError, match="more than once"):
        gpu.check()
    ora.resetTerminalEnvs([4, 9, 17])
    assert np.array_equal(gpu.getStates().cpu().numpy(), ora.states)        # 4 was reset exactly once
    ep, _ = gpu.getCounters()
    assert np.array_equal(ep.cpu().numpy(), ora.episodes)
    gpu.check()                                                            # latch cleared
    gpu.resetTerminalEnvs(torch.tensor([1, n, -3], dtype=torch.int32, device=gpu.device))
    with pytest.raises(ValueError, match="outside"):
        gpu.check()
    ora.resetTerminalEnvs([1])
    assert np.array_equal(gpu.getStates().cpu().numpy(), ora.states)
    # the same index in two different calls is fine
    gpu.resetTerminalEnvs(torch.tensor([4], dtype=torch.int32, device=gpu.device))
    gpu.resetTerminalEnvs(torch.tensor([4], dtype=torch.int32, device=gpu.device))
    gpu.check()
    assert not torch.equal(before, gpu.getStates())
    gpu.close()


def test_p_error_zero_is_rejected_or_latched(T):
    import ctypes as C
    d, n = 5, 64
    env = T.make("toric-code-v0", {"size": d, "p_error": 0.1})
    gpu = T.EnvSet(env, n, seed=1, numpy_io=False)
    L = T.load()
    assert L.tq_set_params(gpu._h, 0.0, 100.0, 75) == -1 and b"(0,1]" in L.tq_last_error()
    with pytest.raises(ValueError):
        gpu.set_perror_schedule("linear", 0.0, 0.2, 0.01)
    # p = 0 smuggled in through the device array: bounded number of rounds, then TQ_E_RESET is latched
    gpu.resetAll(torch.zeros(n, dtype=torch.float64, device=gpu.device))
    with pytest.raises(T.ToricEnvError, match="rounds"):
        gpu.check()
    gpu.resetAll()
    gpu.check()
    assert C.c_int(L.tq_version()).value == 200
    gpu.close()


def test_every_slot_is_written_and_noop_slots_are_empty(T):
    """tq_actor_step writes its slot every step: a lattice given a no-op leaves an EMPTY slot (action
    word 0, zero planes), never the record of an earlier flush; wire.decode drops it."""
    from toric_rl_decoder_amd import wire
    d, n = 7, 500
    gpu, ora = make_pair(T, d, n, seed=12, numpy_io=False, max_steps_per_episode=1000)
    gpu.resetAll()
    blk = gpu.newTransitionBlock(steps=1)
    gpu.perspectiveCounts()
    gpu.actorStep(None, block=blk, slot=0)                    # flush 1: every slot holds a transition
    assert (blk.unpack()["action"][:, 3] >= 1).all()
    per, pos, cnt = gpu.generatePerspective(dtype=torch.uint8)
    act, _ = gpu.selectAction(None, 1.0, positions=pos)
    act = act.clone()
    hole = torch.arange(0, n, 7, device=gpu.device)
    act[hole] = 0                                             # no-op for every 7th lattice
    gpu.actorStep(act, block=blk, slot=0)                     # flush 2 re-uses the block
    gpu.check()                                               # a no-op is not an error
    u = blk.unpack()
    a = u["action"].cpu().numpy()
    h = hole.cpu().numpy()
    assert (a[h] == 0).all() and not u["perspective"][hole].any() and not u["next_perspective"][hole].any()
    assert not u["reward"][hole].any() and not u["terminal"][hole].any()
    dec = wire.decode(blk.buf.cpu().numpy(), d, n)
    assert dec["perspective"].shape[0] == n - h.size and np.array_equal(dec["slot"], np.setdiff1d(np.arange(n), h))
    gpu.close()


def test_alignment_and_scratch_contract(T):
    """16-byte alignment of vector-accessed arguments is checked (TQ_E_INVALID, no kernel launched);
    the stateless scratch is sized by tq_states_reserve and never re-allocated by a hot-path call;
    tq_create leaves the caller's current device alone."""
    import ctypes as C
    L = T.load()
    d, n = 7, 256
    cur = torch.cuda.current_device()
    gpu, _ = make_pair(T, d, n, seed=2, numpy_io=False)
    assert torch.cuda.current_device() == cur
    gpu.resetAll()
    offs = torch.zeros(n + 3, dtype=torch.int64, device=gpu.device)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = L.tq_persp_count(gpu._h, None, C.c_void_p(offs.data_ptr() + 8), stream)
    assert rc == -1 and b"16-byte" in L.tq_last_error()
    assert L.tq_persp_count(gpu._h, None, C.c_void_p(offs.data_ptr() + 16), stream) == 0
    acts = torch.zeros(4 * n + 4, dtype=torch.int32, device=gpu.device)
    assert L.tq_step(gpu._h, C.c_void_p(acts.data_ptr() + 4), None, None, stream) == -1
    out = torch.zeros(n * 98 * 98 + 64, dtype=torch.uint8, device=gpu.device)
    rc = L.tq_persp_write(gpu._h, C.c_void_p(offs.data_ptr() + 16), C.c_void_p(out.data_ptr() + 4), None, n * 98, 3, stream)
    assert rc == -1
    # stateless scratch: too small -> error code, not a re-allocation in the middle of the stream
    big = 1 << 16
    st = torch.zeros((big, 2, 11, 11), dtype=torch.uint8, device=gpu.device)
    o2 = torch.zeros(big + 1, dtype=torch.int64, device=gpu.device)
    rc = L.tq_states_persp_count(11, big, C.c_void_p(st.data_ptr()), None, C.c_void_p(o2.data_ptr()), stream)
    if rc != 0:                                               # (0 only if an earlier test already reserved this much)
        assert rc == -3 and b"tq_states_reserve" in L.tq_last_error()
    assert L.tq_states_reserve(11, big) == 0
    assert L.tq_states_persp_count(11, big, C.c_void_p(st.data_ptr()), None, C.c_void_p(o2.data_ptr()), stream) == 0
    torch.cuda.synchronize()
    assert int(o2[-1]) == 0
    gpu.close()


def test_generate_perspective_of_explicit_states(T):
    """EnvSet.generatePerspective(states=...): the reference function's explicit-state form."""
    d, n = 5, 200
    gpu, ora = make_pair(T, d, n, seed=3)
    gpu.resetAll()
    _, other = O.reset_lattices(77, np.arange(50), 0, 0.2, d)
    per, pos, cnt = gpu.generatePerspective(states=other.astype(np.int64))
    bp, bpos, bcnt, _ = O.generate_perspective_batch(other)
    assert per.dtype == np.float32 and np.array_equal(per, bp.astype(np.float32))
    assert np.array_equal(pos, bpos) and np.array_equal(cnt, bcnt)
    gpu.close()


@pytest.mark.parametrize("d,n_err", [(3, 1), (5, 3), (7, 4), (9, 162)])
def test_min_qubit_errors_sampler(T, d, n_err):
    """gym config "min_qubit_errors" = n > 0: every reset (resetAll, indexed, and the auto-reset fused
    into the actor step) places exactly n errors; bit-exact against the oracle's fixed-n sampler."""
    n = 700
    env = T.make("toric-code-v0", {"size": d, "min_qubit_errors": n_err, "p_error": 0.1})
    gpu = T.EnvSet(env, n, seed=9, first_env_id=3, numpy_io=False, max_steps_per_episode=4)
    ora = O.OracleEnvSet(d, n, 0.1, seed=9, first_env_id=3, min_qubit_errors=n_err)
    gpu.resetAll()
    ora.resetAll()
    q = gpu.getQubits().cpu().numpy()
    assert np.array_equal(q, ora.qubits) and ((q != 0).reshape(n, -1).sum(1) == n_err).all()
    idx = np.arange(5, n, 9)
    gpu.resetTerminalEnvs(torch.as_tensor(idx, dtype=torch.int32, device=gpu.device))
    ora.resetTerminalEnvs(idx)
    assert np.array_equal(gpu.getQubits().cpu().numpy(), ora.qubits)
    for t in range(12):                                       # fused steps: lattices time out after 4 steps and are re-drawn
        bp, bpos, bcnt, boff = O.generate_perspective_batch(ora.states)
        oact, _, _ = O.select_action_batch(np.zeros((bp.shape[0], 3), np.float32), boff, bpos, 1.0, ora.seed,
                                           ora.env_ids, ora.episodes, ora.steps)
        gpu.perspectiveCounts()
        act, rew, term = gpu.actorStep(None)
        _, orew, oterm, _ = ora.step(oact)
        assert np.array_equal(act.cpu().numpy(), oact) and np.array_equal(term.cpu().numpy().astype(bool), oterm)
        ridx = np.nonzero(oterm | (ora.steps > 4))[0]
        if ridx.size:
            ora.resetTerminalEnvs(ridx)
        assert np.array_equal(gpu.getQubits().cpu().numpy(), ora.qubits)
        assert np.array_equal(gpu.getStates().cpu().numpy(), ora.states)
    assert ora.episodes.max() >= 3
    gpu.check()
    gpu.close()


@pytest.mark.parametrize("d", (3, 7, 9))
def test_stack_written_in_lattice_ranges(T, d):
    """tq_persp_write_range: the batch walked in ragged chunks of lattices (chunk borders fall inside
    128-byte lines of the one-shot stack; empty lattices at the borders), every chunk into the same
    small buffer: the concatenation is the one-shot stack, positions included, and nothing is written
    past a chunk's own perspectives."""
    n = 1000
    gpu, ora = make_pair(T, d, n, seed=21, numpy_io=False)
    gpu.resetAll()
    q = gpu.getQubits().clone()
    q[100:110] = 0                                            # a run of empty lattices, one chunk border inside it
    q[399] = 0
    gpu.setQubits(q)
    per, pos, cnt = gpu.generatePerspective(dtype=torch.float32)
    off = gpu._offsets.clone()
    offh = off.cpu().numpy()
    nq = 2 * d * d
    cuts = [0, 1, 105, 400, 401, 777, n]
    for a, b in zip(cuts, cuts[1:]):
        k = int(offh[b] - offh[a])
        buf = torch.full((k * nq + 300,), 7.0, dtype=torch.float32, device=gpu.device)
        pbuf = torch.full((3 * k + 70,), -5, dtype=torch.int32, device=gpu.device)
        gpu.writePerspectives(buf[:k * nq].view(k, 2, d, d) if k else buf[:0].view(0, 2, d, d), pbuf, off, first=a, count=b - a)
        gpu.check()
        assert torch.equal(buf[:k * nq], per[offh[a]:offh[b]].reshape(-1)) and bool((buf[k * nq:] == 7).all())
        assert torch.equal(pbuf[:3 * k], pos[offh[a]:offh[b]].reshape(-1)) and bool((pbuf[3 * k:] == -5).all())
    with pytest.raises(ValueError):
        gpu.writePerspectives(per, pos, off, first=n - 5, count=6)
    gpu.close()


def test_fresh_handle_is_usable_at_once_on_a_non_blocking_stream(T):
    """tq_create allocates AND synchronises: a handle created while another one is busy can be reset at once on a
    non-blocking stream (PyTorch side streams are) -- its counters must already be zero, or the Philox counters of
    the first episode would be wrong."""
    d, n = 7, 65536
    busy, _ = make_pair(T, d, n, seed=8, numpy_io=False)
    busy.resetAll()
    side = torch.cuda.Stream()                                # hipStreamNonBlocking
    for rep in range(4):
        for _ in range(6):
            busy.actorStep(None, want_actions=False)          # keep the null-stream side of the device busy
        with torch.cuda.stream(side):
            fresh, ora = make_pair(T, d, n, seed=100 + rep, first=5 * n, numpy_io=False)
            fresh.resetAll()
            st = fresh.getStates().clone()
            ep, steps = fresh.getCounters()
            ep, steps = ep.clone(), steps.clone()
        side.synchronize()
        assert np.array_equal(st.cpu().numpy(), ora.resetAll())
        assert bool((ep == 1).all()) and bool((steps == 0).all())
        fresh.close()
    busy.close()


def test_fixed_n_sampler_accepts_p_error_zero(T):
    """{min_qubit_errors: n, p_error: 0} is a valid config (the small-p regime of results/small_p_error_test.py):
    the fixed-n sampler does not use p_error; with the depolarizing sampler p_error = 0 stays rejected."""
    d, n = 5, 200
    env = T.make("toric-code-v0", {"size": d, "min_qubit_errors": 3, "p_error": 0.0})
    gpu = T.EnvSet(env, n, seed=4, numpy_io=False)
    ora = O.OracleEnvSet(d, n, 0.0, seed=4, min_qubit_errors=3)
    gpu.resetAll()
    ora.resetAll()
    assert np.array_equal(gpu.getQubits().cpu().numpy(), ora.qubits)
    gpu.check()
    L = T.load()
    assert L.tq_set_min_qubit_errors(gpu._h, 0) == -1 and b"depolarizing" in L.tq_last_error()   # would leave p = 0 in charge
    gpu.close()
    with pytest.raises(ValueError):
        T.EnvSet(T.make("toric-code-v0", {"size": d, "min_qubit_errors": 0, "p_error": 0.0}), n)
    # createSyndromOpt re-uses one scratch lattice
    e1 = T.make("toric-code-v0", {"size": d, "p_error": 0.1})
    qm = np.zeros((2, d, d), np.int64)
    qm[1, 3, 1] = 2
    a = e1.createSyndromOpt(qm)
    scratch = e1._scratch
    b = e1.createSyndromOpt(np.zeros((2, d, d), np.int64))
    assert e1._scratch is scratch and np.array_equal(a, O.syndrome(qm)) and not b.any()


def test_stack_ranges_whose_lattice_count_is_a_multiple_of_64(T):
    """Lattice ranges of 64, 128, 4096 lattices (the cut-point search of a range probes 64 lattices per round: spans
    that are multiples of 64 once left the range's last lattice unprobed) -- every range equals its part of the
    one-shot stack."""
    d, n = 7, 8256
    gpu, _ = make_pair(T, d, n, seed=33, numpy_io=False)
    gpu.resetAll()
    for _ in range(2):
        gpu.actorStep(None, want_actions=False)
    per, pos, cnt = gpu.generatePerspective(dtype=torch.float32)
    off = gpu._offsets.clone()
    offh = off.cpu().numpy()
    nq = 2 * d * d
    cuts = [0, 64, 192, 4288, 4352, 8256 - 64, 8256]
    for a, b in zip(cuts, cuts[1:]):
        k = int(offh[b] - offh[a])
        buf = torch.full((k * nq + 300,), 7.0, dtype=torch.float32, device=gpu.device)
        pbuf = torch.full((3 * k + 70,), -5, dtype=torch.int32, device=gpu.device)
        gpu.writePerspectives(buf[:k * nq].view(k, 2, d, d), pbuf, off, first=a, count=b - a)
        gpu.check()
        assert torch.equal(buf[:k * nq], per[offh[a]:offh[b]].reshape(-1)) and bool((buf[k * nq:] == 7).all()), (a, b)
        assert torch.equal(pbuf[:3 * k], pos[offh[a]:offh[b]].reshape(-1)) and bool((pbuf[3 * k:] == -5).all()), (a, b)
    gpu.close()


def test_chunked_stack_buffer_and_placement_probe(T):
    """T.alloc_stack (tq_stack_alloc: 2 MiB physical chunks behind one virtual range) is an ordinary device buffer for
    the stack write -- same bytes as a torch.empty buffer, nothing written past the stack -- and
    EnvSet.pickStackBuffer returns one of its candidates with the report of the probe."""
    import ctypes as C
    d, n = 7, 3000
    gpu, ora = make_pair(T, d, n, seed=15, numpy_io=False)
    gpu.resetAll()
    ora.resetAll()
    cnt, off = gpu.perspectiveCounts()
    P = int(off[-1].item())
    bp, bpos, _, _ = O.generate_perspective_batch(ora.states)
    cap = P + 3
    ref = torch.full((cap, 2, d, d), 7.0, dtype=torch.float32, device=gpu.device)
    chk = T.alloc_stack(cap, d, torch.float32, gpu.device)
    assert chk.shape == (cap, 2, d, d) and chk.is_cuda and chk.data_ptr() % (2 << 20) == 0
    chk.fill_(7.0)
    pos = torch.empty((cap, 3), dtype=torch.int32, device=gpu.device)
    gpu.writePerspectives(ref, pos, off)
    gpu.writePerspectives(chk, pos, off)
    gpu.check()
    assert torch.equal(ref, chk) and bool((chk[P:] == 7).all())
    assert np.array_equal(chk[:P].cpu().numpy(), bp.astype(np.float32)) and np.array_equal(pos[:P].cpu().numpy(), bpos)
    for dt in (torch.bfloat16, torch.uint8):
        c2 = T.alloc_stack(P, d, dt, gpu.device)
        gpu.writePerspectives(c2, None, off)
        assert np.array_equal(c2.float().cpu().numpy(), bp.astype(np.float32))
        del c2
    best, rep = gpu.pickStackBuffer(4, capacity=cap, positions=pos)
    assert rep["candidates"] == 4 and len(rep["write_ms"]) == 4 and 0 <= rep["chosen"] < 4
    assert rep["kinds"][0] == "torch.empty" and "2 MiB" in rep["kinds"][1] and best.shape == (cap, 2, d, d)
    gpu.writePerspectives(best, pos, off)
    assert torch.equal(best[:P], ref[:P])
    L = T.load()
    assert L.tq_stack_free(C.c_void_p(ref.data_ptr())) == -1 and b"tq_stack_alloc" in L.tq_last_error()   # not one of ours
    assert L.tq_stack_free(None) == 0
    # allocate-and-free cycles return the memory, failures too
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(4):
        w = T.alloc_chunked((300 << 20,), torch.uint8, gpu.device)
        assert w.data_ptr() % (2 << 20) == 0 and bool((w == 0).all())             # zero-filled
        w[::4097] = 7
        assert int(w[::4097].sum().item()) == 7 * len(w[::4097])
        del w
    tiny = T.alloc_chunked((3, 5), torch.int32, gpu.device)
    tiny[:] = 5
    assert int(tiny.sum().item()) == 75
    del tiny
    p_ = C.c_void_p(None)
    assert L.tq_stack_alloc(0, 0, C.byref(p_)) == -1                    # bytes == 0
    assert L.tq_stack_alloc(0, 1 << 46, C.byref(p_)) in (-1, -2) and not p_.value     # 64 TiB: refused, nothing left behind
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    assert abs(free1 - free0) < (256 << 20), (free0, free1)
    del chk, best
    gpu.close()


@pytest.mark.parametrize("d,strategy", [(5, "linear"), (7, "random"), (13, "fixed")])
def test_hip_equals_the_host_twin_call_for_call(T, d, strategy):
    """The same ABI calls on the GPU (libtoricenv.so) and on host memory (oracle/host_twin.cpp, the twin built on the
    product's csrc/lattice.hpp): states, qubits, counters, counts, offsets, f32 stack, positions and the packed
    transition block BYTE FOR BYTE after 24 fused steps with the p_error schedule."""
    from oracle import host_twin as H
    n, steps, max_steps = 3000, 24, 7
    p0 = P_OF[d]
    gpu, _ = make_pair(T, d, n, p=p0, seed=91, first=4000, numpy_io=False, max_steps_per_episode=max_steps)
    tw = H.HostEnvSet(d, n, p_error=p0, seed=91, first_env_id=4000, max_steps_per_episode=max_steps)
    gpu.set_perror_schedule(strategy, 0.04, 0.22, 0.02)
    tw.set_perror_schedule(strategy, 0.04, 0.22, 0.02)
    assert np.array_equal(gpu.resetAll().cpu().numpy(), tw.reset_all())
    gblk = gpu.newTransitionBlock(steps=steps)
    hblk, hcap = tw.new_block(steps=steps)
    for t in range(steps):
        per, pos, cnt = gpu.generatePerspective()
        hper, hpos, hcnt, hoff = tw.perspectives(np.float32)
        assert np.array_equal(cnt.cpu().numpy(), hcnt) and np.array_equal(gpu._offsets.cpu().numpy(), hoff)
        assert np.array_equal(per.cpu().numpy(), hper) and np.array_equal(pos.cpu().numpy(), hpos)
        act, rew, term = gpu.actorStep(None, block=gblk, slot=t)
        hact, hrew, hterm = tw.actor_step(None, block=hblk, block_cap=hcap, slot=t)
        assert np.array_equal(act.cpu().numpy(), hact) and np.array_equal(rew.cpu().numpy(), hrew)
        assert np.array_equal(term.cpu().numpy(), hterm)
    assert np.array_equal(gpu.getStates().cpu().numpy(), tw.states()) and np.array_equal(gpu.getQubits().cpu().numpy(), tw.qubits())
    ep, st = gpu.getCounters()
    hep, hst = tw.counters()
    assert np.array_equal(ep.cpu().numpy().astype(np.uint32), hep) and np.array_equal(st.cpu().numpy().astype(np.uint32), hst)
    assert int(hep.max()) >= 3
    gpu.check()
    tw.check()
    assert np.array_equal(gblk.buf.cpu().numpy()[:hblk.size], hblk)        # priorities: zeros on both sides
    gpu.close()
    tw.close()


def test_reused_stack_buffer_of_the_policy_glue(T):
    """EnvSet.generatePerspectiveReused: the stack in ONE buffer the EnvSet keeps, chosen by a bounded probe at first
    use (pickStackBuffer, 4 candidates), sized from the observed perspective count with headroom, returned as views --
    same contents as generatePerspective at every step, same storage from step to step, re-allocated larger when a
    step outgrows it and when the dtype changes."""
    d, n = 7, 1500
    gpu, ora = make_pair(T, d, n, seed=8, numpy_io=False)
    gpu.resetAll()
    ptrs = set()
    for t in range(5):
        per, pos, cnt = gpu.generatePerspective()
        rper, rpos, rcnt = gpu.generatePerspectiveReused()
        assert torch.equal(per, rper) and torch.equal(pos, rpos) and torch.equal(cnt, rcnt)
        ptrs.add(rper.data_ptr())
        act, qv = gpu.selectAction(None, np.ones(n))               # the positions the glue remembers are the views
        gpu.actorStep(act)
    assert len(ptrs) == 1
    cache = gpu._stack_cache
    rep = cache["probe"]
    assert rep["candidates"] == 4 and rep["kinds"][0] == "torch.empty" and rep["writes_per_candidate"] >= 6
    assert per.shape[0] <= cache["capacity"] < n * 2 * d * d       # observed count x headroom, not the worst case
    assert gpu.reusedStackBacking().shape[0] == cache["capacity"]
    # a step that outgrows the buffer: dense syndromes (every qubit a hit)
    q = np.zeros((n, 2, d, d), np.uint8)
    q[:, 0, ::2, ::2] = 1
    q[:, 1, 1::2, 1::2] = 3
    gpu.setQubits(q)
    per, pos, cnt = gpu.generatePerspective()
    assert per.shape[0] > cache["capacity"]
    rper, rpos, rcnt = gpu.generatePerspectiveReused()
    assert torch.equal(per, rper) and torch.equal(pos, rpos) and gpu._stack_cache["capacity"] >= per.shape[0]
    bp, bpos, _, _ = O.generate_perspective_batch(O.syndrome(q))
    assert np.array_equal(rper.cpu().numpy(), bp.astype(np.float32)) and np.array_equal(rpos.cpu().numpy(), bpos)
    h8, _, _ = gpu.generatePerspectiveReused(dtype=torch.uint8)    # one cache entry: the f32 buffer is gone
    assert torch.equal(h8.float(), per) and gpu._stack_cache["dtype"] == torch.uint8
    big, _, _ = gpu.generatePerspectiveReused(dtype=torch.uint8, capacity=n * 2 * d * d)
    assert gpu._stack_cache["capacity"] == n * 2 * d * d and torch.equal(big.float(), per)
    gpu.check()
    gpu.close()


def test_chunked_buffers_never_alias(T):
    """tq_stack_alloc buffers of changing sizes, allocated, written with patterns of their own and freed in turn: no
    buffer ever shows another buffer's bytes, and position-dependent patterns come back intact.  (The HIP virtual
    memory API leaves stale address translations behind when an address is used again for another physical chunk -- after
    hipMemAddressFree + a later reservation of the same range, or after hipMemUnmap + hipMemMap: a later buffer read and
    wrote the previous tenant's pages.  The library never re-uses an address and checks every buffer before handing it
    out, csrc/toricenv.hip.)"""
    dev = torch.device("cuda:0")
    sizes = [(600 * 18 * 72, 600 * 18 * 12), (600 * 50 * 200, 600 * 50 * 12), (600 * 98 * 392, 600 * 98 * 12),
             (600 * 162 * 648, 600 * 162 * 12), (600 * 50 * 200, 600 * 50 * 12), (600 * 98 * 392, 600 * 98 * 12)]
    for rep in range(2):
        for na, nb in sizes:
            a = T.alloc_chunked((na,), torch.uint8, dev)
            b = T.alloc_chunked((nb,), torch.uint8, dev)
            a.fill_(0x3F)
            torch.cuda.synchronize()
            assert not bool(b.any()), (rep, na, nb)
            b.fill_(0x11)
            torch.cuda.synchronize()
            assert bool((a == 0x3F).all()), (rep, na, nb)
            del a, b
    for mib in (130, 200, 333, 2500):
        n = (mib << 20) // 4
        w = T.alloc_chunked((n,), torch.int32, dev)
        pat = torch.arange(n, dtype=torch.int32, device=dev)
        w.copy_(pat)                                               # straight after the allocation, no other call in between
        torch.cuda.synchronize()
        assert torch.equal(w, pat), mib
        del w, pat
    torch.cuda.empty_cache()


# ------------------------------------------------------------------ the path that prevents a hang
@pytest.mark.parametrize("d,dtype", [(7, torch.float32), (5, torch.uint8), (9, torch.bfloat16)])
def test_offsets_that_do_not_belong_to_the_lattices_are_refused(T, d, dtype):
    """tq_persp_write with an offsets array that is not the scan of the lattices' counts (shifted by one lattice, all
    zeros, decreasing, a gap in the middle, stale after a step): the call returns, the grid drains (no wave waits
    for ever on a commit turn that never comes), nothing is stored outside [0, offsets[N]) perspectives, tq_check
    reports TQ_E_INVALID, and the next correct write on the same handle is bit-exact (stream_write.hpp: range check,
    per-lattice count check, bounded waits)."""
    import time
    n = 5000
    gpu, ora = make_pair(T, d, n, numpy_io=False)
    gpu.resetAll()
    ora.resetAll()
    bp, bpos, bcnt, boff = O.generate_perspective_batch(ora.states)
    _, off = gpu.perspectiveCounts()
    good = off.clone()
    assert np.array_equal(good.cpu().numpy(), boff)
    P, tail = int(boff[-1]), 64
    nq = 2 * d * d
    sent = 0x5A if dtype == torch.uint8 else 0x5A5A
    stack = torch.empty((P + tail, 2, d, d), dtype=dtype, device=gpu.device)
    pos = torch.empty((P + tail, 3), dtype=torch.int32, device=gpu.device)
    raw = stack.view(torch.uint8 if dtype == torch.uint8 else torch.int16 if dtype != torch.float32 else torch.int32)
    half = torch.zeros_like(good)
    half[n // 2:] = 5
    bad_tables = {"shifted by one lattice": torch.cat([good[1:], good[-1:]]),
                  "all zeros": torch.zeros_like(good),
                  "decreasing": good.flip(0).contiguous(),
                  "a gap of five perspectives in the middle": good + half}
    for name, bad in bad_tables.items():
        raw.fill_(sent)
        pos.fill_(-7)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gpu.writePerspectives(stack[:P + 8], pos, bad)      # capacity P + 8 perspectives
        torch.cuda.synchronize()
        assert time.perf_counter() - t0 < 5.0, name          # returned: every wait in the kernel is bounded
        with pytest.raises(ValueError, match="offsets"):
            gpu.check()
        assert bool((raw[P + 8:] == sent).all()) and bool((pos[P + 8:] == -7).all()), name   # nothing behind the capacity
        gpu.check()                                          # the latch is cleared by the read
    # stale offsets: the lattices moved on (a step) and the old scan is handed in again
    gpu.actorStep(None, want_actions=False)
    gpu.writePerspectives(stack[:P], pos, good)
    with pytest.raises(ValueError, match="offsets"):
        gpu.check()
    # the handle is as good as new: a correct write is bit-exact
    oact, _, _ = O.select_action_batch(np.zeros((P, 3), np.float32), boff, bpos, 1.0, ora.seed, ora.env_ids, ora.episodes, ora.steps)
    _, _, oterm, _ = ora.step(oact)
    idx = np.nonzero(oterm | (ora.steps > 75))[0]
    if idx.size:
        ora.resetTerminalEnvs(idx)
    per, ppos, cnt = gpu.generatePerspective(dtype=dtype)
    bp2, bpos2, bcnt2, _ = O.generate_perspective_batch(ora.states)
    assert np.array_equal(per.float().cpu().numpy(), bp2.astype(np.float32)) and np.array_equal(ppos.cpu().numpy(), bpos2)
    gpu.check()
    # an honest empty stack (no lattice has a defect) is not an error
    gpu.setQubits(np.zeros((n, 2, d, d), np.uint8))
    _, off0 = gpu.perspectiveCounts()
    assert int(off0[-1]) == 0
    gpu.writePerspectives(stack[:P], pos, off0)
    gpu.check()
    gpu.close()
    # the stateless entry points refuse a foreign table as well
    st = torch.as_tensor(ora.states.astype(np.uint8), device="cuda")
    per, ppos, cnt, soff = T.generatePerspectiveBatch(d // 2, d, st, dtype=dtype, return_offsets=True)
    import ctypes as C
    L = T.load()
    wrong = torch.cat([soff[1:], soff[-1:]]).contiguous()
    out = torch.empty((int(soff[-1]) + 8, 2, d, d), dtype=dtype, device="cuda")
    from toric_rl_decoder_amd.envset import _DTYPES, _ptr, _stream
    assert L.tq_states_persp_write(d, n, _ptr(st), _ptr(wrong), _ptr(out), C.c_void_p(0), int(soff[-1]), _DTYPES[dtype], _stream()) == 0
    assert L.tq_states_check(_stream()) == -1 and b"offsets" in L.tq_last_error()
    assert L.tq_states_check(_stream()) == 0


def test_scan_after_a_large_stack_leaves_no_stale_cut_points(T):
    """One handle: a large stack, then a state with fewer perspectives than the stack write has workgroups (P < 256).
    Every entry of the cut-point table is rewritten by every scan (k_scan_final), so the small stack is exact."""
    d, n = 5, 6000
    gpu, ora = make_pair(T, d, n, numpy_io=False)
    gpu.resetAll()
    ora.resetAll()
    per, pos, cnt = gpu.generatePerspective()
    bp, bpos, _, _ = O.generate_perspective_batch(ora.states)
    assert np.array_equal(per.cpu().numpy(), bp.astype(np.float32))
    rng = np.random.default_rng(5)
    for hot in (1, 3, 9):                                    # lattices that keep one error: 4-8 perspectives each
        q = np.zeros((n, 2, d, d), np.uint8)
        where = rng.choice(n, hot, replace=False)
        q[where, rng.integers(0, 2, hot), rng.integers(0, d, hot), rng.integers(0, d, hot)] = rng.integers(1, 4, hot)
        gpu.setQubits(q)
        per, pos, cnt = gpu.generatePerspective()
        bp, bpos, bcnt, _ = O.generate_perspective_batch(O.syndrome(q))
        assert 0 < bp.shape[0] < 256
        assert np.array_equal(per.cpu().numpy(), bp.astype(np.float32)) and np.array_equal(pos.cpu().numpy(), bpos)
        assert np.array_equal(cnt.cpu().numpy(), bcnt)
        reused, rpos, _ = gpu.generatePerspectiveReused()
        assert np.array_equal(reused.cpu().numpy(), bp.astype(np.float32)) and np.array_equal(rpos.cpu().numpy(), bpos)
    gpu.check()
    gpu.close()


@pytest.mark.parametrize("d,n,dtype", [(7, 24576, torch.float32), (9, 8192, torch.bfloat16), (11, 8192, torch.float16), (7, 32768, torch.uint8)])
def test_unequal_workgroup_shares_write_the_same_stack(T, d, n, dtype):
    """tq_set_xcd_bias: the workgroups of even XCDs take more of the stack than those of odd XCDs (d >= 7, stacks of
    64 MB and more).  Whatever the shares -- equal, the default, the extreme 48 : 16 -- the stack, the positions and
    the latch are the same, through the handle's table and through a lattice range (cut points found in the kernel);
    the equal-share stack is held against the C oracle."""
    from oracle.c_oracle import CEnvBatch
    L = T._lib.load()
    default = L.tq_get_xcd_bias()
    gpu, _ = make_pair(T, d, n, p=0.12, seed=77, numpy_io=False)
    try:
        gpu.resetAll()
        for _ in range(2):
            gpu.actorStep(None)
        st_np = gpu.getStates().cpu().numpy()
        counts, offsets = gpu.perspectiveCounts()
        P = int(offsets[-1].item())
        esize = torch.empty((), dtype=dtype).element_size()
        assert P * 2 * d * d * esize >= 64 << 20             # large enough for the shares to apply
        out = {}
        for bias in (0, default, 16, 3):
            assert L.tq_set_xcd_bias(bias) == 0 and L.tq_get_xcd_bias() == bias
            stack = torch.full((P + 8, 2, d, d), 7, dtype=dtype, device=gpu.device)
            pos = torch.full((P + 8, 3), -1, dtype=torch.int32, device=gpu.device)
            gpu.writePerspectives(stack, pos, offsets)
            gpu.check()
            half = torch.full_like(stack, 7)
            hpos = torch.full_like(pos, -1)
            gpu.writePerspectives(half, hpos, offsets, first=0, count=n // 2 + 3)    # a lattice range: no table
            gpu.check()
            ph = int(offsets[n // 2 + 3].item())
            assert torch.equal(half[:ph], stack[:ph]) and torch.equal(hpos[:ph], pos[:ph])
            assert bool((half[ph:] == 7).all()) and bool((stack[P:] == 7).all()) and bool((pos[P:] == -1).all())
            out[bias] = (stack, pos)
        for bias in out:
            assert torch.equal(out[bias][0], out[0][0]) and torch.equal(out[bias][1], out[0][1]), f"bias {bias} writes another stack"
        ce = CEnvBatch(d, n, 0.12, seed=77)
        cper, cpos, ccnt, coff = ce.perspectives(states=st_np, dtype=np.uint8)
        assert cper.shape[0] == P and np.array_equal(cpos, out[0][1][:P].cpu().numpy())
        step = 1 << 20
        for i in range(0, P, step):
            want = torch.as_tensor(cper[i:i + step], device=gpu.device)
            assert torch.equal(out[0][0][:P][i:i + step].to(torch.float32), want.to(torch.float32))
        assert L.tq_set_xcd_bias(17) < 0 and L.tq_set_xcd_bias(-1) < 0 and L.tq_get_xcd_bias() == 3
        # one handle's own setting goes before the process-wide one; -1 follows it again
        h = gpu._h
        assert L.tq_env_get_xcd_bias(h) == 3 and L.tq_env_set_xcd_bias(h, 9) == 0 and L.tq_env_get_xcd_bias(h) == 9 and L.tq_get_xcd_bias() == 3
        stack9 = torch.full_like(out[0][0], 7)
        gpu.writePerspectives(stack9, None, offsets)
        gpu.check()
        assert torch.equal(stack9, out[0][0])
        assert L.tq_env_set_xcd_bias(h, 17) < 0 and L.tq_env_set_xcd_bias(h, -2) < 0 and L.tq_env_set_xcd_bias(h, -1) == 0 and L.tq_env_get_xcd_bias(h) == 3
        if d == 7:
            # the probe: candidates of ONE kind write at one rate -> the search is extended once, the first ones still allocated;
            # the share setting is checked on the kept buffer (and whatever it decides, the stack is the same)
            L.tq_set_xcd_bias(default)
            best, rep = gpu.pickStackBuffer(3, capacity=P + 8, kinds=("torch",), launches=4)
            added = rep["candidates_added_because_uniform"]
            assert added in (0, 3) and rep["candidates"] == 3 + added == len(rep["write_ms"]) == len(rep["kinds"])
            assert (added == 3) == (min(rep["write_ms"][:3]) > 0.93 * rep["write_ms"][0] and rep["write_ms"][0] >= 0.1)
            assert "xcd_bias" in rep and rep["xcd_bias"]["bias"] in (0, default) and L.tq_get_xcd_bias() == default   # the process-wide setting is left alone
            assert L.tq_env_get_xcd_bias(h) == rep["xcd_bias"]["bias"]
            gpu.writePerspectives(best, None, offsets)
            gpu.check()
            assert torch.equal(best[:P], out[0][0][:P])
    finally:
        L.tq_set_xcd_bias(default)
        gpu.close()


def test_stack_write_in_a_replayed_graph_takes_its_shares_again(T):
    """A stack write large enough for the unequal shares, captured into a HIP graph (three writes: not the number of
    counter sets the handle goes round) and replayed: every replay finds the slot counters of the captured launches as the
    launch before left them -- zero -- and writes the whole stack again."""
    d, n = 7, 24576
    gpu, _ = make_pair(T, d, n, p=0.12, seed=31, numpy_io=False)
    try:
        gpu.resetAll()
        gpu.actorStep(None)
        counts, offsets = gpu.perspectiveCounts()
        P = int(offsets[-1].item())
        assert P * 2 * d * d * 4 >= 64 << 20 and T._lib.load().tq_get_xcd_bias() > 0
        ref = torch.empty((P, 2, d, d), dtype=torch.float32, device=gpu.device)
        rpos = torch.empty((P, 3), dtype=torch.int32, device=gpu.device)
        gpu.writePerspectives(ref, rpos, offsets)
        gpu.check()
        bufs = [torch.zeros_like(ref) for _ in range(3)]
        poss = [torch.zeros_like(rpos) for _ in range(3)]
        torch.cuda.synchronize(gpu.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for b, q in zip(bufs, poss):
                gpu.writePerspectives(b, q, offsets)
        for replay in range(4):
            for b, q in zip(bufs, poss):
                b.zero_()
                q.zero_()
            g.replay()
            torch.cuda.synchronize(gpu.device)
            gpu.check()
            for b, q in zip(bufs, poss):
                assert torch.equal(b, ref) and torch.equal(q, rpos), f"replay {replay} wrote another stack"
        gpu.writePerspectives(bufs[0], poss[0], offsets)         # and eagerly afterwards
        gpu.check()
        assert torch.equal(bufs[0], ref)
    finally:
        gpu.close()


# ------------------------------------------------------------------ the two-stream loop
@pytest.mark.parametrize("d,n,chunks", [(7, 16384, 1), (5, 8192, 1), (9, 4096, 4)])
def test_explore_loop_on_two_streams_equals_the_serial_loop_and_the_oracle(T, d, n, chunks):
    """T.ExploreLoop(overlap=True): the fused step and the next scan run on a second stream BESIDE the stack write (two
    plane buffers and two cut-point tables in the handle, two events per step).  Free-running for 160 steps -- no host
    synchronisation inside -- it must leave, step for step, the same stack and positions (integer checksums taken on
    the write's stream), the same packed transition blocks, and at the end the same lattices and counters as the
    same loop on ONE stream and as the C oracle's actor loop."""
    from oracle.c_oracle import CEnvBatch
    steps, flush, p = 160, 8, P_OF[d]
    env = T.make("toric-code-v0", {"size": d, "p_error": p})
    nq = 2 * d * d
    runs = []
    for overlap in (True, False):
        gpu = T.EnvSet(env, n, seed=4321, first_env_id=5, numpy_io=False)
        gpu.resetAll()
        cap = (n // chunks) * nq
        stack = torch.zeros((cap, 2, d, d), dtype=torch.float32, device=gpu.device)
        pos = torch.zeros((cap, 3), dtype=torch.int32, device=gpu.device)
        offs = torch.zeros((steps + 2, (n + 2) & ~1), dtype=torch.int64, device=gpu.device)
        blocks = [gpu.newTransitionBlock(steps=flush) for _ in range(2)]
        flushed = []
        loop = T.ExploreLoop(gpu, stack, pos, offs, blocks=blocks, flush=flush, chunks=chunks, overlap=overlap,
                             on_flush=lambda b: flushed.append(b.buf.clone()))
        sums = torch.zeros((steps, 2), dtype=torch.int64, device=gpu.device)
        for t in range(steps):
            loop.step()
            sums[t, 0] = stack.view(torch.int32).sum(dtype=torch.int64)      # on the write's stream, behind write(t)
            sums[t, 1] = pos.sum(dtype=torch.int64)
        loop.drain()
        torch.cuda.synchronize()
        gpu.check()
        ep, st = gpu.getCounters()
        runs.append(dict(sums=sums.cpu(), P=offs[:steps, n].cpu(), flushed=[f.cpu() for f in flushed], qubits=gpu.getQubits().cpu().numpy(),
                         states=gpu.getStates().cpu().numpy(), ep=ep.cpu().numpy(), st=st.cpu().numpy(), stack=stack.cpu(), pos=pos.cpu()))
        assert loop.overlap == overlap
        gpu.close()
    a, b = runs
    assert torch.equal(a["P"], b["P"]) and torch.equal(a["sums"], b["sums"])
    assert len(a["flushed"]) == len(b["flushed"]) == steps // flush and all(torch.equal(x, y) for x, y in zip(a["flushed"], b["flushed"]))
    assert torch.equal(a["stack"], b["stack"]) and torch.equal(a["pos"], b["pos"])
    for k in ("qubits", "states", "ep", "st"):
        assert np.array_equal(a[k], b[k]), k
    ce = CEnvBatch(d, n, p, seed=4321, first_env_id=5)
    ce.reset()
    P, _ = ce.actor_steps(steps)
    assert int(a["P"].sum()) == P
    assert np.array_equal(a["qubits"], ce.qubits) and np.array_equal(a["states"], ce.states)
    assert np.array_equal(a["ep"].astype(np.uint32), ce.episodes) and np.array_equal(a["st"].astype(np.uint32), ce.steps)
