"""Policy-side callers of the hot path on the GPU against oracle loops.
The Q-network is replaced by an integer-weight linear map: exact in fp32 on CPU and GPU, with many
ties, so the first-maximum rule is exercised and results are bit-identical."""
import numpy as np
import pytest
import torch

from oracle import toric_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import toric_rl_decoder_amd as T
    assert torch.cuda.is_available()
    T.load()
    return T


class IntQ(torch.nn.Module):
    def __init__(self, d, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.w = torch.nn.Parameter(torch.randint(-2, 3, (2 * d * d, 3), generator=g).float(), requires_grad=False)

    def forward(self, x):
        return x.flatten(1).float() @ self.w


def oracle_q(model, persp):
    return (persp.reshape(persp.shape[0], -1).astype(np.float32) @ model.w.cpu().numpy()).astype(np.float32)


@pytest.mark.parametrize("d", (3, 5, 7))
def test_select_action_batch_with_model(T, d):
    n = 600
    env = T.make("toric-code-v0", {"size": d, "p_error": 0.1})
    gpu = T.EnvSet(env, n, seed=8, numpy_io=True)
    ora = O.OracleEnvSet(d, n, 0.1, seed=8)
    model = IntQ(d).to(gpu.device)
    assert np.array_equal(gpu.resetAll(), ora.resetAll())
    rng = np.random.default_rng(d)
    for t in range(10):
        eps = rng.random(n) * 0.6
        act, qv = T.selectActionEnvSet(gpu, model, eps)
        bp, bpos, bcnt, boff = O.generate_perspective_batch(ora.states)
        oact, oqv, _ = O.select_action_batch(oracle_q(model, bp), boff, bpos, eps, ora.seed, ora.env_ids,
                                             ora.episodes, ora.steps)
        assert np.array_equal(act, oact) and np.array_equal(qv, oqv)
        ns, r, term, _ = gpu.step(act)                      # solved lattices carry op 0: no-op, no error
        ons, orr, oterm, _ = ora.step(oact)
        assert np.array_equal(ns, ons) and np.array_equal(r, orr) and np.array_equal(term, oterm)
    assert (act[term][:, 3] == 0).sum() >= 0
    gpu.close()


@pytest.mark.parametrize("d", (5, 7, 9))
def test_predict_max_optimized(T, d):
    """util_learner.py:48-111 incl. the zero-padding quirk and terminal states."""
    rng = np.random.default_rng(50 + d)
    n = 300
    _, st = O.reset_lattices(5, np.arange(n), 0, 0.08, d)
    st[::7] = 0                                              # terminal states in the batch
    model = IntQ(d, seed=3).cuda()
    got = T.predictMaxOptimized(model, st, d // 2, d, "cuda").cpu().numpy()
    bp, _, cnt, off = O.generate_perspective_batch(st)
    q = oracle_q(model, bp)
    largest = max(1, int(cnt.max()))
    want = np.zeros(n, np.float32)
    for i in range(n):
        if cnt[i] == 0:
            continue
        m = q[off[i]:off[i + 1]].max()
        want[i] = max(m, 0.0) if cnt[i] < largest else m     # reference pads shorter slices with zero rows
    assert np.array_equal(got, want)
    # plain segmented maximum (no padding)
    qd = torch.as_tensor(q, device="cuda")
    plain = T.segment_max(qd, torch.as_tensor(off, device="cuda")).cpu().numpy()
    ref = np.array([q[off[i]:off[i + 1]].max() if cnt[i] else 0.0 for i in range(n)], np.float32)
    assert np.array_equal(plain, ref)


@pytest.mark.parametrize("d", (3, 5, 7, 9, 11, 13, 15, 17, 19, 21))
def test_learner_goldens_from_the_reference(T, golden_dir, d):
    """predictMaxOptimized and computePrioritiesParallel against vectors produced by running the
    reference's own functions (tests/golden/make_golden.py): the device path, the numpy drop-in and
    the packed-block kernel."""
    import os
    from toric_rl_decoder_amd import wire
    g = np.load(os.path.join(golden_dir, f"learner_d{d}.npz"), allow_pickle=False)

    class Lin(torch.nn.Module):
        def __init__(self, w):
            super().__init__()
            self.w = torch.nn.Parameter(torch.as_tensor(w), requires_grad=False)

        def forward(self, x):
            return x.flatten(1).float() @ self.w

    got = T.predictMaxOptimized(Lin(g["pm_w"]).cuda(), g["pm_states"], d // 2, d, "cuda").cpu().numpy()
    assert np.array_equal(got, g["pm_out"])
    A, R, Q, disc = g["pr_A"].astype(np.int64), g["pr_R"], g["pr_Q"].astype(np.float64), float(g["pr_discount"])
    pr = T.computePrioritiesParallel(A, R, Q[:, :-1], np.roll(Q, -1, axis=1)[:, :-1], disc)
    assert pr.dtype == np.float64 and np.array_equal(pr, g["pr_out"])
    n, steps = A.shape[0], A.shape[1]
    zeros = np.zeros((n * steps, 2, d, d), np.uint8)
    buf = wire.encode(d, zeros, zeros, A.transpose(1, 0, 2).reshape(-1, 4), R.T.reshape(-1), np.zeros(n * steps, bool))
    blk = T.TransitionBlock(d, n * steps, torch.device("cuda"))
    blk.buf.copy_(torch.as_tensor(buf, device="cuda"))
    blk.computePriorities(n, steps, torch.as_tensor(g["pr_Q"].transpose(1, 0, 2).copy(), device="cuda"), disc)
    assert np.array_equal(blk.unpack()["priority"].cpu().numpy(), g["pr_out"].T.reshape(-1).astype(np.float32))


def test_evaluate_matches_oracle_loop(T):
    """evaluation.py:10-124, episodes batched; compared with the same loop on the oracle."""
    d, episodes, max_steps = 5, 400, 12
    model = IntQ(d, seed=1)
    ps = [0.05, 0.12]
    corrected, ground, steps_avg, mean_q, failed = T.evaluate(model, "toric-code-v0", {"size": d, "min_qubit_errors": 0},
                                                             d // 2, "cuda", ps, num_of_episodes=episodes,
                                                             num_of_steps=max_steps, seed=21)
    for i, p in enumerate(ps):
        env = O.OracleEnvSet(d, episodes, p, seed=21 + i)
        env.resetAll()
        done = np.zeros(episodes, bool)
        steps = np.zeros(episodes, np.int64)
        qs, qn = 0.0, 0
        for _ in range(max_steps):
            bp, bpos, bcnt, boff = O.generate_perspective_batch(env.states)
            act, qv, _ = O.select_action_batch(oracle_q(model, bp), boff, bpos, 0.0, env.seed, env.env_ids,
                                               env.episodes, env.steps)
            live = ~done
            chosen = qv[np.arange(episodes), np.clip(act[:, 3] - 1, 0, 2)]
            qs += float((chosen.astype(np.float64) * live).sum())
            qn += int(live.sum())
            steps += live
            _, _, term, _ = env.step(act)
            done |= term
            if done.all():
                break
        gs = O.eval_ground_state(env.qubits)
        assert np.isclose(corrected[i], done.mean(), atol=1e-12) and np.isclose(ground[i], gs.mean(), atol=1e-12)
        assert steps_avg[i] == np.round(steps.mean(), 1) and mean_q[i] == np.round(qs / max(qn, 1), 3)
    assert len(failed) % 2 == 0


def test_actor_loop_matches_oracle_loop(T):
    """run_actor (Actor_mp.py:104-185 on the device) against the same loop spelled with oracle
    calls: the transitions of the buffer columns that are sent and, in the same packed block, the
    priorities of computePrioritiesParallel (f64 arithmetic as upstream, stored as f32: bit-equal)."""
    from toric_rl_decoder_amd import wire
    d, n, buf, flushes, max_steps = 5, 300, 4, 3, 6
    env = T.make("toric-code-v0", {"size": d, "p_error": 0.1})
    gpu = T.EnvSet(env, n, seed=31, numpy_io=False, max_steps_per_episode=max_steps)
    ora = O.OracleEnvSet(d, n, 0.1, seed=31)
    model = IntQ(d, seed=2).to(gpu.device)
    gpu.resetAll()
    ora.resetAll()
    eps = 0.3
    gen = T.run_actor(gpu, model, flushes, buf, eps, discount_factor=0.95)
    for f in range(flushes):
        blk = next(gen)
        A = np.zeros((n, buf + 1, 4), np.int64)
        Q = np.zeros((n, buf + 1), dtype=(np.float64, 3))       # local_buffer_Q is an f64 array (Actor_mp.py:69)
        R = np.zeros((n, buf + 1))
        trans = []
        for t in range(buf + 1):
            bp, bpos, bcnt, boff = O.generate_perspective_batch(ora.states)
            act, qv, _ = O.select_action_batch(oracle_q(model, bp), boff, bpos, eps, ora.seed, ora.env_ids,
                                               ora.episodes, ora.steps)
            prev = ora.states.copy()
            _, rew, term, _ = ora.step(act)
            trans.append(O.generate_transition_batch(act, prev, ora.states) + (rew.astype(np.float32), term))
            A[:, t], Q[:, t], R[:, t] = act, qv, rew
            idx = np.nonzero(term | (ora.steps > max_steps))[0]
            if idx.size:
                ora.resetTerminalEnvs(idx)
        want = O.compute_priorities(A[:, :-1], R[:, :-1], Q[:, :-1], np.roll(Q, -1, axis=1)[:, :-1], 0.95)
        for t in range(buf):
            u = blk.unpack(first=t * n, count=n)
            per, act_c, nper, rew, term = trans[t]
            assert np.array_equal(u["perspective"].cpu().numpy(), per)
            assert np.array_equal(u["next_perspective"].cpu().numpy(), nper)
            assert np.array_equal(u["action"].cpu().numpy(), act_c)
            assert np.array_equal(u["reward"].cpu().numpy(), rew)
            assert np.array_equal(u["terminal"].cpu().numpy().astype(bool), term)
            assert np.array_equal(u["priority"].cpu().numpy(), want[:, t].astype(np.float32))
        # host ingest of the same block: the (transition, priority) list of Actor_mp.py:152
        rec, pr = wire.to_records(wire.decode(blk.buf.cpu().numpy(), d, blk.capacity), d)
        assert rec.shape[0] == n * buf and np.array_equal(pr.reshape(buf, n).T, want.astype(np.float32))
        assert np.array_equal(gpu.getStates().cpu().numpy(), ora.states)
    gpu.close()


def test_actor_mp_call_sequence_with_reference_signatures(T):
    """The loop body of src/Actor_mp.py:104-183 typed out with the reference's own names, keyword
    arguments and numpy buffers -- only the imports differ -- against the same loop on the oracle."""
    from toric_rl_decoder_amd import (EnvSet, computePrioritiesParallel, generateTransitionParallel, make,
                                      seed_select, selectActionBatch)
    args = {"device": "cuda", "discount_factor": 0.95, "no_envs": 150, "epsilon_final": [0.2],
            "env_p_error_start": 0.1, "env_p_error_final": 0.2, "env_p_error_delta": 0.02,
            "env_p_error_strategy": "linear", "env": "toric-code-v0",
            "env_config": {"size": 5, "min_qubit_errors": 0, "p_error": 0.1},
            "size_local_memory_buffer": 3, "max_actions_per_episode": 5}
    device, discount_factor, no_envs = args["device"], args["discount_factor"], args["no_envs"]
    epsilon = np.ones(no_envs) * 0.35
    env_p_error_start, env_p_error_final = args["env_p_error_start"], args["env_p_error_final"]
    env_p_error_delta, env_p_error_strategy = args["env_p_error_delta"], args["env_p_error_strategy"]
    env_p_errors = np.ones(no_envs) * env_p_error_start
    env = make(args["env"], config=args["env_config"], seed=404)
    envs = EnvSet(env, no_envs)
    size = env.system_size
    action_type = np.dtype([('position', (np.int64, 3)), ('op', np.int64)])          # src/util.py:10
    transition_type = np.dtype([('perspective', (np.int64, (2, size, size))), ('action', action_type),
                                ('reward', np.float64), ('next_perspective', (np.int64, (2, size, size))),
                                ('terminal', np.bool_)])
    no_actions = int(env.action_space.high[-1])
    grid_shift = int(size / 2)
    model = IntQ(size, seed=6).to(device)
    state = envs.resetAll(p_errors=env_p_errors)
    steps_per_episode = np.zeros(no_envs)
    size_local_memory_buffer = args["size_local_memory_buffer"] + 1
    local_buffer_T = np.empty((no_envs, size_local_memory_buffer), dtype=transition_type)
    local_buffer_A = np.empty((no_envs, size_local_memory_buffer, 4), dtype=np.int64)
    local_buffer_Q = np.empty((no_envs, size_local_memory_buffer), dtype=(np.float64, 3))
    local_buffer_R = np.empty((no_envs, size_local_memory_buffer))
    buffer_idx = 0
    sent = []

    # the oracle's copy of the loop state
    ora = O.OracleEnvSet(size, no_envs, 0.1, seed=404)
    o_state = ora.resetAll(p_errors=env_p_errors)
    assert np.array_equal(state, o_state) and state.dtype == np.int64
    o_T = np.empty((no_envs, size_local_memory_buffer), dtype=transition_type)
    o_A, o_Q, o_R = np.empty_like(local_buffer_A), np.empty_like(local_buffer_Q), np.empty_like(local_buffer_R)
    seed_select(2024)
    resets = 0
    for it in range(3 * size_local_memory_buffer):
        steps_per_episode += 1
        action, q_values = selectActionBatch(number_of_actions=no_actions, epsilon=epsilon, grid_shift=grid_shift,
                                             toric_size=size, state=state, model=model, device=device)
        next_state, reward, terminal_state, _ = envs.step(action)
        transition = generateTransitionParallel(action, reward, state, next_state, terminal_state, grid_shift,
                                                transition_type)
        # ---- oracle: the same three calls
        bp, bpos, bcnt, boff = O.generate_perspective_batch(o_state.astype(np.uint8))
        o_action, o_q, _ = O.select_action_batch(oracle_q(model, bp), boff, bpos, epsilon, 2024, np.arange(no_envs),
                                                 it & 0xFFFFFFFF, it >> 32, domain=O.DOMAIN_SEL_CALL)
        o_next, o_reward, o_term, _ = ora.step(o_action)
        o_tr = O.generate_transition_ref(o_action, o_reward, o_state, o_next, o_term, grid_shift)
        assert action.dtype == np.int64 and q_values.dtype == np.float64 and q_values.shape == (no_envs, 3)
        assert np.array_equal(action, o_action) and np.array_equal(q_values, o_q.astype(np.float64))
        assert np.array_equal(next_state, o_next) and np.array_equal(reward, o_reward)
        assert np.array_equal(terminal_state, o_term) and reward.dtype == np.float64
        assert transition.dtype == transition_type
        for k in ("perspective", "next_perspective"):
            assert np.array_equal(transition[k], o_tr[k])
        assert np.array_equal(transition["action"]["position"], o_tr["position"])
        assert np.array_equal(transition["action"]["op"], o_tr["op"])
        assert np.array_equal(transition["reward"], o_reward) and np.array_equal(transition["terminal"], o_term)

        local_buffer_T[:, buffer_idx] = transition
        local_buffer_A[:, buffer_idx] = action
        local_buffer_Q[:, buffer_idx] = q_values
        local_buffer_R[:, buffer_idx] = reward
        o_T[:, buffer_idx] = transition
        o_A[:, buffer_idx], o_Q[:, buffer_idx], o_R[:, buffer_idx] = o_action, o_q, o_reward
        buffer_idx += 1
        if buffer_idx >= size_local_memory_buffer:
            priorities = computePrioritiesParallel(local_buffer_A[:, :-1], local_buffer_R[:, :-1], local_buffer_Q[:, :-1],
                                                   np.roll(local_buffer_Q, -1, axis=1)[:, :-1], discount_factor)
            to_send = [*zip(local_buffer_T[:, :-1].flatten(), priorities.flatten())]
            want = O.compute_priorities(o_A[:, :-1], o_R[:, :-1], o_Q[:, :-1], np.roll(o_Q, -1, axis=1)[:, :-1],
                                        discount_factor)
            assert priorities.dtype == np.float64 and np.array_equal(priorities, want)
            assert len(to_send) == no_envs * (size_local_memory_buffer - 1)
            sent.append(to_send)
            buffer_idx = 0
        too_many_steps = steps_per_episode > args["max_actions_per_episode"]
        if np.any(terminal_state) or np.any(too_many_steps):
            idx = np.argwhere(np.logical_or(terminal_state, too_many_steps)).flatten()
            env_p_errors[idx] = np.minimum(env_p_error_final, env_p_errors[idx] + env_p_error_delta)
            if env_p_error_strategy == 'random':
                p_errors = np.random.uniform(env_p_error_start, env_p_errors[idx])
            else:
                p_errors = env_p_errors[idx]
            reset_states = envs.resetTerminalEnvs(idx, p_errors=p_errors)
            o_reset = ora.resetTerminalEnvs(idx, p_errors)
            assert reset_states.dtype == np.float64 and np.array_equal(reset_states, o_reset)   # EnvSet.py:20 quirk
            next_state[idx] = reset_states
            o_next[idx] = o_reset
            steps_per_episode[idx] = 0
            resets += idx.size
        state = next_state
        o_state = o_next
    assert len(sent) == 3 and resets > no_envs
    envs.close()


def test_stateless_select_action_edge_cases(T):
    """selectActionBatch over explicit states: scalar epsilon, greedy ties (first maximum), a state
    without defects (the reference would raise, numba/util_actor.py:93: here op 0), argument checks."""
    d, n = 7, 64
    _, st = O.reset_lattices(9, np.arange(n), 0, 0.1, d)
    st[5] = 0
    model = IntQ(d, seed=4).cuda()
    T.seed_select(7, calls=(1 << 32) + 5)                      # a call counter beyond 32 bits
    act, qv = T.selectActionBatch(3, 0.0, d // 2, d, st.astype(np.int64), model, "cuda")
    bp, bpos, bcnt, boff = O.generate_perspective_batch(st)
    oact, oqv, _ = O.select_action_batch(oracle_q(model, bp), boff, bpos, 0.0, 7, np.arange(n), 5, 1,
                                         domain=O.DOMAIN_SEL_CALL)
    assert np.array_equal(act, oact) and np.array_equal(qv, oqv.astype(np.float64))
    assert tuple(act[5]) == (0, 0, 0, 0) and not qv[5].any()
    act2, _ = T.selectActionBatch(3, 1.0, d // 2, d, st, model, "cuda")         # next call: new draws
    oact2, _, _ = O.select_action_batch(oracle_q(model, bp), boff, bpos, 1.0, 7, np.arange(n), 6, 1,
                                        domain=O.DOMAIN_SEL_CALL)
    assert np.array_equal(act2, oact2) and not np.array_equal(act2, act)
    with pytest.raises(ValueError):
        T.selectActionBatch(4, 0.0, d // 2, d, st, model, "cuda")
    with pytest.raises(ValueError):
        T.selectActionBatch(3, 0.0, 1, d, st, model, "cuda")
    with pytest.raises(ValueError):
        T.selectActionBatch(3, 0.0, d // 2, d, st, model, "cpu")


def test_compute_priorities_numpy_and_block_kernel(T):
    """computePrioritiesParallel: numpy in/out (reference dtypes) and the device kernel over a packed
    block, both against the oracle restatement; empty slots get priority 0."""
    from toric_rl_decoder_amd import wire
    rng = np.random.default_rng(3)
    d, n, steps = 7, 333, 5
    A = np.stack((rng.integers(0, 2, (n, steps)), rng.integers(0, d, (n, steps)), rng.integers(0, d, (n, steps)),
                  rng.integers(1, 4, (n, steps))), axis=2)
    Qall = (rng.standard_normal((n, steps + 1, 3)) * 50).astype(np.float32).astype(np.float64)   # f32 values, as the NN gives
    R = rng.integers(-4, 5, (n, steps)).astype(np.float64)
    R[rng.random((n, steps)) < 0.1] = 100.0
    want = O.compute_priorities(A, R, Qall[:, :-1], Qall[:, 1:], 0.95)
    got = T.computePrioritiesParallel(A, R, Qall[:, :-1], Qall[:, 1:], 0.95)
    assert got.dtype == np.float64 and np.array_equal(got, want)
    dev = torch.device("cuda")
    got_t = T.computePrioritiesParallel(torch.as_tensor(A, device=dev), torch.as_tensor(R, device=dev),
                                        torch.as_tensor(Qall[:, :-1], device=dev), torch.as_tensor(Qall[:, 1:], device=dev), 0.95)
    assert np.array_equal(got_t.cpu().numpy(), want)
    # block kernel: slots t*n + e
    zeros = np.zeros((n * steps, 2, d, d), np.uint8)
    act_slots = A.transpose(1, 0, 2).reshape(-1, 4).copy()
    hole = rng.random(n * steps) < 0.05
    act_slots[hole] = 0
    buf = wire.encode(d, zeros, zeros, act_slots, R.T.reshape(-1), np.zeros(n * steps, bool))
    blk = T.TransitionBlock(d, n * steps, dev)
    blk.buf.copy_(torch.as_tensor(buf, device=dev))
    q_dev = torch.as_tensor(Qall.transpose(1, 0, 2).astype(np.float32).copy(), device=dev)     # (steps+1, n, 3)
    blk.computePriorities(n, steps, q_dev, 0.95)
    pr = blk.unpack()["priority"].cpu().numpy()
    w32 = want.T.reshape(-1).astype(np.float32)
    w32[hole] = 0
    assert np.array_equal(pr, w32)
    blk.computePriorities(n, steps, None, 0.95)                 # no Q-values: |reward|
    pr0 = blk.unpack()["priority"].cpu().numpy()
    assert np.array_equal(pr0, np.where(hole, 0, np.abs(R.T.reshape(-1))).astype(np.float32))
    with pytest.raises(ValueError):
        blk.computePriorities(n, steps, q_dev[:-1], 0.95)


def test_transition_gather_rccl_single_rank_with_host_drain(T):
    """The RCCL path of TransitionGather (world of one rank on this box) incl. the asynchronous drain
    of every gathered slot to the pinned host ring; bytes must arrive unchanged and decode."""
    import os
    import torch.distributed as dist
    from toric_rl_decoder_amd import gather, wire
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(sock.getsockname()[1])       # a free port, not a pinned one
    sock.close()
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        d, n = 7, 512
        env = T.make("toric-code-v0", {"size": d, "p_error": 0.1})
        gpu = T.EnvSet(env, n, seed=77, numpy_io=False)
        gpu.resetAll()
        blocks = [gpu.newTransitionBlock(steps=2) for _ in range(2)]
        tg = gather.TransitionGather(blocks[0].nbytes, dev, ring_slots=2, host_drain=True)
        sent = []
        for f in range(5):                                    # 5 flushes through a 2-slot ring
            blk = blocks[f & 1]
            for t in range(2):
                gpu.perspectiveCounts()
                gpu.actorStep(None, block=blk, slot=t)
            blk.computePriorities(n, 2, None, 0.95)           # eps = 1: no Q-values, priority = |reward|
            slot = tg.gather(blk.buf)
            torch.cuda.synchronize()
            sent.append((slot, blk.buf.clone()))
        tg.wait()
        for slot, want in sent[-2:]:                          # the two newest flushes are still in the ring
            assert torch.equal(tg.slot_view(slot, 0), want)
            assert torch.equal(tg.slot_view(slot, 0, host=True), want.cpu())
        rec = wire.decode(tg.slot_view(sent[-1][0], 0, host=True).numpy(), d, 2 * n)
        assert rec["perspective"].shape == (2 * n, 2, d, d) and set(np.unique(rec["action"][:, 3])) <= {1, 2, 3}
        # the priority arrives with its transition, bit-equal to what the device computed
        want = blocks[(len(sent) - 1) & 1].unpack()["priority"].cpu().numpy()
        records, prio = wire.to_records(rec, d)
        assert np.array_equal(prio.view(np.uint32), want.view(np.uint32)) and np.array_equal(prio, np.abs(rec["reward"]))
        assert records.shape[0] == 2 * n and prio.max() > 0
        # weights the other way (Learner_mp.py:124-130 -> Actor_mp.py:133-144): one RCCL broadcast of the flat vector
        from torch.nn.utils import parameters_to_vector
        torch.manual_seed(3)
        model = T.NN_11(d, 3).to(dev)
        w0 = parameters_to_vector(model.parameters()).detach().clone()
        buf = gather.broadcast_weights(model, src=0)
        torch.cuda.synchronize()
        assert buf.is_cuda and buf.numel() == w0.numel() > 800000 and torch.equal(buf, w0)
        assert torch.equal(parameters_to_vector(model.parameters()).detach(), w0)
        gpu.close()
    finally:
        dist.destroy_process_group()


def test_nn11_on_the_device_stack(T):
    """BASELINE configs[2] as written, small: generatePerspective feeds NN_11 (the reference's
    architecture, src/nn/torch/NN.py:10-45, random init) on the device; its forward must agree with
    the same module on the CPU in fp32 (1e-4 absolute: Q-values of a random-init net are O(0.1)),
    and the device selection on those Q-values must be the oracle's choice."""
    d, n = 7, 256
    env = T.make("toric-code-v0", {"size": d, "p_error": 0.1})
    gpu = T.EnvSet(env, n, seed=17, numpy_io=False)
    ora = O.OracleEnvSet(d, n, 0.1, seed=17)
    gpu.resetAll()
    ora.resetAll()
    torch.manual_seed(0)
    model = T.NN_11(d, 3).eval()
    model_gpu = T.NN_11(d, 3).eval()
    model_gpu.load_state_dict(model.state_dict())
    model_gpu.to(gpu.device)
    per, pos, cnt = gpu.generatePerspective(dtype=torch.float32)
    bp, bpos, bcnt, boff = O.generate_perspective_batch(ora.states)
    assert np.array_equal(per.cpu().numpy(), bp.astype(np.float32))
    with torch.no_grad():
        q_gpu = model_gpu(per).float()
        q_cpu = model(torch.from_numpy(bp.astype(np.float32)))
    assert q_gpu.shape == (bp.shape[0], 3)
    assert float((q_gpu.cpu() - q_cpu).abs().max()) < 1e-4              # tolerance stated: fp32 conv, different summation order
    eps = np.full(n, 0.25)
    act, qv = gpu.selectAction(q_gpu, eps, positions=pos)
    oact, oqv, _ = O.select_action_batch(q_gpu.cpu().numpy(), boff, bpos, eps, ora.seed, ora.env_ids, ora.episodes, ora.steps)
    assert np.array_equal(act.cpu().numpy(), oact) and np.array_equal(qv.cpu().numpy(), oqv)
    # and through the reference-signature entry point (numpy in / numpy out)
    T.seed_select(5)
    a2, q2 = T.selectActionBatch(3, 0.0, d // 2, d, ora.states.astype(np.int64), model_gpu, "cuda")
    o2, oq2, _ = O.select_action_batch(q_gpu.cpu().numpy(), boff, bpos, 0.0, 5, np.arange(n), 0, 0, domain=O.DOMAIN_SEL_CALL)
    # the entry point runs its own forward pass (fixed-shape chunks, zero-padded): another batch shape, possibly another
    # MIOpen kernel and summation order than q_gpu above -- Q-values to 1e-5 absolute, the greedy choice exact
    assert np.array_equal(a2, o2) and float(np.abs(q2 - oq2.astype(np.float64)).max()) < 1e-5
    gpu.close()


def test_forward_chunked_padding_in_the_buffer_slack(T, golden_dir):
    """policy._forward_chunked: with pad_to=1 it IS the plain forward (bit-identical); with the ragged last chunk run
    at a padded row count -- surplus rows taken from the slack of the re-used stack buffer, or from a zero-filled copy
    -- the Q-values agree to 1e-4 absolute (trained weights: Q ~ 90, i.e. ~1e-6 relative; a different batch shape may
    pick a different convolution kernel) and the greedy choice of every lattice is the same."""
    import os
    from safetensors.torch import load_file
    from toric_rl_decoder_amd.policy import _forward_chunked
    d, n, chunk = 7, 700, 1 << 14
    model = T.NN_11(d, 3)
    model.load_state_dict(load_file(os.path.join(golden_dir, f"nn11_d{d}_converged.safetensors")))
    model = model.cuda().eval()
    env = T.make("toric-code-v0", {"size": d, "p_error": 0.1})
    gpu = T.EnvSet(env, n, seed=23, numpy_io=False)
    gpu.resetAll()
    per, pos, cnt = gpu.generatePerspectiveReused()
    P = per.shape[0]
    assert P > 2 * chunk and P % 1024 and gpu.reusedStackBacking().shape[0] >= (P + 1023) // 1024 * 1024
    with torch.no_grad():
        direct = torch.cat([model(per[i:i + chunk]) for i in range(0, P, chunk)]).float()
    exact = _forward_chunked(model, per, chunk, pad_to=1)
    assert torch.equal(exact, direct)                                   # same batch shapes: bit-identical
    in_slack = _forward_chunked(model, per, chunk, pad_to=1024, backing=gpu.reusedStackBacking())
    zero_pad = _forward_chunked(model, per.clone(), chunk, pad_to=1024)  # not a view of a larger buffer: zero-filled copy
    out = torch.empty((P + 7, 3), dtype=torch.float32, device=per.device)
    into = _forward_chunked(model, per, chunk, pad_to=1024, backing=gpu.reusedStackBacking(), out=out)
    assert into.data_ptr() == out.data_ptr() and torch.equal(into, in_slack)
    for q in (in_slack, zero_pad):
        assert q.shape == (P, 3) and float((q - exact).abs().max()) < 1e-4
        a = gpu.selectAction(q, 0.0, positions=pos)[0].clone()          # (the method returns its own scratch tensor)
        b = gpu.selectAction(exact, 0.0, positions=pos)[0].clone()
        assert torch.equal(a, b) and int((a[:, 3] > 0).sum()) == n      # greedy choice unchanged by the padding
    gpu.close()
