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
        act, qv = T.selectActionBatch(gpu, model, eps)
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
    calls: transitions of every buffer column and the priorities of computePrioritiesParallel."""
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
        blk, prio = next(gen)
        A = np.zeros((n, buf + 1, 4), np.int64)
        Q = np.zeros((n, buf + 1, 3), np.float32)
        R = np.zeros((n, buf + 1), np.float32)
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
        q_taken = np.take_along_axis(Q[:, :-1], (A[:, :-1, 3] - 1)[..., None], axis=2)[..., 0]
        want = np.abs(R[:, :-1] + np.float32(0.95) * np.roll(Q, -1, axis=1)[:, :-1].max(axis=2) - q_taken)
        assert np.allclose(prio.cpu().numpy(), want, rtol=0, atol=1e-6)
        for t in (0, buf):
            u = blk.unpack(first=t * n, count=n)
            per, act_c, nper, rew, term = trans[t]
            assert np.array_equal(u["perspective"].cpu().numpy(), per)
            assert np.array_equal(u["next_perspective"].cpu().numpy(), nper)
            assert np.array_equal(u["action"].cpu().numpy(), act_c)
            assert np.array_equal(u["reward"].cpu().numpy(), rew)
        assert np.array_equal(gpu.getStates().cpu().numpy(), ora.states)
    gpu.close()


def test_transition_gather_rccl_single_rank_with_host_drain(T):
    """The RCCL path of TransitionGather (world of one rank on this box) incl. the asynchronous drain
    of every gathered slot to the pinned host ring; bytes must arrive unchanged and decode."""
    import os
    import torch.distributed as dist
    from toric_rl_decoder_amd import gather, wire
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
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
            slot = tg.gather(blk.buf)
            torch.cuda.synchronize()
            sent.append((slot, blk.buf.clone()))
        tg.wait()
        for slot, want in sent[-2:]:                          # the two newest flushes are still in the ring
            assert torch.equal(tg.slot_view(slot, 0), want)
            assert torch.equal(tg.slot_view(slot, 0, host=True), want.cpu())
        rec = wire.decode(tg.slot_view(sent[-1][0], 0, host=True).numpy(), d, 2 * n)
        assert rec["perspective"].shape == (2 * n, 2, d, d) and set(np.unique(rec["action"][:, 3])) <= {1, 2, 3}
        gpu.close()
    finally:
        dist.destroy_process_group()
