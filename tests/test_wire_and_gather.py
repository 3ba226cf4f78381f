"""Host logic: wire format round trip (numpy) and the N>1 transition gather on gloo, world_size 2."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import toric_oracle as O
from toric_rl_decoder_amd import gather, wire


def make_transitions(d, n, seed):
    env = O.OracleEnvSet(d, n, 0.1, seed=seed, first_env_id=seed * 1000)
    env.resetAll()
    bp, bpos, bcnt, boff = O.generate_perspective_batch(env.states)
    act, _, _ = O.select_action_batch(np.zeros((bp.shape[0], 3), np.float32), boff, bpos, 1.0, env.seed,
                                      env.env_ids, env.episodes, env.steps)
    prev = env.states.copy()
    _, rew, term, _ = env.step(act)
    per, a2, nper = O.generate_transition_batch(act, prev, env.states)
    return per, nper, a2, rew.astype(np.float32), term


def make_priorities(n, seed):
    """f32 priorities with awkward bit patterns (denormal, large, exact integers)."""
    rng = np.random.default_rng(1000 + seed)
    p = np.abs(rng.standard_normal(n) * 10.0 ** rng.integers(-3, 3, n)).astype(np.float32)
    p[::11] = np.float32(1e-40)
    p[5::13] = 100.0
    return p


@pytest.mark.parametrize("d", (3, 5, 7, 9, 11, 13, 15, 17, 19, 21))
def test_wire_round_trip(d):
    n, cap = 37, 64
    per, nper, act, rew, term = make_transitions(d, n, 3)
    prio = make_priorities(n, 3)
    buf = wire.encode(d, per, nper, act, rew, term, cap=cap, priority=prio)
    assert buf.size == wire.block_bytes(d, cap)
    out = wire.decode(buf, d, cap, 0, n)
    assert np.array_equal(out["perspective"], per) and np.array_equal(out["next_perspective"], nper)
    assert np.array_equal(out["action"], act) and np.array_equal(out["reward"], rew)
    assert np.array_equal(out["terminal"], term)
    assert np.array_equal(out["priority"].view(np.uint32), prio.view(np.uint32))   # bit-equal
    assert np.array_equal(out["slot"], np.arange(n))
    part = wire.decode(buf, d, cap, 5, 9)
    assert np.array_equal(part["perspective"], per[5:14]) and np.array_equal(part["action"], act[5:14])
    assert np.array_equal(part["priority"], prio[5:14])
    # the unused tail of the block (slots n..cap) holds no transitions: dropped by default
    assert wire.decode(buf, d, cap)["perspective"].shape[0] == n
    assert wire.decode(buf, d, cap, drop_empty=False)["perspective"].shape[0] == cap
    rec, pr = wire.to_records(out, d)                        # the (transition, priority) pairs of IO_mp.py:60-66
    assert rec.dtype == wire.transition_type(d)
    assert rec.dtype.itemsize == {3: 329, 5: 841, 7: 1609, 9: 2633, 11: 3913, 13: 5449, 15: 7241, 17: 9289, 19: 11593, 21: 14153}[d]
    assert np.array_equal(rec["action"]["position"][:, 1], np.full(n, d // 2))
    assert pr.dtype == np.float32 and np.array_equal(pr, prio) and len(list(zip(rec, pr))) == n
    assert wire.block_bytes(7, 1 << 16) == (1 << 16) * 45                          # 45 B / transition at d=7


def test_empty_slots_are_dropped():
    """A slot with action word 0 (op = 0: the lattice took no action that step) is not a transition."""
    d, n = 5, 20
    per, nper, act, rew, term = make_transitions(d, n, 4)
    act = act.copy()
    hole = np.array([0, 7, 19])
    act[hole] = 0
    buf = wire.encode(d, per, nper, act, rew, term, priority=make_priorities(n, 4))
    out = wire.decode(buf, d, n)
    keep = np.setdiff1d(np.arange(n), hole)
    assert np.array_equal(out["slot"], keep) and np.array_equal(out["perspective"], per[keep])
    rec, pr = wire.to_records(out, d)
    assert rec.shape[0] == n - 3 and pr.shape[0] == n - 3 and (rec["action"]["op"] >= 1).all()


def test_shard_ranges_cover_all_envs():
    for total, world in ((65536 * 8, 8), (1000, 3), (7, 8)):
        spans = [gather.shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        for (f0, c0), (f1, _) in zip(spans, spans[1:]):
            assert f0 + c0 == f1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker_two_roots(rank, world, port, d, n, q):
    """roots=2: flush i lands on rank i % 2; five flushes, two ring slots per root."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nbytes = wire.block_bytes(d, n)
        g = gather.TransitionGather(nbytes, "cpu", ring_slots=2, roots=2)
        where = []
        for flush in range(5):
            per, nper, act, rew, term = make_transitions(d, n, 10 * flush + rank)
            buf = torch.from_numpy(wire.encode(d, per, nper, act, rew, term, priority=make_priorities(n, 10 * flush + rank)))
            slot = g.gather(buf)
            where.append((g.last_root, slot))
        g.wait()
        ok = where == [(0, 0), (1, 0), (0, 1), (1, 1), (0, 0)] and g.is_root
        # what each root's ring holds at the end: rank 0 has flushes 4 (slot 0) and 2 (slot 1); rank 1 has 1 and 3
        mine = {0: ((0, 4), (1, 2)), 1: ((0, 1), (1, 3))}[rank]
        for slot, flush in mine:
            for r in range(world):
                per, nper, act, rew, term = make_transitions(d, n, 10 * flush + r)
                out = wire.decode(g.slot_view(slot, r).numpy(), d, n)
                ok &= np.array_equal(out["perspective"], per) and np.array_equal(out["action"], act)
                ok &= np.array_equal(out["priority"].view(np.uint32), make_priorities(n, 10 * flush + r).view(np.uint32))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_transition_gather_two_roots_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_two_roots, args=(r, 2, port, 5, 40, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def _worker(rank, world, port, d, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nbytes = wire.block_bytes(d, n)
        g = gather.TransitionGather(nbytes, "cpu", ring_slots=2)
        slots = []
        for flush in range(3):                              # three flushes through a 2-slot ring
            per, nper, act, rew, term = make_transitions(d, n, 10 * flush + rank)
            buf = torch.from_numpy(wire.encode(d, per, nper, act, rew, term,
                                               priority=make_priorities(n, 10 * flush + rank)))
            slots.append(g.gather(buf))
        g.wait()
        ok = True
        if rank == 0:
            assert slots == [0, 1, 0]
            for r in range(world):                          # slot 0 now holds flush 2, slot 1 flush 1
                for slot, flush in ((0, 2), (1, 1)):
                    per, nper, act, rew, term = make_transitions(d, n, 10 * flush + r)
                    out = wire.decode(g.slot_view(slot, r).numpy(), d, n)
                    ok &= np.array_equal(out["perspective"], per) and np.array_equal(out["next_perspective"], nper)
                    ok &= np.array_equal(out["action"], act) and np.array_equal(out["reward"], rew)
                    # the priority travels in the same block and arrives bit-equal (Actor_mp.py:152, IO_mp.py:60-66)
                    want = make_priorities(n, 10 * flush + r)
                    ok &= np.array_equal(out["priority"].view(np.uint32), want.view(np.uint32))
                    rec, pr = wire.to_records(out, d)
                    ok &= rec.shape[0] == n and np.array_equal(pr, want)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_transition_gather_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, 50, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def _worker_weights(rank, world, port, q):
    """Learner (rank 0) -> actor (rank 1): the flattened NN_11 parameter vector in one broadcast
    (Learner_mp.py:124-130 -> Actor_mp.py:133-144)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch.nn.utils import parameters_to_vector
        from toric_rl_decoder_amd.policy import NN_11
        torch.manual_seed(100 + rank)                        # every rank starts from different weights
        model = NN_11(5, 3)
        torch.manual_seed(100)
        learner = NN_11(5, 3)                                # what rank 0 holds
        before = parameters_to_vector(model.parameters()).detach().clone()
        buf = gather.broadcast_weights(model, src=0)
        after = parameters_to_vector(model.parameters()).detach()
        want = parameters_to_vector(learner.parameters()).detach()
        ok = torch.equal(after.view(torch.int32), want.view(torch.int32))          # bit for bit
        ok &= buf.dtype == torch.float32 and buf.numel() == want.numel() and torch.equal(buf, want)
        ok &= (rank == 0) == bool(torch.equal(before, after))                      # only the actor's weights changed
        # a second round with the learner's weights perturbed re-uses the staging buffer
        if rank == 0:
            with torch.no_grad():
                for p in model.parameters():
                    p.mul_(1.5)
        buf2 = gather.broadcast_weights(model, src=0, buffer=buf)
        ok &= buf2 is buf and torch.equal(parameters_to_vector(model.parameters()).detach(), want * 1.5)
        q.put((rank, bool(ok), int(want.numel())))
    finally:
        dist.destroy_process_group()


def test_weight_broadcast_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_weights, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[:2] for r in res] == [(0, True), (1, True)]
    assert res[0][2] == res[1][2] > 500000                   # NN_11 at d=5: ~0.88 M parameters in one message
