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


@pytest.mark.parametrize("d", (3, 5, 7, 9, 11))
def test_wire_round_trip(d):
    n, cap = 37, 64
    per, nper, act, rew, term = make_transitions(d, n, 3)
    buf = wire.encode(d, per, nper, act, rew, term, cap=cap)
    assert buf.size == wire.block_bytes(d, cap)
    out = wire.decode(buf, d, cap, 0, n)
    assert np.array_equal(out["perspective"], per) and np.array_equal(out["next_perspective"], nper)
    assert np.array_equal(out["action"], act) and np.array_equal(out["reward"], rew)
    assert np.array_equal(out["terminal"], term)
    part = wire.decode(buf, d, cap, 5, 9)
    assert np.array_equal(part["perspective"], per[5:14]) and np.array_equal(part["action"], act[5:14])
    rec = wire.to_records(out, d)
    assert rec.dtype.itemsize == {3: 329, 5: 841, 7: 1609, 9: 2633, 11: 3913}[d]
    assert np.array_equal(rec["action"]["position"][:, 1], np.full(n, d // 2))
    assert wire.block_bytes(7, 1 << 16) == (1 << 16) * 41                          # 41 B / transition at d=7


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
            buf = torch.from_numpy(wire.encode(d, per, nper, act, rew, term))
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
