"""BASELINE configs[4] at its own per-GPU size on the one GPU of the box: the 131 072-lattice shard
of RANK 7 (global env ids 917 504 ... 1 048 575), d=7, p_error=0.10.

  (a) exact counts / offsets / positions and the exact stack of all 131 072 lattices against the C
      oracle (which tests/test_oracle.py ties to the golden-pinned numpy oracle); the shard is
      partition invariant: two handles of 65 536 lattices == one handle of 131 072.
  (b) the flush of that shard: 8 fused steps into ONE full-size packed block (131 072 x 8 slots,
      47 MB), computePrioritiesParallel on the device, through the RCCL transition gather
      (world of the one rank this box has) with the drain to the pinned host ring, then the host
      ingest -- wire.decode + wire.to_records on the pinned bytes -- compared record for record
      with the oracle's generateTransitionParallel / computePrioritiesParallel.
Reference: Actor_mp.py:146-169 (flush), IO_mp.py:60-66 (what the replay process saves).
"""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import toric_oracle as O
from oracle.c_oracle import CEnvBatch

pytestmark = pytest.mark.gpu

D, P_ERR, SHARD, RANK, SEED = 7, 0.10, 131072, 7, 2020
FIRST = RANK * SHARD


@pytest.fixture(scope="module")
def T():
    import toric_rl_decoder_amd as T
    assert torch.cuda.is_available(), "these tests need the GPU"
    T.load()
    return T


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _handle(T, n, first):
    env = T.make("toric-code-v0", {"size": D, "min_qubit_errors": 0, "p_error": P_ERR})
    return T.EnvSet(env, n, seed=SEED, first_env_id=first, numpy_io=False)


def test_configs4_rank7_shard_exact_stack_and_partition_invariance(T):
    n, nq = SHARD, 2 * D * D
    gpu = _handle(T, n, FIRST)
    lo, hi = _handle(T, n // 2, FIRST), _handle(T, n // 2, FIRST + n // 2)
    ce = CEnvBatch(D, n, P_ERR, seed=SEED, first_env_id=FIRST)
    for e in (gpu, lo, hi):
        e.resetAll()
    ce.reset()
    for _ in range(3):
        for e in (gpu, lo, hi):
            e.actorStep(None, want_actions=False)
    ce.actor_steps(3)
    states = gpu.getStates().clone()
    st_np = states.cpu().numpy()
    # the shard's lattices are those of global ids 917 504..: same qubits / syndromes / counters as the oracle's
    assert np.array_equal(gpu.getQubits().cpu().numpy(), ce.qubits) and np.array_equal(st_np, ce.states)
    ep, st = gpu.getCounters()
    assert np.array_equal(ep.cpu().numpy().astype(np.uint32), ce.episodes)
    assert np.array_equal(st.cpu().numpy().astype(np.uint32), ce.steps)
    # ... and differ from rank 0's (the RNG really is keyed by the global id)
    r0 = CEnvBatch(D, 256, P_ERR, seed=SEED, first_env_id=0)
    r0.reset()
    assert not np.array_equal(r0.qubits, _handle_qubits_after_reset(T, 256, FIRST))

    per, pos, cnt = gpu.generatePerspective(dtype=torch.float32)
    off = gpu._offsets.clone()
    cper, cpos, ccnt, coff = ce.perspectives(states=st_np, dtype=np.uint8)
    assert np.array_equal(cnt.cpu().numpy(), ccnt) and np.array_equal(off.cpu().numpy(), coff)
    assert np.array_equal(pos.cpu().numpy(), cpos)
    assert per.shape[0] == cper.shape[0] == int(coff[-1])
    step = 1 << 20
    for i in range(0, cper.shape[0], step):
        want = torch.as_tensor(cper[i:i + step], device=per.device)
        assert torch.equal(per[i:i + step], want.to(torch.float32)), f"stack differs in perspectives [{i}, {i + step})"
    # the numpy batch oracle (golden-pinned) on a strided subset of the same shard
    sel = np.unique(np.concatenate((np.arange(0, n, 257), [n - 1])))
    bp, _, _, _ = O.generate_perspective_batch(st_np[sel])
    rows = np.concatenate([np.arange(coff[e], coff[e + 1]) for e in sel])
    assert np.array_equal(cper[rows], bp)
    del cper, want

    # two 65 536-lattice handles == the one 131 072-lattice handle: lattices, offsets and the stack itself
    assert torch.equal(states, torch.cat((lo.getStates().clone(), hi.getStates().clone())))
    assert torch.equal(gpu.getQubits(), torch.cat((lo.getQubits().clone(), hi.getQubits().clone())))
    pl, posl, cntl = lo.generatePerspective(dtype=torch.float32)
    cut = int(pl.shape[0])
    assert cut == int(coff[n // 2]) and torch.equal(per[:cut], pl) and torch.equal(pos[:cut], posl)
    del pl
    ph, posh, cnth = hi.generatePerspective(dtype=torch.float32)
    assert torch.equal(per[cut:], ph) and torch.equal(pos[cut:], posh)
    assert torch.equal(cnt, torch.cat((cntl, cnth)))
    for e in (gpu, lo, hi):
        e.check()
        e.close()


def _handle_qubits_after_reset(T, n, first):
    h = _handle(T, n, first)
    h.resetAll()
    q = h.getQubits().cpu().numpy()
    h.close()
    return q


def test_configs4_full_size_flush_through_rccl_gather_to_host_records(T):
    import torch.distributed as dist
    from toric_rl_decoder_amd import gather, wire
    n, steps, disc = SHARD, 8, 0.95
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        gpu = _handle(T, n, FIRST)
        ce = CEnvBatch(D, n, P_ERR, seed=SEED, first_env_id=FIRST)
        gpu.resetAll()
        ce.reset()
        blk = gpu.newTransitionBlock(steps=steps)
        assert blk.nbytes == n * steps * 45 == wire.block_bytes(D, n * steps)          # 47 185 920 B
        tg = gather.TransitionGather(blk.nbytes, dev, ring_slots=2, host_drain=True)
        # Q-values of every step plus the step after (f32 values as a network would give them)
        rng = np.random.default_rng(44)
        Q = (rng.standard_normal((steps + 1, n, 3)) * 30).astype(np.float32)
        log_per, log_nper, log_act, log_rew, log_term = [], [], [], [], []
        for t in range(steps):
            pos, cnt, off = ce.positions()
            oact, _ = ce.select(np.zeros((pos.shape[0], 3), np.float32), off, pos, 1.0)
            prev = ce.states.copy()
            _, orew, oterm = ce.step(oact)
            tper, tact, tnper = ce.transition(oact, prev, ce.states)
            log_per.append(tper); log_nper.append(tnper); log_act.append(tact)
            log_rew.append(orew.copy()); log_term.append(oterm.copy())
            idx = np.nonzero(oterm | (ce.steps > 75))[0]
            if idx.size:
                ce.reset(idx)
            act, rew, term = gpu.actorStep(None, block=blk, slot=t)
            assert np.array_equal(act.cpu().numpy(), oact)
        assert np.array_equal(gpu.getStates().cpu().numpy(), ce.states)
        blk.computePriorities(n, steps, torch.as_tensor(Q, device=dev), disc)
        slot = tg.gather(blk.buf)
        tg.wait()
        gpu.check()
        host = tg.slot_view(slot, 0, host=True)
        assert host.is_pinned() and host.numel() == blk.nbytes
        dec = wire.decode(host.numpy(), D, n * steps)
        assert dec["perspective"].shape[0] == n * steps and np.array_equal(dec["slot"], np.arange(n * steps))
        assert np.array_equal(dec["perspective"], np.concatenate(log_per))
        assert np.array_equal(dec["next_perspective"], np.concatenate(log_nper))
        assert np.array_equal(dec["action"], np.concatenate(log_act))
        assert np.array_equal(dec["reward"], np.concatenate(log_rew))
        assert np.array_equal(dec["terminal"], np.concatenate(log_term))
        # priorities: the oracle's computePrioritiesParallel on (N, T) buffers, f64, stored as f32
        A = np.stack(log_act, axis=1).astype(np.int64)                                 # (n, T, 4)
        R = np.stack(log_rew, axis=1).astype(np.float64)
        Qn = Q.transpose(1, 0, 2).astype(np.float64)                                   # (n, T+1, 3)
        want = O.compute_priorities(A, R, Qn[:, :-1], Qn[:, 1:], disc)
        assert np.array_equal(dec["priority"].reshape(steps, n).T, want.astype(np.float32))
        # the (transition, priority) records the replay process saves (IO_mp.py:60-66)
        rec, prio = wire.to_records(dec, D)
        assert rec.dtype == wire.transition_type(D) and rec.dtype.itemsize == 1609 and rec.shape[0] == n * steps
        assert np.array_equal(rec["perspective"], np.concatenate(log_per))
        assert np.array_equal(rec["next_perspective"], np.concatenate(log_nper))
        assert np.array_equal(rec["action"]["position"], np.concatenate(log_act)[:, :3])
        assert np.array_equal(rec["action"]["op"], np.concatenate(log_act)[:, 3])
        assert np.array_equal(rec["reward"], np.concatenate(log_rew).astype(np.float64))
        assert np.array_equal(rec["terminal"], np.concatenate(log_term))
        assert np.array_equal(prio, want.T.reshape(-1).astype(np.float32))
        gpu.close()
    finally:
        dist.destroy_process_group()
