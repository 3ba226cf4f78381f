"""tq_persp_count_write (scan folded into the stack write, one launch) against the two launches
tq_persp_count + tq_persp_write and against the oracle, bit-exact, on a real MI355X.

generatePerspectiveBatch + cumsum + concatenate: numba/util_actor.py:33-39,56-67.
"""
import os

import numpy as np
import pytest
import torch

from oracle import toric_oracle as O

pytestmark = pytest.mark.gpu

SIZES = (3, 5, 7, 9, 11, 13, 15)
P_OF = {3: 0.1, 5: 0.1, 7: 0.1, 9: 0.15, 11: 0.1, 13: 0.1, 15: 0.08}


@pytest.fixture(scope="module")
def T():
    import toric_rl_decoder_amd as T
    assert torch.cuda.is_available(), "these tests need the GPU"
    assert os.path.exists(T.LIB_PATH), "libtoricenv.so must be built (no fallback path exists)"
    T.load()
    return T


def handle(T, d, n, seed=77, **kw):
    env = T.make("toric-code-v0", {"size": d, "min_qubit_errors": 0, "p_error": P_OF[d]})
    return T.EnvSet(env, n, seed=seed, numpy_io=False, **kw)


def both_ways(gpu, dtype=torch.float32, cap=None, with_pos=True, expect_capacity=False):
    """Two launches, then one, into canary-filled buffers of the same capacity -> (stack, positions, offsets, counts) of
    the one-launch form after asserting that everything, canaries included, equals the two-launch form."""
    d, n = gpu.size, gpu.no_envs
    nq = 2 * d * d
    cnt, off = gpu.perspectiveCounts()
    cnt, off = cnt.clone(), off.clone()
    P = int(off[-1].item())
    cap = P if cap is None else cap
    cap_alloc = max(cap, 1)

    def fresh():
        return (torch.full((cap_alloc * nq + 300,), 7, dtype=dtype, device=gpu.device),
                torch.full((3 * cap_alloc + 70,), -5, dtype=torch.int32, device=gpu.device) if with_pos else None)

    ref, refp = fresh()
    gpu.writePerspectives(ref[:cap * nq].view(cap, 2, d, d) if cap else ref[:nq].view(1, 2, d, d)[:0], refp, off)
    if expect_capacity:
        with pytest.raises(T_ERR):
            gpu.check()
    else:
        gpu.check()
    out, outp = fresh()
    off2 = torch.full((n + 1,), -1, dtype=torch.int64, device=gpu.device)
    view = out[:cap * nq].view(cap, 2, d, d) if cap else out[:nq].view(1, 2, d, d)[:0]
    c2, o2 = gpu.countAndWritePerspectives(view, outp, off2)
    if expect_capacity:
        with pytest.raises(T_ERR):
            gpu.check()
    else:
        gpu.check()
    assert o2.data_ptr() == off2.data_ptr()
    assert torch.equal(off2, off), "offsets"
    assert torch.equal(c2, cnt), "counts"
    assert torch.equal(out, ref), "stack (canaries included)"
    if with_pos:
        assert torch.equal(outp, refp), "positions (canaries included)"
    return out, outp, off2, c2


T_ERR = None


@pytest.fixture(autouse=True)
def _err_type(T):
    global T_ERR
    T_ERR = T.ToricEnvError


@pytest.mark.parametrize("d", SIZES)
@pytest.mark.parametrize("dtype", (torch.float32, torch.bfloat16, torch.uint8))
def test_one_launch_equals_two_and_the_oracle(T, d, dtype):
    """Empty, tiny and dense lattices mixed, empties at both ends: stack, positions, offsets, counts."""
    rng = np.random.default_rng(2000 + d)
    n = 777
    q = np.zeros((n, 2, d, d), np.uint8)
    kind = rng.integers(0, 4, n)
    for e in range(n):
        if kind[e] == 1:
            q[e, rng.integers(0, 2), rng.integers(0, d), rng.integers(0, d)] = rng.integers(1, 4)
        elif kind[e] == 2:
            q[e] = (rng.random((2, d, d)) < 0.1) * rng.integers(1, 4, (2, d, d))
        elif kind[e] == 3:
            q[e] = (rng.random((2, d, d)) < 0.5) * rng.integers(1, 4, (2, d, d))
    q[:5] = 0
    q[-3:] = 0
    gpu = handle(T, d, n)
    gpu.setQubits(q)
    out, outp, off, cnt = both_ways(gpu, dtype)
    bp, bpos, bcnt, _ = O.generate_perspective_batch(O.syndrome(q))
    P, nq = bp.shape[0], 2 * d * d
    assert np.array_equal(cnt.cpu().numpy(), bcnt)
    assert np.array_equal(out[:P * nq].float().cpu().numpy(), bp.astype(np.float32).reshape(-1))
    assert np.array_equal(outp[:3 * P].cpu().numpy(), bpos.reshape(-1))
    both_ways(gpu, dtype, with_pos=False)
    gpu.close()


@pytest.mark.parametrize("n", (1, 2, 63, 64, 255, 256, 257, 1000, 1024, 1025, 4101))
def test_lattice_counts_around_the_block_sizes(T, n):
    """256 counts per level-1 sum, 1024 / 768 lattices per pass of the workgroup's own scan."""
    for d in (5, 13):
        gpu = handle(T, d, n, seed=n)
        gpu.resetAll()
        for _ in range(3):
            gpu.actorStep(None, want_actions=False)
        both_ways(gpu)
        gpu.close()


def test_mostly_empty_and_all_empty_batches(T):
    """A batch at the end of an evaluation: nearly every lattice solved (no perspectives), the survivors far apart --
    workgroups whose lattice range is thousands of empty lattices long, and most workgroups with no range at all."""
    d, n = 7, 20000
    gpu = handle(T, d, n)
    q = np.zeros((n, 2, d, d), np.uint8)
    gpu.setQubits(q)
    out, outp, off, cnt = both_ways(gpu, cap=4)                # P = 0
    assert int(off[-1].item()) == 0 and int(cnt.sum().item()) == 0
    for live in ([17], [0, n - 1], [5, 9000, 9001, 19990], list(range(3000, 3003)) + [n - 1]):
        q[:] = 0
        for e in live:
            q[e, 0, 1, 2] = 2
            q[e, 1, 4, 4] = 1
        gpu.setQubits(q)
        out, outp, off, cnt = both_ways(gpu)
        bp, bpos, bcnt, _ = O.generate_perspective_batch(O.syndrome(q[live]))
        nq = 2 * d * d
        assert np.array_equal(out[:bp.shape[0] * nq].cpu().numpy(), bp.astype(np.float32).reshape(-1))
        assert np.array_equal(cnt.cpu().numpy()[live], bcnt)
    gpu.close()


@pytest.mark.parametrize("d,n", [(5, 3000), (9, 700)])
def test_stack_that_does_not_fit(T, d, n):
    """capacity < P: TQ_E_CAPACITY is latched, the lattices that fit whole are written exactly as by the two-launch
    form, nothing behind them, and the offsets are still complete."""
    gpu = handle(T, d, n)
    gpu.resetAll()
    gpu.actorStep(None, want_actions=False)
    _, off = gpu.perspectiveCounts()
    P = int(off[-1].item())
    for cap in (P - 1, P // 2, P // 3 + 1, 5, 1):
        both_ways(gpu, cap=cap, expect_capacity=True)
    both_ways(gpu, cap=P)
    both_ways(gpu, cap=P + 1000)
    gpu.close()


def test_after_an_indexed_reset_and_with_a_stale_table(T):
    """tq_reset_idx leaves the level-1 sums stale (recomputed inside the call); the cut-point table of an earlier
    tq_persp_count must not be used for offsets the one-launch form has rewritten."""
    d, n = 7, 5000
    gpu = handle(T, d, n)
    gpu.resetAll()
    gpu.actorStep(None, want_actions=False)
    idx = torch.arange(3, n, 7, dtype=torch.int32, device=gpu.device)
    gpu.resetTerminalEnvs(idx)
    nq = 2 * d * d
    first = torch.empty((n * nq, 2, d, d), dtype=torch.float32, device=gpu.device)
    gpu.countAndWritePerspectives(first)                       # level-1 sums stale at this point
    gpu.check()
    out, _, off, _ = both_ways(gpu)
    P = int(off[-1].item())
    assert torch.equal(first.reshape(-1)[:P * nq], out[:P * nq])
    off = torch.zeros(n + 2, dtype=torch.int64, device=gpu.device)[:n + 1]
    gpu.perspectiveCounts(off)                                 # table for `off`
    for _ in range(5):
        gpu.actorStep(None, want_actions=False)                # different counts now
    stack = torch.empty((n * nq, 2, d, d), dtype=torch.float32, device=gpu.device)
    pos = torch.empty((n * nq, 3), dtype=torch.int32, device=gpu.device)
    gpu.countAndWritePerspectives(stack, pos, off)             # rewrites `off` in place
    P = int(off[-1].item())
    again = torch.empty((P, 2, d, d), dtype=torch.float32, device=gpu.device)
    gpu.writePerspectives(again, None, off)                    # must find its own cut points
    gpu.check()
    assert torch.equal(again, stack[:P])
    act, qv = gpu.selectAction(None, np.ones(n), positions=pos, offsets=off)    # the policy glue reads the same offsets / positions
    a = act.cpu().numpy()
    st = gpu.getStates().cpu().numpy()
    hit0 = np.maximum.reduce([st[:, 0], np.roll(st[:, 0], -1, 1), st[:, 1], np.roll(st[:, 1], 1, 2)])
    hit1 = np.maximum.reduce([st[:, 0], np.roll(st[:, 0], -1, 2), st[:, 1], np.roll(st[:, 1], 1, 1)])
    hit = np.stack([hit0, hit1], 1)
    assert bool(hit[np.arange(n), a[:, 0], a[:, 1], a[:, 2]].all())    # every action sits on a perspective (util.py:68-69,77-78)
    gpu.close()


def test_more_lattices_than_the_prologue_holds(T):
    """> 4096 level-1 sums: the call falls back to the two launches, same results."""
    d, n = 3, 4096 * 256 + 300
    gpu = handle(T, d, n)
    gpu.resetAll()
    both_ways(gpu)
    gpu.close()
    gpu = handle(T, d, 4096 * 256)                             # the largest batch the prologue takes
    gpu.resetAll()
    both_ways(gpu)
    gpu.close()


@pytest.mark.parametrize("d,n", [(7, 65536), (9, 65536), (7, 131072)])
def test_full_size_batches(T, d, n):
    gpu = handle(T, d, n)
    gpu.resetAll()
    for _ in range(4):
        gpu.actorStep(None, want_actions=False)
    both_ways(gpu)
    gpu.close()
