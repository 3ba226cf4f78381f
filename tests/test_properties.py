"""Property tests (hypothesis) of the oracle's lattice algebra and of the host-side wire format:
invariants listed in SURVEY 8(a)/(c) that must hold for ANY qubit configuration, not just sampled ones."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import toric_oracle as O
from toric_rl_decoder_amd import gather, wire

SIZES = st.sampled_from([3, 5, 7, 9])


@st.composite
def lattices(draw, max_n=6):
    d = draw(SIZES)
    n = draw(st.integers(1, max_n))
    seed = draw(st.integers(0, 2**32 - 1))
    dens = draw(st.sampled_from([0.0, 0.05, 0.2, 0.6, 1.0]))
    rng = np.random.default_rng(seed)
    q = rng.integers(1, 4, (n, 2, d, d)).astype(np.uint8) * (rng.random((n, 2, d, d)) < dens)
    return d, q.astype(np.uint8), rng


@settings(max_examples=60, deadline=None)
@given(lattices())
def test_syndrome_invariants(arg):
    d, q, rng = arg
    n = q.shape[0]
    s = O.syndrome(q)
    # every Pauli flips two checks of a sector: defect parity is even per sector
    assert (s[:, 0].reshape(n, -1).sum(1) % 2 == 0).all() and (s[:, 1].reshape(n, -1).sum(1) % 2 == 0).all()
    # linearity over GF(2): syndrome(a xor b) = syndrome(a) xor syndrome(b)   (Pauli product = XOR of codes)
    q2 = rng.integers(0, 4, q.shape).astype(np.uint8)
    assert np.array_equal(O.syndrome(q ^ q2), O.syndrome(q) ^ O.syndrome(q2))
    # translation covariance: rolling the qubits rolls the syndrome
    a, b = int(rng.integers(0, d)), int(rng.integers(0, d))
    assert np.array_equal(O.syndrome(np.roll(q, (a, b), axis=(2, 3))), np.roll(s, (a, b), axis=(2, 3)))


@settings(max_examples=60, deadline=None)
@given(lattices())
def test_perspective_invariants(arg):
    d, q, rng = arg
    s = O.syndrome(q)
    n = s.shape[0]
    per, pos, cnt, off = O.generate_perspective_batch(s)
    hm = O.hit_masks(s).reshape(n, -1)
    assert np.array_equal(cnt, hm.sum(1)) and off[-1] == per.shape[0]
    # a qubit is a hit iff its Pauli X, Y or Z would change an existing defect: flipping it with Y touches a defect
    for e in range(n):
        rows = slice(off[e], off[e + 1])
        # every perspective is a permutation of the syndrome (same number of defects)
        assert (per[rows].reshape(-1, 2 * d * d).sum(1) == s[e].sum()).all()
        # positions are sorted: layer-major, then row-major (np.argwhere order of the reference)
        key = pos[rows, 0].astype(np.int64) * d * d + pos[rows, 1] * d + pos[rows, 2]
        assert (np.diff(key) > 0).all()
    gs = d // 2
    if per.shape[0]:
        centre = per[:, 0, gs, gs] | per[:, 0, (gs + 1) % d, gs] | per[:, 1, gs, gs] | per[:, 1, gs, gs - 1]
        assert centre.all()                                   # centred-frame property
    # both forms of the oracle agree
    rp, rpos, rcnt = O.generate_perspective_batch_ref(gs, d, s.astype(np.int64))
    assert np.array_equal(rp, per) and np.array_equal(rpos, pos) and np.array_equal(rcnt, cnt)


@settings(max_examples=40, deadline=None)
@given(lattices(max_n=4))
def test_step_and_transition_invariants(arg):
    d, q, rng = arg
    s = O.syndrome(q)
    n = q.shape[0]
    act = np.stack((rng.integers(0, 2, n), rng.integers(0, d, n), rng.integers(0, d, n), rng.integers(1, 4, n)), 1)
    q1, s1, r1, t1 = O.step_lattices(q, s, act)
    q2, s2, _, _ = O.step_lattices(q1, s1, act)
    assert np.array_equal(q2, q) and np.array_equal(s2, s)     # P.P = I
    delta = s.reshape(n, -1).sum(1).astype(int) - s1.reshape(n, -1).sum(1)
    assert np.array_equal(r1, np.where(t1, 100.0, delta)) and np.array_equal(t1, s1.reshape(n, -1).sum(1) == 0)
    per, a2, nper = O.generate_transition_batch(act, s, s1)
    gs = d // 2
    for e in range(n):
        diff = {tuple(x) for x in np.argwhere(per[e] != nper[e])}
        op = act[e, 3]
        want = set()
        if op in (2, 3):
            want |= {(0, gs, gs), (0, (gs + 1) % d, gs)}       # the acted qubit's two vertices in the centred frame
        if op in (1, 2):
            want |= {(1, gs, gs), (1, gs, gs - 1)}             # and its two plaquettes, for either layer
        assert diff == want and a2[e].tolist() == [act[e, 0], gs, gs, op]


@settings(max_examples=40, deadline=None)
@given(st.sampled_from([3, 5, 7, 9, 11]), st.integers(1, 50), st.integers(0, 30), st.integers(0, 2**31))
def test_wire_round_trip_any_shape(d, n, extra, seed):
    rng = np.random.default_rng(seed)
    per = (rng.random((n, 2, d, d)) < 0.3).astype(np.uint8)
    nper = (rng.random((n, 2, d, d)) < 0.3).astype(np.uint8)
    act = np.stack((rng.integers(0, 2, n), np.full(n, d // 2), np.full(n, d // 2), rng.integers(1, 4, n)), 1).astype(np.int32)
    rew = rng.integers(-4, 101, n).astype(np.float32)
    term = rng.random(n) < 0.3
    cap = n + extra
    buf = wire.encode(d, per, nper, act, rew, term, cap=cap)
    assert buf.size == wire.block_bytes(d, cap)
    out = wire.decode(buf, d, cap, 0, n)
    assert np.array_equal(out["perspective"], per) and np.array_equal(out["next_perspective"], nper)
    assert np.array_equal(out["action"], act) and np.array_equal(out["reward"], rew) and np.array_equal(out["terminal"], term)


@settings(max_examples=50, deadline=None)
@given(st.integers(1, 10**7), st.integers(1, 64))
def test_shard_ranges_partition(total, world):
    spans = [gather.shard_range(total, world, r) for r in range(world)]
    assert spans[0][0] == 0 and sum(c for _, c in spans) == total
    assert all(f0 + c0 == f1 for (f0, c0), (f1, _) in zip(spans, spans[1:]))
    assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


@settings(max_examples=60, deadline=None)
@given(bias=st.integers(0, 16), seed=st.integers(0, 2**31 - 1), irregular=st.integers(0, 40))
def test_workgroup_shares_cover_the_fine_parts_whatever_the_dispatch(bias, seed, irregular):
    """csrc/stream_write.hpp, "this workgroup's range": 256 workgroups, 128 large slots of 32 + bias fine parts and 128
    small ones of 32 - bias; a workgroup on an even XCD takes the next large slot, one on an odd XCD the next small
    one, a kind that has run out sends it to the other kind.  Whatever XCDs the dispatcher puts the workgroups on
    (round-robin from any start, or `irregular` of them anywhere) and in whatever order their atomics arrive, the
    ranges [f_lo, f_hi) are a partition of the 8192 fine parts."""
    rng = np.random.default_rng(seed)
    G, RR = 256, 32
    start = int(rng.integers(0, 8))
    xcd = (start + np.arange(G)) % 8
    if irregular:
        xcd[rng.choice(G, irregular, replace=False)] = rng.integers(0, 8, irregular)
    counters = [0, 0]                                        # slots[large], as in the kernel: index 1 = large
    covered = np.zeros(G * RR, np.int32)
    for b in rng.permutation(G):                             # arrival order of the workgroups' atomics
        large = int(xcd[b] % 2 == 0)
        t = counters[large]
        counters[large] += 1
        if t >= G // 2:
            large ^= 1
            t = counters[large]
            counters[large] += 1
        assert t < G // 2
        f_lo = t * 2 * RR + (0 if large else RR + bias)
        f_hi = f_lo + (RR + bias if large else RR - bias)
        covered[f_lo:f_hi] += 1
    assert (covered == 1).all()
    if not irregular:                                        # round-robin: exactly the even XCDs' workgroups hold the large slots
        assert counters == [G // 2, G // 2]
