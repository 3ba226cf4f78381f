"""Unit test of the product's bit-plane header (toric-rl-decoder_amd/csrc/lattice.hpp) on the CPU:
tests/host_lattice_shim.cpp instantiates the same templates the HIP kernels use, g++ builds it
into a temp dir, and every operation is compared with the oracle.  Test-only build: the product
itself has no CPU path."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import toric_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = (3, 5, 7, 9, 11, 13, 15, 17, 19, 21)


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    out = tmp_path_factory.mktemp("shim") / "libshim.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                           "-I", os.path.join(ROOT, "toric-rl-decoder_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host_lattice_shim.cpp"), "-o", str(out)])
    return C.CDLL(str(out))


def P(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("d", SIZES)
def test_bitplane_algebra(shim, d):
    rng = np.random.default_rng(d)
    n, nq = 300, 2 * d * d
    q = rng.integers(0, 4, (n, 2, d, d)).astype(np.uint8)
    q[rng.random(q.shape) < 0.7] = 0
    st = np.zeros_like(q)
    assert shim.shim_syndrome(d, n, P(q), P(st)) == 0
    assert np.array_equal(st, O.syndrome(q))

    cnt = np.zeros(n, np.int32)
    masks = np.zeros_like(q)
    shim.shim_counts(d, n, P(st), P(cnt), P(masks))
    hm = O.hit_masks(st)
    assert np.array_equal(masks.astype(bool), hm) and np.array_equal(cnt, hm.reshape(n, -1).sum(1))

    lut = np.zeros((nq, nq), np.int32)
    shim.shim_lut(d, P(lut))
    assert np.array_equal(lut, O.perspective_source_index(d).reshape(nq, nq))

    act = np.stack((rng.integers(0, 2, n), rng.integers(0, d, n), rng.integers(0, d, n), rng.integers(1, 4, n)),
                   1).astype(np.int32)
    dense = (rng.random((n, 2, d, d)) < 0.5).astype(np.uint8)
    out = np.zeros_like(dense)
    shim.shim_perspective(d, n, P(dense), P(act), P(out))
    per, _, _ = O.generate_transition_batch(act, dense, dense)
    assert np.array_equal(out, per)

    qq, st3, g = q.copy(), np.zeros_like(q), np.zeros(n, np.int32)
    shim.shim_step(d, n, P(qq), P(act), P(st3), P(g))
    oq, ns, _, _ = O.step_lattices(q, st, act)
    assert np.array_equal(qq, oq) and np.array_equal(st3, ns)
    assert np.array_equal(g.astype(bool), O.eval_ground_state(oq))


@pytest.mark.parametrize("d", SIZES)
def test_perspective_bitstream_algebra(shim, d):
    """PStream (the stack-write kernel's algebra: rotate once, roll rows from a table, masked column
    rolls, OR into the lattice's bit string, window reads) reproduces the oracle's stack bit for bit,
    for sparse, dense, empty and full syndromes."""
    rng = np.random.default_rng(200 + d)
    n = 120
    _, st = O.reset_lattices(3, np.arange(n), 0, 0.12, d)
    st[40:80] = (rng.random((40, 2, d, d)) < 0.5)
    st[80] = 0
    st[81] = 1
    st[82] = 0
    st[82, 1, d - 1, d - 1] = 1
    per, pos, cnt, off = O.generate_perspective_batch(st)
    out = np.zeros_like(per)
    assert shim.shim_stream_stack(d, n, P(st), P(off), P(out)) == 0
    assert np.array_equal(out, per)


@pytest.mark.parametrize("d", SIZES)
def test_reset_matches_oracle(shim, d):
    rng = np.random.default_rng(100 + d)
    n = 300
    ep = rng.integers(0, 5, n).astype(np.uint32)
    p = np.full(n, 0.1)
    p[:60] = 0.02
    q, st, rounds = np.zeros((n, 2, d, d), np.uint8), np.zeros((n, 2, d, d), np.uint8), np.zeros(n, np.int32)
    shim.shim_reset(d, n, C.c_uint64(99), C.c_int64(5), P(ep), P(p), P(q), P(st), P(rounds))
    oq, os_ = O.reset_lattices(99, np.arange(5, 5 + n), ep, p, d)
    assert np.array_equal(q, oq) and np.array_equal(st, os_)
    assert rounds.min() >= 1 and (d > 3 or rounds.max() > 1)


@pytest.mark.parametrize("d", SIZES)
def test_fixed_n_sampler_matches_oracle(shim, d):
    """config "min_qubit_errors" = n: exactly n errors per reset, uniform over positions and Paulis."""
    rng = np.random.default_rng(300 + d)
    n = 400
    ep = rng.integers(0, 5, n).astype(np.uint32)
    for n_err in (1, d // 2 + 1, 2 * d * d):
        q, st = np.zeros((n, 2, d, d), np.uint8), np.zeros((n, 2, d, d), np.uint8)
        shim.shim_reset_n(d, n, C.c_uint64(7), C.c_int64(11), P(ep), n_err, P(q), P(st))
        oq, os_ = O.reset_lattices(7, np.arange(11, 11 + n), ep, 0.1, d, min_errors=n_err)
        assert np.array_equal(q, oq) and np.array_equal(st, os_)
        assert ((q != 0).reshape(n, -1).sum(1) == n_err).all() and st.reshape(n, -1).any(1).all()
    # uniformity over positions (one error, many lattices): every qubit is hit about equally often
    big = max(20000, 120 * 2 * d * d)                       # ~120 expected hits per qubit at least
    q1, _ = O.reset_lattices(1, np.arange(big), 0, 0.1, d, min_errors=1)
    hits = (q1 != 0).reshape(big, -1).sum(0)
    assert hits.sum() == big and hits.min() > 0.6 * big / (2 * d * d)


def test_philox_header(shim):
    out = np.zeros(4, np.uint32)
    shim.shim_philox(P(np.array([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], np.uint32)),
                     P(np.array([0xa4093822, 0x299f31d0], np.uint32)), P(out))
    assert [int(x) for x in out] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
