"""Host-side logic of the evaluation callers (no GPU): the forced-errors sampler of results/small_p_error_test.py."""
import numpy as np

import toric_rl_decoder_amd as T


def test_forced_error_sampler_statistics():
    """generateNPlusQRandomErrors (results/small_p_error_test.py:22-52), batched on the host: exactly q forced errors
    on distinct qubits, noise only elsewhere at rate p, Paulis uniform."""
    rng = np.random.default_rng(3)
    d, q, p, n = 7, 4, 0.05, 40000
    m = T.generateNPlusQRandomErrors(q, p, np.zeros((n, 2, d, d), np.int64), rng)
    flips = (m != 0).reshape(n, -1).sum(1)
    assert flips.min() >= q
    assert abs(flips.mean() - (q + (2 * d * d - q) * p)) < 0.05
    codes = np.bincount(m.ravel(), minlength=4)[1:]
    assert codes.min() / codes.max() > 0.97
    only = T.generateNRandomErrors(np.zeros((n, 2, d, d), np.int64), q, rng)
    assert ((only != 0).reshape(n, -1).sum(1) == q).all()
    per_site = (only != 0).reshape(n, -1).mean(0)
    assert abs(per_site - q / (2 * d * d)).max() < 0.006        # uniform over the 2 d^2 qubits
