"""Statistical end-to-end check of the env half of the oracle (the part with no bit-level
reference: gym_ToricCode is absent upstream).  The reference's committed NN_11 checkpoints were
trained against the real env; driving the ORACLE env greedily with them must reproduce the
reference's recorded success rates (results/results_mats/RL_{5,7}.txt at p = linspace(.05,.19,8)).
A wrong syndrome geometry, Pauli table, perspective order/rotation or ground-state rule would not.

Opt-in (minutes of CPU conv work): TORIC_SLOW=1, and only where the reference tree is present
(weights are read with torch.load(weights_only=True); nothing is copied).  Last run in the
authoring container is recorded in DESIGN.md section 6.
"""
import os

import numpy as np
import pytest
import torch

from oracle import toric_oracle as O

REF = os.environ.get("TORIC_REFERENCE", "/root/reference")
WEIGHTS = {5: "network/converged/Size_5_NN_11_17_Mar_2020_22_33_59.pt",
           7: "network/converged/Size_7_NN_11_random_18_Mar_2020_18_17_52.pt"}
RECORDED = {5: {0.05: 0.9929, 0.11: 0.8690, 0.19: 0.4787},       # results/results_mats/RL_5.txt
            7: {0.05: 0.9977, 0.11: 0.9094, 0.15: 0.6947}}       # results/results_mats/RL_7.txt

pytestmark = pytest.mark.skipif(os.environ.get("TORIC_SLOW") != "1" or not os.path.isdir(REF),
                                reason="opt-in slow test (TORIC_SLOW=1) that needs the reference's weights")


def run_episodes(d, p, episodes, max_steps=75, seed=11):
    from toric_rl_decoder_amd.policy import NN_11
    model = NN_11(d, 3)
    model.load_state_dict(torch.load(os.path.join(REF, WEIGHTS[d]), map_location="cpu", weights_only=True))
    model.eval()
    env = O.OracleEnvSet(d, episodes, p, seed=seed)
    env.resetAll()
    done = np.zeros(episodes, bool)
    steps = np.zeros(episodes, np.int64)
    for _ in range(max_steps):
        per, pos, cnt, off = O.generate_perspective_batch(env.states, dtype=np.float32)
        with torch.no_grad():
            q = torch.cat([model(torch.from_numpy(per[i:i + 4096])) for i in range(0, per.shape[0], 4096)]).numpy() \
                if per.shape[0] else np.zeros((0, 3), np.float32)
        act, _, _ = O.select_action_batch(q, off, pos, 0.0, seed, env.env_ids, env.episodes, env.steps)
        steps += ~done
        _, _, term, _ = env.step(act)                    # op 0 for lattices that are already solved
        done |= term
        if done.all():
            break
    ground = O.eval_ground_state(env.qubits) & done
    return ground.mean(), done.mean(), steps.mean()


@pytest.mark.parametrize("d,p,episodes", [(5, 0.05, 2000), (5, 0.11, 2000), (5, 0.19, 1500), (7, 0.05, 1000), (7, 0.15, 600)])
def test_trained_weights_reproduce_recorded_success_rates(d, p, episodes):
    ps, cleared, mean_steps = run_episodes(d, p, episodes)
    want = RECORDED[d][p]
    sigma = np.sqrt(want * (1 - want) / episodes)
    print(f"d={d} p={p}: ground-state success {ps:.4f} (recorded {want:.4f}, sigma {sigma:.4f}), "
          f"syndrome cleared {cleared:.4f}, mean steps {mean_steps:.2f}")
    assert abs(ps - want) < 4 * sigma + 0.01
    assert cleared > 0.97
