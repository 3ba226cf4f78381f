#!/usr/bin/env python3
"""Observed-vs-recorded accuracy of the HIP path with the reference's trained NN_11 weights (the committed data
fixtures): all 8 rows of results/results_mats/RL_{5,7}.txt (plain depolarizing sampler, evaluation.py) and the
forced-errors + noise sampler of results/small_p_error_test.py (results/evaluation_size_{5,7}.txt:2).
Writes a JSON table; tests/test_gpu_accuracy.py asserts the same quantities with stated tolerances.
Usage (GPU box): python tools/accuracy_table.py [episodes] [out.json]"""
import json
import os
import sys

os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from safetensors.torch import load_file  # noqa: E402
import toric_rl_decoder_amd as T  # noqa: E402

RL = {5: [0.9929, 0.9699, 0.9286, 0.8690, 0.7809, 0.6830, 0.5752, 0.4787],
      7: [0.9977, 0.9888, 0.9602, 0.9094, 0.8109, 0.6947, 0.5665, 0.4278]}
NQ = {5: dict(ground=0.9159, cleared=0.99998, mean_q=92.602, steps=5.3), 7: dict(ground=0.978138, cleared=0.999966, mean_q=91.137, steps=8.6)}


def main():
    episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "accuracy_table.json")
    ps = [float(x) for x in np.linspace(0.05, 0.19, 8)]
    res = {"episodes": episodes, "rl": {}, "nq": {}}
    for d in (5, 7):
        model = T.NN_11(d, 3)
        model.load_state_dict(load_file(os.path.join(ROOT, "tests", "golden", f"nn11_d{d}_converged.safetensors")))
        rows = []
        for i, p in enumerate(ps):
            c, g, st, mq, failed = T.evaluate(model, "toric-code-v0", {"size": d, "min_qubit_errors": 0}, d // 2, "cuda", [p],
                                              num_of_episodes=episodes, epsilon=0.0, num_of_steps=75, seed=777 + i, chunk=1 << 14,
                                              round_like_reference=False)
            success = 1.0 - (len(failed) // 2) / episodes
            want = RL[d][i]
            sigma = float(np.sqrt(want * (1 - want) / episodes))
            rows.append(dict(p=p, success=success, ground=float(g[0]), cleared=float(c[0]), steps=float(st[0]), mean_q=float(mq[0]),
                             recorded=want, sigma=sigma, z=(success - want) / sigma))
            print(f"d={d} p={p:.2f}: success {success:.4f} recorded {want:.4f} z={rows[-1]['z']:+.2f} cleared {c[0]:.4f} steps {st[0]:.2f} meanQ {mq[0]:.2f}", flush=True)
        res["rl"][d] = rows
        c, g, st, mq, table, n_fail, p_l, failed = T.prediction_smart(
            model, "toric-code-v0", {"size": d, "min_qubit_errors": 0}, d // 2, "cuda", [0.05], num_of_episodes=episodes, epsilon=0.0,
            num_of_steps=75, nbr_of_qubit_errors=d // 2 + 1, seed=4242 + d, chunk=1 << 14, round_like_reference=False)
        want = NQ[d]
        sigma = float(np.sqrt(want["ground"] * (1 - want["ground"]) / episodes))
        res["nq"][d] = dict(ground=float(g[0]), cleared=float(c[0]), steps=float(st[0]), mean_q=float(mq[0]), recorded=want, sigma=sigma,
                            z=(float(g[0]) - want["ground"]) / sigma, P_l=float(p_l[0]), N_fail=float(n_fail))
        print(f"d={d} N+Q p=0.05: ground {g[0]:.4f} (recorded {want['ground']}) z={res['nq'][d]['z']:+.2f} cleared {c[0]:.5f} steps {st[0]:.3f} "
              f"(recorded {want['steps']}) meanQ {mq[0]:.3f} (recorded {want['mean_q']})", flush=True)
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
