"""A/B in ONE process on ONE stack buffer, alternating blocks of steps of the bench's plain actor-loop pass:
  * scan + write as two launches (tq_persp_count, tq_persp_write) against the one-launch experiment
    (tq_persp_count_write: the scan in the write kernel's prologue -- commit 2d73a3d has it, the library no longer does;
    the one-launch rows are skipped when the entry point is absent),
  * with and without a pair of HIP events around the write of every step.
    python tools/ab_scan.py [d] [p_error] [lattices] [rounds] [steps per block]
Prints wall-clock microseconds per step of each form (synchronised around every block) and the HIP-event time of the
(count +) write launch(es).  Result (profiles/r03_stack_write_ab.txt, 10): the one-launch form gains nothing, a pair
of event records costs 6 us per step."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toric_rl_decoder_amd as T  # noqa: E402


def main():
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.10
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 60
    dev = torch.device("cuda:0")
    nq = 2 * d * d
    env = T.make("toric-code-v0", {"size": d, "min_qubit_errors": 0, "p_error": p})
    envs = T.EnvSet(env, n, device=dev, seed=1234, numpy_io=False)
    envs.resetAll()
    for t in range(75):
        idx = torch.arange(t, n, 75, dtype=torch.int32, device=dev)
        envs.resetTerminalEnvs(idx)
        envs.actorStep(None, want_actions=False)
    pos = torch.empty((n * nq, 3), dtype=torch.int32, device=dev)
    stack, rep = envs.pickStackBuffer(8, positions=pos, park=True)
    print("buffer probe:", ["%.1f" % (1e3 * x) for x in rep["write_ms"]], "us; chosen", rep["chosen"], flush=True)
    off = torch.zeros(n + 2, dtype=torch.int64, device=dev)[:n + 1]
    blk = envs.newTransitionBlock(steps=8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def block(fused, k, events=True):
        torch.cuda.synchronize()
        ev = []
        t0 = time.perf_counter()
        for t in range(k):
            if events:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            if fused:
                envs.countAndWritePerspectives(stack, pos, off)
            else:
                envs.perspectiveCounts(off)
                envs.writePerspectives(stack, pos, off)
            if events:
                b.record()
                ev.append((a, b))
            envs.actorStep(None, block=blk, slot=t % 8, want_actions=True)
            if t % 8 == 7:
                blk.computePriorities(n, 8, None, 0.95)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return 1e6 * dt / k, 1e3 * float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else float("nan")

    have_fused = hasattr(envs, "countAndWritePerspectives")
    forms = [("two launches, events", False, True), ("one launch, events", True, True),
             ("two launches, no events", False, False), ("one launch, no events", True, False)]
    forms = [f for f in forms if have_fused or not f[1]]
    for f in forms:
        block(f[1], 20, f[2])
    res = {f[0]: [] for f in forms}
    for r in range(rounds):
        for name, fused, events in forms:
            res[name].append(block(fused, steps, events))
        print("round %d: " % r + "   ".join("%s %.1f (%.1f)" % (name, res[name][-1][0], res[name][-1][1]) for name, _, _ in forms), flush=True)
    print("means, us per step (HIP-event time of count + write):")
    for name, _, _ in forms:
        a = np.array(res[name])
        print("  %-26s %8.2f  (%.2f)" % (name, a[:, 0].mean(), a[:, 1].mean()))
    envs.check()
    envs.close()


if __name__ == "__main__":
    main()
