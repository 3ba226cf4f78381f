"""What the fused actor step spends its time on: with / without transition records, with / without time-outs.
    python tools/actor_step_cost.py [d]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import toric_rl_decoder_amd as T

d = int(sys.argv[1]) if len(sys.argv) > 1 else 7
n = 65536
env = T.make("toric-code-v0", {"size": d, "p_error": 0.1 if d == 7 else 0.15})


def run(transitions, max_steps, label):
    gpu = T.EnvSet(env, n, seed=5, numpy_io=False)
    T.load().tq_set_params(gpu._h, 0.1 if d == 7 else 0.15, 100.0, max_steps)
    gpu.resetAll()
    for t in range(76):                                   # stagger the episodes like bench.py does
        idx = torch.arange(t, n, 76, dtype=torch.int32, device=gpu.device)
        gpu.resetTerminalEnvs(idx)
        gpu.actorStep(None, want_actions=False)
    blk = gpu.newTransitionBlock(steps=8) if transitions else None
    big = torch.empty(int(3e9) // 4, dtype=torch.float32, device=gpu.device)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
    for t in range(200):
        big.zero_()                                       # the step's planes are not in any cache, as after a stack write
        ev[t][0].record()
        gpu.actorStep(None, block=blk, slot=t % 8, want_actions=True)
        ev[t][1].record()
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in ev][20:])
    ep, st = gpu.getCounters()
    print("%-46s %.2f us (median %.2f); lattices past step 75: %d" % (label, 1e3 * ms.mean(), 1e3 * np.median(ms), int((st > 75).sum().item())), flush=True)
    gpu.close()


run(True, 75, "transition records, time-out at 75 steps")
run(False, 75, "no transition records, time-out at 75 steps")
run(True, 10 ** 6, "transition records, no time-outs")
run(False, 10 ** 6, "no transition records, no time-outs")
