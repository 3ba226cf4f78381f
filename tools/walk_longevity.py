"""Does a walked-in buffer stay fast?  The step loop for a few minutes, the stack-write rate on the walked-in buffer and on
a torch.empty buffer every 15 s.   python tools/walk_longevity.py [seconds]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import toric_rl_decoder_amd as T

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 180
d, n = 7, 65536
nq = 2 * d * d
env = T.make("toric-code-v0", {"size": d, "p_error": 0.1})
gpu = T.EnvSet(env, n, seed=11, numpy_io=False)
gpu.resetAll()
for t in range(76):
    gpu.resetTerminalEnvs(torch.arange(t, n, 76, dtype=torch.int32, device=gpu.device))
    gpu.actorStep(None, want_actions=False)
cap = n * nq
pos = torch.empty((cap, 3), dtype=torch.int32, device=gpu.device)
walked = T.alloc_stack(cap, d, torch.float32, gpu.device)
plain = torch.empty((cap, 2, d, d), dtype=torch.float32, device=gpu.device)
off = torch.zeros(n + 2, dtype=torch.int64, device=gpu.device)[:n + 1]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def rate(stack):
    t = []
    for r in range(4):
        gpu.actorStep(None, want_actions=False)
        gpu.perspectiveCounts(off)
        e0.record(); gpu.writePerspectives(stack, pos, off); e1.record(); e1.synchronize()
        P = int(off[-1].item())
        t.append((P * (nq * 4 + 12) + n * nq) / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    return float(np.mean(t[1:]))


t0 = time.perf_counter()
nxt = 0.0
steps = 0
while True:
    now = time.perf_counter() - t0
    if now >= nxt:
        print("%6.1f s, %7d steps: walked-in %.0f GB/s   torch.empty %.0f GB/s" % (now, steps, rate(walked), rate(plain)), flush=True)
        nxt += 15.0
        if now >= secs:
            break
    for _ in range(500):
        gpu.actorStep(None, want_actions=False)
        gpu.perspectiveCounts(off)
        gpu.writePerspectives(walked, pos, off)
    torch.cuda.synchronize()
    steps += 500
gpu.check()
print("done")
