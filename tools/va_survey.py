"""Which tq_stack_alloc buffers are the fast ones?  Many candidates in one process, their virtual addresses and write times.
    python tools/va_survey.py [d] [candidates]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import toric_rl_decoder_amd as T

d = int(sys.argv[1]) if len(sys.argv) > 1 else 7
k = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n = 65536
env = T.make("toric-code-v0", {"size": d, "p_error": 0.1 if d == 7 else 0.15})
gpu = T.EnvSet(env, n, seed=3, numpy_io=False)
gpu.resetAll()
for _ in range(30):
    gpu.actorStep(None, want_actions=False)
cnt, off = gpu.perspectiveCounts()
P = int(off[-1].item())
cap = P + 1000                                             # small candidates: many of them fit
best, rep = gpu.pickStackBuffer(k, capacity=cap, park=True)
ms = np.array(rep["write_ms"])
print("capacity %.2f GB; torch.empty %.1f us at %s" % (cap * 2 * d * d * 4 / 1e9, 1e3 * ms[0], rep["addresses"][0]))
for a, t in zip(rep["addresses"][1:], ms[1:]):
    v = int(a, 16)
    print("%s  %6.1f us  %s   GiB-offset %.3f  (addr>>21)&1023=%4d  &63=%2d" % (a, 1e3 * t, "FAST" if t < 0.9 * ms[0] else "slow", (v % (1 << 30)) / (1 << 30), (v >> 21) & 1023, (v >> 21) & 63))
