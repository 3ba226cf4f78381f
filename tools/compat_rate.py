#!/usr/bin/env python3
"""PCIe-inclusive rate of the numpy-compat surface (host buffers in and out every call), for DESIGN.md:
the reference-shaped loop  generatePerspective() -> numpy, selectAction(numpy q) , step(numpy actions)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import toric_rl_decoder_amd as T

n, d, steps = 16384, 7, 10
env = T.make("toric-code-v0", {"size": d, "p_error": 0.1})
envs = T.EnvSet(env, n, seed=1, numpy_io=True)
state = envs.resetAll()
t0 = time.perf_counter()
P = 0
for _ in range(steps):
    persp, pos, cnt = envs.generatePerspective()                       # D2H of the f32 stack
    P += persp.shape[0]
    act, qv = envs.selectAction(np.zeros((persp.shape[0], 3), np.float32), 1.0)
    state, rew, term, _ = envs.step(act)                              # H2D actions, D2H states
    tr = envs.generateTransition(act)
    idx = np.nonzero(term)[0]
    if idx.size:
        envs.resetTerminalEnvs(idx)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"numpy-compat loop: {n * steps / dt:.0f} env-steps/s, {P / dt:.0f} perspectives/s "
      f"({P * 2 * d * d * 4 / dt / 1e9:.2f} GB/s of stack over PCIe + numpy conversions), {n} lattices x {steps} steps")
