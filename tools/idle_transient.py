#!/usr/bin/env python3
"""Does the stack write run slower for a while after the GPU was idle?  One ExploreLoop, one buffer; series of 40 write
times (HIP events) taken (a) right after the previous series, (b) after torch.cuda.synchronize() only, (c) after 2 ms,
(d) after 50 ms of host sleep.  Usage (GPU box): python tools/idle_transient.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import toric_rl_decoder_amd as T  # noqa: E402

n, d = 65536, 7
env = T.make("toric-code-v0", {"size": d, "p_error": 0.1})
envs = T.EnvSet(env, n, seed=5, numpy_io=False)
envs.resetAll()
for t in range(76):
    idx = torch.arange(t, n, 76, dtype=torch.int32, device=envs.device)
    if idx.numel():
        envs.resetTerminalEnvs(idx)
    envs.actorStep(None, want_actions=False)
nq = 2 * d * d
pos = torch.empty((n * nq, 3), dtype=torch.int32, device=envs.device)
offs = torch.zeros((8, (n + 2) & ~1), dtype=torch.int64, device=envs.device)
blocks = [envs.newTransitionBlock(steps=8) for _ in range(2)]
for overlap in (True, False):
    loop = T.ExploreLoop(envs, None, pos, offs, blocks=blocks, flush=8, overlap=overlap)
    stack, rep = envs.pickStackBuffer(8, positions=pos, timer=loop.time_writes, park=True)
    loop.stack = stack
    print("overlap" if overlap else "one stream", "probe %.4f" % rep["probe_ms_chosen"])
    loop.time_writes(stack, 60)
    for name, pause in (("back to back", None), ("after synchronize", 0.0), ("after 2 ms idle", 0.002), ("after 50 ms idle", 0.05), ("after 500 ms idle", 0.5),
                        ("back to back again", None)):
        if pause is not None:
            torch.cuda.synchronize()
            if pause:
                time.sleep(pause)
        x = np.array(loop.time_writes(stack, 40, skip=0))
        print("  %-20s first %s   mean of 10s: %s" % (name, [round(float(v), 4) for v in x[:5]], [round(float(x[i:i + 10].mean()), 4) for i in range(0, 40, 10)]), flush=True)
    loop.drain()
    torch.cuda.synchronize()
    envs.releaseParked()
