#!/usr/bin/env python3
"""What does freeing a lot of device memory do to a loop that runs right afterwards?  One ExploreLoop on 65 536 lattices of
d=7 (bf16 stack: 0.16 ms per step, the leg of bench.py's default line that was seen losing 60 us per step between its
writes); wall time per step and write time per step of 40-step series: before anything, right after 24 x 5 GB of
tq_stack_alloc buffers were allocated and freed again, and 0.5 / 2 / 5 s later.  Usage (GPU box): python tools/free_aftermath.py"""
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
loop = T.ExploreLoop(envs, None, pos, offs, blocks=blocks, flush=8)
stack, rep = envs.pickStackBuffer(6, dtype=torch.bfloat16, positions=pos, timer=loop.time_writes)
loop.stack = stack
print("probe %.4f ms" % rep["probe_ms_chosen"], flush=True)


def series(name, steps=40):
    for _ in range(8):
        loop.step()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    loop.drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e in evs:
        loop.step(e)
    loop.drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    w = np.array([a.elapsed_time(b) for a, b in evs])
    print("  %-44s wall %.4f ms/step   write %.4f ms   between the writes %.1f us/step" % (name, 1e3 * dt / steps, w.mean(), 1e3 * (1e3 * dt / steps - w.mean())), flush=True)


series("before")
series("before, again")
for size_gb, count in ((5, 24), (2, 24)):
    bufs = [T.alloc_chunked((size_gb << 30,), torch.uint8, envs.device) for _ in range(count)]
    torch.cuda.synchronize()
    series("with %d x %d GB allocated" % (count, size_gb))
    del bufs
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    t_free = time.perf_counter()
    series("right after freeing them")
    series("the next 48 steps")
    for pause in (0.5, 2.0, 5.0):
        time.sleep(pause)
        series("%.1f s after the free" % (time.perf_counter() - t_free))
loop.drain()
torch.cuda.synchronize()
