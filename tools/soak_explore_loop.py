#!/usr/bin/env python3
"""Soak of T.ExploreLoop (two streams, free-running): thousands of passes, then the lattices, counters and the total
perspective count against the C oracle's actor loop.  Usage (GPU box): python tools/soak_explore_loop.py [steps=3000]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import toric_rl_decoder_amd as T  # noqa: E402
from oracle.c_oracle import CEnvBatch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for d, n, p, dtype, pace in ((7, 16384, 0.1, torch.float32, "host"), (5, 32768, 0.1, torch.uint8, "host"), (9, 8192, 0.15, torch.bfloat16, "device"),
                             (3, 65536, 0.1, torch.float32, "host"), (13, 2048, 0.1, torch.float32, "host")):
    env = T.make("toric-code-v0", {"size": d, "p_error": p})
    gpu = T.EnvSet(env, n, seed=99, first_env_id=3, numpy_io=False)
    gpu.resetAll()
    nq = 2 * d * d
    stack = torch.empty((n * nq, 2, d, d), dtype=dtype, device=gpu.device)
    pos = torch.empty((n * nq, 3), dtype=torch.int32, device=gpu.device)
    offs = torch.zeros((64, (n + 2) & ~1), dtype=torch.int64, device=gpu.device)
    blocks = [gpu.newTransitionBlock(steps=8) for _ in range(2)]
    loop = T.ExploreLoop(gpu, stack, pos, offs, blocks=blocks, flush=8, overlap=True, pace=pace)
    ptot = torch.zeros((), dtype=torch.int64, device=gpu.device)
    t0 = time.perf_counter()
    for t in range(steps):
        loop.step()
        ptot += offs[t % 64, n]            # on stream A, behind write(t): row t was scanned before write(t) started
    loop.drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gpu.check()
    ce = CEnvBatch(d, n, p, seed=99, first_env_id=3)
    ce.reset()
    P, _ = ce.actor_steps(steps)
    ep, st = gpu.getCounters()
    ok = (int(ptot.item()) == P and np.array_equal(gpu.getQubits().cpu().numpy(), ce.qubits) and np.array_equal(gpu.getStates().cpu().numpy(), ce.states)
          and np.array_equal(ep.cpu().numpy().astype(np.uint32), ce.episodes) and np.array_equal(st.cpu().numpy().astype(np.uint32), ce.steps))
    # and the stack of the final lattices
    per, ppos, _ = gpu.generatePerspective(dtype=dtype)
    cp, cpos, _, _ = ce.perspectives(dtype=np.uint8)
    ok = ok and np.array_equal(ppos.cpu().numpy(), cpos) and bool((per.float().cpu() == torch.from_numpy(cp).float()).all())
    print(f"d={d} n={n} {str(dtype).split('.')[-1]} pace={pace}: {steps} passes in {dt:.2f} s ({n * steps / dt / 1e6:.1f} M env-steps/s), "
          f"P={P}: {'EQUAL to the C oracle' if ok else 'MISMATCH'}", flush=True)
    gpu.close()
    if not ok:
        sys.exit(1)
