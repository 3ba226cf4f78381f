"""Set-up experiment in a PyTorch process: the stack write on tq_stack_alloc buffers made with different walks
(TORIC_STACK_WALK = "steps,shift MiB"; "0,0" = mapped once, never moved), and how long the fast state lasts.
    python tools/walk_probe.py [d] [lattices]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import toric_rl_decoder_amd as T

d = int(sys.argv[1]) if len(sys.argv) > 1 else 7
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
nq = 2 * d * d
env = T.make("toric-code-v0", {"size": d, "p_error": 0.1 if d == 7 else 0.15})
gpu = T.EnvSet(env, n, seed=2020, numpy_io=False)
gpu.resetAll()
for _ in range(30):
    gpu.actorStep(None, want_actions=False)
cnt, off = gpu.perspectiveCounts()
P = int(off[-1].item())
alg = P * (nq * 4 + 12) + n * nq
cap = n * nq
pos = torch.empty((cap, 3), dtype=torch.int32, device=gpu.device)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def rate(stack, reps=4):
    t = []
    for r in range(reps + 1):
        e0.record(); gpu.writePerspectives(stack, pos, off); e1.record(); e1.synchronize()
        t.append(e0.elapsed_time(e1))
    return alg / (np.mean(t[1:]) * 1e-3) / 1e9


print("d=%d, %d lattices, %.2f GB stack (capacity %.2f GB)" % (d, n, P * nq * 4 / 1e9, cap * nq * 4 / 1e9), flush=True)
ref = torch.empty((cap, 2, d, d), dtype=torch.float32, device=gpu.device)
print("torch.empty: %.0f GB/s" % rate(ref), flush=True)
keep = {}
WALKS = os.environ.get("WALKS", "0,0 16,128 8,128 32,128 16,256 16,64 16,32 4,128 16,128").split()
for walk in WALKS:
    os.environ["TORIC_STACK_WALK"] = walk
    rs, ts = [], []
    for k in range(2):
        t0 = time.perf_counter()
        s = T.alloc_stack(cap, d, torch.float32, gpu.device)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        rs.append(rate(s))
        keep[(walk, k, len(keep))] = s
    print("walk %-8s: %s GB/s   (alloc %.2f s)" % (walk, " ".join("%5.0f" % r for r in rs), np.mean(ts)), flush=True)
del os.environ["TORIC_STACK_WALK"]
print("the same buffers again:", " ".join("%5.0f" % rate(s) for s in keep.values()), flush=True)
gpu.writePerspectives(ref, pos, off)
ok = []
for s in keep.values():
    s.fill_(3.0)
    gpu.writePerspectives(s, pos, off)
    ok.append(bool(torch.equal(ref[:P], s[:P])) and bool((s[P:] == 3).all()))
print("content == torch.empty buffer's, nothing written behind the stack:", ok, flush=True)
best = max(keep.values(), key=lambda s: rate(s, 2))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 6:
    for _ in range(1500):
        gpu.actorStep(None, want_actions=False)
        gpu.perspectiveCounts(off)
        gpu.writePerspectives(best, pos, off)
    torch.cuda.synchronize()
    P = int(off[-1].item()); alg = P * (nq * 4 + 12) + n * nq
    print("  after %4.1f s of the step loop: %.0f GB/s on the walked buffer, %.0f on torch.empty (%.1f perspectives per lattice)" %
          (time.perf_counter() - t0, rate(best), rate(ref), P / n), flush=True)
junk = [torch.empty(int(8e9), dtype=torch.uint8, device=gpu.device) for _ in range(6)]
for j in junk:
    j.zero_()
del junk
torch.cuda.empty_cache()
time.sleep(1.0)
print("after allocating, writing and freeing 48 GB elsewhere: %.0f GB/s" % rate(best), flush=True)
gpu.check()
print("done")
