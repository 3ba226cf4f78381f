"""Set-up experiment: does the state of the device's free memory decide which stack buffers are fast?
Times the stack write on chunked candidates (T.alloc_stack) before and after big allocate-and-free cycles.
    python tools/precondition_probe.py [d] [candidates per phase]
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import toric_rl_decoder_amd as T

d = int(sys.argv[1]) if len(sys.argv) > 1 else 7
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = 65536
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


def rate(stack):
    t = []
    for r in range(4):
        e0.record(); gpu.writePerspectives(stack, pos, off); e1.record(); e1.synchronize()
        t.append(e0.elapsed_time(e1))
    return alg / (np.mean(t[1:]) * 1e-3) / 1e9


def phase(name, kind="chunked", k=K, hold=True):
    keep, rates = [], []
    for _ in range(k):
        s = T.alloc_stack(cap, d, torch.float32, gpu.device) if kind == "chunked" else torch.empty((cap, 2, d, d), dtype=torch.float32, device=gpu.device)
        rates.append(rate(s))
        if hold:
            keep.append(s)
        else:
            del s
    free, total = torch.cuda.mem_get_info()
    print("%-44s %s GB/s   (free %.0f of %.0f GB)" % (name, " ".join("%5.0f" % r for r in rates), free / 1e9, total / 1e9), flush=True)
    return keep


def cycle(gb, wait=1.0):
    t0 = time.perf_counter()
    big = torch.empty(int(gb * 1e9), dtype=torch.uint8, device=gpu.device)
    big[::1 << 20].zero_()
    torch.cuda.synchronize()
    del big
    torch.cuda.empty_cache()
    time.sleep(wait)
    print("   -- allocated and freed %.0f GB (%.2f s incl. %.1f s wait)" % (gb, time.perf_counter() - t0, wait), flush=True)


a = phase("fresh process, chunked")
b = phase("fresh process, torch.empty", kind="torch", k=3)
del a, b
torch.cuda.empty_cache()
time.sleep(1.0)
c = phase("after freeing those (1 s later), chunked")
del c
torch.cuda.empty_cache()
free, _ = torch.cuda.mem_get_info()
cycle(0.8 * free / 1e9)
c = phase("after a cycle of 80 % of the free memory")
del c
torch.cuda.empty_cache()
cycle(0.8 * free / 1e9, wait=3.0)
c = phase("after a second cycle, 3 s wait")
c2 = phase("  ... the next ones (first ones held)")
del c, c2
torch.cuda.empty_cache()
time.sleep(1.0)
c = phase("alloc, measure, free at once (same memory again?)", hold=False)
gpu.check()
print("done")
