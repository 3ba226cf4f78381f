"""Is a slow buffer slow everywhere?  Fill rates (torch zero_) of the eighths of fast and slow tq_stack_alloc buffers and of
the stack write on lattice sub-ranges that land in them.   python tools/slow_where.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import toric_rl_decoder_amd as T

d, n = 7, 65536
nq = 2 * d * d
env = T.make("toric-code-v0", {"size": d, "p_error": 0.1})
gpu = T.EnvSet(env, n, seed=3, numpy_io=False)
gpu.resetAll()
for _ in range(30):
    gpu.actorStep(None, want_actions=False)
cnt, off = gpu.perspectiveCounts()
P = int(off[-1].item())
cap = P + 1000
pos = torch.empty((cap, 3), dtype=torch.int32, device=gpu.device)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def t_write(s):
    t = []
    for r in range(4):
        e0.record(); gpu.writePerspectives(s, pos, off); e1.record(); e1.synchronize(); t.append(e0.elapsed_time(e1))
    return 1e3 * float(np.mean(t[1:]))


def fill_gbps(x, reps=30):
    x.zero_()
    e0.record()
    for _ in range(reps):
        x.zero_()
    e1.record(); e1.synchronize()
    return x.numel() * x.element_size() * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


bufs = [T.alloc_stack(cap, d, torch.float32, gpu.device) for _ in range(16)]
times = [t_write(b) for b in bufs]
order = np.argsort(times)
print("stack write, us:", ["%.0f" % t for t in times])
def rate_of(fn, nbytes, reps=20):
    fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


src = torch.empty((cap, 2, d, d), dtype=torch.float32, device=gpu.device)
gpu.writePerspectives(src, pos, off)
rnd = torch.randint(0, 2, (P * nq,), device=gpu.device, dtype=torch.int32).float()
for name, i in (("fastest", order[0]), ("second fastest", order[1]), ("slowest", order[-1]), ("second slowest", order[-2])):
    b = bufs[i].view(-1)[:P * nq]
    nb = b.numel() * 4
    print("%-15s (stack write %.0f us = %.0f GB/s)  zero_ %5.0f   fill_(1.0) %5.0f   copy_ of a stack %5.0f   copy_ of random 0/1 %5.0f GB/s written" %
          (name, times[i], (P * (nq * 4 + 12) + n * nq) / times[i] / 1e3, rate_of(lambda: b.zero_(), nb), rate_of(lambda: b.fill_(1.0), nb),
           rate_of(lambda: b.copy_(src.view(-1)[:P * nq]), nb), rate_of(lambda: b.copy_(rnd), nb)), flush=True)
