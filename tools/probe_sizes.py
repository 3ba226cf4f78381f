"""Does the write rate depend on the SIZE CLASS of the allocation that holds the stack?  (set-up experiment)"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import toric_rl_decoder_amd as T
d, n = int(sys.argv[1]) if len(sys.argv) > 1 else 7, 65536
nq = 2 * d * d
env = T.make("toric-code-v0", {"size": d, "p_error": 0.1 if d == 7 else 0.15})
gpu = T.EnvSet(env, n, seed=2020, numpy_io=False)
gpu.resetAll()
for _ in range(30):
    gpu.actorStep(None, want_actions=False)
cnt, off = gpu.perspectiveCounts()
P = int(off[-1].item())
alg = P * (nq * 4 + 12) + n * nq
pos = torch.empty((n * nq, 3), dtype=torch.int32, device=gpu.device)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
cap = n * nq
keep = []
for gb in [float(x) for x in (sys.argv[2:] or ["2.5", "4", "7", "12"])]:
    rates = []
    for k in range(4):
        elems = max(cap * nq, int(gb * 1e9 / 4))
        arena = torch.empty(elems, dtype=torch.float32, device=gpu.device)
        keep.append(arena)
        stack = arena[:cap * nq].view(cap, 2, d, d)
        t = []
        for r in range(4):
            e0.record(); gpu.writePerspectives(stack, pos, off); e1.record(); e1.synchronize()
            t.append(e0.elapsed_time(e1))
        rates.append(alg / (np.mean(t[1:]) * 1e-3) / 1e9)
    print("arena %5.1f GB: %s GB/s" % (gb, " ".join("%5.0f" % r for r in rates)), flush=True)
rates = []
for k in range(6):
    stack = T.alloc_stack(cap, d, torch.float32, gpu.device)
    keep.append(stack)
    t = []
    for r in range(4):
        e0.record(); gpu.writePerspectives(stack, pos, off); e1.record(); e1.synchronize()
        t.append(e0.elapsed_time(e1))
    rates.append(alg / (np.mean(t[1:]) * 1e-3) / 1e9)
print("alloc_stack (2 MiB chunks): %s GB/s" % " ".join("%5.0f" % r for r in rates), flush=True)
ref = torch.empty((cap, 2, d, d), dtype=torch.float32, device=gpu.device)
gpu.writePerspectives(ref, pos, off); gpu.writePerspectives(keep[-1], pos, off)
torch.cuda.synchronize()
print("alloc_stack content == torch.empty content:", bool(torch.equal(ref[:P], keep[-1][:P])))
gpu.check()
del keep, stack, ref
import gc; gc.collect()
print("freed")
