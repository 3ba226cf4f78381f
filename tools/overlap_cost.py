#!/usr/bin/env python3
"""What the two events per step of T.ExploreLoop cost on the write's stream (MI355X): the loop timed (wall clock over
200 steps, the stack write's own time from every 8th step's events subtracted) with the cross-stream events left out
one by one.  Leaving one out breaks the ORDER the loop needs (results are not checked here): this is a cost table only.
Usage (GPU box): python tools/overlap_cost.py [lattices=65536] [d=7]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import toric_rl_decoder_amd as T  # noqa: E402


class CostLoop(T.ExploreLoop):
    """ExploreLoop.step with the two cross-stream orderings switchable (a cost experiment: the results are NOT valid
    without them)."""
    wait_on_A = True       # stream A waits for scan(t) of stream B
    record_on_A = True     # stream A records "write(t) done", stream B waits for it before step(t+1)

    def step(self, bracket=None):
        envs, t, k = self.envs, self.t, self.t & 1
        off = self._row(t)
        if self.overlap and self.wait_on_A == "host":
            self.scanned[k].synchronize()
        elif self.overlap and self.wait_on_A:
            self.A.wait_event(self.scanned[k])
        if bracket is not None:
            bracket[0].record(self.A)
        envs.writePerspectives(self.stack, self.positions, off)
        if bracket is not None:
            bracket[1].record(self.A)
        if self.overlap and self.record_on_A:
            self.written[k].record(self.A)
        with torch.cuda.stream(self.B):
            if self.overlap and t > 0 and self.record_on_A:
                self.B.wait_event(self.written[k ^ 1])
            blk = self.blocks[(t // self.flush) % len(self.blocks)]
            envs.actorStep(None, block=blk, slot=t % self.flush, want_actions=True)
            if (t + 1) % self.flush == 0:
                blk.computePriorities(envs.no_envs, self.flush, None, 0.95)
            envs.perspectiveCounts(self._row(t + 1))
            if self.overlap:
                self.scanned[k ^ 1].record(self.B)
        self.t = t + 1


def run(n, d, mode, steps=200):
    env = T.make("toric-code-v0", {"size": d, "p_error": 0.1})
    envs = T.EnvSet(env, n, seed=5, numpy_io=False)
    envs.resetAll()
    for t in range(76):
        idx = torch.arange(t, n, 76, dtype=torch.int32, device=envs.device)
        if idx.numel():
            envs.resetTerminalEnvs(idx)
        envs.actorStep(None, want_actions=False)
    nq = 2 * d * d
    stack = T.alloc_stack(n * nq, d, torch.float32, envs.device)
    pos = torch.empty((n * nq, 3), dtype=torch.int32, device=envs.device)
    offs = torch.zeros((8, (n + 2) & ~1), dtype=torch.int64, device=envs.device)
    blocks = [envs.newTransitionBlock(steps=8) for _ in range(2)]
    loop = CostLoop(envs, stack, pos, offs, blocks=blocks, flush=8, overlap=mode != "serial", pace="device")
    loop.wait_on_A = "host" if mode == "host_wait" else mode not in ("no_wait_on_A", "no_events")
    loop.record_on_A = mode not in ("no_record_on_A", "no_events")
    for _ in range(20):
        loop.step()
    loop.drain()
    torch.cuda.synchronize()
    evs = []
    t0 = time.perf_counter()
    for i in range(steps):
        if i % 8 == 0:
            e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            evs.append(e)
            loop.step(e)
        else:
            loop.step()
    loop.drain()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    w = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    envs.close()
    return 1e3 * dt, w


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    print(f"{n} lattices, d={d}: ms per step, stack write ms (events, every 8th step), difference in us")
    for rep in range(2):
        for mode in ("serial", "overlap", "host_wait", "no_wait_on_A"):
            step_ms, write_ms = run(n, d, mode)
            print(f"  {mode:16s} {step_ms:.4f}  {write_ms:.4f}  {1e3 * (step_ms - write_ms):6.1f}", flush=True)


if __name__ == "__main__":
    main()
