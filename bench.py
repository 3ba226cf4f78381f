#!/usr/bin/env python3
"""Headline benchmark of the toric-code env hot path on MI355X.

A "step" is one pass of the hot path over one batch of lattices, the body of the reference's actor loop
(src/Actor_mp.py:104-185) with the policy network left out of the headline (it is stock torch conv work, out of scope;
epsilon starts at 1 upstream, Actor_mp.py:37, where the Q-values never influence the action):

    perspective stack write (P,2,d,d) f32 + positions            (stream A)
    eps=1 selection, env step, transition record, auto-reset, next counts (one fused kernel) -> exclusive scan of the
    counts -> every --flush steps: priorities into the packed block (computePrioritiesParallel with Q = 0)   (stream B)

through toric_rl_decoder_amd.ExploreLoop: the fused step does not depend on the stack at eps = 1, so it and the next scan
run on a second HIP stream BESIDE the stack write (--no-overlap = one stream, the reference's call order;
small batches run on one stream anyway).

Workload at N=1: BASELINE.json configs[2], the configuration the metric is quoted on: 65 536 lattices, d=7,
p_error=0.10.  Lattices are resident in HBM when the timed region starts; nothing crosses PCIe in the N=1 loop.
Set-up, untimed, per leg: (1) burn-in -- the episodes are staggered (lattice e is reset at step e mod 76), so the population
and with it perspectives per lattice is stationary and `value` does not depend on --steps; (2) the stack-buffer probe
(EnvSet.pickStackBuffer: --stack-candidates buffers, all allocated first, the write timed inside the loop, the fastest
kept; on MI355X a buffer has a write rate of its own); (3) 64 passes of the loop in one go (the ~25 passes after the
bursty probe run 2-3 % slower: profiles/r04_first_steps_after_setup.txt).  Then W warm-up passes (their writes are held
against the probe: > 8 % slower -> one re-probe; plus at most flush-1 passes so that the timed region starts on a
transition-block boundary on every rank), barrier + synchronize, exactly K timed passes, barrier + synchronize.

N>1: `python bench.py --gpus N` starts N ranks by itself (a `python -m torch.distributed.run` child, before this process
touches the GPU) and relays rank 0's JSON line; under torch.distributed.run (WORLD_SIZE set) it runs as a rank.  One rank
per GPU; every rank owns a contiguous block of global env ids (weak scaling).  Default shape for N>1: BASELINE.json
configs[4] -- 131 072 lattices per GPU, and every transition (with its priority) is delivered to the HOST replay ring:
RCCL gather of the packed blocks to rank 0 over xGMI every --flush steps (issued from stream B right behind the
priorities), then rank 0's copy stream drains each gathered slot to pinned host memory.  The rate with the ring kept in
rank 0's HBM is measured right after and reported beside it (`hbm_ring`); `ranks[]` carries every rank's own roofline.

Prints ONE JSON line on rank 0 (contract in the task statement).  Every leg -- the headline, `configs4_shard_on_one_gpu`,
`bf16_stack`, `configs3_on_one_gpu.{one_shot,chunks_4}` -- carries `stack_buffer_probe`, `stack_verified` (the timed buffer
holds the right bytes; a failure voids the line) and its own `roofline` (HIP events on the write's stream around every
--event-every-th write; `probe_ms_chosen`, `timed_write_ms`, `timed_over_probe`, `default_buffer`, `workgroup_shares` = the XCD bias in force and the probe's biased-against-equal timing); `cpu_baseline` (the
oracle's ports on the host cores, rank 0, N=1 only) and `nn_in_loop` (configs[2] as written: the stack fed to NN_11 and
device-side selection in the loop; N=1 only, --nn-steps 0 to skip).  --shards > 1, --graph and --policy nn11 take the
serial path (run_serial).
"""
import argparse
import contextlib
import gc
import json
import os
import socket
import subprocess
import sys
import time
import types

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes), before HIP initialises
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")             # NN_11 leg: no exhaustive solver search on a fresh box

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 measured copy
STACK_KERNEL = "k_persp_stream"  # the stack-write kernel libtoricenv launches (csrc/stream_write.hpp)
EPISODE = 76                    # a lattice is auto-reset once its step counter exceeds 75 (Distributed_mp.py:44)
ENVS_N1, ENVS_MULTI = 65536, 131072     # BASELINE configs[2] / configs[4] lattices per GPU
NN_CHUNK = 16384                # perspectives per NN_11 forward call
SETTLE = int(os.environ.get("TORIC_BENCH_SETTLE", "64"))   # loop passes at the end of set-up, after the bursty probe (ExploreLeg.pick_stack)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=None,
                    help="lattices per GPU (default: 65536 at N=1 = configs[2]; 131072 at N>1 = configs[4])")
    ap.add_argument("--size", type=int, default=7)
    ap.add_argument("--p-error", type=float, default=0.10)
    ap.add_argument("--seed", type=int, default=2020)
    ap.add_argument("--out-dtype", default="f32", choices=["f32", "f16", "bf16", "u8"])
    ap.add_argument("--flush", type=int, default=8, help="steps per transition block / priorities / gather")
    ap.add_argument("--shards", type=int, default=1, help="independent sub-shards (HIP streams) per GPU")
    ap.add_argument("--chunks", type=int, default=1,
                    help="write the stack of a step in this many lattice ranges, one after the other, into ONE buffer "
                         "of 1/chunks the size (tq_persp_write_range; a consumer with a small buffer, SURVEY 8d C4)")
    ap.add_argument("--event-every", type=int, default=4,
                    help="bracket every E-th stack write of the timed region with HIP events (roofline). A hipEventRecord "
                         "costs ~3 us of stream time on MI355X (tools/ab_scan.py), so bracketing every launch adds 6 us to "
                         "every step of the timed region; 1 = every launch")
    ap.add_argument("--no-transitions", action="store_true", help="do not write transition records")
    ap.add_argument("--delivery", default="auto", choices=["auto", "host", "hbm"],
                    help="N>1: where gathered transition blocks end up. host = pinned host replay ring (default for "
                         "N>1, north_star), hbm = ring in rank 0's HBM only")
    ap.add_argument("--roots", type=int, default=0,
                    help="N>1: ranks that take turns as root of the transition gather, each draining to the host over "
                         "its own PCIe link (0 = auto: 2 from 8 ranks on, where one Gen5 x16 link no longer carries "
                         "the ~63 GB/s of packed records; 1 below)")
    ap.add_argument("--no-burn-in", action="store_true", help="skip the episode-staggering burn-in")
    ap.add_argument("--stack-candidates", type=int, default=24,
                    help="stack buffers to allocate at set-up; the one the write kernel is fastest on is kept, the others are "
                         "freed (the write rate depends on the buffer: 5.1-5.5 TB/s into a plain allocation, 6.5-6.8 into most "
                         "tq_stack_alloc buffers on most boxes, profiles/r03_stack_write_ab.txt).  1 = take the first allocation as it comes")
    ap.add_argument("--stack-kinds", default="torch,chunked",
                    help="where the candidates come from (first entry: candidate 0, the rest cyclically for the others): torch = "
                         "torch.empty, chunked = T.alloc_stack (2 MiB physical "
                         "chunks, tq_stack_alloc).  Profile runs use --stack-candidates 1 --stack-kinds chunked so that every "
                         "launch of the process writes the same buffer")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--nn-steps", type=int, default=3,
                    help="N=1: timed steps of the NN_11-in-the-loop leg (configs[2] as written); 0 = skip")
    ap.add_argument("--no-shard-leg", action="store_true",
                    help="N=1: skip the extra timing of the N>1 per-GPU shape (131072 lattices); profiling runs use this "
                         "so that every k_persp_stream launch of the process has the headline shape")
    ap.add_argument("--no-events", action="store_true", help="no per-launch HIP events (pure wall clock)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="default path: run the fused step and the scan BEHIND the stack write on one stream (the reference's order) "
                         "instead of beside it on a second stream (T.ExploreLoop)")
    ap.add_argument("--bf16-steps", type=int, default=20,
                    help="N=1: timed steps of the bf16-stack leg (the stack the bf16 nn_in_loop variant consumes), with its own roofline; 0 = skip")
    ap.add_argument("--graph", action="store_true",
                    help="capture --flush steps in a HIP graph and replay it (launch-bound small batches; implies "
                         "--no-events: no per-launch timing inside a graph, so no roofline object)")
    ap.add_argument("--policy", default="explore", choices=["explore", "nn11"],
                    help="explore: eps=1 selection in the fused kernel (default, the env path alone); "
                         "nn11: NN_11 forward on the stack + device eps-greedy selection in the main loop (NN-bound)")
    ap.add_argument("--eps", type=float, default=0.1, help="epsilon of the nn11 policy")
    ap.add_argument("--nn-dtype", default="f32", choices=["f32", "bf16"],
                    help="dtype of the NN_11 forward: f32 as upstream (default; ~18 TFLOP/s in stock MIOpen = 23 s per "
                         "step at 65 536 lattices) or bf16 autocast")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------- self-launch (N > 1)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` without torch.distributed.run around it: start the N ranks as a CHILD
    process tree and relay rank 0's JSON line.  This parent never initialises the GPU (no torch.cuda
    call, not even torch is imported here) and never re-execs itself."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    if p.returncode != 0:
        sys.exit(p.returncode)
    if line is None:
        sys.exit("bench.py: the ranks exited cleanly but rank 0 printed no result line")


# ---------------------------------------------------------------------------- CPU baseline
def cpu_baseline(d, p, seed, budget_s):
    """The oracle's C actor loop (EnvSet.step + generatePerspectiveBatch + generateTransitionParallel
    restated, oracle/toric_oracle.c) timed on the host cores: bounded sample of the same workload, on
    all cores this process may use (`value`) and on one core; plus the same loop in the reference's
    own shape (per-lattice python + np.roll, int64, fresh allocations) on 256 lattices x 20 steps
    (SURVEY 8d), one process like the reference's actor."""
    from oracle.c_oracle import CEnvBatch, lib
    L = lib()
    # host cores this process may use (the GPU box gives one GPU's share of a big host)
    threads = max(1, min(len(os.sched_getaffinity(0)), L.tor_num_threads(), int(os.environ.get("TORIC_CPU_THREADS", "16"))))

    def timed(n, nthreads, budget):
        L.tor_set_threads(nthreads)
        env = CEnvBatch(d, n, p, seed=seed)
        env.reset()
        env.actor_steps(2)                                    # page in, spin up the thread team
        t0 = time.perf_counter()
        env.actor_steps(4)
        probe = (time.perf_counter() - t0) / 4
        steps = int(max(4, min(2000, budget / max(probe, 1e-6))))
        t0 = time.perf_counter()
        P, _ = env.actor_steps(steps)
        dt = time.perf_counter() - t0
        return n * steps / dt, P / dt, steps, dt

    v, pps, steps, dt = timed(4096, threads, 0.6 * budget_s)
    out = {"value": v, "unit": "env-steps/s", "cores": int(threads), "kind": "port",
           "sample": f"4096 lattices x {steps} steps, d={d}, p={p}, C oracle actor loop (OpenMP, {threads} threads), {dt:.1f} s",
           "perspectives_per_sec": pps}
    v1, pps1, steps1, dt1 = timed(512, 1, 0.2 * budget_s)
    out["one_core"] = {"value": v1, "cores": 1, "perspectives_per_sec": pps1,
                       "sample": f"512 lattices x {steps1} steps, 1 thread, {dt1:.1f} s"}
    # the same C-ABI on host memory (oracle/host_twin.cpp over the product's csrc/lattice.hpp: SURVEY 8b / 8d "C++ host
    # backend ... on 1 core and on all cores"): counts -> f32 stack + positions -> fused step with transition records
    try:
        from oracle import host_twin as H

        def twin_timed(n, nthreads, budget):
            L.tor_set_threads(nthreads)                           # one OpenMP runtime serves both libraries
            tw = H.HostEnvSet(d, n, p_error=p, seed=seed)
            tw.reset_all()
            nq = 2 * d * d
            cap = n * nq
            stack, pos = np.empty((cap, 2, d, d), np.float32), np.empty((cap, 3), np.int32)
            blk, bc = tw.new_block(steps=8)

            def step(t):
                off = tw.perspectives(out=stack, positions=pos, capacity=cap)[3]
                tw.actor_step(None, block=blk, block_cap=bc, slot=t % 8, want_actions=False)
                return int(off[-1])
            for t in range(2):
                step(t)
            t0 = time.perf_counter()
            step(2)
            probe = time.perf_counter() - t0
            steps = int(max(3, min(2000, budget / max(probe, 1e-6))))
            t0 = time.perf_counter()
            P = sum(step(t) for t in range(steps))
            dt = time.perf_counter() - t0
            tw.check()
            tw.close()
            return n * steps / dt, P / dt, steps, dt

        vt, ppt, st_, dtt = twin_timed(4096, threads, 0.15 * budget_s)
        v1t, pp1t, s1t, dt1t = twin_timed(512, 1, 0.1 * budget_s)
        out["host_twin"] = {"value": vt, "unit": "env-steps/s", "cores": int(threads), "kind": "port", "perspectives_per_sec": ppt,
                            "sample": f"4096 lattices x {st_} steps, d={d}, p={p}: the C-ABI's hot path on host memory "
                                      f"(oracle/host_twin.cpp, bit-plane algebra of csrc/lattice.hpp, OpenMP, {threads} threads), {dtt:.1f} s",
                            "one_core": {"value": v1t, "cores": 1, "perspectives_per_sec": pp1t,
                                         "sample": f"512 lattices x {s1t} steps, 1 thread, {dt1t:.1f} s"}}
        out["c_oracle"] = {"value": out["value"], "cores": out["cores"], "sample": out["sample"],
                           "perspectives_per_sec": out["perspectives_per_sec"], "one_core": dict(out["one_core"])}
        if vt > out["value"]:                                     # the headline CPU number is the faster of the two ports
            out.update({"value": vt, "sample": out["host_twin"]["sample"], "perspectives_per_sec": ppt, "which": "host_twin"})
        else:
            out["which"] = "c_oracle"
        if v1t > out["one_core"]["value"]:
            out["one_core"] = dict(out["host_twin"]["one_core"], which="host_twin")
    except Exception as e:                                        # the twin is an extra; the baseline above stands without it
        out["host_twin"] = {"error": repr(e)}
    L.tor_set_threads(threads)
    from oracle import toric_oracle as O
    n_ref, s_ref = (256, 20) if budget_s >= 5 else (32, 2)    # full SURVEY sample only with a real budget
    oe = O.OracleEnvSet(d, n_ref, p, seed=seed)
    oe.resetAll()
    t0 = time.perf_counter()
    O.run_actor_steps_ref(oe, s_ref, eps=1.0)
    dtr = time.perf_counter() - t0
    out["numpy_reference_shaped"] = {"value": n_ref * s_ref / dtr, "cores": 1,
                                     "sample": f"{n_ref} lattices x {s_ref} steps, per-lattice python + np.roll/np.rot90 "
                                               f"(the reference's algorithmic form), {dtr:.1f} s"}
    out["numpy_reference_shaped_steps_per_sec"] = n_ref * s_ref / dtr
    return out


class Shard:
    """One sub-shard of this GPU's lattices with its stream and its caller-owned output buffers (legacy serial path:
    --shards > 1, --graph, --policy nn11)."""


def hbm_roofline(alg_bytes, write_ms, extra=None, first_of=None):
    """The `roofline` object of one leg: algorithmic bytes per launch over the stack write's average duration
    from HIP events on the stream the kernel runs on.  ``first_of`` = K: write_ms[0] is the FIRST launch of a timed
    region of K launches (the first kernel on its stream after the barrier: a few per cent slower) and the others a
    regular sample of the remaining K - 1; the average over all K launches is estimated with those weights -- the first
    launch counts 1 / K, not 1 / (number of samples)."""
    write_ms = np.ravel(np.asarray(write_ms, dtype=np.float64))
    if first_of is not None and write_ms.size > 1 and first_of > 1:
        ms = float((write_ms[0] + (first_of - 1) * write_ms[1:].mean()) / first_of)
    else:
        ms = float(write_ms.mean())
    ach = alg_bytes / (ms * 1e-3) / 1e9
    r = {"bound": "hbm", "kernel": STACK_KERNEL, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
         "bytes_per_launch": alg_bytes, "avg_launch_ms": ms, "median_launch_ms": float(np.median(write_ms)), "launches_timed": int(write_ms.size),
         "first_launches_ms": [round(float(x), 4) for x in write_ms[:(100000 if os.environ.get("TORIC_BENCH_SERIES") == "1" else 8)]]}
    if first_of is not None:
        r["first_launch_ms"] = float(write_ms[0])
    if extra:
        r.update(extra)
    return r


class ExploreLeg:
    """One timed leg of the eps = 1 actor loop on `n` lattices of this GPU through T.ExploreLoop: lattices, burn-in
    (episodes staggered: lattice e is reset at step e mod 76), the caller-owned buffers, the stack-buffer probe (in
    the loop, all candidates allocated first), warm-up with a check of the probe's prediction (one bounded re-probe),
    the timed steps, the byte-for-byte verification of the timed buffer and the leg's own `roofline`."""

    def __init__(self, T, torch, env, n, d, seed, first, tdtype, flush, device, rows, chunks=1, overlap=True, transitions=True,
                 burn_in=True, on_flush=None):
        self.T, self.torch, self.n, self.d, self.device, self.tdtype, self.chunks, self.flush = T, torch, n, d, device, tdtype, chunks, flush
        self.nq = 2 * d * d
        self.esize = torch.empty((), dtype=tdtype).element_size()
        self.envs = T.EnvSet(env, n, device=device, seed=seed, first_env_id=first, numpy_io=False)
        self.envs.resetAll()
        self.cap = (n // chunks) * self.nq      # worst case every qubit is a hit: 2.5 GB at d=7 f32, 6.9 GB at d=9 for 65 536 lattices
        self.stack = torch.empty((self.cap, 2, d, d), dtype=tdtype, device=device)
        self.positions = torch.empty((self.cap, 3), dtype=torch.int32, device=device)
        self.offs = torch.zeros((rows, (n + 2) & ~1), dtype=torch.int64, device=device)     # one scan per step: P = row[n]
        self.blocks = [self.envs.newTransitionBlock(steps=flush) for _ in range(2)] if transitions else None
        if burn_in:
            for t in range(EPISODE):
                idx = torch.arange(t, n, EPISODE, dtype=torch.int32, device=device)
                if idx.numel():
                    self.envs.resetTerminalEnvs(idx)
                self.envs.actorStep(None, want_actions=False)
        torch.cuda.synchronize(device)
        self.loop = T.ExploreLoop(self.envs, self.stack, self.positions, self.offs, blocks=self.blocks, flush=flush, chunks=chunks,
                                  overlap=overlap, on_flush=on_flush)
        self.probe = None
        self.ev, self.t0, self.K, self.every = [], 0, 0, 1

    def pick_stack(self, candidates, kinds):
        """Set-up, untimed: the stack buffer is re-used every step, so its placement is chosen by timing the write, inside
        the loop, on every candidate (EnvSet.pickStackBuffer with ExploreLoop.time_writes)."""
        torch = self.torch
        if candidates <= 1 and kinds[0] == "torch":
            return
        del self.stack
        self.loop.stack = None
        torch.cuda.empty_cache()
        time.sleep(0.3)                         # the driver wipes freed memory in the background
        self.stack, self.probe = self.envs.pickStackBuffer(candidates, dtype=self.tdtype, capacity=self.cap, positions=self.positions, kinds=kinds,
                                                           park=True, timer=self.loop.time_writes, launches=10, passes=2)
        self.loop.stack = self.stack
        # The probe ran in bursts (6 steps, host synchronisation, next candidate).  Set-up ends with SETTLE passes of the loop in
        # one go: for ~25 steps after such a phase the same write takes 2-3 % longer (a hump that decays by itself,
        # profiles/r04_first_steps_after_setup.txt) -- it belongs to the change of regime, not to the W warm-up steps or the
        # timed region that follow.
        for _ in range(SETTLE):
            self.loop.step()
        # ... and the probe's figure for the kept buffer is taken again HERE, in the loop's own regime: the candidates were
        # timed in bursts between allocations and other candidates' writes, 2-4 % slower than the same write a moment later
        self.probe["probe_ms_in_bursts"] = self.probe["probe_ms_chosen"]
        self.probe["probe_ms_chosen"] = float(np.median(self.loop.time_writes(self.stack, 8, skip=1)))
        self.loop.drain()
        torch.cuda.synchronize(self.device)
        self.probe["settle_steps"] = SETTLE
        self.probe["note"] = ("set-up, untimed (EnvSet.pickStackBuffer): every candidate allocated first, then the stack write timed INSIDE the "
                              "loop (HIP events around the write, the env kernels beside it) on each candidate in two passes of 5 writes; the "
                              "median decides; candidate 0 is the allocation a caller gets by default (torch.empty), the others T.alloc_stack "
                              "= tq_stack_alloc (2 MiB physical chunks); rejected candidates stay parked until the timed region is over")

    def warm(self, W, reprobe=True):
        """W untimed steps; the last ones' writes are timed and held against the probe: more than 8 % slower -> ONE
        re-probe among the candidates that are still parked, a few more untimed steps."""
        torch = self.torch
        k = min(max(W - 1, 1), 8)              # the first write after the probe (another buffer was written last) is not held against it
        for _ in range(max(0, W - k - 1)):
            self.loop.step()
        ms = self.loop.time_writes(self.stack, k, skip=1 if W > 1 else 0)
        out = {"warm_write_ms": float(np.mean(ms))}
        if self.probe is not None and reprobe:
            p = self.probe["probe_ms_chosen"]
            out["warm_over_probe"] = out["warm_write_ms"] / p
            parked = list(getattr(self.envs, "_parked", []))
            if out["warm_over_probe"] > 1.08 and parked:
                self.stack, rep = self.envs.pickStackBuffer(among=[self.stack] + parked, dtype=self.tdtype, capacity=self.cap, positions=self.positions,
                                                            park=True, timer=self.loop.time_writes, launches=10, passes=2)
                self.loop.stack = self.stack
                self.probe["reprobe"] = {"why": "the warm-up's writes ran %.1f %% slower than the probe predicted" % (100 * (out["warm_over_probe"] - 1)),
                                         "write_ms": rep["write_ms"], "chosen": rep["chosen"], "probe_ms_chosen": rep["probe_ms_chosen"]}
                self.probe["probe_ms_chosen"] = rep["probe_ms_chosen"]
                ms = self.loop.time_writes(self.stack, 4, skip=1)
                out["warm_write_ms_after_reprobe"] = float(np.mean(ms))
        # Nothing else may be keeping the GPU's front end busy when the timed region starts: right after device memory is freed
        # the driver wipes it, and for that long (45 ms for 120 GB, tools/free_aftermath.py) kernels are dispatched late -- the
        # writes keep their rate, the gaps between them grow.  One flush window at a time is held against the probe's figure
        # until it passes (or 2 s are over); what was seen is reported.
        if self.probe is not None:
            quiet = {"windows": 0, "waited_ms": 0.0}
            for _ in range(40):
                self.loop.drain()
                torch.cuda.synchronize(self.device)
                t0 = time.perf_counter()
                for _ in range(self.flush):
                    self.loop.step()
                self.loop.drain()
                torch.cuda.synchronize(self.device)
                per_step = 1e3 * (time.perf_counter() - t0) / self.flush
                quiet["windows"] += 1
                quiet["last_window_ms_per_step"] = per_step
                if per_step <= 1.15 * self.probe["probe_ms_chosen"] * self.chunks + 0.03:
                    break
                time.sleep(0.05)
                quiet["waited_ms"] += 50.0
            out["quiet_check"] = quiet
        while self.loop.t % self.flush:         # the timed region starts on a block boundary: every rank flushes (and gathers) at the
            self.loop.step()                    # same steps of it, however many passes its probe and its re-probe took
        self.loop.drain()
        torch.cuda.synchronize(self.device)
        return out

    def prepare(self, K, every):
        """Create the timed region's events BEFORE it starts (torch creates the HIP event at the first record: that call
        belongs to set-up, not between two launches of the timed region)."""
        torch = self.torch
        self.K, self.every = K, max(1, min(every, K))
        assert self.offs.shape[0] > K + 1, "one offsets row per timed step"
        # which launches are bracketed: the first one, and the middle launch of every further group of `every`
        self.sampled = list(range(K)) if self.every == 1 else [0] + [i for i in range(1, K) if i % self.every == self.every // 2]
        self.ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in self.sampled]
        for a, b in self.ev:
            a.record(self.loop.A)
            b.record(self.loop.A)
        torch.cuda.synchronize(self.device)

    def run(self, K, every):
        """K steps; HIP events around every `every`-th stack write.  No synchronisation (the caller brackets)."""
        if self.K != K or self.every != max(1, min(every, K)) or not self.ev:
            self.prepare(K, every)
        self.t0 = self.loop.t
        at = {i: j for j, i in enumerate(self.sampled)}
        for i in range(K):
            self.loop.step(self.ev[at[i]] if i in at else None)
        self.loop.drain()

    def perspectives(self):
        """(P of every timed step, f64 tensor; mean P over ALL timed steps -- the scan's row of every step is kept)."""
        R = self.offs.shape[0]
        rows = self.torch.tensor([(self.t0 + i) % R for i in range(self.K)], device=self.device)
        p = self.offs[rows, self.n].to(self.torch.float64)
        return p, float(p.mean().item())

    def write_ms(self):
        return np.array([a.elapsed_time(b) for a, b in self.ev])

    def roofline(self, extra=None):
        _, p_mean = self.perspectives()
        alg = p_mean * (self.nq * self.esize + 12) + self.n * self.nq       # P*(B_p+12) + N*2d^2, per step
        ms = self.write_ms()
        r = hbm_roofline(alg, ms, {"perspectives_per_launch": p_mean / self.chunks, "launches_per_step": self.chunks,
                                   "lattices_per_launch": self.n // self.chunks, "bytes_per_step": alg,
                                   "timed_launches": "HIP events on the write's stream around the stack write(s) of the first timed step and of the middle "
                                                     "step of every further group of %d (%d of %d steps); avg_launch_ms estimates the mean over ALL %d: "
                                                     "(first + (K-1) * mean(others)) / K%s" % (
                                       self.every, len(self.sampled), self.K, self.K,
                                       "" if self.chunks == 1 else "; the %d range launches of a step and their gaps included" % self.chunks)},
                         first_of=self.K)
        if self.chunks > 1:
            r["bytes_per_launch"] = alg / self.chunks
        if self.probe is not None:
            r["probe_ms_chosen"] = self.probe["probe_ms_chosen"]
            r["timed_write_ms"] = r["avg_launch_ms"]
            r["timed_over_probe"] = r["avg_launch_ms"] / self.probe["probe_ms_chosen"]
            r["default_buffer"] = {"kind": self.probe["kinds"][0], "write_ms": self.probe["write_ms"][0],
                                   "frac": r["frac"] * r["avg_launch_ms"] / self.probe["write_ms"][0],
                                   "note": "candidate 0 of the probe: what a caller gets without pickStackBuffer (same loop, probe's clock)"}
        r["workgroup_shares"] = {"xcd_bias": int(self.envs._L.tq_env_get_xcd_bias(self.envs._h)) if self.envs.size >= 7 and self.tdtype != self.torch.uint8 else 0,
                                 "note": "of every pair of the write's workgroups the one on the even XCD takes 32 + bias, the other 32 - bias of "
                                         "the pair's 64 fine parts of the stack (include/toricenv.h: tq_set_xcd_bias; 0 = equal shares; d <= 5 and u8 stacks: always 0)"}
        if self.probe is not None and "xcd_bias" in self.probe:
            r["workgroup_shares"]["probe"] = self.probe["xcd_bias"]
        if extra:
            r.update(extra)
        return r

    def verify(self):
        """The buffer that was timed holds the right bytes: the stack of the current lattices written into it and into a
        fresh torch.empty buffer, compared byte for byte (untimed).  Not a formality: a buffer reached through stale
        address translations takes writes at 7 TB/s and is wrong in 70 % of its elements (profiles/r03_stack_write_ab.txt 12)."""
        torch, envs, n, d = self.torch, self.envs, self.n, self.d
        torch.cuda.synchronize(self.device)
        off_v = self.offs[0][:n + 1]
        envs.perspectiveCounts(off_v)
        first_v, count_v = (0, n) if self.chunks == 1 else ((self.chunks - 1) * (n // self.chunks), n // self.chunks)
        Pv = int((off_v[first_v + count_v] - off_v[first_v]).item())
        ref_s = torch.empty((Pv, 2, d, d), dtype=self.tdtype, device=self.device)
        ref_p = torch.empty((Pv, 3), dtype=torch.int32, device=self.device)
        envs.writePerspectives(ref_s, ref_p, off_v, first=first_v, count=count_v)
        self.stack.view(torch.uint8).fill_(0x5A)
        envs.writePerspectives(self.stack, self.positions, off_v, first=first_v, count=count_v)
        torch.cuda.synchronize(self.device)
        wrong = int((self.stack[:Pv].view(torch.uint8) != ref_s.view(torch.uint8)).sum().item()) + int((self.positions[:Pv] != ref_p).sum().item())
        return {"ok": wrong == 0, "wrong_bytes": wrong, "perspectives": Pv,
                "how": "after the timed region: stack + positions of the current lattices written into the timed buffer and into "
                       "a fresh torch.empty buffer, compared byte for byte"}

    def close(self):
        if self.loop is not None:
            self.loop.drain()
        self.torch.cuda.synchronize(self.device)       # nothing of the side stream may still run when the handle goes
        self.envs.check()
        self.envs.close()                       # frees the parked candidates of the probe as well ...
        self.loop = None
        self.stack = None
        self.torch.cuda.empty_cache()
        time.sleep(0.5)                         # ... and the driver wipes freed memory in the background


def time_explore_leg(T, torch, env, n, d, seed, tdtype, flush, device, steps, warm, chunks=1, candidates=1, kinds=("torch", "chunked"),
                     event_every=4, overlap=True):
    """An extra leg of the N=1 line on its own lattices, no collective: -> dict with value, ms_per_step, perspectives/s,
    the probe's report, `stack_verified` and the leg's own `roofline` (probe_ms_chosen vs timed_write_ms included)."""
    leg = ExploreLeg(T, torch, env, n, d, seed, 0, tdtype, flush, device, rows=steps + 4, chunks=chunks, overlap=overlap)
    leg.pick_stack(candidates, kinds)
    w = leg.warm(warm)
    leg.prepare(steps, event_every)
    gc.disable()                        # no collection inside the timed region: what it frees (tq_stack_free synchronises the device; the driver
                                        # wipes freed memory in the background) would land in it.  Not gc.collect() here either -- measured: the
                                        # wipe of what THAT frees slows the first timed writes by 5-38 %
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    leg.run(steps, event_every)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    gc.enable()
    p_all, _ = leg.perspectives()
    P = float(p_all.sum().item())
    out = {"lattices": n, "d": d, "steps": steps, "warmup": warm, "value": n * steps / dt, "unit": "env-steps/s", "ms_per_step": 1e3 * dt / steps,
           "perspectives_per_sec": P / dt, "perspectives_per_lattice": P / (steps * n), "stack_buffer_probe": leg.probe, "warm_up": w,
           "stack_verified": leg.verify(), "roofline": leg.roofline()}
    out["non_write_us_per_step"] = 1e3 * (out["ms_per_step"] - out["roofline"]["avg_launch_ms"])
    leg.close()
    return out


def load_trained_weights(d):
    """The reference's trained NN_11 state_dict for size d, committed as a data fixture (tests/golden/nn11_d*.safetensors,
    made by tests/golden/make_weights.py from network/converged/*.pt with weights_only=True), or None."""
    path = os.path.join(ROOT, "tests", "golden", "nn11_d%d_converged.safetensors" % d)
    if not os.path.exists(path):
        return None
    from safetensors.torch import load_file
    return load_file(path)


def config_name(world, n, d, p):
    """Which BASELINE.json config a (lattices per GPU, d, p_error) triple is -- by all three, not by n alone."""
    if d == 7 and abs(p - 0.10) < 1e-12:
        if n == ENVS_N1 and world == 1:
            return "configs[2]"
        if n == ENVS_MULTI:
            return "configs[4] shape"
    if d == 9 and abs(p - 0.15) < 1e-12 and n == ENVS_N1 and world == 1:
        return "configs[3]"
    if d == 5 and abs(p - 0.10) < 1e-12 and n == 4096 and world == 1:
        return "configs[1]"
    return "custom"


def pmc_traffic(d, out_dtype, n, p_error, launches_per_step, p_mean):
    """HBM bytes per launch from the committed counters of a PROFILED run of the same shape
    (profiles/pmc_latest.json: separate --pmc passes, WRITE_SIZE + 2*FETCH_SIZE, tools/pmc_profile.sh), scaled by
    this run's perspectives per launch.  None unless lattice size, dtype, lattices, p_error and launch shape all
    match: the counters are not collected in this process, so they are only quoted for the run they describe."""
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if launches_per_step != 1 or not os.path.exists(pmc):
        return None, None
    try:
        for ent in json.load(open(pmc)).get("entries", []):
            if (ent.get("d") == d and ent.get("out_dtype") == out_dtype and ent.get("envs") == n
                    and abs(ent.get("p_error", -1) - p_error) < 1e-12):
                return ent["hbm_bytes_per_perspective"] * p_mean, ent
    except Exception:
        pass
    return None, None


def dry_run(args, world, rank, result_out):
    """TORIC_BENCH_DRY_RUN=1: launcher / rendezvous check only (CPU test of the N>1 command form on a
    box without a GPU): init the process group, agree on the world size, print a line that says so.
    Nothing is measured and `value` is null."""
    import torch
    import torch.distributed as dist
    backend = os.environ.get("TORIC_DIST_BACKEND", "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend, rank=rank, world_size=world)
    seen = torch.ones(1)
    dist.all_reduce(seen)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "env steps/sec (batched) at d=%d p=%g" % (args.size, args.p_error), "value": None,
                          "unit": "env-steps/s", "n_gpus": dist.get_world_size(), "steps": args.steps,
                          "warmup": args.warmup, "dry_run": True, "ranks_seen": int(seen.item()),
                          "config": {"workload": "launcher dry run: no GPU work", "collective": backend}}),
              file=result_out, flush=True)
    dist.destroy_process_group()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args)
    # ONE JSON line on stdout, nothing else: native libraries print there too (RCCL writes its version
    # banner to stdout when the communicator comes up), so file descriptor 1 is pointed at stderr for
    # the rest of the process and the result line goes through a private copy of the real stdout.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if os.environ.get("TORIC_BENCH_DRY_RUN") == "1":
        return dry_run(args, world, rank, result_out)

    import torch
    import torch.distributed as dist
    import toric_rl_decoder_amd as T
    from toric_rl_decoder_amd import gather as G

    # Rehearsal on a one-GPU box: TORIC_DIST_BACKEND=gloo TORIC_SHARE_GPU=1 runs every rank on cuda:0
    # with host-staged collectives (RCCL refuses two ranks on one device).  Real runs use nccl = RCCL.
    backend = os.environ.get("TORIC_DIST_BACKEND", "nccl")
    if os.environ.get("TORIC_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    red_dev = device if backend == "nccl" else torch.device("cpu")      # where scalar reductions live
    # TORIC_FORCE_DIST=1: run the collective path even with one rank (exercises the RCCL calls on a one-GPU box)
    dist_on = world > 1 or os.environ.get("TORIC_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))       # a world of one rank rendezvouses with itself
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == world

    ctx = types.SimpleNamespace(args=args, world=world, rank=rank, device=device, red_dev=red_dev, dist_on=dist_on, backend=backend,
                                result_out=result_out)
    if max(1, args.shards) > 1 or args.graph or args.policy == "nn11":
        return run_serial(ctx)
    return run_explore(ctx)


def run_explore(ctx):
    """The default path: the eps = 1 actor loop through T.ExploreLoop (stack write on the caller's stream, fused step +
    scan beside it on a second stream), every leg with its own probe report, `roofline` and `stack_verified`."""
    import torch
    import torch.distributed as dist
    import toric_rl_decoder_amd as T
    from toric_rl_decoder_amd import gather as G
    from toric_rl_decoder_amd.policy import _forward_chunked
    args, world, rank, device, red_dev, dist_on, backend, result_out = (ctx.args, ctx.world, ctx.rank, ctx.device, ctx.red_dev, ctx.dist_on,
                                                                         ctx.backend, ctx.result_out)
    d, K, W = args.size, args.steps, args.warmup
    n = args.envs if args.envs is not None else (ENVS_N1 if world == 1 else ENVS_MULTI)
    nq = 2 * d * d
    tdtype = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16, "u8": torch.uint8}[args.out_dtype]
    esize = {"f32": 4, "f16": 2, "bf16": 2, "u8": 1}[args.out_dtype]
    flush = max(1, args.flush)
    CH = max(1, args.chunks)
    if n % CH:
        sys.exit("--envs must be divisible by --chunks")
    EV = max(1, min(args.event_every, K))
    # two streams pay off when a step is long against the host's launch calls and the ~13 us of env kernels that then run
    # beside the write: from ~0.25 GB of stack per step on (16 384 lattices of d=7 in f32; 65 536 lattices of d=3 write 64 MB
    # in 40 us and are faster on one stream: profiles/r04_overlap_cost.txt)
    overlap = not args.no_overlap and n * 0.75 * nq * nq * esize >= 256e6
    kinds = tuple(k for k in args.stack_kinds.split(",") if k in ("torch", "chunked")) or ("torch",)
    host_delivery = dist_on and not args.no_transitions and args.delivery in ("auto", "host") and backend == "nccl"
    roots = args.roots if args.roots > 0 else (2 if world >= 8 else 1)
    roots = max(1, min(roots, world))

    env = T.make("toric-code-v0", {"size": d, "min_qubit_errors": 0, "p_error": args.p_error})
    first, _ = G.shard_range(n * world, world, rank)
    state = {"tg": None}

    def on_flush(blk):                                            # on the loop's side stream, after the priorities
        if state["tg"] is not None:
            state["tg"].gather(blk.buf)

    leg = ExploreLeg(T, torch, env, n, d, args.seed, first, tdtype, flush, device, rows=max(K, 8) + 4, chunks=CH, overlap=overlap,
                     transitions=not args.no_transitions, burn_in=not args.no_burn_in, on_flush=on_flush)
    have_blocks = leg.blocks is not None

    def make_gather(host):
        if not (dist_on and have_blocks):
            return None
        return G.TransitionGather(leg.blocks[0].nbytes, device, ring_slots=2, host_drain=host, roots=roots)

    def barrier():
        torch.cuda.synchronize(device)
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(device)

    def warm_collective(g):     # RCCL sets its communicator up lazily: pay for that before anything is timed
        if g is not None:
            g.gather(leg.blocks[1].buf)
            g.wait()

    tg = make_gather(host_delivery)
    warm_collective(tg)
    barrier()
    leg.pick_stack(args.stack_candidates, kinds)                 # set-up, untimed; every rank probes its own GPU
    barrier()

    def timed_region():
        """W untimed + K timed steps, bracketed by barrier + synchronize; -> (seconds, max over ranks; warm-up report)."""
        # the warm-up (and a rank's own re-probe) takes a rank-dependent number of passes: no collective in it -- every
        # rank issues exactly the gathers of the K timed steps, K // flush of them, at the same steps
        gather_, state["tg"] = state["tg"], None
        w = leg.warm(W, reprobe=not args.no_events)
        leg.prepare(K, EV)
        state["tg"] = gather_
        gc.disable()                    # no collection (and no free of device memory) inside the timed region
        barrier()
        t0 = time.perf_counter()
        leg.run(K, EV)
        if state["tg"] is not None:
            with torch.cuda.stream(leg.loop.B):
                state["tg"].wait()
        barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
        gc.enable()
        if dist_on:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item()), w

    state["tg"] = tg
    elapsed, warm_report = timed_region()
    leg.envs.check()                                              # capacity / action / reset / offsets latch
    p_all, p_mean = leg.perspectives()
    p_sum = p_all.sum().reshape(1).to(red_dev)
    if dist_on:
        dist.all_reduce(p_sum, op=dist.ReduceOp.SUM)
    total_steps = float(n) * world * K
    roof = leg.roofline() if not args.no_events else None
    verified = leg.verify()

    # ---- every rank's own roofline (N>1): frac, write ms, probe's prediction, gathered to rank 0
    rank_rows = None
    if dist_on and roof is not None:
        mine = torch.tensor([roof["frac"], roof["avg_launch_ms"], roof.get("probe_ms_chosen", float("nan")), roof.get("timed_over_probe", float("nan")),
                             p_mean, 1.0 if verified["ok"] else 0.0], dtype=torch.float64, device=red_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        num = lambda v: float(v) if float(v) == float(v) else None          # NaN (no probe) -> null
        rank_rows = [{"rank": r, "frac": float(x[0]), "avg_launch_ms": float(x[1]), "probe_ms_chosen": num(x[2]), "timed_over_probe": num(x[3]),
                      "perspectives_per_launch": float(x[4]), "stack_verified": bool(x[5] > 0.5)} for r, x in enumerate(allr)]

    # ---- N>1 with host delivery: the same region again with the ring kept in rank 0's HBM
    hbm_ring = None
    if host_delivery:
        state["tg"] = make_gather(False)
        warm_collective(state["tg"])
        barrier()
        el2, _ = timed_region()
        hbm_ring = {"value": total_steps / el2, "ms_per_step": 1e3 * el2 / K,
                    "note": "same run, transition ring left in rank 0's HBM (no D2H drain)"}
    state["tg"] = None
    leg.envs.releaseParked()
    time.sleep(0.5)

    n1_legs = world == 1 and not dist_on and args.envs is None and not args.no_shard_leg and CH == 1
    # ---- N=1 on the default shape: one GPU on the per-GPU shape of the N>1 runs (configs[4]), like for like
    shard_leg = None
    if n1_legs:
        print("[bench] configs[4] shard leg (131072 lattices on this GPU) ...", file=sys.stderr, flush=True)
        shard_leg = time_explore_leg(T, torch, env, ENVS_MULTI, d, args.seed, tdtype, flush, device, max(8, min(K, 40)), 8,
                                     candidates=args.stack_candidates, kinds=kinds, event_every=args.event_every, overlap=overlap)
        shard_leg["envs_per_gpu"] = ENVS_MULTI
        shard_leg["note"] = ("this GPU alone on the per-GPU shape of the N>1 runs (BASELINE configs[4]: 131 072 lattices), no collective: "
                             "the like-for-like base of the scaling curve")

    # ---- N=1: the bf16 stack (TQ_BF16, what the bf16 nn_in_loop variant consumes), same lattices' shape, its own probe and roofline
    bf16_leg = None
    if n1_legs and args.bf16_steps > 0 and args.out_dtype == "f32":
        print("[bench] bf16-stack leg ...", file=sys.stderr, flush=True)
        bf16_leg = time_explore_leg(T, torch, env, n, d, args.seed, torch.bfloat16, flush, device, args.bf16_steps, 8,
                                    candidates=args.stack_candidates, kinds=kinds, event_every=args.event_every, overlap=overlap)
        bf16_leg["note"] = "the same pass with the stack written as bf16 by the kernel itself (numba/util_actor.py:39's cast absorbed)"

    # ---- N=1 on the default shape: BASELINE configs[3] (65 536 lattices, d=9, p=0.15) timed by this very run,
    # one shot and with the stack written in 4 lattice ranges into a buffer of a quarter of the size
    c3_leg = None
    if n1_legs and d == 7 and args.out_dtype == "f32":
        d3, p3, n3, k3, w3 = 9, 0.15, ENVS_N1, max(8, min(K, 20)), 5
        env3 = T.make("toric-code-v0", {"size": d3, "min_qubit_errors": 0, "p_error": p3})
        c3_leg = {"workload": "BASELINE configs[3]: %d lattices, d=%d, p_error=%g, f32 stack; same actor-loop pass" % (n3, d3, p3),
                  "steps": k3, "warmup": w3}
        for name, ch in (("one_shot", 1), ("chunks_4", 4)):
            print("[bench] configs[3] leg (65536 lattices, d=9, p=0.15), %s ..." % name, file=sys.stderr, flush=True)
            c3_leg[name] = time_explore_leg(T, torch, env3, n3, d3, args.seed, tdtype, flush, device, k3, w3, chunks=ch,
                                            candidates=args.stack_candidates, kinds=kinds, event_every=args.event_every, overlap=overlap)

    # ---- N=1: configs[2] as written -- generatePerspective feeding NN_11 for selectAction, measured at size in
    # f32 (what upstream runs) and with the bf16 stack the kernels can write directly + bf16 autocast
    nn_leg = None
    if world == 1 and args.nn_steps > 0 and CH == 1:
        from toric_rl_decoder_amd.policy import NN_11
        flop_per_persp = 2.0 * sum(ci * co * 9 * ((d - 2) ** 2 if i == 10 else d * d)
                                   for i, (ci, co) in enumerate(zip((2, 128, 128, 120, 111, 104, 103, 90, 80, 73, 71),
                                                                    (128, 128, 120, 111, 104, 103, 90, 80, 73, 71, 64))))
        weights = load_trained_weights(d)
        nn_leg = {"workload": "configs[2] as written: %d lattices, d=%d: stack -> NN_11 (%s, stock torch conv) -> device "
                              "eps=%g greedy selection -> fused step" %
                              (n, d, "the reference's trained d=%d weights" % d if weights is not None else "random init", args.eps),
                  "steps": args.nn_steps, "variants": {}}
        envs = leg.envs
        eps_t = torch.full((n,), args.eps, dtype=torch.float64, device=device)
        q_buf = torch.empty((leg.cap, 3), dtype=torch.float32, device=device)          # pre-sized: no torch.cat per step
        p_pin = torch.zeros(1, dtype=torch.int64).pin_memory()
        p_ready = torch.cuda.Event()
        off = leg.offs[0][:n + 1]
        p_nn = []

        def nn_step(model, stack, nn_dtype, t):
            envs.perspectiveCounts(off)
            p_pin.copy_(off[n:n + 1], non_blocking=True)          # 8 bytes, behind the scan only
            p_ready.record()
            envs.writePerspectives(stack, leg.positions, off)     # enqueued BEFORE the host waits: no bubble behind the read-back
            p_ready.synchronize()
            P = int(p_pin.item())
            p_nn.append(P)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=nn_dtype == "bf16"):
                # fixed-shape chunks (one MIOpen problem per layer): the last chunk runs into the buffer's slack, its surplus rows are cut off
                q = _forward_chunked(model, stack[:P], NN_CHUNK, pad_to=NN_CHUNK, backing=stack, out=q_buf)
            act, _ = envs.selectAction(q, eps_t, positions=leg.positions, offsets=off)
            blk = leg.blocks[(t // flush) & 1] if have_blocks else None
            envs.actorStep(act, block=blk, slot=t % flush, want_actions=True)
            if blk is not None and (t + 1) % flush == 0:
                blk.computePriorities(n, flush, None, 0.95)       # actor.run_actor passes the Q rows here

        for vname, nn_dtype in (("f32", "f32"), ("bf16", "bf16")):
            torch.manual_seed(0)
            m = NN_11(d, 3).to(device).eval()
            if weights is not None:
                m.load_state_dict(weights)
            stack = leg.stack if nn_dtype == "f32" and tdtype == torch.float32 else torch.empty(leg.stack.shape, dtype=torch.bfloat16 if nn_dtype == "bf16" else torch.float32, device=device)
            print("[bench] nn_in_loop %s: warm-up step (MIOpen picks its kernels) ..." % vname, file=sys.stderr, flush=True)
            t0 = time.perf_counter()
            nn_step(m, stack, nn_dtype, 0)
            torch.cuda.synchronize(device)
            print("[bench] nn_in_loop %s: warm-up took %.1f s; timing %d step(s) ..." % (vname, time.perf_counter() - t0, args.nn_steps),
                  file=sys.stderr, flush=True)
            del p_nn[:]
            t0 = time.perf_counter()
            for i in range(args.nn_steps):
                nn_step(m, stack, nn_dtype, 1 + i)
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
            P_nn = float(sum(p_nn))
            nn_leg["variants"][vname] = {"stack_dtype": "bf16" if nn_dtype == "bf16" else args.out_dtype, "nn_dtype": nn_dtype,
                                         "env_steps_per_sec": n * args.nn_steps / dt, "perspectives_per_sec_into_nn": P_nn / dt,
                                         "ms_per_step": 1e3 * dt / args.nn_steps, "nn_tflops": P_nn * flop_per_persp / dt / 1e12}
            del m, stack
            envs.check()
        for k_ in ("env_steps_per_sec", "perspectives_per_sec_into_nn", "ms_per_step", "nn_tflops", "nn_dtype"):
            nn_leg[k_] = nn_leg["variants"]["f32"][k_]                # top level = the f32 run, as upstream

    if rank == 0:
        cfg_name = config_name(world, n, d, args.p_error)
        collective = None
        if dist_on:
            collective = "transition gather (packed blocks incl. priorities) to %s (%s, %d ranks) every %d steps%s" % (
                "rank 0" if roots == 1 else "ranks 0..%d in turn" % (roots - 1), backend, dist.get_world_size(), flush,
                " + D2H drain of every gathered slot to the root's pinned host replay ring" if host_delivery else "; ring in the root's HBM")
        void = not verified["ok"]
        res = {
            "metric": "env steps/sec (batched) at d=%d p=%g" % (d, args.p_error),
            "value": None if void else total_steps / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "BASELINE %s: %d lattices/GPU, d=%d, p_error=%g; actor-loop pass = "
                                   "perspective stack (%s) + positions -> selection -> step -> transition "
                                   "record -> auto-reset (max 75 steps/episode), priorities every %d steps; "
                                   "policy NN excluded (eps=1 selection in the fused kernel)" %
                                   (cfg_name, n, d, args.p_error, args.out_dtype, flush),
                       "policy": args.policy, "envs_per_gpu": n, "d": d, "p_error": args.p_error,
                       "out_dtype": args.out_dtype, "transitions": have_blocks, "flush_steps": flush,
                       "streams_per_gpu": 2 if overlap else 1, "stack_chunks": CH, "hip_graph": False, "parallelism": "env-shard x%d" % world,
                       "loop": ("T.ExploreLoop: the stack write on one HIP stream, the fused step + next scan beside it on a second "
                                "(two plane buffers and two cut-point tables in the handle; two events per step, the host paces the write behind the scan)" if overlap else
                                "T.ExploreLoop(overlap=False): one stream, the reference's call order"),
                       "steady_state": not args.no_burn_in, "delivery": ("host" if host_delivery else "hbm") if dist_on else None,
                       "gather_roots": roots if dist_on else None,
                       "collective": collective},
            "perspectives_per_sec": float(p_sum.item()) / elapsed,
            "perspectives_per_lattice": float(p_sum.item()) / total_steps,
            "warm_up": warm_report,
        }
        if void:
            res["void"] = "the timed stack buffer holds wrong bytes (stack_verified): no value is reported"
        if leg.probe is not None:
            res["stack_buffer_probe"] = leg.probe
        if hbm_ring is not None:
            res["hbm_ring"] = hbm_ring
        res["stack_verified"] = verified
        if rank_rows is not None:
            res["ranks"] = rank_rows
        if shard_leg is not None:
            res["configs4_shard_on_one_gpu"] = shard_leg
        if bf16_leg is not None:
            res["bf16_stack"] = bf16_leg
        if c3_leg is not None:
            res["configs3_on_one_gpu"] = c3_leg
        if roof is not None:
            alg = roof["bytes_per_step"]
            traffic, ent = pmc_traffic(d, args.out_dtype, n, args.p_error, CH, p_mean)
            # context: the box's own streaming-fill bandwidth, measured after the timed region
            # on the same buffer (hipMemsetAsync through torch), so frac can be read against it as well
            fb = leg.stack.view(torch.uint8).reshape(-1)[:int(alg) & ~4095]
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fill_ms = []
            for _ in range(6):
                f0.record(); fb.zero_(); f1.record(); f1.synchronize()
                fill_ms.append(f0.elapsed_time(f1))
            fill_gbps = fb.numel() / (min(fill_ms[1:]) * 1e-3) / 1e9
            roof.update({"traffic": traffic, "traffic_over_algorithmic": None if traffic is None else traffic / alg,
                         "traffic_from_profile_run": None if ent is None else True,
                         "traffic_source": None if ent is None else "%s; counters of the profiled run %s, not of this process" % (
                             ent.get("source"), ent.get("profiled_at", "profiles/pmc_latest.json")),
                         "measured_fill_gbps": fill_gbps, "frac_of_measured_fill": roof["achieved"] / fill_gbps})
            res["roofline"] = roof
            res["non_write_us_per_step"] = 1e3 * (res["ms_per_step"] - roof["avg_launch_ms"])
        # every leg's timed write against what its probe promised: the legs outside 3 % are named
        checks = {"headline": roof}
        if shard_leg is not None:
            checks["configs4_shard_on_one_gpu"] = shard_leg["roofline"]
        if bf16_leg is not None:
            checks["bf16_stack"] = bf16_leg["roofline"]
        if c3_leg is not None:
            checks["configs3_on_one_gpu.one_shot"] = c3_leg["one_shot"]["roofline"]
            checks["configs3_on_one_gpu.chunks_4"] = c3_leg["chunks_4"]["roofline"]
        res["probe_vs_timed"] = {k_: v["timed_over_probe"] for k_, v in checks.items() if v is not None and "timed_over_probe" in v}
        res["legs_outside_3pct_of_probe"] = sorted(k_ for k_, v in res["probe_vs_timed"].items() if abs(v - 1.0) > 0.03)
        bad_legs = [k_ for k_, lg in (("configs4_shard_on_one_gpu", shard_leg), ("bf16_stack", bf16_leg),
                                       ("configs3.one_shot", c3_leg and c3_leg["one_shot"]), ("configs3.chunks_4", c3_leg and c3_leg["chunks_4"]))
                    if lg is not None and not lg["stack_verified"]["ok"]]
        if bad_legs:
            res["void_legs"] = bad_legs
        if nn_leg is not None:
            res["nn_in_loop"] = nn_leg
        if world == 1 and args.cpu_seconds > 0:
            res["cpu_baseline"] = cpu_baseline(d, args.p_error, args.seed, args.cpu_seconds)
        print(json.dumps(res), file=result_out, flush=True)
        if void:
            print("[bench] THE TIMED STACK BUFFER HOLDS WRONG BYTES (%d): the numbers of this run are void" % verified["wrong_bytes"], file=sys.stderr, flush=True)
    leg.close()
    if dist_on:
        dist.destroy_process_group()
    if not verified["ok"]:
        sys.exit(3)


def run_serial(ctx):
    """The serial loop on one stream in the reference's call order (perspective counts -> stack write -> selection ->
    step ...): --shards > 1 (independent sub-shards on their own streams), --graph (a flush window captured into a HIP
    graph) and --policy nn11 (NN_11 in the main loop).  The default path is run_explore."""
    import torch
    import torch.distributed as dist
    import toric_rl_decoder_amd as T
    from toric_rl_decoder_amd import gather as G
    args, world, rank, device, red_dev, dist_on, backend, result_out = (ctx.args, ctx.world, ctx.rank, ctx.device, ctx.red_dev, ctx.dist_on,
                                                                         ctx.backend, ctx.result_out)
    d, K, W = args.size, args.steps, args.warmup
    n = args.envs if args.envs is not None else (ENVS_N1 if world == 1 else ENVS_MULTI)
    nq = 2 * d * d
    tdtype = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16, "u8": torch.uint8}[args.out_dtype]
    esize = {"f32": 4, "f16": 2, "bf16": 2, "u8": 1}[args.out_dtype]
    S = max(1, args.shards)
    if n % S:
        sys.exit("--envs must be divisible by --shards")
    ns = n // S
    flush = max(1, args.flush)
    use_events = not args.no_events and not args.graph
    EV = max(1, min(args.event_every, K))
    if args.policy == "nn11" and args.chunks > 1:
        sys.exit("--policy nn11 reads the whole stack: use --chunks 1")
    if args.graph and (args.policy != "explore" or world > 1 or args.shards != 1):
        sys.exit("--graph supports the single-GPU, single-stream explore policy only")
    if args.graph:
        K = max(flush, K - K % flush)                                  # whole replays
        W = max(flush, W - W % flush)
    host_delivery = dist_on and not args.no_transitions and args.delivery in ("auto", "host") and backend == "nccl"
    roots = args.roots if args.roots > 0 else (2 if world >= 8 else 1)
    roots = max(1, min(roots, world))

    def make_model():
        from toric_rl_decoder_amd.policy import NN_11
        torch.manual_seed(0)                                          # random-init weights of the NN_11 architecture
        return NN_11(d, 3).to(device).eval()

    model = make_model() if args.policy == "nn11" else None

    env = T.make("toric-code-v0", {"size": d, "min_qubit_errors": 0, "p_error": args.p_error})
    first, _ = G.shard_range(n * world, world, rank)
    # worst case every qubit is a hit (exploration grows the defect density): size each stack for
    # that -- 2.5 GB at d=7 f32, 6.9 GB at d=9 for 65 536 lattices -- out of 288 GB of HBM
    CH = max(1, args.chunks)
    if ns % CH:
        sys.exit("--envs / --shards must be divisible by --chunks")
    cap = (ns // CH) * nq
    row = (ns + 2) & ~1                 # offsets rows of even length: every row starts 16-byte aligned (toricenv.h)
    shards = []
    for k in range(S):
        sh = Shard()
        sh.stream = torch.cuda.Stream(device=device) if S > 1 else torch.cuda.current_stream(device)
        with torch.cuda.stream(sh.stream):
            sh.envs = T.EnvSet(env, ns, device=device, seed=args.seed, first_env_id=first + k * ns, numpy_io=False)
            sh.envs.resetAll()
            sh.stack = torch.empty((cap, 2, d, d), dtype=tdtype, device=device)
            sh.positions = torch.empty((cap, 3), dtype=torch.int32, device=device)
            sh.offs = torch.zeros((2 * (W + K) + 2 * args.nn_steps + 4, row), dtype=torch.int64, device=device)  # one scan per step: P = row[ns]
            sh.blocks = None if args.no_transitions else [sh.envs.newTransitionBlock(steps=flush) for _ in range(2)]
            sh.ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                     for _ in range((K + EV - 1) // EV)] if use_events else []
            sh.wrote = torch.cuda.Event()
            sh.eps = torch.full((ns,), args.eps, dtype=torch.float64, device=device)
            sh.burn_idx = [torch.arange(t, ns, EPISODE, dtype=torch.int32, device=device) for t in range(EPISODE)]
        shards.append(sh)
    torch.cuda.synchronize(device)
    have_blocks = shards[0].blocks is not None

    def make_gathers(host):
        if not (dist_on and have_blocks):
            return None
        return [G.TransitionGather(sh.blocks[0].nbytes, device, ring_slots=2, host_drain=host, roots=roots) for sh in shards]

    tg = make_gathers(host_delivery)
    state = {"tg": tg, "model": model}

    def one_step(k, t, timed_idx=None):
        sh = shards[k]
        tg_, model_ = state["tg"], state["model"]
        # one shard: stay on torch's current stream (inside torch.cuda.graph() that is the capture stream)
        with (torch.cuda.stream(sh.stream) if S > 1 else contextlib.nullcontext()):
            envs, off = sh.envs, sh.offs[t][:ns + 1]
            envs.perspectiveCounts(off)
            if S > 1:
                sh.stream.wait_event(shards[(k - 1) % S].wrote)          # one stack write at a time
            bracket = timed_idx is not None and use_events and timed_idx % EV == 0
            if bracket:
                sh.ev[timed_idx // EV][0].record(sh.stream)
            if CH == 1:
                envs.writePerspectives(sh.stack, sh.positions, off)
            else:                                                     # the consumer would read the buffer between two chunks
                for c in range(CH):
                    envs.writePerspectives(sh.stack, sh.positions, off, first=c * (ns // CH), count=ns // CH)
            if bracket:
                sh.ev[timed_idx // EV][1].record(sh.stream)
            if S > 1:
                sh.wrote.record(sh.stream)
            blk = sh.blocks[(t // flush) & 1] if have_blocks else None
            act = None
            if model_ is not None:                                    # configs[2] as written: stack -> NN_11 -> selection
                P = int(off[-1].item())
                # fixed-shape chunks (one MIOpen problem per layer): the rows past P in the last chunk are
                # stale stack rows whose Q-values are cut off again
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.nn_dtype == "bf16"):
                    q = torch.cat([model_(sh.stack[i:i + NN_CHUNK]) for i in range(0, P, NN_CHUNK)])[:P].float()
                act, _ = envs.selectAction(q, sh.eps, positions=sh.positions, offsets=off)
            envs.actorStep(act, block=blk, slot=t % flush, want_actions=True)
            if blk is not None and (t + 1) % flush == 0:
                # computePrioritiesParallel into the block (eps = 1: no Q-values, priority = |reward|);
                # with the NN in the loop the Q rows would be passed here (actor.run_actor does)
                blk.computePriorities(ns, flush, None, 0.95)
                if tg_ is not None:
                    tg_[k].gather(blk.buf)

    def barrier():
        torch.cuda.synchronize(device)
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(device)

    def warm_collective(gs):     # RCCL sets its communicator up lazily: pay for that before anything is timed
        if gs is not None:
            for k, g in enumerate(gs):
                g.gather(shards[k].blocks[1].buf)
                g.wait()

    warm_collective(tg)
    barrier()

    # ---- burn-in: stagger the episodes so the population is stationary (not part of warm-up or timing)
    if not args.no_burn_in:
        for t in range(EPISODE):
            for sh in shards:
                with (torch.cuda.stream(sh.stream) if S > 1 else contextlib.nullcontext()):
                    if sh.burn_idx[t].numel():
                        sh.envs.resetTerminalEnvs(sh.burn_idx[t])
                    sh.envs.actorStep(None, want_actions=False)
        barrier()

    # ---- placement probe (set-up, untimed): the same stack write on every candidate buffer, keep the fastest
    probe = None
    kinds = tuple(k for k in args.stack_kinds.split(",") if k in ("torch", "chunked")) or ("torch",)
    if (args.stack_candidates > 1 or kinds[0] != "torch") and S == 1 and CH == 1 and not args.graph:
        sh0 = shards[0]
        del sh0.stack
        torch.cuda.empty_cache()
        # rejected candidates stay parked until the timed region is over: the driver wipes freed memory in the background
        sh0.stack, probe = sh0.envs.pickStackBuffer(args.stack_candidates, dtype=tdtype, capacity=cap, positions=sh0.positions, kinds=kinds,
                                                    park=True, launches=6)
        probe["note"] = ("set-up, untimed (EnvSet.pickStackBuffer): 3 stack writes timed on each candidate buffer, the fastest kept, "
                         "the others freed; candidate 0 is the allocation a caller gets by default (torch.empty), 'chunked' is "
                         "T.alloc_stack = tq_stack_alloc (2 MiB physical chunks)")

    graph = None
    if args.graph:
        # the capture stream is torch's current stream inside torch.cuda.graph(); every ABI call
        # enqueues on it, no call allocates or synchronises, so the whole flush window is capturable
        sh0 = shards[0]
        p_acc = torch.zeros(flush, dtype=torch.int64, device=device)
        for t in range(flush):                                        # eager once (lazy init, LUT)
            one_step(0, t)
        torch.cuda.synchronize(device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for t in range(flush):
                one_step(0, t)
                p_acc[t:t + 1].add_(sh0.offs[t, ns:ns + 1])           # running sum of P per slot of the window
        p_acc.zero_()

    def run_steps(first_t, count, timed):
        if graph is not None:
            for _ in range(count // flush):
                graph.replay()
            return
        for i in range(count):
            for k in range(S):
                one_step(k, first_t + i, i if timed else None)

    def timed_region(first_t):
        """W untimed + K timed steps, bracketed by barrier + synchronize; -> seconds (max over ranks)."""
        run_steps(first_t, W, False)
        barrier()
        if graph is not None:
            p_acc.zero_()
            torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        run_steps(first_t + W, K, True)
        if state["tg"] is not None:
            for k, g in enumerate(state["tg"]):
                with torch.cuda.stream(shards[k].stream):
                    g.wait()
        barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
        if dist_on:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item())

    elapsed = timed_region(0)
    for sh in shards:
        sh.envs.check()                                               # capacity / action / reset latch
        sh.envs.releaseParked()
    time.sleep(0.5)

    if graph is not None:      # slot t of the captured window accumulated its P over the K/flush replays
        p_timed = (p_acc.to(torch.float64) / (K // flush)).repeat(K // flush).reshape(1, -1)
    else:
        p_timed = torch.stack([sh.offs[W:W + K, ns] for sh in shards]).to(torch.float64)   # (S, K) perspectives per launch
    p_sum = p_timed.sum().reshape(1).to(red_dev)
    if dist_on:
        dist.all_reduce(p_sum, op=dist.ReduceOp.SUM)
    total_steps = float(n) * world * K
    ev_ms = None
    if use_events:
        ev_ms = np.array([[a.elapsed_time(b) for a, b in sh.ev] for sh in shards])         # (S, K) launch durations

    # ---- the buffer that was timed holds the right bytes: the stack of the current lattices written into it and into a
    # fresh torch.empty buffer, compared element for element (untimed).  Not a formality: a buffer reached through stale
    # address translations takes writes at 7 TB/s and is wrong in 70 % of its elements (profiles/r03_stack_write_ab.txt 12).
    verified = None
    if S == 1 and CH == 1 and graph is None:
        sh0 = shards[0]
        off_v = sh0.offs[-1][:ns + 1]
        sh0.envs.perspectiveCounts(off_v)
        Pv = int(off_v[-1].item())
        ref_s = torch.empty((Pv, 2, d, d), dtype=tdtype, device=device)
        ref_p = torch.empty((Pv, 3), dtype=torch.int32, device=device)
        sh0.envs.writePerspectives(ref_s, ref_p, off_v)
        sh0.stack.view(torch.uint8).fill_(0x5A)
        sh0.envs.writePerspectives(sh0.stack, sh0.positions, off_v)
        torch.cuda.synchronize(device)
        wrong = int((sh0.stack[:Pv].view(torch.uint8) != ref_s.view(torch.uint8)).sum().item()) + int((sh0.positions[:Pv] != ref_p).sum().item())
        verified = {"ok": wrong == 0, "wrong_bytes": wrong, "perspectives": Pv,
                    "how": "after the timed region: stack + positions of the current lattices written into the timed buffer and into "
                           "a fresh torch.empty buffer, compared byte for byte"}
        del ref_s, ref_p
        if wrong:
            print("[bench] THE TIMED STACK BUFFER HOLDS WRONG BYTES (%d): the numbers of this run are void" % wrong, file=sys.stderr, flush=True)

    # ---- N>1 with host delivery: the same region again with the ring kept in rank 0's HBM
    hbm_ring = None
    if host_delivery and graph is None:
        state["tg"] = make_gathers(False)
        warm_collective(state["tg"])
        barrier()
        el2 = timed_region(W + K)
        hbm_ring = {"value": total_steps / el2, "ms_per_step": 1e3 * el2 / K,
                    "note": "same run, transition ring left in rank 0's HBM (no D2H drain)"}

    shard_leg = c3_leg = nn_leg = None

    if rank == 0:
        policy_txt = "policy NN excluded (eps=1 selection in the fused kernel)" if model is None else \
            "NN_11 (random init, %s) forward + eps=%g greedy selection IN the loop" % (args.nn_dtype, args.eps)
        cfg_name = config_name(world, n, d, args.p_error)
        collective = None
        if dist_on:
            collective = "transition gather (packed blocks incl. priorities) to %s (%s, %d ranks) every %d steps%s" % (
                "rank 0" if roots == 1 else "ranks 0..%d in turn" % (roots - 1), backend, dist.get_world_size(), flush,
                " + D2H drain of every gathered slot to the root's pinned host replay ring" if host_delivery else "; ring in the root's HBM")
        res = {
            "metric": "env steps/sec (batched) at d=%d p=%g" % (d, args.p_error),
            "value": total_steps / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "BASELINE %s: %d lattices/GPU, d=%d, p_error=%g; actor-loop pass = "
                                   "perspective stack (%s) + positions -> selection -> step -> transition "
                                   "record -> auto-reset (max 75 steps/episode), priorities every %d steps; %s" %
                                   (cfg_name, n, d, args.p_error, args.out_dtype, flush, policy_txt),
                       "policy": args.policy, "envs_per_gpu": n, "d": d, "p_error": args.p_error,
                       "out_dtype": args.out_dtype, "transitions": have_blocks, "flush_steps": flush,
                       "streams_per_gpu": S, "stack_chunks": CH, "hip_graph": bool(args.graph), "parallelism": "env-shard x%d" % world,
                       "steady_state": not args.no_burn_in, "delivery": ("host" if host_delivery else "hbm") if dist_on else None,
                       "gather_roots": roots if dist_on else None,
                       "collective": collective},
            "perspectives_per_sec": float(p_sum.item()) / elapsed,
            "perspectives_per_lattice": float(p_sum.item()) / total_steps,
        }
        if probe is not None:
            res["stack_buffer_probe"] = probe
        if hbm_ring is not None:
            res["hbm_ring"] = hbm_ring
        if verified is not None:
            res["stack_verified"] = verified
        if shard_leg is not None:
            res["configs4_shard_on_one_gpu"] = shard_leg
        if c3_leg is not None:
            res["configs3_on_one_gpu"] = c3_leg
        if use_events:
            p_mean = float(p_timed[:, ::EV].mean().item())              # of the launches the events bracket
            alg = p_mean * (nq * esize + 12) + ns * nq                 # P*(B_p+12) + N*2d^2, per launch
            achieved = alg / (ev_ms.mean() * 1e-3) / 1e9
            traffic, ent = pmc_traffic(d, args.out_dtype, n, args.p_error, S * CH, p_mean)
            # context: the box's own streaming-fill bandwidth, measured after the timed region
            # on the same buffer (hipMemsetAsync through torch), so frac can be read against it as well
            fb = shards[0].stack.view(torch.uint8).reshape(-1)[:int(alg) & ~4095]
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fill_ms = []
            for _ in range(6):
                f0.record(); fb.zero_(); f1.record(); f1.synchronize()
                fill_ms.append(f0.elapsed_time(f1))
            fill_gbps = fb.numel() / (min(fill_ms[1:]) * 1e-3) / 1e9
            res["roofline"] = {"bound": "hbm", "kernel": STACK_KERNEL, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                               "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                               "traffic_over_algorithmic": None if traffic is None else traffic / alg,
                               "traffic_from_profile_run": None if ent is None else True,
                               "traffic_source": None if ent is None else "%s; counters of the profiled run %s, not of this process" % (
                                   ent.get("source"), ent.get("profiled_at", "profiles/pmc_latest.json")),
                               "measured_fill_gbps": fill_gbps, "frac_of_measured_fill": achieved / fill_gbps,
                               "bytes_per_launch": alg, "avg_launch_ms": float(ev_ms.mean()),
                               "median_launch_ms": float(np.median(ev_ms)), "perspectives_per_launch": p_mean,
                               "launches_per_step": S * CH, "lattices_per_launch": ns // CH,
                               "launches_timed": int(ev_ms.size),
                               "timed_launches": "every %d%s step of the timed region (a HIP event record costs ~3 us of stream time; "
                                                 "--event-every 1 brackets every launch)" % (EV, "th" if EV > 3 else ("st", "nd", "rd")[EV - 1])}
        if nn_leg is not None:
            res["nn_in_loop"] = nn_leg
        if world == 1 and args.cpu_seconds > 0:
            res["cpu_baseline"] = cpu_baseline(d, args.p_error, args.seed, args.cpu_seconds)
        print(json.dumps(res), file=result_out, flush=True)
    for sh in shards:
        sh.envs.close()
    if dist_on:
        dist.destroy_process_group()



if __name__ == "__main__":
    main()
