#!/usr/bin/env python3
"""Headline benchmark of the toric-code env hot path on MI355X.

A "step" is one pass of the hot path over one batch of lattices, in the call order of the
reference's actor loop (src/Actor_mp.py:104-185) with the policy network left out of the headline
(it is stock torch conv work, out of scope; epsilon starts at 1 upstream, Actor_mp.py:37, where the
Q-values never influence the action):

    perspective counts -> exclusive scan -> perspective stack write (P,2,d,d) f32 + positions
    -> eps=1 selection, env step, transition record, auto-reset, next counts (one fused kernel)
    -> every --flush steps: priorities into the packed block (computePrioritiesParallel with Q = 0)

Workload at N=1: BASELINE.json configs[2], the configuration the metric is quoted on: 65 536
lattices, d=7, p_error=0.10.  Lattices are resident in HBM when the timed region starts; nothing
crosses PCIe in the N=1 loop.  Before the warm-up the episodes are staggered (burn-in: lattice e is
reset at step e mod 76), so the population -- and with it perspectives per lattice -- is stationary
and `value` does not depend on --steps.

N>1: `python bench.py --gpus N` starts N ranks by itself (a `python -m torch.distributed.run` child,
before this process touches the GPU) and relays rank 0's JSON line; under torch.distributed.run
(WORLD_SIZE set) it runs as a rank.  One rank per GPU; every rank owns a contiguous block of global
env ids (weak scaling).  Default shape for N>1: BASELINE.json configs[4] -- 131 072 lattices per
GPU, and every transition (with its priority) is delivered to the HOST replay ring: RCCL gather of
the packed blocks to rank 0 over xGMI every --flush steps, then rank 0's copy stream drains each
gathered slot to pinned host memory.  The rate with the ring kept in rank 0's HBM is measured right
after and reported beside it (`hbm_ring`).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (perspective write
kernel, HIP events around every launch in the timed region), `cpu_baseline` (the C oracle's actor
loop on the host cores, rank 0, N=1 only) and `nn_in_loop` (configs[2] as written: the stack fed
to NN_11 and device-side selection in the loop; N=1 only, --nn-steps 0 to skip).
"""
import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

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
    ap.add_argument("--stack-candidates", type=int, default=12,
                    help="stack buffers to allocate at set-up; the one the write kernel is fastest on is kept, the others are "
                         "freed (the write rate depends on the buffer: 5.1-5.5 TB/s into a plain allocation, 6.5-6.8 into most "
                         "tq_stack_alloc buffers on most boxes, profiles/r03_stack_write_ab.txt).  1 = take the first allocation as it comes")
    ap.add_argument("--stack-good-enough", type=float, default=0.80,
                    help="the probe stops early at a candidate whose write takes less than this fraction of candidate 0's "
                         "(well-placed buffers take 0.79-0.83 of a plain allocation's time and differ by ~2 %% among themselves: "
                         "the default tries nearly always all candidates, ~1 s of set-up)")
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
    """One sub-shard of this GPU's lattices with its stream and its caller-owned output buffers."""


def time_plain_loop(T, torch, env, n, d, seed, first, tdtype, flush, device, steps, warm, chunks=1, events=False, candidates=1,
                    kinds=("torch", "chunked"), event_every=4, good_enough=0.80):
    """The same pass over a batch of `n` lattices on the current stream, no collective: burn-in, `warm` untimed
    and `steps` timed steps.  -> (seconds, perspectives in the timed steps, per-step stack-write milliseconds
    from HIP events or None).  Used at N=1 for the extra legs of the line: one GPU on the per-GPU shape of the
    N>1 runs (configs[4]: 131 072 lattices) and BASELINE configs[3] (65 536 lattices, d=9, p=0.15), one shot
    and with the stack written in `chunks` lattice ranges into a buffer of 1/chunks the size."""
    nq = 2 * d * d
    envs = T.EnvSet(env, n, device=device, seed=seed, first_env_id=first, numpy_io=False)
    envs.resetAll()
    stack = torch.empty(((n // chunks) * nq, 2, d, d), dtype=tdtype, device=device)
    positions = torch.empty(((n // chunks) * nq, 3), dtype=torch.int32, device=device)
    offs = torch.zeros((warm + steps, (n + 2) & ~1), dtype=torch.int64, device=device)
    blocks = [envs.newTransitionBlock(steps=flush) for _ in range(2)]
    every = max(1, min(int(event_every), steps))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range((steps + every - 1) // every)] if events else None
    for t in range(EPISODE):
        idx = torch.arange(t, n, EPISODE, dtype=torch.int32, device=device)
        if idx.numel():
            envs.resetTerminalEnvs(idx)
        envs.actorStep(None, want_actions=False)
    probe = None
    if candidates > 1:                                               # placement probe, as in the main loop
        del stack
        torch.cuda.empty_cache()
        if chunks == 1:
            stack, probe = envs.pickStackBuffer(candidates, dtype=tdtype, positions=positions, kinds=kinds, park=True, good_enough=good_enough)
        else:                                                        # the small buffer of the range-by-range consumer, probed with the first range
            stack, probe = envs.pickStackBuffer(candidates, dtype=tdtype, capacity=(n // chunks) * nq, positions=positions, kinds=kinds,
                                                park=True, first=0, count=n // chunks, good_enough=good_enough)

    def step(t):
        off = offs[t][:n + 1]
        envs.perspectiveCounts(off)
        timed = ev is not None and t >= warm and (t - warm) % every == 0
        if timed:
            ev[(t - warm) // every][0].record()
        if chunks == 1:
            envs.writePerspectives(stack, positions, off)
        else:
            for c in range(chunks):
                envs.writePerspectives(stack, positions, off, first=c * (n // chunks), count=n // chunks)
        if timed:
            ev[(t - warm) // every][1].record()
        blk = blocks[(t // flush) & 1]
        envs.actorStep(None, block=blk, slot=t % flush, want_actions=True)
        if (t + 1) % flush == 0:
            blk.computePriorities(n, flush, None, 0.95)

    for t in range(warm):
        step(t)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for t in range(warm, warm + steps):
        step(t)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    P = float(offs[warm:, n].sum().item())
    # the timed buffer holds the right bytes (see main): whole batch, or the last range of the range-by-range consumer
    off_v = offs[0][:n + 1]
    envs.perspectiveCounts(off_v)
    first_v, count_v = (0, n) if chunks == 1 else ((chunks - 1) * (n // chunks), n // chunks)
    Pv = int((off_v[first_v + count_v] - off_v[first_v]).item())
    ref_s = torch.empty((Pv, 2, d, d), dtype=tdtype, device=device)
    ref_p = torch.empty((Pv, 3), dtype=torch.int32, device=device)
    envs.writePerspectives(ref_s, ref_p, off_v, first=first_v, count=count_v)
    stack.view(torch.uint8).fill_(0x5A)
    envs.writePerspectives(stack, positions, off_v, first=first_v, count=count_v)
    torch.cuda.synchronize(device)
    wrong = int((stack[:Pv].view(torch.uint8) != ref_s.view(torch.uint8)).sum().item()) + int((positions[:Pv] != ref_p).sum().item())
    time_plain_loop.last_verified = {"ok": wrong == 0, "wrong_bytes": wrong, "perspectives": Pv}
    del ref_s, ref_p
    time_plain_loop.last_p_bracketed = float(offs[warm::every, n].double().mean().item())   # per step, of the steps the events bracket
    envs.check()
    envs.close()                                                    # frees the parked candidates of the probe as well ...
    del stack
    torch.cuda.empty_cache()
    time.sleep(0.5)                                                 # ... and the driver wipes freed memory in the background
    ev_ms = np.array([a.elapsed_time(b) for a, b in ev]) if ev is not None else None
    time_plain_loop.last_probe = probe
    return dt, P, ev_ms


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
                                                    park=True, good_enough=args.stack_good_enough)
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

    # ---- N=1 on the default shape: one GPU on the per-GPU shape of the N>1 runs (configs[4]), like for like
    shard_leg = None
    if world == 1 and not dist_on and args.envs is None and graph is None and args.policy == "explore" and not args.no_shard_leg:
        k2, w2 = max(8, min(K, 40)), 8
        print("[bench] configs[4] shard leg (131072 lattices on this GPU) ...", file=sys.stderr, flush=True)
        dt2, P2, _ = time_plain_loop(T, torch, env, ENVS_MULTI, d, args.seed, 0, tdtype, flush, device, k2, w2, candidates=args.stack_candidates,
                                     good_enough=args.stack_good_enough)
        shard_leg = {"envs_per_gpu": ENVS_MULTI, "steps": k2, "value": ENVS_MULTI * k2 / dt2, "ms_per_step": 1e3 * dt2 / k2,
                     "perspectives_per_sec": P2 / dt2, "stack_buffer_probe": time_plain_loop.last_probe,
                     "stack_verified": time_plain_loop.last_verified,
                     "note": "this GPU alone on the per-GPU shape of the N>1 runs (BASELINE configs[4]: 131 072 lattices), "
                             "no collective: the like-for-like base of the scaling curve"}

    # ---- N=1 on the default shape: BASELINE configs[3] (65 536 lattices, d=9, p=0.15) timed by this very run,
    # one shot and with the stack written in 4 lattice ranges into a buffer of a quarter of the size
    c3_leg = None
    if (world == 1 and not dist_on and args.envs is None and d == 7 and graph is None and args.policy == "explore"
            and not args.no_shard_leg and args.out_dtype == "f32"):
        d3, p3, n3, k3, w3 = 9, 0.15, ENVS_N1, max(8, min(K, 20)), 5
        env3 = T.make("toric-code-v0", {"size": d3, "min_qubit_errors": 0, "p_error": p3})
        c3_leg = {"workload": "BASELINE configs[3]: %d lattices, d=%d, p_error=%g, f32 stack; same actor-loop pass" % (n3, d3, p3),
                  "steps": k3, "warmup": w3}
        for name, ch in (("one_shot", 1), ("chunks_4", 4)):
            print("[bench] configs[3] leg (65536 lattices, d=9, p=0.15), %s ..." % name, file=sys.stderr, flush=True)
            dt3, P3, ev3 = time_plain_loop(T, torch, env3, n3, d3, args.seed, 0, tdtype, flush, device, k3, w3, chunks=ch, events=True,
                                           candidates=args.stack_candidates, event_every=args.event_every, good_enough=args.stack_good_enough)
            alg3 = time_plain_loop.last_p_bracketed * (2 * d3 * d3 * 4 + 12) + n3 * 2 * d3 * d3
            c3_leg[name] = {"value": n3 * k3 / dt3, "unit": "env-steps/s", "ms_per_step": 1e3 * dt3 / k3,
                            "stack_buffer_probe": time_plain_loop.last_probe, "stack_verified": time_plain_loop.last_verified,
                            "perspectives_per_sec": P3 / dt3, "perspectives_per_lattice": P3 / (k3 * n3),
                            "roofline": {"bound": "hbm", "kernel": STACK_KERNEL, "achieved": alg3 / (ev3.mean() * 1e-3) / 1e9,
                                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg3 / (ev3.mean() * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                         "bytes_per_step": alg3, "avg_write_ms_per_step": float(ev3.mean()),
                                         "launches_per_step": ch,
                                         "note": "HIP events around the stack write(s) of every %d. timed step%s" % (max(1, min(args.event_every, k3)),
                                                 ("" if ch == 1 else ": the %d range launches and their gaps included (every workgroup finds its own cut points)" % ch))}}

    # ---- N=1: configs[2] as written -- generatePerspective feeding NN_11 for selectAction, measured at size in
    # f32 (what upstream runs) and with the bf16 stack the kernels can write directly + bf16 autocast
    nn_leg = None
    if world == 1 and args.nn_steps > 0 and args.policy == "explore" and graph is None and S == 1 and CH == 1:
        sh0 = shards[0]
        flop_per_persp = 2.0 * sum(ci * co * 9 * ((d - 2) ** 2 if i == 10 else d * d)
                                   for i, (ci, co) in enumerate(zip((2, 128, 128, 120, 111, 104, 103, 90, 80, 73, 71),
                                                                    (128, 128, 120, 111, 104, 103, 90, 80, 73, 71, 64))))
        weights = load_trained_weights(d)
        nn_leg = {"workload": "configs[2] as written: %d lattices, d=%d: stack -> NN_11 (%s, stock torch conv) -> device "
                              "eps=%g greedy selection -> fused step" %
                              (n, d, "the reference's trained d=%d weights" % d if weights is not None else "random init", args.eps),
                  "steps": args.nn_steps, "variants": {}}
        base = W + K
        for vname, nn_dtype in (("f32", "f32"), ("bf16", "bf16")):
            m = make_model()
            if weights is not None:
                m.load_state_dict(weights)
            stack_save = sh0.stack
            if nn_dtype == "bf16":                                    # the stack written as bf16 by the kernel itself (TQ_BF16)
                sh0.stack = torch.empty(sh0.stack.shape, dtype=torch.bfloat16, device=device)
            state["tg"], state["model"], args.nn_dtype = None, m, nn_dtype
            print("[bench] nn_in_loop %s: warm-up step (MIOpen picks its kernels) ..." % vname, file=sys.stderr, flush=True)
            t0 = time.perf_counter()
            one_step(0, base)
            torch.cuda.synchronize(device)
            print("[bench] nn_in_loop %s: warm-up took %.1f s; timing %d step(s) ..." % (vname, time.perf_counter() - t0, args.nn_steps),
                  file=sys.stderr, flush=True)
            t0 = time.perf_counter()
            for i in range(args.nn_steps):
                one_step(0, base + 1 + i)
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
            P_nn = float(sh0.offs[base + 1:base + 1 + args.nn_steps, ns].sum().item())
            nn_leg["variants"][vname] = {"stack_dtype": "bf16" if nn_dtype == "bf16" else args.out_dtype, "nn_dtype": nn_dtype,
                                         "env_steps_per_sec": n * args.nn_steps / dt, "perspectives_per_sec_into_nn": P_nn / dt,
                                         "ms_per_step": 1e3 * dt / args.nn_steps, "nn_tflops": P_nn * flop_per_persp / dt / 1e12}
            base += 1 + args.nn_steps
            sh0.stack = stack_save
            state["model"] = None
            del m
            sh0.envs.check()
        for k_ in ("env_steps_per_sec", "perspectives_per_sec_into_nn", "ms_per_step", "nn_tflops", "nn_dtype"):
            nn_leg[k_] = nn_leg["variants"]["f32"][k_]                # top level = the f32 run, as upstream

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
