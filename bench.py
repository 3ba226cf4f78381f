#!/usr/bin/env python3
"""Headline benchmark of the toric-code env hot path on MI355X.

A "step" is one pass of the hot path over one batch of lattices, in the call order of the
reference's actor loop (src/Actor_mp.py:104-185) with the policy network left out (it is stock
torch conv work, out of scope; epsilon starts at 1 upstream, Actor_mp.py:37, where the Q-values
never influence the action):

    perspective counts -> exclusive scan -> perspective stack write (P,2,d,d) f32 + positions
    -> eps=1 selection, env step, transition record, auto-reset, next counts (one fused kernel)

Workload at N=1: BASELINE.json configs[2], the configuration the metric is quoted on:
65 536 lattices, d=7, p_error=0.10.  Inputs are resident in HBM when the timed region starts
(the lattices live on the device; nothing crosses PCIe in the loop).

Optional (--shards S > 1, default 1): lattices are independent, so the batch can be processed as S
sub-shards on separate HIP streams, one shard's latency-bound kernels (scan, fused step) running
beside another's stack write; writes are ordered against each other with events so their HIP-event
timing stays clean.  Measured slower than one stream, hence off by default.

N>1 (launched by torch.distributed.run, one rank per GPU): every rank owns a contiguous block of
global env ids (weak scaling: 65 536 lattices per GPU); the only exchange is the gather of packed
transition blocks to rank 0's HBM replay ring (RCCL over xGMI) every --flush steps, issued async
so it overlaps the next steps.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (perspective
write kernel, HIP events around every launch in the timed region) and `cpu_baseline` (the C
oracle's actor loop on the host cores, rank 0, N=1 only).
"""
import argparse
import contextlib
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes), before HIP initialises

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 measured copy


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=65536, help="lattices per GPU")
    ap.add_argument("--size", type=int, default=7)
    ap.add_argument("--p-error", type=float, default=0.10)
    ap.add_argument("--seed", type=int, default=2020)
    ap.add_argument("--out-dtype", default="f32", choices=["f32", "f16", "bf16", "u8"])
    ap.add_argument("--flush", type=int, default=8, help="steps per transition block / gather (N>1)")
    ap.add_argument("--shards", type=int, default=1, help="independent sub-shards (HIP streams) per GPU")
    ap.add_argument("--no-transitions", action="store_true", help="do not write transition records")
    ap.add_argument("--host-drain", action="store_true",
                    help="N>1: rank 0 also copies every gathered block to a pinned host ring (replay process side)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--no-events", action="store_true", help="no per-launch HIP events (pure wall clock)")
    ap.add_argument("--graph", action="store_true",
                    help="capture --flush steps in a HIP graph and replay it (launch-bound small batches; implies "
                         "--no-events: no per-launch timing inside a graph, so no roofline object)")
    ap.add_argument("--policy", default="explore", choices=["explore", "nn11"],
                    help="explore: eps=1 selection in the fused kernel (default, the env path alone); "
                         "nn11: NN_11 forward on the stack + device eps-greedy selection in the loop (NN-bound)")
    ap.add_argument("--eps", type=float, default=0.1, help="epsilon of the nn11 policy")
    ap.add_argument("--nn-dtype", default="bf16", choices=["f32", "bf16"], help="autocast dtype of the nn11 forward")
    return ap.parse_args()


def cpu_baseline(d, p, seed, budget_s):
    """The oracle's C actor loop (EnvSet.step + generatePerspectiveBatch + generateTransitionParallel
    restated, oracle/toric_oracle.c) timed on the host cores: bounded sample of the same workload."""
    from oracle.c_oracle import CEnvBatch, lib
    L = lib()
    # host cores this process may use (the GPU box gives one GPU's share of a big host)
    threads = max(1, min(len(os.sched_getaffinity(0)), L.tor_num_threads(), int(os.environ.get("TORIC_CPU_THREADS", "16"))))
    L.tor_set_threads(threads)
    n = 4096
    env = CEnvBatch(d, n, p, seed=seed)
    env.reset()
    env.actor_steps(2)                                        # page in, spin up the thread team
    t0 = time.perf_counter()
    env.actor_steps(8)
    probe = (time.perf_counter() - t0) / 8
    steps = int(max(4, min(2000, budget_s / max(probe, 1e-6))))
    t0 = time.perf_counter()
    P, _ = env.actor_steps(steps)
    dt = time.perf_counter() - t0
    out = {"value": n * steps / dt, "unit": "env-steps/s", "cores": int(threads), "kind": "port",
           "sample": f"{n} lattices x {steps} steps, d={d}, p={p}, C oracle actor loop (OpenMP, {threads} threads), {dt:.1f} s",
           "perspectives_per_sec": P / dt}
    # the same loop in the reference's own shape (per-lattice python + np.roll), tiny sample
    from oracle import toric_oracle as O
    oe = O.OracleEnvSet(d, 64, p, seed=seed)
    oe.resetAll()
    t0 = time.perf_counter()
    O.run_actor_steps_ref(oe, 4, eps=1.0)
    out["numpy_reference_shaped_steps_per_sec"] = 64 * 4 / (time.perf_counter() - t0)
    return out


class Shard:
    """One sub-shard of this GPU's lattices with its stream and its caller-owned output buffers."""


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
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
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    d, n, K, W = args.size, args.envs, args.steps, args.warmup
    nq = 2 * d * d
    tdtype = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16, "u8": torch.uint8}[args.out_dtype]
    esize = {"f32": 4, "f16": 2, "bf16": 2, "u8": 1}[args.out_dtype]
    S = max(1, args.shards)
    if n % S:
        sys.exit("--envs must be divisible by --shards")
    ns = n // S
    flush = max(1, args.flush)
    use_events = not args.no_events and not args.graph
    if args.graph and (args.policy != "explore" or world > 1 or args.shards != 1):
        sys.exit("--graph supports the single-GPU, single-stream explore policy only")
    if args.graph:
        K = max(flush, K - K % flush)                                  # whole replays
        W = max(flush, W - W % flush)

    model = None
    if args.policy == "nn11":
        from toric_rl_decoder_amd.policy import NN_11
        torch.manual_seed(0)                                          # random-init weights of the NN_11 architecture
        model = NN_11(d, 3).to(device).eval()

    env = T.make("toric-code-v0", {"size": d, "min_qubit_errors": 0, "p_error": args.p_error})
    first, _ = G.shard_range(n * world, world, rank)
    # worst case every qubit is a hit (exploration grows the defect density): size each stack for
    # that -- 2.5 GB at d=7 f32, 6.9 GB at d=9 for 65 536 lattices -- out of 288 GB of HBM
    cap = ns * nq
    shards = []
    for k in range(S):
        sh = Shard()
        sh.stream = torch.cuda.Stream(device=device) if S > 1 else torch.cuda.current_stream(device)
        with torch.cuda.stream(sh.stream):
            sh.envs = T.EnvSet(env, ns, device=device, seed=args.seed, first_env_id=first + k * ns, numpy_io=False)
            sh.envs.resetAll()
            sh.stack = torch.empty((cap, 2, d, d), dtype=tdtype, device=device)
            sh.positions = torch.empty((cap, 3), dtype=torch.int32, device=device)
            sh.offs = torch.zeros((W + K, ns + 1), dtype=torch.int64, device=device)     # one scan per step: P = row[-1]
            sh.blocks = None if args.no_transitions else [sh.envs.newTransitionBlock(steps=flush) for _ in range(2)]
            sh.ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                     for _ in range(K)] if use_events else []
            sh.wrote = torch.cuda.Event()
            sh.eps = torch.full((ns,), args.eps, dtype=torch.float64, device=device) if model is not None else None
        shards.append(sh)
    torch.cuda.synchronize(device)
    have_blocks = shards[0].blocks is not None
    tg = None
    if dist_on and have_blocks:
        tg = [G.TransitionGather(sh.blocks[0].nbytes, device, ring_slots=2, host_drain=args.host_drain) for sh in shards]

    def one_step(k, t, timed_idx=None):
        sh = shards[k]
        # one shard: stay on torch's current stream (inside torch.cuda.graph() that is the capture stream)
        with (torch.cuda.stream(sh.stream) if S > 1 else contextlib.nullcontext()):
            envs, off = sh.envs, sh.offs[t]
            envs.perspectiveCounts(off)
            if S > 1:
                sh.stream.wait_event(shards[(k - 1) % S].wrote)          # one stack write at a time
            if timed_idx is not None and use_events:
                sh.ev[timed_idx][0].record(sh.stream)
            envs.writePerspectives(sh.stack, sh.positions, off)
            if timed_idx is not None and use_events:
                sh.ev[timed_idx][1].record(sh.stream)
            if S > 1:
                sh.wrote.record(sh.stream)
            blk = sh.blocks[(t // flush) & 1] if have_blocks else None
            act = None
            if model is not None:                                     # configs[2] as written: stack -> NN_11 -> selection
                P = int(off[-1].item())
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.nn_dtype == "bf16"):
                    q = torch.cat([model(sh.stack[i:i + 32768]) for i in range(0, P, 32768)]).float()
                act, _ = envs.selectAction(q, sh.eps, positions=sh.positions, offsets=off)
            envs.actorStep(act, block=blk, slot=t % flush, want_actions=True)
            if tg is not None and (t + 1) % flush == 0:
                tg[k].gather(blk.buf)

    def barrier():
        torch.cuda.synchronize(device)
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(device)

    if tg is not None:        # RCCL sets its communicator up lazily: pay for that before anything is timed
        for k, g in enumerate(tg):
            g.gather(shards[k].blocks[1].buf)
            g.wait()
    barrier()
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
                p_acc[t:t + 1].add_(sh0.offs[t, -1:])                 # running sum of P per slot of the window
        p_acc.zero_()

    def run_steps(first, count, timed):
        if graph is not None:
            for _ in range(count // flush):
                graph.replay()
            return
        for i in range(count):
            for k in range(S):
                one_step(k, first + i, i if timed else None)

    run_steps(0, W, False)
    barrier()
    if graph is not None:
        p_acc.zero_()
        torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    run_steps(W, K, True)
    if tg is not None:
        for k, g in enumerate(tg):
            with torch.cuda.stream(shards[k].stream):
                g.wait()
    barrier()
    elapsed = time.perf_counter() - t0
    for sh in shards:
        sh.envs.check()                                               # capacity / action latch

    el = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    if dist_on:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    if graph is not None:      # slot t of the captured window accumulated its P over the K/flush replays
        p_timed = (p_acc.to(torch.float64) / (K // flush)).repeat(K // flush).reshape(1, -1)
    else:
        p_timed = torch.stack([sh.offs[W:, -1] for sh in shards]).to(torch.float64)   # (S, K) perspectives per launch
    p_sum = p_timed.sum().reshape(1).to(red_dev)
    if dist_on:
        dist.all_reduce(p_sum, op=dist.ReduceOp.SUM)
    total_steps = float(n) * world * K

    if rank == 0:
        policy_txt = "policy NN excluded (eps=1 selection in the fused kernel)" if model is None else \
            "NN_11 (random init, %s) forward + eps=%g greedy selection IN the loop" % (args.nn_dtype, args.eps)
        res = {
            "metric": "env steps/sec (batched) at d=%d p=%g" % (d, args.p_error),
            "value": total_steps / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: %d lattices/GPU, d=%d, p_error=%g; actor-loop pass = "
                                   "perspective stack (%s) + positions -> selection -> step -> transition "
                                   "record -> auto-reset (max 75 steps/episode); %s" %
                                   (n, d, args.p_error, args.out_dtype, policy_txt),
                       "policy": args.policy, "envs_per_gpu": n, "d": d, "p_error": args.p_error,
                       "out_dtype": args.out_dtype, "transitions": have_blocks, "flush_steps": flush,
                       "streams_per_gpu": S, "hip_graph": bool(args.graph), "parallelism": "env-shard x%d" % world,
                       "collective": None if not dist_on else
                       "transition gather to rank 0 (%s) every %d steps%s" % (backend, flush, " + D2H drain to pinned host ring" if args.host_drain else "")},
            "perspectives_per_sec": float(p_sum.item()) / elapsed,
        }
        if use_events:
            ms = np.array([[a.elapsed_time(b) for a, b in sh.ev] for sh in shards])     # (S, K) launch durations
            p_mean = float(p_timed.mean().item())
            alg = p_mean * (nq * esize + 12) + ns * nq                 # P*(B_p+12) + N*2d^2, per launch
            achieved = alg / (ms.mean() * 1e-3) / 1e9
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
            if os.path.exists(pmc):
                try:
                    j = json.load(open(pmc))
                    if j.get("envs") == ns and j.get("d") == d and j.get("out_dtype") == args.out_dtype:
                        traffic = j.get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            # context: the box's own streaming-fill bandwidth, measured after the timed region
            # on the same buffer (hipMemsetAsync through torch), so frac can be read against it as well
            fb = shards[0].stack.view(torch.uint8).reshape(-1)[:int(alg) & ~4095]
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fill_ms = []
            for _ in range(6):
                f0.record(); fb.zero_(); f1.record(); f1.synchronize()
                fill_ms.append(f0.elapsed_time(f1))
            fill_gbps = fb.numel() / (min(fill_ms[1:]) * 1e-3) / 1e9
            res["roofline"] = {"bound": "hbm", "kernel": "k_persp_write", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                               "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                               "measured_fill_gbps": fill_gbps, "frac_of_measured_fill": achieved / fill_gbps,
                               "bytes_per_launch": alg, "avg_launch_ms": float(ms.mean()),
                               "median_launch_ms": float(np.median(ms)), "perspectives_per_launch": p_mean,
                               "launches_per_step": S, "lattices_per_launch": ns}
        if world == 1 and args.cpu_seconds > 0:
            res["cpu_baseline"] = cpu_baseline(d, args.p_error, args.seed, args.cpu_seconds)
        print(json.dumps(res), flush=True)
    for sh in shards:
        sh.envs.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
