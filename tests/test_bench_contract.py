"""bench.py prints exactly one JSON line with the contract's keys (GPU box) and its CPU-baseline leg
runs on its own (no GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "nn_in_loop", "perspectives_per_lattice"}


def test_cpu_baseline_leg_runs_without_gpu():
    sys.path.insert(0, ROOT)
    import bench
    out = bench.cpu_baseline(5, 0.1, 1, 0.3)
    assert out["kind"] == "port" and out["value"] > 0 and out["cores"] >= 1 and "lattices" in out["sample"]
    assert out["numpy_reference_shaped_steps_per_sec"] > 0


def test_roofline_arithmetic_and_config_names():
    """The `roofline` object of a leg: algorithmic bytes per launch over the mean launch time, against 8 TB/s."""
    sys.path.insert(0, ROOT)
    import bench
    alg = 4790616 * (98 * 4 + 12) + 65536 * 98                # SURVEY 8(d) at configs[2]: P * (2 d^2 * 4 + 12) + N * 2 d^2
    r = bench.hbm_roofline(alg, [0.29, 0.30, 0.28], {"probe_ms_chosen": 0.29})
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["launches_timed"] == 3
    assert abs(r["achieved"] - alg / 0.29e-3 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert r["median_launch_ms"] == 0.29 and r["bytes_per_launch"] == alg and r["probe_ms_chosen"] == 0.29
    # the first launch of a region of K counts 1/K, the regular sample of the others (K-1)/K
    w = bench.hbm_roofline(alg, [0.32, 0.29, 0.30, 0.28], first_of=20)
    assert abs(w["avg_launch_ms"] - (0.32 + 19 * 0.29) / 20) < 1e-12 and w["first_launch_ms"] == 0.32 and w["launches_timed"] == 4
    assert bench.config_name(1, 65536, 7, 0.10) == "configs[2]" and bench.config_name(1, 65536, 9, 0.15) == "configs[3]"
    assert bench.config_name(8, 131072, 7, 0.10) == "configs[4] shape" and bench.config_name(1, 8192, 7, 0.10) == "custom"
    assert bench.config_name(1, 4096, 5, 0.10) == "configs[1]"


def test_self_launcher_starts_the_ranks_and_relays_one_line():
    """`python bench.py --gpus 2` with no torch.distributed.run around it (the driver's command form)
    must start the two ranks itself and print exactly one JSON line with n_gpus == 2.  No GPU here:
    TORIC_BENCH_DRY_RUN=1 makes the ranks stop after the rendezvous (gloo) -- the launcher, the
    argument pass-through, the relay of rank 0's line and the exit code are what is checked."""
    env = dict(os.environ, TORIC_BENCH_DRY_RUN="1", TORIC_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "7", "--warmup", "3"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["steps"] == 7 and j["warmup"] == 3 and j["dry_run"] is True


def test_self_launcher_propagates_a_rank_failure():
    env = dict(os.environ, TORIC_BENCH_DRY_RUN="1", TORIC_DIST_BACKEND="no-such-backend")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_bench_two_ranks_share_the_gpu():
    """The N>1 path for real on a one-GPU box: the self-launcher starts two ranks that both use cuda:0
    (TORIC_SHARE_GPU=1) and exchange their packed transition blocks, priorities included, through
    gloo (RCCL refuses two ranks on one device)."""
    env = dict(os.environ, TORIC_DIST_BACKEND="gloo", TORIC_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16", "--warmup", "8",
                        "--envs", "32768", "--cpu-seconds", "0", "--nn-steps", "0", "--roots", "2"], capture_output=True, text=True, timeout=900,
                       env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["envs_per_gpu"] == 32768
    assert j["config"]["streams_per_gpu"] == 2        # the two-stream loop: the gathers are issued from its side stream
    assert "2 ranks" in j["config"]["collective"] and j["value"] > 1e6 and j["config"]["gather_roots"] == 2
    assert [r_["rank"] for r_ in j["ranks"]] == [0, 1] and all(r_["stack_verified"] and r_["frac"] > 0 for r_ in j["ranks"])
    assert abs(j["value"] - 2 * 32768 * 16 / (j["ms_per_step"] * 16e-3)) / j["value"] < 1e-6


@pytest.mark.gpu
def test_bench_rccl_path_keeps_stdout_to_one_line():
    """The collective path over RCCL (a world of one rank on this box, TORIC_FORCE_DIST=1) with delivery to
    the pinned host ring: RCCL prints its version banner on stdout when the communicator comes up -- the
    bench must still print exactly one line there -- and the line carries the HBM-ring rate beside it."""
    env = dict(os.environ, TORIC_FORCE_DIST="1", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):   # no port given: bench.py picks a free one
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "16", "--warmup", "8", "--envs", "32768",
                        "--cpu-seconds", "0", "--nn-steps", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[:500]
    j = json.loads(lines[0])
    assert j["config"]["delivery"] == "host" and "nccl, 1 ranks" in j["config"]["collective"] and j["config"]["streams_per_gpu"] == 2
    assert j["hbm_ring"]["value"] > 1e6 and j["value"] > 1e6
    rk = j["ranks"]                                           # every rank's own roofline
    assert len(rk) == 1 and rk[0]["rank"] == 0 and 0 < rk[0]["frac"] < 1 and rk[0]["stack_verified"] is True
    assert rk[0]["probe_ms_chosen"] > 0 and rk[0]["timed_over_probe"] > 0


@pytest.mark.gpu
def test_bench_json_contract():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--envs", "8192",
                        "--cpu-seconds", "0.5", "--nn-steps", "1"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert KEYS <= set(j)
    assert j["n_gpus"] == 1 and j["steps"] == 6 and j["warmup"] == 2 and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["achieved"] > 100
    assert r["traffic"] is None or 0.9 < r["traffic_over_algorithmic"] < 1.3
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["one_core"]["value"] > 0
    nn = j["nn_in_loop"]
    assert nn["steps"] == 1 and nn["perspectives_per_sec_into_nn"] > 1e4 and nn["env_steps_per_sec"] > 100
    assert set(nn["variants"]) == {"f32", "bf16"} and nn["variants"]["bf16"]["stack_dtype"] == "bf16"
    pr = j["stack_buffer_probe"]                              # set-up probe of the stack buffer's placement, reported in full
    assert pr["kinds"][0] == "torch.empty" and len(pr["write_ms"]) == pr["candidates"] >= 2 and 0 <= pr["chosen"] < pr["candidates"]
    # the figure the timed region is held against is taken behind the settling passes; the one from the candidates' bursts is kept beside it
    assert pr["writes_per_candidate"] >= 10 and pr["probe_ms_chosen"] > 0 and pr["probe_ms_in_bursts"] > 0 and pr["settle_steps"] >= 1
    assert pr["candidates_added_because_uniform"] in (0, pr["candidates"] // 2)
    assert r["workgroup_shares"]["xcd_bias"] in range(0, 17)
    # every leg explains itself: what the probe promised, what the timed region delivered, what a default allocation would give
    assert r["probe_ms_chosen"] > 0 and abs(r["timed_write_ms"] - r["avg_launch_ms"]) < 1e-9 and r["first_launch_ms"] > 0
    assert abs(r["timed_over_probe"] - r["timed_write_ms"] / r["probe_ms_chosen"]) < 1e-9
    assert r["default_buffer"]["kind"] == "torch.empty" and 0 < r["default_buffer"]["frac"] <= 1.0
    assert set(j["probe_vs_timed"]) == {"headline"} and isinstance(j["legs_outside_3pct_of_probe"], list)
    assert j["config"]["streams_per_gpu"] == 1 and "ExploreLoop" in j["config"]["loop"]      # 8192 lattices: below the two-stream threshold
    assert j["non_write_us_per_step"] == pytest.approx(1e3 * (j["ms_per_step"] - r["avg_launch_ms"]))
    assert "warm_write_ms" in j["warm_up"]
    assert j["stack_verified"]["ok"] is True and j["stack_verified"]["wrong_bytes"] == 0      # the timed buffer holds the right bytes
    assert "host_twin" in c and (c["host_twin"].get("value", 0) > 0 or "error" in c["host_twin"])
    assert r["kernel"] == "k_persp_stream" and "custom" in j["config"]["workload"]      # 8192 lattices: not a BASELINE config
    assert j["config"]["steady_state"] is True and 40 < j["perspectives_per_lattice"] < 98
    assert j["value"] > 1e6 and abs(j["value"] - 8192 * 6 / (j["ms_per_step"] * 6e-3)) / j["value"] < 1e-6


@pytest.mark.gpu
def test_bench_graph_mode():
    """--graph: the flush window captured into a HIP graph (every ABI call is capture-safe: no allocation,
    no synchronisation, caller's stream) and replayed; same JSON contract minus the per-launch roofline."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "32", "--warmup", "8", "--envs", "2048",
                        "--cpu-seconds", "0", "--nn-steps", "0", "--graph"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert j["config"]["hip_graph"] is True and "roofline" not in j
    assert j["steps"] == 32 and j["value"] > 1e6 and j["perspectives_per_sec"] > 30 * j["value"]
