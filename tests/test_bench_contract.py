"""bench.py prints exactly one JSON line with the contract's keys (GPU box) and its CPU-baseline leg
runs on its own (no GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def test_cpu_baseline_leg_runs_without_gpu():
    sys.path.insert(0, ROOT)
    import bench
    out = bench.cpu_baseline(5, 0.1, 1, 0.3)
    assert out["kind"] == "port" and out["value"] > 0 and out["cores"] >= 1 and "lattices" in out["sample"]
    assert out["numpy_reference_shaped_steps_per_sec"] > 0


@pytest.mark.gpu
def test_bench_json_contract():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--envs", "8192",
                        "--cpu-seconds", "0.5"], capture_output=True, text=True, timeout=600)
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
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert j["value"] > 1e6 and abs(j["value"] - 8192 * 6 / (j["ms_per_step"] * 6e-3)) / j["value"] < 1e-6


@pytest.mark.gpu
def test_bench_graph_mode():
    """--graph: the flush window captured into a HIP graph (every ABI call is capture-safe: no allocation,
    no synchronisation, caller's stream) and replayed; same JSON contract minus the per-launch roofline."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "32", "--warmup", "8", "--envs", "2048",
                        "--cpu-seconds", "0", "--graph"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert j["config"]["hip_graph"] is True and "roofline" not in j
    assert j["steps"] == 32 and j["value"] > 1e6 and j["perspectives_per_sec"] > 30 * j["value"]
