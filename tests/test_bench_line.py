"""bench.py prints one JSON line with the agreed fields (run on a small grid so that it takes seconds)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_line_with_roofline_and_cpu_baseline():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n-side", "60", "--res", "640x360", "--steps", "3", "--warmup", "1",
                        "--cpu-seconds", "2"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "Mrays/s" and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    roof = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    # the bound is HBM, priced with the survey's algorithmic bytes against the 8 TB/s peak; on a scene that fits the
    # Infinity Cache those bytes are served by caches and the fraction may exceed 1, which the object says itself
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and roof["unit"] == "GB/s" and roof["kernel_ms"] > 0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and roof["frac"] > 0.0
    assert roof["cache_resident"] is True and roof["walked_bytes"] <= roof["scene_bytes"] < (256 << 20)
    assert roof["served_by_caches"] is None or 0.0 <= roof["served_by_caches"] < 1.0
    # every other fraction in the object is a fraction of something: in (0, 1]
    assert 0.0 < roof["fp32"]["frac"] <= 1.0 and 0.0 < roof["l1_probe"]["frac"] <= 1.0
    assert roof["hbm_side"]["frac"] is None or 0.0 < roof["hbm_side"]["frac"] <= 1.0
    dev = d["ms_per_step_device"]
    assert dev["n"] == 3 and dev["sd"] >= 0.0 and dev["min"] <= dev["mean"] <= dev["max"]
    # value = rays of all kinds per second over the timed frames
    assert abs(d["value"] - d["config"]["rays_per_frame"] / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    cpu = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] in ("port", "reference") and cpu["value"] > 0 and cpu["cores"] >= 1
    # three trials per CPU leg with their spread, and the real reference's figures quoted beside them
    assert cpu["trials"] >= 3 and cpu["seconds_sd"] >= 0.0 and cpu["single_thread"]["trials"] >= 3
    assert cpu["brute_force_single_thread"]["trials"] >= 3 and "cpu_reference_survey" in cpu
