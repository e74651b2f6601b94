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
    # the bound is the vector-memory request path (node fetches served by L1 / L2), priced against a rate measured in the
    # same run; every fraction in the object is a fraction of something: in (0, 1]
    assert roof["bound"] == "l1" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and roof["kernel_ms"] > 0
    assert 0.0 < roof["frac"] <= 1.0
    assert 0.0 < roof["fp32"]["frac"] <= 1.0
    assert roof["hbm"]["frac"] is None or 0.0 < roof["hbm"]["frac"] <= 1.0

    def fracs(o):
        for k, v in o.items():
            if isinstance(v, dict):
                yield from fracs(v)
            elif k.endswith("frac") and v is not None:
                yield v
    assert all(0.0 < f <= 1.0 for f in fracs(roof))
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
