"""Measured fetch rates of the vector-memory path (rt_probe_fetch_rate): prints one JSON object.

    PYTHONPATH=. python profiles/probe_rates.py > profiles/r2/fetch_rates.json
"""
import json
from raytracer_amd import capi

ctx = capi.Context(0)
out = {"unit": "GB/s", "note": "pattern 0/1: 128-byte nodes per second x 128 B; stream: 16-byte loads per second x 16 B"}
for name, pattern, sizes in (("node_per_lane", 0, (16 << 10, 1 << 20, 16 << 20, 1 << 30)),
                             ("node_per_wave", 1, (16 << 10, 1 << 20, 16 << 20, 1 << 30)),
                             ("stream", 2, (16 << 10, 1 << 20, 16 << 20, 1 << 30))):
    for sz in sizes:
        best = 0.0
        for _ in range(3):
            r = ctx.probe_fetch_rate(pattern, sz, 2048 if sz > (1 << 20) else 8192)
            best = max(best, r)
        out["%s_%dKiB" % (name, sz >> 10)] = best * (16 if pattern == 2 else 128) / 1e9
print(json.dumps(out, indent=1))
