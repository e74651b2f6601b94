"""HBM-side traffic of the dominant kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs).

    python profiles/make_traffic.py gpurun_out/r1 profiles/r1_traffic.json

Both counters are in KiB per dispatch.  On gfx950 FETCH_SIZE tallies each 128-byte fabric read request as 64 bytes, so
the read side is doubled (calibrated for 16-byte-per-lane loads, which is what the ray kernels issue); WRITE_SIZE is
taken as is.  bench.py reports the resulting bytes per launch as roofline.traffic for the same workload."""
import collections
import csv
import json
import os
import sys

SHADOW = "k_rays<1, false, 0,"      # the plain (non-counting) shadow launch, whichever tree variant the workload runs


def mean_counter(path, counter):
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            vals[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}, {k: len(v) for k, v in vals.items()}


def main(src, dst):
    fetch, nf = mean_counter(os.path.join(src, "fetch", "fetch_counter_collection.csv"), "FETCH_SIZE")
    write, nw = mean_counter(os.path.join(src, "write", "write_counter_collection.csv"), "WRITE_SIZE")
    bench = json.loads(open(os.path.join(src, "fetch_bench.json")).read().strip().splitlines()[-1])
    KERNEL = max((k for k in fetch if k.startswith(SHADOW)), key=lambda k: fetch[k])
    out = {
        "kernel": KERNEL,
        "workload": {k: bench["config"].get(k) for k in ("n_side", "n_prims", "rays_per_frame", "max_scatters", "instanced", "device_build")},
        "kernel_source_sha16": bench["config"].get("kernel_source_sha16"),
        "res": bench["metric"].rsplit(" ", 1)[-1],
        "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
        "metric": bench["metric"],
        "fetch_size_kib_per_launch": fetch[KERNEL], "write_size_kib_per_launch": write[KERNEL],
        "launches_averaged": [nf[KERNEL], nw[KERNEL]],
        "read_correction": 2.0,
        "traffic_bytes_per_launch": (2.0 * fetch[KERNEL] + write[KERNEL]) * 1024.0,
        "per_kernel_kib": {k: {"FETCH_SIZE": fetch.get(k), "WRITE_SIZE": write.get(k)} for k in sorted(set(fetch) | set(write))
                           if not k.startswith("__amd")},
        "note": "HBM/fabric-side bytes (Infinity Cache hits included); L1 and L2 hits never reach these counters",
    }
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
