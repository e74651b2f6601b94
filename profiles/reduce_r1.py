"""Copy what is kept under profiles/ out of the raw rocprofv3 output of collect_r1.sh / collect_r1_extra.sh.

    python profiles/reduce_r1.py            (from the repository root, after gpurun merged gpurun_out/)

Kernel names are cut at the argument list (the rocPRIM ones to the pass they belong to) so the tables stay readable."""
import csv
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW, EXTRA, OUT = (os.path.join(ROOT, p) for p in ("gpurun_out/r1", "gpurun_out/r1_extra", "profiles/r1"))


def short(name):
    if "rocprim" in name:
        for tag in ("onesweep_iteration", "onesweep_global_offsets", "radix_sort"):
            if tag in name:
                return "rocprim %s (63-bit keys, 32-bit values)" % tag
        return "rocprim kernel"
    return name.split("(DevScene")[0].replace("(anonymous namespace)::", "").split("(")[0] if "k_" in name else name


def stats(src, dst):
    rows = list(csv.reader(open(src)))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(rows[0])
        for r in rows[1:]:
            w.writerow([short(r[0])] + r[1:])


def main():
    os.makedirs(OUT, exist_ok=True)
    shutil.copy(os.path.join(RAW, "bench.json"), os.path.join(OUT, "bench.json"))
    stats(os.path.join(RAW, "stats", "stats_kernel_stats.csv"), os.path.join(OUT, "kernel_stats.csv"))
    with open(os.path.join(OUT, "pmc_traffic_cache_summary.txt"), "w") as f:
        for sub in ("cache", "fetch", "write"):
            f.write(subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "summarise_pmc.py"), os.path.join(RAW, sub)],
                                   check=True, stdout=subprocess.PIPE, text=True).stdout)
    shutil.copy(os.path.join(RAW, "r1_traffic.json"), os.path.join(ROOT, "profiles", "r1_traffic.json"))    # made on the box, before the bench line
    if os.path.isdir(EXTRA):
        stats(os.path.join(EXTRA, "inst", "inst_kernel_stats.csv"), os.path.join(OUT, "kernel_stats_instanced.csv"))
        stats(os.path.join(EXTRA, "devbuild", "devbuild_kernel_stats.csv"), os.path.join(OUT, "kernel_stats_device_build.csv"))
        for name in ("inst_bench.json", "devbuild_bench.json"):
            lines = open(os.path.join(EXTRA, name)).read().strip().splitlines()
            open(os.path.join(OUT, name), "w").write(lines[-1] + "\n")


if __name__ == "__main__":
    main()
