"""Copy what is kept under profiles/r2/ out of the raw rocprofv3 output of collect_r2.sh (and collect_r2_extra.sh).

    python profiles/reduce_r2.py            (from the repository root, after gpurun merged gpurun_out/)

Kernel names are cut at the argument list so the tables stay readable."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles"))
from reduce_r1 import stats  # noqa: E402

RAW, EXTRA, OUT = (os.path.join(ROOT, p) for p in ("gpurun_out/r2", "gpurun_out/r2_extra", "profiles/r2"))


def summary(subs, dst):
    with open(dst, "w") as f:
        for sub in subs:
            f.write(subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "summarise_pmc.py"), os.path.join(RAW, sub)],
                                   check=True, stdout=subprocess.PIPE, text=True).stdout)


def main():
    os.makedirs(OUT, exist_ok=True)
    shutil.copy(os.path.join(RAW, "bench.json"), os.path.join(OUT, "bench.json"))
    stats(os.path.join(RAW, "stats", "stats_kernel_stats.csv"), os.path.join(OUT, "kernel_stats.csv"))
    summary(("cache", "fetch", "write"), os.path.join(OUT, "pmc_traffic_cache_summary.txt"))
    summary(("sq1", "sq2", "mem"), os.path.join(OUT, "pmc_sq_summary.txt"))
    shutil.copy(os.path.join(RAW, "r2_traffic.json"), os.path.join(ROOT, "profiles", "r2_traffic.json"))    # made on the box, before the bench line
    if os.path.isdir(EXTRA):
        for tag in ("inst", "devbuild"):
            src = os.path.join(EXTRA, tag, tag + "_kernel_stats.csv")
            if os.path.exists(src):
                stats(src, os.path.join(OUT, "kernel_stats_%s.csv" % {"inst": "instanced", "devbuild": "device_build"}[tag]))
            name = tag + "_bench.json"
            if os.path.exists(os.path.join(EXTRA, name)):
                lines = open(os.path.join(EXTRA, name)).read().strip().splitlines()
                open(os.path.join(OUT, name), "w").write(lines[-1] + "\n")


if __name__ == "__main__":
    main()
