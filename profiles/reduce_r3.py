"""Copy what is kept under profiles/ out of the raw rocprofv3 output of collect_r3.sh.

    python profiles/reduce_r3.py r3 profiles/r3            (from the repository root, after gpurun merged gpurun_out/)
    python profiles/reduce_r3.py r3_hbm profiles/r3/hbm

Kernel names are cut at the argument list so the tables stay readable."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles"))
from reduce_r1 import stats  # noqa: E402


def summary(raw, subs, dst):
    subs = [s for s in subs if os.path.isdir(os.path.join(raw, s))]
    if not subs:
        return
    with open(dst, "w") as f:
        for sub in subs:
            f.write(subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "summarise_pmc.py"), os.path.join(raw, sub)],
                                   check=True, stdout=subprocess.PIPE, text=True).stdout)


def main(tag, out):
    raw = os.path.join(ROOT, "gpurun_out", tag)
    out = os.path.join(ROOT, out)
    os.makedirs(out, exist_ok=True)
    lines = open(os.path.join(raw, "bench.json")).read().strip().splitlines()
    open(os.path.join(out, "bench.json"), "w").write(lines[-1] + "\n")
    stats(os.path.join(raw, "stats", "stats_kernel_stats.csv"), os.path.join(out, "kernel_stats.csv"))
    summary(raw, ("cache", "fetch", "write"), os.path.join(out, "pmc_traffic_cache_summary.txt"))
    summary(raw, ("sq1", "sq2", "mem"), os.path.join(out, "pmc_sq_summary.txt"))
    tj = os.path.join(raw, tag + "_traffic.json")
    if os.path.exists(tj):
        shutil.copy(tj, os.path.join(ROOT, "profiles", tag + "_traffic.json"))      # made on the box, before the bench line


def extra(out):
    """The bench lines and kernel stats of collect_r3_extra.sh."""
    raw = os.path.join(ROOT, "gpurun_out", "r3_extra")
    out = os.path.join(ROOT, out)
    for tag, name in (("devbuild", "device_build"),):
        src = os.path.join(raw, tag, tag + "_kernel_stats.csv")
        if os.path.exists(src):
            stats(src, os.path.join(out, "kernel_stats_%s.csv" % name))
    for name in ("devbuild_bench.json", "res8k_bench.json", "bounce2_bench.json", "config2.json", "tile_share_4k_8k.txt"):
        if os.path.exists(os.path.join(raw, name)):
            if name.endswith(".txt"):
                shutil.copy(os.path.join(raw, name), os.path.join(out, name))
                continue
            lines = open(os.path.join(raw, name)).read().strip().splitlines()
            open(os.path.join(out, name), "w").write(lines[-1] + "\n")


if __name__ == "__main__":
    if sys.argv[1] == "extra":
        extra(sys.argv[2])
    else:
        main(sys.argv[1], sys.argv[2])
