#!/bin/bash
# Kernel stats of the two workloads beside the headline one, run on the GPU box from the repository root:
#   bash profiles/collect_r2_extra.sh
# (a) the 100 M-triangle two-level scene with two mirror bounces, (b) the headline scene with the hierarchy built on the GPU.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r2_extra
mkdir -p $OUT
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/inst -o inst -- python3 -u bench.py --instanced --max-scatters 2 --steps 10 --warmup 2 > $OUT/inst_bench.json 2> $OUT/inst.err || { echo "instanced pass failed"; tail -5 $OUT/inst.err; exit 1; }
echo "instanced done"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/devbuild -o devbuild -- python3 -u bench.py --device-build --steps 10 --warmup 2 > $OUT/devbuild_bench.json 2> $OUT/devbuild.err || { echo "device-build pass failed"; tail -5 $OUT/devbuild.err; exit 1; }
echo "device build done"
find $OUT -name "*kernel_stats.csv"
