#!/bin/bash
# Bench lines and kernel stats of the workloads beside the headline one, run on the GPU box from the repository root:
#   bash profiles/collect_r3_extra.sh
# (a) the 100 M-triangle two-level scene with two mirror bounces, (b) the headline scene with the hierarchy built on the
# GPU, (c) the 7680x4320 frame, (d) the headline scene with two mirror bounces, (e) config 2 (100k triangles, primary rays).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r3_extra
mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/inst -o inst -- python3 -u bench.py --instanced --max-scatters 2 --steps 10 --warmup 2 > $OUT/inst_bench.json 2> $OUT/inst.err || { echo "instanced pass failed"; tail -5 $OUT/inst.err; exit 1; }
echo "instanced done"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/devbuild -o devbuild -- python3 -u bench.py --device-build --steps 10 --warmup 2 > $OUT/devbuild_bench.json 2> $OUT/devbuild.err || { echo "device-build pass failed"; tail -5 $OUT/devbuild.err; exit 1; }
echo "device build done"
timeout -k 10 300 python3 -u bench.py --res 7680x4320 --steps 10 --warmup 2 --cpu-seconds 0 > $OUT/res8k_bench.json 2> $OUT/res8k.err || { echo "8K failed"; tail -5 $OUT/res8k.err; exit 1; }
echo "8K done"
timeout -k 10 300 python3 -u bench.py --max-scatters 2 --steps 10 --warmup 2 --cpu-seconds 0 > $OUT/bounce2_bench.json 2> $OUT/bounce2.err || { echo "bounce failed"; tail -5 $OUT/bounce2.err; exit 1; }
echo "two bounces done"
PYTHONPATH=. timeout -k 10 300 python3 -u profiles/config2.py > $OUT/config2.json 2> $OUT/config2.err || { echo "config 2 failed"; tail -5 $OUT/config2.err; exit 1; }
echo "config 2 done"
