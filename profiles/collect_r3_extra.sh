#!/bin/bash
# Bench lines and kernel stats of the workloads beside the headline one, run on the GPU box from the repository root:
#   bash profiles/collect_r3_extra.sh
# (a) the headline scene with the hierarchy built on the GPU, (b) the 7680x4320 frame, (c) the headline scene with two mirror
# bounces, (d) config 2 (100k triangles, primary rays), (e) one rank's share of the tiled frame for 1, 2, 4, 8 ranks at
# 3840x2160 and 7680x4320, whole and by stage.  (The 100 M-triangle two-level scene: collect_r3.sh r3_inst --instanced --max-scatters 2.)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r3_extra
mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/devbuild -o devbuild -- python3 -u bench.py --device-build --steps 10 --warmup 2 > $OUT/devbuild_bench.json 2> $OUT/devbuild.err || { echo "device-build pass failed"; tail -5 $OUT/devbuild.err; exit 1; }
echo "device build done"
timeout -k 10 300 python3 -u bench.py --res 7680x4320 --steps 10 --warmup 2 --cpu-seconds 0 > $OUT/res8k_bench.json 2> $OUT/res8k.err || { echo "8K failed"; tail -5 $OUT/res8k.err; exit 1; }
echo "8K done"
timeout -k 10 300 python3 -u bench.py --max-scatters 2 --steps 10 --warmup 2 --cpu-seconds 0 > $OUT/bounce2_bench.json 2> $OUT/bounce2.err || { echo "bounce failed"; tail -5 $OUT/bounce2.err; exit 1; }
echo "two bounces done"
PYTHONPATH=. timeout -k 10 300 python3 -u profiles/config2.py > $OUT/config2.json 2> $OUT/config2.err || { echo "config 2 failed"; tail -5 $OUT/config2.err; exit 1; }
echo "config 2 done"
: > $OUT/tile_share_4k_8k.txt
for R in 3840x2160 7680x4320; do
  echo "== $R" >> $OUT/tile_share_4k_8k.txt
  PYTHONPATH=. timeout -k 10 300 python3 -u profiles/tile_share.py $R >> $OUT/tile_share_4k_8k.txt 2> $OUT/share.err || { echo "tile share failed"; tail -5 $OUT/share.err; exit 1; }
  PYTHONPATH=. timeout -k 10 300 python3 -u profiles/tile_share_stages.py $R >> $OUT/tile_share_4k_8k.txt 2>> $OUT/share.err || { echo "tile share stages failed"; tail -5 $OUT/share.err; exit 1; }
done
echo "tile shares done"
