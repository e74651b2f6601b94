#!/bin/bash
# Measurement recipe, run on the GPU box from the repository root:
#   bash profiles/collect_r1.sh
# bench line, rocprofv3 kernel stats of the same command, and two PMC passes (read and write side) for HBM traffic.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r1
mkdir -p $OUT
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 -u bench.py --steps 10 --warmup 2 --cpu-seconds 0 > $OUT/stats_bench.json 2> $OUT/stats.err || { echo "stats pass failed"; tail -5 $OUT/stats.err; exit 1; }
echo "stats done"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 -u bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/fetch_bench.json 2> $OUT/fetch.err || { echo "fetch pass failed"; tail -5 $OUT/fetch.err; exit 1; }
echo "fetch done"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 -u bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/write_bench.json 2> $OUT/write.err || { echo "write pass failed"; tail -5 $OUT/write.err; exit 1; }
echo "write done"
timeout -s KILL 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/cache -o cache -- python3 -u bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/cache_bench.json 2> $OUT/cache.err || { echo "cache pass failed"; tail -5 $OUT/cache.err; exit 1; }
echo "cache done"
# HBM-side bytes per launch from the two passes above, then the bench line that reports them as roofline.traffic
python3 profiles/make_traffic.py $OUT $OUT/r1_traffic.json > /dev/null && cp $OUT/r1_traffic.json profiles/r1_traffic.json || { echo "traffic reduction failed"; exit 1; }
timeout -k 10 300 python3 -u bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
echo "bench done"
find $OUT -name "*.csv" | head -30
