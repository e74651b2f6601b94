#!/bin/bash
# Measurement recipe behind profiles/r3 (and profiles/r3/hbm), run on the GPU box from the repository root:
#   bash profiles/collect_r3.sh r3                       the headline workload (1M cells, 3840x2160, 4 lights)
#   bash profiles/collect_r3.sh r3_hbm --n-side 4000     16M cells: nodes and records far beyond the 256 MiB Infinity Cache
#   bash profiles/collect_r3.sh r3_inst --instanced --max-scatters 2     100 M instanced triangles, two mirror bounces
# Kernel stats of the bench command (rocprofv3 --kernel-trace --stats), then one PMC pass per counter group, each in a run
# of its own (FETCH_SIZE, WRITE_SIZE, cache hits / L1 requests, issue mix and lane utilisation, waits, the L1 request
# path), then the HBM-side bytes per launch (make_traffic.py) and the bench line that reports them.
# RT_PASSES names the PMC passes to run (default: all).
set -u
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
B="python3 -u bench.py --cpu-seconds 0 $*"
PASSES=${RT_PASSES:-"stats fetch write cache sq1 sq2 mem"}
pass() {   # name, step count, rocprofv3 options...
    local name=$1 steps=$2; shift 2
    case " $PASSES " in *" $name "*) ;; *) return 0;; esac
    timeout -s KILL 500 rocprofv3 "$@" --output-format csv -d $OUT/$name -o $name -- $B --steps $steps --warmup 1 > $OUT/${name}_bench.json 2> $OUT/$name.err \
        || { echo "$name pass failed"; tail -5 $OUT/$name.err; exit 1; }
    echo "$name done"
}
pass stats 10 --kernel-trace --stats
pass fetch 2 --pmc FETCH_SIZE
pass write 2 --pmc WRITE_SIZE
pass cache 2 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass sq1 2 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD
pass sq2 2 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INST_CYCLES_VMEM
pass mem 2 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum
python3 profiles/make_traffic.py $OUT $OUT/${TAG}_traffic.json > /dev/null && cp $OUT/${TAG}_traffic.json profiles/${TAG}_traffic.json || { echo "traffic reduction failed"; exit 1; }
timeout -k 10 500 python3 -u bench.py --steps 20 --warmup 3 "$@" > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
echo "bench done"
