#!/bin/bash
# SQ-side counter passes for the ray kernels (issue mix, lane utilisation, waits); run from the repository root on the GPU box.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-sq}
mkdir -p $OUT
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq1 -o sq1 -- python3 -u bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/sq1_bench.json 2> $OUT/sq1.err || { echo "sq1 failed"; tail -5 $OUT/sq1.err; exit 1; }
echo sq1 done
timeout -s KILL 240 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq2 -o sq2 -- python3 -u bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/sq2_bench.json 2> $OUT/sq2.err || { echo "sq2 failed"; tail -5 $OUT/sq2.err; exit 1; }
echo sq2 done
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum --output-format csv -d $OUT/mem -o mem -- python3 -u bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/mem_bench.json 2> $OUT/mem.err || { echo "mem failed"; tail -5 $OUT/mem.err; exit 1; }
echo mem done
