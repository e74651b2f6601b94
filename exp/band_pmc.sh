set -u
export TMPDIR=/tmp
OUT=gpurun_out/band
mkdir -p $OUT
for R in 2160 256; do
pass() { local name=$1; shift
  PYTHONPATH=. timeout -s KILL 200 rocprofv3 "$@" --output-format csv -d $OUT/${name}_$R -o ${name}_$R -- python3 -u exp/band_prog.py $R 4 > $OUT/${name}_$R.out 2> $OUT/${name}_$R.err || { echo "$name $R failed"; tail -5 $OUT/${name}_$R.err; exit 1; }
  echo "$name $R done"; }
pass sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD
pass sq2 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INST_CYCLES_VMEM
pass cache --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass mem --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
done
