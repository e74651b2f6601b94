# experiment: shadow launch with its rays set up by a dense pre-pass (exp/librt_hip_prep.so), at several swap thresholds;
# kernel times from rocprofv3 (the pre-pass is a kernel of its own)
export TMPDIR=/tmp
for T in 8 16 24 32; do
  OUT=gpurun_out/prep_$T; rm -rf $OUT; mkdir -p $OUT
  RT_REFILL_MIN=$T RT_HIP_LIB=exp/librt_hip_prep.so timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o st -- python3 -u bench.py --cpu-seconds 0 --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/err.txt || { echo "failed"; tail -5 $OUT/err.txt; exit 1; }
  python3 - <<PY
import csv
rows=list(csv.reader(open("$OUT/st_kernel_stats.csv")))
for r in rows[1:12]:
    n=r[0].split("(")[0]
    if n.startswith("void k_rays<1, false") or "k_prep" in n or n.startswith("void k_rays<0, false"): print("refill $T  %-44s avg %9.1f us" % (n[:44], float(r[3])/1e3))
PY
done
