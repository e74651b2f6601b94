# usage: bash exp/pmc_hbm.sh VARIANT [bench args]: FETCH_SIZE and WRITE_SIZE per launch of every kernel (two PMC passes, 2 frames each)
export TMPDIR=/tmp
V=$1; shift
if [ "$V" = base ]; then export RT_HIP_LIB=""; else export RT_HIP_LIB="exp/librt_hip_$V.so"; fi
OUT=gpurun_out/hbm_$V
rm -rf $OUT; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o c -- python3 -u bench.py --cpu-seconds 0 --steps 2 --warmup 1 "$@" > $OUT/bench_$C.json 2> $OUT/err_$C.txt || { echo "pmc pass $C failed"; tail -5 $OUT/err_$C.txt; exit 1; }
done
python3 - <<PY
import csv, collections, glob
v = collections.defaultdict(lambda: collections.defaultdict(list))
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % C, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if r["Counter_Name"] == C and not k.startswith("__amd") and "probe" not in k and k: v[k][C].append(float(r["Counter_Value"]))
for k in sorted(v):
    print("$V %-40s" % k, {c: "%.1f MB (n=%d)" % (sum(x)/len(x)*1024/1e6 * (2 if c == "FETCH_SIZE" else 1), len(x)) for c, x in v[k].items()})
PY
