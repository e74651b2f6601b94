# usage: bash exp/pmc_l1.sh VARIANT [bench args]: L1 look-ups and L1->L2 requests per launch of the ray kernels (one PMC pass, 2 frames)
export TMPDIR=/tmp
V=$1; shift
if [ "$V" = base ]; then export RT_HIP_LIB=""; else export RT_HIP_LIB="exp/librt_hip_$V.so"; fi
OUT=gpurun_out/pmc_$V
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum --output-format csv -d $OUT -o l1 -- python3 -u bench.py --cpu-seconds 0 --steps 2 --warmup 1 "$@" > $OUT/bench.json 2> $OUT/err.txt || { echo "pmc pass failed"; tail -5 $OUT/err.txt; exit 1; }
python3 - <<PY
import csv, collections, json, glob
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
v = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith("k_rays") and ", false," in k[:16]: v[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])["config"]
for k in sorted(v):
    rays = d["shadow_rays"] if k.startswith("k_rays<1") else d["primary_rays"]
    print("$V", k, {c: round(sum(x)/len(x)/rays, 2) for c, x in v[k].items()}, "per ray")
PY
