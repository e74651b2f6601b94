# usage: stats_py.sh TAG script.py [args]: rocprofv3 kernel stats of a python script
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/st_$TAG
mkdir -p $OUT
PYTHONPATH=. timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o st -- python3 -u "$@" > $OUT/out.txt 2> $OUT/err.txt || { echo "stats pass failed"; tail -5 $OUT/err.txt; exit 1; }
python3 - <<PY
import csv
rows=list(csv.reader(open("$OUT/st_kernel_stats.csv")))
for r in rows[1:14]:
    if "probe" in r[0]: continue
    print("%-60s calls %4s avg %10.1f us  min %10.1f max %10.1f" % (r[0].split("(")[0][:60], r[1], float(r[3])/1e3, float(r[5])/1e3, float(r[6])/1e3))
PY
