# usage: run_args.sh TAG "bench args" [TAG "bench args" ...]
while [ $# -ge 2 ]; do
TAG=$1; ARGS=$2; shift 2
timeout -k 10 150 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 $ARGS > gpurun_out/b_$TAG.json 2> gpurun_out/b_$TAG.err || { echo "bench $TAG failed"; tail -3 gpurun_out/b_$TAG.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/b_$TAG.json'));r=d['roofline'];print('$TAG', d['ms_per_step'],r['stage_ms_per_frame'],r['nodes_per_primary_ray'],r['prims_per_primary_ray'],r['nodes_per_shadow_ray'],r['prims_per_shadow_ray'])"
done
