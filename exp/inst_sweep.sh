# usage: bash exp/inst_sweep.sh   (on the GPU box, from the repository root)
# swap thresholds of the shadow and of the closest-hit launches on the two-level scene (100 M instanced triangles, two bounces)
run() {
  env "$@" timeout -k 10 200 python bench.py --instanced --max-scatters 2 --steps 6 --warmup 2 --cpu-seconds 0 > gpurun_out/inst_sw.json 2> gpurun_out/inst_sw.err || { echo "failed: $*"; tail -3 gpurun_out/inst_sw.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/inst_sw.json'));r=d['roofline'];print('$*', round(d['ms_per_step'],3),{k:round(v,3) for k,v in r['stage_ms_per_frame'].items()})"
}
run RT_NONE=1
for s in 32 40 56; do run RT_REFILL_MIN=$s RT_REFILL_MIN_PRIMARY=48; done
for p in 32 40 56; do run RT_REFILL_MIN=48 RT_REFILL_MIN_PRIMARY=$p; done
