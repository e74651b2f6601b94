# usage: bash exp/bench_variants.sh VARIANT...   (base = the regular library): the bench line's stage times only, no tests
for V in "$@"; do
  if [ "$V" = base ]; then LIBV=""; else LIBV="exp/librt_hip_$V.so"; fi
  RT_HIP_LIB=$LIBV timeout -k 10 120 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 > gpurun_out/b_$V.json 2> gpurun_out/b_$V.err || { echo "bench $V failed"; tail -3 gpurun_out/b_$V.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/b_$V.json'));r=d['roofline'];print('$V', round(d['ms_per_step'],3),{k:round(v,3) for k,v in r['stage_ms_per_frame'].items()},round(r['nodes_per_primary_ray'],2),round(r['nodes_per_shadow_ray'],2))"
done
