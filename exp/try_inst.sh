# usage: bash exp/try_inst.sh VARIANT...   (base = the regular library): two-level scenes - parity tests, then the bench line of the
# 100 M-triangle frame with two bounces
for V in "$@"; do
  if [ "$V" = base ]; then LIBV=""; else LIBV="exp/librt_hip_$V.so"; fi
  RT_HIP_LIB=$LIBV timeout -k 10 500 python -u -m pytest tests/test_instancing.py tests/test_gpu_large.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/ti_$V.log 2>&1 || { echo "tests $V FAILED"; tail -15 gpurun_out/ti_$V.log; exit 1; }
  tail -1 gpurun_out/ti_$V.log
  RT_HIP_LIB=$LIBV timeout -k 10 200 python bench.py --instanced --max-scatters 2 --steps 8 --warmup 2 --cpu-seconds 0 > gpurun_out/bi_$V.json 2> gpurun_out/bi_$V.err || { echo "bench $V failed"; tail -3 gpurun_out/bi_$V.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/bi_$V.json'));r=d['roofline'];print('$V', round(d['ms_per_step'],3),{k:round(v,3) for k,v in r['stage_ms_per_frame'].items()},round(r['nodes_per_primary_ray'],2),round(r['nodes_per_shadow_ray'],2))"
done
