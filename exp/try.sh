# usage: bash exp/try.sh VARIANT...   (base = the regular library)
# quick parity subset with each variant library, then its bench line (stage times, boxes per ray)
for V in "$@"; do
  if [ "$V" = base ]; then LIBV=""; else LIBV="exp/librt_hip_$V.so"; fi
  RT_HIP_LIB=$LIBV timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/t_$V.log 2>&1 || { echo "tests $V FAILED"; tail -15 gpurun_out/t_$V.log; exit 1; }
  tail -1 gpurun_out/t_$V.log
  RT_HIP_LIB=$LIBV timeout -k 10 120 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 > gpurun_out/b_$V.json 2> gpurun_out/b_$V.err || { echo "bench $V failed"; tail -3 gpurun_out/b_$V.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/b_$V.json'));r=d['roofline'];print('$V', round(d['ms_per_step'],3),{k:round(v,3) for k,v in r['stage_ms_per_frame'].items()},round(r['nodes_per_primary_ray'],2),round(r['nodes_per_shadow_ray'],2))"
done
