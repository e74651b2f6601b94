# usage: bash exp/bench_b2.sh VARIANT...: the 1 M-cell frame with two mirror bounces, and the 100 M-triangle frame, per variant library
for V in "$@"; do
  if [ "$V" = base ]; then LIBV=""; else LIBV="exp/librt_hip_$V.so"; fi
  for A in "--max-scatters 2" "--instanced --max-scatters 2"; do
    RT_HIP_LIB=$LIBV timeout -k 10 300 python bench.py --steps 8 --warmup 2 --cpu-seconds 0 $A > gpurun_out/b3.json 2> gpurun_out/b3.err || { echo "bench $V $A failed"; tail -3 gpurun_out/b3.err; exit 1; }
    python -c "
import json;d=json.load(open('gpurun_out/b3.json'));r=d['roofline'];print('$V [$A]', round(d['ms_per_step'],3),{k:round(v,3) for k,v in r['stage_ms_per_frame'].items()})"
  done
done
