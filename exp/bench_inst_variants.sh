# usage: bash exp/bench_inst_variants.sh VARIANT...: bench line of the two-level scene (100 M instanced triangles, two bounces) per variant library
for V in "$@"; do
  if [ "$V" = base ]; then LIBV=""; else LIBV="exp/librt_hip_$V.so"; fi
  RT_HIP_LIB=$LIBV timeout -k 10 200 python bench.py --instanced --max-scatters 2 --steps 8 --warmup 2 --cpu-seconds 0 > gpurun_out/bi_$V.json 2> gpurun_out/bi_$V.err || { echo "bench $V failed"; tail -3 gpurun_out/bi_$V.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/bi_$V.json'));r=d['roofline'];print('$V', round(d['ms_per_step'],3),{k:round(v,3) for k,v in r['stage_ms_per_frame'].items()})"
done
