# usage: bash exp/refill_sweep.sh VARIANT THRESHOLD...: bench stage times at several swap thresholds (RT_REFILL_MIN)
V=$1; shift
if [ "$V" = base ]; then LIBV=""; else LIBV="exp/librt_hip_$V.so"; fi
for T in "$@"; do
  RT_REFILL_MIN=$T RT_HIP_LIB=$LIBV timeout -k 10 120 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 > gpurun_out/rf_${V}_$T.json 2> gpurun_out/rf_${V}_$T.err || { echo "bench $V $T failed"; tail -3 gpurun_out/rf_${V}_$T.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/rf_${V}_$T.json'));r=d['roofline'];print('$V refill $T', round(d['ms_per_step'],3),{k:round(v,3) for k,v in r['stage_ms_per_frame'].items()})"
done
