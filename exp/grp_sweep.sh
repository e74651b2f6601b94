# group-granular refill: variants x swap thresholds
for V in "$@"; do
for T in 16 24 32 40; do
RT_HIP_LIB=exp/librt_hip_$V.so RT_REFILL_MIN=$T timeout -k 10 120 python bench.py --steps 8 --warmup 2 --cpu-seconds 0 > gpurun_out/b_${V}_$T.json 2> gpurun_out/b_${V}_$T.err || { echo "bench $V T=$T failed"; tail -3 gpurun_out/b_${V}_$T.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/b_${V}_$T.json'));r=d['roofline'];print('$V', $T, round(d['ms_per_step'],3),{k:round(v,3) for k,v in r['stage_ms_per_frame'].items()})"
done
done
