# closest-hit launch at several swap thresholds (RT_REFILL_MIN_PRIMARY), plain frame and two bounces
for T in 28 32 36 40 44 48; do
  for A in "" "--max-scatters 2"; do
  RT_REFILL_MIN_PRIMARY=$T timeout -k 10 120 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 $A > gpurun_out/rp.json 2> gpurun_out/rp.err || { echo failed; tail -3 gpurun_out/rp.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/rp.json'));r=d['roofline'];print('primary refill $T $A', round(d['ms_per_step'],3),{k:round(v,3) for k,v in r['stage_ms_per_frame'].items()})"
  done
done
