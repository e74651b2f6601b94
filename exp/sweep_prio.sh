# swap thresholds and the inner loop's letting-go ratio again, on the kernels with issue priorities
for r in 24 28 36 40; do
  RT_REFILL_MIN=$r timeout -k 10 120 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 > gpurun_out/sp.json 2> gpurun_out/sp.err || { echo failed; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/sp.json'));r=d['roofline'];print('refill $r', round(d['ms_per_step'],3),{k:round(v,3) for k,v in r['stage_ms_per_frame'].items()})"
done
bash exp/bench_variants.sh base rel3 rel5 rel6
