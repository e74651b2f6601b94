for L in 0 20000 26000 40000 80000; do
  RT_RESOLVE_LDS=$L timeout -k 10 120 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 > gpurun_out/b_ro.json 2> gpurun_out/b_ro.err || { echo "bench failed"; tail -3 gpurun_out/b_ro.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/b_ro.json'));r=d['roofline'];print('lds $L', round(d['ms_per_step'],3),{k:round(v,3) for k,v in r['stage_ms_per_frame'].items()})"
done
