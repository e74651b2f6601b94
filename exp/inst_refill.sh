for T in 32 40 48; do
RT_REFILL_MIN=$T timeout -k 10 150 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --instanced --max-scatters 2 > gpurun_out/b_ir$T.json 2> gpurun_out/b_ir$T.err || { tail -3 gpurun_out/b_ir$T.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/b_ir$T.json'));r=d['roofline'];print('inst T=$T', round(d['ms_per_step'],3),{k:round(v,3) for k,v in r['stage_ms_per_frame'].items()})"
RT_REFILL_MIN=$T timeout -k 10 150 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --max-scatters 2 > gpurun_out/b_ir$T.json 2> gpurun_out/b_ir$T.err || { tail -3 gpurun_out/b_ir$T.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/b_ir$T.json'));r=d['roofline'];print('bounce2 T=$T', round(d['ms_per_step'],3),{k:round(v,3) for k,v in r['stage_ms_per_frame'].items()})"
done
