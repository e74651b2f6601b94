RT_WAVES_PER_CU=20 RT_HIP_LIB=exp/librt_hip_w5.so timeout -k 10 120 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 > gpurun_out/b_w5.json 2> gpurun_out/b_w5.err || { tail -3 gpurun_out/b_w5.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/b_w5.json'));r=d['roofline'];print('w5/20', round(d['ms_per_step'],3),{k:round(v,3) for k,v in r['stage_ms_per_frame'].items()})"
RT_WAVES_PER_CU=16 RT_HIP_LIB=exp/librt_hip_w5.so timeout -k 10 120 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 > gpurun_out/b_w5.json 2> gpurun_out/b_w5.err || { tail -3 gpurun_out/b_w5.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/b_w5.json'));r=d['roofline'];print('w5/16', round(d['ms_per_step'],3),{k:round(v,3) for k,v in r['stage_ms_per_frame'].items()})"
