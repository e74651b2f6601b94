for V in base it2 it8; do
  if [ "$V" = base ]; then LIBV=""; else LIBV="exp/librt_hip_$V.so"; fi
  for E in 0 1; do
    if [ $E = 1 ]; then export RT_ENTRY_ONE_LANE=1; else unset RT_ENTRY_ONE_LANE; fi
    RT_HIP_LIB=$LIBV timeout -k 10 120 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 > gpurun_out/b_ei.json 2> gpurun_out/b_ei.err || { echo "bench failed"; exit 1; }
    python -c "
import json;d=json.load(open('gpurun_out/b_ei.json'));r=d['roofline'];print('$V one_lane=$E', round(d['ms_per_step'],3),{k:round(v,3) for k,v in r['stage_ms_per_frame'].items()},round(r['nodes_per_primary_ray'],2),round(r['nodes_per_shadow_ray'],2))"
  done
done
