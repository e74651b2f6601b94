# four lanes per beam against one lane per beam in the entry search: parity subset, then the bench line either way
bash exp/try.sh base || exit 1
RT_ENTRY_ONE_LANE=1 timeout -k 10 120 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 > gpurun_out/b_one.json 2> gpurun_out/b_one.err || { echo "bench one-lane failed"; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/b_one.json'));r=d['roofline'];print('one lane', round(d['ms_per_step'],3),{k:round(v,3) for k,v in r['stage_ms_per_frame'].items()},round(r['nodes_per_primary_ray'],2),round(r['nodes_per_shadow_ray'],2))"
