# usage: bash exp/bench2.sh TAG: full GPU tests (minus the 16M-cell one), then the bench's stage times on the headline scene and on 16M cells
TAG=$1
mkdir -p gpurun_out/$TAG
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread --deselect tests/test_gpu_hbm.py > gpurun_out/$TAG/pytest.log 2>&1; tail -2 gpurun_out/$TAG/pytest.log
grep -q passed gpurun_out/$TAG/pytest.log && ! grep -q failed gpurun_out/$TAG/pytest.log || exit 1
for A in "" "--n-side 4000"; do
  timeout -k 10 400 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 $A > gpurun_out/$TAG/b.json 2> gpurun_out/$TAG/b.err || { echo "bench failed"; tail -3 gpurun_out/$TAG/b.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/$TAG/b.json'));r=d['roofline'];print('$TAG $A', round(d['ms_per_step'],3),{k:round(v,3) for k,v in r['stage_ms_per_frame'].items()},round(r['nodes_per_shadow_ray'],2),round(r['prims_per_shadow_ray'],3))"
done
