set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -x -q --timeout 300 --timeout-method thread -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -60 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
timeout -k 10 600 python -u bench.py --steps 10 --warmup 2 --cpu-seconds 0 > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err || { tail -30 gpurun_out/bench_b.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/bench_b.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"
