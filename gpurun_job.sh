set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -x -v --timeout 300 --timeout-method thread -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -60 gpurun_out/gpu_tests.log; exit 1; }
tail -45 gpurun_out/gpu_tests.log
