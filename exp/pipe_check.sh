# variant library against the parity suite, then the bench beside the regular build
V=${1:-pipe}
RT_HIP_LIB=exp/librt_hip_$V.so timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_instancing.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/parity_$V.log 2>&1 || { tail -20 gpurun_out/parity_$V.log; exit 1; }
tail -2 gpurun_out/parity_$V.log
bash exp/run_variants.sh sort5 $V
