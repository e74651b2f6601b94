set -e
RT_HIP_LIB=exp/librt_hip_grp4.so timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_instancing.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/parity_grp4.log 2>&1 || { tail -20 gpurun_out/parity_grp4.log; exit 1; }
tail -2 gpurun_out/parity_grp4.log
bash exp/grp_sweep.sh grp4 grp2 grp8 grp16
