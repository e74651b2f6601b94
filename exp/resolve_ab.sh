# two-pass resolve against the FP64 single-pass build: frames byte for byte, then parity subset and bench line
PYTHONPATH=. RT_HIP_LIB=exp/librt_hip_fp64res.so timeout -k 10 300 python exp/resolve_diff.py gpurun_out/frames_fp64.npy || exit 1
PYTHONPATH=. timeout -k 10 300 python exp/resolve_diff.py gpurun_out/frames_new.npy gpurun_out/frames_fp64.npy || exit 1
rm -f gpurun_out/frames_fp64.npy gpurun_out/frames_new.npy
bash exp/try.sh base
