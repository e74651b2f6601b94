for F in 0 1 2 4 8; do
RT_HIP_LIB=exp/librt_hip_fine$F.so PYTHONPATH=. timeout -k 10 100 python exp/fine_sweep.py 2>&1 | grep "sum" || exit 1
done
