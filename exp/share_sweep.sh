for L in exp/librt_hip_c512.so exp/librt_hip_c1024.so; do for D in 8 16; do
RT_HIP_LIB=$L RT_WAVE_DIV=$D PYTHONPATH=. timeout -k 10 100 python exp/share_sweep.py 2>&1 | grep ranks || exit 1
done; done
