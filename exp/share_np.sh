# usage: bash exp/share_np.sh: one rank's share (1 of 8, 4K) with persistent waves (default) against one wave per group of 64 rays
run() { echo "== $*"; env "$@" PYTHONPATH=. timeout -k 10 200 python profiles/tile_share_stages.py 2>&1 | grep "^ranks [18]"; }
run RT_X=0
run RT_HIP_LIB=exp/librt_hip_chunk64.so RT_WAVE_DIV=1 RT_MAX_WAVES=1000000 RT_REFILL_MIN=64
run RT_HIP_LIB=exp/librt_hip_chunk64.so RT_WAVE_DIV=2 RT_MAX_WAVES=1000000 RT_REFILL_MIN=48
run RT_HIP_LIB=exp/librt_hip_chunk64.so RT_WAVE_DIV=1 RT_MAX_WAVES=1000000 RT_REFILL_MIN=32
