for V in "$@"; do echo "== $V"; RT_HIP_LIB=exp/librt_hip_$V.so PYTHONPATH=. timeout -k 10 200 python profiles/tile_share.py 2>/dev/null | grep "tile 64" || exit 1; done
