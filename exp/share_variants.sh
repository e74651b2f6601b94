# usage: bash exp/share_variants.sh VARIANT...: stage times of 1..8 ranks' shares of the 4K frame per variant library
for V in "$@"; do
  if [ "$V" = base ]; then LIBV=""; else LIBV="exp/librt_hip_$V.so"; fi
  echo "== $V"
  PYTHONPATH=. RT_HIP_LIB=$LIBV timeout -k 10 300 python3 -u profiles/tile_share_stages.py 3840x2160 || exit 1
done
