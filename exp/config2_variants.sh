for V in "$@"; do
  if [ "$V" = base ]; then LIBV=""; else LIBV="exp/librt_hip_$V.so"; fi
  echo "== $V"
  PYTHONPATH=. RT_HIP_LIB=$LIBV timeout -k 10 300 python3 -u profiles/config2.py | cut -c1-260 || exit 1
done
