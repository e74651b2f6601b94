# usage: bash exp/phase2.sh VARIANT...: where a shadow wave's cycles go (libraries built with -DRT_PHASE_TIMING)
for V in "$@"; do
  echo "== $V"
  PYTHONPATH=. RT_HIP_LIB=exp/librt_hip_$V.so timeout -k 10 200 python profiles/phase_split.py 2>&1 | tail -3 || exit 1
done
