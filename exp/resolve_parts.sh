export TMPDIR=/tmp
for V in "$@"; do
  if [ "$V" = base ]; then LIBV=""; else LIBV="exp/librt_hip_$V.so"; fi
  RT_HIP_LIB=$LIBV timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rp_$V -o rp -- python3 -u bench.py --cpu-seconds 0 --steps 5 --warmup 1 > gpurun_out/rp_$V.json 2> gpurun_out/rp_$V.err || { echo "failed $V"; exit 1; }
  echo "== $V"; grep -i "k_resolve" gpurun_out/rp_$V/rp_kernel_stats.csv | cut -c1-20,150-200
done
