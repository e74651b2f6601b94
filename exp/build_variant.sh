# usage: bash exp/build_variant.sh NAME [-DFLAG...]: exp/librt_hip_NAME.so = the library with rt_kernels.hip compiled with the flags
# (run in the build container after python -m raytracer_amd.build; RT_HIP_LIB=exp/librt_hip_NAME.so selects it)
N=$1; shift
B=raytracer_amd/build
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Iinclude -Iraytracer_amd/csrc -Wno-unused-function "$@" --offload-arch=gfx950 -c raytracer_amd/csrc/rt_kernels.hip -o /tmp/rt_kernels_$N.o || exit 1
OBJS=$(ls $B/*.o | grep -v rt_kernels_hip.o)
/opt/rocm/bin/hipcc -shared -o exp/librt_hip_$N.so --offload-arch=gfx950 /tmp/rt_kernels_$N.o $OBJS -ldl && echo exp/librt_hip_$N.so
