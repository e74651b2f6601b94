# usage: bash exp/build_variant_all.sh NAME [-DFLAG...]: like build_variant.sh, with the flags applied to the host side (rt_capi.cpp) too -
# for constants of rt_device.h that both sides see
N=$1; shift
B=raytracer_amd/build
F="-O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Iinclude -Iraytracer_amd/csrc -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" --offload-arch=gfx950 -c raytracer_amd/csrc/rt_kernels.hip -o /tmp/rt_kernels_$N.o || exit 1
/opt/rocm/bin/hipcc $F "$@" -c raytracer_amd/csrc/rt_capi.cpp -o /tmp/rt_capi_$N.o || exit 1
OBJS=$(ls $B/*.o | grep -v "rt_kernels_hip.o\|rt_capi_cpp.o")
/opt/rocm/bin/hipcc -shared -o exp/librt_hip_$N.so --offload-arch=gfx950 /tmp/rt_kernels_$N.o /tmp/rt_capi_$N.o $OBJS -ldl && echo exp/librt_hip_$N.so
