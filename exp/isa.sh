# usage: bash exp/isa.sh NAME [-DFLAG...]: device assembly of rt_kernels.hip with the flags -> /tmp/rt_NAME.s, register report of the ray kernels
N=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Iinclude -Iraytracer_amd/csrc "$@" --offload-arch=gfx950 --cuda-device-only -S raytracer_amd/csrc/rt_kernels.hip -o /tmp/rt_$N.s 2>/dev/null
python profiles/isa_report.py /tmp/rt_$N.s k_rays
