"""Build a variant of librt_hip.so for an experiment: rt_kernels.hip (and optionally other sources) recompiled with
extra flags, everything else taken from the regular build.

    python profiles/variant.py NAME [-DFLAG ...] [--also rt_capi.cpp]

writes exp/librt_hip_NAME.so; run with RT_HIP_LIB=exp/librt_hip_NAME.so (raytracer_amd/capi.py honours it).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import build as B  # noqa: E402


def main():
    name = sys.argv[1]
    flags, also = [], []
    it = iter(sys.argv[2:])
    for a in it:
        if a == "--also":
            also.append(next(it))
        else:
            flags.append(a)
    B.build()
    cc = B.hipcc()
    outdir = os.path.join(ROOT, "exp")
    os.makedirs(outdir, exist_ok=True)
    objs = []
    for src in B.DEVICE_SOURCES + B.HOST_SOURCES:
        obj = os.path.join(B.OBJDIR, src.replace(".", "_") + ".o")
        if src == "rt_kernels.hip" or src in also:
            obj = os.path.join(outdir, name + "_" + src.replace(".", "_") + ".o")
            cmd = [cc] + B.COMMON + flags + ["-c", os.path.join(B.CSRC, src), "-o", obj]
            if src.endswith(".hip"):
                cmd += ["--offload-arch=" + B.ARCH]
            subprocess.check_call(cmd)
        objs.append(obj)
    lib = os.path.join(outdir, "librt_hip_%s.so" % name)
    subprocess.check_call([cc, "-shared", "-o", lib, "--offload-arch=" + B.ARCH] + objs + ["-ldl"])
    print(lib)


if __name__ == "__main__":
    main()
