"""Build librt_hip.so (HIP kernels for gfx950 + host-side scene code + the C ABI) in-tree.

    python -m raytracer_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off everywhere: the FP64 intersectors and
the host-side precomputation must round exactly like the reference's plain x86-64 build.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "librt_hip.so")
HOSTLIB = os.path.join(LIBDIR, "librt_host.so")

ARCH = "gfx950"
# RT_EXTRA_FLAGS: extra compiler flags for experiments (e.g. -DRT_RAYS_MIN_WAVES=5 -DRT_STACK_LDS=12)
COMMON = ["-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
          "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wall", "-Wno-unused-function"] + os.environ.get("RT_EXTRA_FLAGS", "").split()
DEVICE_SOURCES = ["rt_kernels.hip", "rt_gpu_build.hip", "rt_probe.hip"]
HOST_SOURCES = ["rt_capi.cpp", "rt_comm.cpp", "rt_scene.cpp", "rt_prep.cpp", "rt_bvh.cpp", "rt_json.cpp", "rt_image.cpp"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    hs = [os.path.join(ROOT, "include", "rt_hip.h")]
    for base, _, files in os.walk(CSRC):
        hs += [os.path.join(base, f) for f in files if f.endswith(".h")]
    return hs


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build step failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def build(force=False, verbose=False):
    cc = hipcc()
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    headers = _headers()
    objs = []
    for src in DEVICE_SOURCES + HOST_SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJDIR, src.replace(".", "_") + ".o")
        objs.append(obj)
        if force or _stale(obj, [path] + headers):
            cmd = [cc] + COMMON + ["-c", path, "-o", obj]
            if src.endswith(".hip"):
                cmd += ["--offload-arch=" + ARCH]
            if verbose:
                print(" ".join(cmd))
            _run(cmd)
    if force or _stale(LIB, objs):
        _run([cc, "-shared", "-o", LIB, "--offload-arch=" + ARCH] + objs + ["-ldl"])
    # C++ mirror of the reference's class API + its two command-line drivers, on top of the C ABI
    host_dir = os.path.join(CSRC, "host")
    if os.path.isdir(host_dir):
        srcs = sorted(os.path.join(host_dir, f) for f in os.listdir(host_dir) if f.endswith(".cpp") and not f.endswith("_driver.cpp"))
        if srcs and (force or _stale(HOSTLIB, srcs + headers + [LIB])):
            _run([cc] + COMMON + ["-shared", "-o", HOSTLIB] + srcs + ["-I" + host_dir, "-L" + LIBDIR, "-lrt_hip", "-Wl,-rpath,$ORIGIN"])
        for drv, name in (("render_driver", "render_driver"), ("time_driver", "time_driver"), ("mirror_check_driver", "mirror_check")):
            src = os.path.join(host_dir, drv + ".cpp")
            exe = os.path.join(LIBDIR, name)
            if os.path.exists(src) and (force or _stale(exe, [src, HOSTLIB] + headers)):
                _run([cc] + COMMON + ["-o", exe, src, "-I" + host_dir, "-L" + LIBDIR, "-lrt_host", "-lrt_hip", "-Wl,-rpath,$ORIGIN"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
