"""Build libupsp_gpu.so (HIP kernels + C ABI) and the pybind11 `raycast` module for gfx950.

    python -m upsp_processing_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical
contract (the watertight triangle test falls back to double precision when an
edge function is exactly 0.0f -- cpp/raycast/pspRT.cpp:133 in the reference --
so no FMA contraction may change which rays take that path).
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libupsp_gpu.so")
ARCH = "gfx950"

HIP_SOURCES = ["raycast.hip", "frames.hip", "imageops.hip", "ecc.hip", "pipeline.hip", "ktimer.hip", "video.hip", "phase2.hip", "geom.hip", "feed.hip",
               "exchange.hip"]
CXX_SOURCES = ["bvh_build.cpp"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
          "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_lib(force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "upsp_gpu.h"))
    objs = []
    for src in HIP_SOURCES + CXX_SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + headers):
            cmd = [HIPCC] + COMMON + ["--offload-arch=" + ARCH, "-c", sp, "-o", obj]
            if src.endswith(".cpp"):
                cmd.insert(1, "-x")
                cmd.insert(2, "hip")
            _run(cmd)
    if force or _stale(LIB, objs):
        _run([HIPCC, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs + ["-ldl"])   # (RCCL: dlopen, see exchange.hip)
    return LIB


def build_pybind(force=False):
    import pybind11
    src = os.path.join(CSRC, "pybind_raycast.cpp")
    if not os.path.exists(src):
        return None
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    out = os.path.join(HERE, "raycast" + ext)
    if force or _stale(out, [src, LIB, os.path.join(ROOT, "include", "upsp_gpu.h")]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
              "-I" + os.path.join(ROOT, "include"), "-I" + pybind11.get_include(),
              "-I" + sysconfig.get_paths()["include"], src, "-o", out,
              "-L" + LIBDIR, "-lupsp_gpu", "-Wl,-rpath,$ORIGIN/lib"])
    return out


def build_cli(force=False):
    """bin/psp_process_cpp: the C++ phase-1 driver (csrc/psp_process_main.cpp) over the C ABI -- no Python, no torch."""
    src = os.path.join(CSRC, "psp_process_main.cpp")
    out = os.path.join(ROOT, "bin", "psp_process_cpp")
    if force or _stale(out, [src, LIB, os.path.join(ROOT, "include", "upsp_gpu.h")]):
        _run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"),
              "-I/opt/rocm/include", src, "-o", out, "-L" + LIBDIR, "-lupsp_gpu", "-L/opt/rocm/lib", "-lamdhip64",
              "-Wl,-rpath,$ORIGIN/../upsp_processing_amd/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return out


def build_all(force=False):
    lib = build_lib(force)
    mod = build_pybind(force)
    build_cli(force)
    return lib, mod


if __name__ == "__main__":
    print(build_all("--force" in sys.argv))
