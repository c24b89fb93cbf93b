"""Builds ozimmu_amd/libozimmu_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m ozimmu_amd.build [--force]

The .so is git-ignored but travels with the tree (gpurun snapshot); nothing is installed or JIT-cached.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libozimmu_hip.so")
SOURCES = ["slice_gemm.hip", "split.hip", "convert.hip", "api.cpp", "interpose.cpp"]
HEADERS = ["kernels.h", "layout.h", "handle.h", "slice_gemm_kernel.h", "slice_gemm_w_kernel.h", os.path.join("..", "..", "include", "ozimmu_hip.h")]
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm: /opt/rocm/bin/hipcc)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    deps = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs = []
    common = ["-O3", "-std=c++17", "-fPIC", "-I" + CSRC, "-D__HIP_PLATFORM_AMD__", "-Wall",
              "-Wno-unused-function", "-Wno-unused-value"]
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(HERE, "build", s.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + deps):
            dev = ["-x", "hip", f"--offload-arch={ARCH}"] if s.endswith(".hip") else []
            cmd = [hipcc] + common + dev + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + ["-ldl", "-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
