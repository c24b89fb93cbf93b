"""Builds the library in-tree with hipcc for gfx950 (cross-compiles without a GPU), in two flavours from the same sources:

    ozimmu_amd/libozimmu_hip.so        what ships: no test hook anywhere (objects in build/).  The bindings, bench.py, smoke(),
                                       the LD_PRELOAD tests and every parity test that needs no hook load THIS file.
    ozimmu_amd/libozimmu_hip_test.so   -DOZIMMU_HIP_TEST_HOOKS (objects in build_test/): the INT32 diagonal-sum dump of the
                                       kernels' epilogues, launch-failure injection, the epoch jump - loaded only by the
                                       tests that drive those hooks (tests/conftest.py: `ozh`).

    python -m ozimmu_amd.build [--force] [--no-test-flavour | --test-flavour-only]

The .so is git-ignored but travels with the tree (gpurun snapshot); nothing is installed or JIT-cached.
"""
import concurrent.futures
import hashlib
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libozimmu_hip.so")
LIB_TEST = os.path.join(HERE, "libozimmu_hip_test.so")
GEMM_PARTS = ["slice_gemm_s3_4.hip", "slice_gemm_s5_6.hip", "slice_gemm_s7_7.hip", "slice_gemm_s8_8.hip", "slice_gemm_s9_9.hip", "slice_gemm_s10_10.hip", "slice_gemm_s11_11.hip", "slice_gemm_s12_12.hip", "slice_gemm_s13_13.hip", "slice_gemm_s14_14.hip", "slice_gemm_s15_15.hip", "slice_gemm_s16_16.hip", "slice_gemm_s17_17.hip", "slice_gemm_s18_18.hip"]
SOURCES = GEMM_PARTS + ["slice_gemm.hip", "slice_gemm_one_launch.hip", "topology.hip", "split.hip", "convert.hip", "api.cpp", "config.cpp", "kernel_policy.cpp", "kernel_tuner.cpp", "interpose.cpp"]
HEADERS = ["kernels.h", "config.h", "topology.h", "tile_plan.h", "kernel_policy.h", "kernel_tuner.h", "layout.h", "handle.h", "slice_gemm_kernel.h", "slice_gemm_w_kernel.h", "slice_gemm_x_tile.h", "slice_gemm_y_tile.h", "slice_gemm_k2_kernel.h",
           "slice_gemm_launch.h", "split_resident.h", "one_launch.h", "diagnostics.h", os.path.join("..", "..", "include", "ozimmu_hip.h")]
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


_INCLUDE = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _closure(path, seen=None):
    """the file and every project header it includes, transitively: an object is stale when one of THESE is newer (a slice-GEMM
    unit takes minutes; a change of split_resident.h or api.cpp must not recompile all fourteen of them)"""
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path) as f:
        for inc in _INCLUDE.findall(f.read()):
            _closure(os.path.join(os.path.dirname(path), inc), seen)
    return seen


def device_asm_path(source):
    """where the build leaves the gfx950 assembly of a .hip source (read by tests/test_isa_invariants.py)"""
    return os.path.join(HERE, "build", source.rsplit(".", 1)[0] + "-hip-amdgcn-amd-amdhsa-gfx950.s")


def _compile(hipcc, common, s, verbose, bdir="build"):
    src = os.path.join(CSRC, s)
    stem = s.rsplit(".", 1)[0]
    obj = os.path.join(HERE, bdir, stem + ".o")
    # -save-temps=obj keeps the device assembly (device_asm_path) that the ISA test inspects, so that the test does not
    # compile the kernels a second time; the other intermediates are deleted again (tens of MB)
    dev = ["-x", "hip", f"--offload-arch={ARCH}", "-save-temps=obj"] if s.endswith(".hip") else []
    cmd = [hipcc] + common + dev + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    keep = {obj, os.path.join(HERE, bdir, os.path.basename(device_asm_path(s)))}
    for f in os.listdir(os.path.join(HERE, bdir)):
        path = os.path.join(HERE, bdir, f)
        if (f.startswith(stem + "-h") or f.startswith(stem + ".hip-")) and path not in keep:
            os.remove(path)
    return obj


def _flavour(hooks):
    bdir = "build_test" if hooks else "build"
    common = ["-O3", "-std=c++17", "-fPIC", "-I" + CSRC, "-D__HIP_PLATFORM_AMD__", "-Wall",
              "-Wno-unused-function", "-Wno-unused-value"] + (["-DOZIMMU_HIP_TEST_HOOKS"] if hooks else [])
    objs = [os.path.join(HERE, bdir, s.rsplit(".", 1)[0] + ".o") for s in SOURCES]
    return bdir, common, objs, (LIB_TEST if hooks else LIB)


def _stamp_path(lib):
    return lib + ".flags"


def _flags_stamp(common):
    """what a library was compiled with, besides its sources: the flag list (include path made relative), the architecture and
    the source list.  Stored next to the library (it travels to the GPU box with it); a library whose stamp differs is stale
    even when it is newer than every source (after a change of the flags or of ARCH a stale .so was reused silently
    unless --force was passed - and the GPU box never has objects to notice by)."""
    flags = [f.replace(CSRC, "csrc") for f in common]
    return hashlib.sha256("\n".join(flags + [ARCH] + SOURCES).encode()).hexdigest()


def _stamp_stale(lib, common):
    try:
        with open(_stamp_path(lib)) as f:
            return f.read().strip() != _flags_stamp(common)
    except OSError:
        return False  # a library from before the stamps: taken as built with these flags, stamped by this run


def build(force=False, verbose=False, test_flavour=True, product=True):
    """compiles the stale translation units of both flavours side by side (the slice-GEMM parts take minutes each), links
    libozimmu_hip.so (the product) and libozimmu_hip_test.so (with the test hooks); returns the product's path"""
    hipcc = _hipcc()
    deps = [os.path.join(CSRC, h) for h in HEADERS]  # (compiler flags and ARCH: the stamp file next to each library)
    flavours = [_flavour(h) for h in ([False] if product else []) + ([True] if test_flavour else [])]
    jobs = []
    all_src = [os.path.join(CSRC, s) for s in SOURCES] + deps
    # a library newer than every source is current even where its objects are absent (the GPU box: build/ and build_test/ -
    # objects and kept assembly, 200 MB - are listed in .gpurunignore and do not travel with the snapshot)
    # ... but where the object directory is present (a development box), a missing or outdated object means the library is too
    def _objects_stale(bdir, objs):
        return os.path.isdir(os.path.join(HERE, bdir)) and any(
            _stale(obj, sorted(_closure(os.path.join(CSRC, s)))) for s, obj in zip(SOURCES, objs))
    restamp = {f[3] for f in flavours if _stamp_stale(f[3], f[1])}  # other flags / architecture: everything of that flavour again
    flavours = [f for f in flavours if force or f[3] in restamp or _stale(f[3], all_src) or _objects_stale(f[0], f[2])]
    for bdir, common, objs, lib in flavours:
        os.makedirs(os.path.join(HERE, bdir), exist_ok=True)
        jobs += [(common, s, bdir) for s, obj in zip(SOURCES, objs)
                 if force or lib in restamp or _stale(obj, sorted(_closure(os.path.join(CSRC, s))))]
    jobs.sort(key=lambda j: j[1] not in GEMM_PARTS)  # the long ones first
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as pool:
        for f in [pool.submit(_compile, hipcc, common, s, verbose, bdir) for common, s, bdir in jobs]:
            f.result()
    for bdir, common, objs, lib in flavours:
        if force or lib in restamp or _stale(lib, objs):
            cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs + ["-ldl", "-lpthread"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(_stamp_path(lib), "w") as f:
                f.write(_flags_stamp(common) + "\n")
    for hooks in ([False] if product else []) + ([True] if test_flavour else []):
        _, common, _, lib = _flavour(hooks)
        if os.path.exists(lib) and not os.path.exists(_stamp_path(lib)):
            with open(_stamp_path(lib), "w") as f:
                f.write(_flags_stamp(common) + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, test_flavour="--no-test-flavour" not in sys.argv,
                product="--test-flavour-only" not in sys.argv))
