#!/usr/bin/env python
"""tools/ab_libs.py — the SAME calls through TWO builds of the library in one process, alternating legs (round 6: the round-5
library, built from commit 59b2ae3 in a worktree and copied to tools/bin/r5/libozimmu_hip.so, against the tree's).  Boxes of the
pool differ by +-5 % and a power-capped part drifts over seconds: only legs that alternate on one box in one run compare two
builds.  Each library is its own image with its own handle (ozimmu_amd's bindings executed twice, as test_flavour() does).

    python tools/ab_libs.py --old tools/bin/r5/libozimmu_hip.so [--shapes 8192 32768x32768x1024:NT 8192x8192x256 ...] [--legs 7] [--reps 6]

Shape syntax: M[xNxK][:OPS][:bBETA][:z] (z: complex).  Prints the median leg of both (DGEMM-equivalent TFLOP/s), new / old, the
kernels that ran, and whether the two results are bit-identical."""
import argparse
import importlib.util
import os
import sys
import time

os.environ.setdefault("OZIMMU_HIP_ENV_PER_CALL", "1")
os.environ.setdefault("OZIMMU_HIP_AUTOTUNE", "0")   # the model's pick on both sides: what a shape seen once runs
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def bindings(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "ozimmu_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.LIB_PATH = os.path.abspath(path)
    return mod


def parse(s):
    parts = s.split(":")
    dims = [int(x) for x in parts[0].split("x")]
    m, n, k = dims if len(dims) == 3 else (dims[0],) * 3
    ops, beta, cplx = "NN", 0.0, False
    for p in parts[1:]:
        if p == "z":
            cplx = True
        elif p.startswith("b"):
            beta = float(p[1:])
        else:
            ops = p.upper()
    return m, n, k, ops, beta, cplx


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--old", required=True)
    ap.add_argument("--new", default=os.path.join(ROOT, "ozimmu_amd", "libozimmu_hip.so"))
    ap.add_argument("--shapes", nargs="+", default=["8192", "32768x32768x1024:NT", "8192x8192x256", "8192x8192x512", "16384x16384x512:NT:b1",
                                                    "4096x4096x1024", "4096", "2048", "1024", "4096:z"])
    ap.add_argument("--mode", default="fp64_int8_9")
    ap.add_argument("--legs", type=int, default=7)
    ap.add_argument("--reps", type=int, default=6)
    a = ap.parse_args()
    import numpy as np
    import torch
    libs = {"old": bindings(a.old, "ozimmu_amd_old"), "new": bindings(a.new, "ozimmu_amd_new")}
    hs = {k: v.create() for k, v in libs.items()}
    print(f"# old = {a.old}  ({libs['old'].version()})\n# new = {a.new}  ({libs['new'].version()})\n# {a.mode}, model's pick on both sides "
          f"(OZIMMU_HIP_AUTOTUNE=0), {a.legs} alternating legs of {a.reps} queued calls, median leg")
    for spec in a.shapes:
        m, n, k, ops, beta, cplx = parse(spec)
        dt = torch.complex128 if cplx else torch.float64
        g = torch.Generator(device="cuda")
        g.manual_seed(1)

        def rnd(r, c):
            x = torch.rand(c, r, dtype=torch.float64, device="cuda", generator=g) * 2 - 1      # column-major r x c
            if cplx:
                x = torch.complex(x, torch.rand(c, r, dtype=torch.float64, device="cuda", generator=g) * 2 - 1)
            return x
        A = rnd(m, k) if ops[0] == "N" else rnd(k, m)
        B = rnd(k, n) if ops[1] == "N" else rnd(n, k)
        lda, ldb = A.shape[1], B.shape[1]
        C0 = rnd(m, n)
        out, kern, legs = {}, {}, {"old": [], "new": []}
        kind = libs["new"].complx if cplx else libs["new"].real
        for name, L in libs.items():
            C = C0.clone()
            st = L.gemm(hs[name], ops[0], ops[1], m, n, k, 1.0, A, lda, B, ldb, beta, C, m, a.mode, kind)
            torch.cuda.synchronize()
            assert st == 0, (name, st)
            out[name] = C
            kern[name] = L.last_kernel(hs[name])[0]
        same = bool(torch.equal(torch.view_as_real(out["old"]).view(torch.int64) if cplx else out["old"].view(torch.int64),
                                torch.view_as_real(out["new"]).view(torch.int64) if cplx else out["new"].view(torch.int64)))
        C = C0.clone()
        for leg in range(2 * a.legs + 2):
            name = ("old", "new")[leg & 1] if (leg // 2) % 2 == 0 else ("new", "old")[leg & 1]
            L = libs[name]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                L.gemm(hs[name], ops[0], ops[1], m, n, k, 1.0, A, lda, B, ldb, beta, C, m, a.mode, kind)
            torch.cuda.synchronize()
            if leg >= 2:
                legs[name].append((time.perf_counter() - t0) / a.reps)
        fl = (8.0 if cplx else 2.0) * m * n * k
        tf = {nm: fl / float(np.median(v)) / 1e12 for nm, v in legs.items()}
        print(f"{spec:28s} old {tf['old']:7.2f} ({kern['old']:>13s})  new {tf['new']:7.2f} ({kern['new']:>13s})  new / old {tf['new'] / tf['old']:6.3f}  "
              f"bit-identical {same}", flush=True)
        del A, B, C, C0, out
        torch.cuda.empty_cache()
    for k_, v in libs.items():
        v.destroy(hs[k_])


if __name__ == "__main__":
    main()
