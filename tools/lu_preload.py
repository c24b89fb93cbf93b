#!/usr/bin/env python3
"""An end-to-end caller of the LD_PRELOAD boundary (VERDICT r5 "missing" 2 / next 5): right-looking blocked LU with partial
pivoting, FP64, written against plain PyTorch - panels by rocSOLVER (torch.linalg.lu_factor), the row block by a unit-lower
triangular solve, the trailing update A22 -= L21 U12 by torch.addmm_ = one rocblas_dgemm per block column.  That DGEMM is what
the reference exists to replace (/root/reference/README.md:17-20 "LD_PRELOAD an application", /root/reference/src/cublas.cu:280-295
cublasDgemm_v2): K = NB panels under a shrinking trailing matrix, every shape seen ONCE - the call stream a factorisation
feeds a DGEMM interposer, and the place where the library's defaults (kernel choice, tuner, thresholds, workspace growth)
meet a realistic shape distribution.

    python tools/lu_preload.py --n 16384 --nb 512 --modes native fp64_int8_9 fp64_int8_8 fp64_int8_auto

For every mode the parent starts one child process (LD_PRELOAD = ozimmu_amd/libozimmu_hip.so, OZIMMU_COMPUTE_MODE = the mode,
OZIMMU_INTERCEPT_THRESHOLD_{M,N,K} = NB: the reference's default thresholds of 1024, src/handle.cu:25-30, would leave every
K = 512 update to the vendor), which factorises the same seeded matrix twice (the first run grows the workspace and loads
rocSOLVER's kernels; the second is timed), and reports
    backward error   ||P A - L U||_F / ||A||_F   (L U recomputed by the VENDOR dgemm: the check never runs on the Ozaki path)
    wall time of the factorisation and the part of it spent in the trailing updates (events around the addmm_ calls)
    DGEMM calls the factorisation issued itself / calls the shim saw / took / left to the vendor, and which slice-GEMM kernels ran
    (ozimmu_hip_intercept_stats).
`native` = the same program without the preload.  Nothing here imports the oracle."""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ozimmu_amd", "libozimmu_hip.so")
KERNEL_CODES = {0: "k2", 1: "classic", 2: "wide", 3: "x16", 4: "k64", 12: "k64_breg", 16: "k2_one_launch"}


def _stats():
    """counters of the preloaded shim (None without the preload)"""
    if LIB not in os.environ.get("LD_PRELOAD", ""):
        return None
    lib = ctypes.CDLL(LIB)  # already mapped by the preload: the same instance
    out = (ctypes.c_ulonglong * 28)()
    lib.ozimmu_hip_intercept_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    if lib.ozimmu_hip_intercept_stats(out, 28) != 0:
        return None
    v = list(out)
    return {"seen": v[0], "taken": v[1], "declined": v[2], "failed": v[3],
            "kernels": {KERNEL_CODES.get(c, str(c)): v[4 + c] for c in range(24) if v[4 + c]}}


def blocked_lu(a, nb, timers=None):
    """in place: a -> L\\U (unit lower L below the diagonal), returns the row permutation `perm` with (P A)[i] = A[perm[i]].
    Right-looking: factor the panel, swap the rows of the other block columns, solve for the row block, update the rest."""
    import torch
    n = a.shape[0]
    perm = torch.arange(n, device=a.device)
    calls = 0
    for j in range(0, n, nb):
        w = min(nb, n - j)
        lu, piv = torch.linalg.lu_factor(a[j:, j:j + w])          # rocSOLVER getrf on the tall panel
        a[j:, j:j + w] = lu
        # LAPACK pivots (1-based, sequential swaps) -> the rows of this panel's range that move
        p = (piv.to(torch.int64) - 1).cpu().tolist()
        local = list(range(n - j))
        for i, pi in enumerate(p):
            if pi != i:
                local[i], local[pi] = local[pi], local[i]
        moved = [i for i, s in enumerate(local) if s != i]
        if moved:
            dst = torch.tensor(moved, device=a.device) + j
            src = torch.tensor([local[i] for i in moved], device=a.device) + j
            perm[dst] = perm[src]
            if j > 0:
                a[dst, :j] = a[src, :j]
            if j + w < n:
                a[dst, j + w:] = a[src, j + w:]
        if j + w < n:
            # U12 = L11^-1 A12 (unit lower), then the trailing update: ONE dgemm, m = n = N - j - w, k = w, beta = 1
            a[j:j + w, j + w:] = torch.linalg.solve_triangular(a[j:j + w, j:j + w], a[j:j + w, j + w:], upper=False,
                                                               unitriangular=True)
            if timers is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            a[j + w:, j + w:].addmm_(a[j + w:, j:j + w], a[j:j + w, j + w:], alpha=-1.0)
            calls += 1
            if timers is not None:
                e1.record()
                timers.append((e0, e1))
    return perm, calls


def backward_error(a0, lu, perm):
    """||P A - L U||_F / ||A||_F with L U from the vendor GEMM (OZIMMU_COMPUTE_MODE=dgemm is read per call: src/cublas.cu:18-48),
    block column by block column (no third N x N matrix)"""
    import torch
    saved = os.environ.get("OZIMMU_COMPUTE_MODE")
    os.environ["OZIMMU_COMPUTE_MODE"] = "dgemm"
    try:
        n = a0.shape[0]
        low = torch.tril(lu, -1)
        low.diagonal().fill_(1.0)
        num = 0.0
        step = 2048
        for c in range(0, n, step):
            u = torch.triu(lu[:, c:c + step], -c)
            r = a0[perm, c:c + step] - low @ u
            num += float((r * r).sum())
        return (num ** 0.5) / float(torch.linalg.matrix_norm(a0))
    finally:
        if saved is None:
            os.environ.pop("OZIMMU_COMPUTE_MODE", None)
        else:
            os.environ["OZIMMU_COMPUTE_MODE"] = saved


def child(args):
    import torch
    torch.manual_seed(args.seed)
    n, nb = args.n, args.nb
    a0 = torch.rand(n, n, dtype=torch.float64, device="cuda") * 2 - 1
    out = {"mode": os.environ.get("OZIMMU_COMPUTE_MODE", "native") if LIB in os.environ.get("LD_PRELOAD", "") else "native",
           "n": n, "nb": nb}
    best = None
    for rep in range(args.reps + 1):     # rep 0: warm-up (workspace growth, rocSOLVER's first-use costs)
        a = a0.clone()
        before = _stats()
        timers = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        perm, calls = blocked_lu(a, nb, timers)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        after = _stats()
        upd = sum(e0.elapsed_time(e1) for e0, e1 in timers) * 1e-3
        if rep > 0 and (best is None or wall < best["wall_s"]):
            best = {"wall_s": wall, "update_s": upd}
            if after:
                best["shim"] = {k: after[k] - before[k] for k in ("seen", "taken", "declined", "failed")}
                best["kernels"] = {k: v - before["kernels"].get(k, 0) for k, v in after["kernels"].items()
                                   if v - before["kernels"].get(k, 0)}
    out.update(best)
    out["dgemm_calls_issued"] = calls
    flops = sum(2.0 * (n - j - min(nb, n - j)) ** 2 * min(nb, n - j) for j in range(0, n, nb))
    out["update_tflops"] = flops / out["update_s"] / 1e12 if out["update_s"] > 0 else None
    out["backward_error"] = backward_error(a0, a, perm)
    print("LU_RESULT " + json.dumps(out), flush=True)


def run_mode(mode, n, nb, reps=1, seed=0, extra_env=None, timeout=3600):
    """one child process; returns its result dict"""
    env = {k: v for k, v in os.environ.items() if not k.startswith("OZIMMU_")}
    if mode != "native":
        env.update(LD_PRELOAD=LIB, OZIMMU_COMPUTE_MODE=mode, OZIMMU_INTERCEPT_THRESHOLD_M=str(nb), OZIMMU_INTERCEPT_THRESHOLD_N=str(nb),
                   OZIMMU_INTERCEPT_THRESHOLD_K=str(nb))
        if mode == "fp64_int8_auto":
            env.setdefault("OZIMMU_AUTO_AVG_MANTISSA_LOSS_THRESHOLD", "1.5")
    env.update(extra_env or {})
    cmd = [sys.executable, os.path.abspath(__file__), "--child", "--n", str(n), "--nb", str(nb), "--reps", str(reps), "--seed", str(seed)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("LU_RESULT ")]
    if p.returncode != 0 or not lines:
        raise RuntimeError(f"lu_preload child ({mode}) failed rc={p.returncode}\n{p.stdout[-2000:]}\n{p.stderr[-2000:]}")
    r = json.loads(lines[-1][len("LU_RESULT "):])
    r["mode"] = mode
    return r


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--nb", type=int, default=512)
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--modes", nargs="+", default=["native", "fp64_int8_9", "fp64_int8_8", "fp64_int8_auto"])
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    rows = [run_mode(m, args.n, args.nb, args.reps, args.seed) for m in args.modes]
    if args.json:
        print(json.dumps(rows))
        return
    base = next((r for r in rows if r["mode"] == "native"), None)
    print(f"# blocked LU, N = {args.n}, NB = {args.nb}, FP64, partial pivoting; trailing updates = rocblas_dgemm (m = n = N - j - NB, k = NB, beta = 1)")
    print(f"# {'mode':16s} {'||PA-LU||/||A||':>16s} {'wall s':>8s} {'x native':>8s} {'updates s':>9s} {'TFLOP/s':>8s} "
          f"{'issued':>6s} {'seen':>5s} {'taken':>5s} {'left':>5s}  kernels")
    for r in rows:
        shim = r.get("shim", {})
        rel = f"{base['wall_s'] / r['wall_s']:.3f}" if base else "-"
        print(f"  {r['mode']:16s} {r['backward_error']:16.3e} {r['wall_s']:8.3f} {rel:>8s} {r['update_s']:9.3f} {r['update_tflops']:8.2f} "
              f"{r['dgemm_calls_issued']:6d} {shim.get('seen', 0):5d} {shim.get('taken', 0):5d} {shim.get('declined', 0):5d}  "
              f"{r.get('kernels', {})}")


if __name__ == "__main__":
    main()
