#!/usr/bin/env python
"""tools/ab.py — one parametrised A/B harness for whole GEMM calls (replaces the per-question scripts of rounds 1-3).

For every shape x mode it times a list of VARIANTS of the same call in alternating legs (A, B, C, C, B, A, ...: neither
side owns the cool part) of queued back-to-back calls between device synchronisations, and prints the median leg per
variant in DGEMM-equivalent TFLOP/s, the best variant, and the time the first variant ("auto" = the library's own kernel
choice) loses against the best forced one ("policy regret").

    python tools/ab.py --shapes 8192 4096x4096x1024 --modes fp64_int8_9 --variants auto k64 wide classic rocblas
    python tools/ab.py --preset short_k --variants auto classic wide k64
    python tools/ab.py --preset mid --variants auto OZIMMU_HIP_K64_TILE=0 rocblas
    python tools/ab.py --preset random:40:0 --variants auto k2 classic wide k64 --losses 3
    python tools/ab.py --shapes 8192 --variants OZIMMU_HIP_K64_BREG=0 OZIMMU_HIP_K64_BREG=1 rocblas --legs 9
    python tools/ab.py --shapes 4096 --complex --ops CN --variants auto rocblas

Variants:  auto | k2 | classic | wide | x16 | k64 (OZIMMU_HIP_GEMM_KERNEL=...) | rocblas (native rocBLAS through the library's
own vendor handle) | NAME=VALUE[,NAME=VALUE...] (environment switches, read per call: csrc/config.h).
Presets:   squares (1024 ... 16384) | mid (1280 ... 7168: off the tile-count sweet spots) | short_k (panel updates, K = 128 ... 1024)
           | panel (beta = 1, N/T) | random:COUNT:SEED (rectangular shapes, K and S drawn at random).
"""
import argparse
import os
import sys
import time

os.environ.setdefault("OZIMMU_HIP_ENV_PER_CALL", "1")  # the variants flip switches between calls of one process
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

KERNELS = ("k2", "classic", "wide", "x16", "k64")
SWITCHES = ("OZIMMU_HIP_GEMM_KERNEL",)


def preset(name, rng_mod=None):
    import numpy as np
    if name == "squares":
        return [((n, n, n), None) for n in (1024, 1536, 2048, 3072, 4096, 6144, 8192, 16384)]
    if name == "mid":
        return [((n, n, n), None) for n in (1280, 1536, 1792, 2048, 2304, 2560, 2816, 3072, 3328, 3584, 3840, 4096, 4608, 5120,
                                             6144, 7168)]
    if name == "short_k":
        return [(s, None) for s in [(8192, 8192, 128), (8192, 8192, 256), (8192, 8192, 384), (8192, 8192, 512),
                                    (16384, 16384, 128), (16384, 16384, 256), (4096, 4096, 128), (4096, 4096, 256),
                                    (4096, 4096, 512), (4096, 4096, 1024), (2048, 2048, 512), (32768, 32768, 256)]]
    if name == "bench":  # the cases of bench.py's extra.policy_regret
        return [((4096, 4096, 1024), "fp64_int8_9"), ((2048, 2048, 2048), "fp64_int8_9"), ((1024, 1024, 1024), "fp64_int8_9"),
                ((3000, 5000, 512), "fp64_int8_6"), ((4096, 4096, 256), "fp64_int8_4"), ((2500, 1800, 2048), "fp64_int8_12"),
                ((8192, 8192, 256), "fp64_int8_9"), ((1536, 1536, 1536), "fp64_int8_8"), ((6000, 3000, 128), "fp64_int8_8"),
                ((900, 1300, 4096), "fp64_int8_6")]
    if name == "panel":
        return [((m, m, k), None) for m in (8192, 16384) for k in (128, 256, 512, 1024, 2048)]
    if name.startswith("random"):
        _, count, seed = (name.split(":") + ["40", "0"])[:3]
        rng = np.random.default_rng(int(seed))
        out = []
        for _ in range(int(count)):
            m, n = (int(rng.integers(200, 5000)) for _ in range(2))
            k = int(rng.choice([128, 256, 512, 1024, 2048, 4096, 8192]))
            S = int(rng.choice([4, 6, 8, 9, 12]))
            out.append(((m, n, k), f"fp64_int8_{S}"))
        return out
    raise SystemExit(f"unknown preset {name}")


def parse_shape(s):
    p = [int(x) for x in s.lower().split("x")]
    return (p[0],) * 3 if len(p) == 1 else tuple(p)


class Harness:
    def __init__(self):
        import torch
        import ozimmu_amd as oz
        self.torch, self.oz = torch, oz
        self.h = oz.create()
        oz.set_cuda_stream(self.h, torch.cuda.current_stream())

    def close(self):
        self.torch.cuda.synchronize()
        self.oz.destroy(self.h)

    def set_variant(self, v):
        """applies a variant's environment; returns True when the call is the native vendor GEMM"""
        for k in list(self._applied):
            os.environ.pop(k, None)
        self._applied = []
        if v in ("auto", "rocblas"):
            return v == "rocblas"
        pairs = [("OZIMMU_HIP_GEMM_KERNEL", v)] if v in KERNELS else [kv.split("=", 1) for kv in v.split(",")]
        for k, val in pairs:
            os.environ[k] = val
            self._applied.append(k)
        return False

    _applied = []

    def time_shape(self, shape, mode, variants, ops="NN", beta=0.0, cplx=False, legs=4, leg_seconds=0.3):
        """{variant: median seconds per call}"""
        torch, oz, h = self.torch, self.oz, self.h
        m, n, k = shape
        opa, opb = ops[0].upper(), ops[1].upper()

        def operand(rows, cols, op):   # column-major storage of an operand whose op(X) is rows x cols
            r, c = (rows, cols) if op == "N" else (cols, rows)
            x = torch.rand(c, r, dtype=torch.float64, device="cuda") * 2 - 1
            if cplx:
                x = torch.complex(x, torch.rand(c, r, dtype=torch.float64, device="cuda") * 2 - 1)
            return x, r
        a, lda = operand(m, k, opa)
        b, ldb = operand(k, n, opb)
        c = torch.zeros(n, m, dtype=torch.complex128 if cplx else torch.float64, device="cuda")
        kind = oz.complx if cplx else oz.real
        al, be = (1.0 + 0j, complex(beta)) if cplx else (1.0, float(beta))
        flops = (8.0 if cplx else 2.0) * m * n * k

        def make(v):
            native = self.set_variant(v)
            if native and not cplx:
                return lambda: oz.native_dgemm(h, opa, opb, m, n, k, 1.0, a, lda, b, ldb, beta, c, m)
            if native:  # rocBLAS ZGEMM through torch (row-major view of C = op(A) op(B): C^T = op(B)^T op(A)^T)
                ta = {"N": lambda x: x, "T": lambda x: x.mT, "C": lambda x: x.mH}
                # a is stored (cols, ld) row-major = column-major X; X as a torch matrix is a.mT
                A_, B_ = ta[opa](a.mT), ta[opb](b.mT)
                return lambda: torch.matmul(A_, B_)
            return lambda: oz.gemm(h, opa, opb, m, n, k, al, a, lda, b, ldb, be, c, m, mode, kind)
        times = {v: [] for v in variants}
        est = None
        for leg in range(legs + 1):           # leg 0: warm-up, discarded
            order = variants if leg % 2 == 0 else variants[::-1]
            for v in order:
                fn = make(v)
                self.set_variant(v)
                if est is None:
                    fn(); fn()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    fn()
                    torch.cuda.synchronize()
                    est = max(time.perf_counter() - t0, 1e-5)
                reps = max(3, min(5000, int(leg_seconds / est)))
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    st = fn()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / reps
                if isinstance(st, int) and st != 0:
                    raise RuntimeError(f"{v}: status {st}")
                if leg:
                    times[v].append(dt)
        self.set_variant("auto")
        return {v: sorted(t)[len(t) // 2] for v, t in times.items()}, flops


def policy_regret(harness, cases, forced=("k2", "classic", "wide", "k64"), legs=2, leg_seconds=0.15):
    """[(shape, mode)] -> summary of how much time the library's kernel choice loses against the best forced kernel.
    Used by bench.py (`extra.policy_regret`) and by --preset random."""
    rows = []
    for shape, mode in cases:
        med, _ = harness.time_shape(shape, mode, ["auto"] + list(forced), legs=legs, leg_seconds=leg_seconds)
        best = min(forced, key=lambda v: med[v])
        rows.append({"shape": "x".join(map(str, shape)), "mode": mode, "best": best,
                     "regret_pct": round(max(0.0, (med["auto"] / med[best] - 1) * 100), 2)})
    r = sorted(x["regret_pct"] for x in rows)
    return {"cases": len(rows), "median_pct": r[len(r) // 2], "max_pct": r[-1], "over_3pct": sum(1 for x in r if x > 3.0),
            "worst": sorted(rows, key=lambda x: -x["regret_pct"])[:3], "forced_kernels": list(forced)}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--shapes", nargs="*", default=[], help="N or MxNxK")
    ap.add_argument("--preset", default=None)
    ap.add_argument("--modes", nargs="*", default=["fp64_int8_9"])
    ap.add_argument("--variants", nargs="+", default=["auto", "rocblas"])
    ap.add_argument("--ops", default="NN")
    ap.add_argument("--beta", type=float, default=0.0)
    ap.add_argument("--complex", action="store_true", dest="cplx")
    ap.add_argument("--legs", type=int, default=4)
    ap.add_argument("--leg-seconds", type=float, default=0.3)
    ap.add_argument("--losses", type=float, default=None, help="list the cases where variant 1 loses more than this many %% of time")
    ap.add_argument("--regret-json", action="store_true",
                    help="print ONE JSON object: the policy's regret against every forced kernel over the cases (bench.py)")
    args = ap.parse_args()
    cases = [(parse_shape(s), None) for s in args.shapes] + (preset(args.preset) if args.preset else [])
    if not cases:
        raise SystemExit("no shapes: --shapes or --preset")
    H = Harness()
    if args.regret_json:
        import json
        fixed = [(sh, md or args.modes[0]) for sh, md in cases]
        print(json.dumps(policy_regret(H, fixed, legs=max(1, min(args.legs, 2)), leg_seconds=min(args.leg_seconds, 0.08))), flush=True)
        H.close()
        return
    losses = []
    for shape, fixed_mode in cases:
        for mode in ([fixed_mode] if fixed_mode else args.modes):
            med, flops = H.time_shape(shape, mode, args.variants, args.ops, args.beta, args.cplx, args.legs, args.leg_seconds)
            tf = {v: flops / t / 1e12 for v, t in med.items()}
            ours = [v for v in args.variants if v != "rocblas"]
            best = min(ours, key=lambda v: med[v])
            first = args.variants[0]
            line = f"{'x'.join(map(str, shape)):>18} {mode:13s} " + "  ".join(f"{v} {tf[v]:7.2f}" for v in args.variants)
            line += f"   best {best} ({(med[first] / med[best] - 1) * 100:+.1f} % time {first} vs best)"
            if "rocblas" in med:
                line += f"   {first} / rocBLAS {med['rocblas'] / med[first]:.3f}"
            print(line, flush=True)
            if args.losses is not None and (med[first] / med[best] - 1) * 100 > args.losses:
                losses.append(line)
    if args.losses is not None:
        print(f"LOSSES > {args.losses} %: {len(losses)} of {len(cases)}")
        print("\n".join(losses) or "none")
    H.close()


if __name__ == "__main__":
    main()
