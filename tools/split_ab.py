"""A/B of the split stage alone (stage profiler: split_A + split_B per call) under environment switches, alternating legs.
    python tools/split_ab.py [--shapes 8192 4096x4096x8192 ...] [--ops NN] [--variants NAME=VALUE ...] [--legs 7] [--calls 6]
e.g. the cut pass walking the operand from its end (what the row-maximum pass left in the memory-side cache) vs from its start:
    python tools/split_ab.py --variants OZIMMU_HIP_SPLIT_REVERSE=1 OZIMMU_HIP_SPLIT_REVERSE=0"""
import argparse
import os
import sys

os.environ.setdefault("OZIMMU_HIP_ENV_PER_CALL", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ozimmu_amd as oz  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="*", default=["8192"])
    ap.add_argument("--ops", default="NN")
    ap.add_argument("--mode", default="fp64_int8_9")
    ap.add_argument("--variants", nargs="+", default=["OZIMMU_HIP_SPLIT_REVERSE=1", "OZIMMU_HIP_SPLIT_REVERSE=0"])
    ap.add_argument("--legs", type=int, default=7)
    ap.add_argument("--calls", type=int, default=6)
    a = ap.parse_args()
    h = oz.create()
    oz.enable_profiling(h)
    for shape in a.shapes:
        m, n, k = (int(shape),) * 3 if "x" not in shape else map(int, shape.split("x"))
        A = torch.rand((k, m) if a.ops[0] == "N" else (m, k), dtype=torch.float64, device="cuda") * 2 - 1   # column-major storage
        B = torch.rand((n, k) if a.ops[1] == "N" else (k, n), dtype=torch.float64, device="cuda") * 2 - 1
        Cm = torch.zeros((n, m), dtype=torch.float64, device="cuda")
        lda, ldb = A.shape[1], B.shape[1]
        res = {v: [] for v in a.variants}
        for leg in range(a.legs + 1):
            for v in a.variants:
                sets = dict(kv.split("=") for kv in v.split(",")) if "=" in v else {}
                for kk, vv in sets.items():
                    os.environ[kk] = vv
                t = []
                for _ in range(a.calls):
                    assert oz.gemm(h, a.ops[0], a.ops[1], m, n, k, 1.0, A, lda, B, ldb, 0.0, Cm, m, a.mode) == 0
                    s = oz.last_stage_ms(h)
                    t.append((s["split_A"] + s["split_B"], s["int8tc"]))
                for kk in sets:
                    os.environ.pop(kk, None)
                if leg:  # leg 0 warms up
                    res[v].append(np.median([x[0] for x in t]))
        line = f"{m}x{n}x{k} {a.ops} {a.mode}: split stage ms (median of {a.legs} legs x {a.calls} calls)"
        for v in a.variants:
            line += f"   {v} {np.median(res[v]):.4f} [{min(res[v]):.4f}..{max(res[v]):.4f}]"
        print(line, flush=True)
    oz.destroy(h)


if __name__ == "__main__":
    main()
