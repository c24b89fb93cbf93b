#!/usr/bin/env python
"""tools/trace.py — kernel-by-kernel timeline of a few calls (which slice-GEMM kernel the policy launches, how long the split
kernels take, the gaps between launches).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $REPO/tools/trace.py run 1024 1024 1024
    python $REPO/tools/trace.py summ /tmp/tr [--last 24]

`run m n k [--mode fp64_int8_9] [--opb N|T] [--beta b] [--reps r] [--rocblas r]`: r calls through the C ABI (then r native DGEMMs).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run(a):
    import torch
    import ozimmu_amd as oz
    m, n, k = a.m, a.n, a.k
    h = oz.create()
    oz.set_cuda_stream(h, torch.cuda.current_stream())
    A = torch.rand(k, m, dtype=torch.float64, device="cuda") * 2 - 1
    B = torch.rand(k * n, dtype=torch.float64, device="cuda") * 2 - 1
    C = torch.zeros(n, m, dtype=torch.float64, device="cuda")
    ldb = n if a.opb == "T" else k
    for _ in range(a.reps):
        assert oz.gemm(h, "N", a.opb, m, n, k, 1.0, A, m, B, ldb, a.beta, C, m, a.mode) == 0
    torch.cuda.synchronize()
    for _ in range(a.rocblas):
        oz.native_dgemm(h, "N", a.opb, m, n, k, 1.0, A, m, B, ldb, a.beta, C, m)
    torch.cuda.synchronize()
    oz.destroy(h)


def summ(a):
    import csv
    import glob
    f = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    keep = ("ozhip", "Cijk", "memset", "fill")
    rows = [r for r in rows if any(s in r["Kernel_Name"] or s in r["Kernel_Name"].lower() for s in keep)]
    prev_end = None
    for r in rows[-a.last:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print(f"{r['Kernel_Name'][:72]:72s} dur {(e - s) / 1e3:9.1f} us  gap {gap:7.1f} us  grid {r.get('Grid_Size', '?')} "
              f"wg {r.get('Workgroup_Size', '?')}")
        prev_end = e


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    r = sub.add_parser("run")
    for x in ("m", "n", "k"):
        r.add_argument(x, type=int)
    r.add_argument("--mode", default="fp64_int8_9")
    r.add_argument("--opb", default="N")
    r.add_argument("--beta", type=float, default=0.0)
    r.add_argument("--reps", type=int, default=4)
    r.add_argument("--rocblas", type=int, default=2)
    s = sub.add_parser("summ")
    s.add_argument("dir")
    s.add_argument("--last", type=int, default=24)
    args = ap.parse_args()
    (run if args.cmd == "run" else summ)(args)
