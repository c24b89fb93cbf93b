#!/usr/bin/env python
"""Generates tests/golden/*.npz.

The reference cannot be built or run in this image (CUDA C++, needs nvcc + cuBLAS + un-vendored cutf) and
its tree ships no golden vectors, so these fixtures are produced by the CPU oracle (oracle/ozaki_oracle.c,
the line-by-line restatement of the reference algorithm) after that oracle has been pinned against the
reference's own CI gate and exact identities (tests/test_oracle.py).  They freeze inputs and every
intermediate of the hot path -- slices, max_exp, INT32 diagonal sums, the FP64 result in both summation
groupings, the auto-mode counters -- so that (a) the oracle cannot drift silently and (b) the HIP path is
compared against committed data, not only against a freshly built checker.

    python tests/golden/make_golden.py        # rewrites the .npz files (deterministic: seeded)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from tests.util import ColMajor, exp_rand, operand, uniform_pm1, wide_exponent  # noqa: E402

CASES = [
    # name, op_a, op_b, m, n, k, S, fill, alpha, beta, special rows
    ("g1_nn_s6_uniform", "N", "N", 33, 35, 37, 6, uniform_pm1, 1.0, 0.0, False),
    ("g2_tn_s9_exprand2", "T", "N", 33, 35, 37, 9, exp_rand(2.0), 1.0, 0.0, False),
    ("g3_nt_s9_special", "N", "T", 64, 64, 64, 9, uniform_pm1, 1.0, 0.0, True),
    ("g4_tt_s13_wide_alpha_beta", "T", "T", 20, 50, 70, 13, wide_exponent(8), -1.5, 0.5, False),
]


def build_case(name, op_a, op_b, m, n, k, S, fill, alpha, beta, special, seed):
    rng = np.random.default_rng(seed)
    a = operand(op_a, m, k, rng, fill=fill)
    b = operand(op_b, k, n, rng, fill=fill)
    if special:
        av = a.view  # op N: (m, k)
        av[0, :] = 0.0                       # zero row
        av[1, :] *= 2.0 ** -600              # tiny row
        av[1, 3] = 4e-320                    # subnormal element in a (not that) tiny row: vanishes
        av[2, :] = rng.uniform(-1, 1, k) * 10.0 ** rng.uniform(-30, 30, k)  # 60 decades in one row
        av[3, 5] = 0.0
        av[4, :] = -np.abs(av[4, :])
        bv = b.view  # op T: storage (n, k)
        bv[7, :] = 0.0
        bv[8, :] *= 2.0 ** 300
    L = O.bits_per_int8(k)
    pa, ea = O.split("A", op_a, a.view, S, L)
    pb, eb = O.split("B", op_b, b.view, S, L)
    d = O.diagonal_sums(pa, pb)
    c0 = ColMajor(m, n, fill=uniform_pm1, rng=rng)
    c_ref = ColMajor(m, n)
    c_ref.buf[...] = c0.buf
    c_diag = ColMajor(m, n)
    c_diag.buf[...] = c0.buf
    assert O.gemm(op_a, op_b, m, n, k, alpha, a.view, b.view, beta, c_ref.view, S, O.ORDER_REFERENCE) == 0
    assert O.gemm(op_a, op_b, m, n, k, alpha, a.view, b.view, beta, c_diag.view, S, O.ORDER_DIAGONAL) == 0
    sel, cnt = O.auto_select(op_a, op_b, m, n, k, a.view, b.view, 1.5)
    return dict(op_a=op_a, op_b=op_b, m=m, n=n, k=k, S=S, L=L, alpha=alpha, beta=beta,
                a=np.asfortranarray(a.view), b=np.asfortranarray(b.view), c0=np.asfortranarray(c0.view),
                planes_a=pa, planes_b=pb, max_exp_a=ea, max_exp_b=eb, diag=d,
                c_reference_order=np.asfortranarray(c_ref.view), c_diagonal_order=np.asfortranarray(c_diag.view),
                auto_counters=cnt, auto_selected=sel)


def main():
    for i, case in enumerate(CASES):
        data = build_case(*case, seed=100 + i)
        path = os.path.join(HERE, case[0] + ".npz")
        np.savez_compressed(path, **data)
        print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
