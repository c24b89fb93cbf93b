"""tools/residual.py — sampled relative residual of a GEMM result against a long-double product.

    residual = ||C - C_true||_F / ||C_true||_F   over `ns` random entries (i, j)

the metric of the reference's harness (mateval::relative_residual, test/main_test.cu:101-117), evaluated on a sample
so that it stays cheap at 16384^3.  Plain numpy (x87 80-bit long double on the host); deliberately independent of
`oracle/` so that bench.py and tools/ozimmu_eval.py can report accuracy without touching the test oracle.
Arguments are column-major views: `a` is the stored A (m x k for op 'N', k x m for op 'T'), likewise `b`; `c` is m x n.
"""
import numpy as np


def sampled_relative_residual(op_a, op_b, m, n, k, a, b, c, ns=1024, seed=1, alpha=1.0):
    cplx = np.iscomplexobj(c) or np.iscomplexobj(a) or np.iscomplexobj(b)
    ld = np.clongdouble if cplx else np.longdouble
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, m, ns)
    cols = rng.integers(0, n, ns)
    num = np.longdouble(0)
    den = np.longdouble(0)
    for s0 in range(0, ns, 256):  # bounded temporaries: 256 x k long doubles per operand
        r, q = rows[s0:s0 + 256], cols[s0:s0 + 256]
        aa = (a[r, :] if op_a.upper() == "N" else a[:, r].T).astype(ld)      # rows of op(A)
        bb = (b[:, q].T if op_b.upper() == "N" else b[q, :]).astype(ld)      # columns of op(B)
        truth = ld(alpha) * (aa * bb).sum(axis=1)
        d = c[r, q].astype(ld) - truth
        num += (np.abs(d) ** 2).sum()
        den += (np.abs(truth) ** 2).sum()
    return float(np.sqrt(num / den)) if den > 0 else float(np.sqrt(num))
