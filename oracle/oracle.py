"""ctypes front-end of the CPU oracle (oracle/ozaki_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the product package `ozimmu_amd`.

Matrices are column-major: pass 2-D numpy float64 arrays in Fortran order; the leading dimension is
taken from the array (arr.strides[1] // 8), so views into larger buffers work.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboz_oracle.so")

OP_N, OP_T, OP_C = 0, 1, 2   # OP_C: conjugate transpose (complex entry points; the real ones treat it as OP_T)
ORDER_REFERENCE, ORDER_DIAGONAL = 0, 1
QUIRK_REF_SUBNORMAL = 1


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("ozaki_oracle.c", "ozaki_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        sz, i, d, vp = C.c_size_t, C.c_int, C.c_double, C.c_void_p
        L.oz_oracle_bits_per_int8.restype = C.c_uint32
        L.oz_oracle_bits_per_int8.argtypes = [C.c_uint32]
        L.oz_oracle_num_split_from_mode.restype = i
        L.oz_oracle_num_split_from_mode.argtypes = [C.c_char_p]
        L.oz_oracle_pair_list.restype = i
        L.oz_oracle_pair_list.argtypes = [i, vp, vp]
        L.oz_oracle_pad4.restype = sz
        L.oz_oracle_pad4.argtypes = [sz]
        L.oz_oracle_split_A.restype = None
        L.oz_oracle_split_A.argtypes = [i, sz, sz, vp, sz, i, i, vp, sz, vp, i]
        L.oz_oracle_split_B.restype = None
        L.oz_oracle_split_B.argtypes = [i, sz, sz, vp, sz, i, i, vp, sz, vp, i]
        L.oz_oracle_int8_gemm.restype = None
        L.oz_oracle_int8_gemm.argtypes = [vp, vp, sz, sz, sz, vp]
        L.oz_oracle_diagonal_sums.restype = None
        L.oz_oracle_diagonal_sums.argtypes = [vp, vp, sz, sz, sz, i, sz, sz, vp]
        L.oz_oracle_gemm.restype = i
        L.oz_oracle_gemm.argtypes = [i, i, sz, sz, sz, d, vp, sz, vp, sz, d, vp, sz, i, i, sz, i]
        L.oz_oracle_auto_select.restype = i
        L.oz_oracle_auto_select.argtypes = [i, i, sz, sz, sz, vp, sz, vp, sz, d, vp]
        L.oz_oracle_relative_residual.restype = d
        L.oz_oracle_relative_residual.argtypes = [i, i, sz, sz, sz, vp, sz, vp, sz, vp, sz]
        L.oz_oracle_relative_residual_sampled.restype = d
        L.oz_oracle_relative_residual_sampled.argtypes = [i, i, sz, sz, sz, vp, sz, vp, sz, vp, sz,
                                                          sz, vp, vp]
        L.oz_oracle_max_threads.restype = i
        _lib = L
    return _lib


def _ld(a):
    """leading dimension (elements) of a column-major 2-D float64 array or view"""
    assert a.dtype == np.float64 and a.ndim == 2 and a.strides[0] == 8, "need column-major float64"
    return max(a.strides[1] // 8, 1) if a.shape[1] > 1 else max(a.shape[0], 1)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def op_code(op):
    if isinstance(op, str):
        return {"N": OP_N, "T": OP_T, "C": OP_C}[op.upper()]
    return int(op)


def bits_per_int8(k):
    return int(lib().oz_oracle_bits_per_int8(int(k)))


def num_split_from_mode(mode):
    return int(lib().oz_oracle_num_split_from_mode(mode.encode()))


def pair_list(S):
    P = S * (S + 1) // 2
    a = np.zeros(P, dtype=np.int32)
    b = np.zeros(P, dtype=np.int32)
    n = lib().oz_oracle_pair_list(S, _p(a), _p(b))
    assert n == P
    return list(zip(a.tolist(), b.tolist()))


def pad4(k):
    return int(lib().oz_oracle_pad4(k))


def split(which, op, a, S, L=None, quirks=0, ldo=None):
    """Slices of op(a).  which='A': a holds op-storage of an (m x k) op(A); returns planes [S, m, ldo]
    and max_exp [m].  which='B': a holds storage of a (k x n) op(B); planes [S, n, ldo], max_exp [n]."""
    op = op_code(op)
    if which == "A":
        rows, k = (a.shape if op == OP_N else a.shape[::-1])
    else:
        k, rows = (a.shape if op == OP_N else a.shape[::-1])
    if L is None:
        L = bits_per_int8(k)
    if ldo is None:
        ldo = pad4(k)
    planes = np.zeros((S, rows, ldo), dtype=np.int8)
    mx = np.zeros(rows, dtype=np.float64)
    if which == "A":
        lib().oz_oracle_split_A(op, rows, k, _p(a), _ld(a), S, L, _p(planes), ldo, _p(mx), quirks)
    else:
        lib().oz_oracle_split_B(op, k, rows, _p(a), _ld(a), S, L, _p(planes), ldo, _p(mx), quirks)
    return planes, mx


def int8_gemm(a_plane, b_plane):
    m, kp = a_plane.shape
    n, kp2 = b_plane.shape
    assert kp == kp2
    a_plane = np.ascontiguousarray(a_plane)
    b_plane = np.ascontiguousarray(b_plane)
    c = np.zeros((m, n), dtype=np.int32, order="F")
    lib().oz_oracle_int8_gemm(_p(a_plane), _p(b_plane), m, n, kp, _p(c))
    return c


def diagonal_sums(a_planes, b_planes, k0=0, k1=None):
    S, m, ldo = a_planes.shape
    n = b_planes.shape[1]
    if k1 is None:
        k1 = ldo
    a_planes = np.ascontiguousarray(a_planes)
    b_planes = np.ascontiguousarray(b_planes)
    d = np.zeros((S, n, m), dtype=np.int64)
    lib().oz_oracle_diagonal_sums(_p(a_planes), _p(b_planes), m, n, ldo, S, k0, k1, _p(d))
    return d.transpose(0, 2, 1)  # [S][m][n]


def gemm(op_a, op_b, m, n, k, alpha, a, b, beta, c, S, order=ORDER_REFERENCE, kchunk=0, quirks=0):
    """In place on c (column-major).  Returns the reference's int status (0 ok, 1 bad shape)."""
    return int(lib().oz_oracle_gemm(op_code(op_a), op_code(op_b), m, n, k, float(alpha), _p(a),
                                    _ld(a), _p(b), _ld(b), float(beta), _p(c), _ld(c), S, order,
                                    kchunk, quirks))


def auto_select(op_a, op_b, m, n, k, a, b, threshold):
    cnt = np.zeros(16, dtype=np.uint64)
    s = lib().oz_oracle_auto_select(op_code(op_a), op_code(op_b), m, n, k, _p(a), _ld(a), _p(b),
                                    _ld(b), float(threshold), _p(cnt))
    return int(s), cnt


def relative_residual(op_a, op_b, m, n, k, a, b, c):
    return float(lib().oz_oracle_relative_residual(op_code(op_a), op_code(op_b), m, n, k, _p(a),
                                                   _ld(a), _p(b), _ld(b), _p(c), _ld(c)))


def relative_residual_sampled(op_a, op_b, m, n, k, a, b, c, ns=2048, seed=1234):
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, m, ns).astype(np.int64)
    cols = rng.integers(0, n, ns).astype(np.int64)
    return float(lib().oz_oracle_relative_residual_sampled(
        op_code(op_a), op_code(op_b), m, n, k, _p(a), _ld(a), _p(b), _ld(b), _p(c), _ld(c), ns,
        _p(rows), _p(cols)))


def max_threads():
    return int(lib().oz_oracle_max_threads())


# ---- complex (ZGEMM) ------------------------------------------------------------------------------------

def _zld(a):
    """leading dimension (complex elements) of a column-major 2-D complex128 array or view"""
    assert a.dtype == np.complex128 and a.ndim == 2 and a.strides[0] == 16, "need column-major complex128"
    return max(a.strides[1] // 16, 1) if a.shape[1] > 1 else max(a.shape[0], 1)


def zgemm(op_a, op_b, m, n, k, alpha, a, b, beta, c, S, order=ORDER_REFERENCE, kchunk=0, quirks=0):
    """In place on c (column-major complex128).  Restates gemm_int8<cuDoubleComplex> (src/gemm.cu:412-521)."""
    L = lib()
    L.oz_oracle_zgemm.restype = C.c_int
    L.oz_oracle_zgemm.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                  C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                  C.c_int, C.c_size_t, C.c_int]
    al = np.array([complex(alpha).real, complex(alpha).imag])
    be = np.array([complex(beta).real, complex(beta).imag])
    return int(L.oz_oracle_zgemm(op_code(op_a), op_code(op_b), m, n, k, _p(al), _p(a), _zld(a), _p(b), _zld(b),
                                 _p(be), _p(c), _zld(c), S, order, kchunk, quirks))


def auto_select_z(op_a, op_b, m, n, k, a, b, threshold):
    L = lib()
    L.oz_oracle_auto_select_z.restype = C.c_int
    L.oz_oracle_auto_select_z.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                          C.c_size_t, C.c_void_p, C.c_size_t, C.c_double, C.c_void_p]
    cnt = np.zeros(16, dtype=np.uint64)
    s = L.oz_oracle_auto_select_z(op_code(op_a), op_code(op_b), m, n, k, _p(a), _zld(a), _p(b), _zld(b),
                                  float(threshold), _p(cnt))
    return int(s), cnt


def relative_residual_sampled_z(op_a, op_b, m, n, k, a, b, c, ns=2048, seed=1234):
    L = lib()
    L.oz_oracle_relative_residual_sampled_z.restype = C.c_double
    L.oz_oracle_relative_residual_sampled_z.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t,
                                                        C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                                        C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, m, ns).astype(np.int64)
    cols = rng.integers(0, n, ns).astype(np.int64)
    return float(L.oz_oracle_relative_residual_sampled_z(op_code(op_a), op_code(op_b), m, n, k, _p(a), _zld(a), _p(b),
                                                         _zld(b), _p(c), _zld(c), ns, _p(rows), _p(cols)))
