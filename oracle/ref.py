"""ctypes front-end of oracle/_ref/libref_config.so: the one translation unit of the reference that compiles in the build
container (/root/reference/src/config.cu, built where it lies by `make -C oracle ref`; see oracle/Makefile and
oracle/ref_config_shim.cpp).

TEST INFRASTRUCTURE ONLY (tests/test_ref_pin.py).  What it pins is integer tables: mode -> slice count, the ordered slice-pair
list, the padded plane geometry, the enum order of the public header.  The floating-point path of the reference (split.cu,
gemm.cu) stays unbuildable here, so the oracle's arithmetic remains "parity unpinned" (oracle/ozaki_oracle.h).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libref_config.so")
REFERENCE = os.environ.get("OZIMMU_REFERENCE_DIR", "/root/reference")

OP_N, OP_T = 0, 1


def buildable():
    return os.path.exists(os.path.join(REFERENCE, "src", "config.cu"))


def build(force=False):
    """compiles the reference's config.cu from /root/reference (build container only); returns the path or None when neither
    the reference tree nor a prebuilt library is there (the GPU box has the prebuilt file only)"""
    if buildable():
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref", "REF=" + REFERENCE] + (["-B"] if force else []))
    return LIB_PATH if os.path.exists(LIB_PATH) else None


_lib = None


def lib():
    global _lib
    if _lib is None:
        if build() is None:
            raise FileNotFoundError("oracle/_ref/libref_config.so: not built and /root/reference is absent")
        L = C.CDLL(LIB_PATH)
        i, u, vp = C.c_int, C.c_uint32, C.c_void_p
        L.ref_get_split_config.restype = i
        L.ref_get_split_config.argtypes = [i, vp, vp, vp, vp, vp, i]
        L.ref_split_type.restype = i
        L.ref_split_type.argtypes = [i, i, i]
        L.ref_gemm_mode_str.restype = i
        L.ref_gemm_mode_str.argtypes = [i, C.c_char_p, i]
        L.ref_padded_ld_i8.restype = u
        L.ref_padded_ld_i8.argtypes = [u]
        L.ref_slice_ld_i8.restype = u
        L.ref_slice_ld_i8.argtypes = [u, u, i]
        L.ref_slice_num_elements_i8.restype = u
        L.ref_slice_num_elements_i8.argtypes = [u, u, i]
        L.ref_enum_value.restype = i
        L.ref_enum_value.argtypes = [C.c_char_p]
        _lib = L
    return _lib


def enum_value(name):
    v = int(lib().ref_enum_value(name.encode()))
    if v == -1000:
        raise KeyError(name)
    return v


def split_config(mode):
    """get_split_config(mode) (src/config.cu:4-100): (A split types, B split types, [(A_id, B_id, gemm_mode), ...] in order)"""
    L = lib()
    cap = 512
    na, nb = C.c_int(0), C.c_int(0)
    a = (C.c_int * cap)()
    b = (C.c_int * cap)()
    g = (C.c_int * cap)()
    p = L.ref_get_split_config(int(mode), C.addressof(na), C.addressof(nb), C.addressof(a), C.addressof(b), C.addressof(g), cap)
    assert p >= 0
    ta = [int(L.ref_split_type(int(mode), 0, j)) for j in range(na.value)]
    tb = [int(L.ref_split_type(int(mode), 1, j)) for j in range(nb.value)]
    return ta, tb, [(a[j], b[j], g[j]) for j in range(p)]


def gemm_mode_str(gemm_mode):
    buf = C.create_string_buffer(64)
    n = lib().ref_gemm_mode_str(int(gemm_mode), buf, 64)
    assert n >= 0
    return buf.value.decode()


def padded_ld_i8(n):
    return int(lib().ref_padded_ld_i8(n))


def slice_ld_i8(m, n, op):
    return int(lib().ref_slice_ld_i8(m, n, op))


def slice_num_elements_i8(m, n, op):
    return int(lib().ref_slice_num_elements_i8(m, n, op))
