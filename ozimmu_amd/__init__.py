"""ozimmu_amd — Python host-side mirror of libozimmu_hip.so (Ozaki-scheme INT8-MFMA DGEMM for MI355X).

The product is the C-ABI shared library (include/ozimmu_hip.h): an LD_PRELOAD drop-in for
rocBLAS/hipBLAS DGEMM plus the direct API of the reference's include/ozimmu/ozimmu.hpp:47-100.
This module only binds that ABI with ctypes, using the reference's names and argument order
(`create`, `destroy`, `set_cuda_stream`, `gemm`, `auto_mode_select`, ...), so tests read like the
reference's own harness (test/main_test.cu:242).  PyTorch is used for device memory and streams only.

There is NO CPU fallback: if the HIP library is missing the import of any compute entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# libozimmu_hip.so is what ships: no test hook compiled in.  OZIMMU_HIP_LIBRARY points these bindings at another build.
LIB_PATH = os.environ.get("OZIMMU_HIP_LIBRARY") or os.path.join(_HERE, "libozimmu_hip.so")
# the same sources with -DOZIMMU_HIP_TEST_HOOKS (INT32 diagonal-sum dump, launch-failure injection, epoch jump): only the tests
# that drive those hooks load it, through test_flavour() below
TEST_LIB_PATH = os.path.join(_HERE, "libozimmu_hip_test.so")

# include/ozimmu/ozimmu.hpp:12 (op_c: conjugate transpose, include/ozimmu_hip.h - the reference has no such value)
op_n, op_t, op_c = 0, 1, 2
# include/ozimmu/ozimmu.hpp:14-37
(sgemm, dgemm, fp64_int8_3, fp64_int8_4, fp64_int8_5, fp64_int8_6, fp64_int8_7, fp64_int8_8,
 fp64_int8_9, fp64_int8_10, fp64_int8_11, fp64_int8_12, fp64_int8_13, fp64_int8_14, fp64_int8_15,
 fp64_int8_16, fp64_int8_17, fp64_int8_18, fp64_int8_auto) = range(19)
malloc_sync, malloc_async = 0, 1
real, complx = 0, 1
matrix_A, matrix_B = 0, 1

_lib = None


class OzimmuLibraryMissing(RuntimeError):
    pass


def lib():
    """Loads libozimmu_hip.so (built by `python -m ozimmu_amd.build`).  Import torch first in a process
    that uses both, so a single HIP runtime (libamdhip64.so.7) is shared."""
    global _lib
    if _lib is not None:
        return _lib
    try:  # PyTorch bundles its own libamdhip64.so.7: load it FIRST so both share one HIP runtime
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise OzimmuLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -m ozimmu_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, sz, i, d, u32 = C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_uint32
    L.ozimmu_hip_version.restype = C.c_char_p
    L.ozimmu_hip_create.restype = i
    L.ozimmu_hip_create.argtypes = [C.POINTER(vp), i]
    L.ozimmu_hip_destroy.restype = i
    L.ozimmu_hip_destroy.argtypes = [vp]
    L.ozimmu_hip_set_stream.restype = None
    L.ozimmu_hip_set_stream.argtypes = [vp, vp]
    for f in ("enable_profiling", "disable_profiling", "clear_profiler_result"):
        getattr(L, "ozimmu_hip_" + f).restype = None
        getattr(L, "ozimmu_hip_" + f).argtypes = [vp]
    L.ozimmu_hip_print_profiler_result.restype = None
    L.ozimmu_hip_print_profiler_result.argtypes = [vp, C.c_char_p, i]
    L.ozimmu_hip_set_auto_mantissa_loss_threashold.restype = None
    L.ozimmu_hip_set_auto_mantissa_loss_threashold.argtypes = [vp, d]
    L.ozimmu_hip_get_auto_mantissa_loss_threashold.restype = d
    L.ozimmu_hip_get_auto_mantissa_loss_threashold.argtypes = [vp]
    L.ozimmu_hip_reallocate_working_memory.restype = sz
    L.ozimmu_hip_reallocate_working_memory.argtypes = [vp, sz]
    L.ozimmu_hip_working_memory_size.restype = sz
    L.ozimmu_hip_working_memory_size.argtypes = [i, i, sz, sz, sz, i, i]
    L.ozimmu_hip_gemm.restype = i
    L.ozimmu_hip_gemm.argtypes = [vp, i, i, sz, sz, sz, vp, vp, sz, vp, sz, vp, vp, sz, i, i]
    L.ozimmu_hip_gemm_f32.restype = i
    ll = C.c_longlong
    L.ozimmu_hip_gemm_on_stream.restype = i
    L.ozimmu_hip_gemm_on_stream.argtypes = [vp, vp, i, i, sz, sz, sz, vp, vp, sz, vp, sz, vp, vp, sz, i, i]
    L.ozimmu_hip_gemm_strided_batched.restype = i
    L.ozimmu_hip_gemm_strided_batched.argtypes = [vp, vp, i, i, sz, sz, sz, vp, vp, sz, ll, vp, sz, ll, vp, vp, sz, ll,
                                                  sz, i, i]
    L.ozimmu_hip_gemm_f32.argtypes = [vp, i, i, sz, sz, sz, vp, vp, sz, vp, sz, vp, vp, sz, i]
    L.ozimmu_hip_auto_mode_select.restype = i
    L.ozimmu_hip_auto_mode_select.argtypes = [vp, i, i, sz, sz, sz, vp, sz, vp, sz, i, d]
    L.ozimmu_hip_get_compute_mode_name_str.restype = C.c_char_p
    L.ozimmu_hip_get_compute_mode_name_str.argtypes = [i]
    L.ozimmu_hip_compute_mode_from_str.restype = i
    L.ozimmu_hip_compute_mode_from_str.argtypes = [C.c_char_p]
    L.ozimmu_hip_get_bits_per_int8.restype = u32
    L.ozimmu_hip_get_bits_per_int8.argtypes = [u32]
    L.ozimmu_hip_get_num_split.restype = i
    L.ozimmu_hip_get_num_split.argtypes = [i]
    L.ozimmu_hip_split_int8.restype = i
    L.ozimmu_hip_split_int8.argtypes = [vp, vp, u32, vp, sz, sz, vp, sz, i, i, C.c_uint, C.c_uint]
    L.ozimmu_hip_diagonal_sums.restype = i
    L.ozimmu_hip_diagonal_sums.argtypes = [vp, i, i, sz, sz, sz, vp, sz, vp, sz, C.c_uint, vp]
    L.ozimmu_hip_mantissa_loss.restype = i
    L.ozimmu_hip_mantissa_loss.argtypes = [vp, i, i, sz, sz, sz, vp, sz, vp, sz, C.POINTER(C.c_uint64)]
    L.ozimmu_hip_native_dgemm.restype = i
    L.ozimmu_hip_native_dgemm.argtypes = [vp, i, i, sz, sz, sz, vp, vp, sz, vp, sz, vp, vp, sz]
    L.ozimmu_hip_device_info.restype = i
    L.ozimmu_hip_device_info.argtypes = [vp, C.POINTER(C.c_double)]
    L.ozimmu_hip_policy_predict.restype = i
    L.ozimmu_hip_policy_predict.argtypes = [vp, i, i, sz, sz, sz, sz, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.ozimmu_hip_policy_params.restype = i
    L.ozimmu_hip_policy_params.argtypes = [C.POINTER(C.c_double), i, i]
    L.ozimmu_hip_last_kernel.restype = i
    L.ozimmu_hip_last_kernel.argtypes = [vp, C.POINTER(C.c_int)]
    L.ozimmu_hip_last_stage_ms.restype = i
    L.ozimmu_hip_last_stage_ms.argtypes = [vp, C.POINTER(C.c_float)]
    _lib = L
    return L


def _ptr(x):
    """device pointer of a torch tensor, or an int passed through"""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    return x.data_ptr()


def _op(op):
    if isinstance(op, str):
        return {"N": op_n, "T": op_t, "C": op_c}[op.upper()]
    return int(op)


def _mode(mode):
    if isinstance(mode, str):
        m = lib().ozimmu_hip_compute_mode_from_str(mode.encode())
        if mode != "dgemm" and m == dgemm:
            raise ValueError(f"unknown compute mode {mode!r}")
        return m
    return int(mode)


class handle_t:
    """mtk::ozimmu::handle_t"""

    def __init__(self, ptr):
        self.ptr = ptr


def create(malloc_mode=malloc_sync):
    p = C.c_void_p()
    st = lib().ozimmu_hip_create(C.byref(p), malloc_mode)
    if st != 0 or not p.value:
        raise RuntimeError(f"ozimmu_hip_create failed (status {st}); is a GPU visible?")
    return handle_t(p)


def destroy(handle):
    if handle is not None and handle.ptr:
        lib().ozimmu_hip_destroy(handle.ptr)
        handle.ptr = None
    return 0


def set_cuda_stream(handle, stream):
    """`stream`: a torch.cuda.Stream, or a raw hipStream_t as int (0 = default stream)."""
    raw = getattr(stream, "cuda_stream", stream)
    lib().ozimmu_hip_set_stream(handle.ptr, C.c_void_p(int(raw) if raw else 0))


set_stream = set_cuda_stream


def enable_profiling(handle):
    lib().ozimmu_hip_enable_profiling(handle.ptr)


def disable_profiling(handle):
    lib().ozimmu_hip_disable_profiling(handle.ptr)


def print_profiler_result(handle, tag, csv=False):
    lib().ozimmu_hip_print_profiler_result(handle.ptr, tag.encode(), int(csv))


def clear_profiler_result(handle):
    lib().ozimmu_hip_clear_profiler_result(handle.ptr)


def last_stage_ms(handle):
    ms = (C.c_float * 3)()
    lib().ozimmu_hip_last_stage_ms(handle.ptr, ms)
    return {"split_A": ms[0], "split_B": ms[1], "int8tc": ms[2]}


def device_info(handle):
    """what the launch policy plans with on the handle's device (include/ozimmu_hip.h: ozimmu_hip_device_info)"""
    v = (C.c_double * 4)()
    if lib().ozimmu_hip_device_info(handle.ptr, v):
        raise RuntimeError("ozimmu_hip_device_info failed")
    return {"cus": int(v[0]), "xcds": int(v[1]), "mfma32_us": v[2], "mfma32_measured_us": v[3]}


# (16 = k2_one_launch: the K-split tile behind the split of the same call in ONE kernel, csrc/slice_gemm_one_launch.hip)
KERNEL_NAMES = {0: "k2", 1: "classic", 2: "wide", 3: "x16", 4: "k64", 12: "k64_breg", 16: "k2_one_launch", -1: None}
POLICY_PARAMS = 40


def policy_predict(handle, num_split, m, n, k, batch=1, pass_index=0):
    """the cost model's prediction (us) per kernel and the policy's pick (include/ozimmu_hip.h: ozimmu_hip_policy_predict)"""
    out = (C.c_double * 6)()
    pick = C.c_int(-1)
    if lib().ozimmu_hip_policy_predict(handle.ptr if handle is not None else None, num_split, pass_index, m, n, k, batch, out,
                                       C.byref(pick)):
        raise RuntimeError("ozimmu_hip_policy_predict failed")
    names = ["k2", "classic", "wide", "x16", "k64", "k64_breg"]
    return {nm: out[j] for j, nm in enumerate(names) if out[j] >= 0}, KERNEL_NAMES.get(pick.value, str(pick.value))


def policy_params(new=None):
    """read (new=None) or replace the fitted constants of the kernel-choice model"""
    v = (C.c_double * POLICY_PARAMS)()
    if new is not None:
        for j, x in enumerate(new):
            v[j] = float(x)
        if lib().ozimmu_hip_policy_params(v, len(new), 1):
            raise RuntimeError("ozimmu_hip_policy_params failed")
        return list(new)
    lib().ozimmu_hip_policy_params(v, POLICY_PARAMS, 0)
    return list(v)


def last_kernel(handle):
    """names of the kernels the last slice-GEMM launch of this handle ran (first / second diagonal pass)"""
    v = (C.c_int * 2)()
    lib().ozimmu_hip_last_kernel(handle.ptr, v)
    return [KERNEL_NAMES.get(v[0], str(v[0])), KERNEL_NAMES.get(v[1], str(v[1]))]


def mfma_ceiling(seconds=2.0, device=-1):
    """INT8 TOPS the matrix pipe sustains on v_mfma_i32_16x16x64_i8 alone (random operands, no memory traffic), steady state"""
    L = lib()
    L.ozimmu_hip_mfma_ceiling.restype = C.c_int
    L.ozimmu_hip_mfma_ceiling.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_double)]
    v = C.c_double(0.0)
    if L.ozimmu_hip_mfma_ceiling(device, float(seconds), C.byref(v)) != 0:
        raise RuntimeError("ozimmu_hip_mfma_ceiling failed")
    return v.value


def intercept_stats():
    """process-wide counters of the LD_PRELOAD boundary + the kernel histogram (include/ozimmu_hip.h: ozimmu_hip_intercept_stats)"""
    L = lib()
    out = (C.c_ulonglong * 28)()
    L.ozimmu_hip_intercept_stats.restype = C.c_int
    L.ozimmu_hip_intercept_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    L.ozimmu_hip_intercept_stats(out, 28)
    v = list(out)
    return {"seen": v[0], "taken": v[1], "declined": v[2], "failed": v[3],
            "kernels": {KERNEL_NAMES.get(c, str(c)): v[4 + c] for c in range(24) if v[4 + c]}}


def tuner_state(handle, mode, m, n, k, op_a=None, op_b=None, beta_nonzero=False, full=False):
    """measured kernel choice of this handle for a plain real GEMM shape: (state, slot, candidates); state -1 = unknown shape,
    0 = counting calls or measuring, 1 = decided on prediction slot `slot` (index into KERNELS of policy_predict).  Without
    op_a / op_b: the most recently used entry of any layout / beta class.  full=True: + (calls seen, measurements finished)"""
    v = (C.c_int * 4)()
    L = lib()
    L.ozimmu_hip_tuner_state_ex.restype = C.c_int
    L.ozimmu_hip_tuner_state_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t,
                                            C.POINTER(C.c_int)]
    S = int(str(mode).rsplit("_", 1)[1])
    oa = -1 if op_a is None else _op(op_a)
    ob = -1 if op_b is None else _op(op_b)
    st = L.ozimmu_hip_tuner_state_ex(handle.ptr, S, oa, ob, 1 if beta_nonzero else 0, m, n, k, v)
    return (st, v[0], v[1], v[2], v[3]) if full else (st, v[0], v[1])


def set_auto_mantissa_loss_threashold(handle, threshold):
    lib().ozimmu_hip_set_auto_mantissa_loss_threashold(handle.ptr, float(threshold))


def get_auto_mantissa_loss_threashold(handle):
    return float(lib().ozimmu_hip_get_auto_mantissa_loss_threashold(handle.ptr))


def reallocate_working_memory(handle, size_or_gemm_list):
    """size in bytes, or the reference's gemm_list_t: [(op_A, op_B, m, n, k, element_kind, mode), ...]"""
    if isinstance(size_or_gemm_list, int):
        size = size_or_gemm_list
    else:
        size = 0
        for (oa, ob, m, n, k, kind, mode) in size_or_gemm_list:
            size = max(size, lib().ozimmu_hip_working_memory_size(_op(oa), _op(ob), m, n, k, kind, _mode(mode)))
    return int(lib().ozimmu_hip_reallocate_working_memory(handle.ptr, size))


def gemm(handle, op_A, op_B, m, n, k, alpha, a_ptr, lda, b_ptr, ldb, beta, c_ptr, ldc, compute_mode,
         element_kind=real):
    """mtk::ozimmu::gemm (include/ozimmu/ozimmu.hpp:75-82).  alpha/beta: Python floats (host scalars);
    a/b/c: column-major device buffers (torch tensors or raw pointers).  Returns the int status."""
    if element_kind == complx:  # cuDoubleComplex scalars: {re, im}
        al = (C.c_double * 2)(complex(alpha).real, complex(alpha).imag)
        be = (C.c_double * 2)(complex(beta).real, complex(beta).imag)
    else:
        al, be = C.c_double(alpha), C.c_double(beta)
    return int(lib().ozimmu_hip_gemm(handle.ptr, _op(op_A), _op(op_B), m, n, k, C.addressof(al), _ptr(a_ptr), lda,
                                     _ptr(b_ptr), ldb, C.addressof(be), _ptr(c_ptr), ldc, _mode(compute_mode),
                                     element_kind))


def _scalars(alpha, beta, element_kind):
    if element_kind == complx:
        return ((C.c_double * 2)(complex(alpha).real, complex(alpha).imag),
                (C.c_double * 2)(complex(beta).real, complex(beta).imag))
    return C.c_double(alpha), C.c_double(beta)


def _stream_ptr(stream):
    return C.c_void_p(getattr(stream, "cuda_stream", stream) or None)


def gemm_on_stream(handle, stream, op_A, op_B, m, n, k, alpha, a_ptr, lda, b_ptr, ldb, beta, c_ptr, ldc, compute_mode,
                   element_kind=real):
    """gemm with the stream passed along (stream switch and enqueue under one lock); stream: torch stream or raw handle"""
    al, be = _scalars(alpha, beta, element_kind)
    return int(lib().ozimmu_hip_gemm_on_stream(handle.ptr, _stream_ptr(stream), _op(op_A), _op(op_B), m, n, k,
                                               C.addressof(al), _ptr(a_ptr), lda, _ptr(b_ptr), ldb, C.addressof(be),
                                               _ptr(c_ptr), ldc, _mode(compute_mode), element_kind))


def gemm_strided_batched(handle, stream, op_A, op_B, m, n, k, alpha, a_ptr, lda, stride_a, b_ptr, ldb, stride_b, beta,
                         c_ptr, ldc, stride_c, batch_count, compute_mode, element_kind=real):
    """cublas{D,Z}gemmStridedBatched semantics (src/cublas.cu:315-512); strides in elements"""
    al, be = _scalars(alpha, beta, element_kind)
    return int(lib().ozimmu_hip_gemm_strided_batched(
        handle.ptr, _stream_ptr(stream), _op(op_A), _op(op_B), m, n, k, C.addressof(al), _ptr(a_ptr), lda, stride_a,
        _ptr(b_ptr), ldb, stride_b, C.addressof(be), _ptr(c_ptr), ldc, stride_c, batch_count, _mode(compute_mode),
        element_kind))


def gemm_f32(handle, op_A, op_B, m, n, k, alpha, a_ptr, lda, b_ptr, ldb, beta, c_ptr, ldc, element_kind=real):
    """mtk::ozimmu::dgemm_f32 (src/cublas_helper.cu:83-133): the `sgemm` compute mode."""
    if element_kind == complx:
        al = (C.c_double * 2)(complex(alpha).real, complex(alpha).imag)
        be = (C.c_double * 2)(complex(beta).real, complex(beta).imag)
    else:
        al, be = C.c_double(alpha), C.c_double(beta)
    return int(lib().ozimmu_hip_gemm_f32(handle.ptr, _op(op_A), _op(op_B), m, n, k, C.addressof(al), _ptr(a_ptr), lda,
                                         _ptr(b_ptr), ldb, C.addressof(be), _ptr(c_ptr), ldc, element_kind))


def native_dgemm(handle, op_A, op_B, m, n, k, alpha, a_ptr, lda, b_ptr, ldb, beta, c_ptr, ldc):
    al, be = C.c_double(alpha), C.c_double(beta)
    return int(lib().ozimmu_hip_native_dgemm(handle.ptr, _op(op_A), _op(op_B), m, n, k, C.addressof(al),
                                             _ptr(a_ptr), lda, _ptr(b_ptr), ldb, C.addressof(be), _ptr(c_ptr), ldc))


def auto_mode_select(handle, op_A, op_B, m, n, k, a_ptr, lda, b_ptr, ldb, element_kind, mantissa_loss_threshold):
    return int(lib().ozimmu_hip_auto_mode_select(handle.ptr, _op(op_A), _op(op_B), m, n, k, _ptr(a_ptr), lda,
                                                 _ptr(b_ptr), ldb, element_kind, float(mantissa_loss_threshold)))


def mantissa_loss(handle, op_A, op_B, m, n, k, a_ptr, lda, b_ptr, ldb):
    cnt = (C.c_uint64 * 16)()
    st = lib().ozimmu_hip_mantissa_loss(handle.ptr, _op(op_A), _op(op_B), m, n, k, _ptr(a_ptr), lda, _ptr(b_ptr),
                                        ldb, cnt)
    if st:
        raise RuntimeError(f"ozimmu_hip_mantissa_loss failed ({st})")
    return list(cnt)


def split_int8(handle, out_ptr, ldo, max_exp_ptr, m, n, in_ptr, ld, op, matrix, num_split, bits_per_int8):
    """mtk::ozimmu::split_int8<double> (src/split.hpp:12-19); reference output layout."""
    return int(lib().ozimmu_hip_split_int8(handle.ptr, _ptr(out_ptr), ldo, _ptr(max_exp_ptr), m, n, _ptr(in_ptr), ld,
                                           _op(op), matrix, num_split, bits_per_int8))


def diagonal_sums(handle, op_A, op_B, m, n, k, a_ptr, lda, b_ptr, ldb, num_split, out_ptr):
    return int(lib().ozimmu_hip_diagonal_sums(handle.ptr, _op(op_A), _op(op_B), m, n, k, _ptr(a_ptr), lda,
                                              _ptr(b_ptr), ldb, num_split, _ptr(out_ptr)))


def get_compute_mode_name_str(mode):
    s = lib().ozimmu_hip_get_compute_mode_name_str(int(mode))
    if s is None:
        raise ValueError("Not implemented (invalid compute mode)")
    return s.decode()


def get_bits_per_int8(k):
    return int(lib().ozimmu_hip_get_bits_per_int8(int(k)))


def get_num_split(mode):
    return int(lib().ozimmu_hip_get_num_split(_mode(mode)))


def version():
    return lib().ozimmu_hip_version().decode()


PRELOAD_ENV = {"LD_PRELOAD": LIB_PATH}

_test_flavour = None


def test_flavour():
    """A second, independent copy of these bindings over libozimmu_hip_test.so (own library image, own handles): the module
    source executed again under another name with LIB_PATH = TEST_LIB_PATH.  For the tests that need a hook; everything
    else - including every parity test that only calls the product API - stays on the library that ships."""
    global _test_flavour
    if _test_flavour is None:
        import importlib.util
        spec = importlib.util.spec_from_file_location("ozimmu_amd_test_flavour", os.path.abspath(__file__))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.LIB_PATH = TEST_LIB_PATH
        mod.PRELOAD_ENV = {"LD_PRELOAD": TEST_LIB_PATH}
        _test_flavour = mod
    return _test_flavour
