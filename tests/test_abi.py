"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol that
include/ozimmu_hip.h declares plus the interposed rocBLAS/hipBLAS entry points, and its host-only logic
(mode names, slice width, workspace sizing) follows the reference.  No compute calls here."""
import ctypes
import os
import re
import subprocess

import pytest

import ozimmu_amd
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ozimmu_hip.h")

_FAMILIES = ["rocblas_dgemm", "rocblas_zgemm", "rocblas_gemm_ex", "rocblas_dgemm_strided_batched",
             "rocblas_zgemm_strided_batched", "rocblas_gemm_strided_batched_ex", "hipblasDgemm", "hipblasZgemm",
             "hipblasGemmEx", "hipblasGemmExWithFlags", "hipblasDgemmStridedBatched", "hipblasZgemmStridedBatched",
             "hipblasGemmStridedBatchedEx", "hipblasGemmStridedBatchedExWithFlags"]
# every FP64 GEMM entry point the vendor libraries export, 32-bit and ILP64 (`_64`) index twins
INTERPOSED = ["rocblas_create_handle", "rocblas_destroy_handle", "hipblasCreate", "hipblasDestroy"] + _FAMILIES + [f + "_64" for f in _FAMILIES]


@pytest.fixture(scope="module")
def lib():
    from ozimmu_amd import build
    build.build()
    return ozimmu_amd.lib()


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ozimmu_hip_\w+)\s*\(", text)))


def test_header_declares_the_reference_api():
    names = declared_functions()
    # include/ozimmu/ozimmu.hpp:47-100, flattened to C
    for f in ["create", "destroy", "set_stream", "enable_profiling", "disable_profiling", "print_profiler_result",
              "clear_profiler_result", "set_auto_mantissa_loss_threashold", "get_auto_mantissa_loss_threashold",
              "reallocate_working_memory", "gemm", "auto_mode_select", "get_compute_mode_name_str",
              "get_output_type", "get_data_size_in_byte", "get_bits_per_int8"]:
        assert "ozimmu_hip_" + f in names, f


def test_output_type_and_data_size_follow_the_reference(lib):
    """mtk::ozimmu::get_output_type / get_data_size_in_byte (include/ozimmu/ozimmu.hpp:96-98, src/handle.cu:195-245)"""
    lib.ozimmu_hip_get_output_type.restype = ctypes.c_int
    lib.ozimmu_hip_get_output_type.argtypes = [ctypes.c_int]
    lib.ozimmu_hip_get_data_size_in_byte.restype = ctypes.c_size_t
    lib.ozimmu_hip_get_data_size_in_byte.argtypes = [ctypes.c_int]
    FP64, FP32, FP16, INT8, ORIGINAL, NONE = range(6)          # data_t, ozimmu.hpp:38
    assert lib.ozimmu_hip_get_output_type(0) == FP32           # sgemm
    for mode in range(1, 19):                                  # dgemm, fp64_int8_3..18, fp64_int8_auto
        assert lib.ozimmu_hip_get_output_type(mode) == FP64
    assert lib.ozimmu_hip_get_output_type(19) == ORIGINAL and lib.ozimmu_hip_get_output_type(-1) == ORIGINAL
    assert [lib.ozimmu_hip_get_data_size_in_byte(d) for d in (FP64, FP32, FP16, INT8, ORIGINAL, NONE, 99)] == [8, 4, 2, 1, 0, 0, 0]
    # a forwarder written against the reference sizes its output buffer as m * n * get_data_size_in_byte(get_output_type(mode))
    assert lib.ozimmu_hip_get_data_size_in_byte(lib.ozimmu_hip_get_output_type(8)) == 8


def test_library_exports_every_declared_symbol(lib):
    for name in declared_functions() + INTERPOSED:
        assert hasattr(lib, name), f"{name} is declared in include/ozimmu_hip.h but not exported"


@pytest.mark.parametrize("flavour", ["product", "test"])
def test_exports_are_unmangled_c_symbols(lib, flavour):
    """both flavours of the library (ozimmu_amd/build.py) export the same C ABI: what ships, and the build with the test hooks"""
    path = ozimmu_amd.LIB_PATH if flavour == "product" else ozimmu_amd.TEST_LIB_PATH
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    for name in declared_functions() + INTERPOSED:
        assert name in exported


def test_bindings_load_the_library_that_ships(lib):
    """`ozimmu_amd.lib()` - what bench.py, smoke() and the parity tests run - is libozimmu_hip.so, built WITHOUT
    -DOZIMMU_HIP_TEST_HOOKS: its hook entry points say so, and the hook environment variables are not even read"""
    assert os.path.basename(ozimmu_amd.LIB_PATH) == "libozimmu_hip.so" or os.environ.get("OZIMMU_HIP_LIBRARY")
    lib.ozimmu_hip_tile_plan.restype = ctypes.c_int
    lib.ozimmu_hip_tile_plan.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_double)]
    out = (ctypes.c_double * 3)()
    assert lib.ozimmu_hip_tile_plan(512, 512, 2, 256, 1, out) == 2          # the literal simulation is a test hook
    blob = open(ozimmu_amd.LIB_PATH, "rb").read()
    hooked = open(ozimmu_amd.TEST_LIB_PATH, "rb").read()
    for name in (b"OZIMMU_HIP_TEST_FAIL_LAUNCH", b"OZIMMU_HIP_TEST_EXP_EPOCH", b"OZIMMU_HIP_TEST_NO_STREAM_ORDER"):
        assert name not in blob and name in hooked


def test_interposed_surface_covers_the_vendor_fp64_gemm_exports():
    """every {d,z}gemm / gemm_ex (plain, strided-batched, _64, WithFlags) symbol that the installed rocBLAS / hipBLAS
    export is defined by the shim: a caller of any of them reaches the intercept predicate"""
    import glob
    for lib_glob, pat in (("/opt/rocm/lib/librocblas.so.*", r" T (rocblas_(?:d|z)?gemm(?:_strided_batched)?(?:_ex)?(?:_64)?)$"),
                          ("/opt/rocm/lib/libhipblas.so.*", r" T (hipblas(?:D|Z)?[Gg]emm(?:StridedBatched)?(?:Ex)?(?:WithFlags)?(?:_64)?)$")):
        libs = sorted(glob.glob(lib_glob))
        if not libs:
            pytest.skip("vendor library not installed")
        out = subprocess.check_output(["nm", "-D", "--defined-only", libs[0]], text=True)
        names = set(re.findall(pat, out, flags=re.M))
        names = {n for n in names if not re.match(r"(rocblas_gemm|hipblasGemm)$", n)}
        assert names, lib_glob
        missing = sorted(names - set(INTERPOSED))
        assert not missing, missing


def test_no_link_time_dependency_on_vendor_blas():
    """originals are resolved with dlsym(RTLD_NEXT) at run time (src/utils.hpp:117-141), never linked"""
    out = subprocess.check_output(["readelf", "-d", ozimmu_amd.LIB_PATH], text=True)
    needed = re.findall(r"NEEDED.*\[(.*?)\]", out)
    assert not any("rocblas" in n or "hipblas" in n for n in needed), needed
    assert any("amdhip64" in n for n in needed)


def test_compute_mode_enum_and_names(lib):
    # enum order of include/ozimmu/ozimmu.hpp:14-37
    expect = ["sgemm", "dgemm"] + [f"fp64_int8_{s}" for s in range(3, 19)] + ["fp64_int8_auto"]
    for i, name in enumerate(expect):
        assert ozimmu_amd.get_compute_mode_name_str(i) == name
        assert lib.ozimmu_hip_compute_mode_from_str(name.encode()) == i
        assert O.num_split_from_mode(name) == {"sgemm": -2, "dgemm": -1, "fp64_int8_auto": 0}.get(
            name, ozimmu_amd.get_num_split(i))
    assert lib.ozimmu_hip_get_compute_mode_name_str(19) is None
    # src/cublas.cu:18-48: unknown / unset -> dgemm
    for bad in [b"", b"fp64_int8_2", b"fp64_int8_19", b"FP64_INT8_9", b"auto"]:
        assert lib.ozimmu_hip_compute_mode_from_str(bad) == ozimmu_amd.dgemm
    assert lib.ozimmu_hip_compute_mode_from_str(None) == ozimmu_amd.dgemm
    assert ozimmu_amd.fp64_int8_9 == 8 and ozimmu_amd.fp64_int8_auto == 18


def test_bits_per_int8_matches_oracle(lib):
    for k in list(range(0, 70)) + [1000, 1 << 17, (1 << 17) + 1, 1 << 19, (1 << 19) + 1, 1 << 21, 1 << 23,
                                   (1 << 23) + 5, 1 << 25, 1 << 27, 1 << 29, (1 << 30)]:
        assert ozimmu_amd.get_bits_per_int8(k) == O.bits_per_int8(k), k


def test_working_memory_size(lib):
    # grows with S, m, n, k; zero for non-Ozaki modes (the workspace is library-owned: src/handle.cu:95-144)
    f = lib.ozimmu_hip_working_memory_size
    base = f(0, 0, 1024, 1024, 1024, ozimmu_amd.real, ozimmu_amd.fp64_int8_9)
    assert base >= 2 * 9 * 1024 * 1024            # at least the slice planes
    assert base < 2 * 9 * 1024 * 1024 * 1.2 + (1 << 20)
    assert f(0, 0, 1024, 1024, 1024, ozimmu_amd.real, ozimmu_amd.fp64_int8_12) > base   # + FP64 partials
    assert f(0, 0, 2048, 1024, 1024, ozimmu_amd.real, ozimmu_amd.fp64_int8_9) > base
    assert f(0, 0, 1024, 1024, 1024, ozimmu_amd.real, ozimmu_amd.dgemm) == 0
    # complex: Re and Im planes of both operands (src/config.cu:145: "* (real ? 1 : 2)")
    assert 1.9 * base < f(0, 0, 1024, 1024, 1024, ozimmu_amd.complx, ozimmu_amd.fp64_int8_9) < 2.1 * base
    assert f(0, 0, 1024, 1024, 1024, ozimmu_amd.real, ozimmu_amd.fp64_int8_auto) >= \
        f(0, 0, 1024, 1024, 1024, ozimmu_amd.real, ozimmu_amd.fp64_int8_18)


def test_null_handle_is_rejected_not_dereferenced(lib):
    """the reference dereferences its global handle unchecked (src/cublas.cu:144); the C ABI must not"""
    al = ctypes.c_double(1.0)
    assert lib.ozimmu_hip_gemm(None, 0, 0, 4, 4, 4, ctypes.addressof(al), None, 4, None, 4, ctypes.addressof(al),
                               None, 4, ozimmu_amd.fp64_int8_6, ozimmu_amd.real) == 1
    assert lib.ozimmu_hip_destroy(None) == 0
    assert lib.ozimmu_hip_reallocate_working_memory(None, 1 << 20) == 0
    lib.ozimmu_hip_set_stream(None, None)
    out = (ctypes.c_int * 2)()
    lib.ozimmu_hip_tuner_state.restype = ctypes.c_int
    lib.ozimmu_hip_tuner_state.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                                           ctypes.POINTER(ctypes.c_int)]
    assert lib.ozimmu_hip_tuner_state(None, 9, 64, 64, 64, out) == -1  # (the measured kernel choice: no handle, no table)


def test_product_has_no_cpu_fallback():
    """the package never imports the oracle; a missing library is a loud error"""
    src = open(os.path.join(ROOT, "ozimmu_amd", "__init__.py")).read()
    for f in os.listdir(os.path.join(ROOT, "ozimmu_amd", "csrc")):
        text = open(os.path.join(ROOT, "ozimmu_amd", "csrc", f)).read()
        assert "ozaki_oracle" not in text and "liboz_oracle" not in text, f
    assert "liboz_oracle" not in src and "import oracle" not in src and "from oracle" not in src
    import importlib
    saved = ozimmu_amd.LIB_PATH
    try:
        ozimmu_amd._lib = None
        ozimmu_amd.LIB_PATH = saved + ".does-not-exist"
        with pytest.raises(ozimmu_amd.OzimmuLibraryMissing):
            ozimmu_amd.lib()
    finally:
        ozimmu_amd.LIB_PATH = saved
        ozimmu_amd._lib = None
        importlib.reload(ozimmu_amd)


def test_tile_plan_closed_form_matches_the_literal_dispatch_simulation(lib):
    """csrc/tile_plan.h: the per-call row partition is searched with a closed-form makespan (microseconds for a
    32768 x 32768 output); the test flavour also carries the literal round-by-round simulation it replaced - same
    partition, same makespan, on shapes from one tile to 131072 tiles.  Host arithmetic only."""
    import random
    import time
    lib = ozimmu_amd.test_flavour().lib()
    lib.ozimmu_hip_tile_plan.restype = ctypes.c_int
    lib.ozimmu_hip_tile_plan.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_double)]
    rng = random.Random(7)
    shapes = [(1, 1), (32, 128), (33, 129), (1024, 1024), (1536, 1536), (4096, 4096), (8192, 8192), (12345, 777),
              (32768, 32768)] + [(rng.randrange(1, 20000), rng.randrange(1, 20000)) for _ in range(40)]
    out_new, out_ref = (ctypes.c_double * 3)(), (ctypes.c_double * 3)()
    for m, n in shapes:
        for wa in (1, 2, 3, 4):
            for cus in (256, 64, 7):
                assert lib.ozimmu_hip_tile_plan(m, n, wa, cus, 0, out_new) == 0
                assert lib.ozimmu_hip_tile_plan(m, n, wa, cus, 1, out_ref) == 0
                assert (out_new[0], out_new[1]) == (out_ref[0], out_ref[1]), (m, n, wa, cus)
                assert abs(out_new[2] - out_ref[2]) < 1e-6
                rows32 = (m + 31) // 32
                assert out_new[0] * wa + out_new[1] * (wa - 1) >= rows32          # the partition covers every row block
    t0 = time.perf_counter()
    assert lib.ozimmu_hip_tile_plan(32768, 32768, 2, 256, 0, out_new) == 0
    assert time.perf_counter() - t0 < 2e-3                                        # was 22 ms
    assert lib.ozimmu_hip_tile_plan(0, 5, 2, 256, 0, out_new) == 1
    assert lib.ozimmu_hip_tile_plan(5, 5, 0, 256, 0, out_new) == 1


def test_kernel_choice_cost_model_is_host_arithmetic(lib):
    """csrc/kernel_policy.cpp: the launch policy predicts the time of every kernel a pass is built for and takes the fastest.
    The prediction needs no GPU (NULL handle = the nominal 256-CU device): eligibility follows the kernels' constraints, the
    pick is the minimum, a forced kernel wins where it exists, and the constants can be read and replaced at run time."""
    pred, pick = ozimmu_amd.policy_predict(None, 9, 8192, 8192, 8192)
    assert set(pred) == {"classic", "wide", "k64", "k64_breg"}          # K-split: at most one 64x64 tile per CU; x16: S >= 10
    assert pick == min(pred, key=pred.get) == "k64_breg"                 # the headline shape runs the k64 tile, B in registers
    assert 10e3 < pred["k64_breg"] < 20e3                                # ~14 ms
    pred, pick = ozimmu_amd.policy_predict(None, 9, 8192, 8192, 8192 + 64)   # k-blocks = 2 mod 4: no register form
    assert "k64_breg" not in pred and "k64" in pred
    pred, _ = ozimmu_amd.policy_predict(None, 9, 8192, 8192, 8192 + 32)      # (planes pad odd k-block counts beyond 32 to even)
    assert "k64" in pred
    pred, _ = ozimmu_amd.policy_predict(None, 9, 512, 512, 96)               # 3 k-blocks: no k64 tile; few tiles: K-split exists
    assert "k64" not in pred and "k2" not in pred                           # (K-split needs 4+ k-blocks)
    pred, pick = ozimmu_amd.policy_predict(None, 9, 1024, 1024, 1024)
    assert "k2" in pred and pick == min(pred, key=pred.get)
    k2_before = pred["k2"]
    pred, _ = ozimmu_amd.policy_predict(None, 13, 4096, 4096, 4096, pass_index=1)   # second diagonal pass: no k64 tile
    assert "k64" not in pred and "wide" in pred
    with pytest.raises(RuntimeError):
        ozimmu_amd.policy_predict(None, 9, 4096, 4096, 4096, pass_index=1)    # fp64_int8_9 has one pass
    # time grows with the work
    t = [min(ozimmu_amd.policy_predict(None, 9, n, n, n)[0].values()) for n in (1024, 2048, 4096, 8192)]
    assert t == sorted(t) and t[3] / t[2] > 6
    # the table can be replaced (tools/policy_fit.py does that in its loop) and restored
    p0 = ozimmu_amd.policy_params()
    assert len(p0) == ozimmu_amd.POLICY_PARAMS
    p1 = list(p0)
    p1[0] *= 2                                                               # K-split: twice as slow per MFMA
    ozimmu_amd.policy_params(p1)
    try:
        assert ozimmu_amd.policy_predict(None, 9, 1024, 1024, 1024)[0]["k2"] > 1.3 * k2_before
        assert ozimmu_amd.policy_params()[0] == p1[0]
    finally:
        ozimmu_amd.policy_params(p0)
    assert ozimmu_amd.policy_params() == p0


def test_every_pick_has_a_forced_gpu_arm(lib, monkeypatch):
    """tests/test_gpu_forced_kernels.py forces every kernel the policy can pick (kernel_policy.h: Pick + the k64 register form)
    against the oracle.  Here, without a GPU: every name the library can report has an arm, and every arm's switches make the
    cost model (host arithmetic) pick exactly that kernel on every shape / mode the GPU tests run it on - a kernel added to the
    policy without a forced parity test, or an arm that silently falls back to another kernel, fails this test."""
    import importlib
    F = importlib.import_module("tests.test_gpu_forced_kernels")
    names = {v for v in ozimmu_amd.KERNEL_NAMES.values() if v}
    assert names == set(F.FORCED_ARMS) == set(F.ARM_MODES)
    # the C++ enum behind the names: one name per Pick value + the register form
    src = open(os.path.join(ROOT, "ozimmu_amd", "csrc", "kernel_policy.h")).read()
    picks = re.search(r"enum class Pick \{([^}]*)\}", src).group(1)
    # (+ the one-launch form of the K-split tile, which the cost model sees as "k2": slice_gemm_one_launch.hip asks it)
    assert len(re.findall(r"\w+\s*=\s*\d+", picks)) + 2 == len(names)
    monkeypatch.setenv("OZIMMU_HIP_ENV_PER_CALL", "1")
    for arm, S in F._arm_cases():
        for k in ("OZIMMU_HIP_GEMM_KERNEL", "OZIMMU_HIP_PAIRED_TILE", "OZIMMU_HIP_K64_BREG", "OZIMMU_HIP_K64_TILE", "OZIMMU_HIP_ONE_LAUNCH"):
            monkeypatch.delenv(k, raising=False)
        for k, v in F.FORCED_ARMS[arm].items():
            monkeypatch.setenv(k, v)
        for m, n, k in F.SHAPES + [(1024, 1024, 1024), (700, 520, 128), (333, 900, 256), (2048, 1536, 1024)]:
            if arm in ("k2", "k2_one_launch") and m * n > 256 * 4096 and (m, n, k) not in F.SHAPES:
                continue
            _, pick = ozimmu_amd.policy_predict(None, S, m, n, k)
            assert pick == ("k2" if arm == "k2_one_launch" else arm), (arm, S, m, n, k, pick)


def test_launch_policy_is_bound_to_the_handles_device_id(lib):
    """csrc/topology.h: the topology a launch plans with is the slot of the device its handle was created on
    (SliceGemmArgs::device), not whatever device is current.  Two fake devices (test flavour: the slots can be set) with
    different CU / XCD counts give different plans for the same product, each independent of the other slot and of the calling
    thread's current device (there is none on this box); the library that ships refuses to have its slots set."""
    T = ozimmu_amd.test_flavour().lib()
    P = ozimmu_amd.lib()
    for L in (T, P):
        L.ozimmu_hip_device_topology.restype = ctypes.c_int
        L.ozimmu_hip_device_topology.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
        L.ozimmu_hip_policy_predict_device.restype = ctypes.c_int
        L.ozimmu_hip_policy_predict_device.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t,
                                                       ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double),
                                                       ctypes.POINTER(ctypes.c_int)]
    v = (ctypes.c_double * 4)(64, 2, 0.0194, 0)
    assert P.ozimmu_hip_device_topology(1, v, 1) == 2                       # product: read-only
    assert T.ozimmu_hip_device_topology(5, (ctypes.c_double * 4)(256, 8, 0.0194, 0), 1) == 0
    assert T.ozimmu_hip_device_topology(6, v, 1) == 0
    assert T.ozimmu_hip_device_topology(64, v, 1) == 1 and T.ozimmu_hip_device_topology(-1, v, 0) == 1
    got = (ctypes.c_double * 4)()
    assert T.ozimmu_hip_device_topology(6, got, 0) == 0 and list(got)[:2] == [64.0, 2.0]
    assert T.ozimmu_hip_device_topology(5, got, 0) == 0 and list(got)[:2] == [256.0, 8.0]

    def predict(dev):
        out, pick = (ctypes.c_double * 6)(), ctypes.c_int(-1)
        assert T.ozimmu_hip_policy_predict_device(dev, 9, 0, 4096, 4096, 4096, 1, out, ctypes.byref(pick)) == 0
        return list(out), pick.value
    big, small = predict(5), predict(6)
    best = lambda r: min(x for x in r[0] if x >= 0)
    assert 3.0 < best(small) / best(big) < 5.0                               # a quarter of the CUs: about four times as long
    assert predict(5) == big                                                 # ... and slot 6 did not leak into slot 5
    # 1024^3: 256 tiles of 64 x 64 fit the 256-CU device's K-split kernel (one tile per CU) and not the 64-CU one's
    out, pick = (ctypes.c_double * 6)(), ctypes.c_int(-1)
    assert T.ozimmu_hip_policy_predict_device(5, 9, 0, 1024, 1024, 1024, 1, out, ctypes.byref(pick)) == 0 and out[0] >= 0
    assert T.ozimmu_hip_policy_predict_device(6, 9, 0, 1024, 1024, 1024, 1, out, ctypes.byref(pick)) == 0 and out[0] < 0
