"""Pins against the REFERENCE ITSELF, compiled: oracle/_ref/libref_config.so is /root/reference/src/config.cu built where it lies
(oracle/Makefile `ref`, oracle/ref_config_shim.cpp) - the one translation unit of the reference this image can compile (the
others need nvcc / cuBLAS / the un-vendored cutf headers).  It pins integer tables only:

    §8 A2   mode -> slice count S and the slice-pair list IN ORDER      src/config.cu:4-100
    §8 A5/A9 padded plane geometry                                      src/utils.hpp:30-72
    §8 (b)  the enum order the C ABI mirrors                            include/ozimmu/ozimmu.hpp:12-46, src/config.hpp:9-18

against (1) the CPU oracle's restatement and (2) the product's own tables (include/ozimmu_hip.h, libozimmu_hip.so - no GPU
needed).  The arithmetic rows A3-A8 and A11 stay unpinned by compiled reference code.
Build container only: skipped where neither /root/reference nor a prebuilt oracle/_ref exists."""
import os
import re

import pytest

from oracle import oracle as O
from oracle import ref as R

pytestmark = pytest.mark.skipif(R.build() is None, reason="oracle/_ref not built and /root/reference absent")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = ["sgemm", "dgemm"] + [f"fp64_int8_{s}" for s in range(3, 19)] + ["fp64_int8_auto"]


def test_ref_library_is_the_reference_compiled_not_a_restatement():
    """the recipe compiles $(REF)/src/config.cu itself; nothing under oracle/ holds a copy of it"""
    mk = open(os.path.join(ROOT, "oracle", "Makefile")).read()
    assert "$(REF)/src/config.cu -o _ref/config.o" in mk and "-z,defs" in mk
    shim = open(os.path.join(ROOT, "oracle", "ref_config_shim.cpp")).read()
    assert "gemm_pair_list" not in shim and "push_back" not in shim           # no pair loop of its own
    assert "get_data_size_in_byte(" not in shim.replace("calls get_data_size_in_byte", "")  # no stand-in symbol
    for f in os.listdir(os.path.join(ROOT, "oracle")):
        assert not f.endswith(".cu"), f


def test_mode_enum_order_matches_reference():
    import ozimmu_amd
    for i, name in enumerate(MODES):
        assert R.enum_value(name) == i
        assert getattr(ozimmu_amd, name) == i
        assert ozimmu_amd.get_compute_mode_name_str(i) == name
    assert (R.enum_value("op_n"), R.enum_value("op_t")) == (O.OP_N, O.OP_T) == (0, 1)
    assert (R.enum_value("real"), R.enum_value("complx")) == (ozimmu_amd.real, ozimmu_amd.complx)
    assert (R.enum_value("malloc_sync"), R.enum_value("malloc_async")) == (ozimmu_amd.malloc_sync, ozimmu_amd.malloc_async)


def _header_enums():
    """enumerator -> value of include/ozimmu_hip.h, by the C rule (previous + 1 unless given)"""
    text = open(os.path.join(ROOT, "include", "ozimmu_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for body in re.findall(r"typedef\s+enum\s*\{(.*?)\}", text, flags=re.S):
        v = -1
        for item in [x.strip() for x in body.split(",") if x.strip()]:
            name, _, val = [x.strip() for x in item.partition("=")]
            v = int(val, 0) if val else v + 1
            out[name] = v
    return out


def test_c_header_enums_match_reference():
    e = _header_enums()
    for name in MODES:
        assert e["OZIMMU_" + name.upper()] == R.enum_value(name), name
    for ref_name, ours in [("op_n", "OZIMMU_OP_N"), ("op_t", "OZIMMU_OP_T"), ("malloc_sync", "OZIMMU_MALLOC_SYNC"),
                           ("malloc_async", "OZIMMU_MALLOC_ASYNC"), ("real", "OZIMMU_REAL"), ("complx", "OZIMMU_COMPLX"),
                           ("fp64", "OZIMMU_DATA_FP64"), ("fp32", "OZIMMU_DATA_FP32"), ("fp16", "OZIMMU_DATA_FP16"),
                           ("int8", "OZIMMU_DATA_INT8"), ("original", "OZIMMU_DATA_ORIGINAL"), ("none", "OZIMMU_DATA_NONE"),
                           ("matrix_A", "OZIMMU_MATRIX_A"), ("matrix_B", "OZIMMU_MATRIX_B")]:
        assert e[ours] == R.enum_value(ref_name), ref_name
    assert e["OZIMMU_OP_C"] == 2      # the extension (conjugate transpose), outside the reference's range


@pytest.mark.parametrize("mode", range(len(MODES)))
def test_split_config_matches_oracle_and_product(mode):
    import ozimmu_amd
    name = MODES[mode]
    ta, tb, pairs = R.split_config(mode)
    int8, original = R.enum_value("int8"), R.enum_value("original")
    if name in ("sgemm", "dgemm"):
        assert ta == tb == [original]
        assert [(a, b, R.gemm_mode_str(g)) for a, b, g in pairs] == [(0, 0, "cublas_" + name)]
        assert O.num_split_from_mode(name) < 0 and ozimmu_amd.get_num_split(mode) == 0
        return
    if name == "fp64_int8_auto":
        assert (ta, tb, pairs) == ([], [], [])          # resolved to a fixed mode before any split (src/gemm.cu:620-640)
        assert O.num_split_from_mode(name) == 0 and ozimmu_amd.get_num_split(mode) == 0
        return
    S = int(name.rsplit("_", 1)[1])
    assert ta == tb == [original] + [int8] * S
    assert O.num_split_from_mode(name) == S
    assert ozimmu_amd.get_num_split(mode) == S
    assert all(R.gemm_mode_str(g) == "int8tc" for _, _, g in pairs)
    got = [(a, b) for a, b, _ in pairs]
    assert got == O.pair_list(S)                         # same pairs in the same ORDER: the order of the FP64 accumulation
    assert len(got) == S * (S + 1) // 2 == len(set(got))
    sums = [a + b for a, b in got]
    assert sums == sorted(sums) and sums[0] == 2 and sums[-1] == S + 1
    # the product adds the pairs of one diagonal as integers: its diagonals are exactly the reference's runs of equal A_id + B_id
    for t in range(2, S + 2):
        assert [p for p in got if sum(p) == t] == [(j, t - j) for j in range(1, t)]


def test_plane_geometry_matches_reference():
    for n in list(range(0, 70)) + [1023, 1024, 1025, 8191, 8192, 32767, 32768, (1 << 20) + 3, (1 << 31) - 5]:
        assert R.padded_ld_i8(n) == O.pad4(n), n
    for m, n in [(1, 1), (5, 7), (1023, 1025), (1025, 1023), (8192, 8192), (32768, 1024), (1024, 32768)]:
        # A planes are taken with op_t over (m = rows of op(A), n = k), B planes with op_n over (k, n):
        # both k-contiguous with ld = pad4(k)  (src/config.cu:137-141, src/gemm.cu:285-304)
        assert R.slice_ld_i8(m, n, R.OP_T) == O.pad4(n)
        assert R.slice_ld_i8(m, n, R.OP_N) == O.pad4(m)
        assert R.slice_num_elements_i8(m, n, R.OP_T) == O.pad4(n) * m
        assert R.slice_num_elements_i8(m, n, R.OP_N) == O.pad4(m) * n
    # the reference computes a plane's element count in uint32 (src/utils.hpp:66-72): it wraps once rows * pad4(k) >= 2^32
    # (SURVEY §8 A9 quirk).  Oracle and product use size_t; shown here so that the deviation is on record against compiled code.
    assert R.slice_num_elements_i8(70000, 70000, R.OP_N) == (70000 * 70000) % (1 << 32)


def test_product_workspace_holds_the_reference_planes():
    """ozimmu_hip_working_memory_size >= the slice planes the reference's geometry asks for (its own layout pads more)"""
    import ozimmu_amd
    lib = ozimmu_amd.lib()
    for S in (3, 9, 12, 18):
        mode = MODES.index(f"fp64_int8_{S}")
        for m, n, k in [(512, 512, 512), (1023, 1025, 1024), (4096, 2048, 1000)]:
            ref_planes = S * (R.slice_num_elements_i8(m, k, R.OP_T) + R.slice_num_elements_i8(k, n, R.OP_N))
            assert lib.ozimmu_hip_working_memory_size(0, 0, m, n, k, ozimmu_amd.real, mode) >= ref_planes
