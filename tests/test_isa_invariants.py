"""CPU test (hipcc cross-compiles gfx950 without a GPU): properties of the generated ISA that the kernels rely on but
that the compiler does not know about.

* the lead throttle of slice_gemm_kernel issues `s_load_dword ... glc` from inline asm and consumes the SGPR only after
  the loop's own `s_waitcnt lgkmcnt(0)`: no instruction may touch that SGPR in between (a compiler-inserted copy would
  capture a stale value);
* no slice GEMM or split kernel uses scratch (a spill inside the k loop would be a silent 2x slowdown)."""
import concurrent.futures
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ozimmu_amd", "csrc")


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    """{"slice_gemm.hip": assembly of all slice-GEMM translation units, "split.hip": ...}"""
    sys.path.insert(0, ROOT)
    from ozimmu_amd import build as B
    d = tmp_path_factory.mktemp("isa")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]

    def device_asm(src):
        # the library build (python -m ozimmu_amd.build, -save-temps=obj) leaves the device assembly behind: use it if
        # it is newer than every source, else compile
        kept = B.device_asm_path(src)
        deps = sorted(B._closure(os.path.join(CSRC, src)))  # the source and the project headers it includes, like the build
        if os.path.exists(kept) and all(os.path.getmtime(kept) >= os.path.getmtime(f) for f in deps):
            return open(kept).read()
        o = d / (src + ".s")
        subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + CSRC,
                               "-D__HIP_PLATFORM_AMD__", "-x", "hip", "--cuda-device-only", "-S",
                               os.path.join(CSRC, src), "-o", str(o)])
        return o.read_text()

    with concurrent.futures.ThreadPoolExecutor(max_workers=len(B.GEMM_PARTS) + 1) as pool:
        parts = [pool.submit(device_asm, src) for src in B.GEMM_PARTS]
        split = pool.submit(device_asm, "split.hip")
        return {"slice_gemm.hip": "\n".join(f.result() for f in parts), "split.hip": split.result()}


def test_throttle_scalar_load_is_not_touched_before_its_wait(asm):
    lines = asm["slice_gemm.hip"].split("\n")
    loads = 0
    for i, l in enumerate(lines):
        m = re.search(r"s_load_dword (s\d+), s\[\d+:\d+\], 0x0 glc", l)
        if not m:
            continue
        loads += 1
        reg = m.group(1)
        for j in range(i + 1, min(i + 800, len(lines))):
            lj = lines[j]
            if "s_waitcnt" in lj and "lgkmcnt(0)" in lj:
                break
            assert not (re.search(r"\b" + reg + r"\b", lj) and not lj.strip().startswith(";")), \
                f"{reg} used before the wait: {lj.strip()}"
        else:
            pytest.fail("no s_waitcnt lgkmcnt(0) after the throttle load")
    assert loads >= 8          # every prefetch-2 instantiation carries the probe


def _kernels(text, substr):
    """{name: body} of the kernels whose mangled name contains `substr`"""
    out = {}
    for m in re.finditer(r"^(_ZN5ozhip\w+):[^\n]*\n(.*?)\n\.Lfunc_end", text, flags=re.S | re.M):
        if substr in m.group(1):
            out[m.group(1)] = m.group(2)
    return out


def _inner_loops(lines):
    """[text] of the innermost loops of a kernel body (list of lines)"""
    blocks, cur = [], None          # (label, comment, [lines])
    for i, l in enumerate(lines):
        m = re.match(r"(\.LBB\d+_\d+):(.*)", l)
        if m:
            comment = m.group(2) + " " + " ".join(x for x in lines[i + 1:i + 3] if x.strip().startswith(";"))
            cur = [m.group(1), comment, []]
            blocks.append(cur)
        elif cur is not None:
            cur[2].append(l)
    loops = []
    for label, comment, body in blocks:
        if "Inner Loop Header" not in comment:
            continue
        tag = "Header=" + label[2:]     # ".LBB10_51" -> "Header=BB10_51"
        text = ["\n".join(body)]
        text += ["\n".join(b) for lab, c, b in blocks if re.search(re.escape(tag) + r"\b", c)]
        # blocks without an MFMA are the rare side paths of a step (thread 0 publishing the phase hint every 4th step): they
        # may reload an address from scratch; the blocks that carry the MFMAs may not touch scratch or copy accumulators
        loops.append("\n".join(t for t in text if "v_mfma" in t))
    return loops


def test_wide_kernel_register_placement_and_m0(asm):
    """slice_gemm_w_kernel.h relies on three things the compiler does not promise:
    * the accumulators stay where the inline-asm MFMAs pinned them: no v_accvgpr_* copies inside the k loop (the
      builtin form produced hundreds, plus scratch);
    * M0 is written only by the LDS-DMA statements and read by nothing else (it is not saved / restored);
    * the LDS-DMA copies are invisible to the compiler's counters: fragment waits are COUNTED lgkmcnt(n > 0), not
      lgkmcnt(0) drains (what the LDS-DMA builtin causes)."""
    ks = {**_kernels(asm["slice_gemm.hip"], "slice_gemm_w_kernel"), **_kernels(asm["slice_gemm.hip"], "slice_gemm_w_multi_kernel")}
    assert len(ks) >= 12
    for name, body in ks.items():
        lines = body.split("\n")
        # the software-pipelined k loops (one per tile height and step kind): every innermost loop that carries MFMAs = its
        # header block plus every block the compiler marks "in Loop: Header=<that label>" (block placement is free: the
        # two-step loop of the VARW_BREG kernels has its second half laid out in FRONT of the header)
        loops = _inner_loops(lines)
        # (32x32x32 tile function, or the paired 16x16x64 one of slice_gemm_x_tile.h)
        loops = [t for t in loops if t.count("v_mfma_i32_32x32x32_i8") + t.count("v_mfma_i32_16x16x64_i8") >= 20]
        assert loops, name
        text = "\n".join(loops)
        assert "v_accvgpr" not in text, f"{name}: accumulator copies inside the k loop"
        # (one known exception, unchanged since round 3 and only reachable where the k64 tile is not: the 32x32x32 kernels with 7
        # diagonals on 128-row tiles - S = 7, and the first pass of S = 13, 14 - keep 448 accumulator registers and reload
        # the phase-hint pointer once per k-step)
        if re.search(r"slice_gemm_w_(multi_)?kernelILi\d+ELi0ELi7ELi4ELi0E", name):
            assert text.count("scratch_") <= len(loops) and "scratch_store" not in text, name
        else:
            assert "scratch_" not in text, name
        in_asm = False
        for l in lines:
            if "#ASMSTART" in l:
                in_asm = True
            elif "#ASMEND" in l:
                in_asm = False
            elif not in_asm and re.search(r"\bm0\b", l) and not l.strip().startswith(";"):
                pytest.fail(f"{name}: compiler-generated use of m0: {l.strip()}")
        counted = len(re.findall(r"s_waitcnt lgkmcnt\([1-9]\d*\)", text))
        drains = len(re.findall(r"s_waitcnt lgkmcnt\(0\)", text))
        assert counted >= 10 and drains <= 3 * len(loops), (name, counted, drains)


def test_kernels_do_not_spill(asm):
    """no scratch at all in the split kernels and the classic slice GEMM.  The persistent wide kernel keeps ~30 kernel
    arguments alive across its tile loop next to a k loop that wants ~60 SGPRs: the compiler parks a few of them in
    VGPR lanes / scratch BETWEEN tiles (<= 128 bytes per lane); its k loops must stay free of scratch and of accumulator
    copies, which test_wide_kernel_register_placement_and_m0 checks."""
    for src, text in asm.items():
        metas = re.findall(r"\.name:\s+(_ZN5ozhip\w+).*?\.private_segment_fixed_size:\s+(\d+)", text, flags=re.S)
        assert metas, src
        for name, scratch in metas:
            limit = 160 if ("slice_gemm_w_kernel" in name or "slice_gemm_w_multi_kernel" in name) else 0
            assert int(scratch) <= limit, f"{src}: {name} uses {scratch} bytes of scratch per lane"


def test_product_and_test_flavour_kernels_differ_only_in_the_dump_branch():
    """ozimmu_amd/build.py builds the same sources twice: libozimmu_hip.so (what ships, no hook) and libozimmu_hip_test.so
    (-DOZIMMU_HIP_TEST_HOOKS).  In the kernels the only hook is the INT32 diagonal-sum dump at the top of the epilogues:
    same kernels, same MFMA / LDS-DMA / fragment-read / barrier counts in both flavours, and the 4-byte stores of the dump
    exist in the test flavour only (the product keeps the few phase-hint stores).  (The parity tests that need no hook, bench.py and smoke() run the product.)"""
    sys.path.insert(0, ROOT)
    from ozimmu_amd import build as B
    B.build()
    pat = {"mfma": r"\bv_mfma_i32_\w+", "lds_dma": r"\bglobal_load_lds_dwordx4\b", "frag": r"\bds_read_b128\b",
           "barrier": r"\bs_barrier\b"}
    seen = 0
    for part in B.GEMM_PARTS:
        prod = open(B.device_asm_path(part)).read()
        test = open(B.device_asm_path(part).replace(os.sep + "build" + os.sep, os.sep + "build_test" + os.sep)).read()
        kp, kt = _kernels(prod, "slice_gemm"), _kernels(test, "slice_gemm")
        assert kp and set(kp) == set(kt), part
        for name in kp:
            for what, rx in pat.items():
                assert len(re.findall(rx, kp[name])) == len(re.findall(rx, kt[name])), (name, what)
            # 4-byte stores: the product has the phase-hint store of its step bodies at most; the dump adds one per
            # accumulator register and diagonal
            n_prod = len(re.findall(r"\bglobal_store_dword\b", kp[name]))
            n_test = len(re.findall(r"\bglobal_store_dword\b", kt[name]))
            assert n_prod <= 12, f"{name}: {n_prod} 4-byte stores in the product (a dump?)"
            assert n_test >= n_prod + 16, f"{name}: no dump stores in the test flavour"
            seen += 1
    assert seen >= 60


TILE_START = r"v_mfma_i32_16x16x64_i8 a\[32:35\], v\[112:115\], v\[\d+:\d+\], 0\s*$"


def _varw(name):     # slice_gemm_w_[multi_]kernel<S, D0, ND, WA, VARW, ...>: the fifth template argument
    m = re.search(r"kernelILi\d+ELi\d+ELi\d+ELi\d+ELi(\d+)E", name)
    return int(m.group(1)) if m else 0


def _line_vgprs(l):
    regs = {int(m.group(1)) for m in re.finditer(r"\bv(\d+)\b", l)}
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", l):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return regs


def test_named_b_registers_are_untouched_by_the_compiler(asm):
    """VARW_BREG kernels (slice_gemm_y_tile.h) keep the k64 tile's B fragments in two hand-allocated register sets v[80:151],
    v[152:223] and (round 5) their accumulators in a[0:255] + v[224:255]: inline-asm loads write the sets, inline-asm MFMAs read
    them, every statement lists the live named registers as clobbered.  What the compiler may do with them:
      * AGPRs and v[224:255] (the accumulators): nothing, anywhere;
      * inside the k loops: nothing above v79;
      * the first set v[80:151]: free again in the overlapped last step (it multiplies out of the second set and loads nothing) -
        a compiler-generated instruction may name it only if every MFMA up to the tile's end reads its B operand from the second
        set and no named load follows;
      * the second set v[152:223]: only behind the tile's last MFMA (the plain epilogue).
    (hipcc hands out VGPRs in ascending order: the layout puts what becomes free first right above the compiler's own range.)"""
    ks = {**_kernels(asm["slice_gemm.hip"], "slice_gemm_w_kernel"), **_kernels(asm["slice_gemm.hip"], "slice_gemm_w_multi_kernel")}
    breg = {n: b for n, b in ks.items() if _varw(n) & 16384}   # VARW_BREG
    assert len(breg) >= 2, "no VARW_BREG kernel in the library"
    for name, body in breg.items():
        lines = body.split("\n")
        for loop in [t for t in _inner_loops(lines) if t.count("v_mfma_i32_16x16x64_i8") >= 20]:
            in_asm = False
            for l in loop.split("\n"):
                if "#ASMSTART" in l:
                    in_asm = True
                elif "#ASMEND" in l:
                    in_asm = False
                elif not in_asm and not l.strip().startswith(";"):
                    assert not any(r >= 80 for r in _line_vgprs(l)), f"{name}: compiler code above v79 inside a k loop: {l.strip()}"
        # per line: what lies between it and the end of its tile function (the next zero-fill of a0, or the kernel's end)
        in_asm = False
        kinds = []                      # (index, kind, payload) of the asm instructions: mfma B operand / named load / zero-fill
        for i, l in enumerate(lines):
            if "#ASMSTART" in l:
                in_asm = True
            elif "#ASMEND" in l:
                in_asm = False
            elif in_asm:
                m = re.search(r"v_mfma_i32_16x16x64_i8 [av]\[\d+:\d+\], v\[(\d+):\d+\]", l)
                if m:
                    kinds.append((i, "mfma", int(m.group(1))))
                elif "global_load_dwordx4 v[" in l:
                    kinds.append((i, "load", 0))
                elif re.search(r"v_accvgpr_write_b32 a0, 0", l):
                    kinds.append((i, "zero", 0))
                # round 6: the register kernel does not zero its accumulators any more - the first MFMA of a tile on every tuple
                # takes the constant 0 as C; a tile's very first MFMA (slot 0: block 0, slices i = 0, j = 8, column block 0 = tuple 8
                # out of v[112:115]) marks the start of a tile function (it reads the first B set: the order of the checks matters)
                if re.search(TILE_START, l):
                    kinds[-1] = (i, "zero", 0)
            elif re.match(r"\s*(s_branch|s_endpgm|s_setpc)", l):
                kinds.append((i, "zero", 0))   # not fall-through either: the overlapped path leaves the tile function here (the
                                               # plain k loop that follows in the text is the other side of a branch)
        # (the multi-product kernels keep the zero-fill in front of the tile: slice_gemm_y_tile.h, VARW_ZFILL)
        assert any(re.search(TILE_START, l) or "v_accvgpr_write_b32 a0, 0" in l for l in lines), \
            f"{name}: neither a zeroing first MFMA nor an asm zero-fill (named accumulators expected)"
        in_asm = False
        for i, l in enumerate(lines):
            if "#ASMSTART" in l:
                in_asm = True
                continue
            if "#ASMEND" in l:
                in_asm = False
                continue
            if in_asm or l.strip().startswith(";"):
                continue
            assert not re.search(r"\ba\d+\b|\ba\[\d+:\d+\]", l), f"{name}: compiler code names an AGPR: {l.strip()}"
            regs = _line_vgprs(l)
            assert not any(r >= 224 for r in regs), f"{name}: compiler code in v[224:255]: {l.strip()}"
            hi = any(152 <= r <= 223 for r in regs)
            lo = any(80 <= r <= 151 for r in regs)
            if not (hi or lo):
                continue
            for idx, kind, val in kinds:
                if idx < i:
                    continue
                if kind == "zero":
                    break
                if hi:
                    pytest.fail(f"{name}: compiler code in v[152:223] in front of a named load / MFMA of the same tile: {l.strip()}")
                if kind == "load":
                    pytest.fail(f"{name}: compiler code in v[80:151] in front of a named load: {l.strip()}")
                assert val >= 152, f"{name}: compiler code in v[80:151] in front of an MFMA that reads the first set: {l.strip()}"


def test_returning_atomics_are_waited_for_before_their_destination_is_touched(asm):
    """The claim tickets of the persistent kernels are returning atomics (global_atomic_add ... sc0: the pre-op value lands in
    a VGPR asynchronously).  Compiler-generated ones are tracked by the compiler; the hand-written one of the speculative claim
    (slice_gemm_w_kernel.h: draw_ticket) is not - round 4 issued it at a tile boundary and waited a whole tile later, a window
    in which a compiler-inserted copy or spill of the destination would have captured a stale ticket.  Now the wait
    sits in the same asm statement.  Checked on the ISA for EVERY returning atomic of the slice-GEMM kernels: between the
    atomic and the first wait that covers it (vmcnt(0), alone or inside a combined s_waitcnt) no instruction names its
    destination register, and no branch leaves the window."""
    lines = asm["slice_gemm.hip"].split("\n")
    seen = hand = 0
    for i, l in enumerate(lines):
        m = re.match(r"\s*global_atomic_\w+ (v\d+), .*\bsc0\b", l)
        if not m:
            continue
        seen += 1
        reg = m.group(1)
        n = int(reg[1:])
        in_asm_stmt = False
        for j in range(i - 1, max(i - 6, 0), -1):       # the hand-written one sits inside #ASMSTART ... #ASMEND
            if "#ASMSTART" in lines[j]:
                in_asm_stmt = True
                break
            if "#ASMEND" in lines[j]:
                break
        hand += in_asm_stmt
        for j in range(i + 1, min(i + 400, len(lines))):
            t = lines[j].strip()
            if not t or t.startswith(";") or t.startswith("#") or t.startswith("."):
                continue
            if re.match(r"s_waitcnt\b", t) and re.search(r"vmcnt\(0\)", t):
                break
            assert not re.match(r"s_(c)?branch|s_endpgm|s_setpc", t), f"line {j}: control flow before the wait for {reg}: {t}"
            assert not re.search(rf"\b{reg}\b", t), f"line {j}: {reg} touched before its atomic returned: {t}"
            for a, b in re.findall(r"v\[(\d+):(\d+)\]", t):
                assert not (int(a) <= n <= int(b)), f"line {j}: {reg} (in a tuple) touched before its atomic returned: {t}"
        else:
            pytest.fail(f"line {i}: no vmcnt(0) wait after {l.strip()}")
    assert seen >= 20 and hand >= 4, (seen, hand)   # every persistent kernel claims; the k64 kernels also speculate


def test_named_accumulator_kernels_keep_the_compiler_out_of_their_registers(asm):
    """VARW_ACCN kernels (slice_gemm_w_kernel.h: fp64_int8_11 / 12 on the k64 tile) hold their 352 / 384 accumulator registers in
    a[0:255] and v[160:255] / v[128:255] and their in-place B slices in v[144:159] / v[96:127] as NAMED registers: written and read by inline asm only, every statement
    listing the whole set as clobbered.  The compiler must neither use them for its own values (a temporary between two statements
    would destroy a sum) nor spill: no compiler-generated instruction names an AGPR or a VGPR of the plan's range, no scratch in a k loop, and the kernel
    descriptor gives the wave the whole file (256 + 256)."""
    ks = _kernels(asm["slice_gemm.hip"], "slice_gemm_w_kernel")
    accn = {n: b for n, b in ks.items() if (_varw(n) & 32768) and not (_varw(n) & 16384)}   # VARW_ACCN
    assert len(accn) >= 1, "no named-accumulator kernel in the library"
    for name, body in accn.items():
        first = 96 if "kernelILi12E" in name else 144   # register plan 12 / 11: the compiler's range ends below
        for loop in [t for t in _inner_loops(body.split("\n")) if t.count("v_mfma_i32_16x16x64_i8") >= 20]:
            assert "scratch_" not in loop, name        # (fp64_int8_12: one 4-byte spill at the kernel's entry, none in a k loop)
        in_asm = False
        mfmas = 0
        for l in body.split("\n"):
            if "#ASMSTART" in l:
                in_asm = True
            elif "#ASMEND" in l:
                in_asm = False
            elif in_asm:
                mfmas += "v_mfma" in l
            elif not l.strip().startswith(";"):
                assert "v_mfma" not in l, f"{name}: an MFMA outside inline asm"
                regs = {int(m.group(1)) for m in re.finditer(r"\bv(\d+)\b", l)}
                for m in re.finditer(r"\bv\[(\d+):(\d+)\]", l):
                    regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
                assert not any(r >= first for r in regs), f"{name}: compiler code in the named VGPR range: {l.strip()}"
                assert not re.search(r"\ba\d+\b|\ba\[\d+:\d+\]", l), f"{name}: compiler code names an AGPR: {l.strip()}"
        assert mfmas >= 1000, (name, mfmas)
        meta = re.search(re.escape(name) + r"\.kd.*?\.vgpr_count:\s+(\d+)", asm["slice_gemm.hip"], flags=re.S)
        assert meta and int(meta.group(1)) == 512, name


def test_one_launch_kernel_uses_no_cache_maintenance_and_no_scratch(tmp_path):
    """slice_gemm_one_launch.hip hands slice planes from producer to consumer workgroups INSIDE one kernel without flushing or
    invalidating a cache (its header: "Visibility"): the producers' stores must be write-through (sc0 sc1), the READY words
    must be read and written at agent scope (sc1), and the compiler must not have added buffer_wbl2 / buffer_inv (a fence that
    crept back in costs a full-L2 operation per workgroup: measured 84 instead of 37 us) - nor scratch."""
    sys.path.insert(0, ROOT)
    from ozimmu_amd import build as B
    src = "slice_gemm_one_launch.hip"
    kept = B.device_asm_path(src)
    deps = sorted(B._closure(os.path.join(CSRC, src)))
    if os.path.exists(kept) and all(os.path.getmtime(kept) >= os.path.getmtime(f) for f in deps):
        text = open(kept).read()
    else:
        o = tmp_path / "one_launch.s"
        subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + CSRC, "-D__HIP_PLATFORM_AMD__",
                               "-x", "hip", "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", str(o)])
        text = o.read_text()
    kernels = re.findall(r"^(_ZN5ozhip20split_gemm_k2_kernelILi(\d+)E\w*):[^\n]*\n(.*?)\n\.Lfunc_end", text, re.S | re.M)
    assert sorted(int(k[1]) for k in kernels) == list(range(3, 10))
    for name, S, body in kernels:
        assert "buffer_wbl2" not in body and "buffer_inv" not in body, name
        assert "scratch_" not in body, name
        stores = re.findall(r"global_store_dwordx4 .*", body)
        # every 16-byte store of the split phase is write-through; the epilogue stores C with dwordx2 / dwordx4 without sc bits
        wt = [l for l in stores if "sc0 sc1" in l]
        assert len(wt) >= 2, (name, len(wt))                       # one per operand layout
        assert re.search(r"global_store_dwordx2 .* sc0 sc1", body), name   # the row scales
        assert re.search(r"global_load_dword v\d+, .* sc1", body), name    # the READY poll
        assert re.search(r"global_store_dword v\d+, .* sc1", body), name   # the READY store
        assert "v_mfma_i32_32x32x32_i8" in body, name
    meta = re.findall(r"\.name:\s+_ZN5ozhip20split_gemm_k2_kernel\w+\n(?:.*\n)*?\s+\.private_segment_fixed_size: (\d+)", text)
    assert meta and all(int(x) == 0 for x in meta)


def test_overlapped_last_step_is_free_of_scratch_and_compiler_waits(asm):
    """Round 6 (profiles/r6_ablate/): the register kernel's overlapped last step - the recombination stream of block a - 1 fused into
    the MFMA statements of block a - took the SUM of its MFMA time and the chain's time, because the compiler had hoisted the
    lane's offset into C and a VGPR copy of the C pointer to the kernel's entry, spilled both and reloaded them from scratch,
    behind an `s_waitcnt vmcnt(0)`, in front of every store.  The addresses are now a SALU-built uniform base + a 32-bit lane offset
    computed inside the step, the stores asm statements.  Checked: between the first and the last fused statement (an MFMA and a
    v_fmac_f64 / v_ldexp_f64 in ONE asm statement) of every overlapped step there is no scratch access, every store is an asm
    store with its wait state, and the stream's three instructions ride in the MFMA's statement."""
    ks = _kernels(asm["slice_gemm.hip"], "slice_gemm_w_kernel")
    breg = {n: b for n, b in ks.items() if (_varw(n) & 16384) and (_varw(n) & 32768)}
    assert breg
    steps = 0
    for name, body in breg.items():
        lines = body.split("\n")
        stmts, cur = [], None            # (first line, last line, text) of every asm statement
        for i, l in enumerate(lines):
            if "#ASMSTART" in l:
                cur = [i, i, []]
            elif "#ASMEND" in l and cur is not None:
                cur[1] = i
                stmts.append((cur[0], cur[1], "\n".join(cur[2])))
                cur = None
            elif cur is not None:
                cur[2].append(l)
        fused = [(a, b) for a, b, t in stmts if "v_mfma_i32_16x16x64_i8" in t and ("v_fmac_f64" in t or "v_ldexp_f64" in t)]
        if not fused:
            continue                     # (a first pass of a two-pass mode: 9 diagonals, never final - the step is compiled all the same)
        runs, start, prev = [], fused[0][0], fused[0][1]
        for a, b in fused[1:]:
            if a - prev > 400:
                runs.append((start, prev))
                start = a
            prev = b
        runs.append((start, prev))
        for a, b in runs:
            span = lines[a:b + 1]
            n_fused = sum(1 for x, y in fused if a <= x <= b)
            if n_fused < 150:
                continue
            steps += 1
            text = "\n".join(span)
            assert "scratch_" not in text, f"{name}: scratch access inside an overlapped step"
            assert n_fused >= 3 * 8 * 9 - 8, (name, n_fused)      # three blocks x 72 elements accumulate inside MFMA statements
            in_asm = False
            drains = 0
            for l in span:
                if "#ASMSTART" in l:
                    in_asm = True
                elif "#ASMEND" in l:
                    in_asm = False
                elif not in_asm and not l.strip().startswith(";"):
                    assert "global_store" not in l, f"{name}: compiler-generated store inside an overlapped step: {l.strip()}"
                    drains += bool(re.search(r"s_waitcnt vmcnt\(0\)", l))
            # (one wait for the row / column scales loaded at the step's start, in front of their first use a block later; the
            # beta != 0 arm waits for the old C it loaded a block ahead - never one per store)
            assert drains <= 6, f"{name}: {drains} full vmcnt drains inside an overlapped step"
    assert steps >= 2        # the 64 x 128 tile and the half-height tile of fp64_int8_9
