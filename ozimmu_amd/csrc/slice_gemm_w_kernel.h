// slice_gemm_w_kernel.h — "wide" form of the fused INT8 slice GEMM: ONE 4-wave workgroup per CU, one wave per SIMD,
// each wave owning WA vertically stacked 32x32 output blocks whose S INT32 diagonal accumulators fill the unified
// 512-entry VGPR+AGPR file (WA*ND*16 registers: 432 at S=9, WA=3).
//
// Why: the part is power-limited under full-entropy INT8 MFMA, and everything that is not an MFMA
// (HBM/L2 -> LDS copies, LDS fragment reads) costs both stall time and clock.  The register file bounds the outputs a
// CU can keep resident; with two waves per SIMD (slice_gemm_kernel.h) each wave can hold one 32x32 block at S=9
// (8 192 outputs per CU), with one wave per SIMD three (12 288 outputs per CU).  Tile = (32*WA) x 128:
//
//                               staged KiB per wave-MFMA      LDS fragment reads per MFMA
//   2 x (64x64, 4 waves)             0.200 * S                       0.40
//   1 x (96x128, 4 waves, WA=3)      0.117 * S   (-42 %)             0.27   (-33 %)
//
//   * the WA A row-blocks are shared by the 4 waves (staged cooperatively, one barrier per k-step); every wave has
//     its own B row-block, staged by that wave alone: wave-private LDS, ordered by the wave's own vmcnt only;
//   * NA A-buffers (prefetch distance NA-1) and 2 B-buffers.  The B fragments of a k-step are copied to registers at
//     the top of the step, so the B buffer can be refilled (distance 2) as soon as those reads have returned;
//   * A fragments stream through a small register ring, the LDS-DMA copies of the stage being prefetched are spread
//     between the MFMA groups so that no copy waits behind another wave's burst in the CU's texture-address queue.
//
// Same arithmetic as slice_gemm_kernel (same diagonal accumulators, same epilogue): results are bit-identical.
#pragma once
#include <type_traits>

#include "slice_gemm_kernel.h"

namespace ozhip {

template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, F, I + 1>(static_cast<F &&>(f));
  }
}

// c += b (32 rows x 32 k, MFMA "A" operand) * a (MFMA "B" operand), accumulator pinned to the AGPR / VGPR half
__device__ __forceinline__ void mfma_agpr(v16i &c, const v4i &b, const v4i &a) {
  asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(c) : "v"(b), "v"(a));
}
__device__ __forceinline__ void mfma_vgpr(v16i &c, const v4i &b, const v4i &a) {
  asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(c) : "v"(b), "v"(a));
}

// One 1 KiB fragment block HBM/L2 -> LDS: lane l copies 16 bytes from sbase + voff(l) + IMM to LDS lds_a + lds_b + IMM
// + 16*l.  sbase, lds_a, lds_b are wave-uniform (SGPRs): the loop-invariant part of the source lives in sbase and the
// k position in the one VGPR voff (lane*16 + k_block * S KiB), so a copy costs one SALU (M0 = LDS base), the mandatory
// wait state after an M0 write, and the copy itself -- instruction slots between MFMAs are what a lone wave per SIMD
// is short of.  M0 is compiler-reserved; nothing else in these kernels reads it (tests/test_isa_invariants.py checks
// the ISA), so it is not saved / restored (-1.8 % kernel time).
// a wave-uniform pointer, stated as such (inside the persistent tile loop hipcc's divergence analysis no longer proves
// it and would hand the copies a VGPR pair; two v_readfirstlane per tile, outside the k loop)
__device__ __forceinline__ const int8_t *uniform_ptr(const int8_t *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const int8_t *)(((unsigned long long)hi << 32) | lo);
}

template <int IMM>
__device__ __forceinline__ void glds16(const int8_t *sbase, uint32_t voff, uint32_t lds_a, uint32_t lds_b) {
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%c4"
               :
               : "v"(voff), "s"(sbase), "s"(lds_a), "s"(lds_b), "i"(IMM)
               : "memory");
}

// VARW_BREG (slice_gemm_y_tile.h): the B fragments of the k64 tile live in HAND-ALLOCATED registers v[80:223] (two sets of
// 2 * S fragments of 4 registers, S <= 9).  hipcc's allocator, handed 36 live 128-bit tuples next to 72 accumulator tuples,
// splits live ranges between the two parities of the step and spills (1.4 KiB of scratch, hundreds of v_accvgpr copies inside
// the k loop); named registers cost nothing.  Every asm statement of the k loop that reads or writes them lists the whole
// range as clobbered, so the compiler keeps no value there across any of them; tests/test_isa_invariants.py checks that no
// compiler-generated instruction of a BREG kernel touches the range at all.
#define OZ_BREG_FIRST 80
#define OZ_BREG_CLOBBERS \
  "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", \
  "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", \
  "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", \
  "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", \
  "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", \
  "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", \
  "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", \
  "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", \
  "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", \
  "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", \
  "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", \
  "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", \
  "v251", "v252", "v253", "v254", "v255"
// fragment -> VGPR: lane l loads the 16 bytes at sbase + voff(l) + IMM into v[REG : REG + 3].  Inline asm for the same reason
// as glds16: the compiler does not count it, the consumer waits with an explicit vmcnt.
template <int REG, int IMM>
__device__ __forceinline__ void gload16_named(const int8_t *sbase, uint32_t voff) {
  static_assert(REG >= OZ_BREG_FIRST && REG + 3 <= 255 && (REG & 3) == 0, "inside the reserved range");
  asm volatile("global_load_dwordx4 v[%c0:%c1], %2, %3 offset:%c4"
               :
               : "i"(REG), "i"(REG + 3), "v"(voff), "s"(sbase), "i"(IMM)
               : "memory", OZ_BREG_CLOBBERS);
}

// VARW_ACCN (slice_gemm_y_tile.h): the accumulators of the k64 tile as hand-allocated registers.  The compiler never sees an
// accumulator: nothing for its allocator to re-assign between code regions, to copy between the halves of the file or to spill
// (what it did to every attempt to hold 352 accumulator registers - or 288 next to the values of an overlapped epilogue - as C++
// variables).  Two register plans:
//   plan 11 (fp64_int8_11, B through LDS + VARW_BHI): tuple X < 64 in a[4X : 4X + 3], X >= 64 in v[160 + 4 (X - 64) ...] (24 tuples),
//            the in-place B fragments of the slices 9, 10 in v[144:159];
//   plan 9  (VARW_BREG, 9 diagonals): X < 64 in a[...], the 8 tuples X >= 64 in v[224:255], ABOVE the two B sets v[80:151], v[152:223]:
//            hipcc hands out VGPRs in ascending order, so what it may use again comes first - the first B set in the overlapped last
//            step, both sets behind the k loop - and it reaches a live named register only when it needs more than that many
//            (the first layout had the accumulators at v[80:111], right above its own range: the overlapped step's temporaries
//            landed in them).
// Every asm statement of such a tile lists a CLOBBER SET (OZ_CL_*) that contains all named registers alive at that point, so the
// compiler keeps no value of its own in them across any statement; tests/test_isa_invariants.py checks that no compiler-generated
// instruction of such a kernel names an AGPR or a VGPR of the plan's range.
#define OZ_AGPR_ALL \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", \
  "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", \
  "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", \
  "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", \
  "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", \
  "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", \
  "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", \
  "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", \
  "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", \
  "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", \
  "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", \
  "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", \
  "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", \
  "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", \
  "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", \
  "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", \
  "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", \
  "a252", "a253", "a254", "a255"
#define OZ_V144_255 \
  "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", \
  "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", \
  "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", \
  "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", \
  "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", \
  "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", \
  "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", \
  "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
#define OZ_V96_143 \
  "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", \
  "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", \
  "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", \
  "v139", "v140", "v141", "v142", "v143"
#define OZ_V224_255 \
  "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", \
  "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", \
  "v252", "v253", "v254", "v255"
#define OZ_V80_151 \
  "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", \
  "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", \
  "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", \
  "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", \
  "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151"
#define OZ_V152_223 \
  "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", \
  "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", \
  "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", \
  "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", \
  "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", \
  "v222", "v223"
constexpr int OZ_CL_P11 = 0;      // plan 11: a[0:255], v[144:255]
constexpr int OZ_CL_P9 = 1;       // plan 9:  a[0:255], v[80:255]
constexpr int OZ_CL_P9_SET1 = 2;  // plan 9, the overlapped last step: the first B set (v[80:151]) is the compiler's again
constexpr int OZ_CL_P9_ACC = 3;   // plan 9 behind the k loop: the accumulators only (a[0:255], v[224:255])
constexpr int OZ_CL_P12 = 5;      // plan 12 (fp64_int8_12): a[0:255], v[96:255] - 32 tuples in v[128:255], the in-place B slices 8 .. 11 in v[96:127]
constexpr int OZ_CL_P11_ACC = 4;  // plan 11 behind the k loop: a[0:255], v[160:255] (= plan 11: v[144:159] are left alone as well)
__host__ __device__ constexpr int accn_v_first(int cl) { return cl == OZ_CL_P12 ? 128 : (cl == OZ_CL_P11 || cl == OZ_CL_P11_ACC) ? 160 : 224; }
__host__ __device__ constexpr int accn_bhi_first(int cl) { return cl == OZ_CL_P12 ? 96 : 144; }
#define OZ_BHI_FIRST 144
// one asm statement with the clobber set CL (the operand lists are the macro's variadic part: outputs : inputs)
#define OZ_ACCN_ASM_(CL, MEM, TEXT, ...)                                                        \
  do {                                                                                          \
    if constexpr ((CL) == OZ_CL_P11 || (CL) == OZ_CL_P11_ACC)                                   \
      asm volatile(TEXT : __VA_ARGS__ : MEM OZ_AGPR_ALL, OZ_V144_255);                          \
    else if constexpr ((CL) == OZ_CL_P12)                                                       \
      asm volatile(TEXT : __VA_ARGS__ : MEM OZ_AGPR_ALL, OZ_V96_143, OZ_V144_255);              \
    else if constexpr ((CL) == OZ_CL_P9)                                                        \
      asm volatile(TEXT : __VA_ARGS__ : MEM OZ_AGPR_ALL, OZ_V80_151, OZ_V152_223, OZ_V224_255); \
    else if constexpr ((CL) == OZ_CL_P9_SET1)                                                   \
      asm volatile(TEXT : __VA_ARGS__ : MEM OZ_AGPR_ALL, OZ_V152_223, OZ_V224_255);              \
    else                                                                                        \
      asm volatile(TEXT : __VA_ARGS__ : MEM OZ_AGPR_ALL, OZ_V224_255);                           \
  } while (0)
#define OZ_ACCN_ASM(CL, TEXT, ...) OZ_ACCN_ASM_(CL, , TEXT, __VA_ARGS__)
#define OZ_ACCN_ASM_MEM(CL, TEXT, ...) OZ_ACCN_ASM_(CL, "memory" OZ_COMMA, TEXT, __VA_ARGS__)
#define OZ_COMMA ,
template <int CL, int REG, int IMM>
__device__ __forceinline__ void gload16_named_accn(const int8_t *sbase, uint32_t voff) {
  static_assert((REG & 3) == 0 && REG >= 80 && REG + 3 <= 223, "inside a B range");
  OZ_ACCN_ASM_MEM(CL, "global_load_dwordx4 v[%c0:%c1], %2, %3 offset:%c4", : "i"(REG), "i"(REG + 3), "v"(voff), "s"(sbase), "i"(IMM));
}
// accumulator tuple X += b x a.  BREG_ < 0: the B fragment is the compiler's value `b`; else v[BREG_ : BREG_ + 3]
template <int CL, int X, int BREG_, bool ZC = false> // ZC: the tuple is WRITTEN (C = 0): the first MFMA of a tile on it
__device__ __forceinline__ void mfma16_accn(const v4i &b, const v4i &a) {
  if constexpr (ZC) {
    static_assert(BREG_ >= 0, "the zeroing form exists for the register kernel");
    if constexpr (X < 64)
      OZ_ACCN_ASM(CL, "v_mfma_i32_16x16x64_i8 a[%c0:%c1], v[%c2:%c3], %4, 0", : "i"(4 * X), "i"(4 * X + 3), "i"(BREG_), "i"(BREG_ + 3), "v"(a));
    else
      OZ_ACCN_ASM(CL, "v_mfma_i32_16x16x64_i8 v[%c0:%c1], v[%c2:%c3], %4, 0",
                  : "i"(accn_v_first(CL) + 4 * (X - 64)), "i"(accn_v_first(CL) + 4 * (X - 64) + 3), "i"(BREG_), "i"(BREG_ + 3), "v"(a));
  } else if constexpr (X < 64) {
    if constexpr (BREG_ < 0)
      OZ_ACCN_ASM(CL, "v_mfma_i32_16x16x64_i8 a[%c0:%c1], %2, %3, a[%c0:%c1]", : "i"(4 * X), "i"(4 * X + 3), "v"(b), "v"(a));
    else
      OZ_ACCN_ASM(CL, "v_mfma_i32_16x16x64_i8 a[%c0:%c1], v[%c2:%c3], %4, a[%c0:%c1]",
                  : "i"(4 * X), "i"(4 * X + 3), "i"(BREG_), "i"(BREG_ + 3), "v"(a));
  } else {
    constexpr int V = accn_v_first(CL) + 4 * (X - 64);
    static_assert(V + 3 <= 255, "accumulator tuples of the VGPR half");
    if constexpr (BREG_ < 0)
      OZ_ACCN_ASM(CL, "v_mfma_i32_16x16x64_i8 v[%c0:%c1], %2, %3, v[%c0:%c1]", : "i"(V), "i"(V + 3), "v"(b), "v"(a));
    else
      OZ_ACCN_ASM(CL, "v_mfma_i32_16x16x64_i8 v[%c0:%c1], v[%c2:%c3], %4, v[%c0:%c1]",
                  : "i"(V), "i"(V + 3), "i"(BREG_), "i"(BREG_ + 3), "v"(a));
  }
}
template <int CL, int X>
__device__ __forceinline__ void zero_accn() {
  if constexpr (X < 64) {
    OZ_ACCN_ASM(CL, "v_accvgpr_write_b32 a%c0, 0\n\tv_accvgpr_write_b32 a%c1, 0\n\tv_accvgpr_write_b32 a%c2, 0\n\tv_accvgpr_write_b32 a%c3, 0",
                : "i"(4 * X), "i"(4 * X + 1), "i"(4 * X + 2), "i"(4 * X + 3));
  } else {
    constexpr int V = accn_v_first(CL) + 4 * (X - 64);
    OZ_ACCN_ASM(CL, "v_mov_b32 v%c0, 0\n\tv_mov_b32 v%c1, 0\n\tv_mov_b32 v%c2, 0\n\tv_mov_b32 v%c3, 0", : "i"(V), "i"(V + 1), "i"(V + 2), "i"(V + 3));
  }
}
// register v of tuple x -> a compiler value (the epilogue; x, v fold to constants once its loops are unrolled)
template <int CL>
__device__ __forceinline__ int read_accn(int x, int v) {
  int r;
  if (x < 64)
    OZ_ACCN_ASM(CL, "v_accvgpr_read_b32 %0, a%c1", "=v"(r) : "i"(4 * x + v));
  else
    OZ_ACCN_ASM(CL, "v_mov_b32 %0, v%c1", "=v"(r) : "i"(accn_v_first(CL) + 4 * (x - 64) + v));
  return r;
}

// ---- the recombination stream of the overlapped last step (slice_gemm_y_tile.h, round 6) as asm statements ---------------------
// hipcc orders plain C++ conversions and fmas by register pressure, not by the source: written as `x = fma((double)r, sc, x)` the
// stream's conversions and accumulations of six of a block's eight outputs sank to the block's end, one dependent chain (and the
// scale constants went through v_readlane spills).  As asm statements they stay where the schedule puts them; the accumulator
// read rides in the MFMA's own statement (no compiler-inserted wait state between the two).
// One MFMA slot of the overlapped step as ONE statement (hipcc puts an `s_nop` behind every asm statement it cannot look into: as
// separate statements a slot was MFMA, read, nop, fma, cvt, nop - six issue slots against the MFMA's 16 cycles; the wave was
// issue bound at ~37 cycles per slot, profiles/r6_ablate/):
//   [accumulator tuple X += v[BREG_ : BREG_ + 3] x a]              (X < 0: no MFMA - the serial part of the stream)
//   [rd = register RS of the accumulator file]                     (RS < 0: none; RS < 256: AGPR a[RS]; else VGPR v[RS - 256])
//   [x = ldexp(fs, E) | x += 2^E * fs]                             (ACC 1: the chain's first fma(fs, 2^E, 0), exactly; 2: one rounding)
//   [cv = double(cs)]
// 2^E travels as the high word of a double literal (the low word of a power of two is zero: a VOP2 literal).  rd, cv and x are
// read-write operands in every variant: a variant that does not write one leaves it alone.
#define OZ_T_MFMA_A "v_mfma_i32_16x16x64_i8 a[%c3:%c4], v[%c5:%c6], %7, a[%c3:%c4]\n\t"
#define OZ_T_MFMA_V "v_mfma_i32_16x16x64_i8 v[%c3:%c4], v[%c5:%c6], %7, v[%c3:%c4]\n\t"
#define OZ_T_MFMA_N ""
#define OZ_T_RD_A "v_accvgpr_read_b32 %0, a%c8\n\t"
#define OZ_T_RD_V "v_mov_b32 %0, v%c8\n\t"
#define OZ_T_RD_N ""
#define OZ_T_F_1 "v_ldexp_f64 %2, %10, %c11\n\t"
#define OZ_T_F_2 "v_fmac_f64_e32 %2, %c11, %10\n\t"
#define OZ_T_F_0 ""
#define OZ_T_C_1 "v_cvt_f64_i32_e32 %1, %9\n\t"
#define OZ_T_C_0 ""
#define OZ_EPI_CASE(M, MC, R, RC, F, C)                                                                                          \
  else if constexpr (mk == MC && rk == RC && ACC == F && CVT == (C != 0)) OZ_ACCN_ASM(                                            \
      CL, OZ_T_MFMA_##M OZ_T_RD_##R OZ_T_F_##F OZ_T_C_##C, "+v"(rd), "+v"(cv), "+v"(x)                                             \
      : "i"(X0), "i"(X0 + 3), "i"(BREG_ < 0 ? 0 : BREG_), "i"(BREG_ < 0 ? 3 : BREG_ + 3), "v"(a), "i"(R0), "v"(cs), "v"(fs), "i"(ACC == 1 ? E : HI))
#define OZ_EPI_CASES_FC(M, MC, R, RC)                                                                                             \
  OZ_EPI_CASE(M, MC, R, RC, 0, 0);                                                                                                \
  OZ_EPI_CASE(M, MC, R, RC, 0, 1);                                                                                                \
  OZ_EPI_CASE(M, MC, R, RC, 1, 0);                                                                                                \
  OZ_EPI_CASE(M, MC, R, RC, 1, 1);                                                                                                \
  OZ_EPI_CASE(M, MC, R, RC, 2, 0);                                                                                                \
  OZ_EPI_CASE(M, MC, R, RC, 2, 1)
#define OZ_EPI_CASES_R(M, MC)                                                                                                     \
  OZ_EPI_CASES_FC(M, MC, N, 0);                                                                                                   \
  OZ_EPI_CASES_FC(M, MC, A, 1);                                                                                                   \
  OZ_EPI_CASES_FC(M, MC, V, 2)
template <int CL, int X, int BREG_, int RS, bool CVT, int ACC, int E>
__device__ __forceinline__ void epi_slot_asm(const v4i &a, int &rd, double &cv, const int &cs, double &x, const double &fs) {
  constexpr int mk = X < 0 ? 0 : X < 64 ? 1 : 2, rk = RS < 0 ? 0 : RS < 256 ? 1 : 2;
  constexpr int X0 = X < 0 ? 0 : X < 64 ? 4 * X : accn_v_first(CL) + 4 * (X - 64), R0 = RS < 0 ? 0 : RS < 256 ? RS : RS - 256;
  constexpr int HI = (1023 + E) << 20;
  static_assert(E > -1000 && E < 1000 && (ACC != 1 || (E >= -16 && E <= 64)), "exponent as an inline constant / a normal double");
  static_assert(X < 0 || BREG_ >= 0, "the MFMA of a slot takes its B fragment from the named registers");
  if constexpr (mk == 0 && rk == 0 && ACC == 0 && !CVT) {
  }
  OZ_EPI_CASES_R(N, 0);
  OZ_EPI_CASES_R(A, 1);
  OZ_EPI_CASES_R(V, 2);
}
#undef OZ_EPI_CASES_R
#undef OZ_EPI_CASES_FC
#undef OZ_EPI_CASE

// MFMA slots of one k-step of a wave that owns WA blocks: block a outermost, A slice i ascending, B slice j descending
// over the pairs with D0 <= i + j < D0 + ND, i + j <= S - 1
template <int S, int D0, int ND, int WA>
struct WSched {
  static constexpr int SL = (D0 + ND < S) ? (D0 + ND) : S;
  static constexpr int MAXS = WA * SL * SL, MAXG = WA * SL;
  int ns = 0, ng = 0;
  int sa[MAXS] = {}, si[MAXS] = {}, sj[MAXS] = {}, sg[MAXS] = {};
  int gfirst[MAXG] = {}, g_a[MAXG] = {}, g_i[MAXG] = {};
  constexpr WSched() {
    for (int a = 0; a < WA; a++)
      for (int i = 0; i < SL; i++) {
        bool any = false;
        for (int j = SL - 1; j >= 0; j--) {
          const int d = i + j;
          if (d < D0 || d >= D0 + ND || d > S - 1) continue;
          if (!any) {
            gfirst[ng] = ns;
            g_a[ng] = a;
            g_i[ng] = i;
            any = true;
          }
          sa[ns] = a; si[ns] = i; sj[ns] = j; sg[ns] = ng;
          ns++;
        }
        if (any) ng++;
      }
  }
  constexpr int max_j_from(int s0) const {
    int m = 0;
    for (int s = s0; s < ns; s++) m = sj[s] > m ? sj[s] : m;
    return m;
  }
  // last slot in [from, ns) ... of the FIRST group that uses a B slice j < jt (the first group runs j descending)
  constexpr int last_use_of_j_below(int jt, int group) const {
    int last = 0;
    for (int s = 0; s < ns; s++)
      if (sg[s] == group && sj[s] < jt) last = s;
    return last;
  }
};

template <int S, int D0, int ND, int WA>
inline constexpr WSched<S, D0, ND, WA> kWSched{};

// VARW bits
constexpr int VARW_NA3 = 1;        // 3 A buffers (prefetch distance 2); else 2 (distance 1)
constexpr int VARW_NO_GLOBAL = 2;  // ablation: no staging (LDS holds garbage)
constexpr int VARW_MFMA_ONLY = 4;  // ablation: no LDS reads either
constexpr int VARW_NO_EPILOGUE = 128; // measurement: accumulators are only kept alive, nothing is converted or stored
constexpr int VARW_B1 = 1024;      // ONE wave-private B buffer instead of two (see w_tile): passes with 14+ staged slices
constexpr int VARW_HALF_BARRIERS = 2048; // measurement, WRONG RESULTS: the per-step barrier on every second k-step only
constexpr int VARW_EPI_NOSTORE = 256; // measurement: epilogue without its stores
constexpr int VARW_EPI_NOCHAIN = 512; // measurement: epilogue without its FP64 chains
constexpr int VARW_TRACE = 64;     // measurement: cycle stamps of k-steps 100..107 of the first 32 workgroups -> p.acc; k64 tile: wall-clock stamps of every tile's phases (slice_gemm_y_tile.h)
constexpr int VARW_BAND4 = 16;     // measurement: XCD patch of 4 (M) x 8 (N) tiles instead of 8 x 4
constexpr int VARW_BAND16 = 32;    // measurement: 16 x 2
constexpr int VARW_K64 = 8192;     // k64 tile: v_mfma_i32_16x16x64_i8, one slice product over 64 k per instruction (slice_gemm_y_tile.h)
constexpr int VARW_BREG = 16384;   // k64 tile: B fragments global -> VGPR (two register sets) instead of through a wave-private LDS stage
constexpr int VARW_ACCN = 32768;   // k64 tile: the accumulators are NAMED registers (a[0:255] + v[160:255]), not compiler values (slice_gemm_y_tile.h)
constexpr int VARW_BHI = 65536;    // k64 tile: the B slices j >= 9 global -> named VGPRs, refilled IN PLACE behind their last use; the rest through LDS
constexpr int VARW_ZFILL = 131072; // k64 register kernel: zero the accumulators in front of the tile instead of letting its first step write them (the multi-product kernels: slice_gemm_y_tile.h)
constexpr int VARW_X16 = 4096;     // paired tile: v_mfma_i32_16x16x64_i8, two slice products per instruction (slice_gemm_x_tile.h)

// One output tile of (32*WA) x 128: rows start at A row-block rb0, columns at B row-block 4*tn.
template <int S, int D0, int ND, int WA, int VARW, int STAG, int DMA0_, int DMAE_, int TAIL_>
__device__ __forceinline__ void w_tile(const SliceGemmArgs &p, char *smem, const uint32_t rb0, const uint32_t tn,
                                       const uint32_t xcd) {
  constexpr int SL = (D0 + ND < S) ? (D0 + ND) : S;
  constexpr int NA = (VARW & VARW_NA3) ? 3 : 2;
  constexpr int PD = NA - 1;                       // prefetch distance in k-steps
  constexpr int A_STAGE = WA * SL * FRAG_BYTES;    // shared
  constexpr int B_STAGE = 4 * SL * FRAG_BYTES;     // 4 wave-private runs of SL blocks
  constexpr int OFF_B = NA * A_STAGE;
  // B buffers per wave.  A wave holds ALL B fragments of a k-step in registers from the end of the previous step on (they
  // are consumed by the step's first group, j descending, before the first copy of the next stage is issued at slot
  // DMA0 = the first slot of group 1), so the next stage's B may land in the very buffer the current one came from: one
  // buffer is enough at prefetch distance 1.  The shipped configurations keep two (unchanged timing); VARW_B1 is what
  // lets the second diagonal pass of S = 14..18 (14-18 staged slices) use 64x128 tiles in 160 KiB.
  constexpr int NB = (VARW & VARW_B1) ? 1 : 2;
  static_assert(NB == 2 || PD == 1, "a single B buffer needs prefetch distance 1");
  constexpr int NQA = (WA * SL + 3) / 4;           // A blocks copied per wave per stage
  constexpr int NDMA = NQA + SL;                   // copies per wave per stage
  constexpr int R = 4;                             // A fragment ring (registers)
  constexpr int NG = WA * SL;                      // MFMA groups per k-step: one per (A block, A slice)
  constexpr bool NO_GLOBAL = (VARW & (VARW_NO_GLOBAL | VARW_MFMA_ONLY)) != 0;
  constexpr bool MFMA_ONLY = (VARW & VARW_MFMA_ONLY) != 0;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  // ---- staging -------------------------------------------------------------------------------------------------
  // A stage: WA*SL blocks numbered q = a*SL + s; wave w copies the run q = w*NQA .. w*NQA+NQA-1 (clamped: the last
  // wave repeats the final block, so every wave issues the same number of copies and the vmcnt arithmetic is uniform).
  const size_t rb_stride = (size_t)p.KB * (size_t)(S * FRAG_BYTES);
  const uint32_t rba_last = p.rba - 1u; // row-blocks the planes hold (rows are padded to 128, tiles to 32*WA)
  const uint32_t lane_off = (uint32_t)lane * 16u;
  const size_t pass0 = (size_t)p.kb0 * (S * FRAG_BYTES); // first k-block of this pass (k position is relative to it)
  const int8_t *a_src[NQA]; // loop-invariant source of the wave's t-th A block (SGPR pairs)
  uint32_t a_lds[NQA];      // its offset inside an A stage
#pragma unroll
  for (int t = 0; t < NQA; t++) {
    uint32_t q = (uint32_t)(wave * NQA + t);
    if (q > (uint32_t)(WA * SL - 1)) q = WA * SL - 1;
    const uint32_t a = q / SL, s = q - a * SL;
    uint32_t rb = rb0 + a;
    if (rb > rba_last) rb = rba_last;
    a_src[t] = uniform_ptr(p.a_planes + rb * rb_stride + s * FRAG_BYTES + pass0);
    a_lds[t] = q * FRAG_BYTES;
  }
  const int8_t *b_src = uniform_ptr(p.b_planes + (size_t)(4u * tn + wave) * rb_stride + pass0);

  // The copies are inline asm (glds16): hipcc counts an LDS-DMA builtin on BOTH vmcnt and lgkmcnt, after which every
  // wait it places in front of a fragment's first use is lgkmcnt(0) -- draining the ring read issued just before it
  // (measured in the ISA: 18 x lgkmcnt(0) per k-step instead of counted lgkmcnt(3)).  Invisible to the compiler, the
  // copies are ordered by the explicit counted vmcnt waits below.
  const uint32_t lds0 = (uint32_t)(size_t)((OZ_AS3 char *)smem);
  const uint32_t ldsb0 = lds0 + OFF_B + wave * (SL * FRAG_BYTES);
  // voff: lane*16 + k-block * S KiB (32 bits: the host keeps a pass below 2^32 bytes per row-block, slice_gemm.hip);
  // lds_a / lds_b: LDS base of the A / B buffer being filled
  auto copy_a = [&](int t, uint32_t voff, uint32_t lds_a) {
    if constexpr (NO_GLOBAL) return;
    glds16<0>(a_src[t], voff, lds_a, a_lds[t]);
  };
  auto copy_b = [&](auto sc, uint32_t voff, uint32_t lds_b) {
    if constexpr (NO_GLOBAL) return;
    constexpr int s = decltype(sc)::value;
    constexpr int G = 4; // blocks per immediate-offset group (the offset also advances the LDS address): 0 .. 3072
    constexpr int g0 = s / G * G;
    glds16<(s % G) * FRAG_BYTES>(b_src + g0 * FRAG_BYTES, voff, lds_b, (uint32_t)(g0 * FRAG_BYTES));
  };
  // copy number c (0..NDMA-1) of a stage: A share first (the other waves wait for it), then the private B run
  auto copy_n = [&](auto cc, int abuf, int bbuf, uint32_t kstep) {
    constexpr int c = decltype(cc)::value;
    const uint32_t voff = lane_off + kstep * (uint32_t)(S * FRAG_BYTES);
    if constexpr (c < NQA) {
      copy_a(c, voff, lds0 + abuf * A_STAGE);
    } else {
      // one B buffer: the copy overwrites what the current step's B fragments were read from.  Those reads were issued a
      // step's tail ago and have long returned; the wait makes that a guarantee instead of a timing argument.
      if constexpr (NB == 1 && c == NQA) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      copy_b(std::integral_constant<int, c - NQA>{}, voff, ldsb0 + bbuf * B_STAGE);
    }
  };

  // ---- accumulators: WA*ND x 16 registers, placed by hand ---------------------------------------------------------
  // The unified file gives a lone wave 256 VGPRs + 256 AGPRs.  hipcc's MFMA builtin wants every accumulator in ONE of
  // the two halves and shuffles the excess through v_accvgpr copies and scratch; inline-asm MFMAs with explicit register
  // classes pin accumulators 0..15 to the AGPR half and the rest to VGPRs.  Hazards the compiler no longer pads
  // (guide §5.7): MFMA -> MFMA on the same accumulator needs no wait state; the first MFMA after the zero-fill and the
  // first VALU read after the last MFMA are covered by the s_nop statements below.
  constexpr int NACC = WA * ND, NACC_A = NACC < 16 ? NACC : 16, NACC_V = NACC > 16 ? NACC - 16 : 1;
  v16i accA[NACC_A], accV[NACC_V];
#pragma unroll
  for (int x = 0; x < NACC_A; x++)
#pragma unroll
    for (int r = 0; r < 16; r++) accA[x][r] = 0;
#pragma unroll
  for (int x = 0; x < NACC_V; x++)
#pragma unroll
    for (int r = 0; r < 16; r++) accV[x][r] = 0;
  auto mfma = [&](auto xc, const v4i &b, const v4i &a) {
    constexpr int X = decltype(xc)::value;
    if constexpr (X < 16)
      mfma_agpr(accA[X], b, a);
    else
      mfma_vgpr(accV[X - 16], b, a);
  };

  // ---- circular K with the per-XCD phase hint (slice_gemm_kernel.h) ----------------------------------------------
  const uint32_t nk = p.kb1 - p.kb0;
  uint32_t *phase = p.phase ? p.phase + (uint32_t)PHASE_LINE_WORDS * xcd : nullptr;
  uint32_t koff = 0;
  // (SliceGemmArgs::phase_min_kb: short passes run without the hint - its read is a device-scope load and two workgroup
  // barriers, ~1.5 us per tile)
  if (phase && nk > p.phase_min_kb) {
    if (threadIdx.x == 0)
      *(volatile uint32_t *)smem = __hip_atomic_load(phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    koff = (*(volatile uint32_t *)smem + 2u) % nk;
    __syncthreads();
  } else {
    phase = nullptr;
  }
  koff = __builtin_amdgcn_readfirstlane(koff);
  auto koff_next = [&](uint32_t k) { return k + 1 == nk ? 0u : k + 1; };

  v4i cf[MFMA_ONLY ? SL : 1]; // MFMA-only ablation: full-entropy operands in registers
  if constexpr (MFMA_ONLY) {
#pragma unroll
    for (int s = 0; s < SL; s++) {
      uint32_t x = (uint32_t)(lane * SL + s) * 2654435761u + blockIdx.x;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        cf[s][c] = (int)x;
      }
      asm volatile("" : "+v"(cf[s]));
    }
  }

  // ---- schedule of one k-step (compile time) ---------------------------------------------------------------------
  // SC lists the MFMA slots of a k-step: block a outermost, A slice i ascending, B slice j descending; a "group" is the
  // run of slots that share one A fragment (a, i).  Events are attached to slots:
  //   * the fragment of group 0 has its own register af0, (re)loaded right behind the barrier for the next k-step: the
  //     step's first MFMA must not wait for an LDS read issued at the end of the previous step;
  //   * groups 1.. go through a ring of R registers: first slot of group g: read the fragment of group g+R-1 into the
  //     ring slot that group g-1 just released.  Ring positions are padded to a multiple of R per k-step (NRP) so that
  //     a group's slot is the same in every step and the prefetch runs across the step boundary into the next buffer;
  //   * slots DMA0, DMA0+DMAE, ...: one LDS-DMA copy of the stage PD steps ahead (A share first, then the private B);
  //   * slot NS-TAIL ("X"): wait for the own copies of the NEXT stage, barrier, then refresh bf[JT..] from it; the TAIL
  //     remaining MFMAs of this step use bf[0..JT-1] only and cover the barrier and the LDS latency; bf[0..JT-1] are
  //     refreshed behind the last MFMA and are needed last in the next step's first group (j descending).
#define SC (kWSched<S, D0, ND, WA>) /* namespace-scope constexpr object: usable inside the lambdas below */
  constexpr int NS = SC.ns;
  constexpr int NRP = (NG - 1 + R - 1) / R * R; // ring positions per k-step: group g >= 1 has position g - 1
  constexpr int TAIL = TAIL_ < NS ? TAIL_ : NS - 1;
  constexpr int XS = NS - TAIL;
  constexpr int DMA0 = DMA0_ >= 0 ? DMA0_ : SC.gfirst[1 < NG ? 1 : 0]; // default: behind the first group
  constexpr int DMAE_FIT = NDMA > 1 ? (XS - 1 - DMA0) / (NDMA - 1) : 1;
  constexpr int DMAE = DMAE_ < DMAE_FIT ? DMAE_ : (DMAE_FIT > 1 ? DMAE_FIT : 1);
  constexpr int JT = SC.max_j_from(XS) + 1; // B slices still needed after X
  static_assert(SC.ng == NG, "empty (a, i) groups are not supported by the ring arithmetic");
  // next-step group q (1 <= q < R) is read at the first slot of group NRP + q - R + 1 if that group exists (else
  // behind the last slot): never before the barrier
  static_assert(NRP + 2 - R > NG - 1 || SC.gfirst[NRP + 2 - R <= NG - 1 && NRP + 2 - R >= 0 ? NRP + 2 - R : 0] >= XS,
                "the first ring read of the next stage must come after the barrier");
  static_assert(NG >= 2, "at least two groups per k-step");
  static_assert(DMA0 + (NDMA - 1) * DMAE < XS, "every copy of a stage is issued before the barrier slot");
  static_assert(PD == 1 || DMA0 + NQA * DMAE >= SC.last_use_of_j_below(JT, 0) + 1,
                "distance 2: the B copies overwrite the buffer bf[0..JT-1] were read from at the end of the last step");

  const char *la0 = smem + lane * 16;
  const char *lb0 = smem + OFF_B + wave * (SL * FRAG_BYTES) + lane * 16;
  int abuf = 0, bbuf = 0; // buffers of the stage being computed
  v4i bf[SL], af[R], af0;
  auto read_a = [&](auto gc, const char *la) { // A fragment of group g -> af0 (g == 0) or ring slot (g - 1) % R
    constexpr int g = decltype(gc)::value;
    const v4i f = *(const v4i *)(la + (SC.g_a[g] * SL + SC.g_i[g]) * FRAG_BYTES);
    if constexpr (g == 0)
      af0 = f;
    else
      af[(g - 1) % R] = f;
  };

  // ---- prologue: stages 0 .. PD-1 in flight, stage 0 landed, its B fragments and first A fragments in registers --
  uint32_t k_issue = koff;
  int issued = 0;
#pragma unroll
  for (int d = 0; d < PD; d++) {
    if ((uint32_t)d < nk) {
      static_for<NDMA>([&](auto cc) { copy_n(cc, d % NA, NB == 1 ? 0 : d % 2, k_issue); });
      k_issue = koff_next(k_issue);
      issued++;
    }
  }
  if constexpr (!NO_GLOBAL) {
    if (issued == 2)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if constexpr (!MFMA_ONLY) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < SL; j++) bf[j] = *(const v4i *)(lb0 + j * FRAG_BYTES);
    static_for<(R < NG ? R : NG)>([&](auto gc) { read_a(gc, la0); }); // group 0 and the ring's first R - 1 entries
  }
  asm volatile("s_nop 7" ::: "memory"); // zero-fill (VALU / v_accvgpr_write) -> first MFMA reading it as C

  uint32_t it = 0;
  // s_memtime returns through the scalar memory path (lgkmcnt): the stamps of a step are only read after the one
  // `s_waitcnt lgkmcnt(0)` statement at its end that names them all (guide §5.7 form (ii))
  auto stamp = [&]() -> unsigned long long {
    unsigned long long t;
    asm volatile("s_memtime %0" : "=s"(t));
    return t;
  };
  // one k-step.  PF: a stage is prefetched during it; NX: a next stage exists (barrier + refresh at X)
  auto step = [&](auto pf_tag, auto nx_tag) {
    constexpr bool PF = decltype(pf_tag)::value, NX = decltype(nx_tag)::value;
    constexpr bool TRACE = (VARW & VARW_TRACE) != 0;
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (TRACE) ts[5] = stamp();
    const int abuf_n = abuf + 1 == NA ? 0 : abuf + 1;                        // next stage
    const int abuf_pf = (PD == 1) ? abuf_n : (abuf_n + 1 == NA ? 0 : abuf_n + 1); // stage PD ahead
    const int bbuf_pf = NB == 1 ? 0 : (PD == 1) ? (bbuf ^ 1) : bbuf;
    const uint32_t kb_pf = k_issue; // k-step (relative to the pass) of the stage being prefetched
    if constexpr (PF) k_issue = koff_next(k_issue);
    const char *la = la0 + abuf * A_STAGE;
    const char *la_n = la0 + abuf_n * A_STAGE;
    const char *lb_n = lb0 + (NB == 1 ? 0 : (bbuf ^ 1) * B_STAGE);
    static_for<NS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int g = SC.sg[s];
      if constexpr (s == XS && NX && !MFMA_ONLY) {
        if constexpr (TRACE) ts[0] = stamp();
        if constexpr (!NO_GLOBAL) {
          if constexpr (PF && PD > 1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * NDMA) : "memory");
          else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (TRACE) ts[1] = stamp();
        // NA == 2: the buffer this barrier releases is refilled right behind it, so this wave's reads of it must have
        // returned.  NA == 3: it is refilled one k-step later; the reads (issued >= 4 MFMAs ago) are long gone by then.
        if constexpr (NA == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (TRACE) ts[2] = stamp();
        if constexpr ((VARW & VARW_HALF_BARRIERS) != 0) {
          if (it & 1u) __builtin_amdgcn_s_barrier();
        } else
          __builtin_amdgcn_s_barrier(); // A of the next stage visible; nobody reads A of this stage from LDS any more
        asm volatile("" ::: "memory");
        if constexpr (TRACE) ts[3] = stamp();
        if constexpr (STAG > 0) // de-phase the 4 lockstep waves so that their copies do not queue in the TA
          for (int q = 0; q < wave; q++) asm volatile("s_nop %0" ::"n"(STAG - 1));
        read_a(std::integral_constant<int, 0>{}, la_n); // next step's first fragment (group 0 of this step is long done)
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (SC.gfirst[g] == s && g >= 1 && !MFMA_ONLY) {
        constexpr int gn = g + R - 1; // ring position gn - 1 = (g - 1) + R - 1
        if constexpr (gn < NG) {
          read_a(std::integral_constant<int, gn>{}, la);
          __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (gn - 1 >= NRP && NX) {
          read_a(std::integral_constant<int, gn - NRP>{}, la_n);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (PF && s >= DMA0 && (s - DMA0) % DMAE == 0 && (s - DMA0) / DMAE < NDMA) {
        if constexpr (TRACE && s == DMA0 + 4 * DMAE) ts[6] = stamp();
        copy_n(std::integral_constant<int, (s - DMA0) / DMAE>{}, abuf_pf, bbuf_pf, kb_pf);
        if constexpr (TRACE && s == DMA0 + 4 * DMAE) ts[7] = stamp();
        __builtin_amdgcn_sched_barrier(0);
      }
      constexpr int a = SC.sa[s], i = SC.si[s], j = SC.sj[s];
      if constexpr (MFMA_ONLY)
        mfma(std::integral_constant<int, a * ND + i + j - D0>{}, cf[j], cf[i]);
      else
        mfma(std::integral_constant<int, a * ND + i + j - D0>{}, bf[j], g == 0 ? af0 : af[(g >= 1 ? g - 1 : 0) % R]);
      // behind the barrier the matrix pipe gets its next MFMA first; the refresh of bf[JT..] from the next stage (and
      // the phase hint) follow in the shadow of the TAIL MFMAs, RPT reads per slot
      if constexpr (s >= XS && NX && !MFMA_ONLY) {
        constexpr int RPT = (SL - JT + TAIL - 1) / TAIL;
#pragma unroll
        for (int q = 0; q < RPT; q++) {
          const int jj = JT + (s - XS) * RPT + q;
          if (jj < SL) bf[jj] = *(const v4i *)(lb_n + jj * FRAG_BYTES);
        }
        if constexpr (s == XS + 1 || (TAIL == 1 && s == XS))
          if (phase && (it & 3u) == 0 && threadIdx.x == 0)
            __hip_atomic_store(phase, koff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NX && !MFMA_ONLY) {
#pragma unroll
      for (int j = 0; j < JT; j++) bf[j] = *(const v4i *)(lb_n + j * FRAG_BYTES);
      // next-stage ring fragments whose trigger group does not exist (padding positions)
      static_for<R - 1>([&](auto qc) {
        constexpr int q = decltype(qc)::value + 1; // next-step groups 1 .. R-1
        if constexpr (NRP + q - R + 1 > NG - 1 && q < NG) read_a(std::integral_constant<int, q>{}, la_n);
      });
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (TRACE) {
      ts[4] = stamp();
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+s"(ts[0]), "+s"(ts[1]), "+s"(ts[2]), "+s"(ts[3]), "+s"(ts[4]), "+s"(ts[5]), "+s"(ts[6]), "+s"(ts[7]));
      if (blockIdx.x < 32 && it >= 100 && it < 108 && lane == 0) {
        uint32_t *tr = reinterpret_cast<uint32_t *>(p.acc) + ((blockIdx.x * 4 + wave) * 8 + (it - 100)) * 8;
#pragma unroll
        for (int q = 0; q < 8; q++) tr[q] = (uint32_t)ts[q];
      }
    }
    koff = koff_next(koff);
    abuf = abuf_n;
    bbuf ^= 1;
  };
  for (; it + PD < nk; it++) step(std::true_type{}, std::true_type{});
  for (; it + 1 < nk; it++) step(std::false_type{}, std::true_type{});
  for (; it < nk; it++) step(std::false_type{}, std::false_type{});

  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); // last MFMA (16 passes) -> VALU reads of its accumulator
  // One explicit v_accvgpr_read per element: left to itself the compiler copies the whole remaining part of a 16-register
  // AGPR tuple to VGPRs for every element it extracts (5x the reads, plus spills).
  auto acc = [&](int a, int d, int r) -> int {
    const int x = a * ND + d;
    if (x >= 16) return accV[x >= 16 ? x - 16 : 0][r];
    int v;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(accA[x < 16 ? x : 0][r]));
    return v;
  };
  if constexpr ((VARW & VARW_NO_EPILOGUE) != 0) {
#pragma unroll
    for (int x = 0; x < NACC_A; x++) asm volatile("" ::"a"(accA[x]));
#pragma unroll
    for (int x = 0; x < (NACC > 16 ? NACC_V : 0); x++) asm volatile("" ::"v"(accV[x]));
    return;
  }
  recombine_and_store<D0, ND, WA, (VARW >> 8) & 3>(p, acc, rb0 * 32 + (lane & 31), tn * 128 + wave * 32 + 4 * (lane >> 5));
}

} // namespace ozhip
#include "slice_gemm_x_tile.h"
#include "slice_gemm_y_tile.h"
namespace ozhip {

// contiguous run of logical ids for XCD x (of nx) out of n (bijective form of the guide's T1 swizzle)
__device__ __forceinline__ uint32_t xcd_run_start(uint32_t x, uint32_t n, uint32_t nx) {
  const uint32_t q = n / nx, r = n % nx;
  return x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
}
// logical id inside a region of rows x cols tiles -> (row, col): bands of 8 rows, columns outer inside a band, so
// that the 32 workgroups an XCD runs concurrently form an 8 (M) x 4 (N) patch that shares A and B panels in its L2
template <uint32_t BH = 8>
__device__ __forceinline__ void band_order(uint32_t lid, uint32_t rows, uint32_t cols, uint32_t &r, uint32_t &c) {
  const uint32_t band_tiles = BH * cols, nbands = (rows + BH - 1u) / BH;
  uint32_t band = lid / band_tiles;
  if (band > nbands - 1) band = nbands - 1;
  const uint32_t rem = lid - band * band_tiles;
  const uint32_t h = (rows - band * BH) < BH ? (rows - band * BH) : BH;
  c = rem / h;
  r = band * BH + rem % h;
}

// Tiles: tiles_m x tiles_n "big" tiles of WA blocks + tiles_m2 x tiles_n "small" tiles of WA-1 blocks below them.  The
// mix lets the host fit the row count and the number of CU rounds (8192 rows = 80 x 96 + 8 x 64: 20 + 2 full rounds
// on 256 CUs instead of 21.5 -> 22 with 96-row tiles only).  Every XCD owns a contiguous run of each region (L2 panel
// sharing), big tiles first.
//
// p.queue == nullptr: one tile per workgroup, grid = number of tiles; the extras of the small region start at the XCD
// where the big region's extras stop, so the per-XCD totals equal what the round-robin dispatch hands each XCD.
// p.queue != nullptr: PERSISTENT workgroups (grid = one per CU) claim tiles from their XCD's run through two counters per
// XCD and, when it is exhausted, from the other XCDs' runs.  The XCDs do not run at the same speed (measured: the odd
// ones finish an 8192^3 launch up to 0.9 ms after the even ones, profiles/r2_ablate/*timeline*); a static partition
// leaves that as an idle tail on 3 % of the CU time.
// `g`: the products of this launch (COUNT = 1: one slice GEMM; up to 4: the real products of a ZGEMM, accumulated into the
// same C in the given order, every claimed tile walked through all of them before the next tile is claimed - each
// element of C sees the same sequence of updates as with one launch per product).  Geometry, queues and phase hints are
// those of g[0].
// The barriers of the tile loop order LDS traffic only (the previous tile's fragment reads against the claim's two words, the words
// against the next tile's staging).  __syncthreads() also drains vmcnt - i.e. waits for the tile's stores of C to be acknowledged
// (~0.5 us, profiles/r6_ablate/: "claim + entry") although nothing that follows depends on them: the next tile's prologue issues
// for 2 us before its own vmcnt(0).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// the prologue hook of the persistent kernels' tiles (slice_gemm_y_tile.h: prologue_hook): phase 1 draws the NEXT tile's ticket
template <bool SPEC>
struct TicketHook {
  uint32_t *ticket;      // thread 0's spec_t
  uint32_t **counter;    // ... spec_cnt
  const uint32_t *region; // ... spec_region (0: nothing to draw)
  template <class P>
  __device__ __forceinline__ void operator()(P) const {
    if constexpr (SPEC && P::value == 1) {
      if (threadIdx.x == 0 && *region) {
        const uint32_t one = 1u;
        uint32_t t;
        asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(t) : "v"(*counter), "v"(one) : "memory");
        *ticket = t;
      }
    }
  }
};

template <int S, int D0, int ND, int WA, int VARW, int STAG, int DMA0, int DMAE, int TAIL_, bool MULTI>
__device__ __forceinline__ void w_persistent(const SliceGemmArgs *g, const int count, char *smem) {
  const SliceGemmArgs &p = g[0];
  auto tile = [&](auto wa_tag, uint32_t rb0, uint32_t c, uint32_t xcd, auto &&hook) {
    constexpr int W = decltype(wa_tag)::value;
    auto one = [&](const SliceGemmArgs &q) {
      if constexpr ((VARW & VARW_K64) != 0)
        y_tile<S, D0, ND, W, (VARW & ~VARW_K64) | (MULTI ? VARW_ZFILL : 0), STAG, DMA0, DMAE, TAIL_, OZ_Y_RING>(q, smem, rb0, c, xcd, hook);
      else if constexpr ((VARW & VARW_X16) != 0)
        x_tile<S, D0, ND, W, VARW & ~VARW_X16, STAG, DMA0, DMAE, TAIL_>(q, smem, rb0, c, xcd);
      else
        w_tile<S, D0, ND, W, VARW, STAG, DMA0, DMAE, TAIL_>(q, smem, rb0, c, xcd);
    };
    if constexpr (!MULTI) {
      one(p);
    } else {
#pragma unroll 1
      for (int i = 0; i < count; i++) {
        if (i) __syncthreads(); // the previous product's LDS reads are done
        one(g[i]);
      }
    }
  };
  unsigned long long wg_t0 = 0;
  if constexpr ((VARW & VARW_TRACE) != 0) wg_t0 = wall_clock64();
  const uint32_t nbig = p.tiles_m * p.tiles_n, nsmall = p.tiles_m2 * p.tiles_n;
  constexpr uint32_t BH = (VARW & VARW_BAND4) ? 4u : (VARW & VARW_BAND16) ? 16u : 8u;
  const uint32_t nx = p.nxcd; // XCDs the host planned for (topology.h): >= 1; emulated counts fold the hardware id onto it
  uint32_t xcd = blockIdx.x % nx;
  if (p.queue) { // the XCD this workgroup really runs on (placement is not architecturally tied to blockIdx)
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    xcd = __builtin_amdgcn_readfirstlane((x & 0xfu) % nx);
  }
  const uint32_t small_shift = nbig % nx; // the small region's runs are rotated by the big region's remainder
  // speculative claim (see below): only the k64 kernels with at most 320 accumulator registers carry its five extra live
  // values through their k loop without spilling (the 432 / 448-register kernels flip into scratch on ONE more: DESIGN.md 4.2)
  constexpr bool SPECULATE = !MULTI && (VARW & VARW_K64) != 0 && 2 * WA * 2 * ND * 4 <= 320 &&
                             (VARW & (VARW_NO_GLOBAL | VARW_MFMA_ONLY)) == 0; // (the tile-boundary trace of the k64 tile keeps it)
  uint32_t spec_region = 0, spec_from = 0, spec_len = 0, spec_start = 0, spec_t = 0; // meaningful in thread 0 only
  uint32_t *spec_cnt = nullptr;
  // The ticket of the NEXT tile is drawn inside the current tile's prologue (y_tile: prologue_hook), between the issue of the
  // first stage's copies and the wait for them, and the SAME asm statement waits for it: the fetch-add's round trip overlaps
  // the copies' latency, which the wave waits out anyway, and the compiler only ever sees a register whose value has arrived.
  // (Round 4 issued the atomic at the tile boundary and waited for it one tile later: between the two the destination VGPR was
  // in flight while the compiler believed it defined - a copy or a spill inside that window would have captured a stale
  // ticket; ADVICE r4.  tests/test_isa_invariants.py: every returning atomic of these kernels is followed by its wait.)
  // (Round 6 also tried issuing the atomic in FRONT of the tile's copies - into v255, an accumulator register that is dead until the
  // first k-step - and collecting it behind them: the round trip then overlaps the ~1 us the 27 copies take to issue.  Measured
  // (profiles/r6_ablate/r6p_*): the ticket's wait went 0.72 -> 0.36 us, but wave 0's copies queue behind the atomic - "copies issued"
  // 1.0 -> 1.1 us at 32768^2 x 1024 and -> 1.7 us at 8192^2 x 256, where every workgroup of an XCD draws from one counter at the same
  // moment: neutral to negative.  Dropped.)
  const TicketHook<SPECULATE> draw_ticket{&spec_t, &spec_cnt, &spec_region};
  for (;;) {
    uint32_t kind = 0, lid = 0; // 1: big tile `lid` of its region, 2: small tile, 0: nothing left
    if (!p.queue) {
      const uint32_t idx = blockIdx.x / nx;
      const uint32_t nbig_x = nbig / nx + (xcd < nbig % nx ? 1u : 0u);
      if (idx < nbig_x) {
        kind = 1;
        lid = xcd_run_start(xcd, nbig, nx) + idx;
      } else {
        kind = 2;
        lid = xcd_run_start((xcd + nx - small_shift) % nx, nsmall, nx) + (idx - nbig_x);
      }
    } else {
      lds_barrier(); // the previous tile's LDS reads are done
      if (threadIdx.x == 0) {
        uint32_t k = 0, l = 0;
        if constexpr (SPECULATE) {
          if (spec_region) { // the ticket drawn in the previous tile's prologue (draw_ticket)
            if (spec_t < spec_len) {
              k = spec_region;
              l = spec_start + spec_t;
            }
          }
        }
        for (uint32_t region = 1; region <= 2 && !k; region++) {
          const uint32_t n = region == 1 ? nbig : nsmall;
          for (uint32_t v = 0; v < nx && !k; v++) { // own run first, then the neighbours'
            const uint32_t x = (xcd + v) % nx, xr = region == 1 ? x : (x + nx - small_shift) % nx;
            const uint32_t len = n / nx + (xr < n % nx ? 1u : 0u);
            uint32_t *cnt = p.queue + (uint32_t)PHASE_LINE_WORDS * x + (region - 1);
            if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= len) continue;
            const uint32_t t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t < len) {
              k = region;
              l = xcd_run_start(xr, n, nx) + t;
              if constexpr (SPECULATE) { // where this tile came from is where the next ticket is drawn
                spec_cnt = cnt;
                spec_len = len;
                spec_start = xcd_run_start(xr, n, nx);
                spec_from = region;
              }
            }
          }
        }
        reinterpret_cast<volatile uint32_t *>(smem)[0] = k;
        reinterpret_cast<volatile uint32_t *>(smem)[1] = l;
        if constexpr (SPECULATE) {
          // Short k loops (K <= 2048): a tile lasts tens of microseconds and the ~2 us round trips of its claim (a device-scope
          // load and a fetch-add, nothing else running on the CU) are a few percent of it.  The NEXT tile's ticket is drawn
          // from the counter this tile came from during this tile's prologue (draw_ticket above) and looked at at the next
          // boundary.  A ticket beyond the run's end is harmless (readers compare with >=); the full claim above then looks
          // elsewhere.  At most one tile per workgroup is held early, so the stealing granularity suffers by less than one
          // short tile at the kernel's end.
          spec_region = (k && spec_from == k && p.kb1 - p.kb0 <= p.spec_claim_kb) ? k : 0u;
        }
      }
      lds_barrier();
      kind = __builtin_amdgcn_readfirstlane(reinterpret_cast<volatile uint32_t *>(smem)[0]);
      lid = __builtin_amdgcn_readfirstlane(reinterpret_cast<volatile uint32_t *>(smem)[1]);
      lds_barrier(); // smem is staging space again
      if (!kind) break;
    }
    uint32_t r, c;
    if (kind == 1) {
      band_order<BH>(lid, p.tiles_m, p.tiles_n, r, c);
      r = __builtin_amdgcn_readfirstlane(r); // wave-uniform by construction; say so (the copies take SGPR operands)
      c = __builtin_amdgcn_readfirstlane(c);
      tile(std::integral_constant<int, WA>{}, WA * r, c, xcd, draw_ticket);
    } else {
      if constexpr (WA > 1) {
        band_order<BH>(lid, p.tiles_m2, p.tiles_n, r, c);
        r = __builtin_amdgcn_readfirstlane(r);
        c = __builtin_amdgcn_readfirstlane(c);
        tile(std::integral_constant<int, WA - 1>{}, WA * p.tiles_m + (WA - 1) * r, c, xcd, draw_ticket);
      }
    }
    if (!p.queue) break;
  }
  if constexpr ((VARW & VARW_TRACE) != 0) { // per-workgroup placement and wall-clock span (100 MHz), after the step stamps
    if (threadIdx.x == 0) {
      unsigned xcc, hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      unsigned long long *w = reinterpret_cast<unsigned long long *>(p.acc) + 4096 + (size_t)blockIdx.x * 3;
      w[0] = ((unsigned long long)(xcc & 0xf) << 32) | (hw & 0xff00u);
      w[1] = wg_t0;
      w[2] = wall_clock64();
    }
  }
}

template <int S, int D0, int ND, int WA, int VARW, int STAG = 0, int DMA0 = -1, int DMAE = 4, int TAIL_ = 6>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void slice_gemm_w_kernel(
    const SliceGemmArgs p_in) {
  const SliceGemmArgs p = batch_view(p_in);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  w_persistent<S, D0, ND, WA, VARW, STAG, DMA0, DMAE, TAIL_, false>(&p, 1, smem);
}

// the real products of a ZGEMM in ONE persistent launch (kernels.h: SliceGemmMulti; no batches)
template <int S, int D0, int ND, int WA, int VARW, int STAG = 0, int DMA0 = -1, int DMAE = 4, int TAIL_ = 6>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void slice_gemm_w_multi_kernel(
    const SliceGemmMulti m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  w_persistent<S, D0, ND, WA, VARW, STAG, DMA0, DMAE, TAIL_, true>(m.g, m.count, smem);
}

#undef SC

} // namespace ozhip
