// slice-GEMM kernels and launch policy of fp64_int8_11 (see slice_gemm_launch.h, slice_gemm.hip: OZ_GEMM_PARTS)
#define OZ_S_LO 11
#define OZ_S_HI 11
#define OZ_PART launch_slice_gemm_s11_11
#define OZ_PART_FUSED launch_slice_gemm_fused_s11_11
#define OZ_PART_TRAITS slice_gemm_traits_s11_11
#include "slice_gemm_launch.h"
