// slice-GEMM kernels and launch policy of fp64_int8_10 (see slice_gemm_launch.h, slice_gemm.hip: OZ_GEMM_PARTS)
#define OZ_S_LO 10
#define OZ_S_HI 10
#define OZ_PART launch_slice_gemm_s10_10
#define OZ_PART_FUSED launch_slice_gemm_fused_s10_10
#define OZ_PART_TRAITS slice_gemm_traits_s10_10
#include "slice_gemm_launch.h"
