// slice-GEMM kernels and launch policy of fp64_int8_18 (see slice_gemm_launch.h, slice_gemm.hip: OZ_GEMM_PARTS)
#define OZ_S_LO 18
#define OZ_S_HI 18
#define OZ_PART launch_slice_gemm_s18_18
#define OZ_PART_FUSED launch_slice_gemm_fused_s18_18
#define OZ_PART_TRAITS slice_gemm_traits_s18_18
#include "slice_gemm_launch.h"
