// slice-GEMM kernels and launch policy of fp64_int8_16 (see slice_gemm_launch.h, slice_gemm.hip: OZ_GEMM_PARTS)
#define OZ_S_LO 16
#define OZ_S_HI 16
#define OZ_PART launch_slice_gemm_s16_16
#define OZ_PART_FUSED launch_slice_gemm_fused_s16_16
#define OZ_PART_TRAITS slice_gemm_traits_s16_16
#include "slice_gemm_launch.h"
