// slice-GEMM kernels and launch policy of fp64_int8_12 (see slice_gemm_launch.h, slice_gemm.hip: OZ_GEMM_PARTS)
#define OZ_S_LO 12
#define OZ_S_HI 12
#define OZ_PART launch_slice_gemm_s12_12
#define OZ_PART_FUSED launch_slice_gemm_fused_s12_12
#define OZ_PART_TRAITS slice_gemm_traits_s12_12
#include "slice_gemm_launch.h"
