// slice-GEMM kernels and launch policy of fp64_int8_5 .. fp64_int8_6 (see slice_gemm_launch.h, slice_gemm.hip: OZ_GEMM_PARTS)
#define OZ_S_LO 5
#define OZ_S_HI 6
#define OZ_PART launch_slice_gemm_s5_6
#define OZ_PART_FUSED launch_slice_gemm_fused_s5_6
#define OZ_PART_TRAITS slice_gemm_traits_s5_6
#include "slice_gemm_launch.h"
