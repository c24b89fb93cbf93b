// slice-GEMM kernels and launch policy of fp64_int8_3 .. fp64_int8_4 (see slice_gemm_launch.h, slice_gemm.hip: OZ_GEMM_PARTS)
#define OZ_S_LO 3
#define OZ_S_HI 4
#define OZ_PART launch_slice_gemm_s3_4
#define OZ_PART_FUSED launch_slice_gemm_fused_s3_4
#define OZ_PART_TRAITS slice_gemm_traits_s3_4
#include "slice_gemm_launch.h"
