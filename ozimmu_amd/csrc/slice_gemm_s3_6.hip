// slice-GEMM kernels and launch policy of fp64_int8_3 .. fp64_int8_6 (see slice_gemm_launch.h, slice_gemm.hip)
#define OZ_S_LO 3
#define OZ_S_HI 6
#define OZ_PART launch_slice_gemm_s3_6
#define OZ_PART_FUSED launch_slice_gemm_fused_s3_6
#define OZ_PART_TRAITS slice_gemm_traits_s3_6
#include "slice_gemm_launch.h"
