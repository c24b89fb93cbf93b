// slice-GEMM kernels and launch policy of fp64_int8_7 .. fp64_int8_10 (see slice_gemm_launch.h, slice_gemm.hip)
#define OZ_S_LO 7
#define OZ_S_HI 10
#define OZ_PART launch_slice_gemm_s7_10
#define OZ_PART_FUSED launch_slice_gemm_fused_s7_10
#define OZ_PART_TRAITS slice_gemm_traits_s7_10
#include "slice_gemm_launch.h"
