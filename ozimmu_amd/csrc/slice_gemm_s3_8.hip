// slice-GEMM kernels and launch policy of fp64_int8_3 .. fp64_int8_8 (see slice_gemm_launch.h, slice_gemm.hip)
#define OZ_S_LO 3
#define OZ_S_HI 8
#define OZ_PART launch_slice_gemm_s3_8
#define OZ_PART_FUSED launch_slice_gemm_fused_s3_8
#include "slice_gemm_launch.h"
