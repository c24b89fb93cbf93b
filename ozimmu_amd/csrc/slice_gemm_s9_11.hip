// slice-GEMM kernels and launch policy of fp64_int8_9 .. fp64_int8_11 (see slice_gemm_launch.h, slice_gemm.hip)
#define OZ_S_LO 9
#define OZ_S_HI 11
#define OZ_PART launch_slice_gemm_s9_11
#define OZ_PART_FUSED launch_slice_gemm_fused_s9_11
#include "slice_gemm_launch.h"
