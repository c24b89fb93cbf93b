// slice-GEMM kernels and launch policy of fp64_int8_15 .. fp64_int8_18 (see slice_gemm_launch.h, slice_gemm.hip)
#define OZ_S_LO 15
#define OZ_S_HI 18
#define OZ_PART launch_slice_gemm_s15_18
#define OZ_PART_FUSED launch_slice_gemm_fused_s15_18
#include "slice_gemm_launch.h"
