// slice-GEMM kernels and launch policy of fp64_int8_12 .. fp64_int8_14 (see slice_gemm_launch.h, slice_gemm.hip)
#define OZ_S_LO 12
#define OZ_S_HI 14
#define OZ_PART launch_slice_gemm_s12_14
#define OZ_PART_FUSED launch_slice_gemm_fused_s12_14
#include "slice_gemm_launch.h"
