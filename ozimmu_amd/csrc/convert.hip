// convert.hip — FP64 <-> FP32 matrix conversion for the `sgemm` compute mode.
//
// Replaces convert_dtype_kernel / convert_dtype of /root/reference/src/cublas_helper.cu:20-66: a column-major matrix
// (rows x cols, leading dimension ld) is converted element by element; complex matrices are handled as 2*rows real
// scalars per column (make_cuComplex(s.x, s.y), :9-18, is the same two conversions).  HBM-bound: 12 bytes per scalar.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace ozhip {

template <class D, class S>
__global__ __launch_bounds__(256) void convert_kernel(D *__restrict__ dst, size_t ldd, const S *__restrict__ src,
                                                      size_t lds, size_t rows, size_t cols) {
  // one workgroup row per column chunk: x runs along the contiguous dimension
  const size_t r0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  for (size_t c = blockIdx.y; c < cols; c += gridDim.y) {
    const S *s = src + c * lds;
    D *d = dst + c * ldd;
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (r0 + i < rows) d[r0 + i] = static_cast<D>(s[r0 + i]);
  }
}

template <class D, class S>
static hipError_t launch_convert(D *dst, size_t ldd, const S *src, size_t lds, size_t rows, size_t cols,
                                 hipStream_t stream) {
  if (rows == 0 || cols == 0) return hipSuccess;
  const dim3 grid((unsigned)((rows + 1023) / 1024), (unsigned)(cols < 65535 ? cols : 65535));
  hipLaunchKernelGGL((convert_kernel<D, S>), grid, dim3(256), 0, stream, dst, ldd, src, lds, rows, cols);
  return hipGetLastError();
}

hipError_t launch_convert_f64_to_f32(float *dst, size_t ldd, const double *src, size_t lds, size_t rows, size_t cols,
                                     hipStream_t stream) {
  return launch_convert(dst, ldd, src, lds, rows, cols, stream);
}
hipError_t launch_convert_f32_to_f64(double *dst, size_t ldd, const float *src, size_t lds, size_t rows, size_t cols,
                                     hipStream_t stream) {
  return launch_convert(dst, ldd, src, lds, rows, cols, stream);
}

} // namespace ozhip
