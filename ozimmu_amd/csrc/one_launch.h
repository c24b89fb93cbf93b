// one_launch.h — split + slice GEMM of a small problem in ONE launch (slice_gemm_one_launch.hip): the interface api.cpp uses.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

#include "kernels.h"

namespace ozhip {

// READY words (one per 8-row strip of op(A) and op(B)) the launch needs in the handle's epoch-tagged word buffer
size_t one_launch_ready_words(size_t m, size_t n);

// gemm_int8<double> (/root/reference/src/gemm.cu:344-410) of one real product as ONE kernel: `a` as for launch_slice_gemm
// (a single pass over all of K that writes C), job[0] / job[1] the views of op(A) / op(B) as for launch_split_resident,
// `ready` the word buffer (zeroed when allocated, only ever written with epoch-tagged words), `tag` this call's epoch << 11.
// hipErrorNotSupported: the form does not apply (mode, K, more tiles than CUs, the policy prefers another kernel,
// OZIMMU_HIP_ONE_LAUNCH=0) - nothing was launched, the caller runs the two-launch form.
hipError_t launch_split_gemm_one(int S, const SliceGemmArgs &a, const SplitJob job[2], int L, uint32_t *ready, uint32_t tag,
                                 hipStream_t stream);

} // namespace ozhip
