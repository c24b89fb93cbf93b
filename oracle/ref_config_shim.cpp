// C-ABI window onto the ONE translation unit of the reference that compiles in this image: /root/reference/src/config.cu
// (plain C++ once <cuda_runtime.h> is force-included the way nvcc does it; the CUDA headers are the real ones that ship inside
// the triton wheel).  TEST INFRASTRUCTURE ONLY, built by `make -C oracle ref` into oracle/_ref/libref_config.so, build container
// only (/root/reference does not exist on the GPU box).  This file holds no reference code: it calls the reference's own
// functions from its compiled object and instantiates its header-only templates where they lie.
//
// What it exposes (= what tests/test_ref_pin.py pins the oracle and the product's tables against):
//   get_split_config(mode)            /root/reference/src/config.cu:4-100      S and the slice-pair list, IN ORDER      (§8 row A2)
//   gemm_mode_str                     /root/reference/src/config.cu:102-118
//   padded_ld / get_slice_ld / get_slice_num_elements<int8>   /root/reference/src/utils.hpp:30-72   plane geometry       (A5, A9)
//   the enum values of include/ozimmu/ozimmu.hpp:12-46        the order include/ozimmu_hip.h mirrors                      (§8 b)
// Not exposed: calculate_working_memory_size (src/config.cu:121-146) calls get_data_size_in_byte, defined in src/handle.cu:227,
// a file that needs the un-vendored cutf headers; providing that symbol here would be a stand-in, so the function is linked
// out (--gc-sections) and stays unpinned.
#include <cstdint>
#include <cstring>
#include <ozimmu/ozimmu.hpp>
#include "config.hpp"
#include "utils.hpp"

namespace oz = mtk::ozimmu;

extern "C" {

// number of slices (split types minus the "original" entry) and the ordered pair list of a compute mode;
// returns the pair count, or -1 when cap is too small
int ref_get_split_config(int mode, int* num_a_types, int* num_b_types, int* a_ids, int* b_ids, int* gemm_modes, int cap) {
  const auto cfg = oz::detail::get_split_config(static_cast<oz::compute_mode_t>(mode));
  *num_a_types = static_cast<int>(cfg.matrix_A_split_types.size());
  *num_b_types = static_cast<int>(cfg.matrix_B_split_types.size());
  const int p = static_cast<int>(cfg.gemm_pair_config_list.size());
  if (p > cap) return -1;
  for (int i = 0; i < p; i++) {
    a_ids[i] = cfg.gemm_pair_config_list[i].A_id;
    b_ids[i] = cfg.gemm_pair_config_list[i].B_id;
    gemm_modes[i] = static_cast<int>(cfg.gemm_pair_config_list[i].gemm_mode);
  }
  return p;
}

// data_t of entry `idx` of the A (which = 0) or B (which = 1) split-type vector, -1 out of range
int ref_split_type(int mode, int which, int idx) {
  const auto cfg = oz::detail::get_split_config(static_cast<oz::compute_mode_t>(mode));
  const auto& v = which ? cfg.matrix_B_split_types : cfg.matrix_A_split_types;
  return idx >= 0 && idx < static_cast<int>(v.size()) ? static_cast<int>(v[idx]) : -1;
}

int ref_gemm_mode_str(int gemm_mode, char* out, int cap) {
  const std::string s = oz::detail::gemm_mode_str(static_cast<oz::detail::gemm_t>(gemm_mode));
  if (static_cast<int>(s.size()) + 1 > cap) return -1;
  std::memcpy(out, s.c_str(), s.size() + 1);
  return static_cast<int>(s.size());
}

uint32_t ref_padded_ld_i8(uint32_t n) { return oz::padded_ld<std::int8_t>(n); }
uint32_t ref_slice_ld_i8(uint32_t m, uint32_t n, int op) { return oz::get_slice_ld<std::int8_t>(m, n, static_cast<oz::operation_t>(op)); }
uint32_t ref_slice_num_elements_i8(uint32_t m, uint32_t n, int op) {
  return oz::get_slice_num_elements<std::int8_t>(m, n, static_cast<oz::operation_t>(op));
}

// the reference's enumerators by name (compile-time: a renamed or reordered enumerator breaks this file or the test)
int ref_enum_value(const char* name) {
#define OZ_E(scope, id) \
  if (!std::strcmp(name, #id)) return static_cast<int>(scope::id)
  OZ_E(oz, op_n); OZ_E(oz, op_t);
  OZ_E(oz, sgemm); OZ_E(oz, dgemm);
  OZ_E(oz, fp64_int8_3); OZ_E(oz, fp64_int8_4); OZ_E(oz, fp64_int8_5); OZ_E(oz, fp64_int8_6); OZ_E(oz, fp64_int8_7);
  OZ_E(oz, fp64_int8_8); OZ_E(oz, fp64_int8_9); OZ_E(oz, fp64_int8_10); OZ_E(oz, fp64_int8_11); OZ_E(oz, fp64_int8_12);
  OZ_E(oz, fp64_int8_13); OZ_E(oz, fp64_int8_14); OZ_E(oz, fp64_int8_15); OZ_E(oz, fp64_int8_16); OZ_E(oz, fp64_int8_17);
  OZ_E(oz, fp64_int8_18); OZ_E(oz, fp64_int8_auto);
  OZ_E(oz, fp64); OZ_E(oz, fp32); OZ_E(oz, fp16); OZ_E(oz, int8); OZ_E(oz, original); OZ_E(oz, none);
  OZ_E(oz, malloc_sync); OZ_E(oz, malloc_async);
  OZ_E(oz, real); OZ_E(oz, complx);
  OZ_E(oz::detail, matrix_A); OZ_E(oz::detail, matrix_B); OZ_E(oz::detail, matrix_C);
  OZ_E(oz::detail, cublas_dgemm); OZ_E(oz::detail, cublas_sgemm); OZ_E(oz::detail, cublas_tf32); OZ_E(oz::detail, cublas_fp16);
  OZ_E(oz::detail, cublas_bf16); OZ_E(oz::detail, int8tc);
#undef OZ_E
  return -1000;
}

}  // extern "C"
