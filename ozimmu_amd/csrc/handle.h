// handle.h — library state shared by api.cpp (direct API) and interpose.cpp (LD_PRELOAD boundary).
// Mirrors /root/reference/src/handle.hpp:6-31 (struct mtk::ozimmu::handle).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ozimmu_hip.h"
#include "config.h"

namespace ozhip {
struct Tuner;
}

struct ozimmu_hip_handle {
  hipStream_t stream = nullptr;
  int device = 0; // the device this handle was created on: workspace, topology and every launch belong to it

  // grow-only device workspace (src/handle.hpp:12-13, src/handle.cu:63-93)
  void *working_memory_ptr = nullptr;
  size_t current_working_memory_size = 0;
  ozimmu_malloc_mode_t malloc_mode = OZIMMU_MALLOC_SYNC;

  // auto mode: 16 counters for S = 3..18 (the reference allocates 8: src/handle.hpp:22)
  unsigned long long *d_mantissa_loss_counter_ptr = nullptr;
  unsigned long long *h_mantissa_loss_pinned = nullptr; // the counters' landing place on the host (pinned: the blocking copy of src/split.cu:404-408 without a staging hop)

  // row exponent words of the split (kernels.h: SplitJobs::tag): a buffer that holds nothing else, written with a tag
  // that grows from call to call, so that it is zeroed once when it is allocated and never per call
  uint32_t *exp_words = nullptr;
  size_t exp_words_bytes = 0;
  uint32_t exp_epoch = 0;

  // fp64_int8_auto: the statistic pass computes the row maxima of exactly the operands the GEMM it selects will split;
  // it leaves them (tagged with exp_epoch) in exp_words and describes them here, and the GEMM that follows under the same
  // lock skips its own row-maximum pass when the description matches.  Cleared by every other use of exp_words.
  struct ExpReuse {
    bool valid = false;
    bool armed = false; // only the statistic pass of an fp64_int8_auto GEMM call publishes (the operands cannot change
                        // between it and the GEMM: both run under one lock inside one API call)
    const void *a = nullptr, *b = nullptr;
    size_t lda = 0, ldb = 0, m = 0, n = 0, k = 0;
    int op_a = 0, op_b = 0, parts = 0;
  } exp_reuse;

  // Once a call of this handle has been captured into a graph, the graph holds pointers into the workspace and the
  // exponent-word buffer of that moment: blocks that are outgrown later are kept until the handle is destroyed instead of
  // freed, so that a replay never touches released memory.
  bool seen_capture = false;
  std::vector<void *> retired_blocks;
  // A graph replays without the library seeing it.  On a handle that has seen more than one stream, the first EAGER call
  // after a capture therefore leaves the blocks the capture used (workspace, exponent words) to the graph for good and
  // continues on fresh ones: replays and eager calls never share memory, so they need no ordering (api.cpp: WorkspaceUse).
  bool capture_dirty = false;      // a captured call has used the current blocks
  int capture_retirements = 0;     // how often that happened ...
  size_t capture_retired_bytes = 0; // ... and what it keeps allocated until the handle is destroyed (bounded: api.cpp)
  double avg_mantissa_loss_threshold = 0; // src/handle.hpp:26

  // src/handle.hpp:28-30, read at creation (src/handle.cu:25-30)
  uint32_t intercept_threshold_m = 1024;
  uint32_t intercept_threshold_n = 1024;
  uint32_t intercept_threshold_k = 1024;

  // stage breakdown (src/handle.hpp:16: cutf time_breakdown profiler; labels src/gemm.cu:38-48, :393-407)
  bool profiling = false;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  double stage_total_ms[3] = {0, 0, 0};
  float stage_last_ms[3] = {0, 0, 0};
  unsigned long long stage_calls = 0;

  // the workspace is shared by every call on this handle: when the caller switches streams the new stream first
  // waits for the last user of the workspace (the reference would let the two streams race on it)
  hipEvent_t tail_ev = nullptr;
  hipStream_t tail_stream = nullptr;
  bool tail_valid = false;        // tail_ev was recorded at the end of the previous call
  bool tail_stream_known = false; // a previous call exists (tail_stream is its stream)
  bool multi_stream = false;      // this handle has seen calls on more than one stream: record an event per call
  hipStream_t first_stream = nullptr; // the first stream any call (eager or captured) arrived on ...
  bool first_stream_known = false;
  bool several_streams = false;   // ... and whether any later call arrived on another one

  // diagnostics: the kernel (kernel_policy.h: Pick, + 8 = k64 with B in registers) the last slice-GEMM launch of this handle
  // ran for its first / second diagonal pass (-1: none)
  int last_kernel[2] = {-1, -1};

  // measured kernel choice for the shapes this handle keeps calling (kernel_tuner.h); touched under mtx only
  ozhip::Tuner *tuner = nullptr;

  // private vendor BLAS handle for the `dgemm` mode (src/handle.hpp:8), created lazily
  void *rocblas_handle = nullptr;

  // The reference is not thread safe (one unguarded global handle, src/cublas.cu:58).  Every entry point that touches
  // the stream, the workspace or the private vendor handle holds this lock for the whole enqueue; recursive because
  // the entry points call each other (auto mode -> gemm, gemm -> reallocate_working_memory, ...).
  std::recursive_mutex mtx;
};

namespace ozhip {

// ---- env + logging (src/utils.hpp:77-115) ------------------------------------------------------------
inline std::string load_env_if_defined(const char *name, const char *default_v = "") {
  const char *e = counted_getenv(name);
  return e ? std::string(e) : std::string(default_v);
}
inline bool env_enabled(const char *name, bool default_v) {
  const char *e = counted_getenv(name);
  return (e != nullptr && std::string(e) != "0") || (e == nullptr && default_v);
}
inline void log_info(const std::string &s) { // ozIMMU_log
  if (env_enabled("OZIMMU_INFO", false)) {
    std::fprintf(stdout, "[ozIMMU LOG] %s\n", s.c_str());
    std::fflush(stdout);
  }
}
inline void log_error(const std::string &s) { // ozIMMU_error: on unless OZIMMU_ERROR=0
  if (env_enabled("OZIMMU_ERROR", true)) {
    std::fprintf(stdout, "[ozIMMU ERROR] %s\n", s.c_str());
    std::fflush(stdout);
  }
}

// original vendor symbol: dlsym(RTLD_NEXT) first (preload case, src/utils.hpp:117-141), then the
// already-loaded / default librocblas / libhipblas (direct-API case)
void *vendor_symbol(const char *name);

// mode helpers
int num_split_of_mode(ozimmu_compute_mode_t mode);
bool is_int8_mode(ozimmu_compute_mode_t mode);

} // namespace ozhip
