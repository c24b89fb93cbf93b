// api.cpp — direct library API (C ABI of include/ozimmu_hip.h) and the Ozaki DGEMM host pipeline.
//
// Mirrors the reference's L3/L2 host layers (citations relative to /root/reference):
//   handle lifecycle + grow-only workspace   src/handle.cu:6-144
//   mtk::ozimmu::gemm dispatcher             src/gemm.cu:524-653
//   gemm_int8<double> orchestration          src/gemm.cu:344-410
//   auto_mode_select                         src/split.cu:454-518
// The device stages are in split.hip and slice_gemm.hip.  No exception crosses the C ABI and nothing
// here synchronises the device except where the reference's semantics need a host value (auto mode).
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rocblas/rocblas.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <new>
#include <string>

#include "config.h"
#include "handle.h"
#include "kernel_policy.h"
#include "kernel_tuner.h"
#include "kernels.h"
#include "layout.h"
#include "one_launch.h"
#include "tile_plan.h"
#include "topology.h"

using namespace ozhip;

// ---- small helpers -----------------------------------------------------------------------------------

namespace ozhip {

void *vendor_symbol(const char *name) {
  void *f = dlsym(RTLD_NEXT, name);
  if (f) return f;
  const bool hipblas = std::strncmp(name, "hipblas", 7) == 0;
  const char *lib = hipblas ? "libhipblas.so.3" : "librocblas.so.5";
  void *h = dlopen(lib, RTLD_NOW | RTLD_NOLOAD); // the copy the process already uses (e.g. PyTorch's)
  if (!h) h = dlopen(lib, RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    log_error(std::string("Failed to load ") + lib + ".");
    return nullptr;
  }
  f = dlsym(h, name);
  if (!f)
    log_error(std::string("Failed to load a function ") + name +
              " during selecting hijacking function. Default rule will be used.");
  return f;
}

static const char *const kModeNames[] = {
    "sgemm",        "dgemm",        "fp64_int8_3",  "fp64_int8_4",  "fp64_int8_5",
    "fp64_int8_6",  "fp64_int8_7",  "fp64_int8_8",  "fp64_int8_9",  "fp64_int8_10",
    "fp64_int8_11", "fp64_int8_12", "fp64_int8_13", "fp64_int8_14", "fp64_int8_15",
    "fp64_int8_16", "fp64_int8_17", "fp64_int8_18", "fp64_int8_auto"};

int num_split_of_mode(ozimmu_compute_mode_t mode) { // src/config.cu:28-80
  if (mode >= OZIMMU_FP64_INT8_3 && mode <= OZIMMU_FP64_INT8_18) return (int)mode - (int)OZIMMU_FP64_INT8_3 + 3;
  return 0;
}
bool is_int8_mode(ozimmu_compute_mode_t mode) { return num_split_of_mode(mode) != 0; }

static inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

// largest K whose per-diagonal INT32 sums cannot overflow: at most S products of magnitude <= (2^L-1)^2 per k.  (The
// reference bounds a single pair: k*2^(2L) <= 2^31, src/split.cu:520-536.)  A multiple of 64 (of 32 below 64): a pass then
// holds an EVEN number of k-blocks (the k64 tile function walks two per step), and every k <= max_k_per_pass has
// k_blocks(k) <= kb_per_pass - the planes pad odd block counts beyond 32 up to even (layout.h) - so "single pass" means
// the same thing to the workspace layout (needs_acc) and to the K loops below.
static size_t max_k_per_pass(int S, int L) {
  if (L <= 0 || S <= 0) return 32; // no safe slice width (k == 0 or k > 2^30): callers route such calls elsewhere
  const unsigned long long q = (1ull << L) - 1ull;
  const unsigned long long kc = 2147483647ull / ((unsigned long long)S * q * q);
  return (size_t)(kc >= 64ull ? kc / 64ull * 64ull : 32ull);
}
// k-blocks one launch may cover (even, or 1)
static uint32_t kb_per_pass(int S, int L) { return (uint32_t)(max_k_per_pass(S, L) / FRAG_K); }

struct Workspace {
  double *ea, *eb;
  int8_t *planes_a, *planes_b;
  double *acc;
  uint32_t *phase;
  size_t total;
};

static Workspace carve(void *base, size_t m, size_t n, size_t k, int S, bool need_acc) {
  Workspace w{};
  size_t off = 0;
  auto take = [&](size_t bytes) {
    void *p = base ? (void *)((char *)base + off) : nullptr;
    off += align256(bytes);
    return p;
  };
  w.phase = (uint32_t *)take(PHASE_LINES_BYTES); // one 256-byte line per XCD (kernels.h)
  w.ea = (double *)take(8 * m);
  w.eb = (double *)take(8 * n);
  w.planes_a = (int8_t *)take(tiled_plane_bytes(m, k, S));
  w.planes_b = (int8_t *)take(tiled_plane_bytes(n, k, S));
  w.acc = need_acc ? (double *)take(8 * m * n) : nullptr;
  w.total = off;
  return w;
}

// lead throttle of the fused kernel (slice_gemm_kernel.h): only long-K, many-tile problems drift enough to gain
static uint32_t throttle_for(size_t m, size_t n, size_t k) {
  if (config().no_throttle) return 0u;
  return (k >= 6144 && ((m + 63) / 64) * ((n + 63) / 64) >= 2048) ? 1u : 0u;
}

// slice width for a K of this length; 0 when none is safe (k == 0, k > 2^30: src/split.cu:520-536 wraps there)
static int bits_for_k(size_t k) {
  return k > (size_t)1 << 30 ? 0 : (int)ozimmu_hip_get_bits_per_int8((uint32_t)k);
}

static bool needs_acc(size_t k, int S) {
  const int L = bits_for_k(k);
  if (L == 0) return false;
  return S > SINGLE_PASS_MAX_S || k_blocks(k) > (size_t)kb_per_pass(S, L);
}

static OperandView view_A(ozimmu_operation_t op, size_t m, size_t k, const double *a, size_t lda) {
  // src/split.cu:254: col_major = (op == op_n): element (row, kk) at a[kk*lda + row]
  return op == OZIMMU_OP_N ? OperandView{a, m, k, 1, lda} : OperandView{a, m, k, lda, 1};
}
static OperandView view_B(ozimmu_operation_t op, size_t k, size_t n, const double *b, size_t ldb) {
  // src/split.cu:277-281: op flipped, m <-> n swapped: rows = n
  return op == OZIMMU_OP_N ? OperandView{b, n, k, ldb, 1} : OperandView{b, n, k, 1, ldb};
}

static bool hip_ok(hipError_t e, const char *what) {
  if (e == hipSuccess) return true;
  log_error(std::string("HIP failure in ") + what + ": " + hipGetErrorString(e));
  // The failure is reported through this library's own status (3: the interposer lets the vendor routine run); it must not
  // stay behind as the runtime's "last error" for the application to trip over (PyTorch checks hipGetLastError after its own
  // launches: a failed workspace hipMalloc here made the NEXT torch call raise "out of memory" although the vendor GEMM had
  // run and nothing was wrong - tests/test_gpu_robustness.py, round 6)
  (void)hipGetLastError();
  return false;
}

// Test hook (tests/test_gpu_robustness.py): OZIMMU_HIP_TEST_FAIL_LAUNCH=n makes the n-th slice-GEMM launch of a call fail
// the way a rejected launch does, so that the error paths (C untouched -> vendor fallback; C already modified -> error
// status, no fallback) can be exercised without breaking the device.
static bool launch_gemm_checked(ozimmu_hip_handle_t h, int S, const SliceGemmArgs &g, hipStream_t stream, int &launch_index) {
  launch_index++;
#ifdef OZIMMU_HIP_TEST_HOOKS
  if (config().test_fail_launch == launch_index) {
    log_error("HIP failure in slice_gemm: injected by OZIMMU_HIP_TEST_FAIL_LAUNCH");
    return false;
  }
#endif
  note_pick(0, -1);
  note_pick(1, -1);
  const bool ok = hip_ok(launch_slice_gemm(S, g, stream), "slice_gemm");
  h->last_kernel[0] = last_pick(0); // diagnostics (ozimmu_hip_last_kernel): what the policy launched for this handle
  h->last_kernel[1] = last_pick(1);
  return ok;
}

// src/utils.hpp:143-168
static int check_gemm_shape(ozimmu_operation_t op, size_t m, size_t n, size_t ld, const char *mat) {
  if ((op == OZIMMU_OP_N ? m : n) > ld) {
    log_error(std::string("The leading dimension of ") + mat + " (" + std::to_string(ld) +
              ") must be larger or equal to the number of " + (op == OZIMMU_OP_N ? "rows" : "cols") + " (" +
              std::to_string(op == OZIMMU_OP_N ? m : n) + ")");
    return 1;
  }
  return 0;
}
static int check_address_alignment(const void *p, size_t elem, const char *mat) {
  if (reinterpret_cast<uintptr_t>(p) % elem) {
    log_error(std::string("Invalid address alignment for matrix ") + mat);
    return 1;
  }
  return 0;
}

// split of one operand into the workspace; stream ordered
// Two streaming passes (row maxima, then cut).  OZIMMU_HIP_SPLIT_BAND_BYTES > 0 walks the operand in row bands of that
// size, so that the cut pass of a band re-reads what its row-max pass just brought into the Infinity Cache.  Measured
// with 64 MiB bands it LOSES (8192^3: 16.90 vs 16.69 ms per GEMM, 6144^3: 7.34 vs 7.22: more, smaller launches cost more
// than the cache hits return), so the default is one band; the knob stays for experiments and the parity tests.
static bool run_split(ozimmu_hip_handle_t h, const OperandView &v, uint32_t *exps, int S, int L,
                      int8_t *planes, double *max_exp, const Batch &batch = Batch(), bool have_row_max = false,
                      uint32_t *zero_ptr = nullptr, uint32_t zero_words = 0) {
  const size_t band_bytes = config().split_band_bytes;
  const size_t row_bytes = 8 * std::max<size_t>(v.K, 1) * std::max<uint32_t>(batch.count, 1);
  size_t band_rows = band_bytes ? std::max<size_t>(TILE_ROWS, band_bytes / row_bytes / TILE_ROWS * TILE_ROWS) : v.rows;
  if (band_rows >= v.rows || v.rows * row_bytes <= 2 * band_bytes) band_rows = v.rows;
  const size_t KB = k_blocks(v.K);
  for (size_t r0 = 0; r0 < v.rows; r0 += band_rows) {
    OperandView b = v;
    b.in = v.in + r0 * v.stride_r;
    b.rows = std::min(band_rows, v.rows - r0);
    int8_t *pl = planes + (r0 / FRAG_ROWS) * KB * (size_t)S * FRAG_BYTES;
    if ((!have_row_max && !hip_ok(launch_row_max_exp(b, exps + r0, h->stream, batch, r0 == 0 ? zero_ptr : nullptr, zero_words),
                                  "row_max_exp")) ||
        !hip_ok(launch_cut(b, exps + r0, S, L, pl, max_exp + r0, h->stream, batch), "cut"))
      return false;
  }
  return true;
}

// One-pass split (split.hip: split_fused_kernel): a workgroup owns a 32-row strip for all of K, so HBM sees the operand
// once and a GEMM call needs 2 launches instead of 6 -- but a strip is walked by only 4 waves, and measured it loses to
// the two streaming passes at every size (1024^2: +29 us, 2048^2: +59 us per GEMM; tools/bench_kernel_choice.py), so it
// is OFF unless OZIMMU_HIP_SPLIT_ONE_PASS_BYTES raises the limit (parity tests run both forms).
static bool one_pass_split(size_t operand_bytes) {
  return operand_bytes <= config().split_one_pass_bytes;
}

// K <= 2048: ONE read of the operands, every strip of 8 / 16 / 32 rows held in the registers of one workgroup (split.hip:
// split_resident_kernel): 8 + S bytes per element instead of 16 + S, one split launch per call instead of two, no exponent
// words.  It pays where the split is latency- and launch-bound (1024^3 +7 % of the call, 1024 x 1024 x 2048 +3.5 %, 2048^3
// +2.5 %, 4096 x 4096 x 1024 +1 %) and is neutral to slightly worse beyond (3072^3 ... 4096^3 -0.3 %; the 8-row strips that
// K = 4096 needs -1.8 %): by policy up to K = 2048; OZIMMU_HIP_SPLIT_RESIDENT=8 / 16 / 32 forces it up to K = 4096 / 2048 /
// 1024, =0 keeps the two-pass forms (the parity tests run every form).
static bool resident_split(size_t k) {
  const int f = config().split_resident;
  if (f == 0 || config().split_one_pass_bytes != 0) return false;
  return k <= (f > 0 ? resident_split_max_k() : (size_t)2048);
}

// Problems whose operands total at most this many bytes split all their operand views with ONE launch per pass
// (row_max_kernel / cut_multi_kernel): they are bounded by launch gaps.  Larger ones keep one launch per view (each
// layout at its own occupancy) and walk row bands.  OZIMMU_HIP_SPLIT_MULTI_BYTES overrides (0: never).
static bool multi_view_split(size_t operand_bytes) {
  return operand_bytes <= config().split_multi_bytes;
}

// a strided batch in BLAS terms: matrix i of an operand starts stride * i ELEMENTS after matrix 0
struct BatchSpec {
  size_t count = 1;
  long long stride_a = 0, stride_b = 0, stride_c = 0;
};

static bool stream_is_capturing(hipStream_t stream);

// Row exponent words of one call (kernels.h: SplitJobs::tag): `parts` views of A (m rows) and of B (n rows) per matrix,
// `count` matrices, in the handle's dedicated buffer.  The buffer only ever holds words tagged with the epoch of the call
// that wrote them and the epoch grows with every call, so nothing is zeroed per call: once when the buffer is (re)allocated
// and once every 2^21 calls when the epoch wraps.
struct ExpWords {
  uint32_t *a[2] = {nullptr, nullptr}, *b[2] = {nullptr, nullptr};
  size_t pitch = 0; // bytes per matrix
  uint32_t tag = 0;
};
// pointers into the buffer for the CURRENT epoch (no new epoch): the layout exp_words() hands out
static void exp_words_layout(ozimmu_hip_handle_t h, size_t m, size_t n, int parts, ExpWords &x) {
  const size_t ea = align256(4 * m), eb = align256(4 * n);
  x.pitch = (size_t)parts * (ea + eb);
  x.tag = h->exp_epoch << 11;
  char *base = reinterpret_cast<char *>(h->exp_words);
  for (int i = 0; i < parts; i++) {
    x.a[i] = reinterpret_cast<uint32_t *>(base + (size_t)i * ea);
    x.b[i] = reinterpret_cast<uint32_t *>(base + (size_t)parts * ea + (size_t)i * eb);
  }
}
// fp64_int8_auto (handle.h: ExpReuse): do the words of the current epoch already hold the row maxima of these operands?
static bool exp_words_reusable(ozimmu_hip_handle_t h, ozimmu_operation_t op_A, ozimmu_operation_t op_B, size_t m, size_t n,
                               size_t k, const void *a, size_t lda, const void *b, size_t ldb, int parts, size_t count) {
  const auto &r = h->exp_reuse;
  return r.valid && count == 1 && r.a == a && r.b == b && r.lda == lda && r.ldb == ldb && r.m == m && r.n == n && r.k == k &&
         r.op_a == (int)op_A && r.op_b == (int)op_B && r.parts == parts && !config().no_exp_reuse;
}
static bool exp_words(ozimmu_hip_handle_t h, size_t m, size_t n, int parts, size_t count, ExpWords &x) {
  h->exp_reuse.valid = false; // a new epoch: whatever an earlier statistic pass left is history
  const size_t ea = align256(4 * m), eb = align256(4 * n);
  x.pitch = (size_t)parts * (ea + eb);
  const size_t bytes = x.pitch * std::max<size_t>(count, 1);
  if (bytes > h->exp_words_bytes) {
    if (stream_is_capturing(h->stream)) return false; // allocation is illegal while the stream is captured into a graph
    // OZIMMU_MALLOC_ASYNC: release, allocation and the one-time zero fill are stream ordered like the workspace's
    // (ozimmu_hip_reallocate_working_memory): growth never synchronises the device; the default mode keeps
    // the reference's synchronising hipFree / hipMalloc (src/handle.cu:71-75)
    const bool async = h->malloc_mode == OZIMMU_MALLOC_ASYNC;
    if (h->exp_words && h->seen_capture)
      h->retired_blocks.push_back(h->exp_words); // a captured graph may still point into it (handle.h)
    else if (h->exp_words && async)
      hipFreeAsync(h->exp_words, h->stream); // earlier calls on other streams are ordered in front of this one (WorkspaceUse)
    else if (h->exp_words)
      hipFree(h->exp_words); // device-synchronising: earlier calls are done with it
    h->exp_words = nullptr;
    h->exp_words_bytes = 0;
    const size_t cap = std::max<size_t>(bytes, (size_t)1 << 20);
    // The zero fill is ordered ON THE CALLER'S STREAM in both modes: a plain hipMemset of device memory runs on the null
    // stream, asynchronously to the host, and a non-blocking stream (PyTorch's side streams) does not wait for it - the
    // row-maximum kernel of this very call could have its words wiped (seen as 1-ulp differences in 1 run of 6 of
    // tests/test_gpu_robustness.py::test_epoch_wrap_around_with_a_captured_graph_in_flight).
    const bool ok = (async ? hip_ok(hipMallocAsync((void **)&h->exp_words, cap, h->stream), "exponent words")
                           : hip_ok(hipMalloc((void **)&h->exp_words, cap), "exponent words")) &&
                    hip_ok(hipMemsetAsync(h->exp_words, 0, cap, h->stream), "memset");
    if (!ok) {
      if (h->exp_words) hipFree(h->exp_words);
      h->exp_words = nullptr;
      return false;
    }
    h->exp_words_bytes = cap;
    h->exp_epoch = 0;
  }
#ifdef OZIMMU_HIP_TEST_HOOKS
  if (const uint32_t e = config().test_exp_epoch) // test hook: jump close to the wrap-around
    if (h->exp_epoch < e) h->exp_epoch = e;
#endif
  if (++h->exp_epoch >= (1u << 21)) { // the tag field is 21 bits: start over on zeroed words (stream ordered)
    if (h->seen_capture) {
      // A graph captured earlier replays with its old (large) tag and leaves such words behind; an eager call of the new,
      // small epochs would lose its atomicMax against them and read the row as empty.  Graphs keep the old buffer
      // (retired, freed with the handle), the new epochs get a fresh zeroed one.
      if (stream_is_capturing(h->stream)) return false; // allocation is illegal inside a capture: vendor fallback
      uint32_t *fresh = nullptr;
      if (!hip_ok(hipMalloc((void **)&fresh, h->exp_words_bytes), "exponent words") ||
          !hip_ok(hipMemsetAsync(fresh, 0, h->exp_words_bytes, h->stream), "memset")) { // stream ordered: see above
        if (fresh) hipFree(fresh);
        --h->exp_epoch;
        return false;
      }
      h->retired_blocks.push_back(h->exp_words);
      h->exp_words = fresh;
    } else if (!hip_ok(hipMemsetAsync(h->exp_words, 0, h->exp_words_bytes, h->stream), "memset")) {
      return false;
    }
    h->exp_epoch = 1;
  }
  x.tag = h->exp_epoch << 11;
  char *base = reinterpret_cast<char *>(h->exp_words);
  // A call that is being captured into a graph will be replayed later with THIS tag, after other calls have left words of
  // later epochs (or an earlier replay words of the same one) in the buffer: the graph zeroes its words itself.
  if (stream_is_capturing(h->stream)) {
    h->seen_capture = true;
    if (!hip_ok(launch_zero_words(base, bytes, 0, 1, h->stream), "zero_words")) return false;
  }
  for (int i = 0; i < parts; i++) {
    x.a[i] = reinterpret_cast<uint32_t *>(base + (size_t)i * ea);
    x.b[i] = reinterpret_cast<uint32_t *>(base + (size_t)parts * ea + (size_t)i * eb);
  }
  return true;
}

// the phase hints / claim counters of a call: one 256-byte line per XCD that the slice GEMM expects zeroed
static bool zero_phase_lines(ozimmu_hip_handle_t h, uint32_t *phase) {
  return hip_ok(launch_zero_words(phase, (size_t)topology(h->device).xcds * PHASE_LINE_WORDS * 4, 0, 1, h->stream), "zero_words");
}

// The per-XCD phase hints and the tile queues of the persistent wide kernel only pay for problems with more tiles than
// CUs; below that the launch is one tile per workgroup (K-split kernel, or a single round of the classic / wide kernel)
// and the call saves the zeroing launch: three launches per small DGEMM (row maxima, cut, GEMM).
// Nor for a short K: with fewer than 32 k-steps per tile the claim of a tile and the read of the hint cost more than
// stealing and phase alignment return (8192^2 x 128..512: 3-6 % slower with them, tools/ab_short_k.py).
// (OZIMMU_HIP_WIDE_GRID = g, tests: g persistent workgroups walk the tiles of ANY single product - claims, stealing and the
// ticket drawn one tile ahead on shapes the oracle checks in a second)
static bool wants_phase(size_t m, size_t n, size_t k, size_t batch) {
  if (batch != 1 || config().no_phase_hint) return false;
  return (m * n >= (size_t)2560 * 1024 && k >= 1024) || config().wide_grid > 0;
}

// WorkspaceUse below: hand the blocks a capture has used over to the graph(s) - nothing is freed, the handle forgets them
static bool leave_blocks_to_graphs(ozimmu_hip_handle_t h) {
  const size_t bytes = h->current_working_memory_size + h->exp_words_bytes;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  if (h->capture_retirements >= 8 || h->capture_retired_bytes + bytes > total_b / 8 || free_b < 2 * bytes) return false;
  if (h->working_memory_ptr) h->retired_blocks.push_back(h->working_memory_ptr);
  if (h->exp_words) h->retired_blocks.push_back(h->exp_words);
  h->working_memory_ptr = nullptr;
  h->current_working_memory_size = 0;
  h->exp_words = nullptr;
  h->exp_words_bytes = 0;
  h->exp_epoch = 0;
  h->exp_reuse.valid = false;
  h->capture_retirements++;
  h->capture_retired_bytes += bytes;
  return true;
}

// Scope of one use of the handle's workspace on h->stream (construct before ensure_workspace, under h->mtx).
// Calls on one stream are ordered by the stream.  As long as a handle has only ever seen ONE stream (the common case: a
// BLAS handle bound to a stream, PyTorch's current stream) nothing else is done: an event recorded after every call
// costs each call ~6 us of dispatch gap in front of its first kernel (rocprofv3 kernel trace, 1024^3), a tenth of a
// small GEMM.  The first time a call arrives on a different stream than its predecessor the device is synchronised
// once (no event of the earlier call exists, and its stream may be gone by now: hipEventRecord on a destroyed stream
// crashes inside the runtime, tests/test_gpu_robustness.py), and from then on every call records an event on its own
// stream when it ends and a call on another stream waits for it.
struct WorkspaceUse {
  ozimmu_hip_handle_t h;
  // false: the call cannot be ordered behind the previous user of the workspace -> the caller returns 3 ("failed, C
  // untouched"; the interposer then lets the vendor routine run / be captured)
  bool ok = true;
  bool capturing = false;
  int restore_device = -1; // the caller's current device, when it is not the handle's
  explicit WorkspaceUse(ozimmu_hip_handle_t handle) : h(handle) {
    // the workspace, the stream, the topology the launch policy plans with and every launch belong to the handle's device:
    // make it current for the duration of the call if the caller has another one selected
    int cur = h->device;
    if (hipGetDevice(&cur) == hipSuccess && cur != h->device && hipSetDevice(h->device) == hipSuccess) restore_device = cur;
    // A stream that is being captured into a graph executes nothing now: no synchronisation is legal on it (a
    // hipDeviceSynchronize under global capture mode invalidates the user's capture), and an event recorded on it becomes a
    // graph node that an eager hipStreamWaitEvent cannot wait for.  Captured calls therefore never touch the tail state.
    // A call captured on a stream OTHER than its predecessor's has no ordering against that predecessor's eager work:
    // refuse it (the common flow - eager matmuls on the default stream, then torch.cuda.graph on a side stream - then
    // captures the vendor GEMM).  A graph REPLAY is ordered against later eager calls by a device synchronisation in front
    // of them once the handle has seen a capture and more than one stream (below); two graphs replaying CONCURRENTLY on
    // different streams share the workspace they were captured with and remain the caller's business.
    capturing = stream_is_capturing(h->stream);
    if (!h->first_stream_known) {
      h->first_stream = h->stream;
      h->first_stream_known = true;
    } else if (h->stream != h->first_stream) {
      h->several_streams = true;
    }
    const bool other_stream = h->tail_stream_known && h->tail_stream != h->stream && !config().test_no_stream_order;
    if (capturing) {
      ok = !other_stream;
      if (ok) h->capture_dirty = true; // the graph being captured will point into the current blocks
      return;
    }
    bool synced = false;
    if (h->seen_capture && h->several_streams && h->capture_dirty) {
      // A graph captured from this handle replays without the library seeing it: a replay records no tail event, and it may
      // run on any stream.  Once a handle has been captured AND has seen more than one stream (capture streams included), an
      // eager call has no way to order itself against a replay in flight - so it does not share memory with one: the first
      // eager call after a capture leaves the workspace and the exponent words the capture used to the graph (kept until the
      // handle is destroyed, like every block a graph may point into) and runs, as every later eager call, on fresh blocks.
      // No synchronisation.  (Rounds 3-5 synchronised the DEVICE in front of every eager call from then on, for ever: ADVICE
      // r4)  What a pathological caller - capture, eager, capture, eager ... - can pin this way is bounded
      // (8 hand-overs, an eighth of the device's memory); beyond that the old behaviour: a device synchronisation per call.
      // (An application that keeps its graph replays and its eager GEMMs on ONE stream never gets here.)
      if (leave_blocks_to_graphs(h)) {
        h->capture_dirty = false;
      } else {
        hipDeviceSynchronize();
        (void)hipGetLastError();
        h->multi_stream = true;
        synced = true;
      }
    }
    if (synced) {
    } else if (other_stream) {
      if (h->multi_stream && h->tail_valid) {
        hipStreamWaitEvent(h->stream, h->tail_ev, 0);
      } else {
        hipDeviceSynchronize();
        (void)hipGetLastError();
        h->multi_stream = true;
      }
    }
  }
  ~WorkspaceUse() {
    if (restore_device >= 0) hipSetDevice(restore_device);
    if (capturing) return; // nothing ran; the tail stays what the last eager call left
    h->tail_valid = h->multi_stream && h->tail_ev && hipEventRecord(h->tail_ev, h->stream) == hipSuccess;
    if (h->multi_stream && !h->tail_valid) (void)hipGetLastError();
    h->tail_stream = h->stream;
    h->tail_stream_known = true;
  }
};

bool stream_is_capturing(hipStream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &st) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return st != hipStreamCaptureStatusNone;
}

static bool ensure_workspace(ozimmu_hip_handle_t h, size_t bytes) {
  // growing the workspace (hipFree / hipMalloc) is illegal while the caller's stream is being captured into a graph and
  // would invalidate the capture: report "failed, C untouched" instead (the interposer then captures the vendor GEMM);
  // once the workspace is large enough, every launch of the call is an ordinary capturable kernel
  // (in every malloc mode: hipMallocAsync on a capturing stream becomes a graph-owned allocation node, and its pointer
  // would outlive the graph execution in h->working_memory_ptr)
  if (bytes > h->current_working_memory_size && stream_is_capturing(h->stream)) return false;
  ozimmu_hip_reallocate_working_memory(h, bytes);
  return h->working_memory_ptr != nullptr && h->current_working_memory_size >= bytes;
}

// gemm_int8<double> (src/gemm.cu:344-410), fused MI355X form.  dump != nullptr: test hook.
// `bs`: the matrices [0, bs.count) of a strided batch run through ONE set of launches (batch index = a grid dimension,
// one workspace slot per matrix); the caller keeps count * slot within its workspace budget.
static int gemm_int8_real(ozimmu_hip_handle_t h, ozimmu_operation_t op_A, ozimmu_operation_t op_B, size_t m,
                          size_t n, size_t k, double alpha, const double *a, size_t lda, const double *b,
                          size_t ldb, double beta, double *c, size_t ldc, int S, int32_t *dump,
                          const BatchSpec &bs = BatchSpec()) {
  const int L = bits_for_k(k); // src/gemm.cu:357
  if (L == 0) return 3;        // callers route k == 0 / k > 2^30 to the vendor GEMM
  const size_t kc = max_k_per_pass(S, L);
  const bool acc_needed = needs_acc(k, S);
  if (dump && (k > kc || bs.count != 1)) return 2; // whole-K INT32 sums would not be exact
  Workspace sz = carve(nullptr, m, n, k, S, acc_needed);
  const size_t slot = sz.total; // 256-byte aligned
  WorkspaceUse use(h);
  if (!use.ok || !ensure_workspace(h, slot * bs.count)) return 3;
  Workspace w = carve(h->working_memory_ptr, m, n, k, S, acc_needed);
  Batch ba, bb;
  ba.count = bb.count = (uint32_t)bs.count;
  ba.in_stride = bs.stride_a;
  bb.in_stride = bs.stride_b;
  ba.ws_stride = bb.ws_stride = slot;
  ExpWords xw;
  // fp64_int8_auto: the statistic pass that selected this mode has just computed these row maxima
  const bool have_row_max = exp_words_reusable(h, op_A, op_B, m, n, k, a, lda, b, ldb, 1, bs.count);
  if (have_row_max) {
    exp_words_layout(h, m, n, 1, xw);
    h->exp_reuse.valid = false;
  } else if (!exp_words(h, m, n, 1, bs.count, xw)) {
    return 3;
  }
  ba.exps_stride = bb.exps_stride = xw.pitch;
  ba.tag = bb.tag = xw.tag;

  const bool prof = h->profiling && !stream_is_capturing(h->stream); // the stage timer synchronises on its events
  // measured kernel choice for shapes the handle sees again (kernel_tuner.h): plain single-pass real GEMMs only
  struct TunedCall {
    TuneTicket tk;
    bool ok = false;
    Tuner *t = nullptr;
    ~TunedCall() { tuner_end(t, tk, ok); }
  } tuned;
  // (never on a handle that has been captured into a graph: the tuner's hipEventQuery / hipEventRecord next to a capture that
  // may be in flight on another thread are not worth finding out about - ADVICE r5)
  bool tune = bs.count == 1 && !dump && !prof && !have_row_max && k <= kc && S <= SINGLE_PASS_MAX_S && !h->seen_capture &&
              !stream_is_capturing(h->stream);
#ifdef OZIMMU_HIP_TEST_HOOKS
  tune = tune && config().test_fail_launch == 0;
#endif
  if (tune) {
    TuneShape shape;
    shape.S = S;
    shape.op_a = (int)op_A;
    shape.op_b = (int)op_B;
    shape.beta_nonzero = beta != 0.0;
    shape.m = m;
    shape.n = n;
    shape.k = k;
    tuned.tk = tuner_begin(h->tuner, h->device, shape, (unsigned)k_blocks(k), h->stream);
    tuned.t = h->tuner;
  }
  if (prof && !hip_ok(hipEventRecord(h->ev[0], h->stream), "event")) return 3;
  const bool use_phase = wants_phase(m, n, k, bs.count);
  // the phase hints / claim counters are cleared by the first row-maximum launch on the side (kernels.h: SplitJobs::zero_ptr);
  // a call without that launch (one-pass split, row maxima left by the auto-mode statistic) clears them with a launch of its own
  const bool resident = !have_row_max && resident_split(k);
  const bool zero_in_row_max = use_phase && !have_row_max && (resident || !one_pass_split(8 * (m + n) * k * bs.count));
  uint32_t *const zp = zero_in_row_max ? w.phase : nullptr;
  const uint32_t zw = (uint32_t)topology(h->device).xcds * PHASE_LINE_WORDS;
  if (use_phase && !zero_in_row_max && !zero_phase_lines(h, w.phase)) return 3;
  SliceGemmArgs g{};
  g.device = h->device;
  g.a_planes = w.planes_a;
  g.b_planes = w.planes_b;
  g.KB = (uint32_t)k_blocks(k);
  g.M = (uint32_t)m;
  g.N = (uint32_t)n;
  g.tiles_m = (uint32_t)((m + TILE_ROWS - 1) / TILE_ROWS);
  g.tiles_n = (uint32_t)((n + TILE_ROWS - 1) / TILE_ROWS);
  g.L = L;
  g.ea = w.ea;
  g.eb = w.eb;
  g.alpha = alpha;
  g.beta = beta;
  g.c = c;
  g.ldc = ldc;
  g.acc = w.acc;
  // the phase hint coordinates the workgroups of ONE product: a batch runs without it
  g.phase = use_phase ? w.phase : nullptr;
  g.throttle = bs.count > 1 ? 0u : throttle_for(m, n, k);
  g.batch = (uint32_t)bs.count;
  g.ws_stride = slot;
  g.c_stride = bs.stride_c;
  g.dump = dump;
  g.dump_only = dump ? 1 : 0;

  // Small problems (at most one 64 x 64 tile per CU, K <= 2048, S <= 9): split AND slice GEMM as ONE kernel
  // (slice_gemm_one_launch.hip); its per-strip ready words are epoch-tagged words of this call in the exponent-word buffer,
  // which the resident split leaves unused.  Not under capture (a replay would meet its own old tag), not with the stage
  // timer (it brackets the stages), not with the launch-failure hook (it counts slice-GEMM launches).
  // (a forced strip height of 16 / 32 rows asks for the resident split kernel proper: this form cuts 8-row strips)
  bool one_launch_ok = resident && (config().split_resident < 0 || config().split_resident == 8) && bs.count == 1 && !prof && g.KB <= kb_per_pass(S, L) && !stream_is_capturing(h->stream) &&
                       4 * one_launch_ready_words(m, n) <= h->exp_words_bytes;
#ifdef OZIMMU_HIP_TEST_HOOKS
  one_launch_ok = one_launch_ok && config().test_fail_launch == 0;
#endif
  if (one_launch_ok) {
    const SplitJob jobs[2] = {{view_A(op_A, m, k, a, lda), w.planes_a, w.ea, 0, nullptr},
                              {view_B(op_B, k, n, b, ldb), w.planes_b, w.eb, 0, nullptr}};
    g.kb0 = 0;
    g.kb1 = g.KB;
    g.acc_in = 0;
    g.final = 1;
    g.qslot = 0;
    note_pick(0, -1);
    note_pick(1, -1);
    const hipError_t e = launch_split_gemm_one(S, g, jobs, L, h->exp_words, xw.tag, h->stream);
    if (e == hipSuccess) {
      h->last_kernel[0] = last_pick(0);
      h->last_kernel[1] = -1;
      tuned.ok = true;
      return 0;
    }
    if (e != hipErrorNotSupported) {
      hip_ok(e, "split_gemm_one");
      return 3; // nothing ran: C untouched
    }
  }
  if (resident) {
    // one read, both operands (and every matrix of the batch) in one launch
    const SplitJob jobs[2] = {{view_A(op_A, m, k, a, lda), w.planes_a, w.ea, bs.stride_a, nullptr},
                              {view_B(op_B, k, n, b, ldb), w.planes_b, w.eb, bs.stride_b, nullptr}};
    if (!hip_ok(launch_split_resident(jobs, 2, S, L, h->stream, (uint32_t)bs.count, slot, zp, zw, topology(h->device).cus), "split"))
      return 3;
    if (prof && !hip_ok(hipEventRecord(h->ev[1], h->stream), "event")) return 3; // split_A + split_B -> split_A
  } else if (one_pass_split(8 * (m + n) * k * bs.count)) {
    // both operands (and every matrix of the batch) in one launch
    const SplitJob jobs[2] = {{view_A(op_A, m, k, a, lda), w.planes_a, w.ea, bs.stride_a, nullptr},
                              {view_B(op_B, k, n, b, ldb), w.planes_b, w.eb, bs.stride_b, nullptr}};
    if (!hip_ok(launch_split_fused(jobs, 2, S, L, h->stream, (uint32_t)bs.count, slot), "split")) return 3;
    if (prof && !hip_ok(hipEventRecord(h->ev[1], h->stream), "event")) return 3; // split_A + split_B -> split_A
  } else if (multi_view_split(8 * (m + n) * k * bs.count)) {
    // A and B in one launch per pass: row maxima, cut, GEMM = 3 launches (in the stage report the row-max pass is
    // booked under split_A and the cut pass under split_B)
    const SplitJob jobs[2] = {{view_A(op_A, m, k, a, lda), w.planes_a, w.ea, bs.stride_a, xw.a[0]},
                              {view_B(op_B, k, n, b, ldb), w.planes_b, w.eb, bs.stride_b, xw.b[0]}};
    if (!have_row_max &&
        !hip_ok(launch_row_max_multi(jobs, 2, h->stream, (uint32_t)bs.count, slot, xw.pitch, xw.tag, zp, zw), "row_max_exp"))
      return 3;
    if (prof && !hip_ok(hipEventRecord(h->ev[1], h->stream), "event")) return 3;
    if (!hip_ok(launch_cut_multi(jobs, 2, S, L, h->stream, (uint32_t)bs.count, slot, xw.pitch, xw.tag), "cut")) return 3;
  } else {
    if (!run_split(h, view_A(op_A, m, k, a, lda), xw.a[0], S, L, w.planes_a, w.ea, ba, have_row_max, zp, zw)) return 3;
    if (prof && !hip_ok(hipEventRecord(h->ev[1], h->stream), "event")) return 3;
    if (!run_split(h, view_B(op_B, k, n, b, ldb), xw.b[0], S, L, w.planes_b, w.eb, bb, have_row_max)) return 3;
  }
  if (prof && !hip_ok(hipEventRecord(h->ev[2], h->stream), "event")) return 3;

  // (an even number of k-blocks per pass: the k64 tile function walks two per step)
  const uint32_t kbp = kb_per_pass(S, L);
  int launches = 0;
  for (uint32_t kb0 = 0; kb0 < g.KB; kb0 += kbp) {
    g.kb0 = kb0;
    g.kb1 = std::min(g.KB, kb0 + kbp);
    g.acc_in = kb0 != 0;
    g.final = g.kb1 == g.KB;
    g.qslot = 2u * (uint32_t)launches; // per-launch claim counters of the wide kernel (two per K chunk: S > 12)
    // C is written by the final launch(es) only (earlier passes go to the FP64 workspace); S > 12 splits the final
    // pass into two launches of which only the second writes C: a failed launch leaves C untouched
    if (!launch_gemm_checked(h, S, g, h->stream, launches)) return 3;
  }
  if (prof) {
    if (!hip_ok(hipEventRecord(h->ev[3], h->stream), "event")) return 3;
    if (!hip_ok(hipEventSynchronize(h->ev[3]), "event sync")) return 3; // reference: stop_timer_sync
    for (int i = 0; i < 3; i++) {
      float ms = 0;
      hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]);
      h->stage_last_ms[i] = ms;
      h->stage_total_ms[i] += ms;
    }
    h->stage_calls++;
  }
  tuned.ok = true;
  return 0;
}

// Re (part 0) or Im (part 1) of an interleaved complex operand, as a strided real view
static OperandView view_A_part(ozimmu_operation_t op, size_t m, size_t k, const double *a, size_t lda, int part) {
  return op == OZIMMU_OP_N ? OperandView{a + part, m, k, 2, 2 * lda} : OperandView{a + part, m, k, 2 * lda, 2};
}
static OperandView view_B_part(ozimmu_operation_t op, size_t k, size_t n, const double *b, size_t ldb, int part) {
  return op == OZIMMU_OP_N ? OperandView{b + part, n, k, 2 * ldb, 2} : OperandView{b + part, n, k, 2, 2 * ldb};
}

struct WorkspaceZ {
  uint32_t *phase;
  double *ea[2], *eb[2];
  int8_t *planes_a[2], *planes_b[2];
  double *acc;
  size_t total;
};

static WorkspaceZ carve_z(void *base, size_t m, size_t n, size_t k, int S, bool need_acc) {
  WorkspaceZ w{};
  size_t off = 0;
  auto take = [&](size_t bytes) {
    void *p = base ? (void *)((char *)base + off) : nullptr;
    off += align256(bytes);
    return p;
  };
  w.phase = (uint32_t *)take(PHASE_LINES_BYTES);
  for (int i = 0; i < 2; i++) w.ea[i] = (double *)take(8 * m);
  for (int i = 0; i < 2; i++) w.eb[i] = (double *)take(8 * n);
  for (int i = 0; i < 2; i++) w.planes_a[i] = (int8_t *)take(tiled_plane_bytes(m, k, S));
  for (int i = 0; i < 2; i++) w.planes_b[i] = (int8_t *)take(tiled_plane_bytes(n, k, S));
  w.acc = need_acc ? (double *)take(8 * m * n) : nullptr;
  w.total = off;
  return w;
}

// gemm_int8<cuDoubleComplex> (src/gemm.cu:412-521): Re and Im are split separately (own row exponents), C is
// scaled by beta first, then four real Ozaki products (Im,Im), (Re,Re), (Im,Re), (Re,Im) are added with the
// factors -alpha, alpha, i*alpha, i*alpha -- in that order (:479-518).  Each product runs the same fused kernel
// as the real path; its epilogue adds the scaled product into the complex C.
// OZIMMU_OP_C (conjugate transpose; the reference runs it as a plain transpose, src/cublas.cu:50-56, which is the wrong
// product): conj(X) = Re(X) - i Im(X), and the slice cut is odd in its argument (the sign is applied after the truncation,
// src/split.cu:159, :176-181) with row maxima that ignore the sign, so the slices of -Im(X) are the negated slices of Im(X):
// the operand is split exactly like op T and every product that contains its imaginary part changes the sign of its factor.
static int gemm_int8_complex(ozimmu_hip_handle_t h, ozimmu_operation_t op_A, ozimmu_operation_t op_B, size_t m,
                             size_t n, size_t k, const double *alpha, const double *a, size_t lda, const double *b,
                             size_t ldb, const double *beta, double *c, size_t ldc, int S,
                             const BatchSpec &bs = BatchSpec()) {
  const int L = bits_for_k(k);
  if (L == 0) return 3;
  const bool acc_needed = needs_acc(k, S);
  WorkspaceZ sz = carve_z(nullptr, m, n, k, S, acc_needed);
  const size_t slot = sz.total;
  WorkspaceUse use(h);
  if (!use.ok || !ensure_workspace(h, slot * bs.count)) return 3;
  WorkspaceZ w = carve_z(h->working_memory_ptr, m, n, k, S, acc_needed);
  Batch ba, bb; // strides of the real views: 2 doubles per complex element
  ba.count = bb.count = (uint32_t)bs.count;
  ba.in_stride = 2 * bs.stride_a;
  bb.in_stride = 2 * bs.stride_b;
  ba.ws_stride = bb.ws_stride = slot;
  ExpWords xw;
  const bool have_row_max = exp_words_reusable(h, op_A, op_B, m, n, k, a, lda, b, ldb, 2, bs.count);
  if (have_row_max) {
    exp_words_layout(h, m, n, 2, xw);
    h->exp_reuse.valid = false;
  } else if (!exp_words(h, m, n, 2, bs.count, xw)) {
    return 3;
  }
  ba.exps_stride = bb.exps_stride = xw.pitch;
  ba.tag = bb.tag = xw.tag;

  const bool prof = h->profiling && !stream_is_capturing(h->stream); // the stage timer synchronises on its events
  if (prof && !hip_ok(hipEventRecord(h->ev[0], h->stream), "event")) return 3;
  const bool use_phase = wants_phase(m, n, k, bs.count);
  const bool resident = !have_row_max && resident_split(k);
  if (use_phase && !resident && !zero_phase_lines(h, w.phase)) return 3;
  if (resident) {
    const SplitJob jobs[4] = {{view_A_part(op_A, m, k, a, lda, 0), w.planes_a[0], w.ea[0], ba.in_stride, nullptr},
                              {view_A_part(op_A, m, k, a, lda, 1), w.planes_a[1], w.ea[1], ba.in_stride, nullptr},
                              {view_B_part(op_B, k, n, b, ldb, 0), w.planes_b[0], w.eb[0], bb.in_stride, nullptr},
                              {view_B_part(op_B, k, n, b, ldb, 1), w.planes_b[1], w.eb[1], bb.in_stride, nullptr}};
    if (!hip_ok(launch_split_resident(jobs, 4, S, L, h->stream, (uint32_t)bs.count, slot, use_phase ? w.phase : nullptr,
                                      (uint32_t)topology(h->device).xcds * PHASE_LINE_WORDS, topology(h->device).cus), "split"))
      return 3;
    if (prof && !hip_ok(hipEventRecord(h->ev[1], h->stream), "event")) return 3;
  } else if (one_pass_split(16 * (m + n) * k * bs.count)) {
    const SplitJob jobs[4] = {{view_A_part(op_A, m, k, a, lda, 0), w.planes_a[0], w.ea[0], ba.in_stride, nullptr},
                              {view_A_part(op_A, m, k, a, lda, 1), w.planes_a[1], w.ea[1], ba.in_stride, nullptr},
                              {view_B_part(op_B, k, n, b, ldb, 0), w.planes_b[0], w.eb[0], bb.in_stride, nullptr},
                              {view_B_part(op_B, k, n, b, ldb, 1), w.planes_b[1], w.eb[1], bb.in_stride, nullptr}};
    if (!hip_ok(launch_split_fused(jobs, 4, S, L, h->stream, (uint32_t)bs.count, slot), "split")) return 3;
    if (prof && !hip_ok(hipEventRecord(h->ev[1], h->stream), "event")) return 3;
  } else if (multi_view_split(16 * (m + n) * k * bs.count)) {
    const SplitJob jobs[4] = {{view_A_part(op_A, m, k, a, lda, 0), w.planes_a[0], w.ea[0], ba.in_stride, xw.a[0]},
                              {view_A_part(op_A, m, k, a, lda, 1), w.planes_a[1], w.ea[1], ba.in_stride, xw.a[1]},
                              {view_B_part(op_B, k, n, b, ldb, 0), w.planes_b[0], w.eb[0], bb.in_stride, xw.b[0]},
                              {view_B_part(op_B, k, n, b, ldb, 1), w.planes_b[1], w.eb[1], bb.in_stride, xw.b[1]}};
    if (!have_row_max &&
        !hip_ok(launch_row_max_multi(jobs, 4, h->stream, (uint32_t)bs.count, slot, xw.pitch, xw.tag), "row_max_exp"))
      return 3;
    if (prof && !hip_ok(hipEventRecord(h->ev[1], h->stream), "event")) return 3;
    if (!hip_ok(launch_cut_multi(jobs, 4, S, L, h->stream, (uint32_t)bs.count, slot, xw.pitch, xw.tag), "cut")) return 3;
  } else {
    for (int part = 0; part < 2; part++)
      if (!run_split(h, view_A_part(op_A, m, k, a, lda, part), xw.a[part], S, L, w.planes_a[part], w.ea[part], ba, have_row_max))
        return 3;
    if (prof && !hip_ok(hipEventRecord(h->ev[1], h->stream), "event")) return 3;
    for (int part = 0; part < 2; part++)
      if (!run_split(h, view_B_part(op_B, k, n, b, ldb, part), xw.b[part], S, L, w.planes_b[part], w.eb[part], bb, have_row_max))
        return 3;
  }
  if (prof && !hip_ok(hipEventRecord(h->ev[2], h->stream), "event")) return 3;

  // From here on C is modified in place (beta scaling, then four accumulating products): a failure below is reported
  // as status 4 so that no caller hands the half-updated C to another GEMM (beta would be applied twice).
  if (!hip_ok(launch_scale_c_complex(m, n, c, ldc, beta[0], beta[1], h->stream, (uint32_t)bs.count, bs.stride_c),
              "scale_c"))
    return 3; // :477
  int launches = 0;

  static const int order[4][2] = {{1, 1}, {0, 0}, {1, 0}, {0, 1}}; // src/gemm.cu:479-480
  SliceGemmArgs prod[4];
  for (int q = 0; q < 4; q++) {
    const int *pq = order[q];
    SliceGemmArgs &g = prod[q];
    g = SliceGemmArgs{};
    g.device = h->device;
    g.a_planes = w.planes_a[pq[0]];
    g.b_planes = w.planes_b[pq[1]];
    g.KB = (uint32_t)k_blocks(k);
    g.M = (uint32_t)m;
    g.N = (uint32_t)n;
    g.L = L;
    g.ea = w.ea[pq[0]];
    g.eb = w.eb[pq[1]];
    g.cplx = 1;
    if (pq[0] == 0 && pq[1] == 0) { // src/gemm.cu:501-512
      g.alpha = alpha[0];
      g.alpha_im = alpha[1];
    } else if (pq[0] == 1 && pq[1] == 1) {
      g.alpha = -alpha[0];
      g.alpha_im = -alpha[1];
    } else {
      g.alpha = -alpha[1];
      g.alpha_im = alpha[0];
    }
    if ((pq[0] == 1 && op_A == OZIMMU_OP_C) != (pq[1] == 1 && op_B == OZIMMU_OP_C)) { // one conjugated imaginary part
      g.alpha = -g.alpha;
      g.alpha_im = -g.alpha_im;
    }
    g.c = c;
    g.ldc = ldc;
    g.acc = w.acc;
    g.phase = use_phase ? w.phase : nullptr;
    g.throttle = bs.count > 1 ? 0u : throttle_for(m, n, k);
    g.batch = (uint32_t)bs.count;
    g.ws_stride = slot;
    g.c_stride = bs.stride_c;
  }
  // (an even number of k-blocks per pass: the k64 tile function walks two per step)
  const uint32_t kbp = kb_per_pass(S, L);
  bool fused = false;
  if (prod[0].KB <= kbp && !config().test_fail_launch) {
    // one K chunk: the four products may run as ONE launch when the K-split kernel applies (small problems: three launch
    // and drain rounds less); same order of updates per element of C.  (The fault-injection test addresses launches by
    // number and keeps the one-by-one form.)
    for (SliceGemmArgs &g : prod) {
      g.kb0 = 0;
      g.kb1 = g.KB;
      g.acc_in = 0;
      g.final = 1;
    }
    note_pick(0, -1);
    const hipError_t fe = launch_slice_gemm_fused(S, prod, 4, h->stream);
    if (fe == hipSuccess) {
      h->last_kernel[0] = last_pick(0);
      h->last_kernel[1] = -1;
      fused = true;
      launches = 4;
    } else if (fe != hipErrorNotSupported) {
      (void)hip_ok(fe, "slice_gemm");
      return 4;
    } else {
      (void)hipGetLastError();
    }
  }
  for (int q = 0; q < 4 && !fused; q++) {
    SliceGemmArgs &g = prod[q];
    for (uint32_t kb0 = 0; kb0 < g.KB; kb0 += kbp) {
      g.kb0 = kb0;
      g.kb1 = std::min(g.KB, kb0 + kbp);
      g.acc_in = kb0 != 0;
      g.final = g.kb1 == g.KB;
      g.qslot = 2u * (uint32_t)launches;
      if (!launch_gemm_checked(h, S, g, h->stream, launches)) return 4;
    }
  }
  if (prof) {
    if (!hip_ok(hipEventRecord(h->ev[3], h->stream), "event")) return 4;
    if (!hip_ok(hipEventSynchronize(h->ev[3]), "event sync")) return 4;
    for (int i = 0; i < 3; i++) {
      float ms = 0;
      hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]);
      h->stage_last_ms[i] = ms;
      h->stage_total_ms[i] += ms;
    }
    h->stage_calls++;
  }
  return 0;
}

} // namespace ozhip

// ---- C ABI ---------------------------------------------------------------------------------------------

extern "C" {

const char *ozimmu_hip_version(void) { return "ozimmu_hip 0.1 (gfx950)"; }

uint32_t ozimmu_hip_get_bits_per_int8(uint32_t k) { // src/split.cu:520-536
  if (k == 0) return 0;
  uint32_t log2_k = 0; // ceil(log2(k))
  while (log2_k < 31 && (1u << (log2_k + 1)) <= k) log2_k++;
  if ((1u << log2_k) != k) log2_k++;
  if (log2_k >= 31) return 0; // k > 2^30: the reference's unsigned (31 - log2_k) wraps; no slice width is safe
  return std::min<uint32_t>(7, (31 - log2_k) / 2);
}

const char *ozimmu_hip_get_compute_mode_name_str(ozimmu_compute_mode_t mode) { // src/handle.cu:146-192
  if ((int)mode < 0 || (int)mode > (int)OZIMMU_FP64_INT8_AUTO) return nullptr;
  return kModeNames[(int)mode];
}

ozimmu_data_t ozimmu_hip_get_output_type(ozimmu_compute_mode_t mode) { // src/handle.cu:195-226
  if ((int)mode < 0 || (int)mode > (int)OZIMMU_FP64_INT8_AUTO) return OZIMMU_DATA_ORIGINAL;
  return mode == OZIMMU_SGEMM ? OZIMMU_DATA_FP32 : OZIMMU_DATA_FP64;
}

size_t ozimmu_hip_get_data_size_in_byte(ozimmu_data_t d) { // src/handle.cu:228-245
  switch (d) {
  case OZIMMU_DATA_FP64: return 8;
  case OZIMMU_DATA_FP32: return 4;
  case OZIMMU_DATA_FP16: return 2;
  case OZIMMU_DATA_INT8: return 1;
  default: return 0;
  }
}

unsigned long long ozimmu_hip_getenv_calls(void) { return getenv_calls(); }

ozimmu_compute_mode_t ozimmu_hip_compute_mode_from_str(const char *name) { // src/cublas.cu:18-48
  if (name)
    for (int i = 0; i <= (int)OZIMMU_FP64_INT8_AUTO; i++)
      if (std::strcmp(name, kModeNames[i]) == 0) return (ozimmu_compute_mode_t)i;
  return OZIMMU_DGEMM;
}

int ozimmu_hip_get_num_split(ozimmu_compute_mode_t mode) { return num_split_of_mode(mode); }

int ozimmu_hip_create(ozimmu_hip_handle_t *handle, ozimmu_malloc_mode_t mm) { // src/handle.cu:6-33
  if (!handle) return 1;
  log_info("Initializing ozIMMU handle");
  ozimmu_hip_handle *h = new (std::nothrow) ozimmu_hip_handle;
  if (!h) return 1;
  h->malloc_mode = mm;
  if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
  if (!hip_ok(hipMalloc((void **)&h->d_mantissa_loss_counter_ptr, sizeof(unsigned long long) * 16),
              "hipMalloc(counters)")) {
    delete h;
    *handle = nullptr;
    return 3;
  }
  if (hipHostMalloc((void **)&h->h_mantissa_loss_pinned, sizeof(unsigned long long) * 16, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    h->h_mantissa_loss_pinned = nullptr; // (the statistic then lands in pageable memory, as before)
  }
  for (auto &e : h->ev) hipEventCreate(&e);
  hipEventCreateWithFlags(&h->tail_ev, hipEventDisableTiming);
  probe_topology(h->device); // once per device: CU / XCD count, sustained MFMA time (the launch policy plans with them)
  auto read_thr = [](const char *name) -> uint32_t { // std::stoul in the reference (throws); here: default
    const std::string s = load_env_if_defined(name, "1024");
    char *end = nullptr;
    const unsigned long v = std::strtoul(s.c_str(), &end, 10);
    if (end == s.c_str()) {
      log_error(std::string("Invalid value for ") + name + ": " + s + " (using 1024)");
      return 1024u;
    }
    return (uint32_t)v;
  };
  h->intercept_threshold_m = read_thr("OZIMMU_INTERCEPT_THRESHOLD_M");
  h->intercept_threshold_n = read_thr("OZIMMU_INTERCEPT_THRESHOLD_N");
  h->intercept_threshold_k = read_thr("OZIMMU_INTERCEPT_THRESHOLD_K");
  *handle = h;
  return 0;
}

int ozimmu_hip_device_info(ozimmu_hip_handle_t h, double out[4]) {
  if (!h || !out) return 1;
  std::lock_guard<std::recursive_mutex> lock(h->mtx);
  const Topology t = topology(h->device);
  out[0] = t.cus;
  out[1] = t.xcds;
  out[2] = t.mfma32_us;
  out[3] = t.mfma32_measured_us;
  return 0;
}

static int predict_with(const Topology &topo, int num_split, int pass, size_t m, size_t n, size_t k, size_t batch,
                        double out_us[6], int *pick) {
  if (!out_us || num_split < 3 || num_split > 18 || m == 0 || n == 0 || m >= ((size_t)1 << 31) || n >= ((size_t)1 << 31))
    return 1;
  const int L = bits_for_k(k);
  PassTraits t;
  if (L == 0 || !slice_gemm_traits(num_split, pass, &t)) return 1;
  PolicyInput in;
  in.M = (uint32_t)m;
  in.N = (uint32_t)n;
  in.nkb = std::min<uint32_t>((uint32_t)k_blocks(k), kb_per_pass(num_split, L)); // the first K chunk
  in.batch = (uint32_t)std::max<size_t>(1, batch);
  const Prediction r = policy_predict(t, in, topo, config());
  for (int i = 0; i < POLICY_KERNELS; i++) out_us[i] = r.us[i];
  if (pick) *pick = (int)r.pick + (r.breg ? 8 : 0);
  return 0;
}

int ozimmu_hip_policy_predict(ozimmu_hip_handle_t h, int num_split, int pass, size_t m, size_t n, size_t k, size_t batch,
                              double out_us[6], int *pick) {
  Topology topo; // no handle (tools/policy_fit.py on a box without a GPU): the nominal device, or what the caller put into the
                 // last two entries of the parameter table (CUs, MFMA time of the device the measurements came from)
  if (h) {
    std::lock_guard<std::recursive_mutex> lock(h->mtx);
    topo = topology(h->device); // the device the handle was created on, whatever is current now
  } else {
    double p[POLICY_PARAMS];
    policy_params_get(p);
    if (p[POLICY_PARAMS - 2] > 0) topo.cus = (int)p[POLICY_PARAMS - 2];
    if (p[POLICY_PARAMS - 1] > 0) topo.mfma32_us = p[POLICY_PARAMS - 1];
  }
  return predict_with(topo, num_split, pass, m, n, k, batch, out_us, pick);
}

int ozimmu_hip_policy_predict_device(int device, int num_split, int pass, size_t m, size_t n, size_t k, size_t batch,
                                     double out_us[6], int *pick) {
  if (device < 0 || device >= 64) return 1;
  return predict_with(topology(device), num_split, pass, m, n, k, batch, out_us, pick);
}

int ozimmu_hip_device_topology(int device, double inout[4], int set) {
  if (!inout) return 1;
  Topology t;
  if (!set) {
    if (!topology_slot(device, &t, false)) return 1;
    inout[0] = t.cus;
    inout[1] = t.xcds;
    inout[2] = t.mfma32_us;
    inout[3] = t.mfma32_measured_us;
    return 0;
  }
#ifdef OZIMMU_HIP_TEST_HOOKS
  if (!(inout[0] >= 1 && inout[0] <= 4096 && inout[1] >= 1 && inout[1] <= MAX_XCDS && inout[2] > 0)) return 1;
  t.cus = (int)inout[0];
  t.xcds = (int)inout[1];
  t.mfma32_us = inout[2];
  t.mfma32_measured_us = inout[3];
  t.probed = true;
  return topology_slot(device, &t, true) ? 0 : 1;
#else
  return 2; // the library that ships plans with what it probed
#endif
}

int ozimmu_hip_policy_params(double *params, int count, int set) {
  if (!params || count < 0 || count > POLICY_PARAMS) return 1;
  if (set) {
    policy_params_set(params, count);
  } else {
    double p[POLICY_PARAMS];
    policy_params_get(p);
    for (int i = 0; i < count; i++) params[i] = p[i];
  }
  return 0;
}

int ozimmu_hip_last_kernel(ozimmu_hip_handle_t h, int out[2]) {
  if (!h || !out) return 1;
  std::lock_guard<std::recursive_mutex> lock(h->mtx);
  out[0] = h->last_kernel[0];
  out[1] = h->last_kernel[1];
  return 0;
}

int ozimmu_hip_tuner_state_ex(ozimmu_hip_handle_t h, int num_split, int op_A, int op_B, int beta_nonzero, size_t m, size_t n,
                              size_t k, int out[4]) {
  if (!h || !out) return -1;
  std::lock_guard<std::recursive_mutex> lock(h->mtx);
  int cur = h->device; // (collecting finished samples queries events of the handle's device)
  const bool sw = hipGetDevice(&cur) == hipSuccess && cur != h->device && hipSetDevice(h->device) == hipSuccess;
  out[0] = out[1] = out[2] = out[3] = -1;
  const int st = tuner_state(h->tuner, num_split, op_A, op_B, beta_nonzero, m, n, k, out);
  if (sw) hipSetDevice(cur);
  return st;
}

int ozimmu_hip_tuner_state(ozimmu_hip_handle_t h, int num_split, size_t m, size_t n, size_t k, int out[2]) {
  if (!h || !out) return -1;
  int v[4];
  const int st = ozimmu_hip_tuner_state_ex(h, num_split, -1, -1, 0, m, n, k, v);
  out[0] = v[0];
  out[1] = v[1];
  return st;
}

int ozimmu_hip_destroy(ozimmu_hip_handle_t h) { // src/handle.cu:35-52
  if (h) {
    log_info("Destroying ozIMMU handle");
    if (h->rocblas_handle) {
      auto destroy = (rocblas_status(*)(rocblas_handle))vendor_symbol("rocblas_destroy_handle");
      if (destroy) destroy((rocblas_handle)h->rocblas_handle);
    }
    if (h->working_memory_ptr) hipFree(h->working_memory_ptr);
    if (h->d_mantissa_loss_counter_ptr) hipFree(h->d_mantissa_loss_counter_ptr);
    if (h->h_mantissa_loss_pinned) hipHostFree(h->h_mantissa_loss_pinned);
    if (h->exp_words) hipFree(h->exp_words);
    for (void *p : h->retired_blocks) hipFree(p);
    for (auto &e : h->ev)
      if (e) hipEventDestroy(e);
    if (h->tail_ev) hipEventDestroy(h->tail_ev);
    tuner_forget(h->tuner);
    delete h;
  }
  return 0;
}

void ozimmu_hip_set_stream(ozimmu_hip_handle_t h, void *hip_stream) { // src/handle.cu:54-61
  if (!h) return;
  std::lock_guard<std::recursive_mutex> lock(h->mtx);
  h->stream = (hipStream_t)hip_stream;
}

void ozimmu_hip_enable_profiling(ozimmu_hip_handle_t h) {
  if (h) h->profiling = true;
}
void ozimmu_hip_disable_profiling(ozimmu_hip_handle_t h) {
  if (h) h->profiling = false;
}
void ozimmu_hip_clear_profiler_result(ozimmu_hip_handle_t h) {
  if (!h) return;
  for (auto &t : h->stage_total_ms) t = 0;
  h->stage_calls = 0;
}
void ozimmu_hip_print_profiler_result(ozimmu_hip_handle_t h, const char *tag, int csv) {
  if (!h) return;
  // labels of the reference's breakdown (src/gemm.cu:38-48, src/config.cu:102-118, src/gemm.cu:393-407);
  // accumulate_in_f64 and copy_result are fused into the int8tc kernel here and report 0.
  static const char *labels[5] = {"split_A", "split_B", "int8tc", "accumulate_in_f64", "copy_result"};
  const double t[5] = {h->stage_total_ms[0], h->stage_total_ms[1], h->stage_total_ms[2], 0.0, 0.0};
  const double sum = t[0] + t[1] + t[2];
  if (csv) {
    std::printf("tag,label,calls,total_ms,share\n");
    for (int i = 0; i < 5; i++)
      std::printf("%s,%s,%llu,%.6f,%.4f\n", tag ? tag : "", labels[i], h->stage_calls, t[i],
                  sum > 0 ? t[i] / sum : 0.0);
  } else {
    std::printf("# ozIMMU-HIP profiling result [%s] (%llu calls)\n", tag ? tag : "", h->stage_calls);
    for (int i = 0; i < 5; i++)
      std::printf("%20s : %12.4f ms (%6.2f%%)\n", labels[i], t[i], sum > 0 ? 100.0 * t[i] / sum : 0.0);
  }
  std::fflush(stdout);
}
int ozimmu_hip_last_stage_ms(ozimmu_hip_handle_t h, float ms[3]) {
  if (!h || !ms) return 1;
  for (int i = 0; i < 3; i++) ms[i] = h->stage_last_ms[i];
  return 0;
}

void ozimmu_hip_set_auto_mantissa_loss_threashold(ozimmu_hip_handle_t h, double threshold) {
  if (h) h->avg_mantissa_loss_threshold = threshold;
}
double ozimmu_hip_get_auto_mantissa_loss_threashold(ozimmu_hip_handle_t h) {
  return h ? h->avg_mantissa_loss_threshold : 0.0;
}

size_t ozimmu_hip_reallocate_working_memory(ozimmu_hip_handle_t h, size_t size_in_byte) { // src/handle.cu:63-93
  if (!h) return 0;
  // the old block is released here: no other thread may be between carving it and enqueueing its launches
  std::lock_guard<std::recursive_mutex> lock(h->mtx);
  if (size_in_byte <= h->current_working_memory_size) return 0;
  log_info("Reallocated memory : " + std::to_string(size_in_byte) + " B");
  if (h->working_memory_ptr) {
    // kernels already enqueued on the stream still use the old block: release it in stream order
    // (hipFree would device-synchronise, src/handle.cu:71-75 does exactly that)
    if (h->seen_capture)
      h->retired_blocks.push_back(h->working_memory_ptr); // a captured graph may still point into it (handle.h)
    else if (h->malloc_mode == OZIMMU_MALLOC_SYNC)
      hipFree(h->working_memory_ptr);
    else
      hipFreeAsync(h->working_memory_ptr, h->stream);
    h->working_memory_ptr = nullptr;
    h->current_working_memory_size = 0;
  }
  hipError_t e = h->malloc_mode == OZIMMU_MALLOC_SYNC
                     ? hipMalloc(&h->working_memory_ptr, size_in_byte)
                     : hipMallocAsync(&h->working_memory_ptr, size_in_byte, h->stream);
  if (!hip_ok(e, "workspace allocation")) {
    h->working_memory_ptr = nullptr;
    return 0;
  }
  h->current_working_memory_size = size_in_byte;
  return size_in_byte;
}

size_t ozimmu_hip_working_memory_size(ozimmu_operation_t, ozimmu_operation_t, size_t m, size_t n, size_t k,
                                      ozimmu_element_kind_t element_kind, ozimmu_compute_mode_t mode) {
  int S = num_split_of_mode(mode);
  if (mode == OZIMMU_FP64_INT8_AUTO) S = 18; // worst case of what auto may select
  if (S == 0) return 0;
  if (bits_for_k(k) == 0) return 0; // k == 0 or k > 2^30: no Ozaki path (ozimmu_hip_gemm runs the vendor GEMM)
  if (element_kind != OZIMMU_REAL) return carve_z(nullptr, m, n, k, S, needs_acc(k, S)).total;
  return carve(nullptr, m, n, k, S, needs_acc(k, S)).total;
}

static rocblas_operation to_rocblas_op(ozimmu_operation_t op) {
  return op == OZIMMU_OP_N ? rocblas_operation_none
                           : op == OZIMMU_OP_C ? rocblas_operation_conjugate_transpose : rocblas_operation_transpose;
}

int ozimmu_hip_native_dgemm(ozimmu_hip_handle_t h, ozimmu_operation_t op_A, ozimmu_operation_t op_B, size_t m,
                            size_t n, size_t k, const double *alpha, const double *a, size_t lda,
                            const double *b, size_t ldb, const double *beta, double *c, size_t ldc) {
  if (!h) return 1;
  typedef rocblas_status (*create_t)(rocblas_handle *);
  typedef rocblas_status (*set_stream_t)(rocblas_handle, hipStream_t);
  typedef rocblas_status (*dgemm_t)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int,
                                    rocblas_int, rocblas_int, const double *, const double *, rocblas_int,
                                    const double *, rocblas_int, const double *, double *, rocblas_int);
  static create_t create = (create_t)vendor_symbol("rocblas_create_handle");
  static set_stream_t set_stream = (set_stream_t)vendor_symbol("rocblas_set_stream");
  static dgemm_t dgemm = (dgemm_t)vendor_symbol("rocblas_dgemm");
  if (!create || !set_stream || !dgemm) return (int)rocblas_status_internal_error;
  std::lock_guard<std::recursive_mutex> lock(h->mtx); // lazy private handle + its stream are per-handle state
  if (!h->rocblas_handle) {
    if (stream_is_capturing(h->stream)) return (int)rocblas_status_internal_error; // creating it allocates device memory
    rocblas_handle rh = nullptr;
    const rocblas_status st = create(&rh);
    if (st != rocblas_status_success) return (int)st;
    h->rocblas_handle = rh;
  }
  set_stream((rocblas_handle)h->rocblas_handle, h->stream);
  return (int)dgemm((rocblas_handle)h->rocblas_handle, to_rocblas_op(op_A), to_rocblas_op(op_B),
                    (rocblas_int)m, (rocblas_int)n, (rocblas_int)k, alpha, a, (rocblas_int)lda, b,
                    (rocblas_int)ldb, beta, c, (rocblas_int)ldc);
}

// native complex GEMM for the `dgemm` mode with complex operands (src/gemm.cu:639-645 with CUDA_C_64F)
static int native_zgemm(ozimmu_hip_handle_t h, ozimmu_operation_t op_A, ozimmu_operation_t op_B, size_t m, size_t n,
                        size_t k, const void *alpha, const void *a, size_t lda, const void *b, size_t ldb,
                        const void *beta, void *c, size_t ldc) {
  typedef rocblas_status (*create_t)(rocblas_handle *);
  typedef rocblas_status (*set_stream_t)(rocblas_handle, hipStream_t);
  typedef rocblas_status (*zgemm_t)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int,
                                    rocblas_int, const rocblas_double_complex *, const rocblas_double_complex *,
                                    rocblas_int, const rocblas_double_complex *, rocblas_int,
                                    const rocblas_double_complex *, rocblas_double_complex *, rocblas_int);
  static create_t create = (create_t)vendor_symbol("rocblas_create_handle");
  static set_stream_t set_stream = (set_stream_t)vendor_symbol("rocblas_set_stream");
  static zgemm_t zgemm = (zgemm_t)vendor_symbol("rocblas_zgemm");
  if (!create || !set_stream || !zgemm) return (int)rocblas_status_internal_error;
  std::lock_guard<std::recursive_mutex> lock(h->mtx);
  if (!h->rocblas_handle) {
    if (stream_is_capturing(h->stream)) return (int)rocblas_status_internal_error; // creating it allocates device memory
    rocblas_handle rh = nullptr;
    const rocblas_status st = create(&rh);
    if (st != rocblas_status_success) return (int)st;
    h->rocblas_handle = rh;
  }
  set_stream((rocblas_handle)h->rocblas_handle, h->stream);
  return (int)zgemm((rocblas_handle)h->rocblas_handle, to_rocblas_op(op_A), to_rocblas_op(op_B), (rocblas_int)m,
                    (rocblas_int)n, (rocblas_int)k, (const rocblas_double_complex *)alpha,
                    (const rocblas_double_complex *)a, (rocblas_int)lda, (const rocblas_double_complex *)b,
                    (rocblas_int)ldb, (const rocblas_double_complex *)beta, (rocblas_double_complex *)c,
                    (rocblas_int)ldc);
}

// ---- `sgemm` compute mode (src/cublas_helper.cu:83-133) -------------------------------------------------------
int ozimmu_hip_gemm_f32(ozimmu_hip_handle_t h, ozimmu_operation_t op_A, ozimmu_operation_t op_B, size_t m, size_t n,
                        size_t k, const void *alpha, const void *a, size_t lda, const void *b, size_t ldb,
                        const void *beta, void *c, size_t ldc, ozimmu_element_kind_t element_kind) {
  if (!h || !alpha || !beta) return 1;
  if (check_gemm_shape(op_A, m, k, lda, "A") | check_gemm_shape(op_B, k, n, ldb, "B") |
      check_gemm_shape(OZIMMU_OP_N, m, n, ldc, "C"))
    return 1;
  if (m == 0 || n == 0) return 0;
  typedef rocblas_status (*create_t)(rocblas_handle *);
  typedef rocblas_status (*set_stream_t)(rocblas_handle, hipStream_t);
  typedef rocblas_status (*sgemm_t)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int,
                                    rocblas_int, const float *, const float *, rocblas_int, const float *, rocblas_int,
                                    const float *, float *, rocblas_int);
  typedef rocblas_status (*cgemm_t)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int,
                                    rocblas_int, const rocblas_float_complex *, const rocblas_float_complex *,
                                    rocblas_int, const rocblas_float_complex *, rocblas_int,
                                    const rocblas_float_complex *, rocblas_float_complex *, rocblas_int);
  static create_t create = (create_t)vendor_symbol("rocblas_create_handle");
  static set_stream_t set_stream = (set_stream_t)vendor_symbol("rocblas_set_stream");
  static sgemm_t sgemm = (sgemm_t)vendor_symbol("rocblas_sgemm");
  static cgemm_t cgemm = (cgemm_t)vendor_symbol("rocblas_cgemm");
  if (!create || !set_stream || !sgemm || !cgemm) return 3;
  std::lock_guard<std::recursive_mutex> lock(h->mtx);
  if (!h->rocblas_handle) {
    if (stream_is_capturing(h->stream)) return 3; // creating it allocates device memory: not while capturing a graph
    rocblas_handle rh = nullptr;
    if (create(&rh) != rocblas_status_success) return 3;
    h->rocblas_handle = rh;
  }
  const bool cplx = element_kind != OZIMMU_REAL;
  const size_t w = cplx ? 2 : 1; // FP32 scalars per element
  // workspace: A32 | B32 | C32, each 256-byte aligned (the reference packs them, :97-100)
  const size_t a_bytes = align256(4 * w * m * k), b_bytes = align256(4 * w * k * n), c_bytes = align256(4 * w * m * n);
  WorkspaceUse use(h);
  if (!use.ok || !ensure_workspace(h, a_bytes + b_bytes + c_bytes)) return 3;
  float *a32 = (float *)h->working_memory_ptr;
  float *b32 = (float *)((char *)h->working_memory_ptr + a_bytes);
  float *c32 = (float *)((char *)h->working_memory_ptr + a_bytes + b_bytes);
  // stored shapes (:102-106): A is m x k when op_A == N, else k x m; the FP32 copies are dense (ld = rows)
  const size_t ar = op_A == OZIMMU_OP_N ? m : k, ac = op_A == OZIMMU_OP_N ? k : m;
  const size_t br = op_B == OZIMMU_OP_N ? k : n, bc = op_B == OZIMMU_OP_N ? n : k;
  const double *al = (const double *)alpha, *be = (const double *)beta;
  const bool beta_zero = be[0] == 0 && (!cplx || be[1] == 0);
  if (!hip_ok(launch_convert_f64_to_f32(a32, w * ar, (const double *)a, w * lda, w * ar, ac, h->stream), "convert A"))
    return 3;
  if (!hip_ok(launch_convert_f64_to_f32(b32, w * br, (const double *)b, w * ldb, w * br, bc, h->stream), "convert B"))
    return 3;
  if (!beta_zero && // :109-112
      !hip_ok(launch_convert_f64_to_f32(c32, w * m, (const double *)c, w * ldc, w * m, n, h->stream), "convert C"))
    return 3;
  set_stream((rocblas_handle)h->rocblas_handle, h->stream);
  rocblas_status st;
  if (!cplx) {
    const float al32 = (float)al[0], be32 = (float)be[0];
    st = sgemm((rocblas_handle)h->rocblas_handle, to_rocblas_op(op_A), to_rocblas_op(op_B), (rocblas_int)m,
               (rocblas_int)n, (rocblas_int)k, &al32, a32, (rocblas_int)ar, b32, (rocblas_int)br, &be32, c32,
               (rocblas_int)m);
  } else {
    const rocblas_float_complex al32((float)al[0], (float)al[1]), be32((float)be[0], (float)be[1]);
    st = cgemm((rocblas_handle)h->rocblas_handle, to_rocblas_op(op_A), to_rocblas_op(op_B), (rocblas_int)m,
               (rocblas_int)n, (rocblas_int)k, &al32, (const rocblas_float_complex *)a32, (rocblas_int)ar,
               (const rocblas_float_complex *)b32, (rocblas_int)br, &be32, (rocblas_float_complex *)c32,
               (rocblas_int)m);
  }
  if (st != rocblas_status_success) return 3;
  if (!hip_ok(launch_convert_f32_to_f64((double *)c, w * ldc, c32, w * m, w * m, n, h->stream), "convert C back"))
    return 3;
  return 0;
}

// real: a/b are double arrays; complex (src/split.cu:367-374): Re and Im counted separately, same 16 counters
static int mantissa_loss_impl(ozimmu_hip_handle_t h, ozimmu_operation_t op_A, ozimmu_operation_t op_B, size_t m,
                              size_t n, size_t k, const double *a, size_t lda, const double *b, size_t ldb, bool cplx,
                              uint64_t counters[16]) {
  std::lock_guard<std::recursive_mutex> lock(h->mtx);
  const int L = (int)ozimmu_hip_get_bits_per_int8((uint32_t)k); // src/split.cu:461
  const int parts = cplx ? 2 : 1;
  if (stream_is_capturing(h->stream)) return 3; // the statistic is read back on the host: not capturable into a graph
  WorkspaceUse use(h);
  // the row maxima go to the tagged exponent words, in the layout the GEMM that follows (fp64_int8_auto) will ask for:
  // it finds them there and skips its own row-maximum pass (handle.h: ExpReuse)
  ExpWords xw;
  if (!exp_words(h, m, n, parts, 1, xw)) return 3;
  Batch tagged;
  tagged.tag = xw.tag;
  bool ok = hip_ok(hipMemsetAsync(h->d_mantissa_loss_counter_ptr, 0, 16 * sizeof(unsigned long long), h->stream),
                   "memset"); // all 16 zeroed (src/split.cu:302-315 zeroes 8)
  for (int part = 0; part < parts && ok; part++) {
    uint32_t *xa = xw.a[part], *xb = xw.b[part];
    const OperandView va = cplx ? view_A_part(op_A, m, k, a, lda, part) : view_A(op_A, m, k, a, lda);
    const OperandView vb = cplx ? view_B_part(op_B, k, n, b, ldb, part) : view_B(op_B, k, n, b, ldb);
    ok = hip_ok(launch_row_max_exp(va, xa, h->stream, tagged), "row_max_exp") &&
         hip_ok(launch_row_max_exp(vb, xb, h->stream, tagged), "row_max_exp") &&
         hip_ok(launch_mantissa_loss(va, xa, L, h->d_mantissa_loss_counter_ptr, h->stream, xw.tag), "loss") &&
         hip_ok(launch_mantissa_loss(vb, xb, L, h->d_mantissa_loss_counter_ptr, h->stream, xw.tag), "loss");
  }
  if (!ok) return 3;
  if (h->exp_reuse.armed) {
    h->exp_reuse.valid = true;
    h->exp_reuse.a = a, h->exp_reuse.b = b, h->exp_reuse.lda = lda, h->exp_reuse.ldb = ldb;
    h->exp_reuse.m = m, h->exp_reuse.n = n, h->exp_reuse.k = k;
    h->exp_reuse.op_a = (int)op_A, h->exp_reuse.op_b = (int)op_B, h->exp_reuse.parts = parts;
  }
  unsigned long long stack_host[16];
  unsigned long long *host = h->h_mantissa_loss_pinned ? h->h_mantissa_loss_pinned : stack_host;
  // blocking download, as src/split.cu:404-408
  if (!hip_ok(hipMemcpyAsync(host, h->d_mantissa_loss_counter_ptr, sizeof(stack_host), hipMemcpyDeviceToHost, h->stream),
              "memcpy") ||
      !hip_ok(hipStreamSynchronize(h->stream), "sync"))
    return 3;
  for (int i = 0; i < 16; i++) counters[i] = host[i];
  return 0;
}

int ozimmu_hip_mantissa_loss(ozimmu_hip_handle_t h, ozimmu_operation_t op_A, ozimmu_operation_t op_B, size_t m,
                             size_t n, size_t k, const double *a, size_t lda, const double *b, size_t ldb,
                             uint64_t counters[16]) {
  if (!h || !counters) return 1;
  return mantissa_loss_impl(h, op_A, op_B, m, n, k, a, lda, b, ldb, false, counters);
}

ozimmu_compute_mode_t ozimmu_hip_auto_mode_select(ozimmu_hip_handle_t h, ozimmu_operation_t op_A,
                                                  ozimmu_operation_t op_B, size_t m, size_t n, size_t k,
                                                  const void *a, size_t lda, const void *b, size_t ldb,
                                                  ozimmu_element_kind_t element_kind, double threshold) {
  if (!h) return OZIMMU_DGEMM;
  uint64_t cnt[16];
  if (mantissa_loss_impl(h, op_A, op_B, m, n, k, (const double *)a, lda, (const double *)b, ldb,
                         element_kind != OZIMMU_REAL, cnt))
    return OZIMMU_DGEMM;
  const double denom = (double)(m * k + k * n); // src/split.cu:486
  for (int s = 3; s <= 18; s++)
    if ((double)cnt[s - 3] / denom <= threshold) return (ozimmu_compute_mode_t)((int)OZIMMU_FP64_INT8_3 + s - 3);
  return OZIMMU_DGEMM; // src/split.cu:493
}

int ozimmu_hip_gemm(ozimmu_hip_handle_t h, ozimmu_operation_t op_A, ozimmu_operation_t op_B, size_t m, size_t n,
                    size_t k, const void *alpha, const void *a, size_t lda, const void *b, size_t ldb,
                    const void *beta, void *c, size_t ldc, ozimmu_compute_mode_t mode,
                    ozimmu_element_kind_t element_kind) {
  if (!h) return 1;
  // src/gemm.cu:535-556
  int arg_error = 0;
  arg_error |= check_gemm_shape(op_A, m, k, lda, "A");
  arg_error |= check_gemm_shape(op_B, k, n, ldb, "B");
  arg_error |= check_gemm_shape(OZIMMU_OP_N, m, n, ldc, "C");
  const size_t elem = element_kind == OZIMMU_REAL ? 8 : 16;
  arg_error |= check_address_alignment(a, elem, "A");
  arg_error |= check_address_alignment(b, elem, "B");
  arg_error |= check_address_alignment(c, elem, "B"); // sic: the reference labels C as "B" too
  if (arg_error || !alpha || !beta) return 1;
  if ((int)mode < 0 || (int)mode > (int)OZIMMU_FP64_INT8_AUTO) {
    log_error("Not implemented (unknown compute mode)"); // OZIMMU_NOT_IMPLEMENTED throws in the reference
    return 2;
  }
  if (m == 0 || n == 0) return 0;
  const bool cplx = element_kind != OZIMMU_REAL;
  std::lock_guard<std::recursive_mutex> lock(h->mtx); // the whole enqueue, including auto mode's recursion

  // Calls the Ozaki path does not cover run on the vendor GEMM, which implements the BLAS semantics for them:
  //   * k == 0 or alpha == 0: C = beta * C without reading A and B (the split would turn a NaN in an unused A into a
  //     NaN row of C);
  //   * k > 2^30: no slice width keeps the INT32 products exact (get_bits_per_int8 returns 0, src/split.cu:520-536);
  //   * m or n >= 2^31: the kernels index rows and columns with 32 bits.
  const double *al = (const double *)alpha;
  const bool alpha_zero = al[0] == 0.0 && (!cplx || al[1] == 0.0);
  const bool vendor_only = k == 0 || alpha_zero || bits_for_k(k) == 0 || m >= ((size_t)1 << 31) || n >= ((size_t)1 << 31);

  if (mode == OZIMMU_FP64_INT8_AUTO && !vendor_only) { // src/gemm.cu:628-638
    {
      // the statistic is read back on the host (blocking copy): impossible while the stream is captured into a graph
      std::lock_guard<std::recursive_mutex> lock(h->mtx);
      if (stream_is_capturing(h->stream)) {
        // The interposer captures the VENDOR routine for this call: the eager and the replayed runs of one program then
        // differ in numerics.  Said once per process even without OZIMMU_INFO (OZIMMU_ERROR=0 silences it), every time with it.
        static std::atomic<bool> said{false};
        const char *msg = "AUTO: the call is being captured into a graph (the mantissa-loss statistic needs a host read-back): "
                          "left to the vendor GEMM; select a fixed fp64_int8_N mode to capture the Ozaki path";
        if (env_enabled("OZIMMU_INFO", false))
          log_info(msg);
        else if (!said.exchange(true) && env_enabled("OZIMMU_ERROR", true)) {
          std::fprintf(stdout, "[ozIMMU LOG] %s\n", msg);
          std::fflush(stdout);
        }
        return 3;
      }
    }
    // statistic and GEMM under one lock: the GEMM may take over the row maxima the statistic pass has just computed
    std::lock_guard<std::recursive_mutex> lock(h->mtx);
    h->exp_reuse.armed = true;
    const ozimmu_compute_mode_t auto_mode = ozimmu_hip_auto_mode_select(
        h, op_A, op_B, m, n, k, a, lda, b, ldb, element_kind, h->avg_mantissa_loss_threshold);
    h->exp_reuse.armed = false;
    log_info(std::string("AUTO selected mode = ") + ozimmu_hip_get_compute_mode_name_str(auto_mode) +
             ", threshold average mantissa loss = " + std::to_string(h->avg_mantissa_loss_threshold));
    const int err = ozimmu_hip_gemm(h, op_A, op_B, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, auto_mode, element_kind);
    h->exp_reuse.valid = false;
    return err;
  }
  if (mode == OZIMMU_SGEMM) // src/cublas.cu:169-186 (the reference's library entry throws NOT_IMPLEMENTED here)
    return ozimmu_hip_gemm_f32(h, op_A, op_B, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, element_kind);
  const int S = num_split_of_mode(mode);
  if (S == 0 || vendor_only) { // `dgemm` (src/gemm.cu:639-645)
    const int st = cplx ? native_zgemm(h, op_A, op_B, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc)
                        : ozimmu_hip_native_dgemm(h, op_A, op_B, m, n, k, (const double *)alpha, (const double *)a,
                                                  lda, (const double *)b, ldb, (const double *)beta, (double *)c, ldc);
    return st == 0 ? 0 : 3;
  }
  if (cplx)
    return gemm_int8_complex(h, op_A, op_B, m, n, k, (const double *)alpha, (const double *)a, lda, (const double *)b,
                             ldb, (const double *)beta, (double *)c, ldc, S);
  return gemm_int8_real(h, op_A, op_B, m, n, k, *(const double *)alpha, (const double *)a, lda,
                        (const double *)b, ldb, *(const double *)beta, (double *)c, ldc, S, nullptr);
}

int ozimmu_hip_gemm_on_stream(ozimmu_hip_handle_t h, void *hip_stream, ozimmu_operation_t op_A,
                              ozimmu_operation_t op_B, size_t m, size_t n, size_t k, const void *alpha, const void *a,
                              size_t lda, const void *b, size_t ldb, const void *beta, void *c, size_t ldc,
                              ozimmu_compute_mode_t mode, ozimmu_element_kind_t element_kind) {
  if (!h) return 1;
  std::lock_guard<std::recursive_mutex> lock(h->mtx); // stream switch + enqueue are one critical section
  h->stream = (hipStream_t)hip_stream;
  return ozimmu_hip_gemm(h, op_A, op_B, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, mode, element_kind);
}

int ozimmu_hip_gemm_strided_batched(ozimmu_hip_handle_t h, void *hip_stream, ozimmu_operation_t op_A,
                                    ozimmu_operation_t op_B, size_t m, size_t n, size_t k, const void *alpha,
                                    const void *a, size_t lda, long long stride_a, const void *b, size_t ldb,
                                    long long stride_b, const void *beta, void *c, size_t ldc, long long stride_c,
                                    size_t batch_count, ozimmu_compute_mode_t mode, ozimmu_element_kind_t element_kind) {
  if (!h) return 1;
  int arg_error = 0; // src/gemm.cu:535-556, once for the whole batch
  arg_error |= check_gemm_shape(op_A, m, k, lda, "A");
  arg_error |= check_gemm_shape(op_B, k, n, ldb, "B");
  arg_error |= check_gemm_shape(OZIMMU_OP_N, m, n, ldc, "C");
  const size_t es = element_kind == OZIMMU_REAL ? 8 : 16;
  arg_error |= check_address_alignment(a, es, "A");
  arg_error |= check_address_alignment(b, es, "B");
  arg_error |= check_address_alignment(c, es, "B");
  if (arg_error || !alpha || !beta) return 1;
  if ((int)mode < 0 || (int)mode > (int)OZIMMU_FP64_INT8_AUTO) {
    log_error("Not implemented (unknown compute mode)");
    return 2;
  }
  if (batch_count == 0 || m == 0 || n == 0) return 0;
  std::lock_guard<std::recursive_mutex> lock(h->mtx);
  h->stream = (hipStream_t)hip_stream;
  const bool cplx = element_kind != OZIMMU_REAL;
  const double *al = (const double *)alpha;
  const bool alpha_zero = al[0] == 0.0 && (!cplx || al[1] == 0.0);
  const int S = num_split_of_mode(mode);
  auto at = [&](const void *p, long long stride, size_t i) {
    return (const void *)((const char *)p + (long long)i * stride * (long long)es);
  };
  // Per-matrix decisions (fp64_int8_auto), the FP32 mode, native modes and the BLAS quick returns take the reference's
  // sequential form (src/cublas.cu:380-406).  A failure after the first matrix leaves earlier ones updated: status 4.
  if (S == 0 || batch_count == 1 || k == 0 || alpha_zero || bits_for_k(k) == 0 || m >= ((size_t)1 << 31) ||
      n >= ((size_t)1 << 31) || config().batch_loop) {
    for (size_t i = 0; i < batch_count; i++) {
      const int st = ozimmu_hip_gemm(h, op_A, op_B, m, n, k, alpha, at(a, stride_a, i), lda, at(b, stride_b, i), ldb, beta,
                                     (void *)at(c, stride_c, i), ldc, mode, element_kind);
      if (st) return (i == 0 && st != 4) ? st : 4;
    }
    return 0;
  }
  // One set of launches per chunk of the batch: as many matrices as fit the workspace budget (default 4 GiB, at
  // least one) and the grid's y / z range.
  const size_t slot = cplx ? carve_z(nullptr, m, n, k, S, needs_acc(k, S)).total : carve(nullptr, m, n, k, S, needs_acc(k, S)).total;
  size_t budget = (size_t)4 << 30;
  if (config().batch_workspace_bytes) budget = config().batch_workspace_bytes;
  size_t chunk = std::max<size_t>(1, budget / std::max<size_t>(slot, 1));
  chunk = std::min<size_t>(chunk, 65535);
  for (size_t i0 = 0; i0 < batch_count; i0 += chunk) {
    BatchSpec bs;
    bs.count = std::min(chunk, batch_count - i0);
    bs.stride_a = stride_a;
    bs.stride_b = stride_b;
    bs.stride_c = stride_c;
    const int st = cplx ? gemm_int8_complex(h, op_A, op_B, m, n, k, (const double *)alpha, (const double *)at(a, stride_a, i0),
                                            lda, (const double *)at(b, stride_b, i0), ldb, (const double *)beta,
                                            (double *)at(c, stride_c, i0), ldc, S, bs)
                        : gemm_int8_real(h, op_A, op_B, m, n, k, *(const double *)alpha, (const double *)at(a, stride_a, i0),
                                         lda, (const double *)at(b, stride_b, i0), ldb, *(const double *)beta,
                                         (double *)at(c, stride_c, i0), ldc, S, nullptr, bs);
    if (st) return (i0 == 0 && st != 4) ? st : 4;
  }
  return 0;
}

int ozimmu_hip_diagonal_sums(ozimmu_hip_handle_t h, ozimmu_operation_t op_A, ozimmu_operation_t op_B, size_t m,
                             size_t n, size_t k, const double *a, size_t lda, const double *b, size_t ldb,
                             unsigned num_split, int32_t *out) {
  if (!h || !out || num_split < 3 || num_split > 18 || m == 0 || n == 0 || k == 0) return 1;
  if (m >= ((size_t)1 << 31) || n >= ((size_t)1 << 31) || bits_for_k(k) == 0) return 1; // 32-bit row / column indices
  if (check_gemm_shape(op_A, m, k, lda, "A") | check_gemm_shape(op_B, k, n, ldb, "B")) return 1;
  std::lock_guard<std::recursive_mutex> lock(h->mtx);
#ifdef OZIMMU_HIP_TEST_HOOKS
  return gemm_int8_real(h, op_A, op_B, m, n, k, 1.0, a, lda, b, ldb, 0.0, nullptr, m, (int)num_split, out);
#else
  return 2; // the release flavour carries no hook in its kernels
#endif
}

int ozimmu_hip_tile_plan(size_t m, size_t n, int wa, int cus, int reference, double *out) {
  if (!out || m == 0 || n == 0 || m >= ((size_t)1 << 31) || n >= ((size_t)1 << 31) || wa < 1 || wa > 8 || cus < 1) return 1;
  WidePlan pl;
  if (reference) {
#ifdef OZIMMU_HIP_TEST_HOOKS
    pl = plan_wide_with((uint32_t)m, (uint32_t)n, wa, cus, simulate_rounds_reference);
#else
    return 2;
#endif
  } else {
    pl = plan_wide((uint32_t)m, (uint32_t)n, wa, cus);
  }
  out[0] = pl.n_big;
  out[1] = pl.n_small;
  out[2] = pl.makespan;
  return 0;
}

int ozimmu_hip_split_int8(ozimmu_hip_handle_t h, int8_t *out_ptr, uint32_t ldo, double *max_exp_ptr, size_t m,
                          size_t n, const double *in_ptr, size_t ld, ozimmu_operation_t op,
                          ozimmu_matrix_t matrix, unsigned num_split, unsigned bits_per_int8) {
  if (!h || !out_ptr || !max_exp_ptr || num_split < 1 || num_split > 18 || bits_per_int8 < 1 || bits_per_int8 > 7)
    return 1;
  // src/split.cu:274-282: A: (m x n) = (rows x k) of op(A); B: (m x n) = (k x cols) of op(B)
  const OperandView v = matrix == OZIMMU_MATRIX_A ? view_A(op, m, n, in_ptr, ld) : view_B(op, m, n, in_ptr, ld);
  if (ldo < v.K || v.rows >= ((size_t)1 << 31) || v.K > ((size_t)1 << 30)) return 1;
  if (v.rows == 0) return 0;
  std::lock_guard<std::recursive_mutex> lock(h->mtx);
  const size_t exps_bytes = align256(4 * v.rows);
  const size_t plane_bytes = tiled_plane_bytes(v.rows, v.K, (int)num_split);
  WorkspaceUse use(h);
  if (!use.ok || !ensure_workspace(h, exps_bytes + plane_bytes + 256)) return 3;
  uint32_t *exps = (uint32_t *)h->working_memory_ptr;
  int8_t *planes = (int8_t *)h->working_memory_ptr + exps_bytes;
  bool ok = hip_ok(hipMemsetAsync(exps, 0, exps_bytes, h->stream), "memset");
  if (v.K == 0) // nothing to cut: max_exp of an empty row is 0
    ok = ok && hip_ok(hipMemsetAsync(max_exp_ptr, 0, 8 * v.rows, h->stream), "memset");
  if (v.K > 0 && resident_split(v.K)) { // the same kernel choice as the GEMM path makes for this K
    const SplitJob job{v, planes, max_exp_ptr, 0, nullptr};
    ok = ok && hip_ok(launch_split_resident(&job, 1, (int)num_split, (int)bits_per_int8, h->stream, 1, 0, nullptr, 0,
                                            topology(h->device).cus), "split");
  } else if (one_pass_split(8 * v.rows * v.K)) {
    const SplitJob job{v, planes, max_exp_ptr, 0, nullptr};
    ok = ok && hip_ok(launch_split_fused(&job, 1, (int)num_split, (int)bits_per_int8, h->stream), "split");
  } else {
    ok = ok && run_split(h, v, exps, (int)num_split, (int)bits_per_int8, planes, max_exp_ptr);
  }
  ok = ok &&
       hip_ok(launch_untile(planes, v.rows, v.K, (int)num_split, out_ptr, ldo, h->stream), "untile");
  return ok ? 0 : 3;
}

} // extern "C"
