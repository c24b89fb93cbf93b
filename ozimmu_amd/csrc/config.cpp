// config.cpp — reads the OZIMMU_HIP_* development switches (config.h) once per process.
#include "config.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include <cstdint>

extern char **environ;

namespace ozhip {

static std::atomic<unsigned long long> g_getenv_calls{0};

const char *counted_getenv(const char *name) {
  g_getenv_calls.fetch_add(1, std::memory_order_relaxed);
  return getenv(name);
}
unsigned long long getenv_calls() { return g_getenv_calls.load(std::memory_order_relaxed); }

static bool flag(const char *name, bool dflt) { // set and not "0" = on
  const char *e = counted_getenv(name);
  return e ? std::strcmp(e, "0") != 0 : dflt;
}
static long long number(const char *name, long long dflt) {
  const char *e = counted_getenv(name);
  return e ? std::strtoll(e, nullptr, 10) : dflt;
}

static Config read_config() {
  Config c;
  c.env_per_call = flag("OZIMMU_HIP_ENV_PER_CALL", false);
  if (const char *e = counted_getenv("OZIMMU_HIP_GEMM_KERNEL")) {
    c.gemm_kernel = !std::strcmp(e, "wide") ? Config::WIDE : !std::strcmp(e, "classic") ? Config::CLASSIC
                  : !std::strcmp(e, "k2") ? Config::K2 : !std::strcmp(e, "x16") ? Config::X16
                  : !std::strcmp(e, "k64") ? Config::K64 : Config::AUTO;
  }
  if (const char *e = counted_getenv("OZIMMU_HIP_PAIRED_TILE")) c.paired_tile = e[0] == '1' ? 1 : 0;
  if (const char *e = counted_getenv("OZIMMU_HIP_K64_TILE")) c.k64_tile = e[0] == '1' ? 1 : 0;
  if (const char *e = counted_getenv("OZIMMU_HIP_K64_BREG")) c.k64_breg = e[0] == '1' ? 1 : 0;
  if (const char *e = counted_getenv("OZIMMU_HIP_FUSED_PRODUCTS")) c.fused_products = e[0] != '0';
  c.wide_small_rows = (int)number("OZIMMU_HIP_WIDE_SMALL_ROWS", -1);
  c.wide_static = number("OZIMMU_HIP_WIDE_STATIC", 0) != 0;
  c.wide_grid = (int)number("OZIMMU_HIP_WIDE_GRID", 0);
  c.xcds = (int)number("OZIMMU_HIP_XCDS", 0);
  c.no_throttle = flag("OZIMMU_HIP_NO_THROTTLE", false);
  c.no_exp_reuse = flag("OZIMMU_HIP_NO_EXP_REUSE", false);
  c.no_phase_hint = flag("OZIMMU_HIP_NO_PHASE_HINT", false);
  c.phase_min_kb = (int)number("OZIMMU_HIP_PHASE_MIN_KB", c.phase_min_kb);
  c.spec_claim_kb = (int)number("OZIMMU_HIP_SPEC_CLAIM_KB", c.spec_claim_kb);
  c.epi_overlap = flag("OZIMMU_HIP_EPI_OVERLAP", true);
  c.static_rounds = (int)number("OZIMMU_HIP_STATIC_ROUNDS", c.static_rounds);
  c.batch_loop = flag("OZIMMU_HIP_BATCH_LOOP", false);
  c.split_band_bytes = (size_t)number("OZIMMU_HIP_SPLIT_BAND_BYTES", 0);
  c.split_one_pass_bytes = (size_t)number("OZIMMU_HIP_SPLIT_ONE_PASS_BYTES", 0);
  c.split_multi_bytes = (size_t)number("OZIMMU_HIP_SPLIT_MULTI_BYTES", (long long)c.split_multi_bytes);
  c.batch_workspace_bytes = (size_t)number("OZIMMU_HIP_BATCH_WORKSPACE_BYTES", 0);
  c.split_strip = (int)number("OZIMMU_HIP_SPLIT_STRIP", 0);
  c.split_resident = (int)number("OZIMMU_HIP_SPLIT_RESIDENT", -1);
#ifdef OZIMMU_HIP_TEST_HOOKS
  c.test_fail_launch = (int)number("OZIMMU_HIP_TEST_FAIL_LAUNCH", 0);
  c.test_exp_epoch = (uint32_t)number("OZIMMU_HIP_TEST_EXP_EPOCH", 0);
  c.test_no_stream_order = flag("OZIMMU_HIP_TEST_NO_STREAM_ORDER", false);
#endif
  return c;
}

// By value: in the follow-the-environment mode of the tests the static copy is rewritten on every call, and a reference
// handed out earlier (to another thread, or to an expression that calls config() twice) would see it change underneath.
// Production (OZIMMU_HIP_ENV_PER_CALL unset): `c` is written once, before `ready`, and read without the lock ever after.
// Follow-the-environment mode: `per_call` (its own atomic, set before `ready`) sends EVERY reader through the lock, and a
// re-read builds the new Config in a local - with env_per_call already set - and assigns it once (a reader could see
// the freshly parsed copy with env_per_call still false and return a torn snapshot).
Config config() {
  static std::mutex mtx;
  static Config c;
  static std::atomic<bool> ready{false}, per_call{false};
  if (!ready.load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lock(mtx);
    if (!ready.load(std::memory_order_relaxed)) {
      c = read_config();
      per_call.store(c.env_per_call, std::memory_order_relaxed);
      ready.store(true, std::memory_order_release);
    }
    return c;
  }
  if (!per_call.load(std::memory_order_relaxed)) return c;
  // tests / A-B tools: follow the environment - re-read only when it changed.  setenv / unsetenv replace or move entries of
  // `environ`, so the pointers themselves are a fingerprint (~100 ns per call instead of 36 getenv calls x the dozen uses of
  // config() per GEMM, which made the HOST the bottleneck of every sub-50-us problem the A/B tools timed); the first bytes of
  // every entry are hashed with them, so that a putenv() buffer edited in place is seen too.
  std::lock_guard<std::mutex> lock(mtx);
  static unsigned long long seen = 0;
  unsigned long long fp = 1469598103934665603ull;
  for (char **e = environ; e && *e; e++) {
    fp = (fp ^ (unsigned long long)(uintptr_t)*e) * 1099511628211ull;
    if (std::strncmp(*e, "OZIMMU_", 7) == 0)
      for (const char *q = *e; *q; q++) fp = (fp ^ (unsigned char)*q) * 1099511628211ull;
  }
  if (fp != seen) {
    Config fresh = read_config();
    fresh.env_per_call = true;
    c = fresh;
    seen = fp;
  }
  return c;
}

} // namespace ozhip
