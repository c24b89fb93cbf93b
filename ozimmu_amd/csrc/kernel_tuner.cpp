// kernel_tuner.cpp — see kernel_tuner.h.
#include "kernel_tuner.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "config.h"
#include "kernel_policy.h"
#include "topology.h"

namespace ozhip {

namespace {

constexpr int TUNE_MAX_CAND = 4;      // candidates per shape (8192^2 x 256: the model's order is classic, k64, wide, k64 in registers - and the
                                      // last one is the fastest by 3 %)
constexpr int TUNE_SAMPLES = 3;       // whole-call times per candidate that count; the median decides
constexpr int TUNE_ROUNDS = TUNE_SAMPLES + 1; // rounds of one sample per candidate; round 0 is thrown away (a part that comes out of
                                      // idle ramps its clock over the first calls: whoever is timed later looks faster) and each
                                      // round starts one candidate further (no candidate always sits behind the same neighbour)
constexpr double TUNE_BAND_SHORT = 1.25; // candidates: predicted within this factor of the model's best for k loops of <= 16 k-blocks (the
                                      // model's error on short loops under large outputs reaches 20 %: profiles/r5_policy/second_fit_r5.md)
constexpr double TUNE_BAND = 1.12;    // ... and within this one beyond (BASELINE C5, 32768^2 x 1024: the third-best kernel is 14 % off
                                      // and 7 % slower - timing it costs more than any choice there can return)
constexpr unsigned TUNE_SHORT_KB = 16;
constexpr double TUNE_MARGIN = 0.985; // another kernel replaces the model's pick only below this fraction of its median ...
constexpr float TUNE_MARGIN_MS = 0.001f; // ... minus a microsecond (calls of tens of microseconds: event granularity)
constexpr unsigned TUNE_MAX_KB = 64;  // k loops of at most this many k-blocks (K <= 2048): where the model's misses are (its regret on
                                      // longer loops is < 0.5 %, and a sample there costs milliseconds)
constexpr double TUNE_MIN_US = 100.0; // calls the model predicts shorter than this are not tuned: twelve of them end inside the clock ramp of
                                      // a part that comes out of idle, and their times decided 1024^3 (29 us predicted) wrongly in one run
                                      // of three (-5 %, for good: a decision is never revisited)
constexpr int TUNE_RECHECK_CALLS = 64; // a shape still in use this many calls after its first decision is measured once more, and that
                                      // second measurement stands: the first sixteen calls of a part that is still warming up do not always
                                      // rank kernels 3-5 % apart the way the steady state does (profiles/r5_policy/bench_regret_x2_r5k.txt)
constexpr size_t TUNE_MAX_SHAPES = 64; // per handle; further shapes run the model's pick
constexpr size_t TUNE_MAX_PENDING = 32; // event pairs in flight per handle

struct Key {
  int S;
  size_t m, n, k;
  bool operator==(const Key &o) const { return S == o.S && m == o.m && n == o.n && k == o.k; }
};

struct Entry {
  Key key{};
  int ncand = 0;
  int slot[TUNE_MAX_CAND] = {-1, -1, -1, -1};
  int samples[TUNE_MAX_CAND] = {0, 0, 0, 0};
  int issued = 0;   // samples asked for so far (round = issued / ncand)
  int inflight = 0; // ... of which not collected yet
  float ms[TUNE_MAX_CAND][TUNE_SAMPLES] = {};
  bool decided = false;
  int winner = -1; // prediction slot
  int phase = 0;   // 0: first measurement, 1: second (final) one
  int calls = 0;   // decided calls since the first decision
};

struct Pending {
  int entry, cand;
  bool counts;
  hipEvent_t start, stop;
};

struct Tuner {
  std::vector<Entry> entries;
  std::deque<Pending> pending;
  std::vector<hipEvent_t> pool;
};

std::mutex g_mtx;
std::unordered_map<const void *, Tuner> g_tuners;

bool enabled() {
  static std::atomic<int> once{-1}; // read once per process, like every OZIMMU_HIP_* switch (csrc/config.h); tests follow the environment
  int v = once.load(std::memory_order_relaxed);
  if (config().env_per_call || v < 0) {
    const char *e = std::getenv("OZIMMU_HIP_AUTOTUNE");
    v = (e && e[0] == '0' && e[1] == 0) ? 0 : 1;
    once.store(v, std::memory_order_relaxed);
  }
  return v == 1;
}

hipEvent_t take_event(Tuner &t) {
  if (!t.pool.empty()) {
    hipEvent_t e = t.pool.back();
    t.pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return e;
}

float median_of(const float *v, int count) {
  if (count <= 0) return -1.f;
  if (count == 1) return v[0];
  if (count == 2) return 0.5f * (v[0] + v[1]);
  const float a = v[0], b = v[1], c = v[2];
  return std::max(std::min(a, b), std::min(std::max(a, b), c));
}

void decide(Entry &e) {
  static_assert(TUNE_SAMPLES == 3, "median_of");
  int best = 0;
  const float base = median_of(e.ms[0], e.samples[0]); // candidate 0 = the model's pick
  float best_ms = base;
  for (int c = 1; c < e.ncand && base > 0.f; c++) {
    const float t = median_of(e.ms[c], e.samples[c]);
    // ... and round by round (sample r of both comes from the same round: a drift of the clock over the exploration cancels)
    int rounds_won = 0, rounds = std::min(e.samples[0], e.samples[c]);
    for (int r = 0; r < rounds; r++) rounds_won += e.ms[c][r] < e.ms[0][r] ? 1 : 0;
    if (t > 0.f && t < best_ms && t < (float)TUNE_MARGIN * base - TUNE_MARGIN_MS && 2 * rounds_won > rounds) {
      best = c;
      best_ms = t;
    }
  }
  e.winner = e.slot[best];
  e.decided = true;
}

// finished event pairs -> samples, oldest first (one stream: they finish in order; a pair that is not ready ends the sweep)
void collect(Tuner &t) {
  while (!t.pending.empty()) {
    Pending &p = t.pending.front();
    const hipError_t q = hipEventQuery(p.stop);
    if (q == hipErrorNotReady) {
      (void)hipGetLastError();
      break;
    }
    Entry &e = t.entries[(size_t)p.entry];
    e.inflight--;
    float ms = 0.f;
    if (q == hipSuccess && hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess && ms > 0.f) {
      if (p.counts && !e.decided && e.samples[p.cand] < TUNE_SAMPLES) e.ms[p.cand][e.samples[p.cand]++] = ms;
    } else {
      (void)hipGetLastError(); // (a destroyed stream, a failed call: the sample is dropped)
    }
    t.pool.push_back(p.start);
    t.pool.push_back(p.stop);
    t.pending.pop_front();
    // every round asked for and collected (dropped samples included: the decision takes what there is)
    if (!e.decided && e.issued >= TUNE_ROUNDS * e.ncand && e.inflight == 0) decide(e);
  }
}

} // namespace

TuneTicket tuner_begin(const void *owner, int device, int S, size_t m, size_t n, size_t k, unsigned nkb, hipStream_t stream) {
  TuneTicket tk;
  policy_override(-1);
  const Config cfg = config();
  if (!enabled() || nkb > TUNE_MAX_KB || cfg.forced_kernel() || cfg.paired_tile >= 0 || cfg.k64_tile >= 0 || cfg.k64_breg >= 0 || cfg.wide_grid > 0 ||
      cfg.wide_static || cfg.wide_small_rows >= 0 || cfg.xcds > 0)
    return tk;
  std::lock_guard<std::mutex> lock(g_mtx);
  Tuner &t = g_tuners[owner];
  collect(t);
  const Key key{S, m, n, k};
  int idx = -1;
  for (size_t i = 0; i < t.entries.size(); i++)
    if (t.entries[i].key == key) {
      idx = (int)i;
      break;
    }
  if (idx < 0) {
    if (t.entries.size() >= TUNE_MAX_SHAPES) return tk;
    PassTraits tr;
    if (!slice_gemm_traits(S, 0, &tr)) return tk;
    PolicyInput in;
    in.M = (uint32_t)m;
    in.N = (uint32_t)n;
    in.nkb = nkb;
    in.batch = 1;
    const Prediction r = policy_predict(tr, in, topology(device), cfg);
    Entry e;
    e.key = key;
    const int model = r.breg ? 5 : (int)r.pick;
    e.slot[0] = model;
    e.ncand = 1;
    // the other kernels predicted within the band, nearest first
    const double band = nkb <= TUNE_SHORT_KB ? TUNE_BAND_SHORT : TUNE_BAND;
    int order[POLICY_KERNELS];
    for (int s = 0; s < POLICY_KERNELS; s++) order[s] = s;
    std::sort(order, order + POLICY_KERNELS, [&](int a, int b) { return r.us[a] < r.us[b]; });
    for (int i = 0; i < POLICY_KERNELS && e.ncand < TUNE_MAX_CAND; i++) {
      const int s = order[i];
      if (s == model || r.us[s] < 0 || r.us[model] <= 0 || r.us[s] > band * r.us[model]) continue;
      e.slot[e.ncand++] = s;
    }
    if (r.us[model] < TUNE_MIN_US) e.ncand = 1;
    if (e.ncand == 1) {
      e.decided = true;
      e.winner = model;
    }
    t.entries.push_back(e);
    idx = (int)t.entries.size() - 1;
  }
  Entry &e = t.entries[(size_t)idx];
  if (e.decided && e.phase == 0 && e.ncand > 1 && ++e.calls >= TUNE_RECHECK_CALLS) {
    e.phase = 1; // (nothing of this entry is in flight: a decision needs every asked-for sample collected)
    e.decided = false;
    e.issued = 0;
    for (int c = 0; c < e.ncand; c++) e.samples[c] = 0;
  }
  if (e.decided) {
    policy_override(e.winner);
    return tk;
  }
  if (e.issued >= TUNE_ROUNDS * e.ncand || t.pending.size() >= TUNE_MAX_PENDING) {
    policy_override(e.winner >= 0 ? e.winner : e.slot[0]); // everything asked for is in flight: the first decision (or the model's pick)
    return tk;                                              // until the times are in
  }
  // round r times every candidate once, starting with candidate r (the first call of a shape runs the model's pick)
  const int round = e.issued / e.ncand, c = (e.issued % e.ncand + round) % e.ncand;
  hipEvent_t ev0 = take_event(t), ev1 = ev0 ? take_event(t) : nullptr;
  if (!ev1 || hipEventRecord(ev0, stream) != hipSuccess) {
    (void)hipGetLastError();
    if (ev0) t.pool.push_back(ev0);
    if (ev1) t.pool.push_back(ev1);
    policy_override(e.slot[0]);
    return tk;
  }
  e.issued++;
  e.inflight++;
  tk.counts = round > 0;
  policy_override(e.slot[c]);
  tk.entry = idx;
  tk.cand = c;
  tk.start = ev0;
  tk.stop = ev1;
  tk.stream = stream;
  tk.owner = owner;
  return tk;
}

void tuner_end(TuneTicket &tk, bool ok) {
  policy_override(-1);
  if (tk.entry < 0) return;
  std::lock_guard<std::mutex> lock(g_mtx);
  auto it = g_tuners.find(tk.owner);
  if (it == g_tuners.end()) return;
  Tuner &t = it->second;
  if (ok && hipEventRecord(tk.stop, tk.stream) == hipSuccess) {
    t.pending.push_back(Pending{tk.entry, tk.cand, tk.counts, tk.start, tk.stop});
  } else {
    (void)hipGetLastError();
    Entry &e = t.entries[(size_t)tk.entry];
    e.inflight--;
    if (!e.decided && e.issued >= TUNE_ROUNDS * e.ncand && e.inflight == 0) decide(e);
    t.pool.push_back(tk.start);
    t.pool.push_back(tk.stop);
  }
  tk.entry = -1;
}

void tuner_forget(const void *owner) {
  std::lock_guard<std::mutex> lock(g_mtx);
  auto it = g_tuners.find(owner);
  if (it == g_tuners.end()) return;
  for (Pending &p : it->second.pending) {
    (void)hipEventDestroy(p.start);
    (void)hipEventDestroy(p.stop);
  }
  for (hipEvent_t e : it->second.pool) (void)hipEventDestroy(e);
  (void)hipGetLastError();
  g_tuners.erase(it);
}

int tuner_state(const void *owner, int S, size_t m, size_t n, size_t k, int *slot, int *candidates) {
  std::lock_guard<std::mutex> lock(g_mtx);
  auto it = g_tuners.find(owner);
  if (it == g_tuners.end()) return -1;
  collect(it->second);
  const Key key{S, m, n, k};
  for (const Entry &e : it->second.entries)
    if (e.key == key) {
      if (slot) *slot = e.decided ? e.winner : -1;
      if (candidates) *candidates = e.ncand;
      return e.decided ? 1 : 0;
    }
  return -1;
}

} // namespace ozhip
