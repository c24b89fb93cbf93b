// kernel_tuner.cpp — see kernel_tuner.h.
#include "kernel_tuner.h"

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
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
                                      // of three (-5 %)
constexpr int TUNE_FIRST_CALL = 8;    // the first measured call of a shape (a shape seen 2-15 times ran candidates
                                      // predicted up to 25 % slower on calls 2, 3, 4 ... and never reached the decision that pays for them)
constexpr int TUNE_RECHECK_CALLS = 64; // a shape still in use this many calls after a decision is measured again: the first sixteen calls
                                      // of a part that is still warming up do not always rank kernels 3-5 % apart the way the steady
                                      // state does (profiles/r5_policy/bench_regret_x2_r5k.txt) ...
constexpr int TUNE_RECHECK_GROWTH = 8; // ... and then 512, 4096, ... calls later: whatever shared the device with the sampled calls
                                      // (another stream's kernels) is in the samples, so no measurement is final; a measurement is
                                      // <= 16 calls of which the non-winners run up to 25 % slower: < 0.1 % of the calls between two looks
constexpr size_t TUNE_MAX_SHAPES = 256; // per handle; the least recently used shape without a sample in flight makes room
constexpr size_t TUNE_MAX_PENDING = 32; // event pairs in flight per handle

struct Key {
  int S = 0, op_a = 0, op_b = 0, beta_nz = 0;
  size_t m = 0, n = 0, k = 0;
  bool operator==(const Key &o) const {
    return S == o.S && op_a == o.op_a && op_b == o.op_b && beta_nz == o.beta_nz && m == o.m && n == o.n && k == o.k;
  }
};
struct KeyHash {
  size_t operator()(const Key &x) const {
    uint64_t h = 1469598103934665603ull;
    for (uint64_t v : {(uint64_t)x.S, (uint64_t)x.op_a, (uint64_t)x.op_b, (uint64_t)x.beta_nz, (uint64_t)x.m, (uint64_t)x.n, (uint64_t)x.k})
      h = (h ^ v) * 1099511628211ull;
    return (size_t)h;
  }
};

struct Entry {
  Key key{};
  int ncand = 0;
  int slot[TUNE_MAX_CAND] = {-1, -1, -1, -1};
  int samples[TUNE_MAX_CAND] = {0, 0, 0, 0};
  int issued = 0;   // samples asked for so far in this measurement (round = issued / ncand)
  int inflight = 0; // ... of which not collected yet
  float ms[TUNE_MAX_CAND][TUNE_SAMPLES] = {};
  bool decided = false;
  int winner = -1;     // prediction slot
  int measurements = 0; // finished ones
  int seen = 0;        // calls of this shape so far (saturating)
  long long calls = 0; // decided calls since the last decision
  long long recheck_after = TUNE_RECHECK_CALLS;
  unsigned long long last_use = 0;
};

struct Pending {
  int entry, cand;
  bool counts;
  hipEvent_t start, stop;
};

} // namespace

struct Tuner {
  std::vector<Entry> entries;
  std::unordered_map<Key, int, KeyHash> index;
  std::deque<Pending> pending;
  std::vector<hipEvent_t> pool;
  unsigned long long tick = 0;
};

namespace {

struct Switches {
  bool on;
  int first_call;
};
Switches switches() {
  // read once per process, like every OZIMMU_HIP_* switch (csrc/config.h); tests follow the environment
  static std::atomic<int> once{-1}, first{TUNE_FIRST_CALL};
  int v = once.load(std::memory_order_relaxed);
  if (config().env_per_call || v < 0) {
    const char *e = std::getenv("OZIMMU_HIP_AUTOTUNE");
    v = (e && e[0] == '0' && e[1] == 0) ? 0 : 1;
    const char *a = std::getenv("OZIMMU_HIP_AUTOTUNE_AFTER");
    first.store(a ? std::max(1, std::atoi(a)) : TUNE_FIRST_CALL, std::memory_order_relaxed);
    once.store(v, std::memory_order_relaxed);
  }
  return Switches{v == 1, first.load(std::memory_order_relaxed)};
}

// all the events a handle's samples can have in flight, created in one go when its first measurement starts (never per call)
bool fill_pool(Tuner &t) {
  if (!t.pool.empty() || !t.pending.empty()) return true;
  t.pool.reserve(2 * TUNE_MAX_PENDING);
  for (size_t i = 0; i < 2 * TUNE_MAX_PENDING; i++) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) {
      (void)hipGetLastError();
      break;
    }
    t.pool.push_back(e);
  }
  return t.pool.size() >= 2;
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
    // ... and round by round (sample r of both comes from the same round: a drift of the clock over the exploration cancels,
    // and so does a neighbour on another stream that ran through one round and not the next)
    int rounds_won = 0, rounds = std::min(e.samples[0], e.samples[c]);
    for (int r = 0; r < rounds; r++) rounds_won += e.ms[c][r] < e.ms[0][r] ? 1 : 0;
    if (t > 0.f && t < best_ms && t < (float)TUNE_MARGIN * base - TUNE_MARGIN_MS && 2 * rounds_won > rounds) {
      best = c;
      best_ms = t;
    }
  }
  // a measurement that lost the model's own samples (failed calls, a destroyed stream) keeps what was decided before
  if (base > 0.f || e.winner < 0) e.winner = e.slot[best];
  e.decided = true;
  e.calls = 0;
  if (e.measurements++ > 0) e.recheck_after *= TUNE_RECHECK_GROWTH;
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

// the slot of a new shape: a free one, or the least recently used entry that has no sample in flight
int make_room(Tuner &t) {
  if (t.entries.size() < TUNE_MAX_SHAPES) {
    t.entries.emplace_back();
    return (int)t.entries.size() - 1;
  }
  int victim = -1;
  for (size_t i = 0; i < t.entries.size(); i++)
    if (t.entries[i].inflight == 0 && (victim < 0 || t.entries[i].last_use < t.entries[(size_t)victim].last_use)) victim = (int)i;
  if (victim >= 0) {
    t.index.erase(t.entries[(size_t)victim].key);
    t.entries[(size_t)victim] = Entry();
  }
  return victim;
}

} // namespace

TuneTicket tuner_begin(Tuner *&tp, int device, const TuneShape &shape, unsigned nkb, hipStream_t stream) {
  TuneTicket tk;
  policy_override(-1);
  const Config cfg = config();
  const Switches sw = switches();
  if (!sw.on || nkb > TUNE_MAX_KB || cfg.forced_kernel() || cfg.paired_tile >= 0 || cfg.k64_tile >= 0 || cfg.k64_breg >= 0 || cfg.wide_grid > 0 ||
      cfg.wide_static || cfg.wide_small_rows >= 0 || cfg.xcds > 0)
    return tk;
  if (!tp) tp = new Tuner();
  Tuner &t = *tp;
  collect(t);
  const Key key{shape.S, shape.op_a, shape.op_b, shape.beta_nonzero ? 1 : 0, shape.m, shape.n, shape.k};
  int idx = -1;
  auto it = t.index.find(key);
  if (it != t.index.end()) idx = it->second;
  if (idx < 0) {
    PassTraits tr;
    if (!slice_gemm_traits(shape.S, 0, &tr)) return tk;
    idx = make_room(t);
    if (idx < 0) return tk;
    PolicyInput in;
    in.M = (uint32_t)shape.m;
    in.N = (uint32_t)shape.n;
    in.nkb = nkb;
    in.batch = 1;
    const Prediction r = policy_predict(tr, in, topology(device), cfg);
    Entry &e = t.entries[(size_t)idx];
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
    t.index.emplace(key, idx);
  }
  Entry &e = t.entries[(size_t)idx];
  e.last_use = ++t.tick;
  if (e.seen < (1 << 30)) e.seen++;
  if (e.ncand > 1 && !e.decided && e.measurements == 0 && e.issued == 0 && e.seen < sw.first_call) {
    return tk; // not worth a measurement yet: no override (the model's pick), no event
  }
  if (e.decided && e.ncand > 1 && ++e.calls >= e.recheck_after) {
    e.decided = false; // (nothing of this entry is in flight: a decision needs every asked-for sample collected)
    e.issued = 0;
    for (int c = 0; c < e.ncand; c++) e.samples[c] = 0;
  }
  if (e.decided) {
    policy_override(e.winner);
    return tk;
  }
  if (e.issued >= TUNE_ROUNDS * e.ncand || t.pending.size() >= TUNE_MAX_PENDING || !fill_pool(t) || t.pool.size() < 2) {
    policy_override(e.winner >= 0 ? e.winner : e.slot[0]); // everything asked for is in flight: the last decision (or the model's pick)
    return tk;                                              // until the times are in
  }
  // round r times every candidate once, starting with candidate r (the first sample of a measurement is the model's pick)
  const int round = e.issued / e.ncand, c = (e.issued % e.ncand + round) % e.ncand;
  hipEvent_t ev0 = t.pool.back();
  t.pool.pop_back();
  hipEvent_t ev1 = t.pool.back();
  t.pool.pop_back();
  if (hipEventRecord(ev0, stream) != hipSuccess) {
    (void)hipGetLastError();
    t.pool.push_back(ev0);
    t.pool.push_back(ev1);
    policy_override(e.winner >= 0 ? e.winner : e.slot[0]);
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
  return tk;
}

void tuner_end(Tuner *tp, TuneTicket &tk, bool ok) {
  policy_override(-1);
  if (tk.entry < 0 || !tp) return;
  Tuner &t = *tp;
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

void tuner_forget(Tuner *&tp) {
  if (!tp) return;
  for (Pending &p : tp->pending) {
    (void)hipEventDestroy(p.start);
    (void)hipEventDestroy(p.stop);
  }
  for (hipEvent_t e : tp->pool) (void)hipEventDestroy(e);
  (void)hipGetLastError();
  delete tp;
  tp = nullptr;
}

int tuner_state(Tuner *tp, int S, int op_a, int op_b, int beta_nonzero, size_t m, size_t n, size_t k, int out[4]) {
  if (!tp) return -1;
  collect(*tp);
  const Entry *hit = nullptr;
  for (const Entry &e : tp->entries) {
    const Key &x = e.key;
    if (e.ncand == 0 || x.S != S || x.m != m || x.n != n || x.k != k) continue;
    if (op_a >= 0 && (x.op_a != op_a || x.op_b != op_b || x.beta_nz != (beta_nonzero ? 1 : 0))) continue;
    if (!hit || e.last_use > hit->last_use) hit = &e;
  }
  if (!hit) return -1;
  if (out) {
    out[0] = hit->decided ? hit->winner : -1;
    out[1] = hit->ncand;
    out[2] = hit->seen;
    out[3] = hit->measurements;
  }
  return hit->decided ? 1 : 0;
}

} // namespace ozhip
