// kernel_tuner.h — measured kernel choice for shapes a handle sees again (round 5).
//
// The cost model of kernel_policy.cpp predicts a kernel's time within ~11 % rms; where two or three kernels are predicted
// within that band (k loops of <= 16 k-blocks under large outputs: 8192^2 x 256 runs 5 % faster on the wide tile than on the
// 64 x 64 tiles the model picks, 16384^2 x 256: 8 %) no closed-form term separates them (profiles/r5_policy/second_fit_r5.md).
// Every kernel returns the SAME bits (tests/test_gpu_forced_kernels.py), so the choice can simply be measured: the first
// calls of a (mode, m, n, k) on a handle run the candidates the model predicts within 25 % (k loops of <= 16 k-blocks; 12 % beyond) of its best in turn (the
// model's own pick first; four rounds, the first one thrown away), each whole call bracketed by two events on the caller's
// stream; later calls of the shape collect the finished pairs WITHOUT waiting (hipEventQuery) and, once the rounds are in,
// keep the kernel with the smallest median - the model's pick unless another beats it by TUNE_MARGIN and in most rounds.
// A shape still in use 64 calls later is measured once more (the part is warm by then) and that result stands.  No host synchronisation, no extra launch; a
// shape seen once (HPL's shrinking trailing matrix) runs what the model picks, as before.  Not tuned: forced kernels and the
// development switches, batches, ZGEMM products, K > 2048, calls predicted under 100 us, two-pass modes, captured streams, the stage timer, the test
// hooks.  OZIMMU_HIP_AUTOTUNE=0 switches it off.  The reference has no counterpart (cuBLAS plans its own kernels,
// src/gemm.cu:315-329).
#pragma once
#include <cstddef>

#include <hip/hip_runtime_api.h>

namespace ozhip {

// kernel_policy.cpp: while >= 0, policy_predict on this thread takes that prediction slot (0 .. POLICY_KERNELS-1) wherever it
// is eligible, instead of the minimum.  Set for the duration of one tuned call.
void policy_override(int slot);

struct TuneTicket {
  int entry = -1, cand = -1; // >= 0: a sample is being taken
  bool counts = false;       // (round 0 of a shape is a warm-up: timed like the others, not kept)
  hipEvent_t start = nullptr, stop = nullptr;
  hipStream_t stream = nullptr;
  const void *owner = nullptr;
};

// before the first launch of a call (the handle's device current, its lock held): collects finished samples, sets the
// override for this call, records the start event when the call is a sample
TuneTicket tuner_begin(const void *owner, int device, int S, size_t m, size_t n, size_t k, unsigned nkb, hipStream_t stream);
// after the last launch: records the stop event (ok) or drops the sample; clears the override
void tuner_end(TuneTicket &t, bool ok);
// ozimmu_hip_destroy: events and table of that handle
void tuner_forget(const void *owner);
// diagnostics (tests): state of the entry for a shape: -1 unknown, 0 exploring, 1 decided; *slot = the decided prediction slot
int tuner_state(const void *owner, int S, size_t m, size_t n, size_t k, int *slot, int *candidates);

} // namespace ozhip
