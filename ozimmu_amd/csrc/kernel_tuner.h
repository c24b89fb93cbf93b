// kernel_tuner.h — measured kernel choice for shapes a handle keeps calling (round 5; made safe for real call streams in round 6).
//
// The cost model of kernel_policy.cpp predicts a kernel's time within ~11 % rms; where two or three kernels are predicted
// within that band (k loops of <= 16 k-blocks under large outputs: 8192^2 x 256 runs 5 % faster on the wide tile than on the
// 64 x 64 tiles the model picks, 16384^2 x 256: 8 %) no closed-form term separates them (profiles/r5_policy/second_fit_r5.md).
// Every kernel returns the SAME bits (tests/test_gpu_forced_kernels.py), so the choice can simply be measured.
//
// A call stream decides what is worth measuring.  A shape is (mode, op_A, op_B, m, n, k, beta != 0): the split in front of the
// GEMM is part of every sample and its cost depends on the operand layouts, beta != 0 adds a read of C to the epilogue.  The
// first TUNE_FIRST_CALL - 1 calls of a shape run what the model picks, untimed: a factorisation's shrinking trailing matrix
// (every shape seen once or a few times) never pays for an exploration whose result it would not live to use.  From the
// TUNE_FIRST_CALL-th call on the kernels the model predicts within 25 % (k loops of <= 16 k-blocks; 12 % beyond) of its best
// run in turn (the model's own pick first; four rounds, the first one thrown away), each whole call bracketed by two events
// on the caller's stream; later calls of the shape collect the finished pairs WITHOUT waiting (hipEventQuery) and, once the
// rounds are in, keep the kernel with the smallest median - the model's pick unless another beats it by TUNE_MARGIN and in
// most rounds.  A shape still in use 64 calls later is measured again (the part is warm by then), then 512, 4096, ... calls
// after that: no measurement stands for ever (a sample is a whole call's time on the caller's stream, and whatever ran on the
// device's other streams at that moment is in it), and the cost of looking again falls below 0.1 % of the calls it serves.
// No host synchronisation, no extra launch.  The table keeps the TUNE_MAX_SHAPES most recently used shapes.
//
// The table belongs to the handle and every function here runs under the handle's lock (api.cpp): no process-wide lock.
// Not tuned: forced kernels and the development switches, batches, ZGEMM products, K > 2048, calls predicted under 100 us,
// two-pass modes, the stage timer, the test hooks, and every call of a handle that has ever been captured into a graph
// (hipEventQuery / hipEventRecord next to a capture in flight on another thread are not something to find out in
// production: INTEGRATION.md).  OZIMMU_HIP_AUTOTUNE=0 switches it off; OZIMMU_HIP_AUTOTUNE_AFTER=n moves the first measured
// call (default 8; tests).  The reference has no counterpart (cuBLAS plans its own kernels, src/gemm.cu:315-329).
#pragma once
#include <cstddef>

#include <hip/hip_runtime_api.h>

namespace ozhip {

// kernel_policy.cpp: while >= 0, policy_predict on this thread takes that prediction slot (0 .. POLICY_KERNELS-1) wherever it
// is eligible, instead of the minimum.  Set for the duration of one tuned call.
void policy_override(int slot);

struct Tuner; // the table of one handle (ozimmu_hip_handle::tuner), created by the first eligible call

struct TuneShape {
  int S = 0;
  int op_a = 0, op_b = 0; // ozimmu_operation_t
  bool beta_nonzero = false;
  size_t m = 0, n = 0, k = 0;
};

struct TuneTicket {
  int entry = -1, cand = -1; // >= 0: a sample is being taken
  bool counts = false;       // (round 0 of a measurement is a warm-up: timed like the others, not kept)
  hipEvent_t start = nullptr, stop = nullptr;
  hipStream_t stream = nullptr;
};

// before the first launch of a call (the handle's device current, its lock held): collects finished samples, sets the
// override for this call, records the start event when the call is a sample
TuneTicket tuner_begin(Tuner *&t, int device, const TuneShape &shape, unsigned nkb, hipStream_t stream);
// after the last launch: records the stop event (ok) or drops the sample; clears the override
void tuner_end(Tuner *t, TuneTicket &tk, bool ok);
// ozimmu_hip_destroy: events and table of that handle
void tuner_forget(Tuner *&t);
// diagnostics (tests): state of the most recently used entry for (S, m, n, k) [and, when op_a >= 0, exactly that layout /
// beta class]: -1 unknown, 0 counting sightings or measuring, 1 decided; out = {decided prediction slot or -1, candidates,
// calls seen, measurements finished}
int tuner_state(Tuner *t, int S, int op_a, int op_b, int beta_nonzero, size_t m, size_t n, size_t k, int out[4]);

} // namespace ozhip
