// diagnostics.h — process-wide counters behind ozimmu_hip_intercept_stats (include/ozimmu_hip.h): what an end-to-end caller
// under LD_PRELOAD (tools/lu_preload.py) reports next to its residual and wall time.  The reference has no counterpart
// beyond its log lines (src/cublas.cu:153-166).
#pragma once

namespace ozhip {

constexpr int PICK_HIST_SLOTS = 24; // kernel codes of ozimmu_hip_last_kernel: Pick (0..4), + 8 k64 in registers, + 16 one launch

// kernel_policy.cpp: how often each kernel code was launched for the first (or only) diagonal pass of a slice GEMM
void pick_histogram(unsigned long long out[PICK_HIST_SLOTS]);

} // namespace ozhip
