// topology.hip — one-time device probes behind topology.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <mutex>

#include "config.h"
#include "topology.h"

namespace ozhip {

typedef int tv4i __attribute__((ext_vector_type(4)));
typedef int tv16i __attribute__((ext_vector_type(16)));

// every workgroup reports the id of the XCD (accelerator complex die) it runs on: max + 1 = XCDs of this device / partition
__global__ void xcc_probe_kernel(unsigned *out) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) atomicMax(out, (x & 0xfu) + 1u);
}

// 12 independent accumulators, operands with every bit random (the regime of real slices: the clock under it is what the
// policy needs), one wave per SIMD
__global__ __launch_bounds__(256) void mfma_calibration_kernel(int iters, int *sink) {
  tv4i a[3], b[3];
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int i = 0; i < 3; i++)
    for (int c = 0; c < 4; c++) {
      x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
      a[i][c] = (int)x;
      b[i][c] = (int)(x * 2654435761u);
    }
  tv16i acc[12];
  for (int i = 0; i < 12; i++)
    for (int r = 0; r < 16; r++) acc[i][r] = 0;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 12; i++) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i % 3], b[(i + 1) % 3], acc[i], 0, 0, 0);
  int s = 0;
  for (int i = 0; i < 12; i++) s += acc[i][i];
  if (s == 0x12345678) sink[0] = s;
}

// bench.py's roofline.frac_of_measured_mfma_ceiling: what the matrix pipe alone sustains on THIS part, now -
// v_mfma_i32_16x16x64_i8 (the headline kernel's instruction), 24 independent accumulators, every operand bit random, four
// waves per SIMD, no memory traffic.  The part is power limited under this load (profiles/r3_power_bound.md: 3 948 TOPS =
// 0.784 of the 5 033 nominal), so the ceiling has to be measured in steady state: launches of ~10 ms for `seconds`, the
// last 60 % timed.
__global__ __launch_bounds__(256) void mfma_ceiling_kernel(int iters, int *sink) {
  tv4i a[9], b[9];
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 777u;
  for (int i = 0; i < 9; i++) {
    for (int c = 0; c < 4; c++) {
      x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
      a[i][c] = (int)x;
      b[i][c] = (int)(x * 2654435761u);
    }
    asm volatile("" : "+v"(a[i]), "+v"(b[i]));
  }
  tv4i acc[24];
  for (int i = 0; i < 24; i++)
    for (int r = 0; r < 4; r++) acc[i][r] = 0;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 24; i++) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i % 9], b[(i * 5 + 1) % 9], acc[i], 0, 0, 0);
  int s = 0;
  for (int i = 0; i < 24; i++)
    for (int r = 0; r < 4; r++) s += acc[i][r];
  if (s == 0x12345678) sink[0] = s;
}

static std::mutex g_mtx;
static Topology g_topo[64];

static bool valid(int dev) { return dev >= 0 && dev < 64; }

Topology topology(int device) {
  Topology t;
  if (valid(device)) {
    std::lock_guard<std::mutex> lock(g_mtx);
    t = g_topo[device];
  }
  const int forced = config().xcds;
  if (forced > 0) t.xcds = std::min(forced, 16);
  return t;
}

bool topology_slot(int device, Topology *inout, bool set) {
  if (!valid(device) || !inout) return false;
  std::lock_guard<std::mutex> lock(g_mtx);
  if (set) g_topo[device] = *inout;
  else *inout = g_topo[device];
  return true;
}

void probe_topology(int dev) {
  if (!valid(dev)) return;
  {
    std::lock_guard<std::mutex> lock(g_mtx);
    if (g_topo[dev].probed) return;
  }
  Topology t;
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) t.cus = v;
  unsigned *d = nullptr;
  if (hipMalloc((void **)&d, 256) == hipSuccess) {
    unsigned h = 0;
    if (hipMemset(d, 0, 256) == hipSuccess) {
      hipLaunchKernelGGL(xcc_probe_kernel, dim3(4 * t.cus), dim3(64), 0, 0, d);
      if (hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost) == hipSuccess && h >= 1 && h <= 16) t.xcds = (int)h;
      // sustained MFMA time: one untimed pass (clock ramp), then the FASTEST of three timed passes of ~1 ms (another tenant
      // of the device can only make a pass slower), accepted only within +-25 % of the nominal 32 cycles at 1.65 GHz: the
      // kernel choice must not swing with whatever else ran during this millisecond
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
        const int iters = 4000; // 48 000 MFMAs per wave ~ 0.8 - 1 ms
        hipLaunchKernelGGL(mfma_calibration_kernel, dim3(t.cus), dim3(256), 0, 0, iters, (int *)d);
        double best = 0;
        for (int rep = 0; rep < 3; rep++) {
          hipEventRecord(e0, 0);
          hipLaunchKernelGGL(mfma_calibration_kernel, dim3(t.cus), dim3(256), 0, 0, iters, (int *)d);
          hipEventRecord(e1, 0);
          float ms = 0;
          if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0) {
            const double us = (double)ms * 1e3 / ((double)iters * 12.0);
            if (best == 0 || us < best) best = us;
          }
        }
        t.mfma32_measured_us = best;
        const double nominal = Topology().mfma32_us;
        if (best > 0) t.mfma32_us = std::min(std::max(best, 0.75 * nominal), 1.25 * nominal);
      }
      if (e0) hipEventDestroy(e0);
      if (e1) hipEventDestroy(e1);
    }
    hipFree(d);
  }
  (void)hipGetLastError();
  t.probed = true;
  std::lock_guard<std::mutex> lock(g_mtx);
  g_topo[dev] = t;
}

} // namespace ozhip

extern "C" int ozimmu_hip_mfma_ceiling(int device, double seconds, double *tops) {
  if (!tops || !(seconds > 0) || seconds > 30.0) return 1;
  int cur = 0, cus = 0;
  if (hipGetDevice(&cur) != hipSuccess) return 3;
  if (device < 0) device = cur;
  if (device != cur && hipSetDevice(device) != hipSuccess) return 3;
  int rc = 3;
  int *d = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0 &&
      hipMalloc((void **)&d, 256) == hipSuccess && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
    const int iters = 12000; // ~10 ms per launch
    const int launches = std::max(5, (int)(seconds / 0.010)), untimed = (launches * 2) / 5;
    for (int i = 0; i < launches; i++) {
      if (i == untimed) hipEventRecord(e0, 0);
      hipLaunchKernelGGL(ozhip::mfma_ceiling_kernel, dim3(4 * cus), dim3(256), 0, 0, iters, d);
    }
    hipEventRecord(e1, 0);
    float ms = 0;
    if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0) {
      const double ops = 2.0 * 16384.0 * 24.0 * (double)iters * 4.0 * (4.0 * cus) * (double)(launches - untimed);
      *tops = ops / ((double)ms * 1e-3) / 1e12;
      rc = 0;
    }
  }
  if (e0) hipEventDestroy(e0);
  if (e1) hipEventDestroy(e1);
  if (d) hipFree(d);
  (void)hipGetLastError();
  if (device != cur) hipSetDevice(cur);
  return rc;
}

