// topology.h — what the launch policy knows about the device it plans for: CU count, XCD count (L2 domains) and the time one
// v_mfma_i32_32x32x32_i8 takes under sustained full-entropy load.  Probed ONCE per device when a handle is created
// (topology.hip: a grid of one-wave workgroups reads HW_REG_XCC_ID; a ~1 ms MFMA loop on every CU is timed with events),
// never inside a stream capture.  Nothing in the kernels or in the tile partition assumes 8 XCDs x 32 CUs: the kernels
// take the XCD count as an argument (SliceGemmArgs::nxcd) and the per-XCD phase / queue lines are sized for MAX_XCDS.
// OZIMMU_HIP_XCDS overrides the XCD count (tests: 1, 2, 4 emulated XCDs give the same bits).
#pragma once
#include <hip/hip_runtime_api.h>

namespace ozhip {

struct Topology {
  int cus = 256;
  int xcds = 8;
  double mfma32_us = 0.0194; // one 32x32x32 INT8 MFMA on one SIMD, sustained (32 cycles at ~1.65 GHz) until calibrated
  double mfma32_measured_us = 0; // what the probe measured (mfma32_us is this value clamped to +-25 % of the nominal one)
  bool probed = false;
};

// Topology of device `device` (defaults until probe_topology() has run for it).  The id is the one the HANDLE was created on
// (handle.h: ozimmu_hip_handle::device) and travels with every launch (kernels.h: SliceGemmArgs::device): nothing on the launch
// path asks the runtime which device happens to be current (a process that drives several GPUs plans every call for the part
// the call runs on, whatever another thread has selected meanwhile).
Topology topology(int device);
// probe `device` if that has not happened yet; it must be the calling thread's current device (device allocations +
// synchronisation: handle creation only)
void probe_topology(int device);
// the slot of `device` as it stands / replaced (tests: fake devices on a box without a GPU; tools)
bool topology_slot(int device, Topology *inout, bool set);

} // namespace ozhip
