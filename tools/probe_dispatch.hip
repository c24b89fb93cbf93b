// tools/probe_dispatch.hip — where does the hardware put workgroup b?  (development probe)
// 1024 workgroups of 256 threads with 72 KiB LDS each (=> 2 per CU, like slice_gemm_kernel<9>), each
// records XCC_ID, HW_ID and its start time, then spins ~100 us so that the first 512 are co-resident.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_dispatch.hip -o tools/bin/probe_dispatch
#include <hip/hip_runtime.h>

#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) void probe(unsigned *out, unsigned long long spin) {
  extern __shared__ char smem[];
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[blockIdx.x * 4 + 0] = xcc;
    out[blockIdx.x * 4 + 1] = hw;
    out[blockIdx.x * 4 + 2] = (unsigned)(wall_clock64() & 0xffffffffu);
    smem[0] = 1;
  }
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(10);
}

int main() {
  const int nb = 1024;
  unsigned *d;
  hipMalloc(&d, nb * 16);
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 72 * 1024, 0, d, 10000ull); // 100 MHz wall clock: 100 us
  std::vector<unsigned> h(nb * 4);
  hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
  // gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0]
  std::map<unsigned, std::vector<int>> by_cu;
  unsigned t0 = h[2];
  for (int b = 0; b < nb; b++) t0 = h[b * 4 + 2] < t0 ? h[b * 4 + 2] : t0;
  for (int b = 0; b < nb; b++) {
    const unsigned xcc = h[b * 4] & 0xf, hw = h[b * 4 + 1];
    const unsigned key = (xcc << 16) | (hw & 0xff00);
    by_cu[key].push_back(b);
    if (b < 24 || (b >= 512 && b < 520))
      std::printf("b=%4d xcc=%u hw=0x%08x cu=%u sh=%u se=%u simd=%u t=%u\n", b, xcc, hw, (hw >> 8) & 0xf, (hw >> 12) & 1,
                  (hw >> 13) & 7, (hw >> 4) & 3, h[b * 4 + 2] - t0);
  }
  std::printf("distinct (xcc,se,sh,cu) keys: %zu\n", by_cu.size());
  int shown = 0;
  for (auto &kv : by_cu) {
    if (shown++ >= 12) break;
    std::printf("key %06x:", kv.first);
    for (int b : kv.second) std::printf(" %d", b);
    std::printf("\n");
  }
  // histogram of id distance between the first two workgroups of a CU
  std::map<int, int> hist;
  for (auto &kv : by_cu)
    if (kv.second.size() >= 2) hist[kv.second[1] - kv.second[0]]++;
  for (auto &kv : hist) std::printf("distance %d: %d CUs\n", kv.first, kv.second);
  return 0;
}
