// Probe (run on the GPU box): where do the two wavefronts of a 128-thread workgroup land?  Prints, per SIMD of a CU, how
// many "wave 0" (the inflate kernel's decoders) and "wave 1" (copiers) it holds when 8 such workgroups share a CU.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/wave_placement.hip -o /tmp/wave_placement && /tmp/wave_placement
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <map>
__global__ __launch_bounds__(128) void probe(uint32_t *out, int spin) {
  __shared__ uint32_t pad[19728 / 4];
  uint32_t hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  uint32_t xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  pad[threadIdx.x] = hw;
  // stay resident for a while so that the chip fills up like the real kernel
  uint64_t t0 = clock64();
  while (clock64() - t0 < (uint64_t)spin) __builtin_amdgcn_s_sleep(8);
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 2 + threadIdx.x / 64) * 2] = hw;
    out[(blockIdx.x * 2 + threadIdx.x / 64) * 2 + 1] = xcc + pad[0] * 0;
  }
}
int main() {
  const int n = 2048;
  uint32_t *d;
  hipMalloc(&d, n * 4 * 4);
  hipLaunchKernelGGL(probe, dim3(n), dim3(128), 0, 0, d, 2000000);
  hipDeviceSynchronize();
  std::vector<uint32_t> h(n * 4);
  hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
  // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
  std::map<uint32_t, std::vector<int>> cu;  // key: xcc, se, sh, cu -> counts [simd][wave]
  int same = 0;
  for (int b = 0; b < n; b++) {
    uint32_t simd[2];
    for (int w = 0; w < 2; w++) {
      const uint32_t hw = h[(b * 2 + w) * 2], xcc = h[(b * 2 + w) * 2 + 1] & 15;
      simd[w] = (hw >> 4) & 3;
      const uint32_t key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
      auto &v = cu[key];
      if (v.empty()) v.assign(8, 0);
      v[simd[w] * 2 + w]++;
    }
    same += simd[0] == simd[1];
  }
  printf("CUs seen: %zu, workgroups with both waves on one SIMD: %d of %d\n", cu.size(), same, n);
  int shown = 0, hist[9][2] = {{0}};
  for (auto &kv : cu) {
    if (shown++ < 6) printf("cu %06x: simd0 w0/w1 %d/%d  simd1 %d/%d  simd2 %d/%d  simd3 %d/%d\n", kv.first, kv.second[0], kv.second[1], kv.second[2],
                            kv.second[3], kv.second[4], kv.second[5], kv.second[6], kv.second[7]);
    for (int s = 0; s < 4; s++) {
      hist[kv.second[s * 2] > 8 ? 8 : kv.second[s * 2]][0]++;
      hist[kv.second[s * 2 + 1] > 8 ? 8 : kv.second[s * 2 + 1]][1]++;
    }
  }
  for (int k = 0; k <= 8; k++) printf("SIMDs holding %d decoders: %d, %d copiers: %d\n", k, hist[k][0], k, hist[k][1]);
  return 0;
}
