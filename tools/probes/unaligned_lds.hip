// probe: do misaligned LDS / global accesses work on gfx950 (ROCm 7.2)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

__global__ void probe(uint8_t *g, uint32_t *res) {
  __shared__ __align__(16) uint8_t lds[1024];
  uint32_t lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = (uint8_t)(i * 7 + 3);
  __syncthreads();
  uint32_t bad = 0;
  // unaligned 8-byte LDS read at offset lane*9+1
  uint32_t o = lane * 9 + 1;
  uint64_t v;
  __builtin_memcpy(&v, lds + o, 8);
  for (int k = 0; k < 8; k++)
    if (((v >> (8 * k)) & 0xff) != (uint8_t)((o + k) * 7 + 3)) bad |= 1;
  // unaligned 4-byte
  uint32_t v4;
  __builtin_memcpy(&v4, lds + o + 2, 4);
  for (int k = 0; k < 4; k++)
    if (((v4 >> (8 * k)) & 0xff) != (uint8_t)((o + 2 + k) * 7 + 3)) bad |= 2;
  __syncthreads();
  // unaligned 8-byte LDS write
  uint64_t wv = 0x0102030405060708ull + lane;
  __builtin_memcpy(lds + 11 * lane + 3, &wv, 8);
  __syncthreads();
  for (int k = 0; k < 8; k++)
    if (lds[11 * lane + 3 + k] != ((wv >> (8 * k)) & 0xff)) bad |= 4;
  // unaligned global 16B store/load
  uint4 gv = make_uint4(lane, lane + 1, lane + 2, lane + 3);
  __builtin_memcpy(g + 3 + 16 * lane, &gv, 16);
  __threadfence();
  uint4 rv;
  __builtin_memcpy(&rv, g + 3 + 16 * lane, 16);
  if (rv.x != lane || rv.w != lane + 3) bad |= 8;
  res[lane] = bad;
}

int main() {
  uint8_t *g; uint32_t *r; uint32_t h[64];
  hipMalloc(&g, 4096); hipMalloc(&r, 256);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, g, r);
  hipMemcpy(h, r, 256, hipMemcpyDeviceToHost);
  uint32_t all = 0; for (int i = 0; i < 64; i++) all |= h[i];
  printf("unaligned probe: bad mask = %u (0 = all misaligned accesses correct)\n", all);
  return 0;
}
