// valu_rate.hip — issue-rate / latency probe for the integer VALU ops and LDS reads the inflate
// slot body is made of (gfx950).  hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP 256
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *out, uint64_t *cyc, uint32_t seed) {
  __shared__ uint32_t lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (i * 2654435761u + seed) & 2047;
  __syncthreads();
  uint32_t a = threadIdx.x + seed, b = seed * 3 + 1, c = 7, d = 11, e = 13, f = 17, g = 19, h = 23;
  uint64_t t0 = clock64();
  for (int it = 0; it < 64; it++) {
    if (MODE == 0) {  // 8 independent v_add_u32 chains, 256 instructions
#pragma unroll
      for (int r = 0; r < REP / 8; r++)
        asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                     "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(seed));
    } else if (MODE == 1) {  // one dependent chain, 256 instructions
#pragma unroll
      for (int r = 0; r < REP / 8; r++)
        asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                     "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                     : "+v"(a) : "s"(seed));
    } else if (MODE == 2) {  // dependent LDS pointer chase, 32 reads
#pragma unroll
      for (int r = 0; r < REP / 8; r++) a = lds[a & 2047];
    } else if (MODE == 3) {  // alignbit / bfe / cndmask / lshl_add / and, 8 independent, 256 instructions
#pragma unroll
      for (int r = 0; r < REP / 8; r++)
        asm volatile("v_alignbit_b32 %0, %0, %1, %2\n v_bfe_u32 %1, %1, 3, 9\n v_cndmask_b32 %2, %2, %3, vcc\n v_lshl_add_u32 %3, %3, 2, %4\n"
                     "v_and_b32 %4, 0xffff, %4\n v_lshrrev_b32 %5, 5, %5\n v_cmp_lt_u32 vcc, %6, %7\n v_lshlrev_b32 %7, 1, %7\n"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : : "vcc");
    } else if (MODE == 4) {  // LDS chase + 16 independent VALU per step
#pragma unroll
      for (int r = 0; r < REP / 16; r++) {
        a = lds[a & 2047];
        asm volatile("v_add_u32 %0, %0, %7\n v_add_u32 %1, %1, %7\n v_add_u32 %2, %2, %7\n v_add_u32 %3, %3, %7\n"
                     "v_add_u32 %4, %4, %7\n v_add_u32 %5, %5, %7\n v_add_u32 %6, %6, %7\n v_add_u32 %0, %0, %7\n"
                     "v_add_u32 %0, %0, %7\n v_add_u32 %1, %1, %7\n v_add_u32 %2, %2, %7\n v_add_u32 %3, %3, %7\n"
                     "v_add_u32 %4, %4, %7\n v_add_u32 %5, %5, %7\n v_add_u32 %6, %6, %7\n v_add_u32 %0, %0, %7\n"
                     : "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(seed));
      }
    } else if (MODE == 5) {  // v_cmp + s_and_saveexec-like SALU mix: 128 VALU + 128 SALU
#pragma unroll
      for (int r = 0; r < REP / 8; r++)
        asm volatile("v_add_u32 %0, %0, %4\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, %4\n s_and_b32 s21, s21, s20\n"
                     "v_add_u32 %2, %2, %4\n s_lshl_b32 s22, s20, 1\n v_add_u32 %3, %3, %4\n s_or_b32 s23, s22, s21\n"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(seed) : "s20", "s21", "s22", "s23", "scc");
    }
  }
  uint64_t t1 = clock64();
  out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + g + h;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int waves_per_simd, double ops_per_it) {
  int nblk = 256 * 4 * waves_per_simd;
  uint32_t *out; uint64_t *cyc;
  hipMalloc(&out, nblk * 64 * 4); hipMalloc(&cyc, nblk * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<nblk, 64>>>(out, cyc, 12345);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<nblk, 64>>>(out, cyc, 12345);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> h(nblk); hipMemcpy(h.data(), cyc, nblk * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= nblk;
  printf("%-28s waves/SIMD %d: %8.0f clk per wave for %5.0f ops -> %.2f clk/op/wave, %.2f clk/op/SIMD ; kernel %.3f ms\n", name, waves_per_simd, avg,
         ops_per_it * 64, avg / (ops_per_it * 64), avg / (ops_per_it * 64) / waves_per_simd, ms);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("8 indep v_add chains", w, 256);
    run<1>("dependent v_add chain", w, 256);
    run<2>("LDS dependent chase", w, 32);
    run<3>("alignbit/bfe/cndmask/.. mix", w, 256);
    run<4>("LDS chase + 16 VALU (x16)", w, 16);
    run<5>("VALU+SALU interleaved (256)", w, 256);
  }
  return 0;
}
