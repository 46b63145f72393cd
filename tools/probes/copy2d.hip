// Probe (GPU box): strided host<->device copies against contiguous ones, pinned memory.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/copy2d.hip -o /tmp/copy2d && /tmp/copy2d
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <chrono>
#include <string.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin(uint64_t cycles, uint32_t *sink) {
  uint64_t t0 = wall_clock64();
  uint32_t x = threadIdx.x;
  while (wall_clock64() - t0 < cycles) x = x * 1664525u + 1013904223u;
  if (x == 12345) *sink = x;
}
int main() {
  const size_t rows = 4096, pitch = 1 << 20, w = 256 << 10;
  uint8_t *h, *d;
  hipHostMalloc((void **)&h, rows * pitch, hipHostMallocDefault);
  hipMalloc((void **)&d, rows * pitch);
  memset(h, 1, rows * pitch);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now();
    hipMemcpyAsync(d, h, rows * w, hipMemcpyHostToDevice, s);
    hipStreamSynchronize(s);
    double t1 = now();
    hipMemcpy2DAsync(d, pitch, h, pitch, w, rows, hipMemcpyHostToDevice, s);
    hipStreamSynchronize(s);
    double t2 = now();
    hipMemcpy2DAsync(h, pitch, d, pitch, w, rows, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    double t3 = now();
    hipMemcpyAsync(h, d, rows * w, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    double t4 = now();
    for (size_t r = 0; r < rows; r++) hipMemcpyAsync(d + r * pitch, h + r * pitch, w, hipMemcpyHostToDevice, s);
    double t5 = now();
    hipStreamSynchronize(s);
    double t6 = now();
    printf("1 GiB: 1D h2d %.1f ms, 2D h2d %.1f ms, 2D d2h %.1f ms, 1D d2h %.1f ms, 4096 row copies h2d: submit %.1f ms, done %.1f ms\n", (t1 - t0) * 1e3,
           (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t6 - t4) * 1e3);
  }
  // the same copies under a kernel that keeps every CU busy (16 wavefronts per CU, 40 ms)
  hipStream_t k, k2;
  hipStreamCreateWithFlags(&k, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&k2, hipStreamNonBlocking);
  uint32_t *sink;
  hipMalloc((void **)&sink, 4);
  for (int mode = 0; mode < 6; mode++) {
    hipLaunchKernelGGL(spin, dim3(4096), dim3(64), 0, k, (uint64_t)100000000 * 40 / 1000, sink);  // 100 MHz clock64
    double t0 = now();
    if (mode == 0) hipMemcpyAsync(d, h, rows * w, hipMemcpyHostToDevice, s);
    if (mode == 1) hipMemcpy2DAsync(d, pitch, h, pitch, w, rows, hipMemcpyHostToDevice, s);
    if (mode == 2) hipMemcpy2DAsync(h, pitch, d, pitch, w, rows, hipMemcpyDeviceToHost, s);
    if (mode == 3) hipMemcpyAsync(h, d, rows * w, hipMemcpyDeviceToHost, s);
    if (mode == 4) { hipMemcpy2DAsync(d, pitch, h, pitch, w, rows, hipMemcpyHostToDevice, s); hipMemcpy2DAsync(h + w, pitch, d + w, pitch, w, rows, hipMemcpyDeviceToHost, k2); hipStreamSynchronize(k2); }
    if (mode == 5) { hipMemcpyAsync(d, h, rows * w, hipMemcpyHostToDevice, s); hipMemcpyAsync(h + rows * w, d + rows * w, rows * w, hipMemcpyDeviceToHost, k2); hipStreamSynchronize(k2); }
    hipStreamSynchronize(s);
    double t1 = now();
    hipStreamSynchronize(k);
    double t2 = now();
    printf("under a busy chip, mode %d (0 1D h2d, 1 2D h2d, 2 2D d2h, 3 1D d2h, 4 2D both ways, 5 1D both ways): copy %.1f ms, kernel done at %.1f ms\n", mode, (t1 - t0) * 1e3, (t2 - t0) * 1e3);
  }
  return 0;
}
