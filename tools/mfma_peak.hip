// Sustained-rate probe for v_mfma_f32_32x32x2_f32 on gfx950: registers only (no LDS, no memory), N independent
// accumulators per wave.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256, 2) void probe(float *out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int t = 0; t < NACC; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int t = 0; t < NACC; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 123.456f) out[0] = s;
}

template <int NACC>
void run(const char *name, int blocks) {
  float *d;
  hipMalloc(&d, 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * NACC * 4096.0;
    printf("%s blocks=%d rep=%d  %.3f ms  %.1f TFLOP/s\n", name, blocks, rep, ms, flop / ms / 1e9);
  }
  hipFree(d);
}

int main() {
  run<4>("acc4 ", 512);
  run<4>("acc4 ", 256);
  run<8>("acc8 ", 512);
  run<4>("acc4 long", 2048);
  return 0;
}
