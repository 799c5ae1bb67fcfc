// How much single-issue work hides behind v_mfma_f32_32x32x2_f32 on gfx950?  Registers only: 8 independent accumulators per
// wave, after every MFMA NV plain v_fma_f32 (independent chains, or ONE dependent chain) and NL ds_read_b128.
// One wave per SIMD (256 threads, launch_bounds(256, 1)) or two (512 threads).  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int THREADS, int NV, int DEP, int NL>
__global__ __launch_bounds__(THREADS, THREADS / 256) void probe(float *out, int iters, float a0, float b0) {
  __shared__ f32x4 lds[1024];
  lds[threadIdx.x] = f32x4{a0, b0, a0, b0};
  __syncthreads();
  f32x16 acc[8];
  for (int t = 0; t < 8; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = a0 * j;
  f32x4 l4 = {0.f, 0.f, 0.f, 0.f};
  typedef const volatile f32x4 __attribute__((address_space(3))) *lp4;
  const f32x4 *lp = lds + (threadIdx.x & 255);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a), "v"(b));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (DEP)
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[0]) : "v"(a), "v"(b));
        else
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[j & 7]) : "v"(a), "v"(b));
      }
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        f32x4 v = *(lp4)(lp + 256 * j);
        l4 += v;
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s = l4[0] + l4[1] + l4[2] + l4[3];
  for (int t = 0; t < 8; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int j = 0; j < 8; ++j) s += x[j];
  if (s == 123.456f) out[0] = s;
}

template <int THREADS, int NV, int DEP, int NL>
void run() {
  float *d;
  hipMalloc(&d, 4);
  const int iters = 4000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<THREADS, NV, DEP, NL>), dim3(blocks), dim3(THREADS), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double nm = (double)iters * 8 * (THREADS / 256);             // MFMAs per SIMD
  const double flop = (double)blocks * 4 * nm * 4096.0;
  printf("waves/SIMD %d  valu/gap %2d %s  ds_read_b128/gap %d : %7.3f ms  %6.1f ns/MFMA/SIMD  %6.1f TFLOP/s\n", THREADS / 256, NV,
         DEP ? "dep  " : "indep", NL, best, best * 1e6 / nm, flop / best / 1e9);
  hipFree(d);
}

int main() {
  run<256, 0, 0, 0>();
  run<256, 2, 0, 0>();
  run<256, 4, 0, 0>();
  run<256, 6, 0, 0>();
  run<256, 8, 0, 0>();
  run<256, 10, 0, 0>();
  run<256, 12, 0, 0>();
  run<256, 16, 0, 0>();
  run<256, 4, 1, 0>();
  run<256, 8, 1, 0>();
  run<256, 0, 0, 1>();
  run<256, 0, 0, 2>();
  run<256, 6, 0, 1>();
  run<512, 0, 0, 0>();
  run<512, 4, 0, 0>();
  run<512, 8, 0, 0>();
  run<512, 12, 0, 0>();
  run<512, 6, 0, 1>();
  return 0;
}
