// What a three-limb Winograd F(4x4,3x3) main loop could cost on gfx950, measured on registers only.
//   (1) cycles per v_mfma_f32_32x32x16_bf16 and per the legacy K=8 form v_mfma_f32_32x32x8_bf16_1k (2-register operands);
//   (2) FINE interleave: after every MFMA, NV independent plain VALU ops (the transform + limb split mix: fma / and / sub / perm),
//       one or two waves per SIMD - how many VALU ops a bf16 MFMA hides (the f32 MFMA hides none: tools/mfma_valu_overlap.hip);
//   (3) PHASES: NT VALU ops then NM MFMAs per iteration (a wave that cannot hold two V sets in registers), two waves per SIMD
//       free-running, in lockstep behind a barrier, or ping-pong (one half of the workgroup transforms while the other multiplies).
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_bf16_valu_overlap.hip -o /tmp/ovl16 && /tmp/ovl16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int K8>
__device__ __forceinline__ void mma(f32x16 &acc, const s16x8 &a, const s16x8 &b) {
  if (K8) {
    const s16x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
    asm volatile("v_mfma_f32_32x32x8_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a4), "v"(b4));
  } else {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  }
}

// the limb-split / transform instruction mix: 4 fma, 2 and, 2 sub, 1 perm per 9 (all independent chains over x[0..7])
__device__ __forceinline__ void valu(float (&x)[8], float a, float b, int j) {
  const int m = j % 9;
  if (m < 4)
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[j & 7]) : "v"(a), "v"(b));
  else if (m < 6)
    asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x[j & 7]));
  else if (m < 8)
    asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[j & 7]) : "v"(b));
  else
    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[j & 7]) : "v"(a), "v"(b));
}

// MODE 0: fine interleave (NV VALU after every MFMA, NACC accumulators round-robin)
// MODE 1: phases, free-running;  MODE 2: phases, __syncthreads() after each phase pair (lockstep)
// MODE 3: ping-pong: waves 0-3 do T then M, waves 4-7 do M then T, a barrier between the half iterations
template <int THREADS, int K8, int MODE, int NV, int NM, int NACC>
__global__ __launch_bounds__(THREADS, THREADS / 256) void probe(float *out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int t = 0; t < NACC; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = a0 * j;
  s16x8 fa, fb;
  for (int j = 0; j < 8; ++j) {
    fa[j] = (short)(threadIdx.x + j);
    fb[j] = (short)(threadIdx.x * 3 + j);
  }
  const int late = (THREADS == 512 && threadIdx.x >= 256) ? 1 : 0;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int t = 0; t < NM; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        mma<K8>(acc[t % NACC], fa, fb);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NV; ++j) valu(x, a, b, j);
      }
    } else {
      auto T = [&]() {                      // chunks of 65 (a full unroll of 390+ ops is refused and x[] would go to scratch)
        static_assert(MODE == 0 || NV % 65 == 0, "phase modes: NV in multiples of 65");
#pragma unroll 1
        for (int c = 0; c < NV / 65; ++c) {
#pragma unroll
          for (int j = 0; j < 65; ++j) valu(x, a, b, j);
        }
      };
      auto M = [&]() {
#pragma unroll
        for (int t = 0; t < NM; ++t) mma<K8>(acc[t % NACC], fa, fb);
      };
      if (MODE == 3) {
        if (late) M(); else T();
        __builtin_amdgcn_s_barrier();
        if (late) T(); else M();
        __builtin_amdgcn_s_barrier();
      } else {
        T();
        M();
        if (MODE == 2) __builtin_amdgcn_s_barrier();
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s = 0.f;
  for (int t = 0; t < NACC; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int j = 0; j < 8; ++j) s += x[j];
  if (s == 123.456f) out[0] = s;
}

template <int THREADS, int K8, int MODE, int NV, int NM, int NACC>
void run(const char *what) {
  float *d;
  hipMalloc(&d, 4);
  const int iters = MODE == 0 ? 2000 : 400, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<THREADS, K8, MODE, NV, NM, NACC>), dim3(blocks), dim3(THREADS), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double nm = (double)iters * NM * (THREADS / 256);             // MFMAs per SIMD
  const double macs = K8 ? 8192.0 : 16384.0;
  const double nv = MODE == 0 ? (double)NV : (double)NV / NM;
  printf("%-28s K=%2d waves/SIMD %d  VALU/MFMA %5.1f  acc %d : %8.3f ms  %6.1f ns/MFMA/SIMD  %7.1f TFLOP/s bf16\n", what, K8 ? 8 : 16,
         THREADS / 256, nv, NACC, best, best * 1e6 / nm, 2.0 * blocks * 4 * nm * macs / best / 1e9);
  hipFree(d);
}

int main() {
  // (1) bare MFMA rate
  run<512, 0, 0, 0, 8, 8>("bare");
  run<512, 0, 0, 14, 8, 8>("fine");
  run<512, 0, 0, 7, 8, 8>("fine");
  // (3) phases, K=16: 780 VALU + 54 MFMAs = one 16-channel step of a (position block, tile half) wave that transforms all 8 of its
  // lane's channels; 390 + 54 = the transform shared by two k blocks (each wave half the channels, V through LDS)
  run<512, 0, 1, 780, 54, 2>("phases free");
  run<512, 0, 2, 780, 54, 2>("phases lockstep");
  run<512, 0, 3, 780, 54, 2>("phases ping-pong");
  run<256, 0, 1, 780, 54, 2>("phases, one wave");
  run<512, 0, 1, 390, 54, 2>("phases free, half VALU");
  run<512, 0, 2, 390, 54, 2>("phases lockstep, half VALU");
  run<512, 0, 3, 390, 54, 2>("ping-pong, half VALU");
  run<256, 0, 1, 390, 54, 2>("one wave, half VALU");
  run<512, 0, 1, 195, 54, 2>("phases free, quarter VALU");
  run<512, 0, 3, 195, 54, 2>("ping-pong, quarter VALU");
  return 0;
}
