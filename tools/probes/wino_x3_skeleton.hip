// the MAIN LOOP of the one three-limb F(4x4,3x3) design that fits the register file and the LDS,
// as a performance skeleton (real instruction mix, real LDS / L2 / HBM traffic, random data, no epilogue, no result check).
//
// Why this design (tools/probes/mfma_bf16_valu_overlap.hip, profiles/r6a_*): the legacy K = 8 bf16 MFMA runs at HALF rate on
// gfx950 (16 ns for 8 k, like the K = 16 form), so a lane must hold 8 channels of V limbs per position: 9 positions x 3 limbs x
// 4 registers = 108 beside 144 accumulator registers - the (position block, tile half) wave of conv_wino4.h cannot keep V in
// registers with two waves per SIMD.  V therefore goes through LDS, and then the transform can be SHARED by two k blocks:
//   workgroup = (image, tile half: 32 tiles, 64 output channels): 8 waves = 4 position blocks x 2 k blocks of 32;
//   per 16-channel k-step: T phase - every wave transforms 4 of its lanes' 8 channels for its 9 positions (48 VALU + 10 LDS reads
//   per channel), splits the 9 values x 4 channels into bf16 limbs (RN, exact: 11 VALU per value pair) and writes them into the
//   B-fragment image of LDS (27 x 2 ds_write_b32); barrier; M phase - per position 3 ds_read_b128 (V limbs) + 3 buffer_load_b128
//   (U limbs, L2 -> registers, nothing shares them) + 6 MFMAs; the raw rows of the next k-step travel global -> registers -> LDS
//   during the M phase; barrier.  LDS: V 108 KB + raw rows 45 KB.  7.2 VALU per MFMA (the probe: 22 ns per MFMA in lockstep
//   phases on registers alone, 16.8 bare).
// What is NOT here: the inverse transform, the InstanceNorm (a half-image workgroup cannot fuse it: + one pass of the output),
// correctness.  The number to beat: 0.80 ms per N = 256 launch of wino4_f3x3_kernel WITH both (keep <= 0.62, stop > 0.75).
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize tools/probes/wino_x3_skeleton.hip -o tools/probes/wino_x3_skeleton
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define LDW 40
#define ROWS 18
#define CH_F (ROWS * LDW)                 // floats per staged channel (half image + halo rows)
#define RAW_F (16 * CH_F)                 // 16 channels
#define V_BYTES (36 * 3 * 1024)
#define LDS_BYTES (V_BYTES + RAW_F * 4)

template <int B>
__device__ __forceinline__ void xf(float e0, float e1, float e2, float e3, float e4, float &o0, float &o1, float &o2) {
  if (B == 0) {
    o0 = fmaf(4.f, e0, fmaf(-5.f, e2, e4));
    const float a = fmaf(-4.f, e2, e4), b = fmaf(-4.f, e1, e3);
    o1 = a + b;
    o2 = a - b;
  } else {
    const float c = e3 - e1, f = e2 - e0;
    o0 = fmaf(2.f, f, c);
    o1 = fmaf(-2.f, f, c);
    o2 = fmaf(4.f, e0, fmaf(-5.f, e2, e4));
  }
}

__device__ __forceinline__ unsigned cvt_pk(float a, float b) {      // [bf16(b) | bf16(a)], round to nearest even
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// three exact limbs of the pair (x0, x1): hi + mid + lo = x (conv_types.h: split3), packed per limb
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
  h = cvt_pk(x0, x1);
  const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
  m = cvt_pk(r0, r1);
  const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
  l = cvt_pk(s0, s1);
}

template <int V> struct int_c { static constexpr int value = V; };

// PHASES: 1 = T | barrier | M | barrier (the design); 0 = no T phase (M only: what the multiply side alone costs); 2 = no M phase
template <int PHASES, int UD, int XNT>
__global__ __launch_bounds__(512, 1) void skeleton(const float *X, const unsigned short *U3, float *out, int N) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float *raw = reinterpret_cast<float *>(lds + V_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave & 3, kb = wave >> 2, t = lane & 31, g = lane >> 5, tr = t >> 3, tc = t & 7;
  const int b8 = blockIdx.x & 7, ksg = b8 & 3, th = b8 >> 2, n = blockIdx.x >> 3;
  const __amdgpu_buffer_rsrc_t xrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X + (long)n * 256 * 1024), 0, 256 * 1024 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short *>(U3 + (long)ksg * 16 * 216 * 512), 0, 16 * 216 * 1024, 0x00020000);
  // raw-row staging: 16 channels x 16 rows x 8 segments of 16 B = 4 per thread (+ the two halo rows: threads 0..255 one more)
  f32x4 sreg[5];
  const int srow = (tid >> 3) & 15, sseg = tid & 7, sch = tid >> 7;          // channel sch + 4 i
  const unsigned s_off = (unsigned)((16 * th + srow) * 32 + sseg * 4) * 4u;
  const int s_lds = (srow + 1) * LDW + 1 + sseg * 4;
  const int hrow = (tid >> 3) & 1, hch = (tid >> 4) & 15;                     // halo: row -1 / 16 of the half
  const int hy = 16 * th + (hrow ? 16 : -1);
  const unsigned h_off = (unsigned)(hy * 32 + sseg * 4) * 4u;
  const bool h_in = tid < 256 && hy >= 0 && hy < 32;
  auto load_raw = [&](int step) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      sreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, s_off, (step * 16 + sch + 4 * i) * 4096, XNT ? 2 : 0));
    sreg[4] = h_in ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, h_off, (step * 16 + hch) * 4096, XNT ? 2 : 0))
                   : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto store_raw = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float *d = raw + (sch + 4 * i) * CH_F + s_lds;
      d[0] = sreg[i][0];
      *reinterpret_cast<f32x2 *>(d + 1) = f32x2{sreg[i][1], sreg[i][2]};
      d[3] = sreg[i][3];
    }
    if (tid < 256) {
      float *d = raw + hch * CH_F + (hrow ? 17 : 0) * LDW + 1 + sseg * 4;
      d[0] = sreg[4][0];
      *reinterpret_cast<f32x2 *>(d + 1) = f32x2{sreg[4][1], sreg[4][2]};
      d[3] = sreg[4][3];
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int q = 0; q < 9; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  load_raw(0);
  for (int u = tid; u < LDS_BYTES / 16; u += 512) reinterpret_cast<f32x4 *>(lds)[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  store_raw();
  __syncthreads();

  auto body = [&](auto bi_c, auto bj_c) {
    constexpr int BI = decltype(bi_c)::value, BJ = decltype(bj_c)::value;
    typedef const volatile f32x4 __attribute__((address_space(3))) *lp4;
    typedef const volatile f32x2 __attribute__((address_space(3))) *lp2;
    const float *rd = raw + (8 * g + 4 * kb) * CH_F + (4 * tr + BI) * LDW + 4 * tc;
    unsigned *vw = reinterpret_cast<unsigned *>(lds + (wp * 9) * 3072 + lane * 16 + 8 * kb);
    const u32x4 *vr = reinterpret_cast<const u32x4 *>(lds + (wp * 9) * 3072 + lane * 16);
    const unsigned u_off = (unsigned)(((wp * 2 + kb) * 27) * 1024 + lane * 16);
    for (int s = 0; s < 16; ++s) {
      // ---------------- T: 4 channels of this lane's 8, 9 positions, limbs -> LDS
      if (PHASES != 0) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          float va[9], vb[9];
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) {
            float Q[5][3];
#pragma unroll
            for (int l = 0; l < 5; ++l) {
              const f32x4 r4 = *(lp4)(rd + (2 * pr + cc) * CH_F + l * LDW);
              const f32x2 r2 = *(lp2)(rd + (2 * pr + cc) * CH_F + l * LDW + 4);
              const float e[6] = {r4[0], r4[1], r4[2], r4[3], r2[0], r2[1]};
              xf<BJ>(e[BJ], e[BJ + 1], e[BJ + 2], e[BJ + 3], e[BJ + 4], Q[l][0], Q[l][1], Q[l][2]);
            }
            float *v = cc ? vb : va;
#pragma unroll
            for (int j = 0; j < 3; ++j) xf<BI>(Q[0][j], Q[1][j], Q[2][j], Q[3][j], Q[4][j], v[j], v[3 + j], v[6 + j]);
          }
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            unsigned h, m, l;
            split_pair(va[q], vb[q], h, m, l);
            vw[(q * 3 + 0) * 256 + pr] = h;
            vw[(q * 3 + 1) * 256 + pr] = m;
            vw[(q * 3 + 2) * 256 + pr] = l;
          }
        }
      }
      __syncthreads();
      // ---------------- M: 9 positions x 6 products; U limbs L2 -> registers two positions ahead; raw rows of step s + 1
      if (s + 1 < 16) load_raw(s + 1);
      if (PHASES != 2) {
        u32x4 ua[UD + 1][3];
        auto load_u = [&](int q, int slot) {
#pragma unroll
          for (int lm = 0; lm < 3; ++lm)
            ua[slot][lm] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, u_off + (q * 3 + lm) * 1024, s * 216 * 1024, 0));
        };
        load_u(0, 0);
        if (UD > 1) load_u(1, 1);
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          if (q + UD < 9) load_u(q + UD, (q + UD) % (UD + 1));
          const u32x4 vh = vr[(q * 3 + 0) * 64], vm = vr[(q * 3 + 1) * 64], vl = vr[(q * 3 + 2) * 64];
          const s16x8 Uh = __builtin_bit_cast(s16x8, ua[q % (UD + 1)][0]), Um = __builtin_bit_cast(s16x8, ua[q % (UD + 1)][1]),
                      Ul = __builtin_bit_cast(s16x8, ua[q % (UD + 1)][2]);
          const s16x8 Vh = __builtin_bit_cast(s16x8, vh), Vm = __builtin_bit_cast(s16x8, vm), Vl = __builtin_bit_cast(s16x8, vl);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Uh, Vh, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Uh, Vm, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Um, Vh, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Um, Vm, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Uh, Vl, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ul, Vh, acc[q], 0, 0, 0);
        }
      }
      if (s + 1 < 16) store_raw();
      __syncthreads();
    }
  };
  switch (wp) {
    case 0: body(int_c<0>(), int_c<0>()); break;
    case 1: body(int_c<0>(), int_c<1>()); break;
    case 2: body(int_c<1>(), int_c<0>()); break;
    default: body(int_c<1>(), int_c<1>()); break;
  }
  float a = 0.f;
#pragma unroll
  for (int q = 0; q < 9; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) a += acc[q][r];
  out[(long)blockIdx.x * 512 + tid] = a;
}

template <int PHASES, int UD, int XNT>
static void run(const char *what, const float *X, const unsigned short *U3, float *out, int N) {
  hipFuncSetAttribute((const void *)skeleton<PHASES, UD, XNT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((skeleton<PHASES, UD, XNT>), dim3(N * 8), dim3(512), LDS_BYTES, 0, X, U3, out, N);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  hipError_t e = hipGetLastError();
  const double gflop = 2.0 * N * 256.0 * 256.0 * 1024.0 * 9.0 / 1e9;
  printf("%-44s N=%3d : %7.3f ms  (%6.1f TFLOP/s algorithmic; wino4_f3x3_kernel with epilogue + InstanceNorm: 0.80 ms at N=256)  %s\n", what,
         N, best, gflop / best, e == hipSuccess ? "" : hipGetErrorString(e));
}

int main() {
  const int N = 256;
  float *X, *out;
  unsigned short *U3;
  hipMalloc(&X, (size_t)N * 256 * 1024 * 4);
  hipMalloc(&out, (size_t)N * 8 * 512 * 4);
  hipMalloc(&U3, (size_t)4 * 16 * 216 * 1024);
  std::vector<float> hx((size_t)N * 256 * 1024);
  srand(1);
  for (auto &v : hx) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  std::vector<unsigned short> hu((size_t)4 * 16 * 216 * 512);
  for (auto &v : hu) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));     // finite bf16 around +-0.01
  hipMemcpy(U3, hu.data(), hu.size() * 2, hipMemcpyHostToDevice);
  run<1, 1, 0>("T | M phases (the design), U one position ahead", X, U3, out, N);
  run<1, 2, 0>("T | M phases, U two positions ahead", X, U3, out, N);
  run<0, 1, 0>("M only (multiply side: V reads, U loads, raw)", X, U3, out, N);
  run<0, 2, 0>("M only, U two positions ahead", X, U3, out, N);
  run<2, 1, 0>("T only (transform + split + V writes, raw)", X, U3, out, N);
  // image rows with the non-temporal policy: the 3.5 MB U slice of an XCD keeps its 4 MB L2 against 32 streaming images
  run<1, 1, 1>("T | M phases, image loads nt", X, U3, out, N);
  run<0, 1, 1>("M only, image loads nt", X, U3, out, N);
  run<1, 1, 0>("T | M phases, N = 128", X, U3, out, 128);
  return 0;
}
