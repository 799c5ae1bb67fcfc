// Shared types of the conv kernels: vector typedefs, the math mode, the gather-GEMM parameter blocks (included by igemm.hip).
#ifndef LSPS_CONV_TYPES_H
#define LSPS_CONV_TYPES_H
#include "common.h"

namespace lsps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: HIP's float4 struct defeats SROA (scratch)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Math mode of the MFMA conv kernels: 0 = exact f32 MFMA (default), 1 = operands rounded to bf16 in registers
// (v_cvt_pk_bf16_f32, RNE) and v_mfma_f32_32x32x16_bf16 with f32 accumulation (BASELINE config 5).  HBM and LDS
// tensors stay f32 in both modes.
static int g_math_mode = 0;

// x = hi + mid + lo with three bf16 values (8+8+8 significand bits: exact for f32).  Used by math mode 2.
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 &hi, bf16x8 &mid, bf16x8 &lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h = (__bf16)x[e];
    const float r1 = x[e] - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    hi[e] = h;
    mid[e] = m;
    lo[e] = (__bf16)r2;
  }
}

// acc += a*b to ~f32 accuracy from six bf16 MFMAs (dropped terms mid*lo, lo*mid, lo*lo are < 2^-24 relative)
__device__ __forceinline__ f32x16 mfma_split6(const bf16x8 &ah, const bf16x8 &am, const bf16x8 &al, const bf16x8 &bh,
                                              const bf16x8 &bm, const bf16x8 &bl, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  return acc;
}

#define LSPS_MAXT 49
#define BK_F 32   // reduction chunk of the F kernel
#define BK_W 64   // reduction (pixel) chunk of the W kernel
#define LDW (BK_W + 1)

struct Taps {
  int T;
  int toff[LSPS_MAXT];          // dh*Wx + dw
  signed char dh[LSPS_MAXT], dw[LSPS_MAXT];
};

struct FParams {
  const float *X, *Wp, *bias;
  float *Y;
  const int2 *gtab;              // [REDp] (element offset c*HxWx + toff[t], tap index t; t = 63 for padding rows)
  const float *zero;             // >= 1 float of zeros: where masked-out gathers read from
  int Cx, Hx, Wx, HxWx;          // gather source [N][Cx][Hx][Wx]
  long xns, yns;                 // floats between samples of X / Y (= Cx*HxWx / M*HyWy unless X / Y are channel slices of wider
                                 // tensors: the groups of a grouped conv)
  int PH, PW, P, NPIX;           // output pixel lattice per sample, P = PH*PW, NPIX = N*P
  int ist;                       // input step per lattice step
  int RED, REDp, Mp;             // RED = Cx*T ; packed weights are [REDp][Mp], zero padded
  unsigned magicT;               // floor(2^32/T)+1 (T>1)
  int M, HyWy, Wy, h0, hs, w0, ws;   // D[m][pix] -> Y[n][m][h0+hs*ph][w0+ws*pw]
  int act;
  float slope;
  // split over the reduction (few-workgroup problems, e.g. the Post head: 20 x n outputs, 8192-long reduction):
  // blockIdx.z handles chunks [z*chunks_per_split, ...) and writes raw partial sums to part[z][m][pix]
  int ksplit, chunks_per_split;
  float *part;
  Taps taps;
};

struct WParams {
  const float *Small, *Big;
  float *part;                   // [splits][M][J]
  const int2 *jtab;              // [Jp = J rounded up to 128] (offset, tap) per column j = (c,t); tap 63 = padding
  const float *zero;
  int Cx, Hx, Wx, HxWx;          // Big = [N][Cx][Hx][Wx]
  long sns, bns;                 // floats between samples of Small / Big (channel slices of wider tensors: grouped conv)
  int PH, PW, P, NPIX;           // Small = [N][M][PH][PW]
  int ist;
  int M, J;                      // J = Cx*T
  unsigned magicT;
  int nchunks, chunks_per_split;
  Taps taps;
};

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  if (act == LSPS_ACT_LRELU) return v > 0.f ? v : v * slope;
  if (act == LSPS_ACT_TANH) return tanhf(v);
  return v;
}

}  // namespace lsps
#endif
