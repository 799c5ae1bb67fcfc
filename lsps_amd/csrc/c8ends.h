// The generator's output head on a C8 tensor (bf16 math mode): ConvTranspose2d(C, 1, kernel 1) + Tanh
// (reference: src/trainers/lsps_nets.py:226-229).  1.05 MMAC against 2 MB of bf16 activation per sample: HBM-bound
// direct kernels, one 16-byte channel-group unit per lane and load.
#ifndef LSPS_C8ENDS_H
#define LSPS_C8ENDS_H
#include "c8conv.h"

namespace lsps {

// y[n][pix] = act(b + sum_c w[c] x[n][c][pix]);  x [N][C/8][HW][8] bf16, y [N][HW] f32.  One pixel per thread.
__global__ __launch_bounds__(256) void c8_pw1_fwd_kernel(const unsigned short *__restrict__ x, const float *__restrict__ w,
                                                         const float *__restrict__ b, float *__restrict__ y, int C, int HW, int act,
                                                         float slope) {
  const int n = blockIdx.y, px = blockIdx.x * 256 + threadIdx.x;
  if (px >= HW) return;
  const u32x4 *xp = reinterpret_cast<const u32x4 *>(x) + (long)n * (C >> 3) * HW + px;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int g = 0; g < (C >> 3); ++g) {
    const bf16x8 v = __builtin_bit_cast(bf16x8, xp[(long)g * HW]);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e & 3] = fmaf(w[g * 8 + e], (float)v[e], acc[e & 3]);
  }
  y[(long)n * HW + px] = apply_act((acc[0] + acc[1]) + (acc[2] + acc[3]) + (b ? b[0] : 0.f), act, slope);
}

// dx[n][c][pix] = w[c] * dpre[n][pix]
__global__ __launch_bounds__(256) void c8_pw1_dgrad_kernel(const float *__restrict__ dpre, const float *__restrict__ w,
                                                           unsigned short *__restrict__ dx, int C, int HW) {
  const int n = blockIdx.y, px = blockIdx.x * 256 + threadIdx.x;
  if (px >= HW) return;
  const float d = dpre[(long)n * HW + px];
  u32x4 *xp = reinterpret_cast<u32x4 *>(dx) + (long)n * (C >> 3) * HW + px;
#pragma unroll 8
  for (int g = 0; g < (C >> 3); ++g) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)(w[g * 8 + e] * d);
    xp[(long)g * HW] = __builtin_bit_cast(u32x4, v);
  }
}

// dx[n][c][pix] = w[c] * dpre[n][pix] * (y[n][c][pix] > 0 ? 1 : slope): the head's input gradient with the LeakyReLU backward of
// the layer in front of it (whose saved output is y) fused, and that layer's bias gradient as partial sums part[row][c]
// (C <= 64).  Workgroup = one of S pixel segments of an image (row = n * S + segment); a 32-lane group owns ONE channel group
// (16 accumulators per lane instead of 128: eight waves per SIMD keep the two 1 GB streams in flight), 32 consecutive pixels
// per step = 512 contiguous bytes, four steps of loads in flight.
// wpart (nullable): partial sums of the head's OWN weight gradient, wpart[row][c] = sum_pix y[n][c][pix] * dpre[n][pix]
// (y is the head's input), and wpart[row][C] = sum_pix dpre (its bias gradient): the same two operands stream through here
// anyway, so the separate c8_pw1_wgrad pass over y disappears.
__global__ __launch_bounds__(256) void c8_pw1_dgrad_act_kernel(const float *__restrict__ dpre, const float *__restrict__ w,
                                                               const unsigned short *__restrict__ y, unsigned short *__restrict__ dx,
                                                               float *__restrict__ part, float *__restrict__ wpart, int C, int HW,
                                                               int S, float slope) {
  const int row = blockIdx.x, n = row / S, seg = row - n * S, tid = threadIdx.x;
  const int g = tid >> 5, lp = tid & 31, G = C >> 3;
  if (g >= G) return;                                           // no barrier below
  const int per = (HW + S - 1) / S, px0 = seg * per, px1 = min(HW, px0 + per);
  const u32x4 *yp = reinterpret_cast<const u32x4 *>(y) + ((long)n * G + g) * HW;
  u32x4 *xp = reinterpret_cast<u32x4 *>(dx) + ((long)n * G + g) * HW;
  const float *dp = dpre + (long)n * HW;
  float wg[8], s[8], sw[8], sd = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    wg[e] = w[g * 8 + e];
    s[e] = sw[e] = 0.f;
  }
  for (int px = px0 + lp; px < px1; px += 128) {
    u32x4 yv[4];
    float d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = px + 32 * u;
      const bool in = q < px1;
      yv[u] = in ? __builtin_nontemporal_load(yp + q) : u32x4{0u, 0u, 0u, 0u};
      d[u] = in ? dp[q] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = px + 32 * u;
      const bf16x8 a = __builtin_bit_cast(bf16x8, yv[u]);
      bf16x8 v;
      sd += d[u];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = wg[e] * d[u];
        v[e] = (__bf16)c8_sel_nonpos((float)a[e], x * slope, x);
        s[e] += (float)v[e];
        sw[e] = fmaf((float)a[e], d[u], sw[e]);
      }
      if (q < px1) __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), xp + q);
    }
  }
  // sums over the 32 lanes of the group
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s[e] += __shfl_xor(s[e], off, 64);
      sw[e] += __shfl_xor(sw[e], off, 64);
    }
    sd += __shfl_xor(sd, off, 64);
  }
  if (lp == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) part[(long)row * C + g * 8 + e] = s[e];
    if (wpart) {
#pragma unroll
      for (int e = 0; e < 8; ++e) wpart[(long)row * (C + 1) + g * 8 + e] = sw[e];
      if (g == 0) wpart[(long)row * (C + 1) + C] = sd;
    }
  }
}

// part[split][c] = sum over the split's images and all pixels of x[n][c][pix] * dpre[n][pix];  part[split][C] = sum of dpre
// (the bias gradient; written by the blocks of channel group 0).  grid (C / 8, splits).
__global__ __launch_bounds__(256) void c8_pw1_wgrad_kernel(const unsigned short *__restrict__ x, const float *__restrict__ dpre,
                                                           float *__restrict__ part, int N, int C, int HW, int imgs_per_split) {
  __shared__ float red[4][9];
  const int g = blockIdx.x, split = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = split * imgs_per_split, n1 = min(N, n0 + imgs_per_split);
  float s[9];
#pragma unroll
  for (int e = 0; e < 9; ++e) s[e] = 0.f;
  for (int n = n0; n < n1; ++n) {
    const u32x4 *xp = reinterpret_cast<const u32x4 *>(x) + ((long)n * (C >> 3) + g) * HW;
    const float *dp = dpre + (long)n * HW;
    for (int px = tid; px < HW; px += 256) {
      const bf16x8 v = __builtin_bit_cast(bf16x8, xp[px]);
      const float d = dp[px];
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] = fmaf((float)v[e], d, s[e]);
      s[8] += d;
    }
  }
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    s[e] = wave_sum(s[e]);
    if (lane == 0) red[wave][e] = s[e];
  }
  __syncthreads();
  if (tid < 8) part[(long)split * (C + 1) + g * 8 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
  if (tid == 8 && g == 0) part[(long)split * (C + 1) + C] = red[0][8] + red[1][8] + red[2][8] + red[3][8];
}

// dW[c] = sum_s part[s][c] (c < C), db[0] = sum_s part[s][C]
__global__ __launch_bounds__(256) void c8_pw1_wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ dW, float *__restrict__ db,
                                                                  int C, int splits) {
  __shared__ float red[4];
  const int c = blockIdx.x, tid = threadIdx.x;                  // one workgroup per column (c == C: the bias gradient)
  float s = 0.f;
  for (int i = tid; i < splits; i += 256) s += part[(long)i * (C + 1) + c];
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    s = red[0] + red[1] + red[2] + red[3];
    if (c < C)
      dW[c] = s;
    else if (db)
      db[0] = s;
  }
}

}  // namespace lsps
#endif
