// Data step of the depth path on the GPU: `normalize` and the image part of `augmentCrop`
// (reference src/data/dataset_hand2.py:27-31, 34-119) for a whole batch of 128x128 crops in one launch each.
// Pure HBM-bound float work: every crop is read once and written once; the per-sample geometry (inverse warp
// matrix, z-thresholds, old / new CoM depth and cube) is computed on the host (lsps_amd/data.py) and passed as a
// small table of doubles.
//
// Bit-exactness notes.  The reference does this arithmetic in numpy float32 with one rounding per operation, and
// OpenCV forms the nearest-neighbour source coordinates in double (warpPerspective) or 10-bit fixed point
// (warpAffine).  A fused multiply-add would change roundings, so contraction is switched off for this file.
#pragma clang fp contract(off)
#include "common.h"

namespace lsps {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void crop_normalize_kernel(const float *__restrict__ dpt, const float *__restrict__ com_z,
                                                             const float *__restrict__ half, float *__restrict__ out,
                                                             int HW4) {
  // one block column per image: grid (ceil(HW4/256), N)
  const int n = blockIdx.y;
  const float cz = com_z[n], hf = half[n];
  const float bg = cz + hf;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HW4) return;
  const f32x4 v = reinterpret_cast<const f32x4 *>(dpt)[(long)n * HW4 + i];
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float d = v[e] == 0.f ? bg : v[e];
    d = d - cz;
    r[e] = d / hf;
  }
  reinterpret_cast<f32x4 *>(out)[(long)n * HW4 + i] = r;
}

__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

// cvRound of a double already clamped to the int range
__device__ __forceinline__ int cv_round(double v) {
  v = fmin(fmax(v, -2147483648.0), 2147483647.0);
  return __double2int_rn(v);
}

// prm (per sample, LSPS_AUG_STRIDE doubles): 0 kind (0 none, 1 perspective, 2 affine); 1 com_z_in; 2 half_in;
// 3 com_z_out; 4 half_out; 5 zstart; 6 zend; 7.. inverse map (9 or 6 values)
__global__ __launch_bounds__(256) void crop_augment_kernel(const float *__restrict__ x, const double *__restrict__ prm,
                                                           float *__restrict__ out, int H, int W, int bh0, int bw0,
                                                           int rows_per_block) {
  __shared__ float red[4];
  // grid (N, row slices): small batches are cut into row slices so that the chip is filled; every slice recomputes
  // the crop's pre-max itself (the crop is 64 KB: L2-resident after the first slice touched it)
  const int n = blockIdx.x, tid = threadIdx.x;
  const double *q = prm + (long)n * LSPS_AUG_STRIDE;
  const int kind = (int)q[0];
  const float cz_in = (float)q[1], hf_in = (float)q[2], cz_out = (float)q[3], hf_out = (float)q[4];
  const float zstart = (float)q[5], zend = (float)q[6];
  const float *xi = x + (long)n * H * W;
  float *oi = out + (long)n * H * W;
  const int HW = H * W;
  const int i_begin = blockIdx.y * rows_per_block * W;
  const int i_end = min(HW, i_begin + rows_per_block * W);

  // premax = max of the de-normalised crop (dataset_hand2.py:66-67)
  float mx = -INFINITY;
  if ((HW & 3) == 0) {
    for (int i = tid; i < HW / 4; i += 256) {
      const f32x4 v = reinterpret_cast<const f32x4 *>(xi)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) mx = fmaxf(mx, v[e] * hf_in + cz_in);
    }
  } else {
    for (int i = tid; i < HW; i += 256) mx = fmaxf(mx, xi[i] * hf_in + cz_in);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  const float premax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));

  const float bg = cz_out + hf_out, nearp = cz_out - hf_out;
  double m[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) m[i] = q[7 + i];

  for (int i = i_begin + tid; i < i_end; i += 256) {
    const int y = i / W, xq = i - y * W;
    float v;
    if (kind == 0) {
      v = xi[i] * hf_in + cz_in;
    } else {
      int X, Y;
      if (kind == 2) {               // cv2.warpAffine, INTER_NEAREST: AB_BITS = 10, round_delta = 512
        const int ad = cv_round(m[0] * xq * 1024.0), bd = cv_round(m[3] * xq * 1024.0);
        const int X0 = cv_round((m[1] * y + m[2]) * 1024.0) + 512, Y0 = cv_round((m[4] * y + m[5]) * 1024.0) + 512;
        X = sat_short((X0 + ad) >> 10);
        Y = sat_short((Y0 + bd) >> 10);
      } else {                       // cv2.warpPerspective, INTER_NEAREST: block origin + in-block offset, in double
        const int bx = (xq / bw0) * bw0, x1 = xq - bx;
        const double X0 = m[0] * bx + m[1] * y + m[2], Y0 = m[3] * bx + m[4] * y + m[5], W0 = m[6] * bx + m[7] * y + m[8];
        double w = W0 + m[6] * x1;
        w = w != 0.0 ? 1.0 / w : 0.0;
        X = sat_short(cv_round((X0 + m[0] * x1) * w));
        Y = sat_short(cv_round((Y0 + m[3] * x1) * w));
      }
      v = 0.f;                       // BORDER_CONSTANT, pad_value = 0
      if (X >= 0 && X < W && Y >= 0 && Y < H) v = xi[Y * W + X] * hf_in + cz_in;
      if (kind == 1) {               // recropHand (handdetector.py:786-805): nv_val = 32000, z-thresholds
        if (fabs((double)v - 32000.0) <= 1e-8 + 1e-5 * 32000.0) v = 0.f;
        const bool lo = v < zstart && v != 0.f, hi = v > zend && v != 0.f;
        if (lo) v = zstart;
        if (hi) v = 0.f;
      }
    }
    // tail (dataset_hand2.py:109-114)
    if (v == premax) v = bg;
    if (v == 0.f) v = bg;
    if (v >= bg) v = bg;
    if (v <= nearp) v = nearp;
    v = v - cz_out;
    oi[i] = v / hf_out;
  }
}

}  // namespace lsps

using namespace lsps;

extern "C" int lsps_crop_normalize(const float *dpt, const float *com_z, const float *half, float *out, int N, int HW,
                                   void *stream) {
  (void)hipGetLastError();
  if (N < 0 || HW < 0 || (HW & 3)) {
    set_error("crop_normalize: bad sizes N=%d HW=%d (HW must be a multiple of 4)", N, HW);
    return LSPS_E_ARG;
  }
  if (N == 0 || HW == 0) return 0;
  if (!dpt || !com_z || !half || !out) {
    set_error("crop_normalize: null pointer");
    return LSPS_E_ARG;
  }
  const int HW4 = HW / 4;
  hipLaunchKernelGGL(crop_normalize_kernel, dim3((HW4 + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, dpt, com_z,
                     half, out, HW4);
  LSPS_CHECK_LAUNCH("crop_normalize");
  return 0;
}

extern "C" int lsps_crop_augment(const float *x, const double *prm, float *out, int N, int H, int W, void *stream) {
  (void)hipGetLastError();
  if (N < 0 || H <= 0 || W <= 0 || H > 32767 || W > 32767) {
    set_error("crop_augment: bad sizes N=%d H=%d W=%d", N, H, W);
    return LSPS_E_ARG;
  }
  if (N == 0) return 0;
  if (!x || !prm || !out || x == out) {
    set_error("crop_augment: null or aliased pointer (the warp gathers: it cannot run in place)");
    return LSPS_E_ARG;
  }
  // the destination blocks cv::warpPerspective walks (BLOCK_SZ = 32): the source coordinate of a pixel is formed as
  // (block origin term) + (in-block offset term), which fixes the double roundings
  int bh0 = H < 16 ? H : 16;
  int bw0 = 1024 / bh0 < W ? 1024 / bh0 : W;
  bh0 = 1024 / bw0 < H ? 1024 / bw0 : H;
  int slices = 1;                       // aim at >= 2048 workgroups (256 CUs x 8)
  while (slices < 8 && (long)N * slices < 2048 && H % (slices * 2) == 0) slices *= 2;
  hipLaunchKernelGGL(crop_augment_kernel, dim3(N, slices), dim3(256), 0, (hipStream_t)stream, x, prm, out, H, W, bh0,
                     bw0, H / slices);
  LSPS_CHECK_LAUNCH("crop_augment");
  return 0;
}
