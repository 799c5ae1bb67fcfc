// InstanceNorm2d(affine=False) fused with LeakyReLU / residual add, activation backward, axpy.
// All HBM-bound: one pass over the data, float4 accesses, wavefront-shuffle reductions.
//
// Reference call sites: src/trainers/common_net.py:168-171 (InstanceNorm2d + in-place LeakyReLU),
// :177-181 (InstanceNorm2d then `out += residual`), :32-40 (GaussianNoiseLayer add),
// :252,264 and src/trainers/lsps_nets.py:228 (activation backward inside autograd).
#include "common.h"

namespace lsps {

// One workgroup (256 threads) per (n,c) plane.  VPT float4 per thread held in registers when the
// plane has exactly 256*4*VPT elements (the 32x32 latent: VPT=1); generic strided path otherwise.
template <int VPT>
__global__ __launch_bounds__(256) void inorm_fwd_kernel(const float *__restrict__ y, const float *__restrict__ res,
                                                        float *__restrict__ out, float *__restrict__ rstd_out, int hw,
                                                        float eps, float slope) {
  __shared__ float red[4];
  const long base = (long)blockIdx.x * hw;
  const float inv = 1.f / (float)hw;
  if (VPT > 0) {
    float4 v[VPT > 0 ? VPT : 1];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      v[i] = *reinterpret_cast<const float4 *>(y + base + (threadIdx.x + 256 * i) * 4);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = block_sum_256(s, red) * inv;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
    const float var = block_sum_256(q, red) * inv;
    const float rstd = 1.f / sqrtf(var + eps);
    if (threadIdx.x == 0) rstd_out[blockIdx.x] = rstd;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const long o = base + (threadIdx.x + 256 * i) * 4;
      float4 r;
      r.x = (v[i].x - mean) * rstd;
      r.y = (v[i].y - mean) * rstd;
      r.z = (v[i].z - mean) * rstd;
      r.w = (v[i].w - mean) * rstd;
      if (slope >= 0.f) {
        r.x = r.x > 0.f ? r.x : r.x * slope;
        r.y = r.y > 0.f ? r.y : r.y * slope;
        r.z = r.z > 0.f ? r.z : r.z * slope;
        r.w = r.w > 0.f ? r.w : r.w * slope;
      }
      if (res) {
        const float4 x = *reinterpret_cast<const float4 *>(res + o);
        r.x += x.x;
        r.y += x.y;
        r.z += x.z;
        r.w += x.w;
      }
      *reinterpret_cast<float4 *>(out + o) = r;
    }
  } else {
    float s = 0.f;
    for (int i = threadIdx.x; i < hw; i += 256) s += y[base + i];
    const float mean = block_sum_256(s, red) * inv;
    float q = 0.f;
    for (int i = threadIdx.x; i < hw; i += 256) {
      const float d = y[base + i] - mean;
      q += d * d;
    }
    const float var = block_sum_256(q, red) * inv;
    const float rstd = 1.f / sqrtf(var + eps);
    __syncthreads();   // all reads of y done before (possibly aliased) writes
    if (threadIdx.x == 0) rstd_out[blockIdx.x] = rstd;
    for (int i = threadIdx.x; i < hw; i += 256) {
      float r = (y[base + i] - mean) * rstd;
      if (slope >= 0.f) r = r > 0.f ? r : r * slope;
      if (res) r += res[base + i];
      out[base + i] = r;
    }
  }
}

// Backward from the OUTPUT.  act variant (slope>=0): g = dout*lrelu'(out), xhat = out>0 ? out : out/slope.
// residual variant (res != null, slope<0): g = dout, xhat = out - res.  plain (no res, slope<0): xhat = out.
// dy = rstd * (g - mean(g) - xhat*mean(g*xhat))
// `dy` MAY alias `dout` (lsps_conv2d_dgrad_inbwd's fallback path runs in place): neither is __restrict__; every thread
// reads exactly the elements it later writes, and in the VPT == 0 form all reads of the second loop precede its write.
template <int VPT>
__global__ __launch_bounds__(256) void inorm_bwd_kernel(const float *dout, const float *__restrict__ out,
                                                        const float *__restrict__ res, const float *__restrict__ rstd_in,
                                                        float *dy, int hw, float slope) {
  __shared__ float red[4];
  const long base = (long)blockIdx.x * hw;
  const float inv = 1.f / (float)hw;
  const float rstd = rstd_in[blockIdx.x];
  const float inv_slope = slope > 0.f ? 1.f / slope : 0.f;
  auto xhat_g = [&](float o, float r, float d, float &xh, float &g) {
    if (slope >= 0.f) {
      const bool pos = o > 0.f;
      xh = pos ? o : o * inv_slope;
      g = pos ? d : d * slope;
    } else {
      xh = o - r;
      g = d;
    }
  };
  if (VPT > 0) {
    float xh[VPT > 0 ? VPT * 4 : 1], g[VPT > 0 ? VPT * 4 : 1];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const long o = base + (threadIdx.x + 256 * i) * 4;
      const float4 ov = *reinterpret_cast<const float4 *>(out + o);
      const float4 dv = *reinterpret_cast<const float4 *>(dout + o);
      float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (res) rv = *reinterpret_cast<const float4 *>(res + o);
      xhat_g(ov.x, rv.x, dv.x, xh[4 * i + 0], g[4 * i + 0]);
      xhat_g(ov.y, rv.y, dv.y, xh[4 * i + 1], g[4 * i + 1]);
      xhat_g(ov.z, rv.z, dv.z, xh[4 * i + 2], g[4 * i + 2]);
      xhat_g(ov.w, rv.w, dv.w, xh[4 * i + 3], g[4 * i + 3]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s1 += g[4 * i + k];
        s2 += g[4 * i + k] * xh[4 * i + k];
      }
    }
    const float m1 = block_sum_256(s1, red) * inv;
    const float m2 = block_sum_256(s2, red) * inv;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const long o = base + (threadIdx.x + 256 * i) * 4;
      float4 r;
      r.x = rstd * (g[4 * i + 0] - m1 - xh[4 * i + 0] * m2);
      r.y = rstd * (g[4 * i + 1] - m1 - xh[4 * i + 1] * m2);
      r.z = rstd * (g[4 * i + 2] - m1 - xh[4 * i + 2] * m2);
      r.w = rstd * (g[4 * i + 3] - m1 - xh[4 * i + 3] * m2);
      *reinterpret_cast<float4 *>(dy + o) = r;
    }
  } else {
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < hw; i += 256) {
      float xh, g;
      xhat_g(out[base + i], res ? res[base + i] : 0.f, dout[base + i], xh, g);
      s1 += g;
      s2 += g * xh;
    }
    const float m1 = block_sum_256(s1, red) * inv;
    const float m2 = block_sum_256(s2, red) * inv;
    __syncthreads();
    for (int i = threadIdx.x; i < hw; i += 256) {
      float xh, g;
      xhat_g(out[base + i], res ? res[base + i] : 0.f, dout[base + i], xh, g);
      dy[base + i] = rstd * (g - m1 - xh * m2);
    }
  }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ out,
                                                      float *__restrict__ dx, long n, int kind, float slope) {
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const float4 d = *reinterpret_cast<const float4 *>(dy + i);
      const float4 o = *reinterpret_cast<const float4 *>(out + i);
      float4 r;
      if (kind == LSPS_ACT_LRELU) {
        r.x = o.x > 0.f ? d.x : d.x * slope;
        r.y = o.y > 0.f ? d.y : d.y * slope;
        r.z = o.z > 0.f ? d.z : d.z * slope;
        r.w = o.w > 0.f ? d.w : d.w * slope;
      } else if (kind == LSPS_ACT_TANH) {
        r.x = d.x * (1.f - o.x * o.x);
        r.y = d.y * (1.f - o.y * o.y);
        r.z = d.z * (1.f - o.z * o.z);
        r.w = d.w * (1.f - o.w * o.w);
      } else {                                   // softplus: sigmoid(z) = 1 - exp(-softplus(z))
        r.x = o.x > 20.f ? d.x : d.x * (1.f - expf(-o.x));
        r.y = o.y > 20.f ? d.y : d.y * (1.f - expf(-o.y));
        r.z = o.z > 20.f ? d.z : d.z * (1.f - expf(-o.z));
        r.w = o.w > 20.f ? d.w : d.w * (1.f - expf(-o.w));
      }
      *reinterpret_cast<float4 *>(dx + i) = r;
    } else {
      for (long k = i; k < n; ++k) {
        const float o = out[k], d = dy[k];
        dx[k] = kind == LSPS_ACT_LRELU ? (o > 0.f ? d : d * slope)
                                       : (kind == LSPS_ACT_TANH ? d * (1.f - o * o) : (o > 20.f ? d : d * (1.f - expf(-o))));
      }
    }
  }
}

// act_bwd fused with the bias gradient of the same layer: dx = dy * act'(out) and part[s][c] = sum of dx over this block's
// slice of channel c (conv bias gradient = sum of the pre-activation gradient over n, h, w).  One pass over (dy, out)
// instead of act_bwd followed by a separate reduction pass over dx.  grid (C, S).
__global__ __launch_bounds__(256) void act_bwd_bias_kernel(const float *__restrict__ dy, const float *__restrict__ out,
                                                           float *__restrict__ dx, float *__restrict__ part, int N, int C,
                                                           int HW, long slice, int kind, float slope) {
  __shared__ float red[4];
  const int c = blockIdx.x, sidx = blockIdx.y;
  const long total = (long)N * HW;
  const long e0 = (long)sidx * slice;
  long e1 = e0 + slice;
  if (e1 > total) e1 = total;
  float s = 0.f;
  if ((HW & 3) == 0) {
    // (n, p) = (e / HW, e % HW) advanced incrementally: one 64-bit division per thread instead of one per float4;
    // two float4 per iteration so that four loads are in flight per thread (four per iteration measured slower: 4.9 vs 5.1 TB/s)
    long e = e0 + (long)threadIdx.x * 4;
    long n = e / HW;
    int p = (int)(e - n * HW);
    const int dn = 1024 / HW, dp = 1024 - dn * HW;
    auto next_off = [&]() {
      const long off = (n * C + c) * HW + p;
      n += dn;
      p += dp;
      if (p >= HW) {
        p -= HW;
        ++n;
      }
      return off;
    };
    auto bwd = [&](float4 d, float4 o) {
      float4 r;
      if (kind == LSPS_ACT_LRELU) {
        r.x = o.x > 0.f ? d.x : d.x * slope;
        r.y = o.y > 0.f ? d.y : d.y * slope;
        r.z = o.z > 0.f ? d.z : d.z * slope;
        r.w = o.w > 0.f ? d.w : d.w * slope;
      } else {
        r.x = d.x * (1.f - o.x * o.x);
        r.y = d.y * (1.f - o.y * o.y);
        r.z = d.z * (1.f - o.z * o.z);
        r.w = d.w * (1.f - o.w * o.w);
      }
      return r;
    };
    for (; e + 1024 < e1; e += 2048) {
      const long off0 = next_off(), off1 = next_off();
      const float4 d0 = *reinterpret_cast<const float4 *>(dy + off0), o0 = *reinterpret_cast<const float4 *>(out + off0);
      const float4 d1 = *reinterpret_cast<const float4 *>(dy + off1), o1 = *reinterpret_cast<const float4 *>(out + off1);
      const float4 r0 = bwd(d0, o0), r1 = bwd(d1, o1);
      *reinterpret_cast<float4 *>(dx + off0) = r0;
      *reinterpret_cast<float4 *>(dx + off1) = r1;
      s += (r0.x + r0.y) + (r0.z + r0.w);          // same summation order as one float4 per iteration
      s += (r1.x + r1.y) + (r1.z + r1.w);
    }
    if (e < e1) {
      const long off = next_off();
      const float4 r = bwd(*reinterpret_cast<const float4 *>(dy + off), *reinterpret_cast<const float4 *>(out + off));
      *reinterpret_cast<float4 *>(dx + off) = r;
      s += (r.x + r.y) + (r.z + r.w);
    }
  } else {
    for (long e = e0 + threadIdx.x; e < e1; e += 256) {
      const long n = e / HW;
      const long off = (n * C + c) * HW + (e - n * HW);
      const float o = out[off], d = dy[off];
      const float r = kind == LSPS_ACT_LRELU ? (o > 0.f ? d : d * slope) : d * (1.f - o * o);
      dx[off] = r;
      s += r;
    }
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) part[(long)sidx * C + c] = s;
}

__global__ __launch_bounds__(256) void sum_slices_kernel(const float *__restrict__ part, float *__restrict__ out, int C, int S) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int k = 0; k < S; ++k) s += part[(long)k * C + c];
  out[c] = s;
}

// out = (x ? x : 0) + t * m   (dropout applied to a residual branch: nn.Dropout after the second InstanceNorm of a
// residual block, common_net.py:171-172, followed by `out += residual`, :180; m = keep mask / (1 - p))
__global__ __launch_bounds__(256) void mul_add_kernel(const float *__restrict__ x, const float *__restrict__ t,
                                                      const float *__restrict__ m, float *__restrict__ out, long n) {
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const float4 a = *reinterpret_cast<const float4 *>(t + i);
      const float4 b = *reinterpret_cast<const float4 *>(m + i);
      float4 r = make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
      if (x) {
        const float4 c = *reinterpret_cast<const float4 *>(x + i);
        r.x += c.x;
        r.y += c.y;
        r.z += c.z;
        r.w += c.w;
      }
      *reinterpret_cast<float4 *>(out + i) = r;
    } else {
      for (long k = i; k < n; ++k) out[k] = (x ? x[k] : 0.f) + t[k] * m[k];
    }
  }
}

// -------------------------------------------------------------------------------------------
// BatchNorm (nn.BatchNorm2d / nn.BatchNorm1d of the BN block variants, common_net.py:183-322): per-channel statistics
// over (N, HW).  Not on the shipped configs' path (they use InstanceNorm); kept simple: one statistics pass with
// double accumulators (sum, sum of squares), one apply pass.  x: [N][C][HW].
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_stats_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                       const float *__restrict__ mean, double *__restrict__ part, int N,
                                                       int C, int HW, long slice) {
  // g == nullptr: part = (sum x, sum x^2);  g != nullptr (backward): xh = (x - mean) -> part = (sum g, sum g*(x - mean))
  __shared__ double red[2][4];
  const int c = blockIdx.x, sidx = blockIdx.y;
  const long total = (long)N * HW;
  const long e0 = (long)sidx * slice;
  long e1 = e0 + slice;
  if (e1 > total) e1 = total;
  const float mu = g ? mean[c] : 0.f;
  double s0 = 0.0, s1 = 0.0;
  for (long e = e0 + threadIdx.x; e < e1; e += 256) {
    const long n = e / HW;
    const long off = (n * C + c) * HW + (e - n * HW);
    const float v = x[off];
    if (g) {
      const float gv = g[off];
      s0 += gv;
      s1 += (double)gv * (double)(v - mu);
    } else {
      s0 += v;
      s1 += (double)v * (double)v;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s0 += __shfl_xor(s0, o, 64);
    s1 += __shfl_xor(s1, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s0;
    red[1][threadIdx.x >> 6] = s1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[((long)sidx * C + c) * 2 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    part[((long)sidx * C + c) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// forward finalize: batch mean / biased var -> mean, rstd; running stats updated with the UNBIASED variance
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double *__restrict__ part, int S, int C, long count, float eps,
                                                          float momentum, float *__restrict__ mean, float *__restrict__ rstd,
                                                          float *__restrict__ run_mean, float *__restrict__ run_var) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s0 = 0.0, s1 = 0.0;
  for (int k = 0; k < S; ++k) {
    s0 += part[((long)k * C + c) * 2];
    s1 += part[((long)k * C + c) * 2 + 1];
  }
  const double m = s0 / (double)count;
  double var = s1 / (double)count - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean) {
    const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)m;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unbiased;
  }
}

__global__ __launch_bounds__(256) void bn_eval_stats_kernel(const float *__restrict__ run_mean, const float *__restrict__ run_var,
                                                            float eps, int C, float *__restrict__ mean, float *__restrict__ rstd) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  mean[c] = run_mean[c];
  rstd[c] = 1.f / sqrtf(run_var[c] + eps);
}

// y = act((x - mean) * rstd * gamma + beta)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                                       const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, float *__restrict__ y, long total, int C,
                                                       int HW, float slope) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)((i / HW) % C);
    float v = (x[i] - mean[c]) * rstd[c];
    if (gamma) v *= gamma[c];
    if (beta) v += beta[c];
    if (slope >= 0.f) v = v > 0.f ? v : v * slope;
    y[i] = v;
  }
}

// backward finalize: per channel a = sum g, b = sum g*(x-mean) -> dgamma = b*rstd, dbeta = a, and the two means the
// apply pass needs (train mode): ma = a/count, mb = b*rstd^2/count
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double *__restrict__ part, int S, int C, long count,
                                                              const float *__restrict__ rstd, float *__restrict__ dgamma,
                                                              float *__restrict__ dbeta, float *__restrict__ ma,
                                                              float *__restrict__ mb) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int k = 0; k < S; ++k) {
    a += part[((long)k * C + c) * 2];
    b += part[((long)k * C + c) * 2 + 1];
  }
  if (dgamma) dgamma[c] = (float)(b * (double)rstd[c]);
  if (dbeta) dbeta[c] = (float)a;
  ma[c] = (float)(a / (double)count);
  mb[c] = (float)(b * (double)rstd[c] * (double)rstd[c] / (double)count);
}

// dx = gamma * rstd * (g - ma - (x - mean) * mb)   (training)   |   gamma * rstd * g   (eval: constants)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *__restrict__ g, const float *__restrict__ x,
                                                           const float *__restrict__ mean, const float *__restrict__ rstd,
                                                           const float *__restrict__ gamma, const float *__restrict__ ma,
                                                           const float *__restrict__ mb, float *__restrict__ dx, long total,
                                                           int C, int HW, int training) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)((i / HW) % C);
    const float sc = rstd[c] * (gamma ? gamma[c] : 1.f);
    float v = g[i];
    if (training) v = v - ma[c] - (x[i] - mean[c]) * mb[c];
    dx[i] = v * sc;
  }
}

// out = act(x): standalone activation forward (ReLU = LRELU with slope 0, Softplus, Tanh) for the block variants whose
// activation does not follow a conv directly (common_net.py:146,361; GaussianVAE2D :71-80)
__global__ __launch_bounds__(256) void act_fwd_kernel(const float *__restrict__ x, float *__restrict__ out, long n, int kind,
                                                      float slope) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = x[i];
    float r;
    if (kind == LSPS_ACT_LRELU)
      r = v > 0.f ? v : v * slope;
    else if (kind == LSPS_ACT_TANH)
      r = tanhf(v);
    else
      r = v > 20.f ? v : log1pf(expf(v));
    out[i] = r;
  }
}

__global__ __launch_bounds__(256) void axpy_kernel(const float *__restrict__ x, const float *__restrict__ y, float alpha,
                                                   float *__restrict__ out, long n) {
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const float4 a = *reinterpret_cast<const float4 *>(x + i);
      const float4 b = *reinterpret_cast<const float4 *>(y + i);
      float4 r;
      r.x = a.x + alpha * b.x;
      r.y = a.y + alpha * b.y;
      r.z = a.z + alpha * b.z;
      r.w = a.w + alpha * b.w;
      *reinterpret_cast<float4 *>(out + i) = r;
    } else {
      for (long k = i; k < n; ++k) out[k] = x[k] + alpha * y[k];
    }
  }
}

static int ew_grid(long n) {
  long b = (n / 4 + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace lsps

using namespace lsps;

extern "C" {

int lsps_inorm_fwd(const float *y, const float *residual, float *out, float *rstd, int planes, int hw, float eps,
                   float slope, void *stream) {
  (void)hipGetLastError();   // clear stale sticky errors left by other users of the runtime
  LSPS_CHECK_ARG(y && out && rstd && planes > 0 && hw > 0, "inorm_fwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const bool al = (((uintptr_t)y | (uintptr_t)out | (uintptr_t)residual) & 15) == 0;
  if (hw == 1024 && al)
    hipLaunchKernelGGL(inorm_fwd_kernel<1>, dim3(planes), dim3(256), 0, st, y, residual, out, rstd, hw, eps, slope);
  else if (hw == 4096 && al)
    hipLaunchKernelGGL(inorm_fwd_kernel<4>, dim3(planes), dim3(256), 0, st, y, residual, out, rstd, hw, eps, slope);
  else
    hipLaunchKernelGGL(inorm_fwd_kernel<0>, dim3(planes), dim3(256), 0, st, y, residual, out, rstd, hw, eps, slope);
  LSPS_CHECK_LAUNCH("inorm_fwd");
  return 0;
}

int lsps_inorm_bwd(const float *dout, const float *out, const float *residual, const float *rstd, float *dy, int planes,
                   int hw, float slope, void *stream) {
  (void)hipGetLastError();   // clear stale sticky errors left by other users of the runtime
  LSPS_CHECK_ARG(dout && out && rstd && dy && planes > 0 && hw > 0, "inorm_bwd: bad argument");
  LSPS_CHECK_ARG(!(slope == 0.f), "inorm_bwd: slope 0 (ReLU) cannot be inverted from the output");
  hipStream_t st = (hipStream_t)stream;
  const bool al = (((uintptr_t)dout | (uintptr_t)out | (uintptr_t)residual | (uintptr_t)dy) & 15) == 0;
  if (hw == 1024 && al)
    hipLaunchKernelGGL(inorm_bwd_kernel<1>, dim3(planes), dim3(256), 0, st, dout, out, residual, rstd, dy, hw, slope);
  else if (hw == 4096 && al)
    hipLaunchKernelGGL(inorm_bwd_kernel<4>, dim3(planes), dim3(256), 0, st, dout, out, residual, rstd, dy, hw, slope);
  else
    hipLaunchKernelGGL(inorm_bwd_kernel<0>, dim3(planes), dim3(256), 0, st, dout, out, residual, rstd, dy, hw, slope);
  LSPS_CHECK_LAUNCH("inorm_bwd");
  return 0;
}

int lsps_act_bwd(const float *dy, const float *out, float *dx, long n, int kind, float slope, void *stream) {
  (void)hipGetLastError();   // clear stale sticky errors left by other users of the runtime
  LSPS_CHECK_ARG(dy && out && dx && n >= 0, "act_bwd: bad argument");
  LSPS_CHECK_ARG(kind == LSPS_ACT_LRELU || kind == LSPS_ACT_TANH || kind == LSPS_ACT_SOFTPLUS, "act_bwd: unknown activation");
  LSPS_CHECK_ARG((((uintptr_t)dy | (uintptr_t)out | (uintptr_t)dx) & 15) == 0, "act_bwd: pointers must be 16-byte aligned");
  if (n == 0) return 0;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dy, out, dx, n, kind, slope);
  LSPS_CHECK_LAUNCH("act_bwd");
  return 0;
}

int lsps_act_bwd_bias(const float *dy, const float *out, float *dx, float *db, int N, int C, int HW, int kind, float slope,
                      void *ws, size_t ws_bytes, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(dy && out && dx && db && N >= 0 && C > 0 && HW > 0, "act_bwd_bias: bad argument");
  LSPS_CHECK_ARG(kind == LSPS_ACT_LRELU || kind == LSPS_ACT_TANH, "act_bwd_bias: unknown activation");
  LSPS_CHECK_ARG((((uintptr_t)dy | (uintptr_t)out | (uintptr_t)dx) & 15) == 0, "act_bwd_bias: pointers must be 16-byte aligned");
  const long total = (long)N * HW;
  long S = (total + 32767) / 32768;            // >= 32 K elements per block, at most 64 slices
  if (S > 64) S = 64;
  if (S < 1) S = 1;
  if (S > 1 && (ws == nullptr || (size_t)S * C * sizeof(float) > ws_bytes)) {
    set_error("act_bwd_bias: workspace too small: need %zu, have %zu", (size_t)S * C * sizeof(float), ws_bytes);
    return LSPS_E_WS;
  }
  if (total == 0) {
    hipError_t e = hipMemsetAsync(db, 0, (size_t)C * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) {
      set_error("hipMemsetAsync: %s", hipGetErrorString(e));
      return LSPS_E_HIP;
    }
    return 0;
  }
  long slice = (total + S - 1) / S;
  slice = (slice + 3) / 4 * 4;
  float *part = S == 1 ? db : (float *)ws;
  hipLaunchKernelGGL(act_bwd_bias_kernel, dim3(C, (int)S), dim3(256), 0, (hipStream_t)stream, dy, out, dx, part, N, C, HW,
                     slice, kind, slope);
  LSPS_CHECK_LAUNCH("act_bwd_bias");
  if (S > 1) {
    hipLaunchKernelGGL(sum_slices_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float *)part, db,
                       C, (int)S);
    LSPS_CHECK_LAUNCH("act_bwd_bias_reduce");
  }
  return 0;
}

int lsps_mul_add(const float *x, const float *t, const float *m, float *out, long n, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(t && m && out && n >= 0, "mul_add: bad argument");
  LSPS_CHECK_ARG((((uintptr_t)x | (uintptr_t)t | (uintptr_t)m | (uintptr_t)out) & 15) == 0,
                 "mul_add: pointers must be 16-byte aligned");
  if (n == 0) return 0;
  hipLaunchKernelGGL(mul_add_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, t, m, out, n);
  LSPS_CHECK_LAUNCH("mul_add");
  return 0;
}

static long bn_slices(long total) {
  long S = (total + 16383) / 16384;
  if (S > 64) S = 64;
  return S < 1 ? 1 : S;
}

size_t lsps_bnorm_workspace_bytes(int C) { return (size_t)64 * C * 2 * sizeof(double) + 2 * (size_t)C * sizeof(float) + 256; }

int lsps_bnorm_fwd(const float *x, const float *gamma, const float *beta, float *run_mean, float *run_var, float *y,
                   float *mean, float *rstd, int N, int C, int HW, int training, float eps, float momentum, float slope,
                   void *ws, size_t ws_bytes, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(x && y && mean && rstd && N > 0 && C > 0 && HW > 0, "bnorm_fwd: bad argument");
  LSPS_CHECK_ARG(training || (run_mean && run_var), "bnorm_fwd: eval mode needs running statistics");
  LSPS_CHECK_ARG(ws && ws_bytes >= lsps_bnorm_workspace_bytes(C), "bnorm_fwd: workspace too small");
  const long total = (long)N * HW;
  hipStream_t st = (hipStream_t)stream;
  if (training) {
    const long S = bn_slices(total);
    const long slice = (total + S - 1) / S;
    double *part = (double *)ws;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C, (int)S), dim3(256), 0, st, x, (const float *)nullptr, (const float *)nullptr,
                       part, N, C, HW, slice);
    LSPS_CHECK_LAUNCH("bn_stats");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const double *)part, (int)S, C, total, eps,
                       momentum, mean, rstd, run_mean, run_var);
    LSPS_CHECK_LAUNCH("bn_finalize");
  } else {
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const float *)run_mean,
                       (const float *)run_var, eps, C, mean, rstd);
    LSPS_CHECK_LAUNCH("bn_eval_stats");
  }
  const long n = total * C;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x, (const float *)mean, (const float *)rstd, gamma,
                     beta, y, n, C, HW, slope);
  LSPS_CHECK_LAUNCH("bn_apply");
  return 0;
}

int lsps_bnorm_bwd(const float *g, const float *x, const float *mean, const float *rstd, const float *gamma, float *dx,
                   float *dgamma, float *dbeta, int N, int C, int HW, int training, void *ws, size_t ws_bytes, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(g && x && mean && rstd && dx && N > 0 && C > 0 && HW > 0, "bnorm_bwd: bad argument");
  LSPS_CHECK_ARG(ws && ws_bytes >= lsps_bnorm_workspace_bytes(C), "bnorm_bwd: workspace too small");
  const long total = (long)N * HW;
  hipStream_t st = (hipStream_t)stream;
  const long S = bn_slices(total);
  const long slice = (total + S - 1) / S;
  double *part = (double *)ws;
  float *ma = (float *)((char *)ws + (size_t)64 * C * 2 * sizeof(double));
  float *mb = ma + C;
  hipLaunchKernelGGL(bn_stats_kernel, dim3(C, (int)S), dim3(256), 0, st, x, g, mean, part, N, C, HW, slice);
  LSPS_CHECK_LAUNCH("bn_bwd_stats");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const double *)part, (int)S, C, total,
                     rstd, dgamma, dbeta, ma, mb);
  LSPS_CHECK_LAUNCH("bn_bwd_finalize");
  const long n = total * C;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n)), dim3(256), 0, st, g, x, mean, rstd, gamma, (const float *)ma,
                     (const float *)mb, dx, n, C, HW, training);
  LSPS_CHECK_LAUNCH("bn_bwd_apply");
  return 0;
}

int lsps_act_fwd(const float *x, float *out, long n, int kind, float slope, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(x && out && n >= 0, "act_fwd: bad argument");
  LSPS_CHECK_ARG(kind == LSPS_ACT_LRELU || kind == LSPS_ACT_TANH || kind == LSPS_ACT_SOFTPLUS, "act_fwd: unknown activation");
  if (n == 0) return 0;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, out, n, kind, slope);
  LSPS_CHECK_LAUNCH("act_fwd");
  return 0;
}

int lsps_axpy(const float *x, const float *y, float alpha, float *out, long n, void *stream) {
  (void)hipGetLastError();   // clear stale sticky errors left by other users of the runtime
  LSPS_CHECK_ARG(x && y && out && n >= 0, "axpy: bad argument");
  LSPS_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)out) & 15) == 0, "axpy: pointers must be 16-byte aligned");
  if (n == 0) return 0;
  hipLaunchKernelGGL(axpy_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, alpha, out, n);
  LSPS_CHECK_LAUNCH("axpy");
  return 0;
}

}  // extern "C"
