// Pose-MLP head (nn.Linear + LeakyReLU / Softplus; src/trainers/lsps_nets.py:44-50,73-83) and the
// Adam step over a flat parameter arena (torch.optim.Adam, src/trainers/lsps_trainer.py:26-29).
#include "common.h"

namespace lsps {

// y[r][o] = act(b[o] + sum_i x[r][i] W[o][i]) ; one thread per output element (14 k-parameter MLP)
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                         const float *__restrict__ b, float *__restrict__ y, int n,
                                                         int in, int out, int act, float slope) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)n * out) return;
  const int r = (int)(idx / out), o = (int)(idx - (long)r * out);
  const float *xr = x + (long)r * in, *wo = w + (long)o * in;
  float s = 0.f;
  for (int i = 0; i < in; ++i) s = fmaf(xr[i], wo[i], s);
  if (b) s += b[o];
  if (act == LSPS_ACT_LRELU)
    s = s > 0.f ? s : s * slope;
  else if (act == LSPS_ACT_SOFTPLUS)
    s = s > 20.f ? s : log1pf(expf(s));   // nn.Softplus(beta=1, threshold=20)
  y[idx] = s;
}

// dz = dy * act'(.) recovered from the saved output y
__global__ __launch_bounds__(256) void linear_dz_kernel(const float *__restrict__ y, const float *__restrict__ dy,
                                                        float *__restrict__ dz, long n, int act, float slope) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float o = y[i];
  float g = dy[i];
  if (act == LSPS_ACT_LRELU)
    g = o > 0.f ? g : g * slope;
  else if (act == LSPS_ACT_SOFTPLUS)
    g = o > 20.f ? g : g * (1.f - expf(-o));   // sigmoid(z) = 1 - exp(-softplus(z))
  dz[i] = g;
}

__global__ __launch_bounds__(256) void linear_dx_kernel(const float *__restrict__ dz, const float *__restrict__ w,
                                                        float *__restrict__ dx, int n, int in, int out) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)n * in) return;
  const int r = (int)(idx / in), i = (int)(idx - (long)r * in);
  float s = 0.f;
  for (int o = 0; o < out; ++o) s = fmaf(dz[(long)r * out + o], w[(long)o * in + i], s);
  dx[idx] = s;
}

// dW[o][i] = sum_r dz[r][o] x[r][i] ; db[o] = sum_r dz[r][o]   (idx over out*(in+1))
__global__ __launch_bounds__(256) void linear_dw_kernel(const float *__restrict__ dz, const float *__restrict__ x,
                                                        float *__restrict__ dw, float *__restrict__ db, int n, int in,
                                                        int out) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)out * (in + 1)) return;
  const int o = (int)(idx / (in + 1)), i = (int)(idx - (long)o * (in + 1));
  float s = 0.f;
  if (i < in) {
    for (int r = 0; r < n; ++r) s = fmaf(dz[(long)r * out + o], x[(long)r * in + i], s);
    dw[(long)o * in + i] = s;
  } else if (db) {
    for (int r = 0; r < n; ++r) s += dz[(long)r * out + o];
    db[o] = s;
  }
}

// Adam, torch.optim.Adam semantics (coupled L2 weight decay, bias-corrected), per-segment step.
#define ADAM_CHUNK 4096
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v,
                                                   const long *__restrict__ seg_off, const int *__restrict__ seg_len,
                                                   const float *__restrict__ seg_bc1, const float *__restrict__ seg_bc2s,
                                                   float lr, float b1, float b2, float eps, float wd, float gscale) {
  const int seg = blockIdx.y;
  const int len = seg_len[seg];
  const long start = (long)blockIdx.x * ADAM_CHUNK;
  if (start >= len) return;
  const long off = seg_off[seg];
  const float step_size = lr / seg_bc1[seg];
  const float inv_bc2s = 1.f / seg_bc2s[seg];
  long end = start + ADAM_CHUNK;
  if (end > len) end = len;
  for (long i = start + threadIdx.x; i < end; i += 256) {
    const long k = off + i;
    const float pk = p[k];
    const float gk = g[k] * gscale + wd * pk;
    const float mk = m[k] + (gk - m[k]) * (1.f - b1);          // exp_avg.lerp_(grad, 1-beta1)
    const float vk = v[k] * b2 + (1.f - b2) * gk * gk;         // mul_(beta2).addcmul_(grad, grad, 1-beta2)
    const float denom = sqrtf(vk) * inv_bc2s + eps;
    m[k] = mk;
    v[k] = vk;
    p[k] = pk - step_size * (mk / denom);
  }
}

}  // namespace lsps

using namespace lsps;

extern "C" {

int lsps_linear_fwd(const float *x, const float *w, const float *b, float *y, int n, int in, int out, int act,
                    float slope, void *stream) {
  (void)hipGetLastError();   // clear stale sticky errors left by other users of the runtime
  LSPS_CHECK_ARG(x && w && y && n > 0 && in > 0 && out > 0, "linear_fwd: bad argument");
  const long total = (long)n * out;
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, x, w, b, y, n, in,
                     out, act, slope);
  LSPS_CHECK_LAUNCH("linear_fwd");
  return 0;
}

int lsps_linear_bwd(const float *x, const float *w, const float *y, const float *dy, float *dx, float *dw, float *db,
                    int n, int in, int out, int act, float slope, float *ws_dz, void *stream) {
  (void)hipGetLastError();   // clear stale sticky errors left by other users of the runtime
  LSPS_CHECK_ARG(x && w && y && dy && dw && ws_dz && n > 0 && in > 0 && out > 0, "linear_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const long total = (long)n * out;
  hipLaunchKernelGGL(linear_dz_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, y, dy, ws_dz, total, act, slope);
  LSPS_CHECK_LAUNCH("linear_dz");
  if (dx) {
    hipLaunchKernelGGL(linear_dx_kernel, dim3(ceil_div((long)n * in, 256)), dim3(256), 0, st, (const float *)ws_dz, w,
                       dx, n, in, out);
    LSPS_CHECK_LAUNCH("linear_dx");
  }
  hipLaunchKernelGGL(linear_dw_kernel, dim3(ceil_div((long)out * (in + 1), 256)), dim3(256), 0, st,
                     (const float *)ws_dz, x, dw, db, n, in, out);
  LSPS_CHECK_LAUNCH("linear_dw");
  return 0;
}

int lsps_adam_step(float *p, const float *g, float *m, float *v, const long *seg_off, const int *seg_len,
                   const float *seg_bc1, const float *seg_bc2s, int nseg, int max_seg_len, float lr, float beta1,
                   float beta2, float eps, float weight_decay, float gscale, void *stream) {
  (void)hipGetLastError();   // clear stale sticky errors left by other users of the runtime
  LSPS_CHECK_ARG(p && g && m && v && seg_off && seg_len && seg_bc1 && seg_bc2s && nseg > 0 && max_seg_len > 0,
                 "adam_step: bad argument");
  LSPS_CHECK_ARG(nseg <= 65535, "adam_step: too many segments");
  dim3 grid(ceil_div(max_seg_len, ADAM_CHUNK), nseg);
  hipLaunchKernelGGL(adam_kernel, grid, dim3(256), 0, (hipStream_t)stream, p, g, m, v, seg_off, seg_len, seg_bc1,
                     seg_bc2s, lr, beta1, beta2, eps, weight_decay, gscale);
  LSPS_CHECK_LAUNCH("adam");
  return 0;
}

}  // extern "C"
