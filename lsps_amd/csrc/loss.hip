// Loss reductions and their backward (KL / reconstruction / GAN / feature matching / regression).
// Reference: src/trainers/lsps_trainer.py:42-60 (L1Loss, _compute_l2_loss, _compute_kl),
// :107-112,:179-192 (sigmoid + binary_cross_entropy vs all-ones / all-zeros), :172-177,:241-243
// (feature matching L1 against a zero tensor), src/trainers/helpers.py:20-32 (accuracy counters).
// HBM-bound: one float4 pass, wavefront-shuffle + LDS block reduction, deterministic two-stage sum.
#include "common.h"

namespace lsps {

#define LOSS_MAX_BLOCKS 1024

__device__ __forceinline__ float loss_term(int kind, float a, float b) {
  if (kind == LSPS_LOSS_L1) return fabsf(a - b);
  if (kind == LSPS_LOSS_L2) {
    const float d = a - b;
    return d * d;
  }
  if (kind == LSPS_LOSS_SQ) return a * a;
  const float s2 = b * b;   // KLSD: a = mu, b = sd
  return a * a + s2 - logf(s2);
}

__global__ __launch_bounds__(256) void loss_partial_kernel(int kind, const float *__restrict__ a,
                                                           const float *__restrict__ b, long n,
                                                           float *__restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  const long stride = (long)gridDim.x * 256 * 4;
  const bool vec = ((((uintptr_t)a | (uintptr_t)b) & 15) == 0);
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (vec && i + 3 < n) {
      const float4 x = *reinterpret_cast<const float4 *>(a + i);
      float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b) y = *reinterpret_cast<const float4 *>(b + i);
      s += (loss_term(kind, x.x, y.x) + loss_term(kind, x.y, y.y)) + (loss_term(kind, x.z, y.z) + loss_term(kind, x.w, y.w));
    } else {
      for (long k = i; k < n && k < i + 4; ++k) s += loss_term(kind, a[k], b ? b[k] : 0.f);
    }
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// final stage: out[j] = sum_k partial[j*nblocks + k] * scale  (j < nout)
__global__ __launch_bounds__(256) void loss_final_kernel(const float *__restrict__ partial, int nblocks, int nout,
                                                         float scale0, float *__restrict__ out) {
  __shared__ float red[4];
  for (int j = 0; j < nout; ++j) {
    float s = 0.f;
    for (int k = threadIdx.x; k < nblocks; k += 256) s += partial[(long)j * nblocks + k];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) out[j] = j == 0 ? s * scale0 : s;
  }
}

__global__ __launch_bounds__(256) void loss_bwd_kernel(int kind, const float *__restrict__ a,
                                                       const float *__restrict__ b, long n, float inv_denom,
                                                       const float *__restrict__ gout, float *__restrict__ da,
                                                       float *__restrict__ db) {
  const float g = gout[0] * inv_denom;
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float x = a[i], y = b ? b[i] : 0.f;
    float ga, gb;
    if (kind == LSPS_LOSS_L1) {
      const float d = x - y;
      ga = d > 0.f ? g : (d < 0.f ? -g : 0.f);   // torch: sign(a-b), 0 at 0
      gb = -ga;
    } else if (kind == LSPS_LOSS_L2) {
      ga = 2.f * (x - y) * g;
      gb = -ga;
    } else if (kind == LSPS_LOSS_SQ) {
      ga = 2.f * x * g;
      gb = 0.f;
    } else {   // KLSD: d/dmu = 2 mu ; d/dsd = 2 sd - 2/sd
      ga = 2.f * x * g;
      gb = (2.f * y - 2.f / y) * g;
    }
    da[i] = ga;
    if (db) db[i] = gb;
  }
}

// sigmoid + BCE vs constant target, torch semantics: log clamped at -100.
__global__ __launch_bounds__(256) void bce_partial_kernel(const float *__restrict__ x, long n, float target,
                                                          float *__restrict__ partial, int nblocks) {
  __shared__ float red[4];
  float s = 0.f, ge = 0.f, le = 0.f;
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float p = 1.f / (1.f + expf(-x[i]));
    const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);
    s += -(target * lp + (1.f - target) * l1p);
    ge += p >= 0.5f ? 1.f : 0.f;
    le += p <= 0.5f ? 1.f : 0.f;
  }
  s = block_sum_256(s, red);
  ge = block_sum_256(ge, red);
  le = block_sum_256(le, red);
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = s;
    partial[nblocks + blockIdx.x] = ge;
    partial[2 * nblocks + blockIdx.x] = le;
  }
}

__global__ __launch_bounds__(256) void bce_bwd_kernel(const float *__restrict__ x, long n, float target,
                                                      const float *__restrict__ gout, float *__restrict__ dx) {
  const float g = gout[0] / (float)n;
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float p = 1.f / (1.f + expf(-x[i]));
    const float pq = p * (1.f - p);
    // torch: dL/dp = (p - t) / max(p(1-p), 1e-12) ; dp/dx = p(1-p)
    dx[i] = g * (p - target) / fmaxf(pq, 1e-12f) * pq;
  }
}

static int loss_blocks(long n) {
  long b = (n / 4 + 255) / 256;
  if (b > LOSS_MAX_BLOCKS) b = LOSS_MAX_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace lsps

using namespace lsps;

extern "C" {

size_t lsps_loss_workspace_bytes(long n) {
  (void)n;
  return (size_t)3 * LOSS_MAX_BLOCKS * sizeof(float);
}

int lsps_loss_fwd(int kind, const float *a, const float *b, long n, float denom, float *out, void *ws, size_t ws_bytes,
                  void *stream) {
  (void)hipGetLastError();   // clear stale sticky errors left by other users of the runtime
  LSPS_CHECK_ARG(a && out && ws && n > 0 && denom != 0.f, "loss_fwd: bad argument");
  LSPS_CHECK_ARG(kind >= LSPS_LOSS_L1 && kind <= LSPS_LOSS_KLSD, "loss_fwd: unknown kind");
  LSPS_CHECK_ARG(kind != LSPS_LOSS_KLSD || b, "loss_fwd: KLSD needs sd");
  if (ws_bytes < lsps_loss_workspace_bytes(n)) {
    set_error("loss_fwd: workspace too small");
    return LSPS_E_WS;
  }
  hipStream_t st = (hipStream_t)stream;
  const int nb = loss_blocks(n);
  hipLaunchKernelGGL(loss_partial_kernel, dim3(nb), dim3(256), 0, st, kind, a, b, n, (float *)ws);
  LSPS_CHECK_LAUNCH("loss_partial");
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, (const float *)ws, nb, 1, 1.f / denom, out);
  LSPS_CHECK_LAUNCH("loss_final");
  return 0;
}

int lsps_loss_bwd(int kind, const float *a, const float *b, long n, float denom, const float *gout, float *da,
                  float *db, void *stream) {
  (void)hipGetLastError();   // clear stale sticky errors left by other users of the runtime
  LSPS_CHECK_ARG(a && gout && da && n > 0 && denom != 0.f, "loss_bwd: bad argument");
  LSPS_CHECK_ARG(kind >= LSPS_LOSS_L1 && kind <= LSPS_LOSS_KLSD, "loss_bwd: unknown kind");
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(loss_bwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, kind, a, b, n, 1.f / denom,
                     gout, da, db);
  LSPS_CHECK_LAUNCH("loss_bwd");
  return 0;
}

int lsps_bce_sigmoid_fwd(const float *logits, long n, float target, float *out3, void *ws, size_t ws_bytes,
                         void *stream) {
  (void)hipGetLastError();   // clear stale sticky errors left by other users of the runtime
  LSPS_CHECK_ARG(logits && out3 && ws && n > 0, "bce_fwd: bad argument");
  if (ws_bytes < lsps_loss_workspace_bytes(n)) {
    set_error("bce_fwd: workspace too small");
    return LSPS_E_WS;
  }
  hipStream_t st = (hipStream_t)stream;
  long nb = (n + 255) / 256;
  if (nb > LOSS_MAX_BLOCKS) nb = LOSS_MAX_BLOCKS;
  hipLaunchKernelGGL(bce_partial_kernel, dim3((int)nb), dim3(256), 0, st, logits, n, target, (float *)ws, (int)nb);
  LSPS_CHECK_LAUNCH("bce_partial");
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, (const float *)ws, (int)nb, 3, 1.f / (float)n, out3);
  LSPS_CHECK_LAUNCH("bce_final");
  return 0;
}

int lsps_bce_sigmoid_bwd(const float *logits, long n, float target, const float *gout, float *dlogits, void *stream) {
  (void)hipGetLastError();   // clear stale sticky errors left by other users of the runtime
  LSPS_CHECK_ARG(logits && gout && dlogits && n > 0, "bce_bwd: bad argument");
  long nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(bce_bwd_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, logits, n, target, gout, dlogits);
  LSPS_CHECK_LAUNCH("bce_bwd");
  return 0;
}

}  // extern "C"
