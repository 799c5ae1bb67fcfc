// C-ABI of the bf16 residual trunk in the channel-group layout "C8" (c8conv.h, c8wgrad.h): BASELINE config 5.
#include "common.h"
#include "c8conv.h"
#include "c8wgrad.h"

namespace lsps {

static bool c8_geom_ok(int N, int C, int H, int W, int K) {
  return N > 0 && H == 32 && W == 32 && C >= 16 && (C & 15) == 0 && K >= 64 && (K & 63) == 0 && (long)(C >> 3) * 1024 * 16 < (1l << 31);
}

// packs w (K, C, 3, 3) for the forward (transposed = 0: M = K outputs, reduction over C) or the transposed direction
// (dgrad, transposed = 1: M = C outputs, reduction over K, taps flipped) into `ws` or the pack-cache scope
static int c8_pack(const float *w, int C, int K, int transposed, void *ws, size_t ws_bytes, hipStream_t st, const unsigned short **out) {
  C8Pack pp;
  pp.W = w;
  if (!transposed) {
    pp.M = K;
    pp.C = C;
    pp.sm = (long)C * 9;
    pp.sc = 9;
    for (int t = 0; t < 9; ++t) pp.tapidx[t] = t;
  } else {
    pp.M = C;
    pp.C = K;
    pp.sm = 9;
    pp.sc = (long)C * 9;
    for (int t = 0; t < 9; ++t) pp.tapidx[t] = 8 - t;
  }
  const size_t need = (size_t)pp.M * pp.C * 9 * sizeof(unsigned short);
  bool hit = false;
  void *slot = pack_cache_slot(w, /*tag: C8 layout*/ (1 << 22) + transposed, pp.M, pp.C, pp.sm, pp.sc, need, &hit, st);
  if (!slot) {
    if (ws_bytes < need || !ws) {
      set_error("c8 conv: workspace too small (%zu < %zu)", ws_bytes, need);
      return LSPS_E_ARG;
    }
    slot = ws;
  }
  *out = (const unsigned short *)slot;
  if (hit) return 0;
  pp.Wq = (unsigned short *)slot;
  const long total = (long)pp.M * pp.C * 9;
  hipLaunchKernelGGL(c8_pack_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, pp);
  LSPS_CHECK_LAUNCH("c8_pack");
  return 0;
}

static int c8_run(const void *x, const float *w, const void *r, void *y, float *rstd, int N, int Cx, int M, int transposed, int mode,
                  float slope, float eps, void *ws, size_t ws_bytes, hipStream_t st) {
  const unsigned short *wq = nullptr;
  // transposed: w is (K = Cx, C = M, 3, 3)
  if (int rc = c8_pack(w, transposed ? M : Cx, transposed ? Cx : M, transposed, ws, ws_bytes, st, &wq)) return rc;
  const void *fn[4] = {reinterpret_cast<const void *>(c8_conv3x3_kernel<0>), reinterpret_cast<const void *>(c8_conv3x3_kernel<1>),
                       reinterpret_cast<const void *>(c8_conv3x3_kernel<2>), reinterpret_cast<const void *>(c8_conv3x3_kernel<3>)};
  if (int rc = lds_optin(fn[mode], C8_LDS_BYTES, "c8_conv3x3")) return rc;
  C8ConvParams p;
  p.X = (const unsigned short *)x;
  p.Wq = wq;
  p.R = (const unsigned short *)r;
  p.Y = (unsigned short *)y;
  p.rstd = rstd;
  p.N = N;
  p.Cx = Cx;
  p.M = M;
  p.mode = mode;
  p.slope = slope;
  p.eps = eps;
  const int KT = M >> 6;
  const dim3 grid(((N + 7) / 8) * 8 * KT);
  switch (mode) {
    case 0: hipLaunchKernelGGL(c8_conv3x3_kernel<0>, grid, dim3(512), C8_LDS_BYTES, st, p); break;
    case 1: hipLaunchKernelGGL(c8_conv3x3_kernel<1>, grid, dim3(512), C8_LDS_BYTES, st, p); break;
    case 2: hipLaunchKernelGGL(c8_conv3x3_kernel<2>, grid, dim3(512), C8_LDS_BYTES, st, p); break;
    default: hipLaunchKernelGGL(c8_conv3x3_kernel<3>, grid, dim3(512), C8_LDS_BYTES, st, p); break;
  }
  LSPS_CHECK_LAUNCH("c8_conv3x3");
  return 0;
}

}  // namespace lsps

using namespace lsps;

extern "C" {

int lsps_c8_conv3x3_ok(int N, int C, int H, int W, int K) { return c8_geom_ok(N, C, H, W, K) ? 1 : 0; }

size_t lsps_c8_conv3x3_workspace_bytes(int C, int K) { return align_up((size_t)C * K * 9 * sizeof(unsigned short), 256); }

int lsps_c8_from_nchw(const float *x, void *y, int N, int C, int HW, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(x && y && N > 0 && C > 0 && (C & 7) == 0 && HW > 0, "c8_from_nchw: bad arguments (C must be a multiple of 8)");
  const long total = (long)N * (C >> 3) * HW;
  hipLaunchKernelGGL(c8_from_nchw_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, x, (unsigned short *)y, C, HW, total);
  LSPS_CHECK_LAUNCH("c8_from_nchw");
  return 0;
}

int lsps_c8_to_nchw(const void *x, float *y, int N, int C, int HW, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(x && y && N > 0 && C > 0 && (C & 7) == 0 && HW > 0, "c8_to_nchw: bad arguments (C must be a multiple of 8)");
  const long total = (long)N * (C >> 3) * HW;
  hipLaunchKernelGGL(c8_to_nchw_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned short *)x, y, C, HW, total);
  LSPS_CHECK_LAUNCH("c8_to_nchw");
  return 0;
}

int lsps_c8_add(const void *a, const void *b, void *out, long n, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(a && b && out && n > 0 && (n & 7) == 0, "c8_add: bad arguments");
  hipLaunchKernelGGL(c8_add_kernel, dim3(ceil_div(n >> 3, 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned short *)a,
                     (const unsigned short *)b, (unsigned short *)out, n >> 3);
  LSPS_CHECK_LAUNCH("c8_add");
  return 0;
}

int lsps_c8_conv3x3_fwd(const void *x, const float *w, const void *addend, void *y, int N, int C, int H, int W, int K, void *ws,
                        size_t ws_bytes, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(x && w && y, "c8_conv3x3_fwd: null pointer");
  LSPS_CHECK_ARG(c8_geom_ok(N, C, H, W, K), "c8_conv3x3_fwd: unsupported geometry (32x32 maps, C %% 16 == 0, K %% 64 == 0)");
  return c8_run(x, w, addend, y, nullptr, N, C, K, 0, 0, 0.f, 0.f, ws, ws_bytes, (hipStream_t)stream);
}

int lsps_c8_conv3x3_in_fwd(const void *x, const float *w, const void *residual, void *y, float *rstd, int N, int C, int H, int W,
                           int K, float slope, float eps, void *ws, size_t ws_bytes, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(x && w && y && rstd, "c8_conv3x3_in_fwd: null pointer");
  LSPS_CHECK_ARG(c8_geom_ok(N, C, H, W, K), "c8_conv3x3_in_fwd: unsupported geometry (32x32 maps, C %% 16 == 0, K %% 64 == 0)");
  return c8_run(x, w, residual, y, rstd, N, C, K, 0, residual ? 2 : 1, slope, eps, ws, ws_bytes, (hipStream_t)stream);
}

int lsps_c8_conv3x3_dgrad_acc(const void *dy, const float *w, const void *addend, void *dx, int N, int C, int H, int W, int K,
                              void *ws, size_t ws_bytes, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(dy && w && dx, "c8_conv3x3_dgrad_acc: null pointer");
  LSPS_CHECK_ARG(c8_geom_ok(N, K, H, W, C), "c8_conv3x3_dgrad_acc: unsupported geometry (32x32 maps, K %% 16 == 0, C %% 64 == 0)");
  return c8_run(dy, w, addend, dx, nullptr, N, K, C, 1, 0, 0.f, 0.f, ws, ws_bytes, (hipStream_t)stream);
}

int lsps_c8_conv3x3_dgrad_inbwd(const void *dy, const float *w, const void *out_saved, const float *rstd, void *dx, int N, int C,
                                int H, int W, int K, float slope, void *ws, size_t ws_bytes, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(dy && w && out_saved && rstd && dx, "c8_conv3x3_dgrad_inbwd: null pointer");
  LSPS_CHECK_ARG(slope > 0.f, "c8_conv3x3_dgrad_inbwd: needs a LeakyReLU slope > 0 (the input is recovered from the output)");
  LSPS_CHECK_ARG(c8_geom_ok(N, K, H, W, C), "c8_conv3x3_dgrad_inbwd: unsupported geometry (32x32 maps, K %% 16 == 0, C %% 64 == 0)");
  return c8_run(dy, w, out_saved, dx, const_cast<float *>(rstd), N, K, C, 1, 3, slope, 0.f, ws, ws_bytes, (hipStream_t)stream);
}

// splits of the weight gradient's image loop: enough workgroups to fill the chip (one workgroup per CU: 118 KB of LDS)
static int c8_wgrad_splits(int N, int C, int K) {
  const int tiles = (K >> 7) * (C >> 6);
  int s = (2 * 256 + tiles - 1) / tiles;           // ~2 workgroups per CU in the queue
  s = (s + 7) / 8 * 8;
  if (s > N) s = N;
  return s < 1 ? 1 : s;
}

size_t lsps_c8_conv3x3_wgrad_workspace_bytes(int N, int C, int K) {
  return align_up((size_t)c8_wgrad_splits(N, C, K) * 9 * K * C * sizeof(float), 256);
}

int lsps_c8_conv3x3_wgrad(const void *x, const void *dy, float *dw, int N, int C, int H, int W, int K, void *ws, size_t ws_bytes,
                          void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(x && dy && dw && ws, "c8_conv3x3_wgrad: null pointer");
  LSPS_CHECK_ARG(N > 0 && H == 32 && W == 32 && C >= 64 && (C & 63) == 0 && K >= 128 && (K & 127) == 0 &&
                     (long)(C >> 3) * 16384 < (1l << 31) && (long)(K >> 3) * 16384 < (1l << 31),
                 "c8_conv3x3_wgrad: unsupported geometry (32x32 maps, C %% 64 == 0, K %% 128 == 0)");
  LSPS_CHECK_ARG(ws_bytes >= lsps_c8_conv3x3_wgrad_workspace_bytes(N, C, K), "c8_conv3x3_wgrad: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (int rc = lds_optin(reinterpret_cast<const void *>(c8_wgrad_kernel), CW8_LDS_BYTES, "c8_wgrad")) return rc;
  C8WgradParams p;
  p.DY = (const unsigned short *)dy;
  p.X = (const unsigned short *)x;
  p.part = (float *)ws;
  p.N = N;
  p.K = K;
  p.C = C;
  p.splits = c8_wgrad_splits(N, C, K);
  p.imgs_per_split = (N + p.splits - 1) / p.splits;
  const int tiles = (K >> 7) * (C >> 6);
  hipLaunchKernelGGL(c8_wgrad_kernel, dim3((p.splits + 7) / 8 * 8 * tiles), dim3(512), CW8_LDS_BYTES, st, p);
  LSPS_CHECK_LAUNCH("c8_wgrad");
  hipLaunchKernelGGL(c8_wgrad_reduce_kernel, dim3(ceil_div((long)K * C, 256)), dim3(256), 0, st, (const float *)p.part, dw, K * C, p.splits);
  LSPS_CHECK_LAUNCH("c8_wgrad_reduce");
  return 0;
}

int lsps_c8_inorm_bwd(const void *dout, const void *out, const void *residual, const float *rstd, void *dy, int N, int C, int HW,
                      float slope, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(dout && out && rstd && dy && N > 0 && C > 0 && (C & 7) == 0 && HW > 0 && HW <= 1024,
                 "c8_inorm_bwd: bad arguments (C %% 8 == 0, planes of <= 1024 pixels)");
  LSPS_CHECK_ARG(residual || slope > 0.f, "c8_inorm_bwd: needs the residual or a LeakyReLU slope > 0");
  hipLaunchKernelGGL(c8_inorm_bwd_kernel, dim3(N * (C >> 3)), dim3(256), 0, (hipStream_t)stream, (const unsigned short *)dout,
                     (const unsigned short *)out, (const unsigned short *)residual, rstd, (unsigned short *)dy, HW, slope);
  LSPS_CHECK_LAUNCH("c8_inorm_bwd");
  return 0;
}

}  // extern "C"
