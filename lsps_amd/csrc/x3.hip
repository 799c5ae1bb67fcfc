// C-ABI of the three-limb ("X3") stride-2 conv prototype (x3s2.h): f32-class arithmetic on the bf16 matrix pipe.
#include <algorithm>
#include "common.h"
#include "x3s2.h"

namespace lsps {

static bool x3_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static bool x3s2_fwd_geom(int N, int Cx, int H, int W, int M, X3S2Params *p) {
  if (N <= 0 || H < 2 || W < 2 || !x3_pow2(H) || !x3_pow2(W) || (Cx & 15) || (M & 127)) return false;
  const int P = H / 2, Q = W / 2, PQ = P * Q;
  int TI, TR, tpi, nt;
  if (PQ >= 128) {
    if (Q > 128) return false;
    TI = 1; TR = 128 / Q; tpi = P / TR; nt = N * tpi;
  } else {
    TI = 128 / PQ; TR = P; tpi = 1; nt = (N + TI - 1) / TI;
  }
  const long bunits = 2l * TI * TR * (2 * Q + 1);
  if (bunits > X3F_BP * 64 || (long)TI * 3 * (Cx >> 3) * H * W * 16 >= (1l << 31)) return false;
  if ((long)(Cx >> 4) * 3 * X3F_ASTAGE >= (1l << 31)) return false;
  if (p) {
    p->H = H; p->W = W; p->P = P; p->Q = Q;
    p->TI = TI; p->TR = TR; p->tiles_per_img = tpi; p->ntiles = nt;
  }
  return true;
}

static int x3_device_cus() {
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    cus = prop.multiProcessorCount >= 8 ? prop.multiProcessorCount : 256;
  }
  return cus;
}

}  // namespace lsps

using namespace lsps;

extern "C" {

int lsps_x3_split_nchw(const float *x, void *xl, int N, int C, int HW, void *stream) {
  LSPS_CHECK_ARG(x && xl, "x3 split: null pointer");
  LSPS_CHECK_ARG(N > 0 && C > 0 && (C & 7) == 0 && HW > 0, "x3 split: C must be a multiple of 8");
  const long units = (long)N * (C >> 3) * HW;
  hipLaunchKernelGGL(x3_split_nchw_kernel, dim3(ceil_div(units, 256)), dim3(256), 0, (hipStream_t)stream, x, (unsigned short *)xl, C, HW,
                     units);
  LSPS_CHECK_LAUNCH("x3_split_nchw");
  return 0;
}

int lsps_x3_join_nchw(const void *xl, float *y, int N, int C, int HW, void *stream) {
  LSPS_CHECK_ARG(xl && y, "x3 join: null pointer");
  LSPS_CHECK_ARG(N > 0 && C > 0 && (C & 7) == 0 && HW > 0, "x3 join: C must be a multiple of 8");
  const long units = (long)N * (C >> 3) * HW;
  hipLaunchKernelGGL(x3_join_nchw_kernel, dim3(ceil_div(units, 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned short *)xl, y, C,
                     HW, units);
  LSPS_CHECK_LAUNCH("x3_join_nchw");
  return 0;
}

int lsps_x3_conv3x3s2_ok(int N, int C, int H, int W, int K) { return x3s2_fwd_geom(N, C, H, W, K, nullptr) ? 1 : 0; }

size_t lsps_x3_conv3x3s2_workspace_bytes(int N, int C, int H, int W, int K) {
  if (!x3s2_fwd_geom(N, C, H, W, K, nullptr)) return 0;
  return align_up((size_t)K * C * 9 * 3 * sizeof(unsigned short), 256);
}

int lsps_x3_conv3x3s2_fwd(const void *xl, const float *w, const float *bias, float *y, void *yl, int N, int C, int H, int W, int K,
                          float slope, void *ws, size_t ws_bytes, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  LSPS_CHECK_ARG(xl && w && (y || yl), "x3 conv: null pointer");
  X3S2Params p;
  LSPS_CHECK_ARG(x3s2_fwd_geom(N, C, H, W, K, &p), "x3 stride-2 conv: unsupported geometry N=%d C=%d %dx%d K=%d", N, C, H, W, K);
  LSPS_CHECK_ARG(!bias || !(reinterpret_cast<uintptr_t>(bias) & 15), "x3 stride-2 conv: the bias must be 16-byte aligned");
  const size_t need = (size_t)K * C * 9 * 3 * sizeof(unsigned short);
  bool hit = false;
  void *slot = pack_cache_slot(w, /*tag: X3 stride-2 layout*/ (1 << 24) + 128, K, C, (long)C * 9, 9, need, &hit, st);
  if (!slot) {
    LSPS_CHECK_ARG(ws && ws_bytes >= need, "x3 stride-2 conv: workspace too small (%zu < %zu)", ws_bytes, need);
    slot = ws;
  }
  if (!hit) {
    X3S2Pack pp;
    pp.W = w; pp.Wq = (unsigned short *)slot; pp.M = K; pp.C = C; pp.sm = (long)C * 9; pp.sc = 9;
    hipLaunchKernelGGL(x3s2_pack_kernel, dim3(ceil_div((long)K * C * 9, 256)), dim3(256), 0, st, pp);
    LSPS_CHECK_LAUNCH("x3s2_pack");
  }
  p.X = (const unsigned short *)xl;
  p.Wq = (const unsigned short *)slot;
  p.bias = bias;
  p.Y = y;
  p.YL = (unsigned short *)yl;
  p.N = N; p.Cx = C; p.M = K;
  p.lrelu = slope >= 0.f ? slope : 1.f;
  const dim3 grid(std::min((p.ntiles + 7) / 8 * 8 * (K >> 7), x3_device_cus() / 8 * 8));
  if (yl) {
    if (int rc = lds_optin(reinterpret_cast<const void *>(x3s2_fwd_kernel<true>), X3F_LDS_BYTES, "x3s2_fwd")) return rc;
    hipLaunchKernelGGL(x3s2_fwd_kernel<true>, grid, dim3(512), X3F_LDS_BYTES, st, p);
  } else {
    if (int rc = lds_optin(reinterpret_cast<const void *>(x3s2_fwd_kernel<false>), X3F_LDS_BYTES, "x3s2_fwd")) return rc;
    hipLaunchKernelGGL(x3s2_fwd_kernel<false>, grid, dim3(512), X3F_LDS_BYTES, st, p);
  }
  LSPS_CHECK_LAUNCH("x3s2_fwd");
  return 0;
}

}  // extern "C"
