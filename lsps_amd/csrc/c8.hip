// C-ABI of the bf16 residual trunk in the channel-group layout "C8" (c8conv.h, c8wgrad.h): BASELINE config 5.
#include <algorithm>
#include <stdlib.h>
#include "common.h"
#include "c8conv.h"
#include "c8wgrad.h"
#include "c8s2.h"
#include "c8ends.h"
#include "c8stem.h"

namespace lsps {

static int c8_wgrad_queue() {                     // workgroups per CU the weight-gradient kernels' grids aim at (experiments)
  return opts().c8w_queue;                          // default 1: one round of workgroups: half the partial sums of 2, measured faster (profiles/r3d)
}

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

// ------------------------------------------------------------------------------------------------------------------
// stride-2 3x3 family (c8s2.h)
// ------------------------------------------------------------------------------------------------------------------
static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// pixel tiles of the SMALL map: `px` consecutive (n, p, q)
static bool c8s2_tile(int N, int P, int Q, int px, int *TI, int *TR, int *tpi, int *ntiles) {
  const int PQ = P * Q;
  if (PQ >= px) {
    if (Q > px) return false;
    *TI = 1;
    *TR = px / Q;
    *tpi = P / *TR;
    *ntiles = N * *tpi;
  } else {
    *TI = px / PQ;
    *TR = P;
    *tpi = 1;
    *ntiles = (N + *TI - 1) / *TI;
  }
  return true;
}

// forward-direction geometry: NJ = 2 (256-pixel tiles) when the de-interleaved image of a tile fits the stage, else NJ = 1
static int c8s2_fwd_nj(int N, int Cx, int H, int W, int M, C8S2Params *p) {
  if (N <= 0 || H < 2 || W < 2 || !pow2(H) || !pow2(W) || (Cx & 15) || (M & 127)) return 0;
  const int P = H / 2, Q = W / 2;
  for (int nj = 2; nj >= 1; --nj) {
    int TI, TR, tpi, nt;
    if (!c8s2_tile(N, P, Q, 128 * nj, &TI, &TR, &tpi, &nt)) continue;
    const long bunits = 2l * TI * (2 * TR + 1) * (2 * Q + 1);
    const long bytes = (long)TI * (Cx >> 3) * H * W * 16;
    if (bunits > C8S2F_BPIECES * 64 || bytes >= (1l << 31)) continue;
    if (p) {
      p->H = H; p->W = W; p->P = P; p->Q = Q;
      p->TI = TI; p->TR = TR; p->tiles_per_img = tpi; p->ntiles = nt;
    }
    return nj;
  }
  return 0;
}

static bool c8s2_tr_geom(int N, int Cx, int H, int W, int M, C8S2Params *p) {      // H x W = the BIG (output) map
  if (N <= 0 || H < 2 || W < 2 || !pow2(H) || !pow2(W) || (Cx & 31) || (M & 63)) return false;
  const int P = H / 2, Q = W / 2;
  int TI, TR, tpi, nt;
  if (!c8s2_tile(N, P, Q, 256, &TI, &TR, &tpi, &nt)) return false;
  const long bunits = 2l * TI * (TR + 1) * (Q + 1);
  const long bytes = (long)TI * (Cx >> 3) * P * Q * 16;
  if (bunits > 1152 || bytes >= (1l << 31) || (long)(M >> 3) * H * W * 16 >= (1l << 40)) return false;
  if (p) {
    p->H = H; p->W = W; p->P = P; p->Q = Q;
    p->TI = TI; p->TR = TR; p->tiles_per_img = tpi; p->ntiles = nt;
  }
  return true;
}

static bool c8s2_wgrad_geom(int N, int K, int C, int H, int W, C8S2WParams *p) {   // small [N][K][H/2][W/2], big [N][C][H][W]
  if (N <= 0 || H < 2 || W < 2 || !pow2(H) || !pow2(W) || (K & 127) || (C & 63)) return false;
  const int P = H / 2, Q = W / 2;
  int TI, TR, cpi, nc;
  if (!c8s2_tile(N, P, Q, 64, &TI, &TR, &cpi, &nc)) return false;
  const int blk = (2 * TR + 1) * (2 * Q + 1);
  int bplane = TI * blk;
  while ((bplane & 7) != 4) ++bplane;                            // plane stride = 64 (mod 128) bytes: bank quarter rotation
  if (8 * bplane > C8S2W_BPIECES * 64) return false;
  if ((long)TI * (C >> 3) * H * W * 16 >= (1l << 31) || (long)TI * (K >> 3) * P * Q * 16 >= (1l << 31)) return false;
  const int tiles = (K >> 7) * (C >> 6);
  int s = (c8_wgrad_queue() * 256 + tiles - 1) / tiles;
  if (s > 1) s = (s + 7) / 8 * 8;                                // a split lives on ONE XCD (workgroup -> tile mapping): use all 8
  if (s > nc) s = nc;                                            // s == 1: the tiles alone fill the chip (kernel: plain tile mapping)
  const int cps = (nc + s - 1) / s;
  s = (nc + cps - 1) / cps;
  if (p) {
    p->N = N; p->K = K; p->C = C; p->H = H; p->W = W; p->P = P; p->Q = Q;
    p->TIW = TI; p->TRW = TR; p->chunks_per_img = cpi; p->nchunks = nc;
    p->splits = s; p->chunks_per_split = cps; p->bplane = bplane;
  }
  return true;
}

static int c8s2_pack(const float *w, int M, int C, long sm, long sc, int BM, void *ws, size_t ws_bytes, hipStream_t st,
                     const unsigned short **out) {
  const size_t need = (size_t)M * C * 9 * sizeof(unsigned short);
  bool hit = false;
  void *slot = pack_cache_slot(w, /*tag: C8 stride-2 layout*/ (1 << 23) + BM, M, C, sm, sc, need, &hit, st);
  if (!slot) {
    if (ws_bytes < need || !ws) {
      set_error("c8 stride-2 conv: workspace too small (%zu < %zu)", ws_bytes, need);
      return LSPS_E_ARG;
    }
    slot = ws;
  }
  *out = (const unsigned short *)slot;
  if (hit) return 0;
  C8S2Pack pp;
  pp.W = w;
  pp.Wq = (unsigned short *)slot;
  pp.M = M; pp.C = C; pp.BM = BM; pp.sm = sm; pp.sc = sc;
  hipLaunchKernelGGL(c8s2_pack_kernel, dim3(ceil_div((long)M * C / 8, 256)), dim3(256), 0, st, pp);
  LSPS_CHECK_LAUNCH("c8s2_pack");
  return 0;
}

// out[c] = sum over `rows` rows of part[rows][C]; `scratch` >= 64 * C floats when rows > 64
static int c8_colsum(const float *part, float *out, int C, int rows, float *scratch, hipStream_t st) {
  if (rows > 64) {
    const int rpc = (rows + 63) / 64, chunks = (rows + rpc - 1) / rpc;
    hipLaunchKernelGGL(c8_colsum_stage1_kernel, dim3(ceil_div(C, 64), chunks), dim3(256), 0, st, part, scratch, C, rows, rpc);
    LSPS_CHECK_LAUNCH("c8_colsum_stage1");
    part = scratch;
    rows = chunks;
  }
  // the last <= 64 rows: the stage-1 kernel with ONE chunk (four row lanes per channel: 16 dependent loads instead of 64 —
  // 15.7 -> 5 us per call, ~16 calls per config-5 step)
  hipLaunchKernelGGL(c8_colsum_stage1_kernel, dim3(ceil_div(C, 64), 1), dim3(256), 0, st, part, out, C, rows, rows);
  LSPS_CHECK_LAUNCH("c8_colsum");
  return 0;
}

// bias-gradient partial sums of a fused-activation dgrad: [ntiles][M] + [64][M] floats at the END of the workspace (the packed
// weights, if they are not in the pack-cache scope, sit at its start)
static float *c8s2_dbpart(void *ws, size_t ws_bytes, int ntiles, int M, size_t pack_bytes) {
  const size_t need = ((size_t)ntiles + 64) * M * sizeof(float);
  if (!ws || ws_bytes < align_up(pack_bytes, 256) + need) return nullptr;
  return reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + ((ws_bytes - need) & ~(size_t)255));
}

// small[n][m] = act(bias + sum Wt[m][kk][r][s] big[n][kk][2p+r-1][2q+s-1]);  W element (m, kk, t) at m*sm + kk*sc + t
static int c8_device_cus() {
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    cus = prop.multiProcessorCount >= 8 ? prop.multiProcessorCount : 256;
  }
  return cus;
}

static int c8s2_run_fwd(const void *big, const float *w, long sm, long sc, const float *bias, void *small, int N, int Cx, int H, int W,
                        int M, float slope, void *ws, size_t ws_bytes, hipStream_t st, const void *act_y = nullptr, float act_slope = 0.f,
                        float *db_prev = nullptr) {
  C8S2Params p;
  const int nj = c8s2_fwd_nj(N, Cx, H, W, M, &p);
  if (!nj) {
    set_error("c8 stride-2 conv (forward direction): unsupported geometry N=%d C=%d %dx%d M=%d", N, Cx, H, W, M);
    return LSPS_E_ARG;
  }
  if (bias && (reinterpret_cast<uintptr_t>(bias) & 15)) {     // the epilogue reads it in 16-byte pieces
    set_error("c8 stride-2 conv: the bias must be 16-byte aligned");
    return LSPS_E_ARG;
  }
  const unsigned short *wq = nullptr;
  if (int rc = c8s2_pack(w, M, Cx, sm, sc, 128, ws, ws_bytes, st, &wq)) return rc;
  p.X = (const unsigned short *)big;
  p.Wq = wq;
  p.bias = bias;
  p.Y = (unsigned short *)small;
  p.N = N; p.Cx = Cx; p.M = M;
  p.lrelu = slope >= 0.f ? slope : 1.f;
  p.ActY = (const unsigned short *)act_y;
  p.act_slope = act_slope;
  p.dbpart = nullptr;
  if (act_y) {
    p.dbpart = c8s2_dbpart(ws, ws_bytes, p.ntiles, M, (size_t)M * Cx * 9 * sizeof(unsigned short));
    if (!p.dbpart) {
      set_error("c8 stride-2 conv: workspace too small for the bias-gradient partial sums");
      return LSPS_E_ARG;
    }
  }
  // persistent workgroups (one per CU: the kernel's LDS footprint admits no second), each walking tiles grid apart
  const dim3 grid(std::min((p.ntiles + 7) / 8 * 8 * (M >> 7), c8_device_cus() / 8 * 8));
  if (nj == 2) {
    if (int rc = lds_optin(reinterpret_cast<const void *>(c8s2_fwd_kernel<2>), C8S2F_LDS_BYTES, "c8s2_fwd")) return rc;
    hipLaunchKernelGGL(c8s2_fwd_kernel<2>, grid, dim3(512), C8S2F_LDS_BYTES, st, p);
  } else {
    if (int rc = lds_optin(reinterpret_cast<const void *>(c8s2_fwd_kernel<1>), C8S2F_LDS_BYTES, "c8s2_fwd")) return rc;
    hipLaunchKernelGGL(c8s2_fwd_kernel<1>, grid, dim3(512), C8S2F_LDS_BYTES, st, p);
  }
  LSPS_CHECK_LAUNCH("c8s2_fwd");
  if (act_y && db_prev) return c8_colsum(p.dbpart, db_prev, M, p.ntiles, p.dbpart + (size_t)p.ntiles * M, st);
  return 0;
}

// big[n][m][2p+r-1][2q+s-1] += Wt[m][kk][r][s] small[n][kk][p][q] (+ bias, activation); H x W = the big map
static int c8s2_run_tr(const void *small, const float *w, long sm, long sc, const float *bias, void *big, int N, int Cx, int H, int W,
                       int M, float slope, void *ws, size_t ws_bytes, hipStream_t st, const void *act_y = nullptr, float act_slope = 0.f,
                       float *db_prev = nullptr) {
  C8S2Params p;
  if (!c8s2_tr_geom(N, Cx, H, W, M, &p)) {
    set_error("c8 stride-2 conv (transposed direction): unsupported geometry N=%d C=%d -> %dx%d M=%d", N, Cx, H, W, M);
    return LSPS_E_ARG;
  }
  if (bias && (reinterpret_cast<uintptr_t>(bias) & 15)) {
    set_error("c8 stride-2 conv: the bias must be 16-byte aligned");
    return LSPS_E_ARG;
  }
  const unsigned short *wq = nullptr;
  if (int rc = c8s2_pack(w, M, Cx, sm, sc, 64, ws, ws_bytes, st, &wq)) return rc;
  p.X = (const unsigned short *)small;
  p.Wq = wq;
  p.bias = bias;
  p.Y = (unsigned short *)big;
  p.N = N; p.Cx = Cx; p.M = M;
  p.lrelu = slope >= 0.f ? slope : 1.f;
  p.ActY = (const unsigned short *)act_y;
  p.act_slope = act_slope;
  p.dbpart = nullptr;
  if (act_y) {
    p.dbpart = c8s2_dbpart(ws, ws_bytes, p.ntiles, M, (size_t)M * Cx * 9 * sizeof(unsigned short));
    if (!p.dbpart) {
      set_error("c8 stride-2 conv: workspace too small for the bias-gradient partial sums");
      return LSPS_E_ARG;
    }
  }
  // persistent workgroups, one per CU
  const dim3 grid(std::min((p.ntiles + 7) / 8 * 8 * (M >> 6), c8_device_cus() / 8 * 8));
  if (act_y) {
    if (int rc = lds_optin(reinterpret_cast<const void *>(c8s2_tr_kernel<true>), C8S2T_LDS_BYTES, "c8s2_tr")) return rc;
    hipLaunchKernelGGL(c8s2_tr_kernel<true>, grid, dim3(512), C8S2T_LDS_BYTES, st, p);
  } else {
    if (int rc = lds_optin(reinterpret_cast<const void *>(c8s2_tr_kernel<false>), C8S2T_LDS_BYTES, "c8s2_tr")) return rc;
    hipLaunchKernelGGL(c8s2_tr_kernel<false>, grid, dim3(512), C8S2T_LDS_BYTES, st, p);
  }
  LSPS_CHECK_LAUNCH("c8s2_tr");
  if (act_y && db_prev) return c8_colsum(p.dbpart, db_prev, M, p.ntiles, p.dbpart + (size_t)p.ntiles * M, st);
  return 0;
}

static int c8s2_run_wgrad(const void *small, const void *big, float *dw, int N, int K, int C, int H, int W, void *ws, size_t ws_bytes,
                          hipStream_t st) {
  C8S2WParams p;
  if (!c8s2_wgrad_geom(N, K, C, H, W, &p)) {
    set_error("c8 stride-2 weight gradient: unsupported geometry N=%d K=%d C=%d %dx%d", N, K, C, H, W);
    return LSPS_E_ARG;
  }
  const size_t need = (size_t)p.splits * 9 * K * C * sizeof(float);
  if (!ws || ws_bytes < need) {
    set_error("c8 stride-2 weight gradient: workspace too small (%zu < %zu)", ws_bytes, need);
    return LSPS_E_ARG;
  }
  p.S = (const unsigned short *)small;
  p.B = (const unsigned short *)big;
  p.part = (float *)ws;
  if (int rc = lds_optin(reinterpret_cast<const void *>(c8s2_wgrad_kernel), C8S2W_LDS_BYTES, "c8s2_wgrad")) return rc;
  const int tiles = (K >> 7) * (C >> 6);
  hipLaunchKernelGGL(c8s2_wgrad_kernel, dim3(p.splits == 1 ? tiles : (p.splits + 7) / 8 * 8 * tiles), dim3(512), C8S2W_LDS_BYTES, st, p);
  LSPS_CHECK_LAUNCH("c8s2_wgrad");
  hipLaunchKernelGGL(c8_wgrad_reduce_kernel, dim3(ceil_div((long)K * C, 64)), dim3(256), 0, st, (const float *)p.part, dw, K * C, p.splits);
  LSPS_CHECK_LAUNCH("c8s2_wgrad_reduce");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// one-input-channel stems on the bf16 matrix pipe (c8stem.h)
// ------------------------------------------------------------------------------------------------------------------
#define C8SW_BLOCKS 512

bool c8_stem_bf16_ok(int N, int H, int W, int K, int R, int S, int stride, int pad) {
  const int on = opts().c8_stem_bf16;
  if (!on || N <= 0 || K != 64 || R > 7 || S > 7 || R < 1 || S < 1 || (stride != 1 && stride != 2) || pad < 0) return false;
  const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
  if (P <= 0 || (Q != 32 && Q != 64 && Q != 128) || (P % (128 / Q)) != 0) return false;
  const int LW = W + 2 * pad + 2, RB = 128 / Q;
  return ((RB - 1) * stride + 8) * LW <= C8SW_XMAX && (long)K * P * Q * 2 < (1L << 31);
}

int c8_stem_fwd_bf16(const float *x, const float *w, const float *bias, void *y, int N, int H, int W, int K, int R, int S, int stride,
                     int pad, float slope, hipStream_t st) {
  C8StemParams p;
  p.X = x; p.W = w; p.bias = bias; p.Y = (unsigned short *)y;
  p.N = N; p.H = H; p.Wd = W; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
  p.P = (H + 2 * pad - R) / stride + 1;
  p.Q = (W + 2 * pad - S) / stride + 1;
  p.LW = W + 2 * pad + 2;
  int tp = 8;                                         // output rows per workgroup: >= ~1024 workgroups, <= 64 KB of staged rows
  while (tp > 1 && ((p.P % tp) != 0 || (long)N * (p.P / tp) < 1024 || (long)((tp - 1) * stride + 8) * p.LW * 4 > 60000)) tp >>= 1;
  p.TP = tp;
  p.rows = (tp - 1) * stride + 8;
  p.lrelu = slope >= 0.f ? slope : 1.f;
  hipLaunchKernelGGL(c8_stem_fwd_kernel, dim3(p.P / tp, N), dim3(256), (size_t)p.rows * p.LW * sizeof(float), st, p);
  LSPS_CHECK_LAUNCH("c8_stem_fwd");
  return 0;
}

int c8_stem_dgrad_bf16(const void *dy, const void *y, const float *w, float *dx, int N, int H, int W, int K, int R, int S, int stride,
                       int pad, float slope, hipStream_t st) {
  C8StemDParams p;
  p.DY = (const unsigned short *)dy; p.Yc = (const unsigned short *)y; p.W = w; p.dX = dx;
  p.N = N; p.H = H; p.Wd = W; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
  p.P = (H + 2 * pad - R) / stride + 1;
  p.Q = (W + 2 * pad - S) / stride + 1;
  p.LW = (p.Q - 1) * stride + 8 + 1;                   // widest LDS column touched: (Q - 1) stride + 7
  if (p.LW < W + pad) p.LW = W + pad;
  p.slope = slope;
  const size_t lds = ((size_t)4 * C8SD_TY * p.LW + 256) * sizeof(float);
  if (int rc = lds_optin(reinterpret_cast<const void *>(c8_stem_dgrad_kernel), (int)lds, "c8_stem_dgrad")) return rc;
  hipLaunchKernelGGL(c8_stem_dgrad_kernel, dim3(ceil_div(H, C8SD_TY), N), dim3(256), lds, st, p);
  LSPS_CHECK_LAUNCH("c8_stem_dgrad");
  return 0;
}

size_t c8_stem_wgrad_bf16_ws_bytes() { return (size_t)C8SW_BLOCKS * 4096 * sizeof(float) + 256; }

int c8_stem_wgrad_bf16(const float *x, const void *dy, const void *y, float *dw, float *db, int N, int H, int W, int K, int R, int S,
                       int stride, int pad, float slope, void *ws, size_t ws_bytes, hipStream_t st) {
  if (!ws || ws_bytes < c8_stem_wgrad_bf16_ws_bytes()) {
    set_error("c8 stem weight gradient: workspace too small (%zu < %zu)", ws_bytes, c8_stem_wgrad_bf16_ws_bytes());
    return LSPS_E_WS;
  }
  C8StemWParams p;
  p.X = x; p.DY = (const unsigned short *)dy; p.Yc = (const unsigned short *)y; p.part = (float *)ws;
  p.N = N; p.H = H; p.Wd = W; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
  p.P = (H + 2 * pad - R) / stride + 1;
  p.Q = (W + 2 * pad - S) / stride + 1;
  p.LW = W + 2 * pad + 2;
  p.RB = 128 / p.Q;
  p.xrows = (p.RB - 1) * stride + 8;
  p.iters_total = N * (p.P / p.RB);
  int blocks = std::min(p.iters_total, C8SW_BLOCKS);
  p.iters_per_block = ceil_div(p.iters_total, blocks);
  blocks = ceil_div(p.iters_total, p.iters_per_block);
  p.slope = slope;
  hipLaunchKernelGGL(c8_stem_wgrad_kernel, dim3(blocks), dim3(256), 0, st, p);
  LSPS_CHECK_LAUNCH("c8_stem_wgrad");
  hipLaunchKernelGGL(c8_stem_wgrad_reduce_kernel, dim3(64), dim3(256), 0, st, (const float *)p.part, dw, db, R, S, blocks);
  LSPS_CHECK_LAUNCH("c8_stem_wgrad_reduce");
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

int lsps_c8_add_nchw(const void *a, const float *b, void *out, int N, int C, int HW, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(a && b && out && N > 0 && C > 0 && (C & 7) == 0 && HW > 0, "c8_add_nchw: bad arguments (C %% 8 == 0)");
  const long units = (long)N * (C >> 3) * HW;
  hipLaunchKernelGGL(c8_add_nchw_kernel, dim3(ceil_div(units, 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned short *)a, b,
                     (unsigned short *)out, HW, units);
  LSPS_CHECK_LAUNCH("c8_add_nchw");
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
  int s = (c8_wgrad_queue() * 256 + tiles - 1) / tiles;
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
  hipLaunchKernelGGL(c8_wgrad_reduce_kernel, dim3(ceil_div((long)K * C, 64)), dim3(256), 0, st, (const float *)p.part, dw, K * C, p.splits);
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

// ---- stride-2 3x3 family ------------------------------------------------------------------------------------------
int lsps_c8_conv3x3s2_ok(int N, int C, int H, int W, int K) {
  return c8s2_fwd_nj(N, C, H, W, K, nullptr) && c8s2_tr_geom(N, K, H, W, C, nullptr) && c8s2_wgrad_geom(N, K, C, H, W, nullptr) ? 1 : 0;
}

int lsps_c8_convT3x3s2_ok(int N, int Ci, int H, int W, int Co) {      // x [N, Ci, H, W] -> y [N, Co, 2H, 2W]
  return c8s2_tr_geom(N, Ci, 2 * H, 2 * W, Co, nullptr) && c8s2_fwd_nj(N, Co, 2 * H, 2 * W, Ci, nullptr) &&
                 c8s2_wgrad_geom(N, Ci, Co, 2 * H, 2 * W, nullptr)
             ? 1
             : 0;
}

// packed weights (forward / transposed direction) or the weight gradient's partial sums, whichever is larger; H x W = big map
size_t lsps_c8_conv3x3s2_workspace_bytes(int N, int C, int H, int W, int K) {
  size_t need = (size_t)C * K * 9 * sizeof(unsigned short);
  {                                    // fused-activation dgrad forms: packed weights + [ntiles + 64][channels] partial sums
    C8S2Params q;
    size_t part = 0;
    if (c8s2_tr_geom(N, K, H, W, C, &q)) part = std::max(part, ((size_t)q.ntiles + 64) * C * sizeof(float));          // conv dgrad -> dx [C]
    if (c8s2_fwd_nj(N, C, H, W, K, &q)) part = std::max(part, ((size_t)q.ntiles + 64) * K * sizeof(float));          // convT dgrad -> dx [K = Ci]
    need = align_up(need, 256) + part + 256;
  }
  C8S2WParams p;
  if (c8s2_wgrad_geom(N, K, C, H, W, &p)) need = std::max(need, (size_t)p.splits * 9 * K * C * sizeof(float));
  return align_up(need, 256);
}

int lsps_c8_conv3x3s2_fwd(const void *x, const float *w, const float *bias, void *y, int N, int C, int H, int W, int K, float slope,
                          void *ws, size_t ws_bytes, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(x && w && y, "c8_conv3x3s2_fwd: null pointer");
  return c8s2_run_fwd(x, w, (long)C * 9, 9, bias, y, N, C, H, W, K, slope, ws, ws_bytes, (hipStream_t)stream);
}

int lsps_c8_conv3x3s2_dgrad(const void *dy, const float *w, void *dx, int N, int C, int H, int W, int K, void *ws, size_t ws_bytes,
                            void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(dy && w && dx, "c8_conv3x3s2_dgrad: null pointer");
  return c8s2_run_tr(dy, w, 9, (long)C * 9, nullptr, dx, N, K, H, W, C, -1.f, ws, ws_bytes, (hipStream_t)stream);
}

int lsps_c8_conv3x3s2_wgrad(const void *x, const void *dy, float *dw, int N, int C, int H, int W, int K, void *ws, size_t ws_bytes,
                            void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(x && dy && dw, "c8_conv3x3s2_wgrad: null pointer");
  return c8s2_run_wgrad(dy, x, dw, N, K, C, H, W, ws, ws_bytes, (hipStream_t)stream);
}

int lsps_c8_convT3x3s2_fwd(const void *x, const float *w, const float *bias, void *y, int N, int Ci, int H, int W, int Co, float slope,
                           void *ws, size_t ws_bytes, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(x && w && y, "c8_convT3x3s2_fwd: null pointer");
  return c8s2_run_tr(x, w, 9, (long)Co * 9, bias, y, N, Ci, 2 * H, 2 * W, Co, slope, ws, ws_bytes, (hipStream_t)stream);
}

int lsps_c8_convT3x3s2_dgrad(const void *dy, const float *w, void *dx, int N, int Ci, int H, int W, int Co, void *ws, size_t ws_bytes,
                             void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(dy && w && dx, "c8_convT3x3s2_dgrad: null pointer");
  return c8s2_run_fwd(dy, w, (long)Co * 9, 9, nullptr, dx, N, Co, 2 * H, 2 * W, Ci, -1.f, ws, ws_bytes, (hipStream_t)stream);
}

int lsps_c8_convT3x3s2_wgrad(const void *x, const void *dy, float *dw, int N, int Ci, int H, int W, int Co, void *ws, size_t ws_bytes,
                             void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(x && dy && dw, "c8_convT3x3s2_wgrad: null pointer");
  return c8s2_run_wgrad(x, dy, dw, N, Ci, Co, 2 * H, 2 * W, ws, ws_bytes, (hipStream_t)stream);
}

size_t lsps_c8_act_bwd_bias_workspace_bytes(int N, int C) { return align_up((size_t)std::min(N, 64) * C * sizeof(float), 256); }

int lsps_c8_act_bwd_bias(const void *dy, const void *y, void *g, float *db, int N, int C, int HW, float slope, void *ws, size_t ws_bytes,
                         void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(dy && y && g && N > 0 && C > 0 && (C & 7) == 0 && HW > 0, "c8_act_bwd_bias: bad arguments (C %% 8 == 0)");
  LSPS_CHECK_ARG(slope >= 0.f, "c8_act_bwd_bias: LeakyReLU slope >= 0");
  // enough workgroups to fill the chip: (C / 8) x splits >= ~1024
  int splits = std::min(N, std::max(1, 1024 / (C >> 3)));
  splits = std::min(splits, 64);
  const int ips = (N + splits - 1) / splits;
  splits = (N + ips - 1) / ips;
  LSPS_CHECK_ARG(ws && ws_bytes >= (size_t)splits * C * sizeof(float), "c8_act_bwd_bias: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(c8_act_bwd_bias_kernel, dim3(C >> 3, splits), dim3(256), 0, st, (const unsigned short *)dy, (const unsigned short *)y,
                     (unsigned short *)g, (float *)ws, N, C, HW, ips, slope);
  LSPS_CHECK_LAUNCH("c8_act_bwd_bias");
  if (db) {
    hipLaunchKernelGGL(c8_colsum_stage1_kernel, dim3(ceil_div(C, 64), 1), dim3(256), 0, st, (const float *)ws, db, C, splits, splits);
    LSPS_CHECK_LAUNCH("c8_colsum");
  }
  return 0;
}

// ---- the generator's 1x1 output head on a C8 tensor (c8ends.h) ------------------------------------------------------
static int c8_pw1_splits(int N, int C) {
  int s = std::min(N, std::max(1, 2048 / (C >> 3)));
  const int ips = (N + s - 1) / s;
  return (N + ips - 1) / ips;
}

size_t lsps_c8_pw1_workspace_bytes(int N, int C) { return align_up((size_t)c8_pw1_splits(N, C) * (C + 1) * sizeof(float), 256); }

int lsps_c8_pw1_fwd(const void *x, const float *w, const float *bias, float *y, int N, int C, int HW, int act, float slope, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(x && w && y && N > 0 && C > 0 && (C & 7) == 0 && HW > 0, "c8_pw1_fwd: bad arguments (C %% 8 == 0)");
  hipLaunchKernelGGL(c8_pw1_fwd_kernel, dim3(ceil_div(HW, 256), N), dim3(256), 0, (hipStream_t)stream, (const unsigned short *)x, w, bias, y,
                     C, HW, act, slope);
  LSPS_CHECK_LAUNCH("c8_pw1_fwd");
  return 0;
}

int lsps_c8_pw1_dgrad(const float *dpre, const float *w, void *dx, int N, int C, int HW, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(dpre && w && dx && N > 0 && C > 0 && (C & 7) == 0 && HW > 0, "c8_pw1_dgrad: bad arguments (C %% 8 == 0)");
  hipLaunchKernelGGL(c8_pw1_dgrad_kernel, dim3(ceil_div(HW, 256), N), dim3(256), 0, (hipStream_t)stream, dpre, w, (unsigned short *)dx, C,
                     HW);
  LSPS_CHECK_LAUNCH("c8_pw1_dgrad");
  return 0;
}

int lsps_c8_pw1_wgrad(const void *x, const float *dpre, float *dw, float *db, int N, int C, int HW, void *ws, size_t ws_bytes,
                      void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(x && dpre && dw && N > 0 && C > 0 && (C & 7) == 0 && HW > 0, "c8_pw1_wgrad: bad arguments (C %% 8 == 0)");
  const int splits = c8_pw1_splits(N, C), ips = (N + splits - 1) / splits;
  LSPS_CHECK_ARG(ws && ws_bytes >= (size_t)splits * (C + 1) * sizeof(float), "c8_pw1_wgrad: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(c8_pw1_wgrad_kernel, dim3(C >> 3, splits), dim3(256), 0, st, (const unsigned short *)x, dpre, (float *)ws, N, C, HW, ips);
  LSPS_CHECK_LAUNCH("c8_pw1_wgrad");
  hipLaunchKernelGGL(c8_pw1_wgrad_reduce_kernel, dim3(C + 1), dim3(256), 0, st, (const float *)ws, dw, db, C, splits);
  LSPS_CHECK_LAUNCH("c8_pw1_wgrad_reduce");
  return 0;
}

// ---- dgrad entries with the PREVIOUS layer's LeakyReLU backward (+ its bias gradient) fused into the epilogue -----------
int lsps_c8_conv3x3s2_dgrad_act(const void *dy, const float *w, const void *act_y, float act_slope, void *dx, float *db_prev, int N, int C,
                                int H, int W, int K, void *ws, size_t ws_bytes, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(dy && w && dx && act_y && act_slope >= 0.f, "c8_conv3x3s2_dgrad_act: null pointer / slope < 0");
  return c8s2_run_tr(dy, w, 9, (long)C * 9, nullptr, dx, N, K, H, W, C, -1.f, ws, ws_bytes, (hipStream_t)stream, act_y, act_slope, db_prev);
}

int lsps_c8_convT3x3s2_dgrad_act(const void *dy, const float *w, const void *act_y, float act_slope, void *dx, float *db_prev, int N,
                                 int Ci, int H, int W, int Co, void *ws, size_t ws_bytes, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(dy && w && dx && act_y && act_slope >= 0.f, "c8_convT3x3s2_dgrad_act: null pointer / slope < 0");
  return c8s2_run_fwd(dy, w, (long)Co * 9, 9, nullptr, dx, N, Co, 2 * H, 2 * W, Ci, -1.f, ws, ws_bytes, (hipStream_t)stream, act_y,
                      act_slope, db_prev);
}

// [rows][C] partials of the previous layer's bias gradient + [64][C] scratch + [rows][C + 1] partials of the head's own gradients
// (rows = N x up to 8 pixel segments per image)
size_t lsps_c8_pw1_dgrad_act_workspace_bytes(int N, int C) { return align_up(((size_t)16 * N + 64) * (C + 1) * sizeof(float), 256); }

int lsps_c8_pw1_dgrad_act(const float *dpre, const float *w, const void *act_y, float act_slope, void *dx, float *db_prev, float *dw,
                          float *db, int N, int C, int HW, void *ws, size_t ws_bytes, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(dpre && w && act_y && dx && N > 0 && C > 0 && (C & 7) == 0 && C <= 64 && HW > 0 && act_slope >= 0.f,
                 "c8_pw1_dgrad_act: bad arguments (C %% 8 == 0, C <= 64)");
  LSPS_CHECK_ARG(ws && ws_bytes >= lsps_c8_pw1_dgrad_act_workspace_bytes(N, C) - 256, "c8_pw1_dgrad_act: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  // pixel segments per image: >= ~4096 workgroups on the chip, >= 512 pixels each
  int S = 1;
  while (S < 8 && (long)N * S < 4096 && HW / (2 * S) >= 512) S *= 2;
  const int rows = N * S;
  float *part = (float *)ws, *scratch = part + (size_t)rows * C, *wpart = dw ? scratch + (size_t)64 * (C + 1) : nullptr;
  hipLaunchKernelGGL(c8_pw1_dgrad_act_kernel, dim3(rows), dim3(256), 0, st, dpre, w, (const unsigned short *)act_y, (unsigned short *)dx,
                     part, wpart, C, HW, S, act_slope);
  LSPS_CHECK_LAUNCH("c8_pw1_dgrad_act");
  if (db_prev)
    if (int rc = c8_colsum(part, db_prev, C, rows, scratch, st)) return rc;
  if (dw) {                                                     // the head's own weight gradient [C] (+ bias gradient [1]) from the same pass
    hipLaunchKernelGGL(c8_pw1_wgrad_reduce_kernel, dim3(C + 1), dim3(256), 0, st, (const float *)wpart, dw, db, C, rows);
    LSPS_CHECK_LAUNCH("c8_pw1_wgrad_reduce");
  }
  return 0;
}

}  // extern "C"
