// C-ABI of the three-limb ("X3") stride-2 conv prototype (x3s2.h): f32-class arithmetic on the bf16 matrix pipe.
#include <algorithm>
#include <numeric>
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


static bool x3s2_tr_geom(int N, int Cx, int H, int W, int M, X3S2TParams *p) {      // H x W = the BIG (output) map
  if (N <= 0 || H < 2 || W < 2 || !x3_pow2(H) || !x3_pow2(W) || (Cx & 15) || (M & 63)) return false;
  const int P = H / 2, Q = W / 2, PQ = P * Q;
  int TI, TR, tpi, nt;
  if (PQ >= 256) {
    if (Q > 256) return false;
    TI = 1; TR = 256 / Q; tpi = P / TR; nt = N * tpi;
  } else {
    TI = 256 / PQ; TR = P; tpi = 1; nt = (N + TI - 1) / TI;
  }
  if (2l * TI * TR * (Q + 1) > X3T_BP * 64 || (long)TI * 3 * (Cx >> 3) * P * Q * 16 >= (1l << 31)) return false;
  if ((long)(Cx >> 4) * 3 * X3T_ASTAGE >= (1l << 31)) return false;
  if (p) {
    p->H = H; p->W = W; p->P = P; p->Q = Q;
    p->TI = TI; p->TR = TR; p->tiles_per_img = tpi; p->ntiles = nt;
  }
  return true;
}

static bool x3s2_wgrad_geom(int N, int K, int C, int H, int W, X3S2WParams *p) {   // small [N][K][H/2][W/2], big [N][C][H][W]
  if (N <= 0 || H < 2 || W < 2 || !x3_pow2(H) || !x3_pow2(W) || (K & 127) || (C & 63)) return false;
  const int P = H / 2, Q = W / 2, PQ = P * Q;
  int TIW, TRW, QW, cb, cpi, nc;
  if (PQ >= 16) {
    TIW = 1;
    QW = Q >= 16 ? 16 : Q;
    TRW = 16 / QW;
    cb = Q / QW;
    cpi = (P / TRW) * cb;
    nc = N * cpi;
  } else {
    TIW = 16 / PQ; TRW = P; QW = Q; cb = 1; cpi = 1;
    nc = (N + TIW - 1) / TIW;
  }
  const int blk = (2 * TRW + 1) * (2 * QW + 1);
  int bplane = TIW * blk;
  while ((bplane & 7) != 4) ++bplane;                            // plane stride = 64 (mod 128) bytes: bank quarter rotation
  if (8 * bplane > X3W_BP * 64) return false;
  if ((long)TIW * 3 * (C >> 3) * H * W * 16 >= (1l << 31) || (long)TIW * 3 * (K >> 3) * P * Q * 16 >= (1l << 31)) return false;
  const int tiles = (K >> 7) * (C >> 6);
  int s = (256 + tiles - 1) / tiles;
  if (s > 1) s = (s + 7) / 8 * 8;                                // a split lives on ONE XCD (workgroup -> tile mapping): use all 8
  if (s > nc) s = nc;
  const int cps = (nc + s - 1) / s;
  s = (nc + cps - 1) / cps;
  if (p) {
    p->N = N; p->K = K; p->C = C; p->H = H; p->W = W; p->P = P; p->Q = Q;
    p->TIW = TIW; p->TRW = TRW; p->QW = QW; p->colblocks = cb; p->chunks_per_img = cpi; p->nchunks = nc;
    p->splits = s; p->chunks_per_split = cps; p->bplane = bplane;
  }
  return true;
}

// packed three-limb weight panel (forward-direction layout: BM = 128, transposed-direction layout: BM = 64) in the pack-cache
// scope or at the start of `ws`
static int x3s2_pack(const float *w, int M, int C, long sm, long sc, int BM, void *ws, size_t ws_bytes, hipStream_t st,
                     const unsigned short **out) {
  const size_t need = (size_t)M * C * 9 * 3 * sizeof(unsigned short);
  bool hit = false;
  void *slot = pack_cache_slot(w, /*tag: X3 stride-2 layout*/ (1 << 24) + BM, M, C, sm, sc, need, &hit, st);
  if (!slot) {
    if (!ws || ws_bytes < need) {
      set_error("x3 stride-2 conv: workspace too small (%zu < %zu)", ws_bytes, need);
      return LSPS_E_ARG;
    }
    slot = ws;
  }
  *out = (const unsigned short *)slot;
  if (hit) return 0;
  X3S2Pack pp;
  pp.W = w; pp.Wq = (unsigned short *)slot; pp.M = M; pp.C = C; pp.sm = sm; pp.sc = sc;
  if (BM == 128)
    hipLaunchKernelGGL(x3s2_pack_kernel<128>, dim3(ceil_div((long)M * C / 8, 256)), dim3(256), 0, st, pp);
  else
    hipLaunchKernelGGL(x3s2_pack_kernel<64>, dim3(ceil_div((long)M * C / 8, 256)), dim3(256), 0, st, pp);
  LSPS_CHECK_LAUNCH("x3s2_pack");
  return 0;
}

// Launch plan of the forward / transposed kernels: how many k ranges (split-K) and which workgroup -> unit walk.
// `ntiles` pixel tiles x `MT` m tiles, `nk` reduction chunks of 16 channels, `out_mb` = 10^6 bytes of f32 output.
//
// Both kernels are persistent (one workgroup per CU) and a unit's time is ~ its chunks, so a launch costs
//   rounds(units over workgroups) x (chunks per unit + fixed part) [+ the partial sums' round trip and x3_splitk_finish_kernel],
// in units of one chunk (~5 us): e.g. 288 units = 2 rounds of which the second is 1/8 full, and with the XCD walk (a pixel tile
// lives on XCD tile % 8) three pixel tiles leave five XCDs without any work.  The plan is the cheapest of {XCD walk, linear walk}
// x {1 .. 8 ranges} under that model; measured on the estimate-mode discriminator (N = 144: 5.76 -> 5.3 ms per estimate3 step,
// `profiles/r5z_*`) — the constants only have to rank the candidates.  LSPS_X3_PLAN=0: the round-5 rule (XCD walk; split only
// below 3/4 of a round and >= 512 channels).
struct X3Plan {
  int ks, kper, linear, grid;
};

static int x3_plan_mode() { return opts().x3_plan; }

static int x3_device_cus();

static X3Plan x3_plan(int ntiles, int MT, int nk, double out_mb, bool tr) {
  const int cus = x3_device_cus() / 8 * 8;
  X3Plan best = {1, nk, 0, 0};
  if (x3_plan_mode() == 0) {
    const int base = ntiles * MT;
    int ks = 1;
    if (base * 4 <= 256 * 3 && nk >= 32) {
      ks = std::min(8, (256 + base - 1) / base);
      ks = std::min(ks, nk / 8);
    }
    if (ks < 1) ks = 1;
    best.kper = (nk + ks - 1) / ks;
    best.ks = (nk + best.kper - 1) / best.kper;
    best.grid = std::min((ntiles + 7) / 8 * 8 * MT * best.ks, cus);
    return best;
  }
  const double fixed = 1.5;                                  // prologue + epilogue of a unit, in chunks
  double best_cost = 0;
  for (int linear = 0; linear < 2; ++linear)
    for (int req = 1; req <= 8; ++req) {
      const int kper = (nk + req - 1) / req, ks = (nk + kper - 1) / kper;
      if (ks != req || (ks > 1 && (kper < 4 || ks * out_mb > 768.))) continue;      // (the partial sums live in the workspace)
      long units, groups, rounds;
      if (linear) {
        units = (long)ntiles * MT * ks;
        groups = std::min<long>(units, cus);
        rounds = (units + groups - 1) / groups;
      } else {
        units = (long)((ntiles + 7) / 8) * MT * ks;          // of the busiest XCD
        groups = std::min<long>((long)(ntiles + 7) / 8 * 8 * MT * ks, cus) / 8;
        rounds = (units + groups - 1) / groups;
      }
      // operand bytes a unit pulls through its XCD's L2 per chunk (KB), divided by how many of the XCD's concurrent workgroups want
      // the same bytes; beyond ~70 KB per chunk and CU (4.5 TB/s over 256 CUs at 4 us per chunk) the chunk takes longer.  Weights are
      // 2.25 x the activations: 384 units of `model_S.3` with nothing shared would stream 2.7 GB of limbs per launch.
      const double wkb = tr ? 55.3 : 110.6, akb = tr ? 24.6 : 49.2;
      const long gpx = std::max<long>(1, linear ? groups / 8 : groups);      // workgroups per XCD
      const long panels = (long)MT * ks;
      double sw, sa;
      if (linear) {
        const long distinct = panels / std::gcd<long, long>(panels, 8);         // panels an XCD ever sees at one time
        sw = std::min<double>(ntiles, std::max<double>(1., (double)gpx / distinct));
        sa = std::max(1., MT / 8.);
      } else {
        sw = std::min<double>((ntiles + 7) / 8, std::max<double>(1., (double)gpx / panels));
        sa = std::min<double>(MT, gpx);
      }
      const double slow = std::max(1., (wkb / sw + akb / sa) / 70.);
      double cost = rounds * (kper * slow + fixed);
      if (ks > 1) cost += (8.0 + (2.0 * ks + 1.5) * out_mb / 2.5) / 5.2;      // partial sums: written, read, + the finish launch
      if (linear || ks > 1) cost *= 1.1;                     // the model ranks, it does not measure: leave the plain launch (one range, an
                                                             // image's rows in one L2) only for a predicted gain of 10 %
      if (best.grid == 0 || cost < best_cost) {
        best_cost = cost;
        best.ks = ks; best.kper = kper; best.linear = linear;
        best.grid = linear ? (int)groups : (int)(groups * 8);
      }
    }
  return best;
}

// x3_splitk_finish_kernel's cut of the N * HW (image, pixel) items: ~2048 workgroups over (C / 8) x ranges, ranges of whole
// 256-item passes, at most X3_FINISH_MAX_RANGES of them (= rows of its bias-gradient partial sums)
#define X3_FINISH_MAX_RANGES 1024
static int x3_finish_ranges(int N, int M, long HW, long *items_per_range) {
  const long total = (long)N * HW;
  long want = std::max<long>(1, (2048 + (M >> 3) - 1) / (M >> 3));
  want = std::min<long>(want, X3_FINISH_MAX_RANGES);
  long ipr = (total + want - 1) / want;
  ipr = (ipr + 255) / 256 * 256;
  *items_per_range = ipr;
  return (int)((total + ipr - 1) / ipr);
}

// out[c] = sum over `rows` rows of part[rows][C]; `scratch`: 64 * C floats (used when rows > 64)
static int x3_colsum(const float *part, float *out, int C, int rows, float *scratch, hipStream_t st) {
  if (rows > 64) {
    const int rpc = (rows + 63) / 64, chunks = (rows + rpc - 1) / rpc;
    hipLaunchKernelGGL(x3_colsum_kernel, dim3(ceil_div(C, 64), chunks), dim3(256), 0, st, part, scratch, C, rows, rpc);
    LSPS_CHECK_LAUNCH("x3_colsum_stage1");
    part = scratch;
    rows = chunks;
  }
  hipLaunchKernelGGL(x3_colsum_kernel, dim3(ceil_div(C, 64), 1), dim3(256), 0, st, part, out, C, rows, rows);
  LSPS_CHECK_LAUNCH("x3_colsum");
  return 0;
}

// bias-gradient partial sums of a MASKED launch: [ntiles + 64][M] floats at the END of the workspace (the last 64 rows: scratch of x3_colsum)
static float *x3s2_dbpart(void *ws, size_t ws_bytes, int ntiles, int M, size_t pack_bytes) {
  const size_t need = ((size_t)ntiles + 64) * M * sizeof(float);
  if (!ws || ws_bytes < align_up(pack_bytes, 256) + need + 256) return nullptr;
  return reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + ((ws_bytes - need) & ~(size_t)255));
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


// small[n][m] = act(bias + sum Wt[m][kk][r][s] big[n][kk][2p+r-1][2q+s-1]);  W element (m, kk, t) at m*sm + kk*sc + t
static int x3s2_run_fwd(const void *big, const float *w, long sm, long sc, const float *bias, float *y, void *yl, int N, int Cx, int H, int W,
                        int M, float slope, void *ws, size_t ws_bytes, hipStream_t st, const void *act_y = nullptr, float act_slope = 0.f,
                        float *db_prev = nullptr) {
  X3S2Params p;
  if (!x3s2_fwd_geom(N, Cx, H, W, M, &p)) {
    set_error("x3 stride-2 conv (forward direction): unsupported geometry N=%d C=%d %dx%d M=%d", N, Cx, H, W, M);
    return LSPS_E_ARG;
  }
  if (bias && (reinterpret_cast<uintptr_t>(bias) & 15)) {
    set_error("x3 stride-2 conv: the bias must be 16-byte aligned");
    return LSPS_E_ARG;
  }
  const unsigned short *wq = nullptr;
  if (int rc = x3s2_pack(w, M, Cx, sm, sc, 128, ws, ws_bytes, st, &wq)) return rc;
  p.X = (const unsigned short *)big;
  p.Wq = wq;
  p.bias = bias;
  p.Y = y;
  p.YL = (unsigned short *)yl;
  p.N = N; p.Cx = Cx; p.M = M;
  p.lrelu = slope >= 0.f ? slope : 1.f;
  p.ActY = (const unsigned short *)act_y;
  p.act_slope = act_slope;
  p.dbpart = nullptr;
  const size_t pack_bytes = align_up((size_t)M * Cx * 9 * 3 * sizeof(unsigned short), 256);
  const X3Plan plan = x3_plan(p.ntiles, M >> 7, Cx >> 4, 4e-6 * N * M * p.P * p.Q, false);
  p.ksplit = plan.ks; p.kper = plan.kper; p.linear = plan.linear;
  p.ysplit = 0;
  if (p.ksplit > 1) {
    // raw partial outputs [range][N][M][P][Q] behind the packed weights; the epilogue runs in x3_splitk_finish_kernel
    const long out_elems = (long)N * M * p.P * p.Q;
    const size_t part_bytes = (size_t)p.ksplit * out_elems * sizeof(float);
    long ips;
    const int nis = x3_finish_ranges(N, M, (long)p.P * p.Q, &ips);
    const size_t db_bytes = act_y ? ((size_t)nis + 64) * M * sizeof(float) : 0;
    if (!ws || ws_bytes < pack_bytes + part_bytes + db_bytes + 256) {
      set_error("x3 stride-2 conv: workspace too small for the split-K partial sums");
      return LSPS_E_ARG;
    }
    float *part = reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + pack_bytes);
    float *dbp = act_y ? reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + pack_bytes + align_up(part_bytes, 256)) : nullptr;
    X3S2Params q = p;
    q.bias = nullptr; q.lrelu = 1.f; q.ActY = nullptr; q.Y = part; q.YL = nullptr; q.ysplit = out_elems;
    const dim3 sgrid(plan.grid);
    if (int rc = lds_optin(reinterpret_cast<const void *>(x3s2_fwd_kernel<false, false>), X3F_LDS_BYTES, "x3s2_fwd")) return rc;
    hipLaunchKernelGGL((x3s2_fwd_kernel<false, false>), sgrid, dim3(512), X3F_LDS_BYTES, st, q);
    LSPS_CHECK_LAUNCH("x3s2_fwd(split-K)");
    hipLaunchKernelGGL(x3_splitk_finish_kernel, dim3(M >> 3, nis), dim3(256), 0, st, (const float *)part, p.ksplit, out_elems, bias, p.lrelu,
                       (const unsigned short *)act_y, act_slope, y, (unsigned short *)yl, dbp, N, M, p.P * p.Q, ips);
    LSPS_CHECK_LAUNCH("x3_splitk_finish");
    if (act_y && db_prev) return x3_colsum(dbp, db_prev, M, nis, dbp + (size_t)nis * M, st);
    return 0;
  }
  if (act_y) {
    p.dbpart = x3s2_dbpart(ws, ws_bytes, p.ntiles, M, (size_t)M * Cx * 9 * 3 * sizeof(unsigned short));
    if (!p.dbpart) {
      set_error("x3 stride-2 conv: workspace too small for the bias-gradient partial sums");
      return LSPS_E_ARG;
    }
  }
  const dim3 grid(plan.grid);
  // round 6: 3-deep image ring + counted waits (x3s2.h: RING); its 9 image pieces per limb hold every geometry but the 2x2 maps
  const bool ring = opts().x3_ring && 2 * p.TI * p.TR * (2 * p.Q + 1) <= X3R_BP * 64;
#define X3F_LAUNCH(O3, MK)                                                                                                  \
  do {                                                                                                                      \
    if (ring) {                                                                                                             \
      if (int rc = lds_optin(reinterpret_cast<const void *>(x3s2_fwd_kernel<O3, MK, true>), X3R_LDS_BYTES, "x3s2_fwd_ring")) return rc; \
      hipLaunchKernelGGL((x3s2_fwd_kernel<O3, MK, true>), grid, dim3(512), X3R_LDS_BYTES, st, p);                          \
    } else {                                                                                                                \
      if (int rc = lds_optin(reinterpret_cast<const void *>(x3s2_fwd_kernel<O3, MK>), X3F_LDS_BYTES, "x3s2_fwd")) return rc; \
      hipLaunchKernelGGL((x3s2_fwd_kernel<O3, MK>), grid, dim3(512), X3F_LDS_BYTES, st, p);                                 \
    }                                                                                                                       \
  } while (0)
  if (yl) {
    if (act_y) X3F_LAUNCH(true, true); else X3F_LAUNCH(true, false);
  } else {
    if (act_y) X3F_LAUNCH(false, true); else X3F_LAUNCH(false, false);
  }
#undef X3F_LAUNCH
  LSPS_CHECK_LAUNCH("x3s2_fwd");
  if (act_y && db_prev) return x3_colsum(p.dbpart, db_prev, M, p.ntiles, p.dbpart + (size_t)p.ntiles * M, st);
  return 0;
}

// big[n][m][2p+r-1][2q+s-1] += Wt[m][kk][r][s] small[n][kk][p][q] (+ bias, activation); H x W = the big map
static int x3s2_run_tr(const void *small, const float *w, long sm, long sc, const float *bias, float *y, void *yl, int N, int Cx, int H,
                       int W, int M, float slope, void *ws, size_t ws_bytes, hipStream_t st, const void *act_y = nullptr,
                       float act_slope = 0.f, float *db_prev = nullptr) {
  X3S2TParams p;
  if (!x3s2_tr_geom(N, Cx, H, W, M, &p)) {
    set_error("x3 stride-2 conv (transposed direction): unsupported geometry N=%d C=%d -> %dx%d M=%d", N, Cx, H, W, M);
    return LSPS_E_ARG;
  }
  if (bias && (reinterpret_cast<uintptr_t>(bias) & 15)) {
    set_error("x3 stride-2 conv: the bias must be 16-byte aligned");
    return LSPS_E_ARG;
  }
  const unsigned short *wq = nullptr;
  if (int rc = x3s2_pack(w, M, Cx, sm, sc, 64, ws, ws_bytes, st, &wq)) return rc;
  p.X = (const unsigned short *)small;
  p.Wq = wq;
  p.bias = bias;
  p.Y = y;
  p.YL = (unsigned short *)yl;
  p.N = N; p.Cx = Cx; p.M = M;
  p.lrelu = slope >= 0.f ? slope : 1.f;
  p.ActY = (const unsigned short *)act_y;
  p.act_slope = act_slope;
  p.dbpart = nullptr;
  const size_t pack_bytes = align_up((size_t)M * Cx * 9 * 3 * sizeof(unsigned short), 256);
  const X3Plan plan = x3_plan(p.ntiles, M >> 6, Cx >> 4, 4e-6 * N * M * H * W, true);
  p.ksplit = plan.ks; p.kper = plan.kper; p.linear = plan.linear;
  p.ysplit = 0;
  if (p.ksplit > 1) {
    const long out_elems = (long)N * M * H * W;
    const size_t part_bytes = (size_t)p.ksplit * out_elems * sizeof(float);
    long ips;
    const int nis = x3_finish_ranges(N, M, (long)H * W, &ips);
    const size_t db_bytes = act_y ? ((size_t)nis + 64) * M * sizeof(float) : 0;
    if (!ws || ws_bytes < pack_bytes + part_bytes + db_bytes + 256) {
      set_error("x3 stride-2 conv: workspace too small for the split-K partial sums");
      return LSPS_E_ARG;
    }
    float *part = reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + pack_bytes);
    float *dbp = act_y ? reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + pack_bytes + align_up(part_bytes, 256)) : nullptr;
    X3S2TParams q = p;
    q.bias = nullptr; q.lrelu = 1.f; q.ActY = nullptr; q.Y = part; q.YL = nullptr; q.ysplit = out_elems;
    const dim3 sgrid(plan.grid);
    if (int rc = lds_optin(reinterpret_cast<const void *>(x3s2_tr_kernel<false, false>), X3T_LDS_BYTES, "x3s2_tr")) return rc;
    hipLaunchKernelGGL((x3s2_tr_kernel<false, false>), sgrid, dim3(512), X3T_LDS_BYTES, st, q);
    LSPS_CHECK_LAUNCH("x3s2_tr(split-K)");
    hipLaunchKernelGGL(x3_splitk_finish_kernel, dim3(M >> 3, nis), dim3(256), 0, st, (const float *)part, p.ksplit, out_elems, bias, p.lrelu,
                       (const unsigned short *)act_y, act_slope, y, (unsigned short *)yl, dbp, N, M, H * W, ips);
    LSPS_CHECK_LAUNCH("x3_splitk_finish");
    if (act_y && db_prev) return x3_colsum(dbp, db_prev, M, nis, dbp + (size_t)nis * M, st);
    return 0;
  }
  if (act_y) {
    p.dbpart = x3s2_dbpart(ws, ws_bytes, p.ntiles, M, (size_t)M * Cx * 9 * 3 * sizeof(unsigned short));
    if (!p.dbpart) {
      set_error("x3 stride-2 conv: workspace too small for the bias-gradient partial sums");
      return LSPS_E_ARG;
    }
  }
  const dim3 grid(plan.grid);
#define X3T_LAUNCH(O3, MK)                                                                                                \
  do {                                                                                                                    \
    if (int rc = lds_optin(reinterpret_cast<const void *>(x3s2_tr_kernel<O3, MK>), X3T_LDS_BYTES, "x3s2_tr")) return rc;  \
    hipLaunchKernelGGL((x3s2_tr_kernel<O3, MK>), grid, dim3(512), X3T_LDS_BYTES, st, p);                                  \
  } while (0)
  if (yl) {
    if (act_y) X3T_LAUNCH(true, true); else X3T_LAUNCH(true, false);
  } else {
    if (act_y) X3T_LAUNCH(false, true); else X3T_LAUNCH(false, false);
  }
#undef X3T_LAUNCH
  LSPS_CHECK_LAUNCH("x3s2_tr");
  if (act_y && db_prev) return x3_colsum(p.dbpart, db_prev, M, p.ntiles, p.dbpart + (size_t)p.ntiles * M, st);
  return 0;
}

static int x3s2_run_wgrad(const void *small, const void *big, float *dw, int N, int K, int C, int H, int W, void *ws, size_t ws_bytes,
                          hipStream_t st) {
  X3S2WParams p;
  if (!x3s2_wgrad_geom(N, K, C, H, W, &p)) {
    set_error("x3 stride-2 weight gradient: unsupported geometry N=%d K=%d C=%d %dx%d", N, K, C, H, W);
    return LSPS_E_ARG;
  }
  const size_t need = (size_t)p.splits * 9 * K * C * sizeof(float);
  if (!ws || ws_bytes < need) {
    set_error("x3 stride-2 weight gradient: workspace too small (%zu < %zu)", ws_bytes, need);
    return LSPS_E_ARG;
  }
  p.S = (const unsigned short *)small;
  p.B = (const unsigned short *)big;
  p.part = (float *)ws;
  if (int rc = lds_optin(reinterpret_cast<const void *>(x3s2_wgrad_kernel), X3W_LDS_BYTES, "x3s2_wgrad")) return rc;
  const int tiles = (K >> 7) * (C >> 6);
  hipLaunchKernelGGL(x3s2_wgrad_kernel, dim3(p.splits == 1 ? tiles : (p.splits + 7) / 8 * 8 * tiles), dim3(512), X3W_LDS_BYTES, st, p);
  LSPS_CHECK_LAUNCH("x3s2_wgrad");
  hipLaunchKernelGGL(x3_wgrad_reduce_kernel, dim3(ceil_div((long)K * C, 64)), dim3(256), 0, st, (const float *)p.part, dw, K * C, p.splits);
  LSPS_CHECK_LAUNCH("x3s2_wgrad_reduce");
  return 0;
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

int lsps_x3_conv3x3s2_ok(int N, int C, int H, int W, int K) {
  return x3s2_fwd_geom(N, C, H, W, K, nullptr) && x3s2_tr_geom(N, K, H, W, C, nullptr) && x3s2_wgrad_geom(N, K, C, H, W, nullptr) ? 1 : 0;
}

size_t lsps_x3_conv3x3s2_workspace_bytes(int N, int C, int H, int W, int K) {
  if (!lsps_x3_conv3x3s2_ok(N, C, H, W, K)) return 0;
  const size_t pack = align_up((size_t)K * C * 9 * 3 * sizeof(unsigned short), 256);
  X3S2TParams q;
  x3s2_tr_geom(N, K, H, W, C, &q);
  size_t need = pack + align_up(((size_t)q.ntiles + 64) * C * sizeof(float), 256) + 512;     // packed weights + dgrad's bias-gradient partials
  X3S2Params f;
  x3s2_fwd_geom(N, C, H, W, K, &f);                                                   // (as a transposed conv's dgrad: tiles x K)
  need = std::max(need, pack + align_up(((size_t)f.ntiles + 64) * K * sizeof(float), 256) + 512);
  // split-K partial outputs (<= 8 ranges) + the finish kernel's bias-gradient partial sums (<= X3_FINISH_MAX_RANGES rows)
  const int ksf = x3_plan(f.ntiles, K >> 7, C >> 4, 4e-6 * N * K * f.P * f.Q, false).ks;      // forward direction: out [N][K][P][Q]
  if (ksf > 1) need = std::max(need, pack + align_up((size_t)ksf * N * K * f.P * f.Q * sizeof(float), 256) + ((size_t)X3_FINISH_MAX_RANGES + 64) * K * sizeof(float) + 512);
  const int kst = x3_plan(q.ntiles, C >> 6, K >> 4, 4e-6 * N * C * H * W, true).ks;           // transposed direction: out [N][C][H][W]
  if (kst > 1) need = std::max(need, pack + align_up((size_t)kst * N * C * H * W * sizeof(float), 256) + ((size_t)X3_FINISH_MAX_RANGES + 64) * C * sizeof(float) + 512);
  X3S2WParams wp;
  x3s2_wgrad_geom(N, K, C, H, W, &wp);
  need = std::max(need, (size_t)wp.splits * 9 * K * C * sizeof(float));
  return need;
}

int lsps_x3_conv3x3s2_plan(int transposed, int N, int C, int H, int W, int K, int plan[4]) {
  LSPS_CHECK_ARG(plan, "x3 conv plan: null pointer");
  X3Plan pl;
  if (transposed) {
    X3S2TParams q;
    LSPS_CHECK_ARG(x3s2_tr_geom(N, K, H, W, C, &q), "x3 conv plan: unsupported geometry");
    pl = x3_plan(q.ntiles, C >> 6, K >> 4, 4e-6 * N * C * H * W, true);
  } else {
    X3S2Params f;
    LSPS_CHECK_ARG(x3s2_fwd_geom(N, C, H, W, K, &f), "x3 conv plan: unsupported geometry");
    pl = x3_plan(f.ntiles, K >> 7, C >> 4, 4e-6 * N * K * f.P * f.Q, false);
  }
  plan[0] = pl.ks; plan[1] = pl.kper; plan[2] = pl.linear; plan[3] = pl.grid;
  return 0;
}

int lsps_x3_conv3x3s2_fwd(const void *xl, const float *w, const float *bias, float *y, void *yl, int N, int C, int H, int W, int K,
                          float slope, void *ws, size_t ws_bytes, void *stream) {
  LSPS_CHECK_ARG(xl && w && (y || yl), "x3 conv: null pointer");
  return x3s2_run_fwd(xl, w, (long)C * 9, 9, bias, y, yl, N, C, H, W, K, slope, ws, ws_bytes, (hipStream_t)stream);
}

int lsps_x3_conv3x3s2_dgrad(const void *dyl, const float *w, float *dx, void *dxl, const void *act_yl, float act_slope, float *db_prev,
                            int N, int C, int H, int W, int K, void *ws, size_t ws_bytes, void *stream) {
  LSPS_CHECK_ARG(dyl && w && (dx || dxl), "x3 conv dgrad: null pointer");
  return x3s2_run_tr(dyl, w, 9, (long)C * 9, nullptr, dx, dxl, N, K, H, W, C, -1.f, ws, ws_bytes, (hipStream_t)stream, act_yl, act_slope,
                     db_prev);
}

int lsps_x3_conv3x3s2_wgrad(const void *xl, const void *dyl, float *dw, int N, int C, int H, int W, int K, void *ws, size_t ws_bytes,
                            void *stream) {
  LSPS_CHECK_ARG(xl && dyl && dw, "x3 conv wgrad: null pointer");
  return x3s2_run_wgrad(dyl, xl, dw, N, K, C, H, W, ws, ws_bytes, (hipStream_t)stream);
}

int lsps_x3_convT3x3s2_ok(int N, int Ci, int H, int W, int Co) {
  return x3s2_tr_geom(N, Ci, 2 * H, 2 * W, Co, nullptr) && x3s2_fwd_geom(N, Co, 2 * H, 2 * W, Ci, nullptr) &&
                 x3s2_wgrad_geom(N, Ci, Co, 2 * H, 2 * W, nullptr)
             ? 1 : 0;
}

/* ConvTranspose2d(Ci, Co, 3, 2, 1, 1): x [N,Ci,H,W] -> y [N,Co,2H,2W]; w (Ci,Co,3,3).  Workspace: lsps_x3_conv3x3s2_workspace_bytes
 * (N, Co, 2H, 2W, Ci). */
int lsps_x3_convT3x3s2_fwd(const void *xl, const float *w, const float *bias, float *y, void *yl, int N, int Ci, int H, int W, int Co,
                           float slope, void *ws, size_t ws_bytes, void *stream) {
  LSPS_CHECK_ARG(xl && w && (y || yl), "x3 convT: null pointer");
  return x3s2_run_tr(xl, w, 9, (long)Co * 9, bias, y, yl, N, Ci, 2 * H, 2 * W, Co, slope, ws, ws_bytes, (hipStream_t)stream);
}

int lsps_x3_convT3x3s2_dgrad(const void *dyl, const float *w, float *dx, void *dxl, const void *act_yl, float act_slope, float *db_prev,
                             int N, int Ci, int H, int W, int Co, void *ws, size_t ws_bytes, void *stream) {
  LSPS_CHECK_ARG(dyl && w && (dx || dxl), "x3 convT dgrad: null pointer");
  return x3s2_run_fwd(dyl, w, (long)Co * 9, 9, nullptr, dx, dxl, N, Co, 2 * H, 2 * W, Ci, -1.f, ws, ws_bytes, (hipStream_t)stream, act_yl,
                      act_slope, db_prev);
}

int lsps_x3_convT3x3s2_wgrad(const void *xl, const void *dyl, float *dw, int N, int Ci, int H, int W, int Co, void *ws, size_t ws_bytes,
                             void *stream) {
  LSPS_CHECK_ARG(xl && dyl && dw, "x3 convT wgrad: null pointer");
  return x3s2_run_wgrad(xl, dyl, dw, N, Ci, Co, 2 * H, 2 * W, ws, ws_bytes, (hipStream_t)stream);
}

size_t lsps_x3_act_bwd_bias_workspace_bytes(int N, int C) {
  if (N <= 0 || C <= 0) return 0;
  return ((size_t)256 + 64 + 8) * C * sizeof(float);
}

/* g (X3) = dy * LeakyReLU'(y) from f32 NCHW dy and the layer's f32 NCHW OUTPUT y (slope < 0: g = dy, y may be NULL);
 * db [C] (nullable) = sum of g over n and pixels. */
int lsps_x3_act_bwd_bias(const float *dy, const float *y, void *gl, float *db, int N, int C, int HW, float slope, void *ws,
                         size_t ws_bytes, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  LSPS_CHECK_ARG(dy && gl && (y || slope < 0.f), "x3 act_bwd_bias: null pointer");
  LSPS_CHECK_ARG(N > 0 && C > 0 && (C & 7) == 0 && HW > 0, "x3 act_bwd_bias: C must be a multiple of 8");
  // enough workgroups to fill the chip: (C / 8) x splits >= ~1024
  int splits = std::max(1, std::min(N, (int)((1024 + (C >> 3) - 1) / (C >> 3))));
  if (splits > 256) splits = 256;
  const int ips = (N + splits - 1) / splits;
  splits = (N + ips - 1) / ips;
  float *part = nullptr;
  if (db) {
    LSPS_CHECK_ARG(ws && ws_bytes >= ((size_t)splits + 64) * C * sizeof(float), "x3 act_bwd_bias: workspace too small");
    part = (float *)ws;
  }
  hipLaunchKernelGGL(x3_act_bwd_bias_nchw_kernel, dim3(C >> 3, splits), dim3(256), 0, st, dy, y, (unsigned short *)gl, part, N, C, HW, ips,
                     slope);
  LSPS_CHECK_LAUNCH("x3_act_bwd_bias");
  if (db) return x3_colsum(part, db, C, splits, part + (size_t)splits * C, st);
  return 0;
}

}  // extern "C"
