// Winograd F(2x2, 3x3) form of the 3x3 / stride 1 / pad 1 / width 32 kernel (forward and dgrad of the residual convs).
#ifndef LSPS_CONV_WINO_H
#define LSPS_CONV_WINO_H
#include "conv_types.h"
#include "conv3x3.h"

namespace lsps {

// -------------------------------------------------------------------------------------------
// The direct kernel (igemm_f3x3_kernel) runs at what the f32 matrix pipe sustains, so the only way down is fewer
// multiplies: F(2x2,3x3) computes a 2x2 output tile from a 4x4 input tile with 16 multiplies per (k, c) pair instead
// of 36, i.e. 16 independent [K x C] x [C x tiles] GEMMs, one per position of the transformed 4x4 tile.
//
// Everything is fused in one kernel, nothing transformed ever goes to HBM:
//   U = G g G^T        once per weight tensor by wino_pack_kernel (cached with the other packed panels)
//   V = B^T d B        per lane, in registers, from the raw input rows staged in LDS (the same "raw rows + zero halo"
//                      staging as the direct kernel), on packed f32 VALU ops
//   M_p += U_p V_p     v_mfma_f32_32x32x2_f32, one accumulator tile (32 k x 32 tiles) per position p; the A operands
//                      (U) go global / L2 -> registers, fetched 4 k-steps ahead, and never pass through LDS
//   Y = A^T M A        in registers in the epilogue
//
// Workgroup: 64 output channels x 64 tiles (8 output rows x 32 columns), 8 waves = 2 (k) x 2 (tile rows) x 2 (halves of
// the 16 positions: rows 2wp, 2wp+1 of the 4x4 grid).  All 16 positions in one wave would need 256 accumulator
// registers = one wave per SIMD, and with a single in-order wave every non-MFMA instruction costs its issue time
// (first version: 187 TFLOP/s); 8 positions = 128 accumulators leave two waves per SIMD.  The two halves' partial 2x2
// output tiles are added through LDS in the epilogue.  A wave's 32 tiles are 2 tile rows x 16 tile columns.
// MFMA operand layout (32x32x2): lane l holds A[k = l%32][c = l/32] and B[c = l/32][tile = l%32], so lane l transforms
// ONE tile of ONE channel per k-step.  LDS per k-step and wave: six ds_read_b64 (three raw rows) for 8 MFMAs.
// Row stride 48 floats: two tile rows apart = 96 floats = 32 banks, so the 32 lanes of a ds_read_b64 group (2 tile rows x
// 16 tile columns x 8 B) cover all 64 banks exactly once.
// -------------------------------------------------------------------------------------------
#define WN_CC 8                          // channels per U chunk (4 k-steps)
#define WN_RC 8                          // channels per staged row chunk (4 k-steps between two barriers; 16 measured 2 % slower)
#define WN_LDW 48                        // floats per staged row: [halo][32 pixels][halo][pad]
#define WN_ROWS 10                       // 8 output rows + 2 halo rows
#define WN_CH (WN_ROWS * WN_LDW)         // floats per staged channel
#define WN_UCH (WN_CC * 16 * 64)         // floats of U per (k-slice, chunk): [8 c][4 position quads][64 k][4]
#define WN_BUF (WN_RC * WN_CH)           // floats per LDS buffer (input rows only: U goes global -> registers)
#define WN_LDS_BYTES(NWK) (32 * 1024 * (NWK))   // 2 row buffers (30 KB) in the main loop; 16 KB per wave pair for the epilogue exchange

struct WinoPack {
  const float *W;
  float *U;                      // [M/64][C/8][8 c][4 quads][64 k][4]
  int M, C;
  long sm, sc;
  int tapidx[9];
};

// one thread per (m, c): U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ __launch_bounds__(256) void wino_pack_kernel(WinoPack p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)p.M * p.C) return;
  const int kl = (int)(idx & 63);
  long rest = idx >> 6;
  const int cl = (int)(rest & 7);
  rest >>= 3;
  const int chunks = p.C / WN_CC;
  const int chunk = (int)(rest % chunks), ks = (int)(rest / chunks);
  const int m = ks * 64 + kl, c = chunk * WN_CC + cl;
  float g[3][3], t[4][3], u[4][4];
  const float *w = p.W + (long)m * p.sm + (long)c * p.sc;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) g[r][s] = w[p.tapidx[r * 3 + s]];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    t[0][s] = g[0][s];
    t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
    t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
    t[3][s] = g[2][s];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    u[i][0] = t[i][0];
    u[i][1] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
    u[i][2] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
    u[i][3] = t[i][2];
  }
  float *dst = p.U + ((long)ks * chunks + chunk) * WN_UCH + (long)cl * (16 * 64) + kl * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    *reinterpret_cast<f32x4 *>(dst + i * 256) = f32x4{u[i][0], u[i][1], u[i][2], u[i][3]};
}

// Packed f32 VALU ops of the data transform, spelled out: hipcc scalarises the float2 form of the column stage
// (element shuffles become v_mov + scalar adds: 187 moves per chunk pair), and every VALU issue slot here is time the
// matrix pipe does not get.  op_sel / op_sel_hi pick the half of each 64-bit source for the low / high result.
// The two column ops feed MFMA operands: a VALU result needs 2 wait states before an MFMA may read it, and the
// hazard recognizer does not look inside inline asm, hence the s_nop 1 in their strings (without it: wrong results).
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {                 // (a.x - b.x, a.y - b.y)
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {        // a * b + c
  f32x2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ f32x2 pk_col01(f32x2 lo, f32x2 hi) {             // (lo.x - hi.x, lo.y + hi.x)
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]\n\ts_nop 1" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ f32x2 pk_col23(f32x2 lo, f32x2 hi) {             // (hi.x - lo.y, lo.y - hi.y)
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]\n\ts_nop 1" : "=v"(r) : "v"(hi), "v"(lo));
  return r;
}

struct WinoParams {
  const float *X, *U, *bias;
  const float *R;                // optional addend with Y's layout
  float *Y;
  int Cx, H, M, NT;              // NT = N * (H/8) pixel tiles
  int tiles_per_img;             // H / 8
  int act;
  float slope;
};

// NWK = waves along the output channels: 2 -> 512 threads, 64 channels x 64 tiles, one workgroup per CU;
// 1 -> 256 threads, 32 channels x 64 tiles, TWO workgroups per CU (same 2 waves per SIMD), so that one workgroup's
// barriers, prologue and epilogue overlap with the other one's MFMAs.
template <int NWK>
__global__ __launch_bounds__(256 * NWK, 2 / NWK) void wino_f3x3_kernel(WinoParams p) {
  constexpr int NT_ = 256 * NWK;
  extern __shared__ __attribute__((aligned(16))) float wn_lds[];
  constexpr int NB4 = WN_RC * WN_ROWS * 8;                      // 16-B segments of input rows per row chunk

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave & 1, wt = (wave >> 1) & 1, wk = wave >> 2;
  const int l31 = lane & 31, half = lane >> 5;
  const int tr = l31 >> 4, tc = l31 & 15;

  // workgroup -> (pixel tile, k slice).  Workgroups go to the 8 XCDs round-robin in launch order and each XCD has its
  // own 4 MB L2: an XCD keeps a fixed part of U resident for the whole launch and streams its share of the pixel tiles
  // (measured HBM fetch per N=256 launch: one slice per XCD 1.21 GB, two slices 0.81 GB, same speed).
  const int lin = blockIdx.x, MT = p.M / (32 * NWK);
  int tile, mt;
  if (MT == 4 && (p.NT & 3) == 0) {
    // two slices per XCD (2 MB of U in its L2), a quarter of the pixel tiles, the two slices of a tile adjacent in
    // time: the second read of the tile's input rows hits L2
    const int xcd = lin & 7, q = lin >> 3;
    mt = 2 * (xcd & 1) + (q & 1);
    tile = (xcd >> 1) * (p.NT >> 2) + (q >> 1);
  } else if (MT == 4 && (p.NT & 1) == 0) {
    const int xcd = lin & 7, q = lin >> 3;
    mt = xcd & 3;
    tile = (xcd >> 2) * (p.NT >> 1) + q;
  } else if (MT == 8 && (p.NT & 3) == 0) {       // 32-channel slices: four per XCD (2 MB of U), a quarter of the tiles
    const int xcd = lin & 7, q = lin >> 3;
    mt = 4 * (xcd & 1) + (q & 3);
    tile = (xcd >> 1) * (p.NT >> 2) + (q >> 2);
  } else if (MT == 8) {                          // one slice per XCD, every XCD streams all pixel tiles
    mt = lin & 7;
    tile = lin >> 3;
  } else {
    tile = lin % p.NT;
    mt = lin / p.NT;
  }
  const int n = tile / p.tiles_per_img;
  const int row0 = (tile - n * p.tiles_per_img) * 8;
  const int HW = p.H * 32;
  const float *xn = p.X + (long)n * p.Cx * HW;
  const int nchunks = p.Cx / WN_CC;
  // this lane's A operands: U[chunk][c = 2s + half][quad = 2 wp + q][k = wk*32 + l31][4]: uniform base + lane offset
  const int k0 = mt * 32 * NWK + wk * 32;        // first output channel of this wave
  const float *ubase = p.U + (long)(k0 >> 6) * nchunks * WN_UCH;
  const unsigned u_lane = (half * (16 * 64) + 2 * wp * 256 + ((k0 & 63) + l31) * 4) * 4;   // bytes (32-bit: scalar base + offset addressing)

  // staging of the input rows: NB4 16-B segments per row chunk, NSEG per thread (the last one only for the first waves)
  constexpr int NSEG = (NB4 + NT_ - 1) / NT_;
  int b_lds[NSEG];
  unsigned b_off[NSEG];
  bool b_ok[NSEG];
#pragma unroll
  for (int i = 0; i < NSEG; ++i) {
    const int u = min(tid + NT_ * i, NB4 - 1);
    const int line = u >> 3, c4 = u & 7;
    const int ch = line / WN_ROWS, r = line - ch * WN_ROWS;
    const int img_row = row0 - 1 + r;
    b_ok[i] = img_row >= 0 && img_row < p.H;
    b_lds[i] = ch * WN_CH + r * WN_LDW + 1 + c4 * 4;
    b_off[i] = (ch * HW + min(max(img_row, 0), p.H - 1) * 32 + c4 * 4) * 4;   // bytes; out-of-image rows: load a valid row, never store it
  }
  const bool last_seg = wave < (NB4 - NT_ * (NSEG - 1)) / 64;   // wave-uniform

  // positions (i, 0..3) for i = 2 wp, 2 wp + 1: 8 accumulator tiles of 32 k x 32 tiles
  f32x16 acc[8];

  // input rows travel global -> registers -> LDS and are fetched TWO row chunks ahead (two register sets): they come
  // from HBM, not from L2 like U, and one chunk (4 k-steps ~ 2 us) does not cover that latency under load
  f32x4 breg[2][NSEG];
  auto load_rows = [&](int rc, int set) {
    const char *xc = reinterpret_cast<const char *>(xn + (long)rc * WN_RC * HW);   // uniform
#pragma unroll
    for (int i = 0; i < NSEG; ++i)
      if (i < NSEG - 1 || last_seg) breg[set][i] = *reinterpret_cast<const f32x4 *>(xc + b_off[i]);
  };
  auto store_seg = [&](float *buf, int i, int set) {
    if (b_ok[i]) {
      float *d = buf + b_lds[i];
      d[0] = breg[set][i][0];
      *reinterpret_cast<f32x2 *>(d + 1) = f32x2{breg[set][i][1], breg[set][i][2]};
      d[3] = breg[set][i][3];
    }
  };
  auto store_rows = [&](float *buf, int set) {
#pragma unroll
    for (int i = 0; i < NSEG; ++i)
      if (i < NSEG - 1 || last_seg) store_seg(buf, i, set);
  };

  // The transform rows a wave needs, in the order (e0, e1, e2) that makes both position halves the same arithmetic:
  //   wp = 0 reads d rows (0, 2, 1): t0 = e0 - e1 = d0 - d2,  t1 = e1 + e2 = d2 + d1
  //   wp = 1 reads d rows (2, 1, 3): t2 = e0 - e1 = d2 - d1,  t3 = e1 - e2 = d1 - d3        (t_hi = e1 + sgn * e2)
  const float sgn = wp ? -1.f : 1.f;
  const f32x2 sgn2 = {sgn, sgn};
  const int er0 = wp ? 2 : 0, er1 = wp ? 1 : 2, er2 = wp ? 3 : 1;
  const int b_base = half * WN_CH + (wt * 4 + 2 * tr) * WN_LDW + 2 * tc;   // + 2s channels, + row
  const float *rd0 = wn_lds + b_base + er0 * WN_LDW, *rd1 = wn_lds + b_base + er1 * WN_LDW, *rd2 = wn_lds + b_base + er2 * WN_LDW;

  // A operands (U) come straight from global memory / L2 into registers, one k-step = 2 x 16 B per lane, fetched a
  // whole chunk (4 k-steps) ahead: a[s] is reloaded for the next chunk right after step s has consumed it.  U never
  // passes through LDS, which leaves LDS and the per-chunk barrier with the input rows only (5 KB instead of 47 KB).
  f32x4 a[4][2];
  auto load_u = [&](int ch, int s) {
    const float *us = ubase + (long)ch * WN_UCH + 2 * s * (16 * 64);   // uniform
    a[s][0] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(us) + u_lane);
    a[s][1] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(us + 256) + u_lane);
  };
  // B operands: raw rows of one k-step (channels 2s + half), double-buffered in registers.  `bo` = compile-time
  // float offset of the row buffer (the chunk loop is unrolled over both buffers so that every LDS address is a
  // loop-invariant register + immediate)
  f32x2 e[2][3][2];
  auto read_step = [&](int bo, int s, int slot) {              // s = channel pair 0..7 of the row chunk
    // volatile: keeps six ds_read_b64 with 16-bit immediate offsets; merged into ds_read2_b64 (8-bit offsets) every
    // read needs a VALU add for its base, and a read2_b64 costs 8 LDS cycles against 2 x 2
    typedef const volatile f32x2 __attribute__((address_space(3))) *vp;
    e[slot][0][0] = *(vp)(rd0 + bo + 2 * s * WN_CH);
    e[slot][0][1] = *(vp)(rd0 + bo + 2 * s * WN_CH + 2);
    e[slot][1][0] = *(vp)(rd1 + bo + 2 * s * WN_CH);
    e[slot][1][1] = *(vp)(rd1 + bo + 2 * s * WN_CH + 2);
    e[slot][2][0] = *(vp)(rd2 + bo + 2 * s * WN_CH);
    e[slot][2][1] = *(vp)(rd2 + bo + 2 * s * WN_CH + 2);
  };
  // one k-step: two rows of V = B^T d B and 8 MFMAs.  The transform is the only VALU work of the loop and VALU issue
  // is not hidden behind the matrix pipe (measured: 16 scalar ops per 8 MFMAs cost 20 %), so it is written on
  // float2 values for v_pk_add_f32 / v_pk_fma_f32: columns (v0, v1) = (t0, t1) + (-t2, t2), (v2, v3) = (t2, t1) - (t1, t3)
  auto mma_step = [&](int s, int slot) {
    f32x2 t[2][2], v[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      t[0][h] = pk_sub(e[slot][0][h], e[slot][1][h]);
      t[1][h] = pk_fma(sgn2, e[slot][2][h], e[slot][1][h]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      v[i][0] = pk_col01(t[i][0], t[i][1]);
      v[i][1] = pk_col23(t[i][0], t[i][1]);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][q >> 2][q & 3], v[q >> 2][(q >> 1) & 1][q & 1], acc[q], 0, 0, 0);
    }
  };

  // one row chunk (WN_RC channels = S k-steps; parity `par` = rc & 1, compile-time) out of row buffer `bo`: stores row
  // chunk rc+1 (register set 1-par) into row buffer `nbo`, fetches row chunk `r2` = rc+2 into register set par.
  // Issue order is pinned with sched_barrier: the LDS reads of step g+1 go out BEFORE the transform + MFMAs of step g
  // (left alone, the scheduler sinks them to just before their use).
  auto rowchunk = [&](int bo, int nbo, int rc, int rn, int r2, int par) {
    constexpr int S = WN_RC / 2;
    load_rows(r2, par);
#pragma unroll
    for (int g = 0; g < S; ++g) {
      if (g < S - 1) {
        read_step(bo, g + 1, (g + 1) & 1);
      } else {
        store_rows(wn_lds + nbo, 1 - par);
        __syncthreads();
        read_step(nbo, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      mma_step(g & 3, g & 1);
      __builtin_amdgcn_sched_barrier(0);
      // a[g & 3] is free: fetch it for k-step g + 4 (of this row chunk or of the next one)
      load_u(g + 4 < S ? rc * (S / 4) + (g + 4) / 4 : rn * (S / 4) + (g + 4 - S) / 4, g & 3);
    }
  };

  // Cx % (2 WN_RC) == 0: an even number of row chunks; past the end the prefetches are redundant reloads nobody consumes
  const int nrc = p.Cx / WN_RC;
  // prologue: the first loads go out before everything that does not depend on them (LDS zero fill, accumulator init)
  load_rows(0, 0);
#pragma unroll
  for (int s = 0; s < 4; ++s) load_u(0, s);
  __builtin_amdgcn_sched_barrier(0);
  // Zero both row buffers once: the halo columns (index 0 and 33) are never written again, and neither are the rows
  // that fall outside the image (their owners skip the store below), so the padding costs nothing in the loop.
  for (int u = tid; u < 2 * WN_BUF / 4; u += NT_) reinterpret_cast<f32x4 *>(wn_lds)[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  __syncthreads();                               // zero fill done
  store_rows(wn_lds, 0);
  load_rows(1, 1);
  __syncthreads();
  read_step(0, 0, 0);

  for (int rc = 0; rc < nrc; rc += 2) {
    rowchunk(0, WN_BUF, rc, rc + 1, min(rc + 2, nrc - 1), 0);
    rowchunk(WN_BUF, 0, rc + 1, min(rc + 2, nrc - 1), min(rc + 3, nrc - 1), 1);
  }
  __syncthreads();                               // the LDS buffers are free: reuse them for the exchange below

  // Epilogue: Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]].  A lane holds rows (x, y) = (M[2wp], M[2wp+1]) of its
  // (channel, tile) pairs; the row transform is s0 = m0 + m1 + m2, s1 = m1 - m2 - m3, so each half contributes
  //   wp = 0: (x + y, y)      wp = 1: (x, -(x + y))
  // and after the (linear) column transform the two halves' 2x2 partial tiles are added.  Each wave finishes 8 of its
  // 16 channels: it hands the partial tiles of the other 8 to its partner through LDS and adds the partner's to its own.
  f32x4 *xch = reinterpret_cast<f32x4 *>(wn_lds) + ((wk * 2 + wt) * 2) * 8 * 64 + lane;   // [pair][owner wp][8][64 lanes]
  const int orow = row0 + wt * 4 + 2 * tr;
  float *y0 = p.Y + (long)n * p.M * HW + (long)orow * 32 + 2 * tc;
  // partial 2x2 tile of accumulator register r (compile-time) for position half W (compile-time)
  auto part_of = [&](int r, int W) {
    float p0[4], p1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x = acc[j][r], y = acc[4 + j][r];
      p0[j] = W ? x : x + y;
      p1[j] = W ? -(x + y) : y;
    }
    return f32x4{p0[0] + p0[1] + p0[2], p0[1] - p0[2] - p0[3], p1[0] + p1[1] + p1[2], p1[1] - p1[2] - p1[3]};
  };
  auto finish = [&](int r, f32x4 o) {
    const int m = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    const float bv = p.bias ? p.bias[m] : 0.f;
    f32x2 o0 = {apply_act(o[0] + bv, p.act, p.slope), apply_act(o[1] + bv, p.act, p.slope)};
    f32x2 o1 = {apply_act(o[2] + bv, p.act, p.slope), apply_act(o[3] + bv, p.act, p.slope)};
    float *ym = y0 + (long)m * HW;
    if (p.R) {
      const float *rm = p.R + (ym - p.Y);
      o0 += *reinterpret_cast<const f32x2 *>(rm);
      o1 += *reinterpret_cast<const f32x2 *>(rm + 32);
    }
    *reinterpret_cast<f32x2 *>(ym) = o0;
    *reinterpret_cast<f32x2 *>(ym + 32) = o1;
  };
  if (wp == 0) {                                 // wave-uniform: wp = 0 owns registers 0..7, hands over 8..15
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) xch[(8 + rr) * 64] = part_of(8 + rr, 0);
  } else {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) xch[rr * 64] = part_of(rr, 1);
  }
  __syncthreads();
  if (wp == 0) {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) finish(rr, part_of(rr, 0) + xch[rr * 64]);
  } else {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) finish(8 + rr, part_of(8 + rr, 1) + xch[(8 + rr) * 64]);
  }
}

// -------------------------------------------------------------------------------------------
// Weight gradient of the same conv in Winograd form.  With Y = A^T [U . V] A the gradient of g is
//   dg = G^T [ (A dY A^T) . (B^T d B) ] G      summed over tiles and images,
// i.e. again 16 independent GEMMs, M_p[k][c] += sum_tiles T_p[k][tile] V_p[c][tile], now with the reduction over
// tiles (MFMA k-dimension = 2 tiles), 16 multiplies per (k, c, tile) instead of 36.
//   T = A dY A^T (2x2 -> 4x4) and V = B^T d B (4x4 -> 4x4): per lane, in registers, from raw dy / x rows in LDS
//   M_p accumulates in registers over the workgroup's share of the tile rows; partial sums go to the workspace
//   dg = G^T M G and the sum over workgroups: wino_w3x3_reduce_kernel
// Signs: A's last row / column is (0, -1); the kernel uses (0, +1) (T' = s s^T . T, s = (1,1,1,-1)), which costs
// nothing and is undone in the reduce kernel by G' = diag(s) G.
// Workgroup: 64 k x 64 c, 8 waves = 2 (k) x 2 (c) x 2 (position halves: rows 2wp, 2wp+1 of the 4x4 grid); one chunk
// = one tile row of one image (16 tiles = 8 k-steps): x rows 2tr-1 .. 2tr+2 of 64 channels, dy rows 2tr, 2tr+1 of 64.
// LDS channel strides are 2 * odd floats: the 32 lanes of a ds_read_b64 group hold 32 different channels at the
// same pixel, so their 8-byte words land on 32 different bank pairs.
// -------------------------------------------------------------------------------------------
#define WW_XS 138                        // floats per x channel: 4 rows x 34 + 2
#define WW_DS 66                         // floats per dy channel: 2 rows x 32 + 2
#define WW_XBUF (64 * WW_XS)
#define WW_BUF (WW_XBUF + 64 * WW_DS)    // floats per LDS buffer
#define WW_LDS_BYTES (2 * WW_BUF * 4)    // double-buffered: 102 KB

struct WinoWParams {
  const float *DY, *X;
  float *part;                   // [split][16 positions][M][C]
  int N, M, C, H;
  int ntr, per_split;            // tile rows in total (N * H/2) and per split
};

__device__ __forceinline__ f32x2 pk_fma_n(f32x2 a, f32x2 b, f32x2 c) {      // a * b + c, result feeds an MFMA
  f32x2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ f32x2 pk_sumdiff(f32x2 a) {                      // (a.x + a.y, a.x - a.y), result feeds an MFMA
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(a));
  return r;
}

__global__ __launch_bounds__(512, 1) void wino_w3x3_kernel(WinoWParams p) {
  extern __shared__ __attribute__((aligned(16))) float ww_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave & 1, wc = (wave >> 1) & 1, wk = wave >> 2;
  const int l31 = lane & 31, half = lane >> 5;
  // XCD-aware mapping (workgroups go to the 8 XCDs round-robin in launch order): all (k, c) blocks of one split of the
  // tile rows sit on ONE XCD, so its dy / x rows come from HBM once and from that XCD's L2 for the other blocks
  int cb = blockIdx.x, kb = blockIdx.y, z = blockIdx.z;
  if ((gridDim.z & 7) == 0) {
    const int nb = gridDim.x * gridDim.y;
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int xcd = lin & 7, q = lin >> 3, b = q % nb;
    z = xcd + 8 * (q / nb);
    cb = b % gridDim.x;
    kb = b / gridDim.x;
  }
  const int H = p.H, trows = H >> 1;                   // tile rows per image
  const int t0 = z * p.per_split, t1 = min(p.ntr, t0 + p.per_split);

  // staging assignment: x 64 c x 4 rows x 8 segments = 4 per thread, dy 64 k x 2 rows x 8 segments = 2 per thread.
  // Byte offsets relative to the chunk's scalar base (image n, tile row tr); the first / last tile row of an image
  // has its row -1 / H outside the image: those lanes load a clamped (valid) row and store zeros.
  unsigned xo_mid[4], xo_top[4], xo_bot[4], dyo[2];
  int x_lds[4], d_lds[2];
  int x_r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = tid + 512 * i, line = u >> 3, seg = u & 7;
    const int c = line >> 2, r = line & 3;
    x_r[i] = r;
    x_lds[i] = c * WW_XS + r * 34 + 1 + seg * 4;
    const unsigned cbase = (unsigned)(cb * 64 + c) * H * 32 + seg * 4;   // relative to row 2tr-1 of channel 0 of image n
    xo_mid[i] = (cbase + r * 32) * 4;
    xo_top[i] = (cbase + max(r, 1) * 32) * 4;          // tr = 0: row -1 is outside, load row 0 instead (stored as zeros)
    xo_bot[i] = (cbase + min(r, 2) * 32) * 4;          // last tile row: row H is outside
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int u = tid + 512 * i, line = u >> 3, seg = u & 7;
    const int k = line >> 1, r = line & 1;
    d_lds[i] = WW_XBUF + k * WW_DS + r * 32 + seg * 4;
    dyo[i] = ((unsigned)(kb * 64 + k) * H * 32 + r * 32 + seg * 4) * 4;
  }
  // halo columns of the x rows (index 0 and 33): zero once in both buffers
  for (int u = tid; u < 2 * 64 * 4 * 2; u += 512) {
    const int b = u >> 9, rr = (u & 511) >> 1;
    ww_lds[b * WW_BUF + (rr >> 2) * WW_XS + (rr & 3) * 34 + (u & 1) * 33] = 0.f;
  }

  f32x16 acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  f32x4 xreg[4], dreg[2];
  bool xz[4];                                          // store zeros instead of the loaded row
  auto load_chunk = [&](int trow) {
    const int n = trow / trows, tr = trow - n * trows;
    const char *xb = reinterpret_cast<const char *>(p.X + ((long)n * p.C * H + 2 * tr - 1) * 32);   // uniform
    const char *db = reinterpret_cast<const char *>(p.DY + ((long)n * p.M * H + 2 * tr) * 32);
    if (tr == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) xreg[i] = *reinterpret_cast<const f32x4 *>(xb + xo_top[i]), xz[i] = x_r[i] == 0;
    } else if (tr == trows - 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) xreg[i] = *reinterpret_cast<const f32x4 *>(xb + xo_bot[i]), xz[i] = x_r[i] == 3;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) xreg[i] = *reinterpret_cast<const f32x4 *>(xb + xo_mid[i]), xz[i] = false;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) dreg[i] = *reinterpret_cast<const f32x4 *>(db + dyo[i]);
  };
  auto store_chunk = [&](float *buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float *d = buf + x_lds[i];
      if (xz[i]) {
        d[0] = 0.f;
        *reinterpret_cast<f32x2 *>(d + 1) = f32x2{0.f, 0.f};
        d[3] = 0.f;
      } else {
        d[0] = xreg[i][0];
        *reinterpret_cast<f32x2 *>(d + 1) = f32x2{xreg[i][1], xreg[i][2]};
        d[3] = xreg[i][3];
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float *d = buf + d_lds[i];
      *reinterpret_cast<f32x2 *>(d) = f32x2{dreg[i][0], dreg[i][1]};
      *reinterpret_cast<f32x2 *>(d + 2) = f32x2{dreg[i][2], dreg[i][3]};
    }
  };

  // B side (x -> V rows 2wp, 2wp+1), as in the forward kernel: wp = 0 reads d rows (0, 2, 1), wp = 1 reads (2, 1, 3);
  // slot 0 = e0 - e1, slot 1 = e1 + sgn * e2.
  // A side (dy -> T' rows 2wp, 2wp+1) from the tile's two dy rows D0, D1 (2 pixels each):
  //   wp = 0: rows 0, 1 = D0, D0 + D1      wp = 1: rows 2, 3' = D0 - D1, D1      -> slot 0 = D0 + b0 D1, slot 1 = a1 D0 + D1
  // and per row (x, y) the four columns are x, x + y, x - y, y.
  const float sgn = wp ? -1.f : 1.f, b0 = wp ? -1.f : 0.f, a1 = wp ? 0.f : 1.f;
  const f32x2 sgn2 = {sgn, sgn}, b02 = {b0, b0}, a12 = {a1, a1};
  const int er0 = wp ? 2 : 0, er1 = wp ? 1 : 2, er2 = wp ? 3 : 1;
  const int xl = (wc * 32 + l31) * WW_XS + 2 * half;             // + 4 s (tile 2s + half), + row * 34
  const float *rd0 = ww_lds + xl + er0 * 34, *rd1 = ww_lds + xl + er1 * 34, *rd2 = ww_lds + xl + er2 * 34;
  const float *rdd = ww_lds + WW_XBUF + (wk * 32 + l31) * WW_DS + 2 * half;   // + 4 s, + row * 32

  f32x2 e[2][3][2], dd[2][2];
  typedef const volatile f32x2 __attribute__((address_space(3))) *vp;
  auto read_step = [&](int bo, int s, int slot) {
    e[slot][0][0] = *(vp)(rd0 + bo + 4 * s);
    e[slot][0][1] = *(vp)(rd0 + bo + 4 * s + 2);
    e[slot][1][0] = *(vp)(rd1 + bo + 4 * s);
    e[slot][1][1] = *(vp)(rd1 + bo + 4 * s + 2);
    e[slot][2][0] = *(vp)(rd2 + bo + 4 * s);
    e[slot][2][1] = *(vp)(rd2 + bo + 4 * s + 2);
    dd[slot][0] = *(vp)(rdd + bo + 4 * s);
    dd[slot][1] = *(vp)(rdd + bo + 4 * s + 32);
  };
  auto mma_step = [&](int slot) {
    f32x2 t[2][2], v[2][2], sr[2], sd[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      t[0][h] = pk_sub(e[slot][0][h], e[slot][1][h]);
      t[1][h] = pk_fma(sgn2, e[slot][2][h], e[slot][1][h]);
    }
    sr[0] = pk_fma_n(b02, dd[slot][1], dd[slot][0]);
    sr[1] = pk_fma_n(a12, dd[slot][0], dd[slot][1]);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      v[i][0] = pk_col01(t[i][0], t[i][1]);
      v[i][1] = pk_col23(t[i][0], t[i][1]);
      sd[i] = pk_sumdiff(sr[i]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      acc[i * 4 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sr[i][0], v[i][0][0], acc[i * 4 + 0], 0, 0, 0);
      acc[i * 4 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sd[i][0], v[i][0][1], acc[i * 4 + 1], 0, 0, 0);
      acc[i * 4 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(sd[i][1], v[i][1][0], acc[i * 4 + 2], 0, 0, 0);
      acc[i * 4 + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(sr[i][1], v[i][1][1], acc[i * 4 + 3], 0, 0, 0);
    }
  };
  // one tile row out of buffer `bo`, staging tile row `nt` into buffer `nbo`; the LDS reads of step s+1 are issued
  // before the transforms + MFMAs of step s (order pinned with sched_barrier, see the forward kernel)
  auto chunk = [&](int bo, int nbo, int nt) {
    load_chunk(nt);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < 7) {
        read_step(bo, s + 1, (s + 1) & 1);
      } else {
        store_chunk(ww_lds + nbo);
        __syncthreads();
        read_step(nbo, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      mma_step(s & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if (t0 < t1) {
    load_chunk(t0);
    __syncthreads();                             // halo zero fill done
    store_chunk(ww_lds);
    __syncthreads();
    read_step(0, 0, 0);
    // the last chunk prefetches a clamped (valid) tile row that nobody consumes
    int t = t0;
    for (; t + 1 < t1; t += 2) {
      chunk(0, WW_BUF, t + 1);
      chunk(WW_BUF, 0, min(t + 2, t1 - 1));
    }
    if (t < t1) chunk(0, WW_BUF, t);
  }

  // partial sums: part[z][(2wp + i) * 4 + j][k][c]
  float *pz = p.part + (long)z * 16 * p.M * p.C + (long)(2 * wp) * 4 * p.M * p.C +
              (long)(kb * 64 + wk * 32 + 4 * half) * p.C + cb * 64 + wc * 32 + l31;
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) pz[(long)q * p.M * p.C + (long)((r & 3) + 8 * (r >> 2)) * p.C] = acc[q][r];
}

// dW[k][c][3][3] = G'^T (sum_z M'_z) G',  G' = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,-1]]
__global__ __launch_bounds__(256) void wino_w3x3_reduce_kernel(const float *__restrict__ part, float *__restrict__ dW, int MC,
                                                               int splits) {
  const int i = blockIdx.x * 256 + threadIdx.x;        // k * C + c
  if (i >= MC) return;
  float m[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) m[q] = 0.f;
  for (int z = 0; z < splits; ++z)
#pragma unroll
    for (int q = 0; q < 16; ++q) m[q] += part[((long)z * 16 + q) * MC + i];
  // columns first: w[i][s] = sum_j m[i][j] G'[j][s]
  float w[4][3];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const float m0 = m[a * 4], m1 = m[a * 4 + 1], m2 = m[a * 4 + 2], m3 = m[a * 4 + 3];
    w[a][0] = m0 + 0.5f * (m1 + m2);
    w[a][1] = 0.5f * (m1 - m2);
    w[a][2] = 0.5f * (m1 + m2) - m3;
  }
  float *o = dW + (long)i * 9;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    o[0 * 3 + s] = w[0][s] + 0.5f * (w[1][s] + w[2][s]);
    o[1 * 3 + s] = 0.5f * (w[1][s] - w[2][s]);
    o[2 * 3 + s] = 0.5f * (w[1][s] + w[2][s]) - w[3][s];
  }
}

}  // namespace lsps
#endif
