// The one-input-channel 7x7 stems (reference: src/trainers/lsps_nets.py:117,184 — LeakyReLUConv2d(1, 64, 7, stride, 3)) in the
// bf16 math mode on v_mfma_f32_32x32x16_bf16: f32 image in, C8 bf16 activation out; weight + bias gradient from C8 dy / y.
// The f32 kernels of conv_c1.h run the 49-tap contraction on the f32 matrix pipe (K = 2 per 64-cycle MFMA: 61 % of their time);
// here K = 64 = an 8 x 8 tap grid (row 7 / column 7 carry zero weights) costs 4 MFMAs of 32 cycles per 32 x 32 tile, which
// leaves both kernels to their memory traffic.  Operands are rounded to bf16 when the fragments are formed, like every other
// layer of the mode.
#ifndef LSPS_C8STEM_H
#define LSPS_C8STEM_H
#include "c8conv.h"
#include "c8wgrad.h"

namespace lsps {

struct C8StemParams {
  const float *X, *W, *bias;     // image [N][H][Wd], weights [64][R*S], bias [64] or null
  unsigned short *Y;             // [N][8][P][Q][8]
  int N, H, Wd, P, Q, R, S, stride, pad;
  int TP, rows, LW;              // output rows per workgroup, staged input rows, LDS row stride (floats)
  float lrelu;                   // v = max(v, v * lrelu) (1: none)
};

__device__ __forceinline__ bf16x8 c8_cvt8(const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
  return o;
}

// grid (P / TP, N); 256 threads; wave w handles the 32-pixel segments w, w + 4, ... of the workgroup's TP output rows
__global__ __launch_bounds__(256) void c8_stem_fwd_kernel(C8StemParams p) {
  extern __shared__ __attribute__((aligned(16))) float st_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int n = blockIdx.y, p0 = blockIdx.x * p.TP;

  // image rows p0*stride - pad ... with the zero halo (rows / columns beyond the image, and the extra row / columns of the
  // 8 x 8 tap grid, are zeros)
  const float *xn = p.X + (long)n * p.H * p.Wd;
  const int row0 = p0 * p.stride - p.pad;
  for (int u = tid; u < p.rows * p.LW; u += 256) {
    const int r = u / p.LW, c = u - r * p.LW;
    const int ih = row0 + r, iw = c - p.pad;
    const bool ok = ih >= 0 && ih < p.H && iw >= 0 && iw < p.Wd;
    const float v = xn[(long)min(max(ih, 0), p.H - 1) * p.Wd + min(max(iw, 0), p.Wd - 1)];
    st_lds[u] = ok ? v : 0.f;
  }
  // A operands: lane (m = 32 i + l31, k-half) of k-step ks holds tap row 2 ks + half, columns 0..7
  bf16x8 af[4][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 2 * ks + half, m = 32 * i + l31;
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = (r < p.R && c < p.S) ? p.W[(long)m * p.R * p.S + min(r, p.R - 1) * p.S + min(c, p.S - 1)] : 0.f;
      af[ks][i] = c8_cvt8(v);
    }
  float bias_r[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[i][r] = p.bias ? p.bias[i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] : 0.f;
  __syncthreads();

  const int qblocks = p.Q / 32, nseg = p.TP * qblocks;
  const long PQ = (long)p.P * p.Q;
  const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(p.Y + (long)n * 64 * PQ, 0, 0x7fffffff, 0x00020000);
  typedef unsigned st_u32x2 __attribute__((ext_vector_type(2)));
  for (int seg = wave; seg < nseg; seg += 4) {
    const int pr = seg / qblocks, q0 = (seg - pr * qblocks) * 32;
    const float *Bp = st_lds + (pr * p.stride + half) * p.LW + (q0 + l31) * p.stride;      // tap row 2 ks + half: + 2 ks * LW
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = bias_r[i][r];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = Bp[2 * ks * p.LW + c];
      const bf16x8 bf = c8_cvt8(v);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bf, acc[i], 0, 0, 0);
    }
    const unsigned vo = (unsigned)(((p0 + pr) * p.Q + q0 + l31) * 16 + half * 8);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = acc[i][rq * 4 + e];
          o[e] = (__bf16)fmaxf(a, a * p.lrelu);
        }
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(st_u32x2, o), crs, vo, (i * 4 + rq) * (int)PQ * 16, 0);
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight + bias gradient:  dW[k][r][c] = sum_{n,p,q} g[n][k][p][q] x[n][p s + r - pad][q s + c - pad],  db[k] = sum g,
// g = dy * (y > 0 ? 1 : slope) formed while the C8 tiles are staged.  M = 64 channels, columns = the 8 x 8 tap grid (column 63
// = a row of ones: the bias gradient), reduction over pixels (16 per MFMA): the g fragments come out of the C8 units by the
// transposing LDS read (c8wgrad.h), the image fragments are 8 pixels of one tap per lane.  One iteration = 128 output pixels
// (RB = 128 / Q rows of one image), one 32-pixel segment per wave; partial sums part[block][64][64].
// ------------------------------------------------------------------------------------------------------------------
#define C8SW_GPLANE (128 * 16 + 64)                  // bytes per channel-group plane of g (128 pixels + bank rotation)
#define C8SW_GBYTES (8 * C8SW_GPLANE)                // 16896
#define C8SW_XMAX 2176                               // floats of staged image rows

struct C8StemWParams {
  const float *X;
  const unsigned short *DY, *Yc;   // [N][8][P][Q][8]
  float *part;                     // [blocks][64 * 64]
  int N, H, Wd, P, Q, R, S, stride, pad;
  int LW, RB, xrows;               // LDS row stride (floats), output rows per iteration, staged input rows
  int iters_total, iters_per_block;
  float slope;
};

__global__ __launch_bounds__(256) void c8_stem_wgrad_kernel(C8StemWParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char sw_lds[C8SW_GBYTES + C8SW_XMAX * 4];
  unsigned char *gs = sw_lds;
  float *xs = reinterpret_cast<float *>(sw_lds + C8SW_GBYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const long PQ = (long)p.P * p.Q;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging assignment: g unit u = tid + 256 i (i < 4) -> (channel group u / 128, pixel u % 128); image element u -> (row, col)
  int g_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = tid + 256 * i, g8 = u >> 7, px = u & 127;
    const int rb = px / p.Q, cq = px - rb * p.Q;
    g_off[i] = g8 * (int)PQ + rb * p.Q + cq;                    // in 16-byte units
  }
  const int xcount = p.xrows * p.LW;
  constexpr int NX = (C8SW_XMAX + 255) / 256;
  int x_r[NX], x_c[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int u = tid + 256 * i;
    x_r[i] = u / p.LW;
    x_c[i] = u - x_r[i] * p.LW - p.pad;
  }
  u32x4 greg[4];
  float xreg[NX];
  auto fetch = [&](int it) {
    const int n = it / (p.P / p.RB), pr = (it - n * (p.P / p.RB)) * p.RB;
    const u32x4 *dq = reinterpret_cast<const u32x4 *>(p.DY) + (long)n * 8 * PQ + (long)pr * p.Q;
    const u32x4 *yq = reinterpret_cast<const u32x4 *>(p.Yc) + (long)n * 8 * PQ + (long)pr * p.Q;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16x8 dv = __builtin_bit_cast(bf16x8, dq[g_off[i]]), yv = __builtin_bit_cast(bf16x8, yq[g_off[i]]);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)dv[e];
        o[e] = (__bf16)c8_sel_nonpos((float)yv[e], d * p.slope, d);
      }
      greg[i] = __builtin_bit_cast(u32x4, o);
    }
    const float *xn = p.X + (long)n * p.H * p.Wd;
    const int row0 = pr * p.stride - p.pad;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int ih = row0 + x_r[i], iw = x_c[i];
      const bool ok = tid + 256 * i < xcount && ih >= 0 && ih < p.H && iw >= 0 && iw < p.Wd;
      const float v = xn[(long)min(max(ih, 0), p.H - 1) * p.Wd + min(max(iw, 0), p.Wd - 1)];
      xreg[i] = ok ? v : 0.f;
    }
  };

  const int it_begin = blockIdx.x * p.iters_per_block;
  const int it_end = min(p.iters_total, it_begin + p.iters_per_block);
  if (it_begin < it_end) fetch(it_begin);

  // this wave's 32-pixel segment of the iteration's 128: row srow, first column sq0
  const int qblocks = p.Q / 32;
  const int srow = wave / qblocks, sq0 = (wave - srow * qblocks) * 32;
  // transposing-read addresses of the g fragments (c8wgrad.h): 16-lane group g4: channels 16 (g4 & 1) .., pixels 8 (g4 >> 1) ..
  const int g4 = lane >> 4, i16 = lane & 15;
  const int cgl = 2 * (g4 & 1) + ((i16 & 3) >> 1), px = 8 * (g4 >> 1) + (i16 >> 2), byte = 8 * (i16 & 1);
  const unsigned a_base = (unsigned)(cgl * C8SW_GPLANE + (wave * 32 + px) * 16 + byte);
  // image fragments: lane (tap t = 32 j + l31 -> (r, c) = (t >> 3, t & 7), k-half): 8 consecutive output pixels of that tap
  int b_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int t = 32 * j + l31;
    b_off[j] = (srow * p.stride + (t >> 3)) * p.LW + (sq0 + 8 * half) * p.stride + (t & 7);
  }
  const bool ones = l31 == 31;                                   // j = 1: tap 63 = the bias-gradient column

  for (int it = it_begin; it < it_end; ++it) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = tid + 256 * i;
      *reinterpret_cast<u32x4 *>(gs + (u >> 7) * C8SW_GPLANE + (u & 127) * 16) = greg[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i)
      if (tid + 256 * i < xcount) xs[tid + 256 * i] = xreg[i];
    __syncthreads();
    if (it + 1 < it_end) fetch(it + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {                             // 16 pixels each
      bf16x8 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = c8_tr_frag(gs + a_base + i * 4 * C8SW_GPLANE + ks * 256, gs + a_base + i * 4 * C8SW_GPLANE + ks * 256 + 64);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = xs[b_off[j] + (ks * 16 + e) * p.stride];
        if (j == 1 && ones) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = 1.f;
        }
        bf[j] = c8_cvt8(v);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }

  // the four waves' partial sums added in a fixed order through LDS (deterministic), then written once per block
  float *red = reinterpret_cast<float *>(sw_lds);                // [64 k][64 t] = 16 KB
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int k = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, t = j * 32 + l31;
            const float v = acc[i][j][r];
            red[k * 64 + t] = w == 0 ? v : red[k * 64 + t] + v;
          }
    }
  }
  __syncthreads();
  float *out = p.part + (long)blockIdx.x * 4096;
  for (int u = tid; u < 4096; u += 256) out[u] = red[u];
}

// dW[k][r * S + c] = sum_b part[b][k][8 r + c],  db[k] = sum_b part[b][k][63]
__global__ __launch_bounds__(256) void c8_stem_wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ dW, float *__restrict__ db,
                                                                   int R, int S, int blocks) {
  // workgroup = 64 consecutive elements x 4 interleaved slices of the partial blocks (fixed summation order)
  __shared__ float red[4][64];
  const int tid = threadIdx.x, u = blockIdx.x * 64 + (tid & 63), sl = tid >> 6;
  const int k = u >> 6, t = u & 63, r = t >> 3, c = t & 7;
  const bool isw = r < R && c < S, isb = t == 63;
  float s = 0.f;
  if (isw || isb)
    for (int b = sl; b < blocks; b += 4) s += part[(long)b * 4096 + u];
  red[sl][tid & 63] = s;
  __syncthreads();
  if (sl != 0) return;
  s = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  if (isw)
    dW[(long)k * R * S + r * S + c] = s;
  else if (isb && db)
    db[k] = s;
}

// ------------------------------------------------------------------------------------------------------------------
// Input gradient of a stem (the discriminator's, inside gen_update: the generated images need it):
//   dx[n][y][x] = sum_{k,r,c : y = p s + r - pad, x = q s + c - pad} w[k][r][c] g[n][k][p][q],   g = dy * (y_saved > 0 ? 1 : slope)
// as a tap GEMM on the bf16 pipe, Z[t = 8 r + c][pixel] = sum_k w[k][t] g[k][pixel] (64 taps x 64 channels: 8 MFMAs per 32 pixels,
// the g fragments straight from the C8 tensors with the activation mask applied in registers), whose rows are then added into the
// image at their tap's offset.  Workgroup = one image x a band of TY image rows; every wave owns a PRIVATE LDS copy of the band
// (ds_add_f32 in program order: lanes of one instruction hit distinct columns, so the sum order is fixed = deterministic), the
// four copies are summed in a fixed order when the band is written.  Replaces: activation pass + C8 -> NCHW conversion + f32
// tap GEMM + col2im.
// ------------------------------------------------------------------------------------------------------------------
#define C8SD_TY 32

struct C8StemDParams {
  const unsigned short *DY, *Yc;   // [N][8][P][Q][8]
  const float *W;                  // [64][R * S]
  float *dX;                       // [N][H][Wd]
  int N, H, Wd, P, Q, R, S, stride, pad, LW;
  float slope;
};

__global__ __launch_bounds__(256) void c8_stem_dgrad_kernel(C8StemDParams p) {
  extern __shared__ __attribute__((aligned(16))) float sd_lds[];      // [4 waves][TY][LW] + [4][64] dummy slots
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int n = blockIdx.y, y0 = blockIdx.x * C8SD_TY;
  const int tile = C8SD_TY * p.LW;
  for (int u = tid; u < 4 * tile + 256; u += 256) sd_lds[u] = 0.f;

  // A operands: Wt[t = 32 i + l31][k = 16 ks + 8 half + e], t = 8 r + c
  bf16x8 af[4][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = 32 * i + l31, r = t >> 3, c = t & 7;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
        v[e] = (r < p.R && c < p.S) ? p.W[(long)(16 * ks + 8 * half + e) * p.R * p.S + min(r, p.R - 1) * p.S + min(c, p.S - 1)] : 0.f;
      af[ks][i] = c8_cvt8(v);
    }
  __syncthreads();

  // small-map rows that reach the band: p * stride + r - pad in [y0, y0 + TY) for some r in [0, R)
  const int p_lo = max(0, (y0 + p.pad - (p.R - 1) + p.stride - 1) / p.stride);
  const int p_hi = min(p.P - 1, (y0 + C8SD_TY - 1 + p.pad) / p.stride);
  const int qblocks = p.Q / 32, nitems = (p_hi - p_lo + 1) * qblocks;
  const long PQ = (long)p.P * p.Q;
  const u32x4 *dq = reinterpret_cast<const u32x4 *>(p.DY) + (long)n * 8 * PQ;
  const u32x4 *yq = reinterpret_cast<const u32x4 *>(p.Yc) + (long)n * 8 * PQ;
  float *mine = sd_lds + wave * tile;
  float *dummy = sd_lds + 4 * tile + wave * 64 + lane;           // where the adds of tap column 7 (zero weights) go
  // the C8 units of an item are fetched one item ahead (a wave's items are independent; with two workgroups per CU nothing
  // else hides the HBM latency of a load -> mask -> MFMA -> scatter chain)
  auto fetch = [&](int item, u32x4 (&dv)[4], u32x4 (&yv)[4]) {
    const int pr = p_lo + item / qblocks, q0 = (item % qblocks) * 32;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const long u = (long)(2 * ks + half) * PQ + (long)pr * p.Q + q0 + l31;
      dv[ks] = dq[u];
      yv[ks] = yq[u];
    }
  };
  auto process = [&](int item, const u32x4 (&dvr)[4], const u32x4 (&yvr)[4]) {
    const int pr = p_lo + item / qblocks, q0 = (item % qblocks) * 32;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 dv = __builtin_bit_cast(bf16x8, dvr[ks]), yv = __builtin_bit_cast(bf16x8, yvr[ks]);
      bf16x8 g;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)dv[e];
        g[e] = (__bf16)c8_sel_nonpos((float)yv[e], d * p.slope, d);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], g, acc[i], 0, 0, 0);
    }
    // acc[i][rr]: tap t = 32 i + (rr & 3) + 8 (rr >> 2) + 4 half -> r = 4 i + (rr >> 2), c = (rr & 3) + 4 half, pixel (pr, q0 + l31)
    const int xb = (q0 + l31) * p.stride;                       // LDS column of tap column 0 (= image column + pad)
    // Scatter-add into the wave's private band by plain read - add - write (LDS float atomics run at well under one lane per
    // cycle on this part: the first build, ds_add_f32, spent 95 % of its time in them).  Four rounds, every lane active in
    // each: lanes 0-31 take tap column e, lanes 32-63 column 4 + ((e + 1) & 3) - columns of opposite parity, so (image columns
    // advance by `stride` per lane) no two lanes of an instruction ever meet on one address; the one column beyond the tap grid
    // (c = 7) goes to a per-lane dummy slot.  A round reads all tap rows, then writes them: rows never alias, and the LDS
    // executes one wave's instructions in order, so round e + 1 reads what round e wrote.
    typedef volatile float __attribute__((address_space(3))) *vlp;
    // (Odd strides: columns of opposite parity can still meet, so the two half-waves take turns: `phases` = 2.)
    const int phases = (p.stride & 1) ? 2 : 1;
    for (int ph = 0; ph < phases; ++ph)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int e1 = (e + 1) & 3;
      const int c = half ? 4 + e1 : e;
      const bool mine_turn = phases == 1 || half == ph;
      float old[8];
      vlp dst[8];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int r = 4 * i + rq, yy = pr * p.stride + r - p.pad - y0;     // uniform
          const bool ok = r < p.R && yy >= 0 && yy < C8SD_TY;
          dst[r] = (vlp)((ok && c < p.S && mine_turn) ? mine + yy * p.LW + xb + c : dummy);
          old[r] = *dst[r];
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int r = 4 * i + rq;
          *dst[r] = old[r] + (half ? acc[i][rq * 4 + e1] : acc[i][rq * 4 + e]);
        }
    }
  };
  u32x4 da[4], ya[4], db_[4], yb[4];
  if (wave < nitems) fetch(wave, da, ya);
  for (int item = wave; item < nitems; item += 8) {
    if (item + 4 < nitems) fetch(item + 4, db_, yb);
    process(item, da, ya);
    if (item + 4 < nitems) {
      if (item + 8 < nitems) fetch(item + 8, da, ya);
      process(item + 4, db_, yb);
    }
  }
  __syncthreads();
  float *dxn = p.dX + (long)n * p.H * p.Wd;
  for (int u = tid; u < C8SD_TY * p.Wd; u += 256) {
    const int yy = u / p.Wd, xx = u - yy * p.Wd;
    if (y0 + yy >= p.H) continue;
    const int o = yy * p.LW + xx + p.pad;
    dxn[(long)(y0 + yy) * p.Wd + xx] = (sd_lds[o] + sd_lds[tile + o]) + (sd_lds[2 * tile + o] + sd_lds[3 * tile + o]);
  }
}

}  // namespace lsps
#endif
