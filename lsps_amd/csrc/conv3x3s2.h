// 3x3 / stride 2 / pad 1 kernels (down-sampling convs, up-sampling transposed convs): forward direction, transposed direction, weight gradient.
#ifndef LSPS_CONV3X3S2_H
#define LSPS_CONV3X3S2_H
#include "conv_types.h"
#include "conv3x3.h"

namespace lsps {

// -------------------------------------------------------------------------------------------
// "Forward direction" kernel specialised for 3x3 / STRIDE 2 / pad 1 (down-sampling convs forward, up-sampling
// transposed convs' dgrad): in = big image [N][Cx][2P][2Q], out = small image [N][M][P][Q], Q % 32 == 0.
// Same structure as igemm_f3x3_kernel (tile 128 channels x 4 output rows x 32 columns, 8 input channels per chunk,
// weights [chunk][tap][8][Mp]); the 9 input rows of the tile are staged DE-INTERLEAVED by column parity like in
// igemm_w3x3s2_kernel (per row: O'[33] = odd columns with the left neighbour first, then E[32] = even columns), so the
// stride-2 taps are unit-stride LDS reads: s=0 -> O'[q], s=1 -> E[q], s=2 -> O'[q+1].
// -------------------------------------------------------------------------------------------
#define FS2_ROW 65
#define FS2_ROWS 9                                   // 2 * 4 + 1 input rows per tile
#define FS2_CH (FS2_ROWS * FS2_ROW)                  // 585 floats per channel (585 % 32 = 9)

struct FS2Params {
  const float *X, *Wp, *bias, *zero;
  float *Y;
  int Cx, P, Q, M, Mp;           // output [P][Q]; input [2P][2Q]
  int qblocks, tiles_per_img;    // Q / 32, (P / 4) * qblocks
  int act;
  float slope;
};

// CC = input channels per chunk.  8 (the bf16 mode needs it: K = 16 = 2 taps x 8 channels): 55.6 KB of LDS and ~210 registers = two
// workgroups per CU.  4: 27.8 KB and half the prefetch registers = three to four per CU, 72 instead of 144 MFMAs per wave
// between the barrier pairs (measured in DESIGN.md 3.9).
template <bool BF16, int CC>
__global__ __launch_bounds__(256, CC == 8 ? 2 : 3) void igemm_f3x3s2_kernel(FS2Params p) {
  constexpr int BM = 128, RC = CC * 9;
  constexpr int A4 = (RC * BM / 4 + 255) / 256;                    // float4 of weights per thread per chunk (9, or 4.5 -> 5)
  constexpr int A_LASTW = ((RC * BM / 4) % 256) ? ((RC * BM / 4) % 256) / 64 : 4;   // waves that take part in the last round
  constexpr int LINES = CC * FS2_ROWS;                          // 72 (channel, row) lines of 64 columns
  constexpr int B4 = (LINES * 16 + 255) / 256;                     // 5 float4 of input per thread per chunk (4.5)
  constexpr int B_LASTW = ((LINES * 16) % 256) ? ((LINES * 16) % 256) / 64 : 4;
  constexpr int H_WAVES = (LINES + 63) / 64;                       // waves that fetch the halo column
  static_assert((RC * BM / 4) % 64 == 0 && (LINES * 16) % 64 == 0, "the last staging round ends on a wave boundary");
  __shared__ __attribute__((aligned(16))) float lds[RC * BM + CC * FS2_CH + 4];
  float *As = lds, *Bs = lds + RC * BM;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.y * BM;
  const int n = blockIdx.x / p.tiles_per_img;
  const int rem = blockIdx.x - n * p.tiles_per_img;
  const int p0 = (rem / p.qblocks) * 4, q0 = (rem - (rem / p.qblocks) * p.qblocks) * 32;
  const int Hx = 2 * p.P, Wx = 2 * p.Q, HWx = Hx * Wx;

  // Staging without a branch or a 64-bit lane address: both operands come through buffer descriptors (uniform
  // base + ONE 32-bit lane offset + a scalar chunk offset); a lane whose element lies outside the image carries an offset
  // past num_records and gets 0 from the range check; the ragged last round of a staging loop ends on a wave boundary, so
  // the only conditions left are wave-uniform.
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.X + (long)n * p.Cx * HWx), 0, p.Cx * HWx * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.Wp + m0), 0, (p.Cx * 9 * p.Mp - m0) * 4, 0x00020000);
  const unsigned a_vo = (unsigned)((tid >> 5) * p.Mp + (tid & 31) * 4) * 4u;       // round i: + 8 i rows (uniform)
  const int a_row8 = 8 * p.Mp * 4, a_chunk = RC * p.Mp * 4, b_chunk = CC * HWx * 4;

  int b_lds[B4];
  unsigned b_vo[B4];
#pragma unroll
  for (int i = 0; i < B4; ++i) {
    const int u = tid + 256 * i;
    const int line = u >> 4, c4 = u & 15;
    const int chn = line / FS2_ROWS, r = line - chn * FS2_ROWS;
    const int ih = 2 * p0 - 1 + r;                                 // ih <= 2 p0 + 7 < 2P always
    b_lds[i] = chn * FS2_CH + r * FS2_ROW + 2 * c4;
    b_vo[i] = (u < LINES * 16 && ih >= 0) ? (unsigned)(chn * HWx + ih * Wx + 2 * q0 + c4 * 4) * 4u : OOB;
  }
  // column 2 q0 - 1 of every line; lanes beyond the last line fetch nothing and store into the spare slot
  const int h_chn = tid / FS2_ROWS, h_r = tid - h_chn * FS2_ROWS;
  const bool h_ok = tid < LINES && (2 * p0 - 1 + h_r) >= 0 && q0 > 0;
  const unsigned h_vo = h_ok ? (unsigned)(h_chn * HWx + (2 * p0 - 1 + h_r) * Wx + 2 * q0 - 1) * 4u : OOB;
  const int h_lds = tid < LINES ? h_chn * FS2_CH + h_r * FS2_ROW : CC * FS2_CH;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 areg[A4], breg[B4];
  float hreg = 0.f;
  const int nchunks = p.Cx / CC;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const float *Ap = As + half * BM + wm * 64 + l31;
  const float *Bp = Bs + half * FS2_CH + wn * 4 * FS2_ROW + l31;     // wave's output rows wn*2 + j -> input rows 2(wn*2+j) + r

  for (int ch = -1; ch < nchunks; ++ch) {
    if (ch >= 0) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < A4; ++i)
        if (i < A4 - 1 || wave < A_LASTW) *reinterpret_cast<f32x4 *>(As + (tid + 256 * i) * 4) = areg[i];
#pragma unroll
      for (int i = 0; i < B4; ++i)
        if (i < B4 - 1 || wave < B_LASTW) {
          float *d = Bs + b_lds[i];
          d[33] = breg[i][0];                                      // E[2 c4]
          d[1] = breg[i][1];                                       // O'[2 c4 + 1]
          d[34] = breg[i][2];                                      // E[2 c4 + 1]
          d[2] = breg[i][3];                                       // O'[2 c4 + 2]
        }
      if (wave < H_WAVES) Bs[h_lds] = hreg;                        // O'[0]
      __syncthreads();
    }
    if (ch + 1 < nchunks) {
      const int ao = (ch + 1) * a_chunk, bo = (ch + 1) * b_chunk;  // uniform
#pragma unroll
      for (int i = 0; i < A4; ++i)
        if (i < A4 - 1 || wave < A_LASTW)
          areg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, a_vo, ao + i * a_row8, 0));
#pragma unroll
      for (int i = 0; i < B4; ++i)
        if (i < B4 - 1 || wave < B_LASTW)
          breg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, b_vo[i], bo, 0));
      if (wave < H_WAVES) hreg = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, h_vo, bo, 0));
    }
    if (ch >= 0 && BF16) {
      // bf16 MFMA mode: K = 16 = (2 taps) x (8 channels): lanes 0-31 carry tap 2g, lanes 32-63 tap 2g+1 (the fifth
      // group's upper half re-reads tap 8 and is zeroed); operands rounded to bf16 in registers
      const float *A0 = As + wm * 64 + l31;
      const float *B0 = Bs + wn * 4 * FS2_ROW + l31;
#pragma unroll
      for (int g = 0; g < 5; ++g) {
        const int t0 = 2 * g, t1 = (2 * g + 1 <= 8) ? 2 * g + 1 : 8;
        const int tsel = half ? t1 : t0;
        const int tr = tsel / 3, ts = tsel - tr * 3;
        const int boff = tr * FS2_ROW + (ts == 1 ? 33 : (ts == 2 ? 1 : 0));
        const bool dead = (2 * g + 1 > 8) && half;
        bf16x8 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) af[i][e] = (__bf16)A0[(tsel * CC + e) * BM + i * 32];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float v = B0[e * FS2_CH + 2 * j * FS2_ROW + boff];
            bf[j][e] = (__bf16)(dead ? 0.f : v);
          }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
    if (ch >= 0 && !BF16) {
      // k-pair kk = (tap t, channel pair cp); the operands of pair kk + 1 are fetched from LDS BEFORE the four MFMAs of pair
      // kk are issued (the compiler puts each read right in front of its first use otherwise).  Measured: 128.9 -> 130.9
      // TFLOP/s in the step and 141 -> 124 registers; the same change on the transposed kernel bought nothing.
      constexpr int NK = 9 * (CC / 2);
      float a_nx[2], b_nx[2];
      auto rd = [&](int kk) {
        const int t = kk / (CC / 2), cp = kk - t * (CC / 2);
        const int tr = t / 3, ts = t - tr * 3;
        const int coff = ts == 1 ? 33 : (ts == 2 ? 1 : 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) a_nx[i] = Ap[2 * kk * BM + i * 32];
#pragma unroll
        for (int j = 0; j < 2; ++j) b_nx[j] = Bp[2 * cp * FS2_CH + (2 * j + tr) * FS2_ROW + coff];
      };
      rd(0);
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        const float a[2] = {a_nx[0], a_nx[1]}, b[2] = {b_nx[0], b_nx[1]};
        __builtin_amdgcn_sched_barrier(0);
        if (kk + 1 < NK) rd(kk + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  const long PQ = (long)p.P * p.Q;
  // Epilogue with little VALU (it runs on the datapath the MFMAs of the other workgroups need): when the output channels come
  // in whole groups of 8 (validity is then wave-uniform per register) the bias is fetched once per register instead of per
  // store, LeakyReLU with 0 <= slope <= 1 is max(v, slope v), and the stores go through a buffer descriptor (lane offset =
  // pixel + the lane half's four channels, channel = scalar offset): no 64-bit lane address, no per-lane branch.
  const bool lrelu01 = p.act == LSPS_ACT_LRELU && p.slope >= 0.f && p.slope <= 1.f;
  if ((p.M & 7) == 0 && (p.act == LSPS_ACT_NONE || lrelu01) && (long)p.M * PQ * 4 < (1L << 31)) {
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.Y + (long)n * p.M * PQ, 0, p.M * (int)PQ * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.bias ? p.bias : p.Y), 0,
                                                                       p.bias ? p.M * 4 : 0, 0x00020000);
    const int pq4 = (int)PQ * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mu = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2);      // + 4 half: uniform validity (M % 8 == 0)
        if (mu >= p.M) continue;
        // (no bias: the descriptor has 0 records and the load returns 0)
        const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (unsigned)(half * 16), mu * 4, 0));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float v = acc[i][j][r] + bv;
          if (lrelu01) v = fmaxf(v, v * p.slope);
          const unsigned vo = (unsigned)(((p0 + wn * 2 + j) * p.Q + q0 + l31) * 4 + half * 4 * pq4);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, vo, mu * pq4, 0);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float *yb = p.Y + (long)n * p.M * PQ + (long)(p0 + wn * 2 + j) * p.Q + q0 + l31;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < p.M) {
          float v = acc[i][j][r];
          if (p.bias) v += p.bias[m];
          yb[(long)m * PQ] = apply_act(v, p.act, p.slope);
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------
// "Transposed direction" kernel specialised for 3x3 / STRIDE 2 / pad 1 (the up-sampling transposed convs forward,
// the down-sampling convs' dgrad): in = small image [N][Cx][Hs][Ws] (Ws % 32 == 0), out = big image [N][M][2Hs][2Ws]
//   out[m][2p+a][2q+b] = sum_c sum_{(r,s) in class(a,b)} W(m,c,r,s) * in[c][p + dh(r)][q + dw(s)]
// with class rows a=0: r=1 (dh 0); a=1: r=0 (dh +1), r=2 (dh 0), and the same for columns.  A workgroup owns a tile
// of the SMALL image (TR rows x 32 columns) for one row parity a (blockIdx.z) and BOTH column parities: the raw
// input rows of 16 channels are staged once in LDS (TR+1 rows x 33 columns, zero past the edges) and the 3 (a=0)
// or 6 (a=1) taps are shifted LDS reads; the two column classes are separate accumulators that the epilogue
// interleaves into float2 stores (full 256-byte rows instead of stride-2 scatter).
// BM = 128: TR = 4, waves 2x2;  BM = 64 (64-channel outputs): TR = 8, waves 1x4.  Each wave: 64 m x 2 rows x 2 classes.
// -------------------------------------------------------------------------------------------
#define TS_CC 16
#define TS_LDS_FLOATS (6 * TS_CC * 128 + TS_CC * 5 * 34)      // a = 1, BM = 128 (the largest of the four variants)

struct TS2Params {
  const float *X, *Wp, *bias, *zero;
  float *Y;
  int Cx, Hs, Ws, M, Mp;
  int qblocks, tiles_per_img;    // Ws / 32, (Hs / TR) * qblocks
  int act;
  float slope;
};

template <int APAR, int BM, bool BF16>
__device__ __forceinline__ void ts2_body(const TS2Params &p, float *lds) {
  constexpr int NT = APAR ? 6 : 3;                           // taps of this row class
  constexpr int WAVES_M = BM / 64, WAVES_N = 4 / WAVES_M, TR = 2 * WAVES_N;
  constexpr int ROWS = TR + 1, CHS = ROWS * 34;
  constexpr int AROWS = NT * TS_CC;
  constexpr int A4 = AROWS * BM / 4 / 256;
  constexpr int B4 = (TS_CC * ROWS * 8 + 255) / 256;
  static_assert(AROWS * BM + TS_CC * CHS <= TS_LDS_FLOATS, "LDS budget");
  float *As = lds, *Bs = lds + AROWS * BM;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.y * BM;
  const int n = blockIdx.x / p.tiles_per_img;
  const int rem = blockIdx.x - n * p.tiles_per_img;
  const int p0 = (rem / p.qblocks) * TR, q0 = (rem - (rem / p.qblocks) * p.qblocks) * 32;
  const int HWs = p.Hs * p.Ws;

  // branch-free staging through buffer descriptors, as in igemm_f3x3s2_kernel: rows past the bottom edge / the column past
  // the right edge carry an out-of-range lane offset and read 0; the ragged last round ends on a wave boundary
  constexpr unsigned OOB = 0x80000000u;
  constexpr int B_REM = (TS_CC * ROWS * 8) % 256, B_LASTW = B_REM ? B_REM / 64 : 4;
  constexpr int H_WAVES = (TS_CC * ROWS + 63) / 64;
  static_assert(B_REM % 64 == 0, "the last staging round ends on a wave boundary");
  static_assert((AROWS * BM / 4) % 256 == 0 && (BM == 128 || BM == 64), "weight rounds are full");
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.X + (long)n * p.Cx * HWs), 0, p.Cx * HWs * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.Wp + m0), 0, (p.Cx * 9 * p.Mp - m0) * 4, 0x00020000);
  const unsigned a_vo = (unsigned)((tid / (BM / 4)) * p.Mp + (tid % (BM / 4)) * 4) * 4u;
  const int mp4 = p.Mp * 4, a_chunk = 9 * TS_CC * mp4, b_chunk = TS_CC * HWs * 4;

  int b_lds[B4];
  unsigned b_vo[B4];
#pragma unroll
  for (int i = 0; i < B4; ++i) {
    const int u = tid + 256 * i;
    const int line = u >> 3, c4 = u & 7;
    const int chn = line / ROWS, r = line - chn * ROWS;
    b_lds[i] = chn * CHS + r * 34 + c4 * 4;
    b_vo[i] = (u < TS_CC * ROWS * 8 && (p0 + r) < p.Hs) ? (unsigned)(chn * HWs + (p0 + r) * p.Ws + q0 + c4 * 4) * 4u : OOB;
  }
  // column q0 + 32 (the right neighbour of the tile): one scalar per (channel, row) line; surplus lanes use the spare slot
  const int h_chn = tid / ROWS, h_r = tid - h_chn * ROWS;
  const bool h_use = tid < TS_CC * ROWS;
  const bool h_ok = h_use && (p0 + h_r) < p.Hs && (q0 + 32) < p.Ws;
  const unsigned h_vo = h_ok ? (unsigned)(h_chn * HWs + (p0 + h_r) * p.Ws + q0 + 32) * 4u : OOB;
  const int h_lds = h_use ? h_chn * CHS + h_r * 34 + 32 : TS_LDS_FLOATS - AROWS * BM;

  f32x16 acc[2][2][2];                                       // [m tile][row][column class]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][c][r] = 0.f;

  f32x4 areg[A4], breg[B4];
  float hreg = 0.f;
  const int nchunks = p.Cx / TS_CC;
  const int wm = WAVES_M == 2 ? (wave >> 1) : 0, wn = WAVES_M == 2 ? (wave & 1) : wave;
  const int l31 = lane & 31, half = lane >> 5;
  const float *Ap = As + half * BM + wm * 64 + l31;
  const float *Bp = Bs + half * CHS + wn * 2 * 34 + l31;

  for (int ch = -1; ch < nchunks; ++ch) {
    if (ch >= 0) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < A4; ++i) *reinterpret_cast<f32x4 *>(As + (tid + 256 * i) * 4) = areg[i];
#pragma unroll
      for (int i = 0; i < B4; ++i)
        if (i < B4 - 1 || wave < B_LASTW) {
          float *d = Bs + b_lds[i];
          d[0] = breg[i][0];
          d[1] = breg[i][1];
          d[2] = breg[i][2];
          d[3] = breg[i][3];
        }
      if (wave < H_WAVES) Bs[h_lds] = hreg;
      __syncthreads();
    }
    if (ch + 1 < nchunks) {
      // packed weights: rows [chunk of 16 channels][tap 0..8][16 channels]; this class uses taps 3..5 (a = 0) or
      // 0..2 and 6..8 (a = 1).  Round i of a thread covers local rows (256 i + tid) / (BM / 4): the tap is uniform per round.
      const int ao = (ch + 1) * a_chunk, bo = (ch + 1) * b_chunk;
#pragma unroll
      for (int i = 0; i < A4; ++i) {
        const int lrow0 = i * (1024 / BM);                     // first local row of the round (8 i or 16 i)
        const int lt = lrow0 / TS_CC;
        const int grow0 = (APAR ? (lt < 3 ? lt : lt + 3) : lt + 3) * TS_CC + (lrow0 - lt * TS_CC);
        areg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, a_vo, ao + grow0 * mp4, 0));
      }
#pragma unroll
      for (int i = 0; i < B4; ++i)
        if (i < B4 - 1 || wave < B_LASTW)
          breg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, b_vo[i], bo, 0));
      if (wave < H_WAVES) hreg = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, h_vo, bo, 0));
    }
    if (ch >= 0 && BF16) {
      // bf16 MFMA mode: K = 16 = the chunk's 16 channels of one tap (lanes 0-31: channels 0-7, lanes 32-63: 8-15),
      // gathered from the same f32 LDS tiles and rounded to bf16 in registers
      const float *A0 = As + 8 * half * BM + wm * 64 + l31;
      const float *B0 = Bs + 8 * half * CHS + wn * 2 * 34 + l31;
#pragma unroll
      for (int lt = 0; lt < NT; ++lt) {
        const int s = lt % 3;
        const int dh = (APAR && lt < 3) ? 1 : 0;
        const int dw = s == 0 ? 1 : 0, cls = s == 1 ? 0 : 1;
        bf16x8 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) af[i][e] = (__bf16)A0[(lt * TS_CC + e) * BM + i * 32];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) bf[j][e] = (__bf16)B0[e * CHS + (j + dh) * 34 + dw];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j][cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j][cls], 0, 0, 0);
      }
    }
    if (ch >= 0 && !BF16) {
#pragma unroll 1
      for (int l3 = 0; l3 < NT; l3 += 3)                     // the row-tap groups stay rolled (register pressure)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int lt = l3 + s;
        const int dh = (APAR && lt < 3) ? 1 : 0;             // a = 1: r = 0 reads the next input row
        const int dw = s == 0 ? 1 : 0, cls = s == 1 ? 0 : 1;
#pragma unroll
        for (int cp = 0; cp < TS_CC / 2; ++cp) {
          float a[2], b[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) a[i] = Ap[(lt * TS_CC + 2 * cp) * BM + i * 32];
#pragma unroll
          for (int j = 0; j < 2; ++j) b[j] = Bp[2 * cp * CHS + (j + dh) * 34 + dw];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j][cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j][cls], 0, 0, 0);
        }
      }
    }
  }

  const int Wb = 2 * p.Ws;
  const long HWb = 4L * HWs;
  // low-VALU epilogue as in igemm_f3x3s2_kernel: bias once per register, max-form LeakyReLU, descriptor stores (float2 = the
  // two column classes of one output pixel pair)
  const bool lrelu01 = p.act == LSPS_ACT_LRELU && p.slope >= 0.f && p.slope <= 1.f;
  if ((p.M & 7) == 0 && (p.act == LSPS_ACT_NONE || lrelu01) && (long)p.M * HWb * 4 < (1L << 31)) {
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.Y + (long)n * p.M * HWb, 0, p.M * (int)HWb * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.bias ? p.bias : p.Y), 0,
                                                                       p.bias ? p.M * 4 : 0, 0x00020000);
    const int hw4 = (int)HWb * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mu = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2);
        if (mu >= p.M) continue;
        const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (unsigned)(half * 16), mu * 4, 0));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float v0 = acc[i][j][0][r] + bv, v1 = acc[i][j][1][r] + bv;
          if (lrelu01) {
            v0 = fmaxf(v0, v0 * p.slope);
            v1 = fmaxf(v1, v1 * p.slope);
          }
          const int orow = 2 * (p0 + wn * 2 + j) + APAR;
          const unsigned vo = (unsigned)((orow * Wb + 2 * (q0 + l31)) * 4 + half * 4 * hw4);
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1)}, yrs, vo,
                                                mu * hw4, 0);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int orow = 2 * (p0 + wn * 2 + j) + APAR;
    float *yb = p.Y + (long)n * p.M * HWb + (long)orow * Wb + 2 * (q0 + l31);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < p.M) {
          float v0 = acc[i][j][0][r], v1 = acc[i][j][1][r];
          if (p.bias) {
            const float bv = p.bias[m];
            v0 += bv;
            v1 += bv;
          }
          f32x2 o;
          o[0] = apply_act(v0, p.act, p.slope);
          o[1] = apply_act(v1, p.act, p.slope);
          *reinterpret_cast<f32x2 *>(yb + (long)m * HWb) = o;
        }
      }
    }
  }
}

template <int BM, bool BF16 = false>
__global__ __launch_bounds__(256, 2) void igemm_t3x3s2_kernel(TS2Params p) {
  __shared__ __attribute__((aligned(16))) float lds[TS_LDS_FLOATS + 4];    // + the spare slot of the halo store
  if (blockIdx.z == 0)
    ts2_body<0, BM, BF16>(p, lds);
  else
    ts2_body<1, BM, BF16>(p, lds);
}

// -------------------------------------------------------------------------------------------
// bf16 MFMA variant of the transposed-direction stride-2 kernel (math mode 1) with K-contiguous operands, like
// igemm_f3x3_bf16_kernel: weights packed once per call by pack_bf16_kernel ([128-row m tile][16-channel chunk][9 taps]
// [128 m][16 c]), input rows rounded to bf16 when staged and stored [row][column][16 c]; one ds_read_b128 per operand
// fragment, one MFMA (K = the chunk's 16 channels) per tap and tile pair.  Same tiling / classes / epilogue as
// ts2_body.  (The f32 body's in-register conversion ran at 185 TFLOP/s.)
// -------------------------------------------------------------------------------------------
struct TS2BParams {
  const float *X, *bias, *zero;
  const unsigned short *Wq;
  float *Y;
  int Cx, Hs, Ws, M;
  int qblocks, tiles_per_img;
  int act;
  float slope;
};

template <int APAR, int BM>
__device__ __forceinline__ void ts2_bf16_body(const TS2BParams &p, unsigned short *lds) {
  constexpr int NT = APAR ? 6 : 3;
  constexpr int WAVES_M = BM / 64, WAVES_N = 4 / WAVES_M, TR = 2 * WAVES_N;
  constexpr int ROWS = TR + 1;
  constexpr int AEL = NT * BM * FB_CC;                       // bf16 elements of this class's weight taps
  constexpr int A16 = (AEL / 8 + 255) / 256;                 // 16-byte units per thread per chunk (6 / 3 / 3 / 2)
  constexpr int NPIX = ROWS * 33;                            // staged pixels (33 columns: the right neighbour too)
  constexpr int BP = (NPIX + 255) / 256;
  unsigned short *Aq = lds, *Bq = lds + AEL;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = blockIdx.y;                                 // BM-row tile; lives in 128-row tile mt*BM/128 of the pack
  const int m0 = mt * BM;
  const int n = blockIdx.x / p.tiles_per_img;
  const int rem = blockIdx.x - n * p.tiles_per_img;
  const int p0 = (rem / p.qblocks) * TR, q0 = (rem - (rem / p.qblocks) * p.qblocks) * 32;
  const int HWs = p.Hs * p.Ws;
  const float *xn = p.X + (long)n * p.Cx * HWs;
  const int nchunks = p.Cx / FB_CC;

  bool b_use[BP], b_ok[BP];
  int b_lds[BP], b_off[BP];
#pragma unroll
  for (int q = 0; q < BP; ++q) {
    const int u = tid + 256 * q;
    b_use[q] = u < NPIX;
    const int r = u / 33, col = u - r * 33;
    b_ok[q] = b_use[q] && (p0 + r) < p.Hs && (q0 + col) < p.Ws;
    b_lds[q] = (r * 34 + col) * FB_CC;
    b_off[q] = (p0 + r) * p.Ws + q0 + col;
  }

  f32x16 acc[2][2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][c][r] = 0.f;

  f32x4 areg[A16];
  float breg[BP][FB_CC];
  const int wm = WAVES_M == 2 ? (wave >> 1) : 0, wn = WAVES_M == 2 ? (wave & 1) : wave;
  const int l31 = lane & 31, half = lane >> 5;
  // this class's taps inside a packed chunk: taps 3..5 (a = 0) or 0..2 and 6..8 (a = 1); rows m0 % 128 .. + BM
  const unsigned short *wq = p.Wq + (long)(m0 / 128) * nchunks * FB_ACH + (long)(m0 % 128) * FB_CC;

  for (int ch = -1; ch < nchunks; ++ch) {
    if (ch >= 0) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < A16; ++i)
        if ((tid + 256 * i) * 8 < AEL) *reinterpret_cast<f32x4 *>(Aq + (tid + 256 * i) * 8) = areg[i];
#pragma unroll
      for (int q = 0; q < BP; ++q)
        if (b_use[q]) {
#pragma unroll
          for (int h8 = 0; h8 < 2; ++h8) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)breg[q][h8 * 8 + e];
            *reinterpret_cast<bf16x8 *>(Bq + b_lds[q] + h8 * 8) = v;
          }
        }
      __syncthreads();
    }
    if (ch + 1 < nchunks) {
      const unsigned short *src = wq + (long)(ch + 1) * FB_ACH;
#pragma unroll
      for (int i = 0; i < A16; ++i) {
        int u8 = (tid + 256 * i) * 8;                        // element index inside [NT][BM][16]
        if (u8 >= AEL) u8 = 0;                               // (odd thread counts: the surplus lanes re-read element 0)
        const int lt = u8 / (BM * FB_CC), rest = u8 - lt * (BM * FB_CC);
        const int t = APAR ? (lt < 3 ? lt : lt + 3) : lt + 3;
        areg[i] = *reinterpret_cast<const f32x4 *>(src + (long)t * 128 * FB_CC + rest);
      }
      const float *xc = xn + (long)(ch + 1) * FB_CC * HWs;
#pragma unroll
      for (int q = 0; q < BP; ++q)
#pragma unroll
        for (int e = 0; e < FB_CC; ++e) {
          const float *s2 = b_ok[q] ? (xc + (long)e * HWs + b_off[q]) : p.zero;
          breg[q][e] = *s2;
        }
    }
    if (ch >= 0) {
#pragma unroll 1
      for (int l3 = 0; l3 < NT; l3 += 3) {                   // row tap groups stay rolled: few live operand fragments
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int lt = l3 + s;
        const int dh = (APAR && lt < 3) ? 1 : 0;
        const int dw = s == 0 ? 1 : 0, cls = s == 1 ? 0 : 1;
        bf16x8 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          af[i] = *reinterpret_cast<const bf16x8 *>(Aq + ((lt * BM) + wm * 64 + i * 32 + l31) * FB_CC + 8 * half);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          bf[j] = *reinterpret_cast<const bf16x8 *>(Bq + ((wn * 2 + j + dh) * 34 + l31 + dw) * FB_CC + 8 * half);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j][cls] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j][cls], 0, 0, 0);
      }
      }
    }
  }

  const int Wb = 2 * p.Ws;
  const long HWb = 4L * HWs;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int orow = 2 * (p0 + wn * 2 + j) + APAR;
    float *yb = p.Y + (long)n * p.M * HWb + (long)orow * Wb + 2 * (q0 + l31);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < p.M) {
          const float bv = p.bias ? p.bias[m] : 0.f;
          f32x2 o;
          o[0] = apply_act(acc[i][j][0][r] + bv, p.act, p.slope);
          o[1] = apply_act(acc[i][j][1][r] + bv, p.act, p.slope);
          *reinterpret_cast<f32x2 *>(yb + (long)m * HWb) = o;
        }
      }
    }
  }
}

template <int BM>
__global__ __launch_bounds__(256, 2) void igemm_t3x3s2_bf16_kernel(TS2BParams p) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[6 * BM * FB_CC + (BM == 128 ? 5 : 9) * 34 * FB_CC];
  if (blockIdx.z == 0)
    ts2_bf16_body<0, BM>(p, lds);
  else
    ts2_bf16_body<1, BM>(p, lds);
}

// -------------------------------------------------------------------------------------------
// bf16 MFMA variant of the forward-direction stride-2 kernel (math mode 1), K-contiguous operands as in
// igemm_f3x3_bf16_kernel: packed bf16 weights [m tile][16-channel chunk][9 taps][128 m][16 c]; the 9 input rows of the
// tile are rounded to bf16 when staged and stored [row][O'(33) | E(32)][16 c] (column-parity de-interleaved like the f32
// kernel), one ds_read_b128 per fragment, one MFMA per tap and tile pair.
// -------------------------------------------------------------------------------------------
#define FS2B_ROW 66                                    // pixels per staged row: O'[0..32], E[33..64], one pad

__global__ __launch_bounds__(256, 2) void igemm_f3x3s2_bf16_kernel(TS2BParams p) {
  // TS2BParams reused: Hs / Ws are the OUTPUT (small) image here, the input is [2 Hs][2 Ws]
  constexpr int A16 = FB_ACH / 8 / 256;                              // 9
  constexpr int NPIX = FS2_ROWS * 65;                                // 585 staged pixels
  constexpr int BP = (NPIX + 255) / 256;                             // 3
  __shared__ __attribute__((aligned(16))) unsigned short lds[FB_ACH + FS2_ROWS * FS2B_ROW * FB_CC];
  unsigned short *Aq = lds, *Bq = lds + FB_ACH;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = blockIdx.y, m0 = mt * 128;
  const int n = blockIdx.x / p.tiles_per_img;
  const int rem = blockIdx.x - n * p.tiles_per_img;
  const int p0 = (rem / p.qblocks) * 4, q0 = (rem - (rem / p.qblocks) * p.qblocks) * 32;
  const int Hx = 2 * p.Hs, Wx = 2 * p.Ws, HWx = Hx * Wx;
  const float *xn = p.X + (long)n * p.Cx * HWx;
  const int nchunks = p.Cx / FB_CC;

  bool b_use[BP], b_ok[BP];
  int b_lds[BP], b_off[BP];
#pragma unroll
  for (int q = 0; q < BP; ++q) {
    const int u = tid + 256 * q;
    b_use[q] = u < NPIX;
    const int r = u / 65, x = u - r * 65;                           // x: input column 2 q0 - 1 + x
    const int ih = 2 * p0 - 1 + r, iw = 2 * q0 - 1 + x;
    b_ok[q] = b_use[q] && ih >= 0 && iw >= 0;                       // ih < 2P, iw < 2Q always
    // odd input columns (x even) -> O'[x / 2], even input columns (x odd) -> E[(x - 1) / 2] at 33 + ...
    b_lds[q] = (r * FS2B_ROW + ((x & 1) ? 33 + (x >> 1) : (x >> 1))) * FB_CC;
    b_off[q] = ih * Wx + iw;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 areg[A16];
  float breg[BP][FB_CC];
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const unsigned short *wq = p.Wq + (long)mt * nchunks * FB_ACH;

  for (int ch = -1; ch < nchunks; ++ch) {
    if (ch >= 0) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < A16; ++i) *reinterpret_cast<f32x4 *>(Aq + (tid + 256 * i) * 8) = areg[i];
#pragma unroll
      for (int q = 0; q < BP; ++q)
        if (b_use[q]) {
#pragma unroll
          for (int h8 = 0; h8 < 2; ++h8) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)breg[q][h8 * 8 + e];
            *reinterpret_cast<bf16x8 *>(Bq + b_lds[q] + h8 * 8) = v;
          }
        }
      __syncthreads();
    }
    if (ch + 1 < nchunks) {
      const unsigned short *src = wq + (long)(ch + 1) * FB_ACH;
#pragma unroll
      for (int i = 0; i < A16; ++i) areg[i] = *reinterpret_cast<const f32x4 *>(src + (tid + 256 * i) * 8);
      const float *xc = xn + (long)(ch + 1) * FB_CC * HWx;
#pragma unroll
      for (int q = 0; q < BP; ++q)
#pragma unroll
        for (int e = 0; e < FB_CC; ++e) {
          const float *s2 = b_ok[q] ? (xc + (long)e * HWx + b_off[q]) : p.zero;
          breg[q][e] = *s2;
        }
    }
    if (ch >= 0) {
#pragma unroll 3
      for (int t = 0; t < 9; ++t) {
        const int tr = t / 3, ts = t - tr * 3;
        const int coff = ts == 1 ? 33 : (ts == 2 ? 1 : 0);
        bf16x8 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(Aq + (t * 128 + wm * 64 + i * 32 + l31) * FB_CC + 8 * half);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          bf[j] = *reinterpret_cast<const bf16x8 *>(Bq + ((2 * (wn * 2 + j) + tr) * FS2B_ROW + coff + l31) * FB_CC + 8 * half);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  const long PQ = (long)p.Hs * p.Ws;
  float *y0 = p.Y + (long)n * p.M * PQ + (long)(p0 + wn * 2) * p.Ws + q0 + l31;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (m < p.M) {
        const float bv = p.bias ? p.bias[m] : 0.f;
        float *ym = y0 + (long)m * PQ;
#pragma unroll
        for (int j = 0; j < 2; ++j) ym[(long)j * p.Ws] = apply_act(acc[i][j][r] + bv, p.act, p.slope);
      }
    }
  }
}

// -------------------------------------------------------------------------------------------
// W kernel specialised for 3x3 / STRIDE 2 / pad 1 (the down-sampling convs and the up-sampling transposed convs:
// "small" image [N][M][Hs][Ws], "big" image [N][C][2Hs][2Ws], Ws % 32 == 0):
//   dW[m][c][r][s] = sum_{n,p,q} small[n][m][p][q] * big[n][c][2p + r - 1][2q + s - 1]
// Chunk = one row segment of 32 small pixels.  The three big rows it touches are staged ONCE in LDS for 64 big
// channels, DE-INTERLEAVED by column parity (odd columns with their left halo: O'[0..32], even columns: E[0..31]),
// so that the stride-2 tap reads become unit-stride LDS reads: tap s=0 -> O'[q], s=1 -> E[q], s=2 -> O'[q+1].
// A big element serves only ~9/4 taps here (9 in the stride-1 kernel), so the tile is 128 (m) x 64 (c) x 9 taps
// on 512 threads (8 waves, 9 accumulators each) to keep ~70 flop per staged byte.
// -------------------------------------------------------------------------------------------
#define WS2_LDA 33                // small tile [128 m][32 px + 1]
#define WS2_ROW 65                // one big row in LDS: O'[33] then E[32]
#define WS2_CH (3 * WS2_ROW)      // 195 floats per big channel (195 % 32 = 3: conflict-free across 32 channels)
#define WS2_LDS_BYTES ((128 * WS2_LDA + 64 * WS2_CH) * sizeof(float))

struct WS2Params {
  const float *Small, *Big, *zero;
  float *part;                   // [splits][M][C][9]
  int N, M, C, Hs, Ws;
  int qblocks;                   // Ws / 32
  int nchunks, chunks_per_split; // chunk = (n, small row, 32-column block)
};

template <bool BF16>
__global__ __launch_bounds__(512, 2) void igemm_w3x3s2_kernel(WS2Params p) {
  extern __shared__ __attribute__((aligned(16))) float ws2_lds[];
  float *As = ws2_lds, *Bs = ws2_lds + 128 * WS2_LDA;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c0 = blockIdx.x * 64, m0 = blockIdx.y * 128, split = blockIdx.z;
  const int HWs = p.Hs * p.Ws, Wb = 2 * p.Ws;
  const long HWb = 4L * HWs;
  const int ch_begin = split * p.chunks_per_split;
  int ch_end = ch_begin + p.chunks_per_split;
  if (ch_end > p.nchunks) ch_end = p.nchunks;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  f32x4 areg[2], breg[6];
  float hreg = 0.f;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const float *Ap = As + (wm * 32 + l31) * WS2_LDA + half;
  const float *Bp = Bs + (wn * 32 + l31) * WS2_CH + half;
  const int per_img = p.Hs * p.qblocks;

  // Branch-free staging with a uniform chunk cursor (n, y, qb) instead of a division per chunk: the descriptors' bases carry
  // the cursor (64-bit scalar arithmetic), a lane carries ONE 32-bit offset per load.  The big image's row 2y - 1 does not
  // exist for y = 0 and the column 2 q0 - 1 not for q0 = 0: lanes of that row carry bit 31 in their offset (out of range for
  // the descriptor -> 0), cleared by a uniform mask when the row exists.
  constexpr unsigned OOB = 0x80000000u;
  unsigned a_vo[2], b_vo[6];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int u = tid + 512 * i;
    a_vo[i] = (unsigned)((u >> 3) * HWs + (u & 7) * 4) * 4u;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int u = tid + 512 * i;
    const int line = u >> 4, c4 = u & 15;
    const int chn = line / 3, r = line - chn * 3;
    b_vo[i] = ((unsigned)((long)chn * HWb + r * Wb + c4 * 4) * 4u) | (r == 0 ? OOB : 0u);
  }
  unsigned h_vo;
  {
    const int chn = tid / 3, r = tid - chn * 3;
    h_vo = ((unsigned)((long)chn * HWb + r * Wb) * 4u) | (r == 0 ? OOB : 0u);
  }
  int cn = 0, cy = 0, cq = 0;                               // cursor of the chunk being FETCHED
  {
    const int nc = ch_begin;
    cn = nc / per_img;
    const int rem = nc - cn * per_img;
    cy = rem / p.qblocks;
    cq = rem - cy * p.qblocks;
  }

  for (int ch = ch_begin - 1; ch < ch_end; ++ch) {
    if (ch >= ch_begin) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int u = tid + 512 * i;
        float *d = As + (u >> 3) * WS2_LDA + (u & 7) * 4;
        d[0] = areg[i][0];
        d[1] = areg[i][1];
        d[2] = areg[i][2];
        d[3] = areg[i][3];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int u = tid + 512 * i;
        const int line = u >> 4, c4 = u & 15;            // line = channel * 3 + row
        float *d = Bs + line * WS2_ROW + 2 * c4;         // (line * 65 == ch * 195 + row * 65)
        d[33] = breg[i][0];                              // E[2 c4]
        d[1] = breg[i][1];                               // O'[2 c4 + 1]
        d[34] = breg[i][2];                              // E[2 c4 + 1]
        d[2] = breg[i][3];                               // O'[2 c4 + 2]
      }
      if (wave < 3) Bs[tid * WS2_ROW] = hreg;            // O'[0]: the column left of the block (zero at the image edge)
      __syncthreads();
    }
    if (ch + 1 < ch_end) {
      const int q0 = cq * 32;
      const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float *>(p.Small + ((long)cn * p.M + m0) * HWs + cy * p.Ws), 0, 0x7fffffff, 0x00020000);
      // base = row 2y - 1 of channel c0 (for y = 0 one row in front of the channel: only the masked lanes would touch it)
      const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float *>(p.Big + ((long)cn * p.C + c0) * HWb + (long)(2 * cy - 1) * Wb), 0, 0x7fffffff, 0x00020000);
      const unsigned rmask = cy > 0 ? 0x7fffffffu : 0xffffffffu;                // uniform
#pragma unroll
      for (int i = 0; i < 2; ++i)
        areg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, a_vo[i], q0 * 4, 0));
#pragma unroll
      for (int i = 0; i < 6; ++i)
        breg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, b_vo[i] & rmask, 2 * q0 * 4, 0));
      if (wave < 3) {
        const unsigned hm = q0 > 0 ? 0u : OOB;                                  // uniform
        hreg = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (h_vo & rmask) | hm, q0 > 0 ? 2 * q0 * 4 - 4 : 0, 0));
      }
      if (++cq == p.qblocks) {
        cq = 0;
        if (++cy == p.Hs) {
          cy = 0;
          ++cn;
        }
      }
    }
    if (ch >= ch_begin && BF16) {
      // bf16 MFMA mode: K = 16 consecutive small pixels per MFMA (lanes 0-31: pixels 0-7 of the group, lanes 32-63:
      // 8-15), operands rounded to bf16 in registers; 2 groups x 9 taps per chunk
      const float *A0 = As + (wm * 32 + l31) * WS2_LDA + 8 * half;
      const float *B0 = Bs + (wn * 32 + l31) * WS2_CH + 8 * half;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        bf16x8 af;
#pragma unroll
        for (int e = 0; e < 8; ++e) af[e] = (__bf16)A0[g * 16 + e];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int r = t / 3, sx = t - r * 3;
          const int off = r * WS2_ROW + (sx == 1 ? 33 : (sx == 2 ? 1 : 0)) + g * 16;
          bf16x8 bf;
#pragma unroll
          for (int e = 0; e < 8; ++e) bf[e] = (__bf16)B0[off + e];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[t], 0, 0, 0);
        }
      }
    }
    if (ch >= ch_begin && !BF16) {
      // k-step kq covers small pixels 2kq + half; the operands of step kq+1 are fetched from LDS before the nine
      // MFMAs of step kq are issued (the compiler does not pipeline the reads across iterations by itself)
      float a_nx = Ap[0], b_nx[9];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        b_nx[3 * r + 0] = Bp[r * WS2_ROW];
        b_nx[3 * r + 1] = Bp[r * WS2_ROW + 33];
        b_nx[3 * r + 2] = Bp[r * WS2_ROW + 1];
      }
#pragma unroll
      for (int kq = 0; kq < 16; ++kq) {
        const float a = a_nx;
        float b[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) b[t] = b_nx[t];
        if (kq + 1 < 16) {
          a_nx = Ap[2 * kq + 2];
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            b_nx[3 * r + 0] = Bp[r * WS2_ROW + 2 * kq + 2];
            b_nx[3 * r + 1] = Bp[r * WS2_ROW + 33 + 2 * kq + 2];
            b_nx[3 * r + 2] = Bp[r * WS2_ROW + 2 * kq + 3];
          }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
      }
    }
  }

  const int c = c0 + wn * 32 + l31;
  // partials in the weight's own layout [split][m][c][t]: a lane's nine taps are 36 contiguous bytes, a wave row is
  // 1152 contiguous bytes (merged in L2), and the reduction over splits is a plain coalesced sum
  float *out = p.part + (long)split * p.M * p.C * 9 + (long)c * 9;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
    for (int t = 0; t < 9; ++t) out[(long)m * p.C * 9 + t] = acc[t][r];
  }
}

}  // namespace lsps
#endif
