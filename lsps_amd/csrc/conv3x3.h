// 3x3 / stride 1 / pad 1 / width 32 kernels (the residual convs): forward-direction kernel, its bf16 / split-precision variant, weight gradient.
#ifndef LSPS_CONV3X3_H
#define LSPS_CONV3X3_H
#include "conv_types.h"

namespace lsps {

// -------------------------------------------------------------------------------------------
// F kernel specialised for the dominant layer class: 3x3 taps, stride 1, pad 1, image width 32
// (the 28 residual convs = 88 % of the generator's MACs, forward and dgrad).
// Instead of gathering every (c,tap) row of the B tile from global memory (9 loads per input element,
// ~10 VALU instructions each: measured 20 % of the kernel), the raw input rows of CC channels are
// staged ONCE in LDS with their zero halo, and the 9 taps are shifted LDS reads with compile-time
// offsets.  Tile: 128 output channels x 128 pixels (4 full image rows), chunk = CC*9 reduction rows.
// -------------------------------------------------------------------------------------------
#define F3_CC 8
#define F3_LDW 34                 // 32 pixels + left/right halo column (always zero: W == 32, pad == 1)

// XCD-aware workgroup -> tile mapping.  Workgroups are handed to the 8 XCDs round-robin in linear launch order and each
// XCD has its own L2.  `lin` = linear workgroup id; this returns (pixel tile, m tile) such that one XCD walks a
// CONTIGUOUS range of pixel tiles with the m tiles of a pixel tile adjacent in time: the second 128-channel tile re-reads
// the input rows from that XCD's L2 instead of HBM, and so does the 2-row halo shared with the next pixel tile.
__device__ __forceinline__ void xcd_tile(int lin, int NT, int MT, int &tile, int &mt) {
  if ((NT & 7) == 0) {
    const int xcd = lin & 7, q = lin >> 3;
    mt = q % MT;
    tile = xcd * (NT >> 3) + q / MT;
  } else {
    tile = lin % NT;
    mt = lin / NT;
  }
}

struct F3Params {
  const float *X, *Wp, *bias, *zero;
  const float *R;                // optional addend with Y's layout (dgrad of a residual block: + skip gradient)
  float *Y;
  int Cx, H, M, Mp, NT;          // NT = N * (H/TR) pixel tiles
  int tiles_per_img;             // H / TR
  int act;
  float slope;
  // small batches (the estimate modes run the generator on 4-8 samples): the channel reduction is split over
  // blockIdx.z; each split writes raw partial sums to part[z][N][M][H][32], f3x3_ksplit_reduce_kernel finishes
  int ksplit, chunks_per_split;
  float *part;
};

// y = act(sum_z part[z] + bias[m]) + R   (second stage of the reduction split above)
__global__ __launch_bounds__(256) void f3x3_ksplit_reduce_kernel(const float *__restrict__ part, const float *__restrict__ bias,
                                                                 const float *__restrict__ R, float *__restrict__ y,
                                                                 long total4, int ks, int M, int HW, int act, float slope) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  f32x4 s = reinterpret_cast<const f32x4 *>(part)[i];
  for (int z = 1; z < ks; ++z) s += reinterpret_cast<const f32x4 *>(part)[(long)z * total4 + i];
  const int m = (int)((i * 4 / HW) % M);
  const float bv = bias ? bias[m] : 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) s[e] = apply_act(s[e] + bv, act, slope);
  if (R) s += reinterpret_cast<const f32x4 *>(R)[i];
  reinterpret_cast<f32x4 *>(y)[i] = s;
}

// TR = output rows per tile: 4 (tile 128 ch x 128 px, waves 2x2, each 64 ch x 64 px) or, when that grid would
// leave CUs idle (estimate modes run the generator on 8 samples), 2 (128 ch x 64 px, waves 4x1, each 32 ch x 64 px).
// KSPLIT is a template parameter: with a run-time split the chunk loop of the main variant lost 4 % (129 vs 135 TFLOP/s)
template <int TR, bool KSPLIT = false>
__global__ __launch_bounds__(256, 2) void igemm_f3x3_kernel(F3Params p) {
  constexpr int BM = 128, RC = F3_CC * 9;            // 72 reduction rows per chunk
  constexpr int A4 = RC * BM / 4 / 256;              // 9 float4 of weights per thread per chunk
  constexpr int ROWS = TR + 2, CH = ROWS * F3_LDW;   // staged rows incl. halo; floats per channel
  constexpr int B4 = (F3_CC * ROWS * 8 + 255) / 256; // float4 of input per thread per chunk (2 or 1)
  constexpr int WM = TR == 4 ? 2 : 1;                // MFMA row tiles per wave
  __shared__ __attribute__((aligned(16))) float lds[RC * BM + F3_CC * CH];
  float *As = lds, *Bs = lds + RC * BM;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile, mt;
  xcd_tile(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x, gridDim.y, tile, mt);
  const int m0 = mt * BM;
  const int n = tile / p.tiles_per_img;
  const int row0 = (tile - n * p.tiles_per_img) * TR;            // first output row of the tile
  const int HW = p.H * 32;
  const float *xn = p.X + (long)n * p.Cx * HW;

  // zero the halo columns once (never overwritten): cols 0 and 33 of every (channel,row)
  if (tid < F3_CC * ROWS * 2) {
    const int rr = tid >> 1;
    Bs[rr * F3_LDW + (tid & 1) * 33] = 0.f;
  }

  // B staging assignment: F3_CC*ROWS (channel,row) lines of 32 pixels = 8 float4 each
  int b_lds[B4];
  long b_off[B4];
  bool b_use[B4], b_ok[B4];
#pragma unroll
  for (int i = 0; i < B4; ++i) {
    const int u = tid + 256 * i;
    b_use[i] = u < F3_CC * ROWS * 8;
    const int line = u >> 3, c4 = u & 7;
    const int ch = line / ROWS, r = line - ch * ROWS;
    const int img_row = row0 - 1 + r;
    b_ok[i] = b_use[i] && img_row >= 0 && img_row < p.H;
    b_lds[i] = ch * CH + r * F3_LDW + 1 + c4 * 4;
    b_off[i] = (long)ch * HW + (long)img_row * 32 + c4 * 4;
  }

  f32x16 acc[WM][2];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 areg[A4], breg[B4];
  const int nchunks_all = p.Cx / F3_CC;
  const int ch0 = KSPLIT ? blockIdx.z * p.chunks_per_split : 0;
  const int nchunks = KSPLIT ? min(nchunks_all, ch0 + p.chunks_per_split) : nchunks_all;   // end of this split
  const int wm = TR == 4 ? (wave >> 1) : wave, wn = TR == 4 ? (wave & 1) : 0;
  const int l31 = lane & 31, half = lane >> 5;
  const float *Ap = As + half * BM + wm * WM * 32 + l31;
  const float *Bp = Bs + half * CH + wn * 2 * F3_LDW + l31;

  for (int ch = ch0 - 1; ch < nchunks; ++ch) {
    if (ch >= ch0) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < A4; ++i) {
        const int u = tid + 256 * i;
        *reinterpret_cast<f32x4 *>(As + u * 4) = areg[i];          // tile rows are contiguous: [72][128]
      }
#pragma unroll
      for (int i = 0; i < B4; ++i)
        if (b_use[i]) {
          float *d = Bs + b_lds[i];
          d[0] = breg[i][0];
          d[1] = breg[i][1];
          d[2] = breg[i][2];
          d[3] = breg[i][3];
        }
      __syncthreads();
    }
    if (ch + 1 < nchunks) {
      const float *wsrc = p.Wp + (long)(ch + 1) * RC * p.Mp + m0;
#pragma unroll
      for (int i = 0; i < A4; ++i) {
        const int u = tid + 256 * i;
        const int row = u >> 5, c4 = u & 31;
        areg[i] = *reinterpret_cast<const f32x4 *>(wsrc + (long)row * p.Mp + c4 * 4);
      }
      const float *xc = xn + (long)(ch + 1) * F3_CC * HW;
#pragma unroll
      for (int i = 0; i < B4; ++i) {
        const float *src = b_ok[i] ? (xc + b_off[i]) : p.zero;     // masked rows read the zero slot
        breg[i] = *reinterpret_cast<const f32x4 *>(src);
      }
    }
    if (ch >= ch0) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int tr = t / 3, ts = t - tr * 3;
#pragma unroll
        for (int cp = 0; cp < F3_CC / 2; ++cp) {                   // channel pair (2cp, 2cp+1): k = half
          const int kk = t * (F3_CC / 2) + cp;                     // k-step: reduction rows 2kk, 2kk+1
          float a[WM], b[2];
#pragma unroll
          for (int i = 0; i < WM; ++i) a[i] = Ap[2 * kk * BM + i * 32];
#pragma unroll
          for (int j = 0; j < 2; ++j) b[j] = Bp[2 * cp * CH + (j + tr) * F3_LDW + ts];
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  }

  // epilogue: wave's pixel rows wn*2 + j, column l31; channel-major so that a channel's bias is fetched once
  float *y0 = p.Y + (long)n * p.M * HW + (long)(row0 + wn * 2) * 32 + l31;
#pragma unroll
  for (int i = 0; i < WM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (wm * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (m < p.M) {
        float *ym = y0 + (long)m * HW;
        if (KSPLIT) {                                            // raw partial sums of this reduction split
          float *pm = p.part + (long)blockIdx.z * p.NT * (long)(TR * 32) * p.M + (ym - p.Y);
#pragma unroll
          for (int j = 0; j < 2; ++j) pm[j * 32] = acc[i][j][r];
          continue;
        }
        const float bv = p.bias ? p.bias[m] : 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float v = apply_act(acc[i][j][r] + bv, p.act, p.slope);
          if (p.R) v += p.R[(ym - p.Y) + j * 32];
          ym[j * 32] = v;
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------
// Split-precision variant of the 3x3 / stride-1 / width-32 kernel (math mode 2, experimental): every f32 operand
// is split ONCE into three bf16 limbs (hi, mid, lo: 8+8+8 significand bits, exact) — the weights by the pack
// kernel, the input rows when they are staged into LDS — and stored K-contiguous ([..][8 channels] bf16), so that
// an MFMA operand fragment is ONE ds_read_b128 per limb.  A product is six v_mfma_f32_32x32x16_bf16 (32 cycles
// each, K = 16 = 2 taps x 8 channels) against eight v_mfma_f32_32x32x2_f32 (64 cycles each) for the same 16
// reduction elements: 192 vs 512 matrix-pipe cycles at f32-class accuracy (dropped limb products < 2^-24).
// -------------------------------------------------------------------------------------------
#define FS_TAPS 10                                   // 9 taps + one all-zero tap so that taps pair up
#define FS_APLANE (FS_TAPS * 128 * 8)                // bf16 elements per limb plane of the weight tile
#define FS_ACHUNK (3 * FS_APLANE)                    // bf16 elements per (m-tile, channel chunk)

struct FSPack {
  const float *W;
  unsigned short *Wq;            // [Mp/128][C/8][np limbs][10 taps][128 m][8 c] bf16
  int M, C, np;                  // np = 3 (f32 split: hi, mid, lo) or 1 (bf16 mode: hi only)
  long sm, sc;
  int tapidx[9];
};

__device__ __forceinline__ void split3_scalar(float x, unsigned short &h, unsigned short &m, unsigned short &l) {
  const __bf16 bh = (__bf16)x;
  const float r1 = x - (float)bh;
  const __bf16 bm = (__bf16)r1;
  const float r2 = r1 - (float)bm;
  const __bf16 bl = (__bf16)r2;
  h = __builtin_bit_cast(unsigned short, bh);
  m = __builtin_bit_cast(unsigned short, bm);
  l = __builtin_bit_cast(unsigned short, bl);
}

__global__ __launch_bounds__(256) void pack_split_kernel(FSPack p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;      // over [mtile][chunk][tap][m_local][c]
  const int chunks = p.C / 8, mtiles = (p.M + 127) / 128;
  const long total = (long)mtiles * chunks * FS_TAPS * 128 * 8;
  if (idx >= total) return;
  const int c = (int)(idx & 7);
  const int ml = (int)((idx >> 3) & 127);
  long rest = idx >> 10;
  const int t = (int)(rest % FS_TAPS);
  rest /= FS_TAPS;
  const int chunk = (int)(rest % chunks), mt = (int)(rest / chunks);
  const int m = mt * 128 + ml;
  float v = 0.f;
  if (t < 9 && m < p.M) v = p.W[(long)m * p.sm + (long)(chunk * 8 + c) * p.sc + p.tapidx[t]];
  unsigned short h, mm, l;
  split3_scalar(v, h, mm, l);
  unsigned short *base = p.Wq + ((long)mt * chunks + chunk) * (p.np * FS_APLANE) + ((long)t * 128 + ml) * 8 + c;
  base[0] = h;
  if (p.np == 3) {
    base[FS_APLANE] = mm;
    base[2 * FS_APLANE] = l;
  }
}

struct FSParams {
  const float *X, *bias, *zero;
  const float *R;                // optional addend with Y's layout
  const unsigned short *Wq;
  float *Y;
  int Cx, H, M, tiles_per_img;
  int act;
  float slope;
};

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// NP = 3: f32 split (six MFMAs per tile pair); NP = 1: plain bf16 mode (operands rounded once at staging / packing)
template <int TR, int NP>
__global__ __launch_bounds__(256, 2) void igemm_f3x3_split_kernel(FSParams p) {
  constexpr int ROWS = TR + 2, BPL = ROWS * F3_LDW * 8;      // bf16 elements per limb plane of the input tile
  constexpr int ACH = NP * FS_APLANE;                        // bf16 elements of weights per (m-tile, channel chunk)
  constexpr int A16 = ACH / 8 / 256;                         // 16-byte units of weights per thread per chunk (5 per limb)
  constexpr int WM = TR >= 4 ? 2 : 1;                        // MFMA row tiles per wave
  constexpr int JN = TR == 8 ? 4 : 2;                        // image rows per wave (TR = 8: 128 ch x 256 px tile,
                                                             // twice the weight reuse: the bf16 mode is L2-bound)
  constexpr int BP = (ROWS * 32 + 255) / 256;                // staged pixels per thread (TR = 8: 320 pixels -> 2)
  __shared__ __attribute__((aligned(16))) unsigned short lds[ACH + NP * BPL];
  unsigned short *Aq = lds, *Bq = lds + ACH;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile, mt;
  xcd_tile(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x, gridDim.y, tile, mt);
  const int m0 = mt * 128;
  const int n = tile / p.tiles_per_img;
  const int row0 = (tile - n * p.tiles_per_img) * TR;
  const int HW = p.H * 32;
  const float *xn = p.X + (long)n * p.Cx * HW;
  const int nchunks = p.Cx / 8;

  // halo columns (0 and 33) of every row, all 8 channels, all 3 limb planes: zero once
  for (int u = tid; u < NP * ROWS * 2 * 8; u += 256) {
    const int c = u & 7, side = (u >> 3) & 1, r = (u >> 4) % ROWS, pl = (u >> 4) / ROWS;
    Bq[pl * BPL + (r * F3_LDW + side * 33) * 8 + c] = 0;
  }

  // B staging: thread (r, col) owns ONE pixel of the staged rows and gathers its 8 channels (8 coalesced dword
  // loads: lanes = consecutive pixels), so that after the split each limb is ONE 16-byte LDS store
  bool b_use[BP], b_ok[BP];
  int b_lds[BP];
  long b_off[BP];
#pragma unroll
  for (int q = 0; q < BP; ++q) {
    const int u = tid + 256 * q;
    b_use[q] = u < ROWS * 32;
    const int b_r = u >> 5, b_col = u & 31;
    const int b_img_row = row0 - 1 + b_r;
    b_ok[q] = b_use[q] && b_img_row >= 0 && b_img_row < p.H;
    b_lds[q] = (b_r * F3_LDW + 1 + b_col) * 8;
    b_off[q] = (long)b_img_row * 32 + b_col;
  }

  f32x16 acc[WM][JN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 areg[A16];
  float breg[BP][8];
  const int wm = TR >= 4 ? (wave >> 1) : wave, wn = TR >= 4 ? (wave & 1) : 0;
  const int l31 = lane & 31, half = lane >> 5;
  const unsigned short *wq = p.Wq + (long)mt * nchunks * ACH;

  for (int ch = -1; ch < nchunks; ++ch) {
    if (ch >= 0) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < A16; ++i) *reinterpret_cast<f32x4 *>(Aq + (tid + 256 * i) * 8) = areg[i];
#pragma unroll
      for (int q = 0; q < BP; ++q)
        if (b_use[q]) {
          u16x8 h8, m8, l8;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            unsigned short h, m, l;
            split3_scalar(breg[q][e], h, m, l);
            h8[e] = h;
            m8[e] = m;
            l8[e] = l;
          }
          *reinterpret_cast<u16x8 *>(Bq + b_lds[q]) = h8;
          if (NP == 3) {
            *reinterpret_cast<u16x8 *>(Bq + BPL + b_lds[q]) = m8;
            *reinterpret_cast<u16x8 *>(Bq + 2 * BPL + b_lds[q]) = l8;
          }
        }
      __syncthreads();
    }
    if (ch + 1 < nchunks) {
#ifdef LSPS_ABL_SPLIT_NOA
      const unsigned short *src = wq;            // ablation: always the first chunk (L1/L2-resident)
#else
      const unsigned short *src = wq + (long)(ch + 1) * ACH;
#endif
#pragma unroll
      for (int i = 0; i < A16; ++i) areg[i] = *reinterpret_cast<const f32x4 *>(src + (tid + 256 * i) * 8);
      const float *xc = xn + (long)(ch + 1) * F3_CC * HW;
#pragma unroll
      for (int q = 0; q < BP; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float *s2 = b_ok[q] ? (xc + (long)e * HW + b_off[q]) : p.zero;
          breg[q][e] = *s2;
        }
    }
    if (ch >= 0) {
#pragma unroll
      for (int g = 0; g < 5; ++g) {
        const int t0 = 2 * g, t1 = 2 * g + 1;                // t1 == 9: the all-zero tap
        const int tb1 = t1 <= 8 ? t1 : 8;
        const int arow = (half ? t1 : t0) * 128 + wm * WM * 32 + l31;
        const int boff = (half ? (tb1 / 3) * F3_LDW + (tb1 % 3) : (t0 / 3) * F3_LDW + (t0 % 3)) + wn * JN * F3_LDW + l31;
        bf16x8 af[NP][WM], bf[NP][JN];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
          for (int i = 0; i < WM; ++i)
            af[pl][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8 *>(Aq + pl * FS_APLANE + (arow + i * 32) * 8));
#pragma unroll
          for (int j = 0; j < JN; ++j)
            bf[pl][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u16x8 *>(Bq + pl * BPL + (boff + j * F3_LDW) * 8));
        }
        // six limb products, smallest first; the tile loop is INSIDE so that consecutive MFMAs hit different
        // accumulators (no dependent-accumulator stall)
        constexpr int NT = NP == 3 ? 6 : 1;
        constexpr int TA[6] = {NP == 3 ? 2 : 0, 0, 1, 1, 0, 0}, TB[6] = {0, NP == 3 ? 2 : 0, 1, 0, 1, 0};
#pragma unroll
        for (int term = 0; term < NT; ++term)
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < JN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[TA[term]][i], bf[TB[term]][j], acc[i][j], 0, 0, 0);
      }
    }
  }

#pragma unroll
  for (int j = 0; j < JN; ++j) {
    float *yb = p.Y + (long)n * p.M * HW + (long)(row0 + wn * JN + j) * 32 + l31;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < p.M) {
          float v = acc[i][j][r];
          if (p.bias) v += p.bias[m];
          v = apply_act(v, p.act, p.slope);
          if (p.R) v += p.R[(yb - p.Y) + (long)m * HW];
          yb[(long)m * HW] = v;
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------
// bf16 MFMA kernel of the 3x3 / stride-1 / width-32 layers (math mode 1), K = 16 CHANNELS of one tap per MFMA:
// weights packed once per call as [m tile][16-channel chunk][9 taps][128 m][16 c] bf16, input rows rounded to bf16 when
// staged and stored [row][column][16 c], so every operand fragment is one ds_read_b128 (lanes 0-31: channels 0-7,
// lanes 32-63: channels 8-15) and a tap costs exactly one MFMA per tile pair (the 8-channel layout of the split kernel
// pairs taps and wastes the tenth half-tap).  16 channels per chunk also halves the barriers per reduction element.
// -------------------------------------------------------------------------------------------
#define FB_CC 16
#define FB_ACH (9 * 128 * FB_CC)                     // bf16 elements of weights per (m tile, chunk): 36 KB

__global__ __launch_bounds__(256) void pack_bf16_kernel(FSPack p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;      // over [mtile][chunk][tap][m_local][c]
  const int chunks = p.C / FB_CC, mtiles = (p.M + 127) / 128;
  const long total = (long)mtiles * chunks * FB_ACH;
  if (idx >= total) return;
  const int c = (int)(idx & 15);
  const int ml = (int)((idx >> 4) & 127);
  long rest = idx >> 11;
  const int t = (int)(rest % 9);
  rest /= 9;
  const int chunk = (int)(rest % chunks), mt = (int)(rest / chunks);
  const int m = mt * 128 + ml;
  float v = 0.f;
  if (m < p.M) v = p.W[(long)m * p.sm + (long)(chunk * FB_CC + c) * p.sc + p.tapidx[t]];
  p.Wq[idx] = __builtin_bit_cast(unsigned short, (__bf16)v);
}

// TR = 4: 256 threads (waves 2 x 2);  TR = 8: 512 threads (waves 2 x 4, the weight tile is reused by twice the pixels)
template <int TR, int NTHR>
__global__ __launch_bounds__(NTHR, 2) void igemm_f3x3_bf16_kernel(FSParams p) {
  constexpr int ROWS = TR + 2;
  constexpr int BEL = ROWS * F3_LDW * FB_CC;                 // bf16 elements of the staged input rows
  constexpr int A16 = (FB_ACH / 8 + NTHR - 1) / NTHR;        // 16-byte units of weights per thread per chunk (9 or 5)
  constexpr int WAVES_N = NTHR / 128;                        // waves along the pixel rows (2 or 4); 2 along the channels
  constexpr int JN = TR / WAVES_N;                           // image rows per wave
  constexpr int BP = (ROWS * 32 + NTHR - 1) / NTHR;          // staged pixels per thread
  __shared__ __attribute__((aligned(16))) unsigned short lds[FB_ACH + BEL];
  unsigned short *Aq = lds, *Bq = lds + FB_ACH;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile, mt;
  xcd_tile(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x, gridDim.y, tile, mt);
  const int m0 = mt * 128;
  const int n = tile / p.tiles_per_img;
  const int row0 = (tile - n * p.tiles_per_img) * TR;
  const int HW = p.H * 32;
  const float *xn = p.X + (long)n * p.Cx * HW;
  const int nchunks = p.Cx / FB_CC;

  // halo columns 0 and 33 of every staged row: zero once
  for (int u = tid; u < ROWS * 2 * FB_CC; u += NTHR) {
    const int c = u & 15, side = (u >> 4) & 1, r = u >> 5;
    Bq[(r * F3_LDW + side * 33) * FB_CC + c] = 0;
  }

  bool b_use[BP], b_ok[BP];
  int b_lds[BP];
  long b_off[BP];
#pragma unroll
  for (int q = 0; q < BP; ++q) {
    const int u = tid + NTHR * q;
    b_use[q] = u < ROWS * 32;
    const int b_r = u >> 5, b_col = u & 31;
    const int img_row = row0 - 1 + b_r;
    b_ok[q] = b_use[q] && img_row >= 0 && img_row < p.H;
    b_lds[q] = (b_r * F3_LDW + 1 + b_col) * FB_CC;
    b_off[q] = (long)img_row * 32 + b_col;
  }

  f32x16 acc[2][JN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 areg[A16];
  float breg[BP][FB_CC];
  const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
  const int l31 = lane & 31, half = lane >> 5;
  const unsigned short *wq = p.Wq + (long)mt * nchunks * FB_ACH;

  for (int ch = -1; ch < nchunks; ++ch) {
    if (ch >= 0) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < A16; ++i)
        if ((tid + NTHR * i) * 8 < FB_ACH) *reinterpret_cast<f32x4 *>(Aq + (tid + NTHR * i) * 8) = areg[i];
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
        const int u8 = (tid + NTHR * i) * 8;
        areg[i] = *reinterpret_cast<const f32x4 *>(src + (u8 < FB_ACH ? u8 : 0));
      }
      const float *xc = xn + (long)(ch + 1) * FB_CC * HW;
#pragma unroll
      for (int q = 0; q < BP; ++q)
#pragma unroll
        for (int e = 0; e < FB_CC; ++e) {
          const float *s2 = b_ok[q] ? (xc + (long)e * HW + b_off[q]) : p.zero;
          breg[q][e] = *s2;
        }
    }
    if (ch >= 0) {
#pragma unroll 3
      for (int t = 0; t < 9; ++t) {                          // partial unroll: keeps the live operand fragments few
        const int arow = t * 128 + wm * 64 + l31;
        const int boff = (t / 3) * F3_LDW + (t % 3) + wn * JN * F3_LDW + l31;
        bf16x8 af[2], bf[JN];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(Aq + (arow + i * 32) * FB_CC + 8 * half);
#pragma unroll
        for (int j = 0; j < JN; ++j) bf[j] = *reinterpret_cast<const bf16x8 *>(Bq + (boff + j * F3_LDW) * FB_CC + 8 * half);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < JN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  // epilogue: channel-major so that a channel's bias is fetched once for all of the wave's rows
  float *y0 = p.Y + (long)n * p.M * HW + (long)(row0 + wn * JN) * 32 + l31;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (m < p.M) {
        const float bv = p.bias ? p.bias[m] : 0.f;
        float *ym = y0 + (long)m * HW;
#pragma unroll
        for (int j = 0; j < JN; ++j) {
          float v = apply_act(acc[i][j][r] + bv, p.act, p.slope);
          if (p.R) v += p.R[(ym - p.Y) + j * 32];
          ym[j * 32] = v;
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------
// W kernel specialised for 3x3 / stride 1 / pad 1 / width 32 (the residual convs): per chunk of 64 pixels
// (2 image rows) the raw rows of 64 input channels (with halo) and of 64 dy channels are staged ONCE in LDS;
// each wave owns a 32(k) x 32(c) output tile for all 9 taps (9 accumulators), so one A operand read feeds
// 9 MFMAs and every input element is loaded from global memory once instead of 9 times.
// -------------------------------------------------------------------------------------------
#define W3_LDA 65                 // dy tile [64 m][64 px + 1]
#define W3_ROWS 4                 // 2 pixel rows + top/bottom halo
#define W3_CH 137                 // floats per channel in LDS: 4 rows x 34, padded to an odd stride (137 % 32 = 9)

struct W3Params {
  const float *DY, *X, *zero;
  float *part;                   // [splits][M][C][9]
  int N, M, C, H;
  int nchunks, chunks_per_split; // chunk = (n, row pair)
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void igemm_w3x3_kernel(W3Params p) {
  constexpr bool BF16 = MODE == 1, SPLIT = MODE == 2;
  __shared__ __attribute__((aligned(16))) float lds[64 * W3_LDA + 64 * W3_CH];
  float *As = lds, *Bs = lds + 64 * W3_LDA;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware mapping: the (C/64) x (M/64) output tiles of ONE pixel split all read the same dy / x chunks, so they are
  // placed on one XCD (shared L2), adjacent in launch order; the splits are dealt round-robin to the XCDs
  int c0, m0, split;
  {
    const int tiles = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    int t;
    if ((gridDim.z & 7) == 0) {
      const int xcd = lin & 7, q = lin >> 3;
      t = q % tiles;
      split = (q / tiles) * 8 + xcd;
    } else {
      t = lin % tiles;
      split = lin / tiles;
    }
    c0 = (t % gridDim.x) * 64;
    m0 = (t / gridDim.x) * 64;
  }
  const int HW = p.H * 32, rows2 = p.H / 2;
  const int ch_begin = split * p.chunks_per_split;
  int ch_end = ch_begin + p.chunks_per_split;
  if (ch_end > p.nchunks) ch_end = p.nchunks;

  // halo columns (0 and 33 of each (channel,row) line) are always zero
  for (int u = tid; u < 64 * W3_ROWS * 2; u += 256) {
    const int line = u >> 1;
    Bs[(line >> 2) * W3_CH + (line & 3) * 34 + (u & 1) * 33] = 0.f;
  }

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  f32x4 areg[4], breg[8];
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const float *Ap = As + (wm * 32 + l31) * W3_LDA + half;
  const float *Bp = Bs + (wn * 32 + l31) * W3_CH + half;

  for (int ch = ch_begin - 1; ch < ch_end; ++ch) {
    if (ch >= ch_begin) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int u = tid + 256 * i;
        float *d = As + (u >> 4) * W3_LDA + (u & 15) * 4;
        d[0] = areg[i][0];
        d[1] = areg[i][1];
        d[2] = areg[i][2];
        d[3] = areg[i][3];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int u = tid + 256 * i;
        const int line = u >> 3;
        float *d = Bs + (line >> 2) * W3_CH + (line & 3) * 34 + 1 + (u & 7) * 4;
        d[0] = breg[i][0];
        d[1] = breg[i][1];
        d[2] = breg[i][2];
        d[3] = breg[i][3];
      }
      __syncthreads();
    }
    if (ch + 1 < ch_end) {
      const int nc = ch + 1;
      const int n = nc / rows2, y0 = (nc - n * rows2) * 2;
      const float *dyb = p.DY + ((long)n * p.M + m0) * HW + y0 * 32;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int u = tid + 256 * i;
        areg[i] = *reinterpret_cast<const f32x4 *>(dyb + (long)(u >> 4) * HW + (u & 15) * 4);
      }
      const float *xb = p.X + ((long)n * p.C + c0) * HW;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int u = tid + 256 * i;
        const int line = u >> 3;
        const int img_row = y0 - 1 + (line & 3);
        const bool ok = img_row >= 0 && img_row < p.H;
        const float *src = ok ? (xb + (long)(line >> 2) * HW + img_row * 32 + (u & 7) * 4) : p.zero;
        breg[i] = *reinterpret_cast<const f32x4 *>(src);
      }
    }
    if (ch >= ch_begin && MODE == 0) {
#pragma unroll
      for (int row = 0; row < 2; ++row) {
        const float *Ar = Ap + row * 32, *Br = Bp + row * 34;
#pragma unroll 2
        for (int kq = 0; kq < 16; ++kq) {           // k-step: pixels (row, 2kq + half)
          const float a = Ar[2 * kq];
          float b[9];
#pragma unroll
          for (int t = 0; t < 9; ++t) b[t] = Br[(t / 3) * 34 + 2 * kq + (t % 3)];
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
        }
      }
    }
    if (ch >= ch_begin && (BF16 || SPLIT)) {
      // bf16 / split modes: K = 16 consecutive pixels of one image row per MFMA (lanes 0-31: pixels 0..7, 32-63: 8..15)
      const float *A0 = As + (wm * 32 + l31) * W3_LDA + 8 * half;
      const float *B0 = Bs + (wn * 32 + l31) * W3_CH + 8 * half;
#pragma unroll 1
      for (int q = 0; q < 4; ++q) {                 // 64-pixel chunk = 4 groups of 16 pixels (2 rows x 2 halves)
        const int row = q >> 1, col = (q & 1) * 16;
        float av[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = A0[row * 32 + col + e];
        bf16x8 ah, am, al;
        if (SPLIT) {
          split3(av, ah, am, al);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) ah[e] = (__bf16)av[e];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          float bv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) bv[e] = B0[(row + t / 3) * 34 + col + (t % 3) + e];
          if (SPLIT) {
            bf16x8 bh, bm, bl;
            split3(bv, bh, bm, bl);
            acc[t] = mfma_split6(ah, am, al, bh, bm, bl, acc[t]);
          } else {
            bf16x8 bf;
#pragma unroll
            for (int e = 0; e < 8; ++e) bf[e] = (__bf16)bv[e];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bf, acc[t], 0, 0, 0);
          }
        }
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

// -------------------------------------------------------------------------------------------
// bf16 MFMA weight-gradient kernel of the 3x3 / stride-1 / width-32 layers (math mode 1).  Same tiling as
// igemm_w3x3_kernel (64 m x 64 c x 9 taps per workgroup, 64-pixel chunks, 9 accumulators per wave), but dy and the
// input rows are rounded to bf16 when they are staged and stored pixel-contiguous, so a K = 16 operand (16 consecutive
// pixels: lanes 0-31 pixels 0-7, lanes 32-63 pixels 8-15) is ONE aligned ds_read_b128.  The column taps s = 0 / 2 need
// the same 8 pixels shifted by one: they are built from the aligned vector and one neighbouring dword with
// v_alignbyte (8 VALU per row) instead of re-reading LDS — the register-conversion variant read 80 dwords per group and
// was LDS-bound at 560 TFLOP/s.
// -------------------------------------------------------------------------------------------
#define WB_LDA 72                 // bf16 elements per dy row: 64 pixels + 8 pad (144 B: 8 lanes cover all banks)
#define WB_ROW 48                 // one input row: 8 zero pad | 32 pixels | 8 zero pad
#define WB_CH (4 * WB_ROW + 8)    // 200 bf16 per channel (400 B)

__global__ __launch_bounds__(256, 2) void igemm_w3x3_bf16_kernel(W3Params p) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * WB_LDA + 64 * WB_CH];
  unsigned short *Aq = lds, *Bq = lds + 64 * WB_LDA;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int c0, m0, split;                                                // XCD-aware mapping as in igemm_w3x3_kernel
  {
    const int tiles = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    int t;
    if ((gridDim.z & 7) == 0) {
      const int xcd = lin & 7, q = lin >> 3;
      t = q % tiles;
      split = (q / tiles) * 8 + xcd;
    } else {
      t = lin % tiles;
      split = lin / tiles;
    }
    c0 = (t % gridDim.x) * 64;
    m0 = (t / gridDim.x) * 64;
  }
  const int HW = p.H * 32, rows2 = p.H / 2;
  const int ch_begin = split * p.chunks_per_split;
  int ch_end = ch_begin + p.chunks_per_split;
  if (ch_end > p.nchunks) ch_end = p.nchunks;

  // the zero pads left and right of every staged input row are written once
  for (int u = tid; u < 64 * 4 * 16; u += 256) {
    const int line = u >> 4, e = u & 15;                            // line = channel * 4 + row
    Bq[(line >> 2) * WB_CH + (line & 3) * WB_ROW + (e < 8 ? e : 32 + e)] = 0;
  }

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  f32x4 areg[4], breg[8];
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const unsigned short *Ap = Aq + (wm * 32 + l31) * WB_LDA + 8 * half;
  const unsigned short *Bp = Bq + (wn * 32 + l31) * WB_CH + 8 + 8 * half;     // + 8: skip the left pad

  for (int ch = ch_begin - 1; ch < ch_end; ++ch) {
    if (ch >= ch_begin) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int u = tid + 256 * i;
        bf16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (__bf16)areg[i][e];
        *reinterpret_cast<bf16x4 *>(Aq + (u >> 4) * WB_LDA + (u & 15) * 4) = v;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int u = tid + 256 * i;
        const int line = u >> 3;
        bf16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (__bf16)breg[i][e];
        *reinterpret_cast<bf16x4 *>(Bq + (line >> 2) * WB_CH + (line & 3) * WB_ROW + 8 + (u & 7) * 4) = v;
      }
      __syncthreads();
    }
    if (ch + 1 < ch_end) {
      const int nc = ch + 1;
      const int n = nc / rows2, y0 = (nc - n * rows2) * 2;
      const float *dyb = p.DY + ((long)n * p.M + m0) * HW + y0 * 32;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int u = tid + 256 * i;
        areg[i] = *reinterpret_cast<const f32x4 *>(dyb + (long)(u >> 4) * HW + (u & 15) * 4);
      }
      const float *xb = p.X + ((long)n * p.C + c0) * HW;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int u = tid + 256 * i;
        const int line = u >> 3;
        const int img_row = y0 - 1 + (line & 3);
        const bool ok = img_row >= 0 && img_row < p.H;
        const float *src = ok ? (xb + (long)(line >> 2) * HW + img_row * 32 + (u & 7) * 4) : p.zero;
        breg[i] = *reinterpret_cast<const f32x4 *>(src);
      }
    }
    if (ch >= ch_begin) {
#pragma unroll 1
      for (int q = 0; q < 4; ++q) {                 // 4 groups of 16 pixels: (row 0 / 1) x (columns 0-15 / 16-31)
        const int row = q >> 1, col = (q & 1) * 16;
        const bf16x8 af = *reinterpret_cast<const bf16x8 *>(Ap + row * 32 + col);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const unsigned short *bp = Bp + (row + r) * WB_ROW + col;
          const u32x4 cur = *reinterpret_cast<const u32x4 *>(bp);
          const unsigned prev = *reinterpret_cast<const unsigned *>(bp - 2);     // pixels -2, -1 of this lane's 8
          const unsigned next = *reinterpret_cast<const unsigned *>(bp + 8);     // pixels +8, +9
          u32x4 lft, rgt;                                                       // the 8 pixels shifted by -1 / +1
          lft[0] = __builtin_amdgcn_alignbyte(cur[0], prev, 2);
          lft[1] = __builtin_amdgcn_alignbyte(cur[1], cur[0], 2);
          lft[2] = __builtin_amdgcn_alignbyte(cur[2], cur[1], 2);
          lft[3] = __builtin_amdgcn_alignbyte(cur[3], cur[2], 2);
          rgt[0] = __builtin_amdgcn_alignbyte(cur[1], cur[0], 2);
          rgt[1] = __builtin_amdgcn_alignbyte(cur[2], cur[1], 2);
          rgt[2] = __builtin_amdgcn_alignbyte(cur[3], cur[2], 2);
          rgt[3] = __builtin_amdgcn_alignbyte(next, cur[3], 2);
          acc[3 * r + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, lft), acc[3 * r + 0], 0, 0, 0);
          acc[3 * r + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, cur), acc[3 * r + 1], 0, 0, 0);
          acc[3 * r + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, rgt), acc[3 * r + 2], 0, 0, 0);
        }
      }
    }
  }

  const int c = c0 + wn * 32 + l31;
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
