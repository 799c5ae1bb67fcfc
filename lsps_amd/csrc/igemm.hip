// Implicit-GEMM Conv2d / ConvTranspose2d for gfx950 on the f32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32, 157 TFLOP/s dense peak).
//
// One gather-GEMM covers every conv-shaped op of the depth path (NCHW, no im2col buffer):
//
//   F kernel  D[m][pix] = sum_{(c,t)} Wp[(c,t)][m] * X[n][c][ph*ist+dh[t]][pw*ist+dw[t]]
//       - Conv2d forward / ConvTranspose2d dgrad:  "forward direction" (big -> small image),
//         all R*S taps, ist = stride, dh = r - pad.
//       - Conv2d dgrad / ConvTranspose2d forward:  "transposed direction" (small -> big image),
//         decomposed by output parity class (a,b) in [0,stride)^2: a class only sees the taps with
//         (a + pad - r) % stride == 0, so no multiply-by-zero work is issued (9 taps total for a
//         3x3 stride-2 layer instead of 36); ist = 1, dh = (a + pad - r)/stride.
//   W kernel  D[m][(c,t)] = sum_{pix} Small[n][m][pix] * Big[n][c][ph*ist+dh[t]][pw*ist+dw[t]]
//       - Conv2d / ConvTranspose2d weight gradient, split over the pixel reduction, partial
//         tiles reduced deterministically by a second kernel.
//
// MFMA roles: A = weights / small-side rows (i = output row m), B = pixels or (c,t) columns.
// The C/D layout then puts 32 consecutive pixels of one output channel in lanes 0..31 of one
// accumulator register => 128-byte coalesced NCHW stores without any LDS transpose.
// Tiles are staged through LDS (register prefetch of the next chunk overlaps the MFMA chain).
//
// Build-time experiment hooks (never defined in the shipped build; see DESIGN.md §3.1 for what they measured):
//   LSPS_ABL_NOLOAD / _NOSTORE / _NOBAR / _SAMEADDR  ablations of the generic F kernel's phases
//   LSPS_ABL_SPLIT_NOA   cache-resident weight tile in the split-precision kernel
//   LSPS_STAGGER_PRIO, LSPS_F_LDS_PAD=<floats>       priority stagger / occupancy cap experiments
//   LSPS_NO_F3X3, LSPS_NO_W3X3                       force the generic kernels for the 3x3 layers
//
// Reference call sites replaced: every nn.Conv2d / nn.ConvTranspose2d on the path
// (src/trainers/common_net.py:162-163,250,262; src/trainers/lsps_nets.py:17-23,123-124,226-227)
// and their autograd backward (total_loss.backward(), src/trainers/lsps_trainer.py:71,130,212,257).
#include "common.h"
#include <stdarg.h>
#include <string.h>

namespace lsps {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: HIP's float4 struct defeats SROA (scratch)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Math mode of the MFMA conv kernels: 0 = exact f32 MFMA (default), 1 = operands rounded to bf16 in registers
// (v_cvt_pk_bf16_f32, RNE) and v_mfma_f32_32x32x16_bf16 with f32 accumulation (BASELINE config 5).  HBM and LDS
// tensors stay f32 in both modes.
static int g_math_mode = 0;

// x = hi + mid + lo with three bf16 values (8+8+8 significand bits: exact for f32).  Used by math mode 2.
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 &hi, bf16x8 &mid, bf16x8 &lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h = (__bf16)x[e];
    const float r1 = x[e] - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    hi[e] = h;
    mid[e] = m;
    lo[e] = (__bf16)r2;
  }
}

// acc += a*b to ~f32 accuracy from six bf16 MFMAs (dropped terms mid*lo, lo*mid, lo*lo are < 2^-24 relative)
__device__ __forceinline__ f32x16 mfma_split6(const bf16x8 &ah, const bf16x8 &am, const bf16x8 &al, const bf16x8 &bh,
                                              const bf16x8 &bm, const bf16x8 &bl, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  return acc;
}

#define LSPS_MAXT 49
#define BK_F 32   // reduction chunk of the F kernel
#define BK_W 64   // reduction (pixel) chunk of the W kernel
#define LDW (BK_W + 1)

struct Taps {
  int T;
  int toff[LSPS_MAXT];          // dh*Wx + dw
  signed char dh[LSPS_MAXT], dw[LSPS_MAXT];
};

struct FParams {
  const float *X, *Wp, *bias;
  float *Y;
  const int2 *gtab;              // [REDp] (element offset c*HxWx + toff[t], tap index t; t = 63 for padding rows)
  const float *zero;             // >= 1 float of zeros: where masked-out gathers read from
  int Cx, Hx, Wx, HxWx;          // gather source [N][Cx][Hx][Wx]
  int PH, PW, P, NPIX;           // output pixel lattice per sample, P = PH*PW, NPIX = N*P
  int ist;                       // input step per lattice step
  int RED, REDp, Mp;             // RED = Cx*T ; packed weights are [REDp][Mp], zero padded
  unsigned magicT;               // floor(2^32/T)+1 (T>1)
  int M, HyWy, Wy, h0, hs, w0, ws;   // D[m][pix] -> Y[n][m][h0+hs*ph][w0+ws*pw]
  int act;
  float slope;
  // split over the reduction (few-workgroup problems, e.g. the Post head: 20 x n outputs, 8192-long reduction):
  // blockIdx.z handles chunks [z*chunks_per_split, ...) and writes raw partial sums to part[z][m][pix]
  int ksplit, chunks_per_split;
  float *part;
  Taps taps;
};

struct WParams {
  const float *Small, *Big;
  float *part;                   // [splits][M][J]
  const int2 *jtab;              // [Jp = J rounded up to 128] (offset, tap) per column j = (c,t); tap 63 = padding
  const float *zero;
  int Cx, Hx, Wx, HxWx;          // Big = [N][Cx][Hx][Wx]
  int PH, PW, P, NPIX;           // Small = [N][M][PH][PW]
  int ist;
  int M, J;                      // J = Cx*T
  unsigned magicT;
  int nchunks, chunks_per_split;
  Taps taps;
};

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  if (act == LSPS_ACT_LRELU) return v > 0.f ? v : v * slope;
  if (act == LSPS_ACT_TANH) return tanhf(v);
  return v;
}

// -------------------------------------------------------------------------------------------
// weight packing: Wp[red=(c,t)][m] = W[m*sm + c*sc + tapidx[t]], zero padded to [REDp][Mp]
// -------------------------------------------------------------------------------------------
#define ZERO_SLOT_FLOATS 64
struct PackParams {
  const float *W;
  float *Wp;                     // [REDp][Mp] followed by ZERO_SLOT_FLOATS zeros
  int2 *gtab;                    // [REDp]
  int M, Mp, RED, REDp, T, HxWx;
  int cc;                        // 0: rows ordered (c,t); >0: rows ordered [c/cc][t][c%cc] (3x3 kernel)
  unsigned magicT;
  long sm, sc;
  int tapidx[LSPS_MAXT];
  int toff[LSPS_MAXT];
};

__global__ __launch_bounds__(256) void pack_weights_kernel(PackParams p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)p.REDp * p.Mp;
  if (idx >= total + ZERO_SLOT_FLOATS) return;
  if (idx >= total) {
    p.Wp[idx] = 0.f;
    return;
  }
  const int red = (int)(idx / p.Mp), m = (int)(idx - (long)red * p.Mp);
  float v = 0.f;
  int2 g = make_int2(0, 63);
  if (red < p.RED) {
    int c, t;
    if (p.cc > 0) {
      const int chunk = red / (p.T * p.cc), rem = red - chunk * p.T * p.cc;
      t = rem / p.cc;
      c = chunk * p.cc + (rem - t * p.cc);
    } else {
      c = (p.T == 1) ? red : (int)__umulhi((unsigned)red, p.magicT);
      t = red - c * p.T;
    }
    if (m < p.M) v = p.W[(long)m * p.sm + (long)c * p.sc + p.tapidx[t]];
    g = make_int2(c * p.HxWx + p.toff[t], t);
  }
  p.Wp[idx] = v;
  if (m == 0) p.gtab[red] = g;
}

// column table of the W kernel: j = (c,t) -> (offset, tap); also zeroes the slot masked gathers read
__global__ __launch_bounds__(256) void build_jtab_kernel(int2 *jtab, float *zero, int J, int Jp, int T, unsigned magicT,
                                                         int HxWx, Taps taps) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < ZERO_SLOT_FLOATS) zero[j] = 0.f;
  if (j >= Jp) return;
  int2 g = make_int2(0, 63);
  if (j < J) {
    const int c = (T == 1) ? j : (int)__umulhi((unsigned)j, magicT);
    const int t = j - c * T;
    g = make_int2(c * HxWx + taps.toff[t], t);
  }
  jtab[j] = g;
}

// -------------------------------------------------------------------------------------------
// F kernel
// -------------------------------------------------------------------------------------------
template <int WM, int WN, int WAVES_M, int WAVES_N, bool BF16 = false>
__global__ __launch_bounds__(256, 2) void igemm_f_kernel(FParams p) {
  constexpr int BM = WM * 32 * WAVES_M, BN = WN * 32 * WAVES_N, BK = BK_F;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(BN == 64 || BN == 128 || BN == 256, "pixel tile");
  constexpr int PIXW = BN / 64;        // waves side by side along the pixel tile
  constexpr int RGROUPS = 4 / PIXW;    // wave groups stacked along the reduction rows
  constexpr int NB = BK / RGROUPS;     // B gathers per thread per chunk: rows rbase*NB .. rbase*NB+NB-1
  constexpr int A4 = BK * BM / 4 / 256;
  static_assert(A4 >= 1, "A tile");

#ifndef LSPS_F_LDS_PAD
#define LSPS_F_LDS_PAD 0
#endif
  __shared__ __attribute__((aligned(16))) float lds[BK * BM + BK * BN + LSPS_F_LDS_PAD];
  float *As = lds, *Bs = lds + BK * BM;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.y * BM;
#ifdef LSPS_STAGGER_PRIO
  // Workgroups sharing a CU otherwise march in lock-step (fair MFMA arbitration) and hit their
  // load/store phases together; distinct static priorities order them so phases interleave.
  switch ((blockIdx.x >> 3) & 3) {
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    case 3: __builtin_amdgcn_s_setprio(3); break;
    default: break;
  }
#endif

  // ---- this thread's gather pixel (one column of the B tile)
  const int pcol = (wave % PIXW) * 64 + lane;
  const int rbase = wave / PIXW;
  const long Jg = (long)blockIdx.x * BN + pcol;
  const bool pv = Jg < p.NPIX;
  int gn = 0, gph = 0, gpw = 0;
  if (pv) {
    gn = (int)(Jg / p.P);
    const int rem = (int)(Jg - (long)gn * p.P);
    gph = rem / p.PW;
    gpw = rem - gph * p.PW;
  }
  const int ih0 = gph * p.ist, iw0 = gpw * p.ist;
  const float *xb = p.X + ((long)gn * p.Cx * p.Hx + ih0) * p.Wx + iw0;
  unsigned long long mask = 0ull;
  for (int t = 0; t < p.taps.T; ++t) {
    const int ih = ih0 + p.taps.dh[t], iw = iw0 + p.taps.dw[t];
    if (pv && ih >= 0 && ih < p.Hx && iw >= 0 && iw < p.Wx) mask |= (1ull << t);
  }

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float breg[NB];
  f32x4 areg[A4];
  const int nchunks = p.REDp / BK;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, half = lane >> 5;

  // Software pipeline, one copy of each phase: iteration `ch` first moves the registers prefetched for
  // chunk ch into LDS, then issues the global loads of chunk ch+1 (in flight during the MFMA chain), then
  // runs the MFMA chain of chunk ch.
  const int ch_first = p.ksplit > 1 ? blockIdx.z * p.chunks_per_split : 0;
  int ch_last = p.ksplit > 1 ? ch_first + p.chunks_per_split : nchunks;
  if (ch_last > nchunks) ch_last = nchunks;
  int2 tabv = make_int2(0, 63);
  if (ch_first < ch_last) tabv = p.gtab[ch_first * BK + rbase * NB + (lane & (NB - 1))];
  for (int ch = ch_first - 1; ch < ch_last; ++ch) {
    if (ch >= ch_first) {
#ifndef LSPS_ABL_NOBAR
      __syncthreads();
#endif
#ifndef LSPS_ABL_NOSTORE
#pragma unroll
      for (int i = 0; i < A4; ++i) {
        const int u = tid + 256 * i;
        const int row = u / (BM / 4), c4 = u % (BM / 4);
        *reinterpret_cast<f32x4 *>(As + row * BM + c4 * 4) = areg[i];
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) Bs[(rbase * NB + i) * BN + pcol] = breg[i];
#endif
#ifndef LSPS_ABL_NOBAR
      __syncthreads();
#endif
    }
#ifdef LSPS_ABL_NOLOAD
    if (ch + 1 < ch_last) {
#pragma unroll
      for (int i = 0; i < A4; ++i) areg[i] = (f32x4){1.f, 2.f, 3.f, (float)ch};
#pragma unroll
      for (int i = 0; i < NB; ++i) breg[i] = (float)(ch + i);
    }
    if (false) {
#else
    if (ch + 1 < ch_last) {
#endif
      const int k0 = (ch + 1) * BK;
#pragma unroll
      for (int i = 0; i < A4; ++i) {
        const int u = tid + 256 * i;
        const int row = u / (BM / 4), c4 = u % (BM / 4);
        areg[i] = *reinterpret_cast<const f32x4 *>(p.Wp + (long)(k0 + row) * p.Mp + m0 + c4 * 4);
      }
      // The NB gather-table rows of this wave were fetched one chunk ago by ONE vector load (lane i holds
      // row i) and are broadcast with v_readlane: no scalar-memory round trip per gather.  Masked-out lanes
      // read the zero slot: no select after the load, so nothing waits for the gathers until they are stored
      // to LDS after the MFMA chain.
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int off = __builtin_amdgcn_readlane(tabv.x, i);
        const int t = __builtin_amdgcn_readlane(tabv.y, i);
        const bool ok = (mask >> t) & 1ull;
#ifdef LSPS_ABL_SAMEADDR
        const float *src = ok ? (p.X + (off & 1023) + lane) : p.zero;     // same issue work, L1-resident data
#else
        const float *src = ok ? (xb + off) : p.zero;
#endif
        breg[i] = *src;
      }
      // table rows of the chunk after this one (clamped: the tail read is never used)
      {
        int nk = k0 + BK;
        if (nk >= p.REDp) nk = 0;
        tabv = p.gtab[nk + rbase * NB + (lane & (NB - 1))];
      }
    }
    if (ch >= ch_first && !BF16) {
#pragma unroll 8
      for (int kk = 0; kk < BK / 2; ++kk) {
        float a[WM], b[WN];
        const int row = 2 * kk + half;
#pragma unroll
        for (int i = 0; i < WM; ++i) a[i] = As[row * BM + (wm * WM + i) * 32 + l31];
#pragma unroll
        for (int j = 0; j < WN; ++j) b[j] = Bs[row * BN + (wn * WN + j) * 32 + l31];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (ch >= ch_first && BF16) {               // bf16 mode: K = 16 reduction rows per MFMA, rounded in registers
#pragma unroll
      for (int k16 = 0; k16 < BK / 16; ++k16) {
        const int row0 = k16 * 16 + 8 * half;
        bf16x8 af[WM], bf[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) af[i][e] = (__bf16)As[(row0 + e) * BM + (wm * WM + i) * 32 + l31];
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) bf[j][e] = (__bf16)Bs[(row0 + e) * BN + (wn * WN + j) * 32 + l31];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: lane holds pixel column l31 of each 32x32 tile, rows (r&3)+8*(r>>2)+4*half
  if (p.ksplit > 1) {                       // raw partial sums; bias / activation are applied by the reducer
    float *part = p.part + (long)blockIdx.z * p.M * p.NPIX;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const long Jo = (long)blockIdx.x * BN + (wn * WN + j) * 32 + l31;
      if (Jo >= p.NPIX) continue;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (m < p.M) part[(long)m * p.NPIX + Jo] = acc[i][j][r];
        }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const long Jo = (long)blockIdx.x * BN + (wn * WN + j) * 32 + l31;
    if (Jo >= p.NPIX) continue;
    const int n = (int)(Jo / p.P);
    const int rem = (int)(Jo - (long)n * p.P);
    const int ph = rem / p.PW, pw = rem - ph * p.PW;
    float *yb = p.Y + (long)n * p.M * p.HyWy + (long)(p.h0 + p.hs * ph) * p.Wy + (p.w0 + p.ws * pw);
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < p.M) {
          float v = acc[i][j][r];
          if (p.bias) v += p.bias[m];
          yb[(long)m * p.HyWy] = apply_act(v, p.act, p.slope);
        }
      }
    }
  }
}



// y[n][m][p] = act(bias[m] + sum_z part[z][m][n*P + p])   (forward direction only: contiguous output lattice)
__global__ __launch_bounds__(256) void ksplit_reduce_kernel(const float *__restrict__ part, const float *__restrict__ bias,
                                                            float *__restrict__ y, int M, int P, long NPIX, int ksplit,
                                                            int act, float slope) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;       // over [M][NPIX]
  if (idx >= (long)M * NPIX) return;
  const int m = (int)(idx / NPIX);
  const long pix = idx - (long)m * NPIX;
  float s = 0.f;
  for (int z = 0; z < ksplit; ++z) s += part[(long)z * M * NPIX + idx];
  if (bias) s += bias[m];
  const long n = pix / P, pp = pix - n * P;
  y[(n * M + m) * P + pp] = apply_act(s, act, slope);
}

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

struct F3Params {
  const float *X, *Wp, *bias, *zero;
  float *Y;
  int Cx, H, M, Mp, NT;          // NT = N * (H/TR) pixel tiles
  int tiles_per_img;             // H / TR
  int act;
  float slope;
};

// TR = output rows per tile: 4 (tile 128 ch x 128 px, waves 2x2, each 64 ch x 64 px) or, when that grid would
// leave CUs idle (estimate modes run the generator on 8 samples), 2 (128 ch x 64 px, waves 4x1, each 32 ch x 64 px).
template <int TR>
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
  const int m0 = blockIdx.y * BM;
  const int n = blockIdx.x / p.tiles_per_img;
  const int row0 = (blockIdx.x - n * p.tiles_per_img) * TR;      // first output row of the tile
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
  const int nchunks = p.Cx / F3_CC;
  const int wm = TR == 4 ? (wave >> 1) : wave, wn = TR == 4 ? (wave & 1) : 0;
  const int l31 = lane & 31, half = lane >> 5;
  const float *Ap = As + half * BM + wm * WM * 32 + l31;
  const float *Bp = Bs + half * CH + wn * 2 * F3_LDW + l31;

  for (int ch = -1; ch < nchunks; ++ch) {
    if (ch >= 0) {
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
    if (ch >= 0) {
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

  // epilogue: wave's pixel rows wn*2 + j, column l31
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float *yb = p.Y + (long)n * p.M * HW + (long)(row0 + wn * 2 + j) * 32 + l31;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < p.M) {
          float v = acc[i][j][r];
          if (p.bias) v += p.bias[m];
          yb[(long)m * HW] = apply_act(v, p.act, p.slope);
        }
      }
    }
  }
}


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

template <bool BF16>
__global__ __launch_bounds__(256, 2) void igemm_f3x3s2_kernel(FS2Params p) {
  constexpr int BM = 128, RC = F3_CC * 9;
  constexpr int A4 = RC * BM / 4 / 256;                            // 9 float4 of weights per thread per chunk
  constexpr int LINES = F3_CC * FS2_ROWS;                          // 72 (channel, row) lines of 64 columns
  constexpr int B4 = (LINES * 16 + 255) / 256;                     // 5 float4 of input per thread per chunk (4.5)
  __shared__ __attribute__((aligned(16))) float lds[RC * BM + F3_CC * FS2_CH];
  float *As = lds, *Bs = lds + RC * BM;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.y * BM;
  const int n = blockIdx.x / p.tiles_per_img;
  const int rem = blockIdx.x - n * p.tiles_per_img;
  const int p0 = (rem / p.qblocks) * 4, q0 = (rem - (rem / p.qblocks) * p.qblocks) * 32;
  const int Hx = 2 * p.P, Wx = 2 * p.Q, HWx = Hx * Wx;
  const float *xn = p.X + (long)n * p.Cx * HWx;

  int b_lds[B4], b_off[B4];
  bool b_use[B4], b_ok[B4];
#pragma unroll
  for (int i = 0; i < B4; ++i) {
    const int u = tid + 256 * i;
    b_use[i] = u < LINES * 16;
    const int line = u >> 4, c4 = u & 15;
    const int chn = line / FS2_ROWS, r = line - chn * FS2_ROWS;
    const int ih = 2 * p0 - 1 + r;
    b_ok[i] = b_use[i] && ih >= 0;                                 // ih <= 2 p0 + 7 < 2P always
    b_lds[i] = chn * FS2_CH + r * FS2_ROW + 2 * c4;
    b_off[i] = chn * HWx + ih * Wx + 2 * q0 + c4 * 4;
  }
  const bool h_use = tid < LINES;                                   // column 2 q0 - 1 of every line
  const int h_chn = tid / FS2_ROWS, h_r = tid - h_chn * FS2_ROWS;
  const bool h_ok = h_use && (2 * p0 - 1 + h_r) >= 0 && q0 > 0;
  const int h_off = h_chn * HWx + (2 * p0 - 1 + h_r) * Wx + 2 * q0 - 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 areg[A4], breg[B4];
  float hreg = 0.f;
  const int nchunks = p.Cx / F3_CC;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const float *Ap = As + half * BM + wm * 64 + l31;
  const float *Bp = Bs + half * FS2_CH + wn * 4 * FS2_ROW + l31;     // wave's output rows wn*2 + j -> input rows 2(wn*2+j) + r

  for (int ch = -1; ch < nchunks; ++ch) {
    if (ch >= 0) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < A4; ++i) *reinterpret_cast<f32x4 *>(As + (tid + 256 * i) * 4) = areg[i];
#pragma unroll
      for (int i = 0; i < B4; ++i)
        if (b_use[i]) {
          float *d = Bs + b_lds[i];
          d[33] = breg[i][0];                                      // E[2 c4]
          d[1] = breg[i][1];                                       // O'[2 c4 + 1]
          d[34] = breg[i][2];                                      // E[2 c4 + 1]
          d[2] = breg[i][3];                                       // O'[2 c4 + 2]
        }
      if (h_use) Bs[h_chn * FS2_CH + h_r * FS2_ROW] = hreg;        // O'[0]
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
      const float *xc = xn + (long)(ch + 1) * F3_CC * HWx;
#pragma unroll
      for (int i = 0; i < B4; ++i) {
        const float *src = b_ok[i] ? (xc + b_off[i]) : p.zero;
        breg[i] = *reinterpret_cast<const f32x4 *>(src);
      }
      {
        const float *src = h_ok ? (xc + h_off) : p.zero;
        hreg = *src;
      }
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
          for (int e = 0; e < 8; ++e) af[i][e] = (__bf16)A0[(tsel * F3_CC + e) * BM + i * 32];
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
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int tr = t / 3, ts = t - tr * 3;
        const int coff = ts == 1 ? 33 : (ts == 2 ? 1 : 0);
#pragma unroll
        for (int cp = 0; cp < F3_CC / 2; ++cp) {
          const int kk = t * (F3_CC / 2) + cp;
          float a[2], b[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) a[i] = Ap[2 * kk * BM + i * 32];
#pragma unroll
          for (int j = 0; j < 2; ++j) b[j] = Bp[2 * cp * FS2_CH + (2 * j + tr) * FS2_ROW + coff];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  }

  const long PQ = (long)p.P * p.Q;
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
  const float *xn = p.X + (long)n * p.Cx * HWs;

  int b_lds[B4];
  int b_off[B4];                                             // element offsets inside one 16-channel slab (< 2^31)
  bool b_use[B4], b_ok[B4];
#pragma unroll
  for (int i = 0; i < B4; ++i) {
    const int u = tid + 256 * i;
    b_use[i] = u < TS_CC * ROWS * 8;
    const int line = u >> 3, c4 = u & 7;
    const int chn = line / ROWS, r = line - chn * ROWS;
    b_ok[i] = b_use[i] && (p0 + r) < p.Hs;
    b_lds[i] = chn * CHS + r * 34 + c4 * 4;
    b_off[i] = chn * HWs + (p0 + r) * p.Ws + q0 + c4 * 4;
  }
  // column q0 + 32 (the right neighbour of the tile): one scalar per (channel, row) line
  const bool h_use = tid < TS_CC * ROWS;
  const int h_chn = tid / ROWS, h_r = tid - h_chn * ROWS;
  const bool h_ok = h_use && (p0 + h_r) < p.Hs && (q0 + 32) < p.Ws;
  const int h_off = h_chn * HWs + (p0 + h_r) * p.Ws + q0 + 32;

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
        if (b_use[i]) {
          float *d = Bs + b_lds[i];
          d[0] = breg[i][0];
          d[1] = breg[i][1];
          d[2] = breg[i][2];
          d[3] = breg[i][3];
        }
      if (h_use) Bs[h_chn * CHS + h_r * 34 + 32] = hreg;
      __syncthreads();
    }
    if (ch + 1 < nchunks) {
      // packed weights: rows [chunk of 16 channels][tap 0..8][16 channels]; this class uses taps 3..5 (a = 0) or
      // 0..2 and 6..8 (a = 1)
      const float *wsrc = p.Wp + (long)(ch + 1) * (9 * TS_CC) * p.Mp + m0;
#pragma unroll
      for (int i = 0; i < A4; ++i) {
        const int u = tid + 256 * i;
        const int row = u / (BM / 4), c4 = u - row * (BM / 4);
        const int lt = row / TS_CC;
        const int grow = (APAR ? (lt < 3 ? lt : lt + 3) : lt + 3) * TS_CC + (row - lt * TS_CC);
        areg[i] = *reinterpret_cast<const f32x4 *>(wsrc + (long)grow * p.Mp + c4 * 4);
      }
      const float *xc = xn + (long)(ch + 1) * TS_CC * HWs;
#pragma unroll
      for (int i = 0; i < B4; ++i) {
        const float *src = b_ok[i] ? (xc + b_off[i]) : p.zero;
        breg[i] = *reinterpret_cast<const f32x4 *>(src);
      }
      {
        const float *src = h_ok ? (xc + h_off) : p.zero;
        hreg = *src;
      }
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
#pragma unroll
      for (int lt = 0; lt < NT; ++lt) {
        const int s = lt % 3;
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
  __shared__ __attribute__((aligned(16))) float lds[TS_LDS_FLOATS];
  if (blockIdx.z == 0)
    ts2_body<0, BM, BF16>(p, lds);
  else
    ts2_body<1, BM, BF16>(p, lds);
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
  const int m0 = blockIdx.y * 128;
  const int n = blockIdx.x / p.tiles_per_img;
  const int row0 = (blockIdx.x - n * p.tiles_per_img) * TR;
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
  const unsigned short *wq = p.Wq + (long)blockIdx.y * nchunks * ACH;

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
          yb[(long)m * HW] = apply_act(v, p.act, p.slope);
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------
// W kernel (weight gradient): tile (64*TW) m x (64*TW) (c,t) columns x 64 pixels, split over pixel chunks.
// TW = 2: 128x128 tile, each wave 2x2 MFMA tiles.  TW = 1: 64x64 tile (one MFMA tile per wave) for the
// layers with <= 64 rows / columns (7x7 stem: 64 x 49; 1x1 head), where a 128x128 tile is >= 75 % padding.
// -------------------------------------------------------------------------------------------
template <int TW, bool BF16 = false>
__global__ __launch_bounds__(256) void igemm_w_kernel(WParams p) {
  constexpr int BM = 64 * TW, BN = 64 * TW, BK = BK_W, RW = 16 * TW;   // RW rows / columns loaded per wave
  __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDW];
  float *As = lds, *Bs = lds + BM * LDW;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.y * BM, j0 = blockIdx.x * BN;
  const int split = blockIdx.z;
  const int ch_begin = split * p.chunks_per_split;
  int ch_end = ch_begin + p.chunks_per_split;
  if (ch_end > p.nchunks) ch_end = p.nchunks;
  const int T = p.taps.T;

  f32x16 acc[TW][TW];
#pragma unroll
  for (int i = 0; i < TW; ++i)
#pragma unroll
    for (int j = 0; j < TW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float areg[RW], breg[RW];
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  // this wave's RW column-table rows (fixed for the whole kernel): lane i holds row i, broadcast by v_readlane
  const int2 tabv = p.jtab[j0 + wave * RW + (lane & (RW - 1))];

  for (int ch = ch_begin - 1; ch < ch_end; ++ch) {
    if (ch >= ch_begin) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        As[(wave * RW + i) * LDW + lane] = areg[i];
        Bs[(wave * RW + i) * LDW + lane] = breg[i];
      }
      __syncthreads();
    }
    if (ch + 1 < ch_end) {
      const long q = (long)(ch + 1) * BK + lane;    // this lane's pixel of the chunk
      const bool pv = q < p.NPIX;
      int n = 0, ph = 0, pw = 0, pp = 0;
      if (pv) {
        n = (int)(q / p.P);
        pp = (int)(q - (long)n * p.P);
        ph = pp / p.PW;
        pw = pp - ph * p.PW;
      }
      const int ih0 = ph * p.ist, iw0 = pw * p.ist;
      unsigned long long mask = 0ull;
      for (int t = 0; t < T; ++t) {
        const int ih = ih0 + p.taps.dh[t], iw = iw0 + p.taps.dw[t];
        if (pv && ih >= 0 && ih < p.Hx && iw >= 0 && iw < p.Wx) mask |= (1ull << t);
      }
      const float *sb = p.Small + (long)n * p.M * p.P + pp;
      const float *xb = p.Big + ((long)n * p.Cx * p.Hx + ih0) * p.Wx + iw0;
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        const int m = m0 + wave * RW + i;           // wave-uniform row
        const bool ok = pv && m < p.M;
        const float *src = ok ? (sb + (long)m * p.P) : p.zero;   // masked lanes read the zero slot: no select
        areg[i] = *src;
      }
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        const int off = __builtin_amdgcn_readlane(tabv.x, i);
        const int t = __builtin_amdgcn_readlane(tabv.y, i);
        const bool ok = (mask >> t) & 1ull;
        const float *src = ok ? (xb + off) : p.zero;
        breg[i] = *src;
      }
    }
    if (ch >= ch_begin && !BF16) {
#pragma unroll 8
      for (int kk = 0; kk < BK / 2; ++kk) {
        float a[TW], b[TW];
        const int col = 2 * kk + half;
#pragma unroll
        for (int i = 0; i < TW; ++i) a[i] = As[((wm * TW + i) * 32 + l31) * LDW + col];
#pragma unroll
        for (int j = 0; j < TW; ++j) b[j] = Bs[((wn * TW + j) * 32 + l31) * LDW + col];
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
          for (int j = 0; j < TW; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (ch >= ch_begin && BF16) {               // bf16 mode: K = 16 consecutive pixels per MFMA
#pragma unroll
      for (int k16 = 0; k16 < BK / 16; ++k16) {
        const int col0 = k16 * 16 + 8 * half;
        bf16x8 af[TW], bf[TW];
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) af[i][e] = (__bf16)As[((wm * TW + i) * 32 + l31) * LDW + col0 + e];
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) bf[j][e] = (__bf16)Bs[((wn * TW + j) * 32 + l31) * LDW + col0 + e];
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
          for (int j = 0; j < TW; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  float *out = p.part + (long)split * p.M * p.J;
#pragma unroll
  for (int j = 0; j < TW; ++j) {
    const int jj = j0 + (wn * TW + j) * 32 + l31;
    if (jj >= p.J) continue;
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TW + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < p.M) out[(long)m * p.J + jj] = acc[i][j][r];
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
  const int c0 = blockIdx.x * 64, m0 = blockIdx.y * 64, split = blockIdx.z;
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

  for (int ch = ch_begin - 1; ch < ch_end; ++ch) {
#ifdef LSPS_ABL_WS2_NOSTAGE
    if (ch == ch_begin) {
#else
    if (ch >= ch_begin) {
#endif
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
      if (tid < 192) Bs[tid * WS2_ROW] = hreg;           // O'[0]: the column left of the block (zero at the image edge)
      __syncthreads();
    }
    if (ch + 1 < ch_end) {
      const int nc = ch + 1;
      const int n = nc / per_img;
      const int rem = nc - n * per_img;
      const int y = rem / p.qblocks, q0 = (rem - y * p.qblocks) * 32;
      const float *sb = p.Small + ((long)n * p.M + m0) * HWs + y * p.Ws + q0;
#ifdef LSPS_ABL_WS2_NOLOAD
      if (ch < ch_begin) {
#endif
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int u = tid + 512 * i;
        areg[i] = *reinterpret_cast<const f32x4 *>(sb + (long)(u >> 3) * HWs + (u & 7) * 4);
      }
      const float *bb = p.Big + ((long)n * p.C + c0) * HWb + 2 * q0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int u = tid + 512 * i;
        const int line = u >> 4, c4 = u & 15;
        const int chn = line / 3, r = line - chn * 3;
        const int rb = 2 * y - 1 + r;
        const float *src = rb >= 0 ? (bb + (long)chn * HWb + (long)rb * Wb + c4 * 4) : p.zero;
        breg[i] = *reinterpret_cast<const f32x4 *>(src);
      }
      if (tid < 192) {
        const int chn = tid / 3, r = tid - chn * 3;
        const int rb = 2 * y - 1 + r;
        const float *src = (rb >= 0 && q0 > 0) ? (bb + (long)chn * HWb + (long)rb * Wb - 1) : p.zero;
        hreg = *src;
      }
#ifdef LSPS_ABL_WS2_NOLOAD
      }
#endif
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

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float *part, float *out, long n, int splits) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  // eight independent loads in flight per thread; fixed summation order (deterministic)
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 8 <= splits; k += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] += part[(long)(k + u) * n + i];
  }
  for (; k < splits; ++k) s[0] += part[(long)k * n + i];
  out[i] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}



// -------------------------------------------------------------------------------------------
// Single-input-channel convolutions (the 7x7 stems: 1 -> 64 channels, stride 1 in the generator, stride 2 in the
// discriminator).  They carry 0.3 % of the flops but stream the largest activations of the net (64 x 128 x 128 floats
// per sample), so they are HBM-bound: the kernels below read the input image / dy and write the output exactly once.
// The contraction (49 taps, padded to 50) still runs on the matrix pipe — K = taps for the forward, K = pixels for the
// weight gradient — with the image rows staged in LDS (zero halo) and the taps as per-lane LDS offsets.
// -------------------------------------------------------------------------------------------
#define C1_KS 25                  // k-steps of 2 taps: up to 50 taps (7x7 = 49)
#define C1_MAXLDS 5400            // floats of staged image rows (21 KB: with the weight tile a workgroup stays under 36 KB)

struct C1Params {
  const float *X, *W, *bias;
  float *Y;
  int N, H, Wd, K, P, Q, R, S, stride, pad;
  int TP, rows, LW;              // output rows per workgroup, staged input rows, LDS row stride (Wd + 2 pad)
  int act;
  float slope;
};

__device__ __forceinline__ void c1_stage_rows(float *xs, const float *xn, int row0, int rows, int LW, int H, int Wd, int pad,
                                               int tid, int nthreads) {
  for (int u = tid; u < rows * LW; u += nthreads) {
    const int r = u / LW, c = u - r * LW;
    const int ih = row0 + r, iw = c - pad;
    const bool ok = ih >= 0 && ih < H && iw >= 0 && iw < Wd;
    // unconditional load from a clamped address + select: a predicated load makes hipcc branch around every load
    const float v = xn[(long)min(max(ih, 0), H - 1) * Wd + min(max(iw, 0), Wd - 1)];
    xs[u] = ok ? v : 0.f;
  }
}

// out[n][k][p][q] = act(bias[k] + sum_t W[k][t] * x[n][p*s + r_t - pad][q*s + c_t - pad]);  grid (P/TP, ceil(K/64), N)
#define C1_WLD 51                 // LDS row stride of the zero-padded weight tile [64 k][50 taps] (51 % 32 = 19: conflict-free)
#define C1_FIXED_LDS ((64 * C1_WLD + 64 + 2 * C1_KS) * sizeof(float))
// The stores are the floor here (64 x 128 x 128 floats per sample; a store-only variant of this kernel runs at 4.0 TB/s,
// a compute-only one at 0.86 of that time).  Weights and tap offsets live in registers for the whole workgroup; a
// variant that re-read them from LDS to run 4 waves per SIMD was not faster.
__global__ __launch_bounds__(256, 2) void c1_fwd_kernel(C1Params p) {
  extern __shared__ __attribute__((aligned(16))) float c1_lds[];
  float *wl = c1_lds, *bl = wl + 64 * C1_WLD;
  int *tl = reinterpret_cast<int *>(bl + 64);
  float *xs = bl + 64 + 2 * C1_KS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int n = blockIdx.z, m0 = blockIdx.y * 64, p0 = blockIdx.x * p.TP;
  const int T = p.R * p.S;

  c1_stage_rows(xs, p.X + (long)n * p.H * p.Wd, p0 * p.stride - p.pad, p.rows, p.LW, p.H, p.Wd, p.pad, tid, 256);
  // weights (coalesced: 64 x T contiguous floats), bias and tap offsets go through LDS once per workgroup
  for (int u = tid; u < 64 * 2 * C1_KS; u += 256) {
    const int k = u / (2 * C1_KS), t = u - k * (2 * C1_KS);
    const bool ok = t < T && m0 + k < p.K;
    const float v = p.W[(long)min(m0 + k, p.K - 1) * T + min(t, T - 1)];
    wl[k * C1_WLD + t] = ok ? v : 0.f;
  }
  if (tid < 64) {
    const float v = p.bias ? p.bias[min(m0 + tid, p.K - 1)] : 0.f;
    bl[tid] = v;
  }
  if (tid < 2 * C1_KS) {
    const int r = tid < T ? tid / p.S : 0, c = tid < T ? tid - r * p.S : 0;
    tl[tid] = r * p.LW + c;
  }
  __syncthreads();

  float a[C1_KS][2];
  int boff[C1_KS];
#pragma unroll
  for (int ks = 0; ks < C1_KS; ++ks) {
    boff[ks] = tl[2 * ks + half];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[ks][i] = wl[(i * 32 + l31) * C1_WLD + 2 * ks + half];
  }
  const int qblocks = p.Q / 32, nseg = p.TP * qblocks;
  const long PQ = (long)p.P * p.Q;
  const bool lrelu = p.act == LSPS_ACT_LRELU, other = p.act != LSPS_ACT_LRELU && p.act != LSPS_ACT_NONE;
  const bool full = m0 + 64 <= p.K;
  for (int seg = wave; seg < nseg; seg += 4) {
    const int pr = seg / qblocks, q0 = (seg - pr * qblocks) * 32;
    const float *Bp = xs + pr * p.stride * p.LW + (q0 + l31) * p.stride;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#ifdef LSPS_ABL_C1_NOMFMA
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#else
#pragma unroll
    for (int ks = 0; ks < C1_KS; ++ks) {
#endif
      const float b = Bp[boff[ks]];
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks][i], b, acc[i], 0, 0, 0);
    }
#ifdef LSPS_ABL_C1_NOSTORE
    if (acc[0][0] != 123.f && acc[1][3] != 77.f) continue;
#endif
    float *yb = p.Y + ((long)n * p.K + m0) * PQ + (long)(p0 + pr) * p.Q + q0 + l31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[i][r] + bl[kl];
        if (lrelu) v = v > 0.f ? v : v * p.slope;
        if (other) v = apply_act(v, p.act, p.slope);
        if (full || m0 + kl < p.K) yb[(long)kl * PQ] = v;
      }
  }
}

// dW[k][t] = sum_{n,p,q} dy[n][k][p][q] * x[n][p*s + r_t - pad][q*s + c_t - pad]   (K <= 64, T <= 64, Q in {32, 64, 128})
// Per iteration: RB = 128 / Q output rows of one image (128 pixels = one 32-pixel segment per wave): dy[64 k][128] and
// the (RB-1)*s + R input rows are staged in LDS; the reduction index of the MFMA is the pixel, its columns are the
// taps (per-lane LDS offsets).  Both operands of the NEXT iteration are fetched into registers before the MFMAs of the
// current one are issued.
#define C1W_LDA 129
#define C1W_XMAX 1536             // floats of staged input rows: 6 per thread
struct C1WParams {
  const float *X, *DY;
  float *part;                   // [blocks][K * T]
  int N, H, Wd, K, P, Q, R, S, stride, pad;
  int LW, RB, xrows;             // LDS row stride (Wd + 2 pad), output rows per iteration, staged input rows
  int iters_total, iters_per_block;      // iterations = N * P / RB
};

__global__ __launch_bounds__(256, 2) void c1_wgrad_kernel(C1WParams p) {
  __shared__ __attribute__((aligned(16))) float lds[64 * C1W_LDA + C1W_XMAX];
  float *dys = lds, *xs = lds + 64 * C1W_LDA;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int T = p.R * p.S;
  const long PQ = (long)p.P * p.Q;
  const int HWx = p.H * p.Wd;

  int toff[2];                    // LDS offset of this lane's tap in column tiles 0 (taps 0..31) and 1 (taps 32..63)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int t = j * 32 + l31;
    const int r = t < T ? t / p.S : 0, c = t < T ? t - r * p.S : 0;
    toff[j] = r * p.LW + c;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging assignment (fixed per thread): dy float4 u = tid + 256 i -> (k, 4 pixels of the 128); x element u -> (row, col)
  const int q4 = p.Q / 4;
  int d_off[8], d_lds[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int u = tid + 256 * i;
    const int k = u >> 5, c4 = u & 31;                     // 32 float4 = 128 pixels per channel
    const int rb = c4 / q4, cq = c4 - rb * q4;             // pixel -> (row in the iteration, column)
    d_off[i] = (k < p.K ? k : 0) * (int)PQ + rb * p.Q + cq * 4;
    d_lds[i] = k * C1W_LDA + c4 * 4;
  }
  const int xcount = p.xrows * p.LW;
  int x_r[6], x_c[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int u = tid + 256 * i;
    x_r[i] = u / p.LW;
    x_c[i] = u - x_r[i] * p.LW - p.pad;
  }

  f32x4 dreg[8];
  float xreg[6];
  auto fetch = [&](int it) {
    const int n = it / (p.P / p.RB), pr = (it - n * (p.P / p.RB)) * p.RB;
    const float *dyn = p.DY + (long)n * p.K * PQ + (long)pr * p.Q;
#pragma unroll
    for (int i = 0; i < 8; ++i) dreg[i] = *reinterpret_cast<const f32x4 *>(dyn + d_off[i]);
    const float *xn = p.X + (long)n * HWx;
    const int row0 = pr * p.stride - p.pad;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int ih = row0 + x_r[i], iw = x_c[i];
      const bool ok = tid + 256 * i < xcount && ih >= 0 && ih < p.H && iw >= 0 && iw < p.Wd;
      const float v = xn[min(max(ih, 0), p.H - 1) * p.Wd + min(max(iw, 0), p.Wd - 1)];     // clamped, unconditional
      xreg[i] = ok ? v : 0.f;
    }
  };

  const int it_begin = blockIdx.x * p.iters_per_block;
  int it_end = it_begin + p.iters_per_block;
  if (it_end > p.iters_total) it_end = p.iters_total;
  if (it_begin < it_end) fetch(it_begin);
  const int qblocks = p.Q / 32;
  const int srow = wave / qblocks, sq0 = (wave - srow * qblocks) * 32;      // this wave's segment: (row, first column)
  const float *Ap = dys + l31 * C1W_LDA + wave * 32 + half;
  const float *Bp = xs + srow * p.stride * p.LW + (sq0 + half) * p.stride;
  for (int it = it_begin; it < it_end; ++it) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool live = (tid + 256 * i) >> 5 < p.K;        // channels >= K are zero rows
      float *d = dys + d_lds[i];
      d[0] = live ? dreg[i][0] : 0.f;
      d[1] = live ? dreg[i][1] : 0.f;
      d[2] = live ? dreg[i][2] : 0.f;
      d[3] = live ? dreg[i][3] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (tid + 256 * i < xcount) xs[tid + 256 * i] = xreg[i];
    __syncthreads();
    if (it + 1 < it_end) fetch(it + 1);
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {             // pixels 2ks + half of the wave's segment
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = Ap[i * 32 * C1W_LDA + 2 * ks];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bp[toff[j] + 2 * ks * p.stride];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  // the four waves' partial sums are added in a fixed order through LDS (deterministic), then written once per block
  float *red = lds;               // [64 k][64 t]
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
  float *out = p.part + (long)blockIdx.x * p.K * T;
  for (int u = tid; u < p.K * T; u += 256) {
    const int k = u / T, t = u - k * T;
    out[u] = red[k * 64 + t];
  }
}

// -------------------------------------------------------------------------------------------
// col2im for the dgrad of a 1-input-channel conv computed as a GEMM over taps:
//   Z[n][t][p][q] = sum_k W[k][t] dy[n][k][p][q]   (MFMA kernel, M = R*S rows)
//   dx[n][h][w]   = sum_{t=(r,s) valid} Z[n][t][(h+pad-r)/st][(w+pad-s)/st]
__global__ __launch_bounds__(256) void col2im_c1_kernel(const float *__restrict__ Z, float *__restrict__ dx, int N, int H,
                                                        int W, int P, int Q, int R, int S, int st, int pad) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)N * H * W) return;
  const int w = (int)(idx % W);
  const long t1 = idx / W;
  const int h = (int)(t1 % H), n = (int)(t1 / H);
  const float *zn = Z + (long)n * R * S * P * Q;
  float acc = 0.f;
  for (int r = 0; r < R; ++r) {
    const int a = h + pad - r;
    if (a < 0 || a % st != 0) continue;
    const int pp = a / st;
    if (pp >= P) continue;
    for (int s2 = 0; s2 < S; ++s2) {
      const int b = w + pad - s2;
      if (b < 0 || b % st != 0) continue;
      const int qq = b / st;
      if (qq >= Q) continue;
      acc += zn[((long)(r * S + s2) * P + pp) * Q + qq];
    }
  }
  dx[idx] = acc;
}

// -------------------------------------------------------------------------------------------
// Pointwise head: 1x1 (transposed) conv with ONE output channel (generator output, lsps_nets.py:226-229).
// 1.05 MMAC per sample against 4 MB of input: HBM-bound, so no MFMA — one float4 of pixels per thread,
// channel loop with the weight in SGPRs, fused bias + tanh.  y[n][pix] = act(b + sum_c w[c] x[n][c][pix]).
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pw1_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                      const float *__restrict__ b, float *__restrict__ y, int N, int C,
                                                      int HW4, int act, float slope) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;       // (n, pixel quad)
  if (idx >= (long)N * HW4) return;
  const int n = (int)(idx / HW4), q = (int)(idx - (long)n * HW4);
  const f32x4 *xp = reinterpret_cast<const f32x4 *>(x) + (long)n * C * HW4 + q;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int c = 0; c < C; ++c) acc += w[c] * xp[(long)c * HW4];
  const float bb = b ? b[0] : 0.f;
  f32x4 r;
  r[0] = apply_act(acc[0] + bb, act, slope);
  r[1] = apply_act(acc[1] + bb, act, slope);
  r[2] = apply_act(acc[2] + bb, act, slope);
  r[3] = apply_act(acc[3] + bb, act, slope);
  reinterpret_cast<f32x4 *>(y)[idx] = r;
}

// dx[n][c][pix] = w[c] * dy[n][pix]
__global__ __launch_bounds__(256) void pw1_dgrad_kernel(const float *__restrict__ dy, const float *__restrict__ w,
                                                        float *__restrict__ dx, int N, int C, int HW4) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)N * HW4) return;
  const int n = (int)(idx / HW4), q = (int)(idx - (long)n * HW4);
  const f32x4 g = reinterpret_cast<const f32x4 *>(dy)[idx];
  f32x4 *xp = reinterpret_cast<f32x4 *>(dx) + (long)n * C * HW4 + q;
#pragma unroll 8
  for (int c = 0; c < C; ++c) xp[(long)c * HW4] = w[c] * g;
}

// part[s][c] = sum over slice s of (n,pix) of x[n][c][pix] * dy[n][pix]   (grid: C x S)
__global__ __launch_bounds__(256) void pw1_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                        float *__restrict__ part, int N, int C, int HW4, long slice4) {
  __shared__ float red[4];
  const int c = blockIdx.x, sidx = blockIdx.y;
  const long total4 = (long)N * HW4;
  const long e0 = (long)sidx * slice4;
  long e1 = e0 + slice4;
  if (e1 > total4) e1 = total4;
  float s = 0.f;
  for (long e = e0 + threadIdx.x; e < e1; e += 256) {
    const long n = e / HW4, q = e - n * HW4;
    const f32x4 a = reinterpret_cast<const f32x4 *>(x)[(n * C + c) * HW4 + q];
    const f32x4 g = reinterpret_cast<const f32x4 *>(dy)[e];
    s += (a[0] * g[0] + a[1] * g[1]) + (a[2] * g[2] + a[3] * g[3]);
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) part[(long)sidx * C + c] = s;
}

// db[c] = sum_{n,p} t[n][c][p].  Stage 1: grid (C, S): block (c,s) sums slice s of the N*HW elements of
// channel c into part[c*S+s]; stage 2 (reduce_partials_kernel with n=C... see run_bias_grad) adds the S slices.
__global__ __launch_bounds__(256) void bias_grad_partial_kernel(const float *__restrict__ t, float *__restrict__ part,
                                                                int N, int C, int HW, long slice) {
  __shared__ float red[4];
  const int c = blockIdx.x, sidx = blockIdx.y;
  const long total = (long)N * HW;
  const long e0 = (long)sidx * slice;
  long e1 = e0 + slice;
  if (e1 > total) e1 = total;
  float s = 0.f;
  if ((HW & 3) == 0) {
    for (long e = e0 + (long)threadIdx.x * 4; e < e1; e += 1024) {
      const long n = e / HW;
      const long i = e - n * HW;
      const float4 v = *reinterpret_cast<const float4 *>(t + (n * C + c) * HW + i);
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (long e = e0 + threadIdx.x; e < e1; e += 256) {
      const long n = e / HW;
      const long i = e - n * HW;
      s += t[(n * C + c) * HW + i];
    }
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) part[(long)sidx * C + c] = s;   // layout [S][C] so that reduce_partials sums over S
}

// -------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------
static unsigned magic_for(int T) { return T <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)T + 1ull); }

// tile configs: 0: 128x128, 1: 64x256, 2: 32x256, 3: 128x64, 4: 64x64 (the last two for small pixel counts:
// the late discriminator layers have N*4..N*64 pixels, and a 128x128 tiling leaves half of the CUs idle)
static int cfg_bm(int cfg) { return cfg == 0 ? 128 : (cfg == 1 ? 64 : (cfg == 2 ? 32 : (cfg == 3 ? 128 : 64))); }
static int cfg_bn(int cfg) { return cfg == 0 ? 128 : (cfg == 1 || cfg == 2 ? 256 : 64); }
static int choose_cfg(int M, long NPIX) {
  if (M < 64) return 2;
  if (M < 128) return 1;
  const long g0 = (long)ceil_div(NPIX, 128) * ceil_div(M, 128);
  if (g0 >= 512) return 0;
  const long g3 = (long)ceil_div(NPIX, 64) * ceil_div(M, 128);
  if (g3 >= 512) return 3;
  return 4;
}

struct TapList {
  int T;
  int dh[LSPS_MAXT], dw[LSPS_MAXT], idx[LSPS_MAXT];
};

static void fill_taps(Taps &t, const TapList &l, int Wx) {
  t.T = l.T;
  for (int i = 0; i < LSPS_MAXT; ++i) {
    t.toff[i] = 0;
    t.dh[i] = 0;
    t.dw[i] = 0;
  }
  for (int i = 0; i < l.T; ++i) {
    t.dh[i] = (signed char)l.dh[i];
    t.dw[i] = (signed char)l.dw[i];
    t.toff[i] = l.dh[i] * Wx + l.dw[i];
  }
}

// workspace carve of one packed class: [gtab: REDp int2][Wp: REDp*Mp floats][zero slot]
static size_t class_bytes(int REDp, int Mp) {
  return align_up((size_t)REDp * sizeof(int2), 256) + align_up(((size_t)REDp * Mp + ZERO_SLOT_FLOATS) * sizeof(float), 256);
}

static int launch_pack(const float *W, void *cls, int M, int Mp, int RED, int REDp, const TapList &l, long sm, long sc,
                       int HxWx, int Wx, hipStream_t st, const float **Wp_out, const int2 **gtab_out,
                       const float **zero_out, int cc = 0) {
  int2 *gtab = (int2 *)cls;
  float *Wp = (float *)((char *)cls + align_up((size_t)REDp * sizeof(int2), 256));
  *Wp_out = Wp;
  *gtab_out = gtab;
  *zero_out = Wp + (size_t)REDp * Mp;
  if (REDp == 0) return 0;
  PackParams pp;
  pp.W = W;
  pp.Wp = Wp;
  pp.gtab = gtab;
  pp.HxWx = HxWx;
  pp.cc = cc;
  for (int i = 0; i < LSPS_MAXT; ++i) pp.toff[i] = i < l.T ? l.dh[i] * Wx + l.dw[i] : 0;
  pp.M = M;
  pp.Mp = Mp;
  pp.RED = RED;
  pp.REDp = REDp;
  pp.T = l.T;
  pp.magicT = magic_for(l.T);
  pp.sm = sm;
  pp.sc = sc;
  for (int i = 0; i < LSPS_MAXT; ++i) pp.tapidx[i] = i < l.T ? l.idx[i] : 0;
  const long total = (long)REDp * Mp + ZERO_SLOT_FLOATS;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, pp);
  LSPS_CHECK_LAUNCH("pack_weights");
  return 0;
}

static int launch_f(const FParams &p, int cfg, hipStream_t st) {
  const int BM = cfg_bm(cfg), BN = cfg_bn(cfg);
  dim3 grid(ceil_div(p.NPIX, BN), ceil_div(p.M, BM), p.ksplit > 1 ? p.ksplit : 1);
  if (grid.x == 0 || grid.y == 0) return 0;
  const bool bf = g_math_mode == 1;
#define LSPS_LAUNCH_F(...)                                                                  \
  do {                                                                                      \
    if (bf)                                                                                 \
      hipLaunchKernelGGL((igemm_f_kernel<__VA_ARGS__, true>), grid, dim3(256), 0, st, p);   \
    else                                                                                    \
      hipLaunchKernelGGL((igemm_f_kernel<__VA_ARGS__, false>), grid, dim3(256), 0, st, p);  \
  } while (0)
  if (cfg == 0)
    LSPS_LAUNCH_F(2, 2, 2, 2);
  else if (cfg == 1)
    LSPS_LAUNCH_F(2, 2, 1, 4);
  else if (cfg == 2)
    LSPS_LAUNCH_F(1, 2, 1, 4);
  else if (cfg == 3)
    LSPS_LAUNCH_F(1, 2, 4, 1);
  else
    LSPS_LAUNCH_F(1, 1, 2, 2);
#undef LSPS_LAUNCH_F
  LSPS_CHECK_LAUNCH("igemm_f");
  return 0;
}

static size_t packed_bytes(int Cin, int taps_total, int classes, int M) {
  const int Mp = (int)align_up((size_t)M, 128);
  // sum over classes of class_bytes(REDp_c, Mp) with sum REDp_c <= Cin*taps_total + 32*classes
  return class_bytes(Cin * taps_total + 32 * classes, Mp) + (size_t)classes * 1024;
}


static bool f3x3_ok(int Cin, int H, int W, int R, int S, int st_, int pad) {
  return R == 3 && S == 3 && st_ == 1 && pad == 1 && W == 32 && (H % 4) == 0 && (Cin % F3_CC) == 0;
}

// in [N][Cin][H][32] -> out [N][M][H][32]; tapidx maps kernel tap t=(dh+1)*3+(dw+1) to the weight's r*3+s
static int run_f3x3(const float *in, const float *W, const float *bias, float *out, int N, int Cin, int H, int M,
                    long sm, long sc, bool flip, int act, float slope, void *ws, size_t ws_bytes, hipStream_t st) {
  TapList l;
  l.T = 9;
  for (int t = 0; t < 9; ++t) {
    l.dh[t] = t / 3 - 1;
    l.dw[t] = t % 3 - 1;
    l.idx[t] = flip ? 8 - t : t;
  }
  const int RED = Cin * 9, REDp = RED;     // Cin % 8 == 0 -> RED % 72 == 0
  const int Mp = (int)align_up(M, 128);
  if (g_math_mode != 0) {                  // bf16 / split-precision variants: weights pre-converted to bf16 limb planes
    const int np = g_math_mode == 2 ? 3 : 1;
    const size_t wq_bytes = (size_t)(Mp / 128) * (Cin / 8) * np * FS_APLANE * sizeof(unsigned short);
    if (256 + wq_bytes > ws_bytes) {
      set_error("conv workspace too small: need %zu, have %zu", 256 + wq_bytes, ws_bytes);
      return LSPS_E_WS;
    }
    hipError_t e = hipMemsetAsync(ws, 0, 256, st);
    if (e != hipSuccess) {
      set_error("hipMemsetAsync: %s", hipGetErrorString(e));
      return LSPS_E_HIP;
    }
    FSPack pk;
    pk.W = W;
    pk.Wq = (unsigned short *)((char *)ws + 256);
    pk.M = M;
    pk.C = Cin;
    pk.np = np;
    pk.sm = sm;
    pk.sc = sc;
    for (int t = 0; t < 9; ++t) pk.tapidx[t] = l.idx[t];
    const long total = (long)(Mp / 128) * (Cin / 8) * FS_TAPS * 128 * 8;
    hipLaunchKernelGGL(pack_split_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, pk);
    LSPS_CHECK_LAUNCH("pack_split");
    FSParams q;
    memset(&q, 0, sizeof(q));
    q.X = in;
    q.bias = bias;
    q.zero = (const float *)ws;
    q.Wq = pk.Wq;
    q.Y = out;
    q.Cx = Cin;
    q.H = H;
    q.M = M;
    int tr2 = ((long)N * (H / 4) * (Mp / 128) >= 512) ? 4 : 2;
    if (np == 1 && (H % 8) == 0 && (long)N * (H / 8) * (Mp / 128) >= 1024) tr2 = 8;   // bf16: L2-bound, reuse weights twice
    q.tiles_per_img = H / tr2;
    q.act = act;
    q.slope = slope;
    const dim3 grid2(N * q.tiles_per_img, Mp / 128);
    if (np == 3) {
      if (tr2 == 4)
        hipLaunchKernelGGL((igemm_f3x3_split_kernel<4, 3>), grid2, dim3(256), 0, st, q);
      else
        hipLaunchKernelGGL((igemm_f3x3_split_kernel<2, 3>), grid2, dim3(256), 0, st, q);
    } else {
      if (tr2 == 8)
        hipLaunchKernelGGL((igemm_f3x3_split_kernel<8, 1>), grid2, dim3(256), 0, st, q);
      else if (tr2 == 4)
        hipLaunchKernelGGL((igemm_f3x3_split_kernel<4, 1>), grid2, dim3(256), 0, st, q);
      else
        hipLaunchKernelGGL((igemm_f3x3_split_kernel<2, 1>), grid2, dim3(256), 0, st, q);
    }
    LSPS_CHECK_LAUNCH("igemm_f3x3_split");
    return 0;
  }
  const size_t need = class_bytes(REDp, Mp);
  if (need > ws_bytes) {
    set_error("conv workspace too small: need %zu, have %zu", need, ws_bytes);
    return LSPS_E_WS;
  }
  F3Params p;
  memset(&p, 0, sizeof(p));
  const int2 *gtab_unused;
  int rc = launch_pack(W, ws, M, Mp, RED, REDp, l, sm, sc, H * 32, 32, st, &p.Wp, &gtab_unused, &p.zero, F3_CC);
  if (rc) return rc;
  p.X = in;
  p.bias = bias;
  p.Y = out;
  p.Cx = Cin;
  p.H = H;
  p.M = M;
  p.Mp = Mp;
  const int tr = ((long)N * (H / 4) * (Mp / 128) >= 512) ? 4 : 2;   // small batches: 2-row tiles, twice the workgroups
  p.tiles_per_img = H / tr;
  p.NT = N * p.tiles_per_img;
  p.act = act;
  p.slope = slope;
  const dim3 grid(p.NT, Mp / 128);
  if (tr == 4)
    hipLaunchKernelGGL(igemm_f3x3_kernel<4>, grid, dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(igemm_f3x3_kernel<2>, grid, dim3(256), 0, st, p);
  LSPS_CHECK_LAUNCH("igemm_f3x3");
  return 0;
}

static bool f3x3s2_ok(int Cb, int Hb, int Wb, int Cs, int Hs, int Ws, int R, int S, int st_, int pad) {
  return R == 3 && S == 3 && st_ == 2 && pad == 1 && Hb == 2 * Hs && Wb == 2 * Ws && (Ws % 32) == 0 && (Hs % 4) == 0 &&
         (Cb % F3_CC) == 0 && Cs >= 128 && (long)F3_CC * Hb * Wb < (1L << 31);
}

// in [N][Cb][2Hs][2Ws] -> out [N][M][Hs][Ws]
static int run_f3x3s2(const float *in, const float *W, const float *bias, float *out, int N, int Cb, int Hs, int Ws, int M,
                      long sm, long sc, int act, float slope, void *ws, size_t ws_bytes, hipStream_t st) {
  TapList l;
  l.T = 9;
  for (int t = 0; t < 9; ++t) {
    l.dh[t] = 0;
    l.dw[t] = 0;
    l.idx[t] = t;
  }
  const int RED = Cb * 9;
  const int Mp = (int)align_up(M, 128);
  const size_t need = class_bytes(RED, Mp);
  if (need > ws_bytes) {
    set_error("conv workspace too small: need %zu, have %zu", need, ws_bytes);
    return LSPS_E_WS;
  }
  FS2Params p;
  memset(&p, 0, sizeof(p));
  const int2 *gtab_unused;
  int rc = launch_pack(W, ws, M, Mp, RED, RED, l, sm, sc, 4 * Hs * Ws, 2 * Ws, st, &p.Wp, &gtab_unused, &p.zero, F3_CC);
  if (rc) return rc;
  p.X = in;
  p.bias = bias;
  p.Y = out;
  p.Cx = Cb;
  p.P = Hs;
  p.Q = Ws;
  p.M = M;
  p.Mp = Mp;
  p.qblocks = Ws / 32;
  p.tiles_per_img = (Hs / 4) * p.qblocks;
  p.act = act;
  p.slope = slope;
  if (g_math_mode == 1)
    hipLaunchKernelGGL(igemm_f3x3s2_kernel<true>, dim3(N * p.tiles_per_img, Mp / 128), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(igemm_f3x3s2_kernel<false>, dim3(N * p.tiles_per_img, Mp / 128), dim3(256), 0, st, p);
  LSPS_CHECK_LAUNCH("igemm_f3x3s2");
  return 0;
}

static bool c1_fwd_ok(int Cb, int Hb, int Wb, int Hs, int Ws, int R, int S, int st_, int pad, long sm) {
  return Cb == 1 && R * S <= 2 * C1_KS && (Ws % 32) == 0 && (st_ == 1 || st_ == 2) && sm == (long)R * S &&
         (long)R * (Wb + 2 * pad) <= C1_MAXLDS;
}

static int run_c1_fwd(const float *in, const float *W, const float *bias, float *out, int N, int Hb, int Wb, int M, int Hs,
                      int Ws, int R, int S, int st_, int pad, int act, float slope, hipStream_t st) {
  C1Params p;
  memset(&p, 0, sizeof(p));
  p.X = in;
  p.W = W;
  p.bias = bias;
  p.Y = out;
  p.N = N;
  p.H = Hb;
  p.Wd = Wb;
  p.K = M;
  p.P = Hs;
  p.Q = Ws;
  p.R = R;
  p.S = S;
  p.stride = st_;
  p.pad = pad;
  p.LW = Wb + 2 * pad;
  // output rows per workgroup: as many as fit in LDS (amortises the weight / tap set-up), while >= 1024 workgroups remain
  int tp = 32;
  while (tp > 1 && ((Hs % tp) != 0 || (long)((tp - 1) * st_ + R) * p.LW > C1_MAXLDS ||
                    ((long)N * (Hs / tp) * ceil_div(M, 64) < 1024 && tp > 4)))
    tp >>= 1;
  p.TP = tp;
  p.rows = (tp - 1) * st_ + R;
  p.act = act;
  p.slope = slope;
  hipLaunchKernelGGL(c1_fwd_kernel, dim3(Hs / tp, ceil_div(M, 64), N), dim3(256),
                     C1_FIXED_LDS + (size_t)p.rows * p.LW * sizeof(float), st, p);
  LSPS_CHECK_LAUNCH("c1_fwd");
  return 0;
}

// "forward direction": in = big image [N][Cb][Hb][Wb], out = small image [N][Cs][Hs][Ws]
//   out[n][m][p][q] = sum_{c,r,s} W(m,c,r,s) * in[n][c][p*st-pad+r][q*st-pad+s]
//   weight element address: W[m*sm + c*sc + r*S + s]
static int run_forward_dir(const float *in, const float *W, const float *bias, float *out, int N, int Cb, int Hb, int Wb,
                           int Cs, int Hs, int Ws, int R, int S, int st_, int pad, long sm, long sc, int act,
                           float slope, void *ws, size_t ws_bytes, hipStream_t st) {
#ifndef LSPS_NO_F3X3
  if (f3x3_ok(Cb, Hb, Wb, R, S, st_, pad) && Cs >= 128)
    return run_f3x3(in, W, bias, out, N, Cb, Hb, Cs, sm, sc, false, act, slope, ws, ws_bytes, st);
#endif
#ifndef LSPS_NO_F3X3S2
  if (f3x3s2_ok(Cb, Hb, Wb, Cs, Hs, Ws, R, S, st_, pad))
    return run_f3x3s2(in, W, bias, out, N, Cb, Hs, Ws, Cs, sm, sc, act, slope, ws, ws_bytes, st);
#endif
#ifndef LSPS_NO_C1
  if (c1_fwd_ok(Cb, Hb, Wb, Hs, Ws, R, S, st_, pad, sm))
    return run_c1_fwd(in, W, bias, out, N, Hb, Wb, Cs, Hs, Ws, R, S, st_, pad, act, slope, st);
#endif
  TapList l;
  l.T = R * S;
  for (int r = 0; r < R; ++r)
    for (int s = 0; s < S; ++s) {
      l.dh[r * S + s] = r - pad;
      l.dw[r * S + s] = s - pad;
      l.idx[r * S + s] = r * S + s;
    }
  const int M = Cs, RED = Cb * l.T;
  const int Mp = (int)align_up(M, 128), REDp = (int)align_up(RED, BK_F);
  const size_t need = class_bytes(REDp, Mp);
  if (need > ws_bytes) {
    set_error("conv workspace too small: need %zu, have %zu", need, ws_bytes);
    return LSPS_E_WS;
  }
  FParams p;
  memset(&p, 0, sizeof(p));
  int rc = launch_pack(W, ws, M, Mp, RED, REDp, l, sm, sc, Hb * Wb, Wb, st, &p.Wp, &p.gtab, &p.zero);
  if (rc) return rc;
  p.X = in;
  p.bias = bias;
  p.Y = out;
  p.Cx = Cb;
  p.Hx = Hb;
  p.Wx = Wb;
  p.HxWx = Hb * Wb;
  p.PH = Hs;
  p.PW = Ws;
  p.P = Hs * Ws;
  p.NPIX = N * Hs * Ws;
  p.ist = st_;
  p.RED = RED;
  p.REDp = REDp;
  p.Mp = Mp;
  p.magicT = magic_for(l.T);
  p.M = M;
  p.HyWy = Hs * Ws;
  p.Wy = Ws;
  p.h0 = 0;
  p.hs = 1;
  p.w0 = 0;
  p.ws = 1;
  p.act = act;
  p.slope = slope;
  fill_taps(p.taps, l, Wb);
  const int cfg = choose_cfg(M, p.NPIX);
  const long wgs = (long)ceil_div(p.NPIX, cfg_bn(cfg)) * ceil_div(M, cfg_bm(cfg));
  const int nchunks = REDp / BK_F;
  if (wgs <= 64 && nchunks >= 32) {          // a handful of workgroups with a long reduction: split it
    int ks = (int)(256 / wgs);
    if (ks > nchunks / 4) ks = nchunks / 4;
    const size_t part_bytes = (size_t)ks * M * p.NPIX * sizeof(float);
    if (ks > 1 && need + part_bytes <= ws_bytes) {
      p.ksplit = ks;
      p.chunks_per_split = ceil_div(nchunks, ks);
      p.part = (float *)((char *)ws + need);
      rc = launch_f(p, cfg, st);
      if (rc) return rc;
      const long total = (long)M * p.NPIX;
      hipLaunchKernelGGL(ksplit_reduce_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, (const float *)p.part, bias,
                         out, M, p.P, (long)p.NPIX, ks, act, slope);
      LSPS_CHECK_LAUNCH("ksplit_reduce");
      return 0;
    }
  }
  return launch_f(p, cfg, st);
}

static bool t3x3s2_ok(int Cb, int Hb, int Wb, int Cs, int Hs, int Ws, int R, int S, int st_, int pad) {
  const int tr = Cb >= 128 ? 4 : 8;
  return R == 3 && S == 3 && st_ == 2 && pad == 1 && Hb == 2 * Hs && Wb == 2 * Ws && (Ws % 32) == 0 && (Hs % tr) == 0 &&
         (Cs % TS_CC) == 0 && Cb >= 64;
}

// in [N][Cs][Hs][Ws] -> out [N][M][2Hs][2Ws]
static int run_t3x3s2(const float *in, const float *W, const float *bias, float *out, int N, int Cs, int Hs, int Ws, int M,
                      long sm, long sc, int act, float slope, void *ws, size_t ws_bytes, hipStream_t st) {
  TapList l;
  l.T = 9;
  for (int t = 0; t < 9; ++t) {
    l.dh[t] = 0;
    l.dw[t] = 0;
    l.idx[t] = t;
  }
  const int RED = Cs * 9;
  const int Mp = (int)align_up(M, 128);
  const size_t need = class_bytes(RED, Mp);
  if (need > ws_bytes) {
    set_error("conv workspace too small: need %zu, have %zu", need, ws_bytes);
    return LSPS_E_WS;
  }
  TS2Params p;
  memset(&p, 0, sizeof(p));
  const int2 *gtab_unused;
  int rc = launch_pack(W, ws, M, Mp, RED, RED, l, sm, sc, Hs * Ws, Ws, st, &p.Wp, &gtab_unused, &p.zero, TS_CC);
  if (rc) return rc;
  p.X = in;
  p.bias = bias;
  p.Y = out;
  p.Cx = Cs;
  p.Hs = Hs;
  p.Ws = Ws;
  p.M = M;
  p.Mp = Mp;
  p.qblocks = Ws / 32;
  p.act = act;
  p.slope = slope;
  const bool bf = g_math_mode == 1;
  if (M >= 128) {
    p.tiles_per_img = (Hs / 4) * p.qblocks;
    const dim3 grid(N * p.tiles_per_img, ceil_div(M, 128), 2);
    if (bf)
      hipLaunchKernelGGL((igemm_t3x3s2_kernel<128, true>), grid, dim3(256), 0, st, p);
    else
      hipLaunchKernelGGL((igemm_t3x3s2_kernel<128, false>), grid, dim3(256), 0, st, p);
  } else {
    p.tiles_per_img = (Hs / 8) * p.qblocks;
    const dim3 grid(N * p.tiles_per_img, ceil_div(M, 64), 2);
    if (bf)
      hipLaunchKernelGGL((igemm_t3x3s2_kernel<64, true>), grid, dim3(256), 0, st, p);
    else
      hipLaunchKernelGGL((igemm_t3x3s2_kernel<64, false>), grid, dim3(256), 0, st, p);
  }
  LSPS_CHECK_LAUNCH("igemm_t3x3s2");
  return 0;
}

// "transposed direction": in = small image [N][Cs][Hs][Ws], out = big image [N][Cb][Hb][Wb]
//   out[n][m][h][w] = sum_{c,r,s} W(m,c,r,s) * in[n][c][(h+pad-r)/st][(w+pad-s)/st]   (divisible, in range)
static int run_transposed_dir(const float *in, const float *W, const float *bias, float *out, int N, int Cb, int Hb,
                              int Wb, int Cs, int Hs, int Ws, int R, int S, int st_, int pad, long sm, long sc, int act,
                              float slope, void *ws, size_t ws_bytes, hipStream_t st) {
#ifndef LSPS_NO_F3X3
  // stride-1 transposed conv == forward 3x3 conv with flipped taps (dh = pad - r)
  if (f3x3_ok(Cs, Hs, Ws, R, S, st_, pad) && Hb == Hs && Wb == Ws && Cb >= 128)
    return run_f3x3(in, W, bias, out, N, Cs, Hs, Cb, sm, sc, true, act, slope, ws, ws_bytes, st);
#endif
#ifndef LSPS_NO_T3X3S2
  if (t3x3s2_ok(Cb, Hb, Wb, Cs, Hs, Ws, R, S, st_, pad))
    return run_t3x3s2(in, W, bias, out, N, Cs, Hs, Ws, Cb, sm, sc, act, slope, ws, ws_bytes, st);
#endif
  const int M = Cb;
  const int Mp = (int)align_up(M, 128);
  size_t used = 0;
  for (int a = 0; a < st_; ++a)
    for (int b = 0; b < st_; ++b) {
      const int PH = (Hb - a + st_ - 1) / st_, PW = (Wb - b + st_ - 1) / st_;
      if (PH <= 0 || PW <= 0) continue;
      TapList l;
      l.T = 0;
      for (int r = 0; r < R; ++r) {
        const int vr = a + pad - r;
        if (((vr % st_) + st_) % st_ != 0) continue;
        for (int s = 0; s < S; ++s) {
          const int vs = b + pad - s;
          if (((vs % st_) + st_) % st_ != 0) continue;
          l.dh[l.T] = vr / st_;
          l.dw[l.T] = vs / st_;
          l.idx[l.T] = r * S + s;
          ++l.T;
        }
      }
      const int RED = Cs * l.T;
      const int REDp = (int)align_up(RED, BK_F);
      const size_t need = class_bytes(REDp, Mp);
      if (used + need > ws_bytes) {
        set_error("conv workspace too small: need >= %zu, have %zu", used + need, ws_bytes);
        return LSPS_E_WS;
      }
      FParams p;
      memset(&p, 0, sizeof(p));
      int rc = launch_pack(W, (char *)ws + used, M, Mp, RED, REDp, l, sm, sc, Hs * Ws, Ws, st, &p.Wp, &p.gtab, &p.zero);
      used += need;
      if (rc) return rc;
      p.X = in;
      p.bias = bias;
      p.Y = out;
      p.Cx = Cs;
      p.Hx = Hs;
      p.Wx = Ws;
      p.HxWx = Hs * Ws;
      p.PH = PH;
      p.PW = PW;
      p.P = PH * PW;
      p.NPIX = N * PH * PW;
      p.ist = 1;
      p.RED = RED;
      p.REDp = REDp;
      p.Mp = Mp;
      p.magicT = magic_for(l.T);
      p.M = M;
      p.HyWy = Hb * Wb;
      p.Wy = Wb;
      p.h0 = a;
      p.hs = st_;
      p.w0 = b;
      p.ws = st_;
      p.act = act;
      p.slope = slope;
      fill_taps(p.taps, l, Ws);
      rc = launch_f(p, choose_cfg(M, p.NPIX), st);
      if (rc) return rc;
    }
  return 0;
}

static int wgrad_tile(int M, int J) { return (M <= 64 && J <= 64) ? 64 : 128; }

static int wgrad_splits(int M, int J, int nchunks) {
  // The W kernel runs 2 workgroups per CU (LDS-bound): aim at just under two full rounds of 256 CUs x 2.
  // Few-tile problems (7x7 stem: 64x49 outputs; 1x1 head) get up to 1024 pixel splits so that the whole chip
  // streams the activations; the partial buffer is capped at 256 MiB.
  const int tile = wgrad_tile(M, J);
  const long tiles = (long)ceil_div(M, tile) * ceil_div(J, tile);
  long s = 1024 / tiles;
  const long cap = ((long)256 << 20) / ((long)M * J * 4);
  if (s > cap) s = cap;
  if (s > nchunks) s = nchunks;
  if (s < 1) s = 1;
  return (int)s;
}


static bool w3x3_ok(int Cb, int Hb, int Wb, int Cs, int Hs, int Ws, int R, int S, int st_, int pad) {
  return R == 3 && S == 3 && st_ == 1 && pad == 1 && Wb == 32 && Ws == 32 && Hb == Hs && (Hb % 2) == 0 &&
         (Cb % 64) == 0 && (Cs % 64) == 0;
}

static int w3x3_splits(int M, int C, int nchunks) {
  const int tiles = (M / 64) * (C / 64);
  int s = 512 / tiles;                    // 2 workgroups per CU x 256 CUs, one round
  if (s > nchunks) s = nchunks;
  if (s < 1) s = 1;
  return s;
}

static size_t w3x3_ws_bytes(int N, int M, int C, int H) {
  return 256 + (size_t)w3x3_splits(M, C, N * H / 2) * 9 * M * C * sizeof(float);
}

static int run_w3x3(const float *dy, const float *x, float *dW, int N, int C, int H, int M, void *ws, size_t ws_bytes,
                    hipStream_t st) {
  W3Params p;
  memset(&p, 0, sizeof(p));
  p.DY = dy;
  p.X = x;
  p.N = N;
  p.M = M;
  p.C = C;
  p.H = H;
  p.nchunks = N * H / 2;
  const int splits = w3x3_splits(M, C, p.nchunks);
  p.chunks_per_split = ceil_div(p.nchunks, splits);
  if (w3x3_ws_bytes(N, M, C, H) > ws_bytes) {
    set_error("wgrad workspace too small: need %zu, have %zu", w3x3_ws_bytes(N, M, C, H), ws_bytes);
    return LSPS_E_WS;
  }
  float *zero = (float *)ws;
  hipError_t e = hipMemsetAsync(zero, 0, 256, st);
  if (e != hipSuccess) {
    set_error("hipMemsetAsync: %s", hipGetErrorString(e));
    return LSPS_E_HIP;
  }
  p.zero = zero;
  p.part = (float *)((char *)ws + 256);
  if (g_math_mode == 1)
    hipLaunchKernelGGL(igemm_w3x3_kernel<1>, dim3(C / 64, M / 64, splits), dim3(256), 0, st, p);
  else if (g_math_mode == 2)
    hipLaunchKernelGGL(igemm_w3x3_kernel<2>, dim3(C / 64, M / 64, splits), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(igemm_w3x3_kernel<0>, dim3(C / 64, M / 64, splits), dim3(256), 0, st, p);
  LSPS_CHECK_LAUNCH("igemm_w3x3");
  const long total = (long)M * C * 9;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, (const float *)p.part, dW, total,
                     splits);
  LSPS_CHECK_LAUNCH("reduce_partials");
  return 0;
}

static bool w3x3s2_ok(int Cb, int Hb, int Wb, int Cs, int Hs, int Ws, int R, int S, int st_, int pad) {
  return R == 3 && S == 3 && st_ == 2 && pad == 1 && Hb == 2 * Hs && Wb == 2 * Ws && (Ws % 32) == 0 && (Cb % 64) == 0 &&
         (Cs % 128) == 0;
}

static int w3x3s2_splits(int M, int C, int nchunks) {
  const int tiles = (M / 128) * (C / 64);
  int s = 256 / tiles;                    // one 512-thread workgroup per CU x 256 CUs
  if (s > nchunks) s = nchunks;
  if (s < 1) s = 1;
  return s;
}

static size_t w3x3s2_ws_bytes(int N, int M, int C, int Hs, int Ws) {
  return 256 + (size_t)w3x3s2_splits(M, C, N * Hs * (Ws / 32)) * 9 * M * C * sizeof(float);
}

static int run_w3x3s2(const float *small, const float *big, float *dW, int N, int C, int M, int Hs, int Ws, void *ws,
                      size_t ws_bytes, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {                        // 67 KB of LDS: above the 64 KB static limit, so dynamic + opt-in
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(igemm_w3x3s2_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)WS2_LDS_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(igemm_w3x3s2_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)WS2_LDS_BYTES);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(igemm_w3x3s2): %s", hipGetErrorString(e));
      return LSPS_E_HIP;
    }
    attr_set = true;
  }
  WS2Params p;
  memset(&p, 0, sizeof(p));
  p.Small = small;
  p.Big = big;
  p.N = N;
  p.M = M;
  p.C = C;
  p.Hs = Hs;
  p.Ws = Ws;
  p.qblocks = Ws / 32;
  p.nchunks = N * Hs * p.qblocks;
  const int splits = w3x3s2_splits(M, C, p.nchunks);
  p.chunks_per_split = ceil_div(p.nchunks, splits);
  const size_t need = w3x3s2_ws_bytes(N, M, C, Hs, Ws);
  if (need > ws_bytes) {
    set_error("wgrad workspace too small: need %zu, have %zu", need, ws_bytes);
    return LSPS_E_WS;
  }
  float *zero = (float *)ws;
  hipError_t e = hipMemsetAsync(zero, 0, 256, st);
  if (e != hipSuccess) {
    set_error("hipMemsetAsync: %s", hipGetErrorString(e));
    return LSPS_E_HIP;
  }
  p.zero = zero;
  p.part = (float *)((char *)ws + 256);
  if (g_math_mode == 1)
    hipLaunchKernelGGL(igemm_w3x3s2_kernel<true>, dim3(C / 64, M / 128, splits), dim3(512), WS2_LDS_BYTES, st, p);
  else
    hipLaunchKernelGGL(igemm_w3x3s2_kernel<false>, dim3(C / 64, M / 128, splits), dim3(512), WS2_LDS_BYTES, st, p);
  LSPS_CHECK_LAUNCH("igemm_w3x3s2");
  const long total = (long)M * C * 9;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, (const float *)p.part, dW, total,
                     splits);
  LSPS_CHECK_LAUNCH("reduce_partials");
  return 0;
}

#define C1W_BLOCKS 512
static bool c1_wgrad_ok(int Cb, int Hb, int Wb, int Cs, int Hs, int Ws, int R, int S, int st_, int pad) {
  if (!(Cb == 1 && Cs <= 64 && R * S <= 64 && (Ws == 32 || Ws == 64 || Ws == 128) && (st_ == 1 || st_ == 2))) return false;
  const int rb = 128 / Ws;
  return (Hs % rb) == 0 && (long)((rb - 1) * st_ + R) * (Wb + 2 * pad) <= C1W_XMAX && (long)Cs * Hs * Ws < (1L << 31);
}

static size_t c1_wgrad_ws_bytes(int Cs, int R, int S) { return (size_t)C1W_BLOCKS * Cs * R * S * sizeof(float) + 256; }

static int run_c1_wgrad(const float *small, const float *big, float *dW, int N, int Hb, int Wb, int Cs, int Hs, int Ws,
                        int R, int S, int st_, int pad, void *ws, size_t ws_bytes, hipStream_t st) {
  C1WParams p;
  memset(&p, 0, sizeof(p));
  p.X = big;
  p.DY = small;
  p.N = N;
  p.H = Hb;
  p.Wd = Wb;
  p.K = Cs;
  p.P = Hs;
  p.Q = Ws;
  p.R = R;
  p.S = S;
  p.stride = st_;
  p.pad = pad;
  p.LW = Wb + 2 * pad;
  p.RB = 128 / Ws;
  p.xrows = (p.RB - 1) * st_ + R;
  p.iters_total = N * (Hs / p.RB);
  int blocks = p.iters_total < C1W_BLOCKS ? p.iters_total : C1W_BLOCKS;
  p.iters_per_block = ceil_div(p.iters_total, blocks);
  blocks = ceil_div(p.iters_total, p.iters_per_block);
  if (c1_wgrad_ws_bytes(Cs, R, S) > ws_bytes) {
    set_error("wgrad workspace too small: need %zu, have %zu", c1_wgrad_ws_bytes(Cs, R, S), ws_bytes);
    return LSPS_E_WS;
  }
  p.part = (float *)ws;
  hipLaunchKernelGGL(c1_wgrad_kernel, dim3(blocks), dim3(256), 0, st, p);
  LSPS_CHECK_LAUNCH("c1_wgrad");
  const long nW = (long)Cs * R * S;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(ceil_div(nW, 256)), dim3(256), 0, st, (const float *)p.part, dW, nW, blocks);
  LSPS_CHECK_LAUNCH("reduce_partials");
  return 0;
}

// dW[m][(c,t)] = sum_{n,p,q} small[n][m][p][q] * big[n][c][p*st-pad+r][q*st-pad+s]
static int run_wgrad(const float *small, const float *big, float *dW, int N, int Cb, int Hb, int Wb, int Cs, int Hs,
                     int Ws, int R, int S, int st_, int pad, void *ws, size_t ws_bytes, hipStream_t st) {
#ifndef LSPS_NO_W3X3
  if (w3x3_ok(Cb, Hb, Wb, Cs, Hs, Ws, R, S, st_, pad)) return run_w3x3(small, big, dW, N, Cb, Hb, Cs, ws, ws_bytes, st);
#endif
#ifndef LSPS_NO_C1
  if (c1_wgrad_ok(Cb, Hb, Wb, Cs, Hs, Ws, R, S, st_, pad))
    return run_c1_wgrad(small, big, dW, N, Hb, Wb, Cs, Hs, Ws, R, S, st_, pad, ws, ws_bytes, st);
#endif
#ifndef LSPS_NO_W3X3S2
  if (w3x3s2_ok(Cb, Hb, Wb, Cs, Hs, Ws, R, S, st_, pad))
    return run_w3x3s2(small, big, dW, N, Cb, Cs, Hs, Ws, ws, ws_bytes, st);
#endif
  WParams p;
  memset(&p, 0, sizeof(p));
  TapList l;
  l.T = R * S;
  for (int r = 0; r < R; ++r)
    for (int s = 0; s < S; ++s) {
      l.dh[r * S + s] = r - pad;
      l.dw[r * S + s] = s - pad;
      l.idx[r * S + s] = r * S + s;
    }
  p.Small = small;
  p.Big = big;
  p.Cx = Cb;
  p.Hx = Hb;
  p.Wx = Wb;
  p.HxWx = Hb * Wb;
  p.PH = Hs;
  p.PW = Ws;
  p.P = Hs * Ws;
  p.NPIX = N * Hs * Ws;
  p.ist = st_;
  p.M = Cs;
  p.J = Cb * l.T;
  p.magicT = magic_for(l.T);
  p.nchunks = ceil_div(p.NPIX, BK_W);
  const int splits = wgrad_splits(p.M, p.J, p.nchunks);
  p.chunks_per_split = ceil_div(p.nchunks, splits);
  fill_taps(p.taps, l, Wb);
  const long nW = (long)p.M * p.J;
  const int Jp = (int)align_up((size_t)p.J, 128);
  const size_t head = align_up((size_t)Jp * sizeof(int2), 256) + 256;
  const size_t need = head + (splits > 1 ? (size_t)splits * nW * sizeof(float) : 0);
  if (need > ws_bytes) {
    set_error("wgrad workspace too small: need %zu, have %zu", need, ws_bytes);
    return LSPS_E_WS;
  }
  int2 *jtab = (int2 *)ws;
  float *zero = (float *)((char *)ws + head - 256);
  hipLaunchKernelGGL(build_jtab_kernel, dim3(ceil_div(Jp, 256)), dim3(256), 0, st, jtab, zero, p.J, Jp, l.T, p.magicT,
                     Hb * Wb, p.taps);
  LSPS_CHECK_LAUNCH("build_jtab");
  p.jtab = jtab;
  p.zero = zero;
  p.part = splits > 1 ? (float *)((char *)ws + head) : dW;
  const int tile = wgrad_tile(p.M, p.J);
  dim3 grid(ceil_div(p.J, tile), ceil_div(p.M, tile), splits);
  if (g_math_mode == 1) {
    if (tile == 64)
      hipLaunchKernelGGL((igemm_w_kernel<1, true>), grid, dim3(256), 0, st, p);
    else
      hipLaunchKernelGGL((igemm_w_kernel<2, true>), grid, dim3(256), 0, st, p);
  } else {
    if (tile == 64)
      hipLaunchKernelGGL((igemm_w_kernel<1, false>), grid, dim3(256), 0, st, p);
    else
      hipLaunchKernelGGL((igemm_w_kernel<2, false>), grid, dim3(256), 0, st, p);
  }
  LSPS_CHECK_LAUNCH("igemm_w");
  if (splits > 1) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(ceil_div(nW, 256)), dim3(256), 0, st, (const float *)p.part, dW, nW,
                       splits);
    LSPS_CHECK_LAUNCH("reduce_partials");
  }
  return 0;
}

#define BIAS_WS_BYTES ((size_t)1 << 20)   // head of every conv workspace: [S<=64][C<=4096] bias partials

static int run_bias_grad(const float *t, float *db, int N, int C, int HW, void *ws, size_t ws_bytes, hipStream_t st) {
  const long total = (long)N * HW;
  long S = (total + 32767) / 32768;           // >= 32 K elements per block
  if (S > 64) S = 64;
  if (S < 1) S = 1;
  if ((size_t)S * C * sizeof(float) > BIAS_WS_BYTES || ws_bytes < BIAS_WS_BYTES) {
    set_error("bias_grad: workspace too small (C=%d)", C);
    return LSPS_E_WS;
  }
  long slice = (total + S - 1) / S;
  slice = (slice + 3) / 4 * 4;
  float *part = (float *)ws;
  hipLaunchKernelGGL(bias_grad_partial_kernel, dim3(C, (int)S), dim3(256), 0, st, t, S == 1 ? db : part, N, C, HW, slice);
  LSPS_CHECK_LAUNCH("bias_grad_partial");
  if (S > 1) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, (const float *)part, db,
                       (long)C, (int)S);
    LSPS_CHECK_LAUNCH("bias_grad_reduce");
  }
  return 0;
}

static size_t conv_ws_bytes(int N, int Cb, int Hb, int Wb, int Cs, int Hs, int Ws, int R, int S, int st_) {
  const size_t fwd = packed_bytes(Cb, R * S, 1, Cs);
  const size_t tr = packed_bytes(Cs, R * S, st_ * st_, Cb);
  const int J = Cb * R * S;
  const int nchunks = ceil_div((long)N * Hs * Ws, BK_W);
  const int splits = wgrad_splits(Cs, J, nchunks);
  const size_t wg = align_up(align_up((size_t)J, 128) * sizeof(int2), 256) + 256 +
                    (splits > 1 ? (size_t)splits * Cs * J * sizeof(float) : 0);
  size_t m = fwd > tr ? fwd : tr;
  if (wg > m) m = wg;
  if (R == 3 && S == 3 && st_ == 1) {   // split-precision weight planes (math mode 2), either direction
    const size_t cmax = Cb > Cs ? Cb : Cs;
    const size_t sp = 256 + (align_up(cmax, 128) / 128) * (cmax / 8 + 1) * FS_ACHUNK * sizeof(unsigned short);
    if (sp > m) m = sp;
  }
  if (w3x3_ok(Cb, Hb, Wb, Cs, Hs, Ws, R, S, st_, R == 3 ? 1 : -1)) {
    const size_t w3 = w3x3_ws_bytes(N, Cs, Cb, Hb);
    if (w3 > m) m = w3;
  }
  if (w3x3s2_ok(Cb, Hb, Wb, Cs, Hs, Ws, R, S, st_, 1)) {
    const size_t w3 = w3x3s2_ws_bytes(N, Cs, Cb, Hs, Ws);
    if (w3 > m) m = w3;
  }
  if (Cb == 1 && Cs <= 64 && c1_wgrad_ws_bytes(Cs, R, S) > m) m = c1_wgrad_ws_bytes(Cs, R, S);
  return 2 * BIAS_WS_BYTES + m + ((size_t)8 << 20) + 1024;   // + 8 MiB: reduction-split partials of tiny forward problems
}

static bool conv_args_ok(int N, int C, int H, int W, int K, int R, int S, int stride, int pad) {
  return N > 0 && C > 0 && H > 0 && W > 0 && K > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0 &&
         R * S <= LSPS_MAXT && stride <= 4;
}



static bool pw1_ok(int Co, int R, int S, int stride, int pad, int outpad, long HW, const void *p0, const void *p1) {
  return Co == 1 && R == 1 && S == 1 && stride == 1 && pad == 0 && outpad == 0 && (HW % 4) == 0 &&
         (((uintptr_t)p0 | (uintptr_t)p1) & 15) == 0;
}

}  // namespace lsps

using namespace lsps;

extern "C" {

int lsps_version(void) { return LSPS_ABI_VERSION; }
const char *lsps_last_error(void) { return lsps::g_err; }

int lsps_set_math_mode(int mode) {
  if (mode < 0 || mode > 2) {
    set_error("set_math_mode: mode must be 0 (f32), 1 (bf16 MFMA operands) or 2 (f32 via 3-limb bf16 split)");
    return LSPS_E_ARG;
  }
  lsps::g_math_mode = mode;
  return 0;
}

int lsps_get_math_mode(void) { return lsps::g_math_mode; }

int lsps_device_cus(void) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return cus;
}

size_t lsps_conv2d_workspace_bytes(int N, int C, int H, int W, int K, int R, int S, int stride, int pad) {
  if (!conv_args_ok(N, C, H, W, K, R, S, stride, pad)) return 0;
  const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
  size_t extra = 0;
  if (C == 1 && R * S >= 9) extra = align_up((size_t)N * R * S * P * Q * sizeof(float), 256) + ((size_t)8 << 20);   // Z of the tap-GEMM dgrad
  return conv_ws_bytes(N, C, H, W, K, P, Q, R, S, stride) + extra;
}

int lsps_conv2d_fwd(const float *x, const float *w, const float *bias, float *y, int N, int C, int H, int W, int K,
                    int R, int S, int stride, int pad, int act, float slope, void *ws, size_t ws_bytes, void *stream) {
  (void)hipGetLastError();   // clear stale sticky errors left by other users of the runtime
  LSPS_CHECK_ARG(x && w && y && ws, "conv2d_fwd: null pointer");
  LSPS_CHECK_ARG(conv_args_ok(N, C, H, W, K, R, S, stride, pad), "conv2d_fwd: unsupported geometry");
  const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
  LSPS_CHECK_ARG(P > 0 && Q > 0, "conv2d_fwd: empty output");
  return run_forward_dir(x, w, bias, y, N, C, H, W, K, P, Q, R, S, stride, pad, (long)C * R * S, (long)R * S, act,
                         slope, ws, ws_bytes, (hipStream_t)stream);
}

int lsps_conv2d_dgrad(const float *dy, const float *w, float *dx, int N, int C, int H, int W, int K, int R, int S,
                      int stride, int pad, void *ws, size_t ws_bytes, void *stream) {
  (void)hipGetLastError();   // clear stale sticky errors left by other users of the runtime
  LSPS_CHECK_ARG(dy && w && dx && ws, "conv2d_dgrad: null pointer");
  LSPS_CHECK_ARG(conv_args_ok(N, C, H, W, K, R, S, stride, pad), "conv2d_dgrad: unsupported geometry");
  const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
  if (C == 1 && R * S >= 9) {
    // one input channel (the 7x7 stems: the generator's is reached by the cycle passes, the discriminator's by
    // gen_update): the direct MFMA tiling would be 31/32 padding, so compute Z[t] = W[:,t]^T dy as a 1x1 conv with
    // R*S output rows (M = 49 -> 77 % of a 64-row tile), then gather-sum the taps (a hand-written direct VALU
    // kernel measured 3x slower than this)
    const size_t zbytes = (size_t)N * R * S * P * Q * sizeof(float);
    if (zbytes + ((size_t)4 << 20) <= ws_bytes) {
      float *Z = (float *)ws;
      int rc = run_forward_dir(dy, w, nullptr, Z, N, K, P, Q, R * S, P, Q, 1, 1, 1, 0, 1L, (long)R * S, LSPS_ACT_NONE, 1.f,
                               (char *)ws + align_up(zbytes, 256), ws_bytes - align_up(zbytes, 256), (hipStream_t)stream);
      if (rc) return rc;
      const long total = (long)N * H * W;
      hipLaunchKernelGGL(col2im_c1_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float *)Z,
                         dx, N, H, W, P, Q, R, S, stride, pad);
      LSPS_CHECK_LAUNCH("col2im_c1");
      return 0;
    }
  }
  // out channel m = c: W[k][c][r][s] -> sm = R*S ; reduction channel k -> sc = C*R*S
  return run_transposed_dir(dy, w, nullptr, dx, N, C, H, W, K, P, Q, R, S, stride, pad, (long)R * S, (long)C * R * S,
                            LSPS_ACT_NONE, 1.f, ws, ws_bytes, (hipStream_t)stream);
}

int lsps_conv2d_wgrad(const float *x, const float *dy, float *dw, float *db, int N, int C, int H, int W, int K, int R,
                      int S, int stride, int pad, void *ws, size_t ws_bytes, void *stream) {
  (void)hipGetLastError();   // clear stale sticky errors left by other users of the runtime
  LSPS_CHECK_ARG(x && dy && dw && ws, "conv2d_wgrad: null pointer");
  LSPS_CHECK_ARG(conv_args_ok(N, C, H, W, K, R, S, stride, pad), "conv2d_wgrad: unsupported geometry");
  const int P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
  LSPS_CHECK_ARG(ws_bytes >= BIAS_WS_BYTES, "conv2d_wgrad: workspace too small");
  int rc = run_wgrad(dy, x, dw, N, C, H, W, K, P, Q, R, S, stride, pad, (char *)ws + BIAS_WS_BYTES,
                     ws_bytes - BIAS_WS_BYTES, (hipStream_t)stream);
  if (rc) return rc;
  if (db) return run_bias_grad(dy, db, N, K, P * Q, ws, ws_bytes, (hipStream_t)stream);
  return 0;
}

size_t lsps_convT2d_workspace_bytes(int N, int Ci, int H, int W, int Co, int R, int S, int stride, int pad,
                                    int outpad) {
  if (!conv_args_ok(N, Ci, H, W, Co, R, S, stride, pad)) return 0;
  const int Ho = (H - 1) * stride - 2 * pad + R + outpad, Wo = (W - 1) * stride - 2 * pad + S + outpad;
  return conv_ws_bytes(N, Co, Ho, Wo, Ci, H, W, R, S, stride);
}

int lsps_convT2d_fwd(const float *x, const float *w, const float *bias, float *y, int N, int Ci, int H, int W, int Co,
                     int R, int S, int stride, int pad, int outpad, int act, float slope, void *ws, size_t ws_bytes,
                     void *stream) {
  (void)hipGetLastError();   // clear stale sticky errors left by other users of the runtime
  LSPS_CHECK_ARG(x && w && y && ws, "convT2d_fwd: null pointer");
  LSPS_CHECK_ARG(conv_args_ok(N, Ci, H, W, Co, R, S, stride, pad) && outpad >= 0 && (outpad == 0 || outpad < stride),
                 "convT2d_fwd: unsupported geometry");
  const int Ho = (H - 1) * stride - 2 * pad + R + outpad, Wo = (W - 1) * stride - 2 * pad + S + outpad;
  LSPS_CHECK_ARG(Ho > 0 && Wo > 0, "convT2d_fwd: empty output");
  if (pw1_ok(Co, R, S, stride, pad, outpad, (long)H * W, x, y)) {
    const int HW4 = H * W / 4;
    hipLaunchKernelGGL(pw1_fwd_kernel, dim3(ceil_div((long)N * HW4, 256)), dim3(256), 0, (hipStream_t)stream, x, w, bias,
                       y, N, Ci, HW4, act, slope);
    LSPS_CHECK_LAUNCH("pw1_fwd");
    return 0;
  }
  // out channel m = co: W[ci][co][r][s] -> sm = R*S ; reduction channel ci -> sc = Co*R*S
  return run_transposed_dir(x, w, bias, y, N, Co, Ho, Wo, Ci, H, W, R, S, stride, pad, (long)R * S, (long)Co * R * S,
                            act, slope, ws, ws_bytes, (hipStream_t)stream);
}

int lsps_convT2d_dgrad(const float *dy, const float *w, float *dx, int N, int Ci, int H, int W, int Co, int R, int S,
                       int stride, int pad, int outpad, void *ws, size_t ws_bytes, void *stream) {
  (void)hipGetLastError();   // clear stale sticky errors left by other users of the runtime
  LSPS_CHECK_ARG(dy && w && dx && ws, "convT2d_dgrad: null pointer");
  LSPS_CHECK_ARG(conv_args_ok(N, Ci, H, W, Co, R, S, stride, pad), "convT2d_dgrad: unsupported geometry");
  const int Ho = (H - 1) * stride - 2 * pad + R + outpad, Wo = (W - 1) * stride - 2 * pad + S + outpad;
  if (pw1_ok(Co, R, S, stride, pad, outpad, (long)H * W, dy, dx)) {
    const int HW4 = H * W / 4;
    hipLaunchKernelGGL(pw1_dgrad_kernel, dim3(ceil_div((long)N * HW4, 256)), dim3(256), 0, (hipStream_t)stream, dy, w, dx,
                       N, Ci, HW4);
    LSPS_CHECK_LAUNCH("pw1_dgrad");
    return 0;
  }
  // dx[n][ci][h][w] = sum_{co,r,s} W[ci][co][r][s] dy[n][co][h*st-pad+r][w*st-pad+s]
  return run_forward_dir(dy, w, nullptr, dx, N, Co, Ho, Wo, Ci, H, W, R, S, stride, pad, (long)Co * R * S, (long)R * S,
                         LSPS_ACT_NONE, 1.f, ws, ws_bytes, (hipStream_t)stream);
}

int lsps_convT2d_wgrad(const float *x, const float *dy, float *dw, float *db, int N, int Ci, int H, int W, int Co,
                       int R, int S, int stride, int pad, int outpad, void *ws, size_t ws_bytes, void *stream) {
  (void)hipGetLastError();   // clear stale sticky errors left by other users of the runtime
  LSPS_CHECK_ARG(x && dy && dw && ws, "convT2d_wgrad: null pointer");
  LSPS_CHECK_ARG(conv_args_ok(N, Ci, H, W, Co, R, S, stride, pad), "convT2d_wgrad: unsupported geometry");
  const int Ho = (H - 1) * stride - 2 * pad + R + outpad, Wo = (W - 1) * stride - 2 * pad + S + outpad;
  LSPS_CHECK_ARG(ws_bytes >= 2 * BIAS_WS_BYTES, "convT2d_wgrad: workspace too small");
  int rc;
  if (pw1_ok(Co, R, S, stride, pad, outpad, (long)H * W, x, dy) && Ci <= 4096) {
    const int HW4 = H * W / 4;
    const long total4 = (long)N * HW4;
    long Sp = (total4 + 16383) / 16384;
    if (Sp > 64) Sp = 64;
    const long slice4 = (total4 + Sp - 1) / Sp;
    float *part = (float *)((char *)ws + BIAS_WS_BYTES);      // [Sp][Ci] <= 1 MiB
    hipLaunchKernelGGL(pw1_wgrad_kernel, dim3(Ci, (int)Sp), dim3(256), 0, (hipStream_t)stream, x, dy, part, N, Ci, HW4,
                       slice4);
    LSPS_CHECK_LAUNCH("pw1_wgrad");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(ceil_div(Ci, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)part, dw, (long)Ci, (int)Sp);
    LSPS_CHECK_LAUNCH("pw1_wgrad_reduce");
    rc = 0;
  } else {
    rc = run_wgrad(x, dy, dw, N, Co, Ho, Wo, Ci, H, W, R, S, stride, pad, (char *)ws + BIAS_WS_BYTES,
                   ws_bytes - BIAS_WS_BYTES, (hipStream_t)stream);
  }
  if (rc) return rc;
  if (db) return run_bias_grad(dy, db, N, Co, Ho * Wo, ws, ws_bytes, (hipStream_t)stream);
  return 0;
}

}  // extern "C"
