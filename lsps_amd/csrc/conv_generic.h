// Generic gather-GEMM kernels (any R x S / stride): weight packing, F kernel, W kernel, partial reductions, the 1x1 head kernels, bias gradient, col2im of the stem dgrad.
#ifndef LSPS_CONV_GENERIC_H
#define LSPS_CONV_GENERIC_H
#include "conv_types.h"

namespace lsps {

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
  const float *xb = p.X + (long)gn * p.xns + (long)ih0 * p.Wx + iw0;
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
    float *yb = p.Y + (long)n * p.yns + (long)(p.h0 + p.hs * ph) * p.Wy + (p.w0 + p.ws * pw);
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
                                                            int act, float slope, int PW, int HyWy, int Wy, int h0, int hs,
                                                            int w0, int ws, long yns) {
  // part[z][m][pix] -> Y[n][m][h0 + hs*ph][w0 + ws*pw]  (pix = n*P + ph*PW + pw; the output lattice of one parity class of
  // the transposed direction, or the whole image with h0 = w0 = 0, hs = ws = 1)
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;       // over [M][NPIX]
  if (idx >= (long)M * NPIX) return;
  const int m = (int)(idx / NPIX);
  const long pix = idx - (long)m * NPIX;
  float s = 0.f;
  for (int z = 0; z < ksplit; ++z) s += part[(long)z * M * NPIX + idx];
  if (bias) s += bias[m];
  const long n = pix / P;
  const int pp = (int)(pix - n * P);
  const int ph = pp / PW, pw = pp - ph * PW;
  y[n * yns + (long)m * HyWy + (long)(h0 + hs * ph) * Wy + w0 + ws * pw] = apply_act(s, act, slope);
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
      const float *sb = p.Small + (long)n * p.sns + pp;
      const float *xb = p.Big + (long)n * p.bns + (long)ih0 * p.Wx + iw0;
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

// y += a (fallback of lsps_conv2d_dgrad_acc for layers outside the fused 3x3 path)
__global__ __launch_bounds__(256) void add_inplace_kernel(float *__restrict__ y, const float *__restrict__ a, long n) {
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n && ((((uintptr_t)y | (uintptr_t)a) & 15) == 0)) {
      f32x4 v = *reinterpret_cast<f32x4 *>(y + i);
      const f32x4 b = *reinterpret_cast<const f32x4 *>(a + i);
      v += b;
      *reinterpret_cast<f32x4 *>(y + i) = v;
    } else {
      for (long k = i; k < n && k < i + 4; ++k) y[k] += a[k];
    }
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
// A block covers 1024 consecutive pixel quads of ONE image (thread t: quads t, t + 256, t + 512, t + 768), so every channel
// plane is read in 16 KB contiguous pieces (1 KB pieces before: 2.7 TB/s) with 16 independent loads in flight per thread.
template <int QPT>   // pixel quads per thread: 4 (16 KB contiguous per plane and block) once that still leaves >= 8 blocks per CU, else 2
__global__ __launch_bounds__(256) void pw1_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                      const float *__restrict__ b, float *__restrict__ y, int N, int C,
                                                      int HW4, int act, float slope) {
  const int bpi = (HW4 + 256 * QPT - 1) / (256 * QPT);         // blocks per image
  const int n = blockIdx.x / bpi, q0 = (blockIdx.x - n * bpi) * 256 * QPT + threadIdx.x;
  const f32x4 *xp = reinterpret_cast<const f32x4 *>(x) + (long)n * C * HW4;
  f32x4 acc[QPT];
  bool ok[QPT];
#pragma unroll
  for (int j = 0; j < QPT; ++j) {
    acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    ok[j] = q0 + 256 * j < HW4;
  }
#pragma unroll 4
  for (int c = 0; c < C; ++c) {
    const float wc = w[c];
#pragma unroll
    for (int j = 0; j < QPT; ++j)
      if (ok[j]) acc[j] += wc * xp[(long)c * HW4 + q0 + 256 * j];
  }
  const float bb = b ? b[0] : 0.f;
#pragma unroll
  for (int j = 0; j < QPT; ++j) {
    if (!ok[j]) continue;
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = apply_act(acc[j][e] + bb, act, slope);
    reinterpret_cast<f32x4 *>(y)[(long)n * HW4 + q0 + 256 * j] = r;
  }
}

// dx[n][c][pix] = w[c] * dy[n][pix]   (same block -> pixel mapping as the forward kernel: 16 KB contiguous stores per plane)
__global__ __launch_bounds__(256) void pw1_dgrad_kernel(const float *__restrict__ dy, const float *__restrict__ w,
                                                        float *__restrict__ dx, int N, int C, int HW4) {
  const int bpi = (HW4 + 1023) / 1024;
  const int n = blockIdx.x / bpi, q0 = (blockIdx.x - n * bpi) * 1024 + threadIdx.x;
  f32x4 g[4];
  bool ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ok[j] = q0 + 256 * j < HW4;
    g[j] = ok[j] ? reinterpret_cast<const f32x4 *>(dy)[(long)n * HW4 + q0 + 256 * j] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 *xp = reinterpret_cast<f32x4 *>(dx) + (long)n * C * HW4;
#pragma unroll 4
  for (int c = 0; c < C; ++c) {
    const float wc = w[c];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (ok[j]) xp[(long)c * HW4 + q0 + 256 * j] = wc * g[j];
  }
}

// dx[n][c][pix] = w[c] * dy[n][pix] * (y[n][c][pix] > 0 ? 1 : slope): the 1x1 head's input gradient with the LeakyReLU backward of
// the layer in front of it (saved output y, dx's shape) fused, and that layer's bias gradient as partial sums part[block][C]
// (C <= 64; grid = 2 blocks per image: 512 pixel quads each for a 128x128 map ... the quads of a block are strided by 256).
// wpart (nullable): partial sums of the head's OWN gradients from the same pass, wpart[block][c] = sum y[c][pix] * dy[pix] (y is its
// input) and wpart[block][C] = sum dy.
// blockIdx.y = slice of PW1_CS channels: with all 64 channels per thread the two running sums per channel are 128 registers
// (two waves per SIMD, 3.4 TB/s on the two 1 GB streams of a bs = 128 step); 16 channels per thread leave room for eight waves
// X3OUT (round 5): dx is written as a three-limb X3 tensor [N][3][C/8][HW][8] bf16 (csrc/x3s2.h) — the operand format of the
// three-limb transposed conv in front of the head, whose LeakyReLU backward this kernel applies; C % 16 == 0.
#define PW1_CS 16
template <bool X3OUT>
__global__ __launch_bounds__(256) void pw1_dgrad_act_kernel(const float *__restrict__ dy, const float *__restrict__ w,
                                                            const float *__restrict__ y, float *__restrict__ dx, float *__restrict__ part,
                                                            float *__restrict__ wpart, int C, int HW4, float slope) {
  __shared__ float red[4][PW1_CS];
  const int n = blockIdx.x >> 1, hb = blockIdx.x & 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.y * PW1_CS, nc = min(PW1_CS, C - c0);       // uniform
  const int q0 = hb * ((HW4 + 1) / 2), q1 = hb ? HW4 : (HW4 + 1) / 2;
  const f32x4 *yp = reinterpret_cast<const f32x4 *>(y) + ((long)n * C + c0) * HW4;
  f32x4 *xp = reinterpret_cast<f32x4 *>(dx) + ((long)n * C + c0) * HW4;
  const f32x4 *gp = reinterpret_cast<const f32x4 *>(dy) + (long)n * HW4;
  float s[PW1_CS], sw[PW1_CS], wv[PW1_CS], sd = 0.f;
#pragma unroll
  for (int c = 0; c < PW1_CS; ++c) {
    s[c] = sw[c] = 0.f;
    wv[c] = c0 + c < C ? w[c0 + c] : 0.f;
  }
  if (X3OUT) {
    // Thread = pixels 4 qb + j * 256 + tid (j < 4) of the block's current 256 quads [qb, qb + 256): every load (4 bytes per lane) and
    // every 16-byte limb store is DENSE over the lanes.  (With thread = one quad — 16-byte loads, but stores 64 bytes apart between
    // lanes, four instructions to fill a line — the kernel moved 3.8 TB/s.)  One channel group at a time: 32 loads in flight, then
    // 12 stores; all 16 channels of the slice at once need 233 registers.
    const long cgs = C >> 3, HW = (long)HW4 * 4, ls = cgs * HW * 8;
    const float *ys = y + ((long)n * C + c0) * HW, *gs = dy + (long)n * HW;
    const long pend = (long)q1 * 4;
    auto pass = [&](long pb, auto full_tag) {                   // pb: this lane's first pixel
      constexpr bool FULL = decltype(full_tag)::value;         // all 1024 pixels of the pass exist: no guards, constant offsets
      constexpr int NJ = FULL ? 4 : 1;                         // (the ragged last pass: one pixel per call, guarded)
      float g[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        g[j] = (FULL || pb + j * 256 < pend) ? gs[pb + j * 256] : 0.f;
        sd += g[j];
      }
#pragma unroll
      for (int grp = 0; grp < PW1_CS / 8; ++grp) {
        float yv[8][NJ], oo[8][NJ];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float *yk = ys + (long)(grp * 8 + k) * HW + pb;
#pragma unroll
          for (int j = 0; j < NJ; ++j) yv[k][j] = (FULL || pb + j * 256 < pend) ? yk[j * 256] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int c = grp * 8 + k;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const float v = wv[c] * g[j];
            oo[k][j] = yv[k][j] > 0.f ? v : v * slope;
          }
          if constexpr (FULL) {
            s[c] += (oo[k][0] + oo[k][1]) + (oo[k][NJ - 2] + oo[k][NJ - 1]);
            sw[c] += (yv[k][0] * g[0] + yv[k][1] * g[1]) + (yv[k][NJ - 2] * g[NJ - 2] + yv[k][NJ - 1] * g[NJ - 1]);
          } else {
            s[c] += oo[k][0];
            sw[c] += yv[k][0] * g[0];
          }
        }
        unsigned short *base = reinterpret_cast<unsigned short *>(dx) + (((long)n * 3 * cgs + ((c0 >> 3) + grp)) * HW + pb) * 8;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (FULL || pb + j * 256 < pend) {
            float v8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v8[k] = oo[k][j];
            bf16x8 h, m, l;
            split3(v8, h, m, l);
            *reinterpret_cast<bf16x8 *>(base + j * 2048) = h;
            *reinterpret_cast<bf16x8 *>(base + ls + j * 2048) = m;
            *reinterpret_cast<bf16x8 *>(base + 2 * ls + j * 2048) = l;
          }
        }
        __builtin_amdgcn_sched_barrier(0);          // the next group's 32 loads stay behind this group's stores
      }
    };
    for (int qb = q0; qb < q1; qb += 256) {
      if (qb + 256 <= q1) {                                     // uniform
        pass((long)qb * 4 + tid, std::true_type());
      } else {
        for (int j = 0; j < 4; ++j) pass((long)qb * 4 + j * 256 + tid, std::false_type());
      }
    }
  }
  for (int q = q0 + tid; !X3OUT && q < q1; q += 256) {
    const f32x4 g = gp[q];
    sd += (g[0] + g[1]) + (g[2] + g[3]);
    f32x4 yv[PW1_CS];
#pragma unroll
    for (int c = 0; c < PW1_CS; ++c)                   // all loads of the slice in flight before the first store
      yv[c] = yp[(long)(c < nc ? c : 0) * HW4 + q];    // (channels beyond C: a duplicate load, never stored)
#pragma unroll
    for (int c = 0; c < PW1_CS; ++c) {
      if (c < nc) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = wv[c] * g[e];
        o[e] = yv[c][e] > 0.f ? v : v * slope;
      }
      xp[(long)c * HW4 + q] = o;
      s[c] += (o[0] + o[1]) + (o[2] + o[3]);
      sw[c] += (yv[c][0] * g[0] + yv[c][1] * g[1]) + (yv[c][2] * g[2] + yv[c][3] * g[3]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < PW1_CS; ++c) {
    const float t = wave_sum(s[c]);
    if (lane == 0) red[wave][c] = t;
  }
  __syncthreads();
  if (tid < PW1_CS && c0 + tid < C) part[(long)blockIdx.x * C + c0 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
  if (wpart) {                                                  // uniform
    __syncthreads();
#pragma unroll
    for (int c = 0; c < PW1_CS; ++c) {
      const float t = wave_sum(sw[c]);
      if (lane == 0) red[wave][c] = t;
    }
    sd = wave_sum(sd);
    __syncthreads();
    if (tid < PW1_CS && c0 + tid < C)
      wpart[(long)blockIdx.x * (C + 1) + c0 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    __syncthreads();
    if (lane == 0) red[wave][0] = sd;
    __syncthreads();
    if (tid == 0 && blockIdx.y == 0) wpart[(long)blockIdx.x * (C + 1) + C] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
  }
}

// dW[c] = sum_b wpart[b][c] (c < C), db[0] = sum_b wpart[b][C]: one workgroup per column c (grid C + 1), fixed summation order
__global__ __launch_bounds__(256) void pw1_wsplit_reduce_kernel(const float *__restrict__ wpart, float *__restrict__ dW, float *__restrict__ db,
                                                                int C, int blocks) {
  __shared__ float red[4];
  const int c = blockIdx.x, tid = threadIdx.x;
  float s = 0.f;
  for (int b = tid; b < blocks; b += 256) s += wpart[(long)b * (C + 1) + c];
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid != 0) return;
  s = (red[0] + red[1]) + (red[2] + red[3]);
  if (c < C)
    dW[c] = s;
  else if (db)
    db[0] = s;
}

// part[s][c] = sum over slice s of (n,pix) of x[n][c][pix] * dy[n][pix]   (grid: C x S).  (image, quad) of a thread's
// element advance incrementally (one division at the start instead of one per element), four independent load pairs in flight.
__global__ __launch_bounds__(256) void pw1_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                        float *__restrict__ part, int N, int C, int HW4, long slice4) {
  __shared__ float red[4];
  const int c = blockIdx.x, sidx = blockIdx.y;
  const long total4 = (long)N * HW4;
  const long e0 = (long)sidx * slice4;
  long e1 = e0 + slice4;
  if (e1 > total4) e1 = total4;
  const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x), *g4 = reinterpret_cast<const f32x4 *>(dy);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  long e = e0 + threadIdx.x;
  long n = e / HW4;
  int q = (int)(e - n * HW4);
  for (; e + 768 < e1; e += 1024) {
    f32x4 a[4], g[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = x4[(n * C + c) * HW4 + q];
      g[j] = g4[e + 256 * j];
      q += 256;
      while (q >= HW4) {
        q -= HW4;
        ++n;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] += (a[j][0] * g[j][0] + a[j][1] * g[j][1]) + (a[j][2] * g[j][2] + a[j][3] * g[j][3]);
  }
  for (; e < e1; e += 256) {
    const f32x4 a = x4[(n * C + c) * HW4 + q], g = g4[e];
    s[0] += (a[0] * g[0] + a[1] * g[1]) + (a[2] * g[2] + a[3] * g[3]);
    q += 256;
    while (q >= HW4) {
      q -= HW4;
      ++n;
    }
  }
  const float t = block_sum_256((s[0] + s[1]) + (s[2] + s[3]), red);
  if (threadIdx.x == 0) part[(long)sidx * C + c] = t;
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

}  // namespace lsps
#endif
