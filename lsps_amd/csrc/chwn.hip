// 3x3 / stride 2 / pad 1 convs on SMALL feature maps (the discriminator trunk: 32x32 -> 16x16 -> 8x8 -> 4x4 -> 2x2 with
// 128 ... 2048 channels, reference lsps_nets.py:119-121 `_make_shared_net`, LeakyReLUConv2d = common_net.py:250-252) in a
// batch-innermost layout  [C][H][W][N]  ("CHWN").
//
// Why a layout: in NCHW a 32-pixel MFMA column group of a 2x2 ... 16x16 map straddles rows and images, so the generic kernel
// gathers every operand element with per-element address arithmetic — VALU work that does not hide behind an f32 MFMA
// (DESIGN.md 3.7): 64 ... 109 TFLOP/s.  With the batch innermost, the pixels of ONE (output position, tap) pair are N
// CONTIGUOUS floats per channel, and the conv is, per output position, a plain GEMM over (valid taps x channels):
//   forward   Y[k][p,q][n]   = act(b[k] + sum_{(r,s) valid} sum_c W[k][c][r][s] X[c][2p+r-1, 2q+s-1][n])
//   dgrad     dX[c][h,w][n]  = sum_{(r,s): (h+1-r, w+1-s) even, in range} sum_k W[k][c][r][s] dY[k][(h+1-r)/2, (w+1-s)/2][n]
//   wgrad     dW[k][c][r][s] = sum_{p,q valid} sum_n dY[k][p,q][n] X[c][2p+r-1, 2q+s-1][n]
// No gather, no multiply-by-zero: taps that fall into the padding are skipped per position (2x2 outputs: 25 of 36).
// Kernels: chwn_gemm_kernel (forward / dgrad: 128 x 128 tile, K-major operand tiles straight from global rows, optional
// split of the (tap, channel) reduction for small batches), chwn_wgrad_kernel (reduction over n: both operand tiles are
// read along n and stored [row][17] so that the MFMA operand reads are conflict-free), pack / transpose / reduce helpers.
#include "common.h"

namespace lsps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------------
// [R][S] -> [S][R] (NCHW <-> CHWN: R = N, S = C*H*W and back), 64 x 64 tiles through LDS
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose2d_kernel(const float *__restrict__ src, float *__restrict__ dst, long R, long S) {
  __shared__ float tile[64][65];
  const long r0 = (long)blockIdx.y * 64, s0 = (long)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const long r = r0 + ty + 4 * i, s = s0 + tx;
    tile[ty + 4 * i][tx] = (r < R && s < S) ? src[r * S + s] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const long s = s0 + ty + 4 * i, r = r0 + tx;
    if (r < R && s < S) dst[s * R + r] = tile[tx][ty + 4 * i];
  }
}

// W[K][C][9] -> Wf[9][C][K] (forward A operand: rows = input channel, columns = output channel)
__global__ __launch_bounds__(256) void chwn_pack_f_kernel(const float *__restrict__ W, float *__restrict__ Wf, int K, int C) {
  __shared__ float tile[64][65];
  const long J = (long)C * 9;                          // W as [K][J], j = c * 9 + t
  const long k0 = (long)blockIdx.y * 64, j0 = (long)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const long k = k0 + ty + 4 * i, j = j0 + tx;
    tile[ty + 4 * i][tx] = (k < K && j < J) ? W[k * J + j] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const long j = j0 + ty + 4 * i, k = k0 + tx;
    if (k < K && j < J) {
      const long c = j / 9, t = j - c * 9;
      Wf[(t * C + c) * K + k] = tile[tx][ty + 4 * i];
    }
  }
}

// W[K][C][9] -> Wt[9][K][C] (dgrad A operand: rows = output channel of the conv, columns = input channel); one block per k
__global__ __launch_bounds__(256) void chwn_pack_t_kernel(const float *__restrict__ W, float *__restrict__ Wt, int K, int C) {
  extern __shared__ float row[];                       // C * 9 floats
  const int k = blockIdx.x;
  const float *w = W + (long)k * C * 9;
  for (int i = threadIdx.x; i < C * 9; i += 256) row[i] = w[i];
  __syncthreads();
  for (int i = threadIdx.x; i < C * 9; i += 256) {
    const int t = i / C, c = i - t * C;
    Wt[((long)t * K + k) * C + c] = row[c * 9 + t];
  }
}

__device__ __forceinline__ float chwn_act(float v, int act, float slope) {
  if (act == LSPS_ACT_LRELU) return v > 0.f ? v : v * slope;
  if (act == LSPS_ACT_TANH) return tanhf(v);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward / dgrad:  Y[m][pos][n] = act(bias[m] + sum_{valid taps} sum_c A_tap[c][m] * B[c][pos_tap][n])
// ---------------------------------------------------------------------------------------------------------------------
#define CG_BK 16                   // 32 KB of LDS and <= 128 registers: four workgroups per CU hide the loads (BK 32, two per CU: slower)

struct CGParams {
  const float *A;                // mode 0: Wf [9][Rd][Md];  mode 1: Wt [9][Rd][Md]  (Rd = reduction channels, Md = output channels)
  const float *B;                // mode 0: x [Rd][Hb][Wb][N];  mode 1: dy [Rd][Hb][Wb][N]
  const float *bias;             // [Md] or null
  float *Y;                      // [Md][Ho][Wo][N]
  float *part;                   // [splits][Md][Ho][Wo][N] when splits > 1
  int Md, Rd, N;
  int Hb, Wb, Ho, Wo;
  int mode;                      // 0: forward (taps by output position), 1: dgrad (taps by input position)
  int act;
  float slope;
  int splits;
  int grouped;                   // mode 1 only: blockIdx.z / splits indexes a GROUP of output positions with ~4 taps in total
};

// dgrad (mode 1) computes one tile per INPUT position of the conv, and a position receives 1, 2, 2 or 4 taps by its (row, column)
// parity: one workgroup per position means workgroups of 1x ... 4x the work, and whenever the grid is about one round of the
// chip's resident slots (N = 128: 1024 workgroups) the kernel lasts as long as its 4-tap workgroups while the 1-tap ones leave
// their slots idle (tools/bench_chwn.py 128: dgrad 69 TFLOP/s next to forward's 96 - 111).  Grouped, a workgroup walks
//   type A: the odd-odd position of one 2x2 block (4 taps) | type B: its two mixed positions (2 + 2) | type C: the even-even
//   positions of four consecutive blocks (1 + 1 + 1 + 1; hence maps whose block count is a multiple of 4)
// one after the other: every workgroup does ~4 taps (border positions lose the taps that fall outside).
__device__ __forceinline__ int chwn_group_size(int g, int Ho, int Wo) {
  const int NB = (Ho >> 1) * (Wo >> 1);
  return g < NB ? 1 : (g < 2 * NB ? 2 : 4);
}
__device__ __forceinline__ int chwn_group_pos(int g, int i, int Ho, int Wo) {       // i-th position of group g (all uniform)
  const int bw = Wo >> 1, NB = (Ho >> 1) * bw;
  if (g < NB) return (2 * (g / bw) + 1) * Wo + 2 * (g % bw) + 1;
  if (g < 2 * NB) {
    const int b = g - NB, y = 2 * (b / bw), x = 2 * (b % bw);
    return i == 0 ? y * Wo + x + 1 : (y + 1) * Wo + x;
  }
  const int b = 4 * (g - 2 * NB) + i;
  return (2 * (b / bw)) * Wo + 2 * (b % bw);
}

template <bool GROUPED>
__global__ __launch_bounds__(256, 4) void chwn_gemm_kernel(CGParams p) {
  __shared__ __attribute__((aligned(16))) float As[2][CG_BK][128], Bs[2][CG_BK][128];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1, l31 = lane & 31, half = lane >> 5;
  const int n0 = blockIdx.x * 128, m0 = blockIdx.y * 128;
  const int zi = blockIdx.z / p.splits, sp = blockIdx.z - zi * p.splits;
  const int npos = GROUPED ? chwn_group_size(zi, p.Ho, p.Wo) : 1;
#pragma clang loop unroll(disable)
  for (int gi = 0; gi < npos; ++gi) {
  const int pos = GROUPED ? __builtin_amdgcn_readfirstlane(chwn_group_pos(zi, gi, p.Ho, p.Wo)) : zi;
  const int oh = pos / p.Wo, ow = pos - oh * p.Wo;

  // valid taps of this position, packed 4 bits each (uniform)
  unsigned long taps = 0;
  int ntap = 0;
  for (int t = 0; t < 9; ++t) {
    const int r = t / 3, s = t - 3 * r;
    bool ok;
    if (p.mode == 0) {
      const int ih = 2 * oh + r - 1, iw = 2 * ow + s - 1;
      ok = ih >= 0 && ih < p.Hb && iw >= 0 && iw < p.Wb;
    } else {
      const int a = oh + 1 - r, b = ow + 1 - s;
      ok = a >= 0 && b >= 0 && !(a & 1) && !(b & 1) && (a >> 1) < p.Hb && (b >> 1) < p.Wb;
    }
    if (ok) taps |= (unsigned long)t << (4 * ntap++);
  }
  const int RC = p.Rd / CG_BK;
  const int total = ntap * RC;
  const int it0 = (int)((long)total * sp / p.splits), it1 = (int)((long)total * (sp + 1) / p.splits);

  // staging: per operand CG_BK rows x 32 float4 = CG_BK / 8 per thread (rows tid/32 + 8 i).  Everything that changes from
  // one chunk to the next is uniform: a (tap, chunk) cursor with byte pointers advanced incrementally (no division, no 64-bit
  // lane arithmetic in the loop: those instructions cost their full issue time beside an f32 MFMA); the lanes keep ONE 32-bit
  // offset per operand and load through buffer descriptors.  Batch columns >= N of the last n tile are never stored, so
  // their loads are only redirected to a valid column, not zeroed.
  const int srow = tid >> 5, scol = (tid & 31) * 4;
  const long ldb = (long)p.Hb * p.Wb * p.N;
  const unsigned a_vo = (unsigned)(srow * p.Md + scol) * 4u;
  const unsigned b_vo = (unsigned)((long)srow * ldb + min(n0 + scol, p.N - 4)) * 4u;
  const int a_row8 = 8 * p.Md * 4, b_row8 = (int)(8 * ldb * 4);           // rows + 8: uniform byte offsets
  const long a_step = (long)CG_BK * p.Md * 4, b_step = (long)CG_BK * ldb * 4;
  int cur_ti = 0, cur_ch = 0;
  const char *a_ptr = nullptr, *b_ptr = nullptr;
  auto seek = [&](int ti, int ch) {                    // byte pointers of chunk `ch` of the ti-th valid tap
    const int t = (int)((taps >> (4 * ti)) & 15), r = t / 3, s = t - 3 * r;
    const int pb = p.mode == 0 ? (2 * oh + r - 1) * p.Wb + 2 * ow + s - 1 : ((oh + 1 - r) >> 1) * p.Wb + ((ow + 1 - s) >> 1);
    a_ptr = reinterpret_cast<const char *>(p.A + ((long)t * p.Rd * p.Md + m0)) + ch * a_step;
    b_ptr = reinterpret_cast<const char *>(p.B + (long)pb * p.N) + ch * b_step;
    cur_ti = ti;
    cur_ch = ch;
  };
  f32x4 ra[CG_BK / 8], rb[CG_BK / 8];
  auto load_tiles = [&]() {                            // the chunk under the cursor, then advance the cursor
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a_ptr), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(b_ptr), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < CG_BK / 8; ++i) {
      ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, a_vo, i * a_row8, 0));
      rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, b_vo, i * b_row8, 0));
    }
    a_ptr += a_step;
    b_ptr += b_step;
    if (++cur_ch == RC && cur_ti + 1 < ntap) seek(cur_ti + 1, 0);
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < CG_BK / 8; ++i) {
      *reinterpret_cast<f32x4 *>(&As[buf][srow + 8 * i][scol]) = ra[i];
      *reinterpret_cast<f32x4 *>(&Bs[buf][srow + 8 * i][scol]) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (it0 < it1) {
    seek(it0 / RC, it0 % RC);
    load_tiles();
    store_tiles(0);
    __syncthreads();
    int buf = 0;
    for (int it = it0; it < it1; ++it) {
      if (it + 1 < it1) load_tiles();
      // operands of k-pair kk + 1 are read BEFORE the MFMAs of kk are issued (order pinned: left alone, the compiler reads,
      // waits out the LDS latency and only then issues the four MFMAs of every pair)
      const float *as = &As[buf][half][wm * 64 + l31], *bs = &Bs[buf][half][wn * 64 + l31];
      float a0 = as[0], a1 = as[32], b0 = bs[0], b1 = bs[32];
#pragma unroll
      for (int kk = 0; kk < CG_BK / 2; ++kk) {
        float a0n = 0.f, a1n = 0.f, b0n = 0.f, b1n = 0.f;
        if (kk + 1 < CG_BK / 2) {
          a0n = as[(kk + 1) * 256];
          a1n = as[(kk + 1) * 256 + 32];
          b0n = bs[(kk + 1) * 256];
          b1n = bs[(kk + 1) * 256 + 32];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a0 = a0n, a1 = a1n, b0 = b0n, b1 = b1n;
      }
      if (it + 1 < it1) store_tiles(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  // epilogue: register r of acc[i][j] = row m0 + wm*64 + i*32 + (r & 3) + 8 (r >> 2) + 4 half, column n0 + wn*64 + j*32 + l31
  long plane = (long)p.Ho * p.Wo * p.N;
  // (inside the position loop the per-lane output addresses are loop-invariant up to `pos`; hoisted, their 32 64-bit values
  // live through the main loop and spill — keep them a post-loop computation, as in the one-position kernel)
  if (GROUPED) asm volatile("" : "+s"(plane));
  float *out = p.splits > 1 ? p.part + (long)sp * p.Md * plane : p.Y;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + l31;
      if (n >= p.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[i][j][r];
        if (p.splits == 1) v = chwn_act(v + (p.bias ? p.bias[m] : 0.f), p.act, p.slope);
        out[(long)m * plane + (long)pos * p.N + n] = v;
      }
    }
  }                                                    // next position of the group (the loop above ended on a barrier)
}

// Y = act(bias + sum of the reduction splits)
__global__ __launch_bounds__(256) void chwn_split_reduce_kernel(const float *__restrict__ part, const float *__restrict__ bias,
                                                                float *__restrict__ Y, long total, long plane, int splits,
                                                                int act, float slope) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= total) return;
  f32x4 v = *reinterpret_cast<const f32x4 *>(part + i);
  for (int s = 1; s < splits; ++s) v += *reinterpret_cast<const f32x4 *>(part + (long)s * total + i);
  const float b = bias ? bias[i / plane] : 0.f;        // plane % 4 == 0: the four elements share their channel
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = chwn_act(v[e] + b, act, slope);
  *reinterpret_cast<f32x4 *>(Y + i) = v;
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient:  part[sp][t][k][c] = sum over this split's (valid position, n chunk) of dY[k][p,q][n] * X[c][ih,iw][n]
// ---------------------------------------------------------------------------------------------------------------------
#define CW_BK 16
#define CW_LD 17
#define CW_LDS_BYTES (4 * 128 * CW_LD * 4)     // 34.8 KB: two operands, double-buffered

struct CWParams {
  const float *DY, *X;
  float *part;                   // [splits][9][K][C]
  int K, C, N, H, W, P, Q;
  int splits;
};

__global__ __launch_bounds__(256, 4) void chwn_wgrad_kernel(CWParams p) {
  extern __shared__ float cw_lds[];
  float(*As)[128 * CW_LD] = reinterpret_cast<float(*)[128 * CW_LD]>(cw_lds);
  float(*Bs)[128 * CW_LD] = reinterpret_cast<float(*)[128 * CW_LD]>(cw_lds + 2 * 128 * CW_LD);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1, l31 = lane & 31, half = lane >> 5;
  const int c0 = blockIdx.x * 128, k0 = blockIdx.y * 128;
  const int t = blockIdx.z / p.splits, sp = blockIdx.z - t * p.splits;
  const int r = t / 3, s = t - 3 * r;
  // output positions whose tap (r, s) reads a real input pixel: a rectangle
  const int pl = r == 0 ? 1 : 0, ql = s == 0 ? 1 : 0;
  const int ph = min(p.P - 1, (p.H - r) / 2), qh = min(p.Q - 1, (p.W - s) / 2);      // 2p + r - 1 <= H - 1
  const int np = max(0, ph - pl + 1), nq = max(0, qh - ql + 1);
  const int NC = (p.N + CW_BK - 1) / CW_BK;
  const int total = np * nq * NC;
  const int it0 = (int)((long)total * sp / p.splits), it1 = (int)((long)total * (sp + 1) / p.splits);

  // staging: per operand 128 rows x CW_BK floats: thread -> row tid/2, CW_BK / 2 floats at (tid & 1) * CW_BK / 2.  As in
  // chwn_gemm_kernel everything that changes per chunk is a uniform cursor (position, n chunk) with incrementally advanced
  // byte pointers; a lane keeps one 32-bit offset per operand.  Here the batch IS the reduction: columns >= N must
  // contribute zeros (only when N is not a multiple of the chunk: a uniform flag).
  const int srow = tid >> 1, scol = (tid & 1) * (CW_BK / 2);
  const long lda = (long)p.P * p.Q * p.N, ldb = (long)p.H * p.W * p.N;
  const unsigned a_vo = (unsigned)((long)(k0 + srow) * lda + scol) * 4u, b_vo = (unsigned)((long)(c0 + srow) * ldb + scol) * 4u;
  const bool tail = (p.N % CW_BK) != 0;
  int cur_pi = 0, cur_nc = 0;
  const char *a_ptr = nullptr, *b_ptr = nullptr;
  auto seek = [&](int pi, int nc) {
    const int pp = pl + pi / nq, qq = ql + pi % nq;
    a_ptr = reinterpret_cast<const char *>(p.DY + (long)(pp * p.Q + qq) * p.N + nc * CW_BK);
    b_ptr = reinterpret_cast<const char *>(p.X + (long)((2 * pp + r - 1) * p.W + 2 * qq + s - 1) * p.N + nc * CW_BK);
    cur_pi = pi;
    cur_nc = nc;
  };
  f32x4 ra[CW_BK / 8], rb[CW_BK / 8];
  auto load_tiles = [&]() {
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a_ptr), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(b_ptr), 0, 0x7fffffff, 0x00020000);
    if (tail && cur_nc == NC - 1) {                    // last n chunk of a ragged batch: in-range float4s only
      const int n = cur_nc * CW_BK + scol;
#pragma unroll
      for (int i = 0; i < CW_BK / 8; ++i) {
        const bool in = n + 4 * i < p.N;
        ra[i] = in ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, a_vo + 16 * i, 0, 0)) : f32x4{0.f, 0.f, 0.f, 0.f};
        rb[i] = in ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, b_vo + 16 * i, 0, 0)) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    } else {
#pragma unroll
      for (int i = 0; i < CW_BK / 8; ++i) {
        ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, a_vo + 16 * i, 0, 0));
        rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, b_vo + 16 * i, 0, 0));
      }
    }
    a_ptr += CW_BK * 4;
    b_ptr += CW_BK * 4;
    if (++cur_nc == NC && cur_pi + 1 < np * nq) seek(cur_pi + 1, 0);
  };
  auto store_tiles = [&](int buf) {
    float *a = &As[buf][srow * CW_LD + scol], *b = &Bs[buf][srow * CW_LD + scol];
#pragma unroll
    for (int i = 0; i < CW_BK / 8; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[4 * i + e] = ra[i][e];
        b[4 * i + e] = rb[i][e];
      }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  if (it0 < it1) {
    seek(it0 / NC, it0 % NC);
    load_tiles();
    store_tiles(0);
    __syncthreads();
    int buf = 0;
    for (int it = it0; it < it1; ++it) {
      if (it + 1 < it1) load_tiles();
      const float *as = &As[buf][(wm * 64 + l31) * CW_LD + half], *bs = &Bs[buf][(wn * 64 + l31) * CW_LD + half];
      float a0 = as[0], a1 = as[32 * CW_LD], b0 = bs[0], b1 = bs[32 * CW_LD];
#pragma unroll
      for (int kk = 0; kk < CW_BK / 2; ++kk) {       // reads of k-pair kk + 1 ahead of the MFMAs of kk (see chwn_gemm_kernel)
        float a0n = 0.f, a1n = 0.f, b0n = 0.f, b1n = 0.f;
        if (kk + 1 < CW_BK / 2) {
          a0n = as[2 * kk + 2];
          a1n = as[2 * kk + 2 + 32 * CW_LD];
          b0n = bs[2 * kk + 2];
          b1n = bs[2 * kk + 2 + 32 * CW_LD];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a0 = a0n, a1 = a1n, b0 = b0n, b1 = b1n;
      }
      if (it + 1 < it1) store_tiles(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }
  float *out = p.part + ((long)sp * 9 + t) * p.K * p.C;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = c0 + wn * 64 + j * 32 + l31;
      if (c >= p.C) continue;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int k = k0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
        if (k < p.K) out[(long)k * p.C + c] = acc[i][j][q];
      }
    }
}

// dW[k][c][t] = sum over splits of part[sp][t][k][c]
__global__ __launch_bounds__(256) void chwn_wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ dW, long KC,
                                                                int splits) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= KC) return;
  float w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = 0.f;
  for (int s = 0; s < splits; ++s)
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] += part[((long)s * 9 + t) * KC + i];
#pragma unroll
  for (int t = 0; t < 9; ++t) dW[i * 9 + t] = w[t];
}

static int chwn_device_cus() {
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cus;
}

static bool chwn_geom_ok(int N, int C, int H, int W, int K) {
  return N > 0 && (N % 4) == 0 && C > 0 && (C % 128) == 0 && K > 0 && (K % 128) == 0 && H >= 2 && W >= 2 && (H % 2) == 0 &&
         (W % 2) == 0 && (long)128 * H * W * N * 4 < (1L << 32);     // 32-bit byte offsets of a lane inside a 128-row tile
}

// reduction splits of the forward / dgrad GEMM: fill the chip twice over when the batch alone does not
static int chwn_gemm_splits(int Md, int Rd, int N, int npos) {
  const long wgs = (long)ceil_div(N, 128) * (Md / 128) * npos;
  const long want = 2L * chwn_device_cus();
  if (wgs >= want) return 1;
  long s = (want + wgs - 1) / wgs;
  const long iters = (long)Rd / CG_BK * 4;             // ~4 taps x channel chunks per position
  if (s > iters / 4) s = iters / 4;
  if (s > 16) s = 16;
  return s < 1 ? 1 : (int)s;
}
static int chwn_wgrad_splits(int K, int C, int N, int P, int Q) {
  const long wgs = (long)ceil_div(C, 128) * ceil_div(K, 128) * 9;
  const long want = 2L * chwn_device_cus();
  if (wgs >= want) return 1;
  long s = (want + wgs - 1) / wgs;
  const long iters = (long)P * Q * ceil_div(N, CW_BK);
  if (s > iters / 4) s = iters / 4;
  if (s > 64) s = 64;
  return s < 1 ? 1 : (int)s;
}

}  // namespace lsps

using namespace lsps;

extern "C" {

int lsps_transpose2d(const float *src, float *dst, long R, long S, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(src && dst && R > 0 && S > 0 && src != dst, "transpose2d: bad argument");
  LSPS_CHECK_ARG(ceil_div(R, 64) <= 65535, "transpose2d: more than 65535 x 64 rows");
  hipLaunchKernelGGL(transpose2d_kernel, dim3(ceil_div(S, 64), ceil_div(R, 64)), dim3(256), 0, (hipStream_t)stream, src, dst, R, S);
  LSPS_CHECK_LAUNCH("transpose2d");
  return 0;
}

size_t lsps_conv3x3s2_chwn_workspace_bytes(int N, int C, int H, int W, int K) {
  if (!chwn_geom_ok(N, C, H, W, K)) return 0;
  const int P = H / 2, Q = W / 2;
  const size_t pack = (size_t)9 * C * K * sizeof(float);
  size_t fs = (size_t)chwn_gemm_splits(K, C, N, P * Q), ds = (size_t)chwn_gemm_splits(C, K, N, H * W);
  const size_t dsg = (size_t)chwn_gemm_splits(C, K, N, (H * W >= 64 && ((H / 2) * (W / 2)) % 4 == 0) ? 9 * (H / 2) * (W / 2) / 4 : H * W);   // dgrad, grouped positions
  if (dsg > ds) ds = dsg;
  const size_t fpart = fs > 1 ? fs * K * P * Q * N * sizeof(float) : 0, dpart = ds > 1 ? ds * C * H * W * N * sizeof(float) : 0;
  const size_t wpart = (size_t)chwn_wgrad_splits(K, C, N, P, Q) * 9 * K * C * sizeof(float);
  size_t m = pack + (fpart > dpart ? fpart : dpart);
  if (wpart > m) m = wpart;
  return m + 1024;
}

static int chwn_run_gemm(const float *A, const float *B, const float *bias, float *Y, int Md, int Rd, int N, int Hb, int Wb, int Ho,
                         int Wo, int mode, int act, float slope, float *part, hipStream_t st) {
  CGParams p;
  p.A = A;
  p.B = B;
  p.bias = bias;
  p.Y = Y;
  p.part = part;
  p.Md = Md;
  p.Rd = Rd;
  p.N = N;
  p.Hb = Hb;
  p.Wb = Wb;
  p.Ho = Ho;
  p.Wo = Wo;
  p.mode = mode;
  p.act = act;
  p.slope = slope;
  const int group_env = opts().chwn_group;
  // dgrad: positions grouped by tap count (see chwn_group_positions); maps of at least 4 x 4 positions (NB % 4 == 0)
  // (measured, tools/bench_chwn.py, profiles/r4s_chwn_grouped_dgrad.txt: 8x8 ... 32x32 input maps gain 8 - 15 % at N = 128 and 768;
  // on the 4x4 map — 9 groups — the lower occupancy costs more than the balance gives: per-position workgroups stay)
  p.grouped = (mode == 1 && group_env && (Ho % 2) == 0 && (Wo % 2) == 0 && Ho * Wo >= 64 && (((Ho / 2) * (Wo / 2)) % 4) == 0) ? 1 : 0;
  const int nz = p.grouped ? 9 * (Ho / 2) * (Wo / 2) / 4 : Ho * Wo;
  p.splits = chwn_gemm_splits(Md, Rd, N, nz);
  if ((long)nz * p.splits > 65535) {
    set_error("conv3x3s2_chwn: %d x %d output positions x %d splits exceed the grid", Ho, Wo, p.splits);
    return LSPS_E_ARG;
  }
  if (p.grouped)
    hipLaunchKernelGGL(chwn_gemm_kernel<true>, dim3(ceil_div(N, 128), Md / 128, nz * p.splits), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(chwn_gemm_kernel<false>, dim3(ceil_div(N, 128), Md / 128, nz * p.splits), dim3(256), 0, st, p);
  LSPS_CHECK_LAUNCH("chwn_gemm");
  if (p.splits > 1) {
    const long plane = (long)Ho * Wo * N, total = (long)Md * plane;
    hipLaunchKernelGGL(chwn_split_reduce_kernel, dim3(ceil_div(total / 4, 256)), dim3(256), 0, st, (const float *)part, bias, Y,
                       total, plane, p.splits, act, slope);
    LSPS_CHECK_LAUNCH("chwn_split_reduce");
  }
  return 0;
}

int lsps_conv3x3s2_chwn_fwd(const float *x, const float *w, const float *bias, float *y, int N, int C, int H, int W, int K, int act,
                            float slope, void *ws, size_t ws_bytes, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(x && w && y && ws, "conv3x3s2_chwn_fwd: null pointer");
  LSPS_CHECK_ARG(chwn_geom_ok(N, C, H, W, K), "conv3x3s2_chwn_fwd: unsupported geometry (need N %% 4, C %% 128, K %% 128, even H, W)");
  LSPS_CHECK_ARG(ws_bytes >= lsps_conv3x3s2_chwn_workspace_bytes(N, C, H, W, K), "conv3x3s2_chwn_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float *Wf = (float *)ws, *part = Wf + (size_t)9 * C * K;
  hipLaunchKernelGGL(chwn_pack_f_kernel, dim3(ceil_div((long)C * 9, 64), ceil_div(K, 64)), dim3(256), 0, st, w, Wf, K, C);
  LSPS_CHECK_LAUNCH("chwn_pack_f");
  return chwn_run_gemm(Wf, x, bias, y, K, C, N, H, W, H / 2, W / 2, 0, act, slope, part, st);
}

int lsps_conv3x3s2_chwn_dgrad(const float *dy, const float *w, float *dx, int N, int C, int H, int W, int K, void *ws,
                              size_t ws_bytes, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(dy && w && dx && ws, "conv3x3s2_chwn_dgrad: null pointer");
  LSPS_CHECK_ARG(chwn_geom_ok(N, C, H, W, K), "conv3x3s2_chwn_dgrad: unsupported geometry");
  LSPS_CHECK_ARG(ws_bytes >= lsps_conv3x3s2_chwn_workspace_bytes(N, C, H, W, K), "conv3x3s2_chwn_dgrad: workspace too small");
  LSPS_CHECK_ARG((size_t)C * 9 * sizeof(float) <= 160 * 1024, "conv3x3s2_chwn_dgrad: more than 4551 input channels");
  hipStream_t st = (hipStream_t)stream;
  float *Wt = (float *)ws, *part = Wt + (size_t)9 * C * K;
  if (int rc = lds_optin(reinterpret_cast<const void *>(chwn_pack_t_kernel), 160 * 1024, "chwn_pack_t")) return rc;
  hipLaunchKernelGGL(chwn_pack_t_kernel, dim3(K), dim3(256), (size_t)C * 9 * sizeof(float), st, w, Wt, K, C);
  LSPS_CHECK_LAUNCH("chwn_pack_t");
  return chwn_run_gemm(Wt, dy, nullptr, dx, C, K, N, H / 2, W / 2, H, W, 1, LSPS_ACT_NONE, 1.f, part, st);
}

int lsps_conv3x3s2_chwn_wgrad(const float *x, const float *dy, float *dw, int N, int C, int H, int W, int K, void *ws,
                              size_t ws_bytes, void *stream) {
  (void)hipGetLastError();
  LSPS_CHECK_ARG(x && dy && dw && ws, "conv3x3s2_chwn_wgrad: null pointer");
  LSPS_CHECK_ARG(chwn_geom_ok(N, C, H, W, K), "conv3x3s2_chwn_wgrad: unsupported geometry");
  LSPS_CHECK_ARG(ws_bytes >= lsps_conv3x3s2_chwn_workspace_bytes(N, C, H, W, K), "conv3x3s2_chwn_wgrad: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  CWParams p;
  p.DY = dy;
  p.X = x;
  p.part = (float *)ws;
  p.K = K;
  p.C = C;
  p.N = N;
  p.H = H;
  p.W = W;
  p.P = H / 2;
  p.Q = W / 2;
  p.splits = chwn_wgrad_splits(K, C, N, p.P, p.Q);
  if (int rc = lds_optin(reinterpret_cast<const void *>(chwn_wgrad_kernel), (int)CW_LDS_BYTES, "chwn_wgrad")) return rc;
  hipLaunchKernelGGL(chwn_wgrad_kernel, dim3(ceil_div(C, 128), ceil_div(K, 128), 9 * p.splits), dim3(256), CW_LDS_BYTES, st, p);
  LSPS_CHECK_LAUNCH("chwn_wgrad");
  hipLaunchKernelGGL(chwn_wgrad_reduce_kernel, dim3(ceil_div((long)K * C, 256)), dim3(256), 0, st, (const float *)p.part, dw,
                     (long)K * C, p.splits);
  LSPS_CHECK_LAUNCH("chwn_wgrad_reduce");
  return 0;
}

}  // extern "C"
