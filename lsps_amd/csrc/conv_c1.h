// Single-input-channel (7x7 stem) kernels: forward and weight gradient.
#ifndef LSPS_CONV_C1_H
#define LSPS_CONV_C1_H
#include "conv_types.h"

namespace lsps {

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
  int out_c8;                    // 1: Y is bf16 in the channel-group layout [N][K/8][P][Q][8] (c8conv.h; bf16 math mode);
                                 // 2: Y is a three-limb X3 tensor [N][3][K/8][P][Q][8] (x3s2.h; f32 math mode, K == 64)
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
  // Epilogue kept off the datapath the MFMAs need (round 2 PMC: 2.65 VALU per MFMA here, the matrix pipe 61 % busy): the
  // bias lives in registers (the accumulators start from it), LeakyReLU with 0 <= slope <= 1 is max(v, slope v), and the
  // stores go through a buffer descriptor: lane offset = pixel (+ the lane half's four channels), channel = scalar offset.
  const bool fast = full && (p.act == LSPS_ACT_NONE || (lrelu && p.slope >= 0.f && p.slope <= 1.f)) && 64 * PQ * 4 < (1L << 31);
  float bias_r[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[i][r] = bl[i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.Y + ((long)n * p.K + m0) * PQ, 0, 0x7fffffff, 0x00020000);
  const int pq4 = (int)PQ * 4;
  for (int seg = wave; seg < nseg; seg += 4) {
    const int pr = seg / qblocks, q0 = (seg - pr * qblocks) * 32;
    const float *Bp = xs + pr * p.stride * p.LW + (q0 + l31) * p.stride;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = fast ? bias_r[i][r] : 0.f;
#pragma unroll
    for (int ks = 0; ks < C1_KS; ++ks) {
      const float b = Bp[boff[ks]];
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks][i], b, acc[i], 0, 0, 0);
    }
    if (fast && p.out_c8) {                                  // wave-uniform.  A register quad = 4 consecutive channels = 8 bytes
      const bool x3 = p.out_c8 == 2;                         // three limb planes per image (x = hi + mid + lo exactly)
      const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<unsigned short *>(p.Y) + ((long)n * (x3 ? 3 : 1) * p.K + m0) * PQ, 0, 0x7fffffff, 0x00020000);
      const unsigned vo = (unsigned)(((p0 + pr) * p.Q + q0 + l31) * 16 + half * 8);
      const int limb_bytes = p.K * (int)PQ * 2;
      typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          bf16x4 v;
          float y4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a = acc[i][rq * 4 + e];
            y4[e] = lrelu ? fmaxf(a, a * p.slope) : a;
            v[e] = (__bf16)y4[e];
          }
          const int so = (i * 4 + rq) * (int)PQ * 16;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), crs, vo, so, 0);
          if (x3) {                                          // wave-uniform
            bf16x4 vm, vl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float r1 = y4[e] - (float)v[e];
              vm[e] = (__bf16)r1;
              vl[e] = (__bf16)(r1 - (float)vm[e]);
            }
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, vm), crs, vo, so + limb_bytes, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, vl), crs, vo, so + 2 * limb_bytes, 0);
          }
        }
      continue;
    }
    if (fast) {                                              // wave-uniform
      const unsigned vo = (unsigned)(((p0 + pr) * p.Q + q0 + l31) * 4 + half * 4 * pq4);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][r];
          if (lrelu) v = fmaxf(v, v * p.slope);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, vo, (i * 32 + (r & 3) + 8 * (r >> 2)) * pq4, 0);
        }
      continue;
    }
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
  // bf16 math mode: dy and the layer's saved OUTPUT y as C8 tensors [N][K/8][P][Q][8] (DY unused).  The LeakyReLU backward
  // g = dy * (y > 0 ? 1 : slope) happens while the tile is staged, and the bias gradient sum(g) comes out as tap column
  // T of the same product (its B operand is a row of ones): no separate activation-backward pass over the largest
  // activation of the net.  part is [blocks][K * (T + 1)] then.
  const unsigned short *DYc, *Yc;
  const float *Yf;               // f32 mode: the layer's saved OUTPUT [N][K][P][Q] (same fusion on f32 tensors, DY = dy); or null
  float slope;
  float *part;                   // [blocks][K * T]
  int N, H, Wd, K, P, Q, R, S, stride, pad;
  int LW, RB, xrows;             // LDS row stride (Wd + 2 pad), output rows per iteration, staged input rows
  int iters_total, iters_per_block;      // iterations = N * P / RB
};

__global__ __launch_bounds__(256, 2) void c1_wgrad_kernel(C1WParams p) {
  __shared__ __attribute__((aligned(16))) float lds[64 * C1W_LDA + 2 * C1W_XMAX];
  float *dys = lds, *xs = lds + 64 * C1W_LDA;
  const bool c8 = p.DYc != nullptr;                       // wave-uniform
  const bool fuse = c8 || p.Yf != nullptr;                // LeakyReLU backward while staging + bias gradient as tap column T
  for (int u = threadIdx.x; u < C1W_XMAX; u += 256) xs[C1W_XMAX + u] = 1.f;      // the bias-gradient "tap" reads ones
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
    toff[j] = (fuse && t == T) ? C1W_XMAX : r * p.LW + c;
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

  // C8 staging assignment: unit u = tid + 256 i (i < 4) -> (channel group u / 128, pixel u % 128 of the iteration's 128)
  int c_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = tid + 256 * i, g8 = u >> 7, px = u & 127;
    const int rb = px / p.Q, cq = px - rb * p.Q;
    c_off[i] = (g8 * 8 < p.K ? g8 : 0) * (int)PQ + rb * p.Q + cq;          // in 16-byte units
  }
  f32x4 dreg[8];
  float xreg[6];
  auto fetch = [&](int it) {
    const int n = it / (p.P / p.RB), pr = (it - n * (p.P / p.RB)) * p.RB;
    if (c8) {                                              // dreg[2 i], dreg[2 i + 1] = the 8 channels of unit i, activation undone
      const u32x4 *dq = reinterpret_cast<const u32x4 *>(p.DYc) + (long)n * (p.K >> 3) * PQ + (long)pr * p.Q;
      const u32x4 *yq = reinterpret_cast<const u32x4 *>(p.Yc) + (long)n * (p.K >> 3) * PQ + (long)pr * p.Q;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8 dv = __builtin_bit_cast(bf16x8, dq[c_off[i]]), yv = __builtin_bit_cast(bf16x8, yq[c_off[i]]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = (float)dv[e];
          dreg[2 * i + (e >> 2)][e & 3] = (float)yv[e] > 0.f ? d : d * p.slope;
        }
      }
    } else {
      const float *dyn = p.DY + (long)n * p.K * PQ + (long)pr * p.Q;
#pragma unroll
      for (int i = 0; i < 8; ++i) dreg[i] = *reinterpret_cast<const f32x4 *>(dyn + d_off[i]);
      if (p.Yf) {
        const float *yn = p.Yf + (long)n * p.K * PQ + (long)pr * p.Q;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f32x4 yv = *reinterpret_cast<const f32x4 *>(yn + d_off[i]);
#pragma unroll
          for (int e = 0; e < 4; ++e) dreg[i][e] = yv[e] > 0.f ? dreg[i][e] : dreg[i][e] * p.slope;
        }
      }
    }
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
    if (c8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int u = tid + 256 * i, g8 = u >> 7, px = u & 127;
        const bool live = g8 * 8 < p.K;
#pragma unroll
        for (int e = 0; e < 8; ++e) dys[(g8 * 8 + e) * C1W_LDA + px] = live ? dreg[2 * i + (e >> 2)][e & 3] : 0.f;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool live = (tid + 256 * i) >> 5 < p.K;      // channels >= K are zero rows
        float *d = dys + d_lds[i];
        d[0] = live ? dreg[i][0] : 0.f;
        d[1] = live ? dreg[i][1] : 0.f;
        d[2] = live ? dreg[i][2] : 0.f;
        d[3] = live ? dreg[i][3] : 0.f;
      }
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
  const int To = fuse ? T + 1 : T;                         // fused form: column T = the bias gradient
  float *out = p.part + (long)blockIdx.x * p.K * To;
  for (int u = tid; u < p.K * To; u += 256) {
    const int k = u / To, t = u - k * To;
    out[u] = red[k * 64 + t];
  }
}

// dW[k][t] = sum_b part[b][k][t] (t < T), db[k] = sum_b part[b][k][T]
__global__ __launch_bounds__(256) void c1_wgrad_c8_reduce_kernel(const float *__restrict__ part, float *__restrict__ dW, float *__restrict__ db,
                                                                 int K, int T, int blocks) {
  // workgroup = 64 consecutive elements x 4 interleaved slices of the partial blocks (fixed summation order)
  __shared__ float red[4][64];
  const int tid = threadIdx.x, u = blockIdx.x * 64 + (tid & 63), sl = tid >> 6, To = T + 1;
  float s = 0.f;
  if (u < K * To)
    for (int b = sl; b < blocks; b += 4) s += part[(long)b * K * To + u];
  red[sl][tid & 63] = s;
  __syncthreads();
  if (sl != 0 || u >= K * To) return;
  s = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  const int k = u / To, t = u - k * To;
  if (t < T)
    dW[k * T + t] = s;
  else if (db)
    db[k] = s;
}

}  // namespace lsps
#endif
