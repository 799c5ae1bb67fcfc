// bf16 residual trunk (BASELINE config 5): the 3x3 / stride 1 / pad 1 convs of LeakyINSResBlock on 32x32 maps with the
// InstanceNorm that follows each of them (reference: src/trainers/common_net.py:160-181), on bf16 activations kept in a
// channel-group layout between the blocks.
#ifndef LSPS_C8CONV_H
#define LSPS_C8CONV_H
#include "conv_types.h"
#include "c8util.h"

namespace lsps {

// -------------------------------------------------------------------------------------------
// Activation layout "C8": [N][C/8][H][W][8] bf16 — 8 consecutive channels of one pixel are one 16-byte unit.
//   * a v_mfma_f32_32x32x16_bf16 operand fragment is 8 consecutive reduction elements per lane (lanes 0-31: k 0..7,
//     lanes 32-63: k 8..15): with the channels innermost one pixel's unit of channel group 2g (+1) IS the B fragment of
//     lane (pixel, k-half), for every tap, so staging is a pure copy: no conversion, no transpose, no VALU;
//   * the copy is LDS-DMA (buffer_load_dwordx4 ... lds): lane-linear 1 KB pieces whose per-lane SOURCE address does the
//     gather.  The LDS image of a 16-channel chunk is two planes (k-halves) of 34 x 34 units (the image with its zero
//     halo); halo units are lanes whose buffer offset is out of range, which the hardware returns as zeros;
//   * fragment reads are ds_read_b128 of 32 consecutive units per half-wave: conflict-free without padding or swizzle;
//   * the C/D layout of the 32x32 MFMA gives a lane 4 consecutive output channels of one pixel per accumulator
//     register quad: the epilogue stores 8-byte pieces, 512 contiguous bytes per wave instruction.
// Weights: packed once per call (cached in the trainer's pack-cache scope) as bf16
//   Wq[k tile of 64][chunk of 16 c][tap][k-half][64 k][8 c]  = the LDS image of a chunk's A operand (18 KB).
//
// Workgroup = ONE image x 64 output channels (512 threads; wave w owns image rows 4w..4w+3 for all 64 channels: 2 x 4
// accumulator tiles = 128 registers, two waves per SIMD), so whole (n, k) planes live in one workgroup's registers and
// InstanceNorm (+ LeakyReLU | + residual), or the BACKWARD of the InstanceNorm + LeakyReLU in front of a dgrad conv,
// happens in the epilogue (same contract as conv_wino4.h).  Main loop: per 16-channel chunk 55 DMA pieces fill the other
// LDS stage while 9 taps x 8 MFMAs run from this one; one barrier per chunk.
// -------------------------------------------------------------------------------------------
#define C8_LDW 34
#define C8_PLANE (C8_LDW * C8_LDW)                   // 1156 units per k-half plane
#define C8_BPIECES 37                                // 2 planes = 2312 units -> 37 DMA pieces of 64 units
#define C8_BUNITS (C8_BPIECES * 64)
#define C8_APIECES 18                                // 9 taps x 2 k-halves x 64 k = 1152 units
#define C8_AUNITS (C8_APIECES * 64)
#define C8_STAGE ((C8_BUNITS + C8_AUNITS) * 16)      // 56320 bytes
#define C8_RBYTES (128 * 1024)                       // epilogue operand tile R (64 channels x 1024 pixels bf16) staged over the dead stages
#define C8_SCRATCH (C8_RBYTES)                       // epilogue reduction scratch behind it (5 KB)
#define C8_LDS_BYTES (C8_RBYTES + 5120)              // 136192 bytes (stages: 112640): one workgroup per CU
#define C8_ACHUNK (C8_AUNITS * 8)                    // bf16 elements of packed weights per (k tile, chunk)

struct C8Pack {
  const float *W;
  unsigned short *Wq;
  int M, C;                      // output / reduction channels of THIS op (dgrad: M = conv input channels)
  long sm, sc;                   // element strides of m and c in W
  int tapidx[9];                 // element offset of tap t = 3 r + s (input offset (r-1, s-1)) in W
};

__global__ __launch_bounds__(256) void c8_pack_kernel(C8Pack p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;      // over [kt][chunk][tap][khalf][64][8]
  const int chunks = p.C >> 4;
  const long total = (long)(p.M >> 6) * chunks * C8_ACHUNK;
  if (idx >= total) return;
  const int e = (int)(idx & 7), kl = (int)((idx >> 3) & 63), kh = (int)((idx >> 9) & 1);
  long rest = idx >> 10;
  const int t = (int)(rest % 9);
  rest /= 9;
  const int chunk = (int)(rest % chunks), kt = (int)(rest / chunks);
  const int m = kt * 64 + kl, c = chunk * 16 + kh * 8 + e;
  p.Wq[idx] = __builtin_bit_cast(unsigned short, (__bf16)p.W[(long)m * p.sm + (long)c * p.sc + p.tapidx[t]]);
}

// f32 [N][C][HW] -> bf16 [N][C/8][HW][8]
__global__ __launch_bounds__(256) void c8_from_nchw_kernel(const float *__restrict__ x, unsigned short *__restrict__ y, int C, int HW,
                                                           long total) {
  const long u = (long)blockIdx.x * 256 + threadIdx.x;        // one 16-byte unit
  if (u >= total) return;
  const int px = (int)(u % HW);
  const long ncg = u / HW;                                     // n * (C/8) + cg
  const float *src = x + ncg * 8 * HW + px;
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (__bf16)src[(long)e * HW];
  *reinterpret_cast<bf16x8 *>(y + u * 8) = v;
}

// bf16 [N][C/8][HW][8] -> f32 [N][C][HW]
__global__ __launch_bounds__(256) void c8_to_nchw_kernel(const unsigned short *__restrict__ x, float *__restrict__ y, int C, int HW,
                                                         long total) {
  const long u = (long)blockIdx.x * 256 + threadIdx.x;
  if (u >= total) return;
  const int px = (int)(u % HW);
  const long ncg = u / HW;
  const bf16x8 v = *reinterpret_cast<const bf16x8 *>(x + u * 8);
  float *dst = y + ncg * 8 * HW + px;
#pragma unroll
  for (int e = 0; e < 8; ++e) dst[(long)e * HW] = (float)v[e];
}

// out = a + b (bf16 elementwise; GaussianNoiseLayer on a C8 tensor, common_net.py:39-40)
__global__ __launch_bounds__(256) void c8_add_kernel(const unsigned short *__restrict__ a, const unsigned short *__restrict__ b,
                                                     unsigned short *__restrict__ out, long units) {
  const long u = (long)blockIdx.x * 256 + threadIdx.x;
  if (u >= units) return;
  const bf16x8 x = *reinterpret_cast<const bf16x8 *>(a + u * 8), y = *reinterpret_cast<const bf16x8 *>(b + u * 8);
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (__bf16)((float)x[e] + (float)y[e]);
  *reinterpret_cast<bf16x8 *>(out + u * 8) = o;
}

// out = a + b with b an f32 [N][C][HW] tensor (GaussianNoiseLayer's fresh draw, common_net.py:39-40): no conversion pass for b
__global__ __launch_bounds__(256) void c8_add_nchw_kernel(const unsigned short *__restrict__ a, const float *__restrict__ b,
                                                          unsigned short *__restrict__ out, int HW, long units) {
  const long u = (long)blockIdx.x * 256 + threadIdx.x;
  if (u >= units) return;
  const int px = (int)(u % HW);
  const long ncg = u / HW;                                     // n * (C/8) + cg
  const float *src = b + ncg * 8 * HW + px;
  const bf16x8 x = *reinterpret_cast<const bf16x8 *>(a + u * 8);
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (__bf16)((float)x[e] + src[(long)e * HW]);
  *reinterpret_cast<bf16x8 *>(out + u * 8) = o;
}

// Backward of InstanceNorm (+ residual | + LeakyReLU) from the OUTPUT on C8 tensors (norm_act.hip: inorm_bwd_kernel in this
// layout).  residual variant (res != null): g = dout, xh = out - res;  activation variant (slope > 0): g = dout * lrelu'(out),
// xh = out > 0 ? out : out / slope;  dy = rstd * (g - mean(g) - xh * mean(g * xh)).  One workgroup per (n, channel group).
__global__ __launch_bounds__(256) void c8_inorm_bwd_kernel(const unsigned short *__restrict__ dout, const unsigned short *__restrict__ out,
                                                           const unsigned short *__restrict__ res, const float *__restrict__ rstd,
                                                           unsigned short *__restrict__ dy, int HW, float slope) {
  __shared__ float red[4][16];
  const long base = (long)blockIdx.x * HW;                     // units
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float inv_slope = slope > 0.f ? 1.f / slope : 0.f;
  float g[4][8], xh[4][8], s[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long u = base + tid + 256 * i;
    const bool ok = tid + 256 * i < HW;
    bf16x8 dv, ov, rv;
#pragma unroll
    for (int e = 0; e < 8; ++e) dv[e] = ov[e] = rv[e] = (__bf16)0.f;
    if (ok) {
      dv = *reinterpret_cast<const bf16x8 *>(dout + u * 8);
      ov = *reinterpret_cast<const bf16x8 *>(out + u * 8);
      if (res) rv = *reinterpret_cast<const bf16x8 *>(res + u * 8);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float o = (float)ov[e], d = (float)dv[e];
      if (res) {
        g[i][e] = d;
        xh[i][e] = o - (float)rv[e];
      } else {
        g[i][e] = c8_sel_nonpos(o, d * slope, d);
        xh[i][e] = fmaxf(o, 0.f) + fminf(o, 0.f) * inv_slope;
      }
      s[e] += g[i][e];
      s[8 + e] = fmaf(g[i][e], xh[i][e], s[8 + e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    s[e] = wave_sum(s[e]);
    if (lane == 0) red[wave][e] = s[e];
  }
  __syncthreads();
  const float inv = 1.f / (float)HW;
  float m1[8], m2[8], rs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    m1[e] = (red[0][e] + red[1][e] + red[2][e] + red[3][e]) * inv;
    m2[e] = (red[0][8 + e] + red[1][8 + e] + red[2][8 + e] + red[3][8 + e]) * inv;
    rs[e] = rstd[(long)blockIdx.x * 8 + e];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (tid + 256 * i >= HW) continue;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)(rs[e] * (g[i][e] - m1[e] - xh[i][e] * m2[e]));
    *reinterpret_cast<bf16x8 *>(dy + (base + tid + 256 * i) * 8) = o;
  }
}

struct C8ConvParams {
  const unsigned short *X;       // [N][Cx/8][32][32][8]
  const unsigned short *Wq;      // c8_pack_kernel's layout
  const unsigned short *R;       // mode 0: optional addend; mode 2: residual; mode 3: saved OUTPUT of the norm layer being differentiated
  unsigned short *Y;             // [N][M/8][32][32][8]
  float *rstd;                   // [N*M]: written by modes 1, 2; READ by mode 3
  int N, Cx, M;
  int mode;                      // 0: y = conv (+ R);  1: y = lrelu_slope(IN(conv)) (slope < 0: none);  2: y = IN(conv) + R;
                                 // 3: y = backward of IN + LeakyReLU(slope) applied to conv, from the saved output R and rstd
  float slope, eps;
};

template <int MODE>
__global__ __launch_bounds__(512, 1) void c8_conv3x3_kernel(C8ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char c8_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;

  // workgroup -> (image, k tile).  Workgroups go to the 8 XCDs round-robin in launch order: the KT k tiles of an image are
  // consecutive on ONE XCD (its L2 serves the image's re-reads; all packed weights, 1.2 MB at 256 x 256, stay resident)
  const int KT = p.M >> 6, lin = blockIdx.x;
  const int xcd = lin & 7, q = lin >> 3;
  const int kt = q % KT, n = xcd + 8 * (q / KT);
  if (n >= p.N) return;
  const int nch = p.Cx >> 4;
  const int img_bytes = (p.Cx >> 3) * 1024 * 16;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short *>(p.X) + (long)n * (img_bytes >> 1), 0, img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short *>(p.Wq) + (long)kt * nch * C8_ACHUNK, 0, nch * C8_ACHUNK * 2, 0x00020000);

  // DMA pieces of this wave: piece = wave + 8 i (i < 7); pieces 0..36 image, 37..54 weights.  Per-lane source offsets once.
  unsigned voff[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int piece = wave + 8 * i;
    if (piece < C8_BPIECES) {
      const int u = piece * 64 + lane;
      const int pl = u >= C8_PLANE ? 1 : 0, rem = u - pl * C8_PLANE;
      const int r = (int)(((unsigned)rem * 1928u) >> 16), c = rem - r * C8_LDW;     // rem / 34 for rem < 2369 (exact)
      const bool in = u < 2 * C8_PLANE && r >= 1 && r <= 32 && c >= 1 && c <= 32;
      voff[i] = in ? (unsigned)(pl * 16384 + ((r - 1) * 32 + (c - 1)) * 16) : 0x80000000u;
    } else {
      voff[i] = (unsigned)(((piece - C8_BPIECES) * 64 + lane) * 16);
    }
  }
  auto issue = [&](int ch, int stage) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int piece = wave + 8 * i;
      if (piece < C8_BPIECES) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (c8_lds_ptr)(c8_lds + stage * C8_STAGE + piece * 1024), 16, voff[i],
                                                 ch * 32768, 0, 0);
      } else if (piece < C8_BPIECES + C8_APIECES) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            wrs, (c8_lds_ptr)(c8_lds + stage * C8_STAGE + C8_BUNITS * 16 + (piece - C8_BPIECES) * 1024), 16, voff[i],
            ch * (C8_ACHUNK * 2), 0, 0);
      }
    }
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const unsigned b_base = (unsigned)((half * C8_PLANE + 4 * wave * C8_LDW + l31) * 16);
  const unsigned a_base = (unsigned)(C8_BUNITS * 16 + (half * 64 + l31) * 16);

  // The epilogue's second operand (addend / residual / saved output: this workgroup's 64 channels of image n, 128 KB) is
  // fetched by LDS-DMA over the main loop's stages: the pieces that do not overlap the LAST chunk's stage while that chunk
  // computes, the rest after the loop — the epilogue reads it from LDS (a per-element global load -> store chain, which
  // the compiler must serialise because Y may alias R, cost 25 - 40 % of the kernel: profiles/r3d_pmc_c8.txt).
  const bool useR = (MODE == 2 || MODE == 3 || (MODE == 0 && p.R != nullptr));
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short *>(p.R ? p.R : p.X) + ((long)n * (p.M >> 3) + kt * 8) * 8192, 0, C8_RBYTES, 0x00020000);
  const int last_lo = ((nch - 1) & 1) * (C8_STAGE / 1024), last_hi = last_lo + C8_STAGE / 1024;     // pieces under the last stage
  auto issueR = [&](bool late) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int piece = wave + 8 * i;
      const bool under = piece >= last_lo && piece < last_hi;
      if (under == late)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rrs, (c8_lds_ptr)(c8_lds + piece * 1024), 16, (unsigned)(lane * 16), piece * 1024, 0, 0);
    }
  };

  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int ch = 0; ch < nch; ++ch) {
    const int stage = ch & 1;
#ifndef C8_ABL_NODMA                     // ablation builds: tools/build_abl_c8.sh (profiles/r3g_c8_ablations.txt)
    if (ch + 1 < nch) issue(ch + 1, stage ^ 1);
    else if (useR) issueR(false);
#endif
    const unsigned char *Bs = c8_lds + stage * C8_STAGE + b_base;
    const unsigned char *As = c8_lds + stage * C8_STAGE + a_base;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      bf16x8 af[2], bf[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(As + (t * 128 + i * 32) * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const bf16x8 *>(Bs + ((j + t / 3) * C8_LDW + t % 3) * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

#ifdef C8_ABL_NOEPI
  {                                     // ablation: no epilogue (the accumulators stay live through an impossible store)
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 1.2345e30f) p.Y[0] = 1;
    return;
  }
#endif
  if (useR) {
    issueR(true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---------------------------------------------------------------------------------------------------------------
  // epilogue.  acc[i][j][r]: channel kt*64 + i*32 + (r&3) + 8*(r>>2) + 4*half, pixel (row 4*wave + j, column l31).
  // A register quad (r>>2 fixed) = 4 consecutive channels = 8 bytes of the unit of channel group kt*8 + i*4 + (r>>2).
  // ---------------------------------------------------------------------------------------------------------------
  const long plane_units = 1024;
  const long y_img = (long)n * (p.M >> 3) * plane_units;
  typedef unsigned long long u64;
  auto loadR = [&](int i, int rq, int j, float (&o)[4]) {        // R's 4 channels of this lane's accumulator quad, from LDS
    const bf16x4 v = __builtin_bit_cast(
        bf16x4, *reinterpret_cast<const u64 *>(c8_lds + (((i * 4 + rq) * 1024 + (4 * wave + j) * 32 + l31) << 4) + 8 * half));
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (float)v[e];
  };
  // Stores: a lane holds 4 channels (8 bytes) of a pixel's unit, lane + 32 the other 4.  Two image rows (j, j + 1) are
  // exchanged across the half-waves (v_permlane32_swap: cdna_hip_programming.md T21) so that lanes 0-31 store the whole
  // 16-byte units of row j and lanes 32-63 those of row j + 1: 16 store instructions of 16 B per lane instead of 32 of 8 B
  // (the store tail of a one-workgroup-per-CU kernel is issue-bound).
  typedef unsigned c8_u32x2 __attribute__((ext_vector_type(2)));
  auto pack4 = [&](const float (&o)[4]) -> c8_u32x2 {
    bf16x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (__bf16)o[e];
    return __builtin_bit_cast(c8_u32x2, v);
  };
  auto store_pair = [&](int i, int rq, int j0, c8_u32x2 A, c8_u32x2 B) {       // A: row 4 wave + j0, B: the row below
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const auto sw = __builtin_amdgcn_permlane32_swap(A[d], B[d], false, false);
      A[d] = sw[0];
      B[d] = sw[1];
    }
    const long unit = y_img + (long)(kt * 8 + i * 4 + rq) * plane_units + (4 * wave + j0 + half) * 32 + l31;
    reinterpret_cast<u32x4 *>(p.Y)[unit] = u32x4{A[0], A[1], B[0], B[1]};
  };

  if (MODE == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
      {
        c8_u32x2 pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = acc[i][j][rq * 4 + e];
          if (useR) {
            float a[4];
            loadR(i, rq, j, a);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += a[e];
          }
          pk[j] = pack4(o);
        }
        store_pair(i, rq, 0, pk[0], pk[1]);
        store_pair(i, rq, 2, pk[2], pk[3]);
      }
    return;
  }

  // plane statistics: two sums per channel over the image's 1024 pixels
  //   modes 1, 2:  s1 = sum v,  s2 = sum v^2             (v = conv output, f32 accumulators)
  //   mode 3:      s1 = sum g,  s2 = sum g * xh          (g = d * lrelu'(o), xh = o > 0 ? o : o / slope; o = saved output)
  const float inv_slope = p.slope > 0.f ? 1.f / p.slope : 0.f;
  const float lrelu = p.slope >= 0.f ? p.slope : 1.f;
  float s1[32], s2[32];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (MODE == 3) {
          float o[4];
          loadR(i, rq, j, o);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // branch-free forms (64 compare masks held in SGPR pairs spill): sign mask + bitfield select, max / min
            const float d = acc[i][j][rq * 4 + e];
            const float g = c8_sel_nonpos(o[e], d * p.slope, d);
            const float xh = fmaxf(o[e], 0.f) + fminf(o[e], 0.f) * inv_slope;
            acc[i][j][rq * 4 + e] = g;
            a1[e] += g;
            a2[e] = fmaf(g, xh, a2[e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = acc[i][j][rq * 4 + e];
            a1[e] += v;
            a2[e] = fmaf(v, v, a2[e]);
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s1[i * 16 + rq * 4 + e] = a1[e];
        s2[i * 16 + rq * 4 + e] = a2[e];
      }
    }
  c8_reduce_scatter32<16>(s1, l31);
  c8_reduce_scatter32<16>(s2, l31);
  // lane (half, l31) now holds the wave's sums of accumulator slot qs = l31: channel kt*64 + (qs>>4)*32 + (qs&3) + 8*((qs>>2)&3) + 4*half
  float *red = reinterpret_cast<float *>(c8_lds + C8_SCRATCH);  // [wave 8][half 2][32][2]   (behind the R tile)
  float *stat = red + 8 * 2 * 32 * 2;                            // [half 2][32][4]
  red[((wave * 2 + half) * 32 + l31) * 2 + 0] = s1[0];
  red[((wave * 2 + half) * 32 + l31) * 2 + 1] = s2[0];
  __syncthreads();
  if (tid < 64) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      t1 += red[((w * 2 + half) * 32 + l31) * 2 + 0];
      t2 += red[((w * 2 + half) * 32 + l31) * 2 + 1];
    }
    const int m = kt * 64 + (l31 >> 4) * 32 + (l31 & 3) + 8 * ((l31 >> 2) & 3) + 4 * half;
    const float m1 = t1 * (1.f / 1024.f), m2 = t2 * (1.f / 1024.f);
    float a, b;
    if (MODE == 3) {
      a = m1;                                                     // mean(g)
      b = m2;                                                     // mean(g * xh)
      stat[(half * 32 + l31) * 4 + 2] = p.rstd[(long)n * p.M + m];
    } else {
      const float var = fmaxf(m2 - m1 * m1, 0.f);
      a = m1;
      b = rsqrtf(var + p.eps);
      p.rstd[(long)n * p.M + m] = b;
    }
    stat[(half * 32 + l31) * 4 + 0] = a;
    stat[(half * 32 + l31) * 4 + 1] = b;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      f32x4 st[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) st[e] = *reinterpret_cast<const f32x4 *>(stat + (half * 32 + i * 16 + rq * 4 + e) * 4);
      c8_u32x2 pk[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float o[4];
        if (MODE == 3) {
          float sv[4];
          loadR(i, rq, j, sv);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xh = fmaxf(sv[e], 0.f) + fminf(sv[e], 0.f) * inv_slope;
            o[e] = st[e][2] * (acc[i][j][rq * 4 + e] - st[e][0] - xh * st[e][1]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = (acc[i][j][rq * 4 + e] - st[e][0]) * st[e][1];
            if (MODE == 1) v = fmaxf(v, v * lrelu);              // LeakyReLU for 0 <= slope <= 1; slope < 0 (none): lrelu = 1
            o[e] = v;
          }
          if (MODE == 2) {
            float a[4];
            loadR(i, rq, j, a);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += a[e];
          }
        }
        pk[j] = pack4(o);
      }
      store_pair(i, rq, 0, pk[0], pk[1]);
      store_pair(i, rq, 2, pk[2], pk[3]);
    }
}

}  // namespace lsps
#endif
