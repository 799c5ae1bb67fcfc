// Weight gradient of the 3x3 / stride 1 / pad 1 convs on 32x32 maps from bf16 activations in the C8 layout (c8conv.h):
//   dW[k][c][r][s] = sum_n sum_{y,x} dY[n][k][y][x] * X[n][c][y+r-1][x+s-1]          (autograd of common_net.py:162-163)
#ifndef LSPS_C8WGRAD_H
#define LSPS_C8WGRAD_H
#include "c8conv.h"

namespace lsps {

// -------------------------------------------------------------------------------------------
// Per tap a [K x pixels] x [pixels x C] product: the REDUCTION runs over pixels, but C8 keeps the channels of a pixel
// contiguous — the transpose of what an MFMA fragment (one channel, 8 consecutive reduction elements per lane) wants.
// gfx950's transposing LDS read does it for free: within 16 lanes, lane i points at 4 contiguous bf16 = S[key i/4][4 (i%4)..]
// of a [4 keys][16 columns] block and lane L receives column L: S[0..3][L] (measured: tools/probes/tr16_probe.hip).  With
// key = pixel and column = channel, the 16 lanes address 4 consecutive pixel units of two channel groups (2 x 64
// contiguous bytes) and each lane ends up with 4 consecutive pixels of ONE channel: two such reads = the 8 reduction
// elements of its v_mfma_f32_32x32x16_bf16 fragment.  A tap is an address offset of whole 16-byte units: no alignment games.
//
// Workgroup = 128 k x 64 c x 9 taps of the output for a range of images ("split"); 8 waves = 4 (k) x 2 (c), each 32 k x
// 32 c x 9 taps = 9 accumulator tiles (144 registers, two waves per SIMD): one dY fragment feeds 9 MFMAs.  Staging: chunks
// of 4 image rows by LDS-DMA (lane-linear 1 KB pieces, the source address does the gather; halo units are out-of-range
// lanes = zeros), two stages, one barrier per chunk.  LDS planes (one per channel group) are strided by 64 (mod 256) bytes
// so that the four planes a 32-lane read phase touches fall on distinct bank quarters.
// Partial sums part[split][tap][K][C] (f32), reduced and permuted to [K][C][3][3] by c8_wgrad_reduce_kernel.
// -------------------------------------------------------------------------------------------
#define CW8_ROWS 4                                    // image rows per chunk
#define CW8_DY_PLANE (CW8_ROWS * 32 * 16 + 64)        // 2112 bytes per k group plane (2048 + 64: bank quarter rotation)
#define CW8_DY_BYTES (16 * CW8_DY_PLANE)              // 33792: 16 k groups = 128 k
#define CW8_X_PLANE ((CW8_ROWS + 2) * C8_LDW * 16)    // 3264 bytes per c group plane: 6 rows x 34 units (3264 % 256 = 192)
#define CW8_X_UNITS (8 * (CW8_ROWS + 2) * C8_LDW)     // 1632 units = 25.5 pieces: 8 c groups = 64 c
#define CW8_X_PIECES 26
#define CW8_X_BYTES (CW8_X_PIECES * 1024)             // 26624
#define CW8_STAGE (CW8_DY_BYTES + CW8_X_BYTES)        // 60416
#define CW8_LDS_BYTES (2 * CW8_STAGE)                 // 120832
#define CW8_PIECES (32 + CW8_X_PIECES)                // 58 per chunk

struct C8WgradParams {
  const unsigned short *DY;      // [N][K/8][32][32][8]
  const unsigned short *X;       // [N][C/8][32][32][8]
  float *part;                   // [splits][9][K][C]
  int N, K, C;
  int splits, imgs_per_split;
};

__global__ __launch_bounds__(512, 1) void c8_wgrad_kernel(C8WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char c8w_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave >> 1, wc = wave & 1;                    // 32-k block (0..3), 32-c block (0..1)

  // workgroup -> (split, output tile); the tiles of one split sit on ONE XCD, adjacent in time: its L2 serves the 4 (dY) and
  // 2 (X) re-reads of the split's images
  const int CT = p.C >> 6, KT = p.K >> 7, tiles = CT * KT;
  const int lin = blockIdx.x, xcd = lin & 7, q = lin >> 3;
  const int tile = q % tiles, split = xcd + 8 * (q / tiles);
  if (split >= p.splits) return;
  const int kt = tile / CT, ct = tile - kt * CT;
  const int n0 = split * p.imgs_per_split;
  const int n1 = min(p.N, n0 + p.imgs_per_split);

  const int dy_img = (p.K >> 3) * 16384, x_img = (p.C >> 3) * 16384;     // bytes per image
  // DMA pieces of this wave: piece = wave + 8 i (i < 8): 0..31 dY (k group = piece / 2, rows 2 (piece & 1) ..), 32..57 X
  int voff[8];
  unsigned topbot[8];                                           // X pieces: bit 0: unit in the top halo row, bit 1: bottom halo row
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int piece = wave + 8 * i;
    topbot[i] = 0;
    if (piece < 32) {
      voff[i] = (kt * 16 + (piece >> 1)) * 16384 + ((piece & 1) * 64 + lane) * 16;
    } else {
      const int u = (piece - 32) * 64 + lane;                   // unit within the 8 planes of 6 x 34
      const int cg = (int)(((unsigned)u * 20561u) >> 22);       // u / 204 for u < 1664 (exact)
      const int rem = u - cg * 204;
      const int r = (int)(((unsigned)rem * 1928u) >> 16), c = rem - r * C8_LDW;      // rem / 34
      const bool in = u < CW8_X_UNITS && c >= 1 && c <= 32;
      voff[i] = in ? (ct * 8 + cg) * 16384 + ((r - 1) * 32 + (c - 1)) * 16 : (int)0x80000000;
      topbot[i] = in ? (r == 0 ? 1u : 0u) | (r == CW8_ROWS + 1 ? 2u : 0u) : 0u;
    }
  }
  auto issue = [&](int n, int rc, int stage, int i0 = 0, int i1 = 8) {      // image n, chunk rc (rows 4 rc ..) -> LDS stage; pieces i0 .. i1-1
    const c8_i32x4 drs = c8_rsrc_words(p.DY + (long)n * (dy_img >> 1), dy_img);
    const c8_i32x4 xrs = c8_rsrc_words(p.X + (long)n * (x_img >> 1), x_img);
    const int row_off = rc * (CW8_ROWS * 512);
    const unsigned sbase = c8_lds_addr(c8w_lds) + stage * CW8_STAGE;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < i0 || i >= i1) continue;
      const int piece = wave + 8 * i;
      if (piece < 32) {
        c8_dma16_asm(drs, sbase + (piece >> 1) * CW8_DY_PLANE + (piece & 1) * 1024, voff[i] + row_off, 0);
      } else if (piece < CW8_PIECES) {
        const bool dead = ((topbot[i] & 1u) && rc == 0) || ((topbot[i] & 2u) && rc == 32 / CW8_ROWS - 1);
        const int vo = dead ? (int)0x80000000 : voff[i] + row_off;     // halo columns stay out of range: 0x80000000 + small
        c8_dma16_asm(xrs, sbase + CW8_DY_BYTES + (piece - 32) * 1024, vo, 0);
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // transposing-read addresses: 16-lane group g = lane / 16: channels 16 (g & 1) .., pixels 8 (g >> 1) ..; lane i = lane % 16
  // points at pixel + i / 4, channel quad i % 4 = group 2 (g & 1) + (i % 4) / 2, byte 8 ((i % 4) & 1)
  const int g = lane >> 4, i16 = lane & 15;
  const int cgl = 2 * (g & 1) + ((i16 & 3) >> 1), px = 8 * (g >> 1) + (i16 >> 2), byte = 8 * (i16 & 1);
  const unsigned a_base = (unsigned)((wk * 4 + cgl) * CW8_DY_PLANE + px * 16 + byte);
  const unsigned b_base = (unsigned)(CW8_DY_BYTES + (wc * 4 + cgl) * CW8_X_PLANE + px * 16 + byte);

  const int chunks_per_img = 32 / CW8_ROWS, total = (n1 - n0) * chunks_per_img;
  if (total > 0) {
    issue(n0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  for (int it = 0; it < total; ++it) {
    const int stage = it & 1;
    const bool more = it + 1 < total;
    const int nn = n0 + (it + 1) / chunks_per_img, nrc = (it + 1) % chunks_per_img;
#if !defined(C8W_ABL_NODMA) && !defined(C8W_SPREAD_DMA)
    if (more) issue(nn, nrc, stage ^ 1);
#endif
    const unsigned char *As = c8w_lds + stage * CW8_STAGE + a_base;
    const unsigned char *Bs = c8w_lds + stage * CW8_STAGE + b_base;
#pragma unroll
    for (int ks = 0; ks < CW8_ROWS * 2; ++ks) {                 // 16 pixels: row ks / 2, columns 16 (ks & 1) ..
      const int rr = ks >> 1, x0 = (ks & 1) * 16;
#if defined(C8W_SPREAD_DMA) && !defined(C8W_ABL_NODMA)          // experiment: one DMA piece per k-step instead of a burst of eight
      if (more) issue(nn, nrc, stage ^ 1, ks, ks + 1);
      __builtin_amdgcn_sched_barrier(0);
#endif
      const bf16x8 af = c8_tr_frag(As + (rr * 32 + x0) * 16, As + (rr * 32 + x0 + 4) * 16);
#ifdef C8W_ABL_ALLREADS                  // the first build: every tap's fragment by its own two transposing reads (18 per k-step)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int o = ((rr + t / 3) * C8_LDW + x0 + t % 3) * 16;
        const bf16x8 bf = c8_tr_frag(Bs + o, Bs + o + 64);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[t], 0, 0, 0);
      }
#else
      // The three column taps of an image row read pixel windows that are shifted by one pixel: ONE 12-pixel window per row
      // (three transposing reads: pixels 0-3, 4-7, 8-11 of this lane's channel) serves all three, the +1 tap by four
      // v_alignbit.  9 + 2 LDS reads per k-step instead of 18 + 2 (measured: the same speed — the LDS read traffic is not what
      // the DMA stream, which costs this kernel 22 %, competes with; profiles/r3g_c8_ablations.txt).
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const unsigned char *row = Bs + ((rr + r) * C8_LDW + x0) * 16;
        const c8_s16x4 q0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c8_s16x4 __attribute__((address_space(3))) *)(row));
        const c8_s16x4 q1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c8_s16x4 __attribute__((address_space(3))) *)(row + 64));
        const c8_s16x4 q2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c8_s16x4 __attribute__((address_space(3))) *)(row + 128));
        const uint2 d0 = __builtin_bit_cast(uint2, q0), d1 = __builtin_bit_cast(uint2, q1), d2 = __builtin_bit_cast(uint2, q2);
        const unsigned w0 = d0.x, w1 = d0.y, w2 = d1.x, w3 = d1.y, w4 = d2.x;
        const u32x4 f0 = {w0, w1, w2, w3}, f2 = {w1, w2, w3, w4};
        const u32x4 f1 = {__builtin_amdgcn_alignbit(w1, w0, 16), __builtin_amdgcn_alignbit(w2, w1, 16), __builtin_amdgcn_alignbit(w3, w2, 16),
                          __builtin_amdgcn_alignbit(w4, w3, 16)};
        acc[3 * r + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, f0), acc[3 * r + 0], 0, 0, 0);
        acc[3 * r + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, f1), acc[3 * r + 1], 0, 0, 0);
        acc[3 * r + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, f2), acc[3 * r + 2], 0, 0, 0);
      }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // partial tile: acc[t][r] = D[k = kt*128 + wk*32 + (r&3) + 8 (r>>2) + 4 half][c = ct*64 + wc*32 + l31]
  const int l31 = lane & 31, half = lane >> 5;
  float *out = p.part + ((long)split * 9) * p.K * p.C + (long)(kt * 128 + wk * 32 + 4 * half) * p.C + ct * 64 + wc * 32 + l31;
#ifdef C8W_ABL_NOSTORE
  {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[q][r];
    if (t == 1.2345e30f) out[0] = t;
    return;
  }
#endif
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(long)t * p.K * p.C + (long)((r & 3) + 8 * (r >> 2)) * p.C] = acc[t][r];
}

// dW[k][c][t] = sum_s part[s][t][k][c]
__global__ __launch_bounds__(256) void c8_wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ dW, int KC, int splits) {
  // workgroup = 64 consecutive (k, c) x 4 interleaved slices of the splits (4 waves per SIMD stream the partial sums; fixed order)
  __shared__ float red[4][9][64];
  const int tid = threadIdx.x, l = tid & 63, sl = tid >> 6, kc = blockIdx.x * 64 + l;
  float s[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) s[t] = 0.f;
  if (kc < KC)
    for (int sp = sl; sp < splits; sp += 4)
#pragma unroll
      for (int t = 0; t < 9; ++t) s[t] += part[((long)sp * 9 + t) * KC + kc];
#pragma unroll
  for (int t = 0; t < 9; ++t) red[sl][t][l] = s[t];
  __syncthreads();
  // 576 outputs of the workgroup, contiguous in dW: element e = l * 9 + t
  for (int e = tid; e < 576; e += 256) {
    const int ll = e / 9, t = e - ll * 9;
    if (blockIdx.x * 64 + ll < KC)
      dW[(long)blockIdx.x * 576 + e] = (red[0][t][ll] + red[1][t][ll]) + (red[2][t][ll] + red[3][t][ll]);
  }
}

}  // namespace lsps
#endif
