// f32-class 3x3 / stride-2 / pad-1 convs on the bf16 matrix pipe (round 5): every f32 operand is carried as
// THREE bf16 limbs (x = hi + mid + lo exactly: 8 + 8 + 8 significand bits, conv_types.h: split3) and a product is six
// v_mfma_f32_32x32x16_bf16 (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi: 192 matrix-pipe cycles per 32x32x16 MACs against the
// 512 of eight v_mfma_f32_32x32x2_f32; the dropped terms are < 2^-24 relative; products of limbs are exact in f32 and the
// accumulation is f32, so the arithmetic is f32-class: tools/split_precision_check.py).
// Layers: LeakyReLUConv2d(., ., 3, 2, 1) of both nets (reference: src/trainers/common_net.py:246-256, lsps_nets.py:119-123,
// 186-192).
//
// "X3" tensor: an f32 activation stored as three C8 planes per image, XL[N][limb][C/8][H][W][8] bf16 (c8conv.h for C8).  With the limbs
// already split in HBM, staging is the C8 family's pure LDS-DMA copy (no conversion VALU in the consumer: a stride-2 layer
// re-uses an input element for only 9/4 taps x K outputs, far too few MFMAs to pay 3.5 VALU per element next to a 32-cycle MFMA).
// The split happens once per element where the tensor is produced (x3_split_nchw_kernel for tensors that arrive as f32 NCHW,
// or the producing kernel's epilogue: template parameter OUT3 below).
//
// x3s2_fwd_kernel: workgroup = 128 m x 128 output pixels (8 waves = 2 (m) x 4 (pixels), wave tile 64 m x 32 pixels);
// a STAGE is (16-channel chunk, tap row r): three limbs of the TR input rows 2p + r - 1 with their columns de-interleaved by
// parity (c8s2.h: a tap s is the wave-uniform unit offset {0, Q + 1, 1}) + three limbs of the 3 taps' weights = 63 KB, double
// buffered; per stage and wave 3 taps x 2 m tiles x 6 MFMAs from 27 ds_read_b128 (half the LDS reads per MFMA of the bf16
// kernel: each fragment feeds three or two products).
#ifndef LSPS_X3S2_H
#define LSPS_X3S2_H
#include "conv_types.h"

namespace lsps {

typedef __attribute__((address_space(3))) void *x3_lds_ptr;

#define X3_OOB 0x80000000u

// f32 [N][C][HW] -> three bf16 limb planes per image: [N][limb][C/8][HW][8]
__global__ __launch_bounds__(256) void x3_split_nchw_kernel(const float *__restrict__ x, unsigned short *__restrict__ y, int C, int HW,
                                                            long units) {
  const long u = (long)blockIdx.x * 256 + threadIdx.x;        // one 16-byte unit per limb
  if (u >= units) return;
  const int px = (int)(u % HW);
  const long ncg = u / HW;                                     // n * (C/8) + cg
  const float *src = x + ncg * 8 * HW + px;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = src[(long)e * HW];
  bf16x8 h, m, l;
  split3(v, h, m, l);
  const long cgs = C >> 3, n = ncg / cgs, cg = ncg - n * cgs;
  const long o = ((n * 3 * cgs + cg) * HW + px) * 8, ls = cgs * HW * 8;      // limb planes of ONE image are adjacent
  *reinterpret_cast<bf16x8 *>(y + o) = h;
  *reinterpret_cast<bf16x8 *>(y + o + ls) = m;
  *reinterpret_cast<bf16x8 *>(y + o + 2 * ls) = l;
}

// three bf16 planes -> f32 [N][C][HW] (checks)
__global__ __launch_bounds__(256) void x3_join_nchw_kernel(const unsigned short *__restrict__ x, float *__restrict__ y, int C, int HW,
                                                           long units) {
  const long u = (long)blockIdx.x * 256 + threadIdx.x;
  if (u >= units) return;
  const int px = (int)(u % HW);
  const long ncg = u / HW;
  const long cgs = C >> 3, n = ncg / cgs, cg = ncg - n * cgs;
  const long o = ((n * 3 * cgs + cg) * HW + px) * 8, ls = cgs * HW * 8;
  const bf16x8 h = *reinterpret_cast<const bf16x8 *>(x + o), m = *reinterpret_cast<const bf16x8 *>(x + o + ls),
               l = *reinterpret_cast<const bf16x8 *>(x + o + 2 * ls);
  float *dst = y + ncg * 8 * HW + px;
#pragma unroll
  for (int e = 0; e < 8; ++e) dst[(long)e * HW] = ((float)l[e] + (float)m[e]) + (float)h[e];
}

struct X3S2Pack {
  const float *W;
  unsigned short *Wq;
  int M, C;
  long sm, sc;                   // element strides of m and of the reduction channel in W; tap t = 3 r + s at offset t
};

// Wq[m tile of 128][chunk of 16 c][tap row r][limb][tap column s][k-half][128 m][8 c]: the LDS image of a stage's A operand
__global__ __launch_bounds__(256) void x3s2_pack_kernel(X3S2Pack p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;      // over [mt][chunk][r][s][kh][128][8]
  const long total = (long)p.M * p.C * 9;
  if (idx >= total) return;
  const int e = (int)(idx & 7), ml = (int)((idx >> 3) & 127), kh = (int)((idx >> 10) & 1);
  long rest = idx >> 11;
  const int s = (int)(rest % 3);
  rest /= 3;
  const int r = (int)(rest % 3);
  rest /= 3;
  const int chunks = p.C >> 4;
  const int chunk = (int)(rest % chunks), mt = (int)(rest / chunks);
  const int m = mt * 128 + ml, c = chunk * 16 + kh * 8 + e;
  const float x = p.W[(long)m * p.sm + (long)c * p.sc + 3 * r + s];
  const __bf16 h = (__bf16)x;
  const float r1 = x - (float)h;
  const __bf16 mi = (__bf16)r1;
  const __bf16 lo = (__bf16)(r1 - (float)mi);
  const long stage = ((long)mt * chunks + chunk) * 3 + r;
  const long o = (stage * 9 + s) * 2048 + kh * 1024 + ml * 8 + e;   // limb 0; limb l at + l * 3 * 2048
  p.Wq[o] = __builtin_bit_cast(unsigned short, h);
  p.Wq[o + 3 * 2048] = __builtin_bit_cast(unsigned short, mi);
  p.Wq[o + 6 * 2048] = __builtin_bit_cast(unsigned short, lo);
}

struct X3S2Params {
  const unsigned short *X;       // [N][3][Cx/8][H][W][8]
  const unsigned short *Wq;      // x3s2_pack_kernel's layout
  const float *bias;             // [M] or null
  float *Y;                      // f32 [N][M][P][Q] (OUT3 = false)
  unsigned short *YL;            // [N][3][M/8][P][Q][8] (OUT3 = true)
  int N, Cx, M;
  int H, W, P, Q;
  int TI, TR;                    // pixel tile (of the output map): TI images x TR rows x Q columns = 128 pixels
  int tiles_per_img, ntiles;
  float lrelu;                   // epilogue: v = max(v, v * lrelu) (1 = no activation)
};

#define X3F_BP 9                                           // image pieces (64 units) per limb and stage: <= 576 units
#define X3F_AP 12                                          // weight pieces per limb and stage: 3 taps x 2 k-halves x 128 m
#define X3F_APIECES (3 * X3F_AP)
#define X3F_BPIECES (3 * X3F_BP)
#define X3F_ASTAGE (X3F_APIECES * 1024)                    // bytes of packed weights per stage
#define X3F_STAGE ((X3F_BPIECES + X3F_APIECES) * 1024)     // 64512
#define X3F_LDS_BYTES (2 * X3F_STAGE)                      // 129024

template <bool OUT3>
__global__ __launch_bounds__(512, 1) void x3s2_fwd_kernel(X3S2Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char x3_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave & 1, wp = wave >> 1;
  typedef unsigned long long u64;

  // persistent workgroups as in c8s2_fwd_kernel: tiles lin = blockIdx.x, + grid, ...; lin -> (pixel tile, m tile)
  const int MT = p.M >> 7, G = gridDim.x, nlin = ((p.ntiles + 7) >> 3) * 8 * MT;
  const int TI = p.TI, TR = p.TR, Q = p.Q;
  const int CB = 2 * Q + 1, blk = TR * CB, plane = TI * blk, bunits = 2 * plane;
  const int W16 = p.W * 16, HW16 = p.H * W16, img_bytes = (p.Cx >> 3) * HW16, nch = p.Cx >> 4;
  const int PQ = p.P * Q, tpi = TR * Q;

  int mt, ptile, n0, p0, nimg;                                  // the tile whose DMA set-up is current
  __amdgpu_buffer_rsrc_t xrs, wrs;                              // image piece wave + 8 i belongs to limb (wave + 8 i) / 9
  unsigned voffb[4], voffa[5];
  int pimg[4], prow[4];
  unsigned pcol[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = wave + 8 * i, pc = j % X3F_BP;
    const int u = pc * 64 + lane;
    pimg[i] = -1; prow[i] = 0; pcol[i] = 0;
    if (j < X3F_BPIECES && u < bunits) {
      const int kh = u >= plane ? 1 : 0, rem = u - kh * plane;
      const int img = rem / blk, rem2 = rem - img * blk;
      const int ri = rem2 / CB, ci = rem2 - ri * CB;
      const int col = ci <= Q ? 2 * ci - 1 : 2 * (ci - Q - 1);
      if (col >= 0 && col < p.W) {
        pimg[i] = img;
        prow[i] = ri;
        pcol[i] = (unsigned)(kh * HW16 + col * 16);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) voffa[i] = (unsigned)(((wave + 8 * i) * 64 + lane) * 16);

  auto decode = [&](int lin, int &mt_, int &ptile_) {
    const int xcd = lin & 7, qq = lin >> 3;
    mt_ = qq % MT;
    ptile_ = xcd + 8 * (qq / MT);
    return lin < nlin && ptile_ < p.ntiles;
  };
  auto setup = [&](int mt_, int ptile_) {
    mt = mt_; ptile = ptile_;
    if (TI == 1) {
      n0 = ptile / p.tiles_per_img;
      p0 = (ptile - n0 * p.tiles_per_img) * TR;
    } else {
      n0 = ptile * TI;
      p0 = 0;
    }
    nimg = min(TI, p.N - n0);
    xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(p.X) + (long)n0 * 3 * (img_bytes >> 1), 0, nimg * 3 * img_bytes,
                                            0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int limb = min((wave + 8 * i) / X3F_BP, 2);
      // offset of the r = 1 row (2 (p0 + ri)); r = 0 / 2: -+ one row at issue time; row -1 (r = 0, first output row) is padding
      voffb[i] = (pimg[i] >= 0 && pimg[i] < nimg)
                     ? (unsigned)((pimg[i] * 3 + limb) * img_bytes + 2 * (p0 + prow[i]) * W16) + pcol[i] : X3_OOB;
    }
    wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(p.Wq) + (long)mt * nch * 3 * (X3F_ASTAGE >> 1), 0,
                                            nch * 3 * X3F_ASTAGE, 0x00020000);
  };
  auto issue = [&](int ch, int r, int buf) {
    unsigned char *base = x3_lds + buf * X3F_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = wave + 8 * i;
      if (piece < X3F_BPIECES) {
        unsigned v = voffb[i];
        if (r == 0) v = (p0 + prow[i] == 0) ? X3_OOB : v - (unsigned)W16;
        if (r == 2) v = v + (unsigned)W16;
        if (voffb[i] == X3_OOB) v = X3_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (x3_lds_ptr)(base + piece * 1024), 16, v, ch * 2 * HW16, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int piece = wave + 8 * i;
      if (piece < X3F_APIECES)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (x3_lds_ptr)(base + (X3F_BPIECES + piece) * 1024), 16, voffa[i],
                                                 (ch * 3 + r) * X3F_ASTAGE, 0, 0);
    }
  };

  // this lane's output pixel: tile pixel t = wp * 32 + l31 -> (image, row, column) of the tile
  const int t_px = wp * 32 + l31;
  const int yil = t_px / tpi, yrem = t_px - yil * tpi;
  const int ypl = yrem / Q, yql = yrem - ypl * Q;
  const unsigned bbase = (unsigned)((half * plane + yil * blk + ypl * CB + yql) * 16);
  const unsigned a_base = (unsigned)(X3F_BPIECES * 1024 + (half * 128 + wm * 64 + l31) * 16);

  int lin = blockIdx.x, buf = 0;
  {
    int m_, t_;
    if (!decode(lin, m_, t_)) return;
    setup(m_, t_);
  }
  issue(0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  while (true) {
    const int mt_c = mt;
    const bool yvalid = yil < nimg;
    const long ypix = (long)(n0 + yil) * p.M * PQ + (long)(p0 + ypl) * Q + yql;          // f32 NCHW: + channel * PQ
    const long yunit = (long)(n0 + yil) * 3 * (p.M >> 3) * PQ + (long)(p0 + ypl) * Q + yql;  // X3: + (limb * M/8 + channel group) * PQ
    int mt_n, ptile_n;
    const bool more = decode(lin + G, mt_n, ptile_n);

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    for (int ch = 0; ch < nch; ++ch) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        if (r < 2) {
          issue(ch, r + 1, buf ^ 1);
        } else if (ch + 1 < nch) {
          issue(ch + 1, 0, buf ^ 1);
        } else if (more) {
          setup(mt_n, ptile_n);
          issue(0, 0, buf ^ 1);
        }
        const unsigned char *S = x3_lds + buf * X3F_STAGE;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int coff = (s == 0 ? 0 : (s == 1 ? Q + 1 : 1)) * 16;
          bf16x8 af[2][3], bf[3];
#pragma unroll
          for (int l = 0; l < 3; ++l) {
            bf[l] = *reinterpret_cast<const bf16x8 *>(S + l * (X3F_BP * 1024) + bbase + coff);
#pragma unroll
            for (int i = 0; i < 2; ++i)
              af[i][l] = *reinterpret_cast<const bf16x8 *>(S + a_base + ((l * 3 + s) * 256 + i * 32) * 16);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i] = mfma_split6(af[i][0], af[i][1], af[i][2], bf[0], bf[1], bf[2], acc[i]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        buf ^= 1;
      }
    }

    // epilogue: acc[i][r] = channel mt*128 + wm*64 + i*32 + (r&3) + 8 (r>>2) + 4 half of this lane's pixel
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int m4 = mt_c * 128 + wm * 64 + i * 32 + 8 * rq + 4 * half;
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) b4 = *reinterpret_cast<const f32x4 *>(p.bias + m4);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = acc[i][rq * 4 + e] + b4[e];
          v[e] = fmaxf(x, x * p.lrelu);
        }
        if (!yvalid) continue;
        if (!OUT3) {
#pragma unroll
          for (int e = 0; e < 4; ++e) p.Y[ypix + (long)(m4 + e) * PQ] = v[e];
        } else {
          bf16x4 h, mi, lo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            h[e] = (__bf16)v[e];
            const float r1 = v[e] - (float)h[e];
            mi[e] = (__bf16)r1;
            lo[e] = (__bf16)(r1 - (float)mi[e]);
          }
          const long o = ((yunit + (long)(m4 >> 3) * PQ) << 1) + half;      // 8-byte pieces
          u64 *Y = reinterpret_cast<u64 *>(p.YL);
          const long ls = (long)(p.M >> 3) * PQ * 2;                         // limb plane of one image in 8-byte pieces
          Y[o] = __builtin_bit_cast(u64, h);
          Y[o + ls] = __builtin_bit_cast(u64, mi);
          Y[o + 2 * ls] = __builtin_bit_cast(u64, lo);
        }
      }
    if (!more) return;
    lin += G;
  }
}

}  // namespace lsps
#endif
