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
#include "c8util.h"

namespace lsps {

typedef c8_lds_ptr x3_lds_ptr;

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

// Wq[m tile of BM][chunk of 16 c][tap row r][limb][tap column s][k-half][BM m][8 c]: the LDS image of a stage's A operand
// (BM = 128: x3s2_fwd_kernel, BM = 64: x3s2_tr_kernel).  One thread = the 8 channels x 9 taps of one (m, k-half): 27 stores of
// 16 bytes, contiguous over the BM threads of an (m tile, chunk, k-half); the 72 source floats are one 288-byte run when the
// reduction channel is the inner dimension of W (sc = 9: float4 loads), nine-float runs contiguous ACROSS the threads otherwise
// (sm = 9).  The deepest discriminator layer (1024 x 2048 x 9 weights: 75 MB in, 113 MB out) is re-packed twice per estimate-mode
// step (forward and dgrad panels), 70 us each with one thread per ELEMENT (4-byte strided loads, 2-byte stores).
template <int BM>
__global__ __launch_bounds__(256) void x3s2_pack_kernel(X3S2Pack p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;      // over [mt][chunk][kh][BM]
  const long total = (long)p.M * p.C >> 3;
  if (idx >= total) return;
  const int ml = (int)(idx % BM);
  long rest = idx / BM;
  const int kh = (int)(rest & 1);
  rest >>= 1;
  const int chunks = p.C >> 4;
  const int chunk = (int)(rest % chunks), mt = (int)(rest / chunks);
  const int m = mt * BM + ml, c0 = chunk * 16 + kh * 8;
  float x[8][9];
  const float *src = p.W + (long)m * p.sm + (long)c0 * p.sc;
  if (p.sc == 9 && !(reinterpret_cast<uintptr_t>(src) & 15)) {       // uniform: W 16-byte aligned, sm and 8 sc multiples of 4 floats
    float v[72];
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const f32x4 q = reinterpret_cast<const f32x4 *>(src)[i];
      v[4 * i] = q[0]; v[4 * i + 1] = q[1]; v[4 * i + 2] = q[2]; v[4 * i + 3] = q[3];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int t = 0; t < 9; ++t) x[e][t] = v[e * 9 + t];
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int t = 0; t < 9; ++t) x[e][t] = src[(long)e * p.sc + t];
  }
  const long stage0 = ((long)mt * chunks + chunk) * 3;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = x[e][t];
    bf16x8 h, mi, lo;
    split3(v, h, mi, lo);
    const int r = t / 3, sx = t - 3 * r;
    unsigned short *o = p.Wq + (((stage0 + r) * 9 + sx) * 2 + kh) * (BM * 8) + ml * 8;       // limb 0; limb l at + l * 3 * 2 BM 8
    *reinterpret_cast<bf16x8 *>(o) = h;
    *reinterpret_cast<bf16x8 *>(o + 3 * 2 * BM * 8) = mi;
    *reinterpret_cast<bf16x8 *>(o + 6 * 2 * BM * 8) = lo;
  }
}

struct X3S2Params {
  const unsigned short *X;       // [N][3][Cx/8][H][W][8]
  const unsigned short *Wq;      // x3s2_pack_kernel<128>'s layout
  const float *bias;             // [M] or null
  float *Y;                      // f32 [N][M][P][Q] (OUT3 = false)
  unsigned short *YL;            // [N][3][M/8][P][Q][8] (OUT3 = true)
  int N, Cx, M;
  int H, W, P, Q;
  int TI, TR;                    // pixel tile (of the output map): TI images x TR rows x Q columns = 128 pixels
  int tiles_per_img, ntiles;
  float lrelu;                   // epilogue: v = max(v, v * lrelu) (1 = no activation)
  // MASKED (ConvTranspose2d dgrad whose result is the gradient w.r.t. the OUTPUT of the X3 layer in front): y *= LeakyReLU'(ActY)
  // from the hi limb of that layer's saved X3 output (Y's shape), dbpart[pixel tile][M] = its bias-gradient partial sums
  const unsigned short *ActY;
  float act_slope;
  float *dbpart;
  // split-K (few tiles, long reductions: the deep discriminator layers at estimate-mode batch sizes): the 16-channel chunks are cut
  // into `ksplit` ranges of `kper` chunks, each (tile, range) is a workgroup that stores its RAW f32 accumulators (no bias /
  // activation / mask: bias = null, lrelu = 1) at Y + range * ysplit; x3_splitk_finish_kernel sums the ranges and applies the epilogue
  int ksplit, kper;
  long ysplit;
  // workgroup -> (pixel tile, m tile, k range) walk.  0: a pixel tile lives on XCD tile % 8 (its image rows are fetched into ONE L2 and
  // shared by the m tiles: the large-map layers); 1: unit = workgroup index, m tile / k range fastest — every CU gets work when
  // there are fewer than 8 pixel tiles (or not a multiple of 8), and with (M tiles x ranges) % 8 == 0 an XCD only ever reads its own
  // eighth of the weight panels (the deep discriminator layers: 113 MB of limbs against a few MB of activations)
  int linear;
};

#define X3F_BP 10                                          // image pieces (64 units) per limb and stage: <= 640 units
#define X3F_AP 12                                          // weight pieces per limb and stage: 3 taps x 2 k-halves x 128 m
#define X3F_APIECES (3 * X3F_AP)
#define X3F_BPIECES (3 * X3F_BP)
#define X3F_ASTAGE (X3F_APIECES * 1024)                    // bytes of packed weights per stage
#define X3F_STAGE ((X3F_BPIECES + X3F_APIECES) * 1024)     // 64512
#define X3F_LDS_BYTES (2 * X3F_STAGE)                      // 129024
// RING variant (round 6): the image pieces of a stage are requested TWO stages ahead into a ring of three image
// buffers, the (L2-resident) weight pieces one stage ahead into two weight buffers; the stage-end wait is a counted vmcnt that leaves
// the newest image requests in flight instead of draining to zero.  9 pieces per limb (<= 576 units: every layer but the 2x2 maps).
#define X3R_BP 9
#define X3R_BPIECES (3 * X3R_BP)
#define X3R_IMG (X3R_BPIECES * 1024)                       // 27648
#define X3R_LDS_BYTES (3 * X3R_IMG + 2 * X3F_ASTAGE)       // 156672

struct X3Stage { int lin, mtk, ptile, ch, ke, r, valid; };

template <bool OUT3, bool MASKED, bool RING = false>
__global__ __launch_bounds__(512, 1) void x3s2_fwd_kernel(X3S2Params p) {
  constexpr int BP = RING ? X3R_BP : X3F_BP, BPIECES = 3 * BP;
  extern __shared__ __attribute__((aligned(16))) unsigned char x3_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave & 1, wp = wave >> 1;
  typedef unsigned long long u64;

  // persistent workgroups as in c8s2_fwd_kernel: tiles lin = blockIdx.x, + grid, ...; lin -> (pixel tile, m tile)
  const int MT = p.M >> 7, KS = p.ksplit, G = gridDim.x, nlin = ((p.ntiles + 7) >> 3) * 8 * MT * KS;
  const int TI = p.TI, TR = p.TR, Q = p.Q;
  const int CB = 2 * Q + 1, blk = TR * CB, plane = TI * blk, bunits = 2 * plane;
  const int W16 = p.W * 16, HW16 = p.H * W16, img_bytes = (p.Cx >> 3) * HW16, nch = p.Cx >> 4;
  const int PQ = p.P * Q, tpi = TR * Q;

  int mt, ptile, n0, p0, nimg, kc0, kc1, ksp;                   // the tile whose DMA set-up is current (chunks kc0 .. kc1 - 1)
  __amdgpu_buffer_rsrc_t xrs, wrs;                              // image piece wave + 8 i belongs to limb (wave + 8 i) / 9
  unsigned voffb[4], voffa[5];
  int pimg[4], prow[4];
  unsigned pcol[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = wave + 8 * i, pc = j % BP;
    const int u = pc * 64 + lane;
    pimg[i] = -1; prow[i] = 0; pcol[i] = 0;
    if (j < BPIECES && u < bunits) {
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

  auto decode = [&](int lin, int &mt_, int &ptile_) {                // mt_ carries (m tile, k range): mt + MT * range
    if (p.linear) {                                                  // weight-stationary walk (x3_plan in x3.hip): see X3S2Params::linear
      mt_ = lin % (MT * KS);
      ptile_ = lin / (MT * KS);
      return ptile_ < p.ntiles;
    }
    const int xcd = lin & 7, qq = lin >> 3;
    mt_ = qq % (MT * KS);
    ptile_ = xcd + 8 * (qq / (MT * KS));
    return lin < nlin && ptile_ < p.ntiles;
  };
  auto setup = [&](int mt_, int ptile_) {
    ksp = mt_ / MT;
    mt = mt_ - ksp * MT; ptile = ptile_;
    kc0 = ksp * p.kper;
    kc1 = min(nch, kc0 + p.kper);
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
      const int limb = min((wave + 8 * i) / BP, 2);
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
      if (piece < BPIECES) {
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
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (x3_lds_ptr)(base + (BPIECES + piece) * 1024), 16, voffa[i],
                                                 (ch * 3 + r) * X3F_ASTAGE, 0, 0);
    }
  };

  // this lane's output pixel: tile pixel t = wp * 32 + l31 -> (image, row, column) of the tile
  const int t_px = wp * 32 + l31;
  const int yil = t_px / tpi, yrem = t_px - yil * tpi;
  const int ypl = yrem / Q, yql = yrem - ypl * Q;
  const unsigned bbase = (unsigned)((half * plane + yil * blk + ypl * CB + yql) * 16);
  const unsigned a_base = (unsigned)(BPIECES * 1024 + (half * 128 + wm * 64 + l31) * 16);

  auto epilogue = [&](f32x16 (&acc)[2], int mt_c, int ptile_c, bool yvalid, long ypix, long yunit, float *red_) {
  // epilogue: acc[i][r] = channel mt*128 + wm*64 + i*32 + (r&3) + 8 (r>>2) + 4 half of this lane's pixel
  float sdb[32];
#pragma unroll
  for (int e = 0; e < 32; ++e) sdb[e] = 0.f;
  u64 am[2][4];                                                  // MASKED: the mask operand's pieces, all fetched before the first store
  if (MASKED) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int m4 = mt_c * 128 + wm * 64 + i * 32 + 8 * rq + 4 * half;
        am[i][rq] = yvalid ? reinterpret_cast<const u64 *>(p.ActY)[((yunit + (long)(m4 >> 3) * PQ) << 1) + half] : 0ull;
      }
  }
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
        float x = acc[i][rq * 4 + e] + b4[e];
        x = fmaxf(x, x * p.lrelu);
        if (MASKED) {
          const bf16x4 mk = __builtin_bit_cast(bf16x4, am[i][rq]);
          x = c8_sel_nonpos((float)mk[e], x * p.act_slope, x);
          if (yvalid) sdb[i * 16 + rq * 4 + e] += x;
        }
        v[e] = x;
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
  if (MASKED) {
    // per-channel sums over the workgroup's pixels: butterfly over the 32 pixel lanes of a half, then over the 4 pixel waves
    c8_reduce_scatter32<16>(sdb, l31);                           // lane (half, l31): slot l31 = i*16 + r of its half
    float *red = red_;                                           // [wave 8][half 2][32] in a dead buffer
    red[(wave * 2 + half) * 32 + l31] = sdb[0];
    __syncthreads();
    if (tid < 128) {                                             // (wm, half, slot)
      const int w_m = tid >> 6, hf = (tid >> 5) & 1, qs = tid & 31;
      float t = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) t += red[((w4 * 2 + w_m) * 2 + hf) * 32 + qs];
      const int m = mt_c * 128 + w_m * 64 + (qs >> 4) * 32 + (qs & 3) + 8 * ((qs >> 2) & 3) + 4 * hf;
      p.dbpart[(long)ptile_c * p.M + m] = t;
    }
    __syncthreads();                                             // before the next tile's DMA lands on `red`
  }
  };

  if constexpr (RING) {
    // ---- round 6: three image buffers (requested two stages ahead) + two weight buffers (one stage ahead), counted waits ----
    typedef X3Stage Stage;                                            // a stage = (tile, 16-channel chunk, tap row); wave-uniform
    auto open_tile = [&](Stage &it, int lin_) {
      it.lin = lin_;
      it.valid = decode(lin_, it.mtk, it.ptile) ? 1 : 0;
      it.ch = (it.mtk / MT) * p.kper;
      it.ke = min(nch, it.ch + p.kper);
      it.r = 0;
    };
    auto advance = [&](Stage &it) {                                   // next stage; true when it entered another tile
      if (!it.valid) return false;
      if (++it.r < 3) return false;
      it.r = 0;
      if (++it.ch < it.ke) return false;
      open_tile(it, it.lin + G);
      return it.valid != 0;
    };
    __amdgpu_buffer_rsrc_t xrs_i, wrs_w;
    unsigned voff_i[4];
    int p0_i = 0;
    auto setup_img = [&](const Stage &it) {
      int n0_;
      if (TI == 1) {
        n0_ = it.ptile / p.tiles_per_img;
        p0_i = (it.ptile - n0_ * p.tiles_per_img) * TR;
      } else {
        n0_ = it.ptile * TI;
        p0_i = 0;
      }
      const int nimg_ = min(TI, p.N - n0_);
      xrs_i = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(p.X) + (long)n0_ * 3 * (img_bytes >> 1), 0, nimg_ * 3 * img_bytes,
                                                0x00020000);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int limb = min((wave + 8 * i) / BP, 2);
        voff_i[i] = (pimg[i] >= 0 && pimg[i] < nimg_)
                        ? (unsigned)((pimg[i] * 3 + limb) * img_bytes + 2 * (p0_i + prow[i]) * W16) + pcol[i] : X3_OOB;
      }
    };
    auto setup_w = [&](const Stage &it) {
      wrs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(p.Wq) + (long)(it.mtk % MT) * nch * 3 * (X3F_ASTAGE >> 1), 0,
                                                nch * 3 * X3F_ASTAGE, 0x00020000);
    };
    auto issue_img = [&](const Stage &it, int slot) {
      unsigned char *base = x3_lds + slot * X3R_IMG;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int piece = wave + 8 * i;
        if (piece < BPIECES) {
          unsigned v = voff_i[i];
          if (it.r == 0) v = (p0_i + prow[i] == 0) ? X3_OOB : v - (unsigned)W16;
          if (it.r == 2) v = v + (unsigned)W16;
          if (voff_i[i] == X3_OOB) v = X3_OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs_i, (x3_lds_ptr)(base + piece * 1024), 16, v, it.ch * 2 * HW16, 0, 0);
        }
      }
    };
    auto issue_w = [&](const Stage &it, int slot) {
      unsigned char *base = x3_lds + 3 * X3R_IMG + slot * X3F_ASTAGE;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int piece = wave + 8 * i;
        if (piece < X3F_APIECES)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs_w, (x3_lds_ptr)(base + piece * 1024), 16, voffa[i], (it.ch * 3 + it.r) * X3F_ASTAGE, 0, 0);
      }
    };
    // the stage-end wait: everything but this wave's newest image requests (pieces wave + 8 i < 27: four for waves 0 - 2, three else)
    auto wait_stage = [&](int img_in_flight) {
      if (!img_in_flight)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (wave < BPIECES - 24)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    };
    const unsigned a_off = (unsigned)((half * 128 + wm * 64 + l31) * 16);

    Stage ci, wi, ii;
    open_tile(ci, blockIdx.x);
    if (!ci.valid) return;
    wi = ci;
    ii = ci;
    setup_img(ii);
    setup_w(wi);
    issue_img(ii, 0);
    issue_w(wi, 0);
    if (advance(ii)) setup_img(ii);
    if (ii.valid) issue_img(ii, 1);                                   // stage 1's image
    if (advance(wi)) setup_w(wi);
    wait_stage(ii.valid);
    int t3 = 0, t2 = 0;
    while (true) {
      const int mt_c = ci.mtk % MT, ksp_c = ci.mtk / MT, ptile_c = ci.ptile;
      int n0_c, p0_c;
      if (TI == 1) {
        n0_c = ptile_c / p.tiles_per_img;
        p0_c = (ptile_c - n0_c * p.tiles_per_img) * TR;
      } else {
        n0_c = ptile_c * TI;
        p0_c = 0;
      }
      const bool yvalid = yil < min(TI, p.N - n0_c);
      const long ypix = (long)ksp_c * p.ysplit + (long)(n0_c + yil) * p.M * PQ + (long)(p0_c + ypl) * Q + yql;
      const long yunit = (long)(n0_c + yil) * 3 * (p.M >> 3) * PQ + (long)(p0_c + ypl) * Q + yql;
      f32x16 acc[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      const int nst = (ci.ke - ci.ch) * 3;
      for (int st = 0; st < nst; ++st) {
        if (wi.valid) issue_w(wi, t2 ^ 1);                            // weights of stage t + 1
        if (advance(ii)) setup_img(ii);
        if (ii.valid) issue_img(ii, t3 == 0 ? 2 : t3 - 1);            // image of stage t + 2 -> slot (t + 2) % 3
        const unsigned char *SI = x3_lds + t3 * X3R_IMG, *SA = x3_lds + 3 * X3R_IMG + t2 * X3F_ASTAGE;
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) {
          const int coff = (sx == 0 ? 0 : (sx == 1 ? Q + 1 : 1)) * 16;
          bf16x8 af[2][3], bf[3];
#pragma unroll
          for (int l = 0; l < 3; ++l) {
            bf[l] = *reinterpret_cast<const bf16x8 *>(SI + l * (BP * 1024) + bbase + coff);
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i][l] = *reinterpret_cast<const bf16x8 *>(SA + a_off + ((l * 3 + sx) * 256 + i * 32) * 16);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i] = mfma_split6(af[i][0], af[i][1], af[i][2], bf[0], bf[1], bf[2], acc[i]);
        }
        wait_stage(ii.valid);
        if (advance(wi)) setup_w(wi);
        t3 = t3 == 2 ? 0 : t3 + 1;
        t2 ^= 1;
      }
      // scratch of the MASKED column sums: the image buffer the last stage consumed (its next request goes out in the next stage)
      epilogue(acc, mt_c, ptile_c, yvalid, ypix, yunit, reinterpret_cast<float *>(x3_lds + (t3 == 0 ? 2 : t3 - 1) * X3R_IMG));
      open_tile(ci, ci.lin + G);
      if (!ci.valid) return;
    }
  }
  int lin = blockIdx.x, buf = 0;
  {
    int m_, t_;
    if (!decode(lin, m_, t_)) return;
    setup(m_, t_);
  }
  issue(kc0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  while (true) {
    const int mt_c = mt, ptile_c = ptile, kb = kc0, ke = kc1;
    const bool yvalid = yil < nimg;
    const long ypix = (long)ksp * p.ysplit + (long)(n0 + yil) * p.M * PQ + (long)(p0 + ypl) * Q + yql;   // f32 NCHW: + channel * PQ
    const long yunit = (long)(n0 + yil) * 3 * (p.M >> 3) * PQ + (long)(p0 + ypl) * Q + yql;  // X3: + (limb * M/8 + channel group) * PQ
    int mt_n, ptile_n;
    const bool more = decode(lin + G, mt_n, ptile_n);

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    for (int ch = kb; ch < ke; ++ch) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        if (r < 2) {
          issue(ch, r + 1, buf ^ 1);
        } else if (ch + 1 < ke) {
          issue(ch + 1, 0, buf ^ 1);
        } else if (more) {
          setup(mt_n, ptile_n);
          issue(kc0, 0, buf ^ 1);
        }
        const unsigned char *S = x3_lds + buf * X3F_STAGE;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int coff = (s == 0 ? 0 : (s == 1 ? Q + 1 : 1)) * 16;
          bf16x8 af[2][3], bf[3];
#pragma unroll
          for (int l = 0; l < 3; ++l) {
            bf[l] = *reinterpret_cast<const bf16x8 *>(S + l * (BP * 1024) + bbase + coff);
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

    epilogue(acc, mt_c, ptile_c, yvalid, ypix, yunit, reinterpret_cast<float *>(x3_lds + (buf ^ 1) * X3F_STAGE));
    if (!more) return;
    lin += G;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// transposed direction (Conv2d dgrad, ConvTranspose2d forward): big[n][m][2p+r-1][2q+s-1] += Wt[m][kk][r][s] small[n][kk][p][q].
// Workgroup = 64 m x 256 SMALL pixels x the four output parity classes of a 2x2 block (c8s2_tr_kernel's tiling: tap (r, s)
// feeds class (r != 1, s != 1) from the small pixel shifted by (r == 0, s == 0): no multiply-by-zero work); 8 waves = 2 (m)
// x 4 (pixels), wave tile 32 m x 64 small pixels x 4 classes = 8 accumulator tiles.  A STAGE is (16-channel k-step, tap row
// r): three limbs of the TR small rows p + (r == 0) (one halo column of zeros per row) + three limbs of the 3 taps' weights.
// Epilogue: bias + LeakyReLU; optionally (MASKED) the LeakyReLU BACKWARD of the layer whose output gradient this is (from the hi
// limb of that layer's saved X3 output: sign(x) = sign(hi)) and that layer's bias-gradient partial sums; output f32 NCHW or X3.
// ------------------------------------------------------------------------------------------------------------------
struct X3S2TParams {
  const unsigned short *X;       // small [N][3][Cx/8][P][Q][8]
  const unsigned short *Wq;      // x3s2_pack_kernel<64>'s layout
  const float *bias;             // [M] or null
  float *Y;                      // f32 [N][M][H][W] (OUT3 = false)
  unsigned short *YL;            // [N][3][M/8][H][W][8] (OUT3 = true)
  int N, Cx, M;
  int H, W, P, Q;                // big (output) map H x W, small map P x Q
  int TI, TR;                    // pixel tile of the SMALL map: TI images x TR rows x Q columns = 256 pixels
  int tiles_per_img, ntiles;
  float lrelu;
  const unsigned short *ActY;    // MASKED: X3 saved output of the previous layer (Y's shape); only its hi limb is read
  float act_slope;
  float *dbpart;                 // MASKED: [ntiles][M]
  int ksplit, kper;              // split-K as in X3S2Params (k-steps of 16 channels); raw f32 accumulators at Y + range * ysplit
  long ysplit;
  int linear;                    // as in X3S2Params
};

#define X3T_BP 12                                          // image pieces per limb and stage: <= 768 units (2 k-halves)
#define X3T_AP 6                                           // weight pieces per limb and stage: 3 taps x 2 k-halves x 64 m
#define X3T_BPIECES (3 * X3T_BP)
#define X3T_APIECES (3 * X3T_AP)
#define X3T_ASTAGE (X3T_APIECES * 1024)
#define X3T_STAGE ((X3T_BPIECES + X3T_APIECES) * 1024)     // 55296
#define X3T_LDS_BYTES (2 * X3T_STAGE)                      // 110592

template <bool OUT3, bool MASKED>
__global__ __launch_bounds__(512, 1) void x3s2_tr_kernel(X3S2TParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char x3_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave & 1, wp = wave >> 1;
  typedef unsigned long long u64;

  const int MT = p.M >> 6, KS = p.ksplit, G = gridDim.x, nlin = ((p.ntiles + 7) >> 3) * 8 * MT * KS;
  const int TI = p.TI, TR = p.TR, Q = p.Q;
  const int CB = Q + 1, blk = TR * CB, plane = TI * blk, bunits = 2 * plane;
  const int Q16 = Q * 16, PQ16 = p.P * Q16, img_bytes = (p.Cx >> 3) * PQ16, nks = p.Cx >> 4;
  const long HWl = (long)p.H * p.W;

  int mt, ptile, n0, p0, nimg, kc0, kc1, ksp;
  __amdgpu_buffer_rsrc_t xrs, wrs;
  constexpr int NB = (X3T_BPIECES + 7) / 8, NA = (X3T_APIECES + 7) / 8;      // 5, 3
  unsigned voffb[NB];                                           // source offset for the un-shifted rows (r = 1, 2), or X3_OOB
  unsigned lastrow = 0;                                         // bit i: piece i's unit is in small row P - 1 (dead under r = 0)
  const unsigned voffa = (unsigned)((wave * 64 + lane) * 16);

  auto decode = [&](int lin, int &mt_, int &ptile_) {                // mt_ carries (m tile, k range): mt + MT * range
    if (p.linear) {                                                  // weight-stationary walk (x3_plan in x3.hip): see X3S2Params::linear
      mt_ = lin % (MT * KS);
      ptile_ = lin / (MT * KS);
      return ptile_ < p.ntiles;
    }
    const int xcd = lin & 7, qq = lin >> 3;
    mt_ = qq % (MT * KS);
    ptile_ = xcd + 8 * (qq / (MT * KS));
    return lin < nlin && ptile_ < p.ntiles;
  };
  auto setup = [&](int mt_, int ptile_) {
    ksp = mt_ / MT;
    mt = mt_ - ksp * MT; ptile = ptile_;
    kc0 = ksp * p.kper;
    kc1 = min(nks, kc0 + p.kper);
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
    wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(p.Wq) + (long)mt * nks * 3 * (X3T_ASTAGE >> 1), 0,
                                            nks * 3 * X3T_ASTAGE, 0x00020000);
    lastrow = 0;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int j = wave + 8 * i, limb = min(j / X3T_BP, 2), pc = j % X3T_BP;
      const int u = pc * 64 + lane;
      unsigned v = X3_OOB;
      if (j < X3T_BPIECES && u < bunits) {
        const int kh = u >= plane ? 1 : 0, rem = u - kh * plane;
        const int img = rem / blk, rem2 = rem - img * blk;
        const int ri = rem2 / CB, ci = rem2 - ri * CB;
        const int row = p0 + ri;
        if (row < p.P && ci < Q && img < nimg) {
          v = (unsigned)((img * 3 + limb) * img_bytes + kh * PQ16 + (row * Q + ci) * 16);
          if (row == p.P - 1) lastrow |= 1u << i;
        }
      }
      voffb[i] = v;
    }
  };
  auto issue = [&](int ks, int r, int buf) {
    unsigned char *base = x3_lds + buf * X3T_STAGE;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int piece = wave + 8 * i;
      if (piece < X3T_BPIECES) {
        unsigned v = voffb[i];
        if (r == 0) v = ((lastrow >> i) & 1u) || v == X3_OOB ? X3_OOB : v + (unsigned)Q16;      // rows p + 1
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (x3_lds_ptr)(base + piece * 1024), 16, v, ks * 2 * PQ16, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int piece = wave + 8 * i;
      if (piece < X3T_APIECES)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (x3_lds_ptr)(base + (X3T_BPIECES + piece) * 1024), 16, voffa,
                                                 (ks * 3 + r) * X3T_ASTAGE + i * 8192, 0, 0);
    }
  };

  unsigned bbase[2];
  int ygeo[2];                                                  // image << 20 | row << 10 | column of this lane's two tile pixels
  const int tpi = TR * Q;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int t = wp * 64 + 32 * j + l31;
    const int il = t / tpi, rem = t - il * tpi;
    const int pl = rem / Q, ql = rem - pl * Q;
    bbase[j] = (unsigned)((half * plane + il * blk + pl * CB + ql) * 16);
    ygeo[j] = il << 20 | pl << 10 | ql;
  }
  const unsigned a_base = (unsigned)(X3T_BPIECES * 1024 + (half * 64 + wm * 32 + l31) * 16);

  int lin = blockIdx.x, buf = 0;
  {
    int m_, t_;
    if (!decode(lin, m_, t_)) return;
    setup(m_, t_);
  }
  issue(kc0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  while (true) {
    const int mt_c = mt, ptile_c = ptile, kb = kc0, ke = kc1;
    float *const Ysp = p.Y + (long)ksp * p.ysplit;               // split-K: this range's raw partial output
    long ypix[2];                                                // pixel index of (n, row 2p, column 2q) in an H x W plane set, or -1
    int yn[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int il = ygeo[j] >> 20, pl = (ygeo[j] >> 10) & 1023, ql = ygeo[j] & 1023;
      yn[j] = n0 + il;
      ypix[j] = il < nimg ? (long)(2 * (p0 + pl)) * p.W + 2 * ql : -1;
    }
    int mt_n, ptile_n;
    const bool more = decode(lin + G, mt_n, ptile_n);

    f32x16 acc[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][j][r] = 0.f;

    for (int ks = kb; ks < ke; ++ks) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        if (r < 2) {
          issue(ks, r + 1, buf ^ 1);
        } else if (ks + 1 < ke) {
          issue(ks + 1, 0, buf ^ 1);
        } else if (more) {
          setup(mt_n, ptile_n);
          issue(kc0, 0, buf ^ 1);
        }
        const unsigned char *S = x3_lds + buf * X3T_STAGE;
        bf16x8 bf[2][2][3];                                      // [column shift][j][limb]
#pragma unroll
        for (int sc = 0; sc < 2; ++sc)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int l = 0; l < 3; ++l)
              bf[sc][j][l] = *reinterpret_cast<const bf16x8 *>(S + l * (X3T_BP * 1024) + bbase[j] + sc * 16);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int cls = (r != 1 ? 2 : 0) + (s != 1 ? 1 : 0), sc = s == 0 ? 1 : 0;
          bf16x8 af[3];
#pragma unroll
          for (int l = 0; l < 3; ++l) af[l] = *reinterpret_cast<const bf16x8 *>(S + a_base + ((l * 3 + s) * 128) * 16);
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[cls][j] = mfma_split6(af[0], af[1], af[2], bf[sc][j][0], bf[sc][j][1], bf[sc][j][2], acc[cls][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        buf ^= 1;
      }
    }

    // epilogue.  acc[cls = 2a + b][j][r]: channel mt*64 + wm*32 + (r&3) + 8 (r>>2) + 4 half at output (2p + a, 2q + b)
    float sdb[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) sdb[e] = 0.f;
    // MASKED: the mask operand (hi limb of the saved output: unit (n, limb 0, group, row 2p + a, column 2q + b), this lane's 8 bytes) of
    // channel group rq + 1 is requested before group rq is processed: one workgroup per CU, nothing else covers an epilogue load
    // (-3 ... -6 % per masked launch; all 32 pieces up front: 152 bytes of scratch)
    u64 amk[4][2][2][2];                                          // [rq][a][j][b]
    auto mask_fetch = [&](auto rq_tag) {
      constexpr int rq = decltype(rq_tag)::value;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const long u0 = ((long)yn[j] * 3 * (p.M >> 3) + ((mt_c * 64 + wm * 32 + 8 * rq) >> 3)) * HWl + ypix[j] + (long)a * p.W;
#pragma unroll
          for (int b = 0; b < 2; ++b)
            amk[rq][a][j][b] = ypix[j] < 0 ? 0ull : reinterpret_cast<const u64 *>(p.ActY)[((u0 + b) << 1) + half];
        }
    };
    if (MASKED) mask_fetch(std::integral_constant<int, 0>());
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int m8 = mt_c * 64 + wm * 32 + 8 * rq;                // channel group's first channel
      if (MASKED) {
        if (rq == 0) mask_fetch(std::integral_constant<int, 1>());
        if (rq == 1) mask_fetch(std::integral_constant<int, 2>());
        if (rq == 2) mask_fetch(std::integral_constant<int, 3>());
        __builtin_amdgcn_sched_barrier(0);                        // ... and stays in front of group rq's stores
      }
      f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) b4 = *reinterpret_cast<const f32x4 *>(p.bias + m8 + 4 * half);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (ypix[j] < 0) continue;
          float v[2][4];                                          // [b][e]
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const bf16x4 mk = __builtin_bit_cast(bf16x4, MASKED ? amk[rq][a][j][b] : 0ull);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = acc[a * 2 + b][j][rq * 4 + e] + b4[e];
              x = fmaxf(x, x * p.lrelu);
              if (MASKED) {
                x = c8_sel_nonpos((float)mk[e], x * p.act_slope, x);
                sdb[rq * 4 + e] += x;
              }
              v[b][e] = x;
            }
          }
          if (!OUT3) {
            // f32 NCHW: a lane's (b = 0, 1) pair of one channel is 8 contiguous bytes; 32 lanes = 256 contiguous bytes
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float *dst = Ysp + ((long)yn[j] * p.M + m8 + 4 * half + e) * HWl + ypix[j] + (long)a * p.W;
              *reinterpret_cast<f32x2 *>(dst) = f32x2{v[0][e], v[1][e]};
            }
          } else {
            unsigned w[3][2][2];                                  // [limb][b][dword]
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              bf16x4 h, mi, lo;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                h[e] = (__bf16)v[b][e];
                const float r1 = v[b][e] - (float)h[e];
                mi[e] = (__bf16)r1;
                lo[e] = (__bf16)(r1 - (float)mi[e]);
              }
              const uint2 uh = __builtin_bit_cast(uint2, h), um = __builtin_bit_cast(uint2, mi), ul = __builtin_bit_cast(uint2, lo);
              w[0][b][0] = uh.x; w[0][b][1] = uh.y;
              w[1][b][0] = um.x; w[1][b][1] = um.y;
              w[2][b][0] = ul.x; w[2][b][1] = ul.y;
            }
            // exchange the two column classes across the half-waves (c8s2_tr_kernel): lanes 0-31 store the whole 16-byte unit of
            // column 2q, lanes 32-63 that of column 2q + 1
            const long unit0 = ((long)yn[j] * 3 * (p.M >> 3) + (m8 >> 3)) * HWl + ypix[j] + (long)a * p.W + half;
#pragma unroll
            for (int l = 0; l < 3; ++l) {
#pragma unroll
              for (int d = 0; d < 2; ++d) {
                const auto sw = __builtin_amdgcn_permlane32_swap(w[l][0][d], w[l][1][d], false, false);
                w[l][0][d] = sw[0];
                w[l][1][d] = sw[1];
              }
              reinterpret_cast<u32x4 *>(p.YL)[unit0 + (long)l * (p.M >> 3) * HWl] = u32x4{w[l][0][0], w[l][0][1], w[l][1][0], w[l][1][1]};
            }
          }
        }
    }
    if (MASKED) {
      // 16 channel slots per lane: add the two 16-lane halves of the 32 pixel lanes, butterfly over the remaining four bits,
      // then sum the 4 pixel waves through LDS (the buffer that is dead until the next tile's second stage is requested)
      float sd32[32];
#pragma unroll
      for (int e = 0; e < 16; ++e) sd32[e] = sdb[e] + __shfl_xor(sdb[e], 16, 64);
#pragma unroll
      for (int e = 16; e < 32; ++e) sd32[e] = 0.f;
      c8_reduce_scatter32<8>(sd32, l31);                           // lane: slot (l31 & 15) = r of its half
      float *red = reinterpret_cast<float *>(x3_lds + (buf ^ 1) * X3T_STAGE);   // [wave 8][half 2][16]
      if ((l31 & 16) == 0) red[(wave * 2 + half) * 16 + l31] = sd32[0];
      __syncthreads();
      if (tid < 64) {                                              // (wm, half, slot)
        const int w_m = tid >> 5, hf = (tid >> 4) & 1, qs = tid & 15;
        float t = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) t += red[((w4 * 2 + w_m) * 2 + hf) * 16 + qs];
        const int m = mt_c * 64 + w_m * 32 + (qs & 3) + 8 * (qs >> 2) + 4 * hf;
        p.dbpart[(long)ptile_c * p.M + m] = t;
      }
      __syncthreads();                                             // before the next tile's DMA lands on `red`
    }
    if (!more) return;
    lin += G;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradient: dW[k][c][r][s] = sum_{n,p,q} small[n][k][p][q] big[n][c][2p+r-1][2q+s-1], both operands X3, the reduction
// runs over pixels: fragments by transposing LDS reads (c8wgrad.h, c8s2.h).  Workgroup = 128 k x 64 c x 9 taps for a range of
// pixel chunks; chunk = 16 small pixels (one k-step: TRW rows x QW columns of one image, or TIW whole images) whose big-tensor
// patch ((2 TRW + 1) x (2 QW + 1) input pixels, columns de-interleaved by parity) is staged per channel group; a stage holds the
// three limbs of both operands (<= 55 KB), double buffered.  8 waves = 4 (k) x 2 (c), 9 accumulator tiles each.
// ------------------------------------------------------------------------------------------------------------------
#define X3W_SBYTES 4096                                     // small tensor per limb: 16 k-groups x 16 pixels = 4 DMA pieces
#define X3W_BP 14                                           // big tensor pieces per limb: 8 planes of <= 108 units = 864 units
#define X3W_LIMB (X3W_SBYTES + X3W_BP * 1024)               // 18432
#define X3W_STAGE (3 * X3W_LIMB)                            // 55296
#define X3W_LDS_BYTES (2 * X3W_STAGE)                       // 110592

struct X3S2WParams {
  const unsigned short *S;       // small [N][3][K/8][P][Q][8]
  const unsigned short *B;       // big [N][3][C/8][H][W][8]
  float *part;                   // [splits][9][K][C]
  int N, K, C;
  int H, W, P, Q;
  int TIW, TRW, QW;              // chunk: TIW images x TRW rows x QW columns = 16 small pixels
  int colblocks;                 // Q / QW
  int chunks_per_img, nchunks;   // nchunks = N * chunks_per_img, or ceil(N / TIW)
  int splits, chunks_per_split;
  int bplane;                    // units per channel-group plane of the big tensor's LDS image (4 mod 8)
};

__global__ __launch_bounds__(512, 1) void x3s2_wgrad_kernel(X3S2WParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char x3_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave >> 1, wc = wave & 1;

  const int CT = p.C >> 6, KT = p.K >> 7, tiles = CT * KT;
  const int lin = blockIdx.x, xcd = lin & 7, qq = lin >> 3;
  const int tile = p.splits == 1 ? lin : qq % tiles, split = p.splits == 1 ? 0 : xcd + 8 * (qq / tiles);
  if (split >= p.splits || tile >= tiles) return;
  const int kt = tile / CT, ct = tile - kt * CT;
  const int c0 = split * p.chunks_per_split, c1 = min(p.nchunks, c0 + p.chunks_per_split);

  const int Q = p.Q, TIW = p.TIW, TRW = p.TRW, QW = p.QW;
  const int CB = 2 * QW + 1, blk = (2 * TRW + 1) * CB;          // units per image of a plane
  const int PQ16 = p.P * Q * 16, HW16 = p.H * p.W * 16;
  const int s_img = (p.K >> 3) * PQ16, b_img = (p.C >> 3) * HW16;  // bytes per image and limb
  const int tpi = TRW * QW;                                     // chunk pixels per image

  // DMA pieces: small tensor 4 per limb (piece q = k-groups 4 q .. 4 q + 3 x 16 pixels), big tensor <= 14 per limb; a wave's
  // list covers all three limbs: small pieces j = wave + 8 i (i < 2, j < 12: limb j / 4, q = j % 4), big pieces j = wave + 8 i
  // (i < 6, j < 42: limb j / 14).  LDS-DMA lands lane-linear (slot = lane), the SOURCE address is free: slot sigma of a small piece
  // holds (pixel quad sigma / 16, k-group (sigma / 4) % 4, pixel sigma % 4 of the quad), i.e. byte quad * 256 + group * 64 +
  // pixel * 16: the four k-groups a transposing read phase touches (same quad) fill one 256-byte row = all 64 banks once.
  int voffs[2], voffb[6];
  unsigned topm = 0, leftm = 0;                                 // bit i: big piece i's unit lies in the patch's first row / first column
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = wave + 8 * i, limb = min(j >> 2, 2), q4 = j & 3;
    const int t = (lane >> 4) * 4 + (lane & 3), kg = q4 * 4 + ((lane >> 2) & 3);
    const int il = t / tpi, rem = t - il * tpi, pl = rem / QW, ql = rem - pl * QW;
    voffs[i] = (il * 3 + limb) * s_img + (kt * 16 + kg) * PQ16 + (pl * Q + ql) * 16;
  }
  const int bunits = 8 * p.bplane;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int j = wave + 8 * i, limb = min(j / X3W_BP, 2), pc = j % X3W_BP;
    const int u = pc * 64 + lane;
    int v = (int)X3_OOB;
    if (j < 3 * X3W_BP && u < bunits) {
      const int cg = u / p.bplane, rem = u - cg * p.bplane;
      if (rem < TIW * blk) {
        const int il = rem / blk, rem2 = rem - il * blk;
        const int ri = rem2 / CB, ci = rem2 - ri * CB;
        const int col = ci <= QW ? 2 * ci - 1 : 2 * (ci - QW - 1);    // relative to 2 q0; -1 = the left neighbour column
        v = (il * 3 + limb) * b_img + (ct * 8 + cg) * HW16 + (ri * p.W + col) * 16;   // row = 2 p0 - 1 + ri, column = 2 q0 + col:
        if (ri == 0) topm |= 1u << i;                                                  // the chunk's origin is added per chunk
        if (ci == 0) leftm |= 1u << i;
      }
    }
    voffb[i] = v;
  }
  auto issue = [&](int chunk, int stage) {
    int n, p0, q0;
    if (TIW == 1) {
      n = chunk / p.chunks_per_img;
      const int rem = chunk - n * p.chunks_per_img;
      const int rb = rem / p.colblocks;
      p0 = rb * TRW;
      q0 = (rem - rb * p.colblocks) * QW;
    } else {
      n = chunk * TIW;
      p0 = 0;
      q0 = 0;
    }
    const int nimg = min(TIW, p.N - n);
    const c8_i32x4 srs = c8_rsrc_words(p.S + (long)n * 3 * (s_img >> 1), (unsigned)(nimg * 3 * s_img));
    const c8_i32x4 brs = c8_rsrc_words(p.B + (long)n * 3 * (b_img >> 1), (unsigned)(nimg * 3 * b_img));
    const unsigned base = c8_lds_addr(x3_lds) + stage * X3W_STAGE;
    const int sdelta = (p0 * Q + q0) * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = wave + 8 * i;
      if (j < 12) c8_dma16_asm(srs, base + (j >> 2) * X3W_LIMB + (j & 3) * 1024, (unsigned)(voffs[i] + sdelta), 0);
    }
    const int bdelta = ((2 * p0 - 1) * p.W + 2 * q0) * 16;      // first staged row is 2 p0 - 1, column origin 2 q0
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int j = wave + 8 * i;
      if (j < 3 * X3W_BP) {
        const bool dead = voffb[i] == (int)X3_OOB || (p0 == 0 && ((topm >> i) & 1u)) || (q0 == 0 && ((leftm >> i) & 1u));
        c8_dma16_asm(brs, base + (j / X3W_BP) * X3W_LIMB + X3W_SBYTES + (j % X3W_BP) * 1024, dead ? X3_OOB : (unsigned)(voffb[i] + bdelta),
                     0);
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // transposing-read addresses (c8wgrad.h): 16-lane group g: channels 16 (g & 1) .., pixels 8 (g >> 1) ..; lane i16 points at
  // pixel + i16 / 4 (second read: + 4 = the next pixel quad = + 256 bytes), channel group 2 (g & 1) + (i16 % 4) / 2, byte 8 (i16 & 1)
  const int g = lane >> 4, i16 = lane & 15;
  const int cgl = 2 * (g & 1) + ((i16 & 3) >> 1), px = 8 * (g >> 1) + (i16 >> 2), byte = 8 * (i16 & 1);
  const unsigned a_base = (unsigned)(wk * 1024 + 2 * (g >> 1) * 256 + cgl * 64 + (i16 >> 2) * 16 + byte);
  auto bunit = [&](int t) -> int {                              // chunk pixel t -> unit of its (2p - 1, 2q - 1) corner in a plane
    const int il = t / tpi, rem = t - il * tpi;
    const int pl = rem / QW, ql = rem - pl * QW;
    return il * blk + 2 * pl * CB + ql;
  };
  const unsigned b_base0 = (unsigned)(X3W_SBYTES + ((wc * 4 + cgl) * p.bplane + bunit(px)) * 16 + byte);
  const unsigned b_base1 = (unsigned)(X3W_SBYTES + ((wc * 4 + cgl) * p.bplane + bunit(px + 4)) * 16 + byte);

  if (c1 > c0) {
    issue(c0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  for (int it = c0; it < c1; ++it) {
    const int stage = (it - c0) & 1;
    if (it + 1 < c1) issue(it + 1, stage ^ 1);
    const unsigned char *St = x3_lds + stage * X3W_STAGE;
    bf16x8 af[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) af[l] = c8_tr_frag(St + l * X3W_LIMB + a_base, St + l * X3W_LIMB + a_base + 256);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int r = t / 3, s = t % 3;
      const int o = (r * CB + (s == 0 ? 0 : (s == 1 ? QW + 1 : 1))) * 16;
      bf16x8 bf[3];
#pragma unroll
      for (int l = 0; l < 3; ++l) bf[l] = c8_tr_frag(St + l * X3W_LIMB + b_base0 + o, St + l * X3W_LIMB + b_base1 + o);
      acc[t] = mfma_split6(af[0], af[1], af[2], bf[0], bf[1], bf[2], acc[t]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  const int l31 = lane & 31, half = lane >> 5;
  float *out = p.part + ((long)split * 9) * p.K * p.C + (long)(kt * 128 + wk * 32 + 4 * half) * p.C + ct * 64 + wc * 32 + l31;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(long)t * p.K * p.C + (long)((r & 3) + 8 * (r >> 2)) * p.C] = acc[t][r];
}

// dW[k][c][t] = sum_s part[s][t][k][c]   (c8_wgrad_reduce_kernel's twin in this translation unit)
__global__ __launch_bounds__(256) void x3_wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ dW, int KC, int splits) {
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
  for (int e = tid; e < 576; e += 256) {
    const int ll = e / 9, t = e - ll * 9;
    if (blockIdx.x * 64 + ll < KC)
      dW[(long)blockIdx.x * 576 + e] = (red[0][t][ll] + red[1][t][ll]) + (red[2][t][ll] + red[3][t][ll]);
  }
}

// Column sums of part[rows][C] (bias-gradient partial sums of the MASKED kernels: up to thousands of rows).  Stage 1: grid
// (ceil(C / 64), chunks): part2[chunk][c] = sum of the chunk's rows (thread = (row lane 0..3, channel)); stage 2 (the same kernel
// with one chunk): out[c] = sum of part2's rows.  Fixed summation order.
__global__ __launch_bounds__(256) void x3_colsum_kernel(const float *__restrict__ part, float *__restrict__ out, int C, int rows,
                                                        int rows_per_chunk) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6, chunk = blockIdx.y;
  const int r0 = chunk * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  float s = 0.f;
  if (c < C)
    for (int r = r0 + rl; r < r1; r += 4) s += part[(long)r * C + c];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < C) out[(long)chunk * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------------------------
// LeakyReLU backward from the OUTPUT + the layer's bias gradient, f32 NCHW in, X3 out: g = dy * (y > 0 ? 1 : slope) written as
// three limbs (the operand of the X3 weight-gradient / dgrad kernels), dbpart[split][c] = sum over the split's images and pixels
// of g.  The X3 twin of act_bwd_bias_kernel (norm_act.hip) for a layer whose output left the X3 family as f32 NCHW.
// grid = (C / 8, splits); slope < 0: no activation (g = dy).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void x3_act_bwd_bias_nchw_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                                   unsigned short *__restrict__ g, float *__restrict__ dbpart, int N, int C,
                                                                   int HW, int imgs_per_split, float slope) {
  __shared__ float red[4][8];
  const int cg = blockIdx.x, split = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = split * imgs_per_split, n1 = min(N, n0 + imgs_per_split);
  const long cgs = C >> 3, ls = cgs * HW * 8;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  // (image, pixel) pairs of the split flattened: small maps (2x2 ... 8x8 at the end of the discriminator) keep every thread busy
  const long items = (long)(n1 - n0) * HW;
  for (long it = tid; it < items; it += 256) {
    const int n = n0 + (int)(it / HW), u = (int)(it - (long)(n - n0) * HW);
    const float *dyp = dy + ((long)n * C + cg * 8) * HW, *yp = y + ((long)n * C + cg * 8) * HW;
    unsigned short *gp = g + ((long)n * 3 * cgs + cg) * HW * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d = dyp[(long)e * HW + u];
      v[e] = slope < 0.f ? d : c8_sel_nonpos(yp[(long)e * HW + u], d * slope, d);
      s[e] += v[e];
    }
    bf16x8 h, m, l;
    split3(v, h, m, l);
    *reinterpret_cast<bf16x8 *>(gp + (long)u * 8) = h;
    *reinterpret_cast<bf16x8 *>(gp + ls + (long)u * 8) = m;
    *reinterpret_cast<bf16x8 *>(gp + 2 * ls + (long)u * 8) = l;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    s[e] = wave_sum(s[e]);
    if (lane == 0) red[wave][e] = s[e];
  }
  __syncthreads();
  if (tid < 8 && dbpart) dbpart[(long)split * C + cg * 8 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// ------------------------------------------------------------------------------------------------------------------
// Epilogue of a split-K launch (x3s2_fwd_kernel / x3s2_tr_kernel with ksplit > 1): v = sum of the `ns` raw partial outputs
// part[s][N][C][HW] (f32) + bias, LeakyReLU (lrelu: max(v, v * lrelu)), optionally the fused mask of the layer in front
// (v *= LeakyReLU'(ActY), ActY = hi limb of that layer's X3 output, + its bias-gradient partial sums), written as f32 NCHW (y) or
// as limbs (yl).  grid = (C / 8, ranges of the flattened (image, pixel) space); dbpart[range][C].
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void x3_splitk_finish_kernel(const float *__restrict__ part, int ns, long sstride,
                                                               const float *__restrict__ bias, float lrelu,
                                                               const unsigned short *__restrict__ act_y, float act_slope,
                                                               float *__restrict__ y, unsigned short *__restrict__ yl, float *__restrict__ dbpart,
                                                               int N, int C, int HW, long items_per_split) {
  __shared__ float red[4][8];
  const int cg = blockIdx.x, split = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long cgs = C >> 3, ls = cgs * HW * 8;
  float s[8], b8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    s[e] = 0.f;
    b8[e] = bias ? bias[cg * 8 + e] : 0.f;
  }
  // (image, pixel) pairs flattened and cut into gridDim.y ranges: small maps keep every thread busy, large ones every CU
  const long it0 = (long)split * items_per_split, it1 = min((long)N * HW, it0 + items_per_split);
  for (long it = it0 + tid; it < it1; it += 256) {
    const int n = (int)(it / HW), u = (int)(it - (long)n * HW);
    const float *pp = part + ((long)n * C + cg * 8) * HW;
    const long ubase = ((long)n * 3 * cgs + cg) * HW;            // unit index of (n, limb 0, cg, pixel 0)
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = b8[e];
      for (int k = 0; k < ns; ++k) a += pp[(long)k * sstride + (long)e * HW + u];
      v[e] = fmaxf(a, a * lrelu);
    }
    if (act_y) {
      const bf16x8 mk = *reinterpret_cast<const bf16x8 *>(act_y + (ubase + u) * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = c8_sel_nonpos((float)mk[e], v[e] * act_slope, v[e]);
        s[e] += v[e];
      }
    }
    if (y) {
#pragma unroll
      for (int e = 0; e < 8; ++e) y[((long)n * C + cg * 8 + e) * HW + u] = v[e];
    }
    if (yl) {
      bf16x8 h, m, l;
      split3(v, h, m, l);
      unsigned short *gp = yl + (ubase + u) * 8;
      *reinterpret_cast<bf16x8 *>(gp) = h;
      *reinterpret_cast<bf16x8 *>(gp + ls) = m;
      *reinterpret_cast<bf16x8 *>(gp + 2 * ls) = l;
    }
  }
  if (!dbpart) return;                                            // uniform
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    s[e] = wave_sum(s[e]);
    if (lane == 0) red[wave][e] = s[e];
  }
  __syncthreads();
  if (tid < 8) dbpart[(long)split * C + cg * 8 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

}  // namespace lsps
#endif
