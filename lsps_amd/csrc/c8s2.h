// bf16 stride-2 3x3 convs (pad 1) and their transposes on activations in the C8 channel-group layout (c8conv.h): the
// generator's down / up-sampling layers, the discriminator's second front conv and its whole trunk in BASELINE config 5
// (reference: src/trainers/common_net.py:246-268 LeakyReLUConv2d / LeakyReLUConvTranspose2d, lsps_nets.py:117-124, 186-192,
// 222-225).  One geometry-generic kernel per direction:
//   c8s2_fwd_kernel    small[n][m][p][q] = act(b[m] + sum_{kk,r,s} Wt[m][kk][r][s] big[n][kk][2p+r-1][2q+s-1])
//                      Conv2d forward; ConvTranspose2d dgrad
//   c8s2_tr_kernel     big[n][m][2p+r-1][2q+s-1] += Wt[m][kk][r][s] small[n][kk][p][q]   (+ bias, act)
//                      Conv2d dgrad; ConvTranspose2d forward
//   c8s2_wgrad_kernel  dW[k][c][r][s] = sum_{n,p,q} small[n][k][p][q] big[n][c][2p+r-1][2q+s-1]
//                      Conv2d wgrad (small = dY, big = X); ConvTranspose2d wgrad (small = X, big = dY)
// with big = [N][./8][H][W][8], small = [N][./8][H/2][W/2][8] bf16, H, W even powers of two.
#ifndef LSPS_C8S2_H
#define LSPS_C8S2_H
#include "c8conv.h"
#include "c8wgrad.h"

namespace lsps {

// -------------------------------------------------------------------------------------------
// Common idea (as in c8conv.h): staging is LDS-DMA of 16-byte channel-group units whose per-lane SOURCE address does every
// gather — here also the stride: the forward kernel's LDS image of a tile is DE-INTERLEAVED by row and column parity
// (odd rows first, then even rows; inside a row odd columns, then even columns), so that the input pixel of output (p, q)
// under tap (r, s) sits at   lanebase(p, q) + rowoff[r] * CB + coloff[s]   units with rowoff = {0, TR + 1, 1}, coloff =
// {0, Q + 1, 1}, CB = 2 Q + 1: a tap is a wave-uniform offset, a fragment of 32 consecutive output pixels is 32
// consecutive units (ds_read_b128, conflict-free).  Padding = lanes whose buffer offset is out of range (hardware zeros).
// Pixel tiles are runs of consecutive (n, p, q): TR rows of one image when a map has >= the tile's pixels, else TI whole
// images — so ONE kernel covers 64x64 ... 2x2 output maps (the discriminator trunk's maps are 16x16 ... 2x2).
// -------------------------------------------------------------------------------------------
struct C8S2Params {
  const unsigned short *X;       // fwd: big [N][Cx/8][H][W][8];  tr: small [N][Cx/8][P][Q][8]
  const unsigned short *Wq;      // c8s2_pack_kernel's layout
  const float *bias;             // [M] or null
  unsigned short *Y;             // fwd: small [N][M/8][P][Q][8];  tr: big [N][M/8][H][W][8]
  int N, Cx, M;
  int H, W, P, Q;                // big map H x W, small map P x Q = H/2 x W/2
  int TI, TR;                    // pixel tile (of the SMALL map): TI images x TR rows x Q columns
  int tiles_per_img, ntiles;
  float lrelu;                   // epilogue: v = max(v, v * lrelu) (1 = no activation; 0 <= slope <= 1)
  // dgrad entries with the PREVIOUS layer's LeakyReLU backward fused (the layer whose output this gradient belongs to):
  // y = y * (ActY > 0 ? 1 : act_slope) with ActY that layer's saved output (Y's shape), and dbpart[pixel tile][M] = the
  // per-channel sums of the masked result over the workgroup's pixels (that layer's bias gradient, summed by c8_colsum_stage1_kernel)
  const unsigned short *ActY;
  float act_slope;
  float *dbpart;
};

struct C8S2Pack {
  const float *W;
  unsigned short *Wq;
  int M, C, BM;                  // output channels, reduction channels, m tile (64 | 128)
  long sm, sc;                   // element strides of m and of the reduction channel in W; tap t = 3 r + s at offset t
};

// Wq[m tile][chunk of 16 c][tap][k-half][BM][8 c].  One thread = the 8 channels x 9 taps of one (m, k-half): nine 16-byte stores,
// contiguous over the BM threads of an (m tile, chunk, tap, k-half) (as x3s2_pack_kernel: one thread per ELEMENT took 60 us for the
// deepest discriminator layer, twice per step)
__global__ __launch_bounds__(256) void c8s2_pack_kernel(C8S2Pack p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;      // over [mt][chunk][kh][BM]
  const long total = (long)p.M * p.C >> 3;
  if (idx >= total) return;
  const int ml = (int)(idx % p.BM);
  long rest = idx / p.BM;
  const int kh = (int)(rest & 1);
  rest >>= 1;
  const int chunks = p.C >> 4;
  const int chunk = (int)(rest % chunks), mt = (int)(rest / chunks);
  const int m = mt * p.BM + ml, c0 = chunk * 16 + kh * 8;
  const float *src = p.W + (long)m * p.sm + (long)c0 * p.sc;
  float x[8][9];
  if (p.sc == 9 && !(reinterpret_cast<uintptr_t>(src) & 15)) {       // one 288-byte run
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
  const long base = ((long)mt * chunks + chunk) * 9;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    bf16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (__bf16)x[e][t];
    *reinterpret_cast<bf16x8 *>(p.Wq + ((((base + t) * 2 + kh) * p.BM + ml) << 3)) = h;
  }
}

#define C8S2_OOB 0x80000000u

// ------------------------------------------------------------------------------------------------------------------
// forward direction: workgroup = 128 m x (128 NJ) output pixels; 8 waves = 2 (m) x 4 (pixels), wave tile 64 m x 32 NJ pixels
// ------------------------------------------------------------------------------------------------------------------
#define C8S2F_BPIECES 42                                   // LDS room for the image part of a stage (2688 units)
#define C8S2F_APIECES 36                                   // 9 taps x 2 k-halves x 128 m = 2304 units
#define C8S2F_ACHUNK (C8S2F_APIECES * 64 * 8)              // bf16 elements of packed weights per (m tile, chunk)
#define C8S2F_STAGE ((C8S2F_BPIECES + C8S2F_APIECES) * 1024)
#define C8S2F_LDS_BYTES (2 * C8S2F_STAGE)                  // 159744

template <int NJ>
__global__ __launch_bounds__(512, 1) void c8s2_fwd_kernel(C8S2Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s2_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave & 1, wp = wave >> 1;
  typedef unsigned long long u64;

  // Persistent workgroups: workgroup b walks the tiles lin = b, b + grid, ... (grid is a multiple of 8, so lin & 7 — the XCD the
  // hardware placed it on — stays put).  lin -> (pixel tile, m tile): the m tiles of a pixel tile are consecutive on ONE XCD (its
  // L2 serves the re-reads).  The first chunk of the next tile is requested during the last chunk of the current one, and the
  // epilogue's stores drain under the next tile's MFMAs: with one workgroup per CU nothing else would overlap them
  // (profiles/r3g_c8_ablations.txt: a third of the kernel on the 64-channel layers).
  const int MT = p.M >> 7, G = gridDim.x, nlin = ((p.ntiles + 7) >> 3) * 8 * MT;
  const int TI = p.TI, TR = p.TR, Q = p.Q;
  const int CB = 2 * Q + 1, blk = (2 * TR + 1) * CB, plane = TI * blk, bunits = 2 * plane;
  const int HW16 = p.H * p.W * 16, img_bytes = (p.Cx >> 3) * HW16, nch = p.Cx >> 4;
  const int PQ = p.P * Q, tpi = TR * Q;
  const bool masked = p.ActY != nullptr;                        // uniform

  int mt, ptile, n0, p0, nimg;                                  // the tile whose DMA set-up is current
  __amdgpu_buffer_rsrc_t xrs, wrs;
  unsigned voffb[6], voffa[5];
  // per-lane geometry of the image pieces that does not depend on the tile (piece wave + 8 i, i < 6: up to 48 >= 42)
  int pimg[6], prow[6];                                         // image within the tile, input row relative to 2 p0  (or -1: dead lane)
  unsigned pcol[6];                                             // byte offset of (k half, column) within an image row set
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int u = (wave + 8 * i) * 64 + lane;
    pimg[i] = -1; prow[i] = 0; pcol[i] = 0;
    if (u < bunits) {
      const int kh = u >= plane ? 1 : 0, rem = u - kh * plane;
      const int img = rem / blk, rem2 = rem - img * blk;
      const int ri = rem2 / CB, ci = rem2 - ri * CB;
      const int col = ci <= Q ? 2 * ci - 1 : 2 * (ci - Q - 1);
      if (col >= 0 && col < p.W) {
        pimg[i] = img;
        prow[i] = ri <= TR ? 2 * ri - 1 : 2 * (ri - TR - 1);
        pcol[i] = (unsigned)(kh * HW16 + col * 16);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) voffa[i] = (unsigned)(((wave + 8 * i) * 64 + lane) * 16);
  const int bpieces = (bunits + 63) >> 6;

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
    xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(p.X) + (long)n0 * (img_bytes >> 1), 0, nimg * img_bytes,
                                            0x00020000);
    wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(p.Wq) + (long)mt * nch * C8S2F_ACHUNK, 0,
                                            nch * C8S2F_ACHUNK * 2, 0x00020000);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int row = 2 * p0 + prow[i];
      voffb[i] = (pimg[i] >= 0 && pimg[i] < nimg && row >= 0 && row < p.H)
                     ? (unsigned)(pimg[i] * img_bytes + row * p.W * 16) + pcol[i] : C8S2_OOB;
    }
  };
  auto issue = [&](int ch, int stage) {
    unsigned char *base = s2_lds + stage * C8S2F_STAGE;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int piece = wave + 8 * i;
      if (piece < bpieces)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (c8_lds_ptr)(base + piece * 1024), 16, voffb[i], ch * 2 * HW16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int piece = wave + 8 * i;
      if (piece < C8S2F_APIECES)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (c8_lds_ptr)(base + (C8S2F_BPIECES + piece) * 1024), 16, voffa[i],
                                                 ch * (C8S2F_ACHUNK * 2), 0, 0);
    }
  };

  // this lane's output pixels: tile pixel t = wp * 32 NJ + 32 j + l31 -> (image, row, column) of the tile
  unsigned bbase[NJ];
  int yil[NJ], yoff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int t = wp * (32 * NJ) + 32 * j + l31;
    const int il = t / tpi, rem = t - il * tpi;
    const int pl = rem / Q, ql = rem - pl * Q;
    bbase[j] = (unsigned)((half * plane + il * blk + pl * CB + ql) * 16);
    yil[j] = il;
    yoff[j] = pl * Q + ql;
  }
  const unsigned a_base = (unsigned)(C8S2F_BPIECES * 1024 + (half * 128 + wm * 64 + l31) * 16);

  int lin = blockIdx.x, stage = 0;
  {
    int m_, t_;
    if (!decode(lin, m_, t_)) return;
    setup(m_, t_);
  }
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  while (true) {
    // the current tile's output addressing (the DMA set-up moves on to the next tile during the last chunk)
    const int mt_c = mt, ptile_c = ptile;
    long ypix[NJ];                                               // unit index of (n, channel group 0, pixel) in Y, or -1
#pragma unroll
    for (int j = 0; j < NJ; ++j) ypix[j] = yil[j] < nimg ? (long)(n0 + yil[j]) * (p.M >> 3) * PQ + p0 * Q + yoff[j] : -1;
    int mt_n, ptile_n;
    const bool more = decode(lin + G, mt_n, ptile_n);

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int ch = 0; ch < nch; ++ch) {
#ifndef C8S2_ABL_NODMA                   // ablation builds: tools/build_abl_c8.sh
      if (ch + 1 < nch) {
        issue(ch + 1, stage ^ 1);
      } else if (more) {
        setup(mt_n, ptile_n);
        issue(0, stage ^ 1);
      }
#endif
      const unsigned char *S = s2_lds + stage * C8S2F_STAGE;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int r = t / 3, s = t % 3;
        const int toff = ((r == 0 ? 0 : (r == 1 ? TR + 1 : 1)) * CB + (s == 0 ? 0 : (s == 1 ? Q + 1 : 1))) * 16;
        bf16x8 af[2], bf[NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(S + a_base + (t * 256 + i * 32) * 16);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[j] = *reinterpret_cast<const bf16x8 *>(S + bbase[j] + toff);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      stage ^= 1;
    }
    // here: `stage` holds the next tile's first chunk (if any); stage ^ 1 is dead until the next tile's second chunk is requested

#ifdef C8S2_ABL_NOEPI
    {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) t += acc[i][j][r];
      if (t == 1.2345e30f) p.Y[0] = 1;
      if (!more) return;
      lin += G;
      continue;
    }
#endif
    // epilogue: acc[i][j][r] = channel mt*128 + wm*64 + i*32 + (r&3) + 8 (r>>2) + 4 half of pixel j; a register quad = 8 bytes
    u64 ay[2][4][NJ];                                              // the mask operand's pieces, all fetched before the first store
    if (masked) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int m4 = mt_c * 128 + wm * 64 + i * 32 + 8 * rq + 4 * half;
            ay[i][rq][j] =
                ypix[j] >= 0 ? reinterpret_cast<const u64 *>(p.ActY)[((ypix[j] + (long)(m4 >> 3) * PQ) << 1) + half] : 0ull;
          }
    }
    float sdb[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) sdb[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int m4 = mt_c * 128 + wm * 64 + i * 32 + 8 * rq + 4 * half;
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) b4 = *reinterpret_cast<const f32x4 *>(p.bias + m4);
        u64 pk[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          bf16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = acc[i][j][rq * 4 + e] + b4[e];
            x = fmaxf(x, x * p.lrelu);
            if (masked) {
              const bf16x4 a = __builtin_bit_cast(bf16x4, ay[i][rq][j]);
              x = c8_sel_nonpos((float)a[e], x * p.act_slope, x);
            }
            v[e] = (__bf16)x;
            if (masked && ypix[j] >= 0) sdb[i * 16 + rq * 4 + e] += (float)v[e];
          }
          pk[j] = __builtin_bit_cast(u64, v);
        }
        if (NJ == 2) {
          // the two half-waves hold channels 4 half .. of BOTH pixels: exchange (v_permlane32_swap) so that lanes 0-31 store the
          // whole 16-byte unit of pixel j = 0 and lanes 32-63 that of j = 1: 8 store instructions of 16 B instead of 16 of 8 B
          typedef unsigned s2_u32x2 __attribute__((ext_vector_type(2)));
          s2_u32x2 A = __builtin_bit_cast(s2_u32x2, pk[0]), B = __builtin_bit_cast(s2_u32x2, pk[NJ - 1]);
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const auto sw = __builtin_amdgcn_permlane32_swap(A[d], B[d], false, false);
            A[d] = sw[0];
            B[d] = sw[1];
          }
          const long yp = half ? ypix[NJ - 1] : ypix[0];
          if (yp >= 0) reinterpret_cast<u32x4 *>(p.Y)[yp + (long)(m4 >> 3) * PQ] = u32x4{A[0], A[1], B[0], B[1]};
        } else {
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            if (ypix[j] >= 0) reinterpret_cast<u64 *>(p.Y)[((ypix[j] + (long)(m4 >> 3) * PQ) << 1) + half] = pk[j];
        }
      }
    if (masked) {
      // per-channel sums over the workgroup's pixels: butterfly over the 32 pixel lanes of a half, then over the 4 pixel waves
      c8_reduce_scatter32<16>(sdb, l31);                           // lane (half, l31): slot l31 = i*16 + r of its half
      float *red = reinterpret_cast<float *>(s2_lds + (stage ^ 1) * C8S2F_STAGE);   // [wave 8][half 2][32] in the dead stage
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
    if (!more) return;
    lin += G;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// transposed direction: workgroup = 64 m x 256 SMALL pixels (= 1024 output pixels: the four parity classes of a 2x2 block);
// 8 waves = 2 (m) x 4 (pixels), wave tile 32 m x 64 small pixels x 4 classes (8 accumulator tiles).  Tap (r, s) feeds class
// (a, b) = (r != 1, s != 1) from the small pixel shifted by (r == 0, s == 0): 9 MFMAs per (m tile, pixel tile, k-step), no
// multiply-by-zero work.  LDS image: (TR + 1) x (Q + 1) units per image and k-half (one halo row / column of zeros).
// Chunks of 32 reduction channels (2 k-steps) per stage.
// ------------------------------------------------------------------------------------------------------------------
#define C8S2T_KS 2                                          // k-steps (16 channels) per chunk
#define C8S2T_BPIECES (18 * C8S2T_KS)                       // image part: <= 1152 units per k-step
#define C8S2T_APIECES (18 * C8S2T_KS)                       // 9 taps x 2 k-halves x 64 m = 1152 units per k-step
#define C8S2T_ACHUNK16 (1152 * 8)                           // bf16 elements of packed weights per (m tile, 16 channels)
#define C8S2T_STAGE ((C8S2T_BPIECES + C8S2T_APIECES) * 1024)
#define C8S2T_LDS_BYTES (2 * C8S2T_STAGE)                   // 147456

template <bool MASKED>
__global__ __launch_bounds__(512, 1) void c8s2_tr_kernel(C8S2Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s2_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave & 1, wp = wave >> 1;
  typedef unsigned long long u64;

  // persistent workgroups, as c8s2_fwd_kernel: tiles lin = blockIdx.x, + grid, ...; the next tile's first chunk is requested
  // during the last chunk of the current one and the epilogue's stores drain under the next tile's MFMAs
  const int MT = p.M >> 6, G = gridDim.x, nlin = ((p.ntiles + 7) >> 3) * 8 * MT;
  const int TI = p.TI, TR = p.TR, Q = p.Q;
  const int CB = Q + 1, blk = (TR + 1) * CB, plane = TI * blk, bunits = 2 * C8S2T_KS * plane;   // 2 KS k-half planes
  const int PQ16 = p.P * Q * 16, img_bytes = (p.Cx >> 3) * PQ16, nch = p.Cx / (16 * C8S2T_KS);
  constexpr bool masked = MASKED;                               // p.ActY != nullptr

  int mt, ptile, n0, p0, nimg;                                  // the tile whose DMA set-up is current
  __amdgpu_buffer_rsrc_t xrs, wrs;
  constexpr int NB = (C8S2T_BPIECES + 7) / 8, NA = (C8S2T_APIECES + 7) / 8;
  unsigned voffb[NB];                                           // recomputed per tile (registers are scarce here: 128 accumulators)
  const unsigned voffa = (unsigned)((wave * 64 + lane) * 16);   // weight piece wave + 8 i: + i * 8192 through the scalar offset
  const int bpieces = (bunits + 63) >> 6;

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
    xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(p.X) + (long)n0 * (img_bytes >> 1), 0, nimg * img_bytes,
                                            0x00020000);
    wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(p.Wq) + (long)mt * (p.Cx >> 4) * C8S2T_ACHUNK16, 0,
                                            (p.Cx >> 4) * C8S2T_ACHUNK16 * 2, 0x00020000);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int u = (wave + 8 * i) * 64 + lane;
      unsigned v = C8S2_OOB;
      if (u < bunits) {
        const int kh = u / plane, rem = u - kh * plane;
        const int img = rem / blk, rem2 = rem - img * blk;
        const int ri = rem2 / CB, ci = rem2 - ri * CB;
        const int row = p0 + ri;
        if (row < p.P && ci < Q && img < nimg) v = (unsigned)(img * img_bytes + kh * PQ16 + (row * Q + ci) * 16);
      }
      voffb[i] = v;
    }
  };
  auto issue = [&](int ch, int stage) {
    unsigned char *base = s2_lds + stage * C8S2T_STAGE;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int piece = wave + 8 * i;
      if (piece < bpieces)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (c8_lds_ptr)(base + piece * 1024), 16, voffb[i], ch * (2 * C8S2T_KS) * PQ16, 0,
                                                 0);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int piece = wave + 8 * i;
      if (piece < C8S2T_APIECES)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (c8_lds_ptr)(base + (C8S2T_BPIECES + piece) * 1024), 16, voffa,
                                                 ch * (C8S2T_KS * C8S2T_ACHUNK16 * 2) + i * 8192, 0, 0);
    }
  };

  unsigned bbase[2];
  int ygeo[2];                                                  // image << 20 | row << 10 | column of this lane's two tile pixels
  const int tpi = TR * Q;
  const long HWl = (long)p.H * p.W;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int t = wp * 64 + 32 * j + l31;
    const int il = t / tpi, rem = t - il * tpi;
    const int pl = rem / Q, ql = rem - pl * Q;
    bbase[j] = (unsigned)((half * plane + il * blk + pl * CB + ql) * 16);
    ygeo[j] = il << 20 | pl << 10 | ql;
  }
  const unsigned a_base = (unsigned)(C8S2T_BPIECES * 1024 + (half * 64 + wm * 32 + l31) * 16);

  int lin = blockIdx.x, stage = 0;
  {
    int m_, t_;
    if (!decode(lin, m_, t_)) return;
    setup(m_, t_);
  }
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  while (true) {
    const int mt_c = mt, ptile_c = ptile;
    long ypix[2];                                                // unit index of (n, channel group 0, row 2p, column 2q) in Y, or -1
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int il = ygeo[j] >> 20, pl = (ygeo[j] >> 10) & 1023, ql = ygeo[j] & 1023;
      ypix[j] = il < nimg ? (long)(n0 + il) * (p.M >> 3) * HWl + (long)(2 * (p0 + pl)) * p.W + 2 * ql : -1;
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

    for (int ch = 0; ch < nch; ++ch) {
#ifndef C8S2_ABL_NODMA
      if (ch + 1 < nch) {
        issue(ch + 1, stage ^ 1);
      } else if (more) {
        setup(mt_n, ptile_n);
        issue(0, stage ^ 1);
      }
#endif
      const unsigned char *S = s2_lds + stage * C8S2T_STAGE;
#pragma unroll
      for (int ks = 0; ks < C8S2T_KS; ++ks) {
        bf16x8 bf[4][2];
#pragma unroll
        for (int sh = 0; sh < 4; ++sh)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            bf[sh][j] = *reinterpret_cast<const bf16x8 *>(S + bbase[j] + (ks * 2 * plane + (sh >> 1) * CB + (sh & 1)) * 16);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int r = t / 3, s = t % 3;
          const int cls = (r != 1 ? 2 : 0) + (s != 1 ? 1 : 0), sh = (r == 0 ? 2 : 0) + (s == 0 ? 1 : 0);
          const bf16x8 af = *reinterpret_cast<const bf16x8 *>(S + a_base + (ks * 1152 + t * 128) * 16);
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[cls][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[sh][j], acc[cls][j], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      stage ^= 1;
    }
    // here: `stage` holds the next tile's first chunk (if any); stage ^ 1 is dead until the next tile's second chunk is requested

#ifdef C8S2_ABL_NOEPI
    {
      float t = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) t += acc[c][j][r];
      if (t == 1.2345e30f) p.Y[0] = 1;
      if (!more) return;
      lin += G;
      continue;
    }
#endif
    // epilogue.  acc[cls][j][r]: channel mt*64 + wm*32 + (r&3) + 8 (r>>2) + 4 half at output (2p + a, 2q + b).  The two column
    // classes of a row are exchanged across the half-waves (v_permlane32_swap: cdna_hip_programming.md T21) so that lanes 0-31
    // store the whole 16-byte unit of column 2q and lanes 32-63 that of column 2q + 1: 1 KB contiguous per wave instruction.
    u64 ay[4][2][2][2];                                            // [rq][a][j][b]: the mask operand's pieces, fetched before the first store
    if (masked) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              const int m8 = mt_c * 64 + wm * 32 + 8 * rq;
              ay[rq][a][j][b] =
                  ypix[j] >= 0
                      ? reinterpret_cast<const u64 *>(p.ActY)[((ypix[j] + (long)(m8 >> 3) * HWl + (long)a * p.W + b) << 1) + half]
                      : 0ull;
            }
    }
    float sdb[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) sdb[e] = 0.f;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int m8 = mt_c * 64 + wm * 32 + 8 * rq;                // channel group's first channel
      f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) b4 = *reinterpret_cast<const f32x4 *>(p.bias + m8 + 4 * half);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          unsigned w[2][2];                                        // [b][dword]
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = acc[a * 2 + b][j][rq * 4 + e] + b4[e];
              x = fmaxf(x, x * p.lrelu);
              if (masked) {
                const bf16x4 m_ = __builtin_bit_cast(bf16x4, ay[rq][a][j][b]);
                x = c8_sel_nonpos((float)m_[e], x * p.act_slope, x);
              }
              v[e] = (__bf16)x;
              if (masked && ypix[j] >= 0) sdb[rq * 4 + e] += (float)v[e];
            }
            const uint2 u = __builtin_bit_cast(uint2, v);
            w[b][0] = u.x;
            w[b][1] = u.y;
          }
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const auto sw = __builtin_amdgcn_permlane32_swap(w[0][d], w[1][d], false, false);
            w[0][d] = sw[0];
            w[1][d] = sw[1];
          }
          if (ypix[j] < 0) continue;
          const long unit = ypix[j] + (long)(m8 >> 3) * HWl + (long)a * p.W + half;
          u32x4 o = {w[0][0], w[0][1], w[1][0], w[1][1]};
          reinterpret_cast<u32x4 *>(p.Y)[unit] = o;
        }
    }
    if (masked) {
      // 16 channel slots per lane: add the two 16-lane halves of the 32 pixel lanes, butterfly over the remaining four bits,
      // then sum the 4 pixel waves through LDS
#pragma unroll
      for (int e = 0; e < 16; ++e) sdb[e] += __shfl_xor(sdb[e], 16, 64);
      c8_reduce_scatter32<8>(sdb, l31);                            // lane: slot (l31 & 15) = r of its half
      float *red = reinterpret_cast<float *>(s2_lds + (stage ^ 1) * C8S2T_STAGE);   // [wave 8][half 2][16] in the dead stage
      if ((l31 & 16) == 0) red[(wave * 2 + half) * 16 + l31] = sdb[0];
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
// weight gradient: workgroup = 128 k x 64 c x 9 taps for a range of pixel chunks ("split"); chunk = 64 small pixels (TRW rows
// of one image, or TIW whole images) = 4 k-steps of 16 pixels; both operands by transposing LDS reads (c8wgrad.h).  The big
// tensor's rows are staged with their columns de-interleaved by parity (odd columns first), so the 4 consecutive pixels a
// 16-lane group addresses are 64 contiguous bytes of a channel-group plane under every tap.
// ------------------------------------------------------------------------------------------------------------------
#define C8S2W_SPLANE 1088                                   // bytes per k-group plane of the small tensor: 64 pixels + 64 (bank rotation)
#define C8S2W_SBYTES (16 * C8S2W_SPLANE)                    // 17408
#define C8S2W_BPIECES 51                                    // big tensor: 8 planes of <= 404 units
#define C8S2W_STAGE (C8S2W_SBYTES + C8S2W_BPIECES * 1024)   // 69632
#define C8S2W_LDS_BYTES (2 * C8S2W_STAGE)                   // 139264

struct C8S2WParams {
  const unsigned short *S;       // small [N][K/8][P][Q][8]
  const unsigned short *B;       // big [N][C/8][H][W][8]
  float *part;                   // [splits][9][K][C]
  int N, K, C;
  int H, W, P, Q;
  int TIW, TRW;                  // chunk: TIW images x TRW rows x Q columns = 64 small pixels
  int chunks_per_img, nchunks;   // nchunks = total chunks (N * chunks_per_img, or ceil(N / TIW))
  int splits, chunks_per_split;
  int bplane;                    // units per channel-group plane of the big tensor's LDS image (padded to 4 mod 8)
};

__global__ __launch_bounds__(512, 1) void c8s2_wgrad_kernel(C8S2WParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s2_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave >> 1, wc = wave & 1;

  const int CT = p.C >> 6, KT = p.K >> 7, tiles = CT * KT;
  const int lin = blockIdx.x, xcd = lin & 7, qq = lin >> 3;
  // splits == 1 (>= 256 output tiles: the last trunk layer): one workgroup per tile over all XCDs, no partial sums to add up
  const int tile = p.splits == 1 ? lin : qq % tiles, split = p.splits == 1 ? 0 : xcd + 8 * (qq / tiles);
  if (split >= p.splits || tile >= tiles) return;
  const int kt = tile / CT, ct = tile - kt * CT;
  const int c0 = split * p.chunks_per_split, c1 = min(p.nchunks, c0 + p.chunks_per_split);

  const int Q = p.Q, TIW = p.TIW, TRW = p.TRW;
  const int CB = 2 * Q + 1, blk = (2 * TRW + 1) * CB;          // units per image of a plane
  const int PQ = p.P * Q, PQ16 = PQ * 16, HW16 = p.H * p.W * 16;
  const int s_img = (p.K >> 3) * PQ16, b_img = (p.C >> 3) * HW16;  // bytes per image
  const int tpi = TRW * Q;                                      // chunk pixels per image

  // DMA pieces: small tensor 16 (one per k group; piece = wave + 8 i, i < 2), big tensor <= 51 (i < 7)
  int voffs[2], voffb[7];
  unsigned topm = 0;                                            // bit i: piece i's unit lies in the chunk's first row (row 2 p0 - 1)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int kg = wave + 8 * i, il = lane / tpi, pix = lane - il * tpi;
    voffs[i] = il * s_img + (kt * 16 + kg) * PQ16 + pix * 16;
  }
  const int bunits = 8 * p.bplane;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int u = (wave + 8 * i) * 64 + lane;
    int v = (int)C8S2_OOB;
    if (u < bunits) {
      const int cg = u / p.bplane, rem = u - cg * p.bplane;
      if (rem < TIW * blk) {
        const int il = rem / blk, rem2 = rem - il * blk;
        const int ri = rem2 / CB, ci = rem2 - ri * CB;
        const int col = ci <= Q ? 2 * ci - 1 : 2 * (ci - Q - 1);
        if (col >= 0) {
          v = il * b_img + (ct * 8 + cg) * HW16 + (ri * p.W + col) * 16;   // row = 2 p0 - 1 + ri: the -1 row is added per chunk
          if (ri == 0) topm |= 1u << i;
        }
      }
    }
    voffb[i] = v;
  }
  const int bpieces = (bunits + 63) >> 6;
  auto issue = [&](int chunk, int stage) {
    int n, p0;
    if (TIW == 1) {
      n = chunk / p.chunks_per_img;
      p0 = (chunk - n * p.chunks_per_img) * TRW;
    } else {
      n = chunk * TIW;
      p0 = 0;
    }
    const int nimg = min(TIW, p.N - n);
    // requests issued from inline asm (c8conv.h, c8_dma16_asm): hipcc would wait for them in front of the next transposing read
    const c8_i32x4 srs = c8_rsrc_words(p.S + (long)n * (s_img >> 1), (unsigned)(nimg * s_img));
    const c8_i32x4 brs = c8_rsrc_words(p.B + (long)n * (b_img >> 1), (unsigned)(nimg * b_img));
    const unsigned base = c8_lds_addr(s2_lds) + stage * C8S2W_STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) c8_dma16_asm(srs, base + (wave + 8 * i) * C8S2W_SPLANE, (unsigned)(voffs[i] + p0 * Q * 16), 0);
    const int delta = (2 * p0 - 1) * p.W * 16;                  // first staged row is 2 p0 - 1 (out of the image for p0 = 0)
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int piece = wave + 8 * i;
      if (piece < bpieces) {
        const bool dead = voffb[i] < 0 || (p0 == 0 && ((topm >> i) & 1u));
        c8_dma16_asm(brs, base + C8S2W_SBYTES + piece * 1024, dead ? C8S2_OOB : (unsigned)(voffb[i] + delta), 0);
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // transposing-read addresses (c8wgrad.h): 16-lane group g: channels 16 (g & 1) .., pixels 8 (g >> 1) ..; lane i16 points at
  // pixel + i16 / 4 (second read: + 4), channel group 2 (g & 1) + (i16 % 4) / 2, byte 8 (i16 & 1)
  const int g = lane >> 4, i16 = lane & 15;
  const int cgl = 2 * (g & 1) + ((i16 & 3) >> 1), px = 8 * (g >> 1) + (i16 >> 2), byte = 8 * (i16 & 1);
  const unsigned a_base = (unsigned)((wk * 4 + cgl) * C8S2W_SPLANE + px * 16 + byte);
  auto bunit = [&](int t) -> int {                              // chunk pixel t -> unit of its (2p, 2q - 1) corner in a plane
    const int il = t / tpi, rem = t - il * tpi;
    const int pl = rem / Q, ql = rem - pl * Q;
    return il * blk + 2 * pl * CB + ql;
  };
  const unsigned b_base0 = (unsigned)(C8S2W_SBYTES + ((wc * 4 + cgl) * p.bplane + bunit(px)) * 16 + byte);
  const unsigned b_base1 = (unsigned)(C8S2W_SBYTES + ((wc * 4 + cgl) * p.bplane + bunit(px + 4)) * 16 + byte);
  int ksoff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) ksoff[ks] = __builtin_amdgcn_readfirstlane(bunit(16 * ks) * 16);

  if (c1 > c0) {
    issue(c0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  for (int it = c0; it < c1; ++it) {
    const int stage = (it - c0) & 1;
#ifndef C8S2_ABL_NODMA
    if (it + 1 < c1) issue(it + 1, stage ^ 1);
#endif
    const unsigned char *St = s2_lds + stage * C8S2W_STAGE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 af = c8_tr_frag(St + a_base + ks * 256, St + a_base + ks * 256 + 64);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int r = t / 3, s = t % 3;
        const int o = ksoff[ks] + (r * CB + (s == 0 ? 0 : (s == 1 ? Q + 1 : 1))) * 16;
        const bf16x8 bf = c8_tr_frag(St + b_base0 + o, St + b_base1 + o);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[t], 0, 0, 0);
      }
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

// ------------------------------------------------------------------------------------------------------------------
// LeakyReLU backward from the OUTPUT + the layer's bias gradient in one pass over C8 tensors (act_bwd_bias_kernel of
// norm_act.hip in this layout): g = dy * (y > 0 ? 1 : slope), dbpart[split][c] = sum over the split's images and pixels of g.
// grid = (C / 8, splits); the partial sums are added up by c8_colsum_stage1_kernel.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void c8_act_bwd_bias_kernel(const unsigned short *__restrict__ dy, const unsigned short *__restrict__ y,
                                                              unsigned short *__restrict__ g, float *__restrict__ dbpart, int N, int C,
                                                              int HW, int imgs_per_split, float slope) {
  __shared__ float red[4][8];
  const int cg = blockIdx.x, split = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = split * imgs_per_split, n1 = min(N, n0 + imgs_per_split);
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  for (int n = n0; n < n1; ++n) {
    const long base = ((long)n * (C >> 3) + cg) * HW;
    for (int u = tid; u < HW; u += 256) {
      const bf16x8 dv = *reinterpret_cast<const bf16x8 *>(dy + (base + u) * 8);
      const bf16x8 yv = *reinterpret_cast<const bf16x8 *>(y + (base + u) * 8);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)dv[e];
        const float v = c8_sel_nonpos((float)yv[e], d * slope, d);
        o[e] = (__bf16)v;
        s[e] += (float)o[e];                                     // the sum of what the weight-gradient kernels will read
      }
      *reinterpret_cast<bf16x8 *>(g + (base + u) * 8) = o;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    s[e] = wave_sum(s[e]);
    if (lane == 0) red[wave][e] = s[e];
  }
  __syncthreads();
  if (tid < 8) dbpart[(long)split * C + cg * 8 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// part2[chunk][c] = sum of rows [chunk * rows_per_chunk, ...) of part[rows][C]: a stage of a column sum over many rows (the last
// stage: one chunk)
// (per-workgroup bias-gradient partials: up to thousands of rows); grid (ceil(C / 64), chunks), thread = (row lane 0..3, channel)
__global__ __launch_bounds__(256) void c8_colsum_stage1_kernel(const float *__restrict__ part, float *__restrict__ part2, int C, int rows,
                                                               int rows_per_chunk) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6, chunk = blockIdx.y;
  const int r0 = chunk * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  float s = 0.f;
  if (c < C)
    for (int r = r0 + rl; r < r1; r += 4) s += part[(long)r * C + c];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < C) part2[(long)chunk * C + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

}  // namespace lsps
#endif
