// Weight gradient of the 3x3 / stride 1 / pad 1 conv on 32-wide feature maps in Winograd F(4x4,3x3) form (the adjoint of
// conv_wino4.h):  with Y = A^T [ (G g G^T) . (B^T d B) ] A  the gradient of g is
//   dg = G^T [ (A dY A^T) . (B^T d B) ] G        summed over tiles and images,
// i.e. 36 independent GEMMs  M_p[k][c] += sum_tiles T_p[k][tile] V_p[c][tile]  with 36 multiplies per (k, c, 4x4 tile)
// instead of 144 (F(2x2,3x3), conv_wino.h: 64).
//   T = A dY A^T  (4x4 -> 6x6)  and  V = B^T d B  (6x6 -> 6x6): per lane, in registers, from raw dy / x rows in LDS
//   M_p accumulates in registers over the workgroup's share of the tile rows; partial sums go to the workspace
//   dg = G^T M G and the sum over workgroups: wino4_w3x3_reduce_kernel (double precision inside)
// Workgroup = 64 k x 32 c x all 36 positions on 256 threads: ONE wave per SIMD, wave = one 3x3 block of the 6x6 positions
// (wp = 2 bi + bj, as in the forward kernel) with 2 (k halves) x 9 accumulator tiles = 288 registers (256 AGPR + 32 VGPR).
// MFMA operand layout (32x32x2): lane l supplies A[k = l%32][tile = l/32] and B[tile = l/32][c = l%32]: a lane transforms
// the dy tile of ITS k (both halves) and the x tile of ITS c, each value feeds exactly one MFMA.  With 64 k per wave the x
// transform (48 ops) is shared by two MFMAs per position: 104 VALU ops per 18 MFMAs (an 8-wave / 9-accumulator split as in
// the forward kernel would need 76 per 9).  Both transforms run row-first so that a raw LDS row is consumed as soon as it
// has arrived (few live registers), and are issued between the MFMAs of the PREVIOUS k-step (order pinned with
// sched_barrier; plain f32 VALU only: packed ops are the expensive fillers beside MFMAs on this part).
// One chunk = one tile row of one image = 8 tiles = 4 k-steps: x rows 4ty-1 .. 4ty+4 of 32 channels, dy rows 4ty .. 4ty+3
// of 64 channels, double-buffered in LDS; the rows of the next chunk travel global -> registers -> LDS spread over the gaps
// of the current one (x: requested one chunk ahead in step 3, stored in step 1; dy: requested in step 0, stored in step 2).
// LDS channel strides are 4 * odd floats: the 16 lanes of a ds_read_b128 group hold 16 different channels at the same
// pixel and land on 16 different bank quads.
#ifndef LSPS_CONV_WINO4W_H
#define LSPS_CONV_WINO4W_H
#include "conv_wino4.h"

namespace lsps {

#define W4W_XS 244                               // floats per staged x channel: 6 rows x W4_LDW + 4
#define W4W_DS 132                               // floats per staged dy channel: 4 rows x 32 + 4
#define W4W_XBUF (32 * W4W_XS)
#define W4W_BUF (W4W_XBUF + 64 * W4W_DS)         // floats per LDS buffer

// three rows (B = 0: rows 0..2, B = 1: rows 3..5) of A = (A^T)^T applied to one dy column / row of four values:
//   A = [1 0 0 0; 1 1 1 1; 1 -1 1 -1; 1 2 4 8; 1 -2 4 -8; 0 0 0 1]
template <int B>
__device__ __forceinline__ void w4_at(float e0, float e1, float e2, float e3, float &o0, float &o1, float &o2) {
  if (B == 0) {
    const float s = e0 + e2, d = e1 + e3;
    o0 = e0;
    o1 = s + d;
    o2 = s - d;
  } else {
    const float a = fmaf(4.f, e2, e0), b = fmaf(4.f, e3, e1);
    o0 = fmaf(2.f, b, a);
    o1 = fmaf(-2.f, b, a);
    o2 = e3;
  }
}

// 18 accumulator tiles = 288 registers: hipcc keeps every MFMA result in AGPRs once a kernel may use them and would shuttle
// the 32 registers beyond the 256 AGPRs through v_accvgpr_read/write around each MFMA (waiting out the full MFMA latency
// every time), so the MFMAs are written in asm with the register class spelled out: 16 tiles in AGPRs, 2 in VGPRs.  The
// hazard recognizer does not see into asm: the operands of every MFMA here are produced at least one MFMA gap earlier, and
// the accumulators are only read after the s_nops that follow the main loop.
template <bool AGPR>
__device__ __forceinline__ void w4w_mfma(float a, float b, f32x16 &c) {
  if (AGPR)
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// KH = k halves per wave.  KH = 2: the design above (256 threads, one wave per SIMD, 18 accumulator tiles per wave).
// KH = 1: 512 threads, two waves per SIMD, wave = (position block, k half) with 9 accumulator tiles: the x transform is
// computed by both k-half waves (76 instead of 52 VALU per 9 MFMAs), but a filler beside an f32 MFMA costs ~2.2 cycles with
// two waves on the SIMD against ~4 with one (tools/mfma_valu_overlap.hip).
template <int KH>
__global__ __launch_bounds__(KH == 2 ? 256 : 512, KH == 2 ? 1 : 2) void wino4_w3x3_kernel(Wino4WParams p) {
  constexpr int NT = KH == 2 ? 256 : 512, XSEG = 1536 / NT, DSEG = 2048 / NT;
  extern __shared__ __attribute__((aligned(16))) float w4w_lds[];
  typedef const volatile f32x4 __attribute__((address_space(3))) *lp4;
  typedef const volatile f32x2 __attribute__((address_space(3))) *lp2;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef W4_PRIO_YOUNG
  if (KH == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
#ifdef W4_PRIO_OLD
  if (KH == 1 && wave < 4) __builtin_amdgcn_s_setprio(1);
#endif
  const int l31 = lane & 31, half = lane >> 5;
  const int wp = wave & 3, kh0 = KH == 1 ? wave >> 2 : 0;
  // XCD-aware mapping (workgroups go to the 8 XCDs round-robin in launch order): all (k, c) blocks of one split of the
  // tile rows sit on ONE XCD, so its dy / x rows come from HBM once and from that XCD's L2 for the other blocks
  int cb = blockIdx.x, kb = blockIdx.y, z = blockIdx.z;
  if ((gridDim.z & 7) == 0) {
    const int nb = gridDim.x * gridDim.y;
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int xcd = lin & 7, q = lin >> 3, b = q % nb;
    z = xcd + 8 * (q / nb);
    cb = b % gridDim.x;
    kb = b / gridDim.x;
  }
  const int H = p.H, trows = H >> 2;                   // tile rows per image
  const int t0 = z * p.per_split, t1 = min(p.ntr, t0 + p.per_split);

  // staging: x 32 c x 6 rows x 8 segments = XSEG per thread (channel (t/8) % 32, segment t%8, rows xrow0 + i (KH = 2) or
  // xrow0 + 2i (KH = 1, xrow0 = t/256)), dy 64 k x 4 rows x 8 segments = DSEG per thread (KH = 2: channel t/8 + 32 (i&1), row
  // i/2; KH = 1: channel t/8, row i).  Buffer descriptors: uniform base + one 32-bit lane offset
  const int sc = (tid >> 3) & 31, sseg = tid & 7, xrow0 = KH == 2 ? 0 : tid >> 8;
  auto xrow = [&](int i) { return KH == 2 ? i : xrow0 + 2 * i; };
  const unsigned sv = (unsigned)(sc * H * 32 + sseg * 4) * 4u;
  const unsigned svd = (unsigned)((tid >> 3) * H * 32 + sseg * 4) * 4u;
  const int xs_lds = sc * W4W_XS + 1 + sseg * 4;                       // + row * W4_LDW
  const int ds_lds = W4W_XBUF + (tid >> 3) * W4W_DS + sseg * 4;        // KH = 2: + (i & 1) * 32 * W4W_DS + (i >> 1) * 32; KH = 1: + i * 32
  const int dk_off = H * 32 * 32 * 4;                                  // bytes between dy channels k and k + 32
  f32x4 xr[XSEG], dr[DSEG];
  bool zt = false, zb = false;       // (uniform) x row 0 / 5 of the chunk requested last (= the one stored next) is outside
  __amdgpu_buffer_rsrc_t xrs, drs;
  // tile row (within its image) and byte offsets of the chunk requested next, advanced incrementally and branch-free; the
  // last chunk is requested again instead of running past the end
  int rq_t = t0, rq_ty = t0 % trows;
  long rq_x = ((((long)(t0 / trows) * p.C + cb * 32) * H + 4 * rq_ty - 1) * 32) * 4;
  long rq_d = ((((long)(t0 / trows) * p.M + kb * 64) * H + 4 * rq_ty) * 32) * 4;
  const long jump_x = ((long)p.C * H * 32 - (trows - 1) * 128) * 4, jump_d = ((long)p.M * H * 32 - (trows - 1) * 128) * 4;
  auto chunk_base = [&]() {
    zt = rq_ty == 0;
    zb = rq_ty == trows - 1;
    xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.X) + rq_x), 0, 0x7fffffff, 0x00020000);
    drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.DY) + rq_d), 0, 0x7fffffff, 0x00020000);
  };
  auto chunk_advance = [&]() {
    const bool adv = rq_t + 1 < t1, wrap = rq_ty + 1 == trows;
    rq_t += adv ? 1 : 0;
    rq_x += adv ? (wrap ? jump_x : 512) : 0;
    rq_d += adv ? (wrap ? jump_d : 512) : 0;
    rq_ty = adv ? (wrap ? 0 : rq_ty + 1) : rq_ty;
  };
  // rows -1 / H of the image do not exist: the load is redirected to the neighbouring (valid) row and the registers are
  // replaced by zeros when they are stored (selects on a uniform condition: no branch, the chunk loop stays one basic
  // block, which is what keeps the sched_barrier-pinned order and the asm MFMAs' operand distances intact)
  auto load_x = [&](int i) {
#ifdef W4W_ABL_NOSTAGE
    return;
#endif
    const int r = xrow(i);
    if (KH == 2) {
      const int so = i == 0 ? (zt ? 128 : 0) : (i == 5 ? (zb ? -128 : 0) : 0);
      xr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, sv + i * 128, so, 0));
    } else {
      const int adj = (r == 0 && zt) ? 128 : ((r == 5 && zb) ? -128 : 0);
      xr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, sv + r * 128 + adj, 0, 0));
    }
  };
  auto load_d = [&](int i) {
#ifdef W4W_ABL_NOSTAGE
    return;
#endif
    if (KH == 2)
      dr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(drs, svd + (i >> 1) * 128, (i & 1) * dk_off, 0));
    else
      dr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(drs, svd + i * 128, 0, 0));
  };
  auto store_x = [&](float *buf, int i) {
#ifdef W4W_ABL_NOSTAGE
    return;
#endif
    const int r = xrow(i);
    float *d = buf + xs_lds + r * W4_LDW;
    f32x4 v = xr[i];
    if (KH == 2) {
      if (i == 0) v = zt ? f32x4{0.f, 0.f, 0.f, 0.f} : v;
      if (i == 5) v = zb ? f32x4{0.f, 0.f, 0.f, 0.f} : v;
    } else {
      v = ((r == 0 && zt) || (r == 5 && zb)) ? f32x4{0.f, 0.f, 0.f, 0.f} : v;
    }
    d[0] = v[0];
    *reinterpret_cast<f32x2 *>(d + 1) = f32x2{v[1], v[2]};
    d[3] = v[3];
  };
  auto store_d = [&](float *buf, int i) {
#ifdef W4W_ABL_NOSTAGE
    return;
#endif
    const int off = KH == 2 ? (i & 1) * 32 * W4W_DS + (i >> 1) * 32 : i * 32;
    *reinterpret_cast<f32x4 *>(buf + ds_lds + off) = dr[i];
  };

  // halo columns of the x rows (index 0 and 33): zero once in both buffers
  for (int u = tid; u < 2 * 32 * 6 * 2; u += NT) {
    const int b = u / 384, rr = (u % 384) >> 1;
    w4w_lds[b * W4W_BUF + (rr / 6) * W4W_XS + (rr % 6) * W4_LDW + (u & 1) * 33] = 0.f;
  }

  auto body = [&](auto bi_c, auto bj_c) {
    constexpr int BI = decltype(bi_c)::value, BJ = decltype(bj_c)::value;
    f32x16 acc[KH][9];
#pragma unroll
    for (int h = 0; h < KH; ++h)
#pragma unroll
      for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][q][r] = 0.f;
    float T[2][KH][9], V[2][9];                  // [set][k half][position], [set][position]
    float P[5][3], Q[KH][4][3];
    f32x4 rx4[5], rd4[KH][4];
    f32x2 rx2[5];
    const float *rdx0 = w4w_lds + l31 * W4W_XS + BI * W4_LDW + 4 * half;   // + buffer, + 8 s (tile 2s + half), + row * W4_LDW
    const float *rdd0 = w4w_lds + W4W_XBUF + (kh0 * 32 + l31) * W4W_DS + 4 * half;   // + buffer, + 8 s, + 32 * W4W_DS (k half), + row * 32

    // the next k-step's operands, in units that fit one MFMA gap
    auto RX = [&](const float *rx, int sn, int l) {                        // raw x row l of the block's five
#ifdef W4W_ABL_NOLDS
      return;
#endif
      rx4[l] = *(lp4)(rx + 8 * sn + l * W4_LDW);
      rx2[l] = *(lp2)(rx + 8 * sn + l * W4_LDW + 4);
    };
    auto RD = [&](const float *rd, int sn, int kh, int l) {                // raw dy row l of k half kh
#ifdef W4W_ABL_NOLDS
      return;
#endif
      rd4[kh][l] = *(lp4)(rd + 8 * sn + kh * 32 * W4W_DS + l * 32);
    };
    auto XR = [&](int l) {                                                 // row l: the block's three columns of d B
#ifdef W4W_ABL_NOXF
      return;
#endif
      auto c = [&](int x) -> float { return x < 4 ? rx4[l][x] : rx2[l][x - 4]; };
      w4_xf<BJ>(c(BJ), c(BJ + 1), c(BJ + 2), c(BJ + 3), c(BJ + 4), P[l][0], P[l][1], P[l][2]);
    };
    auto XC = [&](int set, int jl) {                                       // column jl: the block's three rows of B^T (d B)
#ifdef W4W_ABL_NOXF
      return;
#endif
      w4_xf<BI>(P[0][jl], P[1][jl], P[2][jl], P[3][jl], P[4][jl], V[set][jl], V[set][3 + jl], V[set][6 + jl]);
    };
    auto DR = [&](int kh, int l) {                                         // dy row l: the block's three columns of dY A^T
#ifdef W4W_ABL_NOXF
      return;
#endif
      w4_at<BJ>(rd4[kh][l][0], rd4[kh][l][1], rd4[kh][l][2], rd4[kh][l][3], Q[kh][l][0], Q[kh][l][1], Q[kh][l][2]);
    };
    auto DC = [&](int set, int kh, int jl) {                               // column jl: the block's three rows of A (dY A^T)
#ifdef W4W_ABL_NOXF
      return;
#endif
      w4_at<BI>(Q[kh][0][jl], Q[kh][1][jl], Q[kh][2][jl], Q[kh][3][jl], T[set][kh][jl], T[set][kh][3 + jl], T[set][kh][6 + jl]);
    };

    // One k-step s: 18 MFMAs on operand set SET.  The gaps between them carry (a) the transforms of the raw rows already in
    // registers into set SET ^ 1 (the operands of k-step s + 1), (b) in the last third, once those rows are consumed, the LDS
    // reads of the raw rows of k-step s + 2 (tile pair sn of the buffer behind rx / rd): a wave is alone on its SIMD, so a
    // read must be many gaps old when it is first used, or the wait stalls the matrix pipe, and (c) a share of the staging of
    // the next chunks (STAGE 0: x registers -> LDS buffer sb; 1: dy registers -> sb; 2: request x rows; 3: request dy rows).
    // STAGE 2 also carries the workgroup barrier (chunk j + 1 published) right before its reads, which are the first ones
    // of that buffer.
    auto step = [&](auto set_c, auto stage_c, const float *rx, const float *rd, int sn, float *sb) {
      constexpr int SET = decltype(set_c)::value, NS = SET ^ 1, STAGE = decltype(stage_c)::value;
#define W4W_SB __builtin_amdgcn_sched_barrier(0)
      if constexpr (KH == 1) {
        // two waves per SIMD, 9 MFMAs per k-step (compiler-allocated accumulators: the builtin's hazards are the compiler's)
#define W4W_MMA1(m)                                                                                                       \
  W4W_SB;                                                                                                                 \
  acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(T[SET][0][m], V[SET][m], acc[0][m], 0, 0, 0);                          \
  W4W_SB
        // reads of the raw rows of k-step s + 1 (tile pair sn behind rx / rd) two gaps ahead of their transform: with two
        // waves per SIMD the partner covers what is left of the LDS latency, and the raw registers live for half a k-step
        // only (144 accumulator + 112 other registers is all a wave has at this occupancy)
        if (STAGE == 3) __syncthreads();             // chunk j + 1 published: this step's reads are the first ones of its buffer
        W4W_MMA1(0);
        RX(rx, sn, 0);
        RX(rx, sn, 1);
        if (STAGE == 0) store_x(sb, 0);
        if (STAGE == 1) store_d(sb, 0);
        if (STAGE == 2) load_x(0);
        if (STAGE == 3) load_d(0);
        W4W_MMA1(1);
        RX(rx, sn, 2);
        RX(rx, sn, 3);
        if (STAGE == 0) store_x(sb, 1);
        if (STAGE == 1) store_d(sb, 1);
        if (STAGE == 2) load_x(1);
        if (STAGE == 3) load_d(1);
        W4W_MMA1(2);
        XR(0);
        XR(1);
        RX(rx, sn, 4);
        if (STAGE == 0) store_x(sb, 2);
        if (STAGE == 1) store_d(sb, 2);
        if (STAGE == 2) load_x(2);
        if (STAGE == 3) load_d(2);
        W4W_MMA1(3);
        XR(2);
        XR(3);
        RD(rd, sn, 0, 0);
        RD(rd, sn, 0, 1);
        if (STAGE == 1) store_d(sb, 3);
        if (STAGE == 3) load_d(3);
        W4W_MMA1(4);
        XR(4);
        XC(NS, 0);
        RD(rd, sn, 0, 2);
        RD(rd, sn, 0, 3);
        W4W_MMA1(5);
        XC(NS, 1);
        XC(NS, 2);
        W4W_MMA1(6);
        DR(0, 0);
        DR(0, 1);
        DR(0, 2);
        W4W_MMA1(7);
        DR(0, 3);
        DC(NS, 0, 0);
        DC(NS, 0, 1);
        W4W_MMA1(8);
        DC(NS, 0, 2);
        W4W_SB;
#undef W4W_MMA1
      } else {
#define W4W_MMA(m)                                                                                                        \
  W4W_SB;                                                                                                                 \
  w4w_mfma<((m) < 16)>(T[SET][(m) & 1][(m) >> 1], V[SET][(m) >> 1], acc[(m) & 1][(m) >> 1]);                              \
  W4W_SB
      W4W_MMA(0);
      XR(0);
      if (STAGE == 0) store_x(sb, 0);
      if (STAGE == 1) store_d(sb, 0);
      if (STAGE == 2) load_x(0);
      if (STAGE == 3) load_d(0);
      W4W_MMA(1);
      XR(1);
      if (STAGE == 1) store_d(sb, 1);
      if (STAGE == 3) load_d(1);
      W4W_MMA(2);
      XR(2);
      if (STAGE == 0) store_x(sb, 1);
      if (STAGE == 2) load_x(1);
      if (STAGE == 3) load_d(2);
      W4W_MMA(3);
      XR(3);
      if (STAGE == 1) store_d(sb, 2);
      W4W_MMA(4);
      XR(4);
      if (STAGE == 0) store_x(sb, 2);
      if (STAGE == 1) store_d(sb, 3);
      if (STAGE == 2) load_x(2);
      if (STAGE == 3) load_d(3);
      W4W_MMA(5);
      XC(NS, 0);
      if (STAGE == 1) store_d(sb, 4);
      if (STAGE == 3) load_d(4);
      W4W_MMA(6);
      XC(NS, 1);
      if (STAGE == 0) store_x(sb, 3);
      if (STAGE == 2) load_x(3);
      W4W_MMA(7);
      XC(NS, 2);
      if (STAGE == 1) store_d(sb, 5);
      if (STAGE == 3) load_d(5);
      W4W_MMA(8);
      DR(0, 0);
      DR(0, 1);
      if (STAGE == 0) store_x(sb, 4);
      if (STAGE == 2) load_x(4);
      W4W_MMA(9);
      DR(0, 2);
      DR(0, 3);
      if (STAGE == 1) store_d(sb, 6);
      if (STAGE == 3) load_d(6);
      W4W_MMA(10);
      DC(NS, 0, 0);
      DC(NS, 0, 1);
      if (STAGE == 0) store_x(sb, 5);
      if (STAGE == 1) store_d(sb, 7);
      if (STAGE == 2) load_x(5);
      if (STAGE == 3) load_d(7);
      W4W_MMA(11);
      DC(NS, 0, 2);
      DR(1, 0);
      if (STAGE == 2) __syncthreads();
      W4W_MMA(12);
      DR(1, 1);
      DR(1, 2);
      RX(rx, sn, 0);
      RX(rx, sn, 1);
      W4W_MMA(13);
      DR(1, 3);
      DC(NS, 1, 0);
      RX(rx, sn, 2);
      RX(rx, sn, 3);
      W4W_MMA(14);
      DC(NS, 1, 1);
      DC(NS, 1, 2);
      RX(rx, sn, 4);
      RD(rd, sn, 0, 0);
      RD(rd, sn, 0, 1);
      W4W_MMA(15);
      RD(rd, sn, 0, 2);
      RD(rd, sn, 0, 3);
      RD(rd, sn, 1, 0);
      W4W_MMA(16);
      RD(rd, sn, 1, 1);
      RD(rd, sn, 1, 2);
      RD(rd, sn, 1, 3);
      W4W_MMA(17);
      W4W_SB;
#undef W4W_MMA
      }
    };
    auto read_raw = [&](const float *rx, const float *rd, int sn) {
#pragma unroll
      for (int l = 0; l < 5; ++l) RX(rx, sn, l);
#pragma unroll
      for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int l = 0; l < 4; ++l) RD(rd, sn, kh, l);
    };

    if (t0 < t1) {
      // prologue: chunk t0 -> buffer 0, chunk t0 + 1 requested, operands of k-step 0 (set 0), raw rows of k-step 1
      chunk_base();
#pragma unroll
      for (int i = 0; i < XSEG; ++i) load_x(i);
#pragma unroll
      for (int i = 0; i < DSEG; ++i) load_d(i);
      __syncthreads();                             // halo zero fill done
#pragma unroll
      for (int i = 0; i < XSEG; ++i) store_x(w4w_lds, i);
#pragma unroll
      for (int i = 0; i < DSEG; ++i) store_d(w4w_lds, i);
      chunk_advance();
      chunk_base();
#pragma unroll
      for (int i = 0; i < XSEG; ++i) load_x(i);
#pragma unroll
      for (int i = 0; i < DSEG; ++i) load_d(i);
      __syncthreads();
      read_raw(rdx0, rdd0, 0);
#pragma unroll
      for (int l = 0; l < 5; ++l) XR(l);
#pragma unroll
      for (int jl = 0; jl < 3; ++jl) XC(0, jl);
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) {
#pragma unroll
        for (int l = 0; l < 4; ++l) DR(kh, l);
#pragma unroll
        for (int jl = 0; jl < 3; ++jl) DC(0, kh, jl);
      }
      if (KH == 2) read_raw(rdx0, rdd0, 1);
      asm volatile("s_nop 4" ::: "memory");          // VALU -> asm MFMA operand distance for the first k-step
      // chunk j lives in buffer (j - t0) & 1.  The last chunk stages / transforms a clamped (valid) chunk nobody consumes.
      int bo = 0;
      for (int j = t0; j < t1; ++j) {
        const int nbo = W4W_BUF - bo;
        const float *rx = rdx0 + bo, *rd = rdd0 + bo;
        float *sb = w4w_lds + nbo;
        if constexpr (KH == 2) {                                // raw rows are read one k-step ahead of their transform
          step(w4_int<0>(), w4_int<0>(), rx, rd, 2, sb);        // + x rows of chunk j + 1: registers -> buffer nbo
          step(w4_int<1>(), w4_int<1>(), rx, rd, 3, sb);        // + dy rows of chunk j + 1: registers -> buffer nbo
          chunk_advance();
          chunk_base();
          step(w4_int<0>(), w4_int<2>(), rdx0 + nbo, rdd0 + nbo, 0, sb);   // + request x of chunk j + 2; barrier; first reads of nbo
          step(w4_int<1>(), w4_int<3>(), rdx0 + nbo, rdd0 + nbo, 1, sb);   // + request dy of chunk j + 2
        } else {                                                // raw rows are read in the k-step that transforms them
          step(w4_int<0>(), w4_int<0>(), rx, rd, 1, sb);
          step(w4_int<1>(), w4_int<1>(), rx, rd, 2, sb);
          chunk_advance();
          chunk_base();
          step(w4_int<0>(), w4_int<2>(), rx, rd, 3, sb);
          step(w4_int<1>(), w4_int<3>(), rdx0 + nbo, rdd0 + nbo, 0, sb);   // barrier first; + request dy of chunk j + 2
        }
        bo = nbo;
      }
    }

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");        // the last MFMAs' results (18 wait states before a VALU read)
#ifdef W4W_ABL_NOEPI
    {
      float a = 0.f;
#pragma unroll
      for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int q = 0; q < 9; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) a += acc[kh][q][r];
      if (a == 123.456f) p.part[tid] = a;
      return;
    }
#endif
    // partial sums: part[z][(3 BI + il) * 6 + 3 BJ + jl][k][c]; accumulator register r = row (r & 3) + 8 (r >> 2) + 4 half
    const long MC = (long)p.M * p.C;
    float *pz = p.part + (long)z * 36 * MC + (long)(kb * 64 + 4 * half) * p.C + cb * 32 + l31;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        float *pq = pz + (long)((3 * BI + q / 3) * 6 + 3 * BJ + q % 3) * MC + (long)((kh0 + kh) * 32) * p.C;
#pragma unroll
        for (int r = 0; r < 16; ++r) pq[(long)((r & 3) + 8 * (r >> 2)) * p.C] = acc[kh][q][r];
      }
  };
  switch (wp) {                                  // wave-uniform: four instances of the main loop
    case 0: body(w4_int<0>(), w4_int<0>()); break;
    case 1: body(w4_int<0>(), w4_int<1>()); break;
    case 2: body(w4_int<1>(), w4_int<0>()); break;
    default: body(w4_int<1>(), w4_int<1>()); break;
  }
}

// dW[k][c][3][3] = G^T (sum_z M_z) G,  G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
__global__ __launch_bounds__(256) void wino4_w3x3_reduce_kernel(const float *__restrict__ part, float *__restrict__ dW, int MC,
                                                                int splits) {
  const int i = blockIdx.x * 256 + threadIdx.x;        // k * C + c
  if (i >= MC) return;
  double m[36];
#pragma unroll
  for (int q = 0; q < 36; ++q) m[q] = 0.0;
  for (int z = 0; z < splits; ++z)
#pragma unroll
    for (int q = 0; q < 36; ++q) m[q] += (double)part[((long)z * 36 + q) * MC + i];
  // columns first: t[a][s] = sum_j m[a][j] G[j][s]
  double t[6][3];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const double m0 = m[a * 6], m1 = m[a * 6 + 1], m2 = m[a * 6 + 2], m3 = m[a * 6 + 3], m4 = m[a * 6 + 4], m5 = m[a * 6 + 5];
    t[a][0] = 0.25 * m0 - (m1 + m2) / 6.0 + (m3 + m4) / 24.0;
    t[a][1] = (m2 - m1) / 6.0 + (m3 - m4) / 12.0;
    t[a][2] = -(m1 + m2) / 6.0 + (m3 + m4) / 6.0 + m5;
  }
  float *o = dW + (long)i * 9;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const double t0 = t[0][s], t1 = t[1][s], t2 = t[2][s], t3 = t[3][s], t4 = t[4][s], t5 = t[5][s];
    o[0 * 3 + s] = (float)(0.25 * t0 - (t1 + t2) / 6.0 + (t3 + t4) / 24.0);
    o[1 * 3 + s] = (float)((t2 - t1) / 6.0 + (t3 - t4) / 12.0);
    o[2 * 3 + s] = (float)(-(t1 + t2) / 6.0 + (t3 + t4) / 6.0 + t5);
  }
}

}  // namespace lsps
#endif
