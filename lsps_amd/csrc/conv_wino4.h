// Winograd F(4x4, 3x3) form of the 3x3 / stride 1 / pad 1 conv on 32x32 feature maps (forward and dgrad of the residual
// convs), with the InstanceNorm that follows every such conv in the reference fused into the epilogue.
#ifndef LSPS_CONV_WINO4_H
#define LSPS_CONV_WINO4_H
#include "conv_types.h"
#include "conv_wino4_types.h"

namespace lsps {

// -------------------------------------------------------------------------------------------
// F(4x4,3x3): a 4x4 output tile from a 6x6 input tile with 36 multiplies per (k, c) pair instead of 144 (F(2x2,3x3): 64),
// i.e. 36 independent [K x C] x [C x tiles] GEMMs, one per position of the transformed 6x6 tile:
//   U = G g G^T        6x6 per (k, c), once per weight tensor (wino4_pack_kernel, double precision inside, cached)
//   V = B^T d B        per lane, in registers, from raw input rows staged in LDS
//   M_p += U_p V_p     v_mfma_f32_32x32x2_f32, one 32 k x 32 tiles accumulator per position
//   Y = A^T M A        epilogue
// Interpolation points 0, +-1, +-2, inf (Lavin & Gray):
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
//
// Workgroup = ONE image x 32 output channels: 512 threads, 8 waves = 4 position blocks (the 6x6 positions as 2x2 blocks
// of 3x3: wp = 2 bi + bj) x 2 halves of the image's 64 tiles (wt: tile rows 4wt .. 4wt+3).  A wave holds 9 accumulator
// tiles (144 registers, two waves per SIMD).  MFMA operand layout (32x32x2): lane l supplies A[k = l%32][c = l/32] and
// B[c = l/32][tile = l%32], so lane l transforms ONE tile of ONE channel per k-step: it reads the 5 raw rows x 6 columns
// its position block needs (ds_read_b128 + ds_read_b64 per row, row stride 40 floats: conflict-free b128), turns each row
// into the 3 columns it owns as the row arrives (pass A, one row per MFMA gap) and combines the rows (pass B): 48 VALU ops
// per 9 MFMAs.
// The transform of k-step s+1 is issued between the MFMAs of k-step s (order pinned with sched_barrier): a 64-cycle f32
// MFMA leaves ~10 free issue slots, plain (not packed) f32 VALU ops are the cheap fillers (MI355X_MICROARCH.md).
// The A operands (U) go L2 -> registers, 9 floats per lane per k-step (2 x 16 B + 4 B, coalesced), one k-step ahead.
// Because a workgroup owns whole (n, k) planes, the InstanceNorm statistics and the normalisation (+ LeakyReLU or
// + residual) happen in the epilogue on registers: the pre-norm tensor never reaches HBM and the separate
// read-modify-write pass of inorm_fwd_kernel disappears (reference: common_net.py:166-171, 177-181).
// -------------------------------------------------------------------------------------------
#ifndef W4_AHEAD
#define W4_AHEAD 2                       // MFMA gaps between the LDS read of a raw row and its pass A (see step())
#endif
#define W4_LDW 40                        // floats per staged row: [halo][32 pixels][halo][6 pad]; 4 rows = 32 banks (mod 64)
#define W4_ROWS 34                       // 32 image rows + 2 halo rows
#define W4_CH (W4_ROWS * W4_LDW)         // floats per staged channel
#define W4_BUF (W4_RC * W4_CH)           // floats per LDS row buffer

__global__ __launch_bounds__(256) void wino4_pack_kernel(Wino4Pack p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)p.M * p.C) return;
  const int m = (int)(idx % p.M), c = (int)(idx / p.M);
  double g[3][3], t[6][3], u[6][6];
  const float *w = p.W + (long)m * p.sm + (long)c * p.sc;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) g[r][s] = (double)w[p.tapidx[r * 3 + s]];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const double g0 = g[0][s], g1 = g[1][s], g2 = g[2][s];
    t[0][s] = 0.25 * g0;
    t[1][s] = -(g0 + g1 + g2) / 6.0;
    t[2][s] = -(g0 - g1 + g2) / 6.0;
    t[3][s] = g0 / 24.0 + g1 / 12.0 + g2 / 6.0;
    t[4][s] = g0 / 24.0 - g1 / 12.0 + g2 / 6.0;
    t[5][s] = g2;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const double t0 = t[i][0], t1 = t[i][1], t2 = t[i][2];
    u[i][0] = 0.25 * t0;
    u[i][1] = -(t0 + t1 + t2) / 6.0;
    u[i][2] = -(t0 - t1 + t2) / 6.0;
    u[i][3] = t0 / 24.0 + t1 / 12.0 + t2 / 6.0;
    u[i][4] = t0 / 24.0 - t1 / 12.0 + t2 / 6.0;
    u[i][5] = t2;
  }
  const int ks = m >> 5, l31 = m & 31, cp = c >> 1, half = c & 1;
  float *rec = p.U + ((long)ks * (p.C >> 1) + cp) * W4_UREC;
#pragma unroll
  for (int wp = 0; wp < 4; ++wp) {
    const int bi = wp >> 1, bj = wp & 1;
    float *d8 = rec + ((half * 4 + wp) * 32 + l31) * 8;
#pragma unroll
    for (int q = 0; q < 8; ++q) d8[q] = (float)u[3 * bi + q / 3][3 * bj + q % 3];
    rec[2048 + (half * 4 + wp) * 32 + l31] = (float)u[3 * bi + 2][3 * bj + 2];
  }
}

// three rows (B = 0: rows 0..2 over d0..d4; B = 1: rows 3..5 over d1..d5, e0 = d1) of B^T applied to one column
template <int B>
__device__ __forceinline__ void w4_xf(float e0, float e1, float e2, float e3, float e4, float &o0, float &o1, float &o2) {
  if (B == 0) {
    o0 = fmaf(4.f, e0, fmaf(-5.f, e2, e4));
    const float a = fmaf(-4.f, e2, e4), b = fmaf(-4.f, e1, e3);
    o1 = a + b;
    o2 = a - b;
  } else {
    const float c = e3 - e1, f = e2 - e0;
    o0 = fmaf(2.f, f, c);
    o1 = fmaf(-2.f, f, c);
    o2 = fmaf(4.f, e0, fmaf(-5.f, e2, e4));
  }
}

// z[x] = sum_j m[j] A^T[x][3B + j], x = 0..3
template <int B>
__device__ __forceinline__ void w4_inv(float m0, float m1, float m2, float &z0, float &z1, float &z2, float &z3) {
  if (B == 0) {
    const float s = m1 + m2, d = m1 - m2;
    z0 = m0 + s;
    z1 = d;
    z2 = s;
    z3 = d;
  } else {
    const float s = m0 + m1, d = m0 - m1;
    z0 = s;
    z1 = 2.f * d;
    z2 = 4.f * s;
    z3 = fmaf(8.f, d, m2);
  }
}

template <int V> struct w4_int { static constexpr int value = V; };

__global__ __launch_bounds__(512, 1) void wino4_f3x3_kernel(Wino4Params p) {
  extern __shared__ __attribute__((aligned(16))) float w4_lds[];
  typedef const volatile f32x4 __attribute__((address_space(3))) *lp4;
  typedef const volatile f32x2 __attribute__((address_space(3))) *lp2;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave & 3, wt = wave >> 2;
  const int l31 = lane & 31, half = lane >> 5, tr = l31 >> 3, tc = l31 & 7;
#ifdef W4_PRIO_YOUNG           // experiment (MI355X_MICROARCH.md, two waves per SIMD): static priority for the second-dispatched half
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
#ifdef W4_PRIO_OLD
  if (wave < 4) __builtin_amdgcn_s_setprio(1);
#endif

  // workgroup -> (image, k slice).  Workgroups go to the 8 XCDs round-robin in launch order; each XCD has its own 4 MB
  // L2 and a 32-channel slice of U is 1.18 MB (256 input channels): an XCD works on TWO slices (resident in its L2) and
  // half of the images, the two slices of an image adjacent in time so that its second read of the image hits L2.
  const int KS = p.M >> 5;
  int lin = blockIdx.x, split = 0;
  if (p.ksplit > 1) {                              // uniform: (split, image, k slice), see Wino4Params
    const int per = p.N * KS;
    split = lin / per;
    lin -= split * per;
  }
  const int ctot = p.ksplit > 1 ? p.Ctot : p.Cx;
  int n, ks;
  if (KS == 8 && (p.N & 1) == 0) {
    const int xcd = lin & 7, q = lin >> 3;
    ks = 2 * (xcd & 3) + (q & 1);
    n = (xcd >> 2) * (p.N >> 1) + (q >> 1);
  } else {
    ks = lin % KS;
    n = lin / KS;
  }
  constexpr int HW = 1024;
  const int nsteps = p.Cx >> 1, nchunks = p.Cx / W4_RC;
  // Global loads go through buffer descriptors (uniform base in SGPRs + ONE 32-bit lane offset + scalar offset): no
  // 64-bit per-lane addresses, which the flat form costs in VGPR pairs (this kernel has none to spare).
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.X + ((long)n * ctot + (long)split * p.Cx) * HW), 0, p.Cx * HW * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.U + ((long)ks * (ctot >> 1) + (long)split * nsteps) * W4_UREC), 0, nsteps * W4_UREC * 4, 0x00020000);
  const unsigned u8_off = (unsigned)(((half * 4 + wp) * 32 + l31) * 8) * 4u;
  const unsigned u1_off = (unsigned)(2048 + (half * 4 + wp) * 32 + l31) * 4u;

  // staging of the raw rows: one chunk = 8 channels x 32 rows x 8 16-B segments = 4 per thread (thread t: channels
  // t/256 + 2i), fetched ONE chunk (4 k-steps, ~2 us) ahead: the rows come from HBM / other XCDs, not from L2 like U
  constexpr int NSEG = W4_RC * 32 * 8 / 512;
  static_assert(NSEG == 4, "the k-step schedule spreads exactly four staging segments over its gaps");
  const int s_lds = (tid >> 8) * W4_CH + (((tid >> 3) & 31) + 1) * W4_LDW + 1 + (tid & 7) * 4;
  const unsigned s_off = (unsigned)((tid >> 8) * HW + ((tid >> 3) & 31) * 32 + (tid & 7) * 4) * 4u;
  f32x4 sreg[NSEG];
  auto load_seg = [&](int chunk, int i) {
#if !defined(W4_ABL_NOSTAGE) && !defined(W4_ABL_NOSTLD)
    sreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, s_off, (chunk * W4_RC + 2 * i) * HW * 4, 0));
#endif
  };
  auto store_seg = [&](float *buf, int i) {
#if !defined(W4_ABL_NOSTAGE) && !defined(W4_ABL_NOSTST)
    float *d = buf + s_lds + 2 * i * W4_CH;
    d[0] = sreg[i][0];
    *reinterpret_cast<f32x2 *>(d + 1) = f32x2{sreg[i][1], sreg[i][2]};
    d[3] = sreg[i][3];
#endif
  };
  auto load_rows = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < NSEG; ++i) load_seg(chunk, i);
  };
  auto store_rows = [&](float *buf) {
#pragma unroll
    for (int i = 0; i < NSEG; ++i) store_seg(buf, i);
  };

  f32x16 acc[9];
  f32x4 ac8[2], an8[2];          // U of the current / next k-step: positions 0..7 of the block
  float ac1, an1;                // position 8
  float vc[9], vn[9];            // V of the current / next k-step
  auto load_u = [&](int step) {
#ifdef W4_ABL_NOU
    return;
#endif
    const int so = step * (W4_UREC * 4);                       // uniform
    an8[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, u8_off, so, 0));
    an8[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, u8_off + 16, so, 0));
    an1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(urs, u1_off, so, 0));
  };

  // prologue: first loads go out before the LDS zero fill
  load_rows(0);
  load_u(0);
#ifndef W4_ABL_NOZERO          // (ablations: timing only, results wrong)
  for (int u = tid; u < 2 * W4_BUF / 4; u += 512) reinterpret_cast<f32x4 *>(w4_lds)[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
#pragma unroll
  for (int q = 0; q < 9; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  __syncthreads();                               // zero fill done (halo rows / columns are never written again)
  store_rows(w4_lds);
  load_rows(min(1, nchunks - 1));
  __syncthreads();

  auto body = [&](auto bi_c, auto bj_c) {
    constexpr int BI = decltype(bi_c)::value, BJ = decltype(bj_c)::value;
    const float *rd = w4_lds + half * W4_CH + (16 * wt + 4 * tr + BI) * W4_LDW + 4 * tc;
    f32x4 r4[5];
    f32x2 r2[5];
    float Q[5][3];
    // V = B^T d B row-first: pass A turns raw row l (six columns, ONE ds_read_b128 + ds_read_b64) into the block's three
    // columns Q[l][0..2] = sum_x B^T[3 BJ + j][x] d[l][x] as soon as that row has arrived; pass B combines the five rows per
    // column, V[3 BI + i][3 BJ + j] = sum_l B^T[3 BI + i][l] Q[l][j].  Same 48 ops per 9 MFMAs as column-first, but the ten
    // LDS reads of a k-step are no longer needed all at once: one row per MFMA gap, consumed two gaps later.  (Column-first
    // issued them as two bursts of five that all eight waves of the workgroup - in lockstep after every barrier - sent at the
    // same moment; without the reads that kernel ran 25 % faster.)
    auto read_row = [&](int bo, int g, int l) {
#ifndef W4_ABL_NOLDS
      r4[l] = *(lp4)(rd + bo + 2 * g * W4_CH + l * W4_LDW);
      r2[l] = *(lp2)(rd + bo + 2 * g * W4_CH + l * W4_LDW + 4);
#endif
    };
    auto col = [&](int l, int x) -> float { return x < 4 ? r4[l][x] : r2[l][x - 4]; };
    auto passA = [&](int l) {
#ifndef W4_ABL_NOXF
      w4_xf<BJ>(col(l, BJ), col(l, BJ + 1), col(l, BJ + 2), col(l, BJ + 3), col(l, BJ + 4), Q[l][0], Q[l][1], Q[l][2]);
#endif
    };
    auto passB = [&](int j) {
#ifndef W4_ABL_NOXF
      w4_xf<BI>(Q[0][j], Q[1][j], Q[2][j], Q[3][j], Q[4][j], vn[j], vn[3 + j], vn[6 + j]);
#endif
    };
    auto mma = [&](int q) {
      const float a = q < 4 ? ac8[0][q] : (q < 8 ? ac8[1][q - 4] : ac1);
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, vc[q], acc[q], 0, 0, 0);
    };
    auto rotate = [&]() {
#pragma unroll
      for (int q = 0; q < 9; ++q) vc[q] = vn[q];
      ac8[0] = an8[0];
      ac8[1] = an8[1];
      ac1 = an1;
    };
    auto load_u_part = [&](int step, int part) {
#ifndef W4_ABL_NOU
      const int so = step * (W4_UREC * 4);                       // uniform
      if (part == 0) an8[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, u8_off, so, 0));
      if (part == 1) an8[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, u8_off + 16, so, 0));
      if (part == 2) an1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(urs, u1_off, so, 0));
#endif
    };
    // One k-step: 9 MFMAs on (ac, vc); the gaps between them carry the transform of the NEXT k-step's raw rows (row buffer
    // bo, channel pair g) into vn, that step's U loads (one per gap, right at the start: they have the whole k-step to
    // arrive), and a share of the row staging (STAGE 1: this thread's four segments of chunk `sc` go from registers into row
    // buffer `sb`; STAGE 2: the loads of chunk `sc`), one per gap.  Nothing is issued in bursts: an in-order wave that
    // queues ten LDS reads or twelve LDS writes back to back stalls on the LDS / TA queues with the matrix pipe idle behind
    // it, and the eight waves of a workgroup reach such points together.
    auto step = [&](int bo, int g, int ustep, auto stage_c, int sb, int sc) {
      constexpr int STAGE = decltype(stage_c)::value;
      auto stage = [&](int i) {
        if (STAGE == 1) store_seg(w4_lds + sb, i);
        if (STAGE == 2) load_seg(sc, i);
      };
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        __builtin_amdgcn_sched_barrier(0);
        mma(q);
        __builtin_amdgcn_sched_barrier(0);
        if (q < 5) read_row(bo, g, q);
        if (q < 3) load_u_part(ustep, q);
        if (q >= W4_AHEAD && q < 5 + W4_AHEAD) passA(q - W4_AHEAD);
        if (q >= 3 && q < 7) stage(q - 3);
        if (W4_AHEAD <= 2 && q == 7) {
          passB(0);
          passB(1);
        }
        if (W4_AHEAD == 3 && q == 7) passB(0);
        if (W4_AHEAD == 3 && q == 8) passB(1);
        if (q == 8) passB(2);
      }
      __builtin_amdgcn_sched_barrier(0);
      rotate();
    };
    // V(0): transform of k-step 0 (no MFMAs yet)
#pragma unroll
    for (int l = 0; l < 5; ++l) read_row(0, 0, l);
#pragma unroll
    for (int l = 0; l < 5; ++l) passA(l);
    passB(0);
    passB(1);
    passB(2);
    rotate();                                    // (ac <- U(0), loaded in the prologue)
    // chunk c (k-steps 4c .. 4c+3) lives in buffer c & 1.  K-step s transforms k-step s+1: the first three steps of a
    // chunk stay inside it (channel pairs 1..3); then chunk c+1 is stored and published and chunk c+2 requested; the
    // last step transforms pair 0 of chunk c+1.
    auto chunk = [&](int bo, int nbo, int c) {
      step(bo, 1, min(4 * c + 1, nsteps - 1), w4_int<0>(), 0, 0);
      step(bo, 2, min(4 * c + 2, nsteps - 1), w4_int<0>(), 0, 0);
      step(bo, 3, min(4 * c + 3, nsteps - 1), w4_int<1>(), nbo, 0);          // + chunk c+1: registers -> buffer nbo
      __syncthreads();
      step(nbo, 0, min(4 * c + 4, nsteps - 1), w4_int<2>(), 0, min(c + 2, nchunks - 1));   // + request chunk c+2
    };
    for (int c = 0; c < nchunks; c += 2) {
      chunk(0, W4_BUF, c);
      if (c + 1 < nchunks) chunk(W4_BUF, 0, c + 1);
    }
  };
  // Everything below the main loop is also instantiated per position block (compile-time bi, bj: no selects).  The
  // barriers inside are executed the same number of times by every wave, from different program counters.
  auto epilogue = [&](auto bi_c, auto bj_c) {
  constexpr int bi = decltype(bi_c)::value, bj = decltype(bj_c)::value;
#ifdef W4_ABL_NOEPI
  {
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) a += acc[q][r];
    if (a == 123.456f) p.Y[tid] = a;
    return;
  }
#endif
  __syncthreads();                               // the row buffers are free: reuse LDS for the exchanges below

  // ---- epilogue: Y = A^T M A.  Register r of an accumulator = output channel k0 + (r & 3) + 8 (r >> 2) + 4 half, tile l31.
  // Phase 1 (columns): the two column blocks (wp ^ 1) swap M so that block bj finishes registers 8 bj .. 8 bj + 7:
  //   Z[il][x] = sum_j M[il][j] A^T[x][j] over all six j.   Phase 2 (rows): the two row blocks (wp ^ 2) swap Z so that
  //   block bi finishes registers 8 bj + 4 bi .. + 3:  y[yy][x] = sum_i A^T[yy][i] Z[i][x].
  f32x4 *xch = reinterpret_cast<f32x4 *>(w4_lds);
  float Z[8][3][4];
  {
    f32x4 *dst = xch + (long)(wave ^ 1) * 18 * 64 + lane;      // the partner reads its own region
    // send M of the partner's registers: r = 8 (1 - bj) + rr, slots rr * 9 + q
    float snd[72];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr)
#pragma unroll
      for (int q = 0; q < 9; ++q) snd[rr * 9 + q] = bj ? acc[q][rr] : acc[q][8 + rr];
#pragma unroll
    for (int s4 = 0; s4 < 18; ++s4) dst[s4 * 64] = f32x4{snd[4 * s4], snd[4 * s4 + 1], snd[4 * s4 + 2], snd[4 * s4 + 3]};
  }
  __syncthreads();
  {
    const f32x4 *src = xch + (long)wave * 18 * 64 + lane;
    float rcv[72];
#pragma unroll
    for (int s4 = 0; s4 < 18; ++s4) {
      const f32x4 v = src[s4 * 64];
      rcv[4 * s4] = v[0];
      rcv[4 * s4 + 1] = v[1];
      rcv[4 * s4 + 2] = v[2];
      rcv[4 * s4 + 3] = v[3];
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr)
#pragma unroll
      for (int il = 0; il < 3; ++il) {
        const float o0 = bj ? acc[il * 3][8 + rr] : acc[il * 3][rr];
        const float o1 = bj ? acc[il * 3 + 1][8 + rr] : acc[il * 3 + 1][rr];
        const float o2 = bj ? acc[il * 3 + 2][8 + rr] : acc[il * 3 + 2][rr];
        const float q0 = rcv[rr * 9 + il * 3], q1 = rcv[rr * 9 + il * 3 + 1], q2 = rcv[rr * 9 + il * 3 + 2];
        float a0, a1, a2, a3, b0, b1, b2, b3;
        w4_inv<bj>(o0, o1, o2, a0, a1, a2, a3);      // own block of columns, then the partner's
        w4_inv<1 - bj>(q0, q1, q2, b0, b1, b2, b3);
        Z[rr][il][0] = a0 + b0;
        Z[rr][il][1] = a1 + b1;
        Z[rr][il][2] = a2 + b2;
        Z[rr][il][3] = a3 + b3;
      }
  }
  __syncthreads();                               // phase-1 regions are read: phase 2 may overwrite them
  {
    f32x4 *dst = xch + (long)(wave ^ 2) * 12 * 64 + lane;
    // send Z of the partner's registers: rr = 4 (1 - bi) + i, slots (i * 3 + il)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int il = 0; il < 3; ++il) {
        f32x4 v;
#pragma unroll
        for (int x = 0; x < 4; ++x) v[x] = bi ? Z[i][il][x] : Z[4 + i][il][x];
        dst[(i * 3 + il) * 64] = v;
      }
  }
  __syncthreads();
  float y[4][4][4];                              // [i][yy][x]: channel k0 + i + 8 (2 bj + bi) + 4 half
  {
    const f32x4 *src = xch + (long)wave * 12 * 64 + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 pz[3];
#pragma unroll
      for (int il = 0; il < 3; ++il) pz[il] = src[(i * 3 + il) * 64];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const float o0 = bi ? Z[4 + i][0][x] : Z[i][0][x];
        const float o1 = bi ? Z[4 + i][1][x] : Z[i][1][x];
        const float o2 = bi ? Z[4 + i][2][x] : Z[i][2][x];
        float a0, a1, a2, a3, b0, b1, b2, b3;
        w4_inv<bi>(o0, o1, o2, a0, a1, a2, a3);
        w4_inv<1 - bi>(pz[0][x], pz[1][x], pz[2][x], b0, b1, b2, b3);
        y[i][0][x] = a0 + b0;
        y[i][1][x] = a1 + b1;
        y[i][2][x] = a2 + b2;
        y[i][3][x] = a3 + b3;
      }
    }
  }
  const int kbase = ks * 32 + 8 * (2 * bj + bi) + 4 * half;    // + i
  const int orow = 16 * wt + 4 * tr;
  float *ybase = p.Y + (((long)split * p.N + n) * p.M + kbase) * HW + orow * 32 + 4 * tc;

  if (p.norm == 3) {
    // BACKWARD of the InstanceNorm + LeakyReLU in front of this (dgrad) conv, recovered from that layer's saved OUTPUT o (p.R)
    // and rstd, on the gradient d this conv just produced (norm_act.hip: inorm_bwd_kernel, activation variant):
    //   g = d * lrelu'(o),  xh = o > 0 ? o : o / slope,  out = rstd * (g - mean(g) - xh * mean(g * xh))
    // Both plane sums in ONE pass over the registers; the un-normalised gradient never reaches HBM.
    __syncthreads();                             // phase-2 regions are read
    float *red = w4_lds;                         // [sum 2][wt 2][wp 4][half 2][i 4]
    const float inv_slope = 1.f / p.slope;
    float xh[4][4][4], s[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float *om = p.R + (ybase - p.Y) + (long)i * HW;
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int yy = 0; yy < 4; ++yy) {
        const f32x4 o = *reinterpret_cast<const f32x4 *>(om + yy * 32);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const bool pos = o[x] > 0.f;
          const float h = pos ? o[x] : o[x] * inv_slope;
          const float g = pos ? y[i][yy][x] : y[i][yy][x] * p.slope;
          xh[i][yy][x] = h;
          y[i][yy][x] = g;
          a1 += g;
          a2 = fmaf(g, h, a2);
        }
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        a1 += __shfl_xor(a1, o, 64);
        a2 += __shfl_xor(a2, o, 64);
      }
      s[0][i] = a1;
      s[1][i] = a2;
    }
    if (l31 == 0) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[((q * 2 + wt) * 4 + wp) * 8 + half * 4 + i] = s[q][i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float m1 = (red[((0 * 2 + 0) * 4 + wp) * 8 + half * 4 + i] + red[((0 * 2 + 1) * 4 + wp) * 8 + half * 4 + i]) * (1.f / 1024.f);
      const float m2 = (red[((1 * 2 + 0) * 4 + wp) * 8 + half * 4 + i] + red[((1 * 2 + 1) * 4 + wp) * 8 + half * 4 + i]) * (1.f / 1024.f);
      const float rs = p.rstd[(long)n * p.M + kbase + i];
      float *ym = ybase + (long)i * HW;
#pragma unroll
      for (int yy = 0; yy < 4; ++yy) {
        f32x4 ov;
#pragma unroll
        for (int x = 0; x < 4; ++x) ov[x] = rs * (y[i][yy][x] - m1 - xh[i][yy][x] * m2);
        *reinterpret_cast<f32x4 *>(ym + yy * 32) = ov;
      }
    }
    return;
  }
  if (p.norm) {
    // InstanceNorm over the (n, k) plane = 64 tiles = the 32 lanes of this half in waves (wp, wt = 0) and (wp, wt = 1).
    // Two passes over the registers (mean, then centred sum of squares), each: 16-value lane sum, xor-shuffles inside
    // the 32-lane half, one LDS hand-off between the two waves.
    // the residual's 16 float4 per lane are requested BEFORE the statistics: their latency runs under the two reduction passes
    // (one workgroup per CU: nothing else covers an epilogue load)
    f32x4 res[4][4];
    if (p.norm == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int yy = 0; yy < 4; ++yy) res[i][yy] = *reinterpret_cast<const f32x4 *>(p.R + (ybase - p.Y) + (long)i * HW + yy * 32);
    }
    __syncthreads();                             // phase-2 regions are read
    float *red = w4_lds;                         // [pass 2][wt 2][wp 4][half 2][i 4]
    float mean[4], rs[4];
#ifdef W4_ABL_NOSTATS
#pragma unroll
    for (int i = 0; i < 4; ++i) { mean[i] = 0.f; rs[i] = 1.f; }
#else
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      float s[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float a = 0.f;
#pragma unroll
        for (int yy = 0; yy < 4; ++yy)
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const float v = pass ? (y[i][yy][x] - mean[i]) : y[i][yy][x];
            a += pass ? v * v : v;
          }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        s[i] = a;
      }
      if (l31 == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red[((pass * 2 + wt) * 4 + wp) * 8 + half * 4 + i] = s[i];
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float tot = red[((pass * 2 + 0) * 4 + wp) * 8 + half * 4 + i] + red[((pass * 2 + 1) * 4 + wp) * 8 + half * 4 + i];
        if (pass == 0)
          mean[i] = tot * (1.f / 1024.f);
        else
          rs[i] = rsqrtf(tot * (1.f / 1024.f) + p.eps);
      }
    }
#endif
    if (l31 == 0 && wt == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) p.rstd[(long)n * p.M + kbase + i] = rs[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float *ym = ybase + (long)i * HW;
#pragma unroll
      for (int yy = 0; yy < 4; ++yy) {
        f32x4 o;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          float v = (y[i][yy][x] - mean[i]) * rs[i];
          if (p.norm == 1 && p.slope >= 0.f) v = v > 0.f ? v : v * p.slope;
          o[x] = v;
        }
        if (p.norm == 2) o += res[i][yy];
#ifdef W4_ABL_NOSTORE
        if (o[0] == 123.456f)
#endif
        *reinterpret_cast<f32x4 *>(ym + yy * 32) = o;
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float bv = p.bias ? p.bias[kbase + i] : 0.f;
    float *ym = ybase + (long)i * HW;
#pragma unroll
    for (int yy = 0; yy < 4; ++yy) {
      f32x4 o;
#pragma unroll
      for (int x = 0; x < 4; ++x) o[x] = apply_act(y[i][yy][x] + bv, p.act, p.slope);
      if (p.R) o += *reinterpret_cast<const f32x4 *>(p.R + (ym - p.Y) + yy * 32);
      *reinterpret_cast<f32x4 *>(ym + yy * 32) = o;
    }
  }
  };
#ifdef W4_PRIO
  if (wt) __builtin_amdgcn_s_setprio(1);         // experiment: static priority for the younger half of the workgroup
#endif
  switch (wp) {                                  // wave-uniform: four instances of main loop + epilogue
    case 0: body(w4_int<0>(), w4_int<0>()); epilogue(w4_int<0>(), w4_int<0>()); break;
    case 1: body(w4_int<0>(), w4_int<1>()); epilogue(w4_int<0>(), w4_int<1>()); break;
    case 2: body(w4_int<1>(), w4_int<0>()); epilogue(w4_int<1>(), w4_int<0>()); break;
    default: body(w4_int<1>(), w4_int<1>()); epilogue(w4_int<1>(), w4_int<1>()); break;
  }
}

// Reduction-split launches (Wino4Params::ksplit): out = IN(sum of the ks partial planes) (+ LeakyReLU | + residual), one
// workgroup per (n, k) plane of 1024 pixels held in registers; same two-pass statistics as inorm_fwd_kernel (norm_act.hip).
__global__ __launch_bounds__(256) void w4_split_reduce_in_kernel(const float *__restrict__ part, int ks, long stride,
                                                                 const float *__restrict__ res, float *__restrict__ out,
                                                                 float *__restrict__ rstd_out, float eps, float slope) {
  __shared__ float red[4];
  const long o = (long)blockIdx.x * 1024 + threadIdx.x * 4;
  f32x4 v = *reinterpret_cast<const f32x4 *>(part + o);
  for (int k = 1; k < ks; ++k) v += *reinterpret_cast<const f32x4 *>(part + k * stride + o);
  auto block_sum = [&](float x) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) x += __shfl_xor(x, s, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  const float mean = block_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.f / 1024.f);
  const f32x4 d = v - mean;
  const float var = block_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.f / 1024.f);
  const float rstd = 1.f / sqrtf(var + eps);
  if (threadIdx.x == 0) rstd_out[blockIdx.x] = rstd;
  f32x4 r = d * rstd;
  if (slope >= 0.f) {
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = r[e] > 0.f ? r[e] : r[e] * slope;
  }
  if (res) r += *reinterpret_cast<const f32x4 *>(res + o);
  *reinterpret_cast<f32x4 *>(out + o) = r;
}

}  // namespace lsps
#endif
