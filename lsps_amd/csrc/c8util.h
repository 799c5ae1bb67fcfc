// Device helpers shared by the C8 (bf16) and X3 (three-limb) kernel families: LDS-DMA issue, transposing LDS reads, small
// epilogue utilities.  No kernels in here (this header is included by more than one translation unit).
#ifndef LSPS_C8UTIL_H
#define LSPS_C8UTIL_H
#include "conv_types.h"

namespace lsps {

typedef __attribute__((address_space(3))) void *c8_lds_ptr;

// LDS-DMA issued from inline asm, for the kernels that read their operands with ds_read_b64_tr_b16.  hipcc cannot tell that a
// transposing read does not alias the DMA target and puts `s_waitcnt vmcnt(0)` in front of the first such read after a
// buffer_load ... lds — the next chunk's requests would be waited for before the current chunk's MFMAs start, i.e. no overlap at
// all (profiles/r3g_c8_ablations.txt: 22 % / 40 % of the two weight-gradient kernels).  Issued this way the compiler sees no LDS
// write; the kernels' own `s_waitcnt vmcnt(0)` + barrier at the end of a chunk is the only synchronisation, as designed.
// Untracked VMEM requests only make the compiler's own vmcnt waits more conservative (the counter retires in order).
typedef int c8_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ c8_i32x4 c8_rsrc_words(const void *base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  c8_i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));      // stride 0
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;                                                                // raw buffer, out-of-range reads return 0
  return r;
}
__device__ __forceinline__ unsigned c8_lds_addr(const void *ptr) {
  return (unsigned)(__UINTPTR_TYPE__)(c8_lds_ptr)ptr;
}
// 64 lanes x 16 bytes -> LDS [lds_addr + 16 lane ..); lds_addr and soff wave-uniform
__device__ __forceinline__ void c8_dma16_asm(c8_i32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(voff), "s"(rsrc),
                 "s"(__builtin_amdgcn_readfirstlane((int)soff))
               : "memory", "m0");
}

// x > 0 ? b : a  (x <= 0, either zero included: LeakyReLU'(0) = slope like torch's `out > 0` test) without a compare mask:
// 64 such masks held live in SGPR pairs spill.  ((bits - 1) | bits) is negative exactly for +0, -0 and negative x.
__device__ __forceinline__ float c8_sel_nonpos(float x, float a, float b) {
  const int xb = __builtin_bit_cast(int, x);
  const int m = ((xb - 1) | xb) >> 31;
  return __builtin_bit_cast(float, (__builtin_bit_cast(int, a) & m) | (__builtin_bit_cast(int, b) & ~m));
}

// butterfly reduce-scatter inside the 32-lane half: on return lane l31 holds in v[0] the sum over the half's lanes of v[l31].
// Step CNT (16, 8, .. 1): partners l31 ^ CNT keep one half of their CNT*2 values each and exchange the other half.
// (Template recursion: with a run-time trip count the register array is indexed dynamically = select chains.)
template <int CNT>
__device__ __forceinline__ void c8_reduce_scatter32(float (&v)[32], int l31) {
  const bool up = (l31 & CNT) != 0;
#pragma unroll
  for (int k = 0; k < CNT; ++k) {
    const float send = up ? v[k] : v[k + CNT];
    const float keep = up ? v[k + CNT] : v[k];
    v[k] = keep + __shfl_xor(send, CNT, 64);
  }
  if constexpr (CNT > 1) c8_reduce_scatter32<CNT / 2>(v, l31);
}

typedef short c8_s16x4 __attribute__((ext_vector_type(4)));
typedef short c8_s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 c8_tr_frag(const unsigned char *p0, const unsigned char *p1) {
  const c8_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c8_s16x4 __attribute__((address_space(3))) *)(p0));
  const c8_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c8_s16x4 __attribute__((address_space(3))) *)(p1));
  const c8_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

}  // namespace lsps
#endif
