// Shared helpers for the liblsps_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/lsps_hip.h"

namespace lsps {

void set_error(const char *fmt, ...);

// the switches in force (lsps_set_options; defaults resolved, never -1).  Defined in igemm.hip.
const LspsOptions &opts();

#define LSPS_CHECK_ARG(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      lsps::set_error(__VA_ARGS__);          \
      return LSPS_E_ARG;                     \
    }                                        \
  } while (0)

#define LSPS_CHECK_LAUNCH(name)                                            \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      lsps::set_error("%s: %s", name, hipGetErrorString(e_));              \
      return LSPS_E_HIP;                                                   \
    }                                                                      \
  } while (0)

// Opt-in for > 64 KB of dynamic LDS.  The attribute is PER DEVICE (a process that touches a second GPU must set it there
// too) and launches may come from two host threads (the main stream's and the autograd worker's): remembered per
// (kernel, device) under a mutex.
int lds_optin(const void *kernel, int bytes, const char *name);

// Pack-cache scope of the trainer step (igemm.hip: lsps_pack_cache_begin / _end) for packed weight layouts of other
// translation units: the cached slot for (W, layout tag, geometry) on this stream (*hit = true), a fresh slot of `need`
// bytes the caller must fill (*hit = false), or nullptr (no scope open / arena full: pack into the call's workspace).
void *pack_cache_slot(const float *W, int tag, int M, int C, long sm, long sc, size_t need, bool *hit, hipStream_t st);

// bf16-MFMA forms of the one-input-channel stems (c8stem.h, compiled in c8.hip; called by the lsps_c8_stem_* entries of igemm.hip)
bool c8_stem_bf16_ok(int N, int H, int W, int K, int R, int S, int stride, int pad);
int c8_stem_fwd_bf16(const float *x, const float *w, const float *bias, void *y, int N, int H, int W, int K, int R, int S, int stride,
                     int pad, float slope, hipStream_t st);
size_t c8_stem_wgrad_bf16_ws_bytes();
int c8_stem_wgrad_bf16(const float *x, const void *dy, const void *y, float *dw, float *db, int N, int H, int W, int K, int R, int S,
                       int stride, int pad, float slope, void *ws, size_t ws_bytes, hipStream_t st);

int c8_stem_dgrad_bf16(const void *dy, const void *y, const float *w, float *dx, int N, int H, int W, int K, int R, int S, int stride,
                       int pad, float slope, hipStream_t st);

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// 64-lane wavefront sum (all lanes receive the total)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves). `red` is >= 4 floats of LDS.
// Every thread receives the total.  Safe to call repeatedly with the same `red`.
__device__ __forceinline__ float block_sum_256(float v, float *red) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

}  // namespace lsps
