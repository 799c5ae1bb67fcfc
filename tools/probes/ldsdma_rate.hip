// Probe: per-CU service rate of LDS-DMA (buffer_load_dwordx4 ... lds, 1 KB per wave instruction) against plain
// buffer_load_dwordx4 into VGPRs, from an L2-resident source (each workgroup re-reads its own 64 KB window) and from a
// streaming source (every request a new line).  One 512-thread workgroup per CU, as the C8 kernels run.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/ldsdma_rate.hip -o /tmp/ldsdma_rate && /tmp/ldsdma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void *lds_ptr;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: LDS-DMA, 1: VGPR loads
__global__ __launch_bounds__(512, 1) void probe(const unsigned char *src, unsigned *out, int iters, int window_kb, long wg_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned char *base = src + (long)blockIdx.x * wg_stride;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(base), 0, 0x7fffffff, 0x00020000);
  const int pieces = window_kb;           // 1 KB pieces in the window
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int it = 0; it < iters; ++it) {
    // 8 pieces per wave per iteration
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int piece = (it * 64 + wave * 8 + i) % pieces;
      const unsigned off = (unsigned)(piece * 1024 + lane * 16);
      if (MODE == 0) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + (wave * 8 + i) * 1024), 16, off, 0, 0, 0);
      } else {
        const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        acc ^= v;
      }
    }
    if (MODE == 0 && (it & 3) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // <= 32 pieces in flight per wave
  }
  if (MODE == 0) {
    __syncthreads();
    acc = *reinterpret_cast<u32x4 *>(lds + tid * 16);
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

int main() {
  const int cus = 256;
  size_t bytes = (size_t)cus * (64 << 20) / 64;   // 1 MB per workgroup for the streaming case... 256 MB total
  bytes = (size_t)cus << 20;
  unsigned char *src;
  unsigned *out;
  hipMalloc(&src, bytes + (1 << 20));
  hipMalloc(&out, 4);
  hipMemset(src, 1, bytes + (1 << 20));
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int cfg = 0; cfg < 5; ++cfg) {
      // 32 KB private: L1 resident; 256 KB private (64 MB chip-wide: MALL / partly L2); 256 KB SHARED by all workgroups (L2 resident,
      // the packed weights of a conv layer); 1 MB private = 256 MB: streams from HBM / MALL; 1 MB shared
      const int window_kb = cfg == 0 ? 32 : (cfg == 1 || cfg == 2 ? 256 : 1024);
      const long stride = (cfg == 2 || cfg == 4) ? 0 : (long)(1 << 20);
      const int iters = 2000;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0)
          hipLaunchKernelGGL(probe<0>, dim3(cus), dim3(512), 65536, 0, src, out, iters, window_kb, stride);
        else
          hipLaunchKernelGGL(probe<1>, dim3(cus), dim3(512), 0, 0, src, out, iters, window_kb, stride);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double total = (double)cus * iters * 64 * 1024;
      printf("%s window %4d KB %s: %.3f ms  %.2f TB/s chip  %.1f GB/s per CU\n", mode == 0 ? "LDS-DMA x4 " : "VGPR loadx4", window_kb, stride ? "per WG" : "shared", ms,
             total / ms / 1e9, total / ms / 1e6 / cus);
    }
  return 0;
}
