// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS[i] = i (16-bit); lane l reads at byte address 8*l * STRIDE.
// Prints which source elements each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short *out, int stride_elems) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(lds + lane * stride_elems));
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}
int main() {
  unsigned short *d, h[256];
  hipMalloc(&d, 512);
  for (int stride : {4, 16}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("stride %d elems per lane\n", stride);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
