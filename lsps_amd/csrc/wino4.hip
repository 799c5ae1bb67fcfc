// Winograd F(4x4,3x3) conv kernels (conv_wino4.h) in their own translation unit: built with -fno-slp-vectorize, because
// the SLP vectoriser fuses the per-column transform ops into v_pk_fma_f32 pairs ACROSS the sched_barrier-pinned MFMA gaps
// (collapsing "one column per gap" into one block after five back-to-back MFMAs, plus ~20 v_mov per k-step to form the
// pairs); packed f32 VALU is also the more expensive filler beside MFMAs on this part (MI355X_MICROARCH.md).
#include "conv_wino4.h"
#include "conv_wino4w.h"

namespace lsps {

int wino4_launch_pack(const Wino4Pack &p, hipStream_t st) {
  hipLaunchKernelGGL(wino4_pack_kernel, dim3(ceil_div((long)p.M * p.C, 256)), dim3(256), 0, st, p);
  LSPS_CHECK_LAUNCH("wino4_pack");
  return 0;
}

int wino4_launch(const Wino4Params &p, hipStream_t st) {
  if (int rc = lds_optin(reinterpret_cast<const void *>(wino4_f3x3_kernel), (int)W4_LDS_BYTES, "wino4_f3x3")) return rc;
  hipLaunchKernelGGL(wino4_f3x3_kernel, dim3(p.N * (p.M / 32) * (p.ksplit > 1 ? p.ksplit : 1)), dim3(512), W4_LDS_BYTES, st, p);
  LSPS_CHECK_LAUNCH("wino4_f3x3");
  return 0;
}

int wino4_launch_split_reduce_in(const float *part, int ks, int planes, const float *residual, float *out, float *rstd, float eps,
                                 float slope, hipStream_t st) {
  hipLaunchKernelGGL(w4_split_reduce_in_kernel, dim3(planes), dim3(256), 0, st, part, ks, (long)planes * 1024, residual, out, rstd,
                     eps, slope);
  LSPS_CHECK_LAUNCH("w4_split_reduce_in");
  return 0;
}

int wino4_launch_wgrad(const Wino4WParams &p, int splits, float *dW, int waves, hipStream_t st) {
  for (const void *f : {reinterpret_cast<const void *>(wino4_w3x3_kernel<2>), reinterpret_cast<const void *>(wino4_w3x3_kernel<1>)})
    if (int rc = lds_optin(f, (int)W4W_LDS_BYTES, "wino4_w3x3")) return rc;
  const dim3 grid(p.C / 32, p.M / 64, splits);
  if (waves == 8)
    hipLaunchKernelGGL(wino4_w3x3_kernel<1>, grid, dim3(512), W4W_LDS_BYTES, st, p);
  else
    hipLaunchKernelGGL(wino4_w3x3_kernel<2>, grid, dim3(256), W4W_LDS_BYTES, st, p);
  LSPS_CHECK_LAUNCH("wino4_w3x3");
  hipLaunchKernelGGL(wino4_w3x3_reduce_kernel, dim3(ceil_div((long)p.M * p.C, 256)), dim3(256), 0, st,
                     (const float *)p.part, dW, p.M * p.C, splits);
  LSPS_CHECK_LAUNCH("wino4_w3x3_reduce");
  return 0;
}

}  // namespace lsps
