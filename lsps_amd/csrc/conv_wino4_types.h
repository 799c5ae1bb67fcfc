// Parameter blocks and launchers of the Winograd F(4x4,3x3) kernels (conv_wino4.h, compiled in wino4.hip with its own flags).
#ifndef LSPS_CONV_WINO4_TYPES_H
#define LSPS_CONV_WINO4_TYPES_H
#include "common.h"

namespace lsps {

#define W4_RC 8                          // channels per staged row chunk = 4 k-steps between two barriers
#define W4_UREC 2304                     // floats of U per (32-channel k slice, channel pair): [2 c][4 blocks][32 k][8] + [2][4][32][1]
#define W4_LDS_BYTES 147456              // epilogue exchange (8 waves x 18 x 64 lanes x 16 B) > main loop (2 x 43520 B)

struct Wino4Pack {
  const float *W;
  float *U;
  int M, C;
  long sm, sc;
  int tapidx[9];
};

struct Wino4Params {
  const float *X, *U, *bias;
  const float *R;                // optional addend with Y's layout (dgrad skip connection / residual)
  float *Y;
  float *rstd;                   // [N*M], written when norm is 1 or 2, READ when norm == 3
  int Cx, M, N;
  int act;
  float slope;
  int norm;                      // 0: y = act(conv + bias) + R;  1: y = lrelu_slope(IN(conv)) (slope < 0: none);  2: y = IN(conv) + R
                                 // 3: y = backward of IN + LeakyReLU(slope) applied to conv, from that layer's saved output R and rstd
  float eps;
  // reduction split for launches that would not fill the chip (few images): workgroup (split, image, k slice) convolves
  // channels [split * Cx, (split + 1) * Cx) of the Ctot-channel tensors X / U and writes its PARTIAL plain output (norm 0, no
  // bias / activation / addend) to Y + split * N * M * 1024; w4_split_reduce_in_kernel sums the partials and normalises.
  int ksplit;                    // 0 | 1: no split (Ctot is ignored, Cx = all channels)
  int Ctot;
};

// weight gradient (conv_wino4w.h): partial sums part[split][36 positions][M][C], then dW = G^T (sum of splits) G
#define W4W_LDS_BYTES 130048             // two buffers of 32 x rows-of-6 + 64 dy rows-of-4 channels
struct Wino4WParams {
  const float *DY, *X;
  float *part;
  int N, M, C, H;
  int ntr, per_split;            // tile rows in total (N * H/4) and per split
};

// wino4.hip: 0 or LSPS_E_HIP (lsps_last_error set)
int wino4_launch_pack(const Wino4Pack &p, hipStream_t st);
int wino4_launch(const Wino4Params &p, hipStream_t st);
// sum of `ks` partial conv outputs part[ks][planes][1024] -> InstanceNorm (+ LeakyReLU(slope) | + residual) -> out, rstd[planes]
int wino4_launch_split_reduce_in(const float *part, int ks, int planes, const float *residual, float *out, float *rstd, float eps,
                                 float slope, hipStream_t st);
int wino4_launch_wgrad(const Wino4WParams &p, int splits, float *dW, int waves /*4 or 8 per workgroup*/, hipStream_t st);

}  // namespace lsps
#endif
