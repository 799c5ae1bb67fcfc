/*
 * lsps_hip.h — C-ABI of liblsps_hip.so: the MI355X (gfx950) kernels behind the LSPS depth path.
 *
 * The reference (masabdi/LSPS) has no FFI seam: its depth encoder/decoder path is Python that
 * calls torch.nn built-ins.  Each entry point below replaces ONE torch built-in call site of
 * the reference (cited per function as src/trainers/<file>:<line>); lsps_amd/trainers binds
 * them through ctypes from torch.autograd.Function.forward/backward.
 *
 * Conventions
 *   - plain pointers and sizes only; every tensor is contiguous and resident in device (HBM) memory;
 *     `stream` is a hipStream_t passed as void*;
 *   - two tensor conventions, fixed per entry point by its name:
 *       lsps_<op>...     : float32, NCHW (row-major) — the reference's own layout and dtype;
 *       lsps_c8_<op>...  : the bf16 mode of BASELINE config 5: activations are bfloat16 (passed as
 *                          `const void*` / `void*`, 2 bytes per element) in the channel-group layout
 *                          "C8" [N][C/8][H][W][8] (the 8 consecutive channels of a pixel = one 16-byte
 *                          unit, C a multiple of 8); weights, biases, statistics, gradients of weights,
 *                          loss scalars and everything an optimizer touches stay float32 in torch layout;
 *                          f32 NCHW <-> C8 conversion entries are lsps_c8_from_nchw / lsps_c8_to_nchw;
 *   - every call is asynchronous on `stream`, never allocates and never synchronises: scratch
 *     memory is handed in by the caller (`ws`, sized by the matching *_workspace_bytes);
 *   - return value 0 = launched, negative = rejected (LSPS_E_*); lsps_last_error() explains.
 *   - Conv2d weight is (K, C, R, S); ConvTranspose2d weight is (C_in, C_out, R, S) as in torch.
 */
#ifndef LSPS_HIP_H
#define LSPS_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSPS_ABI_VERSION 1

#define LSPS_E_ARG   (-1)   /* bad argument (null pointer, unsupported size)            */
#define LSPS_E_WS    (-2)   /* workspace too small                                      */
#define LSPS_E_HIP   (-3)   /* a HIP runtime call failed                                */

/* epilogue activations fused into the conv kernels */
#define LSPS_ACT_NONE  0
#define LSPS_ACT_LRELU 1    /* nn.LeakyReLU(slope), common_net.py:252,264               */
#define LSPS_ACT_TANH  2    /* nn.Tanh, lsps_nets.py:228-229                            */

int         lsps_version(void);
const char *lsps_last_error(void);
/* number of compute units of the current device (used by callers to size split counts) */
int         lsps_device_cus(void);
/* Name of the main compute kernel the most recent conv / convT call OF THIS THREAD dispatched to (e.g. "wino4_f3x3_kernel",
 * "igemm_f_kernel<2,2,2,2>"), and in *launches (may be NULL) the number of main-kernel launches since the previous call of
 * this function.  No reference counterpart (torch dispatches inside cuDNN): it exists so that profilers and tests read the
 * dispatch instead of mirroring it (bench.py `roofline.per_kernel`, tests/test_parity_gpu.py). */
const char *lsps_last_kernel(int *launches);

/* Scope in which packed weight panels are cached in caller-owned device memory: between two optimizer steps the weights
 * of the reference's modules do not change, yet one update method runs the same nn.Conv2d several times (sub-batches,
 * forward + backward layouts; lsps_trainer.py:76-262).  Inside begin/end a panel is packed once per (weight pointer,
 * geometry, tap list); the caller promises not to modify the weights in between, and must call _end (or _begin again)
 * before doing so.  Process-wide, like the math mode; stream-ordered (the arena is written and read on the streams of
 * the conv calls).  Without a scope every call packs into its own workspace (stateless, as before). */
int lsps_pack_cache_begin(void *arena, size_t bytes);
int lsps_pack_cache_end(void);
/* Weights in the device address range [lo, hi) stay unchanged ACROSS scopes until the caller says otherwise: their panels
 * are kept in a second caller-owned arena and found again by every later scope.  The estimate modes of the reference step
 * dis_opt only (lsps_trainer.py:220-262: the generator is a frozen feature extractor there), so post_update re-packed the
 * generator's ~45 weight tensors in every step.  The table is dropped when the call names another range, arena or `epoch`
 * (the caller's change counter of those weights); lo = NULL: the scopes that follow have no frozen weights (entries are
 * kept for a later call with the same range / arena / epoch).  Process-wide and host-side like the scope itself. */
int lsps_pack_cache_frozen(const void *lo, const void *hi, void *arena, size_t bytes, unsigned long long epoch);
/* Math mode of the MFMA conv kernels (process-wide; direct HBM-bound kernels are f32 always): 0 = exact f32 MFMA (default);
 * 1 = operands rounded to bf16 in registers, v_mfma_f32_32x32x16_bf16 with f32 accumulation (BASELINE config 5:
 * "bf16 with MFMA conv path").  Tensors stay f32 in HBM in both modes.                                  */
#define LSPS_MATH_F32  0
#define LSPS_MATH_BF16 1
int         lsps_set_math_mode(int mode);
int         lsps_get_math_mode(void);
/* Algorithm of the f32 3x3 / stride-1 / width-32 convs (the residual blocks: common_net.py:162-163, forward and dgrad):
 * 0 = direct implicit GEMM; 1 = Winograd F(2x2,3x3) when the grid fills the chip (default; initial value from the
 * environment variable LSPS_WINO); 2 = Winograd for every eligible shape (forward / dgrad: H % 8 == 0, C % 16 == 0, K % 64 == 0; weight gradient: H >= 4 even,
 * C % 64 == 0, K % 64 == 0).
 * Winograd results differ from the direct kernel by f32 round-off (~5e-7 relative, same size as the direct kernel's
 * own distance from an f64 convolution); bf16 / split math modes are not affected.  Non-finite inputs: the transforms
 * add and subtract neighbouring pixels before multiplying, so an Inf in a tile can surface as NaN in that tile's
 * outputs where the direct kernel would give Inf (finite data are unaffected).                            */
int         lsps_set_winograd(int mode);
int         lsps_get_winograd(void);

/* Dispatch / experiment switches of the library, set by the host in ONE call (round 6: no `getenv` is left in the library;
 * lsps_amd/options.py owns the LSPS_* environment, pushes this block when the library is loaded and whenever its options
 * object changes, hashes it into the hipGraph signature and prints it in the bench line).  A field value of -1 means
 * "keep the library's default" (the value in brackets).  `struct_size` = sizeof(LspsOptions) of the caller (ABI check).
 * Not a per-call argument: the same process-wide scope as lsps_set_math_mode / lsps_set_winograd.              */
typedef struct LspsOptions {
  int struct_size;
  int wino4_split;     /* [1] reduction-split F(4x4,3x3) launches for few-image passes without a backward (igemm.hip)   */
  int fs2_cc;          /* [4] channel chunk (4 | 8) of the exact-f32 3x3 / stride-2 forward kernel                        */
  int wino4w;          /* [1] F(4x4,3x3) weight gradient (0: F(2x2) / direct)                                             */
  int wino4w_waves;    /* [8] waves per workgroup (4 | 8) of the F(4x4,3x3) weight-gradient kernel                        */
  int chwn_group;      /* [1] batch-innermost trunk dgrad with positions grouped by tap count                             */
  int c8w_queue;       /* [1] workgroups per CU the C8 weight-gradient grids aim at                                       */
  int c8_stem_bf16;    /* [1] bf16-MFMA forms of the one-input-channel stems in bf16 mode                                 */
  int x3_plan;         /* [1] per-launch plan (walk order, reduction ranges) of the three-limb stride-2 kernels; 0: plain */
  int x3_ring;         /* [0] three-limb forward kernel with a 3-deep image ring and counted waits (round 6 experiment)    */
} LspsOptions;
int         lsps_set_options(const LspsOptions *opt);
int         lsps_get_options(LspsOptions *out);      /* the values in force (defaults resolved); out->struct_size is set */

/* ---- Conv2d: replaces nn.Conv2d forward + autograd's convolution_backward -----------------
 * call sites: common_net.py:250 (LeakyReLUConv2d), :162-163 (LeakyINSResBlock.conv3x3),
 *             lsps_nets.py:123-124 (Post / D heads).
 * x[N,C,H,W] * w[K,C,R,S] -> y[N,K,P,Q], P = (H+2*pad-R)/stride+1.                           */
size_t lsps_conv2d_workspace_bytes(int N, int C, int H, int W, int K, int R, int S, int stride, int pad);
int lsps_conv2d_fwd(const float *x, const float *w, const float *bias /*nullable*/, float *y,
                    int N, int C, int H, int W, int K, int R, int S, int stride, int pad,
                    int act, float slope, void *ws, size_t ws_bytes, void *stream);
int lsps_conv2d_dgrad(const float *dy, const float *w, float *dx,
                      int N, int C, int H, int W, int K, int R, int S, int stride, int pad,
                      void *ws, size_t ws_bytes, void *stream);
/* dw[K,C,R,S] and (optional) db[K] are OVERWRITTEN */
/* dx = conv2d_dgrad(dy, w) + addend: the input gradient of the FIRST conv of a residual block plus the gradient that
 * arrives over the skip connection (reference: autograd's sum at `out += residual`, common_net.py:177-181).  For the
 * 3x3 residual convs the addend is folded into the kernel's epilogue; otherwise dgrad followed by an add pass. */
int lsps_conv2d_dgrad_acc(const float *dy, const float *w, const float *addend, float *dx,
                          int N, int C, int H, int W, int K, int R, int S, int stride, int pad,
                          void *ws, size_t ws_bytes, void *stream);
/* dx = backward of [InstanceNorm2d + LeakyReLU(slope)] applied to conv2d_dgrad(dy, w): the input gradient of the SECOND
 * conv of a residual block pushed through the norm + activation in front of it (reference: autograd through
 * common_net.py:168-171 into :162), with the norm recovered from its saved OUTPUT `out_saved` [N,C,H,W] and `rstd` [N*C]
 * (as written by lsps_conv2d_in_fwd / lsps_inorm_fwd).  On 32x32 maps it happens in the epilogue of the F(4x4,3x3) dgrad
 * kernel (the un-normalised gradient never reaches HBM); otherwise dgrad followed by lsps_inorm_bwd in place. */
int lsps_conv2d_dgrad_inbwd(const float *dy, const float *w, const float *out_saved, const float *rstd, float *dx,
                            int N, int C, int H, int W, int K, float slope, void *ws, size_t ws_bytes, void *stream);
int lsps_conv2d_wgrad(const float *x, const float *dy, float *dw, float *db /*nullable*/,
                      int N, int C, int H, int W, int K, int R, int S, int stride, int pad,
                      void *ws, size_t ws_bytes, void *stream);

/* ---- grouped conv -------------------------------------------------------------------------------------------------------
 * nn.Conv2d(C, K, R, stride, pad, groups=G) of the ResNeXt block (reference common_net.py:111-132, `LeakyINSResNeXtBlock`:
 * Conv2d(k*inplanes, k*inplanes, 3, 1, 1, groups=cardinality), used by `SharedResXGen`, lsps_nets.py:277-387).
 * x [N,C,H,W], w [K, C/G, R, S], y / dy [N,K,P,Q]; group g maps channels [g C/G, (g+1) C/G) to [g K/G, (g+1) K/G).  One
 * launch of the gather-GEMM kernels per group straight on the channel slices of the full tensors (sample strides C*H*W /
 * K*P*Q): no slice copies, no concatenation.  Workspace: lsps_conv2d_workspace_bytes of the per-group geometry
 * (N, C/G, H, W, K/G, ...).  dw [K,C/G,R,S] and the optional db [K] are overwritten. */
int lsps_conv2d_grouped_fwd(const float *x, const float *w, const float *bias /*nullable*/, float *y,
                            int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int groups,
                            int act, float slope, void *ws, size_t ws_bytes, void *stream);
int lsps_conv2d_grouped_dgrad(const float *dy, const float *w, float *dx,
                              int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int groups,
                              void *ws, size_t ws_bytes, void *stream);
int lsps_conv2d_grouped_wgrad(const float *x, const float *dy, float *dw, float *db /*nullable*/,
                              int N, int C, int H, int W, int K, int R, int S, int stride, int pad, int groups,
                              void *ws, size_t ws_bytes, void *stream);

/* ---- 3x3 / stride 2 / pad 1 convs on small feature maps in batch-innermost layout [C][H][W][N] ("CHWN") ---------------
 * The discriminator trunk (reference lsps_nets.py:119-121 `_make_shared_net`: four LeakyReLUConv2d(tch, 2 tch, 3, 2, 1),
 * common_net.py:250-252) runs on 16x16 ... 2x2 maps with 128 ... 2048 channels; with the batch innermost every
 * (output position, tap) pair is a plain [K x C] x [C x N] GEMM over contiguous rows (no gather, padding taps skipped).
 * x [C][H][W][N], w [K][C][3][3] (the reference's layout), y [K][H/2][W/2][N]; needs N % 4 == 0, C % 128 == 0,
 * K % 128 == 0, even H and W.  lsps_transpose2d converts [R][S] -> [S][R] (NCHW <-> CHWN with R = N, S = C*H*W).
 * The LeakyReLU backward + bias gradient of such a layer is lsps_act_bwd_bias(dy, y, dpre, db, 1, K, (H/2)*(W/2)*N, ...). */
int lsps_transpose2d(const float *src, float *dst, long R, long S, void *stream);
size_t lsps_conv3x3s2_chwn_workspace_bytes(int N, int C, int H, int W, int K);
int lsps_conv3x3s2_chwn_fwd(const float *x, const float *w, const float *bias /*nullable*/, float *y,
                            int N, int C, int H, int W, int K, int act, float slope,
                            void *ws, size_t ws_bytes, void *stream);
int lsps_conv3x3s2_chwn_dgrad(const float *dy, const float *w, float *dx, int N, int C, int H, int W, int K,
                              void *ws, size_t ws_bytes, void *stream);
/* dw[K,C,3,3] is OVERWRITTEN */
int lsps_conv3x3s2_chwn_wgrad(const float *x, const float *dy, float *dw, int N, int C, int H, int W, int K,
                              void *ws, size_t ws_bytes, void *stream);

/* ---- bf16 residual trunk in the channel-group layout "C8" (BASELINE config 5: exps/nicvl.yaml, bf16 MFMA conv path) -----
 * The chain of LeakyINSResBlock layers of SharedResGen (reference common_net.py:160-181, instantiated at
 * lsps_nets.py:176-179,195-229 on 32x32 maps with 4*ch channels) on bf16 activations stored [N][C/8][H][W][8]: the 8
 * channels of a pixel are one 16-byte unit = one v_mfma_f32_32x32x16_bf16 operand fragment, so operands are staged by
 * LDS-DMA without conversion and every tap is a unit offset.  `void *` tensors are bf16 in that layout; weights stay f32
 * (K, C, 3, 3) (the reference's state-dict layout) and are packed to bf16 per call (cached in the pack-cache scope);
 * statistics and accumulation are f32.  Geometry: H = W = 32; conv entries C % 16 == 0 and K % 64 == 0 (outputs), the
 * weight gradient C % 64 == 0 and K % 128 == 0.
 *   lsps_c8_from_nchw / _to_nchw   layout + precision conversion at the ends of the chain (x [N,C,HW] f32)
 *   lsps_c8_add / lsps_c8_add_nchw GaussianNoiseLayer on a C8 tensor (common_net.py:39-40), n = element count; _nchw: the noise
 *                                  as the f32 [N,C,HW] tensor it was drawn into (no conversion pass)
 *   lsps_c8_conv3x3_fwd            y = conv3x3(x, w) (+ addend)                                   (common_net.py:162)
 *   lsps_c8_conv3x3_in_fwd         residual == NULL: y = LeakyReLU_slope(InstanceNorm(conv(x, w))) (:162-169, slope < 0: none)
 *                                  residual != NULL: y = InstanceNorm(conv(x, w)) + residual       (:163-181); rstd [N*K] out
 *   lsps_c8_conv3x3_dgrad_acc      dx = conv3x3_dgrad(dy, w) + addend (addend nullable)            (autograd of :162 + skip)
 *   lsps_c8_conv3x3_dgrad_inbwd    dx = backward of [InstanceNorm + LeakyReLU(slope)] applied to conv3x3_dgrad(dy, w), the
 *                                  norm recovered from its saved output `out_saved` [N,C,..] and `rstd` [N*C]
 *   lsps_c8_conv3x3_wgrad          dw [K,C,3,3] f32 = weight gradient from x [N,C,..] and dy [N,K,..] (OVERWRITTEN)
 *   lsps_c8_inorm_bwd              backward of InstanceNorm (+ residual | + LeakyReLU(slope)) from the output (lsps_inorm_bwd
 *                                  on C8 tensors; planes of <= 1024 pixels) */
int lsps_c8_conv3x3_ok(int N, int C, int H, int W, int K);            /* 1 if the conv entries take this geometry */
size_t lsps_c8_conv3x3_workspace_bytes(int C, int K);
size_t lsps_c8_conv3x3_wgrad_workspace_bytes(int N, int C, int K);
int lsps_c8_from_nchw(const float *x, void *y, int N, int C, int HW, void *stream);
int lsps_c8_to_nchw(const void *x, float *y, int N, int C, int HW, void *stream);
int lsps_c8_add(const void *a, const void *b, void *out, long n, void *stream);
int lsps_c8_add_nchw(const void *a, const float *b /* f32 [N,C,HW] */, void *out, int N, int C, int HW, void *stream);
int lsps_c8_conv3x3_fwd(const void *x, const float *w, const void *addend /*nullable*/, void *y, int N, int C, int H, int W, int K,
                        void *ws, size_t ws_bytes, void *stream);
int lsps_c8_conv3x3_in_fwd(const void *x, const float *w, const void *residual /*nullable*/, void *y, float *rstd,
                           int N, int C, int H, int W, int K, float slope, float eps, void *ws, size_t ws_bytes, void *stream);
int lsps_c8_conv3x3_dgrad_acc(const void *dy, const float *w, const void *addend /*nullable*/, void *dx,
                              int N, int C, int H, int W, int K, void *ws, size_t ws_bytes, void *stream);
int lsps_c8_conv3x3_dgrad_inbwd(const void *dy, const float *w, const void *out_saved, const float *rstd, void *dx,
                                int N, int C, int H, int W, int K, float slope, void *ws, size_t ws_bytes, void *stream);
int lsps_c8_conv3x3_wgrad(const void *x, const void *dy, float *dw, int N, int C, int H, int W, int K,
                          void *ws, size_t ws_bytes, void *stream);
int lsps_c8_inorm_bwd(const void *dout, const void *out, const void *residual /*nullable*/, const float *rstd, void *dy,
                      int N, int C, int HW, float slope, void *stream);

/* ---- bf16 stride-2 3x3 convs on C8 tensors (BASELINE config 5) -------------------------------------------------------
 * The 3x3 / stride 2 / pad 1 LeakyReLUConv2d layers (reference common_net.py:246-256; generator down-sampling
 * lsps_nets.py:186-192, discriminator front + shared trunk lsps_nets.py:117-124 on 64x64 ... 2x2 output maps) and the
 * 3x3 / stride 2 / pad 1 / output_padding 1 LeakyReLUConvTranspose2d layers (common_net.py:258-268, lsps_nets.py:222-225)
 * on bf16 activations in the C8 layout above; weights f32 in the reference's layouts ((K,C,3,3) resp. (Ci,Co,3,3)), packed
 * to bf16 per call (cached in the pack-cache scope); bias f32; f32 accumulation.  H, W (the LARGER map of the layer) are
 * powers of two; channels: conv C % 64 == 0, K % 128 == 0; transposed conv Ci % 128 == 0, Co % 64 == 0 (see the _ok entries).
 * slope < 0: no activation.  One workspace size covers the three directions of a layer.
 *   lsps_c8_conv3x3s2_fwd       y [N,K,H/2,W/2] = LeakyReLU_slope(conv(x [N,C,H,W], w) + bias)       (common_net.py:250-252)
 *   lsps_c8_conv3x3s2_dgrad     dx [N,C,H,W] from dy [N,K,H/2,W/2] (the gradient w.r.t. the conv's OUTPUT, activation undone)
 *   lsps_c8_conv3x3s2_wgrad     dw [K,C,3,3] f32 (OVERWRITTEN)
 *   lsps_c8_convT3x3s2_fwd      y [N,Co,2H,2W] = LeakyReLU_slope(convT(x [N,Ci,H,W], w) + bias)      (common_net.py:262-264)
 *   lsps_c8_convT3x3s2_dgrad    dx [N,Ci,H,W] from dy [N,Co,2H,2W]
 *   lsps_c8_convT3x3s2_wgrad    dw [Ci,Co,3,3] f32 (OVERWRITTEN)
 *   lsps_c8_act_bwd_bias        g = dy * LeakyReLU'(y) from the layer's OUTPUT y (slope > 0 keeps the sign), db [C] = sum of g
 *                               over n and pixels (db nullable); the C8 form of lsps_act_bwd_bias */
int lsps_c8_conv3x3s2_ok(int N, int C, int H, int W, int K);
int lsps_c8_convT3x3s2_ok(int N, int Ci, int H, int W, int Co);
size_t lsps_c8_conv3x3s2_workspace_bytes(int N, int C, int H, int W, int K);     /* convT: (N, Co, 2H, 2W, Ci) */
int lsps_c8_conv3x3s2_fwd(const void *x, const float *w, const float *bias /*nullable*/, void *y, int N, int C, int H, int W, int K,
                          float slope, void *ws, size_t ws_bytes, void *stream);
int lsps_c8_conv3x3s2_dgrad(const void *dy, const float *w, void *dx, int N, int C, int H, int W, int K,
                            void *ws, size_t ws_bytes, void *stream);
int lsps_c8_conv3x3s2_wgrad(const void *x, const void *dy, float *dw, int N, int C, int H, int W, int K,
                            void *ws, size_t ws_bytes, void *stream);
int lsps_c8_convT3x3s2_fwd(const void *x, const float *w, const float *bias /*nullable*/, void *y, int N, int Ci, int H, int W, int Co,
                           float slope, void *ws, size_t ws_bytes, void *stream);
int lsps_c8_convT3x3s2_dgrad(const void *dy, const float *w, void *dx, int N, int Ci, int H, int W, int Co,
                             void *ws, size_t ws_bytes, void *stream);
int lsps_c8_convT3x3s2_wgrad(const void *x, const void *dy, float *dw, int N, int Ci, int H, int W, int Co,
                             void *ws, size_t ws_bytes, void *stream);
size_t lsps_c8_act_bwd_bias_workspace_bytes(int N, int C);
int lsps_c8_act_bwd_bias(const void *dy, const void *y, void *g, float *db /*nullable*/, int N, int C, int HW, float slope,
                         void *ws, size_t ws_bytes, void *stream);

/* ---- the two ends of the generator / discriminator in the bf16 math mode ------------------------------------------------
 *   lsps_c8_stem_fwd     LeakyReLUConv2d(1, 64, 7, stride, 3) stems (lsps_nets.py:117,184): f32 image x [N,1,H,W] in, C8 bf16
 *                        activation y out (slope < 0: no activation)
 *   lsps_c8_stem_wgrad   dw [K,1,R,S] and db [K] (nullable) from the image, the C8 gradient dy w.r.t. the layer's OUTPUT and the
 *                        layer's saved C8 output y: the LeakyReLU backward happens while dy is staged, the bias gradient is
 *                        one more column of the same product (replaces lsps_act_bwd_bias + lsps_conv2d_wgrad)
 *   lsps_c8_pw1_*        ConvTranspose2d(C, 1, kernel 1) + Tanh output head (lsps_nets.py:226-229) on a C8 input x:
 *                        y [N,HW] f32 = act(b + sum_c w[c] x[c]); dx (C8) = w[c] * dpre; dw [C], db [1] from x and dpre, where
 *                        dpre [N,HW] f32 is the gradient w.r.t. the pre-activation (lsps_act_bwd of dy) */
int lsps_c8_stem_ok(int N, int H, int W, int K, int R, int S, int stride, int pad);
size_t lsps_c8_stem_workspace_bytes(int K, int R, int S);
int lsps_c8_stem_fwd(const float *x, const float *w, const float *bias /*nullable*/, void *y, int N, int H, int W, int K, int R, int S,
                     int stride, int pad, float slope, void *stream);
int lsps_c8_stem_wgrad(const float *x, const void *dy, const void *y, float *dw, float *db /*nullable*/, int N, int H, int W, int K,
                       int R, int S, int stride, int pad, float slope, void *ws, size_t ws_bytes, void *stream);
/* dx [N,1,H,W] f32 of a stem from the C8 gradient dy w.r.t. its OUTPUT and its saved C8 output y (LeakyReLU backward applied on
 * the fly): the discriminator stems inside gen_update; bf16 tap GEMM + deterministic in-LDS col2im in one kernel */
int lsps_c8_stem_dgrad_ok(int N, int H, int W, int K, int R, int S, int stride, int pad);
int lsps_c8_stem_dgrad(const void *dy, const void *y, const float *w, float *dx, int N, int H, int W, int K, int R, int S,
                       int stride, int pad, float slope, void *stream);
size_t lsps_c8_pw1_workspace_bytes(int N, int C);
int lsps_c8_pw1_fwd(const void *x, const float *w, const float *bias /*nullable*/, float *y, int N, int C, int HW, int act, float slope,
                    void *stream);
int lsps_c8_pw1_dgrad(const float *dpre, const float *w, void *dx, int N, int C, int HW, void *stream);
int lsps_c8_pw1_wgrad(const void *x, const float *dpre, float *dw, float *db /*nullable*/, int N, int C, int HW,
                      void *ws, size_t ws_bytes, void *stream);

/* dgrad entries with the PREVIOUS layer's LeakyReLU backward fused into the epilogue (bf16 math mode): the layer in front of
 * this one saved its output act_y (the shape of dx); the gradient handed to it is already multiplied by LeakyReLU'(act_y)
 * (slope act_slope >= 0) and its bias gradient db_prev [channels of dx] (nullable) comes out of per-workgroup partial sums —
 * this replaces that layer's lsps_c8_act_bwd_bias pass (autograd of common_net.py:252,264 folded into the consumer's dgrad). */
int lsps_c8_conv3x3s2_dgrad_act(const void *dy, const float *w, const void *act_y, float act_slope, void *dx, float *db_prev,
                                int N, int C, int H, int W, int K, void *ws, size_t ws_bytes, void *stream);
int lsps_c8_convT3x3s2_dgrad_act(const void *dy, const float *w, const void *act_y, float act_slope, void *dx, float *db_prev,
                                 int N, int Ci, int H, int W, int Co, void *ws, size_t ws_bytes, void *stream);
size_t lsps_c8_pw1_dgrad_act_workspace_bytes(int N, int C);
/* dw [C], db [1] (both nullable): the head's OWN weight / bias gradient from the same pass (act_y is its input) */
int lsps_c8_pw1_dgrad_act(const float *dpre, const float *w, const void *act_y, float act_slope, void *dx, float *db_prev,
                          float *dw, float *db, int N, int C, int HW, void *ws, size_t ws_bytes, void *stream);

/* f32 mode: the LeakyReLU backward of two layers folded into kernels that stream their tensors anyway.
 *   lsps_conv2d_stem_wgrad_act  dw [K,1,R,S] and db [K] (nullable) of a one-input-channel LeakyReLUConv2d (the 7x7 stems,
 *                               lsps_nets.py:117,184) from the image x, the gradient dy w.r.t. the layer's OUTPUT and the saved
 *                               output y: replaces lsps_act_bwd_bias + lsps_conv2d_wgrad when no input gradient is needed
 *   lsps_pw1_dgrad_act          dx [N,C,HW] = w[c] * dpre [N,HW] * LeakyReLU'(act_y [N,C,HW]) of the ConvTranspose2d(C, 1, 1) output
 *                               head (lsps_nets.py:226-227) with the backward of the layer in front of it fused; db_prev [C]
 *                               (nullable) = that layer's bias gradient */
int lsps_conv2d_stem_wgrad_act_ok(int N, int H, int W, int K, int R, int S, int stride, int pad);
int lsps_conv2d_stem_wgrad_act(const float *x, const float *dy, const float *y, float *dw, float *db /*nullable*/, int N, int H, int W,
                               int K, int R, int S, int stride, int pad, float slope, void *ws, size_t ws_bytes, void *stream);
size_t lsps_pw1_dgrad_act_workspace_bytes(int N, int C);
int lsps_pw1_dgrad_act(const float *dpre, const float *w, const float *act_y, float act_slope, float *dx, float *db_prev /*nullable*/,
                       float *dw /*nullable: the head's own weight gradient [C]*/, float *db /*nullable: its bias gradient [1]*/,
                       int N, int C, int HW, void *ws, size_t ws_bytes, void *stream);
/* the same with dx written as a three-limb X3 tensor [N][3][C/8][HW][8] (C % 16 == 0): the operand format of the X3 transposed conv
 * in front of the head (lsps_x3_convT3x3s2_*), whose separate split pass this saves */
int lsps_pw1_dgrad_act_x3(const float *dpre, const float *w, const float *act_y, float act_slope, void *dxl, float *db_prev /*nullable*/,
                          float *dw /*nullable*/, float *db /*nullable*/, int N, int C, int HW, void *ws, size_t ws_bytes, void *stream);

/* ---- ConvTranspose2d: replaces nn.ConvTranspose2d forward/backward -------------------------
 * call sites: common_net.py:262 (LeakyReLUConvTranspose2d), lsps_nets.py:226-227 (1x1 output),
 *             lsps_nets.py:17-23 (Mapping).
 * x[N,Ci,H,W] * w[Ci,Co,R,S] -> y[N,Co,Ho,Wo], Ho = (H-1)*stride-2*pad+R+outpad.             */
size_t lsps_convT2d_workspace_bytes(int N, int Ci, int H, int W, int Co, int R, int S, int stride, int pad, int outpad);
int lsps_convT2d_fwd(const float *x, const float *w, const float *bias, float *y,
                     int N, int Ci, int H, int W, int Co, int R, int S, int stride, int pad, int outpad,
                     int act, float slope, void *ws, size_t ws_bytes, void *stream);
int lsps_convT2d_dgrad(const float *dy, const float *w, float *dx,
                       int N, int Ci, int H, int W, int Co, int R, int S, int stride, int pad, int outpad,
                       void *ws, size_t ws_bytes, void *stream);
int lsps_convT2d_wgrad(const float *x, const float *dy, float *dw, float *db,
                       int N, int Ci, int H, int W, int Co, int R, int S, int stride, int pad, int outpad,
                       void *ws, size_t ws_bytes, void *stream);

/* ---- 3x3 / stride 1 / pad 1 Conv2d + InstanceNorm2d(affine=False) [+ LeakyReLU | + residual], ONE call -------------
 * The two conv -> norm pairs of LeakyINSResBlock (common_net.py:162-163 + 166-171; :162-163 + 177-181):
 *   residual == NULL: y = lrelu_slope(IN(conv3x3(x, w)))   (slope < 0: no activation)
 *   residual != NULL: y = IN(conv3x3(x, w)) + residual     (slope must be < 0)
 * The conv bias is not an argument: it cancels exactly under an affine-free InstanceNorm.  rstd[N*K] is saved for
 * lsps_inorm_bwd (which works from y).  On 32x32 maps with K % 32 == 0 the Winograd F(4x4,3x3) kernel owns whole (n, k)
 * planes per workgroup and normalises in its epilogue, so the pre-norm tensor never reaches HBM; every other shape runs the
 * dispatched conv kernel followed by lsps_inorm_fwd in place.  ws as for lsps_conv2d_fwd. */
int lsps_conv2d_in_fwd(const float *x, const float *w, const float *residual /*nullable*/, float *y, float *rstd,
                       int N, int C, int H, int W, int K, float slope, float eps,
                       void *ws, size_t ws_bytes, void *stream);
/* The same call for a pass that NO backward follows (torch.no_grad(): the generator inside dis_update / post_update,
 * lsps_trainer.py:145-146,238-241; evaluation).  Identical contract; the difference is dispatch freedom: the estimate modes run
 * the generator on 4 + 4 samples (lsps_trainer.py:238), 32 - 64 workgroups of the F(4x4,3x3) kernel, so here such launches
 * split the input channels over up to 8 workgroups per (image, k slice) and sum the partial outputs inside the norm kernel
 * (58 -> 42 us per conv + norm at N = 8).  F(4x4,3x3) rounds at 8e-6 of abs-max where the F(2x2,3x3) / direct kernels that
 * lsps_conv2d_in_fwd keeps for few-image launches round at 5e-7: irrelevant for an output (north_star: 1e-3), but a training
 * step's gradient is discontinuous in its forward pass (LeakyReLU kinks, sign() of the L1 losses), and the golden step cases at
 * N = 2 pin Adam's first sign decisions — so passes that are differentiated keep the kernels those vectors were checked with. */
int lsps_conv2d_in_fwd_nograd(const float *x, const float *w, const float *residual /*nullable*/, float *y, float *rstd,
                              int N, int C, int H, int W, int K, float slope, float eps,
                              void *ws, size_t ws_bytes, void *stream);

/* ---- InstanceNorm2d(affine=False) [+ LeakyReLU] [+ residual add], fused --------------------
 * call sites: common_net.py:168-171 (norm + in-place LeakyReLU), :177-181 (norm, out += residual).
 * planes = N*C, hw = H*W.  out = act(IN(y)) (+ residual).  slope < 0 => no activation.
 * `out` may alias `y`.  rstd[planes] is saved for the backward.
 * The backward works from the OUTPUT (x_hat is recovered from out / out-residual).            */
int lsps_inorm_fwd(const float *y, const float *residual /*nullable*/, float *out, float *rstd,
                   int planes, int hw, float eps, float slope, void *stream);
int lsps_inorm_bwd(const float *dout, const float *out, const float *residual /*nullable*/,
                   const float *rstd, float *dy, int planes, int hw, float slope, void *stream);

/* ---- activation backward from the saved OUTPUT (in-place LeakyReLU / Tanh) ------------------
 * dx = dy * act'(.) ; kind = LSPS_ACT_LRELU (uses sign of out) or LSPS_ACT_TANH (1-out^2).
 * dx may alias dy.  (autograd of common_net.py:252,264; lsps_nets.py:228)                     */
int lsps_act_bwd(const float *dy, const float *out, float *dx, long n, int kind, float slope, void *stream);
/* lsps_act_bwd fused with the bias gradient of a conv / transposed-conv layer whose output went through the
 * activation (reference: autograd of nn.LeakyReLU / nn.Tanh followed by the bias term of convolution_backward,
 * common_net.py:250-252, 262-264): dx = dy * act'(out) for [N][C][HW] tensors and db[c] = sum_{n,hw} dx.  One pass.
 * ws: >= 64*C floats of scratch (the layer's conv workspace is large enough). */
int lsps_act_bwd_bias(const float *dy, const float *out, float *dx, float *db, int N, int C, int HW, int kind,
                      float slope, void *ws, size_t ws_bytes, void *stream);

/* ---- losses (lsps_trainer.py:42-60, :107-112, :172-192; helpers.py:20-32) -------------------
 * Forward kernels write ONE float (already divided by `denom`) to out[0].
 *   L1  : sum|a-b| / denom          (nn.L1Loss; b == NULL means zeros: the feature-matching form)
 *   L2  : sum (a-b)^2 / denom       (_compute_l2_loss)
 *   SQ  : sum a^2 / denom           (_compute_kl(mu))
 *   KLSD: sum (mu^2+sd^2-log sd^2)/denom   (a = mu, b = sd; _compute_kl(mu, sd))
 * Backward kernels write da (and db = -da when db != NULL), scaled by gout[0] (device scalar). */
#define LSPS_LOSS_L1   0
#define LSPS_LOSS_L2   1
#define LSPS_LOSS_SQ   2
#define LSPS_LOSS_KLSD 3
size_t lsps_loss_workspace_bytes(long n);
int lsps_loss_fwd(int kind, const float *a, const float *b, long n, float denom, float *out,
                  void *ws, size_t ws_bytes, void *stream);
int lsps_loss_bwd(int kind, const float *a, const float *b, long n, float denom, const float *gout,
                  float *da, float *db, void *stream);
/* sigmoid + binary_cross_entropy against a constant target (1.0 or 0.0), mean over n logits,
 * with torch's clamp of log() at -100.  out[0] = loss, out[1] = #(p >= 0.5), out[2] = #(p <= 0.5).
 * (lsps_trainer.py:107-112,179-192 ; counters feed helpers.py:20-32)                           */
int lsps_bce_sigmoid_fwd(const float *logits, long n, float target, float *out3,
                         void *ws, size_t ws_bytes, void *stream);
int lsps_bce_sigmoid_bwd(const float *logits, long n, float target, const float *gout, float *dlogits, void *stream);

/* ---- pose-MLP head: nn.Linear (+ LeakyReLU / Softplus), lsps_nets.py:44-50,73-83 ------------
 * y[n,out] = act(x[n,in] @ W[out,in]^T + b[out]); act: 0 none, 1 LeakyReLU(slope), 3 Softplus.  */
#define LSPS_ACT_SOFTPLUS 3
int lsps_linear_fwd(const float *x, const float *w, const float *b, float *y, int n, int in, int out,
                    int act, float slope, void *stream);
/* dz = dy * act'(y) is formed internally from the saved output y; dx (nullable), dw, db overwritten */
int lsps_linear_bwd(const float *x, const float *w, const float *y, const float *dy,
                    float *dx, float *dw, float *db, int n, int in, int out, int act, float slope,
                    float *ws_dz /* n*out floats */, void *stream);

/* ---- Adam with coupled L2 weight decay over a flat parameter arena --------------------------
 * replaces torch.optim.Adam.step (lsps_trainer.py:26-29,72,131,213,258).
 * seg_* describe `nseg` parameter tensors laid out back-to-back in p/g/m/v (device arrays):
 *   seg_off[i] (element offset, 64-bit), seg_len[i], seg_bc1[i] = 1-b1^step_i,
 *   seg_bc2s[i] = sqrt(1-b2^step_i); seg_len[i] == 0 marks a tensor without a gradient this
 *   step (skipped entirely, as torch does for grad=None).  max_seg_len = max_i seg_len[i] (host
 *   copy, sizes the grid).  gscale multiplies g first (1/world_size under data parallelism). */
int lsps_adam_step(float *p, const float *g, float *m, float *v,
                   const long *seg_off, const int *seg_len, const float *seg_bc1, const float *seg_bc2s,
                   int nseg, int max_seg_len, float lr, float beta1, float beta2, float eps, float weight_decay,
                   float gscale, void *stream);

/* ---- BatchNorm (nn.BatchNorm2d / nn.BatchNorm1d of the BN block variants of common_net.py:183-322; not instantiated by
 * the shipped configs) [+ LeakyReLU].  x, y: [N][C][HW] (HW = 1 for BatchNorm1d).  training != 0: batch statistics
 * (biased variance for the normalisation, running statistics updated with the unbiased one and `momentum`, as torch
 * does; run_mean / run_var may be NULL); training == 0: running statistics.  gamma / beta nullable (affine=False, or
 * beta alone = the `Bias2d` of the "BNNS" blocks, common_net.py:92-105,296-322).  slope < 0: no activation.
 * mean[C] / rstd[C] are outputs saved for the backward, which returns dx and (nullable) dgamma, dbeta; g is the
 * gradient w.r.t. the PRE-activation output (apply lsps_act_bwd first).  ws: lsps_bnorm_workspace_bytes(C). */
size_t lsps_bnorm_workspace_bytes(int C);
int lsps_bnorm_fwd(const float *x, const float *gamma, const float *beta, float *run_mean, float *run_var, float *y,
                   float *mean, float *rstd, int N, int C, int HW, int training, float eps, float momentum, float slope,
                   void *ws, size_t ws_bytes, void *stream);
int lsps_bnorm_bwd(const float *g, const float *x, const float *mean, const float *rstd, const float *gamma, float *dx,
                   float *dgamma, float *dbeta, int N, int C, int HW, int training, void *ws, size_t ws_bytes, void *stream);
/* out = act(x), kind = LSPS_ACT_LRELU (slope 0 = nn.ReLU, common_net.py:146,361) | LSPS_ACT_TANH | LSPS_ACT_SOFTPLUS
 * (GaussianVAE2D, common_net.py:71-80); lsps_act_bwd is its backward (from the output). */
int lsps_act_fwd(const float *x, float *out, long n, int kind, float slope, void *stream);

/* elementwise helpers used by the trainer glue (GaussianNoiseLayer common_net.py:39-40 etc.) */
int lsps_axpy(const float *x, const float *y, float alpha, float *out, long n, void *stream);  /* out = x + alpha*y */
/* out = (x ? x : 0) + t*m: nn.Dropout on the residual branch of a residual block (common_net.py:171-172, m = keep mask/(1-p))
 * followed by `out += residual` (:180); with x == NULL it is the branch's backward (dt = dy*m). */
int lsps_mul_add(const float *x, const float *t, const float *m, float *out, long n, void *stream);

/* ---- f32-class stride-2 conv on the bf16 matrix pipe: three-limb "X3" tensors (round 5 prototype, csrc/x3s2.h) -------------
 * An X3 tensor carries an f32 activation as three bf16 limbs per element, x = hi + mid + lo EXACTLY (8 + 8 + 8 significand
 * bits), as three C8 planes per image: [N][limb 3][C/8][H][W][8] bf16.  A product of two such operands is formed by six
 * v_mfma_f32_32x32x16_bf16 (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi; the dropped terms are < 2^-24 relative), f32
 * accumulation: f32-class arithmetic at 192 instead of 512 matrix-pipe cycles per 32x32x16 MACs.
 *   lsps_x3_split_nchw      f32 [N,C,HW] -> X3 (C % 8 == 0);  lsps_x3_join_nchw the inverse (exact)
 *   lsps_x3_conv3x3s2_fwd   LeakyReLUConv2d(C, K, 3, 2, 1) (reference: src/trainers/common_net.py:246-256; lsps_nets.py:119-123,
 *                           186-192): y = LeakyReLU_slope(conv(x, w) + bias) from an X3 input; w f32 (K,C,3,3), split into limbs by
 *                           the pack kernel (cached in the pack-cache scope); output f32 NCHW `y` or, when `yl` is given, X3 `yl`
 *                           (the next stride-2 layer's input).  C % 16 == 0, K % 128 == 0, H, W powers of two (_ok). */
int lsps_x3_split_nchw(const float *x, void *xl, int N, int C, int HW, void *stream);
/* the one-input-channel stems (LeakyReLUConv2d(1, 64, 7, s, 3), lsps_nets.py:117,184) writing their activation straight as X3
 * (the f32 kernel of lsps_conv2d_fwd with a limb-splitting epilogue): y = LeakyReLU_slope(conv(x, w) + bias), slope < 0: none */
int lsps_x3_stem_ok(int N, int H, int W, int K, int R, int S, int stride, int pad);
int lsps_x3_stem_fwd(const float *x, const float *w, const float *bias /*nullable*/, void *yl, int N, int H, int W, int K, int R, int S,
                     int stride, int pad, float slope, void *stream);
int lsps_x3_join_nchw(const void *xl, float *y, int N, int C, int HW, void *stream);
int lsps_x3_conv3x3s2_ok(int N, int C, int H, int W, int K);
size_t lsps_x3_conv3x3s2_workspace_bytes(int N, int C, int H, int W, int K);
/* the launch plan of Conv2d(C, K, 3, 2, 1) on [N,C,H,W] (no GPU needed; what the calls below will do): plan[0..3] = {k ranges
 * (1: no split-K), 16-channel chunks per range, workgroup walk (0: pixel tile -> XCD tile % 8, 1: linear, m tile / range fastest),
 * workgroups} of the forward kernel (transposed = 0) or of the dgrad / ConvTranspose2d-forward kernel (transposed = 1).
 * Returns 0, or LSPS_E_ARG for an unsupported geometry. */
int lsps_x3_conv3x3s2_plan(int transposed, int N, int C, int H, int W, int K, int plan[4]);
int lsps_x3_conv3x3s2_fwd(const void *xl, const float *w, const float *bias /*nullable*/, float *y /*nullable*/, void *yl /*nullable*/,
                          int N, int C, int H, int W, int K, float slope, void *ws, size_t ws_bytes, void *stream);
/* dx [N,C,H,W] (f32 NCHW `dx` or X3 `dxl`) from the X3 gradient dyl [N,K,H/2,W/2] w.r.t. the conv's OUTPUT (its own activation
 * already undone).  act_yl != NULL: dx is the gradient w.r.t. the OUTPUT of the layer in front (X3 saved output act_yl, dx's shape):
 * the epilogue multiplies by LeakyReLU'(act_yl) (slope act_slope) and db_prev [C] (nullable) receives that layer's bias gradient.
 * lsps_x3_conv3x3s2_wgrad: dw [K,C,3,3] f32 (OVERWRITTEN) from X3 x and X3 dy (autograd of common_net.py:250). */
int lsps_x3_conv3x3s2_dgrad(const void *dyl, const float *w, float *dx /*nullable*/, void *dxl /*nullable*/, const void *act_yl /*nullable*/,
                            float act_slope, float *db_prev /*nullable*/, int N, int C, int H, int W, int K,
                            void *ws, size_t ws_bytes, void *stream);
int lsps_x3_conv3x3s2_wgrad(const void *xl, const void *dyl, float *dw, int N, int C, int H, int W, int K,
                            void *ws, size_t ws_bytes, void *stream);
/* LeakyReLUConvTranspose2d(Ci, Co, 3, 2, 1, 1) (common_net.py:258-268, lsps_nets.py:222-225): x [N,Ci,H,W] X3 -> y [N,Co,2H,2W]
 * (f32 NCHW `y` or X3 `yl`); w (Ci,Co,3,3); dgrad: dx [N,Ci,H,W] from the X3 gradient w.r.t. the layer's output (activation
 * undone); wgrad: dw (Ci,Co,3,3) OVERWRITTEN.  Ci % 128 == 0, Co % 64 == 0.  Workspace: lsps_x3_conv3x3s2_workspace_bytes(N, Co, 2H, 2W, Ci). */
int lsps_x3_convT3x3s2_ok(int N, int Ci, int H, int W, int Co);
int lsps_x3_convT3x3s2_fwd(const void *xl, const float *w, const float *bias /*nullable*/, float *y /*nullable*/, void *yl /*nullable*/,
                           int N, int Ci, int H, int W, int Co, float slope, void *ws, size_t ws_bytes, void *stream);
int lsps_x3_convT3x3s2_dgrad(const void *dyl, const float *w, float *dx /*nullable*/, void *dxl /*nullable*/, const void *act_yl /*nullable*/,
                             float act_slope, float *db_prev /*nullable*/, int N, int Ci, int H, int W, int Co,
                             void *ws, size_t ws_bytes, void *stream);    /* act_yl / db_prev: as lsps_x3_conv3x3s2_dgrad */
int lsps_x3_convT3x3s2_wgrad(const void *xl, const void *dyl, float *dw, int N, int Ci, int H, int W, int Co,
                             void *ws, size_t ws_bytes, void *stream);
/* g (X3) = dy * LeakyReLU'(y) from f32 NCHW dy and the layer's f32 NCHW output y (slope < 0: g = dy), db [C] (nullable) = sum of g:
 * the X3-emitting form of lsps_act_bwd_bias for a layer whose output left the X3 family as f32 (common_net.py:252). */
size_t lsps_x3_act_bwd_bias_workspace_bytes(int N, int C);
int lsps_x3_act_bwd_bias(const float *dy, const float *y /*nullable*/, void *gl, float *db /*nullable*/, int N, int C, int HW, float slope,
                         void *ws, size_t ws_bytes, void *stream);

/* ---- data step either side of the path -------------------------------------------------
 * lsps_crop_normalize: reference src/data/dataset_hand2.py:27-31 `normalize(img, com, cube)` for a batch:
 *   out = (dpt == 0 ? com_z + half : dpt) - com_z) / half, with half = cube_z / 2.  dpt/out: [N][HW] device floats
 *   (HW % 4 == 0), com_z/half: [N] device floats.  May run in place.
 * lsps_crop_augment: the image part of src/data/dataset_hand2.py:34-119 `augmentCrop(...)` (normZeroOne=False) for
 *   a batch of normalised crops x[N][H][W] -> out[N][H][W] (NOT in place): de-normalise, pre-max, nearest-
 *   neighbour warp (cv2.warpPerspective for 'com'/'sc' via HandDetector.recropHand handdetector.py:786-805 with its
 *   32000 / z-threshold fix-ups, cv2.warpAffine for 'rot' handdetector.py:733-739), then the premax / zero / clip /
 *   re-normalise tail.  prm: [N][LSPS_AUG_STRIDE] device doubles, per sample:
 *     [0] kind: 0 none, 1 perspective, 2 affine      [1] com_z in   [2] cube_z/2 in   [3] com_z out  [4] cube_z/2 out
 *     [5] zstart  [6] zend (kind 1)                   [7..15] the INVERTED map, row-major (3x3, or 2x3 + padding)
 *   The per-sample geometry that fills prm is host work: lsps_amd/data.py (mirror of HandDetector.moveCoM /
 *   rotateHand / scaleHand / comToTransform). */
#define LSPS_AUG_STRIDE 16
int lsps_crop_normalize(const float *dpt, const float *com_z, const float *half, float *out, int N, int HW, void *stream);
int lsps_crop_augment(const float *x, const double *prm, float *out, int N, int H, int W, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LSPS_HIP_H */
