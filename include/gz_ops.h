/* gz_ops.h -- C ABI of libgz_hip.so: the MI355X (gfx950) kernels behind the G+D training step of
 * ebartrum/lightning_gan_zoo.
 *
 * The reference has no FFI of its own: its hot path is torch.nn operator calls made from
 *   core/models/standard_networks.py:9-93        (DCGAN/WGAN generator + discriminator)
 *   core/lightning_module.py:104-128,158-207     (DCGAN / WGAN / WGANGP training_step)
 *   core/utils/utils.py:39-58                    (gradient_penalty)
 * Each entry point below replaces the aten operator(s) named in its comment; the Python side
 * (lightning_gan_zoo_amd/functional/) binds them with ctypes and wraps them in
 * torch.autograd.Function objects (see INTEGRATION.md for the binding a maintainer would add).
 *
 * Conventions
 *   - plain pointers to DEVICE memory, fp32, contiguous NCHW; sizes as int; no torch types.
 *   - every launcher takes the HIP stream explicitly, never synchronises, never allocates, and is
 *     re-entrant (autograd runs backward on its own thread).
 *   - return value: 0 on success, negative error code otherwise (GZ_ERR_*); never throws.
 *   - workspaces are caller-allocated; *_workspace_bytes / *_elems tell how much.
 */
#ifndef GZ_OPS_H
#define GZ_OPS_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

#define GZ_ACT_NONE 0
#define GZ_ACT_RELU 1
#define GZ_ACT_LRELU 2
#define GZ_ACT_TANH 3

/* ---- convolution family -------------------------------------------------------------------
 * Geometry names: x [N,C,H,W] is the image side, y [N,K,OH,OW] the feature side,
 * w [K,C,KH,KW] the Conv2d weight (a ConvTranspose2d weight [Cin,Cout,KH,KW] is the same array
 * with K = Cin, C = Cout).  Supported (KH,KW,S,P): (4,4,2,1) (5,5,2,2) (3,3,1,1) (1,1,1,0).
 */

/* packed-weight sizes in floats */
long long gz_conv2d_pack_fwd_elems(int K, int C, int KH, int KW);
long long gz_conv2d_pack_dgrad_elems(int K, int C, int KH, int KW, int S);

/* w -> [C*KH*KW][round4(K)] GEMM-B image for gz_conv2d_fwd */
int gz_conv2d_pack_fwd(const float* w, float* wpack, int K, int C, int KH, int KW, hipStream_t stream);
/* w -> [S*S phases][K*TY*TX][round4(C)] GEMM-B images for gz_conv2d_dgrad */
int gz_conv2d_pack_dgrad(const float* w, float* wpack, int K, int C, int KH, int KW, int S, int P,
                         hipStream_t stream);
/* Re-packing every weight of a network in ONE launch (the packs follow each optimizer step: ~15 launches per G+D pair).
 * gz_conv2d_pack_job fills one host-side job record (gz_conv2d_pack_job_bytes() bytes) for the pair (w, wp) -- the
 * same image gz_conv2d_pack_fwd / _dgrad would write -- and returns the number of workgroups it needs (its first one is
 * `block0` = the sum over the jobs before it); the records are copied to the device as one array and
 * gz_conv2d_pack_multi runs them all. */
size_t gz_conv2d_pack_job_bytes(void);
int gz_conv2d_pack_job(void* job_out, const float* w, float* wp, int is_dgrad, int K, int C, int KH, int KW, int S, int P,
                       int block0);
int gz_conv2d_pack_multi(const void* jobs_dev, int njobs, int total_blocks, hipStream_t stream);
/* Packed images with the job table passed by value and an optional scale 1 / sigma[0] (sigma on the device, may be
 * NULL): spectral normalisation's w = weight_orig / sigma is a new tensor at every call; one launch writes w (what = 2:
 * plain scaled copy) and the forward (what = 0) and dgrad (what = 1) images of every such layer.  table_host:
 * gz_conv2d_pack_table_bytes() zeroed host bytes, at most gz_conv2d_pack_table_max_jobs() jobs. */
int gz_conv2d_pack_table_max_jobs(void);
size_t gz_conv2d_pack_table_bytes(void);
int gz_conv2d_pack_table_add(void* table_host, const float* w, float* wp, const float* sigma, int what, int K, int C, int KH,
                             int KW, int S, int P);
int gz_conv2d_pack_table_launch(void* table_host, hipStream_t stream);

/* y = act(conv2d(x, w) + bias).  Replaces aten::convolution for nn.Conv2d forward
 * (standard_networks.py:20-24,36-43) and the input gradient of nn.ConvTranspose2d. */
int gz_conv2d_fwd(const float* x, const float* wpack, const float* bias, float* y, float* workspace, size_t ws_bytes,
                  int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P, int act,
                  float slope, hipStream_t stream);
/* Split-K scratch of gz_conv2d_fwd / gz_conv2d_dgrad / gz_gemm: launches whose output has too few tiles to fill
 * the 256 CUs (deep 4x4 / 8x8 maps at small batch, nn.Linear heads with K = 8192) cut the reduction instead and
 * sum the partial tiles in a second pass.  0 = not split.  A NULL / too small workspace is not an error: the
 * launch then runs unsplit. */
size_t gz_conv2d_fwd_workspace_bytes(int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P);
size_t gz_conv2d_dgrad_workspace_bytes(int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S,
                                       int P);
size_t gz_gemm_workspace_bytes(int M, int N, int K);

/* x = act(conv_transpose2d(y, w) + bias).  Replaces nn.ConvTranspose2d forward
 * (standard_networks.py:60-73,80-87) and aten::convolution_backward's grad_input for nn.Conv2d. */
int gz_conv2d_dgrad(const float* y, const float* wpack, const float* bias, float* x, float* workspace,
                    size_t ws_bytes, int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P,
                    int act, float slope, hipStream_t stream);

/* dw[K,C,KH,KW] = weight gradient.  Replaces aten::convolution_backward's grad_weight. */
size_t gz_conv2d_wgrad_workspace_bytes(int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW);
/* dbias[K] (may be NULL) = sum over pixels of y, the bias gradient: produced in the same pass only by the
 * launches gz_conv2d_wgrad_fuses_bias reports (3x3 s1 p1 with few channels); otherwise pass NULL and reduce y */
int gz_conv2d_wgrad_fuses_bias(int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P);
int gz_conv2d_wgrad(const float* x, const float* y, float* dw, float* dbias, float* workspace, size_t ws_bytes, int N,
                    int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P, hipStream_t stream);

/* Round 4 -- CU budget.  The convolution plans size their tile / split choices to whole rounds of workgroup slots on
 * `cu_count` CUs (default 256 = the whole MI355X).  Under data parallelism RCCL's channel kernels occupy part of the
 * chip while backward runs (the overlap ddp.GradSync creates); a plan sized for 256 CUs then spills into a second,
 * nearly empty round.  ddp.GradSync calls gz_set_cu_budget(256 - reserved) when world > 1 (clamped to [64, 256];
 * <= 0 restores 256); returns the value in effect.  Process-wide, may be changed between launches; workspace sizes and
 * gz_conv2d_plan follow it. */
int gz_set_cu_budget(int cu_count);
int gz_get_cu_budget(void);

/* Round 4 -- the dispatch as data.  gz_conv2d_plan writes a one-line description of the kernel a launch of op (0 F, 1 Dg,
 * 2 Wg) with this shape takes -- skeleton (igemm / igemm2 / igemm2w / a direct kernel), tile, operand loaders, number of
 * reduction slabs, BatchNorm statistics rows -- into buf; returns the text length or a negative error.  Pure host logic
 * (callable without a GPU); tests/test_dispatch_plan.py pins it for every layer of the BASELINE configurations, so a
 * heuristic edit that moves a layer onto another kernel shows up as a diff.  Describes the launch for 16-byte aligned
 * tensors with the advertised workspace. */
int gz_conv2d_plan(int op, int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P, char* buf,
                   int buflen);

/* Round 4 -- weight gradients straight into the optimizer's (or the gradient exchange's) buffer.
 * gz_conv2d_wgrad_partial: as gz_conv2d_wgrad, but a launch that splits its reduction leaves the slabs in `workspace`
 * and reports them (*nz_out slabs, *stride_out floats apart); *nz_out == 1 means dw is complete.  gz_reduce_multi then
 * sums the slabs of many parameters -- and of several launches per parameter (a discriminator applied to a real and a
 * fake batch) -- in ONE launch, writing (beta 0) or accumulating into (beta 1) each gradient.  Replaces the per-layer
 * slab reductions plus autograd's `grad += new` launches (reference: torch's AccumulateGrad behind every Conv2d /
 * ConvTranspose2d weight of core/models/standard_networks.py:20-24,36-43,60-73).  The table is plain host memory of
 * gz_reduce_multi_table_bytes() bytes, zero-initialised, filled with gz_reduce_multi_add (at most
 * gz_reduce_multi_max_jobs() gradients of gz_reduce_multi_max_sources() launches each; counts and strides multiples
 * of 4 floats, 16-byte aligned pointers). */
int gz_conv2d_wgrad_partial(const float* x, const float* y, float* dw, float* workspace, size_t ws_bytes, int N, int C,
                            int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P, int* nz_out,
                            long long* stride_out, hipStream_t stream);
/* Round 5 -- the first-order backward of `act(conv(x) + bias)` for the critics' first layer (k4 s2 p1, <= 4 image
 * channels, act = ReLU / LeakyReLU; reference core/models/standard_networks.py:62-66: Conv2d(3, features_d, 4, 2, 1) +
 * LeakyReLU(0.2)) in ONE launch: gy is the gradient with respect to the activation's OUTPUT, masked on load with the
 * saved forward output fwd_out; every slab row holds the K * C * 16 weight-gradient partial sums followed, at
 * *bias_offset_out, by the K bias-gradient partial sums.  Always partial: sum the *nz_out rows (*stride_out floats
 * apart) with gz_reduce_multi or hand them to gz_adam_step_from_slabs.  workspace: gz_conv2d_wgrad_workspace_bytes.
 * gz_conv2d_wgrad_act_fuses reports whether the shape takes this path (otherwise: gz_act_bwd + gz_conv2d_wgrad). */
int gz_conv2d_wgrad_act_fuses(int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P, int act);
int gz_conv2d_wgrad_act_partial(const float* x, const float* gy, const float* fwd_out, int act, float slope,
                                float* workspace, size_t ws_bytes, int N, int C, int H, int W, int K, int OH, int OW,
                                int KH, int KW, int S, int P, int* nz_out, long long* stride_out,
                                long long* bias_offset_out, hipStream_t stream);
/* ... and the same layer's INPUT gradient (the generator step through the frozen critic): x = conv_transpose(gy *
 * act'(fwd_out), w) by the direct few-channel kernel with the mask formed on load.  gz_conv2d_dgrad_act_fuses reports
 * whether the shape takes it (otherwise: gz_act_bwd + gz_conv2d_dgrad); 16-byte aligned tensors, the dgrad weight pack. */
int gz_conv2d_dgrad_act_fuses(int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P, int act);
int gz_conv2d_dgrad_act(const float* gy, const float* fwd_out, int act, float slope, const float* wpack, float* x, int N,
                        int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P, hipStream_t stream);
int gz_reduce_multi_max_jobs(void);
int gz_reduce_multi_max_sources(void);
size_t gz_reduce_multi_table_bytes(void);
int gz_reduce_multi_add(void* table_host, float* out, long long count, int beta, const float* slabs, int nz,
                        long long stride);
int gz_reduce_multi(void* table_host, hipStream_t stream);

/* Convolution + BatchNorm statistics in one launch (standard_networks.py:34-44,80-88: conv / transpose_conv ->
 * batch_norm): the epilogue also writes, per output channel and per group of pixels, (sum, sum of squares) of the
 * raw convolution output to stats[rows][channels][2]; gz_batchnorm_finalize turns them into the coefficients and
 * the running buffers -- the separate read of the whole feature map goes away.  *_stats_rows = number of partial
 * rows the launch produces, 0 when it cannot fuse (split-K launch, 3-channel direct kernel): use the plain entry
 * point + gz_batchnorm_stats then.  No bias, no activation (the activation follows the normalisation). */
int gz_conv2d_fwd_stats_rows(int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P);
int gz_conv2d_fwd_stats(const float* x, const float* wpack, float* y, float* stats, int N, int C, int H, int W, int K,
                        int OH, int OW, int KH, int KW, int S, int P, hipStream_t stream);
int gz_conv2d_dgrad_stats_rows(int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P);
/* Round 4: launches that split their reduction carry the statistics too -- the finish pass runs the same epilogue per
 * 32 x 32 output block (one partial row per 32 pixels) -- and need the op's workspace (gz_conv2d_{fwd,dgrad}_workspace_bytes).
 * The entry points without a workspace remain for unsplit plans. */
int gz_conv2d_fwd_stats_ws(const float* x, const float* wpack, float* y, float* stats, float* workspace, size_t ws_bytes,
                           int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P,
                           hipStream_t stream);
int gz_conv2d_dgrad_stats_ws(const float* y, const float* wpack, float* x, float* stats, float* workspace,
                             size_t ws_bytes, int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S,
                             int P, hipStream_t stream);
int gz_conv2d_dgrad_stats(const float* y, const float* wpack, float* x, float* stats, int N, int C, int H, int W, int K,
                          int OH, int OW, int KH, int KW, int S, int P, hipStream_t stream);

/* diagnostic: tile configuration a launch would use (op 0 F, 1 Dg, 2 Wg) -> 0 128x128, 1 128x64, 2 128x32, 3 64x64 */
int gz_conv2d_tile(int op, int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S);

/* c[M,N] = act(op(a) . op(b) + bias[n]); trans_a: a stored [K][M]; trans_b: b stored [N][K].
 * Replaces the 1x1 -> 4x4 ConvTranspose2d of the generator's first block (standard_networks.py:60),
 * its weight gradient, and nn.Linear. */
int gz_gemm(const float* a, const float* b, const float* bias, float* c, float* workspace, size_t ws_bytes, int M,
            int N, int K, int lda, int ldb, int ldc, int trans_a, int trans_b, int act, float slope,
            hipStream_t stream);

/* ---- cubic 3-D convolution family (HoloGAN ConvTranspose3d k3 s2 p1 op1, hologan_generator.py:29-30) ------
 * x [N,C,D,H,W] image side, y [N,K,OD,OH,OW] feature side, w [K,C,KS,KS,KS]; only (KS,S,P) = (3,2,1). */
long long gz_conv3d_pack_fwd_elems(int K, int C, int KS);
long long gz_conv3d_pack_dgrad_elems(int K, int C, int KS, int S);
int gz_conv3d_pack_fwd(const float* w, float* wpack, int K, int C, int KS, hipStream_t stream);
int gz_conv3d_pack_dgrad(const float* w, float* wpack, int K, int C, int KS, int S, int P, hipStream_t stream);
/* y = act(conv3d(x, w) + bias): input gradient of nn.ConvTranspose3d */
int gz_conv3d_fwd(const float* x, const float* wpack, const float* bias, float* y, float* workspace, size_t ws_bytes,
                  int N, int C, int D, int H, int W, int K, int OD, int OH, int OW, int KS, int S, int P, int act,
                  float slope, hipStream_t stream);
/* split-K scratch of the two launchers above / below (0 = not split; NULL workspace = run unsplit) */
size_t gz_conv3d_fwd_workspace_bytes(int N, int C, int K, int OD, int OH, int OW, int KS);
size_t gz_conv3d_dgrad_workspace_bytes(int N, int C, int K, int OD, int OH, int OW, int KS);
/* x = act(conv_transpose3d(y, w) + bias): nn.ConvTranspose3d forward */
int gz_conv3d_dgrad(const float* y, const float* wpack, const float* bias, float* x, float* workspace,
                    size_t ws_bytes, int N, int C, int D, int H, int W, int K, int OD, int OH, int OW, int KS, int S,
                    int P, int act, float slope, hipStream_t stream);
size_t gz_conv3d_wgrad_workspace_bytes(int N, int C, int K, int OD, int OH, int OW, int KS);
int gz_conv3d_wgrad(const float* x, const float* y, float* dw, float* workspace, size_t ws_bytes, int N, int C, int D,
                    int H, int W, int K, int OD, int OH, int OW, int KS, int S, int P, hipStream_t stream);

/* ---- HoloGAN rigid-body resampling (hologan_generator.py:198-321 + the permute/flip/reshape of :130-133) ----
 * vox [N,C,S,S,S]; minv [N,16] row-major inverse transforms; out2d [N, C*S, S, S] with
 * out2d[n][c*S + (S-1-y)][z][x] = trilinear(vox[n][c], minv[n] . (x,y,z,1)).  idx_out (may be NULL):
 * int64 [8][N*S^3], the clamped corner indices in the reference's idx_a..idx_h order. */
int gz_rigid_resample_fwd(const float* vox, const float* minv, float* out2d, long long* idx_out, int N, int C, int S,
                          hipStream_t stream);
/* gvox [N,C,S,S,S] = adjoint of the above in gather form (no atomics, every element written once): per-voxel hit
 * lists in `workspace` (gz_rigid_resample_bwd_workspace_bytes), gradient volumes staged through LDS.  workspace
 * NULL / too small: the same result from the slower direct gather. */
size_t gz_rigid_resample_bwd_workspace_bytes(int N, int S);
int gz_rigid_resample_bwd(const float* gout2d, const float* minv, float* gvox, float* workspace, size_t ws_bytes,
                          int N, int C, int S, hipStream_t stream);

/* ---- evaluation path: InceptionV3 feature extractor of FID / KID (core/callback_inception_metrics.py:183-246,
 * core/submodules/gan_stability/metrics/inception.py) -- forward only --------------------------------------------
 * y = act(conv2d(x, w) + bias) for ANY rectangular kernel, per-axis stride and padding (3x3 s2 p0, 5x5 s1 p2,
 * 1x7 / 7x1, 1x3 / 3x1 ...).  Eval-mode BatchNorm is folded into w and bias by the caller.  wpack from
 * gz_conv2d_pack_fwd_any (tap-major, channels padded to 16). */
long long gz_conv2d_pack_fwd_any_elems(int K, int C, int KH, int KW);
int gz_conv2d_pack_fwd_any(const float* w, float* wpack, int K, int C, int KH, int KW, hipStream_t stream);
size_t gz_conv2d_fwd_any_workspace_bytes(int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int SH,
                                         int SW, int PH, int PW);
int gz_conv2d_fwd_any(const float* x, const float* wpack, const float* bias, float* y, float* workspace,
                      size_t ws_bytes, int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int SH, int SW,
                      int PH, int PW, int act, float slope, hipStream_t stream);
/* The same convolution written into a channel slice of a concatenated tensor (InceptionV3's torch.cat of branch
 * outputs, inception.py:92-94 etc.): y points at the slice's first channel inside image 0, images are
 * y_image_channels * OH * OW floats apart.  Only the LDS-DMA launch takes a destination stride: GZ_ERR_UNSUPPORTED (-2)
 * when the shape runs elsewhere -- the caller then convolves into a temporary and copies. */
int gz_conv2d_fwd_any_into(const float* x, const float* wpack, const float* bias, float* y, int y_image_channels, int N,
                           int C, int H, int W, int K, int OH, int OW, int KH, int KW, int SH, int SW, int PH, int PW,
                           int act, float slope, hipStream_t stream);
/* square-window pooling over `planes` = N*C images; mode 0 max, 1 average with count_include_pad=False (the FID
 * network's patched pools), 2 average over KS*KS.  nn.AdaptiveAvgPool2d(1) = KS = H, S = 1, P = 0, mode 2. */
int gz_pool2d(const float* x, float* y, long long planes, int H, int W, int OH, int OW, int KS, int S, int P, int mode,
              hipStream_t stream);
/* y = F.interpolate(x, (OH, OW), mode='bilinear', align_corners=False) * mul + add */
int gz_resize_bilinear(const float* x, float* y, long long planes, int H, int W, int OH, int OW, float mul, float add,
                       hipStream_t stream);

/* ---- loss heads and the scalar side of spectral normalisation (csrc/gz_loss.hip) -----------------------------
 * Scalars (loss, sigma, the incoming loss gradient) are 1-element DEVICE arrays: nothing synchronises with the host.
 * loss[0] = mean_i BCE-with-logits(x[i], target), target a constant 0 or 1: criterion(logits, ones_like(logits)) /
 * zeros_like of core/lightning_module.py:114-119,126,221-235 */
int gz_bce_logits_mean(const float* x, float* loss, int n, float target, hipStream_t stream);
/* dx[i] = (sigmoid(x[i]) - target) * gloss[0] / n */
int gz_bce_logits_mean_bwd(const float* x, const float* gloss, float* dx, int n, float target, hipStream_t stream);
/* Round 4: the loss head of a discriminator step on the stacked batch [real; fake] (2 * n_each logits) in one launch.
 * mode 0: (mean BCE(x[:n], t0) + mean BCE(x[n:], t1)) / 2 (core/lightning_module.py:114-120);
 * mode 1: t0 * mean(x[:n]) + t1 * mean(x[n:]) (the WGAN critic loss with t0 = -1, t1 = +1, :168). */
int gz_pair_loss(const float* x, float* loss, int n_each, float t0, float t1, int mode, hipStream_t stream);
int gz_pair_loss_bwd(const float* x, const float* gloss, float* dx, int n_each, float t0, float t1, int mode,
                     hipStream_t stream);
/* loss[0] = mean((a - b)^2): HoloGAN's q_loss (:226,234);  da[i] = 2 (a[i] - b[i]) gloss[0] / n */
/* WGAN-GP penalty tail, core/utils/utils.py:55-57: out[0] = mean_n (sqrt(sumsq[n]) - 1)^2 and its gradient
 * dsumsq[n] = gout[0] (sqrt(s) - 1) / (sqrt(s) N), 0 where s == 0 (torch.norm's subgradient at the origin) */
int gz_gp_penalty(const float* sumsq, float* out, int n, hipStream_t stream);
int gz_gp_penalty_bwd(const float* sumsq, const float* gout, float* dsumsq, int n, hipStream_t stream);
int gz_mse_mean(const float* a, const float* b, float* loss, int n, hipStream_t stream);
int gz_mse_mean_bwd(const float* a, const float* b, const float* gloss, float* da, int n, hipStream_t stream);
/* torch.nn.utils.spectral_norm (core/models/hologan_discriminator.py:15): out = x / max(||x||, eps) (in place
 * allowed; the same values also to out2; out, out2, norm_out may be NULL), norm_out[0] = ||x||; out[0] = <a, b>; out = x / sigma[0]; and the gradient of
 * w = W / sigma(W), sigma = u^T W v:  out[r][l] = (g[r][l] - (sum_r rowdots[r]) u[r] v[l]) / sigma[0] with
 * rowdots[r] = <g[r], w[r]> (gz_rowdot). */
int gz_vec_normalize(const float* x, float* out, float* out2, float* norm_out, int n, float eps, hipStream_t stream);
/* gz_vec_normalize that also returns dot_out[0] = <x / max(||x||, eps), x>: with x = W v this is the power iteration's
 * u update and sigma = u^T W v in one launch */
int gz_vec_normalize_dot(const float* x, float* out, float* out2, float* dot_out, int n, float eps,
                         hipStream_t stream);
int gz_vec_dot(const float* a, const float* b, float* out, int n, hipStream_t stream);
/* The power iteration of SEVERAL spectral-normalised weights in four launches (every spectral-norm layer of one
 * discriminator call, core/models/hologan_discriminator.py:15,32): per job v <- normalise(W^T u), u <- normalise(W v),
 * sigma = u^T W v; u, v are the module's buffers (updated in place), us / vs / sigma the copies the autograd node keeps.
 * table_host: gz_sn_table_bytes() zeroed host bytes filled by gz_sn_add (at most gz_sn_max_jobs()); workspace:
 * gz_sn_workspace_floats(R, L) floats per job.  L % 4 == 0; W, v, vs, workspace 16-byte aligned. */
int gz_sn_max_jobs(void);
size_t gz_sn_table_bytes(void);
long long gz_sn_workspace_floats(int R, int L);
int gz_sn_add(void* table_host, const float* W, float* u, float* v, float* us, float* vs, float* sigma, float* workspace,
              int R, int L);
int gz_sn_power_iteration(void* table_host, float eps, hipStream_t stream);
/* A spectral-normalised convolution followed by InstanceNorm (no affine) without the per-call weight copy
 * (core/models/hologan_discriminator.py:28-38): IN_eps(conv(x, W / sigma) + b) = IN_{eps sigma^2}(conv(x, W)), so the
 * convolution runs on weight_orig's own packed images and the InstanceNorm takes one eps per sample.
 *   gz_rownorm_act_fwd_sigma (below): the InstanceNorm + activation with eps * sigma[g]^2 for the samples of group g
 *     (groups: discriminator calls stacked along n, sigma[groups] on the device)
 *   gz_sn_sigma_term: the gradient's sigma term, term[R][L] = sum_g coefs[g] u[g] v[g]^T with coefs[g] = dL/dsigma_g =
 *     -eps sigma_g sum_{rows of g} rstd^2 S2 from the backward's row sums (gz_rownorm_act_bwd_rows: [rows] pairs
 *     (sum dz, sum dz xh)); rstd = the forward's coef + 3 * rows.  Equal to torch's -(sum g w) u v^T / sigma. */
int gz_sn_sigma_coef_floats(int groups);      /* size of `coefs` (scratch) */
int gz_sn_sigma_term(const float* rowsums, const float* rstd, const float* sigma, const float* u, const float* v,
                     float* coefs, float* term, int rows, int groups, int R, int L, float eps, hipStream_t stream);
int gz_div_scalar(const float* x, const float* sigma, float* out, long long count, hipStream_t stream);
int gz_spectral_norm_bwd(const float* g, const float* rowdots, const float* u, const float* v, const float* sigma,
                         float* out, int R, int L, hipStream_t stream);

/* ---- normalisation + activation -------------------------------------------------------------
 * A tensor [N, C, inner] is N*C rows of `inner` contiguous floats (inner % 4 == 0).
 * coef layout: 4 arrays of `ncoef` floats (scale, shift, mean, rstd); ncoef = C for per-channel
 * statistics (BatchNorm), N*C for per-row statistics (InstanceNorm / AdaIN).
 * workspace: gz_norm_workspace_bytes(N, C) bytes.
 */
size_t gz_norm_workspace_bytes(int N, int C);
int gz_norm_coef_elems(int N, int C, int per_channel);

/* training-mode BatchNorm2d statistics (biased var for normalisation; running stats updated with the
 * unbiased var and `momentum`; *num_batches_tracked += 1).  Replaces the statistics half of
 * aten::native_batch_norm for nn.BatchNorm2d (standard_networks.py:44,87). */
/* the same from per-tile partials [rows][C][2] written by gz_conv2d_*_stats; count = elements per channel (N*H*W) */
int gz_batchnorm_finalize(const float* partials, int rows, long long count, const float* gamma, const float* beta,
                          float* coef, float* running_mean, float* running_var, long long* num_batches_tracked, int C,
                          float eps, float momentum, hipStream_t stream);
int gz_batchnorm_stats(const float* x, const float* gamma, const float* beta, float* coef, float* running_mean,
                       float* running_var, long long* num_batches_tracked, void* workspace, int N, int C,
                       int inner, float eps, float momentum, hipStream_t stream);
/* out[c] = sum over n and the inner dimension of x[n][c][:]: the bias gradient of a convolution (aten's
 * grad_bias); workspace as for the statistics passes (gz_norm_workspace_bytes) */
int gz_channel_sum(const float* x, float* out, void* workspace, int N, int C, int inner, hipStream_t stream);
/* eval-mode coefficients from the running statistics */
int gz_batchnorm_eval_coef(const float* gamma, const float* beta, const float* running_mean,
                           const float* running_var, float* coef, int C, float eps, hipStream_t stream);
/* per-row statistics: nn.InstanceNorm2d (standard_networks.py:46; gamma/beta per channel, biased var)
 * or AdaIN (hologan_generator.py:333-345; gamma/beta per row, unbiased var). */
int gz_rownorm_stats(const float* x, const float* gamma, const float* beta, float* coef, void* workspace, int N,
                     int C, int inner, float eps, int affine_per_row, int unbiased, hipStream_t stream);
/* gz_rownorm_stats + gz_norm_act_fwd(per_channel = 0) in one launch (the row is re-read from cache): coef as above,
 * out = act(x * scale + shift).  affine_per_row: 0 = gamma/beta per channel, 1 = per row ([N][C] arrays, AdaIN),
 * 2 = per row from one packed [N][2C] array (gamma = its base, beta = base + C: HoloGAN's ZMapping output,
 * hologan_generator.py:15-18). */
int gz_rownorm_act_fwd(const float* x, const float* gamma, const float* beta, float* coef, float* out, int N, int C,
                       int inner, float eps, int affine_per_row, int unbiased, int act, float slope,
                       hipStream_t stream);
/* AdaIN + activation of a constant shared by all samples (HoloGAN's learned 4^3 volume, hologan_generator.py:141
 * `self.x.repeat(batch, ...)`): x is [C][inner], sb the packed [N][2C] scale | shift, out / coef as for N*C rows.
 * Backward: dx [C][inner] (summed over the samples), dsb [N][2C]; inner <= 1024. */
int gz_adain_const_fwd(const float* x, const float* sb, float* coef, float* out, int N, int C, int inner, float eps,
                       int act, float slope, hipStream_t stream);
int gz_adain_const_bwd(const float* gout, const float* x, const float* coef, float* dx, float* dsb, int N, int C,
                       int inner, int act, float slope, hipStream_t stream);
/* out = act(x * scale + shift) */
int gz_norm_act_fwd(const float* x, const float* coef, float* out, int N, int C, int inner, int per_channel,
                    int act, float slope, hipStream_t stream);
/* first backward of norm+act: dx (may be NULL), dgamma, dbeta (may be NULL); kbuf: 2*ncoef floats scratch */
/* InstanceNorm (no affine) + activation with eps * sigma[n / (N / groups)]^2 per sample, and its first-order backward
 * that also leaves the row sums (sum dz, sum dz xh) in rowsums[N*C] pairs: the two halves of functional.sn_conv_in_act */
int gz_rownorm_act_fwd_sigma(const float* x, const float* sigma, int groups, float* coef, float* out, int N, int C,
                             int inner, float eps, int act, float slope, hipStream_t stream);
int gz_rownorm_act_bwd_rows(const float* gout, const float* x, const float* coef, float* dx, float* rowsums, int N, int C,
                            int inner, int act, float slope, hipStream_t stream);
int gz_norm_act_bwd(const float* gout, const float* x, const float* coef, float* dx, float* dgamma, float* dbeta,
                    void* workspace, float* kbuf, int N, int C, int inner, int per_channel, int affine_per_row,
                    int unbiased, int act, float slope, hipStream_t stream);
/* Round 4 -- BatchNorm with statistics GROUPS.  A discriminator step of the reference applies D to the real batch and
 * to the fake batch in two calls (core/lightning_module.py:112-119), each call normalising with its own batch
 * statistics and updating the running buffers.  Stacking the two batches along n and running ONE pass with groups = 2
 * is the same arithmetic -- group g = samples [g N/groups, (g+1) N/groups) has its own (scale, shift, mean, rstd) at
 * coef index g*C + c (coef holds 4 * groups * C floats), the running buffers are updated group after group,
 * num_batches_tracked += groups, dgamma / dbeta sum over the groups -- with half the launches and twice the rows per
 * launch.  `partials` of gz_batchnorm_finalize_g: rows of a convolution's fused statistics, the first rows/groups
 * belonging to group 0 etc. (the caller checks that no row straddles two groups); `count` = elements per group and
 * channel.  gz_batchnorm_act_bwd_g: accumulate != 0 adds the affine gradients to dgamma / dbeta (gradient sinks). */
int gz_batchnorm_stats_g(const float* x, const float* gamma, const float* beta, float* coef, float* running_mean,
                         float* running_var, long long* num_batches_tracked, void* workspace, int N, int C,
                         int inner, float eps, float momentum, int groups, hipStream_t stream);
int gz_batchnorm_finalize_g(const float* partials, int rows, long long count, const float* gamma, const float* beta,
                            float* coef, float* running_mean, float* running_var, long long* num_batches_tracked,
                            int C, float eps, float momentum, int groups, hipStream_t stream);
/* round 5: training-mode BatchNorm + activation forward with the finalize folded into the apply launch (one launch
 * when the producing convolution supplied `partials`; partials == NULL: row sums of x into `workspace` first) */
int gz_batchnorm_act_fwd_fused(const float* x, const float* partials, int rows, long long count, const float* gamma,
                               const float* beta, float* coef, float* running_mean, float* running_var,
                               long long* num_batches_tracked, void* workspace, float* out, int N, int C, int inner,
                               float eps, float momentum, int groups, int act, float slope, hipStream_t stream);
int gz_norm_act_fwd_g(const float* x, const float* coef, float* out, int N, int C, int inner, int per_channel,
                      int groups, int act, float slope, hipStream_t stream);
int gz_rownorm_act_bwd_acc(const float* gout, const float* x, const float* coef, float* dx, float* dgamma, float* dbeta,
                           void* workspace, float* kbuf, int N, int C, int inner, int act, float slope, int accumulate,
                           hipStream_t stream);       /* InstanceNorm(affine) first-order backward, same option */
int gz_batchnorm_act_bwd_g(const float* gout, const float* x, const float* coef, float* dx, float* dgamma, float* dbeta,
                           void* workspace, float* kbuf, int N, int C, int inner, int act, float slope, int groups,
                           int accumulate, hipStream_t stream);

/* backward of gz_norm_act_bwd's dx (per-row statistics, per-channel affine): given v = dL/d(dx) returns
 * gg_out = dL/d(gout), gx = dL/dx, ggamma = dL/dgamma (any may be NULL).  This is the InstanceNorm leg of
 * the gradient-penalty double backward (core/utils/utils.py:48-54 with create_graph=True). */
int gz_rownorm_act_bwd2(const float* gout, const float* v, const float* x, const float* coef, float* gg_out,
                        float* gx, float* ggamma, void* workspace, int N, int C, int inner, int act, float slope,
                        hipStream_t stream);
/* dx = g * act'(.) with the derivative expressed through the activation OUTPUT (LeakyReLU, ReLU, tanh) */
int gz_act_bwd(const float* g, const float* out, float* dx, long long count, int act, float slope,
               hipStream_t stream);
/* res = v * g * (-2 out): the derivative of gz_act_bwd's tanh result g * (1 - out^2) with respect to `out`, contracted
 * with v (second-order backward through a fused tanh epilogue) */
int gz_tanh_bwd2(const float* v, const float* g, const float* out, float* res, long long count, hipStream_t stream);

/* ---- row helpers (last discriminator layer, gradient-penalty tail, WGAN clip) -----------------
 * matrices are [R][L] row-major with L % 4 == 0. */
/* y[r] = sum_l a[r][l] * b[r][l]   (b_broadcast: b is a single row [L]) */
int gz_rowdot(const float* a, const float* b, float* y, int R, int L, int b_broadcast, hipStream_t stream);
/* out[r][:] = s[r] * x[r][:] (+ t[r] * x2[r][:], t = 1 - s when one_minus_s else s2); x_broadcast: x is [L] */
int gz_rowscale(const float* x, const float* s, const float* x2, const float* s2, float* out, int R, int L,
                int x_broadcast, int one_minus_s, hipStream_t stream);
/* ---- several small Linear layers over ONE input in one launch ------------------------------------------------
 * HoloGAN's five ZMapping layers, Linear(z_dim -> 2C) + ReLU each on the same z (core/models/hologan_generator.py:7-19,
 * called at :33/:57/:141).  table_host: gz_linear_multi_table_bytes() zeroed bytes of host memory filled through
 * gz_linear_multi_add (at most gz_linear_multi_max_jobs() layers; copied into the kernel argument).
 *   fwd: out_j[N][J_j] = act(x[N][K] . weight_j[J_j][K]^T + bias_j)          (g, dw, db unused: pass NULL)
 *   bwd: dw_j = (g_j * act'(out_j))^T . x,  db_j = its column sums (db NULL: skipped); weight/bias unused.
 * Any N, K, J (scalar loads); rows are summed in a fixed order. */
int gz_linear_multi_max_jobs(void);
size_t gz_linear_multi_table_bytes(void);
int gz_linear_multi_add(void* table_host, const float* weight, const float* bias, float* out, const float* g, float* dw,
                        float* db, int J);
int gz_linear_multi_fwd(void* table_host, const float* x, int N, int K, int act, float slope, hipStream_t stream);
int gz_linear_multi_bwd(void* table_host, const float* x, int N, int K, int act, float slope, hipStream_t stream);
/* out[l] = sum_r x[r][l]  (nn.Linear's bias gradient: core/models/hologan_discriminator.py:41-50, hologan_generator.py:11) */
int gz_colsum(const float* x, float* out, int R, int L, hipStream_t stream);
/* out[l] = sum_r g[r] * x[r][l] */
size_t gz_coldot_workspace_bytes(int R, int L);
int gz_coldot(const float* g, const float* x, float* out, float* workspace, size_t ws_bytes, int R, int L,
              hipStream_t stream);
/* gz_coldot without its slab sum (round 5): *nz_out row slices of L partial sums each are left in `workspace` (L floats
 * apart) for gz_reduce_multi / gz_adam_step_from_slabs; *nz_out == 1 means `out` is complete.  The critics' last layer
 * (Conv2d(8 f_d, 1, 4, 2, 0), reference standard_networks.py:27-30): its weight gradient joins the sinks unreduced. */
int gz_coldot_partial(const float* g, const float* x, float* out, float* workspace, size_t ws_bytes, int R, int L,
                      int* nz_out, hipStream_t stream);
/* dst[i] = src[i] for `words` 4-byte words; src may be PINNED HOST memory (device-mapped): the staging of the per-step
 * host draws -- `noise = distn.sample(...).to(device)`, core/lightning_module.py:107-108, core/utils/utils.py:41 */
int gz_copy_words(const void* src, void* dst, long long words, hipStream_t stream);
/* p = clamp(p, lo, hi) in place (core/lightning_module.py:160-162) */
int gz_clamp_(float* p, long long count, float lo, float hi, hipStream_t stream);

/* ---- R1-regularised ResNet path ------------------------------------------------------------
 * HBM-bound linear maps of reference core/submodules/gan_stability/models/resnet.py; every fwd/bwd pair is also
 * its own double-backward pair (the adjoint of the adjoint is the forward). count % 4 == 0. */
/* y = act(x): the pre-activation `actvn` (resnet.py:131-133) */
int gz_act_fwd(const float* x, float* y, long long count, int act, float slope, hipStream_t stream);
/* out = alpha*a + beta*b (b may be NULL); act_out (may be NULL) = act(out) written in the same pass:
 * the residual tail `x_s + 0.1*dx` (resnet.py:121-122) together with the next block's `actvn(x)` */
int gz_axpby(const float* a, float alpha, const float* b, float beta, float* out, float* act_out, long long count,
             int act, float slope, hipStream_t stream);
/* nn.AvgPool2d(3, stride=2, padding=1) (count_include_pad) over `planes` = N*C maps (resnet.py:72) and its
 * adjoint; OH = (H-1)/2 + 1 */
int gz_avgpool3s2_fwd(const float* x, float* y, long long planes, int H, int W, int OH, int OW, hipStream_t stream);
int gz_avgpool3s2_bwd(const float* gy, float* gx, long long planes, int H, int W, int OH, int OW,
                      hipStream_t stream);
/* nn.Upsample(scale_factor=2) (nearest, resnet.py:31) [planes,H,W] -> [planes,2H,2W] and its adjoint */
int gz_upsample2_fwd(const float* x, float* y, long long planes, int H, int W, hipStream_t stream);
int gz_upsample2_bwd(const float* gy, float* gx, long long planes, int H, int W, hipStream_t stream);

/* ---- input step ---------------------------------------------------------------------------------
 * decoded uint8 images [N,H,W,C] -> float [N,C,H,W] = (x / 255 - mean) / std: ToTensor() + Normalize(mean, std) of
 * core/lightning_module.py:42-47 on the device */
int gz_u8hwc_to_nchw(const unsigned char* in, float* out, int N, int H, int W, int C, float mean, float std,
                     hipStream_t stream);

/* ---- fused multi-tensor optimizer steps (the `optimiser` nodes of conf/expt/<name>.yaml) -------------------------
 * `count` <= GZ_OPT_MAX_TENSORS tensors per call (host arrays of device pointers and element counts);
 * grads are multiplied by grad_scale first (1/world for data-parallel means).  Formulas are torch.optim's
 * single-tensor ones (no weight decay, no amsgrad / momentum / centered).
 * zero_grads != 0: every gradient element is overwritten with 0 after it has been read -- the zero_grad(set_to_none=
 * False) of the reference's harness in the same pass (the flat data-parallel exchange buffer must read zero again). */
#define GZ_OPT_MAX_TENSORS 24
int gz_adam_step(int count, float* const* params, float* const* grads, float* const* exp_avg,
                 float* const* exp_avg_sq, const long long* numel, float lr, float beta1, float beta2, float eps,
                 int step, float grad_scale, int zero_grads, hipStream_t stream);
/* Adam reading UNREDUCED weight-gradient slabs (gz_conv2d_wgrad_partial) instead of a gradient tensor: the slab sum of
 * gz_reduce_multi and the update in one launch (harness.Trainer outside data parallelism).  A zero-initialised host
 * table of gz_adam_src_table_bytes() bytes takes <= gz_adam_src_max_tensors() parameters with <= 4 sources each
 * (gz_adam_src_add per (parameter, source)); the gradient has the bits gz_reduce_multi would have written. */
int gz_adam_src_max_tensors(void);
size_t gz_adam_src_table_bytes(void);
int gz_adam_src_add(void* table_host, float* param, float* exp_avg, float* exp_avg_sq, long long numel,
                    const float* slabs, int nz, long long stride);
int gz_adam_step_from_slabs(void* table_host, float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                            hipStream_t stream);
/* graph-capturable Adam: tick = device float[3] {step, 1 - beta1^step, sqrt(1 - beta2^step)}; gz_adam_tick advances
 * it by one step, gz_adam_step_dev reads the corrections from it (same update as gz_adam_step) */
int gz_adam_tick(float* tick, float beta1, float beta2, hipStream_t stream);
int gz_adam_step_dev(int count, float* const* params, float* const* grads, float* const* exp_avg,
                     float* const* exp_avg_sq, const long long* numel, float lr, float beta1, float beta2, float eps,
                     const float* tick, float grad_scale, int zero_grads, hipStream_t stream);
int gz_rmsprop_step(int count, float* const* params, float* const* grads, float* const* square_avg,
                    const long long* numel, float lr, float alpha, float eps, float grad_scale, int zero_grads,
                    hipStream_t stream);

/* text of the last HIP error seen by a launcher on the calling thread ("" if none) */
const char* gz_last_error(void);

/* library identification: returns the gfx target string the kernels were compiled for */
const char* gz_build_info(void);
/* content digest (sha256 prefix, hex) of the kernel sources, headers and flags this library was built from;
 * lightning_gan_zoo_amd/_lib.py refuses a library whose digest is not the digest of the tree it runs from */
const char* gz_source_digest(void);

#ifdef __cplusplus
}
#endif
#endif /* GZ_OPS_H */
