/* libsrganst.so - C ABI of the MI355X (gfx950) SRGAN-ST training hot path.
 *
 * The reference (SebastianBitsch/SRGAN-ST) has no FFI of its own: its boundary is the Python
 * module surface.  These entry points sit underneath that surface; each one cites
 * the reference code it replaces.  Conventions:
 *   - every function returns int: 0 = ok, <0 = error (message: sst_last_error(), thread-local);
 *   - all pointers are DEVICE pointers to fp32 unless stated; the caller owns every buffer
 *     (inputs, outputs, saved-for-backward, workspace) - the library never allocates;
 *   - launches are asynchronous on `stream` (a hipStream_t passed as void*), re-entrant,
 *     graph-capturable (no sync, no malloc inside);
 *   - `counter` words must be zero on first use; kernels leave them zero again.
 */
#ifndef SRGANST_H
#define SRGANST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* sst_last_error(void);
int sst_version(void);
const char* sst_arch(void);
/* forget the HIP runtime's sticky last error (left behind by e.g. a failed stream capture) before going on in eager mode;
 * returns the pending code (0 = none) */
int sst_clear_error(void);
/* dev switches (SST_* environment variables) are read once, at their first use; this forgets them (tests that toggle a switch) */
int sst_reload_env(void);

/* ---- structure-tensor loss: loss.py:380-413 + utils.py:194-280 (sigma=.5, rho=2 default) -------
 * sr, gt, dsr: NCHW [B,3,H,W].  gS: saved [B,3,H,W].  partials: sst_st_loss_workspace() floats.
 * fwd writes loss[0] = mean_b mean_px d  and gS = d(sum_px d)/d(Jxx,Jyy,Jxy)(sr).
 * bwd: dsr (+)= scale_host * (scale_dev ? *scale_dev : 1) * d loss / d sr.                        */
int sst_st_loss_workspace(int B, int H, int W, int64_t* partial_floats);
int sst_st_loss_fwd(const float* sr, const float* gt, float* loss, float* gS, float* partials,
                    unsigned* counter, int B, int H, int W, float sigma, float rho, int normalize,
                    void* stream);
int sst_st_loss_bwd(const float* sr, const float* gS, float* dsr, const float* scale_dev,
                    float scale_host, int accumulate, int B, int H, int W, float sigma, float rho,
                    void* stream);
/* The same two launches with the pixel criterion of the step riding along (reference train.py:129-140 evaluates "Pixel" =
 * MSE / L1, config.py:88-90, and "ST" on the same sr / gt): pix_mode 0 = MSE, 1 = L1.
 * fwd: also pix_loss[0] = mean over all B*3*H*W elements (pix_partials: as many floats as `partials`).
 * bwd: dsr (+)= (scale_dev ? *scale_dev : 1) * (scale_host * d ST-loss / d sr + pix_weight * d pixel criterion / d sr).   */
int sst_st_pixel_loss_fwd(const float* sr, const float* gt, float* loss, float* gS, float* partials, unsigned* counter,
                          float* pix_loss, float* pix_partials, int pix_mode, int B, int H, int W, float sigma, float rho,
                          int normalize, void* stream);
int sst_st_pixel_loss_bwd(const float* sr, const float* gt, const float* gS, float* dsr, const float* scale_dev, float scale_host,
                          float pix_weight, int pix_mode, int accumulate, int B, int H, int W, float sigma, float rho,
                          void* stream);

/* ---- convolution (fp32 MFMA implicit GEMM, NHWC) --------------------------------------------------
 * Replaces the cuDNN/oneDNN kernels behind nn.Conv2d fwd/bwd of model.py:32-56,101,113,127,159,173,176.
 * Weights are consumed in a fragment-major packed layout produced by sst_conv_pack from the reference
 * layout [Cout][Cin][k][k]:  mode 0 = forward, mode 1 = data-gradient of a stride-1 conv (transposed,
 * rotated 180 deg; then outputs = Cin, inputs = Cout).
 * in_act: 0 none, 1 slope activation (PReLU scalar / LeakyReLU / ReLU=slope 0) applied to
 *         x*in_scale+in_shift while staging (the producer's BatchNorm-apply + activation, fused).
 * out_mode: 0 NHWC, 1 PixelShuffle(2) store (model.py:160), 2 NCHW + clamp(0,1) with pre-clamp copy
 *           (model.py:148-150), 3 inverse pixel-shuffle store.
 * stats: per-tile BatchNorm partials [sst_conv_stat_tiles][2][Cout] (sum, centred M2), stats_cnt [tiles].
 * The packed buffer of a 3x3 conv with 64 inputs and Cout % 16 == 0 carries a second copy of the weights in the layout
 * of the band kernel (csrc/conv_band.hip), which sst_conv_fwd / sst_conv_dgrad_* use for stride-1 NHWC stores when the
 * image rows group into bands of 144 or 48 pixels (the SRResNet trunk shape, model.py:173,176,113). */
int64_t sst_conv_packed_floats(int Cout, int Cin, int ksize);
int sst_conv_pack(const float* w, float* wp, int Cout, int Cin, int ksize, int mode, void* stream);
/* one launch for many tensors: jobs = device array of {const float* w; float* wp; int Cout, Cin, KK, mode;
 * long long total, block_begin;} (48 bytes each; each workgroup packs 1024 floats).  mode 0 / 1: sst_conv_pack forward /
 * data-gradient layout; 2 / 3: sst_conv9_c3_pack mode 0 / 1; 4: sst_conv9_to3_pack (total = the matching *_packed_floats) */
int sst_conv_pack_multi(const void* jobs, int njobs, int total_blocks, void* stream);
int sst_conv_mtiles(int B, int Ho, int Wo);
/* first dimension of stats / stats_cnt / epi_partial written by the NHWC-store conv of this shape (input H x W):
 * one tile per band when the band kernel takes the shape, else sst_conv_mtiles of the output size */
int sst_conv_stat_tiles(int B, int H, int W, int Cin, int Cout, int ksize, int stride);
/* kernel the conv / weight-gradient entries dispatch to for a shape (rocprofv3 spelling, no argument list): labels for
 * bench.py's roofline rows, to be matched against profiles/ */
const char* sst_conv_kernel_name(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int out_mode, int fused_in);
const char* sst_conv_wgrad_kernel_name(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int njobs);
long sst_debug_big_tile_launches(void);   /* test hook: launches of the 64x64-tile conv kernel so far */
long sst_debug_band_launches(void);       /* test hook: launches of the band conv kernel so far */
long sst_debug_wgrad_band_launches(void); /* test hook: launches of the all-taps weight-gradient kernel so far */
/* measurement hook (tools/mfma_peak.py): `blocks` workgroups x 4 waves x iters x 4 v_mfma_f32_32x32x2_f32, no memory traffic */
int sst_debug_mfma_peak(float* out, int blocks, int iters, void* stream);
/* measurement hook (tools/bf16x3_probe.py): the split-operand bf16 MFMA form proposed in DESIGN.md section 8 - mode 0: one 32 x 32 x K
 * product by the fp32 MFMA (C32) and by six bf16 MFMAs per 16 k (C3) for an accuracy comparison on the host; mode 1: rate of the
 * six-MFMA group in a register-only loop (C32 = scratch, K = iterations, blocks workgroups).  Not used by the product path. */
int sst_debug_bf16x3(const float* A, const float* B, float* C32, float* C3, int K, int mode, int blocks, void* stream);
/* measurement hook (tools/stamp_step.py): out[slot] = the device's 100 MHz wall clock at the point of the stream / captured graph */
int sst_debug_stamp(unsigned long long* out, int slot, void* stream);
int sst_conv_fwd(const float* x, const float* wp, float* y, float* y_pre, const float* bias,
                 const float* in_scale, const float* in_shift, const float* in_slope,
                 float in_slope_const, int in_act, const float* residual, float* stats,
                 float* stats_cnt, int out_mode, int B, int H, int W, int Cin, int Cout, int ksize,
                 int stride, void* stream);
/* ---- persistent, software-pipelined form of the 3x3 conv (csrc/conv_pipe.hip) for the discriminator's layers
 * (model.py:30-59) and their stride-1 data-gradients: Cin % 64 == 0, Cout % 32 == 0, NHWC store, stride 1 / 2 (even H, W),
 * at least 256 tiles of 32 px x 32 ch.  The batch is tiled as one tall image (no MFMA lanes on padding pixels of the
 * 12 x 12 / 6 x 6 layers), the next input patch and the weight fragments are in flight while the MFMAs of the current
 * stage run, layers with few tiles split K over workgroups (partial slabs in `ws`, summed in fixed order).
 * sst_conv_pipe_supported: the tile width (8 / 4 / 2) when the shape is taken, else 0 (the 64 -> 64 trunk shape of the band kernel is never taken).  stats / stats_cnt / epi_partial have sst_conv_pipe_stat_tiles rows
 * (a different tiling than sst_conv_stat_tiles); ws = sst_conv_pipe_ws_floats floats (0: not needed).
 * Arguments as sst_conv_fwd (forward statistics) / sst_conv_dgrad_bwdstats (epi_*: backward partials). */
int sst_conv_pipe_supported(int B, int H, int W, int Cin, int Cout, int ksize, int stride);
int sst_conv_pipe_stat_tiles(int B, int H, int W, int Cin, int Cout, int ksize, int stride);
int64_t sst_conv_pipe_ws_floats(int B, int H, int W, int Cin, int Cout, int ksize, int stride);
int sst_conv_pipe_fwd(const float* x, const float* wp, float* y, const float* bias, const float* in_scale,
                      const float* in_shift, const float* in_slope, float in_slope_const, int in_act, float* stats,
                      float* stats_cnt, const float* epi_y, const float* epi_scale, const float* epi_shift,
                      const float* epi_slope, float epi_slope_const, int epi_act, float* epi_partial, float* ws, int B,
                      int H, int W, int Cin, int Cout, int ksize, int stride, void* stream);
/* ... with COEFFICIENT GROUPS: several passes of the network over different inputs with the same weights - the discriminator step's
 * D(gt) and D(sr.detach()), train.py:155-158 - run as ONE batch of B images in which every grp_images consecutive images form a
 * pass with its own train-mode BatchNorm statistics.  The per-channel arrays in_scale / in_shift / epi_scale / epi_shift are then
 * [B / grp_images][channels] (row = pass); stats / epi_partial tiles never straddle a pass (sst_conv_pipe_groups_ok: the pass
 * boundary falls on a tile boundary of the tall image), so the tile rows of pass p are the p-th of B / grp_images equal consecutive
 * ranges.  grp_images = 0 (or B): one row, the call above. */
int sst_conv_pipe_groups_ok(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int grp_images);
int sst_conv_pipe_fwd_grp(const float* x, const float* wp, float* y, const float* bias, const float* in_scale,
                          const float* in_shift, const float* in_slope, float in_slope_const, int in_act, float* stats,
                          float* stats_cnt, const float* epi_y, const float* epi_scale, const float* epi_shift,
                          const float* epi_slope, float epi_slope_const, int epi_act, float* epi_partial, float* ws, int B,
                          int H, int W, int Cin, int Cout, int ksize, int stride, int grp_images, void* stream);
/* "N-split" form of the pipelined kernel (csrc/conv_nsplit.hip) for Cout % 128 == 0 and at least 1024 (tile, 128-channel group) units
 * (512 with out_mode = 1, the PixelShuffle(2) store of sst_conv_fwd - model.py:160 - which the K-split kernel does not have; no
 * statistics / partials with it):
 * the 4 waves of a workgroup share one staged patch and each computes the full K for its own 32 output channels - no K-partial
 * exchange, a quarter of the patch traffic per MFMA.  Same arguments, packed weights and statistics tiling as sst_conv_pipe_fwd_grp
 * (no split-K workspace); sst_conv_ns_supported: tile width when the shape is taken, else 0. */
int sst_conv_ns_supported(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int out_mode);
int sst_conv_ns_fwd(const float* x, const float* wp, float* y, const float* bias, const float* in_scale,
                    const float* in_shift, const float* in_slope, float in_slope_const, int in_act, float* stats,
                    float* stats_cnt, const float* epi_y, const float* epi_scale, const float* epi_shift,
                    const float* epi_slope, float epi_slope_const, int epi_act, float* epi_partial, int B, int H, int W,
                    int Cin, int Cout, int ksize, int stride, int out_mode, int grp_images, void* stream);
/* stride-2 data-gradient (sst_conv_s2_dgrad below) on the pipelined kernel: the four parity classes of a block of class pixels in
 * one unit (they share the dY patch), the 9 (class, tap) pairs in the place of the 9 taps, one accumulator per class.  Even H, W;
 * Cout % 64 == 0, Cin % 32 == 0.  wp = the buffer of sst_conv_s2_dgrad_pack; ws = sst_conv_s2_dgrad_pipe_ws_floats floats (0: none).
 * sst_conv_s2_dgrad_pipe_supported: tile width when the shape is taken, else 0. */
int sst_conv_s2_dgrad_pipe_supported(int B, int H, int W, int Cin, int Cout);
int64_t sst_conv_s2_dgrad_pipe_ws_floats(int B, int H, int W, int Cin, int Cout);
int sst_conv_s2_dgrad_pipe(const float* dy, const float* wp, float* dx, float* ws, int B, int H, int W, int Cin, int Cout,
                           void* stream);
/* ... with the BatchNorm / activation backward partials of dx against epi_y (the saved output of the layer below, [B,H,W,Cin]) written by
 * the epilogue: epi_partial [sst_conv_s2_dgrad_pipe_stat_tiles()][3][Cin], the layout sst_bwd_finalize consumes (null: none). */
int sst_conv_s2_dgrad_pipe_stat_tiles(int B, int H, int W, int Cin, int Cout);
int sst_conv_s2_dgrad_pipe_bwdstats(const float* dy, const float* wp, float* dx, float* ws, const float* epi_y,
                                    const float* epi_scale, const float* epi_shift, const float* epi_slope,
                                    float epi_slope_const, int epi_act, float* epi_partial, int B, int H, int W, int Cin,
                                    int Cout, void* stream);
/* ... with coefficient groups (sst_conv_pipe_fwd_grp): epi_scale / epi_shift [B / grp_images][Cin] */
int sst_conv_s2_dgrad_pipe_groups_ok(int B, int H, int W, int Cin, int Cout, int grp_images);
int sst_conv_s2_dgrad_pipe_bwdstats_grp(const float* dy, const float* wp, float* dx, float* ws, const float* epi_y,
                                    const float* epi_scale, const float* epi_shift, const float* epi_slope,
                                    float epi_slope_const, int epi_act, float* epi_partial, int B, int H, int W, int Cin,
                                    int Cout, int grp_images, void* stream);
/* stride-1 data-gradient (mode 1 weights) whose epilogue also emits the BatchNorm/activation BACKWARD partial sums of
 * its result g against the saved conv output epi_y: epi_partial [sst_conv_stat_tiles][3][Cout] = per-tile sums of
 * (gz, gz*epi_y, g*min(z,0)) - the layout sst_bwd_finalize consumes (replaces a separate sst_bwd_reduce pass). */
int sst_conv_dgrad_bwdstats(const float* x, const float* wp, float* y, const float* residual, const float* epi_y,
                            const float* epi_scale, const float* epi_shift, const float* epi_slope,
                            float epi_slope_const, int epi_act, float* epi_partial, int B, int H, int W, int Cin,
                            int Cout, int ksize, void* stream);
/* ---- accumulator mode of the BatchNorm statistics (trunk shape only: sst_conv_acc_supported): the producer conv adds its
 * per-band (sum, Chan-form sum of squares) into fp64 accumulators acc[nrep][C][2] (zero before the launch) with hardware
 * atomics; the consumer conv derives batch statistics + affine of its input from them in its prologue and publishes
 * mean / rstd / scale / shift (+ running statistics) once.  Replaces sst_bn_finalize launches of model.py:174-177's BatchNorms.
 * in2 == null: staged = act(x*scale+shift); in2 != null: staged = x + in2*scale + shift, also written to side_out. */
int sst_conv_acc_supported(int B, int H, int W, int Cin, int Cout, int ksize, int stride);
int sst_conv_fwd_acc(const float* x, const float* in2, float* side_out, const float* ones, const float* wp, float* y,
                     const float* bias, const float* in_slope, float in_slope_const, int in_act, const double* in_acc,
                     const float* in_gamma, const float* in_beta, float in_n, float eps, float momentum, float* o_mean,
                     float* o_rstd, float* o_scale, float* o_shift, float* run_mean, float* run_var, double* st_acc,
                     int nrep, int B, int H, int W, int Cin, int Cout, int ksize, void* stream);
/* one BatchNorm-backward stage in accumulator mode: coefficients from bw_in_acc [nrep][64][4] + mean/rstd/gamma (prologue),
 * next stage's sums added into bw_st_acc [nrep][Cout][4] (epilogue); dgamma/dbeta/dslope written once.  Replaces the
 * sst_bwd_finalize launch between two sst_conv_dgrad_fused stages. */
int sst_conv_dgrad_fused_acc(const float* g, const float* y2, const float* in_scale, const float* in_shift,
                             const float* in_slope, float in_slope_const, int in_act, float* dy_out, const float* wp,
                             float* out, const float* residual, const float* epi_y, const float* epi_scale,
                             const float* epi_shift, const float* epi_slope, float epi_slope_const, int epi_act,
                             const double* bw_in_acc, const float* mean, const float* rstd, const float* gamma, float n,
                             float* dgamma, float* dbeta, float* dslope, double* bw_st_acc, int nrep, int B, int H,
                             int W, int Cin, int Cout, int ksize, void* stream);
int sst_bn_finalize_acc(const double* acc, int nrep, int C, float n, const float* gamma, const float* beta,
                        float* run_mean, float* run_var, float* mean, float* rstd, float* scale, float* shift,
                        float eps, float momentum, void* stream);
/* forward conv on a not-yet-materialised residual sum h = x + y2*bn_scale + bn_shift (model.py:180-186 -> next block's conv):
 * h is formed while the input tile is staged and also written to h_out; ones = [Cin] vector of 1.0f.  Replaces sst_bn_residual. */
int sst_conv_fwd_resin(const float* x, const float* y2, const float* ones, const float* bn_scale, const float* bn_shift,
                       float* h_out, const float* wp, float* y, const float* bias, float* stats, float* stats_cnt,
                       int B, int H, int W, int Cin, int Cout, int ksize, void* stream);
/* one launch per BatchNorm-backward stage: BN+activation backward apply on load (dy = cA*gz + cB*y2 + cC, also written
 * to dy_out for the weight-gradient kernel), data-gradient conv (+ residual), optional partials for the next stage. */
int sst_conv_dgrad_fused(const float* g, const float* y2, const float* cA, const float* cB, const float* cC,
                         const float* in_scale, const float* in_shift, const float* in_slope, float in_slope_const,
                         int in_act, float* dy_out, const float* wp, float* out, const float* residual,
                         const float* epi_y, const float* epi_scale, const float* epi_shift, const float* epi_slope,
                         float epi_slope_const, int epi_act, float* epi_partial, int B, int H, int W, int Cin, int Cout,
                         int ksize, void* stream);
/* dW[Cout][Cin][k][k] (+)= sum_pixels X'(shifted) * dY ; slab = sst_conv_wgrad_chunks*k*k*Cout*Cin floats */
int sst_conv_wgrad_chunks(int B, int Ho, int Wo, int Cin, int Cout, int ksize);   /* general (per-tap) kernel only */
/* chunk count sst_conv_wgrad (njobs = 1) / sst_conv_wgrad_grouped (njobs layers) actually use; H, W = input size.
 * 3x3 stride-1 layers with Cin, Cout multiples of 64 run the all-taps band kernel (csrc/conv_wgrad.hip). */
int sst_conv_wgrad_chunks2(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int njobs);
int sst_conv_wgrad(const float* x, const float* dy, float* slab, float* dw, const float* in_scale,
                   const float* in_shift, const float* in_slope, float in_slope_const, int in_act,
                   int B, int H, int W, int Cin, int Cout, int stride, int ksize, int accumulate,
                   void* stream);
/* in_scale / in_shift [B / grp_images][Cin]: the images are grp_images-sized passes with their own BatchNorm coefficients (see
 * sst_conv_pipe_fwd_grp), dW sums over all of them.  Taken by the all-taps tile kernel only (sst_conv_wgrad_groups_ok). */
int sst_conv_wgrad_groups_ok(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int grp_images);
int sst_conv_wgrad_grp(const float* x, const float* dy, float* slab, float* dw, const float* in_scale,
                   const float* in_shift, const float* in_slope, float in_slope_const, int in_act,
                   int B, int H, int W, int Cin, int Cout, int stride, int ksize, int accumulate,
                       int grp_images, void* stream);
/* Slab reduces of a whole backward pass in one launch: sst_conv_wgrad_grp with accumulate bit 2 (value 4) leaves its slab un-reduced
 * where sst_conv_wgrad_pending_reduce(...) > 0 (= the slab's chunk count; 0: that launch writes dW itself and ignores the bit); the
 * caller then hands sst_wgrad_reduce_multi a HOST array of up to 24 jobs {const float* slab; float* dw; int nchunk, kk, Cout, Cin,
 * accumulate, reserved;} (40 bytes each).  Same arithmetic per job as the per-layer reduce (bit-identical dW). */
int sst_conv_wgrad_pending_reduce(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int has_in_scale, int in_act);
int sst_wgrad_reduce_multi(const void* jobs, int njobs, void* stream);
/* weight gradient of the two 9x9 convs with a 3-channel side (Generator.conv1 / conv3, model.py:101,127):
 * N = (kx, ch3) = 27 of 32 MFMA columns instead of 3.  kind 0 = conv3 (C->3), kind 1 = conv1 (3->C). */
int sst_wgrad_c3_supported(int C, int ksize);
int64_t sst_wgrad_c3_slab_floats(int B, int H, int W, int C);
int sst_wgrad_c3(const float* big, const float* small, float* slab, float* dw, const float* in_slope,
                 float in_slope_const, int in_act, int kind, int B, int H, int W, int C, int accumulate,
                 void* stream);

/* forward of the same two convs with the folding on the K side (conv1: 3->C, also conv3's data-gradient with
 * mode 1 weights) resp. the N side (conv3: C->3, stores NCHW + clamp like out_mode 2 of sst_conv_fwd). */
int64_t sst_conv9_c3_packed_floats(int Cout_eff);
int sst_conv9_c3_pack(const float* w, float* wp, int Cout, int Cin, int mode, void* stream);
int sst_conv9_c3_fwd(const float* x, const float* wp, float* y, const float* bias, int B, int H, int W,
                     int Cout, void* stream);
int64_t sst_conv9_to3_packed_floats(int C);
int sst_conv9_to3_pack(const float* w, float* wp, int C, void* stream);
int sst_conv9_to3_fwd(const float* x, const float* wp, float* y, float* y_pre, const float* bias,
                      const float* in_slope, float in_slope_const, int in_act, int B, int H, int W, int C,
                      void* stream);

/* data-gradient of a 3x3 / stride-2 / pad-1 conv (Discriminator.features model.py:35,42,49,56): the input-gradient
 * pixels are split into 4 parity classes, each a dense 1x1 / 1x2 / 2x1 / 2x2 correlation over dy. */
int64_t sst_conv_s2_dgrad_packed_floats(int Cout, int Cin);
int sst_conv_s2_dgrad_pack(const float* w, float* wp, int Cout, int Cin, void* stream);
int sst_conv_s2_dgrad(const float* dy, const float* wp, float* dx, int B, int H, int W, int Cin, int Cout,
                      void* stream);
/* one BatchNorm-backward stage around the stride-2 data-gradient (contract of sst_conv_dgrad_fused; g / y2 / dy_out live on the
 * conv's output side [B,Ho,Wo,Cout], dx / epi_y are [B,H,W,Cin]); epi_partial [sst_conv_s2_dgrad_tiles(B,H,W)][3][Cin] */
int sst_conv_s2_dgrad_tiles(int B, int H, int W);
/* kernel the two entries above launch for this shape, as rocprofv3 prints it (profiling labels) */
const char* sst_conv_s2_dgrad_kernel_name(int B, int H, int W, int Cin, int Cout, int fused);
int sst_conv_s2_dgrad_fused(const float* g, const float* y2, const float* cA, const float* cB, const float* cC,
                            const float* in_scale, const float* in_shift, const float* in_slope, float in_slope_const,
                            int in_act, float* dy_out, const float* wp, float* dx, const float* epi_y,
                            const float* epi_scale, const float* epi_shift, const float* epi_slope, float epi_slope_const,
                            int epi_act, float* epi_partial, int B, int H, int W, int Cin, int Cout, void* stream);

/* weight gradients of njobs <= 40 layers of IDENTICAL shape in ONE launch (the 33 trunk-shaped convs of the generator):
 * jobs = HOST array of {const float* x, *dy; float* slab, *dw; const float* in_scale, *in_shift, *in_slope;
 * float in_slope_const; int in_act;} (64 bytes each, device pointers inside); the table is handed to the kernels as a
 * by-value argument, so the call is graph-capturable without any device-side table. */
int sst_conv_wgrad_grouped(const void* jobs, int njobs, int B, int H, int W, int Cin, int Cout, int stride,
                           int ksize, int accumulate, void* stream);

/* ---- BatchNorm (train mode) + elementwise glue, tensors viewed as [R rows, C channels] -----------
 * nn.BatchNorm2d model.py:36-57,114,174,177 (eps 1e-5, momentum .1); PReLU/LeakyReLU backward;
 * residual adds model.py:146,183. */
int sst_bn_finalize(const float* stats, const float* cnt, int ntiles, int C, const float* gamma,
                    const float* beta, float* run_mean, float* run_var, float* mean, float* rstd,
                    float* scale, float* shift, float eps, float momentum, void* stream);
/* `groups` passes batched as one tall image (see sst_conv_pipe_fwd_grp): stats / cnt hold ntiles tiles = groups equal consecutive
 * ranges; mean / rstd / scale / shift are [groups][C]; the running statistics take one momentum step per group, in group order. */
int sst_bn_finalize_grp(const float* stats, const float* cnt, int ntiles, int C, int groups, const float* gamma,
                        const float* beta, float* run_mean, float* run_var, float* mean, float* rstd, float* scale,
                        float* shift, float eps, float momentum, void* stream);
int sst_bn_eval_affine(const float* gamma, const float* beta, const float* run_mean,
                       const float* run_var, float* scale, float* shift, int C, float eps, void* stream);
int sst_bn_residual(const float* y, const float* scale, const float* shift, const float* res,
                    const float* res_slope, float* out, int64_t R, int C, void* stream);
int sst_bwd_reduce_blocks(int64_t R, int C);
int sst_bwd_reduce(const float* g, const float* g2, const float* y, const float* scale,
                   const float* shift, const float* slope, float slope_const, int act, float* partial,
                   int64_t R, int C, void* stream);
int sst_bwd_finalize(const float* partial, int nblk, int C, float n, const float* mean,
                     const float* rstd, const float* gamma, float* dgamma, float* dbeta, float* cA,
                     float* cB, float* cC, float* dslope, int accumulate, void* stream);
/* the three backward steps for `groups` passes batched as one tensor of groups * R rows (R rows per pass): scale / shift / mean / rstd /
 * cA / cB / cC are [groups][C], partial is [groups][sst_bwd_reduce_blocks(R, C)][3][C], n = elements of ONE pass per channel;
 * dgamma / dbeta (the passes share the parameters) sum over the passes in group order. */
int sst_bwd_reduce_grp(const float* g, const float* g2, const float* y, const float* scale, const float* shift,
                       const float* slope, float slope_const, int act, float* partial, int64_t R, int C, int groups,
                       void* stream);
int sst_bwd_finalize_grp(const float* partial, int nblk, int C, float n, int groups, const float* mean, const float* rstd,
                         const float* gamma, float* dgamma, float* dbeta, float* cA, float* cB, float* cC, int accumulate,
                         void* stream);
int sst_bwd_apply_grp(const float* g, const float* g2, const float* y, const float* scale, const float* shift,
                      const float* slope, float slope_const, int act, const float* cA, const float* cB, const float* cC,
                      float* dy, int64_t R, int C, int groups, void* stream);
/* channel-parallel also with dslope (scratch: (C+63)/64 floats, counter: one zeroed word, left zero) */
int sst_bwd_finalize_wide(const float* partial, int nblk, int C, float n, const float* mean, const float* rstd,
                          const float* gamma, float* dgamma, float* dbeta, float* cA, float* cB, float* cC,
                          float* dslope, int accumulate, float* scratch, unsigned* counter, void* stream);
/* activation-only backward (PReLU / LeakyReLU, no BatchNorm) fused with the partial sums of the bias and slope
 * gradients: dy = act'(y)*(g+g2), optionally stored pre-PixelShuffle; partial [sst_act_bwd_partial_blocks][3][C or 4C]
 * in sst_bwd_finalize's layout (replaces reduce + finalize + apply + reduce + finalize of model.py:159-161's backward) */
int sst_act_bwd_partial_blocks(int64_t stored_rows);
int sst_act_bwd_partial(const float* g, const float* g2, const float* y, const float* slope, float slope_const,
                        float* dy, float* partial, int64_t R, int C, int unshuffle_H, int unshuffle_W, void* stream);
int sst_bwd_apply(const float* g, const float* g2, const float* y, const float* scale,
                  const float* shift, const float* slope, float slope_const, int act, const float* cA,
                  const float* cB, const float* cC, float* dy, int64_t R, int C, int unshuffle_H,
                  int unshuffle_W, void* stream);
int sst_add(const float* a, const float* b, float* out, int64_t n, void* stream);

/* ---- layout + criterions ------------------------------------------------------------------------
 * sst_transpose: NCHW <-> NHWC (model.py:138-152 keeps an NCHW surface).  sst_clamp_bwd: backward of
 * clamp_(0,1) model.py:150 fused with the NCHW->NHWC hand-off and the conv3 bias gradient.
 * pixel criterion config.py:88-90 (mode 0 MSE / 1 L1); BCEWithLogits config.py:71-73, train.py:113-161. */
int sst_transpose(const float* src, float* dst, int B, int C, int H, int W, int to_nchw, void* stream);
/* feature criterion on activated, per-channel-affine taps (ContentLossDiscriminator loss.py:231-289: BatchNorm(eval) + LeakyReLU
 * outputs of the discriminator): f(v) = act(v*scale[c]+shift[c]) with slope activation, loss = mean crit(f(x) - f(gt)) over a
 * [rows, C] tensor (mode 0 MSE, 1 L1); bwd: dx (+)= scale_host * scale_dev * dloss/dx */
int sst_feat_loss_fwd(const float* x, const float* gt, const float* scale, const float* shift, float slope, int C,
                      float* loss, float* partials, unsigned* counter, int64_t n, int mode, void* stream);
int sst_feat_loss_bwd(const float* x, const float* gt, const float* scale, const float* shift, float slope, int C,
                      float* dx, const float* scale_dev, float scale_host, int accumulate, int64_t n, int mode,
                      void* stream);
/* VGG19 feature stack pieces of ContentLossVGG (loss.py:11-70): ImageNet normalise fused with the layout change,
 * ReLU+MaxPool2d(2) forward/backward; the feature criterion is sst_pixel_loss_* with mode |= 2 (criterion on relu(x)). */
int sst_transpose_affine(const float* src, float* dst, int B, int C, int H, int W, int to_nchw,
                         const float* scale, const float* shift, void* stream);
int sst_maxpool_relu_fwd(const float* y, float* out, int B, int H, int W, int C, void* stream);
int sst_maxpool_relu_bwd(const float* g, const float* y, float* dy, int B, int H, int W, int C, void* stream);
int sst_clamp_bwd_blocks(int B, int H, int W);
int sst_clamp_bwd(const float* g, const float* pre, float* out, float* partial, float* dbias,
                  int accumulate, int B, int C, int H, int W, void* stream);
int sst_pixel_loss_blocks(int64_t n);
int sst_pixel_loss_fwd(const float* x, const float* gt, float* loss, float* partials, unsigned* counter,
                       int64_t n, int mode, void* stream);
int sst_pixel_loss_bwd(const float* x, const float* gt, float* dx, const float* scale_dev,
                       float scale_host, int accumulate, int64_t n, int mode, void* stream);
int sst_bce_logits(const float* logits, float target, float* loss, float* dlogits,
                   const float* scale_dev, float scale_host, int n, void* stream);
/* MATLAB-style antialiased bicubic resampling of bicubic.py:15-105 (LR synthesis of dataset.py:28) with host-built tap
 * tables: x [planes,H,W] -> y [planes,oh,ow]; wy/iy [oh,Ty], wx/ix [ow,Tx]; round_grid: round to the 1/255 grid (no clamp) */
int sst_bicubic(const float* x, float* y, const float* wy, const int* iy, const float* wx, const int* ix,
                int64_t planes, int H, int W, int oh, int ow, int Ty, int Tx, int round_grid, void* stream);
/* ---- best-buddy losses (loss.py:78-142 BestBuddyLoss, loss.py:145-228 GramLoss, loss.py:292-375
 * PatchwiseStructureTensorLoss; ksize 3, stride 3, pad 0, squared-L2 matching):
 * sst_bb_patches cuts an image [B,3,H,W] into 27-vectors (unfold order) + squared norms inside the candidate table
 * cand [B,ncand,27] / cnrm [B,ncand]; sst_bb_match pairs every SR patch with its best candidate (first minimum of
 * alpha*d(sr_i,c_j) + beta*d(gt_i,c_j), d = clamped expanded squared distance of utils.py:173-187) and produces the selected
 * indices, d(loss)/d(sr) and per-workgroup partial sums of the L1 / L2 criterion (loss = sum of partials) */
int sst_bb_blocks(int B, int H, int W);
/* feature mode: 0 raw patch (27, BestBuddyLoss), 1 gram matrix of the patch (9, GramLoss loss.py:145-228), 2 normalised
 * structure tensor of the 3x3 gray patch (27, PatchwiseStructureTensorLoss loss.py:292-375; st_mats = device [Ax 81][Ay 81][K 81],
 * the three 9x9 maps of utils.py:212-233 restricted to a zero-padded 3x3 image) */
int sst_bb_feature_dim(int mode);
int sst_bb_patches(const float* img, float* cand, float* cnrm, int B, int H, int W, int ncand_total, int cand_off,
                   int mode, const float* st_mats, void* stream);
int sst_bb_match(const float* sr, const float* cand, const float* cnrm, int* ind, float* dsr, float* partials, int B,
                 int H, int W, int ncand, float alpha, float beta, int criterion_l2, int mode, const float* st_mats,
                 void* stream);
/* ... with the matching distance of utils.py:157-191 selectable (dist_l1 = 1: sum |x - y|; 0: the clamped expanded squared L2) */
int sst_bb_match_dist(const float* sr, const float* cand, const float* cnrm, int* ind, float* dsr, float* partials, int B,
                 int H, int W, int ncand, float alpha, float beta, int criterion_l2, int mode, const float* st_mats, int dist_l1,
                      void* stream);
/* General patch geometry for BestBuddyLoss (loss.py:86,116-129: F.unfold with any ksize <= 6, pad, stride; dist_norm 'l1' / 'l2'):
 * features in global tables [B, rows, D = 3 k k] (unfold order, zero padding).  sst_bbg_patches = F.unfold's patch count;
 * sst_bbg_unfold fills rows [row_off, row_off + patches) (+ squared norms or null); sst_bbg_match pairs the SR rows srf [B, np, D]
 * with the candidate table (first np rows = full-resolution GT patches): ind [B, np], gfeat [B, np, D] = d(loss)/d(SR patch entries),
 * partials [sst_bbg_blocks(B, np)] (loss = their sum); sst_bbg_fold = the unfold's adjoint, d(sr) [B,3,H,W] (overlapping patches summed
 * in fixed order). */
int sst_bbg_patches(int H, int W, int k, int pad, int stride);
int sst_bbg_blocks(int B, int np);
int sst_bbg_unfold(const float* img, float* table, float* nrm, int B, int H, int W, int k, int pad, int stride, int nrows_total,
                   int row_off, void* stream);
int sst_bbg_match(const float* srf, const float* cand, const float* cnrm, int* ind, float* gfeat, float* partials, int B, int np,
                  int ncand, int D, float alpha, float beta, int criterion_l2, int dist_l1, void* stream);
int sst_bbg_fold(const float* gfeat, float* dsr, int B, int H, int W, int k, int pad, int stride, void* stream);
int sst_weighted_sum(const float* const* terms, const float* weights, int n, float* out, float* weighted,
                     void* stream);

/* ---- discriminator classifier: flatten + Linear + LeakyReLU + Linear (model.py:61-65,69-70) -------
 * weights in the reference layout [N][K]; x [M][K], y [M][N], M <= 64.
 * sst_linear_dgrad: nhwc_HW > 0 scatters dx from the NCHW-flatten index k=(c,hw) to NHWC [M][HW][C]. */
int sst_linear_ksplit(int M, int N, int K);
int sst_linear_fwd(const float* x, const float* w, const float* bias, float* y, float* slab, int M, int N,
                   int K, void* stream);
int sst_linear_dgrad(const float* dy, const float* w, float* dx, int M, int N, int K, int nhwc_C,
                     int nhwc_HW, void* stream);
int sst_linear_wgrad(const float* dy, const float* x, float* dw, float* db, int M, int N, int K,
                     int accumulate, void* stream);
int sst_head_fwd(const float* h, const float* w, const float* b, float* y, int M, int N, int K, float slope,
                 void* stream);
int sst_head_bwd(const float* h, const float* w, const float* dy, float* dh, float* dw, float* db, int M,
                 int N, int K, float slope, int accumulate, void* stream);
int sst_flatten_act(const float* y, const float* scale, const float* shift, float slope, int act,
                    float* flat, int B, int HW, int C, void* stream);
/* scale / shift [B / grp_images][C] (passes batched as one tensor); grp_images = 0: one row */
int sst_flatten_act_grp(const float* y, const float* scale, const float* shift, float slope, int act,
                        float* flat, int B, int HW, int C, int grp_images, void* stream);

/* ---- optimizer: torch.optim.Adam(lr, betas, eps=1e-4) of train.py:62-75 / warmup.py:34-40 as ONE streaming pass over
 * flat parameter / gradient / moment buffers (n floats, n % 4 == 0).  steps: nsteps device floats, all incremented by
 * one (the per-parameter `step` tensors of torch's optimizer state); lr: device scalar (LR schedulers fill it). */
int sst_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, const float* lr, float* steps,
                  int nsteps, double beta1, double beta2, double eps, double weight_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif
