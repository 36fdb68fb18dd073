/*
 * ufr_hip.h -- C ABI of libufr_hip.so, the MI355X (gfx950) implementation of the native
 * operators on the optical-flow attack hot path of lmb-freiburg/understanding_flow_robustness.
 *
 * Every entry point replaces one function of the reference's pybind11/torch extensions; the
 * reference interface it stands in for is cited as file:line (paths relative to the reference
 * tree).  The ABI is plain C: device pointers, sizes and a HIP stream handle -- no torch types.
 *
 * Conventions
 *   - all pointers are DEVICE pointers to dense row-major buffers in the stated layout;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every call only
 *     enqueues work on that stream (no allocation, no synchronisation -> hipGraph-capturable);
 *   - return value: 0 on success, a negative UFR_E* code otherwise; ufr_last_error() returns a
 *     thread-local human-readable message (mirrors TORCH_CHECK text of the reference);
 *   - dtype codes: UFR_F32 / UFR_F64, and UFR_F16 for the spatial correlation, which the reference's CUDA
 *     op dispatches for half too (correlation_cuda_kernel.cu:262, :297); the attack path is fp32.
 */
#ifndef UFR_HIP_H_
#define UFR_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UFR_ABI_VERSION 9   /* 9: ufr_build_manifest; ufr_igemm_desc: `products` 1 / 3 are served again (RAFT's opt-in reduced precision), `fuse_reduce` (split-K summed by the last workgroup of a tile, no second launch) (round 6); 8: ufr_pwc_warp_backward_owner / _workspace_bytes, ufr_raft_normalize_pair*, ufr_raft_fmap_pyramid_* (round 5); 7: the chunk-range entries of csrc/engine_small.hip take the extents of what they walk (w_chunks / g_chunks) and refuse a range that leaves it (round 5); 6: ufr_igemm_desc gained planes_chunks / f32_first_chunk; the clock probe records 8 words (round 4); 5: ufr_igemm_desc gained out_rowmajor / out_ld; ufr_igemm_clock_probe, ufr_conv1_direct, ufr_patch_paste_placed_rect (round 4); 4: ufr_igemm_desc gained tail / tail_n0 (round 3); 3: k_order (round 2) */

enum { UFR_F32 = 0, UFR_F64 = 1, UFR_F16 = 2 };   /* UFR_F16: the spatial correlation only (generic kernels, float32 sums) */
enum {
  UFR_OK = 0,
  UFR_EINVAL = -1,      /* bad argument (shape, dtype, null pointer) */
  UFR_EUNSUPPORTED = -2,/* valid in the reference but not implemented here */
  UFR_ELAUNCH = -3      /* hipGetLastError() after launch != hipSuccess */
};

typedef void* ufr_stream_t;

int ufr_abi_version(void);
const char* ufr_last_error(void);
/* Number of HIP devices visible to the library (0 when there is no GPU); never throws. */
int ufr_device_count(void);
/* Build manifest: one line "<translation unit> <md5>\n" per object of the library, where md5 is the checksum of the sources the
 * object was COMPILED FROM (csrc/<unit>.hip + csrc/ufr_common.h + include/ufr_hip.h, concatenated in that order).  The Python
 * binding recomputes the checksums from the tree at load and refuses a library holding an object built from other sources (round 5
 * lost two full-suite runs to a library that differed from a clean build of the same tree without anyone being able to say where). */
const char* ufr_build_manifest(void);

/* ---- spatial correlation sampler ------------------------------------------------------------
 * replaces spatial_correlation_sampler_backend.forward / .backward
 *   models/Pytorch-Correlation-extension/Correlation_Module/correlation_sampler.cpp:59-87, :89-124
 *   (CUDA: correlation_cuda_kernel.cu:236-327; CPU semantics: correlation.cpp:75-178)
 * input1,input2: [B,C,H,W]; output / grad_output: [B,patchH,patchW,oH,oW] with
 *   oH = (H + 2*padH - ((kH-1)*dilationH+1)) / dH + 1 (same for W).
 * The callee writes every element of its outputs (the reference zero-initialises them). */
typedef struct {
  int kH, kW, patchH, patchW, padH, padW, dilationH, dilationW, dilation_patchH, dilation_patchW,
      dH, dW;
} ufr_corr_params;

int ufr_corr_forward(const void* input1, const void* input2, void* output, int dtype, int B, int C,
                     int H, int W, const ufr_corr_params* p, ufr_stream_t stream);
int ufr_corr_backward(const void* input1, const void* input2, const void* grad_output,
                      void* grad_input1, void* grad_input2, int dtype, int B, int C, int H, int W,
                      const ufr_corr_params* p, ufr_stream_t stream);
/* Fused epilogue used by models/submodules.py:124-138 (`correlate`: view + divide by C) followed
 * by FlowNetC.py:139 / PWCNet.py LeakyReLU: out = leaky(scale*corr, slope); slope=1 disables. */
int ufr_corr_forward_fused(const void* input1, const void* input2, void* output, int dtype, int B,
                           int C, int H, int W, const ufr_corr_params* p, float scale, float slope,
                           ufr_stream_t stream);

/* ---- RAFT on-the-fly correlation ("alt_cuda_corr") ------------------------------------------
 * replaces alt_cuda_corr.forward / .backward  (models/alt_cuda_corr/correlation.cpp:23-48,
 *   kernels correlation_kernel.cu:18-119, :122-256).  fp32 only, like the reference (:278,:313).
 * fmap1: [B,H1,W1,C]  fmap2: [B,H2,W2,C]  coords: [B,N,H1,W1,2] (x,y)
 * corr / corr_grad: [B,N,(2r+1)^2,H1,W1], channel = oy + (2r+1)*ox.
 * backward: coords_grad is zero-filled, as in the reference (never written, :307,:323). */
int ufr_altcorr_forward(const float* fmap1, const float* fmap2, const float* coords, float* corr,
                        int B, int N, int H1, int W1, int H2, int W2, int C, int radius,
                        ufr_stream_t stream);
int ufr_altcorr_backward(const float* fmap1, const float* fmap2, const float* coords,
                         const float* corr_grad, float* fmap1_grad, float* fmap2_grad,
                         float* coords_grad, int B, int N, int H1, int W1, int H2, int W2, int C,
                         int radius, ufr_stream_t stream);

/* The same operator for ALL pyramid levels of one lookup in one launch (AlternateCorrBlock.__call__, models/raft/corr.py:121-137:
 * per level `alt_cuda_corr.forward(fmap1, fmap2_l, coords / 2^l, r)`, stacked, `/ sqrt(dim)`), on the fp32 matrix cores
 * (csrc/raft_altcorr_mfma.hip).  coords: [B,2,H1,W1] planar (x, y) as RAFT holds them; level l uses coords * coord_scale[l];
 * out / grad_out: [B, L*(2r+1)^2, H1, W1] = scale * the stacked per-level volumes.  C in {128, 256}, radius in {3, 4}, N = 1.
 * backward: fmap1_grad [B,H1,W1,C] and every fmap2_grad[l] [B,H2,W2,C]; accumulate != 0 adds onto the buffers (RAFT's 12
 * lookups share them).  fmap2_grad of a level whose segments are split over workgroups (the coarse levels) is the fixed-order sum of
 * the parts' slabs in the workspace (round 4: no float atomics, bit-reproducible); everything else has one writer per element.
 * The levels must be no larger than RAFT's pooled pyramid (H2[l] <= ceil(H1 / 2^l)): the workspace is sized for that.  workspace: ufr_altcorr_pyramid_workspace_bytes() bytes of
 * device memory (per-pixel window origins and blend adjoints, the tiles' boxes, per-level partial sums). */
typedef struct {
  int num_levels;
  const float* fmap2[4];
  float* fmap2_grad[4];                        /* backward only */
  int H2[4], W2[4];
  float coord_scale[4];
} ufr_altcorr_levels;
int ufr_altcorr_pyramid_forward(const float* fmap1, const ufr_altcorr_levels* levels, const float* coords, float* out, int B,
                                int H1, int W1, int C, int radius, float scale, ufr_stream_t stream);
int ufr_altcorr_pyramid_backward(const float* fmap1, const ufr_altcorr_levels* levels, const float* coords, const float* grad_out,
                                 float* fmap1_grad, void* workspace, int B, int H1, int W1, int C, int radius, float scale,
                                 int accumulate, ufr_stream_t stream);
/* The same with the cost volume's gradient as the update engine holds it (round 5): chunk-major float32 [chunks][B*H1*W1][32], channel
 * o = level * (2r+1)^2 + window point at chunk o / 32, lane o % 32 -- no conversion pass to NCHW between the motion encoder's adjoint and
 * the lookup's. */
int ufr_altcorr_pyramid_backward_cm(const float* fmap1, const ufr_altcorr_levels* levels, const float* coords, const float* grad_out_cm,
                                    float* fmap1_grad, void* workspace, int B, int H1, int W1, int C, int radius, float scale,
                                    int accumulate, ufr_stream_t stream);
long ufr_altcorr_pyramid_workspace_bytes(int B, int H1, int W1, int C, int radius, int num_levels);
/* The lookup on the bf16 matrix cores with float32 accuracy (ABI 9, csrc/raft_altcorr_planes.hip; same values as
 * ufr_altcorr_pyramid_forward to float32 summation order): the feature maps as bf16 split planes [3][C / 32][pixels][32] (the igemm's
 * activation layout, v = p0 + p1 + p2 exactly), made ONCE per forward by ufr_altcorr_planes_prepare from the NHWC float32 maps
 * [pixels][C] (they are constant over RAFT's 12 lookups, models/raft/corr.py:111-119); a workgroup owns 8 x 16 pixels of one level and
 * stages the bounding box of their windows through LDS once (any coordinate field is handled; a discontinuous one walks a larger box).
 * coords: planar [B, 2, H1, W1]; out: [B, L (2r+1)^2, H1, W1] = scale * the stacked per-level volumes.  C in {128, 256}, radius in {3, 4}. */
typedef struct {
  int num_levels;
  const void* planes[4];                       /* fmap2 level l: bf16 [3][C / 32][B * H2 * W2][32] */
  long plane_stride[4];                        /* elements between two planes of level l */
  int H2[4], W2[4];
  float coord_scale[4];
} ufr_altcorr_plane_levels;
int ufr_altcorr_planes_prepare(const float* fmap_nhwc, void* planes, long plane_stride, long npix, int C, ufr_stream_t stream);
int ufr_altcorr_planes_forward(const void* fmap1_planes, long fmap1_plane_stride, const ufr_altcorr_plane_levels* levels,
                               const float* coords, float* out, int B, int H1, int W1, int C, int radius, float scale,
                               ufr_stream_t stream);

/* ---- RAFT all-pairs pyramid lookup -----------------------------------------------------------
 * replaces CorrBlock.__call__ (models/raft/corr.py:72-96): per level a (2r+1)^2 bilinear window
 * (grid_sample align_corners=True, zero padding) around coords/2^l, concatenated over levels.
 * level l volume: [B*H1*W1, Hl, Wl]; coords: [B,2,H1,W1]; out: [B, L*(2r+1)^2, H1, W1].
 * backward ACCUMULATES (+=) into the per-level gradient volumes: the caller zeroes them once and
 * may run the adjoints of several lookups (RAFT's 12 iterations) into the same buffers. */
#define UFR_MAX_LEVELS 8
typedef struct {
  int num_levels;
  const float* vol[UFR_MAX_LEVELS];
  float* grad_vol[UFR_MAX_LEVELS]; /* backward only */
  int Hl[UFR_MAX_LEVELS], Wl[UFR_MAX_LEVELS];
} ufr_pyramid;

int ufr_corr_lookup_forward(const ufr_pyramid* pyr, const float* coords, float* out, int B, int H1,
                            int W1, int radius, ufr_stream_t stream);
int ufr_corr_lookup_backward(const ufr_pyramid* pyr, const float* coords, const float* grad_out,
                             int B, int H1, int W1, int radius, ufr_stream_t stream);

/* ---- Resample2d (FlowNet2 backward warp) ------------------------------------------------------
 * replaces resample2d_cuda.forward / .backward (models/resample2d_package/resample2d_cuda.cc:6-24,
 *   kernels resample2d_kernel.cu:15-72, :75-125, :127-198).  fp32 only (:221-234).
 * input1 (image): [B,C,Hi,Wi]  input2 (flow): [B,2,H,W]  output: [B,C,H,W]
 * backward writes every element of grad_input1 [B,C,Hi,Wi] and grad_input2 [B,2,H,W]. */
int ufr_resample2d_forward(const float* input1, const float* input2, float* output, int B, int C,
                           int Hi, int Wi, int H, int W, int kernel_size, int bilinear,
                           ufr_stream_t stream);
int ufr_resample2d_backward(const float* input1, const float* input2, const float* grad_output,
                            float* grad_input1, float* grad_input2, int B, int C, int Hi, int Wi,
                            int H, int W, int kernel_size, int bilinear, ufr_stream_t stream);
/* The same adjoint for kernel_size 1 and an image of the flow's size WITHOUT global atomics (csrc/resample2d_owner.hip, round 4):
 * launch A writes grad_input2 and a table of the output tiles' sampling boxes into `workspace`
 * (ufr_resample2d_backward_workspace_bytes(B, H, W) bytes, 16-byte aligned), launch B lets every 32 x 64 tile of grad_input1 be
 * computed and WRITTEN by one owner workgroup (no zero fill needed; C <= 12). */
long ufr_resample2d_backward_workspace_bytes(int B, int H, int W);
int ufr_resample2d_backward_owner(const float* input1, const float* input2, const float* grad_output, float* grad_input1,
                                  float* grad_input2, void* workspace, long workspace_bytes, int B, int C, int H, int W,
                                  ufr_stream_t stream);

/* ---- FlowNet2's glue between its sub-networks (csrc/fn2_glue.hip, round 5) --------------------------------------------------
 * models/flownet2_models.py:122-205 strings FlowNetC, two FlowNetS, FlowNetSD and FlowNetFusion together with ~20 elementwise
 * torch operators per stage (mul / interpolate, slice, sub, div, cat around Resample2d and ChannelNorm, and their adjoints).  These
 * entries are those operators as streaming kernels; Resample2d itself stays ufr_resample2d_*.  All tensors NCHW float32, dense.
 *   ufr_flow_upscale4_*: out [B,2,4h,4w] = upsample_x4(flow * scale) (divide = 0) or (flow / scale) (divide != 0), bilinear with
 *     torch's align_corners = False weights (`upsample1`, flownet2_models.py:133) or nearest (`upsample3/4`, :160, :176); the
 *     adjoint gathers (no atomics, overwrites grad_flow).
 *   ufr_fn2_stage_pack: out [B,12,H,W] = cat(x [B,6], resampled [B,3], flow [B,2] / div_flow, ChannelNorm(x[:, :3] - resampled))
 *     (:138-145, :150-157).  ufr_fn2_stage_unpack_grad: grad_x[:, 0:3] (complete) and grad_resampled [B,3,H,W] = what enters
 *     Resample2d's adjoint, from grad_out [B,12,H,W] and the packed tensor.  ufr_fn2_stage_finish_grad: grad_x[:, 3:6] =
 *     grad_out[:, 3:6] + Resample2d's image gradient; grad_flow = Resample2d's flow gradient + grad_out[:, 9:11] / div_flow.
 *   ufr_fn2_fusion_*: the same three steps for FlowNetFusion's input (:183-205):
 *     out [B,11,H,W] = cat(x[:, :3], flow_sd, flow_s2, |flow_sd|, |flow_s2|, |x1 - res_sd|, |x1 - res_s2|). */
int ufr_flow_upscale4_forward(const float* flow, float* out, int B, int h, int w, int bilinear, float scale, int divide,
                              ufr_stream_t stream);
int ufr_flow_upscale4_backward(const float* grad_out, float* grad_flow, int B, int h, int w, int bilinear, float scale, int divide,
                               ufr_stream_t stream);
int ufr_fn2_stage_pack(const float* x, const float* resampled, const float* flow, float* out, int B, int H, int W, float div_flow,
                       ufr_stream_t stream);
int ufr_fn2_stage_unpack_grad(const float* grad_out, const float* packed, float* grad_x, float* grad_resampled, int B, int H, int W,
                              ufr_stream_t stream);
int ufr_fn2_stage_finish_grad(const float* grad_out, const float* grad_image, const float* grad_flow_rs, float* grad_x, float* grad_flow,
                              int B, int H, int W, float div_flow, ufr_stream_t stream);
int ufr_fn2_fusion_pack(const float* x, const float* flow_sd, const float* flow_s2, const float* res_sd, const float* res_s2, float* out,
                        int B, int H, int W, ufr_stream_t stream);
int ufr_fn2_fusion_unpack_grad(const float* grad_out, const float* packed, const float* res_sd, const float* res_s2, float* grad_x,
                               float* grad_res_sd, float* grad_res_s2, float* grad_flow_sd, float* grad_flow_s2, int B, int H, int W,
                               ufr_stream_t stream);
int ufr_fn2_fusion_finish_grad(const float* grad_image_sd, const float* grad_image_s2, const float* grad_flow_rs_sd,
                               const float* grad_flow_rs_s2, float* grad_x, float* grad_flow_sd, float* grad_flow_s2, int B, int H, int W,
                               ufr_stream_t stream);

/* ---- ChannelNorm -------------------------------------------------------------------------------
 * replaces channelnorm_cuda.forward / .backward (models/channelnorm_package/channelnorm_cuda.cc:6-25,
 *   kernels channelnorm_kernel.cu:18-60, :63-96); norm_deg accepted and ignored like the reference.
 * input1: [B,C,H,W] -> output [B,1,H,W]. */
int ufr_channelnorm_forward(const float* input1, float* output, int B, int C, int H, int W,
                            int norm_deg, ufr_stream_t stream);
int ufr_channelnorm_backward(const float* input1, const float* output, const float* grad_output,
                             float* grad_input1, int B, int C, int H, int W, int norm_deg,
                             ufr_stream_t stream);

/* ---- patch-attack inner loop, elementwise stages ----------------------------------------------
 * replaces the tensor arithmetic of attack() in patch_attacks/main.py:537-542 (paste),
 * :557-566 (loss), :581-600 (update, re-paste, clamp).
 * Canvas tensors are [B,3,H,W]; `patch`/`mask` are canvas-sized with batch stride
 * patch_bstride / mask_bstride elements (0 = one patch shared by the whole batch).
 *
 * ufr_patch_paste:  adv = clamp((1-mask)*img + mask*patch, lo, hi) for both frames.
 *   `do_clamp`=0 reproduces the un-clamped first paste (main.py:537-542). */
int ufr_patch_paste(const float* tgt, const float* ref, const float* patch, const float* mask,
                    float* adv_tgt, float* adv_ref, int B, int CHW, long patch_bstride,
                    long mask_bstride, int do_clamp, float lo, float hi, ufr_stream_t stream);
/* ufr_patch_update: patch -= clamp(step * (g_tgt + g_ref), -bound, bound)   (main.py:581-583,
 *   step = 0.5*lr, bound = 2), then re-paste + clamp both frames (main.py:585-600): the reference's
 *   canvas-sized arithmetic, for one pair (patch_bstride 0, B = 1) or B independent attacks with their own
 *   canvas patches (patch_bstride = CHW).  One patch behind several pairs is the patch-coordinate family below. */
int ufr_patch_update(const float* tgt, const float* ref, const float* g_tgt, const float* g_ref,
                     float* patch, const float* mask, float* adv_tgt, float* adv_ref, int B, int CHW,
                     long patch_bstride, long mask_bstride, float step, float bound, float lo, float hi,
                     const float* gate_state, ufr_stream_t stream);
/* Device-side form of the loop control of attack() (main.py:546 `while loss_scalar > 0.1`, :605
 * `loss.item()`, :610 `count > max_count-1`): the reference synchronises the host every iteration
 * to read the loss; here iterations are enqueued back to back and a 3-float state word decides on
 * the device whether an iteration still takes effect.
 *   state[0] = stopped (0/1), state[1] = iterations executed, state[2] = loss of the last one.
 * ufr_patch_update(..., gate_state) is a no-op when gate_state != NULL and gate_state[0] != 0.
 * ufr_attack_gate runs after it: if not stopped { state[1]+=1; state[2]=*loss_cur;
 *   if (*loss_cur <= threshold) state[0]=1; }   -- exactly the reference's order of events. */
int ufr_attack_gate(const float* loss_cur, float* state, float threshold, ufr_stream_t stream);
/* ---- one patch shared by several pairs, in PATCH coordinates ------------------------
 * The reference keeps the patch canvas-sized inside attack() and crops it at the sample's placement
 * (ry, rx) back to patch_shape afterwards (patch_attacks/main.py:396-424, :408 mask, :410-424 crop).  With B pairs
 * (and N ranks) behind ONE patch the common object is therefore the cropped patch P[3,ph,pw] with its mask
 * Mp[3,ph,pw]; pair b shows it at origins[2b] = row, origins[2b+1] = column (int32, device-resident so one
 * captured graph serves every placement; `origins_host`, optional, lets the call validate the placements).
 * ufr_patch_grad_crop: rows[g][e] = [Mp[e] != 0] * sum over the pairs of group g, ascending, of
 *   (g_tgt + g_ref)[b, c, row_b + i, col_b + j]; rows is [groups][3*ph*pw + 1], the last column carries the
 *   rank's loss (*loss_local, row 0).  A rank contributes its rows to an all-gather (31 KB per row at 51x51).
 * ufr_patch_apply: G = sum of n_rows rows in ascending order (bit-identical on every rank),
 *   P -= clamp(step*G, +-bound) (main.py:581-583), *loss = sum of the rows' last column.
 * ufr_patch_paste_placed: adv = clamp((1-M_b)*img + M_b*place(P)) for both frames of every pair
 *   (main.py:537-542 with do_clamp = 0, :585-600 with 1); mask_out (optional) receives the canvas masks M_b. */
int ufr_patch_grad_crop(const float* g_tgt, const float* g_ref, const float* mask_p, const int* origins,
                        const int* origins_host, const float* loss_local, float* rows, int B, int H, int W,
                        int ph, int pw, int groups, ufr_stream_t stream);
/* ufr_patch_grad_crop from the WINDOW gradients of the windowed prefix instead of canvas-sized ones (round 4): gxw [2B,3,wh,ww] =
 * d loss / d (first frames | second frames) on each pair's window, win[b] = {y0, x0, ...} (8 ints per pair, the table of
 * ufr_cone_window; origins clamped into the frame as ufr_window_scatter clamps them); zero gradient outside the window. */
int ufr_patch_grad_crop_window(const float* gxw, const int* win, const float* mask_p, const int* origins, const float* loss_local,
                               float* rows, int B, int H, int W, int wh, int ww, int ph, int pw, int groups, ufr_stream_t stream);
int ufr_patch_apply(const float* rows, int n_rows, float* patch_p, float* loss, int ph, int pw, float step,
                    float bound, const float* gate_state, ufr_stream_t stream);
int ufr_patch_paste_placed(const float* tgt, const float* ref, const float* patch_p, const float* mask_p,
                           const int* origins, const int* origins_host, float* adv_tgt, float* adv_ref,
                           float* mask_out, int B, int H, int W, int ph, int pw, int do_clamp, float lo, float hi,
                           const float* gate_state, ufr_stream_t stream);
/* The same on the B x 3 x ph x pw rectangle pixels only: the re-paste of the second and later iterations of a call, when the
 * canvas outside the rectangles already holds clamp(frame) from the first iteration's full paste (main.py:585-600). */
int ufr_patch_paste_placed_rect(const float* tgt, const float* ref, const float* patch_p, const float* mask_p,
                                const int* origins, float* adv_tgt, float* adv_ref, int B, int H, int W, int ph, int pw,
                                int do_clamp, float lo, float hi, const float* gate_state, ufr_stream_t stream);
/* ufr_flow_loss: loss = mean_b,h,w(1 - cos(flow, target))            (kind 0, main.py:564-566)
 *             or mean(sqrt(sum_c (flow-target)^2 + 1e-8))           (kind 1, main.py:557-562)
 *   flow,target: [B,2,H,W].  Writes d loss / d flow (already scaled by `weight`, = 1-alpha) to
 *   grad_flow and accumulates the scalar loss into *loss (caller zeroes it).  `partials` = workspace of
 *   UFR_LOSS_PARTIALS floats: the workgroups' partial sums, added in a fixed order by a second one-wave
 *   kernel, so the scalar behind `while loss_scalar > 0.1` (main.py:546) is bit-reproducible run to run. */
#define UFR_LOSS_PARTIALS 1024
int ufr_flow_loss(const float* flow, const float* target, float* grad_flow, float* loss, int B,
                  int HW, int kind, float weight, float* partials, ufr_stream_t stream);
/* ufr_flow2_upsampled_loss: the same loss on flow = interpolate(flow2 * flow_scale, scale_factor 4, bilinear,
 *   align_corners False) (models/FlowNetC.py:193-197: the reference upsamples flow2 * div_flow before the loss) WITHOUT
 *   writing the full-size flow: flow2 [B,2,h,w], target [B,2,4h,4w]; writes d loss / d flow2 [B,2,h,w] (the adjoint of
 *   the upsampling and of the scaling applied, a gather per cell: no atomics, fixed order) and accumulates the scalar.
 *   Replaces interpolate + ufr_flow_loss + the autograd adjoint of interpolate (three passes over the full-size flow).
 *   UFR_EUNSUPPORTED when B * ceil(h/16) * ceil(w/16) > UFR_LOSS_PARTIALS (the caller keeps the three-pass form). */
int ufr_flow2_upsampled_loss(const float* flow2, float flow_scale, const float* target, float* grad_flow2, float* loss, int B,
                             int h, int w, int kind, float weight, float* partials, ufr_stream_t stream);

/* ---- RAFT SepConvGRU gate arithmetic -----------------------------------------------------------
 * replaces the elementwise half of models/raft/update.py:61-73 (10 launches forward per half-step):
 *   gates: z = sigmoid(zr_pre[:, :Ch]);  rh = sigmoid(zr_pre[:, Ch:]) * h      zr_pre: [B,2Ch,HW]
 *          rh is written with batch stride rh_bstride (>= Ch*HW) so it can land inside the
 *          [r*h | x] buffer that feeds convq.
 *   blend: h' = (1 - z)*h + z*tanh(q_pre)
 * backward entry points return every input gradient of the fused expression (g_h is the part of
 * d/dh that flows through this stage only; the caller adds the stages together). */
int ufr_gru_gates_forward(const float* zr_pre, const float* h, float* z_out, float* rh_out, int B, int Ch,
                          int HW, long rh_bstride, ufr_stream_t stream);
int ufr_gru_gates_backward(const float* zr_pre, const float* h, const float* g_z, const float* g_rh,
                           float* g_zr, float* g_h, int B, int Ch, int HW, long grh_bstride,
                           ufr_stream_t stream);
int ufr_gru_blend_forward(const float* q_pre, const float* z, const float* h, float* h_out, long total,
                          ufr_stream_t stream);
int ufr_gru_blend_backward(const float* q_pre, const float* z, const float* h, const float* g, float* g_qpre,
                           float* g_z, float* g_h, long total, ufr_stream_t stream);

/* ---- RAFT update block on the engine's chunk-major layout (csrc/raft_update.hip; the convolutions run through ufr_igemm) ----
 * replaces the non-convolution arithmetic of models/raft/update.py:35-162 on activation planes bf16 [3][chunks][M][32] and
 * float32 tensors [chunks][M][32] (M = B*H*W):
 *   ufr_raft_flow_patches: the 7x7x2 neighbourhood of `flow` [B,2,H,W] (convf1, update.py:99,:113) as 98 of 128 plane channels
 *     k = (ky*7 + kx)*2 + c at chunks [chunk0, chunk0 + 4): the 49-tap convolution becomes a 1x1 launch;
 *   ufr_raft_motion_finish: cat([out, flow]) (update.py:120): channels 126 / 127 of the 4 motion chunks at chunk0 of p1 = flow,
 *     and the 4 chunks copied to the same place in p2 (the second half-step's GRU buffer);
 *   ufr_gru_gates_cm_*: zr [2*chunks][M][32] pre-activations -> sigmoid values IN PLACE, rh = r*h (planes); adjoint: g_zr planes
 *     [g_z z(1-z) | g_rh h r(1-r)], g_h += g_rh r; consume_g_rh != 0 leaves zeros in g_rh (it then lives in a running-sum
 *     buffer [h | inp | motion | r*h] whose next writer, an igemm epilogue, adds onto it);
 *   ufr_gru_blend_cm_*: q -> tanh IN PLACE, out = (1-z) h + z q (planes); adjoint: g_q_pre planes, g_z, g_h = g (1-z). */
int ufr_raft_flow_patches(const float* flow, void* planes, long plane_stride, int chunk0, int B, int H, int W, ufr_stream_t stream);
int ufr_raft_motion_finish(void* p1, long plane_stride1, void* p2, long plane_stride2, int chunk0, const float* flow, int B, int H,
                           int W, ufr_stream_t stream);
/* ufr_raft_motion_finish for a motion encoder whose last convolution was a `no_reduce` split-K launch (N <= 126 outputs, Npad 128): the
 * planes of BOTH GRU buffers come straight from the slabs (ascending sum, bias, LeakyReLU(slope), zeros in the padding, flow in 126 / 127). */
int ufr_raft_motion_finish_slabs(const float* ws, int splitk, int Npad, int N, const float* bias, float slope, void* p1, long plane_stride1,
                                 void* p2, long plane_stride2, int chunk0, const float* flow, int B, int H, int W, ufr_stream_t stream);
/* Round 5, the loop's small change: ufr_raft_coords_step = `coords1 += delta` (delta NULL at the loop's entry) + the copy the next
 * lookup's adjoint reads + `flow = coords1 - coords0` (raft.py:190-228) in one kernel of n floats; ufr_grad_finalize_consume =
 * ufr_grad_finalize on `chunks` chunks of a running sum g [chunks][M][32] that is zeroed behind the read (pointers AT the first chunk:
 * g, plane 0 of the mask activation, plane 0 of the output planes). */
int ufr_raft_coords_step(float* coords1, const float* delta, const float* coords0, float* saved, float* flow, long n, ufr_stream_t stream);
int ufr_grad_finalize_consume(float* g, const void* mask_plane0, void* out, long out_plane_stride, long M, int chunks, float slope,
                              ufr_stream_t stream);
int ufr_gru_gates_cm_forward(float* zr, const void* h, long h_plane_stride, int h_chunk0, void* rh, long rh_plane_stride,
                             int rh_chunk0, long M, int chunks, ufr_stream_t stream);
/* The same two kernels reading the pre-activations straight from a `no_reduce` split-K launch's slabs (columns [z | r] resp. q; the
 * float32 pre-activation tensor is then only written -- sigmoid / tanh values for the adjoint -- never read).  `addend` (ABI 7,
 * may be NULL): a float32 tensor in the layout of zr / q added behind the bias -- the share of the pre-activation that does not
 * change between RAFT's iterations (the context features `inp`, models/raft/raft.py:176-180: the same tensor enters the GRU of all
 * 12 iterations, update.py:50-66; its convolution is computed once per forward instead of 24 times). */
int ufr_gru_gates_cm_forward_slabs(const float* ws, int splitk, int Npad, const float* bias, const float* addend, float* zr, const void* h,
                                   long h_plane_stride, int h_chunk0, void* rh, long rh_plane_stride, int rh_chunk0, long M, int chunks,
                                   ufr_stream_t stream);
int ufr_gru_blend_cm_forward_slabs(const float* ws, int splitk, int Npad, const float* bias, const float* addend, float* q, const float* z,
                                   const void* h, long h_plane_stride, int h_chunk0, void* out, long out_plane_stride, int out_chunk0,
                                   long M, int chunks, ufr_stream_t stream);
int ufr_gru_blend_cm_forward(float* q, const float* z, const void* h, long h_plane_stride, int h_chunk0, void* out,
                             long out_plane_stride, int out_chunk0, long M, int chunks, ufr_stream_t stream);
/* acc_gq / acc_gzr (ABI 7, may be NULL): float32 running sums in the layout of q / zr onto which the pre-activation gradients are
 * ADDED -- the adjoint of the iteration-invariant share runs once per backward on the sum over the iterations. */
int ufr_gru_blend_cm_backward(const float* q, const float* z, const void* h, long h_plane_stride, int h_chunk0, const float* g,
                              void* gq, long gq_plane_stride, int gq_chunk0, float* g_z, float* g_h, long M, int chunks,
                              float* acc_gq, ufr_stream_t stream);
int ufr_gru_gates_cm_backward(const float* zr, const void* h, long h_plane_stride, int h_chunk0, const float* g_z, float* g_rh,
                              void* gzr, long gzr_plane_stride, int gzr_chunk0, float* g_h, long M, int chunks, int consume_g_rh,
                              float* acc_gzr, ufr_stream_t stream);

/* ---- RAFT BasicEncoder: normalisation / ReLU / residual arithmetic between the convolutions (csrc/raft_norm.hip) -----------
 * replaces nn.InstanceNorm2d / nn.BatchNorm2d (eval) + ReLU + the residual add of models/raft/extractor.py:5-78, :142-215 on
 * x = float32 [chunks][n*HW][32] (a convolution's output, ufr_igemm out_f32) and activation planes:
 *   ufr_cm_norm_stats: stats[(image * C + c) * 2 + {0,1}] = mean, 1 / sqrt(var + eps) over the image's HW pixels (biased
 *     variance; float64 partial sums in `workspace`, ufr_cm_norm_workspace_doubles() doubles);
 *   ufr_cm_norm_apply: out planes = relu2(res + relu1((x - mean) * rstd)); stats NULL = identity (BatchNorm folded into the
 *     convolution), res NULL = no residual;
 *   ufr_cm_norm_backward: gz planes = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = G * [outmask > 0] * [xhat > 0 if relu1]
 *     (G float32 chunk-major; outmask = plane 0 of the block's output, NULL = none; stats NULL = no statistics terms);
 *   ufr_cm_masked_copy: out = G * [mask > 0] over `elems` float32 elements (the skip connection's share). */
long ufr_cm_norm_workspace_doubles(long HW, int n, int chunks);
int ufr_cm_norm_stats(const float* x, float* stats, double* workspace, long HW, int n, int chunks, float eps, ufr_stream_t stream);
int ufr_cm_norm_apply(const float* x, const float* stats, const void* res, long res_plane_stride, int res_chunk0, void* out,
                      long out_plane_stride, int out_chunk0, long HW, int n, int chunks, int relu1, int relu2, ufr_stream_t stream);
/* ufr_cm_norm_stats + ufr_cm_norm_apply as TWO launches instead of three (ABI 6): the workgroups of the apply kernel add the float64
 * partials of their 32 channels themselves (same order, same arithmetic as the statistics' second stage) and leave `stats` behind for the
 * adjoint; ufr_cm_norm_backward does the same with its sums. */
int ufr_cm_norm_stats_apply(const float* x, float* stats, double* workspace, float eps, const void* res, long res_plane_stride,
                            int res_chunk0, void* out, long out_plane_stride, int out_chunk0, long HW, int n, int chunks, int relu1,
                            int relu2, ufr_stream_t stream);
int ufr_cm_norm_backward(const float* x, const float* G, const void* outmask, int mask_chunk0, const float* stats, float* sums,
                         double* workspace, void* gz, long gz_plane_stride, int gz_chunk0, long HW, int n, int chunks, int relu1,
                         ufr_stream_t stream);
int ufr_cm_masked_copy(const float* G, const void* outmask, long mask_elem_offset, float* out, long elems, ufr_stream_t stream);

/* ---- PWC-Net conv1a straight from the raw frames (csrc/small_cin_conv.hip) ---------------------------------------------------
 * replaces Conv2d(3, N <= 32, 3, stride 2, padding 1) + LeakyReLU(slope) of models/PWCNet.py:55-60, :235 on NCHW float32 frames
 * [n, 3, H, W] (H, W even): out planes chunk `out_chunk0` = split(leaky(conv + bias)) at [n, H/2, W/2]; weight [N][3][3][3] float32 (the
 * caller folds the RGB -> BGR flip of PWCNet.py:230-231 into it).  The chunk's channels >= 8 ceil(N / 8) are NOT written: the planes
 * must hold zeros there (as allocated). */
int ufr_conv3x3s2_c3_planes(const float* frames, const float* weight, const float* bias, float slope, void* out_planes,
                            long plane_stride, int out_chunk0, int n, int N, int H, int W, ufr_stream_t stream);

/* conv1aa / conv1b: Conv2d(16, 16, 3, 1, 1) + LeakyReLU(slope) on activation planes (channels 0-15 of chunk in_chunk0 -> channels 0-15 of
 * chunk out_chunk0 at the same [n, H, W]; channels 16-31 of the output chunk are NOT written).  weight_ct16 = the weight permuted to
 * [16 input channels][9 taps][16 outputs] float32 (PWCNet.py:56-57). */
int ufr_conv3x3_c16_planes(const void* in_planes, long in_plane_stride, int in_chunk0, const float* weight_ct16, const float* bias,
                           float slope, void* out_planes, long out_plane_stride, int out_chunk0, int n, int H, int W, ufr_stream_t stream);

/* ---- PWC-Net backward warp ----------------------------------------------------------------------
 * replaces PWCDCNet.warp (models/PWCNet.py:164-204): grid from the flow (normalised with W-1, sampled with
 * align_corners=False, as the reference does), bilinear grid_sample with zero padding, times the validity mask
 * `grid_sample(ones) >= 0.0001`.  x [B,C,H,W], flow [B,2,H,W]; backward writes grad_x (zero-filled, then
 * scattered) and grad_flow. */
int ufr_pwc_warp_forward(const float* x, const float* flow, float* out, int B, int C, int H, int W,
                         ufr_stream_t stream);
int ufr_pwc_warp_backward(const float* x, const float* flow, const float* grad_out, float* grad_x, float* grad_flow,
                          int B, int C, int H, int W, ufr_stream_t stream);
/* The same adjoint without float atomics and without the zero fill (owner-computes, csrc/pwc_warp.hip): grad_x is bit-reproducible
 * (every cell adds its contributions in the order of their source pixels) unless a cell receives more than 16 corners (a flow that
 * compresses > 4x in both directions): the corners beyond 16 are added with float atomics afterwards.  workspace: 16-byte aligned,
 * ufr_pwc_warp_backward_workspace_bytes(B, H, W) bytes (the sampling boxes of the 8 x 32 tiles), written and read by this call in
 * stream order. */
long ufr_pwc_warp_backward_workspace_bytes(int B, int H, int W);
int ufr_pwc_warp_backward_owner(const float* x, const float* flow, const float* grad_out, float* grad_x, float* grad_flow,
                                void* workspace, long workspace_bytes, int B, int C, int H, int W, ufr_stream_t stream);

/* ---- RAFT's glue in front of the encoders and the on-the-fly correlation (csrc/raft_glue.hip) -------------------------------------
 * ufr_raft_normalize_pair: models/raft/raft.py:128-129 `image = 2 * (image / 255.0) - 1.0` for both frames in one pass, written as the
 * stack [2B,3,H,W] the feature encoder concatenates them to (raft.py:141); bit for bit torch's three operators.  n_each = elements of
 * one frame tensor (a multiple of 4; 16-byte aligned tensors).  _backward: grad1 = grad_stack[:B] * 2 / 255, grad2 likewise (may be NULL).
 * ufr_raft_fmap_pyramid_forward: models/raft/corr.py:97-105, :128-129 (AlternateCorrBlock): levels_nhwc[0] = fmap permuted to
 * [B,H,W,C]; levels_nhwc[l] = avg_pool2d(2, 2) of level l - 1 ([B, H >> l, W >> l, C]), bit for bit torch's; 1 <= levels <= 4.
 * _backward: grad_fmap [B,C,H,W] = the levels' gradients (NULL entries: none) through the poolings' and permutations' adjoints. */
int ufr_raft_normalize_pair(const float* image1, const float* image2, float* stack, long n_each, ufr_stream_t stream);
int ufr_raft_normalize_pair_backward(const float* grad_stack, float* grad1, float* grad2, long n_each, ufr_stream_t stream);
int ufr_raft_fmap_pyramid_forward(const float* fmap, float* const* levels_nhwc, int levels, int B, int C, int H, int W,
                                  ufr_stream_t stream);
int ufr_raft_fmap_pyramid_backward(const float* const* grad_levels_nhwc, int levels, float* grad_fmap, int B, int C, int H, int W,
                                   ufr_stream_t stream);

/* ---- RAFT convex upsampling ------------------------------------------------------------------------
 * replaces RAFT.upsample_flow (models/raft/raft.py:111-122): flow [N,2,H,W], mask [N,576,H,W] (9 x 8 x 8 logits
 * per coarse pixel) -> up [N,2,8H,8W] = softmax-weighted combination of the 3x3 neighbours of 8*flow.
 * backward writes grad_flow [N,2,H,W] and grad_mask [N,576,H,W]; workspace: N*2*9*H*W floats. */
int ufr_convex_upsample_forward(const float* flow, const float* mask, float* up, int N, int H, int W,
                                ufr_stream_t stream);
int ufr_convex_upsample_backward(const float* flow, const float* mask, const float* grad_up, float* grad_flow,
                                 float* grad_mask, float* workspace, int N, int H, int W, ufr_stream_t stream);

/* ---- universal perturbation / I-FGSM inner loop, elementwise stages ------------------------------
 * replaces global_attacks/perturb_model.py:102-145 (compute_flow_loss) and the tensor arithmetic of
 * global_attacks/universal_perturbation.py:477-520 (attack), :667-675 (add_universal_perturbation).
 *
 * ufr_flow_loss_ex: kind 0 cossim, 1 l2 (sqrt(.+10e-8)), 2 l1; gt has 2 channels, or 3 with a
 *   validity mask in the last one.  scale = 1/(averaged element count) or 1/(sum(valid)+1e-8), taken
 *   from *scale_dev when that device pointer is non-NULL (so a captured graph follows new masks),
 *   else from `scale`.  Writes d loss/d flow, accumulates the scalar into *loss (`partials`: as ufr_flow_loss). */
int ufr_flow_loss_ex(const float* flow, const float* gt, float* grad_flow, float* loss, int B, int HW,
                     int gt_channels, int kind, float scale, const float* scale_dev, float* partials,
                     ufr_stream_t stream);
/* ufr_universal_update: one sign-gradient step on both frames.
 *   shared = 0: the reference's per-sample arithmetic; delta is [B,2,3,H,W]
 *       adv = clamp(adv -/+ lr*dir(g), lo, hi); noise = clamp(adv - img, +-eps); adv = img + noise
 *   shared = 1: one perturbation [2,3,H,W] for the whole (sharded) batch, direction from the SUM of
 *       the per-sample gradients; mode 0 fused, 1 sum only -> grad_sum
 *       [2*CHW], 2 apply grad_sum, so an all-reduce can sit between 1 and 2.
 *   use_sign: 1 = I-FGSM (torch.sign), 0 = "ifgm"; frames: bit 0 = frame 0, bit 1 = frame 1
 *   (perturb_mode both/left/right); ascent: 0 = gradient descent (default), 1 = --add_gaussian. */
int ufr_universal_update(const float* img0, const float* img1, const float* g0, const float* g1,
                         float* grad_sum, float* adv0, float* adv1, float* delta, int B, int CHW,
                         float lr, float eps, float lo, float hi, int use_sign, int frames, int ascent,
                         int shared, int mode, ufr_stream_t stream);

/* ---- patch cone-of-influence windows --------------------------------------------------------------
 * No counterpart in the reference: patch_attacks/main.py:537-600 keeps only mask*gradient and changes
 * only masked pixels between iterations, so a convolutional encoder prefix (models/FlowNetC.py:96-104,
 * conv1-3) needs its adjoint -- and from the second iteration of an attack() call its forward -- only
 * inside the patch's cone.  The host runs that prefix on a window; these entry points move data
 * between full tensors and the window at an origin held in DEVICE memory (graph-capturable).
 *
 * ufr_cone_chain: the prefix, input to output (kernel/stride/pad per layer), and the layers whose
 *   outputs leave the prefix ("taps", sorted), each with the width of its inexact rim: the number of
 *   cells next to an interior window edge where zero padding of the window differs from the image.
 * win: int[N][8] = {y0, x0, need_h, need_w, ymin, ymax, xmin, xmax} in input pixels. */
#define UFR_MAX_CONE_LAYERS 8
typedef struct ufr_cone_chain {
  int n_layers;
  int kernel[UFR_MAX_CONE_LAYERS], stride[UFR_MAX_CONE_LAYERS], pad[UFR_MAX_CONE_LAYERS];
  int n_taps;
  int tap_layer[UFR_MAX_CONE_LAYERS], tap_margin[UFR_MAX_CONE_LAYERS];
} ufr_cone_chain;
/* Bounding box of mask != 0 per sample (mask [N,C,H,W], mask_bstride elements between samples), its
 * cone at every tap widened by the rim, and a window origin (multiple of the chain's total stride,
 * window inside the image).  *overflow += 1 when the needed extent exceeds win_h x win_w. */
int ufr_cone_window(const float* mask, int N, long mask_bstride, int C, int H, int W,
                    const ufr_cone_chain* chain, int win_h, int win_w, int* win, float* overflow,
                    ufr_stream_t stream);
/* dst[N,C,wh,ww] = src[N,C,Hs,Ws] at win[n % n_win] / level_stride; rim cells (margin, interior
 * edges only) are written as 0.  scatter is the inverse and skips the rim. */
int ufr_window_gather(const float* src, float* dst, const int* win, int n_win, int N, int C, int Hs, int Ws,
                      int wh, int ww, int level_stride, int margin, ufr_stream_t stream);
int ufr_window_scatter(const float* src, float* dst, const int* win, int n_win, int N, int C, int Hd, int Wd,
                       int wh, int ww, int level_stride, int margin, ufr_stream_t stream);

/* Both adjoints of ufr_corr_forward (kernel 1, stride 1, padding 0, fp32) for the cells of a per-sample
 * window only -- win[n] / level_stride, wh x ww cells, as for ufr_window_gather; every other element of
 * grad_input1 / grad_input2 [B,C,H,W] is written as zero.  Behind a windowed prefix only these cells of
 * correlation_cuda_kernel.cu:86-233's result are read. */
int ufr_corr_backward_window(const float* input1, const float* input2, const float* grad_output,
                             float* grad_input1, float* grad_input2, int B, int C, int H, int W, int patch,
                             int dilation_patch, const int* win, int level_stride, int wh, int ww,
                             ufr_stream_t stream);

/* ---- convolution epilogue --------------------------------------------------------------------------
 * models/submodules.py:18-46, :75-82: Conv2d / ConvTranspose2d(bias=True) + LeakyReLU(0.1).  The host runs
 * the convolution without bias; x [B,C,HW] becomes LeakyReLU(x + bias[c]) in place (one pass instead of a
 * broadcast add and an activation).  ufr_leaky_backward: grad_x = y > 0 ? grad_y : slope * grad_y, from the
 * OUTPUT y like torch's in-place form.  Bit-identical to the torch pair. */
int ufr_bias_leaky_forward(float* x, const float* bias, int B, int C, long HW, float slope, ufr_stream_t stream);
int ufr_leaky_backward(const float* y, const float* grad_y, float* grad_x, long total, float slope,
                       ufr_stream_t stream);

/* ---- 2-channel layers ---------------------------------------------------------------------------------
 * predict_flow* = Conv2d(Cin, 2, 3, 1, 1) and upsampled_flow* = ConvTranspose2d(2, 2, 4, 2, 1)
 * (models/FlowNetC.py:43-50, models/submodules.py:85-90; the same layers close FlowNetS/SD and PWC-Net).
 * One HBM pass each instead of implicit-GEMM tiles sized for wide outputs.  Layouts are torch's:
 * x [B,Cin,H,W], conv weight [2,Cin,3,3], transposed-conv weight [2,2,4,4] (in, out, ky, kx), bias [2].
 * The forward convolution splits the channels over workgroups on small images and adds the partial sums
 * in a fixed order: pass a workspace of ufr_conv3x3_c2_workspace_floats() floats (0 = none needed). */
long ufr_conv3x3_c2_workspace_floats(int B, int Cin, int H, int W);
int ufr_conv3x3_c2_forward(const float* x, const float* weight, const float* bias, float* y, float* workspace,
                           int B, int Cin, int H, int W, ufr_stream_t stream);
int ufr_conv3x3_c2_backward_data(const float* grad_y, const float* weight, float* grad_x, int B, int Cin, int H,
                                 int W, ufr_stream_t stream);
/* x [B,2,H,W] -> y [B,2,2H,2W]; bias may be NULL */
int ufr_deconv4x4s2_c2_forward(const float* x, const float* weight, const float* bias, float* y, int B, int H, int W,
                               ufr_stream_t stream);
int ufr_deconv4x4s2_c2_backward_data(const float* grad_y, const float* weight, float* grad_x, int B, int H, int W,
                                     ufr_stream_t stream);

/* ---- patch placement on the device ------------------------------------------------------------------
 * replaces the host round trip of patch_attacks/utils_patch.py:257-358 (circle_transform: scipy zoom /
 * rotate, three canvas-sized np.zeros + H2D per sample) and patch_attacks/main.py:408-461 (D2H, crop,
 * scipy zoom back).  The patch state is float64 [C,h,w] like the reference's numpy arrays; the host keeps
 * drawing np.random in the reference's order and passes the numbers in.
 *
 * ufr_affine_resample_f64: dst[c,y,x] = src[c] sampled at (m00*y + m01*x + off0, m10*y + m11*x + off1)
 *   with scipy.ndimage semantics for mode='constant', cval=0: order 1 (linear) or 0 (floor(cc+0.5));
 *   a coordinate below 0 or above n-1 gives 0.  zoom: m00 = (Hs-1)/(Hd-1), m11 = (Ws-1)/(Wd-1);
 *   rotate(reshape=False): the 2x2 matrix and offset of scipy.ndimage.rotate.
 * ufr_patch_place: zero the three float32 canvases [C,H,W] and paste patch / mask / patch_init at (y, x).
 * ufr_patch_crop_f64: dst = float64(canvas[crop] * factor[crop]) (float32 product; factor may be NULL). */
int ufr_affine_resample_f64(const double* src, double* dst, int C, int Hs, int Ws, int Hd, int Wd,
                            double m00, double m01, double m10, double m11, double off0, double off1,
                            int order, ufr_stream_t stream);
int ufr_patch_place(const double* patch, const double* mask, const double* init, int C, int h, int w,
                    float* canvas_patch, float* canvas_mask, float* canvas_init, int H, int W, int y, int x,
                    ufr_stream_t stream);
int ufr_patch_crop_f64(const float* canvas, const float* factor, double* dst, int C, int H, int W, int y,
                       int x, int h, int w, ufr_stream_t stream);

/* ---- loaders: image resize and tensor conversion on the device ----------------------------------------
 * replaces dataset_utils/data_utils.py:26-32 (imresize = PIL.Image.resize(BILINEAR) on uint8),
 * dataset_utils/custom_transforms.py:47-57 (ArrayToTensor), :60-122 (flip / crop / Scale) and the
 * arithmetic of flowutils/flow_io.py:104-127 (flow_read_png).  Images are uint8 [H,W,C] (C = 1, 3, 4).
 * bounds[2*i] = first source index, bounds[2*i+1] = tap count, kk[i*ksize + t] = 22-bit fixed-point
 * coefficient of output index i -- Pillow's precompute_coeffs / normalize_coeffs_8bpc tables, computed by
 * the host (input_pipeline.resize_tables).  Each pass rounds to uint8 like Pillow's two-pass resampler;
 * `flip` reads the source row mirrored (RandomHorizontalFlip happens before the resize). */
int ufr_resample_u8_horizontal(const unsigned char* src, unsigned char* dst, int H, int Ws, int Wd, int C, int flip,
                               const int* bounds, const int* kk, int ksize, ufr_stream_t stream);
int ufr_resample_u8_vertical(const unsigned char* src, unsigned char* dst, int Hs, int Hd, int W, int C,
                             const int* bounds, const int* kk, int ksize, ufr_stream_t stream);
/* dst[c,y,x] (float32, [C,crop_h,crop_w]) = float(src[crop_y+y, crop_x+x, c]) / divisor */
int ufr_u8_to_tensor(const unsigned char* src, float* dst, int H, int W, int C, int crop_y, int crop_x, int crop_h,
                     int crop_w, float divisor, ufr_stream_t stream);
/* KITTI flow PNG samples, uint16 [H,W,3] -> float32 [3,H,W]: u = (x-2^15)/64, v likewise, valid as is */
int ufr_kitti_flow_decode(const unsigned short* src, float* dst, int H, int W, ufr_stream_t stream);

/* HOST function (no GPU work): PNG scanline reconstruction (filter types 0-4) for the 16-bit KITTI flow maps
 * that the reference reads with PyPNG (flowutils/flow_io.py:104-127).  data = rows x (1 + stride) inflated
 * bytes, out = rows x stride, bpp = bytes per complete pixel. */
int ufr_host_png_unfilter(const unsigned char* data, unsigned char* out, int rows, int stride, int bpp);

/* ---- native FlowNetC head: implicit-GEMM convolutions on bf16 split planes (csrc/igemm.hip) --------------------
 * replaces, for the attack's frozen networks, the Conv2d / ConvTranspose2d blocks of models/FlowNetC.py:22-50
 * (models/submodules.py:18-46 `conv`, :75-82 `deconv`) that torch runs on MIOpen -- forward AND data gradient -- without
 * the activations leaving the device layout between layers.
 *   activation planes: bf16 [3][chunks][M][32], M = B*H*W pixels, 32 channels per chunk (v = p0 + p1 + p2 exactly);
 *   weights:           bf16 [3][taps*KC][Npad][32] per phase (K tiles ordered as `k_order` says), pre-split; Npad a multiple of 64 (128-column tiles when it is
 *                      a multiple of 128, 64-column tiles otherwise);
 *   gradient sums:     f32 [chunks][M][32].
 * The tile rows are the cells (b, y, x) of a row grid [B,Hr,Wr] (x offset per sample by row_x0[b*row_x0_stride] /
 * row_x0_div when row_x0 != NULL: a column band around the patch); tap t of phase z reads input pixel
 * (y*in_sy + dy[t], x*in_sx + dx[t]) -- zero outside [0,Hi) x [0,Wi), or outside the per-sample input band
 * [in_x0[b*in_x0_stride]/in_x0_div, + in_xw) when in_x0 != NULL -- and the result lands on output pixel
 * (y*out_sy + oy0, x*out_sx + ox0) of the [B,Ho,Wo] grid.
 * Epilogue: act = 1: LeakyReLU(acc + add + bias) (forward; `add` normally NULL); act = 0: (acc + add) * LeakyReLU'(mask) with `add` an fp32
 * chunk-major tensor and `mask` plane 0 of the activation this gradient belongs to (either may be NULL).  The result
 * goes to out_planes (at chunk out_chunk0 of a buffer whose planes are out_plane_stride elements apart) and / or
 * out_f32.  splitk > 1: the phase with the most taps is cut into `splitk` slices of K, the others into proportionally fewer
 * slices of the same length; fp32 slabs in `ws` (at most [nphase*splitk][B*Hr*Wr][Npad]), added in a fixed order by a
 * second kernel (no atomics: bit-reproducible). */
#define UFR_IGEMM_MAX_TAPS 25
typedef struct {
  int ntaps, oy0, ox0;
  long w_off;                                  /* elements from the start of a weight plane */
  signed char dy[UFR_IGEMM_MAX_TAPS], dx[UFR_IGEMM_MAX_TAPS];
} ufr_igemm_phase;
typedef struct {
  const void* x; long x_plane_stride; int in_chunk0, KC;
  int B, Hi, Wi, in_sy, in_sx;
  const int* in_x0; int in_x0_stride, in_x0_div, in_xw;
  const void* w; long w_plane_stride; int Npad, N;
  int Hr, Wr;
  const int* row_x0; int row_x0_stride, row_x0_div;
  int Ho, Wo, out_sy, out_sx;
  int nphase;
  ufr_igemm_phase phase[4];
  int act; const float* bias; float slope;
  const float* add; int add_chunk0;
  const void* mask; int mask_chunk0;
  void* out_planes; long out_plane_stride; int out_chunk0;
  float* out_f32; int out_f32_chunk0;
  float* tail; int tail_n0, tail_accumulate;   /* optional: output columns >= tail_n0 (a multiple of 32) leave as RAW sums into this fp32
                                                  chunk-major tensor [(N - tail_n0) / 32 chunks][B*Ho*Wo][32] instead of passing the epilogue:
                                                  a later layer's partial sum over the same input chunks, which that layer's own launch
                                                  takes back as `add` (with act = 1: LeakyReLU(acc + add + bias)); tail_accumulate != 0
                                                  adds the sums onto the tensor (a gradient sum with other contributors) */
  int splitk; float* ws;
  int products;                                /* 6: the float32-accurate six-product form (the default everywhere); (ABI 9) 3 / 1: the leading three
                                                  products a0b0 + a0b1 + a1b0 (two planes per operand, ~16 significand bits) / the single product
                                                  a0b0 (one bf16 plane per operand: what an autocast to bfloat16 computes) -- RAFT's opt-in reduced
                                                  precision (models/utils_model.py:51 `mixed_precision`, models/raft/raft.py:140,168,195), served
                                                  by the single-stage, 64 x 128 and ping-pong tile forms (variants 5 / 7 / 8 run as 6 / 2) */
  int variant;                                 /* kernel form: 0 / 2 = single-stage LDS-DMA tiles (128 x 128; 128 x 64 when Npad % 128),
                                                  4 = 64 x 128 tiles, 5 = pipelined 128 x 128 (register-held fragments),
                                                  6 = ping-pong: 256 x 128 tiles, two wave groups half a step apart,
                                                  7 = 6 with horizontal runs of taps staged once (256 x 128 or 256 x 64 tiles; launches it
                                                      does not cover -- stride 2, fewer than 22 columns -- run as 6 / 2) */
  int k_order;                                 /* order of the K tiles in the weight image of a phase: 0 = [taps][KC] (tap-major),
                                                  1 = [KC][taps] (the taps of one channel chunk back to back: L2 reuse of the pixels) */
  float* out_rowmajor; long out_ld;            /* optional (ABI 5): the epilogue's result also / instead as ROW-MAJOR fp32 [B*Ho*Wo][out_ld],
                                                  element (pixel, n) -- RAFT's all-pairs volume corr[p][q] = <fmap1[p], fmap2[q]>
                                                  (models/raft/corr.py:57-64) is a 1x1 launch whose "weights" are fmap2's planes; N % 8 == 0 */
  int planes_chunks, f32_first_chunk;          /* (ABI 6) with BOTH out_planes and out_f32: only output chunks < planes_chunks reach the planes
                                                  (0 = all) and only chunks >= f32_first_chunk the fp32 tensor -- conv3_1's data gradient feeds
                                                  conv_redir's (planes, chunk 0) and the correlation's adjoint (fp32, chunks 1 ..): 4.4 instead
                                                  of 10 bytes per element leave the tile, and an epilogue is a chip-wide write burst */
  int no_reduce;                               /* (ABI 6) split-K launch of one phase: stop after the slab kernel; `ws` then holds the raw sums
                                                  [splitk][B*Hr*Wr][Npad] for a consumer that adds them itself in ascending order and then
                                                  the bias (ufr_gru_gates_cm_forward_slabs / ufr_gru_blend_cm_forward_slabs): the epilogue
                                                  fields are not used */
  int* tickets;                                /* (ABI 9) split-K without the second launch: one zero-initialised int per (phase, row tile, column tile)
                                                  -- at least nphase * ceil(B*Hr*Wr / 64) * (Npad / 64) of them, owned by THIS launch (two launches
                                                  in flight on two streams must not share them).  Every slice's workgroup writes its slab, publishes
                                                  it (agent-scope release) and draws a ticket; the workgroup that draws the last one of its tile adds
                                                  the tile's slabs in ascending slice order -- the reduce kernel's arithmetic bit for bit -- runs the
                                                  epilogue and leaves the counter at zero for the next launch.  NULL: the reduce kernel follows. */
} ufr_igemm_desc;
int ufr_igemm(const ufr_igemm_desc* d, ufr_stream_t stream);
/* (ABI 9) Launches since load that asked for variant 8 (direct 3 x 3) or 7 (tap reuse) and ran as a plain tile form because their
 * geometry is not covered: a tuning-table entry that reached a launch it was not swept for shows here (speed only, never results). */
int ufr_igemm_variant_fallbacks(void);
/* Measurement aid (tools/measure_clock.py): with a device buffer of 8 x capacity_workgroups uint64 set, every workgroup of the
 * ping-pong kernel (variant 6) records {s_memtime at entry, at exit, s_memrealtime at entry, at exit, s_memtime before and after
 * the K loop, HW_ID, K steps}: core cycles against the constant 100 MHz counter = the clock the kernel really ran at, and the
 * split of a workgroup's life into set-up / K loop / epilogue.  (NULL, 0) switches it off (the default; synchronous call). */
int ufr_igemm_clock_probe(unsigned long long* buf, int capacity_workgroups);
/* Layout passes at the engine's edges.  ufr_nchw_to_planes: planes[chunk0 + c/32] = split(leaky(scale * x[B,C,H,W]))
 * (scale = 1, slope = 1: a plain conversion).  ufr_chunks_to_nchw: out[B,C,H,W] = scale * leaky'(mask) * v, v from planes
 * (p0 + p1 + p2) or from an fp32 chunk-major tensor (exactly one of `planes`, `f32`).  ufr_grad_finalize: gradient planes =
 * split(g * leaky'(mask)) for `chunks` chunks of M pixels. */
/* A row-major float32 matrix src [rows][ld] (first `cols` columns) as the operand of a GEMM reduced over its COLUMNS:
 * planes[chunk0 + k/32][m][k%32] = split(scale * src[m][k]) in a buffer of `plane_rows` rows per chunk (round 5: the adjoint of RAFT's
 * all-pairs correlation, models/raft/corr.py:57-64 -- the volume's gradient and the feature maps as igemm operands; the reduction
 * over the ROWS of the same matrix is ufr_nchw_to_planes with channels = rows). */
int ufr_rowmajor_to_planes(const float* src, long ld, long rows, int cols, float scale, void* planes, long plane_stride, int chunk0,
                           long plane_rows, ufr_stream_t stream);
int ufr_nchw_to_planes(const float* x, void* planes, long plane_stride, int chunk0, int B, int C, int H, int W,
                       float scale, float slope, const float* bias, ufr_stream_t stream);
/* The windowed prefix's results patched into cached full-frame features that live in the plane layout: as
 * ufr_window_scatter (same origins, clamping and rim rule), destination = chunks [chunk0, ...) of a planes buffer. */
int ufr_window_scatter_planes(const float* src, void* planes, long plane_stride, int chunk0, const int* win, int n_win,
                              int N, int C, int Hd, int Wd, int wh, int ww, int level_stride, int margin,
                              ufr_stream_t stream);
int ufr_chunks_to_nchw(const void* planes, long plane_stride, const float* f32, int chunk0, const void* mask,
                       int mask_chunk0, float* out, int B, int C, int H, int W, float scale, float slope,
                       ufr_stream_t stream);
/* A concatenation of up to four NCHW float32 tensors <-> `chunks` chunks of the plane layout (PWC-Net's stage input
 * x = cat(corr, up_flow, up_feat | c1), models/PWCNet.py:287): member s covers buffer channels [dst_channel0[s], + channels[s]),
 * ascending and disjoint; uncovered channels are zero.  ufr_chunks_to_nchw_cat is the adjoint from a float32 chunk-major
 * gradient sum; member 0 may pass through its activation: g * (act0 > 0 ? pos0 : neg0) with act0 the member's NCHW activation. */
int ufr_nchw_cat_to_planes(const float* const* srcs, const int* channels, const int* dst_channel0, int nseg, void* planes,
                           long plane_stride, int chunk0, int chunks, int B, int H, int W, ufr_stream_t stream);
int ufr_chunks_to_nchw_cat(const float* g, int chunk0, int chunks, float* const* dsts, const int* channels, const int* src_channel0,
                           int nseg, const float* act0, float pos0, float neg0, int B, int H, int W, ufr_stream_t stream);
int ufr_grad_finalize(const float* g, int g_chunk0, const void* mask, int mask_chunk0, void* out, long out_plane_stride,
                      int out_chunk0, long M, int chunks, float slope, ufr_stream_t stream);

/* FlowNetC's conv1 = Conv2d(3, 64, 7, 2, 3) + bias + LeakyReLU (models/FlowNetC.py:100-104, submodules.py:18-46) straight from the
 * RAW frames (frames_a [Ba,3,H,W], then frames_b [Bb,3,H,W] or NULL) to conv1's activation planes [3][2][(Ba+Bb)*H/2*W/2][32] at
 * chunk out_chunk0: the float64 mean subtraction of normalize_correctly (FlowNetC.py:73-79), the zero padding, the im2col
 * (in LDS), the six-product MFMA and the epilogue in ONE kernel (csrc/conv1_direct.hip).  wimg = bf16 [3 planes][7][64][32],
 * k = (kx >> 1) * 8 + (kx & 1) * 4 + c of kernel row ky (igemm.conv1_direct_weights).  H, W even. */
int ufr_conv1_direct(const float* frames_a, const float* frames_b, int Ba, int Bb, int H, int W, const double* mean,
                     const void* wimg, const float* bias, float slope, void* out_planes, long plane_stride, int out_chunk0,
                     ufr_stream_t stream);
/* Raw frames [Ba(+Bb), 3, H, W] -> the PACKED planes conv1 = Conv2d(3, 64, 7, 2, 3) (models/FlowNetC.py:22) reads as an
 * 8-tap ufr_igemm launch: planes [3][1][(Ba + Bb) * (H/2 + 3) * (W/2 + 2)][32], channel j*12 + (c*2 + p)*2 + q of packed
 * pixel (yp, xp) = frame[c, 2 (yp - 2) + p, 2 (xp - 2 + j) + q] - mean[c] in float64 (normalize_correctly, FlowNetC.py:73-79),
 * zero outside the frame and for channels 24..31. */
int ufr_conv1_pack_planes(const float* frames_a, const float* frames_b, void* planes, long plane_stride, int Ba, int Bb, int H,
                          int W, const double* mean, ufr_stream_t stream);
/* Adjoint of the packing: d loss / d frames [N, 3, H, W] from the float32 gradient sum of the packed planes
 * [1][N * (H/2 + 3) * (W/2 + 2)][32] (the output of conv1's data-gradient launch). */
int ufr_conv1_unpack_grad(const float* G, float* grad_frames, int N, int H, int W, ufr_stream_t stream);
/* 2x2 pixel-unshuffle of x [N,C,2H,2W] into planes [N,H,W] with channel (c*2 + p)*2 + q = x[c, 2y+p, 2x+q] (a stride-2 7x7
 * convolution becomes a stride-1 4x4 launch over 4C channels: FlowNetS's 12-channel stem, models/flownet2/FlowNetS.py:24), and the
 * adjoint from the float32 gradient sum G [chunks][N*H*W][32] back to [N,C,2H,2W]. */
int ufr_unshuffle_pack_planes(const float* x, void* planes, long plane_stride, int N, int C, int H, int W, ufr_stream_t stream);
int ufr_unshuffle_unpack_grad(const float* G, float* grad_x, int N, int C, int H, int W, ufr_stream_t stream);
/* NCHW float32 gradient x LeakyReLU'(NCHW activation) -> the engine's gradient planes (three bf16 planes, chunk-major), one pass. */
int ufr_nchw_grad_to_planes(const float* grad, const float* act, void* planes, long plane_stride, int chunk0, int B, int C, int H,
                            int W, float slope, ufr_stream_t stream);
/* ufr_window_gather for a chunk-major float32 tensor [chunks][N*Hs*Ws][32] -> [chunks][N_dst*wh*ww][32] (images >= N untouched). */
int ufr_window_gather_chunks(const float* src, float* dst, const int* win, int n_win, int N, int N_dst, int chunks, int Hs, int Ws,
                             int wh, int ww, int level_stride, int margin, ufr_stream_t stream);

/* models/FlowNetC.py:73-79, :93-94 (normalize_correctly): out[n] = (float)((double)frame - mean[c]) for the Ba first frames and
 * the Bb second frames (frames_b may be NULL when Bb == 0) as one stack [Ba + Bb, C, H, W]; mean = C float64 values on the device. */
int ufr_normalize_frames(const float* frames_a, const float* frames_b, float* out, int Ba, int Bb, int C, int H, int W,
                         const double* mean, ufr_stream_t stream);

/* The incremental form of ufr_corr_forward_planes: recomputes only the columns within the correlation's reach (20 cells) of
 * each sample's window (win[b*8 + 1] / level_stride, `win_cells` wide, at most 23) -- all rows, all 441 displacements; the
 * other columns of `out_planes` keep their values.  Exact when the features changed inside the window only. */
int ufr_corr_forward_planes_window(const void* f1_planes, const void* f2_planes, long in_plane_stride, void* out_planes,
                                   long out_plane_stride, int out_chunk0, int B, int C, int H, int W, int patch,
                                   int dilation_patch, float scale, float slope, const int* win, int level_stride,
                                   int win_cells, ufr_stream_t stream);

/* Both adjoints of FlowNetC's cost volume (correlation_cuda_kernel.cu:86-233; patch 21, dilation_patch 2, 256 channels) on
 * the cells of the prefix window, on the matrix cores (csrc/correlation_window_mfma.hip), fused with everything around it:
 * G = the engine's chunk-major float32 gradient sum of conv3_1's input (the cost volume's channels start at chunk
 * g_chunk0; g_scale = 1 / C of models/submodules.py:124-138), G_redir (nullable) = conv_redir's input gradient, added to
 * the first frames' rows; grad_window [2B, C, wh, ww] = d/d conv3a (rows 0..B-1) and d/d conv3b, window-sized, with the
 * inexact rim of `margin` cells zeroed exactly as ufr_window_gather does.  win / level_stride as ufr_corr_backward_window. */
int ufr_corr_backward_window_fused(const float* f1, const float* f2, const float* G, int g_chunk0, float g_scale,
                                   const float* G_redir, float* grad_window, int B, int C, int H, int W, int patch,
                                   int dilation_patch, const int* win, int level_stride, int wh, int ww, int margin,
                                   ufr_stream_t stream);

/* FlowNetC's cost volume on the matrix cores, planes in, planes out (csrc/correlation_planes.hip): replaces
 * correlation_cuda_forward_kernel (correlation_cuda_kernel.cu:21-83) + `correlate`'s / C (models/submodules.py:124-138) +
 * LeakyReLU (FlowNetC.py:139) for kernel 1, patch 21, dilation_patch 2, 256 channels.  f1 / f2: planes [3][8][B*H*W][32];
 * out channel d = dy*21 + dx of pixel (b, y, x) = leaky(scale * sum_c f1[c,y,x] * f2[c, y + 2(dy-10), x + 2(dx-10)]) lands in
 * chunk out_chunk0 + d/32 of `out_planes`.  Other configurations: UFR_EUNSUPPORTED (use ufr_corr_forward). */
int ufr_corr_forward_planes(const void* f1_planes, const void* f2_planes, long in_plane_stride, void* out_planes,
                            long out_plane_stride, int out_chunk0, int B, int C, int H, int W, int patch,
                            int dilation_patch, float scale, float slope, ufr_stream_t stream);
/* The 2-channel layers of the refinement on the engine's layout (csrc/engine_small.hip):
 * predict_flow* = Conv2d(Cin,2,3,1,1) (models/FlowNetC.py:43-47) reads the chunks [chunk0, chunk0+chunks) of a planes
 * buffer and writes flow [B,2,H,W] fp32; weights repacked wpk[chunk][tap][out][32].  Its data gradient writes (or adds
 * to) the fp32 gradient sum G [.. chunks][B*H*W][32].  upsampled_flow* = ConvTranspose2d(2,2,4,2,1) (:48-50) writes its
 * two channels (+ 30 zeros) into chunk `chunk` of the concatenation at twice the resolution; its data gradient reads
 * channels 0-1 of that chunk of the fp32 gradient sum. */
/* ABI 7: every entry of this group that walks a chunk range knows the extent of what it walks and REFUSES (UFR_EINVAL +
 * ufr_last_error) a range that leaves it, instead of reading or writing past the end: a planes operand holds
 * plane_stride / (pixels * 32) chunks per plane, so chunk0 + chunks must fit in plane_stride; `w_chunks` = the chunks the packed
 * weights hold (chunks <= w_chunks: the weights are indexed by the position inside the range); `g_chunks` = the chunks of the
 * float32 gradient sum G (chunk0 + chunks <= g_chunks, out_chunk < g_chunks). */
int ufr_flow_head_planes_forward(const void* planes, long plane_stride, int chunk0, int chunks, const float* wpk, int w_chunks,
                                 const float* bias, float* out, int B, int H, int W, ufr_stream_t stream);
/* The same convolution on the matrix cores: a per-pixel GEMM T[p, 2k + o] = sum_c x[p, c] w[o, c, k] (float32 = six bf16
 * products, as ufr_igemm) whose A operand is the plane layout as it lies in HBM, then the 9-tap gather through LDS.
 * wmf: bf16 [chunks][3][2][16][32] (plane p of w[o][32 ch + c][k] at n = 2k + o, zeros for n >= 18). */
int ufr_flow_head_planes_forward_mfma(const void* planes, long plane_stride, int chunk0, int chunks, const void* wmf, int w_chunks,
                                      const float* bias, float* out, int B, int H, int W, ufr_stream_t stream);
/* The two flow channels of a ConvTranspose2d(Cin, Cout, 4, 2, 1) data gradient (models/FlowNetC.py:162-183: the last two
 * input channels of deconvK are the upsampled flow), same per-pixel GEMM + gather: grad_planes = the masked output gradient
 * on the fine grid [2H, 2W] (chunks chunk0 .. chunk0 + chunks), wmf as above with n = 2 (4 ky + kx) + o for the weights
 * w[Cin - 2 + o][c][ky][kx]; writes lanes 0-1 of chunk `out_chunk` of the coarse grid's [H, W] float32 gradient sum G. */
int ufr_deconv_flow_tail_backward_mfma(const void* grad_planes, long plane_stride, int chunk0, int chunks, const void* wmf,
                                       int w_chunks, float* G, int g_chunks, int out_chunk, int B, int H, int W, ufr_stream_t stream);
int ufr_flow_head_planes_backward(const float* grad_y, const float* wpk, int w_chunks, float* G, int g_chunks, int chunk0, int chunks,
                                  int B, int H, int W, int accumulate, ufr_stream_t stream);
/* The same with the finalisation of one segment fused (round 4): for the chunks [fin_chunk0, fin_chunk0 + fin_chunks) of the
 * tensor the completed sum x LeakyReLU'(mask_planes plane 0; NULL = linear) also leaves as the three gradient planes `out_planes`
 * (same chunk positions) -- what ufr_grad_finalize did in a launch of its own (FlowNetC's refinement, models/FlowNetC.py:162-183:
 * predict_flowK's adjoint is the last contributor to deconvK's output gradient). */
int ufr_flow_head_planes_backward_finalize(const float* grad_y, const float* wpk, int w_chunks, float* G, int g_chunks, int chunk0,
                                           int chunks, int B, int H, int W, int accumulate, const void* mask_planes, void* out_planes,
                                           long out_plane_stride, int fin_chunk0, int fin_chunks, float slope, ufr_stream_t stream);
/* PWC-Net's `upfeat*` = ConvTranspose2d(C, 2, 4, 2, 1) (models/PWCNet.py:115-143, used at :284,:299,:314,:329) on the engine's planes:
 * forward from `chunks` chunks of the COARSE [B,H,W] planes to out [B,2,2H,2W] (NCHW fp32, + bias) on the matrix cores
 * (wmf: bf16 [chunks][3][2][16][32] with n = 2 (4 ky + kx) + o); backward from grad_y [B,2,2H,2W] into the coarse gradient
 * sum G[chunk0 ..][B*H*W][32] (wpk: fp32 [chunks][16][2][32]; accumulate != 0 adds). */
int ufr_upfeat_planes_forward_mfma(const void* planes, long plane_stride, int chunk0, int chunks, const void* wmf, int w_chunks,
                                   const float* bias, float* out, int B, int H, int W, ufr_stream_t stream);
int ufr_upfeat_planes_backward(const float* grad_y, const float* wpk, int w_chunks, float* G, int g_chunks, int chunk0, int chunks,
                               int B, int H, int W, int accumulate, ufr_stream_t stream);
int ufr_flow_up_planes_forward(const float* x, const float* w, const float* bias, void* planes, long plane_stride,
                               int chunk, int B, int H, int W, ufr_stream_t stream);
int ufr_flow_up_planes_backward(const float* G, int chunk, const float* w, float* grad_x, int B, int H, int W,
                                int accumulate /* ABI 6: grad_x += (1) or = (0) */, ufr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UFR_HIP_H_ */
