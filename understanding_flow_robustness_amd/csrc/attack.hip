// attack.hip -- elementwise stages of the patch-attack inner loop (patch_attacks/main.py:523-613)
// for gfx950.  All three kernels are HBM-streaming: 16-byte accesses when the canvas size allows,
// one pass over each tensor, and the whole post-backward sequence of the reference
//   grad sum -> *0.5*lr -> clamp(+-2) -> patch -= ... -> re-paste both frames -> clamp[0,1]
// (main.py:575-600: 9 elementwise torch kernels and 7 temporaries) is ONE kernel here.
#include "ufr_common.h"

namespace {

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// main.py:537-542 / :585-600.  (1-m)*img and m*patch are rounded separately, then added -- the
// reference's torch.mul + torch.mul + add (no fma).
__global__ void paste_kernel(const float* __restrict__ tgt, const float* __restrict__ ref,
                             const float* __restrict__ patch, const float* __restrict__ mask,
                             float* __restrict__ adv_tgt, float* __restrict__ adv_ref, long total,
                             long CHW, long pstride, long mstride, int do_clamp, float lo, float hi) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long b = i / CHW, e = i - b * CHW;
    const float m = mask[b * mstride + e], pv = patch[b * pstride + e];
    const float mp = m * pv, om = 1.0f - m;
    float a = om * tgt[i] + mp;
    float r = om * ref[i] + mp;
    if (do_clamp) { a = clampf(a, lo, hi); r = clampf(r, lo, hi); }
    adv_tgt[i] = a;
    adv_ref[i] = r;
  }
}

// main.py:575-600 fused (grad sum, step, clamp, patch update, re-paste, clamp: 9 elementwise torch kernels in the
// reference): exactly the reference's B=1 arithmetic, per sample when every sample has its own canvas patch.
__global__ void update_private_kernel(const float* __restrict__ tgt, const float* __restrict__ ref,
                                      const float* __restrict__ g_tgt, const float* __restrict__ g_ref,
                                      float* __restrict__ patch, const float* __restrict__ mask,
                                      float* __restrict__ adv_tgt, float* __restrict__ adv_ref,
                                      long total, long CHW, long pstride, long mstride, float step,
                                      float bound, float lo, float hi, const float* __restrict__ gate) {
  if (gate != nullptr && gate[0] != 0.f) return;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long b = i / CHW, e = i - b * CHW;
    const float gs = g_tgt[i] + g_ref[i];
    const float pv = patch[b * pstride + e] - clampf(step * gs, -bound, bound);
    patch[b * pstride + e] = pv;
    const float m = mask[b * mstride + e];
    const float mp = m * pv, om = 1.0f - m;
    adv_tgt[i] = clampf(om * tgt[i] + mp, lo, hi);
    adv_ref[i] = clampf(om * ref[i] + mp, lo, hi);
  }
}

// Fixed-order second stage of the loss reductions (flow_loss_kernel, universal.hip's flow_loss_ex_kernel): one wave
// adds the workgroups' partial sums -- lane l takes partials l, l+64, ... in order, then a fixed shuffle tree -- so the
// scalar that drives `while loss > 0.1` (main.py:546) is bit-reproducible from run to run (no float atomics).
__global__ void loss_finalize_kernel(const float* __restrict__ partials, int n, float* __restrict__ loss) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += partials[i];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (threadIdx.x == 0) *loss += s;
}

// Loss + its gradient wrt the flow in one pass (replaces the autograd graph of main.py:557-566).
//  kind 0: mean(1 - cos(f,t)),  cos = <f,t> / max(|f|*|t|, 1e-8)   (torch cosine_similarity, eps 1e-8)
//          d/df = -( t/(|f||t|) - cos * f/|f|^2 ) / Npix
//  kind 1: mean(sqrt(|f-t|^2 + 1e-8)),  d/df = (f-t)/sqrt(.) / Npix
// Block-level reduction in LDS; every workgroup stores its partial sum, loss_finalize_kernel adds them in a fixed order.
__global__ void flow_loss_kernel(const float* __restrict__ flow, const float* __restrict__ target,
                                 float* __restrict__ gflow, float* __restrict__ partials, int B, long HW,
                                 int kind, float weight) {
  __shared__ float red[256 / 64];
  const long npix = (long)B * HW;
  const float invn = weight / (float)npix;
  float part = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, p = i - b * HW;
    const size_t o0 = ((size_t)b * 2) * HW + p, o1 = o0 + HW;
    const float fu = flow[o0], fv = flow[o1], tu = target[o0], tv = target[o1];
    if (kind == 0) {
      const float dot = fu * tu + fv * tv;
      const float nf2 = fu * fu + fv * fv, nt2 = tu * tu + tv * tv;
      const float den = fmaxf(sqrtf(nf2 * nt2), 1e-8f);
      const float c = dot / den;
      part += 1.0f - c;
      float gu = 0.f, gv = 0.f;
      if (sqrtf(nf2 * nt2) > 1e-8f) {
        gu = -(tu / den - c * fu / nf2);
        gv = -(tv / den - c * fv / nf2);
      } else {
        gu = -(tu / den);
        gv = -(tv / den);
      }
      gflow[o0] = gu * invn;
      gflow[o1] = gv * invn;
    } else {
      const float du = fu - tu, dv = fv - tv;
      const float s = sqrtf(du * du + dv * dv + 1e-8f);
      part += s;
      gflow[o0] = du / s * invn;
      gflow[o1] = dv / s * invn;
    }
  }
  // wave reduce (64 lanes), then across the 4 waves of the workgroup
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) red[wv] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k];
    partials[blockIdx.x] = s * invn;
  }
}

// ---- shared patch in PATCH coordinates (patch_attacks/main.py:396-424 crops the canvas at (ry, rx) back
// to patch_shape after every sample) -------------------------------------------------------------------------------
// B pairs show ONE patch P[3,ph,pw] (mask Mp[3,ph,pw]) at per-pair origins (oy_b, ox_b).
//   crop:   rows[g][c,i,j] = [Mp != 0] * sum_{b in group g, ascending} (g_tgt + g_ref)[b, c, oy_b+i, ox_b+j]
//           rows[g][3*ph*pw] = this rank's loss (row 0) or 0: the scalar travels with the gradient
//   apply:  G = sum_r rows[r] (ascending r: the fixed order that keeps every rank's patch bit-identical),
//           P -= clamp(step*G, +-bound); *loss = sum_r rows[r][3*ph*pw]
//   paste:  adv_b = clamp((1-M_b)*img_b + M_b*place(P, origin_b)), M_b = place(Mp, origin_b); optionally writes M_b
__global__ void patch_grad_crop_kernel(const float* __restrict__ g_tgt, const float* __restrict__ g_ref,
                                       const float* __restrict__ mask_p, const int* __restrict__ origins,
                                       const float* __restrict__ loss_local, float* __restrict__ rows, int B, int H,
                                       int W, int ph, int pw, int groups) {
  const int n = 3 * ph * pw, per = B / groups;
  const long HW = (long)H * W;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e <= n; e += gridDim.x * blockDim.x) {
    if (e == n) {
      for (int g = 0; g < groups; ++g) rows[(long)g * (n + 1) + n] = (g == 0) ? *loss_local : 0.f;
      continue;
    }
    const int c = e / (ph * pw), r = e - c * ph * pw, i = r / pw, j = r - i * pw;
    const bool shown = mask_p[e] != 0.f;
    for (int g = 0; g < groups; ++g) {
      float s = 0.f;
      if (shown)
        for (int b = g * per; b < (g + 1) * per; ++b) {
          // a placement that leaves the frame (device-resident origins are not validated on the host) shows only the
          // part inside it, exactly as paste_placed_kernel clips it: pixels outside contribute nothing
          const int y = origins[2 * b] + i, x = origins[2 * b + 1] + j;
          if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) continue;
          const long o = ((long)b * 3 + c) * HW + (long)y * W + x;
          s += g_tgt[o] + g_ref[o];
        }
      rows[(long)g * (n + 1) + e] = s;
    }
  }
}

// The same crop straight from the WINDOW gradients of the windowed prefix (patch_attack.py `_backward_cone`): gxw [2B, 3, wh, ww] holds
// d loss / d (first frames | second frames) on each pair's window (origin win[b] = {y0, x0, ..}, clamped into the frame exactly as
// window_copy_kernel clamps it); outside the window the image gradient is zero.  Replaces two window -> canvas scatters (and the two
// canvas zero fills per call behind them) + the canvas crop.
__global__ void patch_grad_crop_window_kernel(const float* __restrict__ gxw, const int* __restrict__ win,
                                              const float* __restrict__ mask_p, const int* __restrict__ origins,
                                              const float* __restrict__ loss_local, float* __restrict__ rows, int B, int H, int W,
                                              int wh, int ww, int ph, int pw, int groups) {
  const int n = 3 * ph * pw, per = B / groups;
  const long wplane = (long)wh * ww;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e <= n; e += gridDim.x * blockDim.x) {
    if (e == n) {
      for (int g = 0; g < groups; ++g) rows[(long)g * (n + 1) + n] = (g == 0) ? *loss_local : 0.f;
      continue;
    }
    const int c = e / (ph * pw), r = e - c * ph * pw, i = r / pw, j = r - i * pw;
    const bool shown = mask_p[e] != 0.f;
    for (int g = 0; g < groups; ++g) {
      float s = 0.f;
      if (shown)
        for (int b = g * per; b < (g + 1) * per; ++b) {
          const int y = origins[2 * b] + i, x = origins[2 * b + 1] + j;
          if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) continue;
          const int y0 = min(max(win[b * 8], 0), H - wh), x0 = min(max(win[b * 8 + 1], 0), W - ww);
          const int wy = y - y0, wx = x - x0;
          if ((unsigned)wy >= (unsigned)wh || (unsigned)wx >= (unsigned)ww) continue;     // zero gradient outside the window
          const long o = (long)c * wplane + (long)wy * ww + wx;
          s += gxw[(long)b * 3 * wplane + o] + gxw[(long)(B + b) * 3 * wplane + o];
        }
      rows[(long)g * (n + 1) + e] = s;
    }
  }
}

__global__ void patch_apply_kernel(const float* __restrict__ rows, int n_rows, float* __restrict__ patch_p,
                                   float* __restrict__ loss, int n, float step, float bound,
                                   const float* __restrict__ gate) {
  const bool stopped = gate != nullptr && gate[0] != 0.f;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e <= n; e += gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int r = 0; r < n_rows; ++r) s += rows[(long)r * (n + 1) + e];
    if (e == n) *loss = s;                                    // the gate reads it even for a void iteration
    else if (!stopped) patch_p[e] -= clampf(step * s, -bound, bound);
  }
}

__global__ void paste_placed_kernel(const float* __restrict__ tgt, const float* __restrict__ ref,
                                    const float* __restrict__ patch_p, const float* __restrict__ mask_p,
                                    const int* __restrict__ origins, float* __restrict__ adv_tgt,
                                    float* __restrict__ adv_ref, float* __restrict__ mask_out, long total, int H, int W,
                                    int ph, int pw, int do_clamp, float lo, float hi, const float* __restrict__ gate) {
  if (gate != nullptr && gate[0] != 0.f) return;
  const long HW = (long)H * W;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long bc = idx / HW, pix = idx - bc * HW;
    const int b = (int)(bc / 3), c = (int)(bc - 3L * b);
    const int y = (int)(pix / W), x = (int)(pix - (long)y * W);
    const int i = y - origins[2 * b], j = x - origins[2 * b + 1];
    float m = 0.f, pv = 0.f;
    if ((unsigned)i < (unsigned)ph && (unsigned)j < (unsigned)pw) {
      const int e = (c * ph + i) * pw + j;
      m = mask_p[e];
      pv = patch_p[e];
    }
    const float mp = m * pv, om = 1.0f - m;
    float a = om * tgt[idx] + mp, r = om * ref[idx] + mp;
    if (do_clamp) { a = clampf(a, lo, hi); r = clampf(r, lo, hi); }
    adv_tgt[idx] = a;
    adv_ref[idx] = r;
    if (mask_out) mask_out[idx] = m;
  }
}

// The re-paste of the SECOND and later iterations of an attack() call: the frames outside the patch rectangles already hold
// clamp(frame) from the first iteration's full-canvas paste and never change again (main.py:585-600 recomputes them to the same
// values), so only the B x 3 x ph x pw rectangle pixels are rewritten -- 188 MB of canvas traffic per iteration become 0.4 MB.
__global__ void paste_placed_rect_kernel(const float* __restrict__ tgt, const float* __restrict__ ref,
                                         const float* __restrict__ patch_p, const float* __restrict__ mask_p,
                                         const int* __restrict__ origins, float* __restrict__ adv_tgt,
                                         float* __restrict__ adv_ref, int B, int H, int W, int ph, int pw, int do_clamp, float lo,
                                         float hi, const float* __restrict__ gate) {
  if (gate != nullptr && gate[0] != 0.f) return;
  const int n = 3 * ph * pw;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < B * n; idx += gridDim.x * blockDim.x) {
    const int b = idx / n, e = idx - b * n;
    const int c = e / (ph * pw), r = e - c * ph * pw, i = r / pw, j = r - i * pw;
    const int y = origins[2 * b] + i, x = origins[2 * b + 1] + j;
    if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) continue;       // clipped like the full-canvas paste
    const long o = ((long)b * 3 + c) * H * W + (long)y * W + x;
    const float m = mask_p[e], mp = m * patch_p[e], om = 1.0f - m;
    float a = om * tgt[o] + mp, q = om * ref[o] + mp;
    if (do_clamp) { a = clampf(a, lo, hi); q = clampf(q, lo, hi); }
    adv_tgt[o] = a;
    adv_ref[o] = q;
  }
}

// models/FlowNetC.py:73-79, :93-94 (`normalize_correctly`): the float64 mean subtraction of both frame stacks, written as
// ONE float32 stack [Ba + Bb, C, H, W] (first frames, then second frames) -- replaces torch.cat + .double() + sub + .float().
__global__ void normalize_frames_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                        long na, long total, int C, long HW, const double* __restrict__ mean) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW) % C);
    const float v = i < na ? a[i] : b[i - na];
    out[i] = (float)((double)v - mean[c]);
  }
}

__global__ void gate_kernel(const float* __restrict__ loss_cur, float* __restrict__ state, float thr) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && state[0] == 0.f) {
    const float l = *loss_cur;
    state[1] += 1.f;
    state[2] = l;
    if (!(l > thr)) state[0] = 1.f;   // `while loss_scalar > 0.1`: a NaN loss also ends the loop
  }
}

}  // namespace

extern "C" int ufr_attack_gate(const float* loss_cur, float* state, float threshold, ufr_stream_t stream) {
  UFR_REQUIRE(loss_cur && state, "attack gate: null pointer argument");
  hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(64), 0, ufr::as_stream(stream), loss_cur, state, threshold);
  return ufr::launched("gate_kernel");
}

extern "C" int ufr_patch_paste(const float* tgt, const float* ref, const float* patch,
                               const float* mask, float* adv_tgt, float* adv_ref, int B, int CHW,
                               long patch_bstride, long mask_bstride, int do_clamp, float lo,
                               float hi, ufr_stream_t stream) {
  UFR_REQUIRE(tgt && ref && patch && mask && adv_tgt && adv_ref, "patch paste: null pointer argument");
  UFR_REQUIRE(B > 0 && CHW > 0, "patch paste: bad shape");
  const long total = (long)B * CHW;
  hipLaunchKernelGGL(paste_kernel, dim3(ufr::stream_grid(total, 256)), dim3(256), 0,
                     ufr::as_stream(stream), tgt, ref, patch, mask, adv_tgt, adv_ref, total,
                     (long)CHW, patch_bstride, mask_bstride, do_clamp, lo, hi);
  return ufr::launched("paste_kernel");
}

extern "C" int ufr_patch_update(const float* tgt, const float* ref, const float* g_tgt,
                                const float* g_ref, float* patch, const float* mask, float* adv_tgt,
                                float* adv_ref, int B, int CHW, long patch_bstride, long mask_bstride,
                                float step, float bound, float lo, float hi, const float* gate_state,
                                ufr_stream_t stream) {
  UFR_REQUIRE(B > 0 && CHW > 0, "patch update: bad shape");
  UFR_REQUIRE(patch_bstride != 0 || B == 1,
              "patch update: one canvas patch behind %d pairs is not defined -- a shared patch lives in patch "
              "coordinates (ufr_patch_grad_crop / ufr_patch_apply / ufr_patch_paste_placed)", B);
  UFR_REQUIRE(tgt && ref && g_tgt && g_ref && patch && mask && adv_tgt && adv_ref,
              "patch update: null pointer argument");
  const long total = (long)B * CHW;
  hipLaunchKernelGGL(update_private_kernel, dim3(ufr::stream_grid(total, 256)), dim3(256), 0, ufr::as_stream(stream),
                     tgt, ref, g_tgt, g_ref, patch, mask, adv_tgt, adv_ref, total, (long)CHW,
                     patch_bstride, mask_bstride, step, bound, lo, hi, gate_state);
  return ufr::launched("update_private_kernel");
}

extern "C" int ufr_flow_loss(const float* flow, const float* target, float* grad_flow, float* loss,
                             int B, int HW, int kind, float weight, float* partials, ufr_stream_t stream) {
  UFR_REQUIRE(flow && target && grad_flow && loss && partials, "flow loss: null pointer argument");
  UFR_REQUIRE(B > 0 && HW > 0 && (kind == 0 || kind == 1), "flow loss: bad argument");
  const long npix = (long)B * HW;
  int grid = ufr::stream_grid(npix, 256);
  if (grid > UFR_LOSS_PARTIALS) grid = UFR_LOSS_PARTIALS;
  hipLaunchKernelGGL(flow_loss_kernel, dim3(grid), dim3(256), 0, ufr::as_stream(stream), flow,
                     target, grad_flow, partials, B, (long)HW, kind, weight);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, ufr::as_stream(stream), partials, grid, loss);
  return ufr::launched("flow_loss_kernel");
}

// ---- loss on the x4-upsampled flow without the full-size flow (ufr_flow2_upsampled_loss) --------------------------------
// torch's upsample_bilinear2d, align_corners = False, scale 1/4 (aten/src/ATen/native/UpSample.h area_pixel_compute_source_index):
//   src = max((dst + 0.5) * 0.25 - 0.5, 0); i0 = (int)src; i1 = i0 + (i0 < in - 1); l1 = src - i0; l0 = 1 - l1
//   out = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11)
// A workgroup owns 16 x 16 cells of flow2 = 64 x 64 pixels: it evaluates the loss gradient on those pixels and a rim of 4
// (the pixels that reach its cells), keeps it in LDS, and every thread gathers one cell's gradient from the <= 10 x 10
// pixels around it.  Loss terms are counted on the owned pixels only.
constexpr int F2_T = 16, F2_R = 4 * F2_T + 8;                              // cells per tile side; pixels per region side
__device__ __forceinline__ void up4_source(int dst, int in, int& i0, int& i1, float& l1) {
  const float src = fmaxf(((float)dst + 0.5f) * 0.25f - 0.5f, 0.f);
  i0 = (int)src;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

__global__ __launch_bounds__(256) void flow2_upsampled_loss_kernel(const float* __restrict__ flow2, float flow_scale,
                                                                   const float* __restrict__ target,
                                                                   float* __restrict__ gflow2, float* __restrict__ partials,
                                                                   int B, int h, int w, int kind, float weight) {
  __shared__ float f2[2][F2_T + 4][F2_T + 4];                              // cells cy0 - 2 .. cy0 + 17 (indices clamped), x flow_scale
  __shared__ float g[2][F2_R][F2_R + 1];                                   // d loss / d flow on the region's pixels
  __shared__ float red[4];
  const int tid = threadIdx.x, b = blockIdx.z;
  const int cy0 = blockIdx.y * F2_T, cx0 = blockIdx.x * F2_T;
  const int H = 4 * h, W = 4 * w;
  const long HW = (long)H * W, hw = (long)h * w;
  const float invn = weight / (float)((long)B * HW);
  for (int i = tid; i < 2 * (F2_T + 4) * (F2_T + 4); i += 256) {
    const int c = i / ((F2_T + 4) * (F2_T + 4)), r = i - c * (F2_T + 4) * (F2_T + 4), yy = r / (F2_T + 4), xx = r - yy * (F2_T + 4);
    const int cy = min(max(cy0 - 2 + yy, 0), h - 1), cx = min(max(cx0 - 2 + xx, 0), w - 1);
    f2[c][yy][xx] = flow2[((long)b * 2 + c) * hw + (long)cy * w + cx] * flow_scale;
  }
  __syncthreads();
  const int py0 = 4 * cy0 - 4, px0 = 4 * cx0 - 4;
  float part = 0.f;
  for (int i = tid; i < F2_R * F2_R; i += 256) {
    const int ry = i / F2_R, rx = i - ry * F2_R, py = py0 + ry, px = px0 + rx;
    float gu = 0.f, gv = 0.f;
    if (py >= 0 && py < H && px >= 0 && px < W) {
      int y0, y1, x0, x1;
      float ly, lx;
      up4_source(py, h, y0, y1, ly);
      up4_source(px, w, x0, x1, lx);
      const float hy0 = 1.f - ly, wx0 = 1.f - lx;
      const int a0 = y0 - (cy0 - 2), a1 = y1 - (cy0 - 2), c0 = x0 - (cx0 - 2), c1 = x1 - (cx0 - 2);
      const float fu = hy0 * (wx0 * f2[0][a0][c0] + lx * f2[0][a0][c1]) + ly * (wx0 * f2[0][a1][c0] + lx * f2[0][a1][c1]);
      const float fv = hy0 * (wx0 * f2[1][a0][c0] + lx * f2[1][a0][c1]) + ly * (wx0 * f2[1][a1][c0] + lx * f2[1][a1][c1]);
      const long o0 = ((long)b * 2) * HW + (long)py * W + px, o1 = o0 + HW;
      const float tu = target[o0], tv = target[o1];
      float term;
      if (kind == 0) {
        const float dot = fu * tu + fv * tv;
        const float nf2 = fu * fu + fv * fv, nt2 = tu * tu + tv * tv;
        const float den = fmaxf(sqrtf(nf2 * nt2), 1e-8f);
        const float c = dot / den;
        term = 1.0f - c;
        if (sqrtf(nf2 * nt2) > 1e-8f) {
          gu = -(tu / den - c * fu / nf2);
          gv = -(tv / den - c * fv / nf2);
        } else {
          gu = -(tu / den);
          gv = -(tv / den);
        }
      } else {
        const float du = fu - tu, dv = fv - tv;
        const float sd = sqrtf(du * du + dv * dv + 1e-8f);
        term = sd;
        gu = du / sd;
        gv = dv / sd;
      }
      gu *= invn;
      gv *= invn;
      const bool owned = ry >= 4 && ry < F2_R - 4 && rx >= 4 && rx < F2_R - 4;
      if (owned) part += term;
    }
    g[0][ry][rx] = gu;
    g[1][ry][rx] = gv;
  }
  __syncthreads();
  {                                                                        // one cell per thread
    const int cyl = tid >> 4, cxl = tid & 15, cy = cy0 + cyl, cx = cx0 + cxl;
    if (cy < h && cx < w) {
      float su = 0.f, sv = 0.f;
      for (int py = 4 * cy - 3; py <= 4 * cy + 6; ++py) {
        if (py < 0 || py >= H) continue;
        int y0, y1;
        float ly;
        up4_source(py, h, y0, y1, ly);
        const float wy = (y0 == cy ? 1.f - ly : 0.f) + (y1 == cy ? ly : 0.f);
        if (wy == 0.f) continue;
        float ru = 0.f, rv = 0.f;
        for (int px = 4 * cx - 3; px <= 4 * cx + 6; ++px) {
          if (px < 0 || px >= W) continue;
          int x0, x1;
          float lx;
          up4_source(px, w, x0, x1, lx);
          const float wx = (x0 == cx ? 1.f - lx : 0.f) + (x1 == cx ? lx : 0.f);
          ru += wx * g[0][py - py0][px - px0];
          rv += wx * g[1][py - py0][px - px0];
        }
        su += wy * ru;
        sv += wy * rv;
      }
      gflow2[((long)b * 2) * hw + (long)cy * w + cx] = su * flow_scale;
      gflow2[((long)b * 2 + 1) * hw + (long)cy * w + cx] = sv * flow_scale;
    }
  }
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = part;
  __syncthreads();
  if (tid == 0) partials[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) * invn;
}

extern "C" int ufr_flow2_upsampled_loss(const float* flow2, float flow_scale, const float* target, float* grad_flow2, float* loss,
                                        int B, int h, int w, int kind, float weight, float* partials, ufr_stream_t stream) {
  UFR_REQUIRE(flow2 && target && grad_flow2 && loss && partials, "upsampled flow loss: null pointer argument");
  UFR_REQUIRE(B > 0 && h > 0 && w > 0 && (kind == 0 || kind == 1), "upsampled flow loss: bad argument");
  const dim3 grid(ufr::ceil_div(w, F2_T), ufr::ceil_div(h, F2_T), B);
  const long n = (long)grid.x * grid.y * grid.z;
  if (n > UFR_LOSS_PARTIALS || B > 65535)
    return ufr::fail(UFR_EUNSUPPORTED, "upsampled flow loss: %ld tiles exceed the %d partial sums of the fixed-order reduction", n,
                     UFR_LOSS_PARTIALS);
  hipLaunchKernelGGL(flow2_upsampled_loss_kernel, grid, dim3(256), 0, ufr::as_stream(stream), flow2, flow_scale, target, grad_flow2,
                     partials, B, h, w, kind, weight);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, ufr::as_stream(stream), partials, (int)n, loss);
  return ufr::launched("flow2_upsampled_loss_kernel");
}

namespace ufr {
void loss_finalize_launch(const float* partials, int n, float* loss, hipStream_t st) {
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, partials, n, loss);
}
}  // namespace ufr

static int check_placed(const int* origins_host, int B, int H, int W, int ph, int pw) {
  if (!origins_host) return 1;
  for (int b = 0; b < B; ++b)
    if (origins_host[2 * b] < 0 || origins_host[2 * b] + ph > H || origins_host[2 * b + 1] < 0 ||
        origins_host[2 * b + 1] + pw > W)
      return 0;
  return 1;
}

extern "C" int ufr_patch_grad_crop(const float* g_tgt, const float* g_ref, const float* mask_p, const int* origins,
                                   const int* origins_host, const float* loss_local, float* rows, int B, int H, int W,
                                   int ph, int pw, int groups, ufr_stream_t stream) {
  UFR_REQUIRE(g_tgt && g_ref && mask_p && origins && loss_local && rows, "patch grad crop: null pointer argument");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && ph > 0 && pw > 0 && ph <= H && pw <= W, "patch grad crop: bad shape");
  UFR_REQUIRE(groups > 0 && B % groups == 0, "patch grad crop: the pairs must split evenly into %d groups", groups);
  UFR_REQUIRE(check_placed(origins_host, B, H, W, ph, pw), "patch grad crop: a placement leaves the frame");
  const int n = 3 * ph * pw + 1;
  hipLaunchKernelGGL(patch_grad_crop_kernel, dim3(ufr::ceil_div(n, 256)), dim3(256), 0, ufr::as_stream(stream), g_tgt,
                     g_ref, mask_p, origins, loss_local, rows, B, H, W, ph, pw, groups);
  return ufr::launched("patch_grad_crop_kernel");
}

extern "C" int ufr_patch_grad_crop_window(const float* gxw, const int* win, const float* mask_p, const int* origins,
                                          const float* loss_local, float* rows, int B, int H, int W, int wh, int ww, int ph, int pw,
                                          int groups, ufr_stream_t stream) {
  UFR_REQUIRE(gxw && win && mask_p && origins && loss_local && rows, "patch grad crop (window): null pointer argument");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && ph > 0 && pw > 0 && ph <= H && pw <= W && wh > 0 && ww > 0 && wh <= H && ww <= W,
              "patch grad crop (window): bad shape");
  UFR_REQUIRE(groups > 0 && B % groups == 0, "patch grad crop (window): the pairs must split evenly into %d groups", groups);
  const int n = 3 * ph * pw + 1;
  hipLaunchKernelGGL(patch_grad_crop_window_kernel, dim3(ufr::ceil_div(n, 256)), dim3(256), 0, ufr::as_stream(stream), gxw, win,
                     mask_p, origins, loss_local, rows, B, H, W, wh, ww, ph, pw, groups);
  return ufr::launched("patch_grad_crop_window_kernel");
}

extern "C" int ufr_patch_apply(const float* rows, int n_rows, float* patch_p, float* loss, int ph, int pw, float step,
                               float bound, const float* gate_state, ufr_stream_t stream) {
  UFR_REQUIRE(rows && patch_p && loss, "patch apply: null pointer argument");
  UFR_REQUIRE(n_rows > 0 && ph > 0 && pw > 0, "patch apply: bad shape");
  const int n = 3 * ph * pw;
  hipLaunchKernelGGL(patch_apply_kernel, dim3(ufr::ceil_div(n + 1, 256)), dim3(256), 0, ufr::as_stream(stream), rows,
                     n_rows, patch_p, loss, n, step, bound, gate_state);
  return ufr::launched("patch_apply_kernel");
}

extern "C" int ufr_patch_paste_placed(const float* tgt, const float* ref, const float* patch_p, const float* mask_p,
                                      const int* origins, const int* origins_host, float* adv_tgt, float* adv_ref,
                                      float* mask_out, int B, int H, int W, int ph, int pw, int do_clamp, float lo,
                                      float hi, const float* gate_state, ufr_stream_t stream) {
  UFR_REQUIRE(tgt && ref && patch_p && mask_p && origins && adv_tgt && adv_ref, "placed paste: null pointer argument");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && ph > 0 && pw > 0 && ph <= H && pw <= W, "placed paste: bad shape");
  UFR_REQUIRE(check_placed(origins_host, B, H, W, ph, pw), "placed paste: a placement leaves the frame");
  const long total = (long)B * 3 * H * W;
  hipLaunchKernelGGL(paste_placed_kernel, dim3(ufr::stream_grid(total, 256)), dim3(256), 0, ufr::as_stream(stream), tgt,
                     ref, patch_p, mask_p, origins, adv_tgt, adv_ref, mask_out, total, H, W, ph, pw, do_clamp, lo, hi,
                     gate_state);
  return ufr::launched("paste_placed_kernel");
}

extern "C" int ufr_patch_paste_placed_rect(const float* tgt, const float* ref, const float* patch_p, const float* mask_p,
                                           const int* origins, float* adv_tgt, float* adv_ref, int B, int H, int W, int ph, int pw,
                                           int do_clamp, float lo, float hi, const float* gate_state, ufr_stream_t stream) {
  UFR_REQUIRE(tgt && ref && patch_p && mask_p && origins && adv_tgt && adv_ref, "placed rect paste: null pointer argument");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && ph > 0 && pw > 0 && ph <= H && pw <= W, "placed rect paste: bad shape");
  const long total = (long)B * 3 * ph * pw;
  UFR_REQUIRE(total < (1L << 30), "placed rect paste: too many patch pixels");
  hipLaunchKernelGGL(paste_placed_rect_kernel, dim3(ufr::stream_grid(total, 256)), dim3(256), 0, ufr::as_stream(stream), tgt, ref,
                     patch_p, mask_p, origins, adv_tgt, adv_ref, B, H, W, ph, pw, do_clamp, lo, hi, gate_state);
  return ufr::launched("paste_placed_rect_kernel");
}

extern "C" int ufr_normalize_frames(const float* frames_a, const float* frames_b, float* out, int Ba, int Bb, int C, int H, int W,
                                    const double* mean, ufr_stream_t stream) {
  UFR_REQUIRE(frames_a && out && mean && (frames_b || Bb == 0), "normalize frames: null pointer argument");
  UFR_REQUIRE(Ba > 0 && Bb >= 0 && C > 0 && H > 0 && W > 0, "normalize frames: bad shape");
  const long HW = (long)H * W, na = (long)Ba * C * HW, total = na + (long)Bb * C * HW;
  normalize_frames_kernel<<<ufr::stream_grid(total, 256), 256, 0, ufr::as_stream(stream)>>>(frames_a, frames_b, out, na, total, C, HW,
                                                                                             mean);
  return ufr::launched("normalize_frames_kernel");
}
