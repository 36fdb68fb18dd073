// universal.hip -- elementwise stages of the universal-perturbation / I-FGSM inner loops
//   global_attacks/universal_perturbation.py:452-530 (attack), :667-675 (add_universal_perturbation)
//   global_attacks/perturb_model.py:102-145 (compute_flow_loss)
// One streaming pass each; the whole post-backward sequence of the reference
//   sign(grad) -> lr*sign -> adv -/+ step -> clamp[0,1] -> noise = clamp(adv-img, +-eps) -> adv = img+noise
// (8 elementwise torch kernels per frame and step) is one kernel for both frames.
#include "ufr_common.h"

namespace {

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
__device__ __forceinline__ float signf(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }  // torch.sign

// perturb_model.py:128-145.  kind 0 cossim, 1 l2 (+10e-8 inside the sqrt), 2 l1 (sum of |du|,|dv|
// averaged over B*2*H*W elements when unmasked -- `loss.mean()` of a [B,2,H,W] tensor).
// gt has Cg = 2 or 3 channels; with 3 the last one is a validity mask and `scale` must be
// 1/(sum(valid)+1e-8), otherwise 1/(number of averaged elements).
__global__ void flow_loss_ex_kernel(const float* __restrict__ flow, const float* __restrict__ gt,
                                    float* __restrict__ gflow, float* __restrict__ partials, int B, long HW,
                                    int Cg, int kind, float scale_val, const float* __restrict__ scale_dev) {
  __shared__ float red[256 / 64];
  const float scale = scale_dev ? *scale_dev : scale_val;   // device-resident so a captured graph sees updates
  const long npix = (long)B * HW;
  float part = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, p = i - b * HW;
    const size_t o0 = ((size_t)b * 2) * HW + p, o1 = o0 + HW;
    const size_t g0 = ((size_t)b * Cg) * HW + p;
    const float fu = flow[o0], fv = flow[o1], tu = gt[g0], tv = gt[g0 + HW];
    const float valid = (Cg == 3) ? gt[g0 + 2 * HW] : 1.0f;
    float l, gu, gv;
    if (kind == 0) {
      const float dot = fu * tu + fv * tv;
      const float nf2 = fu * fu + fv * fv, nt2 = tu * tu + tv * tv;
      const float nn = sqrtf(nf2 * nt2);
      const float den = fmaxf(nn, 1e-8f);
      const float c = dot / den;
      l = 1.0f - c;
      if (nn > 1e-8f) { gu = -(tu / den - c * fu / nf2); gv = -(tv / den - c * fv / nf2); }
      else { gu = -(tu / den); gv = -(tv / den); }
    } else if (kind == 1) {
      const float du = fu - tu, dv = fv - tv;
      const float s = sqrtf(du * du + dv * dv + 10e-8f);
      l = s; gu = du / s; gv = dv / s;
    } else {
      const float du = fu - tu, dv = fv - tv;
      l = fabsf(du) + fabsf(dv); gu = signf(du); gv = signf(dv);
    }
    part += l * valid;
    gflow[o0] = gu * valid * scale;
    gflow[o1] = gv * valid * scale;
  }
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) red[wv] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k];
    partials[blockIdx.x] = s * scale;      // added in a fixed order by attack.hip's loss_finalize_kernel
  }
}

// universal_perturbation.py:477-520 for both frames.  Element e of the [3,H,W] perturbation is owned
// by one thread, which walks the B samples.
//   shared == 0 (reference arithmetic, per sample):
//       adv = clamp(adv -/+ lr*dir(g), 0, 1); noise = clamp(adv - img, -eps, eps); adv = img + noise
//   shared == 1 (batch / multi-rank extension, one perturbation for all samples):
//       d = dir(sum_b g_b)  [mode 1: only write the sum; mode 2: take the sum from grad_sum]
//       delta = clamp(delta -/+ lr*d, -eps, eps); adv_b = clamp(img_b + delta, 0, 1)
// dir = sign for I-FGSM, identity for "ifgm"; frames are masked by perturb_mode (bit 0: frame 0, bit 1: frame 1).
__global__ void universal_update_kernel(const float* __restrict__ img0, const float* __restrict__ img1,
                                        const float* __restrict__ g0, const float* __restrict__ g1,
                                        float* __restrict__ grad_sum, float* __restrict__ adv0,
                                        float* __restrict__ adv1, float* __restrict__ delta, int B, long CHW,
                                        float lr, float eps, float lo, float hi, int use_sign, int frames,
                                        float direction, int shared, int mode) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < CHW; e += (long)gridDim.x * blockDim.x) {
    if (!shared) {
      for (int b = 0; b < B; ++b) {
        const long i = b * CHW + e;
        const float d0 = (frames & 1) ? lr * (use_sign ? signf(g0[i]) : g0[i]) : 0.f;
        const float d1 = (frames & 2) ? lr * (use_sign ? signf(g1[i]) : g1[i]) : 0.f;
        const float a0 = clampf(adv0[i] + direction * d0, lo, hi);
        const float a1 = clampf(adv1[i] + direction * d1, lo, hi);
        const float n0 = clampf(a0 - img0[i], -eps, eps), n1 = clampf(a1 - img1[i], -eps, eps);
        adv0[i] = img0[i] + n0;
        adv1[i] = img1[i] + n1;
        delta[((long)b * 2 + 0) * CHW + e] = n0;      // [B,2,3,H,W] like torch.stack(..., dim=1)
        delta[((long)b * 2 + 1) * CHW + e] = n1;
      }
      continue;
    }
    float s0, s1;
    if (mode == 2) {
      s0 = grad_sum[e]; s1 = grad_sum[CHW + e];
    } else {
      s0 = 0.f; s1 = 0.f;
      for (int b = 0; b < B; ++b) { s0 += g0[b * CHW + e]; s1 += g1[b * CHW + e]; }
      if (grad_sum) { grad_sum[e] = s0; grad_sum[CHW + e] = s1; }
      if (mode == 1) continue;
    }
    const float d0 = (frames & 1) ? lr * (use_sign ? signf(s0) : s0) : 0.f;
    const float d1 = (frames & 2) ? lr * (use_sign ? signf(s1) : s1) : 0.f;
    const float n0 = clampf(delta[e] + direction * d0, -eps, eps);
    const float n1 = clampf(delta[CHW + e] + direction * d1, -eps, eps);
    delta[e] = n0;
    delta[CHW + e] = n1;
    for (int b = 0; b < B; ++b) {
      adv0[b * CHW + e] = clampf(img0[b * CHW + e] + n0, lo, hi);
      adv1[b * CHW + e] = clampf(img1[b * CHW + e] + n1, lo, hi);
    }
  }
}

}  // namespace

extern "C" int ufr_flow_loss_ex(const float* flow, const float* gt, float* grad_flow, float* loss, int B,
                                int HW, int gt_channels, int kind, float scale, const float* scale_dev,
                                float* partials, ufr_stream_t stream) {
  UFR_REQUIRE(flow && gt && grad_flow && loss && partials, "flow loss: null pointer argument");
  UFR_REQUIRE(B > 0 && HW > 0 && kind >= 0 && kind <= 2 && (gt_channels == 2 || gt_channels == 3),
              "flow loss: bad argument (kind %d, gt channels %d)", kind, gt_channels);
  const long npix = (long)B * HW;
  int grid = ufr::stream_grid(npix, 256);
  if (grid > UFR_LOSS_PARTIALS) grid = UFR_LOSS_PARTIALS;
  hipLaunchKernelGGL(flow_loss_ex_kernel, dim3(grid), dim3(256), 0, ufr::as_stream(stream), flow, gt, grad_flow,
                     partials, B, (long)HW, gt_channels, kind, scale, scale_dev);
  ufr::loss_finalize_launch(partials, grid, loss, ufr::as_stream(stream));
  return ufr::launched("flow_loss_ex_kernel");
}

extern "C" int ufr_universal_update(const float* img0, const float* img1, const float* g0, const float* g1,
                                    float* grad_sum, float* adv0, float* adv1, float* delta, int B, int CHW,
                                    float lr, float eps, float lo, float hi, int use_sign, int frames,
                                    int ascent, int shared, int mode, ufr_stream_t stream) {
  UFR_REQUIRE(B > 0 && CHW > 0 && mode >= 0 && mode <= 2 && frames >= 1 && frames <= 3, "universal update: bad argument");
  UFR_REQUIRE(shared || mode == 0, "universal update: per-sample perturbations have no gradient exchange");
  if (mode != 2) UFR_REQUIRE(g0 && g1, "universal update: null gradient pointer");
  if (mode != 0) UFR_REQUIRE(grad_sum, "universal update: mode %d needs grad_sum", mode);
  if (mode != 1) UFR_REQUIRE(img0 && img1 && adv0 && adv1 && delta, "universal update: null pointer argument");
  hipLaunchKernelGGL(universal_update_kernel, dim3(ufr::stream_grid(CHW, 256)), dim3(256), 0,
                     ufr::as_stream(stream), img0, img1, g0, g1, grad_sum, adv0, adv1, delta, B, (long)CHW, lr,
                     eps, lo, hi, use_sign, frames, ascent ? 1.0f : -1.0f, shared, mode);
  return ufr::launched("universal_update_kernel");
}
