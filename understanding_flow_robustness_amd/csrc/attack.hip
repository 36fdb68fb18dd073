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

// main.py:575-600 fused.  One thread per canvas element e (not per batch element): the thread
// sums the B per-sample gradients (batch extension: one shared patch), updates the patch once and
// re-pastes all B frame pairs.
//   mode 0: sum, update, paste     mode 1: sum only -> grad_sum     mode 2: update+paste from grad_sum
__global__ void update_shared_kernel(const float* __restrict__ tgt, const float* __restrict__ ref,
                                     const float* __restrict__ g_tgt, const float* __restrict__ g_ref,
                                     float* __restrict__ grad_sum, float* __restrict__ patch,
                                     const float* __restrict__ mask, float* __restrict__ adv_tgt,
                                     float* __restrict__ adv_ref, int B, long CHW, long mstride,
                                     float step, float bound, float lo, float hi, int mode, int masked,
                                     const float* __restrict__ gate) {
  // attack already converged: the iteration is void (mode 1 still refreshes the exchange buffer so
  // the collective that follows never re-sums stale data)
  if (mode != 1 && gate != nullptr && gate[0] != 0.f) return;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < CHW;
       e += (long)gridDim.x * blockDim.x) {
    float gs;
    if (mode == 2) {
      gs = grad_sum[e];
    } else {
      gs = 0.f;
      // masked: a pair contributes only where ITS mask shows the patch (d batch-loss / d shared patch)
      for (int b = 0; b < B; ++b)
        if (!masked || mask[b * mstride + e] != 0.f) gs += g_tgt[b * CHW + e] + g_ref[b * CHW + e];
      if (grad_sum) grad_sum[e] = gs;
      if (mode == 1) continue;
    }
    const float pv = patch[e] - clampf(step * gs, -bound, bound);
    patch[e] = pv;
    for (int b = 0; b < B; ++b) {
      const float m = mask[b * mstride + e];
      const float mp = m * pv, om = 1.0f - m;
      adv_tgt[b * CHW + e] = clampf(om * tgt[b * CHW + e] + mp, lo, hi);
      adv_ref[b * CHW + e] = clampf(om * ref[b * CHW + e] + mp, lo, hi);
    }
  }
}

// Per-sample patches (patch_bstride != 0): exactly the reference's B=1 arithmetic per sample.
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

// Loss + its gradient wrt the flow in one pass (replaces the autograd graph of main.py:557-566).
//  kind 0: mean(1 - cos(f,t)),  cos = <f,t> / max(|f|*|t|, 1e-8)   (torch cosine_similarity, eps 1e-8)
//          d/df = -( t/(|f||t|) - cos * f/|f|^2 ) / Npix
//  kind 1: mean(sqrt(|f-t|^2 + 1e-8)),  d/df = (f-t)/sqrt(.) / Npix
// Block-level reduction in LDS, one atomicAdd per workgroup (order-dependent in the last bits; the
// scalar is only used for the `loss <= 0.1` early exit and for logging).
__global__ void flow_loss_kernel(const float* __restrict__ flow, const float* __restrict__ target,
                                 float* __restrict__ gflow, float* __restrict__ loss, int B, long HW,
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
    atomicAdd(loss, s * invn);
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
                                const float* g_ref, float* grad_sum, float* patch,
                                const float* mask, float* adv_tgt, float* adv_ref, int B, int CHW,
                                long patch_bstride, long mask_bstride, float step, float bound,
                                float lo, float hi, int mode, const float* gate_state,
                                ufr_stream_t stream) {
  UFR_REQUIRE(B > 0 && CHW > 0, "patch update: bad shape");
  const int masked = (mode & UFR_UPDATE_MASKED_SUM) != 0;
  mode &= ~UFR_UPDATE_MASKED_SUM;
  UFR_REQUIRE(mode >= 0 && mode <= 2, "patch update: bad mode %d", mode);
  hipStream_t st = ufr::as_stream(stream);
  if (patch_bstride != 0) {
    UFR_REQUIRE(mode == 0, "patch update: per-sample patches have no cross-sample gradient sum");
    UFR_REQUIRE(tgt && ref && g_tgt && g_ref && patch && mask && adv_tgt && adv_ref,
                "patch update: null pointer argument");
    const long total = (long)B * CHW;
    hipLaunchKernelGGL(update_private_kernel, dim3(ufr::stream_grid(total, 256)), dim3(256), 0, st,
                       tgt, ref, g_tgt, g_ref, patch, mask, adv_tgt, adv_ref, total, (long)CHW,
                       patch_bstride, mask_bstride, step, bound, lo, hi, gate_state);
    return ufr::launched("update_private_kernel");
  }
  if (mode != 2) UFR_REQUIRE(g_tgt && g_ref, "patch update: null gradient pointer");
  if (mode != 0) UFR_REQUIRE(grad_sum, "patch update: mode %d needs grad_sum", mode);
  if (mode != 1)
    UFR_REQUIRE(tgt && ref && patch && mask && adv_tgt && adv_ref, "patch update: null pointer argument");
  if (masked && mode != 2) UFR_REQUIRE(mask, "patch update: masked sum needs the masks");
  hipLaunchKernelGGL(update_shared_kernel, dim3(ufr::stream_grid(CHW, 256)), dim3(256), 0, st, tgt,
                     ref, g_tgt, g_ref, grad_sum, patch, mask, adv_tgt, adv_ref, B, (long)CHW,
                     mask_bstride, step, bound, lo, hi, mode, masked, gate_state);
  return ufr::launched("update_shared_kernel");
}

extern "C" int ufr_flow_loss(const float* flow, const float* target, float* grad_flow, float* loss,
                             int B, int HW, int kind, float weight, ufr_stream_t stream) {
  UFR_REQUIRE(flow && target && grad_flow && loss, "flow loss: null pointer argument");
  UFR_REQUIRE(B > 0 && HW > 0 && (kind == 0 || kind == 1), "flow loss: bad argument");
  const long npix = (long)B * HW;
  int grid = ufr::stream_grid(npix, 256);
  if (grid > 512) grid = 512;
  hipLaunchKernelGGL(flow_loss_kernel, dim3(grid), dim3(256), 0, ufr::as_stream(stream), flow,
                     target, grad_flow, loss, B, (long)HW, kind, weight);
  return ufr::launched("flow_loss_kernel");
}
