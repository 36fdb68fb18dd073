// pwc_warp.hip -- PWC-Net's backward warp (models/PWCNet.py:164-204) as one kernel forward, one backward.
//
// The reference builds a normalised sampling grid from the flow with FOUR broadcast elementwise ops per axis,
// calls grid_sample twice (features, and a tensor of ones for the validity mask), thresholds the warped ones
// at 0.0001 and multiplies: ~15 kernels and several feature-map-sized temporaries per pyramid level, plus
// their adjoints.  Its arithmetic, kept here operation for operation in float32:
//   vx = 2*(x + flow_x) / max(W-1, 1) - 1            (grid normalised with W-1 ...)
//   ix = ((vx + 1) * W - 1) / 2                      (... but sampled with align_corners=False: a quirk)
//   bilinear with zero padding, weights nw/ne/sw/se as in ATen's grid_sampler
//   mask = (sum of the in-bounds weights >= 0.0001);  out = sampled * mask
// HBM-streaming: a thread computes a pixel's four weights once and walks its slice of the channels.
// Backward: feature gradient scattered with float atomics to the four corners (ATen does the same), flow
// gradient from the four corner values; the mask is piecewise constant and carries no gradient.
#include "ufr_common.h"

namespace {

struct Tap {
  int x0, y0;            // north-west corner
  float nw, ne, sw, se;  // bilinear weights
  bool in_nw, in_ne, in_sw, in_se;
  float keep;            // 1 when the warped ones reach 0.0001, else 0
  float tx, ty;          // ix - x0, iy - y0 pieces needed by the flow gradient
  float ix, iy;
};

__device__ __forceinline__ Tap make_tap(float fx, float fy, int x, int y, int H, int W) {
  Tap t;
  const float dW = (float)max(W - 1, 1), dH = (float)max(H - 1, 1);
  const float vx = 2.0f * ((float)x + fx) / dW - 1.0f;
  const float vy = 2.0f * ((float)y + fy) / dH - 1.0f;
  t.ix = ((vx + 1.f) * (float)W - 1.f) / 2.f;
  t.iy = ((vy + 1.f) * (float)H - 1.f) / 2.f;
  const float fx0 = floorf(t.ix), fy0 = floorf(t.iy);
  // clamp before the int conversion: far-away (or non-finite) positions simply have no in-bounds corner
  t.x0 = (int)fminf(fmaxf(fx0, -2.0e6f), 2.0e6f);
  t.y0 = (int)fminf(fmaxf(fy0, -2.0e6f), 2.0e6f);
  const float x1 = fx0 + 1.f, y1 = fy0 + 1.f;
  t.nw = (x1 - t.ix) * (y1 - t.iy);
  t.ne = (t.ix - fx0) * (y1 - t.iy);
  t.sw = (x1 - t.ix) * (t.iy - fy0);
  t.se = (t.ix - fx0) * (t.iy - fy0);
  const bool xin0 = t.x0 >= 0 && t.x0 < W, xin1 = t.x0 + 1 >= 0 && t.x0 + 1 < W;
  const bool yin0 = t.y0 >= 0 && t.y0 < H, yin1 = t.y0 + 1 >= 0 && t.y0 + 1 < H;
  const bool finite = fabsf(t.ix) < 1.0e6f && fabsf(t.iy) < 1.0e6f;
  t.in_nw = finite && xin0 && yin0; t.in_ne = finite && xin1 && yin0;
  t.in_sw = finite && xin0 && yin1; t.in_se = finite && xin1 && yin1;
  float m = 0.f;                              // grid_sample(ones): same accumulation order as ATen
  if (t.in_nw) m += t.nw;
  if (t.in_ne) m += t.ne;
  if (t.in_sw) m += t.sw;
  if (t.in_se) m += t.se;
  t.keep = m >= 0.0001f ? 1.f : 0.f;
  return t;
}

// Thread = (pixel, channel slice): a workgroup is S waves over the same 64 pixels, wave s walking channels s, s + S, ...
// (coalesced along the pixels; S chosen by the host so that the coarse pyramid levels -- 12 x 40 pixels, 128 channels --
// still fill the chip: one thread per pixel walking every channel took 100-190 us there, latency-bound).
__global__ __launch_bounds__(1024) void pwc_warp_fwd(const float* __restrict__ x, const float* __restrict__ flo,
                                                     float* __restrict__ out, int B, int C, int H, int W, int S) {
  const size_t plane = (size_t)H * W;
  const long npix = (long)B * plane;
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const long idx = (long)blockIdx.x * 64 + lane;
  if (idx >= npix) return;
  const int px = (int)(idx % W), py = (int)((idx / W) % H), b = (int)(idx / (long)plane);
  const size_t pix = (size_t)py * W + px;
  const Tap t = make_tap(flo[((size_t)b * 2 + 0) * plane + pix], flo[((size_t)b * 2 + 1) * plane + pix], px, py, H, W);
  const size_t o_nw = (size_t)t.y0 * W + t.x0;
  for (int c = slice; c < C; c += S) {
    const float* im = x + ((size_t)b * C + c) * plane;
    float v = 0.f;
    if (t.in_nw) v += im[o_nw] * t.nw;
    if (t.in_ne) v += im[o_nw + 1] * t.ne;
    if (t.in_sw) v += im[o_nw + W] * t.sw;
    if (t.in_se) v += im[o_nw + W + 1] * t.se;
    out[((size_t)b * C + c) * plane + pix] = v * t.keep;
  }
}

// gx must be zero-filled by the caller side of this launch (done in the entry point).  The flow gradient's sum over the
// channels: per-slice partials through LDS, added in ascending slice order (fixed order: reproducible).
__global__ __launch_bounds__(1024) void pwc_warp_bwd(const float* __restrict__ x, const float* __restrict__ flo,
                                                     const float* __restrict__ gout, float* __restrict__ gx,
                                                     float* __restrict__ gflo, int B, int C, int H, int W, int S) {
  __shared__ float part[16][64][2];
  const size_t plane = (size_t)H * W;
  const long npix = (long)B * plane;
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const long idx = (long)blockIdx.x * 64 + lane;
  const bool live = idx < npix;
  float gix = 0.f, giy = 0.f;
  int px = 0, py = 0, b = 0;
  size_t pix = 0;
  if (live) {
    px = (int)(idx % W); py = (int)((idx / W) % H); b = (int)(idx / (long)plane);
    pix = (size_t)py * W + px;
    const Tap t = make_tap(flo[((size_t)b * 2 + 0) * plane + pix], flo[((size_t)b * 2 + 1) * plane + pix], px, py, H, W);
    const size_t o_nw = (size_t)t.y0 * W + t.x0;
    const float x0f = floorf(t.ix), y0f = floorf(t.iy), x1f = x0f + 1.f, y1f = y0f + 1.f;
    if (t.keep != 0.f) {
      for (int c = slice; c < C; c += S) {
        const float g = gout[((size_t)b * C + c) * plane + pix];
        const float* im = x + ((size_t)b * C + c) * plane;
        float* gi = gx + ((size_t)b * C + c) * plane;
        if (t.in_nw) { const float v = im[o_nw]; atomicAdd(&gi[o_nw], t.nw * g);
                       gix -= v * (y1f - t.iy) * g; giy -= v * (x1f - t.ix) * g; }
        if (t.in_ne) { const float v = im[o_nw + 1]; atomicAdd(&gi[o_nw + 1], t.ne * g);
                       gix += v * (y1f - t.iy) * g; giy -= v * (t.ix - x0f) * g; }
        if (t.in_sw) { const float v = im[o_nw + W]; atomicAdd(&gi[o_nw + W], t.sw * g);
                       gix -= v * (t.iy - y0f) * g; giy += v * (x1f - t.ix) * g; }
        if (t.in_se) { const float v = im[o_nw + W + 1]; atomicAdd(&gi[o_nw + W + 1], t.se * g);
                       gix += v * (t.iy - y0f) * g; giy += v * (t.ix - x0f) * g; }
      }
    }
  }
  part[slice][lane][0] = gix;
  part[slice][lane][1] = giy;
  __syncthreads();
  if (slice == 0 && live) {
    float sx = 0.f, sy = 0.f;
    for (int k = 0; k < S; ++k) {
      sx += part[k][lane][0];
      sy += part[k][lane][1];
    }
    // d ix / d grid = W / 2 (ATen's unnormalise multiplier), d grid / d flow = 2 / max(W-1, 1)
    const float dW = (float)max(W - 1, 1), dH = (float)max(H - 1, 1);
    gflo[((size_t)b * 2 + 0) * plane + pix] = (sx * (0.5f * (float)W)) / dW * 2.0f;
    gflo[((size_t)b * 2 + 1) * plane + pix] = (sy * (0.5f * (float)H)) / dH * 2.0f;
  }
}

// channel slices per pixel: enough threads for ~4 waves per SIMD on the small grids, at most 16 (and at most C)
static int warp_slices(long npix, int C) {
  int s = 1;
  while (s < 16 && s * 2 <= C && npix * s < 262144) s *= 2;
  return s;
}

}  // namespace

extern "C" int ufr_pwc_warp_forward(const float* x, const float* flow, float* out, int B, int C, int H, int W,
                                    ufr_stream_t stream) {
  UFR_REQUIRE(x && flow && out, "pwc warp forward: null pointer");
  UFR_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "pwc warp forward: bad shape");
  const long npix = (long)B * H * W;
  const int S = warp_slices(npix, C);
  pwc_warp_fwd<<<(unsigned)((npix + 63) / 64), 64 * S, 0, ufr::as_stream(stream)>>>(x, flow, out, B, C, H, W, S);
  return ufr::launched("pwc_warp_fwd");
}

extern "C" int ufr_pwc_warp_backward(const float* x, const float* flow, const float* grad_out, float* grad_x,
                                     float* grad_flow, int B, int C, int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(x && flow && grad_out && grad_x && grad_flow, "pwc warp backward: null pointer");
  UFR_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "pwc warp backward: bad shape");
  hipStream_t st = ufr::as_stream(stream);
  hipError_t e = hipMemsetAsync(grad_x, 0, sizeof(float) * (size_t)B * C * H * W, st);
  if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "pwc warp backward: memset: %s", hipGetErrorString(e));
  const long npix = (long)B * H * W;
  const int S = warp_slices(npix, C);
  pwc_warp_bwd<<<(unsigned)((npix + 63) / 64), 64 * S, 0, st>>>(x, flow, grad_out, grad_x, grad_flow, B, C, H, W, S);
  return ufr::launched("pwc_warp_bwd");
}
