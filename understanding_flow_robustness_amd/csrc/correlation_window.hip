// correlation_window.hip -- both adjoints of the spatial correlation (kernel 1, stride 1, no padding: the
// FlowNetC / PWC-Net configuration, correlation_cuda_kernel.cu:86-233) restricted to a per-sample window
// of pixels, for gfx950.
//
// With the windowed prefix of the patch attack (window.hip, patch_attack.py) only the window's cells of
// d loss/d input1 and d loss/d input2 are ever read; the full adjoint (correlation_mfma.hip, 0.71 ms at
// [8,256,48,160]) computes 30x more.  Here
//   gin1[n,c,y,x] = sum_{i,j} gout[n,i,j,y,x]           * in2[n,c,y+oy,x+ox]           (oy,ox) = DP*(i-R, j-R)
//   gin2[n,c,y,x] = sum_{i,j} gout[n,i,j,y-oy,x-ox]     * in1[n,c,y-oy,x-ox]
// for (y,x) in the window only.  Workgroup = (sample, group of CG channels, adjoint): the source region
// (window grown by DP*R cells, zero outside the image) of its channels is staged in LDS once; a thread owns
// one window cell and walks the P*P displacements: one coalesced gout read feeds CG LDS reads + FMAs.
// Everything outside the window is written as zero (the callers add these to full-size gradients).
#include "ufr_common.h"

namespace {

// CG channels share one staged region and one workgroup of 256 * CG/CT threads; a thread accumulates CT
// channels of one cell.  The CG/CT thread groups read the same gout values: the second read hits the CU's
// vector L1, so the L2 traffic for gout (the kernel's bound: every channel group re-reads the window's
// gout) shrinks by CG/CT.
constexpr int CT = 4;

template <int CG, int P, int DP>
__global__ __launch_bounds__(256 * CG / CT) void corr_bwd_window_kernel(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ gout,
    float* __restrict__ gin1, float* __restrict__ gin2, int C, int H, int W, const int* __restrict__ win,
    int level_stride, int wh, int ww) {
  extern __shared__ __attribute__((aligned(16))) float src[];      // [CG][rh][rw], zero outside the image
  const int n = blockIdx.x, cg = blockIdx.y, adj = blockIdx.z;      // adj 0: gin1 (source in2), 1: gin2 (source in1)
  constexpr int R = (P - 1) / 2, reach = DP * R;
  const int rh = wh + 2 * reach, rw = ww + 2 * reach;
  const int* w = win + n * 8;
  const int y0 = min(max(w[0] / level_stride, 0), H - wh), x0 = min(max(w[1] / level_stride, 0), W - ww);
  const size_t plane = (size_t)H * W;
  const float* s = (adj == 0 ? in2 : in1) + ((size_t)n * C + (size_t)cg * CG) * plane;
  for (int i = threadIdx.x; i < CG * rh * rw; i += blockDim.x) {
    const int rx = i % rw, ry = (i / rw) % rh, c = i / (rw * rh);
    const int yy = y0 - reach + ry, xx = x0 - reach + rx;
    src[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W && cg * CG + c < C) ? s[c * plane + (size_t)yy * W + xx] : 0.f;
  }
  __syncthreads();
  const float* g = gout + (size_t)n * P * P * plane;
  float* out = (adj == 0 ? gin1 : gin2) + ((size_t)n * C + (size_t)cg * CG) * plane;
  const int sgn = adj == 0 ? 1 : -1;
  const int cgrp = threadIdx.x / 256;                                  // which CT channels of the group
  const float* srcg = src + cgrp * CT * rh * rw;
  for (int cell = threadIdx.x % 256; cell < wh * ww; cell += 256) {
    const int ly = cell / ww, lx = cell - ly * ww;
    const int y = y0 + ly, x = x0 + lx;
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = 0.f;
    // No branches in the displacement loops: an out-of-image source reads 0 from LDS, and the matching
    // gout address (adj 1 reads gout at the source cell) is clamped into the image, so the P loads of a row
    // of displacements are independent and in flight together.
    for (int i = 0; i < P; ++i) {
      const int sy = y + sgn * DP * (i - R);
      const int ry = sy - (y0 - reach);                                   // always inside the staged region
      const int gy = adj == 0 ? y : min(max(sy, 0), H - 1);
      float gv[P];
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const int sx = x + sgn * DP * (j - R);
        const int gx = adj == 0 ? x : min(max(sx, 0), W - 1);
        gv[j] = g[(size_t)(i * P + j) * plane + (size_t)gy * W + gx];
      }
      const float* sp = srcg + ry * rw + (x - (x0 - reach));
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const int ox = sgn * DP * (j - R);
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = fmaf(gv[j], sp[c * rh * rw + ox], acc[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (cg * CG + cgrp * CT + c < C) out[(cgrp * CT + c) * plane + (size_t)y * W + x] = acc[c];
  }
}

template <int CG, int P, int DP>
int launch_window(const float* in1, const float* in2, const float* go, float* g1, float* g2, int B, int C, int H,
                  int W, const int* win, int ls, int wh, int ww, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024) {                     // raised once per size class and device, not on every (captured) launch
    hipError_t e = ufr::ensure_dynamic_lds(reinterpret_cast<const void*>(corr_bwd_window_kernel<CG, P, DP>), lds);
    if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "corr backward window: %s", hipGetErrorString(e));
  }
  corr_bwd_window_kernel<CG, P, DP><<<dim3(B, ufr::ceil_div(C, CG), 2), 256 * CG / CT, lds, st>>>(in1, in2, go, g1, g2, C, H, W,
                                                                                         win, ls, wh, ww);
  return ufr::launched("corr_bwd_window");
}

}  // namespace

extern "C" int ufr_corr_backward_window(const float* in1, const float* in2, const float* grad_output, float* gin1,
                                        float* gin2, int B, int C, int H, int W, int patch, int dilation_patch,
                                        const int* win, int level_stride, int wh, int ww, ufr_stream_t stream) {
  UFR_REQUIRE(in1 && in2 && grad_output && gin1 && gin2 && win, "corr backward window: null pointer");
  UFR_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "corr backward window: bad shape");
  UFR_REQUIRE(patch > 0 && (patch & 1) && dilation_patch > 0, "corr backward window: patch %d / dilation %d", patch,
              dilation_patch);
  UFR_REQUIRE(level_stride > 0 && wh > 0 && ww > 0 && wh <= H && ww <= W, "corr backward window: window %dx%d in %dx%d",
              wh, ww, H, W);
  const int reach = dilation_patch * (patch - 1) / 2;
  const size_t per_channel = sizeof(float) * (size_t)(wh + 2 * reach) * (ww + 2 * reach);
  // 8 channels per workgroup when LDS holds them, else 4: fewer groups re-read the window's gout
  // (measured at [8,256,48,160], 16x16 window: 4 -> 396 us, 8 -> 254 us, 12 -> 342 us: 12 waves of 170 VGPRs
  // leave one workgroup per CU without any latency hiding)
  const int CG = 8 * per_channel <= (size_t)ufr::kMaxLds ? 8 : 4;
  const size_t lds = CG * per_channel;
  if (lds > (size_t)ufr::kMaxLds) return ufr::fail(UFR_EUNSUPPORTED, "corr backward window: %zu bytes of LDS", lds);
  hipStream_t st = ufr::as_stream(stream);
  const size_t bytes = sizeof(float) * (size_t)B * C * H * W;
  hipError_t e = hipMemsetAsync(gin1, 0, bytes, st);
  if (e == hipSuccess) e = hipMemsetAsync(gin2, 0, bytes, st);
  if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "corr backward window: %s", hipGetErrorString(e));
#define UFR_WINDOW_CASE(cg, p, dp)                                                                             \
  if (CG == cg && patch == p && dilation_patch == dp)                                                          \
    return launch_window<cg, p, dp>(in1, in2, grad_output, gin1, gin2, B, C, H, W, win, level_stride, wh, ww, lds, st);
  UFR_WINDOW_CASE(8, 21, 2) UFR_WINDOW_CASE(4, 21, 2)
  UFR_WINDOW_CASE(8, 9, 1) UFR_WINDOW_CASE(4, 9, 1)
#undef UFR_WINDOW_CASE
  return ufr::fail(UFR_EUNSUPPORTED, "corr backward window: patch %d / dilation %d (21/2 and 9/1 are built)", patch,
                   dilation_patch);
}
