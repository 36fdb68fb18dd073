// convex_upsample.hip -- RAFT's learned 8x flow upsampling (models/raft/raft.py:111-122) for gfx950.
//
//   up[n,c,8h+i,8w+j] = sum_k softmax_k(mask[n, k*64 + i*8 + j, h, w]) * 8 * flow[n,c,h+ky-1,w+kx-1],  k = 3*ky + kx
// The reference spells it as view + softmax + unfold + mul + sum + permute + reshape (seven kernels and a
// 9x-expanded temporary); here it is one pass over the 576-channel mask forward and two kernels backward.
// HBM-streaming: a workgroup takes 32 consecutive coarse columns of one row and walks the 64 sub-pixels,
// 8 at a time, so the mask (the only large operand, 9*64 floats per coarse pixel) is read in 128-byte runs.
// Backward: kernel A recomputes the softmax, writes d loss/d mask and reduces, per coarse pixel, the 64
// sub-pixel terms of  a[k][c] = sum_ij softmax_k * g_up[c]  in registers + LDS (fixed order, no atomics);
// kernel B gathers d loss/d flow[n,c,h',w'] = 8 * sum_k a[k][c] at (h'-ky+1, w'-kx+1).
#include "ufr_common.h"

namespace {

constexpr int CU_W = 32;          // coarse columns per workgroup
constexpr int CU_NT = 256;        // 8 sub-pixels x 32 columns per pass, 8 passes

__device__ __forceinline__ void softmax9(const float* __restrict__ m, size_t stride, float* s) {
  float mx = m[0];
#pragma unroll
  for (int k = 1; k < 9; ++k) mx = fmaxf(mx, m[k * stride]);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) { s[k] = expf(m[k * stride] - mx); sum += s[k]; }
  const float inv = 1.0f / sum;
#pragma unroll
  for (int k = 0; k < 9; ++k) s[k] *= inv;
}

// 8 * flow at the 3x3 neighbours (zero padding, like F.unfold(padding=1))
__device__ __forceinline__ void neighbours(const float* __restrict__ flow_nc, int h, int w, int H, int W, float* f) {
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int hh = h + k / 3 - 1, ww = w + k % 3 - 1;
    f[k] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? 8.0f * flow_nc[hh * W + ww] : 0.f;
  }
}

__global__ __launch_bounds__(CU_NT) void convex_up_fwd(const float* __restrict__ flow, const float* __restrict__ mask,
                                                       float* __restrict__ up, int H, int W) {
  const int wl = threadIdx.x & (CU_W - 1), sub = threadIdx.x / CU_W;       // 8 sub-pixel lanes
  const int w = blockIdx.x * CU_W + wl, h = blockIdx.y, n = blockIdx.z;
  if (w >= W) return;
  const size_t plane = (size_t)H * W;
  float f0[9], f1[9];
  neighbours(flow + ((size_t)n * 2 + 0) * plane, h, w, H, W, f0);
  neighbours(flow + ((size_t)n * 2 + 1) * plane, h, w, H, W, f1);
  const float* mp = mask + (size_t)n * 576 * plane + (size_t)h * W + w;
  float* u0 = up + ((size_t)n * 2 + 0) * plane * 64;
  float* u1 = up + ((size_t)n * 2 + 1) * plane * 64;
  for (int t = 0; t < 8; ++t) {
    const int ij = t * 8 + sub, i = ij >> 3, j = ij & 7;
    float s[9];
    softmax9(mp + (size_t)ij * plane, 64 * plane, s);
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { a0 = fmaf(s[k], f0[k], a0); a1 = fmaf(s[k], f1[k], a1); }
    const size_t o = (size_t)(8 * h + i) * (8 * W) + 8 * w + j;
    u0[o] = a0; u1[o] = a1;
  }
}

// g_mask [N,576,H,W]; a [N,2,9,H,W]
__global__ __launch_bounds__(CU_NT) void convex_up_bwd_a(const float* __restrict__ flow, const float* __restrict__ mask,
                                                         const float* __restrict__ g_up, float* __restrict__ g_mask,
                                                         float* __restrict__ a_out, int H, int W) {
  __shared__ float part[8][18][CU_W];
  const int wl = threadIdx.x & (CU_W - 1), sub = threadIdx.x / CU_W;
  const int w = blockIdx.x * CU_W + wl, h = blockIdx.y, n = blockIdx.z;
  const size_t plane = (size_t)H * W;
  float acc[18];
#pragma unroll
  for (int q = 0; q < 18; ++q) acc[q] = 0.f;
  if (w < W) {
    float f0[9], f1[9];
    neighbours(flow + ((size_t)n * 2 + 0) * plane, h, w, H, W, f0);
    neighbours(flow + ((size_t)n * 2 + 1) * plane, h, w, H, W, f1);
    const float* mp = mask + (size_t)n * 576 * plane + (size_t)h * W + w;
    float* gm = g_mask + (size_t)n * 576 * plane + (size_t)h * W + w;
    const float* gu0 = g_up + ((size_t)n * 2 + 0) * plane * 64;
    const float* gu1 = g_up + ((size_t)n * 2 + 1) * plane * 64;
    for (int t = 0; t < 8; ++t) {
      const int ij = t * 8 + sub, i = ij >> 3, j = ij & 7;
      float s[9];
      softmax9(mp + (size_t)ij * plane, 64 * plane, s);
      const size_t o = (size_t)(8 * h + i) * (8 * W) + 8 * w + j;
      const float g0 = gu0[o], g1 = gu1[o];
      float d[9], dbar = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) { d[k] = g0 * f0[k] + g1 * f1[k]; dbar = fmaf(s[k], d[k], dbar); }
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        gm[((size_t)k * 64 + ij) * plane] = s[k] * (d[k] - dbar);          // softmax adjoint
        acc[k] = fmaf(s[k], g0, acc[k]);
        acc[9 + k] = fmaf(s[k], g1, acc[9 + k]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 18; ++q) part[sub][q][wl] = acc[q];
  __syncthreads();
  for (int e = threadIdx.x; e < 18 * CU_W; e += CU_NT) {
    const int q = e / CU_W, l = e % CU_W, ww = blockIdx.x * CU_W + l;
    if (ww >= W) continue;
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) v += part[g][q][l];
    a_out[(((size_t)n * 2 + q / 9) * 9 + q % 9) * plane + (size_t)h * W + ww] = v;
  }
}

__global__ void convex_up_bwd_b(const float* __restrict__ a_in, float* __restrict__ g_flow, int N, int H, int W) {
  const size_t plane = (size_t)H * W;
  const long total = (long)N * 2 * plane;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int w = (int)(i % W), h = (int)((i / W) % H);
    const long nc = i / (long)plane;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {                      // the coarse pixel whose k-th neighbour is (h, w)
      const int hh = h - (k / 3 - 1), ww = w - (k % 3 - 1);
      if (hh >= 0 && hh < H && ww >= 0 && ww < W) v += a_in[((size_t)nc * 9 + k) * plane + (size_t)hh * W + ww];
    }
    g_flow[i] = 8.0f * v;
  }
}

}  // namespace

extern "C" int ufr_convex_upsample_forward(const float* flow, const float* mask, float* up, int N, int H, int W,
                                           ufr_stream_t stream) {
  UFR_REQUIRE(flow && mask && up, "convex upsample forward: null pointer");
  UFR_REQUIRE(N > 0 && N <= 65535 && H > 0 && H <= 65535 && W > 0, "convex upsample forward: bad shape");
  convex_up_fwd<<<dim3(ufr::ceil_div(W, CU_W), H, N), CU_NT, 0, ufr::as_stream(stream)>>>(flow, mask, up, H, W);
  return ufr::launched("convex_up_fwd");
}

extern "C" int ufr_convex_upsample_backward(const float* flow, const float* mask, const float* grad_up, float* grad_flow,
                                            float* grad_mask, float* workspace, int N, int H, int W,
                                            ufr_stream_t stream) {
  UFR_REQUIRE(flow && mask && grad_up && grad_flow && grad_mask && workspace, "convex upsample backward: null pointer");
  UFR_REQUIRE(N > 0 && N <= 65535 && H > 0 && H <= 65535 && W > 0, "convex upsample backward: bad shape");
  hipStream_t st = ufr::as_stream(stream);
  convex_up_bwd_a<<<dim3(ufr::ceil_div(W, CU_W), H, N), CU_NT, 0, st>>>(flow, mask, grad_up, grad_mask, workspace, H, W);
  const long total = (long)N * 2 * H * W;
  convex_up_bwd_b<<<ufr::stream_grid(total, 256), 256, 0, st>>>(workspace, grad_flow, N, H, W);
  return ufr::launched("convex_up_bwd");
}
