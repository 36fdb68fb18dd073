// plane_layout.hip -- the layout passes at the edges of the plane layout that csrc/igemm.hip computes in:
// NCHW float32 <-> chunk-major bf16 split planes / float32 gradient sums, with the elementwise work of the boundary fused
// (scale, bias, LeakyReLU, LeakyReLU' of a mask), the packed forms of the 7x7 stride-2 stems (FlowNetC's conv1 with its
// float64 mean subtraction; the generic 2x2 pixel-unshuffle), concatenations of NCHW members (PWC-Net's stage input) and
// the window scatter of the attack step.  One pass over HBM each; 32-channel x 64-pixel tiles through LDS so that both
// sides are contiguous.
#include "ufr_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ void split3(float v, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)v;
  const float r1 = v - (float)a;
  b = (__bf16)r1;
  c = (__bf16)(r1 - (float)b);
}

// ---- layout passes at the engine's edges ---------------------------------------------------------------------------
// x [B][C][HW] float32 (NCHW) -> planes at a chunk offset, y = leaky(scale * x); channels C..Cpad-1 of the last chunk 0.
__global__ __launch_bounds__(256) void nchw_to_planes_kernel(const float* __restrict__ x, __bf16* __restrict__ planes,
                                                             long plane_stride, int chunk0, int B, int C, int HW,
                                                             float scale, float slope, const float* __restrict__ bias,
                                                             const float* __restrict__ act) {
  // act != nullptr: x is a GRADIENT and act the activation it passes through: y = x * LeakyReLU'(act) (no scale / bias)
  __shared__ float tile[32][65];
  const int c0 = blockIdx.x * 32, p0 = blockIdx.y * 64, b = blockIdx.z, tid = threadIdx.x;
  {
    const int p = tid & 63;
#pragma unroll
    for (int cc = tid >> 6; cc < 32; cc += 4) {
      const int c = c0 + cc;
      float v = 0.f;
      if (c < C && p0 + p < HW) {
        const size_t e = ((size_t)b * C + c) * HW + p0 + p;
        v = x[e] * scale;
        if (act) {
          v = act[e] > 0.f ? v : v * slope;
        } else {
          if (bias) v += bias[c];
          v = v > 0.f ? v : v * slope;
        }
      }
      tile[cc][p] = v;
    }
  }
  __syncthreads();
  const int p = tid >> 2, ch = tid & 3;
  if (p0 + p >= HW) return;
  const size_t rows = (size_t)B * HW;
  __bf16* dst = planes + (((size_t)chunk0 + blockIdx.x) * rows + (size_t)b * HW + p0 + p) * 32 + ch * 8;
  bf16x8 q0, q1, q2;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    __bf16 a, bq, c;
    split3(tile[ch * 8 + j][p], a, bq, c);
    q0[j] = a; q1[j] = bq; q2[j] = c;
  }
  *reinterpret_cast<bf16x8*>(dst) = q0;
  *reinterpret_cast<bf16x8*>(dst + plane_stride) = q1;
  *reinterpret_cast<bf16x8*>(dst + 2 * plane_stride) = q2;
}

// A ROW-MAJOR float32 matrix src [rows][ld] (its first `cols` columns) -> planes[chunk0 + k / 32][m][k % 32] = split(scale * src[m][k]):
// the operand layout of a GEMM whose reduction runs over the matrix's COLUMNS (RAFT's all-pairs adjoint: the volume's gradient
// g[p][q] reduced over q, the feature maps [C][HW] as weight images reduced over the pixels).  The reduction over the ROWS is
// nchw_to_planes_kernel on the same matrix read as [channels = rows][pixels = columns].  Workgroup = 64 rows x one chunk: a row's
// 128 bytes in, 4 KB contiguous per plane out; columns >= cols of the last chunk are written as zeros.
__global__ __launch_bounds__(256) void rowmajor_to_planes_kernel(const float* __restrict__ src, long ld, long rows, int cols, float scale,
                                                                 __bf16* __restrict__ planes, long plane_stride, int chunk0,
                                                                 long plane_rows) {
  const int tid = threadIdx.x, p = tid >> 2, ch = tid & 3;
  const long m = (long)blockIdx.y * 64 + p;
  if (m >= rows) return;
  const int k0 = blockIdx.x * 32 + ch * 8;
  const float* s = src + m * ld + k0;
  float v[8];
  if (k0 + 8 <= cols && ((reinterpret_cast<size_t>(s) & 15) == 0)) {
    const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = k0 + j < cols ? s[j] : 0.f;
  }
  bf16x8 q0, q1, q2;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    __bf16 a, b, c;
    split3(v[j] * scale, a, b, c);
    q0[j] = a; q1[j] = b; q2[j] = c;
  }
  __bf16* dst = planes + (((size_t)chunk0 + blockIdx.x) * plane_rows + m) * 32 + ch * 8;
  *reinterpret_cast<bf16x8*>(dst) = q0;
  *reinterpret_cast<bf16x8*>(dst + plane_stride) = q1;
  *reinterpret_cast<bf16x8*>(dst + 2 * plane_stride) = q2;
}

// ---- conv1 (Conv2d(3, 64, 7, 2, 3), models/FlowNetC.py:22) as an igemm launch --------------------------------------------
// A 3-channel pixel would leave 29 of a chunk's 32 channels empty.  The frames are therefore written as PACKED planes: the
// 2 x 2 pixel-unshuffle turns the stride-2 7 x 7 convolution into a stride-1 4 x 4 one over 12 channels on the half grid,
// and two horizontally adjacent half-grid pixels share one chunk (24 of 32 channels), so the convolution is 4 x 2 = 8 taps
// of ONE chunk.  The buffer carries the zero padding physically (2 rows above, 1 below, 2 columns to the left, the j = 1 half of the last column): every tap
// is in range.  channel j*12 + (c*2 + p)*2 + q of packed pixel (yp, xp) = frame[c, 2 (yp - 2) + p, 2 (xp - 2 + j) + q] - mean[c]
// (the float64 mean subtraction of normalize_correctly, FlowNetC.py:73-79, fused), zero outside the frame.
__global__ __launch_bounds__(256) void conv1_pack_kernel(const float* __restrict__ fa, const float* __restrict__ fb, int Ba,
                                                         __bf16* __restrict__ planes, long plane_stride, int N, int H, int W,
                                                         const double* __restrict__ mean) {
  const int Hh = H >> 1, Wh = W >> 1, Hp = Hh + 3, Wp = Wh + 2;
  const long total = (long)N * Hp * Wp * 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i & 3);
    const long pix = i >> 2;
    const int xp = (int)(pix % Wp), yp = (int)((pix / Wp) % Hp), n = (int)(pix / ((long)Wp * Hp));
    const float* img = n < Ba ? fa + (long)n * 3 * H * W : fb + (long)(n - Ba) * 3 * H * W;
    bf16x8 q0, q1, q2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = k * 8 + e;
      float v = 0.f;
      if (ch < 24) {
        const int j = ch / 12, r = ch - 12 * j, c = r >> 2, p = (r >> 1) & 1, q = r & 1;
        const int yh = yp - 2, xh = xp - 2 + j;
        if (yh >= 0 && yh < Hh && xh >= 0 && xh < Wh)
          v = (float)((double)img[((long)c * H + 2 * yh + p) * W + 2 * xh + q] - mean[c]);
      }
      __bf16 a, b2, c2;
      split3(v, a, b2, c2);
      q0[e] = a; q1[e] = b2; q2[e] = c2;
    }
    __bf16* dst = planes + pix * 32 + k * 8;
    *reinterpret_cast<bf16x8*>(dst) = q0;
    *reinterpret_cast<bf16x8*>(dst + plane_stride) = q1;
    *reinterpret_cast<bf16x8*>(dst + 2 * plane_stride) = q2;
  }
}

// Adjoint of conv1_pack_kernel: the gradient with respect to the raw frames from the float32 gradient sum of the packed
// planes G [1][N * Hp * Wp][32]: a frame pixel sits in two packed pixels (j = 0 and j = 1 of the column pair).
__global__ __launch_bounds__(256) void conv1_unpack_grad_kernel(const float* __restrict__ G, float* __restrict__ gx, int N, int H,
                                                                int W) {
  const int Hh = H >> 1, Wh = W >> 1, Hp = Hh + 3, Wp = Wh + 2;
  const long total = (long)N * 3 * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H), c = (int)((i / ((long)W * H)) % 3), n = (int)(i / ((long)3 * W * H));
    const int r = (c * 2 + (y & 1)) * 2 + (x & 1);
    const long row = ((long)n * Hp + (y >> 1) + 2) * Wp + (x >> 1);
    gx[i] = G[(row + 2) * 32 + r] + G[(row + 1) * 32 + 12 + r];
  }
}

// 2x2 pixel-unshuffle of x [N, C, 2H, 2W] into planes [N, H, W, 4C channels]: channel (c*2 + p)*2 + q = x[c, 2y + p, 2x + q].  A
// stride-2 7x7 convolution of x is a stride-1 4x4 one over these planes (FlowNetS's 12-channel stem: 49 taps of one chunk would
// exceed the tap table, 16 taps of 48 channels do not; plane_graph.py).  Thread = (pixel, 8-channel group).
__global__ __launch_bounds__(256) void unshuffle_pack_kernel(const float* __restrict__ x, __bf16* __restrict__ planes, long plane_stride,
                                                             int N, int C, int H, int W, int chunks) {
  const long M = (long)N * H * W, total = M * chunks * 4;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int g8 = (int)(t % (chunks * 4));
    const long m = t / (chunks * 4);
    const int xx = (int)(m % W), yy = (int)((m / W) % H), n = (int)(m / ((long)W * H));
    bf16x8 q0, q1, q2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = g8 * 8 + j, c = ch >> 2, p = (ch >> 1) & 1, q = ch & 1;
      const float v = c < C ? x[(((long)n * C + c) * (2 * H) + 2 * yy + p) * (2 * W) + 2 * xx + q] : 0.f;
      __bf16 a, b, d;
      split3(v, a, b, d);
      q0[j] = a; q1[j] = b; q2[j] = d;
    }
    __bf16* dst = planes + ((long)(g8 >> 2) * M + m) * 32 + (g8 & 3) * 8;
    *reinterpret_cast<bf16x8*>(dst) = q0;
    *reinterpret_cast<bf16x8*>(dst + plane_stride) = q1;
    *reinterpret_cast<bf16x8*>(dst + 2 * plane_stride) = q2;
  }
}

// its adjoint: gx[n, c, 2y + p, 2x + q] = G[(c*2 + p)*2 + q] of pixel (n, y, x), G float32 [chunks][M][32]
__global__ __launch_bounds__(256) void unshuffle_unpack_grad_kernel(const float* __restrict__ G, float* __restrict__ gx, int N, int C, int H,
                                                                    int W) {
  const long M = (long)N * H * W, total = (long)N * C * 4 * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int X = (int)(i % (2 * W)), Y = (int)((i / (2 * W)) % (2 * H)), c = (int)((i / ((long)4 * W * H)) % C), n = (int)(i / ((long)4 * W * H * C));
    const int ch = (c * 2 + (Y & 1)) * 2 + (X & 1);
    const long m = ((long)n * H + (Y >> 1)) * W + (X >> 1);
    gx[i] = G[((long)(ch >> 5) * M + m) * 32 + (ch & 31)];
  }
}

// chunk-major tensor (planes: p0 + p1 + p2, or fp32) -> out [B][C][HW] float32 (NCHW), optionally
// out = scale * leaky'(mask) * v with `mask` = plane 0 of an activation in the same chunk-major geometry.
__global__ __launch_bounds__(256) void chunks_to_nchw_kernel(const __bf16* __restrict__ planes, long plane_stride,
                                                             const float* __restrict__ f32, int chunk0,
                                                             const __bf16* __restrict__ mask, int mask_chunk0,
                                                             float* __restrict__ out, int B, int C, int HW, float scale,
                                                             float slope) {
  __shared__ float tile[32][65];
  const int c0 = blockIdx.x * 32, p0 = blockIdx.y * 64, b = blockIdx.z, tid = threadIdx.x;
  const size_t rows = (size_t)B * HW;
  {
    const int p = tid >> 2, q = tid & 3;
    if (p0 + p < HW) {
      const size_t pix = (size_t)b * HW + p0 + p;
      const size_t o = (((size_t)chunk0 + blockIdx.x) * rows + pix) * 32 + q * 8;
      float v[8];
      if (planes) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(planes + o);
        const bf16x8 bq = *reinterpret_cast<const bf16x8*>(planes + o + plane_stride);
        const bf16x8 c = *reinterpret_cast<const bf16x8*>(planes + o + 2 * plane_stride);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ((float)a[j] + (float)bq[j]) + (float)c[j];
      } else {
        const float4 a = *reinterpret_cast<const float4*>(f32 + o), bq = *reinterpret_cast<const float4*>(f32 + o + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = bq.x; v[5] = bq.y; v[6] = bq.z; v[7] = bq.w;
      }
      if (mask) {
        const bf16x8 m = *reinterpret_cast<const bf16x8*>(mask + (((size_t)mask_chunk0 + blockIdx.x) * rows + pix) * 32 + q * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ((float)m[j] > 0.f) ? v[j] : v[j] * slope;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) tile[q * 8 + j][p] = v[j] * scale;
    }
  }
  __syncthreads();
  const int p = tid & 63;
  if (p0 + p >= HW) return;
#pragma unroll
  for (int cc = tid >> 6; cc < 32; cc += 4) {
    const int c = c0 + cc;
    if (c < C) out[((size_t)b * C + c) * HW + p0 + p] = tile[cc][p];
  }
}

// ---- a concatenation of NCHW tensors <-> chunks (PWC-Net's stage input x = cat(corr, up_flow, up_feat | c1), PWCNet.py:287) ----
// Up to four NCHW float32 members, member s covering buffer channels [dst0[s], dst0[s] + C[s]); channels no member covers are 0.
struct CatSegs {
  int n;
  const float* src[4];      // forward: the members; backward: their gradient outputs (written), cast away below
  int C[4], dst0[4];
};

__global__ __launch_bounds__(256) void nchw_cat_to_planes_kernel(const CatSegs segs, __bf16* __restrict__ planes, long plane_stride,
                                                                 int chunk0, int B, int HW) {
  __shared__ float tile[32][65];
  const int c0 = blockIdx.x * 32, p0 = blockIdx.y * 64, b = blockIdx.z, tid = threadIdx.x;
  {
    const int p = tid & 63;
#pragma unroll
    for (int cc = tid >> 6; cc < 32; cc += 4) {
      const int c = c0 + cc;
      float v = 0.f;
      if (p0 + p < HW) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (s < segs.n && c >= segs.dst0[s] && c < segs.dst0[s] + segs.C[s])
            v = segs.src[s][((size_t)b * segs.C[s] + (c - segs.dst0[s])) * HW + p0 + p];
      }
      tile[cc][p] = v;
    }
  }
  __syncthreads();
  const int p = tid >> 2, ch = tid & 3;
  if (p0 + p >= HW) return;
  const size_t rows = (size_t)B * HW;
  __bf16* dst = planes + (((size_t)chunk0 + blockIdx.x) * rows + (size_t)b * HW + p0 + p) * 32 + ch * 8;
  bf16x8 q0, q1, q2;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    __bf16 a, bq, c;
    split3(tile[ch * 8 + j][p], a, bq, c);
    q0[j] = a; q1[j] = bq; q2[j] = c;
  }
  *reinterpret_cast<bf16x8*>(dst) = q0;
  *reinterpret_cast<bf16x8*>(dst + plane_stride) = q1;
  *reinterpret_cast<bf16x8*>(dst + 2 * plane_stride) = q2;
}

// The adjoint: a float32 chunk-major gradient sum -> the members' NCHW gradients.  Member 0 may pass through the activation
// it carries in the forward (act0 = its NCHW activation: g * (act0 > 0 ? pos0 : neg0); PWC-Net: correlate's / C and LeakyReLU').
__global__ __launch_bounds__(256) void chunks_to_nchw_cat_kernel(const float* __restrict__ g, int chunk0, const CatSegs segs,
                                                                 const float* __restrict__ act0, float pos0, float neg0, int B, int HW) {
  __shared__ float tile[32][65];
  const int c0 = blockIdx.x * 32, p0 = blockIdx.y * 64, b = blockIdx.z, tid = threadIdx.x;
  const size_t rows = (size_t)B * HW;
  {
    const int p = tid >> 2, q = tid & 3;
    if (p0 + p < HW) {
      const size_t o = (((size_t)chunk0 + blockIdx.x) * rows + (size_t)b * HW + p0 + p) * 32 + q * 8;
      const float4 a = *reinterpret_cast<const float4*>(g + o), bq = *reinterpret_cast<const float4*>(g + o + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) tile[q * 8 + j][p] = v[j];
    }
  }
  __syncthreads();
  const int p = tid & 63;
  if (p0 + p >= HW) return;
#pragma unroll
  for (int cc = tid >> 6; cc < 32; cc += 4) {
    const int c = c0 + cc;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (s < segs.n && c >= segs.dst0[s] && c < segs.dst0[s] + segs.C[s]) {
        const size_t e = ((size_t)b * segs.C[s] + (c - segs.dst0[s])) * HW + p0 + p;
        float v = tile[cc][p];
        if (s == 0 && act0) v *= act0[e] > 0.f ? pos0 : neg0;
        const_cast<float*>(segs.src[s])[e] = v;
      }
  }
}

// fp32 chunk-major gradient sum -> gradient planes: g * leaky'(mask), chunk by chunk (a gradient with several sources
// whose last writer is not a GEMM epilogue).
__global__ __launch_bounds__(256) void grad_finalize_kernel(const float* __restrict__ g, int g_chunk0,
                                                            const __bf16* __restrict__ mask, int mask_chunk0,
                                                            __bf16* __restrict__ out, long out_plane_stride, int out_chunk0,
                                                            long M, int chunks, float slope) {
  const long total = (long)chunks * M * 4;            // 8 channels per thread
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long e = i * 8;                               // element inside the [chunks][M][32] tensor
    const float4 a = *reinterpret_cast<const float4*>(g + (long)g_chunk0 * M * 32 + e);
    const float4 b = *reinterpret_cast<const float4*>(g + (long)g_chunk0 * M * 32 + e + 4);
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    if (mask) {
      const bf16x8 m = *reinterpret_cast<const bf16x8*>(mask + (long)mask_chunk0 * M * 32 + e);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ((float)m[j] > 0.f) ? v[j] : v[j] * slope;
    }
    bf16x8 q0, q1, q2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __bf16 x, y, z;
      split3(v[j], x, y, z);
      q0[j] = x; q1[j] = y; q2[j] = z;
    }
    __bf16* o = out + (long)out_chunk0 * M * 32 + e;
    *reinterpret_cast<bf16x8*>(o) = q0;
    *reinterpret_cast<bf16x8*>(o + out_plane_stride) = q1;
    *reinterpret_cast<bf16x8*>(o + 2 * out_plane_stride) = q2;
  }
}

// dst planes [chunk0 + c/32][(n, y0+i, x0+j)][c%32] = split(src[n,c,i,j]) for the window of sample n (origin win[n % n_win] /
// level_stride, clamped into the frame; rim cells next to an interior window edge are skipped, as in ufr_window_scatter):
// the cached full-frame features live in the plane layout, the windowed prefix's results are patched into them.
__global__ __launch_bounds__(256) void window_scatter_planes_kernel(const float* __restrict__ src, __bf16* __restrict__ planes,
                                                                    long plane_stride, int chunk0, const int* __restrict__ win,
                                                                    int n_win, int N, int C, int Hd, int Wd, int wh, int ww,
                                                                    int level_stride, int margin) {
  const int groups = (C + 7) / 8;
  const long total = (long)N * wh * ww * groups;
  const long M = (long)N * Hd * Wd;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int g8 = (int)(t % groups);
    long r = t / groups;
    const int j = (int)(r % ww); r /= ww;
    const int i = (int)(r % wh);
    const int n = (int)(r / wh);
    const int* w = win + (n % n_win) * 8;
    const int y0 = min(max(w[0] / level_stride, 0), Hd - wh), x0 = min(max(w[1] / level_stride, 0), Wd - ww);
    const bool rim = (i < margin && y0 > 0) || (i >= wh - margin && y0 + wh < Hd) || (j < margin && x0 > 0) ||
                     (j >= ww - margin && x0 + ww < Wd);
    if (rim) continue;
    bf16x8 q0, q1, q2;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = g8 * 8 + k;
      const float v = c < C ? src[(((long)n * C + c) * wh + i) * ww + j] : 0.f;
      __bf16 a, b, d;
      split3(v, a, b, d);
      q0[k] = a; q1[k] = b; q2[k] = d;
    }
    const int c0 = g8 * 8;
    __bf16* o = planes + (((long)(chunk0 + (c0 >> 5)) * M) + ((long)n * Hd + y0 + i) * Wd + x0 + j) * 32 + (c0 & 31);
    *reinterpret_cast<bf16x8*>(o) = q0;
    *reinterpret_cast<bf16x8*>(o + plane_stride) = q1;
    *reinterpret_cast<bf16x8*>(o + 2 * plane_stride) = q2;
  }
}


}  // namespace

extern "C" int ufr_window_scatter_planes(const float* src, void* planes, long plane_stride, int chunk0, const int* win,
                                         int n_win, int N, int C, int Hd, int Wd, int wh, int ww, int level_stride, int margin,
                                         ufr_stream_t stream) {
  UFR_REQUIRE(src && planes && win, "window scatter (planes): null pointer");
  UFR_REQUIRE(N > 0 && C > 0 && Hd > 0 && Wd > 0 && wh > 0 && ww > 0 && wh <= Hd && ww <= Wd && chunk0 >= 0,
              "window scatter (planes): bad shape");
  UFR_REQUIRE(level_stride > 0 && margin >= 0 && 2 * margin <= wh && 2 * margin <= ww && n_win > 0 && n_win <= N,
              "window scatter (planes): bad stride / margin / window count");
  const long total = (long)N * wh * ww * ((C + 7) / 8);
  window_scatter_planes_kernel<<<ufr::stream_grid(total, 256), 256, 0, ufr::as_stream(stream)>>>(
      src, static_cast<__bf16*>(planes), plane_stride, chunk0, win, n_win, N, C, Hd, Wd, wh, ww, level_stride, margin);
  return ufr::launched("window_scatter_planes_kernel");
}

extern "C" int ufr_nchw_to_planes(const float* x, void* planes, long plane_stride, int chunk0, int B, int C, int H, int W,
                                  float scale, float slope, const float* bias, ufr_stream_t stream) {
  UFR_REQUIRE(x && planes, "nchw -> planes: null pointer");
  UFR_REQUIRE(B > 0 && B < 65536 && C > 0 && H > 0 && W > 0 && chunk0 >= 0 && plane_stride > 0, "nchw -> planes: bad shape");
  const dim3 grid((C + 31) / 32, (H * W + 63) / 64, B);
  nchw_to_planes_kernel<<<grid, 256, 0, ufr::as_stream(stream)>>>(x, static_cast<__bf16*>(planes), plane_stride, chunk0, B, C,
                                                                   H * W, scale, slope, bias, nullptr);
  return ufr::launched("nchw_to_planes_kernel");
}

extern "C" int ufr_rowmajor_to_planes(const float* src, long ld, long rows, int cols, float scale, void* planes, long plane_stride,
                                      int chunk0, long plane_rows, ufr_stream_t stream) {
  UFR_REQUIRE(src && planes, "row-major -> planes: null pointer");
  UFR_REQUIRE(rows > 0 && cols > 0 && ld >= cols && chunk0 >= 0 && plane_rows >= rows && rows < (1L << 31) * 64, "row-major -> planes: bad shape");
  const int chunks = (cols + 31) / 32;
  UFR_REQUIRE((long)(chunk0 + chunks) * plane_rows * 32 <= plane_stride, "row-major -> planes: chunks [%d, %d) leave the planes operand",
              chunk0, chunk0 + chunks);
  const dim3 grid(chunks, (unsigned)((rows + 63) / 64));
  UFR_REQUIRE(grid.y < 65536, "row-major -> planes: too many rows for one launch (%ld)", rows);
  rowmajor_to_planes_kernel<<<grid, 256, 0, ufr::as_stream(stream)>>>(src, ld, rows, cols, scale, static_cast<__bf16*>(planes),
                                                                       plane_stride, chunk0, plane_rows);
  return ufr::launched("rowmajor_to_planes_kernel");
}

extern "C" int ufr_nchw_grad_to_planes(const float* grad, const float* act, void* planes, long plane_stride, int chunk0, int B, int C,
                                       int H, int W, float slope, ufr_stream_t stream) {
  UFR_REQUIRE(grad && act && planes, "nchw gradient -> planes: null pointer");
  UFR_REQUIRE(B > 0 && B < 65536 && C > 0 && H > 0 && W > 0 && chunk0 >= 0 && plane_stride > 0, "nchw gradient -> planes: bad shape");
  const dim3 grid((C + 31) / 32, (H * W + 63) / 64, B);
  nchw_to_planes_kernel<<<grid, 256, 0, ufr::as_stream(stream)>>>(grad, static_cast<__bf16*>(planes), plane_stride, chunk0, B, C,
                                                                   H * W, 1.0f, slope, nullptr, act);
  return ufr::launched("nchw_to_planes_kernel");
}

extern "C" int ufr_conv1_pack_planes(const float* frames_a, const float* frames_b, void* planes, long plane_stride, int Ba, int Bb,
                                     int H, int W, const double* mean, ufr_stream_t stream) {
  UFR_REQUIRE(frames_a && planes && mean && (frames_b || Bb == 0), "conv1 pack: null pointer");
  UFR_REQUIRE(Ba > 0 && Bb >= 0 && H > 0 && W > 0 && !(H & 1) && !(W & 1) && plane_stride > 0, "conv1 pack: bad shape");
  const long total = (long)(Ba + Bb) * ((H >> 1) + 3) * ((W >> 1) + 2) * 4;
  conv1_pack_kernel<<<ufr::stream_grid(total, 256), 256, 0, ufr::as_stream(stream)>>>(frames_a, frames_b, Ba,
                                                                                      static_cast<__bf16*>(planes), plane_stride,
                                                                                      Ba + Bb, H, W, mean);
  return ufr::launched("conv1_pack_kernel");
}

extern "C" int ufr_conv1_unpack_grad(const float* G, float* grad_frames, int N, int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(G && grad_frames, "conv1 unpack: null pointer");
  UFR_REQUIRE(N > 0 && H > 0 && W > 0 && !(H & 1) && !(W & 1), "conv1 unpack: bad shape");
  const long total = (long)N * 3 * H * W;
  conv1_unpack_grad_kernel<<<ufr::stream_grid(total, 256), 256, 0, ufr::as_stream(stream)>>>(G, grad_frames, N, H, W);
  return ufr::launched("conv1_unpack_grad_kernel");
}

extern "C" int ufr_unshuffle_pack_planes(const float* x, void* planes, long plane_stride, int N, int C, int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(x && planes && N > 0 && C > 0 && H > 0 && W > 0 && plane_stride > 0, "unshuffle pack: bad argument");
  const int chunks = (4 * C + 31) / 32;
  unshuffle_pack_kernel<<<ufr::stream_grid((long)N * H * W * chunks * 4, 256), 256, 0, ufr::as_stream(stream)>>>(
      x, static_cast<__bf16*>(planes), plane_stride, N, C, H, W, chunks);
  return ufr::launched("unshuffle_pack_kernel");
}

extern "C" int ufr_unshuffle_unpack_grad(const float* G, float* grad_x, int N, int C, int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(G && grad_x && N > 0 && C > 0 && H > 0 && W > 0, "unshuffle unpack: bad argument");
  unshuffle_unpack_grad_kernel<<<ufr::stream_grid((long)N * C * 4 * H * W, 256), 256, 0, ufr::as_stream(stream)>>>(G, grad_x, N, C, H, W);
  return ufr::launched("unshuffle_unpack_grad_kernel");
}

extern "C" int ufr_chunks_to_nchw(const void* planes, long plane_stride, const float* f32, int chunk0, const void* mask,
                                  int mask_chunk0, float* out, int B, int C, int H, int W, float scale, float slope,
                                  ufr_stream_t stream) {
  UFR_REQUIRE((planes != nullptr) != (f32 != nullptr) && out, "chunks -> nchw: exactly one source");
  UFR_REQUIRE(B > 0 && B < 65536 && C > 0 && H > 0 && W > 0 && chunk0 >= 0 && mask_chunk0 >= 0, "chunks -> nchw: bad shape");
  const dim3 grid((C + 31) / 32, (H * W + 63) / 64, B);
  chunks_to_nchw_kernel<<<grid, 256, 0, ufr::as_stream(stream)>>>(static_cast<const __bf16*>(planes), plane_stride, f32, chunk0,
                                                                   static_cast<const __bf16*>(mask), mask_chunk0, out, B, C,
                                                                   H * W, scale, slope);
  return ufr::launched("chunks_to_nchw_kernel");
}

static int cat_segs(CatSegs& cs, const float* const* ptrs, const int* channels, const int* dst_channel0, int nseg, int chunks) {
  UFR_REQUIRE(ptrs && channels && dst_channel0 && nseg >= 1 && nseg <= 4 && chunks > 0, "nchw cat: bad segment list");
  cs.n = nseg;
  for (int s = 0; s < 4; ++s) { cs.src[s] = nullptr; cs.C[s] = 0; cs.dst0[s] = 0; }
  for (int s = 0; s < nseg; ++s) {
    UFR_REQUIRE(ptrs[s] && channels[s] > 0 && dst_channel0[s] >= 0 && dst_channel0[s] + channels[s] <= chunks * 32 &&
                    (s == 0 || dst_channel0[s] >= dst_channel0[s - 1] + channels[s - 1]),
                "nchw cat: segment %d out of order or outside the %d chunks", s, chunks);
    cs.src[s] = ptrs[s]; cs.C[s] = channels[s]; cs.dst0[s] = dst_channel0[s];
  }
  return UFR_OK;
}

extern "C" int ufr_nchw_cat_to_planes(const float* const* srcs, const int* channels, const int* dst_channel0, int nseg, void* planes,
                                      long plane_stride, int chunk0, int chunks, int B, int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(planes && plane_stride > 0 && chunk0 >= 0 && B > 0 && B < 65536 && H > 0 && W > 0, "nchw cat -> planes: bad argument");
  CatSegs cs;
  if (int rc = cat_segs(cs, srcs, channels, dst_channel0, nseg, chunks)) return rc;
  const dim3 grid(chunks, (H * W + 63) / 64, B);
  nchw_cat_to_planes_kernel<<<grid, 256, 0, ufr::as_stream(stream)>>>(cs, static_cast<__bf16*>(planes), plane_stride, chunk0, B, H * W);
  return ufr::launched("nchw_cat_to_planes_kernel");
}

extern "C" int ufr_chunks_to_nchw_cat(const float* g, int chunk0, int chunks, float* const* dsts, const int* channels,
                                      const int* src_channel0, int nseg, const float* act0, float pos0, float neg0, int B, int H, int W,
                                      ufr_stream_t stream) {
  UFR_REQUIRE(g && chunk0 >= 0 && B > 0 && B < 65536 && H > 0 && W > 0, "chunks -> nchw cat: bad argument");
  CatSegs cs;
  if (int rc = cat_segs(cs, const_cast<const float* const*>(dsts), channels, src_channel0, nseg, chunks)) return rc;
  const dim3 grid(chunks, (H * W + 63) / 64, B);
  chunks_to_nchw_cat_kernel<<<grid, 256, 0, ufr::as_stream(stream)>>>(g, chunk0, cs, act0, pos0, neg0, B, H * W);
  return ufr::launched("chunks_to_nchw_cat_kernel");
}

extern "C" int ufr_grad_finalize(const float* g, int g_chunk0, const void* mask, int mask_chunk0, void* out,
                                 long out_plane_stride, int out_chunk0, long M, int chunks, float slope, ufr_stream_t stream) {
  UFR_REQUIRE(g && out, "grad finalize: null pointer");
  UFR_REQUIRE(M > 0 && chunks > 0 && g_chunk0 >= 0 && mask_chunk0 >= 0 && out_chunk0 >= 0, "grad finalize: bad shape");
  grad_finalize_kernel<<<ufr::stream_grid((long)chunks * M * 4, 256), 256, 0, ufr::as_stream(stream)>>>(
      g, g_chunk0, static_cast<const __bf16*>(mask), mask_chunk0, static_cast<__bf16*>(out), out_plane_stride, out_chunk0, M,
      chunks, slope);
  return ufr::launched("grad_finalize_kernel");
}
