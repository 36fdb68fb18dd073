// correlation_window_mfma.hip -- both adjoints of FlowNetC's cost volume (21 x 21 displacements, stride 2:
// correlation_cuda_kernel.cu:86-233) on the cells of the patch attack's prefix window, on the bf16 matrix cores, reading
// the head engine's buffers as they lie and writing the WINDOW-sized gradient the windowed prefix's backward consumes.
//
//   gin1[c, y, x] = sum_{i,j} g[i, j, y, x]           * f2[c, y + oy, x + ox]            (oy, ox) = 2 (i - 10, j - 10)
//   gin2[c, y, x] = sum_{i,j} g[i, j, y - oy, x - ox] * f1[c, y - oy, x - ox]
//
// For one window row y and one source row ys both are a BANDED GEMM over the source columns x':
//   out[x, c] += sum_{x'} A[x, x'] * S[x', c],   A[x, x'] = g[i, j, .] where x' = x +- 2 (j - 10), zero elsewhere
// M = the window's 16 cells of the row, N = channels, K = 64 source columns (window + 2 x 20 reach, aligned to 4).
// S is read from the NCHW float32 features (8 consecutive columns of one channel = a lane's B fragment = two 16-byte loads)
// and split into three bf16 planes in registers; A is gathered from the engine's chunk-major gradient sum of conv3_1's
// input (scale 1 / C folded in), split, and shared through LDS by the workgroup's waves.  float32 = six bf16 products.
// A workgroup owns NR same-parity window rows (they share source rows) x 4 waves x TPW channel tiles; per source row one
// barrier (A double-buffered).
//
// Fused around it (all of them were separate full-frame passes): the conversion of the gradient sum to NCHW, the two
// zero-fills of the full-size gradients, conv_redir's contribution to the first frame's gradient, the gather of the
// window and the zeroing of its inexact rim (window.hip: window_copy_kernel<true>).
#include "ufr_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ constexpr int WPA[6] = {2, 0, 1, 1, 0, 0};
__device__ constexpr int WPB[6] = {0, 2, 1, 0, 1, 0};

__device__ __forceinline__ void split3(float v, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)v;
  const float r1 = v - (float)a;
  b = (__bf16)r1;
  c = (__bf16)(r1 - (float)b);
}

constexpr int WP = 21, WR = 10, WC = 256;
constexpr int AST = 72;                           // A row stride in bf16: 64 source columns + 8 (bank spread), 16-byte rows

template <int NR, int TPW>
__global__ __launch_bounds__(256) void corr_bwd_window_mfma_kernel(
    const float* __restrict__ f1, const float* __restrict__ f2, const float* __restrict__ G, int g_chunk0, float g_scale,
    const float* __restrict__ Gredir, float* __restrict__ gwin, int B, int H, int W, const int* __restrict__ win,
    int level_stride, int wh, int ww, int margin) {
  constexpr int CS = 16 / (4 * TPW);              // channel splits over blockIdx.y
  __shared__ __attribute__((aligned(16))) __bf16 As[2][NR][3][16 * AST];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // The 64 workgroups of one frame pair (8 row groups x 2 adjoints x 4 channel quarters) read the same 56 x 64 feature region and the same
  // band of cost-volume gradients.  In launch order (x fastest) consecutive workgroups go round-robin to the 8 XCDs, so EVERY XCD's L2
  // fetched EVERY pair's region: 214 MB of HBM traffic for 72 MB of operands (profiles/r4_corr_window_traffic.json).  The launch index is
  // re-dealt pair-fastest instead: with 8 pairs, pair n lives on XCD n and its region crosses the fabric once.
  const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int n = lin % B, rest = lin / B, bx = rest % (int)gridDim.x, by = rest / (int)gridDim.x;
  const int adj = by / CS, cs = by - adj * CS;
  const int pr = bx & 1, ly0 = pr + 2 * (bx >> 1) * NR;                         // window rows ly0 + 2 r
  const int* w = win + n * 8;
  const int y0 = min(max(w[0] / level_stride, 0), H - wh), x0 = min(max(w[1] / level_stride, 0), W - ww);
  const int xb = ((x0 - 2 * WR) >> 2) << 2;       // first source column of the K range (floor to 4; may be negative)
  const long M = (long)B * H * W, plane = (long)H * W;
  const float* src = (adj == 0 ? f2 : f1) + (long)n * WC * plane;
  for (int i = tid; i < 2 * NR * 3 * 16 * AST / 8; i += 256) reinterpret_cast<bf16x8*>(&As[0][0][0][0])[i] = bf16x8{};
  f32x4 acc[NR][TPW];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const int ybase = y0 + ly0;
  const int ct0 = cs * 4 * TPW + wave * TPW;      // this wave's first channel tile
  for (int s = 0; s < WP + NR - 1; ++s) {
    const int sy = ybase - 2 * WR + 2 * s;        // source row
    if (sy < 0 || sy >= H) continue;              // uniform: nothing to add (zero features / no cost-volume cell)
    const int buf = s & 1;
    // ---- B fragments: 8 source columns of one channel, float32 -> three bf16 planes
    bf16x8 fb[TPW][2][3];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int c = (ct0 + t) * 16 + (lane & 15), col = xb + 32 * ks + 8 * (lane >> 4);
        const float* p = src + ((long)c * H + sy) * W + col;
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (col >= 0 && col < W) v0 = *reinterpret_cast<const float4*>(p);
        if (col + 4 >= 0 && col + 4 < W) v1 = *reinterpret_cast<const float4*>(p + 4);
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          __bf16 a, b2, c2;
          split3(v[e], a, b2, c2);
          fb[t][ks][0][e] = a; fb[t][ks][1][e] = b2; fb[t][ks][2][e] = c2;
        }
      }
    // ---- A = the band of cost-volume gradients of every live (row, displacement row) pair of this source row
    for (int item = tid; item < NR * 16 * WP; item += 256) {
      const int r = item / (16 * WP), rem = item - r * 16 * WP, j = rem >> 4, xl = rem & 15;
      const int i = adj == 0 ? s - r : 2 * WR - s + r;
      const int ly = ly0 + 2 * r;
      if (i < 0 || i >= WP || ly >= wh) continue;
      const int x = x0 + xl, y = y0 + ly;
      const int xs = adj == 0 ? x + 2 * (j - WR) : x - 2 * (j - WR);
      float v = 0.f;
      if (xl < ww && xs >= 0 && xs < W) {
        const int d = i * WP + j;
        const long pix = adj == 0 ? ((long)n * H + y) * W + x : ((long)n * H + sy) * W + xs;
        v = G[((long)(g_chunk0 + (d >> 5)) * M + pix) * 32 + (d & 31)] * g_scale;
      }
      __bf16 a, b2, c2;
      split3(v, a, b2, c2);
      const int o = xl * AST + (xs - xb);
      As[buf][r][0][o] = a; As[buf][r][1][o] = b2; As[buf][r][2][o] = c2;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int i = adj == 0 ? s - r : 2 * WR - s + r;
      if (i < 0 || i >= WP || ly0 + 2 * r >= wh) continue;                       // uniform
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 fa[3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
          fa[p] = *reinterpret_cast<const bf16x8*>(&As[buf][r][p][(lane & 15) * AST + 32 * ks + 8 * (lane >> 4)]);
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
          for (int q = 0; q < 6; ++q)
            acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[WPA[q]], fb[t][ks][WPB[q]], acc[r][t], 0, 0, 0);
      }
    }
  }
  // ---- epilogue: lane holds out[x = 4 (lane >> 4) + e][c = tile * 16 + (lane & 15)]
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int ly = ly0 + 2 * r;
    if (ly >= wh) continue;
    const int y = y0 + ly;
    const bool rim_y = (ly < margin && y0 > 0) || (ly >= wh - margin && y0 + wh < H);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int c = (ct0 + t) * 16 + (lane & 15);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int xl = 4 * (lane >> 4) + e;
        if (xl >= ww) continue;
        const bool rim = rim_y || (xl < margin && x0 > 0) || (xl >= ww - margin && x0 + ww < W);
        float v = acc[r][t][e];
        if (adj == 0 && Gredir) v += Gredir[((long)(c >> 5) * M + ((long)n * H + y) * W + x0 + xl) * 32 + (c & 31)];
        gwin[(((long)(adj * B + n) * WC + c) * wh + ly) * ww + xl] = rim ? 0.f : v;
      }
    }
  }
}

}  // namespace

extern "C" int ufr_corr_backward_window_fused(const float* f1, const float* f2, const float* G, int g_chunk0, float g_scale,
                                              const float* G_redir, float* grad_window, int B, int C, int H, int W, int patch,
                                              int dilation_patch, const int* win, int level_stride, int wh, int ww, int margin,
                                              ufr_stream_t stream) {
  UFR_REQUIRE(f1 && f2 && G && grad_window && win, "corr backward window (fused): null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && g_chunk0 >= 0 && level_stride > 0, "corr backward window (fused): bad shape");
  UFR_REQUIRE(wh > 0 && ww > 0 && wh <= H && ww <= W && margin >= 0 && 2 * margin <= wh && 2 * margin <= ww,
              "corr backward window (fused): window %dx%d (margin %d) in %dx%d", wh, ww, margin, H, W);
  if (C != WC || patch != WP || dilation_patch != 2 || ww > 16 || (W & 3))
    return ufr::fail(UFR_EUNSUPPORTED, "corr backward window (fused): built for 256 channels, patch 21, dilation_patch 2, windows "
                                       "of at most 16 cells across, W a multiple of 4; got C=%d patch=%d dilation=%d ww=%d W=%d",
                     C, patch, dilation_patch, ww, W);
  // (rows per workgroup, channel tiles per wave): 2 x 1 at 8 pairs (measured 0.067 ms; 2 x 2 0.077, 1 x 1 / 1 x 2 0.085),
  // 1 x 1 (4x the workgroups) for one or two pairs
  const int cfg = B <= 2 ? 11 : 21;
#define UFR_CBW_LAUNCH(NR, TPW)                                                                                             \
  corr_bwd_window_mfma_kernel<NR, TPW><<<dim3(2 * ufr::ceil_div(ufr::ceil_div(wh, 2), NR), 2 * (16 / (4 * TPW)), B), 256, 0, \
                                         ufr::as_stream(stream)>>>(f1, f2, G, g_chunk0, g_scale, G_redir, grad_window, B, H, W, \
                                                                   win, level_stride, wh, ww, margin)
  if (cfg == 11) UFR_CBW_LAUNCH(1, 1);
  else UFR_CBW_LAUNCH(2, 1);
#undef UFR_CBW_LAUNCH
  return ufr::launched("corr_bwd_window_mfma_kernel");
}
