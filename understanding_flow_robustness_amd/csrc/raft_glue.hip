// raft_glue.hip -- the elementwise work in front of RAFT's encoders and of its on-the-fly correlation as two kernels each way instead of
// ~35 torch operators per forward / backward:
//   normalize pair   models/raft/raft.py:128-129   image = 2 * (image / 255.0) - 1.0   for both frames, written as the STACK [2B,3,H,W]
//                                                  the feature encoder takes (raft.py:141: fnet([image1, image2]) concatenates them)
//   fmap pyramid     models/raft/corr.py:97-105, :128-129   AlternateCorrBlock: fmap2 average-pooled three times (the reference pools
//                                                  BOTH maps FOUR times and uses fmap1 level 0 and fmap2 levels 0-3 only), each level
//                                                  permuted to NHWC for alt_cuda_corr
// Arithmetic = torch's, operation for operation (the tests compare with torch.equal):
//   x / 255.0 is x * (1.0f / 255.0f) (ATen divides a tensor by a scalar with the reciprocal), * 2 is exact, - 1 rounds once;
//   avg_pool2d(2, 2) adds the four cells row by row -- ((a + b) + c) + d -- and divides by 4 (exact), every level from the ROUNDED level
//   below it; the adjoint of a level is its own gradient plus a quarter of its parent's total (two terms: the order is immaterial).
#include "ufr_common.h"

namespace {

__global__ void raft_normalize_pair_kernel(const float* __restrict__ x1, const float* __restrict__ x2, float* __restrict__ y, long n4) {
  const float r = 1.0f / 255.0f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = i < n4 ? reinterpret_cast<const float4*>(x1)[i] : reinterpret_cast<const float4*>(x2)[i - n4];
    float4 o;
    o.x = __fsub_rn(__fmul_rn(v.x, r) * 2.0f, 1.0f); o.y = __fsub_rn(__fmul_rn(v.y, r) * 2.0f, 1.0f);
    o.z = __fsub_rn(__fmul_rn(v.z, r) * 2.0f, 1.0f); o.w = __fsub_rn(__fmul_rn(v.w, r) * 2.0f, 1.0f);
    reinterpret_cast<float4*>(y)[i] = o;
  }
}

// d x = (d y * 2) * (1 / 255): autograd's two nodes, in its order
__global__ void raft_normalize_pair_bwd_kernel(const float* __restrict__ gy, float* __restrict__ g1, float* __restrict__ g2, long n4) {
  const float r = 1.0f / 255.0f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(gy)[i];
    const float4 o = make_float4(__fmul_rn(v.x * 2.0f, r), __fmul_rn(v.y * 2.0f, r), __fmul_rn(v.z * 2.0f, r), __fmul_rn(v.w * 2.0f, r));
    if (i < n4) reinterpret_cast<float4*>(g1)[i] = o;
    else if (g2) reinterpret_cast<float4*>(g2)[i - n4] = o;
  }
}

struct PyrLevels { float* p[4]; };                 // NHWC levels 0..3 (forward: outputs; backward: the gradients, NULL = none)

// Workgroup = one 8 x 8 block of level-0 pixels, thread = channel (looped past 256): the 64 values of a channel's block are read
// as rows of 8 (NCHW), every level's pixels written [pixel][channel] with the channels of a pixel side by side (coalesced).
// A map of 256 x 48 x 160 is 7.9 MB: the launch replaces 8 pooling + 5 permute-copy launches of ~6 us each.
__global__ __launch_bounds__(256) void raft_fmap_pyramid_fwd_kernel(const float* __restrict__ f, PyrLevels out, int B, int C, int H, int W,
                                                                    int levels) {
  const int bx = (W + 7) >> 3, by = (H + 7) >> 3;
  const int b = blockIdx.x / (bx * by), r = blockIdx.x - b * bx * by, y0 = (r / bx) * 8, x0 = (r % bx) * 8;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float* src = f + ((size_t)b * C + c) * H * W;
    float v0[8][8];
#pragma unroll
    for (int y = 0; y < 8; ++y)
#pragma unroll
      for (int x = 0; x < 8; ++x) v0[y][x] = (y0 + y < H && x0 + x < W) ? src[(size_t)(y0 + y) * W + x0 + x] : 0.f;
#pragma unroll
    for (int y = 0; y < 8; ++y)
#pragma unroll
      for (int x = 0; x < 8; ++x)
        if (y0 + y < H && x0 + x < W) out.p[0][(((size_t)b * H + y0 + y) * W + x0 + x) * C + c] = v0[y][x];
    if (levels < 2) continue;
    float v1[4][4], v2[2][2];
    const int H1 = H >> 1, W1 = W >> 1, H2 = H >> 2, W2 = W >> 2, H3 = H >> 3, W3 = W >> 3;
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        v1[y][x] = (((v0[2 * y][2 * x] + v0[2 * y][2 * x + 1]) + v0[2 * y + 1][2 * x]) + v0[2 * y + 1][2 * x + 1]) * 0.25f;
        const int yy = (y0 >> 1) + y, xx = (x0 >> 1) + x;
        if (yy < H1 && xx < W1) out.p[1][(((size_t)b * H1 + yy) * W1 + xx) * C + c] = v1[y][x];
      }
    if (levels < 3) continue;
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        v2[y][x] = (((v1[2 * y][2 * x] + v1[2 * y][2 * x + 1]) + v1[2 * y + 1][2 * x]) + v1[2 * y + 1][2 * x + 1]) * 0.25f;
        const int yy = (y0 >> 2) + y, xx = (x0 >> 2) + x;
        if (yy < H2 && xx < W2) out.p[2][(((size_t)b * H2 + yy) * W2 + xx) * C + c] = v2[y][x];
      }
    if (levels < 4) continue;
    const float v3 = (((v2[0][0] + v2[0][1]) + v2[1][0]) + v2[1][1]) * 0.25f;
    if ((y0 >> 3) < H3 && (x0 >> 3) < W3) out.p[3][(((size_t)b * H3 + (y0 >> 3)) * W3 + (x0 >> 3)) * C + c] = v3;
  }
}

// d f[NCHW] = g0 + (g1 + (g2 + g3 / 4) / 4) / 4, every level's total rounded as autograd's accumulation rounds it
__global__ __launch_bounds__(256) void raft_fmap_pyramid_bwd_kernel(PyrLevels g, float* __restrict__ gf, int B, int C, int H, int W,
                                                                    int levels) {
  const int bx = (W + 7) >> 3, by = (H + 7) >> 3;
  const int b = blockIdx.x / (bx * by), r = blockIdx.x - b * bx * by, y0 = (r / bx) * 8, x0 = (r % bx) * 8;
  const int H1 = H >> 1, W1 = W >> 1, H2 = H >> 2, W2 = W >> 2, H3 = H >> 3, W3 = W >> 3;
  for (int c = threadIdx.x; c < C; c += 256) {
    float t3 = 0.f, t2[2][2], t1[4][4];
    if (levels >= 4 && g.p[3] && (y0 >> 3) < H3 && (x0 >> 3) < W3) t3 = g.p[3][(((size_t)b * H3 + (y0 >> 3)) * W3 + (x0 >> 3)) * C + c];
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int yy = (y0 >> 2) + y, xx = (x0 >> 2) + x;
        const float own = (levels >= 3 && g.p[2] && yy < H2 && xx < W2) ? g.p[2][(((size_t)b * H2 + yy) * W2 + xx) * C + c] : 0.f;
        t2[y][x] = own + t3 * 0.25f;
      }
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int yy = (y0 >> 1) + y, xx = (x0 >> 1) + x;
        const float own = (levels >= 2 && g.p[1] && yy < H1 && xx < W1) ? g.p[1][(((size_t)b * H1 + yy) * W1 + xx) * C + c] : 0.f;
        // (a level-1 pixel past floor(H / 4) * 2 has no parent: its parent's slot above was read as 0)
        t1[y][x] = own + (((y0 >> 2) + (y >> 1) < H2 && (x0 >> 2) + (x >> 1) < W2) ? t2[y >> 1][x >> 1] * 0.25f : 0.f);
      }
    float* dst = gf + ((size_t)b * C + c) * H * W;
#pragma unroll
    for (int y = 0; y < 8; ++y)
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        if (y0 + y >= H || x0 + x >= W) continue;
        const float own = g.p[0] ? g.p[0][(((size_t)b * H + y0 + y) * W + x0 + x) * C + c] : 0.f;
        const bool parent = (y0 >> 1) + (y >> 1) < H1 && (x0 >> 1) + (x >> 1) < W1;
        dst[(size_t)(y0 + y) * W + x0 + x] = own + (parent ? t1[y >> 1][x >> 1] * 0.25f : 0.f);
      }
  }
}

}  // namespace

extern "C" int ufr_raft_normalize_pair(const float* image1, const float* image2, float* stack, long n_each, ufr_stream_t stream) {
  UFR_REQUIRE(image1 && image2 && stack, "raft normalize pair: null pointer");
  UFR_REQUIRE(n_each > 0 && n_each % 4 == 0, "raft normalize pair: the frames' element count must be a positive multiple of 4");
  UFR_REQUIRE(((reinterpret_cast<size_t>(image1) | reinterpret_cast<size_t>(image2) | reinterpret_cast<size_t>(stack)) & 15) == 0,
              "raft normalize pair: 16-byte aligned tensors expected");
  raft_normalize_pair_kernel<<<ufr::stream_grid(n_each / 2, 256), 256, 0, ufr::as_stream(stream)>>>(image1, image2, stack, n_each / 4);
  return ufr::launched("raft_normalize_pair_kernel");
}

extern "C" int ufr_raft_normalize_pair_backward(const float* grad_stack, float* grad1, float* grad2, long n_each, ufr_stream_t stream) {
  UFR_REQUIRE(grad_stack && grad1, "raft normalize pair backward: null pointer");
  UFR_REQUIRE(n_each > 0 && n_each % 4 == 0, "raft normalize pair backward: the frames' element count must be a positive multiple of 4");
  UFR_REQUIRE(((reinterpret_cast<size_t>(grad_stack) | reinterpret_cast<size_t>(grad1) | reinterpret_cast<size_t>(grad2)) & 15) == 0,
              "raft normalize pair backward: 16-byte aligned tensors expected");
  raft_normalize_pair_bwd_kernel<<<ufr::stream_grid(n_each / 2, 256), 256, 0, ufr::as_stream(stream)>>>(grad_stack, grad1, grad2, n_each / 4);
  return ufr::launched("raft_normalize_pair_bwd_kernel");
}

extern "C" int ufr_raft_fmap_pyramid_forward(const float* fmap, float* const* levels_nhwc, int levels, int B, int C, int H, int W,
                                             ufr_stream_t stream) {
  UFR_REQUIRE(fmap && levels_nhwc, "raft fmap pyramid: null pointer");
  UFR_REQUIRE(levels >= 1 && levels <= 4 && B > 0 && C > 0 && H > 0 && W > 0, "raft fmap pyramid: bad shape (1 - 4 levels)");
  UFR_REQUIRE((H >> (levels - 1)) > 0 && (W >> (levels - 1)) > 0, "raft fmap pyramid: the map is too small for %d levels", levels);
  PyrLevels out{};
  for (int l = 0; l < levels; ++l) {
    UFR_REQUIRE(levels_nhwc[l], "raft fmap pyramid: level %d is null", l);
    out.p[l] = levels_nhwc[l];
  }
  const long blocks = (long)B * ((H + 7) / 8) * ((W + 7) / 8);
  UFR_REQUIRE(blocks < (1L << 31), "raft fmap pyramid: too many pixels");
  raft_fmap_pyramid_fwd_kernel<<<(unsigned)blocks, 256, 0, ufr::as_stream(stream)>>>(fmap, out, B, C, H, W, levels);
  return ufr::launched("raft_fmap_pyramid_fwd_kernel");
}

extern "C" int ufr_raft_fmap_pyramid_backward(const float* const* grad_levels_nhwc, int levels, float* grad_fmap, int B, int C, int H, int W,
                                              ufr_stream_t stream) {
  UFR_REQUIRE(grad_levels_nhwc && grad_fmap, "raft fmap pyramid backward: null pointer");
  UFR_REQUIRE(levels >= 1 && levels <= 4 && B > 0 && C > 0 && H > 0 && W > 0, "raft fmap pyramid backward: bad shape (1 - 4 levels)");
  PyrLevels g{};
  for (int l = 0; l < levels; ++l) g.p[l] = const_cast<float*>(grad_levels_nhwc[l]);      // NULL: that level has no gradient
  const long blocks = (long)B * ((H + 7) / 8) * ((W + 7) / 8);
  UFR_REQUIRE(blocks < (1L << 31), "raft fmap pyramid backward: too many pixels");
  raft_fmap_pyramid_bwd_kernel<<<(unsigned)blocks, 256, 0, ufr::as_stream(stream)>>>(g, grad_fmap, B, C, H, W, levels);
  return ufr::launched("raft_fmap_pyramid_bwd_kernel");
}
