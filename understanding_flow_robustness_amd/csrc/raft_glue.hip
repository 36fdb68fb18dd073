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

// Workgroup = an 8 x 32 block of level-0 pixels x 32 channels, staged through LDS: the NCHW side moves as rows of 32 pixels (128 bytes),
// the NHWC side as 32 channels of one pixel (128 bytes); the pooled levels (4 x 16, 2 x 8, 1 x 4 pixels of the block) are made in LDS.
// (The first form -- thread = channel, 32-byte rows of NCHW -- took 25 / 40 us per 7.9 MB map.)
constexpr int PY_BH = 8, PY_BW = 32, PY_C = 32;
constexpr int PY_S0 = PY_BH * (PY_BW + 1) + 1;      // channel stride of the level-0 tile (odd: the 32 channels of a pixel hit 32 banks)
constexpr int PY_S1 = 4 * 17 + 1, PY_S2 = 2 * 9 + 1, PY_S3 = 5;

__global__ __launch_bounds__(256) void raft_fmap_pyramid_fwd_kernel(const float* __restrict__ f, PyrLevels out, int B, int C, int H, int W,
                                                                    int levels) {
  __shared__ float t0[PY_C * PY_S0], t1[PY_C * PY_S1], t2[PY_C * PY_S2];
  const int bx = (W + PY_BW - 1) / PY_BW, by = (H + PY_BH - 1) / PY_BH, bc = (C + PY_C - 1) / PY_C;
  int r = blockIdx.x;
  const int cb = r % bc; r /= bc;
  const int x0 = (r % bx) * PY_BW; r /= bx;
  const int y0 = (r % by) * PY_BH, b = r / by, c0 = cb * PY_C;
  const int tid = threadIdx.x;
  {                                                  // NCHW rows -> t0[c][y][x]
    const int x = tid & 31, y = tid >> 5;
    const bool in = y0 + y < H && x0 + x < W;
    for (int c = 0; c < PY_C; ++c)
      t0[c * PY_S0 + y * (PY_BW + 1) + x] = (in && c0 + c < C) ? f[((size_t)b * C + c0 + c) * H * W + (size_t)(y0 + y) * W + x0 + x] : 0.f;
  }
  __syncthreads();
  const int c = tid & 31, q = tid >> 5;             // NHWC side: 32 channels of a pixel side by side, 8 pixels per pass
  const bool cin = c0 + c < C;
  for (int p = q; p < PY_BH * PY_BW; p += 8) {
    const int y = p >> 5, x = p & 31;
    if (cin && y0 + y < H && x0 + x < W) out.p[0][(((size_t)b * H + y0 + y) * W + x0 + x) * C + c0 + c] = t0[c * PY_S0 + y * (PY_BW + 1) + x];
  }
  if (levels < 2) return;
  const int H1 = H >> 1, W1 = W >> 1, H2 = H >> 2, W2 = W >> 2, H3 = H >> 3, W3 = W >> 3;
  for (int p = q; p < 4 * 16; p += 8) {
    const int y = p >> 4, x = p & 15;
    const float* s = t0 + c * PY_S0 + (2 * y) * (PY_BW + 1) + 2 * x;
    const float v = (((s[0] + s[1]) + s[PY_BW + 1]) + s[PY_BW + 2]) * 0.25f;
    t1[c * PY_S1 + y * 17 + x] = v;
    const int yy = (y0 >> 1) + y, xx = (x0 >> 1) + x;
    if (cin && yy < H1 && xx < W1) out.p[1][(((size_t)b * H1 + yy) * W1 + xx) * C + c0 + c] = v;
  }
  if (levels < 3) return;
  __syncthreads();
  for (int p = q; p < 2 * 8; p += 8) {
    const int y = p >> 3, x = p & 7;
    const float* s = t1 + c * PY_S1 + (2 * y) * 17 + 2 * x;
    const float v = (((s[0] + s[1]) + s[17]) + s[18]) * 0.25f;
    t2[c * PY_S2 + y * 9 + x] = v;
    const int yy = (y0 >> 2) + y, xx = (x0 >> 2) + x;
    if (cin && yy < H2 && xx < W2) out.p[2][(((size_t)b * H2 + yy) * W2 + xx) * C + c0 + c] = v;
  }
  if (levels < 4) return;
  __syncthreads();
  if (q < 4) {
    const float* s = t2 + c * PY_S2 + 2 * q;
    const float v = (((s[0] + s[1]) + s[9]) + s[10]) * 0.25f;
    const int yy = y0 >> 3, xx = (x0 >> 3) + q;
    if (cin && yy < H3 && xx < W3) out.p[3][(((size_t)b * H3 + yy) * W3 + xx) * C + c0 + c] = v;
  }
}

// d f[NCHW] = g0 + (g1 + (g2 + g3 / 4) / 4) / 4, every level's total rounded as autograd's accumulation rounds it
__global__ __launch_bounds__(256) void raft_fmap_pyramid_bwd_kernel(PyrLevels g, float* __restrict__ gf, int B, int C, int H, int W,
                                                                    int levels) {
  __shared__ float t0[PY_C * PY_S0], t1[PY_C * PY_S1], t2[PY_C * PY_S2], t3[PY_C * PY_S3];
  const int bx = (W + PY_BW - 1) / PY_BW, by = (H + PY_BH - 1) / PY_BH, bc = (C + PY_C - 1) / PY_C;
  int r = blockIdx.x;
  const int cb = r % bc; r /= bc;
  const int x0 = (r % bx) * PY_BW; r /= bx;
  const int y0 = (r % by) * PY_BH, b = r / by, c0 = cb * PY_C;
  const int H1 = H >> 1, W1 = W >> 1, H2 = H >> 2, W2 = W >> 2, H3 = H >> 3, W3 = W >> 3;
  const int tid = threadIdx.x, c = tid & 31, q = tid >> 5;
  const bool cin = c0 + c < C;
  // every level's own gradient -> LDS (0 where a level has none or the pixel does not exist)
  for (int p = q; p < PY_BH * PY_BW; p += 8) {
    const int y = p >> 5, x = p & 31;
    t0[c * PY_S0 + y * (PY_BW + 1) + x] = (g.p[0] && cin && y0 + y < H && x0 + x < W) ? g.p[0][(((size_t)b * H + y0 + y) * W + x0 + x) * C + c0 + c] : 0.f;
  }
  for (int p = q; p < 4 * 16; p += 8) {
    const int y = p >> 4, x = p & 15, yy = (y0 >> 1) + y, xx = (x0 >> 1) + x;
    t1[c * PY_S1 + y * 17 + x] = (levels >= 2 && g.p[1] && cin && yy < H1 && xx < W1) ? g.p[1][(((size_t)b * H1 + yy) * W1 + xx) * C + c0 + c] : 0.f;
  }
  for (int p = q; p < 2 * 8; p += 8) {
    const int y = p >> 3, x = p & 7, yy = (y0 >> 2) + y, xx = (x0 >> 2) + x;
    t2[c * PY_S2 + y * 9 + x] = (levels >= 3 && g.p[2] && cin && yy < H2 && xx < W2) ? g.p[2][(((size_t)b * H2 + yy) * W2 + xx) * C + c0 + c] : 0.f;
  }
  if (q < 4) {
    const int yy = y0 >> 3, xx = (x0 >> 3) + q;
    t3[c * PY_S3 + q] = (levels >= 4 && g.p[3] && cin && yy < H3 && xx < W3) ? g.p[3][(((size_t)b * H3 + yy) * W3 + xx) * C + c0 + c] : 0.f;
  }
  __syncthreads();
  // totals, coarse to fine: own + parent's total / 4 (a pixel past the floor of the halved size has no parent: its slot holds 0)
  for (int p = q; p < 2 * 8; p += 8) {
    const int y = p >> 3, x = p & 7;
    t2[c * PY_S2 + y * 9 + x] += t3[c * PY_S3 + (x >> 1)] * 0.25f;
  }
  __syncthreads();
  for (int p = q; p < 4 * 16; p += 8) {
    const int y = p >> 4, x = p & 15;
    const bool parent = (y0 >> 2) + (y >> 1) < H2 && (x0 >> 2) + (x >> 1) < W2;
    t1[c * PY_S1 + y * 17 + x] += parent ? t2[c * PY_S2 + (y >> 1) * 9 + (x >> 1)] * 0.25f : 0.f;
  }
  __syncthreads();
  {                                                  // NCHW rows out
    const int x = tid & 31, y = tid >> 5;
    if (y0 + y < H && x0 + x < W) {
      const bool parent = (y0 >> 1) + (y >> 1) < H1 && (x0 >> 1) + (x >> 1) < W1;
      for (int cc = 0; cc < PY_C && c0 + cc < C; ++cc)
        gf[((size_t)b * C + c0 + cc) * H * W + (size_t)(y0 + y) * W + x0 + x] =
            t0[cc * PY_S0 + y * (PY_BW + 1) + x] + (parent ? t1[cc * PY_S1 + (y >> 1) * 17 + (x >> 1)] * 0.25f : 0.f);
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
  const long blocks = (long)B * ((H + PY_BH - 1) / PY_BH) * ((W + PY_BW - 1) / PY_BW) * ((C + PY_C - 1) / PY_C);
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
  const long blocks = (long)B * ((H + PY_BH - 1) / PY_BH) * ((W + PY_BW - 1) / PY_BW) * ((C + PY_C - 1) / PY_C);
  UFR_REQUIRE(blocks < (1L << 31), "raft fmap pyramid backward: too many pixels");
  raft_fmap_pyramid_bwd_kernel<<<(unsigned)blocks, 256, 0, ufr::as_stream(stream)>>>(g, grad_fmap, B, C, H, W, levels);
  return ufr::launched("raft_fmap_pyramid_bwd_kernel");
}
