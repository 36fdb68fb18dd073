// small_cout.hip -- the 2-channel layers of the FlowNet family on gfx950.
//
// predict_flow* = Conv2d(Cin, 2, 3, 1, 1) with Cin up to 1026 and upsampled_flow* = ConvTranspose2d(2, 2, 4, 2, 1)
// (models/FlowNetC.py:43-50, models/submodules.py:85-90).  MIOpen runs them through implicit-GEMM tiles built
// for >= 64 output channels: 1.3 ms per attack iteration at [8, *, 96..6, 320..20] for work that is a single
// pass over the activations (352 MB) -- HBM-bound, 18 FMAs per element.  Here:
//   conv3x3_c2_fwd      thread = output pixel, loop over the input channels of its split: 9 row-contiguous
//                       loads (served by L1 / L2 between neighbouring rows) feed 18 FMAs against wave-uniform
//                       weights; small images split the channels over workgroups and a second kernel adds the
//                       partial sums in a fixed order (deterministic; no atomics)
//   conv3x3_c2_bwd_data thread = pixel: its 3x3x2 neighbourhood of the 2-channel gradient lives in registers,
//                       every input channel costs 18 FMAs and one coalesced store
//   deconv4x4s2_c2_*    8 / 32 FMAs per element, one thread per output / input pixel
#include "ufr_common.h"

namespace {

// partial[s][b][o][p] (or y itself when splits == 1, then with bias) over channels [s*cps, min(Cin,(s+1)*cps))
__global__ __launch_bounds__(256) void conv3x3_c2_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out, int Cin,
                                                      int H, int W, int cps, int direct) {
  const int HW = H * W;
  const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y, s = blockIdx.z;
  if (p >= HW) return;
  const int y = p / W, xx = p - y * W;
  const int c_lo = s * cps, c_hi = min(Cin, c_lo + cps);
  // clamped neighbour offsets + validity masks (zero padding)
  int off[9];
  float m[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int yy = y + k / 3 - 1, xk = xx + k % 3 - 1;
    const bool ok = yy >= 0 && yy < H && xk >= 0 && xk < W;
    off[k] = ok ? yy * W + xk : p;
    m[k] = ok ? 1.f : 0.f;
  }
  float a0 = 0.f, a1 = 0.f;
  const float* xb = x + ((size_t)b * Cin + c_lo) * HW;
  const float* w0 = w + (size_t)c_lo * 9;
  const float* w1 = w + ((size_t)Cin + c_lo) * 9;
  for (int c = c_lo; c < c_hi; ++c, xb += HW, w0 += 9, w1 += 9) {
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float v = xb[off[k]] * m[k];
      a0 = fmaf(v, w0[k], a0);
      a1 = fmaf(v, w1[k], a1);
    }
  }
  if (direct) {
    out[((size_t)b * 2 + 0) * HW + p] = a0 + bias[0];
    out[((size_t)b * 2 + 1) * HW + p] = a1 + bias[1];
  } else {
    const size_t B2 = (size_t)gridDim.y * 2;
    out[((size_t)s * B2 + (size_t)b * 2 + 0) * HW + p] = a0;
    out[((size_t)s * B2 + (size_t)b * 2 + 1) * HW + p] = a1;
  }
}

// y[b,o,p] = bias[o] + sum_s partial[s][b][o][p], in split order
__global__ void conv3x3_c2_reduce(const float* __restrict__ partial, const float* __restrict__ bias, float* __restrict__ y,
                                  long per_split, int HW, int splits) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_split; i += (long)gridDim.x * blockDim.x) {
    float a = 0.f;
    for (int s = 0; s < splits; ++s) a += partial[(size_t)s * per_split + i];
    y[i] = a + bias[(i / HW) & 1];
  }
}

// gx[b,c,p] = sum_o sum_k gy[b,o,p - (k - centre)] * w[o,c,k]
__global__ __launch_bounds__(256) void conv3x3_c2_bwd_data(const float* __restrict__ gy, const float* __restrict__ w,
                                                           float* __restrict__ gx, int Cin, int H, int W, int cps) {
  const int HW = H * W;
  const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y, s = blockIdx.z;
  if (p >= HW) return;
  const int y = p / W, xx = p - y * W;
  float g0[9], g1[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {                      // output pixel that used tap k on this input pixel
    const int yy = y - (k / 3 - 1), xk = xx - (k % 3 - 1);
    const bool ok = yy >= 0 && yy < H && xk >= 0 && xk < W;
    g0[k] = ok ? gy[((size_t)b * 2 + 0) * HW + yy * W + xk] : 0.f;
    g1[k] = ok ? gy[((size_t)b * 2 + 1) * HW + yy * W + xk] : 0.f;
  }
  const int c_lo = s * cps, c_hi = min(Cin, c_lo + cps);
  float* o = gx + ((size_t)b * Cin + c_lo) * HW + p;
  const float* w0 = w + (size_t)c_lo * 9;
  const float* w1 = w + ((size_t)Cin + c_lo) * 9;
  for (int c = c_lo; c < c_hi; ++c, o += HW, w0 += 9, w1 += 9) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) a = fmaf(g0[k], w0[k], a);
#pragma unroll
    for (int k = 0; k < 9; ++k) a = fmaf(g1[k], w1[k], a);
    *o = a;
  }
}

// ConvTranspose2d(2, 2, 4, 2, 1): y[b,o,Y,X] = bias[o] + sum_i sum_{ky,kx} x[b,i,(Y+1-ky)/2,(X+1-kx)/2] * w[i,o,ky,kx]
__global__ void deconv4x4s2_c2_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                   float* __restrict__ y, int B, int H, int W, int has_bias) {
  const int Ho = 2 * H, Wo = 2 * W;
  const long total = (long)B * Ho * Wo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int X = (int)(i % Wo), Y = (int)((i / Wo) % Ho), b = (int)(i / ((long)Wo * Ho));
    float a0 = has_bias ? bias[0] : 0.f, a1 = has_bias ? bias[1] : 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ky = ((Y + 1) & 1) + 2 * t, yy = (Y + 1 - ky) / 2;
      if (Y + 1 - ky < 0 || yy >= H) continue;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int kx = ((X + 1) & 1) + 2 * u, xx = (X + 1 - kx) / 2;
        if (X + 1 - kx < 0 || xx >= W) continue;
#pragma unroll
        for (int ic = 0; ic < 2; ++ic) {
          const float v = x[((size_t)b * 2 + ic) * H * W + (size_t)yy * W + xx];
          a0 = fmaf(v, w[((ic * 2 + 0) * 4 + ky) * 4 + kx], a0);
          a1 = fmaf(v, w[((ic * 2 + 1) * 4 + ky) * 4 + kx], a1);
        }
      }
    }
    y[((size_t)b * 2 + 0) * Ho * Wo + (size_t)Y * Wo + X] = a0;
    y[((size_t)b * 2 + 1) * Ho * Wo + (size_t)Y * Wo + X] = a1;
  }
}

// gx[b,i,y,x] = sum_o sum_{ky,kx} gy[b,o,2y-1+ky,2x-1+kx] * w[i,o,ky,kx]
__global__ void deconv4x4s2_c2_bwd_data(const float* __restrict__ gy, const float* __restrict__ w, float* __restrict__ gx,
                                        int B, int H, int W) {
  const int Ho = 2 * H, Wo = 2 * W;
  const long total = (long)B * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W), yy = (int)((i / W) % H), b = (int)(i / ((long)W * H));
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int Y = 2 * yy - 1 + ky;
      if (Y < 0 || Y >= Ho) continue;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int X = 2 * xx - 1 + kx;
        if (X < 0 || X >= Wo) continue;
#pragma unroll
        for (int oc = 0; oc < 2; ++oc) {
          const float g = gy[((size_t)b * 2 + oc) * Ho * Wo + (size_t)Y * Wo + X];
          a0 = fmaf(g, w[((0 * 2 + oc) * 4 + ky) * 4 + kx], a0);
          a1 = fmaf(g, w[((1 * 2 + oc) * 4 + ky) * 4 + kx], a1);
        }
      }
    }
    gx[((size_t)b * 2 + 0) * H * W + (size_t)yy * W + xx] = a0;
    gx[((size_t)b * 2 + 1) * H * W + (size_t)yy * W + xx] = a1;
  }
}

// enough workgroups to fill the chip: split the channels when the image is small
int pick_splits(int B, int HW, int Cin) {
  const long spatial = (long)B * ufr::ceil_div(HW, 256);
  long s = (4L * ufr::kNumCU + spatial - 1) / spatial;
  const long cap = Cin / 16 > 0 ? Cin / 16 : 1;
  if (s > cap) s = cap;
  if (s > 64) s = 64;
  return s < 1 ? 1 : (int)s;
}

}  // namespace

extern "C" long ufr_conv3x3_c2_workspace_floats(int B, int Cin, int H, int W) {
  const int splits = pick_splits(B, H * W, Cin);
  return splits == 1 ? 0 : (long)splits * B * 2 * H * W;
}

extern "C" int ufr_conv3x3_c2_forward(const float* x, const float* w, const float* bias, float* y, float* workspace,
                                      int B, int Cin, int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(x && w && bias && y, "conv3x3 c2 forward: null pointer");
  UFR_REQUIRE(B > 0 && B <= 65535 && Cin > 0 && H > 0 && W > 0, "conv3x3 c2 forward: bad shape");
  const int HW = H * W, splits = pick_splits(B, HW, Cin), cps = ufr::ceil_div(Cin, splits);
  UFR_REQUIRE(splits == 1 || workspace, "conv3x3 c2 forward: %d channel splits need the workspace", splits);
  hipStream_t st = ufr::as_stream(stream);
  conv3x3_c2_fwd<<<dim3(ufr::ceil_div(HW, 256), B, splits), 256, 0, st>>>(x, w, bias, splits == 1 ? y : workspace, Cin, H,
                                                                          W, cps, splits == 1);
  if (splits > 1) {
    const long per = (long)B * 2 * HW;
    conv3x3_c2_reduce<<<ufr::stream_grid(per, 256), 256, 0, st>>>(workspace, bias, y, per, HW, splits);
  }
  return ufr::launched("conv3x3_c2_fwd");
}

extern "C" int ufr_conv3x3_c2_backward_data(const float* grad_y, const float* w, float* grad_x, int B, int Cin, int H,
                                            int W, ufr_stream_t stream) {
  UFR_REQUIRE(grad_y && w && grad_x, "conv3x3 c2 backward: null pointer");
  UFR_REQUIRE(B > 0 && B <= 65535 && Cin > 0 && H > 0 && W > 0, "conv3x3 c2 backward: bad shape");
  const int HW = H * W, splits = pick_splits(B, HW, Cin), cps = ufr::ceil_div(Cin, splits);
  conv3x3_c2_bwd_data<<<dim3(ufr::ceil_div(HW, 256), B, splits), 256, 0, ufr::as_stream(stream)>>>(grad_y, w, grad_x, Cin,
                                                                                                   H, W, cps);
  return ufr::launched("conv3x3_c2_bwd_data");
}

extern "C" int ufr_deconv4x4s2_c2_forward(const float* x, const float* w, const float* bias, float* y, int B, int H, int W,
                                          ufr_stream_t stream) {
  UFR_REQUIRE(x && w && y, "deconv4x4s2 c2 forward: null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0, "deconv4x4s2 c2 forward: bad shape");
  const long total = (long)B * 4 * H * W;
  deconv4x4s2_c2_fwd<<<ufr::stream_grid(total, 256), 256, 0, ufr::as_stream(stream)>>>(x, w, bias, y, B, H, W, bias != nullptr);
  return ufr::launched("deconv4x4s2_c2_fwd");
}

extern "C" int ufr_deconv4x4s2_c2_backward_data(const float* grad_y, const float* w, float* grad_x, int B, int H, int W,
                                                ufr_stream_t stream) {
  UFR_REQUIRE(grad_y && w && grad_x, "deconv4x4s2 c2 backward: null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0, "deconv4x4s2 c2 backward: bad shape");
  const long total = (long)B * H * W;
  deconv4x4s2_c2_bwd_data<<<ufr::stream_grid(total, 256), 256, 0, ufr::as_stream(stream)>>>(grad_y, w, grad_x, B, H, W);
  return ufr::launched("deconv4x4s2_c2_bwd_data");
}
