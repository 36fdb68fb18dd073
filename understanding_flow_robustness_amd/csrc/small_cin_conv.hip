// small_cin_conv.hip -- PWC-Net's first pyramid convolution, conv1a = Conv2d(3, 16, 3, stride 2, padding 1) + LeakyReLU(0.1)
// (models/PWCNet.py:55-60 `conv(3, 16, kernel_size=3, stride=2)`, :235), straight from the raw NCHW frames to the activation planes.
//
// On the implicit GEMM this layer is 97 % padding: 3 of the 32 channels of its one K chunk and 16 of its 64 columns are real, and
// the frames go through a layout pass first -- 0.36 ms per 8 frames of 384x1280 (igemm_glds<128,64>, profiles/r4: `bench.py
// --config c4` runs it for both frame sets of every call).  It is 27 multiply-adds per output and HBM-bound: a thread computes
// one output pixel's N <= 32 channels from its 3 x 3 x 3 inputs with plain fp32 FMAs (weights broadcast from LDS), splits them
// into the three bf16 planes and writes its 2 x N bytes per plane; the chunk's padding channels stay at the zeros the planes
// were allocated with (the igemm's epilogue would write the same zeros).  Reads 12 B and writes 6 N B per input / output pixel.
#include "ufr_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ void split3(float v, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)v;
  const float r1 = v - (float)a;
  b = (__bf16)r1;
  c = (__bf16)(r1 - (float)b);
}

template <int NG_>                               // NG_ groups of 8 output channels
__global__ __launch_bounds__(256) void conv3x3s2_c3_planes_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                                  const float* __restrict__ bias, float slope,
                                                                  __bf16* __restrict__ out, long plane_stride, int out_chunk0, int n,
                                                                  int N, int H, int W) {
  __shared__ float wl[27][NG_ * 8];              // [c * 9 + ky * 3 + kx][output channel]
  __shared__ float bl[NG_ * 8];
  for (int i = threadIdx.x; i < 27 * NG_ * 8; i += 256) {
    const int k = i / (NG_ * 8), o = i - k * (NG_ * 8);
    wl[k][o] = o < N ? wgt[(long)o * 27 + k] : 0.f;
  }
  if (threadIdx.x < NG_ * 8) bl[threadIdx.x] = (int)threadIdx.x < N ? bias[threadIdx.x] : 0.f;
  __syncthreads();
  const int Ho = H / 2, Wo = W / 2;              // (H, W even: the pyramid's levels halve exactly)
  const long M = (long)n * Ho * Wo;
  for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
    const int xo = (int)(m % Wo), yo = (int)((m / Wo) % Ho), b = (int)(m / ((long)Wo * Ho));
    float in[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = 2 * yo - 1 + ky;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int xx = 2 * xo - 1 + kx;
          in[c * 9 + ky * 3 + kx] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? x[(((long)b * 3 + c) * H + yy) * W + xx] : 0.f;
        }
      }
#pragma unroll
    for (int g = 0; g < NG_; ++g) {
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
      for (int k = 0; k < 27; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(in[k], wl[k][g * 8 + j], acc[j]);
      bf16x8 q0, q1, q2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = acc[j] + bl[g * 8 + j];
        v = v > 0.f ? v : v * slope;
        if (g * 8 + j >= N) v = 0.f;
        __bf16 a, bb, c;
        split3(v, a, bb, c);
        q0[j] = a; q1[j] = bb; q2[j] = c;
      }
      __bf16* o = out + ((long)out_chunk0 * M + m) * 32 + g * 8;
      *reinterpret_cast<bf16x8*>(o) = q0;
      *reinterpret_cast<bf16x8*>(o + plane_stride) = q1;
      *reinterpret_cast<bf16x8*>(o + 2 * plane_stride) = q2;
    }
  }
}

// ---- conv1aa / conv1b: Conv2d(16, 16, 3, 1, 1) + LeakyReLU on activation planes (models/PWCNet.py:56-57) --------------------
// Half of the K chunk and three quarters of the 64 columns are padding on the implicit GEMM (0.35 ms per 8 frames of 192x640 for
// 4.5 GFLOP).  Here a workgroup stages a (8 + 2) x (32 + 2) pixel tile of the 16 input channels in LDS as float32 (the three
// planes added back, channel-major: lanes read consecutive pixels), a thread owns one output pixel's 16 channels and walks the
// 144 (channel, tap) pairs: one LDS read + 16 FMAs whose weights sit in SGPRs (a wave-uniform 64-byte scalar load per pair).
constexpr int SC_TH = 8, SC_TW = 32, SC_HW = (SC_TH + 2) * (SC_TW + 2);

__global__ __launch_bounds__(256) void conv3x3_c16_planes_kernel(const __bf16* __restrict__ x, long xs, int x_chunk0,
                                                                 const float* __restrict__ wt /* [16 c][9 taps][16 o] */,
                                                                 const float* __restrict__ bias, float slope, __bf16* __restrict__ out,
                                                                 long os, int out_chunk0, int n, int H, int W) {
  __shared__ float tile[16][SC_HW];
  const int tiles_x = (W + SC_TW - 1) / SC_TW, tiles_y = (H + SC_TH - 1) / SC_TH;
  const int b = blockIdx.x / (tiles_x * tiles_y), tr = blockIdx.x - b * tiles_x * tiles_y;
  const int y0 = (tr / tiles_x) * SC_TH, x0 = (tr % tiles_x) * SC_TW;
  const long M = (long)n * H * W;
  for (int p = threadIdx.x; p < SC_HW; p += 256) {
    const int py = p / (SC_TW + 2), px = p - py * (SC_TW + 2);
    const int yy = y0 - 1 + py, xx = x0 - 1 + px;
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = 0.f;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
      const __bf16* src = x + ((long)x_chunk0 * M + ((long)b * H + yy) * W + xx) * 32;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(src + h * 8), bb = *reinterpret_cast<const bf16x8*>(src + xs + h * 8),
                     c = *reinterpret_cast<const bf16x8*>(src + 2 * xs + h * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[h * 8 + j] = ((float)a[j] + (float)bb[j]) + (float)c[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) tile[j][p] = v[j];
  }
  __syncthreads();
  const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
  const int yo = y0 + ly, xo = x0 + lx;
  f32x2 acc2[8];                                                    // packed FMAs (v_pk_fma_f32): two output channels per instruction
#pragma unroll
  for (int o = 0; o < 8; ++o) acc2[o] = f32x2{0.f, 0.f};
  for (int c = 0; c < 16; ++c) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float v = tile[c][(ly + t / 3) * (SC_TW + 2) + lx + t % 3];
      const f32x2 vv = {v, v};
      const f32x2* w = reinterpret_cast<const f32x2*>(wt + (c * 9 + t) * 16);      // wave-uniform: scalar loads
#pragma unroll
      for (int o = 0; o < 8; ++o) acc2[o] = __builtin_elementwise_fma(vv, w[o], acc2[o]);
    }
  }
  float acc[16];
#pragma unroll
  for (int o = 0; o < 8; ++o) { acc[2 * o] = acc2[o][0]; acc[2 * o + 1] = acc2[o][1]; }
  if (yo < H && xo < W) {
    const long m = ((long)b * H + yo) * W + xo;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      bf16x8 q0, q1, q2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = acc[g * 8 + j] + bias[g * 8 + j];
        v = v > 0.f ? v : v * slope;
        __bf16 a, bb, c;
        split3(v, a, bb, c);
        q0[j] = a; q1[j] = bb; q2[j] = c;
      }
      __bf16* o = out + ((long)out_chunk0 * M + m) * 32 + g * 8;
      *reinterpret_cast<bf16x8*>(o) = q0;
      *reinterpret_cast<bf16x8*>(o + os) = q1;
      *reinterpret_cast<bf16x8*>(o + 2 * os) = q2;
    }
  }
}

}  // namespace

extern "C" int ufr_conv3x3s2_c3_planes(const float* frames, const float* weight, const float* bias, float slope, void* out_planes,
                                       long plane_stride, int out_chunk0, int n, int N, int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(frames && weight && bias && out_planes, "conv3x3 s2 (3 channels): null pointer argument");
  UFR_REQUIRE(n > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && N > 0 && N <= 32 && out_chunk0 >= 0 && plane_stride > 0 &&
                  (long)n * H * W < (1L << 31),
              "conv3x3 s2 (3 channels): bad shape");
  const long M = (long)n * (H / 2) * (W / 2);
  hipStream_t st = ufr::as_stream(stream);
  const int blocks = ufr::stream_grid(M, 256);
  __bf16* out = static_cast<__bf16*>(out_planes);
  const int groups = (N + 7) / 8;
  switch (groups) {
    case 1: conv3x3s2_c3_planes_kernel<1><<<blocks, 256, 0, st>>>(frames, weight, bias, slope, out, plane_stride, out_chunk0, n, N, H, W); break;
    case 2: conv3x3s2_c3_planes_kernel<2><<<blocks, 256, 0, st>>>(frames, weight, bias, slope, out, plane_stride, out_chunk0, n, N, H, W); break;
    case 3: conv3x3s2_c3_planes_kernel<3><<<blocks, 256, 0, st>>>(frames, weight, bias, slope, out, plane_stride, out_chunk0, n, N, H, W); break;
    default: conv3x3s2_c3_planes_kernel<4><<<blocks, 256, 0, st>>>(frames, weight, bias, slope, out, plane_stride, out_chunk0, n, N, H, W);
  }
  return ufr::launched("conv3x3s2_c3_planes_kernel");
}

extern "C" int ufr_conv3x3_c16_planes(const void* in_planes, long in_plane_stride, int in_chunk0, const float* weight_ct16,
                                      const float* bias, float slope, void* out_planes, long out_plane_stride, int out_chunk0, int n,
                                      int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(in_planes && weight_ct16 && bias && out_planes, "conv3x3 (16 -> 16 channels): null pointer argument");
  UFR_REQUIRE(n > 0 && H > 0 && W > 0 && in_chunk0 >= 0 && out_chunk0 >= 0 && in_plane_stride > 0 && out_plane_stride > 0 &&
                  (long)n * H * W < (1L << 31),
              "conv3x3 (16 -> 16 channels): bad shape");
  const long blocks = (long)n * ((H + SC_TH - 1) / SC_TH) * ((W + SC_TW - 1) / SC_TW);
  UFR_REQUIRE(blocks < (1L << 31), "conv3x3 (16 -> 16 channels): too many tiles");
  conv3x3_c16_planes_kernel<<<(unsigned)blocks, 256, 0, ufr::as_stream(stream)>>>(
      static_cast<const __bf16*>(in_planes), in_plane_stride, in_chunk0, weight_ct16, bias, slope, static_cast<__bf16*>(out_planes),
      out_plane_stride, out_chunk0, n, H, W);
  return ufr::launched("conv3x3_c16_planes_kernel");
}
