// imresize.hip -- the loaders' image resize on the device, bit-exact with Pillow (gfx950).
//
// dataset_utils/data_utils.py:26-32 (`imresize`) is `PIL.Image.resize(..., resample=BILINEAR)` on uint8
// images; it runs for every frame of every sample (custom_transforms.py:73-122: Scale to 384x1280 for
// validation, RandomScaleCrop for training) and costs more host time per validation pair than the two
// network forwards cost on the GPU.  Pillow's resampler is integer arithmetic once its coefficient tables
// exist: two separable passes (horizontal, then vertical, each rounding to uint8), coefficients in 22-bit
// fixed point, accumulator started at 1 << 21, result clip8(acc >> 22).  The tables are computed on the host
// in float64 exactly like Pillow's precompute_coeffs (input_pipeline.py); these kernels are the integer part.
// HBM-streaming byte work: one thread per output pixel (all channels), coalesced along the row.
#include <cstdint>

#include "ufr_common.h"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int acc) {
  const int v = acc >> kPrecisionBits;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// dst[y, xx, c] = clip8(half + sum_x src[y, flip(xmin + x), c] * kk[xx, x])
template <int C>
__global__ void resample_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int Ws, int Wd,
                                  int flip, const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
  const long total = (long)H * Wd;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % Wd), y = (int)(i / Wd);
    const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
    const int* k = kk + (long)xx * ksize;
    int acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 1 << (kPrecisionBits - 1);
    const uint8_t* row = src + (long)y * Ws * C;
    for (int x = 0; x < cnt; ++x) {
      const int sx = flip ? Ws - 1 - (xmin + x) : xmin + x;
      const int w = k[x];
#pragma unroll
      for (int c = 0; c < C; ++c) acc[c] += (int)row[sx * C + c] * w;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) dst[i * C + c] = clip8(acc[c]);
  }
}

// dst[yy, x, c] = clip8(half + sum_y src[ymin + y, x, c] * kk[yy, y])
template <int C>
__global__ void resample_v_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int Hd, int W,
                                  const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
  const long total = (long)Hd * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), yy = (int)(i / W);
    const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
    const int* k = kk + (long)yy * ksize;
    int acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 1 << (kPrecisionBits - 1);
    for (int y = 0; y < cnt; ++y) {
      const uint8_t* p = src + ((long)(ymin + y) * W + x) * C;
      const int w = k[y];
#pragma unroll
      for (int c = 0; c < C; ++c) acc[c] += (int)p[c] * w;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) dst[i * C + c] = clip8(acc[c]);
  }
}

// ArrayToTensor (custom_transforms.py:47-57) fused with the crop of RandomScaleCrop / RandomCrop:
// dst[c, y, x] = float(src[cy + y, cx + x, c]) / divisor     (a true division, like `.float() / 255`)
__global__ void u8_to_tensor_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int W, int C, int cy,
                                    int cx, int ch, int cw, float divisor) {
  const long total = (long)C * ch * cw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % cw), y = (int)((i / cw) % ch), c = (int)(i / ((long)cw * ch));
    dst[i] = (float)src[((long)(cy + y) * W + cx + x) * C + c] / divisor;
  }
}

// KITTI flow PNG payload (flowutils/flow_io.py:104-127): 16-bit (u, v, valid) -> float32 [3,H,W]
__global__ void kitti_flow_decode_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst, long HW) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
    // (x - 2^15) / 64 is exact in float32 for 16-bit x
    dst[i] = ((float)src[3 * i + 0] - 32768.0f) / 64.0f;
    dst[HW + i] = ((float)src[3 * i + 1] - 32768.0f) / 64.0f;
    dst[2 * HW + i] = (float)src[3 * i + 2];
  }
}

int check_tables(const char* what, const int* bounds, const int* kk, int ksize) {
  UFR_REQUIRE(bounds && kk && ksize > 0, "%s: missing coefficient tables", what);
  return UFR_OK;
}

}  // namespace

extern "C" int ufr_resample_u8_horizontal(const uint8_t* src, uint8_t* dst, int H, int Ws, int Wd, int C, int flip,
                                          const int* bounds, const int* kk, int ksize, ufr_stream_t stream) {
  UFR_REQUIRE(src && dst, "resample horizontal: null pointer");
  UFR_REQUIRE(H > 0 && Ws > 0 && Wd > 0 && (C == 1 || C == 3 || C == 4), "resample horizontal: bad shape (C=%d)", C);
  if (int rc = check_tables("resample horizontal", bounds, kk, ksize)) return rc;
  const long total = (long)H * Wd;
  const dim3 grid(ufr::stream_grid(total, 256)), block(256);
  hipStream_t st = ufr::as_stream(stream);
  if (C == 1) resample_h_kernel<1><<<grid, block, 0, st>>>(src, dst, H, Ws, Wd, flip, bounds, kk, ksize);
  else if (C == 3) resample_h_kernel<3><<<grid, block, 0, st>>>(src, dst, H, Ws, Wd, flip, bounds, kk, ksize);
  else resample_h_kernel<4><<<grid, block, 0, st>>>(src, dst, H, Ws, Wd, flip, bounds, kk, ksize);
  return ufr::launched("resample_h");
}

extern "C" int ufr_resample_u8_vertical(const uint8_t* src, uint8_t* dst, int Hs, int Hd, int W, int C,
                                        const int* bounds, const int* kk, int ksize, ufr_stream_t stream) {
  UFR_REQUIRE(src && dst, "resample vertical: null pointer");
  UFR_REQUIRE(Hs > 0 && Hd > 0 && W > 0 && (C == 1 || C == 3 || C == 4), "resample vertical: bad shape (C=%d)", C);
  if (int rc = check_tables("resample vertical", bounds, kk, ksize)) return rc;
  const long total = (long)Hd * W;
  const dim3 grid(ufr::stream_grid(total, 256)), block(256);
  hipStream_t st = ufr::as_stream(stream);
  if (C == 1) resample_v_kernel<1><<<grid, block, 0, st>>>(src, dst, Hd, W, bounds, kk, ksize);
  else if (C == 3) resample_v_kernel<3><<<grid, block, 0, st>>>(src, dst, Hd, W, bounds, kk, ksize);
  else resample_v_kernel<4><<<grid, block, 0, st>>>(src, dst, Hd, W, bounds, kk, ksize);
  return ufr::launched("resample_v");
}

extern "C" int ufr_u8_to_tensor(const uint8_t* src, float* dst, int H, int W, int C, int crop_y, int crop_x, int crop_h,
                                int crop_w, float divisor, ufr_stream_t stream) {
  UFR_REQUIRE(src && dst, "u8 to tensor: null pointer");
  UFR_REQUIRE(H > 0 && W > 0 && C > 0 && crop_h > 0 && crop_w > 0, "u8 to tensor: bad shape");
  UFR_REQUIRE(crop_y >= 0 && crop_x >= 0 && crop_y + crop_h <= H && crop_x + crop_w <= W,
              "u8 to tensor: crop %dx%d at (%d,%d) leaves the %dx%d image", crop_h, crop_w, crop_y, crop_x, H, W);
  UFR_REQUIRE(divisor != 0.f, "u8 to tensor: zero divisor");
  const long total = (long)C * crop_h * crop_w;
  u8_to_tensor_kernel<<<ufr::stream_grid(total, 256), 256, 0, ufr::as_stream(stream)>>>(src, dst, W, C, crop_y, crop_x,
                                                                                         crop_h, crop_w, divisor);
  return ufr::launched("u8_to_tensor");
}

extern "C" int ufr_kitti_flow_decode(const uint16_t* src, float* dst, int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(src && dst, "kitti flow decode: null pointer");
  UFR_REQUIRE(H > 0 && W > 0, "kitti flow decode: bad shape");
  const long hw = (long)H * W;
  kitti_flow_decode_kernel<<<ufr::stream_grid(hw, 256), 256, 0, ufr::as_stream(stream)>>>(src, dst, hw);
  return ufr::launched("kitti_flow_decode");
}
