// placement.hip -- the patch's trip between its own SxS frame and the image canvas, on the device
// (gfx950).  Replaces the host round trip of patch_attacks/utils_patch.py:257-358 (circle_transform:
// scipy.ndimage.zoom / rotate on numpy arrays, three canvas-sized np.zeros, three H2D copies per sample)
// and patch_attacks/main.py:408-461 (D2H, crop, scipy zoom back).  Everything here is tiny (a 51x51
// patch); the point is that train() never leaves the GPU, not the kernel time.
//
// scipy.ndimage semantics reproduced (probed against scipy 1.15, tests/test_placement_gpu.py):
//   * coordinates in float64: zoom  cc = k * ((n_in - 1) / (n_out - 1));  affine  cc = M (i, j) + offset
//   * mode='constant', cval=0: a coordinate below 0 or above n-1 on ANY axis gives 0 -- including the
//     last sample of a zoom whose k*zoom product rounds above n-1 (a scipy quirk the reference inherits)
//   * order 1: floor + linear weights;  order 0: floor(cc + 0.5)
#include "ufr_common.h"

namespace {

__global__ void affine_resample_f64_kernel(const double* __restrict__ src, double* __restrict__ dst, int C,
                                           int Hs, int Ws, int Hd, int Wd, double m00, double m01, double m10,
                                           double m11, double off0, double off1, int order) {
  const int total = C * Hd * Wd;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int x = i % Wd, y = (i / Wd) % Hd, c = i / (Wd * Hd);
    const double cy = (m00 * (double)y + m01 * (double)x) + off0;
    const double cx = (m10 * (double)y + m11 * (double)x) + off1;
    double v = 0.0;
    if (!(cy < 0.0 || cy > (double)(Hs - 1) || cx < 0.0 || cx > (double)(Ws - 1))) {
      const double* s = src + (size_t)c * Hs * Ws;
      if (order == 0) {
        const int yy = (int)floor(cy + 0.5), xx = (int)floor(cx + 0.5);
        v = (yy < Hs && xx < Ws) ? s[yy * Ws + xx] : 0.0;
      } else {
        const int y0 = (int)floor(cy), x0 = (int)floor(cx);
        const double ty = cy - (double)y0, tx = cx - (double)x0;
        // separable order-1 spline: weights (1-t, t); a neighbour beyond the edge has weight 0 here
        const int y1 = y0 + 1 < Hs ? y0 + 1 : y0, x1 = x0 + 1 < Ws ? x0 + 1 : x0;
        const double wy1 = y0 + 1 < Hs ? ty : 0.0, wx1 = x0 + 1 < Ws ? tx : 0.0;
        // accumulated like scipy's generic loop: sum over the 2x2 support of coefficient * (wy * wx)
        const double wy0 = 1.0 - ty, wx0 = 1.0 - tx;
        v = s[y0 * Ws + x0] * (wy0 * wx0);
        v += s[y0 * Ws + x1] * (wy0 * wx1);
        v += s[y1 * Ws + x0] * (wy1 * wx0);
        v += s[y1 * Ws + x1] * (wy1 * wx1);
      }
    }
    dst[i] = v;
  }
}

// canvases are [C,H,W] float32, already zeroed; paste the (h x w) double images at (y, x)
__global__ void place_kernel(const double* __restrict__ patch, const double* __restrict__ mask,
                             const double* __restrict__ init, float* __restrict__ cp, float* __restrict__ cm,
                             float* __restrict__ ci, int C, int h, int w, int H, int W, int y, int x) {
  const int total = C * h * w;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int xx = i % w, yy = (i / w) % h, c = i / (w * h);
    const size_t o = ((size_t)c * H + y + yy) * W + x + xx;
    cp[o] = (float)patch[i];
    cm[o] = (float)mask[i];
    ci[o] = (float)init[i];
  }
}

__global__ void crop_f64_kernel(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ dst,
                                int C, int H, int W, int y, int x, int h, int w) {
  const int total = C * h * w;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int xx = i % w, yy = (i / w) % h, c = i / (w * h);
    const size_t o = ((size_t)c * H + y + yy) * W + x + xx;
    const float v = b ? a[o] * b[o] : a[o];          // torch.mul in float32 (main.py:410), then .astype(float64)
    dst[i] = (double)v;
  }
}

}  // namespace

extern "C" int ufr_affine_resample_f64(const double* src, double* dst, int C, int Hs, int Ws, int Hd, int Wd,
                                       double m00, double m01, double m10, double m11, double off0, double off1,
                                       int order, ufr_stream_t stream) {
  UFR_REQUIRE(src && dst, "affine resample: null pointer");
  UFR_REQUIRE(C > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "affine resample: bad shape");
  UFR_REQUIRE(order == 0 || order == 1, "affine resample: order %d unsupported (0 or 1)", order);
  UFR_REQUIRE((long)C * Hd * Wd < 2147483647L && (long)C * Hs * Ws < 2147483647L, "affine resample: image too large");
  const long total = (long)C * Hd * Wd;
  affine_resample_f64_kernel<<<ufr::stream_grid(total, 256), 256, 0, ufr::as_stream(stream)>>>(
      src, dst, C, Hs, Ws, Hd, Wd, m00, m01, m10, m11, off0, off1, order);
  return ufr::launched("affine_resample_f64");
}

extern "C" int ufr_patch_place(const double* patch, const double* mask, const double* init, int C, int h, int w,
                               float* canvas_patch, float* canvas_mask, float* canvas_init, int H, int W, int y,
                               int x, ufr_stream_t stream) {
  UFR_REQUIRE(patch && mask && init && canvas_patch && canvas_mask && canvas_init, "patch place: null pointer");
  UFR_REQUIRE(C > 0 && h > 0 && w > 0 && H > 0 && W > 0, "patch place: bad shape");
  UFR_REQUIRE(y >= 0 && x >= 0 && y + h <= H && x + w <= W, "patch place: %dx%d at (%d,%d) leaves the %dx%d canvas",
              h, w, y, x, H, W);
  hipStream_t st = ufr::as_stream(stream);
  const size_t bytes = sizeof(float) * (size_t)C * H * W;
  hipError_t e = hipMemsetAsync(canvas_patch, 0, bytes, st);
  if (e == hipSuccess) e = hipMemsetAsync(canvas_mask, 0, bytes, st);
  if (e == hipSuccess) e = hipMemsetAsync(canvas_init, 0, bytes, st);
  if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "patch place: memset: %s", hipGetErrorString(e));
  const long total = (long)C * h * w;
  place_kernel<<<ufr::stream_grid(total, 256), 256, 0, st>>>(patch, mask, init, canvas_patch, canvas_mask,
                                                              canvas_init, C, h, w, H, W, y, x);
  return ufr::launched("patch_place");
}

extern "C" int ufr_patch_crop_f64(const float* canvas, const float* factor, double* dst, int C, int H, int W, int y,
                                  int x, int h, int w, ufr_stream_t stream) {
  UFR_REQUIRE(canvas && dst, "patch crop: null pointer");
  UFR_REQUIRE(C > 0 && h > 0 && w > 0 && H > 0 && W > 0, "patch crop: bad shape");
  UFR_REQUIRE(y >= 0 && x >= 0 && y + h <= H && x + w <= W, "patch crop: %dx%d at (%d,%d) leaves the %dx%d canvas",
              h, w, y, x, H, W);
  const long total = (long)C * h * w;
  crop_f64_kernel<<<ufr::stream_grid(total, 256), 256, 0, ufr::as_stream(stream)>>>(canvas, factor, dst, C, H, W, y,
                                                                                     x, h, w);
  return ufr::launched("patch_crop_f64");
}
