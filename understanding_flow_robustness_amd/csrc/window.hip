// window.hip -- "cone of influence" windows for the patch attack (gfx950).
//
// patch_attacks/main.py:537-600 multiplies the image gradient by the patch mask and changes only the
// masked pixels between iterations.  For a purely convolutional encoder prefix (FlowNetC conv1-3,
// models/FlowNetC.py:96-104) that means
//   * backward: d loss/d patch only needs the encoder's adjoint inside the patch's forward cone, and
//   * forward (iterations >= 2 of one attack() call): encoder features outside the cone are unchanged.
// So the encoder runs on a small window around the patch; these kernels move data between the
// full-size tensors and the window at an origin that lives in DEVICE memory (a captured HIP graph
// follows new patch placements without re-capture):
//   ufr_cone_window    mask -> bounding box -> cone through the conv chain -> window origin
//   ufr_window_gather  full tensor -> window (optionally zeroing the inexact rim)
//   ufr_window_scatter window -> full tensor (optionally skipping the inexact rim)
// The "inexact rim": a zero-padded convolution on the window differs from the full-image one within
// `margin` cells of a window edge, except where that edge IS the image edge (same padding).
#include <climits>

#include "ufr_common.h"

namespace {

// win[n] = {y0, x0, need_h, need_w, ymin, ymax, xmin, xmax}; y0/x0/need_* in input pixels
constexpr int kWinInts = 8;

__global__ void bbox_init_kernel(int* __restrict__ win, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int* w = win + n * kWinInts;
  w[0] = w[1] = w[2] = w[3] = 0;
  w[4] = INT_MAX; w[5] = -1; w[6] = INT_MAX; w[7] = -1;
}

// grid (blocks, N): bounding box of mask != 0 over all channels of sample n
__global__ void bbox_kernel(const float* __restrict__ mask, long bstride, int C, int H, int W,
                            int* __restrict__ win) {
  const int n = blockIdx.y;
  const float* m = mask + (long)n * bstride;
  const long total = (long)C * H * W;
  int ymin = INT_MAX, ymax = -1, xmin = INT_MAX, xmax = -1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    if (m[i] != 0.f) {
      const int r = (int)(i % ((long)H * W));
      const int y = r / W, x = r - y * W;
      ymin = min(ymin, y); ymax = max(ymax, y); xmin = min(xmin, x); xmax = max(xmax, x);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    ymin = min(ymin, __shfl_xor(ymin, o)); ymax = max(ymax, __shfl_xor(ymax, o));
    xmin = min(xmin, __shfl_xor(xmin, o)); xmax = max(xmax, __shfl_xor(xmax, o));
  }
  if ((threadIdx.x & 63) == 0 && ymax >= 0) {
    int* w = win + n * kWinInts;
    atomicMin(w + 4, ymin); atomicMax(w + 5, ymax); atomicMin(w + 6, xmin); atomicMax(w + 7, xmax);
  }
}

__device__ __forceinline__ int floor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
__device__ __forceinline__ int ceil_div_i(int a, int b) { return -floor_div(-a, b); }

// One axis: [lo, hi] (input pixels, inclusive) -> window origin / needed extent in pixels.
// Cone of an interval through conv(k, s, p): outputs o with [s*o - p, s*o - p + k - 1] meeting it.
__device__ void cone_axis(int lo, int hi, int size, const ufr_cone_chain& ch, int win, int* origin,
                          int* need) {
  int total = 1;
  for (int l = 0; l < ch.n_layers; ++l) total *= ch.stride[l];
  const int cells = size / total;          // window grid = cells of the deepest level
  if (hi < lo) { *origin = 0; *need = 0; return; }
  int need_lo = INT_MAX, need_hi = -1, n = size, jump = 1, t = 0;
  for (int l = 0; l < ch.n_layers; ++l) {
    const int k = ch.kernel[l], s = ch.stride[l], p = ch.pad[l];
    const int n_out = (n + 2 * p - k) / s + 1;
    lo = max(ceil_div_i(lo + p - (k - 1), s), 0);
    hi = min(floor_div(hi + p, s), n_out - 1);
    n = n_out; jump *= s;
    while (t < ch.n_taps && ch.tap_layer[t] == l) {
      const int per = total / jump;        // cells of this level per window-grid cell
      need_lo = min(need_lo, floor_div(lo - ch.tap_margin[t], per));
      need_hi = max(need_hi, floor_div(hi + ch.tap_margin[t], per));
      ++t;
    }
  }
  need_lo = max(need_lo, 0); need_hi = min(need_hi, cells - 1);
  const int cnt = need_hi - need_lo + 1, wcells = win / total;
  int o = need_lo - max(wcells - cnt, 0) / 2;
  o = min(max(o, 0), max(cells - wcells, 0));
  *origin = o * total; *need = cnt * total;
}

__global__ void cone_finalize_kernel(int* __restrict__ win, int N, int H, int W, ufr_cone_chain ch, int win_h,
                                     int win_w, float* __restrict__ overflow) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int* w = win + n * kWinInts;
  cone_axis(w[4], w[5], H, ch, win_h, w + 0, w + 2);
  cone_axis(w[6], w[7], W, ch, win_w, w + 1, w + 3);
  if ((w[2] > win_h || w[3] > win_w) && overflow) atomicAdd(overflow, 1.0f);
}

// One thread per window element.  GATHER: dst[n,c,i,j] = src[n,c,y0+i,x0+j] (0 on the inexact rim);
// SCATTER: dst[n,c,y0+i,x0+j] = src[n,c,i,j] (rim skipped).
template <bool GATHER>
__global__ void window_copy_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                   const int* __restrict__ win, int n_win, int C, int Hf, int Wf, int wh, int ww,
                                   int level_stride, int margin, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % ww);
    long r = i / ww;
    const int ii = (int)(r % wh); r /= wh;
    const int c = (int)(r % C);
    const int n = (int)(r / C);
    const int* w = win + (n % n_win) * kWinInts;
    // clamped: an origin that was never computed cannot send the copy out of bounds
    const int y0 = min(max(w[0] / level_stride, 0), Hf - wh), x0 = min(max(w[1] / level_stride, 0), Wf - ww);
    const bool rim = (ii < margin && y0 > 0) || (ii >= wh - margin && y0 + wh < Hf) ||
                     (j < margin && x0 > 0) || (j >= ww - margin && x0 + ww < Wf);
    const long full = (((long)n * C + c) * Hf + (y0 + ii)) * Wf + (x0 + j);
    if (GATHER) dst[i] = rim ? 0.f : src[full];
    else if (!rim) dst[full] = src[i];
  }
}

// The window of a chunk-major float32 tensor [chunks][N * Hf * Wf][32] (the engine's gradient sums) as a chunk-major
// window tensor [chunks][N_dst * wh * ww][32], rim zeroed like window_copy_kernel<true>; images n >= N of dst are not touched.
__global__ void window_gather_chunks_kernel(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ win,
                                            int n_win, int N, int N_dst, int chunks, int Hf, int Wf, int wh, int ww, int level_stride,
                                            int margin, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i & 7);                      // float4 of the pixel's 32 channels
    long r = i >> 3;
    const int j = (int)(r % ww); r /= ww;
    const int ii = (int)(r % wh); r /= wh;
    const int n = (int)(r % N);
    const int ch = (int)(r / N);
    const int* w = win + (n % n_win) * kWinInts;
    const int y0 = min(max(w[0] / level_stride, 0), Hf - wh), x0 = min(max(w[1] / level_stride, 0), Wf - ww);
    const bool rim = (ii < margin && y0 > 0) || (ii >= wh - margin && y0 + wh < Hf) ||
                     (j < margin && x0 > 0) || (j >= ww - margin && x0 + ww < Wf);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!rim) v = *reinterpret_cast<const float4*>(src + (((long)ch * N * Hf + (long)n * Hf + y0 + ii) * Wf + x0 + j) * 32 + q * 4);
    *reinterpret_cast<float4*>(dst + (((long)ch * N_dst * wh + (long)n * wh + ii) * ww + j) * 32 + q * 4) = v;
  }
}

int check_window(const char* what, int N, int C, int Hf, int Wf, int wh, int ww, int level_stride, int margin,
                 int n_win) {
  UFR_REQUIRE(N > 0 && C > 0 && Hf > 0 && Wf > 0, "%s: bad shape N=%d C=%d H=%d W=%d", what, N, C, Hf, Wf);
  UFR_REQUIRE(wh > 0 && ww > 0 && wh <= Hf && ww <= Wf, "%s: window %dx%d does not fit %dx%d", what, wh, ww, Hf, Wf);
  UFR_REQUIRE(level_stride > 0 && margin >= 0 && 2 * margin <= wh && 2 * margin <= ww, "%s: bad stride/margin", what);
  UFR_REQUIRE(n_win > 0 && n_win <= N, "%s: n_win=%d out of range", what, n_win);
  return UFR_OK;
}

}  // namespace

extern "C" int ufr_cone_window(const float* mask, int N, long mask_bstride, int C, int H, int W,
                               const ufr_cone_chain* chain, int win_h, int win_w, int* win, float* overflow,
                               ufr_stream_t stream) {
  UFR_REQUIRE(mask && chain && win, "cone_window: null pointer");
  UFR_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "cone_window: bad shape N=%d C=%d H=%d W=%d", N, C, H, W);
  UFR_REQUIRE(chain->n_layers > 0 && chain->n_layers <= UFR_MAX_CONE_LAYERS && chain->n_taps > 0 &&
                  chain->n_taps <= UFR_MAX_CONE_LAYERS, "cone_window: bad chain");
  int total = 1;
  for (int l = 0; l < chain->n_layers; ++l) {
    UFR_REQUIRE(chain->kernel[l] > 0 && chain->stride[l] > 0 && chain->pad[l] >= 0, "cone_window: bad layer %d", l);
    total *= chain->stride[l];
  }
  for (int t = 0; t < chain->n_taps; ++t)
    UFR_REQUIRE(chain->tap_layer[t] >= 0 && chain->tap_layer[t] < chain->n_layers &&
                    (t == 0 || chain->tap_layer[t] >= chain->tap_layer[t - 1]) && chain->tap_margin[t] >= 0,
                "cone_window: taps must be sorted by layer");
  UFR_REQUIRE(H % total == 0 && W % total == 0, "cone_window: %dx%d is not a multiple of the chain stride %d", H, W, total);
  UFR_REQUIRE(win_h > 0 && win_w > 0 && win_h % total == 0 && win_w % total == 0 && win_h <= H && win_w <= W,
              "cone_window: window %dx%d must be a multiple of %d inside %dx%d", win_h, win_w, total, H, W);
  hipStream_t st = ufr::as_stream(stream);
  bbox_init_kernel<<<ufr::ceil_div(N, 64), 64, 0, st>>>(win, N);
  const long per = (long)C * H * W;
  dim3 grid(ufr::stream_grid(per, 256) > 256 ? 256 : ufr::stream_grid(per, 256), N);
  bbox_kernel<<<grid, 256, 0, st>>>(mask, mask_bstride, C, H, W, win);
  cone_finalize_kernel<<<ufr::ceil_div(N, 64), 64, 0, st>>>(win, N, H, W, *chain, win_h, win_w, overflow);
  return ufr::launched("cone_window");
}

extern "C" int ufr_window_gather(const float* src, float* dst, const int* win, int n_win, int N, int C, int Hs,
                                 int Ws, int wh, int ww, int level_stride, int margin, ufr_stream_t stream) {
  UFR_REQUIRE(src && dst && win, "window_gather: null pointer");
  if (int rc = check_window("window_gather", N, C, Hs, Ws, wh, ww, level_stride, margin, n_win)) return rc;
  const long total = (long)N * C * wh * ww;
  window_copy_kernel<true><<<ufr::stream_grid(total, 256), 256, 0, ufr::as_stream(stream)>>>(
      src, dst, win, n_win, C, Hs, Ws, wh, ww, level_stride, margin, total);
  return ufr::launched("window_gather");
}

extern "C" int ufr_window_scatter(const float* src, float* dst, const int* win, int n_win, int N, int C, int Hd,
                                  int Wd, int wh, int ww, int level_stride, int margin, ufr_stream_t stream) {
  UFR_REQUIRE(src && dst && win, "window_scatter: null pointer");
  if (int rc = check_window("window_scatter", N, C, Hd, Wd, wh, ww, level_stride, margin, n_win)) return rc;
  const long total = (long)N * C * wh * ww;
  window_copy_kernel<false><<<ufr::stream_grid(total, 256), 256, 0, ufr::as_stream(stream)>>>(
      src, dst, win, n_win, C, Hd, Wd, wh, ww, level_stride, margin, total);
  return ufr::launched("window_scatter");
}

extern "C" int ufr_window_gather_chunks(const float* src, float* dst, const int* win, int n_win, int N, int N_dst, int chunks,
                                        int Hs, int Ws, int wh, int ww, int level_stride, int margin, ufr_stream_t stream) {
  UFR_REQUIRE(src && dst && win, "window_gather_chunks: null pointer");
  if (int rc = check_window("window_gather_chunks", N, chunks, Hs, Ws, wh, ww, level_stride, margin, n_win)) return rc;
  UFR_REQUIRE(N_dst >= N, "window_gather_chunks: destination holds %d images, source %d", N_dst, N);
  const long total = (long)chunks * N * wh * ww * 8;
  window_gather_chunks_kernel<<<ufr::stream_grid(total, 256), 256, 0, ufr::as_stream(stream)>>>(
      src, dst, win, n_win, N, N_dst, chunks, Hs, Ws, wh, ww, level_stride, margin, total);
  return ufr::launched("window_gather_chunks");
}
