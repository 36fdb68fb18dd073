// pwc_warp.hip -- PWC-Net's backward warp (models/PWCNet.py:164-204) as one kernel forward, one backward.
//
// The reference builds a normalised sampling grid from the flow with FOUR broadcast elementwise ops per axis,
// calls grid_sample twice (features, and a tensor of ones for the validity mask), thresholds the warped ones
// at 0.0001 and multiplies: ~15 kernels and several feature-map-sized temporaries per pyramid level, plus
// their adjoints.  Its arithmetic, kept here operation for operation in float32:
//   vx = 2*(x + flow_x) / max(W-1, 1) - 1            (grid normalised with W-1 ...)
//   ix = ((vx + 1) * W - 1) / 2                      (... but sampled with align_corners=False: a quirk)
//   bilinear with zero padding, weights nw/ne/sw/se as in ATen's grid_sampler
//   mask = (sum of the in-bounds weights >= 0.0001);  out = sampled * mask
// HBM-streaming: a thread computes a pixel's four weights once and walks its slice of the channels.
// Backward: feature gradient scattered with float atomics to the four corners (ATen does the same), flow
// gradient from the four corner values; the mask is piecewise constant and carries no gradient.
#include "ufr_common.h"

namespace {

struct Tap {
  int x0, y0;            // north-west corner
  float nw, ne, sw, se;  // bilinear weights
  bool in_nw, in_ne, in_sw, in_se;
  float keep;            // 1 when the warped ones reach 0.0001, else 0
  float tx, ty;          // ix - x0, iy - y0 pieces needed by the flow gradient
  float ix, iy;
};

__device__ __forceinline__ Tap make_tap(float fx, float fy, int x, int y, int H, int W) {
  Tap t;
  const float dW = (float)max(W - 1, 1), dH = (float)max(H - 1, 1);
  const float vx = 2.0f * ((float)x + fx) / dW - 1.0f;
  const float vy = 2.0f * ((float)y + fy) / dH - 1.0f;
  t.ix = ((vx + 1.f) * (float)W - 1.f) / 2.f;
  t.iy = ((vy + 1.f) * (float)H - 1.f) / 2.f;
  const float fx0 = floorf(t.ix), fy0 = floorf(t.iy);
  // clamp before the int conversion: far-away (or non-finite) positions simply have no in-bounds corner
  t.x0 = (int)fminf(fmaxf(fx0, -2.0e6f), 2.0e6f);
  t.y0 = (int)fminf(fmaxf(fy0, -2.0e6f), 2.0e6f);
  const float x1 = fx0 + 1.f, y1 = fy0 + 1.f;
  t.nw = (x1 - t.ix) * (y1 - t.iy);
  t.ne = (t.ix - fx0) * (y1 - t.iy);
  t.sw = (x1 - t.ix) * (t.iy - fy0);
  t.se = (t.ix - fx0) * (t.iy - fy0);
  const bool xin0 = t.x0 >= 0 && t.x0 < W, xin1 = t.x0 + 1 >= 0 && t.x0 + 1 < W;
  const bool yin0 = t.y0 >= 0 && t.y0 < H, yin1 = t.y0 + 1 >= 0 && t.y0 + 1 < H;
  const bool finite = fabsf(t.ix) < 1.0e6f && fabsf(t.iy) < 1.0e6f;
  t.in_nw = finite && xin0 && yin0; t.in_ne = finite && xin1 && yin0;
  t.in_sw = finite && xin0 && yin1; t.in_se = finite && xin1 && yin1;
  float m = 0.f;                              // grid_sample(ones): same accumulation order as ATen
  if (t.in_nw) m += t.nw;
  if (t.in_ne) m += t.ne;
  if (t.in_sw) m += t.sw;
  if (t.in_se) m += t.se;
  t.keep = m >= 0.0001f ? 1.f : 0.f;
  return t;
}

// Thread = (pixel, channel slice): a workgroup is S waves over the same 64 pixels, wave s walking channels s, s + S, ...
// (coalesced along the pixels; S chosen by the host so that the coarse pyramid levels -- 12 x 40 pixels, 128 channels --
// still fill the chip: one thread per pixel walking every channel took 100-190 us there, latency-bound).
__global__ __launch_bounds__(1024) void pwc_warp_fwd(const float* __restrict__ x, const float* __restrict__ flo,
                                                     float* __restrict__ out, int B, int C, int H, int W, int S) {
  const size_t plane = (size_t)H * W;
  const long npix = (long)B * plane;
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const long idx = (long)blockIdx.x * 64 + lane;
  if (idx >= npix) return;
  const int px = (int)(idx % W), py = (int)((idx / W) % H), b = (int)(idx / (long)plane);
  const size_t pix = (size_t)py * W + px;
  const Tap t = make_tap(flo[((size_t)b * 2 + 0) * plane + pix], flo[((size_t)b * 2 + 1) * plane + pix], px, py, H, W);
  const size_t o_nw = (size_t)t.y0 * W + t.x0;
  for (int c = slice; c < C; c += S) {
    const float* im = x + ((size_t)b * C + c) * plane;
    float v = 0.f;
    if (t.in_nw) v += im[o_nw] * t.nw;
    if (t.in_ne) v += im[o_nw + 1] * t.ne;
    if (t.in_sw) v += im[o_nw + W] * t.sw;
    if (t.in_se) v += im[o_nw + W + 1] * t.se;
    out[((size_t)b * C + c) * plane + pix] = v * t.keep;
  }
}

// gx must be zero-filled by the caller side of this launch (done in the entry point).  The flow gradient's sum over the
// channels: per-slice partials through LDS, added in ascending slice order (fixed order: reproducible).
__global__ __launch_bounds__(1024) void pwc_warp_bwd(const float* __restrict__ x, const float* __restrict__ flo,
                                                     const float* __restrict__ gout, float* __restrict__ gx,
                                                     float* __restrict__ gflo, int B, int C, int H, int W, int S) {
  __shared__ float part[16][64][2];
  const size_t plane = (size_t)H * W;
  const long npix = (long)B * plane;
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const long idx = (long)blockIdx.x * 64 + lane;
  const bool live = idx < npix;
  float gix = 0.f, giy = 0.f;
  int px = 0, py = 0, b = 0;
  size_t pix = 0;
  if (live) {
    px = (int)(idx % W); py = (int)((idx / W) % H); b = (int)(idx / (long)plane);
    pix = (size_t)py * W + px;
    const Tap t = make_tap(flo[((size_t)b * 2 + 0) * plane + pix], flo[((size_t)b * 2 + 1) * plane + pix], px, py, H, W);
    const size_t o_nw = (size_t)t.y0 * W + t.x0;
    const float x0f = floorf(t.ix), y0f = floorf(t.iy), x1f = x0f + 1.f, y1f = y0f + 1.f;
    if (t.keep != 0.f) {
      for (int c = slice; c < C; c += S) {
        const float g = gout[((size_t)b * C + c) * plane + pix];
        const float* im = x + ((size_t)b * C + c) * plane;
        float* gi = gx + ((size_t)b * C + c) * plane;
        if (t.in_nw) { const float v = im[o_nw]; atomicAdd(&gi[o_nw], t.nw * g);
                       gix -= v * (y1f - t.iy) * g; giy -= v * (x1f - t.ix) * g; }
        if (t.in_ne) { const float v = im[o_nw + 1]; atomicAdd(&gi[o_nw + 1], t.ne * g);
                       gix += v * (y1f - t.iy) * g; giy -= v * (t.ix - x0f) * g; }
        if (t.in_sw) { const float v = im[o_nw + W]; atomicAdd(&gi[o_nw + W], t.sw * g);
                       gix -= v * (t.iy - y0f) * g; giy += v * (x1f - t.ix) * g; }
        if (t.in_se) { const float v = im[o_nw + W + 1]; atomicAdd(&gi[o_nw + W + 1], t.se * g);
                       gix += v * (t.iy - y0f) * g; giy += v * (t.ix - x0f) * g; }
      }
    }
  }
  part[slice][lane][0] = gix;
  part[slice][lane][1] = giy;
  __syncthreads();
  if (slice == 0 && live) {
    float sx = 0.f, sy = 0.f;
    for (int k = 0; k < S; ++k) {
      sx += part[k][lane][0];
      sy += part[k][lane][1];
    }
    // d ix / d grid = W / 2 (ATen's unnormalise multiplier), d grid / d flow = 2 / max(W-1, 1)
    const float dW = (float)max(W - 1, 1), dH = (float)max(H - 1, 1);
    gflo[((size_t)b * 2 + 0) * plane + pix] = (sx * (0.5f * (float)W)) / dW * 2.0f;
    gflo[((size_t)b * 2 + 1) * plane + pix] = (sy * (0.5f * (float)H)) / dH * 2.0f;
  }
}

// ---- the adjoint WITHOUT float atomics (owner-computes, as csrc/resample2d_owner.hip) ---------------------------------------------
// The scatter of `pwc_warp_bwd` costs four global float atomics per (pixel, channel) -- ~28 G lanes/s on gfx950 whatever their
// coalescing -- and a zero fill of grad_x in front; its result depends on the order the atomics retire in.  Two launches instead:
//   A  `pwc_warp_bwd_flow_kernel`   the flow gradient (a gather: as above) and, per 8 x 32 tile of SOURCE pixels, the bounding box
//      of the cells its pixels' kept, in-bounds corners land in (four integer atomicMin per wave into a table of boxes);
//   B  `pwc_warp_bwd_owner_kernel`  one workgroup OWNS an 8 x 32 tile of grad_x.  The geometry does not depend on the channel, so it is
//      resolved ONCE: the owner walks the source tiles whose box meets its tile and files every corner that lands in one of its
//      cells as (source pixel, weight) into that cell's slots (one integer LDS atomic per corner reserves the slot); every cell
//      then sorts its <= PW_K = 16 slots by source pixel and, channel by channel, GATHERS  grad_x[c][cell] = sum_s w_s * grad_out[c][src_s]
//      -- a fixed order of additions: the gradient is bit-reproducible -- and writes it with coalesced stores: every element of
//      grad_x exactly once, no zero fill.
// Corners beyond a cell's PW_K slots (a flow that compresses > 4x in both directions, or white noise of several cells; zero padding
// means the frame borders do not pile up here) go to the tile's overflow list and are added with float atomics AFTER the owner's
// stores (those cells are then not reproducible in the last bit -- as the scatter); more than PW_OV of them (a flow that folds the
// frame onto a few cells) and the tile takes the slow path: per channel an LDS accumulator and LDS float atomics.  ANY flow is served.
// The coarse pyramid levels have few tiles and many channels: their owners are cut into channel slices (grid y).
constexpr int PW_TH = 8, PW_TW = 32, PW_CELLS = PW_TH * PW_TW;   // tile of source pixels (the table's granularity) = tile of an owner
constexpr int PW_K = 16;                                          // slots per cell
constexpr int PW_MAX_LIST = 256;                                  // candidate tiles listed per pass
constexpr int PW_OV = 1024;                                       // corners beyond their cell's slots kept per owner

// A: workgroup = 64 pixels (two rows x 32 columns of one tile) x S channel slices; blockIdx = ((b * tiles_y + ty) * tiles_x + tx) * 4 + row pair
__global__ __launch_bounds__(1024) void pwc_warp_bwd_flow_kernel(const float* __restrict__ x, const float* __restrict__ flo,
                                                                 const float* __restrict__ gout, float* __restrict__ gflo,
                                                                 int* __restrict__ boxes, int B, int C, int H, int W, int S) {
  __shared__ float part[16][64][2];
  const size_t plane = (size_t)H * W;
  const int tiles_x = (W + PW_TW - 1) / PW_TW, tiles_y = (H + PW_TH - 1) / PW_TH;
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int tile = blockIdx.x >> 2, rp = blockIdx.x & 3;
  const int b = tile / (tiles_x * tiles_y), tr = tile - b * tiles_x * tiles_y;
  const int py = (tr / tiles_x) * PW_TH + 2 * rp + (lane >> 5), px = (tr % tiles_x) * PW_TW + (lane & 31);
  const bool live = py < H && px < W;
  float gix = 0.f, giy = 0.f;
  const size_t pix = live ? (size_t)py * W + px : 0;
  int mnx = 0x7f7f7f7f, mny = 0x7f7f7f7f, ngx = 0x7f7f7f7f, ngy = 0x7f7f7f7f;      // min x, min y, min -x, min -y of the cells reached
  if (live) {
    const Tap t = make_tap(flo[((size_t)b * 2 + 0) * plane + pix], flo[((size_t)b * 2 + 1) * plane + pix], px, py, H, W);
    const size_t o_nw = (size_t)t.y0 * W + t.x0;
    const float x0f = floorf(t.ix), y0f = floorf(t.iy), x1f = x0f + 1.f, y1f = y0f + 1.f;
    if (t.keep != 0.f) {
      if (slice == 0) {
        if (t.in_nw || t.in_sw) { mnx = min(mnx, t.x0); ngx = min(ngx, -t.x0); }
        if (t.in_ne || t.in_se) { mnx = min(mnx, t.x0 + 1); ngx = min(ngx, -(t.x0 + 1)); }
        if (t.in_nw || t.in_ne) { mny = min(mny, t.y0); ngy = min(ngy, -t.y0); }
        if (t.in_sw || t.in_se) { mny = min(mny, t.y0 + 1); ngy = min(ngy, -(t.y0 + 1)); }
      }
      for (int c = slice; c < C; c += S) {
        const float g = gout[((size_t)b * C + c) * plane + pix];
        const float* im = x + ((size_t)b * C + c) * plane;
        if (t.in_nw) { const float v = im[o_nw]; gix -= v * (y1f - t.iy) * g; giy -= v * (x1f - t.ix) * g; }
        if (t.in_ne) { const float v = im[o_nw + 1]; gix += v * (y1f - t.iy) * g; giy -= v * (t.ix - x0f) * g; }
        if (t.in_sw) { const float v = im[o_nw + W]; gix -= v * (t.iy - y0f) * g; giy += v * (x1f - t.ix) * g; }
        if (t.in_se) { const float v = im[o_nw + W + 1]; gix += v * (t.iy - y0f) * g; giy += v * (t.ix - x0f) * g; }
      }
    }
  }
  if (slice == 0) {                             // (wave-uniform: slice 0 is one whole wave, all of its pixels in one tile)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mnx = min(mnx, __shfl_xor(mnx, off, 64)); mny = min(mny, __shfl_xor(mny, off, 64));
      ngx = min(ngx, __shfl_xor(ngx, off, 64)); ngy = min(ngy, __shfl_xor(ngy, off, 64));
    }
    if (lane == 0 && mnx != 0x7f7f7f7f) {
      int* bx = boxes + (size_t)tile * 4;
      atomicMin(bx + 0, mnx); atomicMin(bx + 1, mny); atomicMin(bx + 2, ngx); atomicMin(bx + 3, ngy);
    }
  }
  part[slice][lane][0] = gix;
  part[slice][lane][1] = giy;
  __syncthreads();
  if (slice == 0 && live) {
    float sx = 0.f, sy = 0.f;
    for (int k = 0; k < S; ++k) {
      sx += part[k][lane][0];
      sy += part[k][lane][1];
    }
    const float dW = (float)max(W - 1, 1), dH = (float)max(H - 1, 1);
    gflo[((size_t)b * 2 + 0) * plane + pix] = (sx * (0.5f * (float)W)) / dW * 2.0f;
    gflo[((size_t)b * 2 + 1) * plane + pix] = (sy * (0.5f * (float)H)) / dH * 2.0f;
  }
}

// An owner whose cells hold at most N_ slots: this cell's slots sorted by source pixel (a source pixel reaches a cell through one
// corner only) = the order of the additions, then the channel loop with N_ x 4 gathers in flight per thread.
template <int N_>
__device__ __forceinline__ void pw_owner_finish(const float* __restrict__ gb, float* __restrict__ ob, size_t plane, int C, int n,
                                                const int* __restrict__ slot_src, const float* __restrict__ slot_w, size_t cell_pix,
                                                bool store) {
  int src[N_];
  float w[N_];
#pragma unroll
  for (int s = 0; s < N_; ++s) { src[s] = s < n ? slot_src[s] : 0x7fffffff; w[s] = s < n ? slot_w[s] : 0.f; }
#pragma unroll
  for (int i = 1; i < N_; ++i)
#pragma unroll
    for (int j = i; j > 0; --j) {
      const bool sw = src[j] < src[j - 1];
      const int a = src[j], b = src[j - 1];
      const float wa = w[j], wb = w[j - 1];
      src[j] = sw ? b : a; src[j - 1] = sw ? a : b;
      w[j] = sw ? wb : wa; w[j - 1] = sw ? wa : wb;
    }
  const int first = n > 0 ? src[0] : (int)cell_pix;                            // unused slots repeat a valid address (a hit)
#pragma unroll
  for (int s = 0; s < N_; ++s) src[s] = s < n ? src[s] : first;
  int c = 0;
  for (; c + 4 <= C; c += 4) {
    float g[4][N_];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int s = 0; s < N_; ++s) g[u][s] = gb[(size_t)(c + u) * plane + src[s]];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float v = 0.f;
#pragma unroll
      for (int s = 0; s < N_; ++s) v += s < n ? w[s] * g[u][s] : 0.f;
      if (store) ob[(size_t)(c + u) * plane + cell_pix] = v;
    }
  }
  for (; c < C; ++c) {
    float v = 0.f;
#pragma unroll
    for (int s = 0; s < N_; ++s) v += s < n ? w[s] * gb[(size_t)c * plane + src[s]] : 0.f;
    if (store) ob[(size_t)c * plane + cell_pix] = v;
  }
}

__global__ __launch_bounds__(PW_CELLS) void pwc_warp_bwd_owner_kernel(const float* __restrict__ flo, const float* __restrict__ gout,
                                                                     const int* __restrict__ boxes, float* __restrict__ gx, int B,
                                                                     int C, int H, int W, int c_per) {
  __shared__ int cnt[PW_CELLS];
  __shared__ int s_src[PW_CELLS * PW_K];
  __shared__ float s_w[PW_CELLS * PW_K];
  __shared__ int ov_cell[PW_OV], ov_src[PW_OV];
  __shared__ float ov_w[PW_OV];
  __shared__ float acc[PW_CELLS];
  __shared__ int list[PW_MAX_LIST];
  __shared__ int n_list, n_ov, nmax_s;
  const int tid = threadIdx.x;
  const int tiles_x = (W + PW_TW - 1) / PW_TW, tiles_y = (H + PW_TH - 1) / PW_TH, nT = tiles_x * tiles_y;
  const int b = blockIdx.x / nT, tr = blockIdx.x - b * nT;
  const int ry0 = (tr / tiles_x) * PW_TH, rx0 = (tr % tiles_x) * PW_TW;
  const int c0 = blockIdx.y * c_per, c1 = min(C, c0 + c_per);                  // this workgroup's channels (the small grids are cut in slices)
  const size_t plane = (size_t)H * W;
  const float* flo_b = flo + (size_t)b * 2 * plane;
  const float* gb = gout + ((size_t)b * C + c0) * plane;
  float* ob = gx + ((size_t)b * C + c0) * plane;
  const int cy = ry0 + (tid >> 5), cx = rx0 + (tid & 31);
  const bool cell_live = cy < H && cx < W;
  const size_t cell_pix = cell_live ? (size_t)cy * W + cx : 0;
  cnt[tid] = 0;
  if (tid == 0) { n_ov = 0; nmax_s = 0; }
  // walks the candidate tiles of one pass of the list; file = true: reserve slots, false (slow path): add channel c into `acc`
  auto walk = [&](int n, bool file, int c) {
    for (int li = 0; li < n; ++li) {
      const int t = list[li];
      const int py = (t / tiles_x) * PW_TH + (tid >> 5), px = (t % tiles_x) * PW_TW + (tid & 31);
      if (py >= H || px >= W) continue;
      const size_t pix = (size_t)py * W + px;
      const Tap tp = make_tap(flo_b[pix], flo_b[plane + pix], px, py, H, W);
      if (tp.keep == 0.f) continue;
      const float g = file ? 0.f : gb[(size_t)c * plane + pix];
      const int lx = tp.x0 - rx0, ly = tp.y0 - ry0;
      const bool inx0 = (unsigned)lx < (unsigned)PW_TW, inx1 = (unsigned)(lx + 1) < (unsigned)PW_TW;
      const bool iny0 = (unsigned)ly < (unsigned)PW_TH, iny1 = (unsigned)(ly + 1) < (unsigned)PW_TH;
      const bool hit[4] = {tp.in_nw && inx0 && iny0, tp.in_ne && inx1 && iny0, tp.in_sw && inx0 && iny1, tp.in_se && inx1 && iny1};
      const int cell[4] = {ly * PW_TW + lx, ly * PW_TW + lx + 1, (ly + 1) * PW_TW + lx, (ly + 1) * PW_TW + lx + 1};
      const float wt[4] = {tp.nw, tp.ne, tp.sw, tp.se};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!hit[q]) continue;
        if (file) {
          const int slot = atomicAdd(&cnt[cell[q]], 1);                       // ds_add_rtn_u32: the fast kind of LDS atomic
          if (slot < PW_K) {
            s_src[cell[q] * PW_K + slot] = (int)pix; s_w[cell[q] * PW_K + slot] = wt[q];
          } else {                                                             // beyond the slots: the tile's overflow list
            const int e = atomicAdd(&n_ov, 1);
            if (e < PW_OV) { ov_cell[e] = cell[q]; ov_src[e] = (int)pix; ov_w[e] = wt[q]; }
          }
        } else {
          atomicAdd(&acc[cell[q]], wt[q] * g);
        }
      }
    }
  };
  auto list_pass = [&](int t0) -> int {                                        // the tiles of [t0, t0 + PW_MAX_LIST) whose box meets this owner
    __syncthreads();                                                           // (the previous pass's list and count are read)
    if (tid == 0) n_list = 0;
    __syncthreads();
    for (int t = t0 + tid; t < min(nT, t0 + PW_MAX_LIST); t += PW_CELLS) {
      const int4 bx = *reinterpret_cast<const int4*>(boxes + ((size_t)b * nT + t) * 4);   // min x, min y, -max x, -max y
      if (bx.x != 0x7f7f7f7f && -bx.z >= rx0 && bx.x < rx0 + PW_TW && -bx.w >= ry0 && bx.y < ry0 + PW_TH) list[atomicAdd(&n_list, 1)] = t;
    }
    __syncthreads();
    return n_list;
  };
  for (int t0 = 0; t0 < nT; t0 += PW_MAX_LIST) {
    const int n = list_pass(t0);
    walk(n, true, 0);
  }
  __syncthreads();
  const int n_mine = min(cnt[tid], PW_K);
  atomicMax(&nmax_s, n_mine);
  __syncthreads();
  if (n_ov <= PW_OV) {
    const int nmax = nmax_s;                                                   // workgroup-uniform
    const int* ss = s_src + tid * PW_K;
    const float* sw = s_w + tid * PW_K;
    const int Cn = c1 - c0;
    if (nmax <= 2) pw_owner_finish<2>(gb, ob, plane, Cn, n_mine, ss, sw, cell_pix, cell_live);
    else if (nmax <= 4) pw_owner_finish<4>(gb, ob, plane, Cn, n_mine, ss, sw, cell_pix, cell_live);
    else if (nmax <= 6) pw_owner_finish<6>(gb, ob, plane, Cn, n_mine, ss, sw, cell_pix, cell_live);
    else if (nmax <= 8) pw_owner_finish<8>(gb, ob, plane, Cn, n_mine, ss, sw, cell_pix, cell_live);
    else if (nmax <= 12) pw_owner_finish<12>(gb, ob, plane, Cn, n_mine, ss, sw, cell_pix, cell_live);
    else pw_owner_finish<16>(gb, ob, plane, Cn, n_mine, ss, sw, cell_pix, cell_live);
    const int nov = n_ov;
    if (nov == 0) return;
    // the corners beyond a cell's slots: float atomics onto the sums just stored (the barrier drains this workgroup's stores first;
    // only this workgroup touches its tile).  Their order is the order they retire in: the last bits of such a cell may differ run to run.
    __syncthreads();
    for (int e = tid; e < nov; e += PW_CELLS) {
      const int cell = ov_cell[e];
      const size_t cp = (size_t)(ry0 + cell / PW_TW) * W + rx0 + cell % PW_TW;
      const float w = ov_w[e];
      const size_t sp = (size_t)ov_src[e];
      for (int c = 0; c < Cn; ++c) atomicAdd(&ob[(size_t)c * plane + cp], w * gb[(size_t)c * plane + sp]);
    }
    return;
  }
  for (int c = 0; c < c1 - c0; ++c) {                                          // the slow path (a flow that folds the frame onto a few cells)
    acc[tid] = 0.f;
    for (int t0 = 0; t0 < nT; t0 += PW_MAX_LIST) {
      const int n = list_pass(t0);
      walk(n, false, c);
    }
    __syncthreads();
    if (cell_live) ob[(size_t)c * plane + cell_pix] = acc[tid];
    __syncthreads();
  }
}

// channel slices per pixel: enough threads for ~4 waves per SIMD on the small grids, at most 16 (and at most C)
static int warp_slices(long npix, int C) {
  int s = 1;
  while (s < 16 && s * 2 <= C && npix * s < 262144) s *= 2;
  return s;
}

}  // namespace

extern "C" int ufr_pwc_warp_forward(const float* x, const float* flow, float* out, int B, int C, int H, int W,
                                    ufr_stream_t stream) {
  UFR_REQUIRE(x && flow && out, "pwc warp forward: null pointer");
  UFR_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "pwc warp forward: bad shape");
  const long npix = (long)B * H * W;
  const int S = warp_slices(npix, C);
  pwc_warp_fwd<<<(unsigned)((npix + 63) / 64), 64 * S, 0, ufr::as_stream(stream)>>>(x, flow, out, B, C, H, W, S);
  return ufr::launched("pwc_warp_fwd");
}

extern "C" int ufr_pwc_warp_backward(const float* x, const float* flow, const float* grad_out, float* grad_x,
                                     float* grad_flow, int B, int C, int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(x && flow && grad_out && grad_x && grad_flow, "pwc warp backward: null pointer");
  UFR_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "pwc warp backward: bad shape");
  hipStream_t st = ufr::as_stream(stream);
  hipError_t e = hipMemsetAsync(grad_x, 0, sizeof(float) * (size_t)B * C * H * W, st);
  if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "pwc warp backward: memset: %s", hipGetErrorString(e));
  const long npix = (long)B * H * W;
  const int S = warp_slices(npix, C);
  pwc_warp_bwd<<<(unsigned)((npix + 63) / 64), 64 * S, 0, st>>>(x, flow, grad_out, grad_x, grad_flow, B, C, H, W, S);
  return ufr::launched("pwc_warp_bwd");
}

extern "C" long ufr_pwc_warp_backward_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return (long)B * ((H + PW_TH - 1) / PW_TH) * ((W + PW_TW - 1) / PW_TW) * 16L;
}

extern "C" int ufr_pwc_warp_backward_owner(const float* x, const float* flow, const float* grad_out, float* grad_x, float* grad_flow,
                                           void* workspace, long workspace_bytes, int B, int C, int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(x && flow && grad_out && grad_x && grad_flow && workspace, "pwc warp backward (owner): null pointer");
  UFR_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && (long)H * W < (1L << 31), "pwc warp backward (owner): bad shape");
  const long need = ufr_pwc_warp_backward_workspace_bytes(B, H, W);
  UFR_REQUIRE(workspace_bytes >= need, "pwc warp backward (owner): workspace too small (%ld < %ld bytes)", workspace_bytes, need);
  UFR_REQUIRE((reinterpret_cast<size_t>(workspace) & 15) == 0, "pwc warp backward (owner): the workspace must be 16-byte aligned");
  hipStream_t st = ufr::as_stream(stream);
  hipError_t e = hipMemsetAsync(workspace, 0x7f, (size_t)need, st);            // every box empty: (min, min, min of -x, min of -y) = 0x7f7f7f7f
  if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "pwc warp backward (owner): memset: %s", hipGetErrorString(e));
  const int tiles = B * ((H + PW_TH - 1) / PW_TH) * ((W + PW_TW - 1) / PW_TW);
  const long npix = (long)B * H * W;
  const int S = warp_slices(npix, C);
  pwc_warp_bwd_flow_kernel<<<tiles * 4, 64 * S, 0, st>>>(x, flow, grad_out, grad_flow, static_cast<int*>(workspace), B, C, H, W, S);
  int rc = ufr::launched("pwc_warp_bwd_flow_kernel");
  if (rc != UFR_OK) return rc;
  int cs = 1;                                   // channel slices: >= ~512 owners where the channels allow (>= 8 per slice)
  while (tiles * cs < 512 && C / (cs * 2) >= 8) cs *= 2;
  const int c_per = ((C + cs - 1) / cs + 3) / 4 * 4;
  pwc_warp_bwd_owner_kernel<<<dim3(tiles, (C + c_per - 1) / c_per), PW_CELLS, 0, st>>>(flow, grad_out, static_cast<const int*>(workspace), grad_x, B, C,
                                                                                     H, W, c_per);
  return ufr::launched("pwc_warp_bwd_owner_kernel");
}
