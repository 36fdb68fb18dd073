// raft_altcorr_mfma.hip -- RAFT's on-the-fly correlation (models/alt_cuda_corr/correlation_kernel.cu:18-256, called per
// pyramid level from models/raft/corr.py:109-137) as gather-GEMMs on the fp32 matrix cores, all pyramid levels in ONE launch.
//
// For pixel p = (h1, w1) with coordinate (x, y) / 2^l on level l, corner (fx, fy) = floor, fraction (dx, dy):
//   s[iy][ix] = <fmap1[p, :], fmap2_l[fy - r + iy, fx - r + ix, :]>      iy, ix in [0, 2r + 1], 0 outside the image
//   corr[oy + rd * ox] = bilinear blend of s[oy..oy+1][ox..ox+1]         (raft_corr.hip has the scalar form)
// The windows of 16 consecutive pixels of a row overlap almost completely when the flow is smooth, so a tile of 16 pixels
// computes S[p, q] = <fmap1[p], fmap2[q]> for the 16-pixel row segments q of the bounding box of its windows as
// `v_mfma_f32_16x16x4_f32` products (M = 16 pixels, N = 16 fmap2 pixels, K = C channels: exact fp32 products, fp32
// accumulation) and every pixel keeps the entries that fall into its own window.  The bounding box is computed on the fly:
// ANY coordinate field is handled (a discontinuous one just walks more segments), nothing is assumed about the flow.
//   forward   tile x level workgroups, the box's segments dealt to the 4 waves; S -> LDS window -> blend -> [B, L*rd*rd, H, W]
//   d/d fmap1 tile workgroups looping over the levels, wave = channel quarter: g1[p, c] (+)= sum_q gs[p, q] fmap2[q, c]
//   d/d fmap2 owner-computes, no atomics on the fine levels: a workgroup owns 16 fmap2 pixels, scans the table of the
//             tiles' boxes for the tiles that reach them and accumulates gs^T fmap1 in registers
// (the reference's kernels do one 4 x 8 pixel block per workgroup with a serial channel loop; round 2's form here was one
// workgroup per pixel on the vector ALU: 0.026 of the fp32 peak, 43x the algorithmic bytes through L2).
#include <climits>

#include "ufr_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int AC_TP = 16;                    // pixels per tile (one MFMA M block)

struct AcLevels {                            // by value in the kernel arguments
  int n;
  const float* f2[4];
  float* g2[4];
  int H2[4], W2[4];
  float cscale[4];                           // coords are multiplied by this (1 / 2^l, corr.py:126)
  int tile0[5];                              // d/d fmap2: first workgroup of each level's fmap2 tiles
  int split[4];                              // d/d fmap2: workgroups per fmap2 tile (each scans a slice of the pixel tiles)
};

struct PixelWindow {
  int cx, cy;                                // first fmap2 column / row of the window (corner - r)
  float dx, dy;
};

// window of pixel (b, h1, w1) on a level; invalid (never inside an image) for w1 >= W1
__device__ __forceinline__ PixelWindow pixel_window(const float* __restrict__ coords, int planar, int b, int h1, int w1, int H1,
                                                    int W1, float cscale, int r) {
  PixelWindow pw{0x3fffffff, 0x3fffffff, 0.f, 0.f};
  if (w1 >= W1) return pw;
  const size_t hw = (size_t)H1 * W1, pix = (size_t)h1 * W1 + w1;
  const float x = (planar ? coords[((size_t)b * 2 + 0) * hw + pix] : coords[((size_t)b * hw + pix) * 2 + 0]) * cscale;
  const float y = (planar ? coords[((size_t)b * 2 + 1) * hw + pix] : coords[((size_t)b * hw + pix) * 2 + 1]) * cscale;
  const float fx = floorf(x), fy = floorf(y);
  pw.dx = x - fx;
  pw.dy = y - fy;
  // clamp far-away (or non-finite) windows so the integer arithmetic cannot overflow; they stay outside every image
  pw.cx = (int)fminf(fmaxf(fx, -1.0e6f), 1.0e6f) - r;
  pw.cy = (int)fminf(fmaxf(fy, -1.0e6f), 1.0e6f) - r;
  if (!(fabsf(x) < 1.0e6f) || !(fabsf(y) < 1.0e6f)) pw.cx = pw.cy = 0x3fffffff;
  return pw;
}

// Bounding box of the windows of a tile's valid pixels that meet the image, clipped to it: rows [y0, y1), columns [x0, x1)
struct Box { int y0, y1, x0, x1; };

__device__ __forceinline__ bool meets(const PixelWindow& pw, int gd, int H2, int W2) {
  return pw.cx < W2 && pw.cx + gd > 0 && pw.cy < H2 && pw.cy + gd > 0;
}

// ------------------------------------------------------------------------------------------------ forward
template <int R, int CPG>                    // CPG = C / 4: channels per lane group
__global__ __launch_bounds__(256) void altcorr_mfma_fwd(const float* __restrict__ f1, const AcLevels lv,
                                                        const float* __restrict__ coords, int planar, float* __restrict__ out,
                                                        int B, int H1, int W1, float scale) {
  constexpr int rd = 2 * R + 1, gd = rd + 1, npt = gd * gd, C = 4 * CPG;
  __shared__ float s[AC_TP][npt];
  __shared__ int cxs[AC_TP], cys[AC_TP];
  __shared__ float dxs[AC_TP], dys[AC_TP];
  __shared__ int bb[4];
  const int l = blockIdx.y, H2 = lv.H2[l], W2 = lv.W2[l];
  const float* __restrict__ f2 = lv.f2[l];
  const int tiles_x = (W1 + AC_TP - 1) / AC_TP;
  const int tile = blockIdx.x, b = tile / (H1 * tiles_x), rem = tile - b * H1 * tiles_x, h1 = rem / tiles_x, w0 = (rem % tiles_x) * AC_TP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { bb[0] = INT_MAX; bb[1] = INT_MIN; bb[2] = INT_MAX; bb[3] = INT_MIN; }
  for (int i = tid; i < AC_TP * npt; i += 256) (&s[0][0])[i] = 0.f;
  __syncthreads();
  if (tid < AC_TP) {
    const PixelWindow pw = pixel_window(coords, planar, b, h1, w0 + tid, H1, W1, lv.cscale[l], R);
    cxs[tid] = pw.cx; cys[tid] = pw.cy; dxs[tid] = pw.dx; dys[tid] = pw.dy;
    if (meets(pw, gd, H2, W2)) {
      atomicMin(&bb[0], pw.cy); atomicMax(&bb[1], pw.cy);
      atomicMin(&bb[2], pw.cx); atomicMax(&bb[3], pw.cx);
    }
  }
  __syncthreads();
  if (bb[0] != INT_MAX) {
    const int y0 = max(bb[0], 0), y1 = min(bb[1] + gd, H2), x0 = max(bb[2], 0), x1 = min(bb[3] + gd, W2);
    const int ncb = (x1 - x0 + 15) >> 4, nblk = (y1 - y0) * ncb;
    const int pi = lane & 15, g = lane >> 4;
    // A operand: this lane's pixel, channels [g * CPG, (g + 1) * CPG) -- the MFMA's k index is the lane group, step j takes
    // channel g * CPG + j of BOTH operands, so every lane reads CPG contiguous floats
    float a[CPG];
    {
      const float* ap = f1 + (((size_t)b * H1 + h1) * W1 + min(w0 + pi, W1 - 1)) * C + g * CPG;
#pragma unroll
      for (int j = 0; j < CPG; j += 4) {
        const float4 v = *reinterpret_cast<const float4*>(ap + j);
        a[j] = v.x; a[j + 1] = v.y; a[j + 2] = v.z; a[j + 3] = v.w;
      }
    }
    int cyr[4], cxr[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) { cyr[r4] = cys[4 * g + r4]; cxr[r4] = cxs[4 * g + r4]; }
    for (int blk = wave; blk < nblk; blk += 4) {
      const int hy = y0 + blk / ncb, xb = x0 + (blk % ncb) * 16, q = xb + pi;
      const float* bp = f2 + (((size_t)b * H2 + hy) * W2 + min(q, W2 - 1)) * C + g * CPG;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < CPG; j += 4) {
        const float4 v = *reinterpret_cast<const float4*>(bp + j);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], v.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 1], v.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 2], v.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 3], v.w, acc1, 0, 0, 0);
      }
      // D[row = pixel 4 g + r][col = fmap2 pixel q]: keep what falls into that pixel's window
      if (q < W2) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int iy = hy - cyr[r4], ix = q - cxr[r4];
          if ((unsigned)iy < (unsigned)gd && (unsigned)ix < (unsigned)gd) s[4 * g + r4][iy * gd + ix] = acc0[r4] + acc1[r4];
        }
      }
    }
  }
  __syncthreads();
  // blend (correlation_kernel.cu:92-115): 16 consecutive pixels per output channel = 64 contiguous bytes
  const size_t plane = (size_t)H1 * W1;
  for (int t = tid; t < AC_TP * rd * rd; t += 256) {
    const int i = t & 15, o = t >> 4, ox = o / rd, oy = o - ox * rd;
    if (w0 + i >= W1) continue;
    const float dx = dxs[i], dy = dys[i];
    const float* si = s[i];
    const float v = (1 - dy) * (1 - dx) * si[oy * gd + ox] + (1 - dy) * dx * si[oy * gd + ox + 1] +
                    dy * (1 - dx) * si[(oy + 1) * gd + ox] + dy * dx * si[(oy + 1) * gd + ox + 1];
    out[(((size_t)b * lv.n + l) * rd * rd + o) * plane + (size_t)h1 * W1 + w0 + i] = v * scale;
  }
}

// gs[i][iy * gd + ix] of the tile's pixels: the adjoint of the bilinear blend (correlation_kernel.cu:196-214)
template <int R>
__device__ __forceinline__ void tile_gs(float (*gs)[(2 * R + 2) * (2 * R + 2)], const float* __restrict__ gout, size_t base,
                                        size_t plane, int w0, int W1, const float* dxs, const float* dys, float scale, int tid,
                                        int nthreads) {
  constexpr int rd = 2 * R + 1, gd = rd + 1, npt = gd * gd;
  for (int t = tid; t < AC_TP * npt; t += nthreads) {
    const int i = t & 15, pt = t >> 4, iy = pt / gd, ix = pt - iy * gd;
    float g = 0.f;
    if (w0 + i < W1) {
      const float* gp = gout + base + w0 + i;
      const float dx = dxs[i], dy = dys[i];
      if (iy > 0 && ix > 0)   g += gp[plane * ((iy - 1) + rd * (ix - 1))] * dy * dx;
      if (iy > 0 && ix < rd)  g += gp[plane * ((iy - 1) + rd * ix)] * dy * (1 - dx);
      if (iy < rd && ix > 0)  g += gp[plane * (iy + rd * (ix - 1))] * (1 - dy) * dx;
      if (iy < rd && ix < rd) g += gp[plane * (iy + rd * ix)] * (1 - dy) * (1 - dx);
    }
    gs[i][pt] = g * scale;
  }
}

// ------------------------------------------------------------------------------------------------ d / d fmap1
// g1[p, c] (+)= sum over levels and fmap2 pixels q of gs_l[p, q] fmap2_l[q, c].  Wave w owns channels [w C/4, (w+1) C/4):
// no reduction between waves, one writer per element (accumulate = read-add-store).
template <int R, int CPG>
__global__ __launch_bounds__(256) void altcorr_mfma_bwd1(const AcLevels lv, const float* __restrict__ coords, int planar,
                                                         const float* __restrict__ gout, float* __restrict__ g1, int B, int H1,
                                                         int W1, float scale, int accumulate) {
  constexpr int rd = 2 * R + 1, gd = rd + 1, npt = gd * gd, C = 4 * CPG, NT = CPG / 16;
  __shared__ float gs[AC_TP][npt];
  __shared__ int cxs[AC_TP], cys[AC_TP];
  __shared__ float dxs[AC_TP], dys[AC_TP];
  __shared__ int bb[4];
  const int tiles_x = (W1 + AC_TP - 1) / AC_TP;
  const int tile = blockIdx.x, b = tile / (H1 * tiles_x), rem = tile - b * H1 * tiles_x, h1 = rem / tiles_x, w0 = (rem % tiles_x) * AC_TP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pi = lane & 15, g = lane >> 4;
  const size_t plane = (size_t)H1 * W1;
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int l = 0; l < lv.n; ++l) {
    const int H2 = lv.H2[l], W2 = lv.W2[l];
    const float* __restrict__ f2 = lv.f2[l];
    __syncthreads();                                     // the previous level's gs / windows are no longer read
    if (tid == 0) { bb[0] = INT_MAX; bb[1] = INT_MIN; bb[2] = INT_MAX; bb[3] = INT_MIN; }
    __syncthreads();
    if (tid < AC_TP) {
      const PixelWindow pw = pixel_window(coords, planar, b, h1, w0 + tid, H1, W1, lv.cscale[l], R);
      cxs[tid] = pw.cx; cys[tid] = pw.cy; dxs[tid] = pw.dx; dys[tid] = pw.dy;
      if (meets(pw, gd, H2, W2)) {
        atomicMin(&bb[0], pw.cy); atomicMax(&bb[1], pw.cy);
        atomicMin(&bb[2], pw.cx); atomicMax(&bb[3], pw.cx);
      }
    }
    __syncthreads();
    tile_gs<R>(gs, gout, (((size_t)b * lv.n + l) * rd * rd) * plane + (size_t)h1 * W1, plane, w0, W1, dxs, dys, scale, tid, 256);
    __syncthreads();
    if (bb[0] == INT_MAX) continue;
    const int y0 = max(bb[0], 0), y1 = min(bb[1] + gd, H2), x0 = max(bb[2], 0), x1 = min(bb[3] + gd, W2);
    const int ncb = (x1 - x0 + 15) >> 4, nblk = (y1 - y0) * ncb;
    const int cyi = cys[pi], cxi = cxs[pi];
    for (int blk = 0; blk < nblk; ++blk) {
      const int hy = y0 + blk / ncb, xb = x0 + (blk % ncb) * 16;
      const int iy = hy - cyi;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        // A[m = pixel pi][k = fmap2 pixel q]: the pixel's gradient for that window point (0 outside window / image)
        const int q = xb + 4 * kk + g, ix = q - cxi;
        const float av = ((unsigned)iy < (unsigned)gd && (unsigned)ix < (unsigned)gd && q < W2) ? gs[pi][iy * gd + ix] : 0.f;
        // B[k = q][n = channel]: lanes 0-15 read 16 consecutive channels
        const float* bp = f2 + (((size_t)b * H2 + hy) * W2 + min(q, W2 - 1)) * C + wave * CPG + pi;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[nt * 16], acc[nt], 0, 0, 0);
      }
    }
  }
  // D[row = pixel 4 g + r][col = channel wave * CPG + nt * 16 + pi]
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int w1 = w0 + 4 * g + r4;
    if (w1 >= W1) continue;
    float* op = g1 + (((size_t)b * H1 + h1) * W1 + w1) * C + wave * CPG + pi;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) op[nt * 16] = (accumulate ? op[nt * 16] : 0.f) + acc[nt][r4];
  }
}

// ------------------------------------------------------------------------------------------------ d / d fmap2
// boxes[(l * ntiles + tile) * 4 ..] = {y0, y1, x0, x1} of every pixel tile's windows on level l (y1 <= y0: empty)
template <int R>
__global__ __launch_bounds__(64) void altcorr_boxes(const AcLevels lv, const float* __restrict__ coords, int planar,
                                                    int* __restrict__ boxes, int B, int H1, int W1) {
  constexpr int gd = 2 * R + 2;
  const int tiles_x = (W1 + AC_TP - 1) / AC_TP, ntiles = B * H1 * tiles_x;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 4);           // 16 lanes per (tile, level)
  const int i = threadIdx.x & 15;
  if (item >= ntiles * lv.n) return;
  const int l = item / ntiles, tile = item - l * ntiles;
  const int b = tile / (H1 * tiles_x), rem = tile - b * H1 * tiles_x, h1 = rem / tiles_x, w0 = (rem % tiles_x) * AC_TP;
  const int H2 = lv.H2[l], W2 = lv.W2[l];
  const PixelWindow pw = pixel_window(coords, planar, b, h1, w0 + i, H1, W1, lv.cscale[l], R);
  const bool m = meets(pw, gd, H2, W2);
  int ymin = m ? pw.cy : INT_MAX, ymax = m ? pw.cy : INT_MIN, xmin = m ? pw.cx : INT_MAX, xmax = m ? pw.cx : INT_MIN;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
    ymin = min(ymin, __shfl_xor(ymin, off, 16)); ymax = max(ymax, __shfl_xor(ymax, off, 16));
    xmin = min(xmin, __shfl_xor(xmin, off, 16)); xmax = max(xmax, __shfl_xor(xmax, off, 16));
  }
  if (i == 0) {
    int* o = boxes + (size_t)item * 4;
    if (ymin == INT_MAX) { o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 0; }
    else { o[0] = max(ymin, 0); o[1] = min(ymax + gd, H2); o[2] = max(xmin, 0); o[3] = min(xmax + gd, W2); }
  }
}

// A workgroup owns fmap2 pixels (hy, xq0 .. xq0 + 15) of one level and (a slice of) the pixel tiles; wave = channel quarter.
// For every pixel tile whose box contains the segment: acc[q, c] += sum_p gs[p, q] fmap1[p, c], gs taken straight from the
// output gradient (four reads per entry).  split == 1: one writer per element (read-add-store: deterministic); split > 1 (the
// coarse levels, where 12-36 segments would otherwise serialise all of the tiles): float atomics for the final add.
template <int R, int CPG>
__global__ __launch_bounds__(256) void altcorr_mfma_bwd2(const float* __restrict__ f1, const AcLevels lv,
                                                         const float* __restrict__ coords, int planar,
                                                         const float* __restrict__ gout, const int* __restrict__ boxes, int B,
                                                         int H1, int W1, float scale, int accumulate) {
  constexpr int rd = 2 * R + 1, gd = rd + 1, C = 4 * CPG, NT = CPG / 16;
  int l = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (k < lv.n && (int)blockIdx.x >= lv.tile0[k]) l = k;
  const int H2 = lv.H2[l], W2 = lv.W2[l], split = lv.split[l];
  const int segs_x = (W2 + 15) >> 4;
  const int local = blockIdx.x - lv.tile0[l], seg = local / split, part = local - seg * split;
  const int b = seg / (H2 * segs_x), rem = seg - b * H2 * segs_x, hy = rem / segs_x, xq0 = (rem % segs_x) * 16;
  const int tiles_x = (W1 + AC_TP - 1) / AC_TP, tiles_b = H1 * tiles_x, ntiles = B * tiles_b;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, pi = lane & 15, g = lane >> 4;
  const size_t plane = (size_t)H1 * W1;
  const int* __restrict__ bx = boxes + ((size_t)l * ntiles + (size_t)b * tiles_b) * 4;
  const float cscale = lv.cscale[l];
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int per = (tiles_b + split - 1) / split, t_lo = part * per, t_hi = min(tiles_b, t_lo + per);
  for (int t0 = t_lo; t0 < t_hi; t0 += 64) {
    const int t = t0 + lane;
    bool hit = false;
    if (t < t_hi) {
      const int4 bxv = *reinterpret_cast<const int4*>(bx + (size_t)t * 4);
      hit = hy >= bxv.x && hy < bxv.y && xq0 < bxv.w && xq0 + 16 > bxv.z;
    }
    unsigned long long m = __ballot(hit);
    while (m) {
      const int tt = t0 + __builtin_ctzll(m);
      m &= m - 1;
      const int h1 = tt / tiles_x, w0 = (tt - h1 * tiles_x) * AC_TP;
      const size_t gbase = (((size_t)b * lv.n + l) * rd * rd) * plane + (size_t)h1 * W1;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        // A[m = fmap2 pixel xq0 + pi][k = pixel w0 + 4 kk + g] = gs[pixel][window point of that fmap2 pixel]
        const int w1 = w0 + 4 * kk + g;
        const PixelWindow pw = pixel_window(coords, planar, b, h1, w1, H1, W1, cscale, R);
        const int iy = hy - pw.cy, ix = xq0 + pi - pw.cx;
        float av = 0.f;
        if ((unsigned)iy < (unsigned)gd && (unsigned)ix < (unsigned)gd && xq0 + pi < W2) {
          const float* gp = gout + gbase + w1;
          if (iy > 0 && ix > 0)   av += gp[plane * ((iy - 1) + rd * (ix - 1))] * pw.dy * pw.dx;
          if (iy > 0 && ix < rd)  av += gp[plane * ((iy - 1) + rd * ix)] * pw.dy * (1 - pw.dx);
          if (iy < rd && ix > 0)  av += gp[plane * (iy + rd * (ix - 1))] * (1 - pw.dy) * pw.dx;
          if (iy < rd && ix < rd) av += gp[plane * (iy + rd * ix)] * (1 - pw.dy) * (1 - pw.dx);
          av *= scale;
        }
        // B[k = pixel][n = channel]
        const float* bp = f1 + (((size_t)b * H1 + h1) * W1 + min(w1, W1 - 1)) * C + wave * CPG + pi;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[nt * 16], acc[nt], 0, 0, 0);
      }
    }
  }
  float* g2 = lv.g2[l];
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int q = xq0 + 4 * g + r4;
    if (q >= W2) continue;
    float* op = g2 + (((size_t)b * H2 + hy) * W2 + q) * C + wave * CPG + pi;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (split > 1) atomicAdd(op + nt * 16, acc[nt][r4]);          // (the buffer was zeroed or holds the running sum)
      else op[nt * 16] = (accumulate ? op[nt * 16] : 0.f) + acc[nt][r4];
    }
  }
}

template <int R, int CPG>
int launch_fwd(const float* f1, const AcLevels& lv, const float* coords, int planar, float* out, int B, int H1, int W1, float scale,
               hipStream_t st) {
  const int tiles = B * H1 * ((W1 + AC_TP - 1) / AC_TP);
  altcorr_mfma_fwd<R, CPG><<<dim3(tiles, lv.n), 256, 0, st>>>(f1, lv, coords, planar, out, B, H1, W1, scale);
  return ufr::launched("altcorr_mfma_fwd");
}

template <int R, int CPG>
int launch_bwd(const float* f1, AcLevels lv, const float* coords, int planar, const float* gout, float* g1, int* boxes, int B, int H1,
               int W1, float scale, int accumulate, hipStream_t st) {
  const int tiles_b = H1 * ((W1 + AC_TP - 1) / AC_TP), tiles = B * tiles_b;
  altcorr_mfma_bwd1<R, CPG><<<tiles, 256, 0, st>>>(lv, coords, planar, gout, g1, B, H1, W1, scale, accumulate);
  int rc = ufr::launched("altcorr_mfma_bwd1");
  if (rc != UFR_OK) return rc;
  altcorr_boxes<R><<<(tiles * lv.n + 3) / 4, 64, 0, st>>>(lv, coords, planar, boxes, B, H1, W1);
  rc = ufr::launched("altcorr_boxes");
  if (rc != UFR_OK) return rc;
  int total = 0;
  for (int l = 0; l < lv.n; ++l) {
    const int segs = B * lv.H2[l] * ((lv.W2[l] + 15) / 16);
    // a segment of level l is reached by ~4^l times as many pixel tiles as one of level 0: spread them over workgroups
    int split = 1;
    while (segs * split < 256 && split * 2 <= tiles_b / 8 && split < 64) split *= 2;
    lv.tile0[l] = total;
    lv.split[l] = split;
    total += segs * split;
    if (split > 1 && !accumulate) {               // atomics add onto the buffer: start from zero
      hipError_t e = hipMemsetAsync(lv.g2[l], 0, sizeof(float) * (size_t)B * lv.H2[l] * lv.W2[l] * 4 * CPG, st);
      if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "altcorr backward: memset: %s", hipGetErrorString(e));
    }
  }
  lv.tile0[lv.n] = total;
  for (int l = lv.n + 1; l < 5; ++l) lv.tile0[l] = total;
  altcorr_mfma_bwd2<R, CPG><<<total, 256, 0, st>>>(f1, lv, coords, planar, gout, boxes, B, H1, W1, scale, accumulate);
  return ufr::launched("altcorr_mfma_bwd2");
}

bool mfma_form_serves(int C, int radius) { return (C == 256 || C == 128) && (radius == 4 || radius == 3); }

}  // namespace

// Entry points shared with raft_corr.hip (the drop-in alt_cuda_corr calls route here when the matrix-core form serves them)
int ufr_altcorr_mfma_forward(const float* f1, const ufr_altcorr_levels* lv_in, const float* coords, int planar, float* out, int B,
                             int H1, int W1, int C, int radius, float scale, hipStream_t st) {
  AcLevels lv{};
  lv.n = lv_in->num_levels;
  for (int l = 0; l < lv.n; ++l) {
    lv.f2[l] = lv_in->fmap2[l]; lv.H2[l] = lv_in->H2[l]; lv.W2[l] = lv_in->W2[l]; lv.cscale[l] = lv_in->coord_scale[l];
  }
  if (C == 256 && radius == 4) return launch_fwd<4, 64>(f1, lv, coords, planar, out, B, H1, W1, scale, st);
  if (C == 128 && radius == 4) return launch_fwd<4, 32>(f1, lv, coords, planar, out, B, H1, W1, scale, st);
  if (C == 256 && radius == 3) return launch_fwd<3, 64>(f1, lv, coords, planar, out, B, H1, W1, scale, st);
  return launch_fwd<3, 32>(f1, lv, coords, planar, out, B, H1, W1, scale, st);
}

int ufr_altcorr_mfma_backward(const float* f1, const ufr_altcorr_levels* lv_in, const float* coords, int planar, const float* gout,
                              float* g1, int* boxes, int B, int H1, int W1, int C, int radius, float scale, int accumulate,
                              hipStream_t st) {
  AcLevels lv{};
  lv.n = lv_in->num_levels;
  for (int l = 0; l < lv.n; ++l) {
    lv.f2[l] = lv_in->fmap2[l]; lv.g2[l] = lv_in->fmap2_grad[l]; lv.H2[l] = lv_in->H2[l]; lv.W2[l] = lv_in->W2[l];
    lv.cscale[l] = lv_in->coord_scale[l];
  }
  if (C == 256 && radius == 4) return launch_bwd<4, 64>(f1, lv, coords, planar, gout, g1, boxes, B, H1, W1, scale, accumulate, st);
  if (C == 128 && radius == 4) return launch_bwd<4, 32>(f1, lv, coords, planar, gout, g1, boxes, B, H1, W1, scale, accumulate, st);
  if (C == 256 && radius == 3) return launch_bwd<3, 64>(f1, lv, coords, planar, gout, g1, boxes, B, H1, W1, scale, accumulate, st);
  return launch_bwd<3, 32>(f1, lv, coords, planar, gout, g1, boxes, B, H1, W1, scale, accumulate, st);
}

bool ufr_altcorr_mfma_serves(int C, int radius) { return mfma_form_serves(C, radius); }

extern "C" int ufr_altcorr_pyramid_forward(const float* fmap1, const ufr_altcorr_levels* levels, const float* coords, float* out,
                                           int B, int H1, int W1, int C, int radius, float scale, ufr_stream_t stream) {
  UFR_REQUIRE(fmap1 && levels && coords && out, "altcorr pyramid forward: null pointer");
  UFR_REQUIRE(B > 0 && H1 > 0 && W1 > 0 && levels->num_levels >= 1 && levels->num_levels <= 4, "altcorr pyramid forward: bad shape");
  UFR_REQUIRE(mfma_form_serves(C, radius), "altcorr pyramid forward: C must be 128 or 256 and the radius 3 or 4 (got %d, %d)", C, radius);
  for (int l = 0; l < levels->num_levels; ++l)
    UFR_REQUIRE(levels->fmap2[l] && levels->H2[l] > 0 && levels->W2[l] > 0, "altcorr pyramid forward: bad level %d", l);
  return ufr_altcorr_mfma_forward(fmap1, levels, coords, 1, out, B, H1, W1, C, radius, scale, ufr::as_stream(stream));
}

extern "C" int ufr_altcorr_pyramid_backward(const float* fmap1, const ufr_altcorr_levels* levels, const float* coords,
                                            const float* grad_out, float* fmap1_grad, int* workspace, int B, int H1, int W1, int C,
                                            int radius, float scale, int accumulate, ufr_stream_t stream) {
  UFR_REQUIRE(fmap1 && levels && coords && grad_out && fmap1_grad && workspace, "altcorr pyramid backward: null pointer");
  UFR_REQUIRE(B > 0 && H1 > 0 && W1 > 0 && levels->num_levels >= 1 && levels->num_levels <= 4, "altcorr pyramid backward: bad shape");
  UFR_REQUIRE(mfma_form_serves(C, radius), "altcorr pyramid backward: C must be 128 or 256 and the radius 3 or 4 (got %d, %d)", C, radius);
  for (int l = 0; l < levels->num_levels; ++l)
    UFR_REQUIRE(levels->fmap2[l] && levels->fmap2_grad[l] && levels->H2[l] > 0 && levels->W2[l] > 0, "altcorr pyramid backward: bad level %d", l);
  return ufr_altcorr_mfma_backward(fmap1, levels, coords, 1, grad_out, fmap1_grad, workspace, B, H1, W1, C, radius, scale, accumulate,
                                   ufr::as_stream(stream));
}

extern "C" long ufr_altcorr_pyramid_workspace_ints(int B, int H1, int W1, int num_levels) {
  return (long)B * H1 * ((W1 + AC_TP - 1) / AC_TP) * 4 * (num_levels > 0 ? num_levels : 1);
}
