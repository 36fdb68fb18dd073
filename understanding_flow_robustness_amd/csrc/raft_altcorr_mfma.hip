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
#include <cstdlib>
#include <climits>
#include <cstdint>

#include "ufr_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int AC_TP = 16;                    // pixels per tile (one MFMA M block)
// (The two adjoint kernels run on the update engine's SIDE stream next to the GRU adjoint of the previous iteration and fill six of a
// SIMD's eight wave slots; the main chain's small streaming kernels then wait for slots -- gates_bwd 9 us alone, up to 90 us beside
// altcorr_mfma_bwd2.  Capping these kernels at 3 / 2 waves per SIMD measured SLOWER, 15.49 -> 15.58 / 16.05 ms per iteration at one pair,
// 85.6 -> 89.1 / 93.2 at eight, one call, gpurun r6_occ: the side chain is the backward's critical path, see profiles/r6_side_occupancy_rejected.txt.)

struct AcLevels {                            // by value in the kernel arguments
  int n;
  const float* f2[4];
  float* g2[4];
  int H2[4], W2[4];
  float cscale[4];                           // coords are multiplied by this (1 / 2^l, corr.py:126)
  int tile0[5];                              // d/d fmap2: first workgroup of each level's fmap2 tiles
  int split[4];                              // d/d fmap2: workgroups per fmap2 tile (each scans a slice of the pixel tiles)
  float* p2[4];                              // d/d fmap2 of a split level: its parts' sums [split][B, H2, W2, C] (workspace)
};

struct PixelWindow {
  int cx, cy;                                // first fmap2 column / row of the window (corner - r)
  float dx, dy;
};

// The hardware deals consecutive workgroups (x fastest) round-robin to the 8 XCDs, each with its own 4 MB L2; a level's fmap2
// (7.9 MB at 48 x 160 x 256) does not fit one L2, and with tiles in launch order every XCD walks every row of it: PMC had
// 216 - 235 MB fetched per lookup and kernel for 28 MB of operands (profiles/r3_c3alt_step_traffic.json).  Pixel ROWS are
// dealt to the XCDs instead: launch index i = 8 j + k becomes the j-th tile of XCD k, which owns the pixel rows r with
// r % 8 == k (all tiles of a row share their fmap2 rows).  One contiguous band of rows per XCD moved fewer bytes still but
// ran slower (image borders and flow discontinuities leave whole XCDs idle: d/d fmap2 0.123 -> 0.150 ms); interleaved
// rows keep the balance.  The grid's x extent is (rows padded to a multiple of 8) x tiles per row; -1 = padding.
__device__ __forceinline__ int xcd_row_tile(int i, int tiles_x, int ntiles) {
  const int k = i & 7, j = i >> 3;
  const int t = ((j / tiles_x) * 8 + k) * tiles_x + j % tiles_x;
  return t < ntiles ? t : -1;
}

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
  const int tile = xcd_row_tile(blockIdx.x, tiles_x, B * H1 * tiles_x);
  if (tile < 0) return;
  const int b = tile / (H1 * tiles_x), rem = tile - b * H1 * tiles_x, h1 = rem / tiles_x, w0 = (rem % tiles_x) * AC_TP;
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
    // A operand: this lane's pixel.  The MFMA's k index is the lane group g, and step 4 i + e takes channel 16 i + 4 g + e of
    // BOTH operands: one 16-byte load per lane covers four steps and the four lane groups of a pixel read 64 contiguous
    // bytes (a [g * CPG, (g + 1) * CPG) split made every load instruction touch 64 different cache lines)
    float a[CPG];
    {
      const float* ap = f1 + (((size_t)b * H1 + h1) * W1 + min(w0 + pi, W1 - 1)) * C + 4 * g;
#pragma unroll
      for (int j = 0; j < CPG; j += 4) {
        const float4 v = *reinterpret_cast<const float4*>(ap + 4 * j);
        a[j] = v.x; a[j + 1] = v.y; a[j + 2] = v.z; a[j + 3] = v.w;
      }
    }
    int cyr[4], cxr[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) { cyr[r4] = cys[4 * g + r4]; cxr[r4] = cxs[4 * g + r4]; }
    for (int blk = wave; blk < nblk; blk += 4) {
      const int hy = y0 + blk / ncb, xb = x0 + (blk % ncb) * 16, q = xb + pi;
      const float* bp = f2 + (((size_t)b * H2 + hy) * W2 + min(q, W2 - 1)) * C + 4 * g;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      float4 bv[CPG / 4];
#pragma unroll
      for (int j = 0; j < CPG / 4; ++j) bv[j] = *reinterpret_cast<const float4*>(bp + 16 * j);
#pragma unroll
      for (int j = 0; j < CPG; j += 4) {
        const float4 v = bv[j / 4];
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

// ------------------------------------------------------------------------------------------------ backward, pre-pass
// One workgroup (256 threads) per (pixel tile, level): every pixel's window origin, gs = the adjoint of the bilinear blend
// (correlation_kernel.cu:196-214) times `scale`, and the tile's box.  Both adjoints read these instead of recomputing them
// per (tile, segment) pair.   gs_all [L][npix][npt] | win [L][npix][2] | boxes [L][ntiles][4]: {y0, y1, x0, x1}, empty = 0s
template <int R>
__global__ __launch_bounds__(256) void altcorr_prepass(const AcLevels lv, const float* __restrict__ coords, int planar,
                                                      const float* __restrict__ gout, float* __restrict__ gs_all,
                                                      int* __restrict__ win, int* __restrict__ boxes, int B, int H1, int W1,
                                                      float scale, int gout_cm) {
  constexpr int rd = 2 * R + 1, gd = rd + 1, npt = gd * gd;
  __shared__ float dxs[AC_TP], dys[AC_TP];
  const int tiles_x = (W1 + AC_TP - 1) / AC_TP, ntiles = B * H1 * tiles_x;
  const int l = blockIdx.y, tile = xcd_row_tile(blockIdx.x, tiles_x, ntiles);
  if (tile < 0) return;
  const int b = tile / (H1 * tiles_x), rem = tile - b * H1 * tiles_x, h1 = rem / tiles_x, w0 = (rem % tiles_x) * AC_TP;
  const int H2 = lv.H2[l], W2 = lv.W2[l], tid = threadIdx.x;
  const size_t plane = (size_t)H1 * W1, npix = (size_t)B * plane;
  {
    const int i = tid & 15;
    const PixelWindow pw = pixel_window(coords, planar, b, h1, w0 + i, H1, W1, lv.cscale[l], R);
    const bool m = meets(pw, gd, H2, W2);
    int ymin = m ? pw.cy : INT_MAX, ymax = m ? pw.cy : INT_MIN, xmin = m ? pw.cx : INT_MAX, xmax = m ? pw.cx : INT_MIN;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      ymin = min(ymin, __shfl_xor(ymin, off, 16)); ymax = max(ymax, __shfl_xor(ymax, off, 16));
      xmin = min(xmin, __shfl_xor(xmin, off, 16)); xmax = max(xmax, __shfl_xor(xmax, off, 16));
    }
    if (tid < AC_TP) {
      dxs[i] = pw.dx; dys[i] = pw.dy;
      if (w0 + i < W1) {
        int* wp = win + ((size_t)l * npix + (size_t)b * plane + (size_t)h1 * W1 + w0 + i) * 2;
        wp[0] = pw.cx; wp[1] = pw.cy;
      }
    }
    if (tid == 0) {
      int* o = boxes + ((size_t)l * ntiles + tile) * 4;
      if (ymin == INT_MAX) { o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 0; }
      else { o[0] = max(ymin, 0); o[1] = min(ymax + gd, H2); o[2] = max(xmin, 0); o[3] = min(xmax + gd, W2); }
    }
  }
  __syncthreads();
  // the cost volume's gradient: NCHW [B, L * rd * rd, H1, W1], or (gout_cm: the update engine's gradient sum as it lies, no
  // conversion pass in between) chunk-major float32 [chunks][B * H1 * W1][32] with channel o at chunk o / 32, lane o % 32
  const size_t gbase = (((size_t)b * lv.n + l) * rd * rd) * plane + (size_t)h1 * W1;
  const size_t prow = (size_t)b * plane + (size_t)h1 * W1;
  for (int t = tid; t < AC_TP * npt; t += blockDim.x) {
    // chunk-major gradient: a pixel's 81 channels of a level lie in three 128-byte lines, and its 100 results are 400 contiguous bytes --
    // consecutive lanes take consecutive window points of ONE pixel; NCHW: consecutive lanes take consecutive pixels of one channel
    const int i = gout_cm ? t / npt : t & 15, pt = gout_cm ? t - i * npt : t >> 4, iy = pt / gd, ix = pt - iy * gd;
    if (w0 + i >= W1) continue;
    const float* gp = gout + gbase + w0 + i;
    auto G = [&](int ol) -> float {
      if (!gout_cm) return gp[plane * ol];
      const int o = l * rd * rd + ol;
      return gout[((size_t)(o >> 5) * npix + prow + w0 + i) * 32 + (o & 31)];
    };
    const float dx = dxs[i], dy = dys[i];
    float g = 0.f;
    if (iy > 0 && ix > 0)   g += G((iy - 1) + rd * (ix - 1)) * dy * dx;
    if (iy > 0 && ix < rd)  g += G((iy - 1) + rd * ix) * dy * (1 - dx);
    if (iy < rd && ix > 0)  g += G(iy + rd * (ix - 1)) * (1 - dy) * dx;
    if (iy < rd && ix < rd) g += G(iy + rd * ix) * (1 - dy) * (1 - dx);
    gs_all[((size_t)l * npix + (size_t)b * plane + (size_t)h1 * W1 + w0 + i) * npt + pt] = g * scale;
  }
}

// ------------------------------------------------------------------------------------------------ d / d fmap1
// part[l][p, c] = sum over fmap2_l pixels q of gs_l[p, q] fmap2_l[q, c]: one workgroup per (pixel tile, level), wave w owns
// channels [w C/4, (w+1) C/4) -- no reduction between waves; the levels are added in a fixed order by altcorr_sum_levels.
// Lane (pi, g) of wave w: A[m = pixel pi][k = fmap2 pixel 4 kk + g]; B[k][n]: channel w CPG + pi NT + nt, so that a lane's NT
// channels are one 16-byte load and its NT results one 16-byte store.
template <int R, int CPG>
__global__ __launch_bounds__(256) void altcorr_mfma_bwd1(const AcLevels lv, const float* __restrict__ gs_all,
                                                         const int* __restrict__ win, const int* __restrict__ boxes,
                                                         float* __restrict__ part, int B, int H1, int W1) {
  constexpr int gd = 2 * R + 2, npt = gd * gd, C = 4 * CPG, NT = CPG / 16;
  static_assert(NT == 4 || NT == 2, "channel quarters of 64 or 32");
  const int tiles_x = (W1 + AC_TP - 1) / AC_TP, ntiles = B * H1 * tiles_x;
  const int l = blockIdx.y, tile = xcd_row_tile(blockIdx.x, tiles_x, ntiles);
  if (tile < 0) return;
  const int b = tile / (H1 * tiles_x), rem = tile - b * H1 * tiles_x, h1 = rem / tiles_x, w0 = (rem % tiles_x) * AC_TP;
  const int H2 = lv.H2[l], W2 = lv.W2[l];
  const float* __restrict__ f2 = lv.f2[l];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, pi = lane & 15, g = lane >> 4;
  const size_t plane = (size_t)H1 * W1, npix = (size_t)B * plane;
  const int4 bx = *reinterpret_cast<const int4*>(boxes + ((size_t)l * ntiles + tile) * 4);
  const int y0 = bx.x, y1 = bx.y, x0 = bx.z, x1 = bx.w;
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int w1 = min(w0 + pi, W1 - 1);
  const size_t pix = (size_t)l * npix + (size_t)b * plane + (size_t)h1 * W1 + w1;
  const bool live = w0 + pi < W1;
  const int cxi = win[pix * 2], cyi = win[pix * 2 + 1];
  const float* __restrict__ gsp = gs_all + pix * npt;
  if (y1 > y0 && x1 > x0) {
    const int ncb = (x1 - x0 + 15) >> 4, nblk = (y1 - y0) * ncb;
    for (int blk = 0; blk < nblk; ++blk) {
      const int hy = y0 + blk / ncb, xb = x0 + (blk % ncb) * 16;
      const int iy = hy - cyi;
      float av[4];
      float bvv[4][NT];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int q = xb + 4 * kk + g, ix = q - cxi;
        av[kk] = (live && (unsigned)iy < (unsigned)gd && (unsigned)ix < (unsigned)gd && q < W2) ? gsp[iy * gd + ix] : 0.f;
        const float* bp = f2 + (((size_t)b * H2 + hy) * W2 + min(q, W2 - 1)) * C + wave * CPG + pi * NT;
        if (NT == 4) {
          const float4 v = *reinterpret_cast<const float4*>(bp);
          bvv[kk][0] = v.x; bvv[kk][1] = v.y; bvv[kk][2 % NT] = v.z; bvv[kk][3 % NT] = v.w;
        } else {
          const float2 v = *reinterpret_cast<const float2*>(bp);
          bvv[kk][0] = v.x; bvv[kk][1] = v.y;
        }
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], bvv[kk][nt], acc[nt], 0, 0, 0);
    }
  }
  // D[row = pixel 4 g + r][col pi] of n-tile nt = channel wave * CPG + pi * NT + nt
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int wq = w0 + 4 * g + r4;
    if (wq >= W1) continue;
    float* op = part + (((size_t)l * B + b) * plane + (size_t)h1 * W1 + wq) * C + wave * CPG + pi * NT;
    if (NT == 4) *reinterpret_cast<float4*>(op) = make_float4(acc[0][r4], acc[1][r4], acc[2 % NT][r4], acc[3 % NT][r4]);
    else *reinterpret_cast<float2*>(op) = make_float2(acc[0][r4], acc[1][r4]);
  }
}

// g1 (+)= part[0] + part[1] + ... in ascending level order
__global__ void altcorr_sum_levels(const float* __restrict__ part, float* __restrict__ g1, long n4, int levels, int accumulate) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 s = accumulate ? reinterpret_cast<const float4*>(g1)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < levels; ++l) {
      const float4 v = reinterpret_cast<const float4*>(part)[(long)l * n4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(g1)[i] = s;
  }
}

// ------------------------------------------------------------------------------------------------ d / d fmap2
// A workgroup owns fmap2 pixels (hy, xq0 .. xq0 + 15) of one level and (a slice of) the pixel tiles; wave = channel quarter.
// For every pixel tile whose box contains the segment: acc[q, c] += sum_p gs[p, q] fmap1[p, c].  split == 1: one writer per
// element (read-add-store: deterministic); split > 1 (the coarse levels, where 12-36 segments would otherwise serialise all
// of the tiles): every part stores ITS sum into its own slab of the workspace (one writer per element, no atomics -- float atomics
// retire at ~28 G lanes/s on this chip and made finer splits SLOWER, gpurun r4_call70), `altcorr_sum_parts` adds the slabs in
// ascending order afterwards: the gradient is bit-reproducible and the coarse levels can be cut as fine as their balance asks.
template <int R, int CPG>
__global__ __launch_bounds__(256) void altcorr_mfma_bwd2(const float* __restrict__ f1, const AcLevels lv,
                                                         const float* __restrict__ gs_all, const int* __restrict__ win,
                                                         const int* __restrict__ boxes, int B, int H1, int W1, int accumulate) {
  constexpr int gd = 2 * R + 2, npt = gd * gd, C = 4 * CPG, NT = CPG / 16;
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
  const size_t plane = (size_t)H1 * W1, npix = (size_t)B * plane;
  const int* __restrict__ bx = boxes + ((size_t)l * ntiles + (size_t)b * tiles_b) * 4;
  const size_t lbase = (size_t)l * npix + (size_t)b * plane;
  const bool qlive = xq0 + pi < W2;
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
      float av[4];
      float bvv[4][NT];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        // A[m = fmap2 pixel xq0 + pi][k = pixel w0 + 4 kk + g] = gs[pixel][window point of that fmap2 pixel]
        const int w1 = w0 + 4 * kk + g, w1c = min(w1, W1 - 1);
        const size_t pix = lbase + (size_t)h1 * W1 + w1c;
        const int2 wv = *reinterpret_cast<const int2*>(win + pix * 2);
        const int iy = hy - wv.y, ix = xq0 + pi - wv.x;
        av[kk] = (w1 < W1 && qlive && (unsigned)iy < (unsigned)gd && (unsigned)ix < (unsigned)gd) ? gs_all[pix * npt + iy * gd + ix] : 0.f;
        // B[k = pixel][n]: channel wave * CPG + pi * NT + nt
        const float* bp = f1 + (((size_t)b * H1 + h1) * W1 + w1c) * C + wave * CPG + pi * NT;
        if (NT == 4) {
          const float4 v = *reinterpret_cast<const float4*>(bp);
          bvv[kk][0] = v.x; bvv[kk][1] = v.y; bvv[kk][2 % NT] = v.z; bvv[kk][3 % NT] = v.w;
        } else {
          const float2 v = *reinterpret_cast<const float2*>(bp);
          bvv[kk][0] = v.x; bvv[kk][1] = v.y;
        }
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], bvv[kk][nt], acc[nt], 0, 0, 0);
    }
  }
  float* g2 = split > 1 ? lv.p2[l] + (size_t)part * B * H2 * W2 * C : lv.g2[l];
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int q = xq0 + 4 * g + r4;
    if (q >= W2) continue;
    float* op = g2 + (((size_t)b * H2 + hy) * W2 + q) * C + wave * CPG + pi * NT;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (split > 1) op[nt] = acc[nt][r4];                     // this part's slab: summed by altcorr_sum_parts
      else op[nt] = (accumulate ? op[nt] : 0.f) + acc[nt][r4];
    }
  }
}

// g2 (+)= parts[0] + parts[1] + ... in ascending order, for every level whose segments were split
__global__ void altcorr_sum_parts(const AcLevels lv, int B, int C4, int accumulate) {
  for (int l = 0; l < lv.n; ++l) {
    const int S = lv.split[l];
    if (S <= 1) continue;
    const long n4 = (long)B * lv.H2[l] * lv.W2[l] * C4;
    const float4* parts = reinterpret_cast<const float4*>(lv.p2[l]);
    float4* g = reinterpret_cast<float4*>(lv.g2[l]);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
      float4 v = accumulate ? g[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      for (int p0 = 0; p0 < S; p0 += 8) {                      // eight slabs in flight, added in ascending order
        float4 a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = p0 + k < S ? parts[(long)(p0 + k) * n4 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 8; ++k) { v.x += a[k].x; v.y += a[k].y; v.z += a[k].z; v.w += a[k].w; }
      }
      g[i] = v;
    }
  }
}

template <int R, int CPG>
int launch_fwd(const float* f1, const AcLevels& lv, const float* coords, int planar, float* out, int B, int H1, int W1, float scale,
               hipStream_t st) {
  const int tiles = (B * H1 + 7) / 8 * 8 * ((W1 + AC_TP - 1) / AC_TP);          // rows padded to a multiple of 8: see xcd_row_tile
  altcorr_mfma_fwd<R, CPG><<<dim3(tiles, lv.n), 256, 0, st>>>(f1, lv, coords, planar, out, B, H1, W1, scale);
  return ufr::launched("altcorr_mfma_fwd");
}

// workspace layout (bytes): gs_all f32 [L][npix][npt] | part f32 [L][npix][C] | win i32 [L][npix][2] | boxes i32 [L][ntiles][4] |
// per split level: the parts' slabs f32 [split][B, H2, W2, C]
struct AcWorkspace { float* gs_all; float* part; int* win; int* boxes; };

// d/d fmap2: workgroups per 16-pixel segment of level l.  First fill the chip (a segment of level l is reached by ~4^l times as
// many pixel tiles as one of level 0); then, since the kernel lasts as long as its busiest workgroup -- a level-l segment collects
// ~4^l x 25 tile hits of ~1 us each, a dependent window -> gradient gather per hit -- the coarse levels are cut finer still,
// up to 2 x 4^l parts while the launch stays below 2048 workgroups per level.
int want_split(int l) { return l > 0 ? min(64, 2 << (2 * l)) : 1; }
int split_for(int l, int segs, int tiles_b) {
  int split = 1;
  while (segs * split < 256 && split * 2 <= tiles_b / 8 && split < 64) split *= 2;
  while (split < want_split(l) && segs * split < 2048 && split * 2 <= tiles_b) split *= 2;
  return split;
}

// upper bound of the parts' slabs of level l (pixels): H2 <= ceil(H1 / 2^l) for RAFT's average-pooled pyramid
long parts_pixels_bound(int l, int B, int H1, int W1) {
  const long H2 = (H1 + (1 << l) - 1) >> l, W2 = (W1 + (1 << l) - 1) >> l;
  const long by_want = want_split(l) > 1 ? min((long)want_split(l) * B * H2 * W2, 2L * 2048 * 16) : 0;   // split * segs < 2 * 2048 under the second rule
  return max(512L * 16, by_want) + 64;                                              // split * segs < 512 under the first
}

long workspace_bytes(int B, int H1, int W1, int C, int radius, int levels) {
  const long npix = (long)B * H1 * W1, ntiles = (long)B * H1 * ((W1 + AC_TP - 1) / AC_TP), npt = (2L * radius + 2) * (2 * radius + 2);
  long parts = 0;
  for (int l = 0; l < levels; ++l) parts += parts_pixels_bound(l, B, H1, W1) * C;
  return 4 * (levels * npix * npt + levels * npix * C + levels * npix * 2 + levels * ntiles * 4 + 256 + parts);
}

template <int R, int CPG>
int launch_bwd(const float* f1, AcLevels lv, const float* coords, int planar, const float* gout, float* g1, void* workspace, int B,
               int H1, int W1, float scale, int accumulate, hipStream_t st, int gout_cm = 0) {
  constexpr int C = 4 * CPG, npt = (2 * R + 2) * (2 * R + 2);
  const int tiles_b = H1 * ((W1 + AC_TP - 1) / AC_TP);
  const long npix = (long)B * H1 * W1;
  AcWorkspace ws;
  ws.gs_all = static_cast<float*>(workspace);
  ws.part = ws.gs_all + (long)lv.n * npix * npt;
  ws.win = reinterpret_cast<int*>(ws.part + (long)lv.n * npix * C);
  ws.boxes = ws.win + (long)lv.n * npix * 2;
  ws.boxes += (4 - ((reinterpret_cast<uintptr_t>(ws.boxes) / 4) & 3)) & 3;            // 16-byte aligned rows
  const int tiles_pad = (B * H1 + 7) / 8 * 8 * (tiles_b / H1);                 // rows padded to a multiple of 8: see xcd_row_tile
  // 256 threads per (tile, level): 1,600 window points of 16 pixels, 4 gathers each -- one wave walked them as 25 dependent round trips
  // (74 us per lookup in the C3 step; 128 / 256 / 512 threads: 51 / 39 / 49 us, c3alt 15.40 / 15.51 -> 15.17 / 15.28 ms, gpurun r5_pp3 / r5_pp4)
  altcorr_prepass<R><<<dim3(tiles_pad, lv.n), 256, 0, st>>>(lv, coords, planar, gout, ws.gs_all, ws.win, ws.boxes, B, H1, W1, scale, gout_cm);
  int rc = ufr::launched("altcorr_prepass");
  if (rc != UFR_OK) return rc;
  altcorr_mfma_bwd1<R, CPG><<<dim3(tiles_pad, lv.n), 256, 0, st>>>(lv, ws.gs_all, ws.win, ws.boxes, ws.part, B, H1, W1);
  rc = ufr::launched("altcorr_mfma_bwd1");
  if (rc != UFR_OK) return rc;
  const long n4 = npix * C / 4;
  altcorr_sum_levels<<<ufr::stream_grid(n4, 256), 256, 0, st>>>(ws.part, g1, n4, lv.n, accumulate);
  rc = ufr::launched("altcorr_sum_levels");
  if (rc != UFR_OK) return rc;
  int total = 0;
  float* parts = reinterpret_cast<float*>(ws.boxes + (long)lv.n * B * tiles_b * 4 + 16);     // behind the boxes [L][B * tiles_b][4]
  parts += (4 - ((reinterpret_cast<uintptr_t>(parts) / 4) & 3)) & 3;                  // 16-byte aligned slabs
  long max_n4 = 0;
  bool any_split = false;
  for (int l = 0; l < lv.n; ++l) {
    const int segs = B * lv.H2[l] * ((lv.W2[l] + 15) / 16);
    const int split = split_for(l, segs, tiles_b);
    lv.tile0[l] = total;
    lv.split[l] = split;
    lv.p2[l] = nullptr;
    total += segs * split;
    if (split > 1) {
      const long px = (long)B * lv.H2[l] * lv.W2[l];
      UFR_REQUIRE((long)split * px <= parts_pixels_bound(l, B, H1, W1), "altcorr backward: level %d (%d x %d) is larger than a pooled level of a %d x %d map",
                  l, lv.H2[l], lv.W2[l], H1, W1);
      lv.p2[l] = parts;
      parts += (long)split * px * C;
      max_n4 = max(max_n4, px * (C / 4));
      any_split = true;
    }
  }
  lv.tile0[lv.n] = total;
  for (int l = lv.n + 1; l < 5; ++l) lv.tile0[l] = total;
  altcorr_mfma_bwd2<R, CPG><<<total, 256, 0, st>>>(f1, lv, ws.gs_all, ws.win, ws.boxes, B, H1, W1, accumulate);
  rc = ufr::launched("altcorr_mfma_bwd2");
  if (rc != UFR_OK || !any_split) return rc;
  altcorr_sum_parts<<<ufr::stream_grid(max_n4, 256), 256, 0, st>>>(lv, B, C / 4, accumulate);
  return ufr::launched("altcorr_sum_parts");
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
                              float* g1, void* workspace, int B, int H1, int W1, int C, int radius, float scale, int accumulate,
                              hipStream_t st, int gout_cm) {
  AcLevels lv{};
  lv.n = lv_in->num_levels;
  for (int l = 0; l < lv.n; ++l) {
    lv.f2[l] = lv_in->fmap2[l]; lv.g2[l] = lv_in->fmap2_grad[l]; lv.H2[l] = lv_in->H2[l]; lv.W2[l] = lv_in->W2[l];
    lv.cscale[l] = lv_in->coord_scale[l];
  }
  if (C == 256 && radius == 4) return launch_bwd<4, 64>(f1, lv, coords, planar, gout, g1, workspace, B, H1, W1, scale, accumulate, st, gout_cm);
  if (C == 128 && radius == 4) return launch_bwd<4, 32>(f1, lv, coords, planar, gout, g1, workspace, B, H1, W1, scale, accumulate, st, gout_cm);
  if (C == 256 && radius == 3) return launch_bwd<3, 64>(f1, lv, coords, planar, gout, g1, workspace, B, H1, W1, scale, accumulate, st, gout_cm);
  return launch_bwd<3, 32>(f1, lv, coords, planar, gout, g1, workspace, B, H1, W1, scale, accumulate, st, gout_cm);
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
                                            const float* grad_out, float* fmap1_grad, void* workspace, int B, int H1, int W1, int C,
                                            int radius, float scale, int accumulate, ufr_stream_t stream) {
  UFR_REQUIRE(fmap1 && levels && coords && grad_out && fmap1_grad && workspace, "altcorr pyramid backward: null pointer");
  UFR_REQUIRE(B > 0 && H1 > 0 && W1 > 0 && levels->num_levels >= 1 && levels->num_levels <= 4, "altcorr pyramid backward: bad shape");
  UFR_REQUIRE(mfma_form_serves(C, radius), "altcorr pyramid backward: C must be 128 or 256 and the radius 3 or 4 (got %d, %d)", C, radius);
  for (int l = 0; l < levels->num_levels; ++l)
    UFR_REQUIRE(levels->fmap2[l] && levels->fmap2_grad[l] && levels->H2[l] > 0 && levels->W2[l] > 0, "altcorr pyramid backward: bad level %d", l);
  return ufr_altcorr_mfma_backward(fmap1, levels, coords, 1, grad_out, fmap1_grad, workspace, B, H1, W1, C, radius, scale, accumulate,
                                   ufr::as_stream(stream), 0);
}

extern "C" int ufr_altcorr_pyramid_backward_cm(const float* fmap1, const ufr_altcorr_levels* levels, const float* coords,
                                               const float* grad_out_cm, float* fmap1_grad, void* workspace, int B, int H1, int W1, int C,
                                               int radius, float scale, int accumulate, ufr_stream_t stream) {
  UFR_REQUIRE(fmap1 && levels && coords && grad_out_cm && fmap1_grad && workspace, "altcorr pyramid backward (chunk-major): null pointer");
  UFR_REQUIRE(B > 0 && H1 > 0 && W1 > 0 && levels->num_levels >= 1 && levels->num_levels <= 4, "altcorr pyramid backward (chunk-major): bad shape");
  UFR_REQUIRE(mfma_form_serves(C, radius), "altcorr pyramid backward (chunk-major): C must be 128 or 256 and the radius 3 or 4 (got %d, %d)", C, radius);
  for (int l = 0; l < levels->num_levels; ++l)
    UFR_REQUIRE(levels->fmap2[l] && levels->fmap2_grad[l] && levels->H2[l] > 0 && levels->W2[l] > 0, "altcorr pyramid backward (chunk-major): bad level %d", l);
  return ufr_altcorr_mfma_backward(fmap1, levels, coords, 1, grad_out_cm, fmap1_grad, workspace, B, H1, W1, C, radius, scale, accumulate,
                                   ufr::as_stream(stream), 1);
}

extern "C" long ufr_altcorr_pyramid_workspace_bytes(int B, int H1, int W1, int C, int radius, int num_levels) {
  return workspace_bytes(B, H1, W1, C, radius, num_levels > 0 ? num_levels : 1);
}
