// raft_corr.hip -- RAFT cost-volume operators for gfx950.
//   (1) alt_cuda_corr: on-the-fly windowed correlation   models/alt_cuda_corr/correlation_kernel.cu
//   (2) CorrBlock lookup: 4-level (2r+1)^2 bilinear window gather  models/raft/corr.py:72-96
//
// alt_corr maths (correlation_kernel.cu:52-115): for pixel (h1,w1) with coordinate (x,y), integer
// corner (fx,fy)=floor, fraction (dx,dy):  s[iy][ix] = <fmap1[h1,w1,:], fmap2[fy-r+iy, fx-r+ix,:]>
// for iy,ix in [0,2r+1] (0 outside the image), and
//   corr[oy + rd*ox] = (1-dy)(1-dx) s[oy][ox] + (1-dy)dx s[oy][ox+1] + dy(1-dx) s[oy+1][ox] + dy dx s[oy+1][ox+1]
// The reference accumulates this per 32-channel slab; here one workgroup owns one pixel and the
// whole channel axis, so the dot products are complete before the bilinear blend.
#include <cstdlib>

#include "ufr_common.h"

// raft_altcorr_mfma.hip
bool ufr_altcorr_mfma_serves(int C, int radius);
int ufr_altcorr_mfma_forward(const float* f1, const ufr_altcorr_levels* lv, const float* coords, int planar, float* out, int B,
                             int H1, int W1, int C, int radius, float scale, hipStream_t st);

namespace {

constexpr int ALT_MAX_GRID = 12;  // rd+1 <= 12  (radius <= 5)

// One 128-thread workgroup per (b,n,h1,w1).  Thread t < (rd+1)^2 owns window point (iy,ix):
// fmap1 row is broadcast from LDS, fmap2 row streamed 16 B at a time.
__global__ void altcorr_fwd(const float* __restrict__ fmap1, const float* __restrict__ fmap2,
                            const float* __restrict__ coords, float* __restrict__ corr, int N,
                            int H1, int W1, int H2, int W2, int C, int r) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* f1 = smem;                         // [C]
  float* s = smem + ufr::kWave * ((C + 63) / 64);  // [(rd+1)^2]  (offset = C rounded to 64)
  const int rd = 2 * r + 1, gd = rd + 1;
  const long pix = blockIdx.x;  // ((b*N + n)*H1 + h1)*W1 + w1
  const int w1 = (int)(pix % W1);
  const int h1 = (int)((pix / W1) % H1);
  const int n = (int)((pix / ((long)W1 * H1)) % N);
  const int b = (int)(pix / ((long)W1 * H1 * N));
  const int tid = threadIdx.x;

  const float* f1g = fmap1 + (((size_t)b * H1 + h1) * W1 + w1) * C;
  for (int c = tid; c < C; c += blockDim.x) f1[c] = f1g[c];
  const float x = coords[pix * 2 + 0], y = coords[pix * 2 + 1];
  const float fx = floorf(x), fy = floorf(y);
  const float dx = x - fx, dy = y - fy;
  __syncthreads();

  if (tid < gd * gd) {
    const int iy = tid / gd, ix = tid - iy * gd;
    const int h2 = (int)fy - r + iy, w2 = (int)fx - r + ix;
    float acc = 0.f;
    if (h2 >= 0 && h2 < H2 && w2 >= 0 && w2 < W2) {
      const float* f2 = fmap2 + (((size_t)b * H2 + h2) * W2 + w2) * C;
      int c = 0;
      if ((C & 3) == 0) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (; c < C; c += 4) {
          const float4 u = *reinterpret_cast<const float4*>(f1 + c);
          const float4 v = *reinterpret_cast<const float4*>(f2 + c);
          a0 = fmaf(u.x, v.x, a0); a1 = fmaf(u.y, v.y, a1);
          a2 = fmaf(u.z, v.z, a2); a3 = fmaf(u.w, v.w, a3);
        }
        acc = (a0 + a1) + (a2 + a3);
      } else {
        for (; c < C; ++c) acc = fmaf(f1[c], f2[c], acc);
      }
    }
    s[tid] = acc;
  }
  __syncthreads();
  if (tid < rd * rd) {
    const int ox = tid / rd, oy = tid - ox * rd;  // channel = oy + rd*ox  (:92-95)
    const float v = (1 - dy) * (1 - dx) * s[oy * gd + ox] + (1 - dy) * dx * s[oy * gd + ox + 1] +
                    dy * (1 - dx) * s[(oy + 1) * gd + ox] + dy * dx * s[(oy + 1) * gd + ox + 1];
    const size_t plane = (size_t)H1 * W1;
    corr[(((size_t)b * N + n) * rd * rd + tid) * plane + (size_t)h1 * W1 + w1] = v;
  }
}

// Adjoint (correlation_kernel.cu:122-256).  One workgroup per pixel, one thread per channel
// (strided when C > blockDim): fmap1_grad is owned by the workgroup (plain store, summed over n
// through `+=` in registers is impossible across n-blocks, so N>1 uses atomics too);
// fmap2_grad is scattered with float atomics, 256 contiguous bytes per wave instruction.
template <bool SCATTER>
__global__ void altcorr_bwd(const float* __restrict__ fmap1, const float* __restrict__ fmap2,
                            const float* __restrict__ coords, const float* __restrict__ corr_grad,
                            float* __restrict__ fmap1_grad, float* __restrict__ fmap2_grad, int N,
                            int H1, int W1, int H2, int W2, int C, int r) {
  __shared__ float gs[ALT_MAX_GRID * ALT_MAX_GRID];
  const int rd = 2 * r + 1, gd = rd + 1;
  const long pix = blockIdx.x;
  const int w1 = (int)(pix % W1);
  const int h1 = (int)((pix / W1) % H1);
  const int n = (int)((pix / ((long)W1 * H1)) % N);
  const int b = (int)(pix / ((long)W1 * H1 * N));
  const int tid = threadIdx.x;
  const float x = coords[pix * 2 + 0], y = coords[pix * 2 + 1];
  const float fx = floorf(x), fy = floorf(y);
  const float dx = x - fx, dy = y - fy;
  const size_t plane = (size_t)H1 * W1;
  const float* gp = corr_grad + (((size_t)b * N + n) * rd * rd) * plane + (size_t)h1 * W1 + w1;

  for (int t = tid; t < gd * gd; t += blockDim.x) {
    const int iy = t / gd, ix = t - iy * gd;
    float g = 0.f;
    if (iy > 0 && ix > 0)   g += gp[plane * ((iy - 1) + rd * (ix - 1))] * dy * dx;
    if (iy > 0 && ix < rd)  g += gp[plane * ((iy - 1) + rd * ix)] * dy * (1 - dx);
    if (iy < rd && ix > 0)  g += gp[plane * (iy + rd * (ix - 1))] * (1 - dy) * dx;
    if (iy < rd && ix < rd) g += gp[plane * (iy + rd * ix)] * (1 - dy) * (1 - dx);
    gs[t] = g;
  }
  __syncthreads();

  const float* f1g = fmap1 + (((size_t)b * H1 + h1) * W1 + w1) * C;
  float* g1g = fmap1_grad + (((size_t)b * H1 + h1) * W1 + w1) * C;
  for (int c = tid; c < C; c += blockDim.x) {
    const float f1 = f1g[c];
    float acc = 0.f;
    for (int iy = 0; iy < gd; ++iy) {
      const int h2 = (int)fy - r + iy;
      if (h2 < 0 || h2 >= H2) continue;
      for (int ix = 0; ix < gd; ++ix) {
        const int w2 = (int)fx - r + ix;
        if (w2 < 0 || w2 >= W2) continue;
        const float g = gs[iy * gd + ix];
        const size_t o = (((size_t)b * H2 + h2) * W2 + w2) * C + c;
        acc = fmaf(g, fmap2[o], acc);
        if (SCATTER) atomicAdd(&fmap2_grad[o], g * f1);
      }
    }
    if (N == 1) g1g[c] = acc; else atomicAdd(&g1g[c], acc);
  }
}

// fmap2 adjoint without one global atomic per (pixel, window point, channel), owner-computes form.
// A workgroup takes a tile of 8x8 pixels and a slab of 64 channels.  Flow fields are smooth, so the 64
// windows of a tile land in a small region of fmap2; a 24x24 region anchored at the tile's smallest
// window corner is OWNED by the workgroup, one thread per cell (576 threads), 64 channel accumulators in
// registers.  For every pixel p of the tile a thread looks up its coefficient
//     A = gs[p][cell - corner_p]   (0 when the cell is outside p's window; gs = adjoint of the bilinear blend)
// and adds A * fmap1[p][0..63], the fmap1 row coming from LDS as broadcast 16-byte reads; a wave whose 64
// cells all miss p's window skips the row.  It is a 576x64x64 product per tile with a 17%-dense A, done
// on the VALU with no atomics and a fixed summation order.  Accumulators are transposed through LDS and
// flushed with one global atomic per touched (cell, channel), 64 contiguous bytes per cell (tiles
// overlap by the window halo).  Window points outside the owned region (discontinuous flow) are added
// directly with global atomics afterwards: always correct, fast when the flow is smooth.
constexpr int AT_TH = 8, AT_TW = 8, AT_PIX = AT_TH * AT_TW;   // pixels per tile
constexpr int AT_CS = 64;                                     // channels per slab
constexpr int AT_RH = 24, AT_RW = 24;                         // region owned by the workgroup
constexpr int AT_NT = AT_RH * AT_RW;                          // 576 threads = 9 waves
constexpr int AT_WAVES = AT_NT / 64;
constexpr int AT_CHUNK = 16;                                  // channels per transpose pass

template <int RADIUS>
__global__ __launch_bounds__(AT_NT) void altcorr_bwd2_tiled(
    const float* __restrict__ fmap1, const float* __restrict__ coords, const float* __restrict__ corr_grad,
    float* __restrict__ fmap2_grad, int N, int H1, int W1, int H2, int W2, int C, int tiles_x) {
  constexpr int r = RADIUS, rd = 2 * r + 1, gd = rd + 1, npt = gd * gd;
  constexpr int kCg = rd * rd * AT_PIX, kGs = AT_PIX * npt, kTr = AT_WAVES * 64 * (AT_CHUNK + 1);
  constexpr int kScratch = (kCg + kGs) > kTr ? (kCg + kGs) : kTr;
  __shared__ __attribute__((aligned(16))) float scratch[kScratch];   // cg | gs, later the transpose buffer
  __shared__ __attribute__((aligned(16))) float f1s[AT_PIX * AT_CS];
  __shared__ int cxs[AT_PIX], cys[AT_PIX];
  __shared__ float dxs[AT_PIX], dys[AT_PIX];
  __shared__ int org[2];
  float* cg = scratch;                                         // corr_grad of the tile, [channel][pixel]
  float* gs = scratch + kCg;                                   // [pixel][iy*gd+ix]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x, c0 = blockIdx.y * AT_CS, bn = blockIdx.z;   // bn = b*N + n
  const int b = bn / N;
  const int ty0 = (tile / tiles_x) * AT_TH, tx0 = (tile % tiles_x) * AT_TW;
  const size_t plane = (size_t)H1 * W1;

  if (tid == 0) { org[0] = 0x7fffffff; org[1] = 0x7fffffff; }
  __syncthreads();
  if (tid < AT_PIX) {
    const int h1 = ty0 + tid / AT_TW, w1 = tx0 + tid % AT_TW;
    int cx = 0x3fffffff, cy = 0x3fffffff;                      // invalid pixel: never inside the image
    float dx = 0.f, dy = 0.f;
    if (h1 < H1 && w1 < W1) {
      const size_t pix = (size_t)bn * plane + (size_t)h1 * W1 + w1;
      const float x = coords[pix * 2 + 0], y = coords[pix * 2 + 1];
      const float fx = floorf(x), fy = floorf(y);
      dx = x - fx; dy = y - fy;
      // clamp far-away windows so the integer arithmetic below cannot overflow; they stay outside
      cx = (int)fminf(fmaxf(fx, -1.0e6f), 1.0e6f) - r;
      cy = (int)fminf(fmaxf(fy, -1.0e6f), 1.0e6f) - r;
      if (cx + gd > 0 && cx < W2 && cy + gd > 0 && cy < H2) {   // window meets the image
        atomicMin(&org[0], max(cy, 0));
        atomicMin(&org[1], max(cx, 0));
      }
    }
    cxs[tid] = cx; cys[tid] = cy; dxs[tid] = dx; dys[tid] = dy;
  }
  for (int i = tid; i < kCg; i += AT_NT) {                     // 8 consecutive pixels = 32 contiguous bytes
    const int p = i % AT_PIX, ch = i / AT_PIX;
    const int h1 = ty0 + p / AT_TW, w1 = tx0 + p % AT_TW;
    cg[i] = (h1 < H1 && w1 < W1) ? corr_grad[((size_t)bn * rd * rd + ch) * plane + (size_t)h1 * W1 + w1] : 0.f;
  }
  for (int i = tid; i < AT_PIX * AT_CS; i += AT_NT) {
    const int p = i / AT_CS, c = i % AT_CS;
    const int h1 = ty0 + p / AT_TW, w1 = tx0 + p % AT_TW;
    f1s[i] = (h1 < H1 && w1 < W1 && c0 + c < C) ? fmap1[(((size_t)b * H1 + h1) * W1 + w1) * C + c0 + c] : 0.f;
  }
  __syncthreads();
  // gs[p][iy*gd+ix] (correlation_kernel.cu:196-214): channel of corr_grad = oy + rd*ox
  for (int i = tid; i < kGs; i += AT_NT) {
    const int p = i % AT_PIX, t = i / AT_PIX;
    const int iy = t / gd, ix = t - iy * gd;
    const float dx = dxs[p], dy = dys[p];
    float g = 0.f;
    if (iy > 0 && ix > 0)   g += cg[((iy - 1) + rd * (ix - 1)) * AT_PIX + p] * dy * dx;
    if (iy > 0 && ix < rd)  g += cg[((iy - 1) + rd * ix) * AT_PIX + p] * dy * (1 - dx);
    if (iy < rd && ix > 0)  g += cg[(iy + rd * (ix - 1)) * AT_PIX + p] * (1 - dy) * dx;
    if (iy < rd && ix < rd) g += cg[(iy + rd * ix) * AT_PIX + p] * (1 - dy) * (1 - dx);
    gs[p * npt + t] = g;
  }
  __syncthreads();
  const int oy = org[0], ox = org[1];
  if (oy == 0x7fffffff) return;                                // no window of this tile meets the image

  const int h2 = oy + tid / AT_RW, w2 = ox + tid % AT_RW;      // this thread's cell
  float acc[AT_CS];
#pragma unroll
  for (int c = 0; c < AT_CS; ++c) acc[c] = 0.f;
  for (int p = 0; p < AT_PIX; ++p) {
    const int iy = h2 - cys[p], ix = w2 - cxs[p];
    const bool hit = (unsigned)iy < (unsigned)gd && (unsigned)ix < (unsigned)gd;
    if (__ballot(hit) == 0) continue;                          // none of the wave's 64 cells is in p's window
    const float A = hit ? gs[p * npt + iy * gd + ix] : 0.f;
    const float4* f1v = reinterpret_cast<const float4*>(f1s + p * AT_CS);
#pragma unroll
    for (int c4 = 0; c4 < AT_CS / 4; ++c4) {
      const float4 f = f1v[c4];                                // same address in every lane: LDS broadcast
      acc[4 * c4 + 0] = fmaf(A, f.x, acc[4 * c4 + 0]);
      acc[4 * c4 + 1] = fmaf(A, f.y, acc[4 * c4 + 1]);
      acc[4 * c4 + 2] = fmaf(A, f.z, acc[4 * c4 + 2]);
      acc[4 * c4 + 3] = fmaf(A, f.w, acc[4 * c4 + 3]);
    }
  }

  // window points outside the owned region: lane = channel, one pixel per wave at a time
  if (c0 + lane < C) {
    for (int p = wave; p < AT_PIX; p += AT_WAVES) {
      const int cy = cys[p], cx = cxs[p];
      if (cy >= oy && cy + gd <= oy + AT_RH && cx >= ox && cx + gd <= ox + AT_RW) continue;   // fully owned
      if (cy + gd <= 0 || cy >= H2 || cx + gd <= 0 || cx >= W2) continue;                     // off the image
      const float f1 = f1s[p * AT_CS + lane];
      for (int iy = 0; iy < gd; ++iy) {
        const int hh = cy + iy;
        if (hh < 0 || hh >= H2) continue;
        for (int ix = 0; ix < gd; ++ix) {
          const int ww = cx + ix;
          if (ww < 0 || ww >= W2) continue;
          if (hh >= oy && hh < oy + AT_RH && ww >= ox && ww < ox + AT_RW) continue;           // owned: in acc
          atomicAdd(&fmap2_grad[(((size_t)b * H2 + hh) * W2 + ww) * C + c0 + lane], gs[p * npt + iy * gd + ix] * f1);
        }
      }
    }
  }
  __syncthreads();                                             // gs / cg are dead: reuse as transpose buffer

  // flush: 16 channels at a time through LDS so that one atomic instruction covers 4 cells x 64 bytes
  float* tr = scratch + wave * 64 * (AT_CHUNK + 1);
#pragma unroll
  for (int ch0 = 0; ch0 < AT_CS; ch0 += AT_CHUNK) {
#pragma unroll
    for (int i = 0; i < AT_CHUNK; ++i) tr[lane * (AT_CHUNK + 1) + i] = acc[ch0 + i];
    __builtin_amdgcn_wave_barrier();
    const int cc = lane % AT_CHUNK;
#pragma unroll
    for (int i = 0; i < 64 / (64 / AT_CHUNK); ++i) {           // 16 passes of 4 cells
      const int cell_l = i * (64 / AT_CHUNK) + lane / AT_CHUNK;
      const float v = tr[cell_l * (AT_CHUNK + 1) + cc];
      const int cell = wave * 64 + cell_l;
      const int hh = oy + cell / AT_RW, ww = ox + cell % AT_RW;
      if (v != 0.f && hh < H2 && ww < W2 && c0 + ch0 + cc < C)
        atomicAdd(&fmap2_grad[(((size_t)b * H2 + hh) * W2 + ww) * C + c0 + ch0 + cc], v);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// CorrBlock lookup (corr.py:72-96).  One thread per (pixel, level): walks the (2r+2)^2 integer grid
// around coords/2^l row by row and emits the (2r+1)^2 bilinear samples; all samples of a window
// share one fractional offset.  Out-of-volume grid points read as 0 (grid_sample zero padding).
// Output channel = l*rd*rd + i*rd + j with i the x offset and j the y offset (corr.py:80-86).
// Writes are coalesced over pixels; reads are row segments of the pixel's own volume slice.
// ------------------------------------------------------------------------------------------------
constexpr int LK_MAX_RD = 9;  // radius <= 4 (RAFT uses 4; small RAFT 3)

template <int RADIUS>
__global__ void lookup_fwd(ufr_pyramid pyr, const float* __restrict__ coords,
                           float* __restrict__ out, int B, int H1, int W1) {
  constexpr int r = RADIUS, rd = 2 * RADIUS + 1;
  const int L = pyr.num_levels;
  const size_t plane = (size_t)H1 * W1;
  const long total = (long)B * plane * L;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const long q = idx % (long)plane;                 // pixel within image (fastest -> coalesced)
    const int l = (int)((idx / (long)plane) % L);
    const int b = (int)(idx / ((long)plane * L));
    const int Hl = pyr.Hl[l], Wl = pyr.Wl[l];
    const float* vol = pyr.vol[l] + ((size_t)b * plane + q) * Hl * Wl;
    const float inv = 1.0f / (float)(1 << l);
    const float cx = coords[((size_t)b * 2 + 0) * plane + q] * inv;
    const float cy = coords[((size_t)b * 2 + 1) * plane + q] * inv;
    const float x0f = floorf(cx), y0f = floorf(cy);
    const float ax = cx - x0f, ay = cy - y0f;
    const int xb = (int)x0f - r, yb = (int)y0f - r;
    float* o = out + ((size_t)b * L * rd * rd + (size_t)l * rd * rd) * plane + q;
    float prev[rd + 1], cur[rd + 1];
#pragma unroll
    for (int gy = 0; gy <= rd; ++gy) {
      const int yy = yb + gy;
      const bool rowok = (yy >= 0 && yy < Hl);
#pragma unroll
      for (int gx = 0; gx <= rd; ++gx) {
        const int xx = xb + gx;
        cur[gx] = (rowok && xx >= 0 && xx < Wl) ? vol[(size_t)yy * Wl + xx] : 0.f;
      }
      if (gy > 0) {
        const int j = gy - 1;
#pragma unroll
        for (int i = 0; i < rd; ++i) {
          const float v = prev[i] * (1 - ax) * (1 - ay) + prev[i + 1] * ax * (1 - ay) +
                          cur[i] * (1 - ax) * ay + cur[i + 1] * ax * ay;
          o[(size_t)(i * rd + j) * plane] = v;
        }
      }
#pragma unroll
      for (int gx = 0; gx <= rd; ++gx) prev[gx] = cur[gx];
    }
  }
}

// Adjoint: volume slice p is read by pixel p only, so the thread that owns (pixel, level) is the
// only writer of its slice -> plain read-modify-write, no atomics.  grad volumes accumulate (+=):
// the caller zeroes them once and may run several lookups' adjoints into the same buffers.
template <int RADIUS>
__global__ void lookup_bwd(ufr_pyramid pyr, const float* __restrict__ coords,
                           const float* __restrict__ gout, int B, int H1, int W1) {
  constexpr int r = RADIUS, rd = 2 * RADIUS + 1;
  const int L = pyr.num_levels;
  const size_t plane = (size_t)H1 * W1;
  const long total = (long)B * plane * L;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const long q = idx % (long)plane;
    const int l = (int)((idx / (long)plane) % L);
    const int b = (int)(idx / ((long)plane * L));
    const int Hl = pyr.Hl[l], Wl = pyr.Wl[l];
    float* gv = pyr.grad_vol[l] + ((size_t)b * plane + q) * Hl * Wl;
    const float inv = 1.0f / (float)(1 << l);
    const float cx = coords[((size_t)b * 2 + 0) * plane + q] * inv;
    const float cy = coords[((size_t)b * 2 + 1) * plane + q] * inv;
    const float x0f = floorf(cx), y0f = floorf(cy);
    const float ax = cx - x0f, ay = cy - y0f;
    const int xb = (int)x0f - r, yb = (int)y0f - r;
    const float* g = gout + ((size_t)b * L * rd * rd + (size_t)l * rd * rd) * plane + q;
    // grid point (gy,gx) receives from samples (j,i) in {gy-1,gy} x {gx-1,gx}
    float gprev[rd + 1], gcur[rd + 1];  // sample-row gradients g[i][j] for j = gy-1 / gy
#pragma unroll
    for (int i = 0; i <= rd; ++i) gprev[i] = 0.f, gcur[i] = 0.f;
#pragma unroll
    for (int gy = 0; gy <= rd; ++gy) {
#pragma unroll
      for (int i = 0; i < rd; ++i)
        gcur[i] = (gy < rd) ? g[(size_t)(i * rd + gy) * plane] : 0.f;
      const int yy = yb + gy;
      if (yy >= 0 && yy < Hl) {
#pragma unroll
        for (int gx = 0; gx <= rd; ++gx) {
          const int xx = xb + gx;
          if (xx < 0 || xx >= Wl) continue;
          float acc = 0.f;
          if (gx < rd) acc += gcur[gx] * (1 - ax) * (1 - ay) + gprev[gx] * (1 - ax) * ay;
          if (gx > 0) acc += gcur[gx - 1] * ax * (1 - ay) + gprev[gx - 1] * ax * ay;
          gv[(size_t)yy * Wl + xx] += acc;
        }
      }
#pragma unroll
      for (int i = 0; i < rd; ++i) gprev[i] = gcur[i];
    }
  }
}

// The same adjoint with the (2r + 2)^2 grid points of a pixel dealt to the LANES (round 6): the one-thread-per-(pixel, level) form above
// walks its 100 read-modify-writes one after the other -- 30,720 threads on a 48 x 160 grid, ~1 us per dependent round trip: 100 us
// per lookup, 1.2 ms of a RAFT iteration (profiles/r6_c3_step_trace.md).  Here a workgroup owns 16 consecutive pixels of one level:
// their (2r+1)^2 x 16 output gradients are staged through LDS (coalesced: 16 pixels of a channel are 64 contiguous bytes), then
// consecutive lanes take consecutive grid points of ONE pixel -- a grid row is one contiguous 40-byte run of that pixel's volume
// slice -- and every (pixel, point) is one independent read-modify-write.  Still one writer per slice (the tile owns its pixels): no
// atomics, and the same four products in the same order as above: bit-identical results.
template <int RADIUS>
__global__ __launch_bounds__(256) void lookup_bwd_tiled(ufr_pyramid pyr, const float* __restrict__ coords, const float* __restrict__ gout,
                                                        int B, int H1, int W1) {
  constexpr int r = RADIUS, rd = 2 * RADIUS + 1, gd = rd + 1, TP = 16;
  __shared__ float sg[rd * rd][TP];
  __shared__ float sax[TP], say[TP];
  __shared__ int sxb[TP], syb[TP];
  const int L = pyr.num_levels;
  const size_t plane = (size_t)H1 * W1;
  const int tiles = (int)((plane + TP - 1) / TP);
  const int l = blockIdx.y, tile = blockIdx.x % tiles, b = blockIdx.x / tiles;
  const size_t q0 = (size_t)tile * TP;
  const int Hl = pyr.Hl[l], Wl = pyr.Wl[l], tid = threadIdx.x;
  const float* g = gout + ((size_t)b * L * rd * rd + (size_t)l * rd * rd) * plane + q0;
  for (int t = tid; t < rd * rd * TP; t += 256) {
    const int ch = t >> 4, i = t & 15;
    sg[ch][i] = q0 + i < plane ? g[(size_t)ch * plane + i] : 0.f;
  }
  if (tid < TP) {
    const size_t q = min(q0 + tid, plane - 1);
    const float inv = 1.0f / (float)(1 << l);
    const float cx = coords[((size_t)b * 2 + 0) * plane + q] * inv, cy = coords[((size_t)b * 2 + 1) * plane + q] * inv;
    const float x0f = floorf(cx), y0f = floorf(cy);
    sax[tid] = cx - x0f; say[tid] = cy - y0f;
    sxb[tid] = (int)x0f - r; syb[tid] = (int)y0f - r;
  }
  __syncthreads();
  for (int t = tid; t < TP * gd * gd; t += 256) {
    const int i = t / (gd * gd), pt = t - i * (gd * gd), gy = pt / gd, gx = pt - gy * gd;
    if (q0 + i >= plane) continue;
    const int yy = syb[i] + gy, xx = sxb[i] + gx;
    if (yy < 0 || yy >= Hl || xx < 0 || xx >= Wl) continue;
    const float ax = sax[i], ay = say[i];
    // grid point (gy, gx) receives from samples (j, i) in {gy-1, gy} x {gx-1, gx}: gcur = g[.][gy], gprev = g[.][gy - 1] (0 outside)
    const float c0 = (gx < rd && gy < rd) ? sg[gx * rd + gy][i] : 0.f, p0 = (gx < rd && gy > 0) ? sg[gx * rd + gy - 1][i] : 0.f;
    const float c1 = (gx > 0 && gy < rd) ? sg[(gx - 1) * rd + gy][i] : 0.f, p1 = (gx > 0 && gy > 0) ? sg[(gx - 1) * rd + gy - 1][i] : 0.f;
    float acc = 0.f;
    if (gx < rd) acc += c0 * (1 - ax) * (1 - ay) + p0 * (1 - ax) * ay;
    if (gx > 0) acc += c1 * ax * (1 - ay) + p1 * ax * ay;
    float* gv = pyr.grad_vol[l] + ((size_t)b * plane + q0 + i) * Hl * Wl;
    gv[(size_t)yy * Wl + xx] += acc;
  }
}

}  // namespace

extern "C" int ufr_altcorr_forward(const float* fmap1, const float* fmap2, const float* coords,
                                   float* corr, int B, int N, int H1, int W1, int H2, int W2, int C,
                                   int radius, ufr_stream_t stream) {
  UFR_REQUIRE(fmap1 && fmap2 && coords && corr, "alt_corr forward: null pointer argument");
  UFR_REQUIRE(B > 0 && N > 0 && H1 > 0 && W1 > 0 && H2 > 0 && W2 > 0 && C > 0, "alt_corr forward: bad shape");
  UFR_REQUIRE(radius >= 0 && 2 * radius + 2 <= ALT_MAX_GRID, "alt_corr forward: radius %d unsupported (max %d)",
              radius, (ALT_MAX_GRID - 2) / 2);
  if (N == 1 && ufr_altcorr_mfma_serves(C, radius)) {     // gather-GEMM on the fp32 matrix cores (raft_altcorr_mfma.hip)
    ufr_altcorr_levels lv{};
    lv.num_levels = 1; lv.fmap2[0] = fmap2; lv.H2[0] = H2; lv.W2[0] = W2; lv.coord_scale[0] = 1.0f;
    return ufr_altcorr_mfma_forward(fmap1, &lv, coords, 0, corr, B, H1, W1, C, radius, 1.0f, ufr::as_stream(stream));
  }
  const int gd = 2 * radius + 2;
  const int nt = ufr::round_up(gd * gd, 64);
  const size_t lds = (size_t)(ufr::kWave * ((C + 63) / 64) + gd * gd) * sizeof(float);
  UFR_REQUIRE(lds <= 64 * 1024, "alt_corr forward: C=%d too large", C);
  const long blocks = (long)B * N * H1 * W1;
  UFR_REQUIRE(blocks < 2147483647L, "alt_corr forward: too many pixels");
  hipLaunchKernelGGL(altcorr_fwd, dim3((unsigned)blocks), dim3(nt), lds, ufr::as_stream(stream),
                     fmap1, fmap2, coords, corr, N, H1, W1, H2, W2, C, radius);
  return ufr::launched("altcorr_fwd");
}

extern "C" int ufr_altcorr_backward(const float* fmap1, const float* fmap2, const float* coords,
                                    const float* corr_grad, float* fmap1_grad, float* fmap2_grad,
                                    float* coords_grad, int B, int N, int H1, int W1, int H2,
                                    int W2, int C, int radius, ufr_stream_t stream) {
  UFR_REQUIRE(fmap1 && fmap2 && coords && corr_grad && fmap1_grad && fmap2_grad && coords_grad,
              "alt_corr backward: null pointer argument");
  UFR_REQUIRE(B > 0 && N > 0 && H1 > 0 && W1 > 0 && H2 > 0 && W2 > 0 && C > 0, "alt_corr backward: bad shape");
  UFR_REQUIRE(radius >= 0 && 2 * radius + 2 <= ALT_MAX_GRID, "alt_corr backward: radius %d unsupported", radius);
  hipStream_t st = ufr::as_stream(stream);
  hipError_t e = hipMemsetAsync(fmap2_grad, 0, sizeof(float) * (size_t)B * H2 * W2 * C, st);
  if (e == hipSuccess) e = hipMemsetAsync(coords_grad, 0, sizeof(float) * (size_t)B * N * H1 * W1 * 2, st);
  if (e == hipSuccess && N > 1) e = hipMemsetAsync(fmap1_grad, 0, sizeof(float) * (size_t)B * H1 * W1 * C, st);
  if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "alt_corr backward: memset: %s", hipGetErrorString(e));
  const long blocks = (long)B * N * H1 * W1;
  UFR_REQUIRE(blocks < 2147483647L, "alt_corr backward: too many pixels");
  const int nt = C >= 256 ? 256 : ufr::round_up(C, 64);
  if ((long)B * N > 65535 || (radius != 4 && radius != 3)) {
    // shapes the tiled form does not cover: one workgroup per pixel does both adjoints, global atomics
    hipLaunchKernelGGL(altcorr_bwd<true>, dim3((unsigned)blocks), dim3(nt), 0, st, fmap1, fmap2, coords,
                       corr_grad, fmap1_grad, fmap2_grad, N, H1, W1, H2, W2, C, radius);
    return ufr::launched("altcorr_bwd");
  }
  hipLaunchKernelGGL(altcorr_bwd<false>, dim3((unsigned)blocks), dim3(nt), 0, st, fmap1, fmap2, coords,
                     corr_grad, fmap1_grad, fmap2_grad, N, H1, W1, H2, W2, C, radius);
  const int tiles_x = ufr::ceil_div(W1, AT_TW), tiles_y = ufr::ceil_div(H1, AT_TH);
  const dim3 grid(tiles_x * tiles_y, ufr::ceil_div(C, AT_CS), B * N);
  if (radius == 4)
    hipLaunchKernelGGL(altcorr_bwd2_tiled<4>, grid, dim3(AT_NT), 0, st, fmap1, coords, corr_grad, fmap2_grad,
                       N, H1, W1, H2, W2, C, tiles_x);
  else
    hipLaunchKernelGGL(altcorr_bwd2_tiled<3>, grid, dim3(AT_NT), 0, st, fmap1, coords, corr_grad, fmap2_grad,
                       N, H1, W1, H2, W2, C, tiles_x);
  return ufr::launched("altcorr_bwd2_tiled");
}

static int check_pyr(const ufr_pyramid* pyr, bool need_grad) {
  UFR_REQUIRE(pyr, "corr lookup: null pyramid");
  UFR_REQUIRE(pyr->num_levels >= 1 && pyr->num_levels <= UFR_MAX_LEVELS, "corr lookup: %d levels unsupported",
              pyr->num_levels);
  for (int l = 0; l < pyr->num_levels; ++l) {
    UFR_REQUIRE(pyr->Hl[l] > 0 && pyr->Wl[l] > 0, "corr lookup: empty level %d", l);
    UFR_REQUIRE(need_grad ? pyr->grad_vol[l] != nullptr : pyr->vol[l] != nullptr,
                "corr lookup: null volume pointer at level %d", l);
  }
  return UFR_OK;
}

extern "C" int ufr_corr_lookup_forward(const ufr_pyramid* pyr, const float* coords, float* out,
                                       int B, int H1, int W1, int radius, ufr_stream_t stream) {
  if (int rc = check_pyr(pyr, false)) return rc;
  UFR_REQUIRE(coords && out && B > 0 && H1 > 0 && W1 > 0, "corr lookup forward: bad argument");
  UFR_REQUIRE(radius >= 0 && 2 * radius + 1 <= LK_MAX_RD, "corr lookup: radius %d unsupported (max %d)", radius,
              (LK_MAX_RD - 1) / 2);
  const long total = (long)B * H1 * W1 * pyr->num_levels;
  const dim3 grid(ufr::stream_grid(total, 128)), block(128);
  hipStream_t st = ufr::as_stream(stream);
  switch (radius) {
    case 0: hipLaunchKernelGGL(lookup_fwd<0>, grid, block, 0, st, *pyr, coords, out, B, H1, W1); break;
    case 1: hipLaunchKernelGGL(lookup_fwd<1>, grid, block, 0, st, *pyr, coords, out, B, H1, W1); break;
    case 2: hipLaunchKernelGGL(lookup_fwd<2>, grid, block, 0, st, *pyr, coords, out, B, H1, W1); break;
    case 3: hipLaunchKernelGGL(lookup_fwd<3>, grid, block, 0, st, *pyr, coords, out, B, H1, W1); break;
    default: hipLaunchKernelGGL(lookup_fwd<4>, grid, block, 0, st, *pyr, coords, out, B, H1, W1); break;
  }
  return ufr::launched("lookup_fwd");
}

extern "C" int ufr_corr_lookup_backward(const ufr_pyramid* pyr, const float* coords,
                                        const float* grad_out, int B, int H1, int W1, int radius,
                                        ufr_stream_t stream) {
  if (int rc = check_pyr(pyr, true)) return rc;
  UFR_REQUIRE(coords && grad_out && B > 0 && H1 > 0 && W1 > 0, "corr lookup backward: bad argument");
  UFR_REQUIRE(radius >= 0 && 2 * radius + 1 <= LK_MAX_RD, "corr lookup: radius %d unsupported", radius);
  const long total = (long)B * H1 * W1 * pyr->num_levels;
  const dim3 grid(ufr::stream_grid(total, 128)), block(128);
  hipStream_t st = ufr::as_stream(stream);
  static const bool tiled = [] { const char* e = getenv("UFR_LOOKUP_BWD_TILED"); return !(e && e[0] == '0'); }();
  if (tiled && (radius == 4 || radius == 3) && (long)B * ((H1 * W1 + 15) / 16) < (1L << 31)) {   // RAFT's radii: grid points over the lanes
    const dim3 gt((unsigned)(B * ((H1 * W1 + 15) / 16)), pyr->num_levels);
    if (radius == 4) lookup_bwd_tiled<4><<<gt, 256, 0, st>>>(*pyr, coords, grad_out, B, H1, W1);
    else lookup_bwd_tiled<3><<<gt, 256, 0, st>>>(*pyr, coords, grad_out, B, H1, W1);
    return ufr::launched("lookup_bwd_tiled");
  }
  switch (radius) {
    case 0: hipLaunchKernelGGL(lookup_bwd<0>, grid, block, 0, st, *pyr, coords, grad_out, B, H1, W1); break;
    case 1: hipLaunchKernelGGL(lookup_bwd<1>, grid, block, 0, st, *pyr, coords, grad_out, B, H1, W1); break;
    case 2: hipLaunchKernelGGL(lookup_bwd<2>, grid, block, 0, st, *pyr, coords, grad_out, B, H1, W1); break;
    case 3: hipLaunchKernelGGL(lookup_bwd<3>, grid, block, 0, st, *pyr, coords, grad_out, B, H1, W1); break;
    default: hipLaunchKernelGGL(lookup_bwd<4>, grid, block, 0, st, *pyr, coords, grad_out, B, H1, W1); break;
  }
  return ufr::launched("lookup_bwd");
}
