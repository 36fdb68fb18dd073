// engine_small.hip -- the 2-channel layers of FlowNetC's refinement on the engine's chunk-major layout (csrc/igemm.hip):
// predict_flow* = Conv2d(Cin, 2, 3, 1, 1) and upsampled_flow* = ConvTranspose2d(2, 2, 4, 2, 1)
// (models/FlowNetC.py:43-50, models/submodules.py:85-90).  HBM-bound: one pass over the concatenation buffer.
//
//   flow_head_planes_fwd   thread = (pixel, 8-channel group): per chunk and tap one 16-byte read from each of the three
//                          planes (a wave reads 16 pixels x 64 B = whole lines), v = p0 + p1 + p2 exactly, 16 FMAs against
//                          weights repacked [chunk][tap][out][32]; the four groups of a pixel are added by two shuffles
//   flow_head_planes_bwd   thread = (pixel, 8-channel group): the pixel's 3x3x2 gradient neighbourhood in registers,
//                          per chunk 144 FMAs and one 32-byte store (or read-add-store) into the fp32 gradient sum
//   flow_up_planes_fwd     ConvTranspose2d(2,2,4,2,1) written straight into the concatenation's last chunk (2 channels
//                          + 30 zeros, all three planes)
//   flow_up_planes_bwd     its data gradient from channels 0-1 of that chunk of the fp32 gradient sum
#include "ufr_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ void split3(float v, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)v;
  const float r1 = v - (float)a;
  b = (__bf16)r1;
  c = (__bf16)(r1 - (float)b);
}

// wpk [chunks][9][2][32] float32: wpk[ch][k][o][c] = weight[o][32*ch + c][k] (zero for padding channels).
// Workgroup = an 8 x 32 pixel tile of one image (thread = pixel, all 32 channels of a chunk) x S chunk slices.  Per chunk
// the tile + its 1-pixel halo (10 x 34 pixels, zero outside the frame) is fetched ONCE -- three 16-byte plane reads per
// 8 channels -- summed to float32 and kept in LDS; the nine taps then read LDS.  Global traffic = the planes once
// (x 1.33 for the halo) instead of nine times through L1/L2.  Slices are added through LDS in ascending order.
constexpr int PF_TH = 8, PF_TW = 32, PF_HH = PF_TH + 2, PF_HW = PF_TW + 2, PF_CS = 36;   // channel stride 36 floats: conflict-free taps
__global__ __launch_bounds__(1024) void flow_head_planes_fwd(const __bf16* __restrict__ x, long plane_stride, int chunk0,
                                                             int chunks, const float* __restrict__ wpk,
                                                             const float* __restrict__ bias, float* __restrict__ out, int B,
                                                             int H, int W, int S) {
  extern __shared__ __attribute__((aligned(16))) float lds_pf[];   // [S][PF_HH*PF_HW][PF_CS] tiles | [S][576] weights | [S][256][2]
  const long M = (long)B * H * W;
  const int tid = threadIdx.x, slice = tid >> 8, t = tid & 255;
  float* tile = lds_pf + (long)slice * (PF_HH * PF_HW * PF_CS);
  float* part = lds_pf + (long)S * (PF_HH * PF_HW * PF_CS + 576);
  const int tiles_x = (W + PF_TW - 1) / PF_TW, tiles_y = (H + PF_TH - 1) / PF_TH;
  const int b = blockIdx.x / (tiles_x * tiles_y), tr = blockIdx.x - b * tiles_x * tiles_y;
  const int y0 = (tr / tiles_x) * PF_TH, x0 = (tr % tiles_x) * PF_TW;
  const int ly = t / PF_TW, lx = t - ly * PF_TW;                     // this thread's pixel inside the tile
  const int per = (chunks + S - 1) / S, c_lo = slice * per, c_hi = min(chunks, c_lo + per);
  float a0 = 0.f, a1 = 0.f;
  for (int ch = c_lo; ch < c_lo + per; ++ch) {                       // uniform trip count: the barriers stay aligned
    const bool work = ch < c_hi;
    __syncthreads();                                                  // previous chunk's taps are done
    if (work) {
      // stage: 340 halo pixels x 4 pieces of 8 channels = 1360 items over 256 threads
      for (int it = t; it < PF_HH * PF_HW * 4; it += 256) {
        const int hp = it >> 2, q = it & 3, hy = hp / PF_HW, hx = hp - hy * PF_HW;
        const int yy = y0 + hy - 1, xx = x0 + hx - 1;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
          const __bf16* src = x + (((long)(chunk0 + ch) * M) + ((long)b * H + yy) * W + xx) * 32 + q * 8;
          const bf16x8 p0 = *reinterpret_cast<const bf16x8*>(src);
          const bf16x8 p1 = *reinterpret_cast<const bf16x8*>(src + plane_stride);
          const bf16x8 p2 = *reinterpret_cast<const bf16x8*>(src + 2 * plane_stride);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = ((float)p0[j] + (float)p1[j]) + (float)p2[j];
        }
        float4* dst = reinterpret_cast<float4*>(tile + hp * PF_CS + q * 8);
        dst[0] = make_float4(v[0], v[1], v[2], v[3]);
        dst[1] = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
    __syncthreads();
    if (work) {
      // the chunk's weights are wave-uniform (thread = pixel): read through the scalar unit, FMAs take them as SGPR operands
      const float* wg = wpk + (long)ch * 576;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float* px = tile + ((ly + k / 3) * PF_HW + lx + k % 3) * PF_CS;
        const float* w0 = wg + (k * 2 + 0) * 32;
        const float* w1 = wg + (k * 2 + 1) * 32;
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
          const float4 v = *reinterpret_cast<const float4*>(px + c4 * 4);
          a0 = fmaf(v.x, w0[c4 * 4 + 0], a0); a0 = fmaf(v.y, w0[c4 * 4 + 1], a0); a0 = fmaf(v.z, w0[c4 * 4 + 2], a0); a0 = fmaf(v.w, w0[c4 * 4 + 3], a0);
          a1 = fmaf(v.x, w1[c4 * 4 + 0], a1); a1 = fmaf(v.y, w1[c4 * 4 + 1], a1); a1 = fmaf(v.z, w1[c4 * 4 + 2], a1); a1 = fmaf(v.w, w1[c4 * 4 + 3], a1);
        }
      }
    }
  }
  part[(slice * 256 + t) * 2 + 0] = a0;
  part[(slice * 256 + t) * 2 + 1] = a1;
  __syncthreads();
  const int yy = y0 + ly, xx = x0 + lx;
  if (slice == 0 && yy < H && xx < W) {
    float r0 = 0.f, r1 = 0.f;
    for (int sl = 0; sl < S; ++sl) {
      r0 += part[(sl * 256 + t) * 2 + 0];
      r1 += part[(sl * 256 + t) * 2 + 1];
    }
    const long HW = (long)H * W, p = (long)yy * W + xx;
    out[((long)b * 2 + 0) * HW + p] = r0 + bias[0];
    out[((long)b * 2 + 1) * HW + p] = r1 + bias[1];
  }
}

// Small grids (1/16 resolution and coarser): thread = (pixel, 8-channel group) x S chunk slices, taps straight from L1 / L2,
// the layer's weights in LDS; slices added through LDS in ascending order.
__global__ __launch_bounds__(1024) void flow_head_planes_fwd_small(const __bf16* __restrict__ x, long plane_stride, int chunk0,
                                                             int chunks, const float* __restrict__ wpk,
                                                             const float* __restrict__ bias, float* __restrict__ out, int B,
                                                             int H, int W, int S) {
  extern __shared__ __attribute__((aligned(16))) float lds_w[];          // [chunks][9][2][32], then [S][64][2] partials
  const long M = (long)B * H * W;
  const int tid = threadIdx.x, slice = tid >> 8, t = tid & 255;
  for (int i = tid; i < chunks * 576 / 4; i += blockDim.x)
    reinterpret_cast<float4*>(lds_w)[i] = reinterpret_cast<const float4*>(wpk)[i];
  __syncthreads();
  const long pix = (long)blockIdx.x * 64 + (t >> 2);
  const int q = t & 3;
  const bool live = pix < M;
  const long pp = live ? pix : 0;
  const int xx = (int)(pp % W), yy = (int)((pp / W) % H);
  int off[9];                                         // neighbour offsets in elements, relative to the pixel's own chunk row
  float msk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {                       // clamped neighbour + validity factor: no branch in the loop
    const int y2 = yy + k / 3 - 1, x2 = xx + k % 3 - 1;
    const bool ok = y2 >= 0 && y2 < H && x2 >= 0 && x2 < W;
    off[k] = ok ? ((k / 3 - 1) * W + (k % 3 - 1)) * 32 : 0;
    msk[k] = ok ? 1.f : 0.f;
  }
  float a0 = 0.f, a1 = 0.f;
  const int per = (chunks + S - 1) / S, c_lo = slice * per, c_hi = min(chunks, c_lo + per);
  for (int ch = c_lo; ch < c_hi; ++ch) {
    const __bf16* xc = x + ((long)(chunk0 + ch) * M + pp) * 32 + q * 8;
    const float* wc = lds_w + ch * 576 + q * 8;
#pragma unroll 3
    for (int k = 0; k < 9; ++k) {
      const __bf16* src = xc + off[k];
      const bf16x8 p0 = *reinterpret_cast<const bf16x8*>(src);
      const bf16x8 p1 = *reinterpret_cast<const bf16x8*>(src + plane_stride);
      const bf16x8 p2 = *reinterpret_cast<const bf16x8*>(src + 2 * plane_stride);
      const float4 w0a = *reinterpret_cast<const float4*>(wc + (k * 2 + 0) * 32), w0b = *reinterpret_cast<const float4*>(wc + (k * 2 + 0) * 32 + 4);
      const float4 w1a = *reinterpret_cast<const float4*>(wc + (k * 2 + 1) * 32), w1b = *reinterpret_cast<const float4*>(wc + (k * 2 + 1) * 32 + 4);
      const float w0[8] = {w0a.x, w0a.y, w0a.z, w0a.w, w0b.x, w0b.y, w0b.z, w0b.w};
      const float w1[8] = {w1a.x, w1a.y, w1a.z, w1a.w, w1b.x, w1b.y, w1b.z, w1b.w};
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = ((float)p0[j] + (float)p1[j]) + (float)p2[j];
        s0 = fmaf(v, w0[j], s0);
        s1 = fmaf(v, w1[j], s1);
      }
      a0 = fmaf(s0, msk[k], a0);
      a1 = fmaf(s1, msk[k], a1);
    }
  }
  a0 += __shfl_xor(a0, 1, 64); a1 += __shfl_xor(a1, 1, 64);
  a0 += __shfl_xor(a0, 2, 64); a1 += __shfl_xor(a1, 2, 64);
  float* part = lds_w + chunks * 576;                   // [S][64][2]
  if (q == 0) {
    part[(slice * 64 + (t >> 2)) * 2 + 0] = a0;
    part[(slice * 64 + (t >> 2)) * 2 + 1] = a1;
  }
  __syncthreads();
  if (slice == 0 && q == 0 && live) {
    float r0 = 0.f, r1 = 0.f;
    for (int sl = 0; sl < S; ++sl) {
      r0 += part[(sl * 64 + (t >> 2)) * 2 + 0];
      r1 += part[(sl * 64 + (t >> 2)) * 2 + 1];
    }
    const long HW = (long)H * W, b = pix / HW, p = pix - b * HW;
    out[(b * 2 + 0) * HW + p] = r0 + bias[0];
    out[(b * 2 + 1) * HW + p] = r1 + bias[1];
  }
}

// ---- predict_flow* on the matrix cores ------------------------------------------------------------------------------
// out[o, y, x] = bias[o] + sum_k sum_c x[c, y + ky, x + kx] w[o, c, k] is computed as a GEMM per PIXEL followed by a
// 9-tap gather:  T[p, n = 2k + o] = sum_c x[p, c] w[o, c, k]   (M = pixels of the tile + halo, N = 18 padded to 32, K = C)
//                out[o, p]      = bias[o] + sum_k T[p + tap k, 2k + o]
// The A operand is the engine's plane layout as it lies in HBM (a lane's fragment = 8 channels of one pixel = one 16-byte
// load per plane, no LDS staging); float32 = six bf16 products (csrc/igemm.hip); T goes through LDS once.  Workgroup =
// TH x 32 pixels (+ halo) x S chunk slices; wave = (slice, tile lane): m-tiles tw, tw + 4, ... of 16 halo pixels.
// wmf: bf16 [chunks][3 planes][2 n-tiles][16][32]: plane p of w[o][32 ch + c][k] at n = 2k + o, zero for n >= 18.
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ constexpr int PFM_A[6] = {2, 0, 1, 1, 0, 0};
__device__ constexpr int PFM_B[6] = {0, 2, 1, 0, 1, 0};
__device__ __attribute__((aligned(64))) unsigned pf_zero_page[16];
constexpr int PFM_MAXT = 6;                        // m-tiles per wave

// MODE 0: predict_flow (3 x 3, stride 1, pad 1): tile = TH x 32 outputs, staged pixels = the tile + halo; N = 18 of 32;
//         out = NCHW [B, 2, H, W] + bias.
// MODE 1: the two flow channels of a ConvTranspose2d(4, 2, 1) data gradient (the last two input channels of deconvK --
//         the upsampled flow; the other channels go through ufr_igemm with an N that is a multiple of 128):
//         g[o, y, x] = sum_{ky,kx} sum_c gz[c, 2y - 1 + ky, 2x - 1 + kx] w[o, c, ky, kx];  tile = TH x 16 coarse outputs,
//         staged pixels = the (2 TH + 2) x 34 fine pixels under it; N = 32; out = lanes 0-1 of chunk `out_chunk` of the
//         coarse grid's float32 gradient sum.  H, W = the OUTPUT grid in modes 0 and 1.
// MODE 2: ConvTranspose2d(C, 2, 4, 2, 1) forward (PWC-Net's `upfeat*`, models/PWCNet.py:115-143): per COARSE pixel
//         T[p, n = 2 (4 ky + kx) + o] = sum_c x[p, c] w[c, o, ky, kx] (N = 32), then fine pixel (Y, X) adds its <= 4 (pixel, tap)
//         pairs: ky = (Y + 1) mod 2 (+ 2), y = (Y + 1 - ky) / 2.  tile = TH x 16 coarse pixels, staged = the tile + halo
//         (TH + 2) x 18; out = NCHW [B, 2, 2H, 2W] + bias.  H, W = the COARSE (input) grid.
// WPS_ = waves per chunk slice.  4 (the default): the slice's m-tiles are dealt to four waves, each of which fetches the slice's
// 6 KB weight image per chunk.  1 (the small grids, <= 12 m-tiles): ONE wave holds all m-tiles of a slice and eight slices share the
// workgroup -- the weights are fetched once per chunk and workgroup instead of four times (a workgroup of these grids is bound by
// its CU's vector-memory port: 51 -> 33 KB per chunk, profiles/r4_predict_flow_small_grids.jsonl)
template <int MODE, int WPS_ = 4, int MAXT_ = PFM_MAXT>
__global__ __launch_bounds__(512) void flow_head_planes_fwd_mfma(const __bf16* __restrict__ x, long plane_stride, int chunk0,
                                                                 int chunks, const __bf16* __restrict__ wmf,
                                                                 const float* __restrict__ bias, float* __restrict__ out,
                                                                 int out_chunk, int B, int H, int W, int TH, int S) {
  constexpr int TW = MODE == 0 ? 32 : 16, TS = MODE == 0 ? 20 : 36;        // T row stride (floats): 18 / 32 used
  constexpr int PFM_HW = MODE == 2 ? 18 : 34;                              // pixels across a staged tile (tile + halo, or 2 x 16 + 2)
  extern __shared__ __attribute__((aligned(16))) float pfm_T[];            // [S][mt * 16][TS]
  const int Hs = MODE == 1 ? 2 * H : H, Ws = MODE == 1 ? 2 * W : W;        // the staged (input) grid
  const long M = (long)B * Hs * Ws;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, slice = wave / WPS_, tw = wave % WPS_;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int b = blockIdx.x / (tiles_x * tiles_y), tr = blockIdx.x - b * tiles_x * tiles_y;
  const int y0 = (tr / tiles_x) * TH, x0 = (tr % tiles_x) * TW;
  const int sy0 = MODE == 1 ? 2 * y0 - 1 : y0 - 1, sx0 = MODE == 1 ? 2 * x0 - 1 : x0 - 1;   // first staged pixel
  const int nh = (MODE == 1 ? 2 * TH + 2 : TH + 2) * PFM_HW, mt = (nh + 15) >> 4;
  const __bf16* zero = reinterpret_cast<const __bf16*>(pf_zero_page);
  // this lane's staged pixel in each of its m-tiles
  const __bf16* abase[MAXT_];
  unsigned okmask = 0;
#pragma unroll
  for (int s = 0; s < MAXT_; ++s) {
    const int hp = (tw + WPS_ * s) * 16 + (lane & 15);
    const int hy = hp / PFM_HW, hx = hp - hy * PFM_HW;
    const int yy = sy0 + hy, xx = sx0 + hx;
    const bool ok = tw + WPS_ * s < mt && hp < nh && yy >= 0 && yy < Hs && xx >= 0 && xx < Ws;
    abase[s] = ok ? x + ((long)chunk0 * M + ((long)b * Hs + yy) * Ws + xx) * 32 + (lane >> 4) * 8 : zero;
    okmask |= ok ? 1u << s : 0u;
  }
  f32x4 acc[MAXT_][2];
#pragma unroll
  for (int s = 0; s < MAXT_; ++s) acc[s][0] = acc[s][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int per = (chunks + S - 1) / S, c_lo = slice * per, c_hi = min(chunks, c_lo + per);
  const __bf16* wl = wmf + (lane & 15) * 32 + (lane >> 4) * 8;
  for (int ch = c_lo; ch < c_hi; ++ch) {
    bf16x8 fb[2][3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) fb[nt][p] = *reinterpret_cast<const bf16x8*>(wl + (((long)ch * 3 + p) * 2 + nt) * 512);
#pragma unroll
    for (int h = 0; h < MAXT_; h += 6) {         // six m-tiles' fragments at a time (WPS_ = 1: twelve tiles in two halves)
      bf16x8 fa[6][3];
#pragma unroll
      for (int s = 0; s < 6; ++s)
        if (tw + WPS_ * (h + s) < mt) {
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const long off = (okmask >> (h + s) & 1u) ? (long)ch * M * 32 + p * plane_stride : 0;   // the zero page does not move
            fa[s][p] = *reinterpret_cast<const bf16x8*>(abase[h + s] + off);
          }
        }
#pragma unroll
      for (int s = 0; s < 6; ++s)
        if (tw + WPS_ * (h + s) < mt) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 6; ++q)
              acc[h + s][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[s][PFM_A[q]], fb[nt][PFM_B[q]], acc[h + s][nt], 0, 0, 0);
        }
    }
  }
  // T: lane holds D[m = 4 (lane >> 4) + r][n = lane & 15]
  float* T = pfm_T + (long)slice * mt * 16 * TS;
#pragma unroll
  for (int s = 0; s < MAXT_; ++s)
    if (tw + WPS_ * s < mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* row = T + ((tw + WPS_ * s) * 16 + (lane >> 4) * 4 + r) * TS;
        row[lane & 15] = acc[s][0][r];
        if (MODE != 0 || (lane & 15) < 2) row[16 + (lane & 15)] = acc[s][1][r];
      }
    }
  __syncthreads();
  if constexpr (MODE == 2) {
    const int Ho = 2 * H, Wo = 2 * W;
    for (int idx = tid; idx < 2 * TH * 32; idx += blockDim.x) {           // the tile's fine pixels
      const int ly = idx >> 5, lx = idx & 31;
      const int Y = 2 * y0 + ly, X = 2 * x0 + lx;
      if (Y >= Ho || X >= Wo) continue;
      float r0 = 0.f, r1 = 0.f;
      for (int sl = 0; sl < S; ++sl) {
        const float* Ts = pfm_T + (long)sl * mt * 16 * TS;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int ky = ((Y + 1) & 1) + 2 * t, hy = (Y + 1 - ky) / 2 - sy0;       // (Y + 1 - ky is even, >= -2: halo row 0)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int kx = ((X + 1) & 1) + 2 * u, hx = (X + 1 - kx) / 2 - sx0;
            const float* e = Ts + (hy * PFM_HW + hx) * TS + 2 * (ky * 4 + kx);
            r0 += e[0];
            r1 += e[1];
          }
        }
      }
      const long HWo = (long)Ho * Wo, p = (long)Y * Wo + X;
      out[((long)b * 2 + 0) * HWo + p] = r0 + bias[0];
      out[((long)b * 2 + 1) * HWo + p] = r1 + bias[1];
    }
    return;
  }
  if (tid < TH * TW) {
    const int ly = tid / TW, lx = tid - ly * TW;
    const int yy = y0 + ly, xx = x0 + lx;
    if (yy < H && xx < W) {
      float r0 = 0.f, r1 = 0.f;
      for (int sl = 0; sl < S; ++sl) {               // slices, then taps, in ascending order
        const float* Ts = pfm_T + (long)sl * mt * 16 * TS;
        if (MODE == 0) {
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            const float* e = Ts + ((ly + k / 3) * PFM_HW + lx + k % 3) * TS + 2 * k;
            r0 += e[0];
            r1 += e[1];
          }
        } else {
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const float* e = Ts + ((2 * ly + k / 4) * PFM_HW + 2 * lx + k % 4) * TS + 2 * k;
            r0 += e[0];
            r1 += e[1];
          }
        }
      }
      if (MODE == 0) {
        const long HW = (long)H * W, p = (long)yy * W + xx;
        out[((long)b * 2 + 0) * HW + p] = r0 + bias[0];
        out[((long)b * 2 + 1) * HW + p] = r1 + bias[1];
      } else {
        float* o = out + ((long)out_chunk * B * H * W + ((long)b * H + yy) * W + xx) * 32;
        o[0] = r0;
        o[1] = r1;
      }
    }
  }
}

// G[chunk0 + ch][pix][c] (+)= sum_o sum_k gy[b, o, pix - (k - centre)] * weight[o][32*ch + c][k]
// thread = (pixel, 8-channel group), blockIdx.y = slice of the chunks (independent outputs: no reduction)
// Optional fused finalisation (round 4): for the chunks [fin_c0, fin_c0 + fin_n) of the tensor -- the segment a deconvolution
// produced, whose gradient is COMPLETE once this kernel has added its share -- the sum x LeakyReLU'(mask) also leaves as the
// three gradient planes the segment's transposed GEMM reads (what ufr_grad_finalize did in a launch of its own).
struct PfFinalize {
  const __bf16* mask; __bf16* out; long out_plane_stride; int c0, n; float slope;
};

__global__ __launch_bounds__(256) void flow_head_planes_bwd(const float* __restrict__ gy, const float* __restrict__ wpk,
                                                            float* __restrict__ G, int chunk0, int chunks, int B, int H, int W,
                                                            int accumulate, int per, const PfFinalize fin) {
  extern __shared__ __attribute__((aligned(16))) float lds_w[];          // this slice's [per][9][2][32]
  const long M = (long)B * H * W;
  const int c_lo = blockIdx.y * per, c_hi = min(chunks, c_lo + per);
  for (int i = threadIdx.x; i < (c_hi - c_lo) * 576 / 4; i += blockDim.x)
    reinterpret_cast<float4*>(lds_w)[i] = reinterpret_cast<const float4*>(wpk + (long)c_lo * 576)[i];
  __syncthreads();
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long pix = t >> 2;
  const int q = (int)(t & 3);
  if (pix >= M) return;
  const long HW = (long)H * W, b = pix / HW, p = pix - b * HW;
  const int yy = (int)(p / W), xx = (int)(p - (long)yy * W);
  float g0[9], g1[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {                       // the output pixel that used tap k on this input pixel
    const int y2 = yy - (k / 3 - 1), x2 = xx - (k % 3 - 1);
    const bool ok = y2 >= 0 && y2 < H && x2 >= 0 && x2 < W;
    g0[k] = ok ? gy[(b * 2 + 0) * HW + (long)y2 * W + x2] : 0.f;
    g1[k] = ok ? gy[(b * 2 + 1) * HW + (long)y2 * W + x2] : 0.f;
  }
  for (int ch = c_lo; ch < c_hi; ++ch) {
    const float* wc = lds_w + (ch - c_lo) * 576 + q * 8;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float4 w0a = *reinterpret_cast<const float4*>(wc + (k * 2 + 0) * 32), w0b = *reinterpret_cast<const float4*>(wc + (k * 2 + 0) * 32 + 4);
      const float w0[8] = {w0a.x, w0a.y, w0a.z, w0a.w, w0b.x, w0b.y, w0b.z, w0b.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = fmaf(g0[k], w0[j], a[j]);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float4 w1a = *reinterpret_cast<const float4*>(wc + (k * 2 + 1) * 32), w1b = *reinterpret_cast<const float4*>(wc + (k * 2 + 1) * 32 + 4);
      const float w1[8] = {w1a.x, w1a.y, w1a.z, w1a.w, w1b.x, w1b.y, w1b.z, w1b.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = fmaf(g1[k], w1[j], a[j]);
    }
    float4* dst = reinterpret_cast<float4*>(G + ((long)(chunk0 + ch) * M + pix) * 32 + q * 8);
    float4 lo = make_float4(a[0], a[1], a[2], a[3]), hi = make_float4(a[4], a[5], a[6], a[7]);
    if (accumulate) {
      const float4 o0 = dst[0], o1 = dst[1];
      lo.x += o0.x; lo.y += o0.y; lo.z += o0.z; lo.w += o0.w;
      hi.x += o1.x; hi.y += o1.y; hi.z += o1.z; hi.w += o1.w;
    }
    dst[0] = lo;
    dst[1] = hi;
    if (fin.out && ch >= fin.c0 && ch < fin.c0 + fin.n) {
      const long e = ((long)(chunk0 + ch) * M + pix) * 32 + q * 8;
      float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      if (fin.mask) {
        const bf16x8 m = *reinterpret_cast<const bf16x8*>(fin.mask + e);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ((float)m[j] > 0.f) ? v[j] : v[j] * fin.slope;
      }
      bf16x8 q0, q1, q2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        __bf16 x, y, z;
        split3(v[j], x, y, z);
        q0[j] = x; q1[j] = y; q2[j] = z;
      }
      *reinterpret_cast<bf16x8*>(fin.out + e) = q0;
      *reinterpret_cast<bf16x8*>(fin.out + e + fin.out_plane_stride) = q1;
      *reinterpret_cast<bf16x8*>(fin.out + e + 2 * fin.out_plane_stride) = q2;
    }
  }
}

// Data gradient of ConvTranspose2d(C, 2, 4, 2, 1) (PWC-Net's `upfeat*`): gy [B, 2, 2H, 2W] -> the coarse grid's gradient sum
// G[chunk0 + ch][pix][c] (+)= sum_o sum_{ky,kx} gy[b, o, 2y - 1 + ky, 2x - 1 + kx] * w[32 ch + c][o][ky][kx]
// wpk [chunks][16][2][32]; thread = (coarse pixel, 8-channel group), blockIdx.y = slice of the chunks.
__global__ __launch_bounds__(256) void flow_tail_planes_bwd(const float* __restrict__ gy, const float* __restrict__ wpk,
                                                            float* __restrict__ G, int chunk0, int chunks, int B, int H, int W,
                                                            int accumulate, int per) {
  extern __shared__ __attribute__((aligned(16))) float lds_w[];          // this slice's [per][16][2][32]
  const long M = (long)B * H * W;
  const int c_lo = blockIdx.y * per, c_hi = min(chunks, c_lo + per);
  for (int i = threadIdx.x; i < (c_hi - c_lo) * 1024 / 4; i += blockDim.x)
    reinterpret_cast<float4*>(lds_w)[i] = reinterpret_cast<const float4*>(wpk + (long)c_lo * 1024)[i];
  __syncthreads();
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long pix = t >> 2;
  const int q = (int)(t & 3);
  if (pix >= M) return;
  const long HW = (long)H * W, b = pix / HW, p = pix - b * HW;
  const int yy = (int)(p / W), xx = (int)(p - (long)yy * W);
  const int Ho = 2 * H, Wo = 2 * W;
  const long HWo = (long)Ho * Wo;
  float g0[16], g1[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int Y = 2 * yy - 1 + (k >> 2), X = 2 * xx - 1 + (k & 3);
    const bool ok = Y >= 0 && Y < Ho && X >= 0 && X < Wo;
    g0[k] = ok ? gy[(b * 2 + 0) * HWo + (long)Y * Wo + X] : 0.f;
    g1[k] = ok ? gy[(b * 2 + 1) * HWo + (long)Y * Wo + X] : 0.f;
  }
  for (int ch = c_lo; ch < c_hi; ++ch) {
    const float* wc = lds_w + (ch - c_lo) * 1024 + q * 8;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float4 w0a = *reinterpret_cast<const float4*>(wc + (k * 2 + 0) * 32), w0b = *reinterpret_cast<const float4*>(wc + (k * 2 + 0) * 32 + 4);
      const float4 w1a = *reinterpret_cast<const float4*>(wc + (k * 2 + 1) * 32), w1b = *reinterpret_cast<const float4*>(wc + (k * 2 + 1) * 32 + 4);
      const float w0[8] = {w0a.x, w0a.y, w0a.z, w0a.w, w0b.x, w0b.y, w0b.z, w0b.w};
      const float w1[8] = {w1a.x, w1a.y, w1a.z, w1a.w, w1b.x, w1b.y, w1b.z, w1b.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = fmaf(g1[k], w1[j], fmaf(g0[k], w0[j], a[j]));
    }
    float4* dst = reinterpret_cast<float4*>(G + ((long)(chunk0 + ch) * M + pix) * 32 + q * 8);
    float4 lo = make_float4(a[0], a[1], a[2], a[3]), hi = make_float4(a[4], a[5], a[6], a[7]);
    if (accumulate) {
      const float4 o0 = dst[0], o1 = dst[1];
      lo.x += o0.x; lo.y += o0.y; lo.z += o0.z; lo.w += o0.w;
      hi.x += o1.x; hi.y += o1.y; hi.z += o1.z; hi.w += o1.w;
    }
    dst[0] = lo;
    dst[1] = hi;
  }
}

// y[b,o,Y,X] = bias[o] + sum_i sum_{ky,kx} x[b,i,(Y+1-ky)/2,(X+1-kx)/2] * w[i,o,ky,kx]  -> chunk `chunk` of `planes`
// (the same arithmetic, operation for operation, as small_cout.hip's deconv4x4s2_c2_fwd); optionally also NCHW fp32
__global__ void flow_up_planes_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                   __bf16* __restrict__ planes, long plane_stride, int chunk, int B, int H, int W, int has_bias) {
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
    __bf16 s0[3], s1[3];
    split3(a0, s0[0], s0[1], s0[2]);
    split3(a1, s1[0], s1[1], s1[2]);
    __bf16* dst = planes + ((long)chunk * total + i) * 32;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      bf16x8 v = {s0[p], s1[p], (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
      const bf16x8 zero = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
      bf16x8* o = reinterpret_cast<bf16x8*>(dst + p * plane_stride);
      o[0] = v; o[1] = zero; o[2] = zero; o[3] = zero;
    }
  }
}

// gx[b,i,y,x] = sum_o sum_{ky,kx} G[chunk][(b,2y-1+ky,2x-1+kx)][o] * w[i,o,ky,kx]
__global__ void flow_up_planes_bwd(const float* __restrict__ G, int chunk, const float* __restrict__ w, float* __restrict__ gx,
                                   int B, int H, int W, int accumulate) {
  const int Ho = 2 * H, Wo = 2 * W;
  const long total = (long)B * H * W, Mf = (long)B * Ho * Wo;
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
        const float2 g = *reinterpret_cast<const float2*>(G + ((long)chunk * Mf + ((long)b * Ho + Y) * Wo + X) * 32);
        a0 = fmaf(g.x, w[((0 * 2 + 0) * 4 + ky) * 4 + kx], a0);
        a1 = fmaf(g.x, w[((1 * 2 + 0) * 4 + ky) * 4 + kx], a1);
        a0 = fmaf(g.y, w[((0 * 2 + 1) * 4 + ky) * 4 + kx], a0);
        a1 = fmaf(g.y, w[((1 * 2 + 1) * 4 + ky) * 4 + kx], a1);
      }
    }
    float* o0 = gx + ((size_t)b * 2 + 0) * H * W + (size_t)yy * W + xx;
    float* o1 = gx + ((size_t)b * 2 + 1) * H * W + (size_t)yy * W + xx;
    *o0 = accumulate ? *o0 + a0 : a0;
    *o1 = accumulate ? *o1 + a1 : a1;
  }
}

// ABI 7: a chunk range [chunk0, chunk0 + chunks) must stay inside what it walks.  A planes operand whose planes lie
// plane_stride elements apart holds plane_stride / (pixels * 32) chunks per plane; the packed weights hold w_chunks (indexed by the
// position inside the range); a float32 gradient sum holds g_chunks.  (tools/bench_pf.py launched 16 chunks on a 13-chunk buffer in
// round 4 and the kernel read 3 x 8 x 48 x 160 x 32 bf16 per plane past the end: a GPU memory-access fault)
bool planes_hold(long plane_stride, long pixels, int chunk0, int chunks) {
  return plane_stride > 0 && (long)(chunk0 + chunks) * pixels * 32 <= plane_stride;
}
#define UFR_REQUIRE_RANGE(what, plane_stride, pixels, chunk0, chunks, w_chunks)                                                       \
  do {                                                                                                                                \
    if (!planes_hold(plane_stride, pixels, chunk0, chunks))                                                                           \
      return ufr::fail(UFR_EINVAL, "%s: chunks [%d, %d) leave the planes operand (%ld chunks per plane)", what, chunk0,               \
                       chunk0 + chunks, (long)(plane_stride) / ((long)(pixels) * 32));                                               \
    if ((chunks) > (w_chunks))                                                                                                        \
      return ufr::fail(UFR_EINVAL, "%s: %d chunks asked of packed weights that hold %d", what, chunks, w_chunks);                     \
  } while (0)
#define UFR_REQUIRE_GRANGE(what, chunk0, chunks, w_chunks, g_chunks)                                                                  \
  do {                                                                                                                                \
    if ((chunk0) + (chunks) > (g_chunks))                                                                                             \
      return ufr::fail(UFR_EINVAL, "%s: chunks [%d, %d) leave the gradient sum (%d chunks)", what, chunk0, chunk0 + chunks, g_chunks); \
    if ((chunks) > (w_chunks))                                                                                                        \
      return ufr::fail(UFR_EINVAL, "%s: %d chunks asked of packed weights that hold %d", what, chunks, w_chunks);                     \
  } while (0)

}  // namespace

extern "C" int ufr_flow_head_planes_forward(const void* planes, long plane_stride, int chunk0, int chunks, const float* wpk, int w_chunks,
                                            const float* bias, float* out, int B, int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(planes && wpk && bias && out, "flow head (planes) forward: null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && chunks > 0 && chunks <= 48 && chunk0 >= 0 && (long)B * H * W < (1L << 29),
              "flow head (planes) forward: bad shape");
  UFR_REQUIRE_RANGE("flow head (planes) forward", plane_stride, (long)B * H * W, chunk0, chunks, w_chunks);
  const int blocks = B * ufr::ceil_div(H, PF_TH) * ufr::ceil_div(W, PF_TW);
  hipStream_t st = ufr::as_stream(stream);
  (void)ufr::ensure_dynamic_lds(reinterpret_cast<const void*>(flow_head_planes_fwd), (2 * (PF_HH * PF_HW * PF_CS + 576) + 2 * 256 * 2) * 4);
  (void)ufr::ensure_dynamic_lds(reinterpret_cast<const void*>(flow_head_planes_fwd_small), 48 * 576 * 4 + 4 * 64 * 2 * 4);
  if (blocks >= 192) {                        // big grids: LDS-tiled kernel, the planes are fetched once
    const int S = (blocks < 512 && chunks >= 8) ? 2 : 1;
    const size_t lds = ((size_t)S * (PF_HH * PF_HW * PF_CS + 576) + (size_t)S * 256 * 2) * 4;
    flow_head_planes_fwd<<<blocks, 256 * S, lds, st>>>(static_cast<const __bf16*>(planes), plane_stride, chunk0, chunks, wpk, bias,
                                                       out, B, H, W, S);
    return ufr::launched("flow_head_planes_fwd");
  }
  const long M = (long)B * H * W;
  const int sblocks = ufr::ceil_div(M, 64);
  int S = 1;                                  // chunk slices per workgroup: more threads on the small grids
  while (S < 4 && (long)sblocks * S * 2 <= 2048 && chunks >= 4 * S) S *= 2;
  const size_t lds = (size_t)chunks * 576 * 4 + (size_t)S * 64 * 2 * 4;
  flow_head_planes_fwd_small<<<sblocks, 256 * S, lds, st>>>(static_cast<const __bf16*>(planes), plane_stride, chunk0, chunks, wpk,
                                                           bias, out, B, H, W, S);
  return ufr::launched("flow_head_planes_fwd_small");
}

namespace {
template <int MODE>
int launch_pf_mfma(const void* planes, long plane_stride, int chunk0, int chunks, const void* wmf, const float* bias, float* out,
                   int out_chunk, int B, int H, int W, hipStream_t st, const char* what) {
  constexpr int TW = MODE == 0 ? 32 : 16, TS = MODE == 0 ? 20 : 36, HWs = MODE == 2 ? 18 : 34;
  auto blocks_of = [&](int th) { return B * ufr::ceil_div(H, th) * ufr::ceil_div(W, TW); };
  int TH;                                       // MODE 0: 10 / 6 / 4 staged rows; MODE 1: 2 TH + 2 <= 10; MODE 2: (TH + 2) x 18 pixels
  if (MODE == 0) TH = blocks_of(8) >= 512 ? 8 : (blocks_of(4) >= 256 ? 4 : 2);
  else if (MODE == 1) TH = blocks_of(4) >= 256 ? 4 : 2;
  else TH = blocks_of(16) >= 512 ? 16 : (blocks_of(8) >= 256 ? 8 : 4);      // 18 x 18 = 324 staged pixels: 21 m-tiles of <= 24
  const int blocks = blocks_of(TH);
  const int S = (blocks >= 512 || chunks < 4) ? 1 : 2;
  const int mt = ((MODE == 1 ? 2 * TH + 2 : TH + 2) * HWs + 15) / 16;
  const size_t lds = (size_t)S * mt * 16 * TS * sizeof(float);
  if (lds > 64 * 1024) {
    hipError_t e = ufr::ensure_dynamic_lds(reinterpret_cast<const void*>(flow_head_planes_fwd_mfma<MODE>), lds);
    if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  }
  if constexpr (MODE == 0) if (mt <= 12 && S == 2 && chunks >= 16 && blocks <= 256) {    // small grids (at most one workgroup per CU): eight one-wave slices
    const size_t lds8 = (size_t)8 * mt * 16 * TS * sizeof(float);
    hipError_t e = ufr::ensure_dynamic_lds(reinterpret_cast<const void*>(flow_head_planes_fwd_mfma<MODE, 1, 12>),
                                           8 * 12 * 16 * TS * sizeof(float));
    if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    flow_head_planes_fwd_mfma<MODE, 1, 12><<<blocks, 512, lds8, st>>>(static_cast<const __bf16*>(planes), plane_stride, chunk0, chunks,
                                                                      static_cast<const __bf16*>(wmf), bias, out, out_chunk, B, H, W, TH, 8);
    return ufr::launched(what);
  }
  flow_head_planes_fwd_mfma<MODE><<<blocks, 256 * S, lds, st>>>(static_cast<const __bf16*>(planes), plane_stride, chunk0, chunks,
                                                                static_cast<const __bf16*>(wmf), bias, out, out_chunk, B, H, W, TH, S);
  return ufr::launched(what);
}
}  // namespace

extern "C" int ufr_flow_head_planes_forward_mfma(const void* planes, long plane_stride, int chunk0, int chunks, const void* wmf,
                                                 int w_chunks, const float* bias, float* out, int B, int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(planes && wmf && bias && out, "flow head (planes, mfma) forward: null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && chunks > 0 && chunks <= 48 && chunk0 >= 0 && (long)B * H * W < (1L << 29),
              "flow head (planes, mfma) forward: bad shape");
  UFR_REQUIRE_RANGE("flow head (planes, mfma) forward", plane_stride, (long)B * H * W, chunk0, chunks, w_chunks);
  return launch_pf_mfma<0>(planes, plane_stride, chunk0, chunks, wmf, bias, out, 0, B, H, W, ufr::as_stream(stream),
                           "flow_head_planes_fwd_mfma");
}

extern "C" int ufr_deconv_flow_tail_backward_mfma(const void* grad_planes, long plane_stride, int chunk0, int chunks, const void* wmf,
                                                  int w_chunks, float* G, int g_chunks, int out_chunk, int B, int H, int W,
                                                  ufr_stream_t stream) {
  UFR_REQUIRE(grad_planes && wmf && G, "deconv flow tail backward: null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && chunks > 0 && chunks <= 48 && chunk0 >= 0 && out_chunk >= 0 &&
                  (long)B * H * W < (1L << 27), "deconv flow tail backward: bad shape");
  UFR_REQUIRE_RANGE("deconv flow tail backward", plane_stride, (long)B * H * W * 4, chunk0, chunks, w_chunks);   // planes on the fine grid
  UFR_REQUIRE(out_chunk < g_chunks, "deconv flow tail backward: chunk %d is not in the gradient sum (%d chunks)", out_chunk, g_chunks);
  return launch_pf_mfma<1>(grad_planes, plane_stride, chunk0, chunks, wmf, nullptr, G, out_chunk, B, H, W, ufr::as_stream(stream),
                           "deconv_flow_tail_bwd_mfma");
}

extern "C" int ufr_upfeat_planes_forward_mfma(const void* planes, long plane_stride, int chunk0, int chunks, const void* wmf,
                                              int w_chunks, const float* bias, float* out, int B, int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(planes && wmf && bias && out, "upfeat (planes, mfma) forward: null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && chunks > 0 && chunks <= 48 && chunk0 >= 0 && (long)B * H * W < (1L << 27),
              "upfeat (planes, mfma) forward: bad shape");
  UFR_REQUIRE_RANGE("upfeat (planes, mfma) forward", plane_stride, (long)B * H * W, chunk0, chunks, w_chunks);
  return launch_pf_mfma<2>(planes, plane_stride, chunk0, chunks, wmf, bias, out, 0, B, H, W, ufr::as_stream(stream),
                           "upfeat_planes_fwd_mfma");
}

extern "C" int ufr_upfeat_planes_backward(const float* grad_y, const float* wpk, int w_chunks, float* G, int g_chunks, int chunk0,
                                          int chunks, int B, int H, int W, int accumulate, ufr_stream_t stream) {
  UFR_REQUIRE(grad_y && wpk && G, "upfeat (planes) backward: null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && chunks > 0 && chunks <= 48 && chunk0 >= 0 && (long)B * H * W < (1L << 27),
              "upfeat (planes) backward: bad shape");
  UFR_REQUIRE_GRANGE("upfeat (planes) backward", chunk0, chunks, w_chunks, g_chunks);
  const long threads = (long)B * H * W * 4;
  const int bx = ufr::ceil_div(threads, 256);
  int slices = 1;
  while ((long)bx * slices < 1024 && slices < chunks) slices *= 2;
  if (slices > chunks) slices = chunks;
  int per = ufr::ceil_div(chunks, slices);
  if (per > 12) per = 12;                       // 12 x 4 KB of weights per slice in LDS
  slices = ufr::ceil_div(chunks, per);
  flow_tail_planes_bwd<<<dim3(bx, slices), 256, (size_t)per * 1024 * 4, ufr::as_stream(stream)>>>(grad_y, wpk, G, chunk0, chunks, B,
                                                                                                 H, W, accumulate, per);
  return ufr::launched("flow_tail_planes_bwd");
}

namespace {
int launch_pf_bwd(const float* grad_y, const float* wpk, float* G, int chunk0, int chunks, int B, int H, int W, int accumulate,
                  const PfFinalize& fin, hipStream_t st) {
  const long threads = (long)B * H * W * 4;
  const int bx = ufr::ceil_div(threads, 256);
  int slices = 1;                             // chunk slices over blockIdx.y: more workgroups on the small grids
  while ((long)bx * slices < 1024 && slices < chunks) slices *= 2;
  if (slices > chunks) slices = chunks;
  const int per = ufr::ceil_div(chunks, slices);
  slices = ufr::ceil_div(chunks, per);
  (void)ufr::ensure_dynamic_lds(reinterpret_cast<const void*>(flow_head_planes_bwd), 48 * 576 * 4);
  flow_head_planes_bwd<<<dim3(bx, slices), 256, (size_t)per * 576 * 4, st>>>(grad_y, wpk, G, chunk0, chunks, B, H, W, accumulate, per, fin);
  return ufr::launched("flow_head_planes_bwd");
}
}  // namespace

extern "C" int ufr_flow_head_planes_backward(const float* grad_y, const float* wpk, int w_chunks, float* G, int g_chunks, int chunk0,
                                             int chunks, int B, int H, int W, int accumulate, ufr_stream_t stream) {
  UFR_REQUIRE(grad_y && wpk && G, "flow head (planes) backward: null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && chunks > 0 && chunks <= 48 && chunk0 >= 0 && (long)B * H * W < (1L << 29),
              "flow head (planes) backward: bad shape");
  UFR_REQUIRE_GRANGE("flow head (planes) backward", chunk0, chunks, w_chunks, g_chunks);
  return launch_pf_bwd(grad_y, wpk, G, chunk0, chunks, B, H, W, accumulate, PfFinalize{nullptr, nullptr, 0, 0, 0, 1.f}, ufr::as_stream(stream));
}

extern "C" int ufr_flow_head_planes_backward_finalize(const float* grad_y, const float* wpk, int w_chunks, float* G, int g_chunks,
                                                      int chunk0, int chunks, int B, int H, int W, int accumulate,
                                                      const void* mask_planes, void* out_planes, long out_plane_stride, int fin_chunk0,
                                                      int fin_chunks, float slope, ufr_stream_t stream) {
  UFR_REQUIRE(grad_y && wpk && G && out_planes, "flow head (planes) backward + finalize: null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && chunks > 0 && chunks <= 48 && chunk0 >= 0 && (long)B * H * W < (1L << 29),
              "flow head (planes) backward + finalize: bad shape");
  UFR_REQUIRE_GRANGE("flow head (planes) backward + finalize", chunk0, chunks, w_chunks, g_chunks);
  UFR_REQUIRE(planes_hold(out_plane_stride, (long)B * H * W, chunk0 + fin_chunk0, fin_chunks),
              "flow head (planes) backward + finalize: the finalised segment leaves the gradient planes");
  UFR_REQUIRE(fin_chunk0 >= 0 && fin_chunks > 0 && fin_chunk0 + fin_chunks <= chunks && out_plane_stride > 0,
              "flow head (planes) backward + finalize: the finalised segment leaves the tensor");
  // the mask (plane 0 of the activation) and the gradient planes share the tensor's geometry: element ((chunk0 + ch) * M + pixel) * 32
  return launch_pf_bwd(grad_y, wpk, G, chunk0, chunks, B, H, W, accumulate,
                       PfFinalize{static_cast<const __bf16*>(mask_planes), static_cast<__bf16*>(out_planes), out_plane_stride, fin_chunk0,
                                  fin_chunks, slope},
                       ufr::as_stream(stream));
}

extern "C" int ufr_flow_up_planes_forward(const float* x, const float* w, const float* bias, void* planes, long plane_stride,
                                          int chunk, int B, int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(x && w && planes, "flow upsample (planes) forward: null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && chunk >= 0, "flow upsample (planes) forward: bad shape");
  flow_up_planes_fwd<<<ufr::stream_grid((long)B * 4 * H * W, 256), 256, 0, ufr::as_stream(stream)>>>(
      x, w, bias, static_cast<__bf16*>(planes), plane_stride, chunk, B, H, W, bias != nullptr);
  return ufr::launched("flow_up_planes_fwd");
}

extern "C" int ufr_flow_up_planes_backward(const float* G, int chunk, const float* w, float* grad_x, int B, int H, int W,
                                           int accumulate, ufr_stream_t stream) {
  UFR_REQUIRE(G && w && grad_x, "flow upsample (planes) backward: null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && chunk >= 0, "flow upsample (planes) backward: bad shape");
  flow_up_planes_bwd<<<ufr::stream_grid((long)B * H * W, 256), 256, 0, ufr::as_stream(stream)>>>(G, chunk, w, grad_x, B, H, W, accumulate);
  return ufr::launched("flow_up_planes_bwd");
}
