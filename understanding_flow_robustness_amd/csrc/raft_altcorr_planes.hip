// raft_altcorr_planes.hip -- RAFT's on-the-fly correlation lookup (models/alt_cuda_corr/correlation_kernel.cu:18-119, called per
// pyramid level from models/raft/corr.py:109-137), round 6: on the bf16 matrix cores with float32 accuracy (three bf16 planes per
// operand, the six leading products -- csrc/igemm.hip's arithmetic) and with the fmap2 pixels a tile needs staged ONCE through LDS.
//
// Round 3's form (raft_altcorr_mfma.hip) gives every 16 pixels of a row their own workgroup x level, multiplies on the exact-fp32
// matrix path (`v_mfma_f32_16x16x4_f32`, 1/16 of the bf16 rate) and lets each of its waves fetch its fmap2 segments from L2 by
// itself: a lookup moved ~650 MB from L2 to the CUs for 22 MB of operands and sat at 24 - 31 % MFMA-busy with 0.7 waves per SIMD
// (profiles/r5_altcorr_counters.txt).  Here:
//   * fmap1 and the fmap2 pyramid are split into planes ONCE per forward (`ufr_altcorr_planes_prepare`; they are constant over the
//     12 lookups): bf16 [3][C / 32][pixels][32], the igemm's activation layout -- a pixel's 32-channel chunk is one 64-byte row;
//   * a workgroup owns a 2-D tile of 8 x 16 pixels (eight waves, one per pixel row) on one level; the eight rows' windows overlap
//     almost completely when the flow is smooth, so the workgroup walks the BOUNDING BOX of all 128 windows once: one fmap2 row x
//     32 columns x 128 channels (24 KB) per stage by LDS-DMA, three stages in flight (XOR-swizzled rows as in igemm.hip), every wave
//     multiplies its 16 pixels against the 16-column segments of the stage that meet ITS OWN box: S[p, q] = <fmap1[p], fmap2[q]>,
//     48 `v_mfma_f32_16x16x32_bf16` per segment (8 chunks x 6 products);
//   * every pixel keeps the entries that fall into its own (2r + 2)^2 window (LDS, wave-private) and blends them bilinearly at
//     the end (correlation_kernel.cu:92-115) -> [B, L (2r+1)^2, H, W].
// ANY coordinate field is handled: a discontinuous one makes the box larger (more rows, more 32-column strips), never wrong.
// L2 -> LDS traffic per lookup at 48 x 160, smooth flow: 240 workgroups x ~14 stages x 48 KB = 160 MB instead of ~650 MB; the
// matrix work 2.7x cheaper per product and ~1.6x fewer products (8 rows share the box's row walk, a 16-pixel tile walks its own).
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "ufr_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ constexpr int PROD_A[6] = {2, 0, 1, 1, 0, 0};   // (a plane, b plane) of each product, smallest first (igemm.hip)
__device__ constexpr int PROD_B[6] = {0, 2, 1, 0, 1, 0};
__device__ __attribute__((aligned(64))) unsigned acp_zero_page[16];

constexpr int TW = 16;                            // pixel tile: TH rows (one wave each) x 16 columns
constexpr int SW = 32;                            // fmap2 columns per stage

struct PlaneLevels {                              // by value in the kernel arguments
  int n;
  const __bf16* f2[4];
  long stride[4];                                 // elements between two planes of a level
  int H2[4], W2[4];
  float cscale[4];
};

__device__ __forceinline__ void glds16(const __bf16* src, __bf16* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(src, lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void split3(float v, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)v;
  const float r1 = v - (float)a;
  b = (__bf16)r1;
  c = (__bf16)(r1 - (float)b);
}

// NHWC float32 [npix][C] -> planes bf16 [3][C / 32][npix][32]; thread = (pixel, 8 channels): two 16-byte reads, three 16-byte writes
__global__ __launch_bounds__(256) void altcorr_planes_prepare_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long plane_stride,
                                                                     long npix, int C) {
  const int c8 = C / 8;
  const long total = npix * c8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / c8;
    const int c0 = (int)(i - pix * c8) * 8;
    const float4 lo = *reinterpret_cast<const float4*>(src + pix * C + c0), hi = *reinterpret_cast<const float4*>(src + pix * C + c0 + 4);
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    bf16x8 q0, q1, q2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __bf16 x, y, z;
      split3(v[j], x, y, z);
      q0[j] = x; q1[j] = y; q2[j] = z;
    }
    __bf16* o = dst + ((long)(c0 >> 5) * npix + pix) * 32 + (c0 & 31);
    *reinterpret_cast<bf16x8*>(o) = q0;
    *reinterpret_cast<bf16x8*>(o + plane_stride) = q1;
    *reinterpret_cast<bf16x8*>(o + 2 * plane_stride) = q2;
  }
}

// A STAGE = one fmap2 row x SW columns x KS = 4 chunks x three planes = 24 KB; NBUF = 4 of them: one being multiplied, three in
// flight (a first form with two whole-K stages of 48 KB -- one in flight -- ran at the LDS-DMA's LATENCY: 66 us per lookup, 1.5 - 2.5 us
// per stage for 0.5 us of matrix work, gpurun r6_tr1).
// -> the kernel takes (TH, KS, NBUF) as template parameters; the launch picks the measured best (UFR_ALTCORR_PLANES_FORM sweeps them).
// LDS (bytes): NBUF stages [3 planes][KS chunks][SW pixels][32 ch] bf16 | s [TH waves][TW pixels][npt] float | cx, cy [TP] int | dx, dy [TP] float | box [4] int
template <int R, int TH, int KS, int NBUF>
constexpr int acp_lds_bytes() { return NBUF * 3 * KS * SW * 32 * 2 + TH * TW * (2 * R + 2) * (2 * R + 2) * 4 + 4 * TH * TW * 4 + 16; }

template <int R, int KCH, int TH, int KS, int NBUF>
__global__ __launch_bounds__(64 * TH) void altcorr_planes_fwd(const __bf16* __restrict__ f1, long f1_stride, const PlaneLevels lv,
                                                          const float* __restrict__ coords, float* __restrict__ out, int B, int H1,
                                                          int W1, float scale) {
  constexpr int rd = 2 * R + 1, gd = rd + 1, npt = gd * gd, TP = TH * TW, NT = 64 * TH;
  constexpr int STAGE = 3 * KS * SW * 32;                                // elements of one stage
  constexpr int NSPLIT = KCH / KS;                                       // stages per (row, strip) piece
  static_assert(SW == 32 && KCH % KS == 0, "whole stages");
  extern __shared__ __attribute__((aligned(16))) unsigned char acp_lds[];
  __bf16* const bufs = reinterpret_cast<__bf16*>(acp_lds);
  float* const s_all = reinterpret_cast<float*>(acp_lds + NBUF * STAGE * 2);
  int* const cxs = reinterpret_cast<int*>(s_all + TH * TW * npt);
  int* const cys = cxs + TP;
  float* const dxs = reinterpret_cast<float*>(cys + TP);
  float* const dys = dxs + TP;
  int* const box = reinterpret_cast<int*>(dys + TP);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (level, sample, tile): XCD k (launch index % 8) owns one contiguous run of the (level, tile) list, so the
  // workgroups that read the same fmap2 rows share an L2 (the bijective form for counts that 8 does not divide: igemm.hip xcd_tile)
  const int tiles_x = (W1 + TW - 1) / TW, tiles_y = (H1 + TH - 1) / TH, ntiles = B * tiles_y * tiles_x;
  int item;
  {
    const int n = gridDim.x, orig = blockIdx.x, q = n / 8, r = n % 8, xcd = orig % 8, idx = orig / 8;
    item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int l = item / ntiles, tile = item - l * ntiles;
  const int b = tile / (tiles_y * tiles_x), rem = tile - b * tiles_y * tiles_x, h0 = (rem / tiles_x) * TH, w0 = (rem % tiles_x) * TW;
  const int H2 = lv.H2[l], W2 = lv.W2[l];
  const __bf16* __restrict__ f2 = lv.f2[l];
  const long f2_stride = lv.stride[l];
  const long npix1 = (long)B * H1 * W1, npix2 = (long)B * H2 * W2;
  const size_t plane1 = (size_t)H1 * W1;
  float* const s = s_all + wave * (TW * npt);

  // ---- the 128 windows and the workgroup's box
  if (tid < 4) box[tid] = (tid & 1) ? INT_MIN : INT_MAX;                  // {ymin, ymax, xmin, xmax} of the window origins that meet the image
  for (int i = lane; i < TW * npt; i += 64) s[i] = 0.f;
  __syncthreads();
  int wy0 = INT_MAX, wy1 = INT_MIN, wx0 = INT_MAX, wx1 = INT_MIN;         // this wave's own box (from lanes 0-15)
  {
    const int pi = lane & 15, h1 = h0 + wave, w1 = w0 + pi;
    int cx = 0x3fffffff, cy = 0x3fffffff;
    float dx = 0.f, dy = 0.f;
    if (h1 < H1 && w1 < W1) {
      const size_t pix = (size_t)h1 * W1 + w1;                           // coords: planar [B, 2, H1, W1]
      const float x = coords[((size_t)b * 2 + 0) * plane1 + pix] * lv.cscale[l], y = coords[((size_t)b * 2 + 1) * plane1 + pix] * lv.cscale[l];
      const float fx = floorf(x), fy = floorf(y);
      dx = x - fx;
      dy = y - fy;
      // far-away (or non-finite) windows are clamped so that the integer arithmetic cannot overflow; they stay outside every image
      cx = (int)fminf(fmaxf(fx, -1.0e6f), 1.0e6f) - R;
      cy = (int)fminf(fmaxf(fy, -1.0e6f), 1.0e6f) - R;
      if (!(fabsf(x) < 1.0e6f) || !(fabsf(y) < 1.0e6f)) {               // (a non-finite coordinate: the pixel's outputs are zeros)
        cx = cy = 0x3fffffff;
        dx = dy = 0.f;
      }
    }
    const bool m = cx < W2 && cx + gd > 0 && cy < H2 && cy + gd > 0;
    int ymin = m ? cy : INT_MAX, ymax = m ? cy : INT_MIN, xmin = m ? cx : INT_MAX, xmax = m ? cx : INT_MIN;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      ymin = min(ymin, __shfl_xor(ymin, off, 16)); ymax = max(ymax, __shfl_xor(ymax, off, 16));
      xmin = min(xmin, __shfl_xor(xmin, off, 16)); xmax = max(xmax, __shfl_xor(xmax, off, 16));
    }
    if (lane < 16) {
      cxs[wave * TW + pi] = cx; cys[wave * TW + pi] = cy; dxs[wave * TW + pi] = dx; dys[wave * TW + pi] = dy;
    }
    if (lane == 0 && ymin != INT_MAX) {
      atomicMin(&box[0], ymin); atomicMax(&box[1], ymax); atomicMin(&box[2], xmin); atomicMax(&box[3], xmax);
    }
    if (ymin != INT_MAX) {                                                // (uniform over the wave: lanes 16-63 repeat lanes 0-15)
      wy0 = max(ymin, 0); wy1 = min(ymax + gd, H2); wx0 = max(xmin, 0); wx1 = min(xmax + gd, W2);
    }
  }
  __syncthreads();
  const bool any = box[0] != INT_MAX;
  const int Y0 = any ? max(box[0], 0) : 0, Y1 = any ? min(box[1] + gd, H2) : 0, X0 = any ? max(box[2], 0) : 0, X1 = any ? min(box[3] + gd, W2) : 0;
  const int nstrip = (X1 - X0 + SW - 1) / SW, npiece = max(Y1 - Y0, 0) * max(nstrip, 0);

  // ---- A operand: this wave's 16 pixels, all chunks, three planes (lane = (pixel lane & 15, 8-channel group lane >> 4))
  const int pi = lane & 15, kg = lane >> 4;
  bf16x8 fa[KCH][3];
  {
    const int h1 = min(h0 + wave, H1 - 1), w1 = min(w0 + pi, W1 - 1);
    const __bf16* ap = f1 + (((long)b * H1 + h1) * W1 + w1) * 32 + kg * 8;
#pragma unroll
    for (int c = 0; c < KCH; ++c)
#pragma unroll
      for (int p = 0; p < 3; ++p) fa[c][p] = *reinterpret_cast<const bf16x8*>(ap + p * f1_stride + (long)c * npix1 * 32);
  }
  int cyr[4], cxr[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { cyr[j] = cys[wave * TW + 4 * kg + j]; cxr[j] = cxs[wave * TW + 4 * kg + j]; }

  // ---- stages: stage i = chunks KS (i % NSPLIT) .. of piece i / NSPLIT = (row Y0 + piece / nstrip, columns X0 + 32 (piece % nstrip) ..)
  const __bf16* zero = reinterpret_cast<const __bf16*>(acp_zero_page);
  constexpr int ITEMS = 3 * KS * SW * 4 / NT;                             // 16-byte transfers per lane and stage
  static_assert(3 * KS * SW * 4 % NT == 0, "whole transfers per lane");
  const int nstage = npiece * NSPLIT;
  auto stage = [&](int i) {
    const int piece = i / NSPLIT, kh = i - piece * NSPLIT;
    const int y = Y0 + piece / nstrip, xs = X0 + (piece - (piece / nstrip) * nstrip) * SW;
    const long rowpix = ((long)b * H2 + y) * W2;
    __bf16* dst = bufs + (i % NBUF) * STAGE;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int it = k * NT + tid, slot = it & 3, px = (it >> 2) & (SW - 1), pc = it >> 7, p = pc / KS, c = pc - p * KS;
      const int x = xs + px;
      const __bf16* src = x < W2 ? f2 + p * f2_stride + ((long)(kh * KS + c) * npix2 + rowpix + x) * 32 + ((slot ^ ((px >> 1) & 3)) << 3) : zero;
      glds16(src, dst + (k * NT + wave * 64) * 8);                      // lane l lands 16 l bytes behind the wave's base
    }
  };
  const int frow = lane & 15;
  for (int i = 0; i < NBUF - 1 && i < nstage; ++i) stage(i);
  f32x4 acc[4];                                                          // two segments x (even, odd products)
  for (int i = 0; i < nstage; ++i) {
    // my transfers of stage i have landed (the younger stages' -- ITEMS instructions each -- may still fly) ...
    const int ahead = min(nstage - 1 - i, NBUF - 2);
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ITEMS) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ITEMS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                      // ... everybody's have, and everybody is done with stage i - 1's buffer
    if (i + NBUF - 1 < nstage) stage(i + NBUF - 1);                       // into the buffer of stage i - 1
    const int piece = i / NSPLIT, kh = i - piece * NSPLIT;
    const int y = Y0 + piece / nstrip, xs = X0 + (piece - (piece / nstrip) * nstrip) * SW;
    if (y < wy0 || y >= wy1) continue;                                    // (wave-uniform; no barrier inside)
    const __bf16* sB = bufs + (i % NBUF) * STAGE;
    // the stage's two 16-column segments that meet this wave's box, their MFMA chains INTERLEAVED and each split over two
    // accumulators (even / odd products): one chain of 48 dependent MFMAs per segment ran at the instruction's latency, not its rate
    const bool m0 = xs < wx1 && xs + 16 > wx0, m1 = xs + 16 < wx1 && xs + 32 > wx0;
    if (kh == 0) {
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int r0 = frow, r1 = 16 + frow;
    const int boff0 = r0 * 32 + ((kg ^ ((r0 >> 1) & 3)) << 3), boff1 = r1 * 32 + ((kg ^ ((r1 >> 1) & 3)) << 3);
    auto run = [&](auto M0, auto M1) {
#pragma unroll
      for (int c = 0; c < KS; ++c) {
        bf16x8 fb0[3], fb1[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          if (decltype(M0)::value) fb0[p] = *reinterpret_cast<const bf16x8*>(sB + (p * KS + c) * (SW * 32) + boff0);
          if (decltype(M1)::value) fb1[p] = *reinterpret_cast<const bf16x8*>(sB + (p * KS + c) * (SW * 32) + boff1);
        }
#pragma unroll
        for (int h = 0; h < NSPLIT; ++h)                                  // (fa[] indexed statically)
          if (h == kh) {
#pragma unroll
            for (int t = 0; t < 6; ++t) {
              if (decltype(M0)::value)
                acc[t & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[h * KS + c][PROD_A[t]], fb0[PROD_B[t]], acc[t & 1], 0, 0, 0);
              if (decltype(M1)::value)
                acc[2 + (t & 1)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[h * KS + c][PROD_A[t]], fb1[PROD_B[t]], acc[2 + (t & 1)], 0, 0, 0);
            }
          }
      }
    };
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    if (m0 && m1) run(T_{}, T_{});
    else if (m0) run(T_{}, F_{});
    else if (m1) run(F_{}, T_{});
    if (kh != NSPLIT - 1) continue;
    // D[row = pixel 4 kg + j][col = fmap2 column q0 + (lane & 15)]: keep what falls into that pixel's window
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      if (!(nt ? m1 : m0)) continue;
      const int q = xs + nt * 16 + frow;
      if (q < W2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int iy = y - cyr[j], ix = q - cxr[j];
          if ((unsigned)iy < (unsigned)gd && (unsigned)ix < (unsigned)gd) s[(4 * kg + j) * npt + iy * gd + ix] = acc[2 * nt][j] + acc[2 * nt + 1][j];
        }
      }
    }
  }
  __syncthreads();                                                        // (every wave past its last stage; the windows are wave-private)
  // ---- blend (correlation_kernel.cu:92-115): 16 consecutive pixels per output channel = 64 contiguous bytes
  const int h1 = h0 + wave;
  if (h1 >= H1) return;
  for (int t = lane; t < TW * rd * rd; t += 64) {
    const int i = t & 15, o = t >> 4, ox = o / rd, oy = o - ox * rd;
    if (w0 + i >= W1) continue;
    const float dx = dxs[wave * TW + i], dy = dys[wave * TW + i];
    const float* si = s + i * npt;
    const float v = (1 - dy) * (1 - dx) * si[oy * gd + ox] + (1 - dy) * dx * si[oy * gd + ox + 1] + dy * (1 - dx) * si[(oy + 1) * gd + ox] +
                    dy * dx * si[(oy + 1) * gd + ox + 1];
    out[(((size_t)b * lv.n + l) * rd * rd + o) * plane1 + (size_t)h1 * W1 + w0 + i] = v * scale;
  }
}

template <int R, int KCH, int TH, int KS, int NBUF>
int launch_planes_form(const __bf16* f1, long f1_stride, const PlaneLevels& lv, const float* coords, float* out, int B, int H1, int W1, float scale,
                       hipStream_t st) {
  constexpr int lds = acp_lds_bytes<R, TH, KS, NBUF>();
  static_assert(lds <= 160 * 1024, "LDS");
  hipError_t e = ufr::ensure_dynamic_lds(reinterpret_cast<const void*>(altcorr_planes_fwd<R, KCH, TH, KS, NBUF>), lds);
  if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "altcorr planes forward: %s", hipGetErrorString(e));
  const int ntiles = B * ((H1 + TH - 1) / TH) * ((W1 + TW - 1) / TW);
  altcorr_planes_fwd<R, KCH, TH, KS, NBUF><<<ntiles * lv.n, 64 * TH, lds, st>>>(f1, f1_stride, lv, coords, out, B, H1, W1, scale);
  return ufr::launched("altcorr_planes_fwd");
}

// forms: 0 = 8 rows, whole-K stages, double buffer (one stage in flight); 1 = 8 rows, 4-chunk stages, three in flight;
//        2 = 4 rows, 4-chunk stages, double buffer (75 KB of LDS: two workgroups per CU); 3 = 4 rows, whole-K stages, double buffer
template <int R, int KCH>
int launch_planes_fwd(const __bf16* f1, long f1_stride, const PlaneLevels& lv, const float* coords, float* out, int B, int H1, int W1, float scale,
                      hipStream_t st) {
  static const int form = [] { const char* e = getenv("UFR_ALTCORR_PLANES_FORM"); return e ? atoi(e) : 0; }();
  if (form == 1) return launch_planes_form<R, KCH, 8, 4, 4>(f1, f1_stride, lv, coords, out, B, H1, W1, scale, st);
  if (form == 2) return launch_planes_form<R, KCH, 4, 4, 2>(f1, f1_stride, lv, coords, out, B, H1, W1, scale, st);
  if (form == 3) return launch_planes_form<R, KCH, 4, KCH, 2>(f1, f1_stride, lv, coords, out, B, H1, W1, scale, st);
  return launch_planes_form<R, KCH, 8, KCH, 2>(f1, f1_stride, lv, coords, out, B, H1, W1, scale, st);
}

}  // namespace

extern "C" int ufr_altcorr_planes_prepare(const float* fmap_nhwc, void* planes, long plane_stride, long npix, int C, ufr_stream_t stream) {
  UFR_REQUIRE(fmap_nhwc && planes, "altcorr planes prepare: null pointer");
  UFR_REQUIRE(npix > 0 && C > 0 && C % 32 == 0 && plane_stride >= npix * C, "altcorr planes prepare: bad shape (C must be a multiple of 32, planes %ld "
              "elements apart for %ld x %d values)", plane_stride, npix, C);
  altcorr_planes_prepare_kernel<<<ufr::stream_grid(npix * (C / 8), 256), 256, 0, ufr::as_stream(stream)>>>(
      fmap_nhwc, static_cast<__bf16*>(planes), plane_stride, npix, C);
  return ufr::launched("altcorr_planes_prepare_kernel");
}

extern "C" int ufr_altcorr_planes_forward(const void* fmap1_planes, long fmap1_plane_stride, const ufr_altcorr_plane_levels* levels,
                                          const float* coords, float* out, int B, int H1, int W1, int C, int radius, float scale,
                                          ufr_stream_t stream) {
  UFR_REQUIRE(fmap1_planes && levels && coords && out, "altcorr planes forward: null pointer");
  UFR_REQUIRE(B > 0 && H1 > 0 && W1 > 0 && levels->num_levels >= 1 && levels->num_levels <= 4, "altcorr planes forward: bad shape");
  UFR_REQUIRE((C == 256 || C == 128) && (radius == 4 || radius == 3), "altcorr planes forward: C must be 128 or 256 and the radius 3 or 4 (got %d, %d)",
              C, radius);
  UFR_REQUIRE(fmap1_plane_stride >= (long)B * H1 * W1 * C, "altcorr planes forward: fmap1's planes overlap");
  PlaneLevels lv{};
  lv.n = levels->num_levels;
  for (int l = 0; l < lv.n; ++l) {
    UFR_REQUIRE(levels->planes[l] && levels->H2[l] > 0 && levels->W2[l] > 0 && levels->plane_stride[l] >= (long)B * levels->H2[l] * levels->W2[l] * C,
                "altcorr planes forward: bad level %d", l);
    lv.f2[l] = static_cast<const __bf16*>(levels->planes[l]); lv.stride[l] = levels->plane_stride[l];
    lv.H2[l] = levels->H2[l]; lv.W2[l] = levels->W2[l]; lv.cscale[l] = levels->coord_scale[l];
  }
  const __bf16* f1 = static_cast<const __bf16*>(fmap1_planes);
  hipStream_t st = ufr::as_stream(stream);
  if (C == 256 && radius == 4) return launch_planes_fwd<4, 8>(f1, fmap1_plane_stride, lv, coords, out, B, H1, W1, scale, st);
  if (C == 128 && radius == 4) return launch_planes_fwd<4, 4>(f1, fmap1_plane_stride, lv, coords, out, B, H1, W1, scale, st);
  if (C == 256 && radius == 3) return launch_planes_fwd<3, 8>(f1, fmap1_plane_stride, lv, coords, out, B, H1, W1, scale, st);
  return launch_planes_fwd<3, 4>(f1, fmap1_plane_stride, lv, coords, out, B, H1, W1, scale, st);
}
