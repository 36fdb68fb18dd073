// split_conv_wide.hip -- EXPERIMENTAL (compiled, not yet run on hardware at the end of round 1; its tests are gated
// by UFR_EXPERIMENTAL=1).  The next schedule step for csrc/split_gemm.hip's implicit-GEMM convolution, written
// from the traffic arithmetic in DESIGN.md 10: a 128 (pixels) x 256 (output channels) tile, so that FlowNetC's
// conv3_1 (N = 256) stages every activation tile once instead of twice, eight waves as 2 x 4 (each still a 64x64
// block of 4x4 `v_mfma_f32_16x16x32_bf16` accumulators), one workgroup per CU and therefore room for TWO LDS
// buffers (2 x 72 KB): the next K tile is written while the current one is multiplied, one barrier per K step.
//
// Same operands as ufr_conv3x3_split (three bf16 planes per operand, row-major or chunk-major), same result
// definition; N must be a multiple of 256.
#include "ufr_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int WM = 128, WN = 256, BK = 32;
constexpr int A_IMG = WM * BK, B_IMG = WN * BK;   // bf16 elements of one plane's tile

// XCD-aware tile order (UFR_SPLIT_XCD=1, off by default until measured): workgroups are dealt round-robin to the 8
// XCDs in launch order, so the launch index i is remapped to tile (i % 8) * ceil(n/8) + i / 8 (the bijective form for
// n % 8 != 0, cdna_hip_programming.md): every XCD then owns one contiguous run of tiles -- the N-tiles of one pixel tile
// and the neighbouring pixel tiles, whose activation rows overlap -- and its private L2 sees their re-reads.
__device__ __forceinline__ void split_tile_of_block(int swz, int& bx, int& by) {
  bx = blockIdx.x;
  by = blockIdx.y;
  if (!swz) return;
  const int gx = gridDim.x, nwg = gx * gridDim.y, orig = by * gx + bx;
  const int q = nwg / 8, r = nwg % 8, xcd = orig % 8, idx = orig / 8;
  const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  bx = wgid % gx;
  by = wgid / gx;
}

static int split_xcd_swizzle() {
  static const int v = [] { const char* e = getenv("UFR_SPLIT_XCD"); return e && e[0] == '1' ? 1 : 0; }();
  return v;
}

__device__ constexpr int PROD_A[6] = {2, 0, 1, 1, 0, 0};   // as in split_gemm.hip: smallest products first
__device__ constexpr int PROD_B[6] = {0, 2, 1, 0, 1, 0};

template <int NPROD>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_split_wide_kernel(
    const __bf16* __restrict__ Xp, const __bf16* __restrict__ Wp, float* __restrict__ Y, int B, int H, int W, int Cpad,
    int N, long rsA, long ksA, long rsB, long ksB, int swz) {
  constexpr int NPL = NPROD == 1 ? 1 : (NPROD == 3 ? 2 : 3);
  constexpr int FIRST = 6 - NPROD;
  constexpr int BUF = NPL * (A_IMG + B_IMG);
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
  int tile_x, tile_y;
  split_tile_of_block(swz, tile_x, tile_y);
  const int bm = tile_y * WM, bn = tile_x * WN;
  const int M = B * H * W, KC = Cpad / BK, KT = 9 * KC;
  const size_t planeA = (size_t)M * Cpad, planeB = (size_t)N * 9 * Cpad;

  // staging: one activation chunk and two weight chunks (rows r and r + 128) per thread and plane
  const int srow = tid >> 2, sch = tid & 3;
  const int pm = bm + srow, px = pm % W, py = pm < M ? (pm / W) % H : -4;
  const __bf16* gb = Wp + (size_t)(bn + srow) * rsB + sch * 8;
  const int soff = srow * BK + ((sch ^ ((srow >> 1) & 3)) << 3);      // + 128 rows keeps (row >> 1) & 3
  u32x4 sa[NPL], sb[NPL][2];
#define UFR_SW_LOAD(kt)                                                                             \
  {                                                                                                 \
    const int tap = (kt) / KC, kc = (kt) - tap * KC, dyo = tap / 3 - 1, dxo = tap % 3 - 1;           \
    const bool ok = (unsigned)(py + dyo) < (unsigned)H && (unsigned)(px + dxo) < (unsigned)W;        \
    const __bf16* src = Xp + (size_t)(ok ? pm + dyo * W + dxo : 0) * rsA + kc * ksA + sch * 8;       \
    _Pragma("unroll") for (int p = 0; p < NPL; ++p) {                                               \
      const u32x4 v = *reinterpret_cast<const u32x4*>(src + p * planeA);                            \
      sa[p] = ok ? v : u32x4{0u, 0u, 0u, 0u};                                                       \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                 \
        sb[p][i] = *reinterpret_cast<const u32x4*>(gb + p * planeB + (size_t)(128 * i) * rsB + (kt) * ksB); \
    }                                                                                               \
  }
#define UFR_SW_STORE(buf)                                                                           \
  _Pragma("unroll") for (int p = 0; p < NPL; ++p) {                                                 \
    *reinterpret_cast<u32x4*>(&lds[(buf) * BUF + p * A_IMG + soff]) = sa[p];                        \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                   \
      *reinterpret_cast<u32x4*>(&lds[(buf) * BUF + NPL * A_IMG + p * B_IMG + soff + 128 * i * BK]) = sb[p][i]; \
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15;
  const int foff = frow * BK + ((((lane >> 4)) ^ ((frow >> 1) & 3)) << 3);

  UFR_SW_LOAD(0)
  UFR_SW_STORE(0)
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < KT;
    if (more) UFR_SW_LOAD(kt + 1)
    const __bf16* la = lds + cur * BUF;
    const __bf16* lb = la + NPL * A_IMG;
    bf16x8 a[NPL][4];
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int m = 0; m < 4; ++m)
        a[p][m] = *reinterpret_cast<const bf16x8*>(&la[p * A_IMG + (wr * 64 + m * 16) * BK + foff]);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      bf16x8 b[NPL];
#pragma unroll
      for (int p = 0; p < NPL; ++p) b[p] = *reinterpret_cast<const bf16x8*>(&lb[p * B_IMG + (wc * 64 + n * 16) * BK + foff]);
#pragma unroll
      for (int t = FIRST; t < 6; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PROD_A[t]][m], b[PROD_B[t]], acc[m][n], 0, 0, 0);
    }
    // the other buffer was last read in iteration kt - 1, and every wave has passed that iteration's barrier
    if (more) { UFR_SW_STORE(cur ^ 1) }
    __syncthreads();
  }
#undef UFR_SW_LOAD
#undef UFR_SW_STORE

#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = bm + wr * 64 + m * 16 + (lane >> 4) * 4 + j;
      if (row < M) {
#pragma unroll
        for (int n = 0; n < 4; ++n) Y[(size_t)row * N + bn + wc * 64 + n * 16 + (lane & 15)] = acc[m][n][j];
      }
    }
}

// General forward convolution (KH x KW taps, stride s, padding p, no dilation / groups) on the measured 128 x 128
// tile of split_gemm.hip: the tile rows are OUTPUT pixels, a tap's row is the input pixel (yo*s + ky - p,
// xo*s + kx - p).  For the strided 3x3 / 5x5 / 7x7 layers and 1x1 projections; their adjoints (transposed
// convolutions) are not this kernel.
constexpr int GM = 128, GN = 128;

template <int NPROD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_split_general_kernel(
    const __bf16* __restrict__ Xp, const __bf16* __restrict__ Wp, float* __restrict__ Y, int B, int Hi, int Wi, int Ho,
    int Wo, int Cpad, int N, int KH, int KW, int stride, int pad, long rsA, long ksA, long rsB, long ksB, int swz) {
  constexpr int NPL = NPROD == 1 ? 1 : (NPROD == 3 ? 2 : 3);
  constexpr int FIRST = 6 - NPROD;
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * NPL][GM * BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  int tile_x, tile_y;
  split_tile_of_block(swz, tile_x, tile_y);
  const int bm = tile_y * GM, bn = tile_x * GN;
  const int M = B * Ho * Wo, Min = B * Hi * Wi, KC = Cpad / BK, KT = KH * KW * KC;
  const size_t planeA = (size_t)Min * Cpad, planeB = (size_t)N * KH * KW * Cpad;

  const int srow0 = tid >> 2, sch = tid & 3;
  int ybase[2], xbase[2], ibase[2];                 // input row / column of tap (0,0), and the image's first pixel
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pm = bm + srow0 + 64 * i;
    const int xo = pm % Wo, yo = (pm / Wo) % Ho, b = pm / (Wo * Ho);
    xbase[i] = xo * stride - pad;
    ybase[i] = pm < M ? yo * stride - pad : -(1 << 20);    // rows past the end never pass the bounds test
    ibase[i] = pm < M ? b * Hi * Wi : 0;
  }
  u32x4 sa[NPL][2], sb[NPL][2];
  const __bf16* gb = Wp + (size_t)(bn + srow0) * rsB + sch * 8;
  const int soff0 = srow0 * BK + ((sch ^ ((srow0 >> 1) & 3)) << 3);
#define UFR_GC_LOAD(kt)                                                                               \
  {                                                                                                   \
    const int tap = (kt) / KC, kc = (kt) - tap * KC, ky = tap / KW, kx = tap - ky * KW;                \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                   \
      const int yi = ybase[i] + ky, xi = xbase[i] + kx;                                               \
      const bool ok = (unsigned)yi < (unsigned)Hi && (unsigned)xi < (unsigned)Wi;                      \
      const __bf16* src = Xp + (size_t)(ok ? ibase[i] + yi * Wi + xi : 0) * rsA + kc * ksA + sch * 8;  \
      _Pragma("unroll") for (int p = 0; p < NPL; ++p) {                                               \
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + p * planeA);                            \
        sa[p][i] = ok ? v : u32x4{0u, 0u, 0u, 0u};                                                    \
        sb[p][i] = *reinterpret_cast<const u32x4*>(gb + p * planeB + (size_t)(64 * i) * rsB + (kt) * ksB);    \
      }                                                                                               \
    }                                                                                                 \
  }
#define UFR_GC_STORE()                                                                              \
  _Pragma("unroll") for (int p = 0; p < NPL; ++p) _Pragma("unroll") for (int i = 0; i < 2; ++i) {   \
    *reinterpret_cast<u32x4*>(&lds[p][soff0 + 64 * i * BK]) = sa[p][i];                             \
    *reinterpret_cast<u32x4*>(&lds[NPL + p][soff0 + 64 * i * BK]) = sb[p][i];                       \
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15;
  const int foff = frow * BK + ((((lane >> 4)) ^ ((frow >> 1) & 3)) << 3);

  UFR_GC_LOAD(0)
  for (int kt = 0; kt < KT; ++kt) {
    __syncthreads();
    UFR_GC_STORE()
    __syncthreads();
    if (kt + 1 < KT) UFR_GC_LOAD(kt + 1)
    bf16x8 a[NPL][4];
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int m = 0; m < 4; ++m)
        a[p][m] = *reinterpret_cast<const bf16x8*>(&lds[p][(wr * 64 + m * 16) * BK + foff]);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      bf16x8 b[NPL];
#pragma unroll
      for (int p = 0; p < NPL; ++p) b[p] = *reinterpret_cast<const bf16x8*>(&lds[NPL + p][(wc * 64 + n * 16) * BK + foff]);
#pragma unroll
      for (int t = FIRST; t < 6; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PROD_A[t]][m], b[PROD_B[t]], acc[m][n], 0, 0, 0);
    }
  }
#undef UFR_GC_LOAD
#undef UFR_GC_STORE
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = bm + wr * 64 + m * 16 + (lane >> 4) * 4 + j;
      if (row < M) {
#pragma unroll
        for (int n = 0; n < 4; ++n) Y[(size_t)row * N + bn + wc * 64 + n * 16 + (lane & 15)] = acc[m][n][j];
      }
    }
}

// Stride-2 transposed convolution by phases (split_gemm.py::deconv_plan): blockIdx.z = phase; the tile rows are the
// COARSE pixels q, every tap reads coarse pixel q + (dy, dx), the result lands on fine pixel (2qy + oy0, 2qx + ox0).
// Chunk-major planes only; weights: per phase a [taps*Cpad/32][N][32] image at w_off[phase].
struct DeconvPlan {
  int ntaps[4], oy0[4], ox0[4];
  int dy[4][16], dx[4][16];
  long w_off[4];
};

template <int NPROD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void deconv_split_kernel(
    const __bf16* __restrict__ Xp, const __bf16* __restrict__ Wp, float* __restrict__ Y, int B, int Hi, int Wi, int Cpad,
    int N, long planeB, DeconvPlan plan, int swz) {
  constexpr int NPL = NPROD == 1 ? 1 : (NPROD == 3 ? 2 : 3);
  constexpr int FIRST = 6 - NPROD;
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * NPL][GM * BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  int tile_x, tile_y;
  split_tile_of_block(swz, tile_x, tile_y);
  const int bm = tile_y * GM, bn = tile_x * GN, z = blockIdx.z;
  const int M = B * Hi * Wi, KC = Cpad / BK, KT = plan.ntaps[z] * KC;
  const size_t planeA = (size_t)M * Cpad;

  const int srow0 = tid >> 2, sch = tid & 3;
  int qy[2], qx[2], ibase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pm = bm + srow0 + 64 * i;
    qx[i] = pm % Wi;
    qy[i] = pm < M ? (pm / Wi) % Hi : -(1 << 20);
    ibase[i] = pm < M ? (pm / (Wi * Hi)) * Hi * Wi : 0;
  }
  u32x4 sa[NPL][2], sb[NPL][2];
  const __bf16* gb = Wp + plan.w_off[z] + (size_t)(bn + srow0) * BK + sch * 8;
  const int soff0 = srow0 * BK + ((sch ^ ((srow0 >> 1) & 3)) << 3);
#define UFR_DC_LOAD(kt)                                                                               \
  {                                                                                                   \
    const int tap = (kt) / KC, kc = (kt) - tap * KC, dyo = plan.dy[z][tap], dxo = plan.dx[z][tap];     \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                   \
      const int yi = qy[i] + dyo, xi = qx[i] + dxo;                                                   \
      const bool ok = (unsigned)yi < (unsigned)Hi && (unsigned)xi < (unsigned)Wi;                      \
      const __bf16* src = Xp + ((size_t)kc * M + (ok ? ibase[i] + yi * Wi + xi : 0)) * BK + sch * 8;   \
      _Pragma("unroll") for (int p = 0; p < NPL; ++p) {                                               \
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + p * planeA);                            \
        sa[p][i] = ok ? v : u32x4{0u, 0u, 0u, 0u};                                                    \
        sb[p][i] = *reinterpret_cast<const u32x4*>(gb + p * planeB + ((size_t)(kt) * N + 64 * i) * BK); \
      }                                                                                               \
    }                                                                                                 \
  }
#define UFR_DC_STORE()                                                                              \
  _Pragma("unroll") for (int p = 0; p < NPL; ++p) _Pragma("unroll") for (int i = 0; i < 2; ++i) {   \
    *reinterpret_cast<u32x4*>(&lds[p][soff0 + 64 * i * BK]) = sa[p][i];                             \
    *reinterpret_cast<u32x4*>(&lds[NPL + p][soff0 + 64 * i * BK]) = sb[p][i];                       \
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15;
  const int foff = frow * BK + ((((lane >> 4)) ^ ((frow >> 1) & 3)) << 3);

  UFR_DC_LOAD(0)
  for (int kt = 0; kt < KT; ++kt) {
    __syncthreads();
    UFR_DC_STORE()
    __syncthreads();
    if (kt + 1 < KT) UFR_DC_LOAD(kt + 1)
    bf16x8 a[NPL][4];
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int m = 0; m < 4; ++m)
        a[p][m] = *reinterpret_cast<const bf16x8*>(&lds[p][(wr * 64 + m * 16) * BK + foff]);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      bf16x8 b[NPL];
#pragma unroll
      for (int p = 0; p < NPL; ++p) b[p] = *reinterpret_cast<const bf16x8*>(&lds[NPL + p][(wc * 64 + n * 16) * BK + foff]);
#pragma unroll
      for (int t = FIRST; t < 6; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PROD_A[t]][m], b[PROD_B[t]], acc[m][n], 0, 0, 0);
    }
  }
#undef UFR_DC_LOAD
#undef UFR_DC_STORE
  const int oy0 = plan.oy0[z], ox0 = plan.ox0[z];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = bm + wr * 64 + m * 16 + (lane >> 4) * 4 + j;
      if (row < M) {
        const int x = row % Wi, y = (row / Wi) % Hi, b = row / (Wi * Hi);
        const size_t orow = ((size_t)b * 2 * Hi + 2 * y + oy0) * (2 * Wi) + 2 * x + ox0;
#pragma unroll
        for (int n = 0; n < 4; ++n) Y[orow * N + bn + wc * 64 + n * 16 + (lane & 15)] = acc[m][n][j];
      }
    }
}

// x[B][C][H*W] float32 -> chunk-major planes[3][Cpad/32][B*H*W][32] bf16 directly (split_gemm.hip's
// nchw_to_nhwc_split3_kernel writes the row-major image, which torch then permutes: one pass saved per layer).
__global__ __launch_bounds__(256) void nchw_to_planes_cm_kernel(const float* __restrict__ x, __bf16* __restrict__ planes,
                                                                int B, int C, int HW, int Cpad) {
  __shared__ float tile[32][65];
  const int c0 = blockIdx.x * 32, p0 = blockIdx.y * 64, b = blockIdx.z, tid = threadIdx.x;
  {
    const int p = tid & 63;
#pragma unroll
    for (int cc = tid >> 6; cc < 32; cc += 4) {
      const int c = c0 + cc;
      tile[cc][p] = (c < C && p0 + p < HW) ? x[((size_t)b * C + c) * HW + p0 + p] : 0.f;
    }
  }
  __syncthreads();
  const int p = tid >> 2, ch = tid & 3;
  if (p0 + p >= HW) return;
  const size_t rows = (size_t)B * HW, plane = rows * Cpad;
  __bf16* dst = planes + ((size_t)blockIdx.x * rows + (size_t)b * HW + p0 + p) * 32 + ch * 8;
  bf16x8 q0, q1, q2;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = tile[ch * 8 + j][p];
    const __bf16 a = (__bf16)v;
    const float r1 = v - (float)a;
    const __bf16 bq = (__bf16)r1;
    q0[j] = a;
    q1[j] = bq;
    q2[j] = (__bf16)(r1 - (float)bq);
  }
  *reinterpret_cast<bf16x8*>(dst) = q0;
  *reinterpret_cast<bf16x8*>(dst + plane) = q1;
  *reinterpret_cast<bf16x8*>(dst + 2 * plane) = q2;
}

// y[B*H*W][Npad] float32 (the convolution's NHWC rows) -> out[B][N][H*W] = act(y + bias[c]); bias may be null,
// slope 1 = no activation (the data-gradient direction).
__global__ __launch_bounds__(256) void rows_to_nchw_kernel(const float* __restrict__ y, const float* __restrict__ bias,
                                                           float* __restrict__ out, int B, int N, int HW, int Npad,
                                                           float slope) {
  __shared__ float tile[32][65];
  const int c0 = blockIdx.x * 32, p0 = blockIdx.y * 64, b = blockIdx.z, tid = threadIdx.x;
  {
    const int p = tid >> 2, q = tid & 3;
    if (p0 + p < HW) {
      const float4* src = reinterpret_cast<const float4*>(y + ((size_t)b * HW + p0 + p) * Npad + c0 + q * 8);
      const float4 v0 = src[0], v1 = src[1];
      tile[q * 8 + 0][p] = v0.x; tile[q * 8 + 1][p] = v0.y; tile[q * 8 + 2][p] = v0.z; tile[q * 8 + 3][p] = v0.w;
      tile[q * 8 + 4][p] = v1.x; tile[q * 8 + 5][p] = v1.y; tile[q * 8 + 6][p] = v1.z; tile[q * 8 + 7][p] = v1.w;
    }
  }
  __syncthreads();
  const int p = tid & 63;
  if (p0 + p >= HW) return;
#pragma unroll
  for (int cc = tid >> 6; cc < 32; cc += 4) {
    const int c = c0 + cc;
    if (c < N) {
      float v = tile[cc][p] + (bias ? bias[c] : 0.f);
      v = v > 0.f ? v : v * slope;
      out[((size_t)b * N + c) * HW + p0 + p] = v;
    }
  }
}

}  // namespace

extern "C" int ufr_conv3x3_split_wide(const void* x_planes, const void* w_planes, float* y, int B, int H, int W,
                                      int Cpad, int N, int products, int chunk_major, ufr_stream_t stream) {
  UFR_REQUIRE(x_planes && w_planes && y, "split conv (wide): null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && Cpad > 0 && Cpad % BK == 0 && N > 0 && N % WN == 0,
              "split conv (wide): Cpad must be a multiple of 32, the output channels of 256");
  UFR_REQUIRE((long)B * H * W < (1L << 31) / 2, "split conv (wide): too many pixels");
  UFR_REQUIRE(products == 6 || products == 3 || products == 1, "split conv (wide): products must be 6, 3 or 1");
  const int M = B * H * W;
  const dim3 grid(N / WN, (M + WM - 1) / WM);
  hipStream_t st = ufr::as_stream(stream);
  const __bf16* a = static_cast<const __bf16*>(x_planes);
  const __bf16* b = static_cast<const __bf16*>(w_planes);
  const long rsA = chunk_major ? BK : Cpad, ksA = chunk_major ? (long)M * BK : BK;
  const long rsB = chunk_major ? BK : 9L * Cpad, ksB = chunk_major ? (long)N * BK : BK;
  if (products == 6) conv3x3_split_wide_kernel<6><<<grid, 512, 0, st>>>(a, b, y, B, H, W, Cpad, N, rsA, ksA, rsB, ksB, split_xcd_swizzle());
  else if (products == 3) conv3x3_split_wide_kernel<3><<<grid, 512, 0, st>>>(a, b, y, B, H, W, Cpad, N, rsA, ksA, rsB, ksB, split_xcd_swizzle());
  else conv3x3_split_wide_kernel<1><<<grid, 512, 0, st>>>(a, b, y, B, H, W, Cpad, N, rsA, ksA, rsB, ksB, split_xcd_swizzle());
  return ufr::launched("conv3x3_split_wide_kernel");
}

extern "C" int ufr_nchw_to_planes_cm(const float* x, void* planes, int B, int C, int H, int W, int Cpad,
                                     ufr_stream_t stream) {
  UFR_REQUIRE(x && planes, "nchw -> chunk-major planes: null pointer");
  UFR_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C && Cpad % BK == 0 && B < 65536,
              "nchw -> chunk-major planes: bad shape (Cpad must be a multiple of 32, >= C)");
  const dim3 grid(Cpad / 32, (H * W + 63) / 64, B);
  nchw_to_planes_cm_kernel<<<grid, 256, 0, ufr::as_stream(stream)>>>(x, static_cast<__bf16*>(planes), B, C, H * W, Cpad);
  return ufr::launched("nchw_to_planes_cm_kernel");
}

extern "C" int ufr_rows_to_nchw(const float* y, const float* bias, float* out, int B, int N, int H, int W, int Npad,
                                float slope, ufr_stream_t stream) {
  UFR_REQUIRE(y && out, "rows -> nchw: null pointer");
  UFR_REQUIRE(B > 0 && N > 0 && H > 0 && W > 0 && Npad >= N && Npad % 32 == 0 && B < 65536,
              "rows -> nchw: bad shape (Npad must be a multiple of 32, >= N)");
  const dim3 grid((N + 31) / 32, (H * W + 63) / 64, B);
  rows_to_nchw_kernel<<<grid, 256, 0, ufr::as_stream(stream)>>>(y, bias, out, B, N, H * W, Npad, slope);
  return ufr::launched("rows_to_nchw_kernel");
}

extern "C" int ufr_conv_split_general(const void* x_planes, const void* w_planes, float* y, int B, int Hi, int Wi,
                                      int Cpad, int N, int KH, int KW, int stride, int pad, int products,
                                      int chunk_major, ufr_stream_t stream) {
  UFR_REQUIRE(x_planes && w_planes && y, "split conv (general): null pointer");
  UFR_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Cpad > 0 && Cpad % BK == 0 && N > 0 && N % GN == 0,
              "split conv (general): Cpad must be a multiple of 32, the output channels of 128");
  UFR_REQUIRE(KH > 0 && KW > 0 && KH <= 11 && KW <= 11 && stride > 0 && pad >= 0 && pad < KH && pad < KW,
              "split conv (general): bad kernel / stride / padding");
  const int Ho = (Hi + 2 * pad - KH) / stride + 1, Wo = (Wi + 2 * pad - KW) / stride + 1;
  UFR_REQUIRE(Ho > 0 && Wo > 0, "split conv (general): empty output");
  UFR_REQUIRE((long)B * Hi * Wi < (1L << 31) / 2, "split conv (general): too many pixels");
  UFR_REQUIRE(products == 6 || products == 3 || products == 1, "split conv (general): products must be 6, 3 or 1");
  const int M = B * Ho * Wo;
  const dim3 grid(N / GN, (M + GM - 1) / GM);
  hipStream_t st = ufr::as_stream(stream);
  const __bf16* a = static_cast<const __bf16*>(x_planes);
  const __bf16* b = static_cast<const __bf16*>(w_planes);
  const long Min = (long)B * Hi * Wi, K = (long)KH * KW * Cpad;
  const long rsA = chunk_major ? BK : Cpad, ksA = chunk_major ? Min * BK : BK;
  const long rsB = chunk_major ? BK : K, ksB = chunk_major ? (long)N * BK : BK;
#define UFR_GC_LAUNCH(P)                                                                                             \
  conv_split_general_kernel<P><<<grid, 256, 0, st>>>(a, b, y, B, Hi, Wi, Ho, Wo, Cpad, N, KH, KW, stride, pad, rsA, ksA, \
                                                     rsB, ksB, split_xcd_swizzle())
  if (products == 6) UFR_GC_LAUNCH(6);
  else if (products == 3) UFR_GC_LAUNCH(3);
  else UFR_GC_LAUNCH(1);
#undef UFR_GC_LAUNCH
  return ufr::launched("conv_split_general_kernel");
}

/* plan: 4 phases x (ntaps, oy0, ox0, w_off, then 16 x (dy, dx)) = 4 x 36 longs, as split_gemm.py::deconv_plan lays it out */
extern "C" int ufr_deconv_split(const void* x_planes_cm, const void* w_planes, float* y, int B, int Hi, int Wi, int Cpad,
                                int N, long w_plane_elems, const long* plan_host, int products, ufr_stream_t stream) {
  UFR_REQUIRE(x_planes_cm && w_planes && y && plan_host, "split deconv: null pointer");
  UFR_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Cpad > 0 && Cpad % BK == 0 && N > 0 && N % GN == 0 && w_plane_elems > 0,
              "split deconv: Cpad must be a multiple of 32, the output channels of 128");
  UFR_REQUIRE((long)B * Hi * Wi < (1L << 31) / 8, "split deconv: too many pixels");
  UFR_REQUIRE(products == 6 || products == 3 || products == 1, "split deconv: products must be 6, 3 or 1");
  DeconvPlan plan;
  for (int z = 0; z < 4; ++z) {
    const long* q = plan_host + z * 36;
    UFR_REQUIRE(q[0] >= 1 && q[0] <= 16 && (q[1] | 1) == 1 && (q[2] | 1) == 1 && q[3] >= 0 &&
                    q[3] + q[0] * (long)Cpad * N <= w_plane_elems,
                "split deconv: bad plan");
    plan.ntaps[z] = (int)q[0]; plan.oy0[z] = (int)q[1]; plan.ox0[z] = (int)q[2]; plan.w_off[z] = q[3];
    for (int t = 0; t < 16; ++t) {
      const long dy = t < q[0] ? q[4 + 2 * t] : 0, dx = t < q[0] ? q[5 + 2 * t] : 0;
      UFR_REQUIRE(dy >= -8 && dy <= 8 && dx >= -8 && dx <= 8, "split deconv: bad tap offset");
      plan.dy[z][t] = (int)dy; plan.dx[z][t] = (int)dx;
    }
  }
  const int M = B * Hi * Wi;
  const dim3 grid(N / GN, (M + GM - 1) / GM, 4);
  hipStream_t st = ufr::as_stream(stream);
  const __bf16* a = static_cast<const __bf16*>(x_planes_cm);
  const __bf16* b = static_cast<const __bf16*>(w_planes);
  if (products == 6) deconv_split_kernel<6><<<grid, 256, 0, st>>>(a, b, y, B, Hi, Wi, Cpad, N, w_plane_elems, plan, split_xcd_swizzle());
  else if (products == 3) deconv_split_kernel<3><<<grid, 256, 0, st>>>(a, b, y, B, Hi, Wi, Cpad, N, w_plane_elems, plan, split_xcd_swizzle());
  else deconv_split_kernel<1><<<grid, 256, 0, st>>>(a, b, y, B, Hi, Wi, Cpad, N, w_plane_elems, plan, split_xcd_swizzle());
  return ufr::launched("deconv_split_kernel");
}
