// split_gemm.hip -- float32-accurate matrix product on the bf16 matrix cores (the round-2 lever).
//
// Not on the product path yet: this is the measured building block for replacing MIOpen's fp32 convolutions
// (which top out at the 157 TFLOP/s fp32 peak) by implicit GEMMs on the 2.5 PFLOP/s bf16 MFMA pipe.
//
//   a = a1 + a2 + a3 exactly, each piece a bf16 (8 significant bits each, 24 together = a float32 significand)
//   a*b ~= a1b1 + (a1b2 + a2b1) + (a2b2 + a1b3 + a3b1)          six bf16 products, float32 accumulation
//
// tools/probe_split_precision.py (CPU): the six-product sum is as close to the float64 result as a plain fp32
// GEMM is (2.6e-7 vs 3.8e-7 of max |out| at K = 4257); three products give 5e-6, one product 2.7e-3.
//
// ufr_split_bf16x3: x[n] -> planes[3][n] bf16 (round-to-nearest-even, residuals exact in float32).
// ufr_gemm_split_nt: C[M,N] = A[M,K] * B[N,K]^T from pre-split planes.  128x128x32 tile, 4 waves as 2x2, each
// wave a 64x64 block of 4x4 `v_mfma_f32_16x16x32_bf16` accumulators.  One LDS image per (operand, plane):
// [128 rows][32 k] bf16 = 64 B per row, the 16-byte k-chunk XOR-swizzled with (row >> 1) & 3, which makes
// every ds_read_b128 lane group of a fragment read hit 16 distinct bank quads (MI355X_MICROARCH.md, LDS
// table).  The next K-tile's global loads are issued before the MFMA block and land in registers while it
// runs.  Per wave and K-tile: 96 MFMAs (1536 cycles) against 24 ds_read_b128 + 12 global loads.
#include "ufr_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ void split3_kernel(const float* __restrict__ x, __bf16* __restrict__ planes, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    const __bf16 p1 = (__bf16)v;
    const float r1 = v - (float)p1;
    const __bf16 p2 = (__bf16)r1;
    const float r2 = r1 - (float)p2;
    planes[i] = p1;
    planes[n + i] = p2;
    planes[2 * n + i] = (__bf16)r2;
  }
}

constexpr int BM = 128, BN = 128, BK = 32;

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

// (a plane, b plane) of each product, smallest magnitude first
__device__ constexpr int PROD_A[6] = {2, 0, 1, 1, 0, 0};
__device__ constexpr int PROD_B[6] = {0, 2, 1, 0, 1, 0};

template <int NPROD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_split_nt_kernel(const __bf16* __restrict__ Ap, const __bf16* __restrict__ Bp,
                                                            float* __restrict__ C, int M, int N, int K, long rsA,
                                                            long ksA, long rsB, long ksB, int swz) {
  constexpr int NPL = NPROD == 1 ? 1 : (NPROD == 3 ? 2 : 3);   // planes needed per operand
  constexpr int FIRST = 6 - NPROD;                              // NPROD leading-order products = the last ones
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * NPL][BM * BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  int tile_x, tile_y;
  split_tile_of_block(swz, tile_x, tile_y);
  const int bm = tile_y * BM, bn = tile_x * BN;
  const size_t planeA = (size_t)M * K, planeB = (size_t)N * K;

  // staging: 512 16-byte chunks per image, two per thread
  const int srow0 = tid >> 2, sch = tid & 3;
  u32x4 sa[NPL][2], sb[NPL][2];
  // row stride rs / K-chunk stride ks: (K, 32) for row-major [rows][K] planes, (32, rows*32) for the
  // chunk-major [K/32][rows][32] layout whose tiles are contiguous 8 KB runs (full 128-byte lines)
  const __bf16* ga = Ap + (size_t)(bm + srow0) * rsA + sch * 8;
  const __bf16* gb = Bp + (size_t)(bn + srow0) * rsB + sch * 8;
  const int soff0 = srow0 * BK + ((sch ^ ((srow0 >> 1) & 3)) << 3);   // row + 64 keeps (row >> 1) & 3
#define UFR_SG_LOAD(k0)                                                                                       \
  _Pragma("unroll") for (int p = 0; p < NPL; ++p) _Pragma("unroll") for (int i = 0; i < 2; ++i) {             \
    sa[p][i] = *reinterpret_cast<const u32x4*>(ga + p * planeA + (size_t)(64 * i) * rsA + ((k0) / BK) * ksA); \
    sb[p][i] = *reinterpret_cast<const u32x4*>(gb + p * planeB + (size_t)(64 * i) * rsB + ((k0) / BK) * ksB); \
  }
#define UFR_SG_STORE()                                                                              \
  _Pragma("unroll") for (int p = 0; p < NPL; ++p) _Pragma("unroll") for (int i = 0; i < 2; ++i) {   \
    *reinterpret_cast<u32x4*>(&lds[p][soff0 + 64 * i * BK]) = sa[p][i];                             \
    *reinterpret_cast<u32x4*>(&lds[NPL + p][soff0 + 64 * i * BK]) = sb[p][i];                       \
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment address inside an image: row = base + (lane & 15), chunk = lane >> 4, swizzle from the row's low bits
  const int frow = lane & 15;
  const int foff = frow * BK + ((((lane >> 4)) ^ ((frow >> 1) & 3)) << 3);

  UFR_SG_LOAD(0)
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();          // everyone is done reading the previous tile
    UFR_SG_STORE()
    __syncthreads();
    if (k0 + BK < K) { UFR_SG_LOAD(k0 + BK) }
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

  // C/D layout of the 16x16 forms: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        C[(size_t)(bm + wr * 64 + m * 16 + (lane >> 4) * 4 + j) * N + bn + wc * 64 + n * 16 + (lane & 15)] = acc[m][n][j];
}

#undef UFR_SG_LOAD
#undef UFR_SG_STORE

// ---- 3x3 stride-1 pad-1 convolution as an implicit GEMM over the same tile machinery ---------------------------
// activations: three bf16 planes, NHWC with the channels zero-padded to a multiple of 32: Xp[3][B*H*W][Cpad]
// weights:     Wp[3][N][9][Cpad] (tap = ky*3 + kx), pre-split once (they are frozen during an attack)
// output:      Y[B*H*W][N] float32 (NHWC).  K tiles run tap-major; a tile row is one pixel's 32-channel chunk of
// one tap, zero when the tap falls outside the frame.
template <int NPROD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NPROD == 6 ? 2 : 3, NPROD == 6 ? 2 : 3))) void conv3x3_split_kernel(
    const __bf16* __restrict__ Xp, const __bf16* __restrict__ Wp, float* __restrict__ Y, int B, int H, int W, int Cpad,
    int N, long rsA, long ksA, long rsB, long ksB, int swz) {
  constexpr int NPL = NPROD == 1 ? 1 : (NPROD == 3 ? 2 : 3);
  constexpr int FIRST = 6 - NPROD;
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * NPL][BM * BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  int tile_x, tile_y;
  split_tile_of_block(swz, tile_x, tile_y);
  const int bm = tile_y * BM, bn = tile_x * BN;
  const int M = B * H * W, K = 9 * Cpad, KC = Cpad / BK, KT = 9 * KC;
  const size_t planeA = (size_t)M * Cpad, planeB = (size_t)N * K;

  const int srow0 = tid >> 2, sch = tid & 3;
  int pm[2], py[2], px[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    pm[i] = bm + srow0 + 64 * i;
    px[i] = pm[i] % W;
    py[i] = pm[i] < M ? (pm[i] / W) % H : -4;      // rows past the end never pass the bounds test
  }
  u32x4 sa[NPL][2], sb[NPL][2];
  const __bf16* gb = Wp + (size_t)(bn + srow0) * rsB + sch * 8;
  const int soff0 = srow0 * BK + ((sch ^ ((srow0 >> 1) & 3)) << 3);
#define UFR_SC_LOAD(kt)                                                                               \
  {                                                                                                   \
    const int tap = (kt) / KC, kc = (kt) - tap * KC, dyo = tap / 3 - 1, dxo = tap % 3 - 1;             \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                   \
      const bool ok = (unsigned)(py[i] + dyo) < (unsigned)H && (unsigned)(px[i] + dxo) < (unsigned)W;  \
      const __bf16* src = Xp + (size_t)(ok ? pm[i] + dyo * W + dxo : 0) * rsA + kc * ksA + sch * 8;    \
      _Pragma("unroll") for (int p = 0; p < NPL; ++p) {                                               \
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + p * planeA);                            \
        sa[p][i] = ok ? v : u32x4{0u, 0u, 0u, 0u};                                                    \
        sb[p][i] = *reinterpret_cast<const u32x4*>(gb + p * planeB + (size_t)(64 * i) * rsB + (kt) * ksB);    \
      }                                                                                               \
    }                                                                                                 \
  }
#define UFR_SC_STORE()                                                                              \
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

  UFR_SC_LOAD(0)
  for (int kt = 0; kt < KT; ++kt) {
    __syncthreads();
    UFR_SC_STORE()
    __syncthreads();
    if (kt + 1 < KT) UFR_SC_LOAD(kt + 1)
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
#undef UFR_SC_LOAD
#undef UFR_SC_STORE

// x[B][C][H*W] float32 -> planes[3][B*H*W][Cpad] bf16 (channels C..Cpad-1 zero).  Tile: 32 channels x 64 pixels
// through LDS so that both the NCHW reads and the NHWC 16-byte writes are contiguous.
__global__ __launch_bounds__(256) void nchw_to_nhwc_split3_kernel(const float* __restrict__ x, __bf16* __restrict__ planes,
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
  const size_t plane = (size_t)B * HW * Cpad;
  __bf16* dst = planes + ((size_t)b * HW + p0 + p) * Cpad + c0 + ch * 8;
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

}  // namespace

extern "C" int ufr_split_bf16x3(const float* x, void* planes, long n, ufr_stream_t stream) {
  UFR_REQUIRE(x && planes, "split bf16x3: null pointer");
  UFR_REQUIRE(n > 0, "split bf16x3: bad size");
  split3_kernel<<<ufr::stream_grid(n, 256), 256, 0, ufr::as_stream(stream)>>>(x, static_cast<__bf16*>(planes), n);
  return ufr::launched("split3_kernel");
}

extern "C" int ufr_gemm_split_nt(const void* a_planes, const void* b_planes, float* c, int M, int N, int K,
                                 int products, int chunk_major, ufr_stream_t stream) {
  UFR_REQUIRE(a_planes && b_planes && c, "split gemm: null pointer");
  UFR_REQUIRE(M > 0 && N > 0 && K > 0 && M % BM == 0 && N % BN == 0 && K % BK == 0,
              "split gemm: M and N must be multiples of 128, K of 32");
  UFR_REQUIRE(products == 6 || products == 3 || products == 1, "split gemm: products must be 6, 3 or 1");
  const dim3 grid(N / BN, M / BM);
  hipStream_t st = ufr::as_stream(stream);
  const __bf16* a = static_cast<const __bf16*>(a_planes);
  const __bf16* b = static_cast<const __bf16*>(b_planes);
  const int swz = split_xcd_swizzle();
  const long rsA = chunk_major ? BK : K, ksA = chunk_major ? (long)M * BK : BK;
  const long rsB = chunk_major ? BK : K, ksB = chunk_major ? (long)N * BK : BK;
  if (products == 6) gemm_split_nt_kernel<6><<<grid, 256, 0, st>>>(a, b, c, M, N, K, rsA, ksA, rsB, ksB, swz);
  else if (products == 3) gemm_split_nt_kernel<3><<<grid, 256, 0, st>>>(a, b, c, M, N, K, rsA, ksA, rsB, ksB, swz);
  else gemm_split_nt_kernel<1><<<grid, 256, 0, st>>>(a, b, c, M, N, K, rsA, ksA, rsB, ksB, swz);
  return ufr::launched("gemm_split_nt_kernel");
}

extern "C" int ufr_nchw_to_nhwc_split3(const float* x, void* planes, int B, int C, int H, int W, int Cpad,
                                       ufr_stream_t stream) {
  UFR_REQUIRE(x && planes, "nchw -> nhwc split: null pointer");
  UFR_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C && Cpad % BK == 0 && B < 65536,
              "nchw -> nhwc split: bad shape (Cpad must be a multiple of 32, >= C)");
  const dim3 grid(Cpad / 32, (H * W + 63) / 64, B);
  nchw_to_nhwc_split3_kernel<<<grid, 256, 0, ufr::as_stream(stream)>>>(x, static_cast<__bf16*>(planes), B, C, H * W, Cpad);
  return ufr::launched("nchw_to_nhwc_split3_kernel");
}

extern "C" int ufr_conv3x3_split(const void* x_planes, const void* w_planes, float* y, int B, int H, int W, int Cpad,
                                 int N, int products, int chunk_major, ufr_stream_t stream) {
  UFR_REQUIRE(x_planes && w_planes && y, "split conv: null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && Cpad > 0 && Cpad % BK == 0 && N > 0 && N % BN == 0,
              "split conv: Cpad must be a multiple of 32, the output channels of 128");
  UFR_REQUIRE((long)B * H * W < (1L << 31) / 2, "split conv: too many pixels");
  UFR_REQUIRE(products == 6 || products == 3 || products == 1, "split conv: products must be 6, 3 or 1");
  const int M = B * H * W;
  const dim3 grid(N / BN, (M + BM - 1) / BM);
  hipStream_t st = ufr::as_stream(stream);
  const __bf16* a = static_cast<const __bf16*>(x_planes);
  const __bf16* b = static_cast<const __bf16*>(w_planes);
  const int swz = split_xcd_swizzle();
  const long rsA = chunk_major ? BK : Cpad, ksA = chunk_major ? (long)M * BK : BK;
  const long rsB = chunk_major ? BK : 9L * Cpad, ksB = chunk_major ? (long)N * BK : BK;
  if (products == 6) conv3x3_split_kernel<6><<<grid, 256, 0, st>>>(a, b, y, B, H, W, Cpad, N, rsA, ksA, rsB, ksB, swz);
  else if (products == 3) conv3x3_split_kernel<3><<<grid, 256, 0, st>>>(a, b, y, B, H, W, Cpad, N, rsA, ksA, rsB, ksB, swz);
  else conv3x3_split_kernel<1><<<grid, 256, 0, st>>>(a, b, y, B, H, W, Cpad, N, rsA, ksA, rsB, ksB, swz);
  return ufr::launched("conv3x3_split_kernel");
}
