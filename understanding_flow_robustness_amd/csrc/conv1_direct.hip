// conv1_direct.hip -- FlowNetC's first layer, Conv2d(3, 64, 7, stride 2, padding 3) + bias + LeakyReLU (models/FlowNetC.py:22-30,
// :100-104; the block is models/submodules.py:18-46), straight from the RAW frames to the engine's activation planes:
//   normalize_correctly (float64 mean subtraction, FlowNetC.py:73-79)  ->  im2col in LDS  ->  six-product bf16 MFMA  ->
//   bias + LeakyReLU -> three bf16 planes of conv1 [3][2 chunks][N*H/2*W/2][32].
// It replaces conv1_pack_kernel + an igemm launch over the packed planes (K = 256 of which 147 real, 8 K steps per tile: the
// launch ran at 0.18 of the six-product ceiling, epilogue-sized) -- round 3's VERDICT item 2b.
//
// Bound: HBM.  Per frame set of N frames: 4*3*H*W*N bytes in, 6*64*(H/2)*(W/2)*N bytes of planes out (8 x 384 x 1280:
// 47 MB in, 377 MB out); the MFMA work (K padded 147 -> 224) is 0.07 ms at the six-product ceiling, under the write time.
//
// Structure (one 512-thread workgroup per CU, persistent over tiles of 8 x 32 output pixels):
//   * LDS holds the tile's input region, 21 rows x 69 columns x 3 channels, already mean-subtracted and split into the three
//     bf16 planes, pixel-major with the channels padded 3 -> 4: one pixel = 8 bytes.  The region starts at input column
//     2*X0 - 3, so the 7 taps of output pixel x start at LDS pixel 2*(x - X0): EVEN, i.e. 16-byte aligned -- a K group of
//     8 = (two adjacent input pixels) x (4 channels) is ONE aligned ds_read_b128.  A K tile of 32 = one kernel row ky: 4 pixel
//     pairs (kx = 0..7; kx = 7 and channel 3 carry zero weights).  K = 7 x 32 = 224.
//   * MFMA roles are swapped against the igemm: A = weights (16 output channels x 32 k), B = pixels (16 pixels x 32 k), so the
//     accumulator of a lane is FOUR CONSECUTIVE CHANNELS of one pixel: the epilogue needs no LDS transpose; round 6 pairs the two runs of
//     a tile row and exchanges between lane groups so that a lane writes 16 bytes (eight consecutive channels) per plane.
//   * a wave owns 16 output channels and keeps their whole weight image in registers (7 K tiles x 3 planes = 84 VGPRs, loaded
//     once per workgroup): no weight traffic in the loop, LDS is read for the pixels only (21 ds_read_b128 per 42 MFMAs).
//     Eight waves = 4 channel quarters x 2 halves of the tile's rows.
//   * double-buffered input tiles: the next tile's pixels are fetched into registers before this tile's MFMAs and written to
//     the other buffer after them; ONE workgroup barrier per tile.
#include <atomic>

#include "ufr_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int TY = 8, TX = 32;                         // output pixels of a tile
constexpr int RY = 2 * TY + 5, RX = 2 * TX + 5;        // input region: 21 x 69
constexpr int LX = 72;                                 // LDS pixels per region row (69 real + 3 zero)
constexpr int PLANE_B = RY * LX * 8;                   // bytes of one plane of one buffer: 12,096
constexpr int BUF_B = 3 * PLANE_B;                     // 36,288
constexpr int PX_PER_THREAD = (RY * RX + 511) / 512;   // 3
__device__ constexpr int PROD_W[6] = {2, 0, 1, 1, 0, 0};   // (weight plane, pixel plane) of each product, smallest first
__device__ constexpr int PROD_X[6] = {0, 2, 1, 0, 1, 0};

__device__ __forceinline__ void split3(float v, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)v;
  const float r1 = v - (float)a;
  b = (__bf16)r1;
  c = (__bf16)(r1 - (float)b);
}

struct Conv1Args {
  const float* fa; const float* fb; int Ba, N, H, W;
  const double* mean;
  const __bf16* wimg;            // [3 planes][7 ky][64 n][32 k] bf16, k = (kx >> 1) * 8 + (kx & 1) * 4 + c
  const float* bias; float slope;
  __bf16* out; long plane_stride; int out_chunk0;
  int tiles_x, tiles_y, tiles;
};

// BUF_: the output planes lie within 2 GB of their start -> stores through a raw buffer resource (see the epilogue)
template <bool BUF_>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv1_direct_kernel(const Conv1Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];           // [2 buffers][3 planes][RY][LX] pixels of 8 bytes
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = wave & 3, half = wave >> 2;                                     // channel quarter, row half of the tile
  const int li = lane & 15, g = lane >> 4;
  const int Hh = a.H >> 1, Wh = a.W >> 1;
  const long M = (long)a.N * Hh * Wh;

  // ---- once: zero both buffers (columns 69..71 and channel 3 stay zero for good), this wave's weights into registers
  for (int i = tid; i < 2 * BUF_B / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
  bf16x8 wf[7][3];
#pragma unroll
  for (int ky = 0; ky < 7; ++ky)
#pragma unroll
    for (int p = 0; p < 3; ++p)
      wf[ky][p] = *reinterpret_cast<const bf16x8*>(a.wimg + (((long)p * 7 + ky) * 64 + q * 16 + li) * 32 + g * 8);
  float bias4[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bias4[r] = a.bias[q * 16 + g * 4 + r];
  const double m0 = a.mean[0], m1 = a.mean[1], m2 = a.mean[2];
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, BUF_ ? (int)(6L * a.plane_stride) : 0, 0x00020000);

  // ---- staging: thread -> PX_PER_THREAD pixels of the region (consecutive threads = consecutive columns)
  float pre[PX_PER_THREAD][3];
  unsigned pre_ok = 0;                          // bit i: pixel i of `pre` lies inside the frame
  auto tile_coords = [&](int t, int& n, int& Y0, int& X0) {
    const int tx = t % a.tiles_x, r = t / a.tiles_x;
    n = r / a.tiles_y;
    Y0 = (r - n * a.tiles_y) * TY;
    X0 = tx * TX;
  };
  auto fetch = [&](int t) {                     // global loads of tile t's region into registers (zeros outside the frame)
    int n, Y0, X0;
    tile_coords(t, n, Y0, X0);
    const float* img = n < a.Ba ? a.fa + (long)n * 3 * a.H * a.W : a.fb + (long)(n - a.Ba) * 3 * a.H * a.W;
    const long HW = (long)a.H * a.W;
    pre_ok = 0;
#pragma unroll
    for (int i = 0; i < PX_PER_THREAD; ++i) {
      const int idx = tid + 512 * i, ry = idx / RX, rx = idx - ry * RX;
      const int iy = 2 * Y0 - 3 + ry, ix = 2 * X0 - 3 + rx;
      const bool ok = idx < RY * RX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const long o = (long)iy * a.W + ix;
      pre[i][0] = ok ? img[o] : 0.f;
      pre[i][1] = ok ? img[o + HW] : 0.f;
      pre[i][2] = ok ? img[o + 2 * HW] : 0.f;
      pre_ok |= ok ? 1u << i : 0u;              // (out-of-frame pixels are zero AFTER the mean subtraction: the convolution pads
    }                                           //  the normalised image)
  };
  auto stage = [&](int buf) {                   // registers -> mean subtraction -> three planes of buffer `buf`
    unsigned char* base = lds + buf * BUF_B;
#pragma unroll
    for (int i = 0; i < PX_PER_THREAD; ++i) {
      const int idx = tid + 512 * i;
      if (idx >= RY * RX) continue;
      const int ry = idx / RX, rx = idx - ry * RX;
      const bool ok = (pre_ok >> i) & 1u;
      const float v0 = ok ? (float)((double)pre[i][0] - m0) : 0.f;
      const float v1 = ok ? (float)((double)pre[i][1] - m1) : 0.f;
      const float v2 = ok ? (float)((double)pre[i][2] - m2) : 0.f;
      bf16x4 p0, p1, p2;
      __bf16 x, y, z;
      split3(v0, x, y, z); p0[0] = x; p1[0] = y; p2[0] = z;
      split3(v1, x, y, z); p0[1] = x; p1[1] = y; p2[1] = z;
      split3(v2, x, y, z); p0[2] = x; p1[2] = y; p2[2] = z;
      p0[3] = (__bf16)0.f; p1[3] = (__bf16)0.f; p2[3] = (__bf16)0.f;
      unsigned char* d = base + (ry * LX + rx) * 8;
      *reinterpret_cast<bf16x4*>(d) = p0;
      *reinterpret_cast<bf16x4*>(d + PLANE_B) = p1;
      *reinterpret_cast<bf16x4*>(d + 2 * PLANE_B) = p2;
    }
  };

  int t = blockIdx.x;
  if (t >= a.tiles) return;
  fetch(t);
  __syncthreads();                              // the zero fill is complete
  stage(0);
  int tn = t + gridDim.x;
  if (tn < a.tiles) fetch(tn);
  __syncthreads();

  for (int it = 0; t < a.tiles; ++it, t = tn, tn += gridDim.x) {
    const int buf = it & 1;
    int n, Y0, X0;
    tile_coords(t, n, Y0, X0);
    const unsigned char* src = lds + buf * BUF_B;
    // ---- this wave: rows half*4 .. half*4 + 3 of the tile, both 16-pixel halves of each row, its 16 channels.
    // 56 steps (8 runs of 16 pixels x 7 kernel rows), fully unrolled: the three pixel fragments of step s + 2 are read while
    // step s multiplies (a ring of three fragment sets), and the MFMAs rotate over FOUR accumulators so that no MFMA waits
    // for the one issued before it (a chain of dependent 16x16x32 MFMAs runs at half rate).
    const unsigned char* base = src + ((2 * half * 4) * LX + 2 * (li + g)) * 8;
    bf16x8 xf[3][3];
    auto read_step = [&](int s, bf16x8 (&dst)[3]) {
      const int pt = s / 7, ky = s - 7 * pt;
      const unsigned char* px = base + ((2 * (pt >> 1) + ky) * LX + 2 * ((pt & 1) * 16)) * 8;
#pragma unroll
      for (int p = 0; p < 3; ++p) dst[p] = *reinterpret_cast<const bf16x8*>(px + p * PLANE_B);
    };
    read_step(0, xf[0]);
    read_step(1, xf[1]);
    // the epilogue of run pt - 1 (bias, LeakyReLU, the three-plane split, three 8-byte stores: ~70 vector instructions) is emitted
    // INSIDE run pt's MFMA stream: an MFMA holds the issue port for 8 of its 16 cycles, so two vector instructions per MFMA
    // ride in its shadow (the sched_group_barrier pattern below asks for exactly that interleaving)
    // Round 6: 16-byte stores.  A lane's accumulator is four consecutive channels of one pixel; the two 16-pixel runs of a tile row (pt even /
    // odd) are paired: lane groups g and g ^ 1 (lanes l and l ^ 16) hold channels 4g .. 4g + 3 of the SAME pixel, so one exchange gives the
    // even group eight consecutive channels of its pixel of the even run and the odd group eight of its pixel of the odd run: three 16-byte
    // stores per lane and pair of runs instead of six 8-byte ones (8-byte accesses run at 0.54 - 0.70 of the 16-byte rate,
    // MI355X_MICROARCH.md).  The even run's epilogue only keeps its four activated values (`vkeep`), the odd run's does the exchange.
    float vkeep[4] = {0.f, 0.f, 0.f, 0.f};
    auto epilogue = [&](int pt, const f32x4 (&acc)[4]) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]) + bias4[r];
        v[r] = v[r] > 0.f ? v[r] : v[r] * a.slope;
      }
      if (!(pt & 1)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) vkeep[r] = v[r];
        return;
      }
      const bool even = !(g & 1);
      float out8[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float got = __shfl_xor(even ? v[r] : vkeep[r], 16);   // the partner group's four channels of MY pixel
        out8[r] = even ? vkeep[r] : got;                            // even group: its pixel of the even run, channels 4g .. 4g + 7
        out8[4 + r] = even ? got : v[r];                            // odd group: its pixel of the odd run, channels 4(g - 1) .. 4(g - 1) + 7
      }
      bf16x8 o0, o1, o2;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        __bf16 b0, b1, b2;
        split3(out8[r], b0, b1, b2);
        o0[r] = b0; o1[r] = b1; o2[r] = b2;
      }
      const int yy = half * 4 + (pt >> 1);
      const int y = Y0 + yy, x = X0 + (even ? 0 : 16) + li;
      const long m = ((long)n * Hh + y) * Wh + x;
      const long e = ((long)(a.out_chunk0 + (q >> 1)) * M + m) * 32 + (q & 1) * 16 + (g & 2) * 4;  // element inside plane 0
      const bool live = y < Hh && x < Wh;
      if constexpr (BUF_) {
        // unconditional buffer stores (a pixel outside the grid gets an out-of-range offset: the hardware drops it), so that the
        // compiler can COUNT the stores between the prefetched loads and their use, not drain them
        const unsigned off = live ? (unsigned)(e * 2) : 0x80000000u;
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), rsrc, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), rsrc, off, (unsigned)(a.plane_stride * 2), 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o2), rsrc, off, (unsigned)(a.plane_stride * 4), 0);
      } else if (live) {
        __bf16* o = a.out + e;
        *reinterpret_cast<bf16x8*>(o) = o0;
        *reinterpret_cast<bf16x8*>(o + a.plane_stride) = o1;
        *reinterpret_cast<bf16x8*>(o + 2 * a.plane_stride) = o2;
      }
    };
    f32x4 acc_prev[4];
#pragma unroll
    for (int pt = 0; pt < 8; ++pt) {
      f32x4 acc[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (pt > 0) epilogue(pt - 1, acc_prev);
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        const int s = pt * 7 + ky;
        if (s + 2 < 56) read_step(s + 2, xf[(s + 2) % 3]);
#pragma unroll
        for (int tp = 0; tp < 6; ++tp) {
          const int c = (ky * 6 + tp) & 3;
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky][PROD_W[tp]], xf[s % 3][PROD_X[tp]], acc[c], 0, 0, 0);
        }
      }
      // 42 MFMAs, each followed by two vector instructions of the previous run's epilogue; the 21 fragment reads spread between
#pragma unroll
      for (int i = 0; i < 42; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);     // two VALU
        if (i & 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // a DS read every other MFMA
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc_prev[c] = acc[c];
    }
    epilogue(7, acc_prev);
    // ---- the next tile: its pixels (fetched before this tile's MFMAs) into the other buffer, then fetch the one after.
    // (Staging it BEFORE this tile's MFMA stream instead -- legal: the other buffer's last readers finished at the previous barrier --
    // so that the mean subtraction / split / LDS writes could ride under the MFMAs measured the same: 5.686 / 5.668 ms against 5.674 with
    // this order in one call, gpurun r5_call24.  VERDICT r4 item 4b's two-wave-group form would have to re-deal rows AND channels.)
    if (tn < a.tiles) {
      stage(buf ^ 1);
      if (tn + (int)gridDim.x < a.tiles) fetch(tn + gridDim.x);
    }
    __syncthreads();                            // buffer buf^1 is complete; buffer buf is free for the tile after next
  }
}

}  // namespace

extern "C" int ufr_conv1_direct(const float* frames_a, const float* frames_b, int Ba, int Bb, int H, int W, const double* mean,
                                const void* wimg, const float* bias, float slope, void* out_planes, long plane_stride,
                                int out_chunk0, ufr_stream_t stream) {
  UFR_REQUIRE(frames_a && mean && wimg && bias && out_planes && (frames_b || Bb == 0), "conv1 direct: null pointer");
  UFR_REQUIRE(Ba > 0 && Bb >= 0 && H > 0 && W > 0 && !(H & 1) && !(W & 1) && plane_stride > 0 && out_chunk0 >= 0, "conv1 direct: bad shape");
  Conv1Args a;
  a.fa = frames_a; a.fb = frames_b; a.Ba = Ba; a.N = Ba + Bb; a.H = H; a.W = W; a.mean = mean;
  a.wimg = static_cast<const __bf16*>(wimg); a.bias = bias; a.slope = slope;
  a.out = static_cast<__bf16*>(out_planes); a.plane_stride = plane_stride; a.out_chunk0 = out_chunk0;
  a.tiles_x = ((W >> 1) + TX - 1) / TX; a.tiles_y = ((H >> 1) + TY - 1) / TY;
  const long tiles = (long)a.N * a.tiles_x * a.tiles_y;
  UFR_REQUIRE(tiles < (1L << 30) && (long)a.N * (H >> 1) * (W >> 1) < (1L << 30), "conv1 direct: too many pixels");
  a.tiles = (int)tiles;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ufr::fail(UFR_ELAUNCH, "conv1 direct: no current device");
  static std::atomic<bool> raised[64] = {};      // (published with release / acquire; the slow path is ufr::ensure_dynamic_lds' mutex)
  if (!raised[dev].load(std::memory_order_acquire)) {
    hipError_t e = ufr::ensure_dynamic_lds(reinterpret_cast<const void*>(conv1_direct_kernel<true>), 2 * BUF_B);
    if (e == hipSuccess) e = ufr::ensure_dynamic_lds(reinterpret_cast<const void*>(conv1_direct_kernel<false>), 2 * BUF_B);
    if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "conv1 direct: %s", hipGetErrorString(e));
    raised[dev].store(true, std::memory_order_release);
  }
  const int grid = (int)(tiles < ufr::kNumCU ? tiles : ufr::kNumCU);          // one persistent workgroup per CU
  // every plane the kernel writes must lie inside the buffer resource: three planes of at least (out_chunk0 + 2) chunks
  if (6L * plane_stride < 0x7fffffffL) conv1_direct_kernel<true><<<grid, 512, 2 * BUF_B, ufr::as_stream(stream)>>>(a);
  else conv1_direct_kernel<false><<<grid, 512, 2 * BUF_B, ufr::as_stream(stream)>>>(a);
  return ufr::launched("conv1_direct_kernel");
}
