// igemm.hip -- ONE implicit-GEMM convolution kernel for the native FlowNetC head (flownetc_engine.py): every
// Conv2d / ConvTranspose2d of models/FlowNetC.py:22-50 (blocks of models/submodules.py:18-46, :75-82) and every one of
// their data gradients, float32-accurate on the bf16 matrix cores.
//
// Arithmetic: a float32 value is held as THREE bf16 planes, v = p0 + p1 + p2 exactly; a product is the
// six leading bf16 products a0b0 + (a0b1 + a1b0) + (a1b1 + a0b2 + a2b0) accumulated in float32 by
// `v_mfma_f32_16x16x32_bf16` (2.5 PFLOP/s dense / 6 = 417 TFLOP/s fp32-equivalent against the 157 TFLOP/s fp32 peak).
// Error against float64 equals a plain fp32 GEMM's (profiles/r1_split_conv_accuracy.jsonl).
//
// Layout in HBM -- activations never leave it between layers:
//   activation planes  bf16 [3][chunks][M][32]   chunk-major NHWC: M = B*H*W pixels, 32 channels per chunk; a K tile
//                                                of the GEMM (128 pixels x 32 channels of one tap) is one contiguous
//                                                8 KB run per plane.  A concatenation (torch.cat in FlowNetC.forward)
//                                                is a chunk offset into a wider buffer.
//   weights            bf16 [3][taps*KC][Npad][32] per phase, pre-split once (frozen during an attack)
//   gradient sums      f32  [chunks][M][32]      same pixel / channel order
//
// Geometry is data: the tile rows are the cells (b, y, x) of a ROW GRID; tap t reads input pixel
// (y*in_sy + dy[t], x*in_sx + dx[t]) and the result lands on output pixel (y*out_sy + oy0, x*out_sx + ox0):
//   Conv2d k, stride s, padding p      rows = output grid, in_s = s, (dy, dx) = (ky - p, kx - p), out_s = 1
//   its data gradient, s = 1           the same with flipped, transposed weights
//   ConvTranspose2d(., ., 4, 2, 1)     rows = the COARSE grid, four phases (oy0, ox0) of 4 taps each, out_s = 2
//   data gradient of a stride-2 conv   the same phase form (1 + 2 + 2 + 4 taps for k = 3)
//   data gradient of that deconv       rows = coarse grid, in_s = 2, 16 taps
// A per-sample column origin (device memory) restricts the rows to a band around the patch (band_conv.py's idea
// without its gather / scatter copies: taps read the full-frame planes directly).
//
// Epilogues: forward  = bias + LeakyReLU -> planes;  gradient = (+ fp32 addend) * LeakyReLU'(mask) -> planes and / or
// fp32.  Split-K (small layers: 6x20 and 12x40 grids cannot fill 256 CUs with 128x128 tiles) writes fp32 slabs that a
// second kernel adds in a fixed order -- no float atomics anywhere, results are bit-reproducible.
#include <atomic>
#include <cstdlib>

#include "ufr_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int BM = 128, BN = 128, BK = 32;
__device__ constexpr int PROD_A[6] = {2, 0, 1, 1, 0, 0};   // (a plane, b plane) of each product, smallest first
__device__ constexpr int PROD_B[6] = {0, 2, 1, 0, 1, 0};

struct RowGeom {            // what the kernels need to turn a tile row into pixels
  int B, Hr, Wr, M;         // row grid, M = B*Hr*Wr
  const int* row_x0;        // optional per-sample first column of the band: x = xr + row_x0[b*stride] / div
  int row_x0_stride, row_x0_div;
  int Ho, Wo, out_sy, out_sx;
};

__device__ __forceinline__ void split3(float v, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)v;
  const float r1 = v - (float)a;
  b = (__bf16)r1;
  c = (__bf16)(r1 - (float)b);
}

// One output element through the epilogue.  `pout` = output pixel index, `n` = output channel.
struct Epilogue {
  const float* bias; float slope; int act;
  const float* add; int add_chunk0;
  const __bf16* mask; int mask_chunk0;
  __bf16* out_planes; long out_plane_stride; int out_chunk0;
  float* out_f32; int out_f32_chunk0;
  int planes_chunks, f32_first_chunk; // planes: output chunks < planes_chunks only; fp32: chunks >= f32_first_chunk only
  float* out_rm; long out_ld;         // optional ROW-MAJOR fp32 output [Mout][out_ld] (RAFT's all-pairs volume: a row = one pixel's N sums)
  float* tail; int tail_n0, tail_acc; // columns >= tail_n0 (a multiple of 32): raw sums to (tail_acc: added onto) this fp32 chunk-major tensor
  long Mout; int N, Nchunks32;        // channels < Nchunks32*32 are written (zeros beyond N: the chunk's padding)
  int nostore;                        // MEASUREMENT ONLY (UFR_IGEMM_DEBUG_SKIP_EPILOGUE=2): the epilogue computes everything and stores nothing
};

__device__ __forceinline__ void epilogue_store(const Epilogue& e, long pout, int n, float v) {
  if (n >= e.Nchunks32 * 32) return;
  if (e.tail && n >= e.tail_n0) {
    float* tp = e.tail + ((long)((n - e.tail_n0) >> 5) * e.Mout + pout) * 32 + (n & 31);
    *tp = (n < e.N ? v : 0.f) + (e.tail_acc ? *tp : 0.f);
    return;
  }
  const long cm = ((long)(n >> 5) * e.Mout + pout) * 32 + (n & 31);          // chunk-major element inside a tensor
  if (e.act) {
    if (e.add) v += e.add[(long)e.add_chunk0 * e.Mout * 32 + cm];
    v += (n < e.N) ? e.bias[n] : 0.f;
    v = v > 0.f ? v : v * e.slope;
  } else {
    if (e.add) v += e.add[(long)e.add_chunk0 * e.Mout * 32 + cm];
    if (e.mask) v = ((float)e.mask[(long)e.mask_chunk0 * e.Mout * 32 + cm] > 0.f) ? v : v * e.slope;
  }
  if (n >= e.N) v = 0.f;
  if (e.out_rm && n < e.N) e.out_rm[pout * e.out_ld + n] = v;
  if (e.out_f32 && (n >> 5) >= e.f32_first_chunk) e.out_f32[(long)e.out_f32_chunk0 * e.Mout * 32 + cm] = v;
  if (e.out_planes && (n >> 5) < e.planes_chunks) {
    __bf16 a, b, c;
    split3(v, a, b, c);
    __bf16* o = e.out_planes + (long)e.out_chunk0 * e.Mout * 32 + cm;
    o[0] = a;
    o[e.out_plane_stride] = b;
    o[2 * e.out_plane_stride] = c;
  }
}

__device__ __forceinline__ long out_pixel(const RowGeom& g, int pm, int oy0, int ox0) {
  const int hw = g.Hr * g.Wr;
  const int b = pm / hw, r = pm - b * hw, yr = r / g.Wr, xr = r - yr * g.Wr;
  const int xg = xr + (g.row_x0 ? g.row_x0[b * g.row_x0_stride] / g.row_x0_div : 0);
  return ((long)b * g.Ho + (yr * g.out_sy + oy0)) * g.Wo + (xg * g.out_sx + ox0);
}

// Eight consecutive channels of one output pixel through the epilogue (n0 a multiple of 8): 16-byte accesses.
__device__ __forceinline__ void epilogue_store8(const Epilogue& e, long pout, int n0, float v[8]) {
  if (n0 >= e.Nchunks32 * 32) return;
  if (e.tail && n0 >= e.tail_n0) {       // a later layer's partial sum over the same input chunks (DenseNet forward push)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (n0 + j >= e.N) v[j] = 0.f;
    float* tp = e.tail + ((long)((n0 - e.tail_n0) >> 5) * e.Mout + pout) * 32 + (n0 & 31);
    if (e.tail_acc) {                    // the tail lives in a running sum (a gradient with other contributors)
      const float4 o0 = *reinterpret_cast<const float4*>(tp), o1 = *reinterpret_cast<const float4*>(tp + 4);
      v[0] += o0.x; v[1] += o0.y; v[2] += o0.z; v[3] += o0.w; v[4] += o1.x; v[5] += o1.y; v[6] += o1.z; v[7] += o1.w;
    }
    *reinterpret_cast<float4*>(tp) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(tp + 4) = make_float4(v[4], v[5], v[6], v[7]);
    return;
  }
  const long cm = ((long)(n0 >> 5) * e.Mout + pout) * 32 + (n0 & 31);
  if (e.act) {
    if (e.add) {                         // the partial sum an earlier launch left in its tail columns
      const float* ap = e.add + (long)e.add_chunk0 * e.Mout * 32 + cm;
      const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
      v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] += (n0 + j < e.N) ? e.bias[n0 + j] : 0.f;
      v[j] = v[j] > 0.f ? v[j] : v[j] * e.slope;
    }
  } else {
    if (e.add) {
      const float* ap = e.add + (long)e.add_chunk0 * e.Mout * 32 + cm;
      const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
      v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
    }
    if (e.mask) {
      const bf16x8 m = *reinterpret_cast<const bf16x8*>(e.mask + (long)e.mask_chunk0 * e.Mout * 32 + cm);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ((float)m[j] > 0.f) ? v[j] : v[j] * e.slope;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (n0 + j >= e.N) v[j] = 0.f;
  if (e.out_rm && n0 < e.N) {              // (N is a multiple of 8 there: the host checks it)
    float* op = e.out_rm + pout * e.out_ld + n0;
    *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
  if (e.out_f32 && (n0 >> 5) >= e.f32_first_chunk) {
    float* op = e.out_f32 + (long)e.out_f32_chunk0 * e.Mout * 32 + cm;
    *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
  if (e.out_planes && (n0 >> 5) < e.planes_chunks) {
    bf16x8 q0, q1, q2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __bf16 x, y, z;
      split3(v[j], x, y, z);
      q0[j] = x; q1[j] = y; q2[j] = z;
    }
    __bf16* o = e.out_planes + (long)e.out_chunk0 * e.Mout * 32 + cm;
    // (write-through `sc1` stores, which drop the line from the XCD's L2 instead of keeping it, measured SLOWER: FlowNetC 4.281 -> 4.315 ms,
    // PWC-Net 16.52 -> 16.60, FlowNet2 9.52 -> 9.66, RAFT 14.90 -> 15.27 per iteration, one call, gpurun r6_wt; non-temporal (`nt`) stores measured the same:
    // FlowNetC 4.297 / 4.285 -> 4.305, FlowNet2 9.432 -> 9.426, gpurun r6_nt)
    if (e.nostore) {                   // measurement only: the planes are computed and dropped
      asm volatile("" ::"v"(q0), "v"(q1), "v"(q2), "v"(o));
      return;
    }
    *reinterpret_cast<bf16x8*>(o) = q0;
    *reinterpret_cast<bf16x8*>(o + e.out_plane_stride) = q1;
    *reinterpret_cast<bf16x8*>(o + 2 * e.out_plane_stride) = q2;
  }
}

// XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs in launch order (x fastest, then y, then z), each
// XCD with its own 4 MB L2.  Launch index i is remapped to tile (i % 8) * ceil(n/8) + i / 8 (the bijective form for
// n % 8 != 0): every XCD then owns ONE contiguous run of tiles -- all N tiles of an M tile and the M tiles above and below
// it, whose taps read the same activation rows -- instead of every XCD fetching every row from the fabric (measured
// before: 22 GB of L2 misses per iteration for ~3.5 GB of activations, profiles/r2_igemm_traffic.json).
__device__ __forceinline__ void xcd_tile(int swz, int& bx, int& by, int& bz) {
  bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
  if (!swz) return;
  const int nx = gridDim.x, nxy = nx * gridDim.y, n = nxy * gridDim.z;
  const int orig = (bz * gridDim.y + by) * nx + bx;
  const int q = n / 8, r = n % 8, xcd = orig % 8, idx = orig / 8;
  const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  // z (phase x split-K slice) varies FASTEST inside an XCD's run: the phases of a stride-2 data gradient reduce over 1, 2,
  // 2 and 4 taps, so a phase-major list would hand two XCDs only the shortest workgroups and two only the longest (measured:
  // conv4 / conv5 / conv6 backward at half the rate of their stride-1 neighbours); the phases of one tile also read the same
  // activation rows.
  bz = w % gridDim.z;
  const int rem = w / gridDim.z;
  by = rem / nx;
  bx = rem - by * nx;
}

struct Phase {
  int ntaps, oy0, ox0;
  long w_off;                       // bf16 elements from the weight plane's start
  int dyx[UFR_IGEMM_MAX_TAPS];      // (dy & 0xffff) | (dx << 16): one dword per tap, so a wave-uniform tap index becomes an
                                    // s_load (a byte table is read with global_load_sbyte, whose wait drains the LDS-DMA too)
  int run[UFR_IGEMM_MAX_TAPS];      // horizontal runs of taps (same dy, dx one apart, <= 3): shift | position << 2 | length << 4
};                                  // (shift = dx - the run's smallest dx): igemm_pp3_kernel stages a run's pixels once

struct Args {
  const __bf16* x; long x_plane_stride; int in_chunk0, KC;
  int Hi, Wi, in_sy, in_sx;
  const int* in_x0; int in_x0_stride, in_x0_div, in_xw;   // optional validity band of the INPUT columns (else [0, Wi))
  const __bf16* w; long w_plane_stride; int Npad;
  RowGeom g;
  Epilogue e;
  int nphase, splitk, xcd;
  int korder;                        // K tile kt = tap * KC + chunk (0) or chunk * ntaps + tap (1: a pixel's taps back to back)
  int per_k, sk[4], zoff[4];         // K tiles per slice; slices of each phase; first slice (z) of each phase
  float* ws;                         // split-K slabs [sum of sk][M][Npad]
  int* tickets;                      // fused split-K reduction: one arrival counter per (phase, row tile, column tile), zero between launches
  int skip_out;                      // MEASUREMENT ONLY (UFR_IGEMM_DEBUG_SKIP_EPILOGUE=1): the accumulators are dropped -- what a launch costs without its epilogue
  Phase ph[4];
};

// The split-K sum of eight consecutive columns of one row: slabs `src + s * sstride`, s = 0 .. S - 1, added in ASCENDING order onto
// zero (the slabs of up to eight slices are requested before the first is added: a deep split on a tiny grid -- 16 .. 32 slabs of a
// 7 x 16 grid -- was a chain of dependent loads, ~0.5 us each; the order of the additions, and with it every bit of the sum, stays).
// Shared by igemm_reduce_kernel and the fused reduction of igemm_write_out: the two forms are bit-identical by construction.
__device__ __forceinline__ void sum_slabs(const float* src, long sstride, int S, float (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
  int s = 0;
  for (; s + 8 <= S; s += 8) {       // (eight where the split has them: a split of 6 - 8 on RAFT's 48 x 160 grids was two round trips)
    float4 lo[8], hi[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      lo[u] = *reinterpret_cast<const float4*>(src + (s + u) * sstride);
      hi[u] = *reinterpret_cast<const float4*>(src + (s + u) * sstride + 4);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      v[0] += lo[u].x; v[1] += lo[u].y; v[2] += lo[u].z; v[3] += lo[u].w; v[4] += hi[u].x; v[5] += hi[u].y; v[6] += hi[u].z; v[7] += hi[u].w;
    }
  }
  if (S - s >= 5) {                  // 5 .. 7 left: all of them together (the loads past the last slice repeat it, their values are not added)
    float4 lo[7], hi[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int su = min(s + u, S - 1);
      lo[u] = *reinterpret_cast<const float4*>(src + su * sstride);
      hi[u] = *reinterpret_cast<const float4*>(src + su * sstride + 4);
    }
#pragma unroll
    for (int u = 0; u < 7; ++u)
      if (s + u < S) {
        v[0] += lo[u].x; v[1] += lo[u].y; v[2] += lo[u].z; v[3] += lo[u].w; v[4] += hi[u].x; v[5] += hi[u].y; v[6] += hi[u].z; v[7] += hi[u].w;
      }
    s = S;
  }
  for (; s + 4 <= S; s += 4) {
    float4 lo[4], hi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      lo[u] = *reinterpret_cast<const float4*>(src + (s + u) * sstride);
      hi[u] = *reinterpret_cast<const float4*>(src + (s + u) * sstride + 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[0] += lo[u].x; v[1] += lo[u].y; v[2] += lo[u].z; v[3] += lo[u].w; v[4] += hi[u].x; v[5] += hi[u].y; v[6] += hi[u].z; v[7] += hi[u].w;
    }
  }
  for (; s < S; ++s) {
    const float4 lo = *reinterpret_cast<const float4*>(src + s * sstride), hi = *reinterpret_cast<const float4*>(src + s * sstride + 4);
    v[0] += lo.x; v[1] += lo.y; v[2] += lo.z; v[3] += lo.w; v[4] += hi.x; v[5] += hi.y; v[6] += hi.z; v[7] += hi.w;
  }
}

// Accumulators -> memory: split-K slabs, or the fused epilogue through an LDS transpose (a lane owns 8 consecutive
// channels of one pixel: 16-byte plane / fp32 stores; row stride 68 floats is conflict-free both ways).
//
// Split-K WITHOUT a second launch (a.tickets, round 6; the recipe of cdna_hip_programming.md 5 "in-launch split-K reduction"): every
// slice's workgroup stores its slab with plain stores, drains them, and after the workgroup's barrier ONE lane publishes them with an
// agent-scope release and draws a ticket of the tile's counter; the workgroup that draws the LAST ticket of its (phase, tile) acquires
// (one lane, agent scope), and all of its waves add the tile's slabs in ascending slice order through `sum_slabs` -- the reduce kernel's
// arithmetic, bit for bit -- and run the epilogue.  Correct for any placement of a tile's slices over CUs / XCDs; xcd_tile keeps them on one
// XCD (speed only).  The last arriver leaves the counter at zero: the next launch of this descriptor (a graph replay) starts clean.
// WGR = rows of the workgroup's tile (the ping-pong forms: 256 = both groups), bm0 = its first row, `flag` = an int of the EXISTING
// LDS array (a second __shared__ object beside an LDS-DMA staging array can de-pipeline the K loop).
template <int NPL, int MT, int WGR, int BNT>
__device__ __forceinline__ void igemm_write_out(const Args& a, const Phase& ph, f32x4 (&acc)[MT][4], float* lds_f32, int* flag, int z, int phase,
                                                int tile_id, int bm0, int bm, int bn0, int bn, int wrow, int wcol, int lane, int wave) {
  // C/D layout of the 16x16 forms: col = lane & 15, row = (lane >> 4) * 4 + reg
  if (a.skip_out) {                  // measurement only: the upper bound of what hiding the epilogue under the next tile's K loop could return
    float keep = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) keep += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    if (keep == 1.2345678e-30f && a.ws) a.ws[0] = keep;      // (never true in practice: keeps the K loop alive)
    return;
  }
  if (a.splitk > 1) {
    float* slab = a.ws + (long)z * a.g.M * a.Npad;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = bm + wrow + m * 16 + (lane >> 4) * 4 + j;
        if (row < a.g.M) {
#pragma unroll
          for (int n = 0; n < 4; ++n) slab[(long)row * a.Npad + bn + wcol + n * 16 + (lane & 15)] = acc[m][n][j];
        }
      }
    if (!a.tickets) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // every wave: its slab stores have left
    __syncthreads();                                                // (also: every wave is done with the K loop's LDS images)
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");            // publish the workgroup's slab (writes back this XCD's L2)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (keep: ROCm 7.2 can drop the fence's own wait)
      const int S = a.sk[phase];
      const int t = __hip_atomic_fetch_add(a.tickets + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = t == S - 1;
      if (last) {
        __hip_atomic_store(a.tickets + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every slice has arrived: clean for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    // the last arriver: rows [bm0, bm0 + WGR) x columns [bn0, bn0 + BNT) of this phase, a thread = (row, 8 columns)
    const float* base = a.ws + (long)a.zoff[phase] * a.g.M * a.Npad;
    const long sstride = (long)a.g.M * a.Npad;
    const int S = a.sk[phase];
    for (int it = threadIdx.x; it < WGR * (BNT / 8); it += blockDim.x) {
      const int r = it / (BNT / 8), n0 = bn0 + (it - r * (BNT / 8)) * 8, row = bm0 + r;
      if (row >= a.g.M || n0 >= a.e.Nchunks32 * 32) continue;
      float v[8];
      sum_slabs(base + (long)row * a.Npad + n0, sstride, S, v);
      epilogue_store8(a.e, out_pixel(a.g, row, ph.oy0, ph.ox0), n0, v);
    }
    return;
  }
  {
    // transpose each wave's 32 x 64 accumulator slabs through LDS so that a lane owns 8 consecutive channels of one
    // pixel: 16-byte plane / fp32 stores instead of 2-byte ones (row stride 68 floats: conflict-free both ways)
    constexpr int TS = 68;
    float* tw = lds_f32 + wave * (32 * TS);
#pragma unroll
    for (int pass = 0; pass < MT / 2; ++pass) {
      __syncthreads();                                   // the K loop's (or the previous pass's) LDS reads are done
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            tw[(mm * 16 + (lane >> 4) * 4 + j) * TS + n * 16 + (lane & 15)] = acc[pass * 2 + mm][n][j];
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int item = it * 64 + lane, r = item >> 3, g8 = item & 7;
        const int row = bm + wrow + pass * 32 + r;
        if (row < a.g.M) {
          const float4 lo = *reinterpret_cast<const float4*>(tw + r * TS + g8 * 8), hi = *reinterpret_cast<const float4*>(tw + r * TS + g8 * 8 + 4);
          float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          epilogue_store8(a.e, out_pixel(a.g, row, ph.oy0, ph.ox0), bn + wcol + g8 * 8, v);
        }
      }
    }
  }
}

// Staging is LDS-DMA (`global_load_lds_dwordx4`: HBM/L2 -> LDS without passing through registers).
// No staging registers and no ds_write pass: 3 workgroups per CU (<= 168 VGPRs, 3 x 48 KB of LDS), so that while one
// workgroup waits for its tile two others have MFMA work (a register-staged form of round 2 left the matrix pipe idle
// half of the time, profiles/r2_igemm_v1_pmc1.txt: SQ_VALU_MFMA_BUSY 50 %).  The LDS image is the same XOR-swizzled one;
// an LDS-DMA writes lane-linearly (wave base + lane * 16 B), so the swizzle moves to the SOURCE address (lane (row, slot)
// fetches 16-byte piece slot ^ ((row >> 1) & 3) of its row).  Rows outside the frame fetch a zero page.
__device__ __attribute__((aligned(64))) unsigned ufr_zero_page[16];

// Clock probe (tools/measure_clock.py; ufr_igemm_clock_probe): when set, thread 0 of every ping-pong workgroup records the
// shader-clock counter (s_memtime: core cycles) and the constant 100 MHz counter (s_memrealtime) at its first and last
// instruction: cycles / real time = the clock the CU actually ran this kernel at.  NULL (the default) costs one scalar load.
__device__ unsigned long long* ufr_clock_probe_buf = nullptr;
__device__ int ufr_clock_probe_cap = 0;

// one LDS-DMA: 16 bytes per lane from `src` (per lane) to `lds_wave_base + 16 * lane` (the base must be wave-uniform)
__device__ __forceinline__ void glds16(const __bf16* src, __bf16* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(src, lds_wave_base, 16, 0, 0);
}

// BM_ x BN_ tiles: 128 x 128 (waves 2 x 2 of 64 x 64), 128 x 64 (waves 4 x 1 of 32 x 64: layers with <= 64 outputs), and
// 64 x 128 (waves 2 x 2 of 32 x 64; 36 KB of LDS: FOUR workgroups per CU, twice the workgroups for the mid-size layers).
//
// PIPE_ (variant 5, the engine's default for 128 x 128 launches): a K step moves ALL of its fragments into registers first
// (24 ds_read_b128 per wave, 96 VGPRs), so the DMA of the NEXT K tile can be issued before the 96 MFMAs instead of after
// them and the L2 round trip runs under the matrix work of the same workgroup -- the plain form relies on two other
// workgroups to cover it.  A second activation image (72 KB of LDS) gives the operand that misses L2 a whole step to
// arrive.  ~210 VGPRs: two workgroups per CU.  Measured on one box (profiles/r2_igemm_layers_v5_*.txt): 8-17 % per layer.
// BUF_: activation rows through a raw buffer resource (see igemm_pp_kernel)
// NP_ = products per float32 product: 6 (three planes per operand, float32-accurate), 3 (a0b0 + a0b1 + a1b0: two planes) or 1 (a0b0: one
// bf16 plane per operand -- RAFT's opt-in reduced precision); the planes a form does not multiply are not staged either.
template <int BM_, int BN_, bool PIPE_ = false, bool BUF_ = false, int NP_ = 6>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PIPE_ ? 2 : (BM_ == 64 ? 4 : 3), PIPE_ ? 2 : (BM_ == 64 ? 4 : 3)))) void igemm_glds_kernel(const Args a) {
  static_assert(!(PIPE_ && BUF_), "the buffer-resource form exists for the single-stage kernel");
  static_assert(NP_ == 6 || NP_ == 3 || NP_ == 1, "six, three or one product");
  constexpr int NPL = NP_ == 6 ? 3 : (NP_ == 3 ? 2 : 1), FIRST = 6 - NP_;
  constexpr int MT = (BN_ == 128 && BM_ == 128) ? 4 : 2;
  constexpr int BPT = BN_ / 64, APT = BM_ / 64;
  constexpr int STAGE = NPL * (BM_ + BN_) * BK;       // elements of one (A, B) stage; PIPE_ appends a second A image
  constexpr int EPI = 4 * 32 * 68 * 2;                // elements the epilogue's transpose needs (one- / two-plane stages are smaller)
  __shared__ __attribute__((aligned(16))) __bf16 lds_static[PIPE_ ? 8 : (STAGE > EPI ? STAGE : EPI)];
  extern __shared__ __attribute__((aligned(16))) __bf16 lds_dynamic[];
  __bf16* lds_all = PIPE_ ? lds_dynamic : lds_static;
  __bf16 (*ldsA)[BM_ * BK] = reinterpret_cast<__bf16 (*)[BM_ * BK]>(lds_all);
  __bf16 (*ldsB)[BN_ * BK] = reinterpret_cast<__bf16 (*)[BN_ * BK]>(lds_all + NPL * BM_ * BK);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wrow = BN_ == 128 ? (wave >> 1) * (BM_ / 2) : wave * 32, wcol = BN_ == 128 ? (wave & 1) * 64 : 0;
  int tx, ty, z;
  xcd_tile(a.xcd, tx, ty, z);
  const int bm = ty * BM_, bn = tx * BN_;
  int phase = 0;                                     // z -> (phase, slice): phases with fewer taps have fewer slices
#pragma unroll
  for (int p = 1; p < 4; ++p)
    if (p < a.nphase && z >= a.zoff[p]) phase = p;
  const int ks = z - a.zoff[phase];
  const Phase& ph = a.ph[phase];
  const int KC = a.KC, KT = ph.ntaps * KC;
  const int kt0 = ks * a.per_k, kt1 = min(KT, kt0 + a.per_k);
  const long Min = (long)a.g.B * a.Hi * a.Wi;
  const long cstride = Min * 32;

  const int srow0 = tid >> 2, sch = tid & 3;
  const int csw = sch ^ ((srow0 >> 1) & 3);             // the piece this lane fetches (64 more rows keep (row >> 1) & 3)
  int yb[APT], xb[APT], xlo[APT], xhi[APT];
  long ibase[APT];
#pragma unroll
  for (int i = 0; i < APT; ++i) {
    const int pm = bm + srow0 + 64 * i;
    const int hw = a.g.Hr * a.g.Wr;
    const int b = pm / hw, r = pm - b * hw, yr = r / a.g.Wr, xr = r - yr * a.g.Wr;
    const bool live = pm < a.g.M;
    const int bb = live ? b : 0;
    const int xg = xr + (a.g.row_x0 ? a.g.row_x0[bb * a.g.row_x0_stride] / a.g.row_x0_div : 0);
    yb[i] = live ? yr * a.in_sy : -(1 << 20);
    xb[i] = xg * a.in_sx;
    ibase[i] = (long)bb * a.Hi * a.Wi;
    xlo[i] = a.in_x0 ? a.in_x0[bb * a.in_x0_stride] / a.in_x0_div : 0;
    xhi[i] = a.in_x0 ? min(a.Wi, xlo[i] + a.in_xw) : a.Wi;
    xlo[i] = max(xlo[i], 0);
  }
  const __bf16* gx = a.x + (long)a.in_chunk0 * cstride + csw * 8;
  const __bf16* zero = reinterpret_cast<const __bf16*>(ufr_zero_page);
  int tap = a.korder ? kt0 % ph.ntaps : kt0 / KC, kc = a.korder ? kt0 / ph.ntaps : kt0 - tap * KC;
  const __bf16* wp = a.w + ph.w_off + ((long)kt0 * a.Npad + bn + srow0) * BK + csw * 8;
  const long wstep = (long)a.Npad * BK;
  const __bf16* xk = gx + (long)kc * cstride;
  bool ok[APT];
  long aoff[APT];
  unsigned voff[APT], rowbase[APT];
#pragma unroll
  for (int i = 0; i < APT; ++i) rowbase[i] = (unsigned)(ibase[i] * 64 + csw * 16);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x), 0, (int)min(6L * a.x_plane_stride, 0x7fffffffL),
                                                                   0x00020000);
  long xkoff = (long)(a.in_chunk0 + kc) * cstride;
  auto set_tap = [&](int t) {
    const int dyx = ph.dyx[t], dyo = (int)(short)(dyx & 0xffff), dxo = dyx >> 16;
#pragma unroll
    for (int i = 0; i < APT; ++i) {
      const int yi = yb[i] + dyo, xi = xb[i] + dxo;
      ok[i] = (unsigned)yi < (unsigned)a.Hi && xi >= xlo[i] && xi < xhi[i];
      if (BUF_) voff[i] = ok[i] ? rowbase[i] + (unsigned)(yi * a.Wi + xi) * 64u : 0x80000000u;
      else aoff[i] = ok[i] ? (ibase[i] + (long)yi * a.Wi + xi) * 32 : 0;
    }
  };
  // wave-uniform LDS destinations: the wave's 16 rows of each image (lane l lands at base + 16 l bytes)
  auto stage_A = [&](int img) {      // the activation rows of the next K tile -> LDS image at element `img`, then advance (tap, chunk)
#pragma unroll
    for (int i = 0; i < APT; ++i)
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        if constexpr (BUF_) {                 // (single-stage form only: the static array)
          typedef __attribute__((address_space(3))) void* lds_void;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void)(lds_static + img + p * (BM_ * BK) + (64 * i + wave * 16) * BK), 16, (int)voff[i],
                                                   (unsigned)((xkoff + p * a.x_plane_stride) * 2), 0, 0);
        } else {
          const __bf16* src = ok[i] ? xk + aoff[i] + p * a.x_plane_stride : zero;
          glds16(src, lds_all + img + p * (BM_ * BK) + (64 * i + wave * 16) * BK);
        }
      }
    if (a.korder) {                  // the taps of one channel chunk back to back: their pixels overlap, so they hit in L2
      if (++tap == ph.ntaps) {
        tap = 0;
        xk += cstride;
        xkoff += cstride;
      }
      set_tap(tap);
    } else {
      xk += cstride;
      xkoff += cstride;
      if (++kc == KC) {
        kc = 0;
        xk = gx;
        xkoff = (long)a.in_chunk0 * cstride;
        if (++tap < ph.ntaps) set_tap(tap);
      }
    }
  };
  auto stage_B = [&]() {             // the weight rows of the next K tile
#pragma unroll
    for (int i = 0; i < BPT; ++i)
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        glds16(wp + p * a.w_plane_stride + (long)(64 * i) * BK, &ldsB[p][(64 * i + wave * 16) * BK]);
    wp += wstep;
  };
  auto stage_tile = [&]() {
    stage_A(0);
    stage_B();
  };

  f32x4 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15;
  const int foff = frow * BK + ((((lane >> 4)) ^ ((frow >> 1) & 3)) << 3);

  if (kt0 < kt1) set_tap(tap);
  if constexpr (PIPE_) {
    // two activation images (the second behind the stage), one weight image.  Step kt: [all DMA landed, barrier] -> DMA of
    // A(kt + 1) into the other image (its last readers finished before the previous step's second barrier) -> 24 fragment
    // reads -> [barrier] -> DMA of B(kt + 1) over B(kt) -> 96 MFMAs.  The activation rows -- the operand that misses L2 --
    // have the whole step to arrive, the weight rows (L2 hits) the MFMA phase.
    if (kt0 < kt1) {
      stage_A(0);
      stage_B();
    }
    for (int kt = kt0; kt < kt1; ++kt) {
      const int cur = (kt - kt0) & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(2); // the read phase is the exposed one: ahead of the other workgroup's MFMA stream
      const __bf16* sA = lds_all + (cur ? STAGE : 0);
      bf16x8 fa[NPL][MT], fb[4][NPL];
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
#pragma unroll
        for (int m = 0; m < MT; ++m) fa[p][m] = *reinterpret_cast<const bf16x8*>(sA + p * (BM_ * BK) + (wrow + m * 16) * BK + foff);
#pragma unroll
        for (int n = 0; n < 4; ++n) fb[n][p] = *reinterpret_cast<const bf16x8*>(&ldsB[p][(wcol + n * 16) * BK + foff]);
      }
      if (kt + 1 < kt1) stage_A(cur ? 0 : STAGE);                  // (its address arithmetic under the reads' latency)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave holds its fragments: the weight image is free
      if (kt + 1 < kt1) stage_B();
      __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int t = FIRST; t < 6; ++t)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[PROD_A[t]][m], fb[n][PROD_B[t]], acc[m][n], 0, 0, 0);
    }
  } else
  for (int kt = kt0; kt < kt1; ++kt) {
    __syncthreads();                 // every wave has read the previous tile's fragments
    stage_tile();
    __syncthreads();                 // hipcc waits vmcnt(0) here: the DMA writes have landed for every wave
    bf16x8 fa[NPL][MT];
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int m = 0; m < MT; ++m)
        fa[p][m] = *reinterpret_cast<const bf16x8*>(&ldsA[p][(wrow + m * 16) * BK + foff]);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      bf16x8 fb[NPL];
#pragma unroll
      for (int p = 0; p < NPL; ++p) fb[p] = *reinterpret_cast<const bf16x8*>(&ldsB[p][(wcol + n * 16) * BK + foff]);
#pragma unroll
      for (int t = FIRST; t < 6; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[PROD_A[t]][m], fb[PROD_B[t]], acc[m][n], 0, 0, 0);
    }
  }
  static_assert(PIPE_ ? 4 * 32 * 68 * 4 <= STAGE * 2 : true, "epilogue staging does not fit");
  igemm_write_out<NPL, MT, BM_, BN_>(a, ph, acc, reinterpret_cast<float*>(lds_all), reinterpret_cast<int*>(lds_all), z, phase,
                                     (phase * gridDim.y + ty) * gridDim.x + tx, bm, bm, bn, bn, wrow, wcol, lane, wave);
}
constexpr int PIPE_LDS_BYTES = (3 * (128 + 128) * BK + 3 * 128 * BK) * 2;     // one stage + the second activation image

// ---- ping-pong form (variant 6): 256 x 128 tiles, EIGHT waves = two groups of four, one workgroup per CU -----------------
// Group g owns rows [128 g, 128 g + 128) of the tile (each wave 64 x 64, as in the pipelined kernel) and runs the same two
// half-steps -- READ (24 fragment reads into registers, DMA of its next activation image; group 0 also the next weight
// image) and MFMA (96 MFMAs) -- but half a step behind the other group, so that on every SIMD (which holds one wave of each
// group) one wave multiplies while the other reads: what two independent pipelined workgroups per CU do when their phases
// happen to alternate, here by construction.  One s_barrier of the whole workgroup separates the half-steps.  LDS: two
// activation images per group + two weight images = 6 x 24 KB = 144 KB; every DMA has a whole MFMA half-step to land.
//   barrier #   1        2        3        4
//   group 0   | READ 0 | MFMA 0 | READ 1 | MFMA 1 | ...
//   group 1   | (idle) | READ 0 | MFMA 0 | READ 1 | ...
// weight image (k + 1) & 1 is refilled by group 0 in READ k: its last readers were group 0's READ k - 1 and group 1's READ
// k - 1, one and two half-steps earlier, both closed by a barrier.
constexpr int PP_IMG = 3 * 128 * BK;                                      // elements of one 128-row image (3 planes): 24 KB
constexpr int pp_lds_bytes(int bn) { return (4 * PP_IMG + 2 * 3 * bn * BK) * 2; }

// A group's waves: 2 x 2 of 64 x 64 (a 64-column form with 12 KB weight images measured slower than the single-stage 128 x 64
// tile and was removed).
// BUF_: the activation rows come through a raw buffer resource (`buffer_load_dwordx4 ... lds`): a row outside the frame /
// band gets an out-of-range offset and the hardware writes zeros -- no zero page, no 64-bit pointer select per plane, the
// plane / chunk displacement in the scalar offset: ~20 instead of ~70 vector instructions per K step in the READ half-step,
// which shares the SIMD's issue port with the other group's MFMA stream.
template <int BN_, bool BUF_ = false, int NP_ = 6>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void igemm_pp_kernel(const Args a) {
  static_assert(NP_ == 6 || NP_ == 3 || NP_ == 1, "six, three or one product");
  constexpr int NPL = NP_ == 6 ? 3 : (NP_ == 3 ? 2 : 1), FIRST = 6 - NP_;      // (the images keep their three-plane size: one workgroup per CU)
  constexpr int MT = BN_ == 128 ? 4 : 2, BPT = BN_ / 64;
  constexpr int PP_IMG_B = 3 * BN_ * BK;
  extern __shared__ __attribute__((aligned(16))) __bf16 lds_pp[];
  unsigned long long* const probe = ufr_clock_probe_buf;
  const unsigned long long probe_c0 = probe ? __builtin_amdgcn_s_memtime() : 0, probe_r0 = probe ? __builtin_amdgcn_s_memrealtime() : 0;
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);             // wave-uniform: LDS destinations on the scalar unit
  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);      // wave-uniform: scalar control flow
  const int wrow = BN_ == 128 ? (wave >> 1) * 64 : wave * 32, wcol = BN_ == 128 ? (wave & 1) * 64 : 0;
  int tx, ty, z;
  xcd_tile(a.xcd, tx, ty, z);
  const int bm = ty * 256 + grp * 128, bn = tx * BN_;
  int phase = 0;
#pragma unroll
  for (int p = 1; p < 4; ++p)
    if (p < a.nphase && z >= a.zoff[p]) phase = p;
  const int ks = z - a.zoff[phase];
  const Phase& ph = a.ph[phase];
  const int KC = a.KC, KT = ph.ntaps * KC;
  const int kt0 = ks * a.per_k, kt1 = min(KT, kt0 + a.per_k);
  const int nk = max(kt1 - kt0, 0);
  const long Min = (long)a.g.B * a.Hi * a.Wi;
  const long cstride = Min * 32;

  const int srow0 = tid >> 2, sch = tid & 3;
  const int csw = sch ^ ((srow0 >> 1) & 3);
  int yb[2], xb[2], xlo[2], xhi[2];
  long ibase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pm = bm + srow0 + 64 * i;
    const int hw = a.g.Hr * a.g.Wr;
    const int b = pm / hw, r = pm - b * hw, yr = r / a.g.Wr, xr = r - yr * a.g.Wr;
    const bool live = pm < a.g.M;
    const int bb = live ? b : 0;
    const int xg = xr + (a.g.row_x0 ? a.g.row_x0[bb * a.g.row_x0_stride] / a.g.row_x0_div : 0);
    yb[i] = live ? yr * a.in_sy : -(1 << 20);
    xb[i] = xg * a.in_sx;
    ibase[i] = (long)bb * a.Hi * a.Wi;
    xlo[i] = a.in_x0 ? a.in_x0[bb * a.in_x0_stride] / a.in_x0_div : 0;
    xhi[i] = a.in_x0 ? min(a.Wi, xlo[i] + a.in_xw) : a.Wi;
    xlo[i] = max(xlo[i], 0);
  }
  const __bf16* gx = a.x + (long)a.in_chunk0 * cstride + csw * 8;
  const __bf16* zero = reinterpret_cast<const __bf16*>(ufr_zero_page);
  int tap = a.korder ? kt0 % ph.ntaps : kt0 / KC, kc = a.korder ? kt0 / ph.ntaps : kt0 - tap * KC;
  const __bf16* wp = a.w + ph.w_off + ((long)kt0 * a.Npad + bn + srow0) * BK + csw * 8;
  const long wstep = (long)a.Npad * BK;
  const __bf16* xk = gx + (long)kc * cstride;
  bool ok[2];
  long aoff[2];
  unsigned voff[2];                                                        // BUF_: byte offset of the row's piece, or out of range
  const unsigned rowbase[2] = {(unsigned)(ibase[0] * 64 + csw * 16), (unsigned)(ibase[1] * 64 + csw * 16)};
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x), 0, (int)min(6L * a.x_plane_stride, 0x7fffffffL),
                                                                   0x00020000);
  long xkoff = (long)(a.in_chunk0 + kc) * cstride;                         // BUF_: elements from plane 0 to the current chunk
  auto set_tap = [&](int t) {
    const int dyx = ph.dyx[t], dyo = (int)(short)(dyx & 0xffff), dxo = dyx >> 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int yi = yb[i] + dyo, xi = xb[i] + dxo;
      ok[i] = (unsigned)yi < (unsigned)a.Hi && xi >= xlo[i] && xi < xhi[i];
      if (BUF_) voff[i] = ok[i] ? rowbase[i] + (unsigned)(yi * a.Wi + xi) * 64u : 0x80000000u;
      else aoff[i] = ok[i] ? (ibase[i] + (long)yi * a.Wi + xi) * 32 : 0;
    }
  };
  const int imgA0 = grp * 2 * PP_IMG, imgB0 = 4 * PP_IMG;                 // element offsets: A[grp][0..1], B[0..1]
  auto stage_A = [&](int img) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        if (BUF_) {
          typedef __attribute__((address_space(3))) void* lds_void;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void)(lds_pp + img + p * (128 * BK) + (64 * i + wave * 16) * BK), 16, (int)voff[i],
                                                   (unsigned)((xkoff + p * a.x_plane_stride) * 2), 0, 0);
        } else {
          const __bf16* src = ok[i] ? xk + aoff[i] + p * a.x_plane_stride : zero;
          glds16(src, lds_pp + img + p * (128 * BK) + (64 * i + wave * 16) * BK);
        }
      }
    if (a.korder) {
      if (++tap == ph.ntaps) {
        tap = 0;
        xk += cstride;
        xkoff += cstride;
      }
      set_tap(tap);
    } else {
      xk += cstride;
      xkoff += cstride;
      if (++kc == KC) {
        kc = 0;
        xk = gx;
        xkoff = (long)a.in_chunk0 * cstride;
        if (++tap < ph.ntaps) set_tap(tap);
      }
    }
  };
  auto stage_B = [&](int img) {
#pragma unroll
    for (int i = 0; i < BPT; ++i)
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        glds16(wp + p * a.w_plane_stride + (long)(64 * i) * BK, lds_pp + img + p * (BN_ * BK) + (64 * i + wave * 16) * BK);
    wp += wstep;
  };

  f32x4 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15;
  const int foff = frow * BK + ((((lane >> 4)) ^ ((frow >> 1) & 3)) << 3);

  if (nk > 0) {
    set_tap(tap);
    stage_A(imgA0);
    if (grp == 0) stage_B(imgB0);
  }
  if (grp == 1) {                    // half a step behind group 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const unsigned long long probe_c2 = probe ? __builtin_amdgcn_s_memtime() : 0;
  for (int i = 0; i < nk; ++i) {
    const int cur = i & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // my DMAs of the previous READ have landed ...
    __builtin_amdgcn_s_barrier();                                         // ... and everybody else's
    // ---- READ half-step
    __builtin_amdgcn_s_setprio(2);
    const __bf16* sA = lds_pp + imgA0 + cur * PP_IMG;
    const __bf16* sB = lds_pp + imgB0 + cur * PP_IMG_B;
    bf16x8 fa[NPL][MT], fb[4][NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
#pragma unroll
      for (int m = 0; m < MT; ++m) fa[p][m] = *reinterpret_cast<const bf16x8*>(sA + p * (128 * BK) + (wrow + m * 16) * BK + foff);
#pragma unroll
      for (int n = 0; n < 4; ++n) fb[n][p] = *reinterpret_cast<const bf16x8*>(sB + p * (BN_ * BK) + (wcol + n * 16) * BK + foff);
    }
    if (i + 1 < nk) {
      stage_A(imgA0 + (cur ^ 1) * PP_IMG);
      if (grp == 0) stage_B(imgB0 + (cur ^ 1) * PP_IMG_B);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---- MFMA half-step (the other group reads)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int t = FIRST; t < 6; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[PROD_A[t]][m], fb[n][PROD_B[t]], acc[m][n], 0, 0, 0);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();                             // group 1's extra barrier at the start
  const unsigned long long probe_c3 = probe ? __builtin_amdgcn_s_memtime() : 0;
  static_assert(4 * 32 * 68 * 4 <= 2 * PP_IMG * 2, "epilogue staging does not fit a group's activation images");
  igemm_write_out<NPL, MT, 256, BN_>(a, ph, acc, reinterpret_cast<float*>(lds_pp + imgA0), reinterpret_cast<int*>(lds_pp), z, phase,
                                     (phase * gridDim.y + ty) * gridDim.x + tx, ty * 256, bm, bn, bn, wrow, wcol, lane, wave);
  if (probe && threadIdx.x == 0) {
    const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (wg < ufr_clock_probe_cap) {                                       // 8 words per workgroup: cycles at entry / exit, 100 MHz
      unsigned long long* o = probe + 8L * wg;                            // ticks at entry / exit, cycles at the K loop's two ends,
      o[0] = probe_c0; o[1] = __builtin_amdgcn_s_memtime(); o[2] = probe_r0; o[3] = __builtin_amdgcn_s_memrealtime();   // HW_ID, K steps
      o[4] = probe_c2; o[5] = probe_c3; o[6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); o[7] = (unsigned long long)nk;
    }
  }
}

// ---- ping-pong with horizontal tap reuse (variant 7) ----------------------------------------------------------------------
// The taps of one run (same dy, dx one apart: the three columns of a 3x3 row, the two of a deconvolution phase) read the SAME
// activation pixels one cell apart, so a group stages a run's pixels ONCE per channel chunk -- image row j = cell (bm + j)'s
// pixel at the run's smallest dx -- and tap `shift` reads row j + shift.  That is exact while cell j + shift lies in the
// same row of the row grid; the last one or two cells of a grid row need the pixels one and two PAST the row's end (zeros
// at a frame edge, real pixels beside a column band), which are staged into fix-up rows behind the image:
//   rows 0..127 the tile's cells | 128, 129 the two cells after the tile | 130 + 2 e + w: pixel w + 1 past the end of the
//   e-th grid row the tile touches (e < 7: the host sends launches with Wr < 22 to igemm_pp_kernel).
// A lane's fragment rows are fixed, so the row it reads for (m, shift) is a per-lane constant: 12 LDS offsets.
// L2 -> LDS bytes per K step: 24 KB of weights + 2 x 27 KB of activations per RUN instead of 72 KB (3x3: 42 KB).
constexpr int PP3_ROWS = 144;
constexpr int PP3_IMG = 3 * PP3_ROWS * BK;                                // elements of one activation image: 27 KB
constexpr int pp3_lds_bytes(int bn) { return (4 * PP3_IMG + 2 * 3 * bn * BK) * 2; }   // 159,744 B at 128 columns

// BN_ = 128: a group's waves 2 x 2 of 64 x 64.  BN_ = 64 (PWC-Net's decoder: 64 and 32 outputs behind 500 input channels, where
// the activation rows are 80 % of a K step's L2 -> LDS bytes and the plain forms are bound by that delivery): 4 x 1 of 32 x 64.
template <int BN_>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void igemm_pp3_kernel(const Args a) {
  constexpr int NPL = 3, MT = BN_ == 128 ? 4 : 2, BPT = BN_ / 64, PP_IMG_B = 3 * BN_ * BK;
  extern __shared__ __attribute__((aligned(16))) __bf16 lds_pp[];
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  const int wrow = BN_ == 128 ? (wave >> 1) * 64 : wave * 32, wcol = BN_ == 128 ? (wave & 1) * 64 : 0;
  int tx, ty, z;
  xcd_tile(a.xcd, tx, ty, z);
  const int bm = ty * 256 + grp * 128, bn = tx * BN_;
  int phase = 0;
#pragma unroll
  for (int p = 1; p < 4; ++p)
    if (p < a.nphase && z >= a.zoff[p]) phase = p;
  const int ks = z - a.zoff[phase];
  const Phase& ph = a.ph[phase];
  const int ntaps = ph.ntaps, KT = ntaps * a.KC;
  const int kt0 = ks * a.per_k, kt1 = min(KT, kt0 + a.per_k);
  const int nk = max(kt1 - kt0, 0);
  const long Min = (long)a.g.B * a.Hi * a.Wi;
  const long cstride = Min * 32;
  const int Wr = a.g.Wr, hw = a.g.Hr * Wr;

  // ---- staging geometry: image rows srow0, srow0 + 64 (cells of the tile) and, on wave 0, extra row 128 + (lane >> 2)
  const int srow0 = tid >> 2, sch = tid & 3;
  const int csw = sch ^ ((srow0 >> 1) & 3);
  const int xrow = lane >> 2, cswx = sch ^ ((xrow >> 1) & 3);             // (128 + xrow) >> 1 & 3 == xrow >> 1 & 3
  int yb[3], xb[3], xlo[3], xhi[3];
  long ibase[3];
  auto cell_geometry = [&](int i, long cell, int xr_override) {            // cell = flat index into [B, Hr, Wr]
    const bool live = cell >= 0 && cell < a.g.M;
    const int c = live ? (int)cell : 0;
    const int b = c / hw, r = c - b * hw, yr = r / Wr;
    const int xr = xr_override >= 0 ? xr_override : r - yr * Wr;
    const int xg = xr + (a.g.row_x0 ? a.g.row_x0[b * a.g.row_x0_stride] / a.g.row_x0_div : 0);
    yb[i] = live ? yr * a.in_sy : -(1 << 20);
    xb[i] = xg * a.in_sx;
    ibase[i] = (long)b * a.Hi * a.Wi;
    xlo[i] = a.in_x0 ? a.in_x0[b * a.in_x0_stride] / a.in_x0_div : 0;
    xhi[i] = a.in_x0 ? min(a.Wi, xlo[i] + a.in_xw) : a.Wi;
    xlo[i] = max(xlo[i], 0);
  };
  cell_geometry(0, (long)bm + srow0, -1);
  cell_geometry(1, (long)bm + srow0 + 64, -1);
  const int grow0 = bm / Wr;                                              // first grid row (flat over B * Hr) the tile touches
  if (xrow < 2) {
    cell_geometry(2, (long)bm + 128 + xrow, -1);
  } else {                                                                // pixel (xrow & 1) + 1 past the end of grid row grow0 + e
    const int e = (xrow - 2) >> 1, w = (xrow - 2) & 1;
    const long rowcell = (long)(grow0 + e) * Wr;                          // that grid row's first cell
    cell_geometry(2, rowcell < a.g.M ? rowcell : -1, Wr + w);
  }
  const __bf16* gx = a.x + (long)a.in_chunk0 * cstride + csw * 8;
  const __bf16* gxx = a.x + (long)a.in_chunk0 * cstride + cswx * 8;
  const __bf16* zero = reinterpret_cast<const __bf16*>(ufr_zero_page);
  const __bf16* wp = a.w + ph.w_off + ((long)kt0 * a.Npad + bn + srow0) * BK + csw * 8;
  const long wstep = (long)a.Npad * BK;

  const int imgA0 = grp * 2 * PP3_IMG, imgB0 = 4 * PP3_IMG;
  // one run's pixels of channel chunk kc -> activation image at element `img`
  auto stage_A = [&](int img, int kc, int tap) {
    const int dyx = ph.dyx[tap], dyo = (int)(short)(dyx & 0xffff), dxo = (dyx >> 16) - (ph.run[tap] & 3) * a.in_sx;   // the run's first pixel
    const long coff = (long)kc * cstride;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int yi = yb[i] + dyo, xi = xb[i] + dxo;
      const bool ok = (unsigned)yi < (unsigned)a.Hi && xi >= xlo[i] && xi < xhi[i];
      const long off = ok ? (ibase[i] + (long)yi * a.Wi + xi) * 32 + coff : 0;
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        glds16(ok ? gx + off + p * a.x_plane_stride : zero, lds_pp + img + p * (PP3_ROWS * BK) + (64 * i + wave * 16) * BK);
    }
    if (wave == 0) {                                                      // the 16 extra rows: lane -> (row 128 + lane / 4, piece)
      const int yi = yb[2] + dyo, xi = xb[2] + dxo;
      const bool ok = (unsigned)yi < (unsigned)a.Hi && xi >= xlo[2] && xi < xhi[2];
      const long off = ok ? (ibase[2] + (long)yi * a.Wi + xi) * 32 + coff : 0;
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        glds16(ok ? gxx + off + p * a.x_plane_stride : zero, lds_pp + img + p * (PP3_ROWS * BK) + 128 * BK);
    }
  };
  auto stage_B = [&](int img) {
#pragma unroll
    for (int i = 0; i < BPT; ++i)
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        glds16(wp + p * a.w_plane_stride + (long)(64 * i) * BK, lds_pp + img + p * (BN_ * BK) + (64 * i + wave * 16) * BK);
    wp += wstep;
  };

  // ---- fragment rows: lane reads image row of cell (wrow + 16 m + frow) shifted by 0, 1, 2 cells
  const int frow = lane & 15;
  int foffs[MT][3];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int i = wrow + m * 16 + frow;
    const int cell = bm + i, gr = cell / Wr, xr = cell - gr * Wr, e = min(gr - grow0, 6);
#pragma unroll
    for (int sft = 0; sft < 3; ++sft) {
      const int t = xr + sft;
      const int r = t < Wr ? i + sft : 130 + 2 * e + min(t - Wr, 1);
      foffs[m][sft] = r * BK + (((lane >> 4) ^ ((r >> 1) & 3)) << 3);
    }
  }
  const int fboff = frow * BK + (((lane >> 4) ^ ((frow >> 1) & 3)) << 3);

  f32x4 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- K steps kt = chunk * ntaps + tap (chunk-major: the host requires k_order = 1)
  int kc = kt0 / ntaps, tap = kt0 - kc * ntaps;
  int ebuf = 0;                                                           // activation image of the current run
  if (nk > 0) {
    stage_A(imgA0, kc, tap);
    if (grp == 0) stage_B(imgB0);
  }
  if (grp == 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  bool first_of_run = true;                                               // this step is the first of its run inside the slice
  for (int i = 0; i < nk; ++i) {
    const int cur = i & 1;
    const int ri = ph.run[tap], sft = ri & 3, pos = (ri >> 2) & 3, len = ri >> 4;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- READ half-step
    __builtin_amdgcn_s_setprio(2);
    const __bf16* sA = lds_pp + imgA0 + ebuf * PP3_IMG;
    const __bf16* sB = lds_pp + imgB0 + cur * PP_IMG_B;
    bf16x8 fa[NPL][MT], fb[4][NPL];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int fo = sft == 0 ? foffs[m][0] : (sft == 1 ? foffs[m][1] : foffs[m][2]);
#pragma unroll
      for (int p = 0; p < NPL; ++p) fa[p][m] = *reinterpret_cast<const bf16x8*>(sA + p * (PP3_ROWS * BK) + fo);
    }
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int n = 0; n < 4; ++n) fb[n][p] = *reinterpret_cast<const bf16x8*>(sB + p * (BN_ * BK) + (wcol + n * 16) * BK + fboff);
    // the next run's image: issued in the first step of this run (its other image was last read in the previous run)
    int ntap = tap - pos + len, nkc = kc;
    if (ntap >= ntaps) { ntap = 0; ++nkc; }
    const int steps_left_in_run = len - 1 - pos;                           // steps of this run after this one
    if (first_of_run && i + 1 + steps_left_in_run < nk) stage_A(imgA0 + (ebuf ^ 1) * PP3_IMG, nkc, ntap);
    if (grp == 0 && i + 1 < nk) stage_B(imgB0 + (cur ^ 1) * PP_IMG_B);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---- MFMA half-step
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[PROD_A[t]][m], fb[n][PROD_B[t]], acc[m][n], 0, 0, 0);
    // advance (kc, tap); a new run flips the image
    first_of_run = pos == len - 1;
    if (first_of_run) ebuf ^= 1;
    if (++tap == ntaps) { tap = 0; ++kc; }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  static_assert(4 * 32 * 68 * 4 <= 2 * PP3_IMG * 2, "epilogue staging does not fit a group's activation images");
  igemm_write_out<NPL, MT, 256, BN_>(a, ph, acc, reinterpret_cast<float*>(lds_pp + imgA0), reinterpret_cast<int*>(lds_pp), z, phase,
                                     (phase * gridDim.y + ty) * gridDim.x + tx, ty * 256, bm, bn, bn, wrow, wcol, lane, wave);
}

// ---- direct 3 x 3 form (variant 8, round 5): stride-1 3 x 3 convolutions with at most 64 output columns -- the full- and
// half-resolution layers of FlowNetFusion / FlowNetSD (6 -> 64, 11 -> 64, 82 -> 16, 64 -> 6 ... channels at 448 x 1024), the 64-channel
// stage of RAFT's encoders, the 64- / 32-output convolutions of PWC-Net's decoders.  On the tile forms above such a launch pads its
// columns to 64 and streams every activation row once per TAP from L2 (nine times per chunk); it is bound by that delivery and by its own
// latency, not by the matrix pipe (0.12 - 0.45 of the ceiling, profiles/r5_engine_launch_sweep_cold_*.jsonl).  Here a workgroup owns a
// 4 x 32 tile of output pixels (8 m-tiles of 16), stages the tile's 6 x 34 halo of ONE input chunk once (three planes, LDS-DMA, the same
// XOR swizzle keyed by the halo pixel) and takes all nine taps from LDS; the WEIGHTS never touch LDS: a wave owns ONE column tile of 16
// and keeps its nine taps x three planes of the chunk in registers (27 fragments, straight from L2), the 4 / NT waves that share a
// column tile split the m-tiles.  LDS = the halo (40 KB): two to three workgroups per CU cover each other's waits.  Epilogue =
// epilogue_store8 through an LDS transpose, as everywhere.  (A first form that staged the weights in LDS three taps at a time lost to
// the tile forms at 64 columns -- its waits were exposed at one workgroup per CU -- gpurun r5_call17 - r5_call19.)
template <int NT>
__global__ __launch_bounds__(256) void igemm_d33_kernel(const Args a) {
  constexpr int TH = 4, TW = 32, HW_ = TW + 2, HPIX = (TH + 2) * HW_, HBLK = (HPIX + 15) / 16, HROWS = HBLK * 16;
  constexpr int MTW = 2 * NT;                                       // m-tiles per wave: 8 per tile, dealt to the 4 / NT waves of a column tile
  extern __shared__ __attribute__((aligned(16))) __bf16 lds_d33[];
  __bf16* const sH = lds_d33;                                       // [3 planes][HROWS][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = wave % NT, pg = wave / NT;                         // this wave's column tile and its share of the m-tiles
  const int H = a.Hi, W = a.Wi;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  int t;
  {                                                                   // XCD k owns one contiguous run of tiles (xcd_tile's 1-D form)
    const int n = gridDim.x, orig = blockIdx.x, q = n / 8, r = n % 8, xcd = orig % 8, idx = orig / 8;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int b = t / (tiles_x * tiles_y), rem = t - b * tiles_x * tiles_y, y0 = (rem / tiles_x) * TH, x0 = (rem % tiles_x) * TW;
  const Phase& ph = a.ph[0];
  const long Min = (long)a.g.B * H * W;
  const __bf16* zero = reinterpret_cast<const __bf16*>(ufr_zero_page);

  f32x4 acc[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int pi = lane & 15, kgrp = lane >> 4;
  for (int kc = 0; kc < a.KC; ++kc) {
    __syncthreads();                                                  // the previous chunk's fragment reads are done
    for (int blk = wave; blk < 3 * HBLK; blk += 4) {                  // halo of chunk kc: 16 pixels x 4 pieces per DMA
      const int p = blk / HBLK, hb = blk - p * HBLK;
      const int hp = hb * 16 + (lane >> 2), slot = lane & 3;
      const int hy = hp / HW_, hx = hp - hy * HW_;
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      const bool ok = hp < HPIX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const int piece = slot ^ ((hp >> 1) & 3);
      const __bf16* src = ok ? a.x + (long)p * a.x_plane_stride + ((long)(a.in_chunk0 + kc) * Min + ((long)b * H + gy) * W + gx) * BK + piece * 8 : zero;
      glds16(src, sH + ((long)p * HROWS + hb * 16) * BK);
    }
    // this wave's column tile of the chunk's weights: lane = (column, 8-channel group), one 16-byte load per tap and plane
    bf16x8 wr[9][3];
    const __bf16* wl = a.w + ph.w_off + ((long)kc * 9 * a.Npad + nt * 16 + pi) * BK + kgrp * 8;
#pragma unroll
    for (int tt = 0; tt < 9; ++tt)
#pragma unroll
      for (int p = 0; p < 3; ++p) wr[tt][p] = *reinterpret_cast<const bf16x8*>(wl + (long)p * a.w_plane_stride + (long)tt * a.Npad * BK);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 9; ++tt) {
      const int dyx = ph.dyx[tt], dy = (int)(short)(dyx & 0xffff), dx = dyx >> 16;
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        const int g = pg * MTW + m;                                   // m-tile g: tile row g >> 1, columns 16 (g & 1) ..
        const int hp = ((g >> 1) + dy + 1) * HW_ + 16 * (g & 1) + pi + dx + 1;
        const int off = hp * BK + ((kgrp ^ ((hp >> 1) & 3)) << 3);
        bf16x8 fa[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) fa[p] = *reinterpret_cast<const bf16x8*>(sH + p * (HROWS * BK) + off);
#pragma unroll
        for (int q = 0; q < 6; ++q) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[PROD_A[q]], wr[tt][PROD_B[q]], acc[m], 0, 0, 0);
      }
    }
  }
  // ---- epilogue: the wave's MTW x 16 pixels x 16 columns through LDS, a lane then owns 8 consecutive channels of one pixel
  constexpr int TS = 20;
  __syncthreads();
  float* tw = reinterpret_cast<float*>(lds_d33) + wave * (MTW * 16 * TS);
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j) tw[(m * 16 + kgrp * 4 + j) * TS + pi] = acc[m][j];
  __syncthreads();
  // (the launch writes whole 32-channel chunks: with one column tile the groups behind it leave as zeros, from the same wave)
  const int G8 = NT == 1 ? max(2, a.e.Nchunks32 * 4) : 2;
  for (int it = lane; it < MTW * 16 * G8; it += 64) {
    const int r = it / G8, g8 = it - r * G8, g = pg * MTW + (r >> 4);
    const int y = y0 + (g >> 1), x = x0 + 16 * (g & 1) + (r & 15);
    if (y < H && x < W) {
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (g8 < 2) {
        const float4 lo = *reinterpret_cast<const float4*>(tw + r * TS + g8 * 8), hi = *reinterpret_cast<const float4*>(tw + r * TS + g8 * 8 + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
      }
      epilogue_store8(a.e, ((long)b * H + y) * W + x, nt * 16 + g8 * 8, v);
    }
  }
}

template <int NT>
constexpr int d33_lds_bytes() {
  constexpr int HROWS = ((6 * 34 + 15) / 16) * 16, stage = 3 * HROWS * BK * 2, epi = 4 * (2 * NT) * 16 * 20 * 4;
  return stage > epi ? stage : epi;
}

// Second stage of split-K: thread = (phase, row, 8 channels); the slabs are added in ascending order.
__global__ __launch_bounds__(256) void igemm_reduce_kernel(const Args a) {
  const int n8 = a.Npad / 8;
  const long total = (long)a.nphase * a.g.M * n8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n0 = (int)(i % n8) * 8;
    const long rest = i / n8;
    const int row = (int)(rest % a.g.M), phase = (int)(rest / a.g.M);
    if (n0 >= a.e.Nchunks32 * 32) continue;
    float v[8];
    sum_slabs(a.ws + ((long)a.zoff[phase] * a.g.M + row) * a.Npad + n0, (long)a.g.M * a.Npad, a.sk[phase], v);
    epilogue_store8(a.e, out_pixel(a.g, row, a.ph[phase].oy0, a.ph[phase].ox0), n0, v);
  }
}

}  // namespace

extern "C" int ufr_igemm_clock_probe(unsigned long long* buf, int capacity_workgroups) {
  UFR_REQUIRE((buf == nullptr) == (capacity_workgroups == 0) && capacity_workgroups >= 0, "igemm clock probe: buffer and capacity disagree");
  if (hipMemcpyToSymbol(HIP_SYMBOL(ufr_clock_probe_cap), &capacity_workgroups, sizeof(int)) != hipSuccess ||
      hipMemcpyToSymbol(HIP_SYMBOL(ufr_clock_probe_buf), &buf, sizeof(buf)) != hipSuccess)
    return ufr::fail(UFR_ELAUNCH, "igemm clock probe: %s", hipGetErrorString(hipGetLastError()));
  return UFR_OK;
}

namespace {
std::atomic<int> g_variant_fallbacks{0};     // launches that asked for variant 8 / 7 and ran as a plain tile form (geometry not covered)
}
extern "C" int ufr_igemm_variant_fallbacks(void) { return g_variant_fallbacks.load(); }

extern "C" int ufr_igemm(const ufr_igemm_desc* d, ufr_stream_t stream) {
  UFR_REQUIRE(d, "igemm: null descriptor");
  UFR_REQUIRE(d->x && d->w, "igemm: null operand");
  UFR_REQUIRE(d->B > 0 && d->Hi > 0 && d->Wi > 0 && d->Hr > 0 && d->Wr > 0 && d->Ho > 0 && d->Wo > 0, "igemm: bad grid");
  UFR_REQUIRE(d->KC > 0 && d->in_chunk0 >= 0 && d->N > 0 && d->Npad % 64 == 0 && d->Npad >= d->N, "igemm: bad channel counts");
  UFR_REQUIRE(d->in_sy > 0 && d->in_sx > 0 && d->out_sy > 0 && d->out_sx > 0, "igemm: bad strides");
  UFR_REQUIRE(d->nphase >= 1 && d->nphase <= 4 && d->splitk >= 1 && d->splitk <= 64, "igemm: bad phase / split count");
  UFR_REQUIRE(d->products == 6 || d->products == 3 || d->products == 1, "igemm: products must be 6, 3 or 1");
  UFR_REQUIRE(d->splitk == 1 || d->ws, "igemm: split-K needs a workspace");
  UFR_REQUIRE(!d->no_reduce || (d->splitk > 1 && d->nphase == 1), "igemm: no_reduce is for single-phase split-K launches");
  UFR_REQUIRE(d->out_planes || d->out_f32 || d->out_rowmajor, "igemm: no output");
  UFR_REQUIRE(!d->out_rowmajor || (d->out_ld >= d->N && d->N % 8 == 0 && d->out_ld % 4 == 0), "igemm: bad row-major output");
  UFR_REQUIRE(!d->act || d->bias, "igemm: the forward epilogue needs the bias");
  UFR_REQUIRE(d->planes_chunks >= 0 && d->f32_first_chunk >= 0 && ((d->planes_chunks == 0 && d->f32_first_chunk == 0) || (d->out_planes && d->out_f32)),
              "igemm: planes_chunks / f32_first_chunk need both outputs");
  UFR_REQUIRE(!d->tail || (d->tail_n0 > 0 && d->tail_n0 % 32 == 0 && d->tail_n0 < d->N), "igemm: bad tail column");
  UFR_REQUIRE(!d->row_x0 || (d->row_x0_div > 0), "igemm: bad band divisor");
  UFR_REQUIRE(!d->in_x0 || (d->in_x0_div > 0 && d->in_xw > 0), "igemm: bad input band");
  const long M = (long)d->B * d->Hr * d->Wr;
  UFR_REQUIRE(M < (1L << 30) && (long)d->B * d->Hi * d->Wi < (1L << 30) && (long)d->B * d->Ho * d->Wo < (1L << 30),
              "igemm: too many pixels");
  // every row must land inside the output grid (without a band the check is exact; with one the origins are device
  // data: the caller guarantees origin + Wr <= the full row-grid width)
  UFR_REQUIRE((d->Hr - 1) * d->out_sy + 1 <= d->Ho && (d->row_x0 || (d->Wr - 1) * d->out_sx + 1 <= d->Wo),
              "igemm: the row grid does not fit the output grid");
  Args a;
  a.x = static_cast<const __bf16*>(d->x); a.x_plane_stride = d->x_plane_stride; a.in_chunk0 = d->in_chunk0; a.KC = d->KC;
  a.Hi = d->Hi; a.Wi = d->Wi; a.in_sy = d->in_sy; a.in_sx = d->in_sx;
  a.in_x0 = d->in_x0; a.in_x0_stride = d->in_x0_stride; a.in_x0_div = d->in_x0_div; a.in_xw = d->in_xw;
  a.w = static_cast<const __bf16*>(d->w); a.w_plane_stride = d->w_plane_stride; a.Npad = d->Npad;
  a.g.B = d->B; a.g.Hr = d->Hr; a.g.Wr = d->Wr; a.g.M = (int)M;
  a.g.row_x0 = d->row_x0; a.g.row_x0_stride = d->row_x0_stride; a.g.row_x0_div = d->row_x0_div;
  a.g.Ho = d->Ho; a.g.Wo = d->Wo; a.g.out_sy = d->out_sy; a.g.out_sx = d->out_sx;
  a.e.bias = d->bias; a.e.slope = d->slope; a.e.act = d->act;
  a.e.add = d->add; a.e.add_chunk0 = d->add_chunk0;
  a.e.mask = static_cast<const __bf16*>(d->mask); a.e.mask_chunk0 = d->mask_chunk0;
  a.e.out_planes = static_cast<__bf16*>(d->out_planes); a.e.out_plane_stride = d->out_plane_stride; a.e.out_chunk0 = d->out_chunk0;
  a.e.out_f32 = d->out_f32; a.e.out_f32_chunk0 = d->out_f32_chunk0;
  a.e.out_rm = d->out_rowmajor; a.e.out_ld = d->out_ld;
  a.e.planes_chunks = d->planes_chunks > 0 ? d->planes_chunks : (1 << 30); a.e.f32_first_chunk = d->f32_first_chunk;
  a.e.tail = d->tail; a.e.tail_n0 = d->tail_n0; a.e.tail_acc = d->tail_accumulate;
  a.e.Mout = (long)d->B * d->Ho * d->Wo; a.e.N = d->N; a.e.Nchunks32 = (d->N + 31) / 32;
  a.nphase = d->nphase; a.splitk = d->splitk; a.ws = d->ws;
  a.tickets = (d->splitk > 1 && !d->no_reduce) ? d->tickets : nullptr;
  static const int skip_out = [] { const char* e = getenv("UFR_IGEMM_DEBUG_SKIP_EPILOGUE"); return e ? atoi(e) : 0; }();
  a.skip_out = skip_out == 1;
  a.e.nostore = skip_out == 2;
  a.xcd = 1;                              // XCD-aware tile order (off: +0.2 ms per iteration, profiles/r2_bench_engine_v4_no_xcd_order)
  a.korder = d->k_order ? 1 : 0;
  for (int z = 0; z < 4; ++z) {
    const ufr_igemm_phase& p = d->phase[z < d->nphase ? z : 0];
    UFR_REQUIRE(p.ntaps >= 1 && p.ntaps <= UFR_IGEMM_MAX_TAPS && p.w_off >= 0, "igemm: bad phase %d", z);
    UFR_REQUIRE(p.oy0 >= 0 && p.oy0 < d->out_sy && p.ox0 >= 0 && p.ox0 < d->out_sx, "igemm: bad phase offset");
    UFR_REQUIRE(p.w_off + (long)p.ntaps * d->KC * d->Npad * BK <= d->w_plane_stride, "igemm: phase %d weights out of range", z);
    a.ph[z].ntaps = p.ntaps; a.ph[z].oy0 = p.oy0; a.ph[z].ox0 = p.ox0; a.ph[z].w_off = p.w_off;
    for (int t = 0; t < UFR_IGEMM_MAX_TAPS; ++t)
      a.ph[z].dyx[t] = t < p.ntaps ? (int)(((unsigned)(int)p.dy[t] & 0xffffu) | ((unsigned)(int)p.dx[t] << 16)) : 0;   // (unsigned: a negative dx shifted left is UB -- found by `make sanitize`)
    for (int t = 0; t < UFR_IGEMM_MAX_TAPS; ++t) a.ph[z].run[t] = 1 << 4;         // default: every tap its own run
    // maximal runs of <= 3 taps with equal dy whose dx step by one CELL of the row grid (+-in_sx input pixels: a stride-2 launch
    // pairs the taps of one parity, igemm.py orders them so); shift = cells behind the run's first pixel
    for (int t = 0; t < p.ntaps;) {
      int len = 1, dir = 0;
      while (len < 3 && t + len < p.ntaps && p.dy[t + len] == p.dy[t]) {
        const int step = p.dx[t + len] - p.dx[t + len - 1];
        if ((step != d->in_sx && step != -d->in_sx) || (dir && step != dir)) break;
        dir = step;
        ++len;
      }
      int dmin = p.dx[t];
      for (int i = 1; i < len; ++i) dmin = p.dx[t + i] < dmin ? p.dx[t + i] : dmin;
      for (int i = 0; i < len; ++i) a.ph[z].run[t + i] = ((p.dx[t + i] - dmin) / d->in_sx) | (i << 2) | (len << 4);
      t += len;
    }
  }
  // split-K: `splitk` slices for the phase with the most taps, proportionally fewer for the others (the phases of a stride-2
  // data gradient reduce over 1, 2, 2 and 4 taps: equal slices per phase would leave the workgroups 4x apart in length)
  int ktmax = 0, nz = 0;
  for (int z = 0; z < d->nphase; ++z) ktmax = a.ph[z].ntaps * d->KC > ktmax ? a.ph[z].ntaps * d->KC : ktmax;
  a.per_k = (ktmax + d->splitk - 1) / d->splitk;
  for (int z = 0; z < 4; ++z) {
    a.zoff[z] = nz;
    a.sk[z] = z < d->nphase ? (a.ph[z].ntaps * d->KC + a.per_k - 1) / a.per_k : 0;
    nz += a.sk[z];
  }
  hipStream_t st = ufr::as_stream(stream);
  // 64-column tiles where a 128-column tile would be mostly padding.  (64-column tiles ON REQUEST for launches whose columns are a
  // multiple of 128 -- twice the workgroups per split-K slice, so half the slabs on RAFT's 48 x 160 grids -- won the cold per-launch
  // sweep on the four GRU launches by 3 - 15 % and LOST inside the step: 16.18 ms against 16.02 / 16.02 with the table without them,
  // same call, gpurun r5_call30.  Removed.)
  const int bn = d->Npad % BN == 0 ? BN : 64;
  const dim3 grid(d->Npad / bn, (unsigned)((M + BM - 1) / BM), nz);
  // kernel forms: 0 / 2 = single-stage LDS-DMA tiles (128 x 128, or 128 x 64 where Npad is not a multiple of 128),
  // 4 = 64 x 128 tiles (four workgroups per CU), 5 = pipelined 128 x 128, 6 = ping-pong 256 x 128 (128-column launches only;
  // the 64-column ones run the single-stage 128 x 64 tile, which measured faster there), 7 = ping-pong with horizontal runs of
  // taps staged once (256 x 128 or 256 x 64: stride-1 launches with >= 22 columns, the narrow-N / long-K layers of PWC-Net)
  const int variant = d->variant ? d->variant : 2;
  UFR_REQUIRE(variant == 2 || variant == 4 || variant == 5 || variant == 6 || variant == 7 || variant == 8, "igemm: unknown kernel variant %d",
              d->variant);
  // activation rows through a raw buffer resource (hardware zeros outside the frame) while the planes stay below 2 GB
  const bool use_buf = 6L * d->x_plane_stride < 0x7fffffffL;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ufr::fail(UFR_ELAUNCH, "igemm: no current device");
  // per device (a process may drive more than one) and safe between host threads: the flag is published with release / acquire,
  // two threads that race here both raise the limits (idempotent) -- the slow path goes through ufr::ensure_dynamic_lds' mutex
  static std::atomic<bool> raised[64] = {};
  if (!raised[dev].load(std::memory_order_acquire)) {
    hipError_t e = ufr::ensure_dynamic_lds(reinterpret_cast<const void*>(igemm_pp_kernel<128>), pp_lds_bytes(128));
    if (e == hipSuccess) e = ufr::ensure_dynamic_lds(reinterpret_cast<const void*>(igemm_pp_kernel<128, true>), pp_lds_bytes(128));
    if (e == hipSuccess) e = ufr::ensure_dynamic_lds(reinterpret_cast<const void*>(igemm_glds_kernel<128, 128, true>), PIPE_LDS_BYTES);
    if (e == hipSuccess) e = ufr::ensure_dynamic_lds(reinterpret_cast<const void*>(igemm_pp3_kernel<128>), pp3_lds_bytes(128));
    if (e == hipSuccess) e = ufr::ensure_dynamic_lds(reinterpret_cast<const void*>(igemm_pp3_kernel<64>), pp3_lds_bytes(64));
    if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "igemm: %s", hipGetErrorString(e));
    raised[dev].store(true, std::memory_order_release);
  }
  // 8 = the direct 3 x 3 form: one phase of nine taps within +-1, stride 1 in and out, row grid = input grid = output grid, no band,
  // no split, no tail / row-major output, chunk-major K order, <= 64 columns (anything else falls through to the tile forms)
  bool d33 = variant == 8 && d->nphase == 1 && d->phase[0].ntaps == 9 && d->in_sx == 1 && d->in_sy == 1 && d->out_sx == 1 && d->out_sy == 1 &&
             d->Hr == d->Hi && d->Wr == d->Wi && d->Ho == d->Hi && d->Wo == d->Wi && !d->row_x0 && !d->in_x0 && d->splitk == 1 && !d->tail &&
             !d->out_rowmajor && d->k_order && d->N <= 64 && d->products == 6;
  for (int t = 0; d33 && t < 9; ++t)
    d33 = d->phase[0].dy[t] >= -1 && d->phase[0].dy[t] <= 1 && d->phase[0].dx[t] >= -1 && d->phase[0].dx[t] <= 1;
  if ((variant == 8 && !d33) || (variant == 7 && !(d->k_order && (d->in_sx == 1 || d->in_sx == 2) && d->Wr >= 22 && d->products == 6)))
    g_variant_fallbacks.fetch_add(1);        // (speed only: the plain forms compute the same sums; igemm.variant_fallbacks() shows it)
  if (d33) {
    const int nt = d->N <= 16 ? 1 : (d->N <= 32 ? 2 : 4);
    const int tiles = d->B * ((d->Hi + 3) / 4) * ((d->Wi + 31) / 32);
    hipError_t e = hipSuccess;
#define UFR_D33(NT_)                                                                                                  \
    e = ufr::ensure_dynamic_lds(reinterpret_cast<const void*>(igemm_d33_kernel<NT_>), d33_lds_bytes<NT_>());          \
    if (e == hipSuccess) igemm_d33_kernel<NT_><<<tiles, 256, d33_lds_bytes<NT_>(), st>>>(a)
    if (nt == 1) { UFR_D33(1); } else if (nt == 2) { UFR_D33(2); } else { UFR_D33(4); }
#undef UFR_D33
    if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "igemm (direct 3 x 3): %s", hipGetErrorString(e));
    return ufr::launched("igemm_d33_kernel");
  }
  if (d->products != 6) {
    // reduced products (RAFT's opt-in bf16 arithmetic): the single-stage tiles, the 64 x 128 tiles and the ping-pong tiles, activation rows
    // through the buffer resource; the pipelined / tap-reuse / direct forms run as their plain twins
    UFR_REQUIRE(use_buf, "igemm: the reduced-product forms need activation planes below 2 GB");
#define UFR_NP_LAUNCH(NP_)                                                                                                    \
    if ((variant == 6 || variant == 7 || variant == 5) && bn == BN) {                                                          \
      const dim3 gpp(d->Npad / BN, (unsigned)((M + 255) / 256), nz);                                                           \
      hipError_t e = ufr::ensure_dynamic_lds(reinterpret_cast<const void*>(igemm_pp_kernel<128, true, NP_>), pp_lds_bytes(128)); \
      if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "igemm: %s", hipGetErrorString(e));                                   \
      igemm_pp_kernel<128, true, NP_><<<gpp, 512, pp_lds_bytes(128), st>>>(a);                                                 \
    } else if (variant == 4 && bn == BN) {                                                                                     \
      const dim3 g64(d->Npad / BN, (unsigned)((M + 63) / 64), nz);                                                             \
      igemm_glds_kernel<64, 128, false, true, NP_><<<g64, 256, 0, st>>>(a);                                                    \
    } else if (bn == BN) {                                                                                                     \
      igemm_glds_kernel<128, 128, false, true, NP_><<<grid, 256, 0, st>>>(a);                                                  \
    } else {                                                                                                                   \
      igemm_glds_kernel<128, 64, false, true, NP_><<<grid, 256, 0, st>>>(a);                                                   \
    }
    if (d->products == 3) { UFR_NP_LAUNCH(3) } else { UFR_NP_LAUNCH(1) }
#undef UFR_NP_LAUNCH
  } else
  if (variant == 7 && d->k_order && (d->in_sx == 1 || d->in_sx == 2) && d->Wr >= 22) {
    // ping-pong + horizontal runs of taps staged once (launches it does not cover fall through to the plain forms)
    const dim3 gpp(d->Npad / bn, (unsigned)((M + 255) / 256), nz);
    if (bn == BN) igemm_pp3_kernel<128><<<gpp, 512, pp3_lds_bytes(128), st>>>(a);
    else igemm_pp3_kernel<64><<<gpp, 512, pp3_lds_bytes(64), st>>>(a);
  } else if (variant == 4 && bn == BN) {                  // 64 x 128 tiles: four workgroups per CU
    const dim3 g64(d->Npad / BN, (unsigned)((M + 63) / 64), nz);
    if (use_buf) igemm_glds_kernel<64, 128, false, true><<<g64, 256, 0, st>>>(a);
    else igemm_glds_kernel<64, 128><<<g64, 256, 0, st>>>(a);
  } else if ((variant == 6 || variant == 7) && bn == BN) {           // ping-pong: 256-row tiles, two wave groups half a step apart
    const dim3 gpp(d->Npad / BN, (unsigned)((M + 255) / 256), nz);
    if (use_buf) igemm_pp_kernel<128, true><<<gpp, 512, pp_lds_bytes(128), st>>>(a);
    else igemm_pp_kernel<128><<<gpp, 512, pp_lds_bytes(128), st>>>(a);
  } else if (variant == 5 && bn == BN) {           // register-held fragments, DMA of the next tile under the MFMAs
    igemm_glds_kernel<128, 128, true><<<grid, 256, PIPE_LDS_BYTES, st>>>(a);
  } else if (bn == BN) {
    if (use_buf) igemm_glds_kernel<128, 128, false, true><<<grid, 256, 0, st>>>(a);
    else igemm_glds_kernel<128, 128><<<grid, 256, 0, st>>>(a);
  } else {
    if (use_buf) igemm_glds_kernel<128, 64, false, true><<<grid, 256, 0, st>>>(a);
    else igemm_glds_kernel<128, 64><<<grid, 256, 0, st>>>(a);
  }
  int rc = ufr::launched("igemm_kernel");
  if (rc != UFR_OK || d->splitk == 1 || d->no_reduce || a.tickets) return rc;     // no_reduce: the caller's next kernel adds the slabs itself;
                                                                                  // tickets: the last workgroup of every tile already has
  const long total = (long)d->nphase * M * (d->Npad / 8);
  igemm_reduce_kernel<<<ufr::stream_grid(total, 256), 256, 0, st>>>(a);
  return ufr::launched("igemm_reduce_kernel");
}
