// correlation_planes.hip -- FlowNetC's cost volume (21 x 21 displacements, stride 2: correlation_cuda_kernel.cu:21-83 with
// models/submodules.py:124-138 `correlate` = / C and FlowNetC.py:139 LeakyReLU fused) on the bf16 matrix cores, reading the
// engine's chunk-major planes of both feature maps and writing conv3_1's input planes directly (flownetc_engine.py: `in31`
// chunks 1..14) -- no NCHW cost volume, no conversion pass.
//
// For one output row y and displacement row dy (source row y2 = y + 2(dy - 10)) the volume is a BANDED GEMM:
//     R[x, x'] = sum_c f1[c, y, x] * f2[c, y2, x'],      needed for x' - x in {-20, -18, ..., 20}
// Only same-parity columns meet, so a workgroup owns (sample, row, column parity): with i = x / 2, j = x' / 2 the band is
// |j - i| <= 10.  A wave owns 16 columns i and keeps its f1 fragments (256 channels x 3 planes = 96 VGPRs) for its whole
// life; per dy the source row's same-parity columns are streamed through LDS by LDS-DMA, one 32-channel chunk per stage,
// double-buffered (XOR-swizzled images like csrc/igemm.hip), and each wave multiplies its 16 columns against 3 tiles of 16 source columns
// (j in [i0 - 16, i0 + 32): 44 % of the MFMA work lands inside the band), float32 = six bf16 products.
// out channel d = dy * 21 + (j - i + 10); value = leaky(R / C).
#include "ufr_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ constexpr int PROD_A[6] = {2, 0, 1, 1, 0, 0};
__device__ constexpr int PROD_B[6] = {0, 2, 1, 0, 1, 0};
__device__ __attribute__((aligned(64))) unsigned corr_zero_page[16];

__device__ __forceinline__ void glds16(const __bf16* src, __bf16* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(src, lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void split3(float v, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)v;
  const float r1 = v - (float)a;
  b = (__bf16)r1;
  c = (__bf16)(r1 - (float)b);
}

constexpr int P = 21, R = 10, KCH = 8;           // 21 displacements per axis, reach 10 same-parity columns, 8 chunks of 32 channels
constexpr int OT = 24;                            // per-wave output tile: 16 columns x 21 displacements, row stride 24 floats

// f1, f2: planes [3][KCH][M][32] (M = B*H*W); out: planes [3][out_chunks][M][32], channels written at chunk out_chunk0 + d / 32.
// grid = (column blocks, 2 parities, B*H); block = NW waves; ONE workgroup per CU is resident (5 waves x 165 VGPRs), so all
// overlap is built into the workgroup:
//  * a STAGE = one 32-channel chunk of one source row (3 planes x NJ columns x 64 B) fetched by LDS-DMA into one of THREE
//    buffers, two stages ahead of its use; the wave that issued a stage's DMA retires it with a COUNTED s_waitcnt vmcnt(n)
//    (n = its DMA instructions of the one younger stage) before a raw s_barrier, and the buffer is read in the next phase
//    (a __syncthreads() would drain every DMA in flight: cdna_hip_programming.md, "Pipelining across barriers");
//  * the nine B fragments of a stage are read in one batch, then 18 MFMAs;
//  * a displacement row's 16 x 21 results per wave go through a wave-private LDS tile and leave as dense 2-byte stores
//    (18 instructions instead of 36 quarter-full ones), BEFORE the stage's DMA is issued so the counted wait stays exact.
// Workgroup -> (row, parity, column block).  The hardware deals consecutive workgroups round-robin to the 8 XCDs, each with
// its own 4 MB L2; a row of f2 is read by the 2 x 21 workgroups of the rows around it, so neighbouring rows must land on the
// SAME XCD: XCD k takes the k-th eighth of the (row, parity, block) list (a bijection when 8 divides it).
// Displacement rows are independent, so every workgroup walks them in a ROTATED order chosen so that all rows y of one parity
// read the SAME source row at the same step (row y + 2 (dy - 10) with dy = (t - y / 2) mod 21 is 2 t - 20 or 2 t + 22 for
// every y): the workgroups resident on an XCD share each fetched row through its L2.
// (The single-wave-per-tile form of this kernel -- one wave keeps all 8 chunks' f1 fragments, 96 VGPRs, 8 barriers per
// displacement row -- measured 0.356 ms against 0.325 and was removed in round 3.)
// ---- the reduction is split over TWO waves per tile ---------------------------------------------------------------------
// 2 NW waves: wave (tile tw, half kh) keeps the f1 fragments of chunks 4 kh .. 4 kh + 3 (48 VGPRs instead of 96) and a
// stage carries one chunk of EACH half, so a displacement row takes 4 barriers instead of 8 with the same 18 MFMAs per
// wave and stage; the upper half's accumulators cross to the lower half's wave through LDS once per row.  With one
// workgroup per CU this doubles the waves that share its issue slots (3,3,2,2 per SIMD instead of 2,1,1,1): 0.356 -> 0.325 ms
// at 8 pairs.  (Five buffers / four stages in flight on the single-wave form measured SLOWER, 0.417 ms: the kernel is bound
// by instruction issue and LDS traffic per stage, not by bytes in flight.)
template <int NW>
constexpr int corr_planes_k2_lds_bytes() { return 3 * 2 * 3 * (16 * NW + 32) * 32 * 2 + NW * 16 * OT * 4 + NW * 64 * 12 * 4; }

template <int NW>
__global__ __launch_bounds__(128 * NW) void corr_fwd_planes_k2_kernel(const __bf16* __restrict__ f1, const __bf16* __restrict__ f2,
                                                                      long in_plane_stride, __bf16* __restrict__ out,
                                                                      long out_plane_stride, int out_chunk0, int B, int H, int W,
                                                                      float scale, float slope, const int* __restrict__ win,
                                                                      int win_div) {
  // win != nullptr: only the 16 NW same-parity columns starting 20 cells left of the sample's window (win[b*8+1] / win_div)
  // are recomputed -- the incremental forward of the attack: between iterations of a call the features change only inside
  // the prefix window, so the volume changes only within the correlation's reach of it; grid.x = 1 then.
  constexpr int NJ = 16 * NW + 32, NRB = NJ / 16;
  constexpr int PLANE = NJ * 32, HBUF = 3 * PLANE, BUF = 2 * HBUF;        // elements: plane, one half's chunk, a stage
  constexpr int KH = KCH / 2;                                             // chunks per half
  extern __shared__ __attribute__((aligned(16))) unsigned char corr_lds[];
  __bf16* lds = reinterpret_cast<__bf16*>(corr_lds);                     // [3 buffers][2 halves][3 planes][NJ * 32]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = wave >= NW ? 1 : 0, tw = wave - kh * NW;
  float* otile = reinterpret_cast<float*>(corr_lds + 3 * BUF * 2) + tw * 16 * OT;
  float* part = reinterpret_cast<float*>(corr_lds + 3 * BUF * 2 + NW * 16 * OT * 4) + (tw * 64 + lane) * 12;
  const int nblk = gridDim.x * gridDim.y * gridDim.z;                    // XCD-contiguous rows: see the notes above
  int item = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  if ((nblk & 7) == 0) item = (item & 7) * (nblk >> 3) + (item >> 3);
  const int bx = item % gridDim.x, par = (item / gridDim.x) % gridDim.y, by = item / (gridDim.x * gridDim.y);
  const int b = by / H, y = by - b * H;
  int i0 = bx * 16 * NW;
  if (win) i0 = min(max((win[b * 8 + 1] / win_div - 2 * R) >> 1, 0), max((W + 1) / 2 - 16 * NW, 0));
  const long M = (long)B * H * W;
  const long rowbase = ((long)b * H + y) * W;
  const int ai = i0 + 16 * tw + (lane & 15), ax = 2 * ai + par;
  const bool a_ok = ax < W;
  const __bf16* zero = reinterpret_cast<const __bf16*>(corr_zero_page);
  bf16x8 fa[KH][3];
#pragma unroll
  for (int i = 0; i < KH; ++i)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const __bf16* src = a_ok ? f1 + p * in_plane_stride + ((long)(kh * KH + i) * M + rowbase + ax) * 32 + (lane >> 4) * 8 : zero;
      fa[i][p] = *reinterpret_cast<const bf16x8*>(src);
    }
  // staging plan: 2 NRB (half, row block) items per plane over 2 NW waves: item = wave + it * 2 NW
  const int srow_in = lane >> 2, spiece = lane & 3;
  int voff[2], sdst[2], shalf[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int itm = wave + it * 2 * NW, h = itm >= NRB ? 1 : 0, rb = itm - h * NRB, r = rb * 16 + srow_in;
    const int j = i0 - 16 + r, xs = 2 * j + par;
    voff[it] = (itm < 2 * NRB && j >= 0 && xs < W) ? xs * 32 + ((spiece ^ ((r >> 1) & 3)) << 3) : -1;
    sdst[it] = h * HBUF + rb * 16 * 32;
    shalf[it] = h;
  }
  const bool two = wave + 2 * NW < 2 * NRB;      // uniform: 6 (else 3) DMA instructions per stage
  const int frow = lane & 15;
  const int foff = frow * 32 + (((lane >> 4) ^ ((frow >> 1) & 3)) << 3);
  auto stage = [&](int dy, int sl, int buf) {    // chunks sl and KH + sl of source row y + 2 (dy - R) -> buffer `buf`
    const long rowoff = ((long)b * H + (y + 2 * (dy - R))) * W;
    __bf16* dst = lds + buf * BUF;
    {
      const __bf16* sbase = f2 + ((long)(shalf[0] * KH + sl) * M + rowoff) * 32;
#pragma unroll
      for (int p = 0; p < 3; ++p) glds16(voff[0] >= 0 ? sbase + p * in_plane_stride + voff[0] : zero, dst + sdst[0] + p * PLANE);
    }
    if (two) {
      const __bf16* sbase = f2 + ((long)(shalf[1] * KH + sl) * M + rowoff) * 32;
#pragma unroll
      for (int p = 0; p < 3; ++p) glds16(voff[1] >= 0 ? sbase + p * in_plane_stride + voff[1] : zero, dst + sdst[1] + p * PLANE);
    }
  };
  const int rot = (P - (y >> 1) % P) % P;        // rotated displacement order: see the notes above
  auto dy_at = [&](int t) { const int d = t + rot; return d >= P ? d - P : d; };
  auto row_ok = [&](int dy) { const int y2 = y + 2 * (dy - R); return y2 >= 0 && y2 < H; };
  auto next_valid = [&](int t) { do { ++t; } while (t < P && !row_ok(dy_at(t))); return t; };
  auto write_row = [&](int d_row, const f32x4 (&acc)[3], bool zeros) {
    if (!zeros) {
#pragma unroll
      for (int tt = 0; tt < 3; ++tt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int il = (lane >> 4) * 4 + rg;
          const int dx = 16 * tt - 16 + (lane & 15) - il + R;
          if (dx >= 0 && dx < P) otile[il * OT + dx] = acc[tt][rg];
        }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int e = lane + 64 * q;
      const int il = e / P, dx = e - il * P;
      const int x = 2 * (i0 + 16 * tw + il) + par;
      if (e < 16 * P && x < W) {
        float v = zeros ? 0.f : otile[il * OT + dx] * scale;
        v = v > 0.f ? v : v * slope;
        __bf16 p0, p1, p2;
        split3(v, p0, p1, p2);
        const int d = d_row * P + dx;
        __bf16* o = out + ((long)(out_chunk0 + (d >> 5)) * M + rowbase + x) * 32 + (d & 31);
        o[0] = p0;
        o[out_plane_stride] = p1;
        o[2 * out_plane_stride] = p2;
      }
    }
  };
  const f32x4 zacc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  if (kh == 0)
    for (int d0 = 0; d0 < P; ++d0)
      if (!row_ok(d0)) write_row(d0, zacc, true);
  int t = next_valid(-1);
  if (t >= P) return;
  int cur = 0;
  stage(dy_at(t), 0, 0);
  stage(dy_at(t), 1, 1);
  if (two) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  while (t < P) {
    const int dy = dy_at(t), tn = next_valid(t);
    f32x4 acc[3];
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sl = 0; sl < KH; ++sl) {            // unrolled: fa[] indexed statically
      const bool ahead = sl + 2 < KH || tn < P;  // the stage two ahead: (dy, sl + 2) or (next valid row, sl + 2 - KH)
      const int nxt = cur == 0 ? 2 : cur - 1;
      if (ahead) stage(sl + 2 < KH ? dy : dy_at(tn), (sl + 2) & (KH - 1), nxt);
      bf16x8 fb[3][3];
      const __bf16* bsrc = lds + cur * BUF + kh * HBUF + tw * 16 * 32 + foff;
#pragma unroll
      for (int tt = 0; tt < 3; ++tt)
#pragma unroll
        for (int p = 0; p < 3; ++p) fb[tt][p] = *reinterpret_cast<const bf16x8*>(bsrc + p * PLANE + tt * 16 * 32);
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[sl][PROD_A[q]], fb[tt][PROD_B[q]], acc[tt], 0, 0, 0);
      if (sl == KH - 1 && kh == 1) {              // upper half: hand the partial sums to the tile's lower-half wave
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) *reinterpret_cast<f32x4*>(part + 4 * tt) = acc[tt];
      }
      if (ahead) {
        if (two) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur = cur == 2 ? 0 : cur + 1;
      if (sl == KH - 1 && kh == 0) {              // lower half: add, then the row leaves (its stores precede the next DMA issue)
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(part + 4 * tt);
          acc[tt][0] += o[0]; acc[tt][1] += o[1]; acc[tt][2] += o[2]; acc[tt][3] += o[3];
        }
        write_row(dy, acc, false);
      }
    }
    t = tn;
  }
}

template <int NW>
int launch_corr_planes(const __bf16* a, const __bf16* b, long in_plane_stride, __bf16* o, long out_plane_stride, int out_chunk0,
                       int B, int H, int W, int ni, float scale, float slope, hipStream_t st, const int* win = nullptr,
                       int win_div = 1) {
  {                                              // above the default dynamic-LDS limit
    hipError_t e = ufr::ensure_dynamic_lds(reinterpret_cast<const void*>(corr_fwd_planes_k2_kernel<NW>), corr_planes_k2_lds_bytes<NW>());
    if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "correlation (planes): %s", hipGetErrorString(e));
  }
  corr_fwd_planes_k2_kernel<NW><<<dim3(win ? 1 : ufr::ceil_div(ni, 16 * NW), 2, B * H), 128 * NW, corr_planes_k2_lds_bytes<NW>(), st>>>(
      a, b, in_plane_stride, o, out_plane_stride, out_chunk0, B, H, W, scale, slope, win, win_div);
  return UFR_OK;
}

}  // namespace

extern "C" int ufr_corr_forward_planes(const void* f1_planes, const void* f2_planes, long in_plane_stride, void* out_planes,
                                       long out_plane_stride, int out_chunk0, int B, int C, int H, int W, int patch,
                                       int dilation_patch, float scale, float slope, ufr_stream_t stream) {
  UFR_REQUIRE(f1_planes && f2_planes && out_planes, "correlation (planes): null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && (long)B * H < 65536 && out_chunk0 >= 0, "correlation (planes): bad shape");
  if (C != 32 * KCH || patch != P || dilation_patch != 2)
    return ufr::fail(UFR_EUNSUPPORTED, "correlation (planes): built for FlowNetC's configuration (256 channels, patch 21, "
                                       "dilation_patch 2); got C=%d patch=%d dilation_patch=%d", C, patch, dilation_patch);
  const int ni = (W + 1) / 2;                    // same-parity columns of a row (parity 0; parity 1 has W / 2)
  hipStream_t st = ufr::as_stream(stream);
  const __bf16* a = static_cast<const __bf16*>(f1_planes);
  const __bf16* b = static_cast<const __bf16*>(f2_planes);
  __bf16* o = static_cast<__bf16*>(out_planes);
  const int waves = ufr::ceil_div(ni, 16);
  int rc;
  if (waves <= 2) rc = launch_corr_planes<2>(a, b, in_plane_stride, o, out_plane_stride, out_chunk0, B, H, W, ni, scale, slope, st);
  else if (waves <= 4 || waves > 5 * 2)                         // 4-wave blocks also tile rows wider than one block
    rc = launch_corr_planes<4>(a, b, in_plane_stride, o, out_plane_stride, out_chunk0, B, H, W, ni, scale, slope, st);
  else                                                          // 160 columns (FlowNetC @1280): one block per row and parity
    rc = launch_corr_planes<5>(a, b, in_plane_stride, o, out_plane_stride, out_chunk0, B, H, W, ni, scale, slope, st);
  if (rc != UFR_OK) return rc;
  return ufr::launched("corr_fwd_planes_kernel");
}

extern "C" int ufr_corr_forward_planes_window(const void* f1_planes, const void* f2_planes, long in_plane_stride, void* out_planes,
                                              long out_plane_stride, int out_chunk0, int B, int C, int H, int W, int patch,
                                              int dilation_patch, float scale, float slope, const int* win, int level_stride,
                                              int win_cells, ufr_stream_t stream) {
  UFR_REQUIRE(f1_planes && f2_planes && out_planes && win, "correlation (planes, window): null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && (long)B * H < 65536 && out_chunk0 >= 0 && level_stride > 0 && win_cells > 0,
              "correlation (planes, window): bad shape");
  if (C != 32 * KCH || patch != P || dilation_patch != 2 || win_cells + 4 * R > 63)
    return ufr::fail(UFR_EUNSUPPORTED, "correlation (planes, window): 256 channels, patch 21, dilation_patch 2, windows of at most "
                                       "23 cells; got C=%d patch=%d dilation_patch=%d cells=%d", C, patch, dilation_patch, win_cells);
  // 2 x 32 same-parity columns from 20 cells left of the window cover [x0 - 20, x0 + cells + 20): every cell whose cost
  // volume can have changed when the features changed inside the window only
  int rc = launch_corr_planes<2>(static_cast<const __bf16*>(f1_planes), static_cast<const __bf16*>(f2_planes), in_plane_stride,
                                 static_cast<__bf16*>(out_planes), out_plane_stride, out_chunk0, B, H, W, (W + 1) / 2, scale, slope,
                                 ufr::as_stream(stream), win, level_stride);
  if (rc != UFR_OK) return rc;
  return ufr::launched("corr_fwd_planes_k2_kernel (window)");
}
