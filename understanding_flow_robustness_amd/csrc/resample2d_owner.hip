// resample2d_owner.hip -- Resample2d's adjoint WITHOUT global atomics (models/resample2d_package/resample2d_kernel.cu:75-125 image
// gradient, :127-198 flow gradient; kernel_size 1, image and flow of one size: FlowNet2's only use).
//
// The reference scatters 4*C float atomics per output pixel into the image gradient; round 2's form privatised them per
// 8 x 32 tile in LDS and still flushed ~1.7 global atomics per pixel and channel -- float atomics retire at ~28 G lanes/s
// on this chip whatever their coalescing, which put the kernel at 0.05 of the HBM roofline (0.50 ms at 8 x 448x1024).
// Here the scatter becomes OWNER-COMPUTES, in two launches:
//   A  `rs_flow_boxes_kernel`  one workgroup per 16 x 64 tile of OUTPUT pixels: the flow gradient (a gather: flow, C gradient
//      values and the 4*C image taps of every pixel, as the forward), and the bounding box of the tile's clamped sampling
//      corners -> a table of boxes [B][tiles] (16 bytes per tile);
//   B  `rs_image_owner_kernel` one workgroup OWNS a 16 x 64 tile of the IMAGE gradient: it scans the table for the output
//      tiles whose box meets its tile (smooth flow: ~4 of them), walks their pixels again (flow + C gradient values, 20
//      bytes per pixel), files the corners that fall inside its tile into per-cell SLOTS in LDS (one integer atomic per corner,
//      see the kernel) and finally adds the slots and WRITES its tile with plain coalesced stores: every element of
//      grad_input1 is written exactly once, no global atomics, no zero fill.
// History (8 x 448x1024, C = 3; profiles/r4_resample_owner_decomposition.txt): the first owner form accumulated with ds_add_f32
// and took 0.34 - 0.37 ms, 0.25 - 0.30 of them in launch B whatever the flow -- LDS float atomics, not bytes or latency.
// Algorithmic bytes per pixel (C = 3): 52 (flow 8 + gradient 12 + image 12 in, 12 + 8 out); moved here ~110 (the output tiles are
// re-read ~3x by kernel B).  Any flow field is handled: a wild one makes more boxes meet a tile (more re-reads), never a
// wrong result.  Arithmetic identical to warp_norm.hip's kernels (weights with int() truncation for the image gradient,
// floor() for the flow gradient, corners clamped to the frame).
#include "ufr_common.h"

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

constexpr int OT_H = 16, OT_W = 64;            // output tiles of kernel A = the table's granularity
constexpr int RT_H = 16, RT_W = 64;            // owner tiles of kernel B
constexpr int RS_MAX_LIST = 512;               // candidate tiles an owner lists in LDS per pass (more are walked in further passes)

struct Sample {                                // one output pixel's sampling geometry (shared by both kernels)
  int xL, xR, yT, yB;
  float xf, yf;
};
__device__ __forceinline__ Sample sample_of(float dx, float dy, int x, int y, int H, int W) {
  Sample s;
  s.xf = (float)x + dx; s.yf = (float)y + dy;
  const int fx = (int)floorf(s.xf), fy = (int)floorf(s.yf);
  s.xL = clampi(fx, 0, W - 1); s.xR = clampi(fx + 1, 0, W - 1);
  s.yT = clampi(fy, 0, H - 1); s.yB = clampi(fy + 1, 0, H - 1);
  return s;
}

// ---- A: flow gradient + boxes ---------------------------------------------------------------------------------------------
// CT_ = the channel count when it is 1..4 (the loops unroll: every load of a pixel is in flight at once), 0 = any count
template <int CT_>
__global__ __launch_bounds__(256) void rs_flow_boxes_kernel(const float* __restrict__ img, const float* __restrict__ flow,
                                                            const float* __restrict__ gout, float* __restrict__ gflow,
                                                            int4* __restrict__ boxes, int B, int Crt, int H, int W) {
  const int C = CT_ ? CT_ : Crt;
  __shared__ int red[4];
  const int tid = threadIdx.x;
  if (tid == 0) { red[0] = 1 << 30; red[1] = -(1 << 30); red[2] = 1 << 30; red[3] = -(1 << 30); }
  __syncthreads();
  const int tiles_x = (W + OT_W - 1) / OT_W, tiles_y = (H + OT_H - 1) / OT_H;
  const int b = blockIdx.x / (tiles_x * tiles_y), tr = blockIdx.x - b * tiles_x * tiles_y;
  const int x = (tr % tiles_x) * OT_W + (tid & 63), yb = (tr / tiles_x) * OT_H + (tid >> 6);
  const size_t plane = (size_t)H * W;
  int mnx = 1 << 30, mxx = -(1 << 30), mny = 1 << 30, mxy = -(1 << 30);
  float dxs[4], dys[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {                 // all four pixels' flow first: independent loads in flight
    const int y = yb + 4 * k;
    const bool live = x < W && y < H;
    const size_t pix = (size_t)(live ? y : 0) * W + (live ? x : 0);
    dxs[k] = live ? flow[((size_t)b * 2 + 0) * plane + pix] : 0.f;
    dys[k] = live ? flow[((size_t)b * 2 + 1) * plane + pix] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int y = yb + 4 * k;
    if (!(x < W && y < H)) continue;
    const size_t pix = (size_t)y * W + x;
    const Sample s = sample_of(dxs[k], dys[k], x, y, H, W);
    mnx = min(mnx, s.xL); mxx = max(mxx, s.xR); mny = min(mny, s.yT); mxy = max(mxy, s.yB);
    // resample2d_kernel.cu:127-198: channel 0 uses gamma = 1 - frac(y), channel 1 gamma = 1 - frac(x)
    const float gx = 1 - (s.yf - floorf(s.yf)), gy = 1 - (s.xf - floorf(s.xf));
    float o0 = 0.f, o1 = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float g = gout[((size_t)b * C + c) * plane + pix];
      const float* im = img + ((size_t)b * C + c) * plane;
      const float tl = im[(size_t)s.yT * W + s.xL], trv = im[(size_t)s.yT * W + s.xR];
      const float bl = im[(size_t)s.yB * W + s.xL], br = im[(size_t)s.yB * W + s.xR];
      o0 += (gx)*g * trv; o0 -= (gx)*g * tl; o0 += (1 - gx) * g * br; o0 -= (1 - gx) * g * bl;
      o1 += (gy)*g * bl;  o1 -= (gy)*g * tl; o1 += (1 - gy) * g * br; o1 -= (1 - gy) * g * trv;
    }
    gflow[((size_t)b * 2 + 0) * plane + pix] = o0;
    gflow[((size_t)b * 2 + 1) * plane + pix] = o1;
  }
  for (int off = 32; off > 0; off >>= 1) {
    mnx = min(mnx, __shfl_xor(mnx, off, 64)); mxx = max(mxx, __shfl_xor(mxx, off, 64));
    mny = min(mny, __shfl_xor(mny, off, 64)); mxy = max(mxy, __shfl_xor(mxy, off, 64));
  }
  if ((tid & 63) == 0) {
    atomicMin(&red[0], mnx); atomicMax(&red[1], mxx); atomicMin(&red[2], mny); atomicMax(&red[3], mxy);
  }
  __syncthreads();
  if (tid == 0) boxes[blockIdx.x] = make_int4(red[0], red[2], red[1], red[3]);      // x0, y0, x1, y1 (inclusive)
}

// ---- B: the image gradient, one owner per tile ------------------------------------------------------------------------------
// LDS FLOAT atomics are the wrong tool on gfx950: ds_add_f32 retires 0.38 lanes per clock and CU, ds_add_rtn_u32 13 (measured,
// profiles/r4_lds_atomics_probe.txt).  So a contribution RESERVES a slot of its cell with one INTEGER atomic (one per corner,
// shared by the C channels) and stores its C products there with plain LDS writes; the owner finally adds each cell's slots.
// A cell of a smooth flow receives four contributions (one per corner role); RS_SLOTS = 5 are kept, anything beyond (flows that
// compress, clamped frame borders) falls back to float atomics into a small overflow accumulator.
// Per candidate tile a thread walks RS_PX pixels (512 threads: 16 waves per CU at two owners per CU); loads never wait on one another: the NEXT tile's flow is fetched while this
// one is processed, and the gradient values of all pixels that reach the owner's tile are requested together.
constexpr int RS_SLOTS = 5;
constexpr int RS_NT = 1024, RS_ROWS = RS_NT / 64, RS_PX = OT_H / RS_ROWS;   // threads of an owner; output rows per step; pixels per thread and candidate tile

template <int CT_>
__global__ __launch_bounds__(RS_NT) void rs_image_owner_kernel(const float* __restrict__ flow, const float* __restrict__ gout,
                                                             const int4* __restrict__ boxes, float* __restrict__ gimg, int B,
                                                             int Crt, int H, int W) {
  constexpr int CR = CT_ ? CT_ : 1;                                            // gradient values kept in registers per pixel
  constexpr int CELLS = RT_H * RT_W;
  const int C = CT_ ? CT_ : Crt;
  extern __shared__ __attribute__((aligned(16))) float rs_smem[];              // over [C][CELLS] | cnt [CELLS] | slots [CELLS][RS_SLOTS][C]
  float* over = rs_smem;
  int* cnt = reinterpret_cast<int*>(rs_smem + C * CELLS);
  float* slots = rs_smem + C * CELLS + CELLS;
  __shared__ int list[RS_MAX_LIST];
  __shared__ int n_list;
  const int tid = threadIdx.x;
  const int rtx = (W + RT_W - 1) / RT_W, rty = (H + RT_H - 1) / RT_H;
  const int b = blockIdx.x / (rtx * rty), tr = blockIdx.x - b * rtx * rty;
  const int ry0 = (tr / rtx) * RT_H, rx0 = (tr % rtx) * RT_W;
  const int otx = (W + OT_W - 1) / OT_W, oty = (H + OT_H - 1) / OT_H, nT = otx * oty;
  const size_t plane = (size_t)H * W;
  const float* flow_b = flow + (size_t)b * 2 * plane;
  const float* gout_b = gout + (size_t)b * C * plane;
  for (int i = tid; i < C * CELLS; i += RS_NT) over[i] = 0.f;
  if (CT_) for (int i = tid; i < CELLS; i += RS_NT) cnt[i] = 0;
  auto fetch_flow = [&](int t, float (&dxs)[RS_PX], float (&dys)[RS_PX], float (&gs)[RS_PX][CR]) {
    const int x = (t % otx) * OT_W + (tid & 63), yb = (t / otx) * OT_H + (tid >> 6);
#pragma unroll
    for (int k = 0; k < RS_PX; ++k) {
      const int y = yb + RS_ROWS * k;
      const bool live = x < W && y < H;
      const size_t pix = (size_t)(live ? y : 0) * W + (live ? x : 0);
      dxs[k] = live ? flow_b[pix] : 0.f;
      dys[k] = live ? flow_b[plane + pix] : 0.f;
      if constexpr (CT_ != 0) {                 // the gradient values ride along (12 bytes per pixel more, one latency less per tile)
#pragma unroll
        for (int c = 0; c < CR; ++c) gs[k][c] = live ? gout_b[(size_t)c * plane + pix] : 0.f;
      }
    }
  };
  for (int t0 = 0; t0 < nT; t0 += RS_MAX_LIST) {                               // (one pass unless > RS_MAX_LIST tiles meet this owner)
    if (tid == 0) n_list = 0;
    __syncthreads();
    for (int t = t0 + tid; t < min(nT, t0 + RS_MAX_LIST); t += RS_NT) {
      const int4 bx = boxes[(size_t)b * nT + t];
      if (bx.z >= rx0 && bx.x < rx0 + RT_W && bx.w >= ry0 && bx.y < ry0 + RT_H) list[atomicAdd(&n_list, 1)] = t;
    }
    __syncthreads();
    const int n = n_list;
    float ndx[RS_PX], ndy[RS_PX], ng[RS_PX][CR];
    if (n > 0) fetch_flow(list[0], ndx, ndy, ng);
    for (int li = 0; li < n; ++li) {
      const int t = list[li];
      float dxs[RS_PX], dys[RS_PX], g[RS_PX][CR];
#pragma unroll
      for (int k = 0; k < RS_PX; ++k) {
        dxs[k] = ndx[k]; dys[k] = ndy[k];
#pragma unroll
        for (int c = 0; c < CR; ++c) g[k][c] = ng[k][c];
      }
      if (li + 1 < n) fetch_flow(list[li + 1], ndx, ndy, ng);                  // in flight while this tile is processed
      const int x = (t % otx) * OT_W + (tid & 63), yb = (t / otx) * OT_H + (tid >> 6);
      int cells[RS_PX][4];                                                        // LDS cell of TL, TR, BL, BR, or -1
      float wts[RS_PX][4];
      bool any[RS_PX];
#pragma unroll
      for (int k = 0; k < RS_PX; ++k) {
        const int y = yb + RS_ROWS * k;
        const Sample s = sample_of(dxs[k], dys[k], x, y, H, W);
        const int cxL = s.xL - rx0, cxR = s.xR - rx0, cyT = s.yT - ry0, cyB = s.yB - ry0;
        const bool live = x < W && y < H;
        const bool inL = (unsigned)cxL < (unsigned)RT_W, inR = (unsigned)cxR < (unsigned)RT_W;
        const bool inT = (unsigned)cyT < (unsigned)RT_H, inB = (unsigned)cyB < (unsigned)RT_H;
        // resample2d_kernel.cu:105-106: the image gradient's weights use int() truncation, not floor()
        const float alpha = s.xf - (float)(int)s.xf, beta = s.yf - (float)(int)s.yf;
        cells[k][0] = live && inT && inL ? cyT * RT_W + cxL : -1; wts[k][0] = (1 - alpha) * (1 - beta);
        cells[k][1] = live && inT && inR ? cyT * RT_W + cxR : -1; wts[k][1] = (alpha) * (1 - beta);
        cells[k][2] = live && inB && inL ? cyB * RT_W + cxL : -1; wts[k][2] = (1 - alpha) * (beta);
        cells[k][3] = live && inB && inR ? cyB * RT_W + cxR : -1; wts[k][3] = (alpha) * (beta);
        any[k] = (cells[k][0] & cells[k][1] & cells[k][2] & cells[k][3]) != -1;
      }
      if constexpr (CT_ != 0) {
#pragma unroll
        for (int k = 0; k < RS_PX; ++k) {
          if (!any[k]) continue;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int cell = cells[k][q];
            if (cell < 0) continue;
            const int slot = atomicAdd(&cnt[cell], 1) & 0x7fffffff;            // ds_add_rtn_u32: the fast kind of LDS atomic
            if (slot < RS_SLOTS) {
#pragma unroll
              for (int c = 0; c < CR; ++c) slots[(cell * RS_SLOTS + slot) * CR + c] = wts[k][q] * g[k][c];
            } else {
              // beyond the slots: float atomics into the overflow sums (slow -- 0.38 lanes per clock -- but rare for a real flow;
              // a per-cell lock bit + plain read-add-write would be the integer-atomic form, but a spin lock between the lanes of one
              // wave is a SIMT deadlock hazard under the compiler's control-flow restructuring: not used)
#pragma unroll
              for (int c = 0; c < CR; ++c) atomicAdd(&over[c * CELLS + cell], wts[k][q] * g[k][c]);
            }
          }
        }
      } else {                                                                 // any channel count: float atomics only
#pragma unroll
        for (int k = 0; k < RS_PX; ++k) {
          if (!any[k]) continue;
          for (int c = 0; c < C; ++c) {
            const float gv = gout_b[(size_t)c * plane + (size_t)(yb + RS_ROWS * k) * W + x];
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (cells[k][q] >= 0) atomicAdd(&over[c * CELLS + cells[k][q]], wts[k][q] * gv);
          }
        }
      }
    }
    __syncthreads();
  }
  // the owner adds each cell's slots and writes its tile: every element of grad_input1 exactly once
  for (int i = tid; i < CELLS; i += RS_NT) {
    const int ly = i / RT_W, lx = i - ly * RT_W;
    const int y = ry0 + ly, x = rx0 + lx;
    if (y >= H || x >= W) continue;
    const int n = CT_ ? min(cnt[i] & 0x7fffffff, RS_SLOTS) : 0;
    for (int c = 0; c < C; ++c) {
      float v = over[c * CELLS + i];
      for (int sl = 0; sl < n; ++sl) v += slots[(i * RS_SLOTS + sl) * CR + c];
      gimg[((size_t)b * C + c) * plane + (size_t)y * W + x] = v;
    }
  }
}

}  // namespace

extern "C" long ufr_resample2d_backward_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return (long)B * ((H + OT_H - 1) / OT_H) * ((W + OT_W - 1) / OT_W) * (long)sizeof(int4);
}

extern "C" int ufr_resample2d_backward_owner(const float* input1, const float* input2, const float* grad_output,
                                             float* grad_input1, float* grad_input2, void* workspace, long workspace_bytes, int B,
                                             int C, int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(input1 && input2 && grad_output && grad_input1 && grad_input2 && workspace, "resample2d backward (owner): null pointer argument");
  UFR_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && (long)B * H * W < (1L << 31), "resample2d backward (owner): bad shape");
  UFR_REQUIRE((size_t)C * RT_H * RT_W * sizeof(float) <= 96 * 1024, "resample2d backward (owner): at most %d channels", 96 * 1024 / (RT_H * RT_W * 4));
  const int CT = C <= 4 ? C : 0;
  UFR_REQUIRE(workspace_bytes >= ufr_resample2d_backward_workspace_bytes(B, H, W), "resample2d backward (owner): workspace too small");
  UFR_REQUIRE((reinterpret_cast<size_t>(workspace) & 15) == 0, "resample2d backward (owner): the workspace must be 16-byte aligned");
  hipStream_t st = ufr::as_stream(stream);
  const int ot = B * ((H + OT_H - 1) / OT_H) * ((W + OT_W - 1) / OT_W);
  int4* boxes = static_cast<int4*>(workspace);
  switch (C) {
    case 1: rs_flow_boxes_kernel<1><<<ot, 256, 0, st>>>(input1, input2, grad_output, grad_input2, boxes, B, C, H, W); break;
    case 2: rs_flow_boxes_kernel<2><<<ot, 256, 0, st>>>(input1, input2, grad_output, grad_input2, boxes, B, C, H, W); break;
    case 3: rs_flow_boxes_kernel<3><<<ot, 256, 0, st>>>(input1, input2, grad_output, grad_input2, boxes, B, C, H, W); break;
    case 4: rs_flow_boxes_kernel<4><<<ot, 256, 0, st>>>(input1, input2, grad_output, grad_input2, boxes, B, C, H, W); break;
    default: rs_flow_boxes_kernel<0><<<ot, 256, 0, st>>>(input1, input2, grad_output, grad_input2, boxes, B, C, H, W);
  }
  int rc = ufr::launched("rs_flow_boxes_kernel");
  if (rc != UFR_OK) return rc;
  const int rt = B * ((H + RT_H - 1) / RT_H) * ((W + RT_W - 1) / RT_W);
  const size_t cells = (size_t)RT_H * RT_W;
  const size_t lds = (CT ? (size_t)C * cells + cells + cells * RS_SLOTS * C : (size_t)C * cells) * sizeof(float);
  {
    const void* fns[5] = {reinterpret_cast<const void*>(rs_image_owner_kernel<0>), reinterpret_cast<const void*>(rs_image_owner_kernel<1>),
                          reinterpret_cast<const void*>(rs_image_owner_kernel<2>), reinterpret_cast<const void*>(rs_image_owner_kernel<3>),
                          reinterpret_cast<const void*>(rs_image_owner_kernel<4>)};
    hipError_t e = ufr::ensure_dynamic_lds(fns[C <= 4 ? C : 0], 120 * 1024);      // per device
    if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "resample2d backward (owner): %s", hipGetErrorString(e));
  }
  switch (C) {
    case 1: rs_image_owner_kernel<1><<<rt, RS_NT, lds, st>>>(input2, grad_output, boxes, grad_input1, B, C, H, W); break;
    case 2: rs_image_owner_kernel<2><<<rt, RS_NT, lds, st>>>(input2, grad_output, boxes, grad_input1, B, C, H, W); break;
    case 3: rs_image_owner_kernel<3><<<rt, RS_NT, lds, st>>>(input2, grad_output, boxes, grad_input1, B, C, H, W); break;
    case 4: rs_image_owner_kernel<4><<<rt, RS_NT, lds, st>>>(input2, grad_output, boxes, grad_input1, B, C, H, W); break;
    default: rs_image_owner_kernel<0><<<rt, RS_NT, lds, st>>>(input2, grad_output, boxes, grad_input1, B, C, H, W);
  }
  return ufr::launched("rs_image_owner_kernel");
}
