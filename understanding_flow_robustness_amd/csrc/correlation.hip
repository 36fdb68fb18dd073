// correlation.hip -- spatial correlation sampler (FlowNetC / PWC-Net cost volume) for gfx950.
//
// Semantics follow the reference's CPU implementation, which is the runnable definition
// (models/Pytorch-Correlation-extension/Correlation_Module/correlation.cpp:75-178):
//   out[n,ph,pw,h,w] = sum_c sum_{i<kH, j<kW} in1[n,c,i1,j1] * in2[n,c,i1+su,j1+sv]
//   i1 = -padH + h*dH + i*dilH, su = (ph-(patchH-1)/2)*dil_patchH  (same for j / w),
//   terms with any index outside the image contribute 0.
// Two implementations:
//   * generic kernels: every parameter combination, fp32/fp64 (used by the check.py style cases);
//   * "fast" kernels for what the flow networks actually call (models/submodules.py:124-138,
//     models/PWCNet.py:42-50): kernel 1, stride 1, pad 0, dilation 1, square odd patch, fp32.
//
// Fast-path design (wave64, LDS tiles, register blocking) -- see DESIGN.md "correlation":
//   With dilation_patch = DP the displacement only connects pixels of equal column residue mod
//   DP, so every image row is staged in LDS *de-interleaved by residue*: plane p holds columns
//   w = DP*u + p.  In that layout a lane that owns 4 consecutive u needs one contiguous,
//   16-byte-aligned window of 4+2R values of the other image's row (R = patch radius) for all
//   P = 2R+1 horizontal displacements: (4+2R)/4 ds_read_b128 feed 4*P FMAs.
#include "ufr_common.h"

namespace {

using ufr::ceil_div;

// ------------------------------------------------------------------------------------------------
// generic kernels
// ------------------------------------------------------------------------------------------------
// T = storage type, A = accumulation type (float for float32 / float16 storage, double for float64)
template <typename T, typename A = T>
__global__ void corr_fwd_generic(const T* __restrict__ in1, const T* __restrict__ in2,
                                 T* __restrict__ out, int B, int C, int H, int W, int oH, int oW,
                                 ufr_corr_params p, float scale, float slope) {
  const long total = (long)B * p.patchH * p.patchW * oH * oW;
  const int radH = (p.patchH - 1) / 2, radW = (p.patchW - 1) / 2;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int w = (int)(idx % oW);
    const int h = (int)((idx / oW) % oH);
    const int pw = (int)((idx / ((long)oW * oH)) % p.patchW);
    const int ph = (int)((idx / ((long)oW * oH * p.patchW)) % p.patchH);
    const int n = (int)(idx / ((long)oW * oH * p.patchW * p.patchH));
    const int su = (ph - radH) * p.dilation_patchH, sv = (pw - radW) * p.dilation_patchW;
    const int u = -p.padH + h * p.dH, v = -p.padW + w * p.dW;
    const T* a = in1 + (size_t)n * C * H * W;
    const T* b = in2 + (size_t)n * C * H * W;
    A acc = 0;
    for (int c = 0; c < C; ++c)
      for (int i = 0; i < p.kH; ++i) {
        const int i1 = u + i * p.dilationH, i2 = i1 + su;
        if (i1 < 0 || i1 >= H || i2 < 0 || i2 >= H) continue;
        for (int j = 0; j < p.kW; ++j) {
          const int j1 = v + j * p.dilationW, j2 = j1 + sv;
          if (j1 < 0 || j1 >= W || j2 < 0 || j2 >= W) continue;
          acc += (A)a[((size_t)c * H + i1) * W + j1] * (A)b[((size_t)c * H + i2) * W + j2];
        }
      }
    const A r = acc * (A)scale;
    out[idx] = (T)(r > (A)0 ? r : r * (A)slope);
  }
}

// One thread per input element (n,c,y,x): gathers both adjoints, no atomics.
template <typename T, typename A = T>
__global__ void corr_bwd_generic(const T* __restrict__ in1, const T* __restrict__ in2,
                                 const T* __restrict__ gout, T* __restrict__ gin1,
                                 T* __restrict__ gin2, int B, int C, int H, int W, int oH, int oW,
                                 ufr_corr_params p) {
  const long total = (long)B * C * H * W;
  const int radH = (p.patchH - 1) / 2, radW = (p.patchW - 1) / 2;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % W);
    const int y = (int)((idx / W) % H);
    const int c = (int)((idx / ((long)W * H)) % C);
    const int n = (int)(idx / ((long)W * H * C));
    const T* a = in1 + ((size_t)n * C + c) * H * W;
    const T* b = in2 + ((size_t)n * C + c) * H * W;
    A acc1 = 0, acc2 = 0;
    for (int ph = 0; ph < p.patchH; ++ph) {
      const int su = (ph - radH) * p.dilation_patchH;
      for (int pw = 0; pw < p.patchW; ++pw) {
        const int sv = (pw - radW) * p.dilation_patchW;
        const T* g = gout + (((size_t)n * p.patchH + ph) * p.patchW + pw) * oH * oW;
        // d/d in1[y,x]: (y,x) is the in1 tap (i1,j1); partner in2 tap at (y+su, x+sv)
        const int y2 = y + su, x2 = x + sv;
        if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) {
          A gs = 0;
          for (int i = 0; i < p.kH; ++i) {
            const int hn = y + p.padH - i * p.dilationH;
            if (hn < 0 || hn % p.dH) continue;
            const int h = hn / p.dH;
            if (h >= oH) continue;
            for (int j = 0; j < p.kW; ++j) {
              const int wn = x + p.padW - j * p.dilationW;
              if (wn < 0 || wn % p.dW) continue;
              const int w = wn / p.dW;
              if (w >= oW) continue;
              gs += (A)g[(size_t)h * oW + w];
            }
          }
          acc1 += gs * (A)b[(size_t)y2 * W + x2];
        }
        // d/d in2[y,x]: (y,x) is the in2 tap (i2,j2); partner in1 tap at (y-su, x-sv)
        const int y1 = y - su, x1 = x - sv;
        if (y1 >= 0 && y1 < H && x1 >= 0 && x1 < W) {
          A gs = 0;
          for (int i = 0; i < p.kH; ++i) {
            const int hn = y1 + p.padH - i * p.dilationH;
            if (hn < 0 || hn % p.dH) continue;
            const int h = hn / p.dH;
            if (h >= oH) continue;
            for (int j = 0; j < p.kW; ++j) {
              const int wn = x1 + p.padW - j * p.dilationW;
              if (wn < 0 || wn % p.dW) continue;
              const int w = wn / p.dW;
              if (w >= oW) continue;
              gs += (A)g[(size_t)h * oW + w];
            }
          }
          acc2 += gs * (A)a[(size_t)y1 * W + x1];
        }
      }
    }
    gin1[idx] = (T)acc1;
    gin2[idx] = (T)acc2;
  }
}

// ------------------------------------------------------------------------------------------------
// fast forward: kernel 1, stride 1, pad 0, patch PxP, dilation_patch DP
//   grid  = (w tiles, H * ceil(P/PHB), B);  block = round_up(PHB*DP*G, 64) threads
//   thread (phl, p, g): 4 pixels u=4g..4g+3 of residue plane p, displacement row ph=phg*PHB+phl,
//                       all P horizontal displacements -> 4*P accumulators
//   LDS   : s1[CK][DP][4G]  (in1 row)  +  s2[CK][PHB][DP][LU]  (in2 rows incl. halo), LU = 4G+2R
// ------------------------------------------------------------------------------------------------
constexpr int kFastMaxThreads = 512;  // 2 waves/SIMD -> up to 256 VGPRs, no spills

template <int P, int DP, int PHB, int CK>
__global__ void __launch_bounds__(kFastMaxThreads) corr_fwd_fast(const float* __restrict__ in1, const float* __restrict__ in2,
                              float* __restrict__ out, int C, int H, int W, int G, float scale,
                              float slope) {
  constexpr int R = (P - 1) / 2;
  constexpr int NB4 = (4 + 2 * R) / 4;  // float4 loads per window
  static_assert((4 + 2 * R) % 4 == 0, "window must be a whole number of float4");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int U = 4 * G;         // pixels per residue plane in this tile
  const int LU = U + 2 * R;    // + halo
  const int TW = U * DP;       // tile width in pixels
  const int SPAN = TW + 2 * R * DP;
  float* s1 = smem;                     // [CK][DP][U]
  float* s2 = smem + CK * DP * U;       // [CK][PHB][DP][LU]

  constexpr int NPHG = (P + PHB - 1) / PHB;
  const int w0 = blockIdx.x * TW;
  const int h = blockIdx.y / NPHG;
  const int phg = blockIdx.y % NPHG;
  const int n = blockIdx.z;
  const int tid = threadIdx.x, NT = blockDim.x;

  const int phl = tid / (DP * G);
  const int rem = tid - phl * (DP * G);
  const int p = rem / G;
  const int g = rem - p * G;
  const int ph = phg * PHB + phl;
  const bool active = (phl < PHB) && (ph < P);

  float acc[4][P];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < P; ++k) acc[i][k] = 0.f;

  const float* a_img = in1 + (size_t)n * C * H * W;
  const float* b_img = in2 + (size_t)n * C * H * W;

  for (int c0 = 0; c0 < C; c0 += CK) {
    __syncthreads();  // previous chunk fully consumed
    // stage in1 row h (natural column order in HBM -> residue planes in LDS)
    for (int e = tid; e < CK * TW; e += NT) {
      const int ck = e / TW, wl = e - ck * TW;
      const int w = w0 + wl, c = c0 + ck;
      float v = 0.f;
      if (w < W && c < C) v = a_img[((size_t)c * H + h) * W + w];
      s1[(ck * DP + (wl % DP)) * U + wl / DP] = v;
    }
    // stage the PHB in2 rows with +-R*DP halo
    for (int e = tid; e < CK * PHB * SPAN; e += NT) {
      const int ck = e / (PHB * SPAN);
      const int r2 = e - ck * (PHB * SPAN);
      const int r = r2 / SPAN, xl = r2 - r * SPAN;
      const int w2 = w0 - R * DP + xl;
      const int h2 = h + (phg * PHB + r - R) * DP;
      const int c = c0 + ck;
      float v = 0.f;
      if (c < C && w2 >= 0 && w2 < W && h2 >= 0 && h2 < H && (phg * PHB + r) < P)
        v = b_img[((size_t)c * H + h2) * W + w2];
      s2[((ck * PHB + r) * DP + (xl % DP)) * LU + xl / DP] = v;
    }
    __syncthreads();
    if (active) {
#pragma unroll 2
      for (int ck = 0; ck < CK; ++ck) {
        const float4 a4 = *reinterpret_cast<const float4*>(&s1[(ck * DP + p) * U + 4 * g]);
        const float a[4] = {a4.x, a4.y, a4.z, a4.w};
        const float4* bp =
            reinterpret_cast<const float4*>(&s2[((ck * PHB + phl) * DP + p) * LU + 4 * g]);
        float b[4 * NB4];
#pragma unroll
        for (int q = 0; q < NB4; ++q) {
          const float4 t = bp[q];
          b[4 * q + 0] = t.x; b[4 * q + 1] = t.y; b[4 * q + 2] = t.z; b[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int k = 0; k < P; ++k) acc[i][k] = fmaf(a[i], b[i + k], acc[i][k]);
      }
    }
  }
  if (active) {
#pragma unroll
    for (int k = 0; k < P; ++k) {
      float* o = out + ((((size_t)n * P + ph) * P + k) * H + h) * W;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int w = w0 + DP * (4 * g + i) + p;
        if (w < W) {
          const float r = acc[i][k] * scale;
          o[w] = r > 0.f ? r : r * slope;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fast backward (one kernel, two roles)
//   WRT2 == false: gin1[n,c,y,x] = sum_{ph,k} g[n,ph,k,y,x] * in2[n,c,y+su,x+DP(k-R)]
//   WRT2 == true : gin2[n,c,y,x] = sum_{ph',k'} g~[ph',k',y,x] * in1[n,c,y+su',x+DP(k'-R)]
//                  g~[ph',k',y,x] = g[n,P-1-ph',P-1-k', y+su', x+DP(k'-R)]   (0 outside the image)
//   i.e. the adjoint wrt the second input is the same gather applied to the "flipped" gradient
//   volume; the flip is done by the staging addresses, nothing is materialised.
//   grid = (w tiles, H * ceil(C/CB), B); block = round_up((CB/CT)*DP*G, 64)
//   thread (cs, p, g): CT channels x 4 pixels; loops over the P displacement rows.
//   LDS: ssrc[CB][DP][LU] (other image's row, halo) + sg[P][DP][U] (gradient rows of this ph)
// ------------------------------------------------------------------------------------------------
template <int P, int DP, int CB, int CT, bool WRT2>
__global__ void __launch_bounds__(kFastMaxThreads) corr_bwd_fast(const float* __restrict__ other, const float* __restrict__ gout,
                              float* __restrict__ gin, int C, int H, int W, int G) {
  constexpr int R = (P - 1) / 2;
  constexpr int NB4 = (4 + 2 * R) / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int U = 4 * G, LU = U + 2 * R, TW = U * DP, SPAN = TW + 2 * R * DP;
  float* ssrc = smem;                // [CB][DP][LU]
  float* sg = smem + CB * DP * LU;   // [P][DP][U]

  const int NCB = (C + CB - 1) / CB;
  const int w0 = blockIdx.x * TW;
  const int y = blockIdx.y / NCB;
  const int c0 = (blockIdx.y % NCB) * CB;
  const int n = blockIdx.z;
  const int tid = threadIdx.x, NT = blockDim.x;

  const int cs = tid / (DP * G);
  const int rem = tid - cs * (DP * G);
  const int p = rem / G;
  const int g = rem - p * G;
  const bool active = cs < CB / CT;

  float acc[CT][4];
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;

  const float* o_img = other + (size_t)n * C * H * W;
  const float* g_img = gout + (size_t)n * P * P * H * W;

  for (int ph = 0; ph < P; ++ph) {
    const int ys = y + (ph - R) * DP;      // source row in the other image
    if (ys < 0 || ys >= H) continue;       // block-uniform
    __syncthreads();
    for (int e = tid; e < CB * SPAN; e += NT) {
      const int cb = e / SPAN, xl = e - cb * SPAN;
      const int w2 = w0 - R * DP + xl, c = c0 + cb;
      float v = 0.f;
      if (c < C && w2 >= 0 && w2 < W) v = o_img[((size_t)c * H + ys) * W + w2];
      ssrc[(cb * DP + (xl % DP)) * LU + xl / DP] = v;
    }
    for (int e = tid; e < P * TW; e += NT) {
      const int k = e / TW, wl = e - k * TW;
      const int w = w0 + wl;
      float v = 0.f;
      if (!WRT2) {
        if (w < W) v = g_img[(((size_t)ph * P + k) * H + y) * W + w];
      } else {
        const int ws = w + DP * (k - R);
        if (w < W && ws >= 0 && ws < W)
          v = g_img[(((size_t)(P - 1 - ph) * P + (P - 1 - k)) * H + ys) * W + ws];
      }
      sg[(k * DP + (wl % DP)) * U + wl / DP] = v;
    }
    __syncthreads();
    if (active) {
      float b[CT][4 * NB4];
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const float4* bp = reinterpret_cast<const float4*>(
            &ssrc[((cs * CT + t) * DP + p) * LU + 4 * g]);
#pragma unroll
        for (int q = 0; q < NB4; ++q) {
          const float4 v4 = bp[q];
          b[t][4 * q + 0] = v4.x; b[t][4 * q + 1] = v4.y; b[t][4 * q + 2] = v4.z; b[t][4 * q + 3] = v4.w;
        }
      }
#pragma unroll
      for (int k = 0; k < P; ++k) {
        const float4 g4 = *reinterpret_cast<const float4*>(&sg[(k * DP + p) * U + 4 * g]);
        const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[t][i] = fmaf(gg[i], b[t][i + k], acc[t][i]);
      }
    }
  }
  if (active) {
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const int c = c0 + cs * CT + t;
      if (c >= C) continue;
      float* o = gin + (((size_t)n * C + c) * H + y) * W;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int w = w0 + DP * (4 * g + i) + p;
        if (w < W) o[w] = acc[t][i];
      }
    }
  }
}

// ---- launch helpers ----------------------------------------------------------------------------
bool fast_eligible(const ufr_corr_params& p) {
  return p.kH == 1 && p.kW == 1 && p.padH == 0 && p.padW == 0 && p.dilationH == 1 &&
         p.dilationW == 1 && p.dH == 1 && p.dW == 1 && p.patchH == p.patchW &&
         p.dilation_patchH == p.dilation_patchW &&
         ((p.patchH == 21 && p.dilation_patchH == 2) || (p.patchH == 9 && p.dilation_patchH == 1));
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "hipFuncSetAttribute(LDS=%zu): %s", bytes,
                                          hipGetErrorString(e));
  }
  return UFR_OK;
}

template <int P, int DP, int PHB, int CK>
int launch_fwd_fast(const float* in1, const float* in2, float* out, int B, int C, int H, int W,
                    float scale, float slope, hipStream_t st) {
  constexpr int R = (P - 1) / 2;
  const int u_all = ceil_div(W, DP);                 // pixels per residue plane in a full row
  int G = ceil_div(u_all, 4);
  const int gmax = kFastMaxThreads / (PHB * DP);
  if (G > gmax) G = gmax;
  // keep the tile in LDS
  while (G > 1 && (size_t)CK * (DP * 4 * G + PHB * DP * (4 * G + 2 * R)) * 4 > (size_t)ufr::kMaxLds) --G;
  const int TW = 4 * G * DP;
  const size_t lds = (size_t)CK * (DP * 4 * G + PHB * DP * (4 * G + 2 * R)) * sizeof(float);
  const int NT = ufr::round_up(PHB * DP * G, 64);
  auto kern = corr_fwd_fast<P, DP, PHB, CK>;
  if (int rc = set_lds(kern, lds)) return rc;
  dim3 grid(ceil_div(W, TW), H * ((P + PHB - 1) / PHB), B);
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds, st, in1, in2, out, C, H, W, G, scale, slope);
  return ufr::launched("corr_fwd_fast");
}

template <int P, int DP, int CB, int CT, bool WRT2>
int launch_bwd_fast(const float* other, const float* gout, float* gin, int B, int C, int H, int W,
                    hipStream_t st) {
  constexpr int R = (P - 1) / 2;
  const int u_all = ceil_div(W, DP);
  int G = ceil_div(u_all, 4);
  const int gmax = kFastMaxThreads / ((CB / CT) * DP);
  if (G > gmax) G = gmax;
  while (G > 1 && (size_t)(CB * DP * (4 * G + 2 * R) + P * DP * 4 * G) * 4 > (size_t)ufr::kMaxLds) --G;
  const int TW = 4 * G * DP;
  const size_t lds = (size_t)(CB * DP * (4 * G + 2 * R) + P * DP * 4 * G) * sizeof(float);
  const int NT = ufr::round_up((CB / CT) * DP * G, 64);
  auto kern = corr_bwd_fast<P, DP, CB, CT, WRT2>;
  if (int rc = set_lds(kern, lds)) return rc;
  dim3 grid(ceil_div(W, TW), H * ceil_div(C, CB), B);
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds, st, other, gout, gin, C, H, W, G);
  return ufr::launched("corr_bwd_fast");
}

int check_common(const void* a, const void* b, const void* c, int dtype, int B, int C, int H, int W,
                 const ufr_corr_params* p, int* oH, int* oW) {
  UFR_REQUIRE(a && b && c && p, "correlation: null pointer argument");
  UFR_REQUIRE(dtype == UFR_F32 || dtype == UFR_F64 || dtype == UFR_F16, "correlation: unsupported dtype code %d", dtype);
  UFR_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "correlation: empty input [%d,%d,%d,%d]", B, C, H, W);
  UFR_REQUIRE(p->kH > 0 && p->kW > 0 && p->patchH > 0 && p->patchW > 0 && p->dH > 0 && p->dW > 0 &&
                  p->dilationH > 0 && p->dilationW > 0 && p->dilation_patchH > 0 &&
                  p->dilation_patchW > 0 && p->padH >= 0 && p->padW >= 0,
              "correlation: non-positive kernel/patch/stride/dilation parameter");
  *oH = (H + 2 * p->padH - ((p->kH - 1) * p->dilationH + 1)) / p->dH + 1;
  *oW = (W + 2 * p->padW - ((p->kW - 1) * p->dilationW + 1)) / p->dW + 1;
  UFR_REQUIRE(*oH > 0 && *oW > 0, "correlation: empty output (%d x %d)", *oH, *oW);
  return UFR_OK;
}

}  // namespace

extern "C" int ufr_corr_forward_fused(const void* input1, const void* input2, void* output,
                                      int dtype, int B, int C, int H, int W,
                                      const ufr_corr_params* p, float scale, float slope,
                                      ufr_stream_t stream) {
  int oH, oW;
  if (int rc = check_common(input1, input2, output, dtype, B, C, H, W, p, &oH, &oW)) return rc;
  hipStream_t st = ufr::as_stream(stream);
  if (dtype == UFR_F32 && fast_eligible(*p)) {
    const float* a = (const float*)input1;
    const float* b = (const float*)input2;
    float* o = (float*)output;
    // aligned single-tile rows: vectorised staging path (correlation_vec.hip); 1 = not covered
    const int vrc = ufr::corr_fwd_vec_launch(a, b, o, B, C, H, W, p->patchH, p->dilation_patchH, scale, slope, st);
    if (vrc <= 0) return vrc;
    // few rows in flight -> split the displacement rows finer so that >= ~2 workgroups per CU exist
    const long rows = (long)B * H;
    if (p->patchH == 21)
      return rows >= 256 ? launch_fwd_fast<21, 2, 7, 8>(a, b, o, B, C, H, W, scale, slope, st)
                         : launch_fwd_fast<21, 2, 3, 8>(a, b, o, B, C, H, W, scale, slope, st);
    return rows >= 512 ? launch_fwd_fast<9, 1, 9, 8>(a, b, o, B, C, H, W, scale, slope, st)
                       : launch_fwd_fast<9, 1, 3, 8>(a, b, o, B, C, H, W, scale, slope, st);
  }
  const long total = (long)B * p->patchH * p->patchW * oH * oW;
  const int grid = ufr::stream_grid(total, 256) * 4;
  if (dtype == UFR_F32)
    hipLaunchKernelGGL(corr_fwd_generic<float>, dim3(grid), dim3(256), 0, st, (const float*)input1,
                       (const float*)input2, (float*)output, B, C, H, W, oH, oW, *p, scale, slope);
  else if (dtype == UFR_F16)          // correlation_cuda_kernel.cu:262 dispatches half too; sums are kept in float32 here
    hipLaunchKernelGGL((corr_fwd_generic<_Float16, float>), dim3(grid), dim3(256), 0, st, (const _Float16*)input1,
                       (const _Float16*)input2, (_Float16*)output, B, C, H, W, oH, oW, *p, scale, slope);
  else
    hipLaunchKernelGGL(corr_fwd_generic<double>, dim3(grid), dim3(256), 0, st,
                       (const double*)input1, (const double*)input2, (double*)output, B, C, H, W,
                       oH, oW, *p, scale, slope);
  return ufr::launched("corr_fwd_generic");
}

extern "C" int ufr_corr_forward(const void* input1, const void* input2, void* output, int dtype,
                                int B, int C, int H, int W, const ufr_corr_params* p,
                                ufr_stream_t stream) {
  return ufr_corr_forward_fused(input1, input2, output, dtype, B, C, H, W, p, 1.0f, 1.0f, stream);
}

extern "C" int ufr_corr_backward(const void* input1, const void* input2, const void* grad_output,
                                 void* grad_input1, void* grad_input2, int dtype, int B, int C,
                                 int H, int W, const ufr_corr_params* p, ufr_stream_t stream) {
  int oH, oW;
  if (int rc = check_common(input1, input2, grad_output, dtype, B, C, H, W, p, &oH, &oW)) return rc;
  UFR_REQUIRE(grad_input1 && grad_input2, "correlation backward: null gradient buffer");
  hipStream_t st = ufr::as_stream(stream);
  if (dtype == UFR_F32 && fast_eligible(*p)) {
    const float* a = (const float*)input1;
    const float* b = (const float*)input2;
    const float* g = (const float*)grad_output;
    int rc = ufr::corr_bwd_vec_launch(a, b, g, (float*)grad_input1, (float*)grad_input2, B, C, H, W,
                                      p->patchH, p->dilation_patchH, st);
    if (rc <= 0) return rc;
    if (p->patchH == 21) {
      rc = launch_bwd_fast<21, 2, 32, 4, false>(b, g, (float*)grad_input1, B, C, H, W, st);
      if (rc) return rc;
      return launch_bwd_fast<21, 2, 32, 4, true>(a, g, (float*)grad_input2, B, C, H, W, st);
    }
    rc = launch_bwd_fast<9, 1, 32, 4, false>(b, g, (float*)grad_input1, B, C, H, W, st);
    if (rc) return rc;
    return launch_bwd_fast<9, 1, 32, 4, true>(a, g, (float*)grad_input2, B, C, H, W, st);
  }
  const long total = (long)B * C * H * W;
  const int grid = ufr::stream_grid(total, 256) * 4;
  if (dtype == UFR_F32)
    hipLaunchKernelGGL(corr_bwd_generic<float>, dim3(grid), dim3(256), 0, st, (const float*)input1,
                       (const float*)input2, (const float*)grad_output, (float*)grad_input1,
                       (float*)grad_input2, B, C, H, W, oH, oW, *p);
  else if (dtype == UFR_F16)          // correlation_cuda_kernel.cu:297
    hipLaunchKernelGGL((corr_bwd_generic<_Float16, float>), dim3(grid), dim3(256), 0, st, (const _Float16*)input1,
                       (const _Float16*)input2, (const _Float16*)grad_output, (_Float16*)grad_input1,
                       (_Float16*)grad_input2, B, C, H, W, oH, oW, *p);
  else
    hipLaunchKernelGGL(corr_bwd_generic<double>, dim3(grid), dim3(256), 0, st,
                       (const double*)input1, (const double*)input2, (const double*)grad_output,
                       (double*)grad_input1, (double*)grad_input2, B, C, H, W, oH, oW, *p);
  return ufr::launched("corr_bwd_generic");
}
