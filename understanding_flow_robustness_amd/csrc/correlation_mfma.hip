// correlation_mfma.hip -- both adjoints of the spatial correlation on the fp32 matrix cores (gfx950).
//
// Maths (kernel 1, stride 1, pad 0; per batch item n, output row y, residue plane p of the columns):
//   gin[c,u] = sum_ph sum_k  G[ph,k,u] * O[c, ys(ph), u + k - R]         (u = column / DP, R = patch radius)
// For a tile of 16 pixels u0..u0+15 and 16 channels this is a small GEMM over the source position
// v = u + k - R in [u0-R, u0+15+R]:
//   D[c,u] += sum_v A[c,v] * Bm[v,u],   A[c,v] = O[c,ys,v],   Bm[v,u] = G[ph, v-u+R, u]  (0 off the band)
// i.e. M = 16 channels, N = 16 pixels, K = 16+2R source positions of which P = 2R+1 carry data per
// column: 58% useful MACs for P=21 -- against the VALU kernel's 100% useful MACs that spend their time
// waiting on LDS (profiles/r1_corr_pmc*.txt: LDS active 3.3x VALU active).  v_mfma_f32_16x16x4_f32 is
// an exact fp32 FMA chain (MI355X_MICROARCH.md), so the result is bit-identical in kind to the VALU
// kernels': parity tolerances do not move.
//
// K ordering: lane (i|j = l&15, kk = l>>4) of K-step s takes v = u0 - R + NS*kk + s (NS = (16+2R)/4 steps),
// so each lane's A operands for all steps are NS CONTIGUOUS floats of its channel row, and its B
// operands NS contiguous floats of a [pixel][k] transposed gradient image:
//   LDS  ssrc[p][c][LUa]   source rows with R zero halo, LUa == 18 (mod 32) -> conflict-free ds_read_b32
//        sgT [p][u][KP]    KP = 15 + 4*NS, gradient of displacement k at index k+15, zeros around the band
// Work: workgroup = (n, y, 32 channels); wave = 2 (plane, 16-pixel tile) groups x 2 channel tiles
// (B operands are shared by the channel tiles); per displacement row ph: stage, barrier, 18*NS/9 MFMAs.
// The adjoint wrt the second input is the same product on the flipped gradient volume
// (correlation.hip::corr_bwd_fast); the flip is done by the staging addresses.
#include <cstdlib>

#include "ufr_common.h"

namespace {

using ufr::ceil_div;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// kCB channels per workgroup, kGPW (plane, pixel-tile) groups per wave; per-thread staging budget:
// kCB*QW/NT ~ kCB*kGPW/16 source pieces and P*QW/NT ~ P*kGPW/16 gradient pieces per displacement row
template <int P, int DP, bool WRT2, int kCB, int kGPW>
__global__ void __launch_bounds__(1024) corr_bwd_mfma(const float* __restrict__ other,
                                                      const float* __restrict__ gout,
                                                      float* __restrict__ gin, int C, int H, int W, int UT,
                                                      int LUa) {
  constexpr int R = (P - 1) / 2, NS = (16 + 2 * R) / 4, KP = 15 + 4 * NS;
  constexpr int kNCT = kCB / 16, kMaxS = kCB * kGPW / 16 + 1, kMaxG = (P * kGPW + 15) / 16 + 1;
  static_assert((16 + 2 * R) % 4 == 0 && R % 2 == 0, "window must split into 4 lane groups; even halo");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int U16 = 16 * UT, QW = W >> 2;
  float* ssrc = smem;                        // [DP][kCB][LUa]
  float* sgT = smem + DP * kCB * LUa;        // [DP][U16][KP]

  const int NCB = (C + kCB - 1) / kCB;
  const int pid = blockIdx.x;
  const int c0 = (pid % NCB) * kCB;          // pid % 8 == channel chunk for 8 chunks: the 21 re-reads of a
  const int y = (pid / NCB) % H;             // source row by the rows y that use it hit one XCD's L2
  const int n = pid / (NCB * H);
  const int tid = threadIdx.x, NT = blockDim.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int li = lane & 15, kk = lane >> 4;
  const long HW = (long)H * W;

  for (int i = tid; i < DP * kCB * LUa + DP * U16 * KP; i += NT) smem[i] = 0.f;

  // staging tables (constant over ph): source pieces (channel, quad) and gradient pieces (k, quad)
  int gos[kMaxS], los[kMaxS];
#pragma unroll
  for (int j = 0; j < kMaxS; ++j) {
    const int e = tid + j * NT;
    gos[j] = -1; los[j] = 0;
    if (e < kCB * QW) {
      const int cb = e / QW, q = e - cb * QW;
      if (c0 + cb < C) {
        gos[j] = cb * (int)HW + 4 * q;
        los[j] = cb * LUa + R + (4 * q) / DP;          // plane 0; plane 1 (DP == 2) is kCB*LUa further
      }
    }
  }
  int gk[kMaxG], gq[kMaxG];
#pragma unroll
  for (int j = 0; j < kMaxG; ++j) {
    const int e = tid + j * NT;
    gk[j] = -1; gq[j] = 0;
    if (e < P * QW) { gk[j] = e / QW; gq[j] = e - gk[j] * QW; }
  }

  f32x4 acc[kGPW][kNCT];
#pragma unroll
  for (int a = 0; a < kGPW; ++a)
#pragma unroll
    for (int b = 0; b < kNCT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* o_img = other + ((size_t)n * C + c0) * HW;
  const float* g_img = gout + (size_t)n * P * P * HW;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int ngrp = DP * UT;

  // displacement rows whose source row y + (ph-R)*DP lies in the image: a contiguous range
  int ph_lo = 0, ph_hi = P - 1;
  while (ph_lo < P && y + (ph_lo - R) * DP < 0) ++ph_lo;
  while (ph_hi >= 0 && y + (ph_hi - R) * DP >= H) --ph_hi;

  float4 vs[kMaxS], vg[kMaxG];
  auto prefetch = [&](int ph) {
    const int ys = y + (ph - R) * DP;
#pragma unroll
    for (int j = 0; j < kMaxS; ++j)
      vs[j] = (gos[j] >= 0) ? *reinterpret_cast<const float4*>(o_img + gos[j] + (size_t)ys * W) : zero4;
#pragma unroll
    for (int j = 0; j < kMaxG; ++j) {
      vg[j] = zero4;
      if (gk[j] >= 0) {
        const size_t off = WRT2 ? (((size_t)(P - 1 - ph) * P + (P - 1 - gk[j])) * H + ys) * W + 4 * gq[j]
                                : (((size_t)ph * P + gk[j]) * H + y) * W + 4 * gq[j];
        vg[j] = *reinterpret_cast<const float4*>(g_img + off);
      }
    }
  };
  // (Measured: issuing row ph+1's loads before row ph's MFMAs -- one register set, write after the
  //  next barrier -- is 18% SLOWER here, 0.90 vs 0.76 ms at B=8: with two workgroups per CU the other
  //  workgroup's MFMAs already cover this one's load latency, and the longer live ranges cost more.)
  for (int ph = ph_lo; ph <= ph_hi; ++ph) {
    prefetch(ph);
    __syncthreads();                                   // every wave is done with the previous ph's tiles
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      if (gos[j] < 0) continue;
      float* d = ssrc + los[j];
      if (DP == 2) {
        *reinterpret_cast<float2*>(d) = make_float2(vs[j].x, vs[j].z);
        *reinterpret_cast<float2*>(d + kCB * LUa) = make_float2(vs[j].y, vs[j].w);
      } else {
        *reinterpret_cast<float2*>(d) = make_float2(vs[j].x, vs[j].y);
        *reinterpret_cast<float2*>(d + 2) = make_float2(vs[j].z, vs[j].w);
      }
    }
#pragma unroll
    for (int j = 0; j < kMaxG; ++j) {
      if (gk[j] < 0) continue;
      const int k = gk[j];
      const float e4[4] = {vg[j].x, vg[j].y, vg[j].z, vg[j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = 4 * gq[j] + e;
        // WRT2: the value read at source position col/DP belongs to output pixel col/DP - (k-R)
        const int u = col / DP - (WRT2 ? (k - R) : 0);
        if (u >= 0 && u < U16) sgT[((col % DP) * U16 + u) * KP + k + 15] = e4[e];
      }
    }
    __syncthreads();
#pragma unroll
    for (int gi = 0; gi < kGPW; ++gi) {
      const int grp = wv * kGPW + gi;
      if (grp >= ngrp) break;                          // wave-uniform
      const int p = grp / UT, u0 = 16 * (grp - p * UT);
      const float* bb = sgT + (size_t)(p * U16 + u0 + li) * KP + NS * kk - li + 15;
      float b[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) b[s] = bb[s];
#pragma unroll
      for (int ct = 0; ct < kNCT; ++ct) {
        const float* ab = ssrc + (size_t)(p * kCB + ct * 16 + li) * LUa + u0 + NS * kk;
        float a[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) a[s] = ab[s];
#pragma unroll
        for (int s = 0; s < NS; ++s)
          acc[gi][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc[gi][ct], 0, 0, 0);
      }
    }
  }

  // D layout of the 16x16 forms: column (pixel) = lane & 15, row (channel) = 4*(lane >> 4) + r
#pragma unroll
  for (int gi = 0; gi < kGPW; ++gi) {
    const int grp = wv * kGPW + gi;
    if (grp >= ngrp) break;
    const int p = grp / UT, u0 = 16 * (grp - p * UT);
    const int x = DP * (u0 + li) + p;
    if (x >= W) continue;
#pragma unroll
    for (int ct = 0; ct < kNCT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = c0 + ct * 16 + 4 * kk + r;
        if (c < C) gin[(((size_t)n * C + c) * H + y) * W + x] = acc[gi][ct][r];
      }
  }
}

template <int P, int DP, bool WRT2, int kCB, int kGPW>
int launch(const float* other, const float* gout, float* gin, int B, int C, int H, int W, hipStream_t st) {
  constexpr int R = (P - 1) / 2, NS = (16 + 2 * R) / 4, KP = 15 + 4 * NS;
  constexpr int kMaxS = kCB * kGPW / 16 + 1, kMaxG = (P * kGPW + 15) / 16 + 1;
  const int UT = ceil_div(ceil_div(W, DP), 16), U16 = 16 * UT;
  const int ngrp = DP * UT, NW = ceil_div(ngrp, kGPW), NT = 64 * NW;
  int LUa = U16 + 2 * R;
  LUa += ((18 - LUa % 32) + 32) % 32;                         // == 18 (mod 32)
  const size_t lds = (size_t)(DP * kCB * LUa + DP * U16 * KP) * sizeof(float);
  const int QW = W / 4;
  if (NT > 1024 || lds > (size_t)ufr::kMaxLds || ceil_div(kCB * QW, NT) > kMaxS || ceil_div(P * QW, NT) > kMaxG)
    return 1;
  if ((long)kCB * H * W >= 2147483647L) return 1;
  const long nblk = (long)B * H * ceil_div(C, kCB);
  if (nblk >= 2147483647L) return 1;
  auto kern = corr_bwd_mfma<P, DP, WRT2, kCB, kGPW>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(NT), lds, st, other, gout, gin, C, H, W, UT, LUa);
  return ufr::launched("corr_bwd_mfma");
}

template <int P, int DP, int kCB, int kGPW>
int launch_pair(const float* in1, const float* in2, const float* gout, float* gin1, float* gin2, int B, int C,
                int H, int W, hipStream_t st) {
  const int rc = launch<P, DP, false, kCB, kGPW>(in2, gout, gin1, B, C, H, W, st);
  if (rc) return rc;
  return launch<P, DP, true, kCB, kGPW>(in1, gout, gin2, B, C, H, W, st);
}

}  // namespace

namespace ufr {

int corr_bwd_mfma_launch(const float* in1, const float* in2, const float* gout, float* gin1, float* gin2,
                         int B, int C, int H, int W, int P, int DP, hipStream_t st) {
  if (W % 4 != 0) return 1;
  // (channels per workgroup, groups per wave) = (32, 2): the measured best of {32, 64} x {1, 2}
  if (P == 21 && DP == 2) return launch_pair<21, 2, 32, 2>(in1, in2, gout, gin1, gin2, B, C, H, W, st);
  if (P == 9 && DP == 1) return launch_pair<9, 1, 32, 2>(in1, in2, gout, gin1, gin2, B, C, H, W, st);
  return 1;
}

}  // namespace ufr
