// correlation_vec.hip -- second-generation fast path of the spatial correlation (gfx950).
//
// Same tiling as correlation.hip's corr_*_fast kernels (residue-plane LDS rows, 4 pixels x P
// displacements per lane) for the common case "row width a multiple of 4, one tile spans the row".
// What changes is how tiles reach LDS:
//   * the (global offset, LDS offset) of every 16-byte piece a thread stages is computed ONCE in the
//     prologue -- the chunk loop does no integer division, only `base + offset`;
//   * HBM/L2 reads are global_load_dwordx4 (one row of W floats = W/4 lanes), LDS writes are
//     ds_write_b64 / b128 into the de-interleaved planes;
//   * halo columns and out-of-image rows are zeroed once, never re-written;
//   * next chunk's loads are issued before the barrier that retires the current chunk's reads, so
//     they are in flight while the other waves of the CU compute (T14-style issue-early/write-late);
//   * the forward's window is streamed 16 bytes at a time (8 live operand registers instead of 24).
#include <cstdlib>

#include "ufr_common.h"

namespace {

using ufr::ceil_div;
constexpr int kVecMaxThreads = 384;

// LDS row strides (in floats) that make the 16-byte window reads bank-conflict free.
// Lane tid = G*j + g reads the 16-byte slot (rowbase(j)/4 + g); ds_read_b128 is serviced in 16-lane
// groups whose lane ids are distinct mod 16, so it is conflict-free iff slot == tid (mod 16), i.e.
// rowbase(j) == 4*G*j (mod 64 floats).  A row of U = 4G pixels + 2R halo is therefore padded to
// U + 64*ceil(2R/64); measured before the padding: SQ_LDS_BANK_CONFLICT = 36% (forward) and 65%
// (backward) of SQ_LDS_IDX_ACTIVE (profiles/r1_corr_pmc*.txt).
__host__ __device__ constexpr int padded_row(int U, int R) { return U + ((2 * R + 63) / 64) * 64; }
// backward: thread-row j = cs*DP + p lives at cs*SC + (t*DP + p)*LU; SC must be == 4*G*DP (mod 64)
__host__ __device__ constexpr int chunk_stride(int LU, int G, int DP, int CT) {
  return CT * DP * LU + ((4 * G * DP - CT * DP * LU) % 64 + 64) % 64;
}

template <int DP>
__device__ __forceinline__ void lds_put_quad(float* plane0, int plane_stride, const float4 v) {
  if (DP == 2) {   // columns c..c+3 -> plane 0 gets (c, c+2), plane 1 gets (c+1, c+3)
    *reinterpret_cast<float2*>(plane0) = make_float2(v.x, v.z);
    *reinterpret_cast<float2*>(plane0 + plane_stride) = make_float2(v.y, v.w);
  } else {
    *reinterpret_cast<float4*>(plane0) = v;
  }
}

// ------------------------------------------------------------------------------------------------
// forward.  grid = (1, H * ceil(P/PHB), B), block = round_up(PHB*DP*G, 64), G = ceil(W/(4*DP))
// LDS: s1[CK][DP][U] + s2[CK][PHB][DP][LU]  (U = 4G, LU = U + 2R)
// ------------------------------------------------------------------------------------------------
template <int P, int DP, int PHB, int CK>
__global__ void __launch_bounds__(kVecMaxThreads) corr_fwd_vec(const float* __restrict__ in1,
                                                                const float* __restrict__ in2,
                                                                float* __restrict__ out, int C, int H,
                                                                int W, int G, float scale, float slope,
                                                                int swz) {
  constexpr int R = (P - 1) / 2, HALO = R * DP, NB4 = (4 + 2 * R) / 4;
  static_assert(HALO % 4 == 0 && (4 + 2 * R) % 4 == 0, "aligned halo/window required");
  constexpr int NPHG = (P + PHB - 1) / PHB;
  constexpr int N2 = CK, N1 = (CK + PHB - 1) / PHB;   // 16-byte pieces per thread and chunk
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int U = 4 * G, LU = padded_row(U, R), QW = W >> 2;
  float* s1 = smem;
  float* s2 = smem + CK * DP * U;

  // Workgroup -> (n, h, displacement-row group).  Workgroups (h, phg) and (h + S, phg - 1), S = DP*PHB,
  // read the SAME in2 rows (h - R*DP + S*phg + DP*{0..PHB-1}): up to NPHG workgroups form a family
  // f = h + S*phg.  swz 1 enumerates (family, phg) instead of (h, phg) and gives every family to one
  // XCD (pid % 8 labels the XCD; its L2 then serves NPHG-1 of the NPHG reads of those rows);
  // (family, phg) pairs whose h falls outside the image are empty workgroups.  Speed only.
  int h, phg, n;
  if (swz == 1) {
    constexpr int S = DP * PHB;
    const int F = H + S * (NPHG - 1), FQ = (F + 7) >> 3;
    const int pid = blockIdx.x, x = pid & 7;
    int t = pid >> 3;
    phg = t % NPHG; t /= NPHG;
    const int f = 8 * (t % FQ) + x;
    n = t / FQ;
    h = f - S * phg;
    if (f >= F || h < 0 || h >= H) return;
  } else {
    h = blockIdx.x / NPHG; phg = blockIdx.x % NPHG; n = blockIdx.y;
  }
  const int tid = threadIdx.x, NT = blockDim.x;
  const int phl = tid / (DP * G);
  const int rem = tid - phl * (DP * G);
  const int p = rem / G;
  const int g = rem - p * G;
  const int ph = phg * PHB + phl;
  const bool active = (phl < PHB) && (ph < P);
  const long HW = (long)H * W;

  for (int i = tid; i < CK * DP * U + CK * PHB * DP * LU; i += NT) smem[i] = 0.f;

  int go2[N2], lo2[N2], go1[N1], lo1[N1];
#pragma unroll
  for (int j = 0; j < N2; ++j) {
    const int e = tid + j * NT;
    go2[j] = -1; lo2[j] = 0;
    if (e < CK * PHB * QW) {
      const int ck = e / (PHB * QW), r2 = e - ck * (PHB * QW);
      const int r = r2 / QW, q = r2 - r * QW;
      const int phr = phg * PHB + r, h2 = h + (phr - R) * DP;
      if (phr < P && h2 >= 0 && h2 < H) {
        go2[j] = (ck * H + h2) * W + 4 * q;
        lo2[j] = ((ck * PHB + r) * DP) * LU + (4 * q + HALO) / DP;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < N1; ++j) {
    const int e = tid + j * NT;
    go1[j] = -1; lo1[j] = 0;
    if (e < CK * QW) {
      const int ck = e / QW, q = e - ck * QW;
      go1[j] = (ck * H + h) * W + 4 * q;
      lo1[j] = (ck * DP) * U + (4 * q) / DP;
    }
  }

  float acc[4][P];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < P; ++k) acc[i][k] = 0.f;

  const float* a_img = in1 + (size_t)n * C * HW;
  const float* b_img = in2 + (size_t)n * C * HW;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

  for (int c0 = 0; c0 < C; c0 += CK) {
    const long lim = (long)(C - c0) * HW;           // pieces at or beyond it belong to channels >= C
    const float* a_c = a_img + (size_t)c0 * HW;
    const float* b_c = b_img + (size_t)c0 * HW;
    float4 v2[N2], v1[N1];
#pragma unroll
    for (int j = 0; j < N2; ++j)
      v2[j] = (go2[j] >= 0 && go2[j] < lim) ? *reinterpret_cast<const float4*>(b_c + go2[j]) : zero4;
#pragma unroll
    for (int j = 0; j < N1; ++j)
      v1[j] = (go1[j] >= 0 && go1[j] < lim) ? *reinterpret_cast<const float4*>(a_c + go1[j]) : zero4;
    __syncthreads();   // every wave has finished reading the previous chunk (and the zero fill)
#pragma unroll
    for (int j = 0; j < N2; ++j)
      if (go2[j] >= 0) lds_put_quad<DP>(s2 + lo2[j], LU, v2[j]);
#pragma unroll
    for (int j = 0; j < N1; ++j)
      if (go1[j] >= 0) lds_put_quad<DP>(s1 + lo1[j], U, v1[j]);
    __syncthreads();
    if (active) {
#pragma unroll 2
      for (int ck = 0; ck < CK; ++ck) {
        const float4 a4 = *reinterpret_cast<const float4*>(&s1[(ck * DP + p) * U + 4 * g]);
        const float a[4] = {a4.x, a4.y, a4.z, a4.w};
        const float4* bp = reinterpret_cast<const float4*>(&s2[((ck * PHB + phl) * DP + p) * LU + 4 * g]);
#pragma unroll
        for (int q = 0; q < NB4; ++q) {
          const float4 t = bp[q];
          const float b[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int k = 4 * q + jj - i;       // window position = pixel i + displacement k
              if (k >= 0 && k < P) acc[i][k] = fmaf(a[i], b[jj], acc[i][k]);
            }
        }
      }
    }
  }
  if (active) {
#pragma unroll
    for (int k = 0; k < P; ++k) {
      float* o = out + ((((size_t)n * P + ph) * P + k) * H + h) * W;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int w = DP * (4 * g + i) + p;
        if (w < W) {
          const float r = acc[i][k] * scale;
          o[w] = r > 0.f ? r : r * slope;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward (both roles, see correlation.hip::corr_bwd_fast for the maths of the WRT2 flip).
// grid = (1, H * ceil(C/CB), B), block = round_up((CB/CT)*DP*G, 64)
// LDS: ssrc[CB][DP][LU] + sg[P][DP][U]
// ------------------------------------------------------------------------------------------------
// block size = (CB/CT)*DP*G rounded up to a wave: a thread stages at most CT source pieces and
// ceil(P/(CB/CT)) gradient pieces per displacement row (QW = W/4 <= DP*G).
constexpr int bwd_max_threads(int P) { return P > 9 ? 384 : 1024; }   // big windows need >128 VGPRs

template <int P, int DP, int CB, int CT, bool WRT2>
__global__ void __launch_bounds__(bwd_max_threads(P)) corr_bwd_vec(const float* __restrict__ other,
                                                                    const float* __restrict__ gout,
                                                                    float* __restrict__ gin, int C, int H,
                                                                    int W, int G, int swz) {
  constexpr int R = (P - 1) / 2, HALO = R * DP, NB4 = (4 + 2 * R) / 4;
  constexpr int MAXS = CT, MAXG = (P + CB / CT - 1) / (CB / CT);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int U = 4 * G, LU = padded_row(U, R), QW = W >> 2;
  const int SC = chunk_stride(LU, G, DP, CT);
  float* ssrc = smem;
  float* sg = smem + (CB / CT) * SC;

  const int NCB = (C + CB - 1) / CB;
  // Workgroup -> (n, y, channel chunk).  Workgroups are dealt round-robin over the 8 XCDs (private
  // L2s), so pid % 8 labels the XCD.  swz selects which workgroups share an L2 (speed only; every
  // choice is a bijection):
  //   swz 0: XCD = channel chunk  -> the 21 re-reads of a source row hit L2, but the gradient rows of
  //          (n, y) are fetched by all 8 XCDs;
  //   swz 1: XCD = 2*(y%4) + chunk/4 -> an XCD owns 4 chunks of every 4th row: gradient rows are
  //          fetched by 2 XCDs, the source rows it re-reads (one row parity, 4 chunks) still fit its L2.
  int y, c0, n;
  {
    const int pid = blockIdx.x;
    if (swz == 1) {
      const int x = pid & 7, slot = pid >> 3;
      const int hq = H >> 2;
      c0 = (4 * (x & 1) + (slot & 3)) * CB;
      y = 4 * ((slot >> 2) % hq) + (x >> 1);
      n = (slot >> 2) / hq;
    } else {
      c0 = (pid % NCB) * CB;
      y = (pid / NCB) % H;
      n = pid / (NCB * H);
    }
  }
  const int tid = threadIdx.x, NT = blockDim.x;
  const int cs = tid / (DP * G);
  const int rem = tid - cs * (DP * G);
  const int p = rem / G;
  const int g = rem - p * G;
  const bool active = cs < CB / CT;
  const long HW = (long)H * W;

  for (int i = tid; i < (CB / CT) * SC + P * DP * U; i += NT) smem[i] = 0.f;

  // source-row pieces: offset relative to (channel c0, row 0); the row term ys*W is added per ph
  int gos[MAXS], los[MAXS];
#pragma unroll
  for (int j = 0; j < MAXS; ++j) {
    const int e = tid + j * NT;
    gos[j] = -1; los[j] = 0;
    if (e < CB * QW) {
      const int cb = e / QW, q = e - cb * QW;
      if (c0 + cb < C) {
        gos[j] = cb * (int)HW + 4 * q;
        los[j] = (cb / CT) * SC + ((cb % CT) * DP) * LU + (4 * q + HALO) / DP;
      }
    }
  }
  // gradient-row pieces: (k, q) pairs; addresses depend on ph, see the loop
  int gk[MAXG], gq[MAXG];
#pragma unroll
  for (int j = 0; j < MAXG; ++j) {
    const int e = tid + j * NT;
    gk[j] = -1; gq[j] = 0;
    if (e < P * QW) { gk[j] = e / QW; gq[j] = e - gk[j] * QW; }
  }

  float acc[CT][4];
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;

  const float* o_img = other + ((size_t)n * C + c0) * HW;
  const float* g_img = gout + (size_t)n * P * P * HW;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

  for (int ph = 0; ph < P; ++ph) {
    const int ys = y + (ph - R) * DP;
    if (ys < 0 || ys >= H) continue;               // block-uniform
    float4 vs[MAXS], vg[MAXG];
#pragma unroll
    for (int j = 0; j < MAXS; ++j)
      vs[j] = (gos[j] >= 0) ? *reinterpret_cast<const float4*>(o_img + gos[j] + (size_t)ys * W) : zero4;
#pragma unroll
    for (int j = 0; j < MAXG; ++j) {
      vg[j] = zero4;
      if (gk[j] >= 0) {
        const size_t off = WRT2 ? (((size_t)(P - 1 - ph) * P + (P - 1 - gk[j])) * H + ys) * W + 4 * gq[j]
                                : (((size_t)ph * P + gk[j]) * H + y) * W + 4 * gq[j];
        vg[j] = *reinterpret_cast<const float4*>(g_img + off);
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MAXS; ++j)
      if (gos[j] >= 0) lds_put_quad<DP>(ssrc + los[j], LU, vs[j]);
#pragma unroll
    for (int j = 0; j < MAXG; ++j) {
      if (gk[j] < 0) continue;
      const int k = gk[j], c = 4 * gq[j];
      if (!WRT2) {
        lds_put_quad<DP>(sg + (k * DP) * U + c / DP, U, vg[j]);
      } else {
        // g~[k][x] = g[..][x + DP(k-R)]: the piece read at source columns c..c+3 belongs to output
        // columns c - DP(k-R) + {0..3}: same residue plane, index shifted by -(k-R); clip to the row.
        const float e4[4] = {vg[j].x, vg[j].y, vg[j].z, vg[j].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int col = c + e;
          const int u = col / DP - (k - R);
          if (u >= 0 && u < U) sg[(k * DP + (col % DP)) * U + u] = e4[e];
        }
      }
    }
    __syncthreads();
    if (active) {
      float b[CT][4 * NB4];
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const float4* bp = reinterpret_cast<const float4*>(&ssrc[cs * SC + (t * DP + p) * LU + 4 * g]);
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
        const int w = DP * (4 * g + i) + p;
        if (w < W) o[w] = acc[t][i];
      }
    }
  }
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return ufr::fail(UFR_ELAUNCH, "hipFuncSetAttribute(LDS=%zu): %s", bytes, hipGetErrorString(e));
  }
  return UFR_OK;
}

template <int P, int DP, int PHB, int CK>
int launch_fwd(const float* in1, const float* in2, float* out, int B, int C, int H, int W, float scale,
               float slope, hipStream_t st) {
  constexpr int R = (P - 1) / 2;
  const int G = ceil_div(ceil_div(W, DP), 4);
  const int NT = ufr::round_up(PHB * DP * G, 64);
  const size_t lds = (size_t)CK * (DP * 4 * G + PHB * DP * padded_row(4 * G, R)) * sizeof(float);
  if (NT > kVecMaxThreads || lds > (size_t)ufr::kMaxLds) return 1;   // not eligible: caller falls back
  auto kern = corr_fwd_vec<P, DP, PHB, CK>;
  if (int rc = set_lds(kern, lds)) return rc;
  constexpr int NPHG = (P + PHB - 1) / PHB;
  const int F = H + DP * PHB * (NPHG - 1), FQ = (F + 7) / 8;     // XCD-swizzled block order (the plain order measured slower)
  const long nblk = (long)B * FQ * NPHG * 8;
  if (nblk >= 2147483647L) return 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(NT), lds, st, in1, in2, out, C, H, W, G, scale, slope, 1);
  return ufr::launched("corr_fwd_vec");
}

template <int P, int DP, int CB, int CT, bool WRT2>
int launch_bwd(const float* other, const float* gout, float* gin, int B, int C, int H, int W, hipStream_t st) {
  constexpr int R = (P - 1) / 2;
  const int G = ceil_div(ceil_div(W, DP), 4);
  const int NT = ufr::round_up((CB / CT) * DP * G, 64);
  const size_t lds = (size_t)((CB / CT) * chunk_stride(padded_row(4 * G, R), G, DP, CT) + P * DP * 4 * G) * sizeof(float);
  if (NT > bwd_max_threads(P) || lds > (size_t)ufr::kMaxLds) return 1;
  if ((long)CB * H * W >= 2147483647L) return 1;   // 32-bit piece offsets
  auto kern = corr_bwd_vec<P, DP, CB, CT, WRT2>;
  if (int rc = set_lds(kern, lds)) return rc;
  const int ncb = ceil_div(C, CB);
  const long nblk = (long)B * H * ncb;
  const int swz = (ncb == 8 && H % 4 == 0 && nblk % 8 == 0) ? 1 : 0;
  if (nblk >= 2147483647L) return 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(NT), lds, st, other, gout, gin, C, H, W, G, swz);
  return ufr::launched("corr_bwd_vec");
}

}  // namespace

namespace ufr {

// Return 0 = launched, 1 = shape not covered by this path (caller uses corr_*_fast), <0 = error.
int corr_fwd_vec_launch(const float* in1, const float* in2, float* out, int B, int C, int H, int W, int P,
                        int DP, float scale, float slope, hipStream_t st) {
  if (W % 4 != 0 || (long)8 * H * W >= 2147483647L) return 1;
  // (displacement rows per block, channels per stage) = (3, 4): the measured best of round 1's sweep
  if (P == 21 && DP == 2) return launch_fwd<21, 2, 3, 4>(in1, in2, out, B, C, H, W, scale, slope, st);
  if (P == 9 && DP == 1) return launch_fwd<9, 1, 3, 4>(in1, in2, out, B, C, H, W, scale, slope, st);
  return 1;
}

int corr_bwd_vec_launch(const float* in1, const float* in2, const float* gout, float* gin1, float* gin2,
                        int B, int C, int H, int W, int P, int DP, hipStream_t st) {
  if (W % 4 != 0) return 1;
  // the fp32-MFMA formulation (correlation_mfma.hip); 1 = shape not covered -> the VALU gather kernels below
  int rc = ufr::corr_bwd_mfma_launch(in1, in2, gout, gin1, gin2, B, C, H, W, P, DP, st);
  if (rc <= 0) return rc;
  if (P == 21 && DP == 2) {
    rc = launch_bwd<21, 2, 16, 4, false>(in2, gout, gin1, B, C, H, W, st);
    if (rc) return rc;
    return launch_bwd<21, 2, 16, 4, true>(in1, gout, gin2, B, C, H, W, st);
  }
  if (P == 9 && DP == 1) {
    rc = launch_bwd<9, 1, 32, 4, false>(in2, gout, gin1, B, C, H, W, st);
    if (rc) return rc;
    return launch_bwd<9, 1, 32, 4, true>(in1, gout, gin2, B, C, H, W, st);
  }
  return 1;
}

}  // namespace ufr
