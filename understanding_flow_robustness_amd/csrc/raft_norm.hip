// raft_norm.hip -- the normalisation / ReLU / residual arithmetic of RAFT's BasicEncoder (models/raft/extractor.py:5-78,
// :142-215) between its convolutions, on the engine's chunk-major layout (convolutions: csrc/igemm.hip writes a float32 tensor
// X [chunks][M][32], M = n*HW pixels in (image, y, x) order):
//   InstanceNorm2d (fnet; no affine, batch statistics always, biased variance, eps 1e-5):
//     stats      per (image, channel) sum x, sum x^2 over HW in float64 partials -> mean, 1 / sqrt(var + eps)
//     apply      out = relu(res + relu((x - mean) * rstd))      (res / either ReLU optional)  -> activation planes
//     adjoint    g = G * [out > 0] * [xhat > 0];  gx = rstd * (g - mean_HW(g) - xhat * mean_HW(g * xhat))  -> gradient planes
//   BatchNorm2d in eval mode (cnet) is folded into the convolution's weights and bias by the host: the same kernels run with
//   `stats` = NULL (mean 0, rstd 1, no statistics terms in the adjoint).
// All HBM-streaming; reductions in a fixed order (bit-reproducible).
#include "ufr_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ void split3(float v, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)v;
  const float r1 = v - (float)a;
  b = (__bf16)r1;
  c = (__bf16)(r1 - (float)b);
}

__device__ __forceinline__ void store_planes8(__bf16* p, long plane_stride, const float v[8]) {
  bf16x8 q0, q1, q2;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    __bf16 x, y, z;
    split3(v[j], x, y, z);
    q0[j] = x; q1[j] = y; q2[j] = z;
  }
  *reinterpret_cast<bf16x8*>(p) = q0;
  *reinterpret_cast<bf16x8*>(p + plane_stride) = q1;
  *reinterpret_cast<bf16x8*>(p + 2 * plane_stride) = q2;
}

__device__ __forceinline__ void load_planes8(const __bf16* p, long plane_stride, float v[8]) {
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
  const bf16x8 b = *reinterpret_cast<const bf16x8*>(p + plane_stride);
  const bf16x8 c = *reinterpret_cast<const bf16x8*>(p + 2 * plane_stride);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = ((float)a[j] + (float)b[j]) + (float)c[j];
}

__device__ __forceinline__ void load_f8(const float* p, float v[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// Partial sums of two per-element quantities over a slice of one image's pixels, for one chunk: workgroup (slice, image, chunk),
// thread = (row offset t / 4, 8-channel group t % 4); partial[((slice * n + image) * C + channel) * 2 + {0, 1}] (float64).
// MODE 0: (x, x^2).  MODE 1: (g, g * xhat) with g = G * [outmask > 0] * [xhat > 0 if relu1].
template <int MODE>
__global__ __launch_bounds__(256) void cm_sums_kernel(const float* __restrict__ x, const float* __restrict__ G,
                                                      const __bf16* __restrict__ outmask, int mask_chunk0,
                                                      const float* __restrict__ stats, double* __restrict__ partial, long HW, int n,
                                                      int chunks, int relu1) {
  const int slice = blockIdx.x, S = gridDim.x, img = blockIdx.y, ch = blockIdx.z;
  const int tid = threadIdx.x, q = tid & 3, ro = tid >> 2;
  const long M = (long)n * HW;
  const long per = (HW + S - 1) / S, r0 = slice * per, r1 = min(HW, r0 + per);
  const int C = chunks * 32;
  float mean[8], rstd[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = ch * 32 + q * 8 + j;
    mean[j] = (MODE == 1 && stats) ? stats[((long)img * C + c) * 2] : 0.f;
    rstd[j] = (MODE == 1 && stats) ? stats[((long)img * C + c) * 2 + 1] : 1.f;
  }
  double a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = b[j] = 0.0;
  for (long r = r0 + ro; r < r1; r += 64) {
    const long e = ((long)ch * M + (long)img * HW + r) * 32 + q * 8;
    float xv[8];
    load_f8(x + e, xv);
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[j] += (double)xv[j]; b[j] += (double)xv[j] * (double)xv[j]; }
    } else {
      float gv[8];
      load_f8(G + e, gv);
      if (outmask) {
        const bf16x8 m = *reinterpret_cast<const bf16x8*>(outmask + ((long)(mask_chunk0 + ch) * M + (long)img * HW + r) * 32 + q * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) gv[j] = (float)m[j] > 0.f ? gv[j] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (xv[j] - mean[j]) * rstd[j];
        const float g = (relu1 && !(xh > 0.f)) ? 0.f : gv[j];
        a[j] += (double)g;
        b[j] += (double)g * (double)xh;
      }
    }
  }
  // reduce the 64 row offsets in a fixed order: thread (ro, q) -> LDS [ro][q][j]; 32 threads (q, j) then sum ro ascending
  __shared__ double buf[64][32][2];
#pragma unroll
  for (int j = 0; j < 8; ++j) { buf[ro][q * 8 + j][0] = a[j]; buf[ro][q * 8 + j][1] = b[j]; }
  __syncthreads();
  if (tid < 64) {
    const int c = tid >> 1, k = tid & 1;
    double s = 0.0;
    for (int r = 0; r < 64; ++r) s += buf[r][c][k];
    partial[(((long)slice * n + img) * C + ch * 32 + c) * 2 + k] = s;
  }
}

// MODE 0: stats[(img * C + c) * 2 + {0, 1}] = mean, 1 / sqrt(var + eps).  MODE 1: = sum g / HW, sum g xhat / HW.
template <int MODE>
__global__ void cm_sums_finalize(const double* __restrict__ partial, float* __restrict__ out, int S, long nC, double HW, double eps) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nC) return;
  double a = 0.0, b = 0.0;
  for (int s = 0; s < S; ++s) { a += partial[((long)s * nC + i) * 2]; b += partial[((long)s * nC + i) * 2 + 1]; }
  if (MODE == 0) {
    const double mean = a / HW, var = fmax(b / HW - mean * mean, 0.0);
    out[i * 2] = (float)mean;
    out[i * 2 + 1] = (float)(1.0 / sqrt(var + eps));
  } else {
    out[i * 2] = (float)(a / HW);
    out[i * 2 + 1] = (float)(b / HW);
  }
}

// out planes = relu2(res + relu1((x - mean) * rstd))
__global__ void cm_norm_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats, const __bf16* __restrict__ res,
                                     long res_stride, int res_chunk0, __bf16* __restrict__ out, long out_stride, int out_chunk0, long HW,
                                     int n, int chunks, int relu1, int relu2) {
  const long M = (long)n * HW, total = (long)chunks * M * 4;
  const int C = chunks * 32;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long e = t * 8;
    const int ch = (int)(e / (M * 32)), c0 = ch * 32 + (int)(e & 31);
    const long m = (e / 32) % M;
    const int img = (int)(m / HW);
    float v[8];
    load_f8(x + e, v);
    if (stats) {
      const float* sp = stats + ((long)img * C + c0) * 2;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (v[j] - sp[2 * j]) * sp[2 * j + 1];
    }
    if (relu1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
    }
    if (res) {
      float r[8];
      load_planes8(res + (long)res_chunk0 * M * 32 + e, res_stride, r);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += r[j];
    }
    if (relu2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
    }
    store_planes8(out + (long)out_chunk0 * M * 32 + e, out_stride, v);
  }
}

// gz planes = rstd * (g - sums[0] - xhat * sums[1]),  g = G * [outmask > 0] * [xhat > 0 if relu1]   (sums = NULL: no statistics terms)
__global__ void cm_norm_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ G, const __bf16* __restrict__ outmask,
                                         int mask_chunk0, const float* __restrict__ stats, const float* __restrict__ sums,
                                         __bf16* __restrict__ gz, long gz_stride, int gz_chunk0, long HW, int n, int chunks, int relu1) {
  const long M = (long)n * HW, total = (long)chunks * M * 4;
  const int C = chunks * 32;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long e = t * 8;
    const int ch = (int)(e / (M * 32)), c0 = ch * 32 + (int)(e & 31);
    const long m = (e / 32) % M;
    const int img = (int)(m / HW);
    float xv[8], gv[8];
    load_f8(x + e, xv);
    load_f8(G + e, gv);
    if (outmask) {
      const bf16x8 mk = *reinterpret_cast<const bf16x8*>(outmask + (long)mask_chunk0 * M * 32 + e);
#pragma unroll
      for (int j = 0; j < 8; ++j) gv[j] = (float)mk[j] > 0.f ? gv[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long sc = ((long)img * C + c0 + j) * 2;
      const float mean = stats ? stats[sc] : 0.f, rstd = stats ? stats[sc + 1] : 1.f;
      const float xh = (xv[j] - mean) * rstd;
      float g = (relu1 && !(xh > 0.f)) ? 0.f : gv[j];
      if (sums) g = g - sums[sc] - xh * sums[sc + 1];
      gv[j] = rstd * g;
    }
    store_planes8(gz + (long)gz_chunk0 * M * 32 + e, gz_stride, gv);
  }
}

// Gs += G * [outmask > 0]  (the skip connection's share of a residual block's output gradient); Gs may be G itself (in place)
__global__ void cm_masked_copy_kernel(const float* __restrict__ G, const __bf16* __restrict__ outmask, int mask_chunk0,
                                      float* __restrict__ out, long total8) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total8; t += (long)gridDim.x * blockDim.x) {
    const long e = t * 8;
    float gv[8];
    load_f8(G + e, gv);
    const bf16x8 mk = *reinterpret_cast<const bf16x8*>(outmask + e);
#pragma unroll
    for (int j = 0; j < 8; ++j) gv[j] = (float)mk[j] > 0.f ? gv[j] : 0.f;
    *reinterpret_cast<float4*>(out + e) = make_float4(gv[0], gv[1], gv[2], gv[3]);
    *reinterpret_cast<float4*>(out + e + 4) = make_float4(gv[4], gv[5], gv[6], gv[7]);
  }
}

int slices_for(long HW, int n, int chunks) {
  int S = 1;
  while ((long)S * n * chunks < 512 && HW / (S * 2) >= 256 && S < 64) S *= 2;
  return S;
}

// ---- the statistics' second stage inside their consumers (round 4) --------------------------------------------------------------
// A launch of its own for `cm_sums_finalize` cost 13 us thirty times per RAFT step for a few hundred numbers.  The consumers are
// launched as workgroup (row slab, image, chunk) instead: 64 threads add the S float64 partials of the workgroup's 32 channels in
// ascending order -- the arithmetic of cm_sums_finalize, bit for bit -- into LDS, slab 0 also writes them out (the adjoint reads
// the forward's statistics again), then every thread streams rows (tid / 4) + 64 k of the slab, 8 channels each.
template <int MODE>
__device__ __forceinline__ void cm_stats_prologue(const double* __restrict__ partial, int S, int n, int C, int img, int ch, double HW,
                                                  double eps, float (*st)[2], float* __restrict__ out, bool write_out) {
  const int tid = threadIdx.x;
  if (tid < 64) {
    const int c = tid >> 1, k = tid & 1;
    const long nC = (long)n * C, i = (long)img * C + ch * 32 + c;
    double v = 0.0;
    for (int s = 0; s < S; ++s) v += partial[((long)s * nC + i) * 2 + k];
    const double other = __shfl_xor(v, 1, 64);                 // k = 0 holds the first sum, k = 1 the second
    float r;
    if (MODE == 0) {
      const double a = k ? other : v, b = k ? v : other;
      const double mean = a / HW, var = fmax(b / HW - mean * mean, 0.0);
      r = k ? (float)(1.0 / sqrt(var + eps)) : (float)mean;
    } else {
      r = (float)(v / HW);
    }
    st[c][k] = r;
    if (write_out) out[i * 2 + k] = r;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void cm_norm_apply_fused_kernel(const float* __restrict__ x, const double* __restrict__ partial, int S,
                                                                  double eps, float* __restrict__ stats, const __bf16* __restrict__ res,
                                                                  long res_stride, int res_chunk0, __bf16* __restrict__ out,
                                                                  long out_stride, int out_chunk0, long HW, int n, int chunks, int relu1,
                                                                  int relu2) {
  __shared__ float st[32][2];
  const int slab = blockIdx.x, slabs = gridDim.x, img = blockIdx.y, ch = blockIdx.z;
  const int tid = threadIdx.x, q = tid & 3, ro = tid >> 2;
  const long M = (long)n * HW;
  cm_stats_prologue<0>(partial, S, n, chunks * 32, img, ch, (double)HW, eps, st, stats, slab == 0);
  const long per = (HW + slabs - 1) / slabs, r0 = slab * per, r1 = min(HW, r0 + per);
  float mean[8], rstd[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { mean[j] = st[q * 8 + j][0]; rstd[j] = st[q * 8 + j][1]; }
  for (long r = r0 + ro; r < r1; r += 64) {
    const long e = ((long)ch * M + (long)img * HW + r) * 32 + q * 8;
    float v[8];
    load_f8(x + e, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (v[j] - mean[j]) * rstd[j];
    if (relu1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
    }
    if (res) {
      float rv[8];
      load_planes8(res + (long)res_chunk0 * M * 32 + e, res_stride, rv);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += rv[j];
    }
    if (relu2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
    }
    store_planes8(out + (long)out_chunk0 * M * 32 + e, out_stride, v);
  }
}

__global__ __launch_bounds__(256) void cm_norm_bwd_apply_fused_kernel(const float* __restrict__ x, const float* __restrict__ G,
                                                                      const __bf16* __restrict__ outmask, int mask_chunk0,
                                                                      const float* __restrict__ stats, const double* __restrict__ partial,
                                                                      int S, float* __restrict__ sums, __bf16* __restrict__ gz,
                                                                      long gz_stride, int gz_chunk0, long HW, int n, int chunks,
                                                                      int relu1) {
  __shared__ float st[32][2];
  const int slab = blockIdx.x, slabs = gridDim.x, img = blockIdx.y, ch = blockIdx.z;
  const int tid = threadIdx.x, q = tid & 3, ro = tid >> 2;
  const long M = (long)n * HW;
  const int C = chunks * 32;
  cm_stats_prologue<1>(partial, S, n, C, img, ch, (double)HW, 0.0, st, sums, slab == 0);
  const long per = (HW + slabs - 1) / slabs, r0 = slab * per, r1 = min(HW, r0 + per);
  float mean[8], rstd[8], s0[8], s1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const long sc = ((long)img * C + ch * 32 + q * 8 + j) * 2;
    mean[j] = stats[sc]; rstd[j] = stats[sc + 1];
    s0[j] = st[q * 8 + j][0]; s1[j] = st[q * 8 + j][1];
  }
  for (long r = r0 + ro; r < r1; r += 64) {
    const long e = ((long)ch * M + (long)img * HW + r) * 32 + q * 8;
    float xv[8], gv[8];
    load_f8(x + e, xv);
    load_f8(G + e, gv);
    if (outmask) {
      const bf16x8 mk = *reinterpret_cast<const bf16x8*>(outmask + (long)mask_chunk0 * M * 32 + e);
#pragma unroll
      for (int j = 0; j < 8; ++j) gv[j] = (float)mk[j] > 0.f ? gv[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (xv[j] - mean[j]) * rstd[j];
      float g = (relu1 && !(xh > 0.f)) ? 0.f : gv[j];
      g = g - s0[j] - xh * s1[j];
      gv[j] = rstd[j] * g;
    }
    store_planes8(gz + (long)gz_chunk0 * M * 32 + e, gz_stride, gv);
  }
}

int slabs_for(long HW, int n, int chunks) {          // ~1024 workgroups, >= 64 rows each
  long s = 1024 / ((long)n * chunks);
  if (s > HW / 64) s = HW / 64;
  return (int)(s < 1 ? 1 : s);
}

}  // namespace

extern "C" long ufr_cm_norm_workspace_doubles(long HW, int n, int chunks) { return 2L * 64 * n * chunks * 32; }

extern "C" int ufr_cm_norm_stats(const float* x, float* stats, double* workspace, long HW, int n, int chunks, float eps,
                                 ufr_stream_t stream) {
  UFR_REQUIRE(x && stats && workspace && HW > 0 && n > 0 && chunks > 0, "norm stats: bad argument");
  const int S = slices_for(HW, n, chunks);
  hipStream_t st = ufr::as_stream(stream);
  cm_sums_kernel<0><<<dim3(S, n, chunks), 256, 0, st>>>(x, nullptr, nullptr, 0, nullptr, workspace, HW, n, chunks, 0);
  int rc = ufr::launched("cm_sums_kernel<0>");
  if (rc != UFR_OK) return rc;
  const long nC = (long)n * chunks * 32;
  cm_sums_finalize<0><<<ufr::ceil_div(nC, 256), 256, 0, st>>>(workspace, stats, S, nC, (double)HW, (double)eps);
  return ufr::launched("cm_sums_finalize<0>");
}

extern "C" int ufr_cm_norm_stats_apply(const float* x, float* stats, double* workspace, float eps, const void* res,
                                       long res_plane_stride, int res_chunk0, void* out, long out_plane_stride, int out_chunk0, long HW,
                                       int n, int chunks, int relu1, int relu2, ufr_stream_t stream) {
  UFR_REQUIRE(x && stats && workspace && out && HW > 0 && n > 0 && chunks > 0, "norm stats + apply: bad argument");
  const int S = slices_for(HW, n, chunks);
  hipStream_t st = ufr::as_stream(stream);
  cm_sums_kernel<0><<<dim3(S, n, chunks), 256, 0, st>>>(x, nullptr, nullptr, 0, nullptr, workspace, HW, n, chunks, 0);
  int rc = ufr::launched("cm_sums_kernel<0>");
  if (rc != UFR_OK) return rc;
  cm_norm_apply_fused_kernel<<<dim3(slabs_for(HW, n, chunks), n, chunks), 256, 0, st>>>(
      x, workspace, S, (double)eps, stats, static_cast<const __bf16*>(res), res_plane_stride, res_chunk0, static_cast<__bf16*>(out),
      out_plane_stride, out_chunk0, HW, n, chunks, relu1, relu2);
  return ufr::launched("cm_norm_apply_fused_kernel");
}

extern "C" int ufr_cm_norm_apply(const float* x, const float* stats, const void* res, long res_plane_stride, int res_chunk0, void* out,
                                 long out_plane_stride, int out_chunk0, long HW, int n, int chunks, int relu1, int relu2,
                                 ufr_stream_t stream) {
  UFR_REQUIRE(x && out && HW > 0 && n > 0 && chunks > 0, "norm apply: bad argument");
  const long total = (long)chunks * n * HW * 4;
  cm_norm_apply_kernel<<<ufr::stream_grid(total, 256), 256, 0, ufr::as_stream(stream)>>>(
      x, stats, static_cast<const __bf16*>(res), res_plane_stride, res_chunk0, static_cast<__bf16*>(out), out_plane_stride, out_chunk0, HW, n,
      chunks, relu1, relu2);
  return ufr::launched("cm_norm_apply_kernel");
}

extern "C" int ufr_cm_norm_backward(const float* x, const float* G, const void* outmask, int mask_chunk0, const float* stats,
                                    float* sums, double* workspace, void* gz, long gz_plane_stride, int gz_chunk0, long HW, int n,
                                    int chunks, int relu1, ufr_stream_t stream) {
  UFR_REQUIRE(x && G && gz && HW > 0 && n > 0 && chunks > 0, "norm backward: bad argument");
  UFR_REQUIRE(!stats || (sums && workspace), "norm backward: the statistics form needs sums and a workspace");
  hipStream_t st = ufr::as_stream(stream);
  if (stats) {
    const int S = slices_for(HW, n, chunks);
    cm_sums_kernel<1><<<dim3(S, n, chunks), 256, 0, st>>>(x, G, static_cast<const __bf16*>(outmask), mask_chunk0, stats, workspace, HW, n,
                                                         chunks, relu1);
    int rc = ufr::launched("cm_sums_kernel<1>");
    if (rc != UFR_OK) return rc;
    cm_norm_bwd_apply_fused_kernel<<<dim3(slabs_for(HW, n, chunks), n, chunks), 256, 0, st>>>(
        x, G, static_cast<const __bf16*>(outmask), mask_chunk0, stats, workspace, S, sums, static_cast<__bf16*>(gz), gz_plane_stride,
        gz_chunk0, HW, n, chunks, relu1);
    return ufr::launched("cm_norm_bwd_apply_fused_kernel");
  }
  const long total = (long)chunks * n * HW * 4;
  cm_norm_bwd_apply_kernel<<<ufr::stream_grid(total, 256), 256, 0, st>>>(x, G, static_cast<const __bf16*>(outmask), mask_chunk0, nullptr,
                                                                        nullptr, static_cast<__bf16*>(gz), gz_plane_stride, gz_chunk0, HW,
                                                                        n, chunks, relu1);
  return ufr::launched("cm_norm_bwd_apply_kernel");
}

extern "C" int ufr_cm_masked_copy(const float* G, const void* outmask, long mask_elem_offset, float* out, long elems,
                                  ufr_stream_t stream) {
  UFR_REQUIRE(G && outmask && out && elems > 0 && elems % 8 == 0, "masked copy: bad argument");
  cm_masked_copy_kernel<<<ufr::stream_grid(elems / 8, 256), 256, 0, ufr::as_stream(stream)>>>(
      G, static_cast<const __bf16*>(outmask) + mask_elem_offset, 0, out, elems / 8);
  return ufr::launched("cm_masked_copy_kernel");
}
