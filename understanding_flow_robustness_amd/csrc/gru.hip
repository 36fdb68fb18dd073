// gru.hip -- RAFT's SepConvGRU gate arithmetic (models/raft/update.py:35-73) as fused kernels, gfx950.
//
// One half-step of the reference (per 1x5 / 5x1 pass, 12 iterations x 2 passes per RAFT forward):
//     hx = cat([h, x]); z = sigmoid(convz(hx)); r = sigmoid(convr(hx))
//     q  = tanh(convq(cat([r*h, x])));  h' = (1 - z)*h + z*q
// costs 10 elementwise launches forward and ~14 backward on [B,128,48,160] tensors (3.9 MB each: pure
// launch latency).  Here the two gate convolutions run as ONE convolution with 2*Ch outputs and the
// arithmetic around the convolutions is two kernels forward, two backward:
//     gates : z = sigmoid(zr[:Ch]),  rh = sigmoid(zr[Ch:]) * h   (rh written straight into the
//             [r*h | x] concat buffer that feeds convq)
//     blend : h' = (1 - z)*h + z*tanh(q_pre)
// All tensors are [B,C,HW] fp32; streaming, 16 bytes per lane where HW allows.
#include "ufr_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void gru_gates_fwd(const float* __restrict__ zr_pre, const float* __restrict__ h,
                              float* __restrict__ z_out, float* __restrict__ rh_out, long n_per_b, long total,
                              long rh_bstride) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / n_per_b, e = i - b * n_per_b;          // n_per_b = Ch*HW
    const float zp = zr_pre[b * 2 * n_per_b + e], rp = zr_pre[b * 2 * n_per_b + n_per_b + e];
    z_out[i] = sigmoidf_(zp);
    rh_out[b * rh_bstride + e] = sigmoidf_(rp) * h[i];
  }
}

// g_zr[:, :Ch] = g_z * z(1-z);  g_zr[:, Ch:] = g_rh * h * r(1-r);  g_h = g_rh * r
__global__ void gru_gates_bwd(const float* __restrict__ zr_pre, const float* __restrict__ h,
                              const float* __restrict__ g_z, const float* __restrict__ g_rh,
                              float* __restrict__ g_zr, float* __restrict__ g_h, long n_per_b, long total,
                              long grh_bstride) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / n_per_b, e = i - b * n_per_b;
    const float z = sigmoidf_(zr_pre[b * 2 * n_per_b + e]);
    const float r = sigmoidf_(zr_pre[b * 2 * n_per_b + n_per_b + e]);
    const float grh = g_rh[b * grh_bstride + e], hv = h[i];
    g_zr[b * 2 * n_per_b + e] = g_z[i] * ((1.0f - z) * z);           // torch: grad * (1 - y) * y
    g_zr[b * 2 * n_per_b + n_per_b + e] = (grh * hv) * ((1.0f - r) * r);
    g_h[i] = grh * r;
  }
}

__global__ void gru_blend_fwd(const float* __restrict__ q_pre, const float* __restrict__ z,
                              const float* __restrict__ h, float* __restrict__ h_out, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float q = tanhf(q_pre[i]), zz = z[i];
    h_out[i] = (1.0f - zz) * h[i] + zz * q;
  }
}

// g_q_pre = g*z*(1-q^2);  g_z = g*(q - h);  g_h = g*(1-z)
__global__ void gru_blend_bwd(const float* __restrict__ q_pre, const float* __restrict__ z,
                              const float* __restrict__ h, const float* __restrict__ g,
                              float* __restrict__ g_qpre, float* __restrict__ g_z, float* __restrict__ g_h,
                              long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float q = tanhf(q_pre[i]), zz = z[i], gg = g[i];
    g_qpre[i] = (gg * zz) * (1.0f - q * q);
    g_z[i] = gg * q - gg * h[i];          // d/dz of (1-z)*h + z*q, in autograd's order: -g*h + g*q
    g_h[i] = gg * (1.0f - zz);
  }
}

}  // namespace

extern "C" int ufr_gru_gates_forward(const float* zr_pre, const float* h, float* z_out, float* rh_out, int B,
                                     int Ch, int HW, long rh_bstride, ufr_stream_t stream) {
  UFR_REQUIRE(zr_pre && h && z_out && rh_out && B > 0 && Ch > 0 && HW > 0 && rh_bstride >= (long)Ch * HW,
              "gru gates forward: bad argument");
  const long npb = (long)Ch * HW, total = npb * B;
  hipLaunchKernelGGL(gru_gates_fwd, dim3(ufr::stream_grid(total, 256)), dim3(256), 0, ufr::as_stream(stream), zr_pre,
                     h, z_out, rh_out, npb, total, rh_bstride);
  return ufr::launched("gru_gates_fwd");
}

extern "C" int ufr_gru_gates_backward(const float* zr_pre, const float* h, const float* g_z, const float* g_rh,
                                      float* g_zr, float* g_h, int B, int Ch, int HW, long grh_bstride,
                                      ufr_stream_t stream) {
  UFR_REQUIRE(zr_pre && h && g_z && g_rh && g_zr && g_h && B > 0 && Ch > 0 && HW > 0 && grh_bstride >= (long)Ch * HW,
              "gru gates backward: bad argument");
  const long npb = (long)Ch * HW, total = npb * B;
  hipLaunchKernelGGL(gru_gates_bwd, dim3(ufr::stream_grid(total, 256)), dim3(256), 0, ufr::as_stream(stream), zr_pre,
                     h, g_z, g_rh, g_zr, g_h, npb, total, grh_bstride);
  return ufr::launched("gru_gates_bwd");
}

extern "C" int ufr_gru_blend_forward(const float* q_pre, const float* z, const float* h, float* h_out, long total,
                                     ufr_stream_t stream) {
  UFR_REQUIRE(q_pre && z && h && h_out && total > 0, "gru blend forward: bad argument");
  hipLaunchKernelGGL(gru_blend_fwd, dim3(ufr::stream_grid(total, 256)), dim3(256), 0, ufr::as_stream(stream), q_pre, z,
                     h, h_out, total);
  return ufr::launched("gru_blend_fwd");
}

extern "C" int ufr_gru_blend_backward(const float* q_pre, const float* z, const float* h, const float* g,
                                      float* g_qpre, float* g_z, float* g_h, long total, ufr_stream_t stream) {
  UFR_REQUIRE(q_pre && z && h && g && g_qpre && g_z && g_h && total > 0, "gru blend backward: bad argument");
  hipLaunchKernelGGL(gru_blend_bwd, dim3(ufr::stream_grid(total, 256)), dim3(256), 0, ufr::as_stream(stream), q_pre, z,
                     h, g, g_qpre, g_z, g_h, total);
  return ufr::launched("gru_blend_bwd");
}
