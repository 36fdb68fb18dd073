// bias_act.hip -- convolution epilogue of the FlowNet family on gfx950: y = LeakyReLU(x + bias[c]) in
// place, and its adjoint.  The reference's `conv(...)` blocks (models/submodules.py:18-46, :75-82) are
// Conv2d(bias=True) + LeakyReLU(0.1); on ROCm the bias is a separate broadcast add after the MIOpen
// convolution and the activation a third pass.  One pass here: 16-byte accesses along the row when the
// plane size allows (every FlowNetC plane does), bias fetched once per 4 elements.  HBM-streaming.
// slope = 0 is the Conv2d + ReLU pair of RAFT's motion encoder and heads (models/raft/update.py:6-14, :83-102).
// Same two roundings as torch (add, then multiply by the slope): results are bit-identical.
#include "ufr_common.h"

namespace {

// LeakyReLU; slope 0 is ReLU and returns +0 for negative inputs like torch's clamp_min (v * 0 would be -0)
__device__ __forceinline__ float act(float v, float slope) { return v > 0.f ? v : (slope == 0.f ? 0.f : v * slope); }

__global__ void bias_leaky_fwd_vec4(float4* __restrict__ x, const float* __restrict__ bias, int C, long hw4,
                                    long total4, float slope) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const float b = bias[(i / hw4) % C];
    float4 v = x[i];
    v.x += b; v.y += b; v.z += b; v.w += b;
    v.x = act(v.x, slope); v.y = act(v.y, slope); v.z = act(v.z, slope); v.w = act(v.w, slope);
    x[i] = v;
  }
}

__global__ void bias_leaky_fwd(float* __restrict__ x, const float* __restrict__ bias, int C, long hw, long total,
                               float slope) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    x[i] = act(x[i] + bias[(i / hw) % C], slope);
  }
}

// adjoint from the OUTPUT (y > 0 <=> pre-activation > 0 for a positive slope), like torch's in-place form
__global__ void leaky_bwd_vec4(const float4* __restrict__ y, const float4* __restrict__ gy, float4* __restrict__ gx,
                               long total4, float slope) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const float4 o = y[i], g = gy[i];
    float4 r;
    r.x = o.x > 0.f ? g.x : g.x * slope; r.y = o.y > 0.f ? g.y : g.y * slope;
    r.z = o.z > 0.f ? g.z : g.z * slope; r.w = o.w > 0.f ? g.w : g.w * slope;
    gx[i] = r;
  }
}

__global__ void leaky_bwd(const float* __restrict__ y, const float* __restrict__ gy, float* __restrict__ gx,
                          long total, float slope) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    gx[i] = y[i] > 0.f ? gy[i] : gy[i] * slope;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int ufr_bias_leaky_forward(float* x, const float* bias, int B, int C, long HW, float slope,
                                      ufr_stream_t stream) {
  UFR_REQUIRE(x && bias, "bias leaky forward: null pointer");
  UFR_REQUIRE(B > 0 && C > 0 && HW > 0 && slope >= 0.f, "bias leaky forward: bad argument (negative slope)");
  const long total = (long)B * C * HW;
  hipStream_t st = ufr::as_stream(stream);
  if ((HW & 3) == 0 && aligned16(x))
    bias_leaky_fwd_vec4<<<ufr::stream_grid(total / 4, 256), 256, 0, st>>>(reinterpret_cast<float4*>(x), bias, C, HW / 4,
                                                                           total / 4, slope);
  else
    bias_leaky_fwd<<<ufr::stream_grid(total, 256), 256, 0, st>>>(x, bias, C, HW, total, slope);
  return ufr::launched("bias_leaky_fwd");
}

extern "C" int ufr_leaky_backward(const float* y, const float* grad_y, float* grad_x, long total, float slope,
                                  ufr_stream_t stream) {
  UFR_REQUIRE(y && grad_y && grad_x && total > 0 && slope >= 0.f, "leaky backward: bad argument");
  hipStream_t st = ufr::as_stream(stream);
  if ((total & 3) == 0 && aligned16(y) && aligned16(grad_y) && aligned16(grad_x))
    leaky_bwd_vec4<<<ufr::stream_grid(total / 4, 256), 256, 0, st>>>(reinterpret_cast<const float4*>(y),
                                                                      reinterpret_cast<const float4*>(grad_y),
                                                                      reinterpret_cast<float4*>(grad_x), total / 4, slope);
  else
    leaky_bwd<<<ufr::stream_grid(total, 256), 256, 0, st>>>(y, grad_y, grad_x, total, slope);
  return ufr::launched("leaky_bwd");
}
