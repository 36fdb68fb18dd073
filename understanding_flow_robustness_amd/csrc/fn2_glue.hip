// fn2_glue.hip -- the elementwise glue between FlowNet2's sub-networks (models/flownet2_models.py:122-205) as a handful of
// streaming kernels instead of ~20 torch dispatches per stage:
//   flow_upscale4      upsample1 / upsample2 ... of `flow * div_flow` resp. `flow / div_flow` (:133-136, :147, :160, :176):
//                      x4 bilinear (torch's align_corners = False weights) or nearest, forward and (gather, no atomics) adjoint
//   stage_pack         cat(x, resampled, flow / div_flow, ChannelNorm(x[:, :3] - resampled)) (:138-145, :150-157) behind Resample2d
//   stage_unpack_grad  its adjoint up to Resample2d's inputs: d/d x[:, :3] complete, the gradient that enters Resample2d's adjoint
//   stage_finish_grad  d/d x[:, 3:] = cat's share + Resample2d's image gradient; d/d flow = Resample2d's + the cat's share / div_flow
//   fusion_pack / _unpack_grad / _finish_grad   the same for FlowNetFusion's 11-channel input (:183-205):
//                      cat(x[:, :3], flow_sd, flow_s2, |flow_sd|, |flow_s2|, |x1 - warp_sd(x2)|, |x1 - warp_s2(x2)|)
// Resample2d itself stays csrc/warp_norm.hip / resample2d_owner.hip (its arithmetic is the reference's, quirk for quirk); the
// ChannelNorm arithmetic here is channelnorm_fwd / channelnorm_bwd's (sum of squares in channel order, sqrtf; g x / (norm + 1e-9)
// with the division in double, channelnorm_kernel.cu:18-96).  One thread per pixel, every access a coalesced row of one plane.
#include "ufr_common.h"

namespace {

// torch upsample_bilinear2d, align_corners = False, scale_factor 4 (area_pixel_compute_source_index with scale 0.25)
__device__ __forceinline__ void bil4(int d, int n, int& i0, int& i1, float& l0, float& l1) {
  float s = 0.25f * ((float)d + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.0f - l1;
}

// torch divides a tensor by a Python scalar as a multiplication by the float reciprocal (div_true_kernel_cuda: `a * (1 / b)` when
// the divisor is a CPU scalar): `flow / self.div_flow` (flownet2_models.py:143, :176) is reproduced bit for bit that way
__device__ __forceinline__ float scaled(float v, float scale, int divide) { return divide ? v * (1.0f / scale) : v * scale; }

// out [B,2,4h,4w] = upsample(scaled(flow [B,2,h,w])): the scaling first, as `interpolate(flow * div_flow)` does
__global__ void flow_upscale4_fwd(const float* __restrict__ flow, float* __restrict__ out, int B, int h, int w, int bilinear,
                                  float scale, int divide) {
  const int H = 4 * h, W = 4 * w;
  const long total = (long)B * 2 * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int X = (int)(i % W), Y = (int)((i / W) % H);
    const long bc = i / ((long)W * H);
    const float* f = flow + bc * h * w;
    if (bilinear) {
      int y0, y1, x0, x1;
      float ly0, ly1, lx0, lx1;
      bil4(Y, h, y0, y1, ly0, ly1);
      bil4(X, w, x0, x1, lx0, lx1);
      const float v00 = scaled(f[y0 * w + x0], scale, divide), v01 = scaled(f[y0 * w + x1], scale, divide);
      const float v10 = scaled(f[y1 * w + x0], scale, divide), v11 = scaled(f[y1 * w + x1], scale, divide);
      out[i] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
    } else {
      out[i] = scaled(f[(Y >> 2) * w + (X >> 2)], scale, divide);
    }
  }
}

// g_flow [B,2,h,w] = scaled(sum over the fine pixels that read this cell of weight * g_out): a cell's <= 8 x 8 readers, fixed order
__global__ void flow_upscale4_bwd(const float* __restrict__ gout, float* __restrict__ gflow, int B, int h, int w, int bilinear,
                                  float scale, int divide) {
  const int H = 4 * h, W = 4 * w;
  const long total = (long)B * 2 * h * w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % w), y = (int)((i / w) % h);
    const long bc = i / ((long)w * h);
    const float* g = gout + bc * H * W;
    float acc = 0.f;
    if (bilinear) {
      const int Y0 = max(4 * y - 2, 0), Y1 = min(4 * y + 5, H - 1), X0 = max(4 * x - 2, 0), X1 = min(4 * x + 5, W - 1);
      for (int Y = Y0; Y <= Y1; ++Y) {
        int a0, a1;
        float la0, la1;
        bil4(Y, h, a0, a1, la0, la1);
        const float wy = (a0 == y ? la0 : 0.f) + (a1 == y ? la1 : 0.f);
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int X = X0; X <= X1; ++X) {
          int b0, b1;
          float lb0, lb1;
          bil4(X, w, b0, b1, lb0, lb1);
          const float wx = (b0 == x ? lb0 : 0.f) + (b1 == x ? lb1 : 0.f);
          row += wx * g[(long)Y * W + X];
        }
        acc += wy * row;
      }
    } else {
      for (int dy = 0; dy < 4; ++dy)
        for (int dx = 0; dx < 4; ++dx) acc += g[(long)(4 * y + dy) * W + 4 * x + dx];
    }
    gflow[i] = scaled(acc, scale, divide);
  }
}

__device__ __forceinline__ float norm3(float a, float b, float c) {
  float acc = 0.f;
  acc += a * a; acc += b * b; acc += c * c;                   // channel order, two roundings each (no contraction: Makefile)
  return sqrtf(acc);
}
__device__ __forceinline__ float norm2(float a, float b) {
  float acc = 0.f;
  acc += a * a; acc += b * b;
  return sqrtf(acc);
}
__device__ __forceinline__ float cn_grad(float g, float x, float norm) { return (float)((double)(g * x) / ((double)norm + 1e-9)); }

// out [B,12,HW] = cat(x [B,6], res [B,3], flow [B,2] / div, |x[:, :3] - res|)
__global__ void stage_pack(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ flow,
                           float* __restrict__ out, int B, long HW, float div) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, p = i - b * HW;
    const float* xb = x + b * 6 * HW + p;
    const float* rb = res + b * 3 * HW + p;
    const float* fb = flow + b * 2 * HW + p;
    float* ob = out + b * 12 * HW + p;
    float d[3];
#pragma unroll
    for (int c = 0; c < 6; ++c) ob[c * HW] = xb[c * HW];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float r = rb[c * HW];
      ob[(6 + c) * HW] = r;
      d[c] = xb[c * HW] - r;
    }
    ob[9 * HW] = fb[0] * (1.0f / div);                       // (torch's tensor / scalar: see `scaled`)
    ob[10 * HW] = fb[HW] * (1.0f / div);
    ob[11 * HW] = norm3(d[0], d[1], d[2]);
  }
}

// gx [B,6,HW]: channels 0-2 = g[0:3] + d|.|/d diff (complete), channels 3-5 untouched; gres [B,3,HW] = g[6:9] - d|.|/d diff
__global__ void stage_unpack_grad(const float* __restrict__ g, const float* __restrict__ packed, float* __restrict__ gx,
                                  float* __restrict__ gres, int B, long HW) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, p = i - b * HW;
    const float* gb = g + b * 12 * HW + p;
    const float* pb = packed + b * 12 * HW + p;
    const float gn = gb[11 * HW], norm = pb[11 * HW];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float gd = cn_grad(gn, pb[c * HW] - pb[(6 + c) * HW], norm);
      gx[b * 6 * HW + c * HW + p] = gb[c * HW] + gd;
      gres[b * 3 * HW + c * HW + p] = gb[(6 + c) * HW] - gd;
    }
  }
}

// gx[:, 3:6] = g[3:6] + gimg (Resample2d's image gradient); gflow = gflow_rs (Resample2d's) + g[9:11] / div
__global__ void stage_finish_grad(const float* __restrict__ g, const float* __restrict__ gimg, const float* __restrict__ gflow_rs,
                                  float* __restrict__ gx, float* __restrict__ gflow, int B, long HW, float div) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, p = i - b * HW;
    const float* gb = g + b * 12 * HW + p;
#pragma unroll
    for (int c = 0; c < 3; ++c) gx[b * 6 * HW + (3 + c) * HW + p] = gb[(3 + c) * HW] + gimg[b * 3 * HW + c * HW + p];
#pragma unroll
    for (int c = 0; c < 2; ++c) gflow[b * 2 * HW + c * HW + p] = gflow_rs[b * 2 * HW + c * HW + p] + gb[(9 + c) * HW] * (1.0f / div);
  }
}

// out [B,11,HW] = cat(x[:, :3], fsd, fs2, |fsd|, |fs2|, |x1 - res_sd|, |x1 - res_s2|)    (x [B,6,HW])
__global__ void fusion_pack(const float* __restrict__ x, const float* __restrict__ fsd, const float* __restrict__ fs2,
                            const float* __restrict__ rsd, const float* __restrict__ rs2, float* __restrict__ out, int B, long HW) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, p = i - b * HW;
    const float* xb = x + b * 6 * HW + p;
    float* ob = out + b * 11 * HW + p;
    const float x0 = xb[0], x1 = xb[HW], x2 = xb[2 * HW];
    const float a0 = fsd[b * 2 * HW + p], a1 = fsd[b * 2 * HW + HW + p], c0 = fs2[b * 2 * HW + p], c1 = fs2[b * 2 * HW + HW + p];
    ob[0] = x0; ob[HW] = x1; ob[2 * HW] = x2;
    ob[3 * HW] = a0; ob[4 * HW] = a1; ob[5 * HW] = c0; ob[6 * HW] = c1;
    ob[7 * HW] = norm2(a0, a1);
    ob[8 * HW] = norm2(c0, c1);
    const float* ra = rsd + b * 3 * HW + p;
    const float* rc = rs2 + b * 3 * HW + p;
    ob[9 * HW] = norm3(x0 - ra[0], x1 - ra[HW], x2 - ra[2 * HW]);
    ob[10 * HW] = norm3(x0 - rc[0], x1 - rc[HW], x2 - rc[2 * HW]);
  }
}

// gx[:, 0:3] complete; gres_sd / gres_s2 [B,3,HW] = what enters the two Resample2d adjoints; gf_sd / gf_s2 [B,2,HW] = the cat's
// and the norms' share of the flow gradients (Resample2d's share is added by fusion_finish_grad)
__global__ void fusion_unpack_grad(const float* __restrict__ g, const float* __restrict__ packed, const float* __restrict__ rsd,
                                   const float* __restrict__ rs2, float* __restrict__ gx, float* __restrict__ gres_sd,
                                   float* __restrict__ gres_s2, float* __restrict__ gf_sd, float* __restrict__ gf_s2, int B, long HW) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, p = i - b * HW;
    const float* gb = g + b * 11 * HW + p;
    const float* pb = packed + b * 11 * HW + p;
    const float gn_sd = gb[7 * HW], gn_s2 = gb[8 * HW], ge_sd = gb[9 * HW], ge_s2 = gb[10 * HW];
    const float n_sd = pb[7 * HW], n_s2 = pb[8 * HW], e_sd = pb[9 * HW], e_s2 = pb[10 * HW];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      gf_sd[b * 2 * HW + c * HW + p] = gb[(3 + c) * HW] + cn_grad(gn_sd, pb[(3 + c) * HW], n_sd);
      gf_s2[b * 2 * HW + c * HW + p] = gb[(5 + c) * HW] + cn_grad(gn_s2, pb[(5 + c) * HW], n_s2);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float xv = pb[c * HW];
      const float da = cn_grad(ge_sd, xv - rsd[b * 3 * HW + c * HW + p], e_sd), dc = cn_grad(ge_s2, xv - rs2[b * 3 * HW + c * HW + p], e_s2);
      gx[b * 6 * HW + c * HW + p] = (gb[c * HW] + da) + dc;
      gres_sd[b * 3 * HW + c * HW + p] = -da;
      gres_s2[b * 3 * HW + c * HW + p] = -dc;
    }
  }
}

// gx[:, 3:6] = gimg_sd + gimg_s2; gf_sd += gflow_rs_sd; gf_s2 += gflow_rs_s2
__global__ void fusion_finish_grad(const float* __restrict__ gimg_sd, const float* __restrict__ gimg_s2, const float* __restrict__ grs_sd,
                                   const float* __restrict__ grs_s2, float* __restrict__ gx, float* __restrict__ gf_sd,
                                   float* __restrict__ gf_s2, int B, long HW) {
  const long total = (long)B * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, p = i - b * HW;
#pragma unroll
    for (int c = 0; c < 3; ++c) gx[b * 6 * HW + (3 + c) * HW + p] = gimg_sd[b * 3 * HW + c * HW + p] + gimg_s2[b * 3 * HW + c * HW + p];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      gf_sd[b * 2 * HW + c * HW + p] += grs_sd[b * 2 * HW + c * HW + p];
      gf_s2[b * 2 * HW + c * HW + p] += grs_s2[b * 2 * HW + c * HW + p];
    }
  }
}

}  // namespace

#define FN2_GRID(n) ufr::stream_grid((n), 256), 256, 0, ufr::as_stream(stream)

extern "C" int ufr_flow_upscale4_forward(const float* flow, float* out, int B, int h, int w, int bilinear, float scale, int divide,
                                         ufr_stream_t stream) {
  UFR_REQUIRE(flow && out, "flow upscale x4 forward: null pointer");
  UFR_REQUIRE(B > 0 && h > 0 && w > 0 && (long)B * h * w < (1L << 26) && scale != 0.f, "flow upscale x4 forward: bad shape");
  flow_upscale4_fwd<<<FN2_GRID((long)B * 32 * h * w)>>>(flow, out, B, h, w, bilinear, scale, divide);
  return ufr::launched("flow_upscale4_fwd");
}

extern "C" int ufr_flow_upscale4_backward(const float* grad_out, float* grad_flow, int B, int h, int w, int bilinear, float scale,
                                          int divide, ufr_stream_t stream) {
  UFR_REQUIRE(grad_out && grad_flow, "flow upscale x4 backward: null pointer");
  UFR_REQUIRE(B > 0 && h > 0 && w > 0 && (long)B * h * w < (1L << 26) && scale != 0.f, "flow upscale x4 backward: bad shape");
  flow_upscale4_bwd<<<FN2_GRID((long)B * 2 * h * w)>>>(grad_out, grad_flow, B, h, w, bilinear, scale, divide);
  return ufr::launched("flow_upscale4_bwd");
}

extern "C" int ufr_fn2_stage_pack(const float* x, const float* resampled, const float* flow, float* out, int B, int H, int W, float div_flow,
                                  ufr_stream_t stream) {
  UFR_REQUIRE(x && resampled && flow && out, "FlowNet2 stage pack: null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && div_flow != 0.f, "FlowNet2 stage pack: bad shape");
  stage_pack<<<FN2_GRID((long)B * H * W)>>>(x, resampled, flow, out, B, (long)H * W, div_flow);
  return ufr::launched("fn2 stage_pack");
}

extern "C" int ufr_fn2_stage_unpack_grad(const float* grad_out, const float* packed, float* grad_x, float* grad_resampled, int B, int H,
                                         int W, ufr_stream_t stream) {
  UFR_REQUIRE(grad_out && packed && grad_x && grad_resampled, "FlowNet2 stage unpack: null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0, "FlowNet2 stage unpack: bad shape");
  stage_unpack_grad<<<FN2_GRID((long)B * H * W)>>>(grad_out, packed, grad_x, grad_resampled, B, (long)H * W);
  return ufr::launched("fn2 stage_unpack_grad");
}

extern "C" int ufr_fn2_stage_finish_grad(const float* grad_out, const float* grad_image, const float* grad_flow_rs, float* grad_x,
                                         float* grad_flow, int B, int H, int W, float div_flow, ufr_stream_t stream) {
  UFR_REQUIRE(grad_out && grad_image && grad_flow_rs && grad_x && grad_flow, "FlowNet2 stage finish: null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0 && div_flow != 0.f, "FlowNet2 stage finish: bad shape");
  stage_finish_grad<<<FN2_GRID((long)B * H * W)>>>(grad_out, grad_image, grad_flow_rs, grad_x, grad_flow, B, (long)H * W, div_flow);
  return ufr::launched("fn2 stage_finish_grad");
}

extern "C" int ufr_fn2_fusion_pack(const float* x, const float* flow_sd, const float* flow_s2, const float* res_sd, const float* res_s2,
                                   float* out, int B, int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(x && flow_sd && flow_s2 && res_sd && res_s2 && out, "FlowNet2 fusion pack: null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0, "FlowNet2 fusion pack: bad shape");
  fusion_pack<<<FN2_GRID((long)B * H * W)>>>(x, flow_sd, flow_s2, res_sd, res_s2, out, B, (long)H * W);
  return ufr::launched("fn2 fusion_pack");
}

extern "C" int ufr_fn2_fusion_unpack_grad(const float* grad_out, const float* packed, const float* res_sd, const float* res_s2,
                                          float* grad_x, float* grad_res_sd, float* grad_res_s2, float* grad_flow_sd,
                                          float* grad_flow_s2, int B, int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(grad_out && packed && res_sd && res_s2 && grad_x && grad_res_sd && grad_res_s2 && grad_flow_sd && grad_flow_s2,
              "FlowNet2 fusion unpack: null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0, "FlowNet2 fusion unpack: bad shape");
  fusion_unpack_grad<<<FN2_GRID((long)B * H * W)>>>(grad_out, packed, res_sd, res_s2, grad_x, grad_res_sd, grad_res_s2, grad_flow_sd,
                                                   grad_flow_s2, B, (long)H * W);
  return ufr::launched("fn2 fusion_unpack_grad");
}

extern "C" int ufr_fn2_fusion_finish_grad(const float* grad_image_sd, const float* grad_image_s2, const float* grad_flow_rs_sd,
                                          const float* grad_flow_rs_s2, float* grad_x, float* grad_flow_sd, float* grad_flow_s2, int B,
                                          int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(grad_image_sd && grad_image_s2 && grad_flow_rs_sd && grad_flow_rs_s2 && grad_x && grad_flow_sd && grad_flow_s2,
              "FlowNet2 fusion finish: null pointer");
  UFR_REQUIRE(B > 0 && H > 0 && W > 0, "FlowNet2 fusion finish: bad shape");
  fusion_finish_grad<<<FN2_GRID((long)B * H * W)>>>(grad_image_sd, grad_image_s2, grad_flow_rs_sd, grad_flow_rs_s2, grad_x, grad_flow_sd,
                                                   grad_flow_s2, B, (long)H * W);
  return ufr::launched("fn2 fusion_finish_grad");
}
