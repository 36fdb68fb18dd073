// raft_update.hip -- the pieces of RAFT's update block (models/raft/update.py:35-162) that are not convolutions, on the
// engine's chunk-major layout (csrc/igemm.hip: activation planes bf16 [3][chunks][M][32], float32 tensors [chunks][M][32]):
//   flow_patches   convf1 = Conv2d(2, 128, 7, padding=3) (update.py:99) has 49 taps of 2 channels: its 7x7x2 neighbourhood is
//                  gathered once into 98 of 128 plane channels, and the convolution becomes a 1x1 igemm launch (K = 4 chunks)
//   motion_finish  cat([out 126, flow 2]) (update.py:120): the flow goes into channels 126 / 127 of the motion chunks, and the
//                  motion features are copied into the second half-step's GRU buffer
//   gates / blend  SepConvGRU's arithmetic (update.py:49-71) around the two gate convolutions of a half-step, forward and adjoint:
//                  z = sigmoid(zr[:Ch]); r = sigmoid(zr[Ch:]); rh = r * h      h' = (1 - z) * h + z * tanh(q)
//                  the sigmoid / tanh VALUES replace the pre-activations in place (the adjoint needs only them)
// All streaming (HBM / L2-bound), 8 channels per thread, 16- / 32-byte accesses.
#include "ufr_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ void split3(float v, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)v;
  const float r1 = v - (float)a;
  b = (__bf16)r1;
  c = (__bf16)(r1 - (float)b);
}

__device__ __forceinline__ void load_planes8(const __bf16* p, long plane_stride, float v[8]) {
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
  const bf16x8 b = *reinterpret_cast<const bf16x8*>(p + plane_stride);
  const bf16x8 c = *reinterpret_cast<const bf16x8*>(p + 2 * plane_stride);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = ((float)a[j] + (float)b[j]) + (float)c[j];
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

__device__ __forceinline__ void load_f8(const float* p, float v[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

__device__ __forceinline__ void store_f8(float* p, const float v[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// planes[chunk0 + k / 32][(b, y, x)][k % 32] = flow[b, k & 1, y + (k >> 1) / 7 - 3, x + (k >> 1) % 7 - 3], k < 98 (0 outside)
__global__ void flow_patches_kernel(const float* __restrict__ flow, __bf16* __restrict__ planes, long plane_stride, int chunk0,
                                    int B, int H, int W) {
  const long M = (long)B * H * W, total = M * 16;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int g8 = (int)(t & 15);
    const long m = t >> 4;
    const int x = (int)(m % W), y = (int)((m / W) % H), b = (int)(m / ((long)W * H));
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = g8 * 8 + j, tap = k >> 1, ch = k & 1, yy = y + tap / 7 - 3, xx = x + tap % 7 - 3;
      v[j] = (k < 98 && yy >= 0 && yy < H && xx >= 0 && xx < W) ? flow[(((long)b * 2 + ch) * H + yy) * W + xx] : 0.f;
    }
    store_planes8(planes + ((long)(chunk0 + (g8 >> 2)) * M + m) * 32 + (g8 & 3) * 8, plane_stride, v);
  }
}

// p2[chunk0 .. chunk0 + 3] = p1[chunk0 .. chunk0 + 3] with channels 126, 127 (lanes 30, 31 of the last chunk) = flow, in both
__global__ void motion_finish_kernel(__bf16* __restrict__ p1, long ps1, __bf16* __restrict__ p2, long ps2, int chunk0,
                                     const float* __restrict__ flow, long M, long HW) {
  const long total = M * 16;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int g8 = (int)(t & 15);
    const long m = t >> 4;
    const long e = ((long)(chunk0 + (g8 >> 2)) * M + m) * 32 + (g8 & 3) * 8;
    bf16x8 q0 = *reinterpret_cast<const bf16x8*>(p1 + e), q1 = *reinterpret_cast<const bf16x8*>(p1 + e + ps1),
           q2 = *reinterpret_cast<const bf16x8*>(p1 + e + 2 * ps1);
    if (g8 == 15) {
      const long b = m / HW, pix = m - b * HW;
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        __bf16 x, y, z;
        split3(flow[(b * 2 + o) * HW + pix], x, y, z);
        q0[6 + o] = x; q1[6 + o] = y; q2[6 + o] = z;
      }
      *reinterpret_cast<bf16x8*>(p1 + e) = q0;
      *reinterpret_cast<bf16x8*>(p1 + e + ps1) = q1;
      *reinterpret_cast<bf16x8*>(p1 + e + 2 * ps1) = q2;
    }
    *reinterpret_cast<bf16x8*>(p2 + e) = q0;
    *reinterpret_cast<bf16x8*>(p2 + e + ps2) = q1;
    *reinterpret_cast<bf16x8*>(p2 + e + 2 * ps2) = q2;
  }
}

// The pre-activation of 8 consecutive output channels of pixel m straight from a split-K launch's raw slabs [S][M][Npad] (ufr_igemm
// with `no_reduce`): the slabs are added in ascending order from zero, then the bias -- the arithmetic of igemm_reduce_kernel + its
// linear epilogue, bit for bit -- so the launch of that kernel between the convolution and the gate arithmetic disappears.
struct SlabSrc {
  const float* ws; const float* bias; long slab_stride; int S, Npad;
  const float* addend;     // optional [chunks][M][32] float32 in the OUTPUT tensor's layout, added behind the bias: the part of the
                           // pre-activation that is the same in every iteration (the context features' share, raft_engine.py)
};
// (addp: the 8 addend values of these channels or nullptr -- added between the slab sum and the bias, the order of igemm's own
// forward epilogue with an `add` tensor, so that the slab form and the launch + reduce form stay bit-identical)
__device__ __forceinline__ void load_slabs8(const SlabSrc& k, long m, int col, float v[8], const float* addp = nullptr) {
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
  const float* src = k.ws + m * k.Npad + col;
  int s = 0;
  if (k.S >= 5 && k.S <= 8) {                  // the usual splits of the 48 x 160 grids: every slab in flight at once (one round trip)
    float4 lo[8], hi[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int su = min(u, k.S - 1);          // (the loads past the last slice repeat it, their values are not added)
      lo[u] = *reinterpret_cast<const float4*>(src + su * k.slab_stride);
      hi[u] = *reinterpret_cast<const float4*>(src + su * k.slab_stride + 4);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (u < k.S) {
        v[0] += lo[u].x; v[1] += lo[u].y; v[2] += lo[u].z; v[3] += lo[u].w; v[4] += hi[u].x; v[5] += hi[u].y; v[6] += hi[u].z; v[7] += hi[u].w;
      }
    s = k.S;
  }
  for (; s + 4 <= k.S; s += 4) {               // four slabs in flight, added in the same ascending order
    float4 lo[4], hi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      lo[u] = *reinterpret_cast<const float4*>(src + (s + u) * k.slab_stride);
      hi[u] = *reinterpret_cast<const float4*>(src + (s + u) * k.slab_stride + 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[0] += lo[u].x; v[1] += lo[u].y; v[2] += lo[u].z; v[3] += lo[u].w; v[4] += hi[u].x; v[5] += hi[u].y; v[6] += hi[u].z; v[7] += hi[u].w;
    }
  }
  for (; s < k.S; ++s) {
    const float4 lo = *reinterpret_cast<const float4*>(src + s * k.slab_stride), hi = *reinterpret_cast<const float4*>(src + s * k.slab_stride + 4);
    v[0] += lo.x; v[1] += lo.y; v[2] += lo.z; v[3] += lo.w; v[4] += hi.x; v[5] += hi.y; v[6] += hi.z; v[7] += hi.w;
  }
  if (addp) {
    float a[8];
    load_f8(addp, a);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += a[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] += k.bias[col + j];
}

// motion_finish_kernel reading the motion encoder's last convolution (126 outputs, ReLU) from its split-K slabs: p1 = p2 =
// split(relu(sum + bias)) with the flow in channels 126, 127 -- the reduce launch and the read-back of p1 disappear
__global__ void motion_finish_slabs_kernel(SlabSrc slabs, int N, float slope, __bf16* __restrict__ p1, long ps1, __bf16* __restrict__ p2,
                                           long ps2, int chunk0, const float* __restrict__ flow, long M, long HW) {
  const long total = M * 16;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int g8 = (int)(t & 15);
    const long m = t >> 4;
    const long e = ((long)(chunk0 + (g8 >> 2)) * M + m) * 32 + (g8 & 3) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    const float* src = slabs.ws + m * slabs.Npad + g8 * 8;
    for (int s = 0; s < slabs.S; ++s) {
      const float4 lo = *reinterpret_cast<const float4*>(src + s * slabs.slab_stride), hi = *reinterpret_cast<const float4*>(src + s * slabs.slab_stride + 4);
      v[0] += lo.x; v[1] += lo.y; v[2] += lo.z; v[3] += lo.w; v[4] += hi.x; v[5] += hi.y; v[6] += hi.z; v[7] += hi.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {                          // igemm's forward epilogue: bias, LeakyReLU(slope), zeros in the padding
      const int n = g8 * 8 + j;
      v[j] += n < N ? slabs.bias[n] : 0.f;
      v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
      if (n >= N) v[j] = 0.f;
    }
    if (g8 == 15) {
      const long b = m / HW, pix = m - b * HW;
      v[6] = flow[(b * 2 + 0) * HW + pix];
      v[7] = flow[(b * 2 + 1) * HW + pix];
    }
    store_planes8(p1 + e, ps1, v);
    store_planes8(p2 + e, ps2, v);
  }
}

// zr [2 chunks_h][M][32] float32 pre-activations -> sigmoid values in place; rh planes = r * h
__global__ void gates_fwd_kernel(float* __restrict__ zr, const __bf16* __restrict__ h, long hs, int h_chunk0,
                                 __bf16* __restrict__ rh, long rs, int rh_chunk0, long M, int chunks, SlabSrc slabs) {
  const long n8 = (long)chunks * M * 4;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n8; t += (long)gridDim.x * blockDim.x) {
    const long e = t * 8;                                 // element inside a [chunks][M][32] tensor
    float z[8], r[8], hv[8];
    if (slabs.ws) {                                       // columns [z | r] of the gate convolution
      const int ch = (int)(e / (M * 32)), col = ch * 32 + (int)(e & 31);
      const long m = (e >> 5) - (long)ch * M;
      load_slabs8(slabs, m, col, z, slabs.addend ? slabs.addend + e : nullptr);
      load_slabs8(slabs, m, chunks * 32 + col, r, slabs.addend ? slabs.addend + (long)chunks * M * 32 + e : nullptr);
    } else {
      load_f8(zr + e, z);
      load_f8(zr + (long)chunks * M * 32 + e, r);
    }
    load_planes8(h + (long)h_chunk0 * M * 32 + e, hs, hv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      z[j] = sigmoidf_(z[j]);
      r[j] = sigmoidf_(r[j]);
      hv[j] = r[j] * hv[j];
    }
    store_f8(zr + e, z);
    store_f8(zr + (long)chunks * M * 32 + e, r);
    store_planes8(rh + (long)rh_chunk0 * M * 32 + e, rs, hv);
  }
}

// q [chunks][M][32] pre-activation -> tanh in place; out planes = (1 - z) h + z q
__global__ void blend_fwd_kernel(float* __restrict__ q, const float* __restrict__ z, const __bf16* __restrict__ h, long hs,
                                 int h_chunk0, __bf16* __restrict__ out, long os, int out_chunk0, long M, int chunks, SlabSrc slabs) {
  const long n8 = (long)chunks * M * 4;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n8; t += (long)gridDim.x * blockDim.x) {
    const long e = t * 8;
    float qv[8], zv[8], hv[8];
    if (slabs.ws) {
      const int ch = (int)(e / (M * 32));
      load_slabs8(slabs, (e >> 5) - (long)ch * M, ch * 32 + (int)(e & 31), qv, slabs.addend ? slabs.addend + e : nullptr);
    } else {
      load_f8(q + e, qv);
    }
    load_f8(z + e, zv);
    load_planes8(h + (long)h_chunk0 * M * 32 + e, hs, hv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      qv[j] = tanhf(qv[j]);
      hv[j] = (1.0f - zv[j]) * hv[j] + zv[j] * qv[j];
    }
    store_f8(q + e, qv);
    store_planes8(out + (long)out_chunk0 * M * 32 + e, os, hv);
  }
}

// g_q_pre planes = g z (1 - q^2);  g_z = g q - g h;  g_h = g (1 - z)      (q = tanh value, z = sigmoid value)
__global__ void blend_bwd_kernel(const float* __restrict__ q, const float* __restrict__ z, const __bf16* __restrict__ h, long hs,
                                 int h_chunk0, const float* __restrict__ g, __bf16* __restrict__ gq, long gqs, int gq_chunk0,
                                 float* __restrict__ g_z, float* __restrict__ g_h, long M, int chunks, float* __restrict__ acc) {
  const long n8 = (long)chunks * M * 4;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n8; t += (long)gridDim.x * blockDim.x) {
    const long e = t * 8;
    float qv[8], zv[8], hv[8], gv[8], a[8], bz[8], bh[8];
    load_f8(q + e, qv);
    load_f8(z + e, zv);
    load_f8(g + e, gv);
    load_planes8(h + (long)h_chunk0 * M * 32 + e, hs, hv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a[j] = (gv[j] * zv[j]) * (1.0f - qv[j] * qv[j]);
      bz[j] = gv[j] * qv[j] - gv[j] * hv[j];
      bh[j] = gv[j] * (1.0f - zv[j]);
    }
    store_planes8(gq + (long)gq_chunk0 * M * 32 + e, gqs, a);
    store_f8(g_z + e, bz);
    store_f8(g_h + e, bh);
    if (acc) {                           // running sum of g_q_pre over the iterations (the context features' adjoint runs once, on it)
      float s[8];
      load_f8(acc + e, s);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += a[j];
      store_f8(acc + e, s);
    }
  }
}

// g_zr planes [2 chunks]: [g_z z (1 - z) | g_rh h r (1 - r)];  g_h += g_rh r   (zr holds the sigmoid values)
__global__ void gates_bwd_kernel(const float* __restrict__ zr, const __bf16* __restrict__ h, long hs, int h_chunk0,
                                 const float* __restrict__ g_z, float* __restrict__ g_rh, __bf16* __restrict__ gzr, long gs,
                                 int gzr_chunk0, float* __restrict__ g_h, long M, int chunks, int consume, float* __restrict__ acc) {
  const long n8 = (long)chunks * M * 4;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n8; t += (long)gridDim.x * blockDim.x) {
    const long e = t * 8;
    float zv[8], rv[8], hv[8], gz[8], grh[8], gh[8], a[8], b[8];
    load_f8(zr + e, zv);
    load_f8(zr + (long)chunks * M * 32 + e, rv);
    load_f8(g_z + e, gz);
    load_f8(g_rh + e, grh);
    load_f8(g_h + e, gh);
    load_planes8(h + (long)h_chunk0 * M * 32 + e, hs, hv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a[j] = gz[j] * ((1.0f - zv[j]) * zv[j]);
      b[j] = (grh[j] * hv[j]) * ((1.0f - rv[j]) * rv[j]);
      gh[j] += grh[j] * rv[j];
    }
    store_planes8(gzr + (long)gzr_chunk0 * M * 32 + e, gs, a);
    store_planes8(gzr + ((long)gzr_chunk0 + chunks) * M * 32 + e, gs, b);
    store_f8(g_h + e, gh);
    if (acc) {                           // running sums of [g_z_pre | g_r_pre] over the iterations
      float sa[8], sb[8];
      load_f8(acc + e, sa);
      load_f8(acc + (long)chunks * M * 32 + e, sb);
#pragma unroll
      for (int j = 0; j < 8; ++j) { sa[j] += a[j]; sb[j] += b[j]; }
      store_f8(acc + e, sa);
      store_f8(acc + (long)chunks * M * 32 + e, sb);
    }
    if (consume) {                       // g_rh sits in a running-sum buffer whose next writer ADDS: leave zeros behind
      const float zero[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      store_f8(g_rh + e, zero);
    }
  }
}

// coords1 (+)= delta; the copy the lookup's adjoint of the NEXT iteration reads; flow = coords1 - coords0 (raft.py:190-228: `coords1 =
// coords1 + delta_flow`, then `coords1.detach()` and `flow = coords1 - coords0` at the top of the next iteration): three torch kernels
// (add_, clone, sub) in one.  delta == nullptr: the loop's entry (coords1 as it stands).
__global__ void coords_step_kernel(float* __restrict__ coords1, const float* __restrict__ delta, const float* __restrict__ coords0,
                                   float* __restrict__ saved, float* __restrict__ flow, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float c = coords1[i];
    if (delta) {
      c += delta[i];
      coords1[i] = c;
    }
    saved[i] = c;
    flow[i] = c - coords0[i];
  }
}

// grad_finalize (plane_layout.hip) on a running sum that is CONSUMED: planes = split(g * ReLU'(mask)) and g := 0 for the next adder
__global__ void finalize_consume_kernel(float* __restrict__ g, const __bf16* __restrict__ mask, long ms, __bf16* __restrict__ out, long os,
                                        long n8, float slope) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n8; t += (long)gridDim.x * blockDim.x) {
    const long e = t * 8;
    float v[8];
    load_f8(g + e, v);
    const bf16x8 m = *reinterpret_cast<const bf16x8*>(mask + e);
    (void)ms;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ((float)m[j] > 0.f) ? v[j] : v[j] * slope;
    store_planes8(out + e, os, v);
    const float zero[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    store_f8(g + e, zero);
  }
}

}  // namespace

extern "C" int ufr_raft_coords_step(float* coords1, const float* delta, const float* coords0, float* saved, float* flow, long n,
                                    ufr_stream_t stream) {
  UFR_REQUIRE(coords1 && coords0 && saved && flow && n > 0, "raft coords step: bad argument");
  coords_step_kernel<<<ufr::stream_grid(n, 256), 256, 0, ufr::as_stream(stream)>>>(coords1, delta, coords0, saved, flow, n);
  return ufr::launched("coords_step_kernel");
}

extern "C" int ufr_grad_finalize_consume(float* g, const void* mask_plane0, void* out, long out_plane_stride, long M, int chunks, float slope,
                                         ufr_stream_t stream) {
  UFR_REQUIRE(g && mask_plane0 && out && M > 0 && chunks > 0 && out_plane_stride > 0, "grad finalize (consume): bad argument");
  finalize_consume_kernel<<<ufr::stream_grid((long)chunks * M * 4, 256), 256, 0, ufr::as_stream(stream)>>>(
      g, static_cast<const __bf16*>(mask_plane0), 0, static_cast<__bf16*>(out), out_plane_stride, (long)chunks * M * 4, slope);
  return ufr::launched("finalize_consume_kernel");
}

extern "C" int ufr_raft_flow_patches(const float* flow, void* planes, long plane_stride, int chunk0, int B, int H, int W,
                                     ufr_stream_t stream) {
  UFR_REQUIRE(flow && planes && B > 0 && H > 0 && W > 0 && chunk0 >= 0 && plane_stride > 0, "raft flow patches: bad argument");
  const long total = (long)B * H * W * 16;
  flow_patches_kernel<<<ufr::stream_grid(total, 256), 256, 0, ufr::as_stream(stream)>>>(flow, static_cast<__bf16*>(planes), plane_stride,
                                                                                       chunk0, B, H, W);
  return ufr::launched("flow_patches_kernel");
}

extern "C" int ufr_raft_motion_finish(void* p1, long plane_stride1, void* p2, long plane_stride2, int chunk0, const float* flow, int B,
                                      int H, int W, ufr_stream_t stream) {
  UFR_REQUIRE(p1 && p2 && flow && B > 0 && H > 0 && W > 0 && chunk0 >= 0, "raft motion finish: bad argument");
  const long M = (long)B * H * W;
  motion_finish_kernel<<<ufr::stream_grid(M * 16, 256), 256, 0, ufr::as_stream(stream)>>>(
      static_cast<__bf16*>(p1), plane_stride1, static_cast<__bf16*>(p2), plane_stride2, chunk0, flow, M, (long)H * W);
  return ufr::launched("motion_finish_kernel");
}

extern "C" int ufr_raft_motion_finish_slabs(const float* ws, int splitk, int Npad, int N, const float* bias, float slope, void* p1,
                                            long plane_stride1, void* p2, long plane_stride2, int chunk0, const float* flow, int B, int H,
                                            int W, ufr_stream_t stream) {
  UFR_REQUIRE(ws && bias && p1 && p2 && flow && B > 0 && H > 0 && W > 0 && chunk0 >= 0 && splitk >= 1 && Npad == 128 && N > 0 && N <= 126,
              "raft motion finish (slabs): bad argument");
  const long M = (long)B * H * W;
  motion_finish_slabs_kernel<<<ufr::stream_grid(M * 16, 256), 256, 0, ufr::as_stream(stream)>>>(
      SlabSrc{ws, bias, M * Npad, splitk, Npad, nullptr}, N, slope, static_cast<__bf16*>(p1), plane_stride1, static_cast<__bf16*>(p2), plane_stride2,
      chunk0, flow, M, (long)H * W);
  return ufr::launched("motion_finish_slabs_kernel");
}

extern "C" int ufr_gru_gates_cm_forward(float* zr, const void* h, long h_plane_stride, int h_chunk0, void* rh, long rh_plane_stride,
                                        int rh_chunk0, long M, int chunks, ufr_stream_t stream) {
  UFR_REQUIRE(zr && h && rh && M > 0 && chunks > 0, "gru gates (chunk-major) forward: bad argument");
  gates_fwd_kernel<<<ufr::stream_grid((long)chunks * M * 4, 256), 256, 0, ufr::as_stream(stream)>>>(
      zr, static_cast<const __bf16*>(h), h_plane_stride, h_chunk0, static_cast<__bf16*>(rh), rh_plane_stride, rh_chunk0, M, chunks,
      SlabSrc{nullptr, nullptr, 0, 0, 0, nullptr});
  return ufr::launched("gates_fwd_kernel");
}

extern "C" int ufr_gru_gates_cm_forward_slabs(const float* ws, int splitk, int Npad, const float* bias, const float* addend, float* zr, const void* h,
                                              long h_plane_stride, int h_chunk0, void* rh, long rh_plane_stride, int rh_chunk0, long M,
                                              int chunks, ufr_stream_t stream) {
  UFR_REQUIRE(ws && bias && zr && h && rh && M > 0 && chunks > 0 && splitk >= 1 && Npad >= 2 * chunks * 32 && Npad % 8 == 0,
              "gru gates (chunk-major, from slabs) forward: bad argument");
  gates_fwd_kernel<<<ufr::stream_grid((long)chunks * M * 4, 256), 256, 0, ufr::as_stream(stream)>>>(
      zr, static_cast<const __bf16*>(h), h_plane_stride, h_chunk0, static_cast<__bf16*>(rh), rh_plane_stride, rh_chunk0, M, chunks,
      SlabSrc{ws, bias, M * Npad, splitk, Npad, addend});
  return ufr::launched("gates_fwd_kernel (slabs)");
}

extern "C" int ufr_gru_blend_cm_forward(float* q, const float* z, const void* h, long h_plane_stride, int h_chunk0, void* out,
                                        long out_plane_stride, int out_chunk0, long M, int chunks, ufr_stream_t stream) {
  UFR_REQUIRE(q && z && h && out && M > 0 && chunks > 0, "gru blend (chunk-major) forward: bad argument");
  blend_fwd_kernel<<<ufr::stream_grid((long)chunks * M * 4, 256), 256, 0, ufr::as_stream(stream)>>>(
      q, z, static_cast<const __bf16*>(h), h_plane_stride, h_chunk0, static_cast<__bf16*>(out), out_plane_stride, out_chunk0, M, chunks,
      SlabSrc{nullptr, nullptr, 0, 0, 0, nullptr});
  return ufr::launched("blend_fwd_kernel");
}

extern "C" int ufr_gru_blend_cm_forward_slabs(const float* ws, int splitk, int Npad, const float* bias, const float* addend, float* q, const float* z,
                                              const void* h, long h_plane_stride, int h_chunk0, void* out, long out_plane_stride,
                                              int out_chunk0, long M, int chunks, ufr_stream_t stream) {
  UFR_REQUIRE(ws && bias && q && z && h && out && M > 0 && chunks > 0 && splitk >= 1 && Npad >= chunks * 32 && Npad % 8 == 0,
              "gru blend (chunk-major, from slabs) forward: bad argument");
  blend_fwd_kernel<<<ufr::stream_grid((long)chunks * M * 4, 256), 256, 0, ufr::as_stream(stream)>>>(
      q, z, static_cast<const __bf16*>(h), h_plane_stride, h_chunk0, static_cast<__bf16*>(out), out_plane_stride, out_chunk0, M, chunks,
      SlabSrc{ws, bias, M * Npad, splitk, Npad, addend});
  return ufr::launched("blend_fwd_kernel (slabs)");
}

extern "C" int ufr_gru_blend_cm_backward(const float* q, const float* z, const void* h, long h_plane_stride, int h_chunk0,
                                         const float* g, void* gq, long gq_plane_stride, int gq_chunk0, float* g_z, float* g_h, long M,
                                         int chunks, float* acc_gq, ufr_stream_t stream) {
  UFR_REQUIRE(q && z && h && g && gq && g_z && g_h && M > 0 && chunks > 0, "gru blend (chunk-major) backward: bad argument");
  blend_bwd_kernel<<<ufr::stream_grid((long)chunks * M * 4, 256), 256, 0, ufr::as_stream(stream)>>>(
      q, z, static_cast<const __bf16*>(h), h_plane_stride, h_chunk0, g, static_cast<__bf16*>(gq), gq_plane_stride, gq_chunk0, g_z, g_h, M,
      chunks, acc_gq);
  return ufr::launched("blend_bwd_kernel");
}

extern "C" int ufr_gru_gates_cm_backward(const float* zr, const void* h, long h_plane_stride, int h_chunk0, const float* g_z,
                                         float* g_rh, void* gzr, long gzr_plane_stride, int gzr_chunk0, float* g_h, long M,
                                         int chunks, int consume_g_rh, float* acc_gzr, ufr_stream_t stream) {
  UFR_REQUIRE(zr && h && g_z && g_rh && gzr && g_h && M > 0 && chunks > 0, "gru gates (chunk-major) backward: bad argument");
  gates_bwd_kernel<<<ufr::stream_grid((long)chunks * M * 4, 256), 256, 0, ufr::as_stream(stream)>>>(
      zr, static_cast<const __bf16*>(h), h_plane_stride, h_chunk0, g_z, g_rh, static_cast<__bf16*>(gzr), gzr_plane_stride, gzr_chunk0, g_h,
      M, chunks, consume_g_rh, acc_gzr);
  return ufr::launched("gates_bwd_kernel");
}
