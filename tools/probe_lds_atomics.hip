// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probe_lds_atomics.hip -o /tmp/lds_atomics && /tmp/lds_atomics
// Result of round 4: profiles/r4_lds_atomics_probe.txt (ds_add_f32 0.38 lanes per clock and CU, ds_add_rtn_u32 13, plain read-add-write 7).
// micro-benchmark: LDS atomic throughput per CU (conflict-free consecutive addresses, one lane per bank)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float f[4096];
  __shared__ int u[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) { f[i] = 0.f; u[i] = 0; }
  __syncthreads();
  int acc = 0;
  for (int it = 0; it < iters; ++it) {
    const int a = (threadIdx.x + 67 * it) & 4095;
    if (MODE == 0) atomicAdd(&f[a], 1.0f);                       // ds_add_f32
    if (MODE == 1) acc += atomicAdd(&u[a], 1);                    // ds_add_rtn_u32
    if (MODE == 2) atomicAdd(&u[a], 1);                           // ds_add_u32
    if (MODE == 3) f[a] += 1.0f;                                  // plain read-modify-write (racy, for the rate only)
    if (MODE == 4) acc += (int)atomicAdd(&f[a], 1.0f);            // ds_add_rtn_f32
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = f[5] + u[7] + acc;
}
template <int MODE>
void run(const char* name) {
  float* out; hipMalloc(&out, 4096 * 4);
  const int iters = 4096, blocks = 256 * 4;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  k<MODE><<<blocks, 256>>>(out, iters); hipDeviceSynchronize();
  hipEventRecord(s); k<MODE><<<blocks, 256>>>(out, iters); hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  const double lanes = (double)blocks * 256 * iters;
  printf("%-28s %8.3f ms  %7.1f G lane-ops/s  = %.2f lanes per clock and CU at 2.1 GHz\n", name, ms, lanes / ms / 1e6, lanes / ms / 1e6 / 256 / 2.1);
}
int main() {
  run<0>("ds_add_f32 (no return)"); run<4>("ds_add_rtn_f32"); run<2>("ds_add_u32 (no return)"); run<1>("ds_add_rtn_u32"); run<3>("plain LDS read-add-write");
  return 0;
}
