// ufr_common.h -- shared host-side helpers of libufr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/ufr_hip.h"

namespace ufr {

// Thread-local message returned by ufr_last_error().
char* err_buf();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(ufr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncAttributeMaxDynamicSharedMemorySize of `fn` raised to >= `bytes` on the CURRENT device: remembered per (kernel,
// device) under a mutex -- a process may drive several devices and several host threads (capi.hip).  hipSuccess when nothing
// had to be done.
hipError_t ensure_dynamic_lds(const void* fn, size_t bytes);

// Build manifest (capi.hip, `ufr_build_manifest()`): every translation unit registers, at load, the checksum of the sources it was
// COMPILED FROM (its .hip + this header + include/ufr_hip.h; the Makefile passes it as -DUFR_TU_SUM).  _lib.py recomputes the
// checksums from the tree and refuses a library that holds an object built from other sources.
void register_tu(const char* name, const char* sum);
struct TuRegistrar {
  TuRegistrar(const char* name, const char* sum) { register_tu(name, sum); }
};

// Post-launch check: launch-configuration errors surface here; nothing is synchronised.
inline int launched(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(UFR_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return UFR_OK;
}

constexpr int kWave = 64;           // CDNA4 wavefront
constexpr int kNumCU = 256;         // MI355X
constexpr int kMaxLds = 160 * 1024; // bytes of LDS per CU / per workgroup

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// Grid size for grid-stride elementwise kernels: enough workgroups to fill 256 CUs x 8.
inline int stream_grid(long n, int block) {
  long g = (n + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

// correlation_vec.hip: 0 = launched, 1 = shape not covered (use the general fast path), <0 = error
int corr_fwd_vec_launch(const float* in1, const float* in2, float* out, int B, int C, int H, int W, int P,
                        int DP, float scale, float slope, hipStream_t st);
int corr_bwd_vec_launch(const float* in1, const float* in2, const float* gout, float* gin1, float* gin2,
                        int B, int C, int H, int W, int P, int DP, hipStream_t st);

// correlation_mfma.hip: both adjoints on the fp32 matrix cores; same return convention
int corr_bwd_mfma_launch(const float* in1, const float* in2, const float* gout, float* gin1, float* gin2,
                         int B, int C, int H, int W, int P, int DP, hipStream_t st);

// attack.hip: *loss += sum of the n workgroup partials, in a fixed order (one wave)
void loss_finalize_launch(const float* partials, int n, float* loss, hipStream_t st);

}  // namespace ufr

#if defined(UFR_TU_NAME) && defined(UFR_TU_SUM)
namespace {
const ufr::TuRegistrar ufr_tu_registrar(UFR_TU_NAME, UFR_TU_SUM);
}
#endif

#define UFR_REQUIRE(cond, ...) \
  do {                         \
    if (!(cond)) return ufr::fail(UFR_EINVAL, __VA_ARGS__); \
  } while (0)
