// capi.hip -- error channel and introspection entry points of libufr_hip.so.
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>

#include "ufr_common.h"

namespace ufr {
char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
hipError_t ensure_dynamic_lds(const void* fn, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> raised;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = raised[{fn, dev}];
  if (bytes <= have) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) have = bytes;
  return e;
}
namespace {
std::map<std::string, std::string>& tu_table() {      // function-local: constructed before the first registrar runs
  static std::map<std::string, std::string> t;
  return t;
}
}  // namespace
void register_tu(const char* name, const char* sum) { tu_table()[name] = sum; }
}  // namespace ufr

// "<translation unit> <md5 of its .hip + ufr_common.h + include/ufr_hip.h at compile time>\n" per object, sorted by name
extern "C" const char* ufr_build_manifest(void) {
  static std::string text;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const auto& kv : ufr::tu_table()) text += kv.first + " " + kv.second + "\n";
  });
  return text.c_str();
}

extern "C" int ufr_abi_version(void) { return UFR_ABI_VERSION; }
extern "C" const char* ufr_last_error(void) { return ufr::err_buf(); }
extern "C" int ufr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
