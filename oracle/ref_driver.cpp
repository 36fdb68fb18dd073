// ref_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A C-ABI door onto the REFERENCE's own CPU correlation
// (models/Pytorch-Correlation-extension/Correlation_Module/correlation.cpp:75-178).
// oracle/Makefile compiles that reference source *where it lies* under
// /root/reference together with this file into oracle/_ref/libufr_corr_ref.so;
// no reference source is copied into this repository.  The library is used to
//   (1) pin oracle/ufr_oracle.c (tests/test_oracle_cpu.py), and
//   (2) serve as the "reference" kind of cpu_baseline in bench.py.
#include <torch/extension.h>

#include <cstring>
#include <vector>

// Defined by the reference translation unit (correlation.cpp:75, :126).
torch::Tensor correlation_cpp_forward(torch::Tensor input1, torch::Tensor input2, int kH, int kW,
                                      int patchH, int patchW, int padH, int padW, int dilationH,
                                      int dilationW, int dilation_patchH, int dilation_patchW,
                                      int dH, int dW);
std::vector<torch::Tensor> correlation_cpp_backward(torch::Tensor input1, torch::Tensor input2,
                                                    torch::Tensor gradOutput, int kH, int kW,
                                                    int patchH, int patchW, int padH, int padW,
                                                    int dilationH, int dilationW,
                                                    int dilation_patchH, int dilation_patchW,
                                                    int dH, int dW);

namespace {
template <typename T>
torch::Tensor wrap(const T* p, std::vector<int64_t> shape) {
  return torch::from_blob(const_cast<T*>(p), shape,
                          torch::TensorOptions().dtype(c10::CppTypeToScalarType<T>::value));
}

template <typename T>
int fwd(const T* in1, const T* in2, T* out, int B, int C, int H, int W, const int* p) {
  auto o = correlation_cpp_forward(wrap(in1, {B, C, H, W}), wrap(in2, {B, C, H, W}), p[0], p[1],
                                   p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11]);
  o = o.contiguous();
  std::memcpy(out, o.template data_ptr<T>(), sizeof(T) * o.numel());
  return 0;
}

template <typename T>
int bwd(const T* in1, const T* in2, const T* gout, T* g1, T* g2, int B, int C, int H, int W,
        int oH, int oW, const int* p) {
  auto r = correlation_cpp_backward(wrap(in1, {B, C, H, W}), wrap(in2, {B, C, H, W}),
                                    wrap(gout, {B, p[2], p[3], oH, oW}), p[0], p[1], p[2], p[3],
                                    p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11]);
  auto a = r[0].contiguous(), b = r[1].contiguous();
  std::memcpy(g1, a.template data_ptr<T>(), sizeof(T) * a.numel());
  std::memcpy(g2, b.template data_ptr<T>(), sizeof(T) * b.numel());
  return 0;
}
}  // namespace

// params = {kH,kW,patchH,patchW,padH,padW,dilationH,dilationW,dilation_patchH,dilation_patchW,dH,dW}
extern "C" {
int ufr_ref_corr_forward_f32(const float* a, const float* b, float* o, int B, int C, int H, int W,
                             const int* params) { return fwd<float>(a, b, o, B, C, H, W, params); }
int ufr_ref_corr_forward_f64(const double* a, const double* b, double* o, int B, int C, int H,
                             int W, const int* params) { return fwd<double>(a, b, o, B, C, H, W, params); }
int ufr_ref_corr_backward_f32(const float* a, const float* b, const float* g, float* g1, float* g2,
                              int B, int C, int H, int W, int oH, int oW, const int* params) {
  return bwd<float>(a, b, g, g1, g2, B, C, H, W, oH, oW, params);
}
int ufr_ref_corr_backward_f64(const double* a, const double* b, const double* g, double* g1,
                              double* g2, int B, int C, int H, int W, int oH, int oW,
                              const int* params) {
  return bwd<double>(a, b, g, g1, g2, B, C, H, W, oH, oW, params);
}
}
