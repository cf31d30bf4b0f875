// attention_api.cpp — the reference's PyTorch binding for the hot path, re-hosted on the C ABI.
//
// Builds the extension module `attention_cutlass` exporting
//     flash_attention_v2_cutlass(q, k, v, is_causal, softmax_scale) -> [out, softmax_lse]
// with the reference's names, positional-only arguments and return shapes
// (flash_attention_cutlass/csrc/attention_api.cpp:6-10, include/attention_api.h:10-11,
//  csrc/flash_attention.cu:741-772).  This file is host glue only: checks (CHECK_INPUT,
// include/attention_api.cuh:12-18), allocation of `out` / `softmax_lse` (flash_attention.cu:756-759)
// and one call of tfa_fwd_bhnd (include/tfa.h) on the current stream.  Unlike the reference it does
// not cudaDeviceSynchronize() and turns errors into exceptions instead of exit(1) (:767-769).
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <torch/extension.h>

#include <vector>

#include "tfa.h"

#define CHECK_CUDA(x) TORCH_CHECK(x.device().is_cuda(), #x " must be a CUDA tensor")
#define CHECK_CONTIGUOUS(x) TORCH_CHECK(x.is_contiguous(), #x " must be contiguous")
#define CHECK_INPUT(x) \
  CHECK_CUDA(x);       \
  CHECK_CONTIGUOUS(x)

std::vector<torch::Tensor> flash_attention_v2_cutlass(torch::Tensor q, torch::Tensor k, torch::Tensor v,
                                                      bool is_causal = false, float softmax_scale = 1) {
  CHECK_INPUT(q);
  CHECK_INPUT(k);
  CHECK_INPUT(v);
  TORCH_CHECK(q.dim() == 4 && k.sizes() == q.sizes() && v.sizes() == q.sizes(),
              "q, k, v must be (B, H, N, D) tensors of the same shape");
  TORCH_CHECK(q.scalar_type() == torch::kFloat16 || q.scalar_type() == torch::kBFloat16,
              "q, k, v must be float16 or bfloat16");
  TORCH_CHECK(k.scalar_type() == q.scalar_type() && v.scalar_type() == q.scalar_type(), "q, k, v must share a dtype");

  const int bs = q.size(0), head = q.size(1), seqlen = q.size(2), dim = q.size(3);
  const c10::DeviceGuard guard(q.device());   // torch-ROCm presents HIP devices as device type "cuda"
  auto out = torch::empty_like(q);
  auto softmax_lse = torch::empty({bs, head, seqlen}, q.options().dtype(torch::kFloat32));

  const int dtype = q.scalar_type() == torch::kBFloat16 ? TFA_BF16 : TFA_F16;
  const int st = tfa_fwd_bhnd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(),
                              softmax_lse.data_ptr<float>(), bs, head, seqlen, dim, softmax_scale,
                              is_causal ? 1 : 0, dtype, c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(q.device().index()).stream());
  TORCH_CHECK(st == 0, tfa_strerror(st));
  return {out, softmax_lse};
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("flash_attention_v2_cutlass", &flash_attention_v2_cutlass, "Flash attention v2 forward (MI355X HIP kernel)");
}
