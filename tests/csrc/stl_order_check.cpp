// Test harness: runs foundpose_amd/csrc/stl_order.hpp (host build) so Python can compare it with the real
// std:: algorithms (oracle/csrc/oracle.cpp::orc_topk_torch) and with torch.topk.
#include <cstdint>
#include <vector>

#include "../../foundpose_amd/csrc/stl_order.hpp"

extern "C" void stl_order_topk(const float* values, int64_t n, int64_t k, int64_t* out_idx) {
  std::vector<stl_order::Elem> a(n);
  for (int64_t j = 0; j < n; ++j) a[j] = {values[j], (int)j};
  stl_order::topk_torch_largest(a.data(), (int)n, (int)k);
  for (int64_t j = 0; j < k; ++j) out_idx[j] = a[j].idx;
}
