// f16f8 (fp16 high halves + fp8 cross terms) instantiations of the GEMM kernel template (gemm_bf16.hip): their own translation unit.
#define FP_GEMM_TU 4
#include "gemm_bf16.hip"
