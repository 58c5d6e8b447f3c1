// Translation unit 5 of the GEMM template: IEEE fp16 operands (the "f16" mode) -- see the end of gemm_bf16.hip.
#define FP_GEMM_TU 5
#include "gemm_bf16.hip"
