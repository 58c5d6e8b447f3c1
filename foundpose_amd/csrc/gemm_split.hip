// split-fp16 (f16x3 mode) instantiations of the GEMM kernel template (gemm_bf16.hip): their own translation unit, see the note above gemm_fp8_launch there.
#define FP_GEMM_TU 3
#include "gemm_bf16.hip"
