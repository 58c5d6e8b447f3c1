// fp8 (e4m3) instantiations of the GEMM kernel template (gemm_bf16.hip): their own translation unit, see the note above gemm_fp8_launch there.
#define FP_GEMM_TU 2
#include "gemm_bf16.hip"
