// Issue cost of the VALU instructions the attention kernels' softmax is made of, on gfx950: cycles per instruction for a wave that
// issues 32 INDEPENDENT copies per loop iteration (throughput), and for a chain of 32 DEPENDENT copies (latency), with one or two
// waves per SIMD.  s_memtime ticks at a constant 100 MHz, so cycles are derived from the event time and reported as ns per instruction
// and as "slots" relative to v_add_f32 in the same run.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int OP, bool DEP>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  float a = threadIdx.x * 1e-3f + 0.5f, b = 1.0001f, c = 0.25f;
  float d0 = a, d1 = a + 1, d2 = a + 2, d3 = a + 3, d4 = a + 4, d5 = a + 5, d6 = a + 6, d7 = a + 7;
  for (int i = 0; i < iters; ++i) {
#define INDEP(INS, ...)                                                                                             \
    REP8(asm volatile(INS "\n" INS "\n" INS "\n" INS : __VA_ARGS__);)
    // one asm statement = 4 instructions; REP8 -> 32 per iteration.  Independent: four different destinations cycling; dependent: one.
    if constexpr (OP == 0) {  // v_add_f32
      if constexpr (DEP) { REP32(asm volatile("v_add_f32 %0, %0, %1" : "+v"(d0) : "v"(b));) }
      else { REP8(asm volatile("v_add_f32 %0, %4, %5\nv_add_f32 %1, %4, %5\nv_add_f32 %2, %4, %5\nv_add_f32 %3, %4, %5" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a), "v"(b));) }
    } else if constexpr (OP == 1) {  // v_fma_f32
      if constexpr (DEP) { REP32(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(d0) : "v"(b), "v"(c));) }
      else { REP8(asm volatile("v_fma_f32 %0, %4, %5, %6\nv_fma_f32 %1, %4, %5, %6\nv_fma_f32 %2, %4, %5, %6\nv_fma_f32 %3, %4, %5, %6" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a), "v"(b), "v"(c));) }
    } else if constexpr (OP == 2) {  // v_exp_f32
      if constexpr (DEP) { REP32(asm volatile("v_exp_f32 %0, %0" : "+v"(d0));) }
      else { REP8(asm volatile("v_exp_f32 %0, %4\nv_exp_f32 %1, %4\nv_exp_f32 %2, %4\nv_exp_f32 %3, %4" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a));) }
    } else if constexpr (OP == 3) {  // v_cvt_pk_f16_f32 (RNE pack, gfx950)
      if constexpr (DEP) { REP32(asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(d0) : "v"(b));) }
      else { REP8(asm volatile("v_cvt_pk_f16_f32 %0, %4, %5\nv_cvt_pk_f16_f32 %1, %4, %5\nv_cvt_pk_f16_f32 %2, %4, %5\nv_cvt_pk_f16_f32 %3, %4, %5" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a), "v"(b));) }
    } else if constexpr (OP == 4) {  // v_cvt_f32_f16
      if constexpr (DEP) { REP32(asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(d0));) }
      else { REP8(asm volatile("v_cvt_f32_f16 %0, %4\nv_cvt_f32_f16 %1, %4\nv_cvt_f32_f16 %2, %4\nv_cvt_f32_f16 %3, %4" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a));) }
    } else if constexpr (OP == 5) {  // v_cvt_f32_f16 sdwa (high half)
      if constexpr (DEP) { REP32(asm volatile("v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(d0));) }
      else { REP8(asm volatile("v_cvt_f32_f16_sdwa %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_f16_sdwa %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_f16_sdwa %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\nv_cvt_f32_f16_sdwa %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a));) }
    } else if constexpr (OP == 6) {  // v_pk_mul_f32 (two fp32 per lane)
      double p0 = __builtin_bit_cast(double, make_float2(d0, d1)), p1 = __builtin_bit_cast(double, make_float2(d2, d3)), p2 = __builtin_bit_cast(double, make_float2(d4, d5)), p3 = __builtin_bit_cast(double, make_float2(d6, d7));
      const double pa = __builtin_bit_cast(double, make_float2(a, b)), pb = __builtin_bit_cast(double, make_float2(b, b));
      if constexpr (DEP) { REP32(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p0) : "v"(pb));) }
      else { REP8(asm volatile("v_pk_mul_f32 %0, %4, %5\nv_pk_mul_f32 %1, %4, %5\nv_pk_mul_f32 %2, %4, %5\nv_pk_mul_f32 %3, %4, %5" : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(pa), "v"(pb));) }
      const float2 f0 = __builtin_bit_cast(float2, p0), f1 = __builtin_bit_cast(float2, p1), f2 = __builtin_bit_cast(float2, p2), f3 = __builtin_bit_cast(float2, p3);
      d0 = f0.x; d1 = f0.y; d2 = f1.x; d3 = f1.y; d4 = f2.x; d5 = f2.y; d6 = f3.x; d7 = f3.y;
    } else if constexpr (OP == 7) {  // v_pk_add_f32
      double p0 = __builtin_bit_cast(double, make_float2(d0, d1)), p1 = __builtin_bit_cast(double, make_float2(d2, d3)), p2 = __builtin_bit_cast(double, make_float2(d4, d5)), p3 = __builtin_bit_cast(double, make_float2(d6, d7));
      const double pa = __builtin_bit_cast(double, make_float2(a, b)), pb = __builtin_bit_cast(double, make_float2(b, b));
      if constexpr (DEP) { REP32(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(pb));) }
      else { REP8(asm volatile("v_pk_add_f32 %0, %4, %5\nv_pk_add_f32 %1, %4, %5\nv_pk_add_f32 %2, %4, %5\nv_pk_add_f32 %3, %4, %5" : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(pa), "v"(pb));) }
      const float2 f0 = __builtin_bit_cast(float2, p0), f1 = __builtin_bit_cast(float2, p1), f2 = __builtin_bit_cast(float2, p2), f3 = __builtin_bit_cast(float2, p3);
      d0 = f0.x; d1 = f0.y; d2 = f1.x; d3 = f1.y; d4 = f2.x; d5 = f2.y; d6 = f3.x; d7 = f3.y;
    } else if constexpr (OP == 8) {  // v_max3_f32
      if constexpr (DEP) { REP32(asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(d0) : "v"(b), "v"(c));) }
      else { REP8(asm volatile("v_max3_f32 %0, %4, %5, %6\nv_max3_f32 %1, %4, %5, %6\nv_max3_f32 %2, %4, %5, %6\nv_max3_f32 %3, %4, %5, %6" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a), "v"(b), "v"(c));) }
    } else if constexpr (OP == 9) {  // v_permlane32_swap_b32 (swaps two registers' halves in place: both are destinations)
      if constexpr (DEP) { REP32(asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(d0), "+v"(d1));) }
      else { REP8(asm volatile("v_permlane32_swap_b32 %0, %1\nv_permlane32_swap_b32 %2, %3\nv_permlane32_swap_b32 %4, %5\nv_permlane32_swap_b32 %6, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));) }
    } else if constexpr (OP == 10) {  // v_cvt_pkrtz_f16_f32
      if constexpr (DEP) { REP32(asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(d0) : "v"(b));) }
      else { REP8(asm volatile("v_cvt_pkrtz_f16_f32 %0, %4, %5\nv_cvt_pkrtz_f16_f32 %1, %4, %5\nv_cvt_pkrtz_f16_f32 %2, %4, %5\nv_cvt_pkrtz_f16_f32 %3, %4, %5" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a), "v"(b));) }
    } else if constexpr (OP == 11) {  // v_pk_fma_f32
      double p0 = __builtin_bit_cast(double, make_float2(d0, d1)), p1 = __builtin_bit_cast(double, make_float2(d2, d3)), p2 = __builtin_bit_cast(double, make_float2(d4, d5)), p3 = __builtin_bit_cast(double, make_float2(d6, d7));
      const double pa = __builtin_bit_cast(double, make_float2(a, b)), pb = __builtin_bit_cast(double, make_float2(b, b));
      if constexpr (DEP) { REP32(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(pb));) }
      else { REP8(asm volatile("v_pk_fma_f32 %0, %4, %5, %5\nv_pk_fma_f32 %1, %4, %5, %5\nv_pk_fma_f32 %2, %4, %5, %5\nv_pk_fma_f32 %3, %4, %5, %5" : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(pa), "v"(pb));) }
      const float2 f0 = __builtin_bit_cast(float2, p0), f1 = __builtin_bit_cast(float2, p1), f2 = __builtin_bit_cast(float2, p2), f3 = __builtin_bit_cast(float2, p3);
      d0 = f0.x; d1 = f0.y; d2 = f1.x; d3 = f1.y; d4 = f2.x; d5 = f2.y; d6 = f3.x; d7 = f3.y;
    } else if constexpr (OP == 12) {  // v_exp_f32 fed by a v_fma_f32 (the softmax's pair), independent pairs
      REP8(asm volatile("v_fma_f32 %0, %4, %5, %6\nv_exp_f32 %0, %0\nv_fma_f32 %1, %4, %5, %6\nv_exp_f32 %1, %1\nv_fma_f32 %2, %4, %5, %6\nv_exp_f32 %2, %2\nv_fma_f32 %3, %4, %5, %6\nv_exp_f32 %3, %3" : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(a), "v"(b), "v"(c));)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
}

template <int OP, bool DEP>
static double run(int threads, float* out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OP, DEP>), dim3(256), dim3(threads), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP, DEP>), dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const int per_iter = OP == 12 ? 64 : 32;
  return ms * 1e6 / ((double)iters * per_iter);  // ns per instruction per wave (the waves of a SIMD share it: 2 waves/SIMD -> ns per 2 instructions)
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  const char* names[] = {"v_add_f32", "v_fma_f32", "v_exp_f32", "v_cvt_pk_f16_f32", "v_cvt_f32_f16", "v_cvt_f32_f16_sdwa", "v_pk_mul_f32", "v_pk_add_f32", "v_max3_f32",
                         "v_permlane32_swap", "v_cvt_pkrtz_f16_f32", "v_pk_fma_f32", "fma+exp pair (per instr)"};
  printf("%-26s %12s %12s %12s %12s   (ns per instruction of one wave; 256 = 1 wave/SIMD, 512 = 2 waves/SIMD)\n", "instruction", "indep 256", "dep 256", "indep 512", "dep 512");
#define ROW(OP)                                                                                                                               \
  printf("%-26s %12.3f %12.3f %12.3f %12.3f\n", names[OP], run<OP, false>(256, out, iters), run<OP, true>(256, out, iters), run<OP, false>(512, out, iters), \
         run<OP, true>(512, out, iters));
  ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12)
  return 0;
}
