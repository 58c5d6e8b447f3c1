// Do a matrix-only wave and a VALU-only wave on the SAME SIMD overlap?  512-thread workgroups (waves w and w + 4 share SIMD w): waves 0-3 run
// a stream of v_mfma_f32_32x32x16_f16 (two independent accumulators), waves 4-7 a stream of plain VALU (v_fma_f32 / v_exp_f32 mix, independent
// registers).  Each role's loop is timed alone (the other half exits) and together; s_memtime ticks per role + wall time.
//   mode 0: MFMA only   1: VALU only   2: both   (accumulators in VGPRs or AGPRs: build with/without -mllvm -amdgpu-mfma-vgpr-form)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_corun.hip -o tools/ubench/mfma_valu_corun
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
template <int VALU_KIND, int MFMA_KIND = 0>
__global__ __launch_bounds__(512, 2) void k(float* out, long long* ticks, int iters, int mode, int swap, int prio) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool mfma_role = swap ? wave >= 4 : wave < 4;
  if (prio == 1 && !mfma_role) __builtin_amdgcn_s_setprio(3);   // the VALU half prioritized
  if (prio == 2 && mfma_role) __builtin_amdgcn_s_setprio(3);    // the matrix half prioritized
  if ((mode == 0 && !mfma_role) || (mode == 1 && mfma_role)) return;
  const long long t0 = clock64();
  if (mfma_role) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) a[e] = (_Float16)(0.01f * (lane + e)), b[e] = (_Float16)(0.02f * (lane - e));
    f32x16 c0, c1;
    for (int r = 0; r < 16; ++r) c0[r] = 0.f, c1[r] = 0.f;
    __shared__ __attribute__((aligned(16))) char L[32768];
    if constexpr (MFMA_KIND == 1) {
      for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<float*>(L)[i] = 1e-3f * (float)(i & 255);
    }
    const int row = lane & 31, khh = lane >> 5;
    for (int i = 0; i < iters; ++i) {
      if constexpr (MFMA_KIND == 0) {
#pragma unroll
        for (int u = 0; u < 24; ++u) {
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
        }
      } else if constexpr (MFMA_KIND >= 2) {  // MFMA_KIND - 1 times `s_nop 7` behind every MFMA: the wave does not ask for the issue port while its MFMA runs
#pragma unroll
        for (int u = 0; u < 24; ++u) {
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int z = 0; z < MFMA_KIND - 1; ++z) asm volatile("s_nop 7");
          __builtin_amdgcn_sched_barrier(0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int z = 0; z < MFMA_KIND - 1; ++z) asm volatile("s_nop 7");
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {  // every pair of MFMAs takes a fresh b128 fragment from LDS (one step ahead), like the attention's matrix phase
        f16x8 f = *reinterpret_cast<const f16x8*>(L + (wave & 3) * 8192 + row * 256 + ((khh ^ (row & 15)) << 4));
#pragma unroll
        for (int u = 0; u < 24; ++u) {
          const f16x8 nf = *reinterpret_cast<const f16x8*>(L + (wave & 3) * 8192 + row * 256 + (((((u + 1) & 7) * 2 + khh) ^ (row & 15)) << 4));
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f, b, c0, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, f, c1, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          f = nf;
        }
      }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    float a = lane * 1e-3f + 0.5f, b = 1.0001f, c = 0.25f, d0 = a, d1 = a + 1, d2 = a + 2, d3 = a + 3;
    float wide[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) wide[j] = a * (float)(j + 1) * 1e-2f;
    for (int i = 0; i < iters; ++i) {
      if constexpr (VALU_KIND == 0) {  // 64 v_fma_f32, independent
#pragma unroll
        for (int u = 0; u < 16; ++u)
          asm volatile("v_fma_f32 %0, %4, %5, %6\nv_fma_f32 %1, %4, %5, %6\nv_fma_f32 %2, %4, %5, %6\nv_fma_f32 %3, %4, %5, %6" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a), "v"(b), "v"(c));
      } else if constexpr (VALU_KIND == 1) {  // 32 (v_fma_f32, v_exp_f32) pairs
#pragma unroll
        for (int u = 0; u < 8; ++u)
          asm volatile("v_fma_f32 %0, %4, %5, %6\nv_exp_f32 %0, %0\nv_fma_f32 %1, %4, %5, %6\nv_exp_f32 %1, %1\nv_fma_f32 %2, %4, %5, %6\nv_exp_f32 %2, %2\nv_fma_f32 %3, %4, %5, %6\nv_exp_f32 %3, %3" : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(a), "v"(b), "v"(c));
      } else if constexpr (VALU_KIND == 2) {  // 64 v_fma_f32 over 64 distinct registers (r[j] = r[j] * b + r[j ^ 1])
        static_assert(true, "");
#pragma unroll
        for (int j = 0; j < 64; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(wide[j]) : "v"(b), "v"(wide[j ^ 1]));
      } else {                                // the split attention's softmax, compiled C (32 scores -> max, exp, row sum, split-fp16 packing)
        float mx = wide[0];
#pragma unroll
        for (int j = 1; j < 32; ++j) mx = fmaxf(mx, wide[j]);
        const float mc = mx * c;
        float ps = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          const float pj = __builtin_amdgcn_exp2f(fmaf(wide[j], c, -mc));
          wide[32 + j] = pj;
          ps += pj;
        }
#pragma unroll
        for (int j = 0; j < 32; j += 2) {
          const f16x2 h = __builtin_convertvector(f32x2{wide[32 + j], wide[33 + j]}, f16x2);
          const f32x2 hf = __builtin_convertvector(h, f32x2);
          const f16x2 l = __builtin_convertvector(f32x2{wide[32 + j] - hf[0], wide[33 + j] - hf[1]}, f16x2);
          wide[j] = __builtin_bit_cast(float, h) * 1e-3f + wide[j] * 0.5f;
          wide[j + 1] = __builtin_bit_cast(float, l) * 1e-3f + wide[j + 1] * 0.5f + ps * 1e-9f;
        }
      }
    }
    float sw = 0.f;
#pragma unroll
    for (int j = 0; j < 64; ++j) sw += wide[j];
    out[blockIdx.x * 512 + threadIdx.x] = d0 + d1 + d2 + d3 + sw;
  }
  const long long t1 = clock64();
  if (blockIdx.x == 100 && lane == 0 && (wave == 0 || wave == 4)) ticks[mfma_role ? 0 : 1] = t1 - t0;
}

template <int VALU_KIND, int MFMA_KIND = 0>
static void run(const char* what, float* out, long long* ticks, int iters) {
  printf("%s: 48 MFMAs per iteration (waves 0-3), 64 VALU per iteration (waves 4-7), %d iterations\n", what, iters);
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int mode = cfg < 2 ? cfg : 2, swap = cfg >= 5, prio = cfg < 2 ? 0 : (cfg - 2) % 3;
    if (cfg >= 2) printf("  [matrix role in waves %s, %s]\n", swap ? "4-7 (younger)" : "0-3 (older)", prio == 0 ? "no priority" : prio == 1 ? "VALU half at s_setprio 3" : "matrix half at s_setprio 3");
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemset(ticks, 0, 16);
    hipLaunchKernelGGL((k<VALU_KIND, MFMA_KIND>), dim3(256), dim3(512), 0, 0, out, ticks, 100, mode, swap, prio);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<VALU_KIND, MFMA_KIND>), dim3(256), dim3(512), 0, 0, out, ticks, iters, mode, swap, prio);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    (void)hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
    printf("  mode %d (%s): wall %.1f us | per iteration: MFMA wave %.0f ticks (%.2f per MFMA), VALU wave %.0f ticks (%.2f per instr) | %.2f ns per iteration\n", mode,
           mode == 0 ? "MFMA only" : mode == 1 ? "VALU only" : "both     ", ms * 1e3, (double)h[0] / iters, (double)h[0] / iters / 48, (double)h[1] / iters, (double)h[1] / iters / 64,
           ms * 1e6 / iters);
  }
}

int main() {
  float* out; long long* ticks;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&ticks, 16);
  run<0>("v_fma_f32 stream", out, ticks, 5000);
  run<1>("(v_fma_f32, v_exp_f32) pairs", out, ticks, 5000);
  run<2>("v_fma_f32 over 64 registers", out, ticks, 5000);
  run<3>("softmax-like compiled code (count its VALU in the ISA)", out, ticks, 5000);
  run<0, 2>("v_fma_f32 stream | 1 x s_nop 7 behind every MFMA", out, ticks, 5000);
  run<0, 3>("v_fma_f32 stream | 2 x s_nop 7 behind every MFMA", out, ticks, 5000);
  run<0, 4>("v_fma_f32 stream | 3 x s_nop 7 behind every MFMA", out, ticks, 5000);
  run<3, 3>("softmax-like | 2 x s_nop 7 behind every MFMA", out, ticks, 5000);
  run<3, 4>("softmax-like | 3 x s_nop 7 behind every MFMA", out, ticks, 5000);
  run<0, 1>("v_fma_f32 stream | MFMAs fed by LDS reads", out, ticks, 5000);
  run<3, 1>("softmax-like | MFMAs fed by LDS reads", out, ticks, 5000);
  return 0;
}
