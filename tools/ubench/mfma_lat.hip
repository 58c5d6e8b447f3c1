// MFMA issue/dependency micro-benchmark: cycles per MFMA for chains over NACC independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC, int KIND>
__global__ void k(float* out, unsigned long long* cyc, int iters, float seed) {
  f32x16 acc32[NACC];
  f32x4 acc16[NACC];
  for (int i = 0; i < NACC; ++i) {
    for (int r = 0; r < 16; ++r) acc32[i][r] = seed * i;
    for (int r = 0; r < 4; ++r) acc16[i][r] = seed * i;
  }
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x); b[i] = (__bf16)(seed * 2 + i); }
  float fa = seed + threadIdx.x, fb = seed * 3;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (KIND == 0) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc32[i], 0, 0, 0);
        else if (KIND == 1) acc16[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc16[i], 0, 0, 0);
        else acc32[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc32[i], 0, 0, 0);
      }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) s += acc32[i][r]; for (int r = 0; r < 4; ++r) s += acc16[i][r]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC, int KIND>
void run(const char* name, int waves_per_block, int blocks) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * waves_per_block * 64 * 4);
  hipMalloc(&cyc, blocks * waves_per_block * 8);
  int iters = 2000;
  hipLaunchKernelGGL((k<NACC, KIND>), dim3(blocks), dim3(waves_per_block * 64), 0, 0, out, cyc, iters, 0.001f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, KIND>), dim3(blocks), dim3(waves_per_block * 64), 0, 0, out, cyc, iters, 0.001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * waves_per_block);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= h.size();
  double n = (double)iters * 8 * NACC;
  printf("%-14s nacc=%d waves/blk=%d blocks=%d : %.1f ticks/MFMA/wave, kernel %.3f ms, %.1f ns/MFMA/wave\n", name, NACC, waves_per_block, blocks, avg / n, ms, ms * 1e6 / n);
  hipFree(out); hipFree(cyc);
}

int main() {
  // one wave per SIMD on every CU (256 CUs x 4 waves), then 2 waves per SIMD
  for (int wpb : {4, 8}) {
    run<1, 0>("bf16 32x32x16", wpb, 256); run<2, 0>("bf16 32x32x16", wpb, 256); run<4, 0>("bf16 32x32x16", wpb, 256); run<8, 0>("bf16 32x32x16", wpb, 256);
    run<1, 1>("f32 16x16x4", wpb, 256); run<2, 1>("f32 16x16x4", wpb, 256); run<4, 1>("f32 16x16x4", wpb, 256);
    run<1, 2>("f32 32x32x2", wpb, 256); run<2, 2>("f32 32x32x2", wpb, 256); run<4, 2>("f32 32x32x2", wpb, 256);
  }
  return 0;
}
