// Per-CU global->LDS DMA throughput vs outstanding depth (1 workgroup of 8 waves per CU, L2-resident source).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

template <int DEPTH, int WIN>  // DMA instructions in flight per wave; number of distinct 1 MiB windows
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, size_t span, int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // each block walks its own 1 MiB window (L2 resident after the first pass), 1 KiB per wave instruction
  const char* base = src + ((size_t)(blockIdx.x % WIN) * (1 << 20)) % span;
  char* lds = smem + wave * DEPTH * 1024;
  for (int d = 0; d < DEPTH; ++d)
    __builtin_amdgcn_global_load_lds((gbl_cvoid*)(base + ((wave * 64 + d) * 1024 + lane * 16) % (1 << 20)), (lds_void*)(lds + d * 1024), 16, 0, 0);
  for (int it = DEPTH; it < iters; ++it) {
    // wait for the oldest, then reuse its slot
    if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (DEPTH == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if (DEPTH == 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
    __builtin_amdgcn_global_load_lds((gbl_cvoid*)(base + (((wave * 64 + it) * 1024) + lane * 16) % (1 << 20)), (lds_void*)(lds + (it % DEPTH) * 1024), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && smem[5] == 77) out[0] = 1;
}

template <int DEPTH, int WIN>
void run(const char* src, size_t span, float* out, int blocks) {
  const int iters = 2048;
  size_t lds = 8 * DEPTH * 1024;
  hipFuncSetAttribute((const void*)k<DEPTH, WIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<DEPTH, WIN>), dim3(blocks), dim3(512), lds, 0, src, span, iters, out);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<DEPTH, WIN>), dim3(blocks), dim3(512), lds, 0, src, span, iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double bytes = (double)blocks * 8 * iters * 1024;
  printf("windows %3d, depth %2d per wave (%3d KiB in flight/CU), %d blocks: %.3f ms  %.1f GB/s per CU  %.2f TB/s chip\n", WIN, DEPTH, 8 * DEPTH, blocks, ms, bytes / blocks / ms / 1e6, bytes / ms / 1e9);
}

int main() {
  char* src; float* out;
  size_t span = (size_t)256 << 20;
  hipMalloc(&src, span); hipMemset(src, 1, span); hipMalloc(&out, 64);
  run<8, 256>(src, span, out, 256);   // 256 MiB footprint: beyond L2 and most of the Infinity Cache
  run<8, 64>(src, span, out, 256);    // 64 MiB: Infinity Cache
  run<8, 24>(src, span, out, 256);    // 24 MiB: 3 MiB per XCD (block b -> XCD b % 8, window b % 24): fits the 4 MiB L2s
  run<8, 8>(src, span, out, 256);     // 8 MiB: 1 MiB per XCD
  run<2, 8>(src, span, out, 256);
  run<16, 8>(src, span, out, 256);
  run<8, 8>(src, span, out, 128);
  run<8, 8>(src, span, out, 64);
  return 0;
}
