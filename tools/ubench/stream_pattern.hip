// Streaming-read pattern test on an [T, W] fp32 matrix (82 MB): which per-wave access shape reaches HBM bandwidth?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int W = 2048;

// pattern 0: the cosine kernel's shape: wave = 16 rows x one 256-float slice, lane (i=l&15,g=l>>4) reads 16 B at col 16j+4g
// pattern 1: wave = one row at a time, 64 lanes x 16 B = 1 KiB contiguous per instruction, 16 rows x 256-float slice
template <int PAT, int NT>
__global__ __launch_bounds__(256) void k(const float* __restrict__ a, int T, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = (blockIdx.x * 4 + wave) * 16, slice = blockIdx.z;
  if (t0 >= T) return;
  f32x4 acc = {0, 0, 0, 0};
  if (PAT == 0) {
    const int i = lane & 15, g = lane >> 4;
    const float* p = a + (size_t)min(t0 + i, T - 1) * W + slice * 256 + 4 * g;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      f32x4 v = NT ? __builtin_nontemporal_load((const f32x4*)(p + 16 * j)) : *(const f32x4*)(p + 16 * j);
      acc += v;
    }
  } else if (PAT == 2) {
    // fully linear: the wave's 16 KiB are one contiguous run (what a [block][slice][16][256] re-layout of the bank gives)
    const size_t wid = ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 4 + wave;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* p = a + wid * 4096 + r * 256 + 4 * lane;
      f32x4 v = NT ? __builtin_nontemporal_load((const f32x4*)p) : *(const f32x4*)p;
      acc += v;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* p = a + (size_t)min(t0 + r, T - 1) * W + slice * 256 + 4 * lane;
      f32x4 v = NT ? __builtin_nontemporal_load((const f32x4*)p) : *(const f32x4*)p;
      acc += v;
    }
  }
  float s = acc[0] + acc[1] + acc[2] + acc[3];
  if (s == 123.456f) out[0] = s;
}

template <int PAT, int NT>
void run(const float* a, int T, float* out, const char* name) {
  dim3 grid((T / 16 + 3) / 4, 1, 8);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<PAT, NT>), grid, dim3(256), 0, 0, a, T, out);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<PAT, NT>), grid, dim3(256), 0, 0, a, T, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
  printf("%-40s %.1f us  %.2f TB/s\n", name, ms * 1e3, (double)T * W * 4 / ms / 1e9);
}

int main() {
  const int T = 10000;
  float *a, *out;
  hipMalloc(&a, (size_t)T * W * 4); hipMalloc(&out, 64);
  hipMemset(a, 0, (size_t)T * W * 4);
  // also a second, larger matrix to defeat the 256 MB Infinity Cache between repetitions
  run<0, 0>(a, T, out, "16 rows x 64 B per instr");
  run<0, 1>(a, T, out, "16 rows x 64 B per instr, nontemporal");
  run<1, 0>(a, T, out, "1 row x 1 KiB per instr");
  run<1, 1>(a, T, out, "1 row x 1 KiB per instr, nontemporal");
  const int T2 = 50000;
  float* b; hipMalloc(&b, (size_t)T2 * W * 4); hipMemset(b, 0, (size_t)T2 * W * 4);
  run<0, 0>(b, T2, out, "T=50k: 16 rows x 64 B");
  run<0, 1>(b, T2, out, "T=50k: 16 rows x 64 B, nt");
  run<1, 0>(b, T2, out, "T=50k: 1 row x 1 KiB");
  run<1, 1>(b, T2, out, "T=50k: 1 row x 1 KiB, nt");
  run<2, 0>(b, T2, out, "T=50k: linear 16 KiB per wave");
  run<2, 1>(b, T2, out, "T=50k: linear 16 KiB per wave, nt");
  run<2, 0>(a, T, out, "T=10k: linear 16 KiB per wave");
  return 0;
}
