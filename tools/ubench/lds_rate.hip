// LDS read throughput on gfx950: cycles per instruction for ds_read_b128 and ds_read_b64_tr_b16 streams (the fragment reads of the attention
// kernels), 1 / 2 / 4 / 8 waves per CU issuing, with the conflict-free address patterns the kernels use.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_rate.hip -o tools/ubench/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, long long* ticks, int iters) {
  __shared__ __attribute__((aligned(16))) char L[65536];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float*>(L)[i] = (float)i;
  __syncthreads();
  float acc = 0.f;
  const long long t0 = clock64();
  if constexpr (KIND == 0) {  // ds_read_b128: lane -> row (lane & 31), chunk swizzled by the row (attn_split_kernel's K image)
    const int row = lane & 31, kh = lane >> 5;
    const char* base = L + wave * 8192 + row * 256;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float4 v = *reinterpret_cast<const float4*>(base + ((((u & 7) * 2 + kh) ^ (row & 15)) << 4) + (u >> 3) * 4096 * 0);
        acc += v.x;
        asm volatile("" ::: "memory");
      }
    }
  } else {                    // ds_read_b64_tr_b16: the V image's transpose reads (attn_split_kernel)
    const int kh = lane >> 5, kq = (lane & 15) >> 2, vb = (lane >> 4) & 1;
    const int vrd0 = (kh * 8 + kq) * 256 + vb * 32 + (lane & 3) * 8;
    const char* base = L + (wave & 3) * 16384 + vrd0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + (((u & 3) ^ kq) << 6) + (u >> 2) * 4096));
        acc += (float)v[0];
        asm volatile("" ::: "memory");
      }
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * 512 + threadIdx.x] = acc;
  if (blockIdx.x == 100 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int KIND>
static void run(const char* name, float* out, long long* ticks) {
  const int iters = 2000;
  for (int threads = 64; threads <= 512; threads *= 2) {
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(threads), 0, 0, out, ticks, 10);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(threads), 0, 0, out, ticks, iters);
    (void)hipDeviceSynchronize();
    long long h;
    (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
    const double per = (double)h / (iters * 16.0);
    printf("%-22s %d wave(s)/CU: %.2f cycles per instruction per wave -> %.2f cycles of LDS per instruction\n", name, threads / 64, per, per / (threads / 64));
  }
}
int main() {
  float* out; long long* ticks;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&ticks, 16);
  run<0>("ds_read_b128", out, ticks);
  run<1>("ds_read_b64_tr_b16", out, ticks);
  return 0;
}
