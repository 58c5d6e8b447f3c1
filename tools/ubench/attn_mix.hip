// What bounds the bf16 attention kernel at head_dim 64?  The kernel's own per-tile instruction stream (csrc/attn.hip, attn_bf16_w64_kernel<2>:
// per wave and 64-key tile, two 32-query blocks: 16 score MFMAs + 16 P.V MFMAs, the fp32 online softmax, the bf16 packing) with everything
// that touches memory removed -- operands are register constants, no LDS, no DMA, no barrier -- at the kernel's occupancy (two waves per SIMD).
//   mode 0  the 32 MFMAs alone                                  -> the matrix pipe's own pace
//   mode 1  the softmax VALU work alone                         -> the vector pipe's own pace
//   mode 2  both, INDEPENDENT (softmax on dummy registers)      -> what perfect co-issue of this instruction mix could reach
//   mode 3  both, with the real dataflow S -> softmax -> P.V    -> what the dependency chain allows at two waves per SIMD, memory-free
// Output: cycles per (wave, tile) and the equivalent TFLOP/s of 4 N^2 D attention at 256 CUs; compare mode 3 with the real kernel
// (tools/bench_kernels.py attn): the difference is LDS fragment reads, LDS-DMA and the per-tile barrier.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/ubench/attn_mix.hip -o tools/ubench/attn_mix && tools/ubench/attn_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  f32x2 p = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(p, bf16x2));
}

template <int MODE, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void mix_kernel(float* out, int tiles, unsigned seed) {
  const int lane = threadIdx.x & 63;
  // register-constant "fragments" (values small enough that nothing overflows over the loop)
  bf16x8 qf[2][4], kf0, vf0;
  for (int e = 0; e < 8; ++e) {
    kf0[e] = (__bf16)(0.01f * (float)((lane * 7 + e * 3 + seed) % 13 - 6));
    vf0[e] = (__bf16)(0.02f * (float)((lane * 5 + e + seed) % 11 - 5));
    for (int qb = 0; qb < 2; ++qb)
      for (int ds = 0; ds < 4; ++ds) qf[qb][ds][e] = (__bf16)(0.03f * (float)((lane + e * 5 + ds + qb * 3) % 9 - 4));
  }
  f32x16 oacc[2][2];
  for (int qb = 0; qb < 2; ++qb)
    for (int i = 0; i < 2; ++i)
      for (int r = 0; r < 16; ++r) oacc[qb][i][r] = 0.f;
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  f32x16 dummy[2];  // mode 1 / 2: what the softmax chews on instead of the scores
  for (int ks = 0; ks < 2; ++ks)
    for (int r = 0; r < 16; ++r) dummy[ks][r] = 0.001f * (float)((lane + r + ks) % 17);
  const float c = 0.125f * 1.44269504088896340736f;

  for (int t = 0; t < tiles; ++t) {
    // loop-variant operands (one VALU each), so nothing is hoisted out of the tile loop
    bf16x8 kf = kf0, vf = vf0;
    kf[0] = (__bf16)(0.01f * (float)((t + lane) & 7));
    vf[0] = (__bf16)(0.02f * (float)((t + lane) & 3));
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 sacc[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[ks][r] = 0.f;
        if (MODE != 1) {
#pragma unroll
          for (int ds = 0; ds < 4; ++ds) sacc[ks] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][ds], sacc[ks], 0, 0, 0);
          if (MODE != 3) asm volatile("" ::"v"(sacc[ks]));  // the scores are not consumed in these modes: keep their MFMAs
        }
      }
      bf16x8 pf[4];
      if (MODE != 0) {
        f32x16* sc = (MODE == 3) ? sacc : dummy;   // compile-time choice
        // ---- the kernel's online softmax, verbatim
        float mx = fmaxf(sc[0][0], sc[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, sc[0][r]), sc[1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run[qb], mx);
        const bool grow = __any(m_new > m_run[qb]);
        float alpha = 1.f;
        if (grow) alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c);
        m_run[qb] = m_new;
        float psum = 0.f;
        const float mc = m_new * c;
        float p[2][16];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            p[ks][r] = __builtin_amdgcn_exp2f(fmaf(sc[ks][r], c, -mc));
            psum += p[ks][r];
          }
        if (grow) {
          l_run[qb] *= alpha;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qb][i][r] *= alpha;
        }
        l_run[qb] += psum;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const int r0 = 8 * kk;
            unsigned a0 = pack_bf16x2(p[ks][r0 + 0], p[ks][r0 + 1]), a1 = pack_bf16x2(p[ks][r0 + 2], p[ks][r0 + 3]);
            unsigned b0 = pack_bf16x2(p[ks][r0 + 4], p[ks][r0 + 5]), b1 = pack_bf16x2(p[ks][r0 + 6], p[ks][r0 + 7]);
            auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            pf[ks * 2 + kk] = __builtin_bit_cast(bf16x8, make_uint4(s0[0], s1[0], s0[1], s1[1]));
          }
        if (MODE != 3) {  // keep the results alive without feeding the MFMAs
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(pf[i]));
          dummy[0][0] -= 1e-6f;  // (a loop-carried nudge that never raises the row maximum: the rescale branch stays as rare as in the kernel)
        }
      }
      if (MODE != 1) {
#pragma unroll
        for (int kstep = 0; kstep < 4; ++kstep)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
            oacc[qb][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, (MODE == 3) ? pf[kstep] : qf[qb][kstep], oacc[qb][dt], 0, 0, 0);
      }
    }
  }
  float acc = l_run[0] + l_run[1] + m_run[0] + m_run[1];
  for (int qb = 0; qb < 2; ++qb)
    for (int i = 0; i < 2; ++i)
      for (int r = 0; r < 16; ++r) acc += oacc[qb][i][r];
  for (int ks = 0; ks < 2; ++ks)
    for (int r = 0; r < 16; ++r) acc += dummy[ks][r];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE, int WPS>
static double run(int wgs, int tiles, float* d_out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  mix_kernel<MODE, WPS><<<wgs, 256>>>(d_out, tiles, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) mix_kernel<MODE, WPS><<<wgs, 256>>>(d_out, tiles, 2 + i);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5.0;
}

int main() {
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* d_out;
  hipMalloc(&d_out, (size_t)cus * 4 * 256 * sizeof(float));
  const int tiles = 2000;
  const double flop_tile = 32.0 * (2.0 * 32 * 32 * 16);  // 32 MFMAs of 32x32x16 per (wave, tile)
  const char* names[4] = {"mfma only", "softmax only", "both, independent", "both, real dataflow"};
  for (int occ = 2; occ >= 1; --occ) {
    const int wgs = cus * occ;  // 4 waves per workgroup = one per SIMD; `occ` workgroups per CU = `occ` waves per SIMD
    double ms[4];
    ms[0] = occ == 2 ? run<0, 2>(wgs, tiles, d_out) : run<0, 1>(wgs, tiles, d_out);
    ms[1] = occ == 2 ? run<1, 2>(wgs, tiles, d_out) : run<1, 1>(wgs, tiles, d_out);
    ms[2] = occ == 2 ? run<2, 2>(wgs, tiles, d_out) : run<2, 1>(wgs, tiles, d_out);
    ms[3] = occ == 2 ? run<3, 2>(wgs, tiles, d_out) : run<3, 1>(wgs, tiles, d_out);
    for (int m = 0; m < 4; ++m) {
      const double us_tile = ms[m] * 1e3 / tiles;                         // per tile, with `occ` waves sharing each SIMD
      const double tf = m == 1 ? 0.0 : flop_tile * (double)wgs * 4 * tiles / (ms[m] * 1e-3) / 1e12;
      printf("%d wave(s)/SIMD  %-22s %8.3f us per tile-round  %8.1f TFLOP/s equivalent (%.3f of 2500)\n", occ, names[m], us_tile, tf, tf / 2500.0);
    }
  }
  return 0;
}
