// What bounds the bf16 attention kernel at head_dim 64?  The kernel's own per-tile instruction stream (csrc/attn.hip, attn_bf16_w64_kernel<2>:
// per wave and 64-key tile, two 32-query blocks: 16 score MFMAs + 16 P.V MFMAs, the fp32 online softmax, the bf16 packing) with everything
// that touches memory removed -- operands are register values, no LDS, no DMA, no barrier -- at the kernel's occupancy (two waves per SIMD).
//   mode 0  the 32 MFMAs alone                                  -> the matrix pipe's own pace under this dependency structure
//   mode 1  the softmax VALU work alone                         -> the vector pipe's own pace
//   mode 2  both, INDEPENDENT (softmax on dummy registers)
//   mode 3  both, with the real dataflow S -> softmax -> P.V    -> the kernel's tile loop, memory-free
// plus: the rescale branch taken every tile (what the exact running maximum costs on random scores), the kernel's launch shape (3072
// workgroups x 22 tiles), and the leaner softmax streams that were candidates (reference folded into the score MFMA's C operand = no fma;
// no running maximum; row sum by v_dot2c over the packed probabilities) -- none of which is faster than the plain stream by more than 6 %.
// Measured (MI355X, us per tile-round = one tile of each of the two waves of a SIMD; fraction of the 2.5 PFLOP/s dense bf16 peak):
//   MFMAs alone 1.30 (0.66) | softmax alone 1.30 | real dataflow 1.84 (0.47) | launch shape 241 us (0.42) -- the same 240 us the real
//   kernel takes with its DMA, barrier and LDS reads compiled out (tools/attn_ablate.sh).  The real kernel: 324 us (0.31).
// (The two 32-key halves of a tile must get different fragments: with one fragment the compiler merges them -- 24 MFMAs, half the softmax.)
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

template <int MODE, int WAVES_PER_SIMD, bool GROW = false, bool ILP = false>
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
    // (the two 32-key halves get DIFFERENT fragments: with one fragment the compiler merges the halves -- 24 MFMAs and half the softmax)
    bf16x8 kfs[2] = {kf0, kf0}, vf = vf0;
    kfs[0][0] = (__bf16)(0.01f * (float)((t + lane) & 7));
    kfs[1][1] = (__bf16)(0.01f * (float)((t + lane + 3) & 7));
    vf[0] = (__bf16)(0.02f * (float)((t + lane) & 3));
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 sacc[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 kf = kfs[ks];
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
        float mx;
        if (ILP) {  // four independent max chains / row-sum chains instead of one serial chain each
          float m4[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) m4[i] = fmaxf(sc[0][i], sc[1][i]);
#pragma unroll
          for (int r = 4; r < 16; ++r) m4[r & 3] = fmaxf(fmaxf(m4[r & 3], sc[0][r]), sc[1][r]);
          mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
        } else {
          mx = fmaxf(sc[0][0], sc[1][0]);
#pragma unroll
          for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, sc[0][r]), sc[1][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run[qb], mx);
        const bool grow = GROW ? __any(m_new >= m_run[qb]) : __any(m_new > m_run[qb]);  // GROW: the rescale branch taken every tile
        float alpha = 1.f;
        if (grow) alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c);
        m_run[qb] = m_new;
        float psum = 0.f;
        const float mc = m_new * c;
        float p[2][16], ps4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            p[ks][r] = __builtin_amdgcn_exp2f(fmaf(sc[ks][r], c, -mc));
            if (ILP) ps4[r & 3] += p[ks][r];
            else psum += p[ks][r];
          }
        if (ILP) psum = (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
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

// The lean stream (what the softmax can shrink to): Q pre-scaled by head_dim^-0.5 log2(e), the exponent's reference folded into the
// score MFMA's C operand (16 registers per query block holding -m), so p = exp2(S) with no fma; no running maximum at all -- the
// row sum of the tile (needed anyway) bounds every p, a sum above 2^9 sends the wave down a rare re-reference path; the row sum
// itself by v_dot2c_f32_bf16 over the PACKED probabilities (two per instruction).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
template <int WAVES_PER_SIMD, bool DOT, bool MAX>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void lean_kernel(float* out, int tiles, unsigned seed) {
  const int lane = threadIdx.x & 63;
  bf16x8 qf[2][4], kf0, vf0;
  for (int e = 0; e < 8; ++e) {
    kf0[e] = (__bf16)(0.01f * (float)((lane * 7 + e * 3 + seed) % 13 - 6));
    vf0[e] = (__bf16)(0.02f * (float)((lane * 5 + e + seed) % 11 - 5));
    for (int qb = 0; qb < 2; ++qb)
      for (int ds = 0; ds < 4; ++ds) qf[qb][ds][e] = (__bf16)(0.03f * (float)((lane + e * 5 + ds + qb * 3) % 9 - 4));
  }
  f32x16 oacc[2][2], minit[2];
  for (int qb = 0; qb < 2; ++qb) {
    for (int i = 0; i < 2; ++i)
      for (int r = 0; r < 16; ++r) oacc[qb][i][r] = 0.f;
    for (int r = 0; r < 16; ++r) minit[qb][r] = -0.25f;
  }
  float l_run[2] = {0.f, 0.f};
  const bf16x2_t ones = {(__bf16)1.0f, (__bf16)1.0f};
  for (int t = 0; t < tiles; ++t) {
    bf16x8 kfs[2] = {kf0, kf0}, vf = vf0;
    kfs[0][0] = (__bf16)(0.01f * (float)((t + lane) & 7));
    kfs[1][1] = (__bf16)(0.01f * (float)((t + lane + 3) & 7));
    vf[0] = (__bf16)(0.02f * (float)((t + lane) & 3));
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 sacc[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 kf = kfs[ks];
        sacc[ks] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][0], minit[qb], 0, 0, 0);
#pragma unroll
        for (int ds = 1; ds < 4; ++ds) sacc[ks] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][ds], sacc[ks], 0, 0, 0);
      }
      bf16x8 pf[4];
      float psum = 0.f, mx = -INFINITY;
      if (MAX) {  // (keep the tile maximum: the current kernel's trigger for moving the reference)
        mx = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, sacc[0][r]), sacc[1][r]);
        auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, mx), __builtin_bit_cast(unsigned, mx), false, false);
        const unsigned m0 = sw[0], m1 = sw[1];
        mx = fmaxf(__builtin_bit_cast(float, m0), __builtin_bit_cast(float, m1));
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int r0 = 8 * kk;
          float p[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) p[i] = __builtin_amdgcn_exp2f(sacc[ks][r0 + i]);
          unsigned a0 = pack_bf16x2(p[0], p[1]), a1 = pack_bf16x2(p[2], p[3]);
          unsigned b0 = pack_bf16x2(p[4], p[5]), b1 = pack_bf16x2(p[6], p[7]);
          if (DOT) {
            psum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a0), ones, psum, false);
            psum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a1), ones, psum, false);
            psum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, b0), ones, psum, false);
            psum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, b1), ones, psum, false);
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) psum += p[i];
          }
          auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
          auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
          pf[ks * 2 + kk] = __builtin_bit_cast(bf16x8, make_uint4(s0[0], s1[0], s0[1], s1[1]));
        }
      if (__any(MAX ? mx > 8.f : psum > 512.f)) {  // the rare path (never taken here): move the reference, rescale O and l
        const float alpha = __builtin_amdgcn_exp2f(-__builtin_amdgcn_logf(psum));
        l_run[qb] *= alpha;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[qb][i][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) minit[qb][r] -= 1.f;
      }
      l_run[qb] += psum;
#pragma unroll
      for (int kstep = 0; kstep < 4; ++kstep)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) oacc[qb][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kstep], oacc[qb][dt], 0, 0, 0);
    }
  }
  float acc = l_run[0] + l_run[1];
  for (int qb = 0; qb < 2; ++qb)
    for (int i = 0; i < 2; ++i)
      for (int r = 0; r < 16; ++r) acc += oacc[qb][i][r] + minit[qb][r];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int WPS, bool DOT, bool MAX>
static double run_lean(int wgs, int tiles, float* d_out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  lean_kernel<WPS, DOT, MAX><<<wgs, 256>>>(d_out, tiles, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) lean_kernel<WPS, DOT, MAX><<<wgs, 256>>>(d_out, tiles, 2 + i);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5.0;
}

template <int MODE, int WPS, bool GROW = false, bool ILP = false>
static double run(int wgs, int tiles, float* d_out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  mix_kernel<MODE, WPS, GROW, ILP><<<wgs, 256>>>(d_out, tiles, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) mix_kernel<MODE, WPS, GROW, ILP><<<wgs, 256>>>(d_out, tiles, 2 + i);
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
  hipMalloc(&d_out, (size_t)4096 * 256 * sizeof(float));
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
  {
    const double a = run<1, 2, false, true>(cus * 2, tiles, d_out), b = run<3, 2, false, true>(cus * 2, tiles, d_out), c1 = run<3, 1, false, true>(cus, tiles, d_out);
    printf("four max / row-sum chains instead of one:  softmax only %8.3f   real dataflow %8.3f (2 waves/SIMD)  %8.3f (1 wave/SIMD) us per tile-round\n",
           a * 1e3 / tiles, b * 1e3 / tiles, c1 * 1e3 / tiles);
  }
  // the same stream with the rescale branch taken in every tile (random scores: some lane of 64 sees a new maximum in most tiles)
  {
    const double ms = run<3, 2, true>(cus * 2, tiles, d_out);
    printf("2 wave(s)/SIMD  real dataflow, rescale every tile   %8.3f us per tile-round\n", ms * 1e3 / tiles);
  }
  {
    const double flop = flop_tile * 4 * tiles;
    const double a = run_lean<2, false, true>(cus * 2, tiles, d_out), b = run_lean<2, false, false>(cus * 2, tiles, d_out), c = run_lean<2, true, false>(cus * 2, tiles, d_out);
    printf("2 wave(s)/SIMD  reference in the MFMA's C operand (no fma), max + add row sum   %8.3f us per tile-round  (%.3f of 2500)\n", a * 1e3 / tiles, flop * cus * 2 / (a * 1e-3) / 2.5e15);
    printf("2 wave(s)/SIMD  ... and no maximum (row sum bounds p)                          %8.3f us per tile-round  (%.3f of 2500)\n", b * 1e3 / tiles, flop * cus * 2 / (b * 1e-3) / 2.5e15);
    printf("2 wave(s)/SIMD  ... and v_dot2c row sum over the packed probabilities          %8.3f us per tile-round  (%.3f of 2500)\n", c * 1e3 / tiles, flop * cus * 2 / (c * 1e-3) / 2.5e15);
    const double l = run_lean<2, false, true>(3072, 22, d_out), l2 = run_lean<2, false, false>(3072, 22, d_out);
    printf("launch shape 3072 x 22 tiles, no fma: %8.1f us;  no fma, no max: %8.1f us\n", l * 1e3, l2 * 1e3);
  }
  // the kernel's launch shape: 3072 workgroups x 22 tiles (ViT-L, batch 32: 512 (image, head) pairs x 6 query tiles), memory-free
  {
    const double ms = run<3, 2>(3072, 22, d_out), msg = run<3, 2, true>(3072, 22, d_out);
    printf("launch shape 3072 x 22 tiles: %8.1f us   (rescale every tile: %8.1f us)   [steady-state pace predicts %.1f us]\n", ms * 1e3, msg * 1e3,
           run<3, 2>(cus * 2, tiles, d_out) * 1e3 / tiles * 22 * 6);
  }
  return 0;
}
