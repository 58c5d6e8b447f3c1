// Probes the operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3, unit scales) on gfx950:
// C = A (32 x 64) * B (64 x 32) with A[i][k], B[k][j] given as small exactly-representable fp8 values; the host checks
// the hypothesis  lane l holds A[i = l & 31][k = 32*(l >> 5) + 0..31]  (byte b of the 8 VGPRs = k offset b), same for B
// with j = l & 31, and the usual 32x32 C map.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/fp8_probe.hip -o /tmp/fp8_probe && /tmp/fp8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void probe(const uint8_t* A, const uint8_t* B, float* C) {  // A [32][64] fp8, B [64][32] fp8 (k-major), C [32][32]
  const int l = threadIdx.x, r = l & 31, g = l >> 5;
  i32x8 a, b;
  for (int v = 0; v < 8; ++v) {
    unsigned wa = 0, wb = 0;
    for (int e = 0; e < 4; ++e) {
      const int k = 32 * g + 4 * v + e;
      wa |= (unsigned)A[r * 64 + k] << (8 * e);
      wb |= (unsigned)B[k * 32 + r] << (8 * e);
    }
    a[v] = wa; b[v] = wb;
  }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * g;
    C[row * 32 + r] = c[i];
  }
}

static uint8_t to_e4m3(float x) {  // exact for the small values used here: sign, 4-bit exponent (bias 7), 3-bit mantissa
  if (x == 0.f) return 0;
  uint8_t s = x < 0 ? 0x80 : 0;
  float a = fabsf(x);
  int e; float m = frexpf(a, &e);  // a = m * 2^e, m in [0.5, 1)
  int E = e - 1 + 7; int M = (int)lrintf((m * 2.f - 1.f) * 8.f);
  return s | (uint8_t)(E << 3) | (uint8_t)M;
}

int main() {
  std::vector<float> fa(32 * 64), fb(64 * 32);
  std::vector<uint8_t> A(32 * 64), B(64 * 32);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) { fa[i * 64 + k] = (float)(((i * 7 + k * 3) % 9) - 4) * 0.5f; A[i * 64 + k] = to_e4m3(fa[i * 64 + k]); }
  for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) { fb[k * 32 + j] = (float)(((k * 5 + j * 11) % 7) - 3) * 0.25f; B[k * 32 + j] = to_e4m3(fb[k * 32 + j]); }
  uint8_t *dA, *dB; float* dC;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 32 * 32 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dC);
  std::vector<float> C(32 * 32);
  hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    float ref = 0; for (int k = 0; k < 64; ++k) ref += fa[i * 64 + k] * fb[k * 32 + j];
    if (C[i * 32 + j] != ref) { if (bad < 5) printf("mismatch C[%d][%d] = %g, want %g\n", i, j, C[i * 32 + j], ref); ++bad; }
  }
  printf("fp8 32x32x64 layout hypothesis: %s (%d mismatches)\n", bad ? "WRONG" : "CONFIRMED", bad);
  return bad != 0;
}
