// bf16 MFMA GEMM for the ViT linear layers: C = A[M,K] * W[N,K]^T (+ fused epilogue), fp32 accumulate.
//
// This is the arithmetic the reference delegates to the DINOv2 backbone's nn.Linear layers
// (call site /root/reference/utils/dinov2_utils.py:257).  MI355X design:
//   * tile 128x128x64, 256 threads = 2x2 waves, each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16
//   * A and W tiles go HBM -> LDS with global_load_lds (16 B/lane, no VGPR round trip), double buffered,
//     one barrier per K-tile, next tile's DMA in flight under the MFMAs
//   * LDS image is row-major [row][64 bf16]; bank conflicts of the ds_read_b128 fragment reads are
//     removed by XOR-swizzling the 16-B chunk index with (row>>1)&7 -- applied on the *source* address
//     (the DMA destination is lane-linear) and again on the read (guide section 5.4 rule 21)
//   * operands are fed to the MFMA swapped (W as the "A" operand) so each lane ends up with 4 consecutive
//     output columns of one row: 8-byte bf16 / 16-byte fp32 epilogue accesses
//   * blockIdx is remapped so that each XCD's L2 sees a contiguous run of tiles sharing A panels.
#include "common.hpp"
#include "kernels.hpp"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

FP_DEVICE int swz(int row) { return (row >> 1) & 7; }

// Issue the DMA of one 128x64 bf16 tile (rows row0.., k0..k0+63) into `lds` (byte offset base).
FP_DEVICE void stage_tile(const __bf16* __restrict__ g, int ld, int row0, int k0, char* lds, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rblk = wave * 4 + i;            // 8-row group handled by this instruction
    const int row = rblk * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ swz(row);  // logical 16-B chunk that must land at physical slot lane&7
    const __bf16* src = g + (size_t)(row0 + row) * ld + k0 + chunk * 8;
    __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)(lds + rblk * 1024), 16, 0, 0);
  }
}

FP_DEVICE bf16x8 read_frag(const char* lds, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((chunk ^ swz(row)) << 4));
}

FP_DEVICE float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmBf16Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A 16K | B 16K]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, kh = lane >> 5;

  const unsigned tiles_n = a.N / BN;
  const unsigned nwg = gridDim.x;
  const unsigned lid = xcd_remap(blockIdx.x, nwg);
  const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = a.K / BK;
  stage_tile(a.A, a.lda, m0, 0, smem, wave, lane);
  stage_tile(a.W, a.ldw, n0, 0, smem + TILE_BYTES, wave, lane);

  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    __syncthreads();  // drains the DMA of tile t (vmcnt(0)) and fences the readers of the other stage
    if (t + 1 < nk) {
      char* nxt = smem + (cur ^ 1) * 2 * TILE_BYTES;
      stage_tile(a.A, a.lda, m0, (t + 1) * BK, nxt, wave, lane);
      stage_tile(a.W, a.ldw, n0, (t + 1) * BK, nxt + TILE_BYTES, wave, lane);
    }
    const char* As = smem + cur * 2 * TILE_BYTES;
    const char* Ws = As + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int chunk = ks * 2 + kh;
      bf16x8 af0 = read_frag(As, wm * 64 + l31, chunk);
      bf16x8 af1 = read_frag(As, wm * 64 + 32 + l31, chunk);
      bf16x8 wf0 = read_frag(Ws, wn * 64 + l31, chunk);
      bf16x8 wf1 = read_frag(Ws, wn * 64 + 32 + l31, chunk);
      // swapped operands: D[i = n][j = m]
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf0, af0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1, af0, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf0, af1, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1, af1, acc[1][1], 0, 0, 0);
    }
  }

  // ---- epilogue: acc[tm][tn][r] = C[m][n],  m = m0 + wm*64 + tm*32 + (lane&31),
  //      n = n0 + wn*64 + tn*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)
  // Column-only operands are fetched once, row operands in one batch per tm, so the tail is a
  // few waits instead of one per access.
  float4 bias[2][4], gam[2][4];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + wn * 64 + tn * 32 + 8 * g + 4 * kh;
      bias[tn][g] = *reinterpret_cast<const float4*>(a.bias + n);
      if constexpr (EPI == GEMM_EPI_LS_RESID_F32) gam[tn][g] = *reinterpret_cast<const float4*>(a.gamma + n);
    }
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int m = m0 + wm * 64 + tm * 32 + l31;
    if (m >= a.M_valid) continue;
    size_t out_row = m;
    int vb = 0, vt = 0, pidx = 0;
    if constexpr (EPI == GEMM_EPI_TOKENS_F32) {
      const int b = m / a.tok_np;
      pidx = m - b * a.tok_np;
      out_row = (size_t)b * a.tok_n + a.tok_skip + pidx;
    }
    if constexpr (EPI == GEMM_EPI_QKV_BF16) {
      vb = m / a.tok_n;
      vt = m - vb * a.tok_n;
    }
    float4 extra[2][4];  // residual row (LS_RESID) or pos-embed row (TOKENS)
    if constexpr (EPI == GEMM_EPI_LS_RESID_F32 || EPI == GEMM_EPI_TOKENS_F32) {
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wn * 64 + tn * 32 + 8 * g + 4 * kh;
          if constexpr (EPI == GEMM_EPI_LS_RESID_F32)
            extra[tn][g] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.out) + out_row * a.ldo + n);
          else
            extra[tn][g] = *reinterpret_cast<const float4*>(a.pos + (size_t)pidx * a.ldo + n);
        }
    }
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + tn * 32 + 8 * g + 4 * kh;
        const float4 bs = bias[tn][g];
        float v0 = acc[tm][tn][4 * g + 0] + bs.x, v1 = acc[tm][tn][4 * g + 1] + bs.y;
        float v2 = acc[tm][tn][4 * g + 2] + bs.z, v3 = acc[tm][tn][4 * g + 3] + bs.w;
        if constexpr (EPI == GEMM_EPI_BIAS_BF16 || EPI == GEMM_EPI_GELU_BF16 || EPI == GEMM_EPI_QKV_BF16) {
          if constexpr (EPI == GEMM_EPI_GELU_BF16) {
            v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
          }
          bool transposed_v = false;
          if constexpr (EPI == GEMM_EPI_QKV_BF16) transposed_v = n >= 2 * a.vit_dim;
          if (!transposed_v) {
            uint2 pk = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
            *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(a.out) + out_row * a.ldo + n) = pk;
          } else {
            // V goes out transposed, Vt[b][head][d][t] (keys contiguous), for the attention P*V operand
            const int nn = n - 2 * a.vit_dim;  // head*64 + d
            __bf16* vtp = a.vt + ((size_t)vb * a.vit_dim + nn) * a.vt_ld + vt;
            vtp[0 * (size_t)a.vt_ld] = (__bf16)v0;
            vtp[1 * (size_t)a.vt_ld] = (__bf16)v1;
            vtp[2 * (size_t)a.vt_ld] = (__bf16)v2;
            vtp[3 * (size_t)a.vt_ld] = (__bf16)v3;
          }
        } else if constexpr (EPI == GEMM_EPI_LS_RESID_F32) {
          const float4 gm = gam[tn][g];
          float4 x = extra[tn][g];
          x.x += gm.x * v0; x.y += gm.y * v1; x.z += gm.z * v2; x.w += gm.w * v3;
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + out_row * a.ldo + n) = x;
        } else if constexpr (EPI == GEMM_EPI_TOKENS_F32) {
          const float4 pe = extra[tn][g];
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + out_row * a.ldo + n) =
              make_float4(v0 + pe.x, v1 + pe.y, v2 + pe.z, v3 + pe.w);
        } else if constexpr (EPI == GEMM_EPI_BIAS_F32) {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + out_row * a.ldo + n) = make_float4(v0, v1, v2, v3);
        }
      }
  }
}

template <int EPI>
int launch(const GemmBf16Args& a, hipStream_t st) {
  const unsigned grid = (a.M / BM) * (a.N / BN);
  const size_t lds = 4 * TILE_BYTES;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, dim3(grid), dim3(256), lds, st, a);
  FP_CHECK_LAUNCH("gemm_bf16_kernel");
  return FP_OK;
}

}  // namespace

int gemm_bf16_launch(int epi, const GemmBf16Args& a, hipStream_t st) {
  FP_REQUIRE(a.M > 0 && a.M % BM == 0, "gemm_bf16: M (%d) must be a positive multiple of %d (pad the activation buffer)", a.M, BM);
  FP_REQUIRE(a.N > 0 && a.N % BN == 0, "gemm_bf16: N (%d) must be a multiple of %d", a.N, BN);
  FP_REQUIRE(a.K > 0 && a.K % BK == 0, "gemm_bf16: K (%d) must be a multiple of %d", a.K, BK);
  FP_REQUIRE(a.bias != nullptr, "gemm_bf16: bias is required (pass zeros)");
  FP_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0 && a.ldo % 4 == 0, "gemm_bf16: leading dims must keep 16-byte alignment");
  switch (epi) {
    case GEMM_EPI_BIAS_BF16: return launch<GEMM_EPI_BIAS_BF16>(a, st);
    case GEMM_EPI_GELU_BF16: return launch<GEMM_EPI_GELU_BF16>(a, st);
    case GEMM_EPI_QKV_BF16: return launch<GEMM_EPI_QKV_BF16>(a, st);
    case GEMM_EPI_LS_RESID_F32: return launch<GEMM_EPI_LS_RESID_F32>(a, st);
    case GEMM_EPI_TOKENS_F32: return launch<GEMM_EPI_TOKENS_F32>(a, st);
    case GEMM_EPI_BIAS_F32: return launch<GEMM_EPI_BIAS_F32>(a, st);
  }
  fp_set_error("gemm_bf16: unknown epilogue %d", epi);
  return FP_ERR_INVALID;
}
