// Fused multi-head self-attention for the DINOv2 blocks (head_dim 64, non-causal, N = 1+R+Np tokens).
//
// Replaces softmax(q k^T / sqrt(64)) v inside the backbone's Attention.forward (the reference reaches it
// through self.model(batch), /root/reference/utils/dinov2_utils.py:257; optional xformers path upstream).
//
// bf16 kernel (flash-style, one pass over the keys, fp32 online softmax):
//   * block = (128 queries, head, image), 4 waves x 32 queries; K tile and V tile (64 keys x 64 d each, rows of the
//     qkv buffer) staged through registers into LDS, next tile's loads in flight under the MFMAs, one barrier per tile
//   * S^T = K Q^T with v_mfma_f32_32x32x16_bf16 (operands swapped so a lane owns ONE query's scores:
//     row max / row sum need a single cross-lane exchange with lane^32)
//   * P -> bf16 in registers: v_cvt_pk_bf16_f32 + v_permlane32_swap builds the B operand of O^T = V^T P^T
//   * V stays row-major ([key][d], exactly as the qkv GEMM wrote it); the A operand V^T of the P*V MFMA is produced by
//     gfx950's hardware transpose read ds_read_b64_tr_b16 (two per fragment: lane (d = l&31, kh) receives keys
//     kh*8 + 4r + j of column d) from an LDS image cut into 16-column blocks.  No pre-transposed V^T copy in HBM
//     (92 MB per layer at the bench batch) and no transposing epilogue in the qkv GEMM.
// fp32 kernel (parity mode): one thread per query, K/V rows broadcast from LDS, exact expf.
#include "common.hpp"
#include "kernels.hpp"

namespace {

FP_DEVICE int swz(int row) { return (row >> 1) & 7; }

FP_DEVICE bf16x8 read_frag(const char* lds, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((chunk ^ swz(row)) << 4));
}

// V image for the transpose reads: [4 blocks of 16 d][64 keys][16 d = 32 B]; block offsets 0 / 2176 / 4416 / 6592 so
// that (a) the two blocks a ds_read_b64_tr_b16 touches are 128 B apart mod 256 (lanes 0-31 cover all 64 banks once) and
// (b) the four blocks start in different bank quarters for the 16-B staging writes.
constexpr int VTR_DT = 4416, VTR_B = 2176, VTR_BYTES = 8704;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__global__ __launch_bounds__(256, 4) void attn_bf16_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char KV[2][2][VTR_BYTES];  // [stage][K: 64 rows x 128 B swizzled | V: block image]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int qt = blockIdx.x, head = blockIdx.y, img = blockIdx.z;
  const int N = a.n_tok, D = a.dim;
  const __bf16* qkv = reinterpret_cast<const __bf16*>(a.qkv) + (size_t)img * N * a.ld_qkv;

  // Q fragments straight from global (once per block): B operand, lane holds Q[query][8 d]
  const int q = qt * 128 + wave * 32 + l31;
  const int qc = q < N ? q : N - 1;
  bf16x8 qf[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds)
    qf[ds] = *reinterpret_cast<const bf16x8*>(qkv + (size_t)qc * a.ld_qkv + head * 64 + (ds * 2 + kh) * 8);

  f32x16 oacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c = 0.125f * 1.44269504088896340736f;  // head_dim^-0.5 * log2(e)

  // staging assignment: 2 K chunks + 2 V^T chunks (16 B each) per thread (rows srow0 and srow0 + 32)
  const int srow0 = tid >> 3, sch = tid & 7;
  const int soff0 = srow0 * 128 + ((sch ^ swz(srow0)) << 4);
  const int soff1 = (srow0 + 32) * 128 + ((sch ^ swz(srow0 + 32)) << 4);
  const __bf16* kbase = qkv + D + head * 64 + sch * 8;
  const __bf16* vbase = kbase + D;
  // V staging offsets: this thread's 8 d's (chunk sch) of key srow0 / srow0 + 32 inside the block image
  const int voff0 = (sch >> 2) * VTR_DT + ((sch >> 1) & 1) * VTR_B + srow0 * 32 + (sch & 1) * 16, voff1 = voff0 + 32 * 32;
  // transpose-read base: source chunk of this lane inside a [4 keys][16 d] block + its block (b = bit 4 of the lane) + key half
  const int vrd = ((lane >> 4) & 1) * VTR_B + (kh * 8 + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
  uint4 kreg0, kreg1, vreg0, vreg1;
#define ATTN_LOAD_TILE(key0_)                                                                   \
  {                                                                                             \
    int k0_ = (key0_) + srow0, k1_ = (key0_) + srow0 + 32;                                      \
    k0_ = k0_ < N ? k0_ : N - 1;                                                                \
    k1_ = k1_ < N ? k1_ : N - 1;                                                                \
    kreg0 = *reinterpret_cast<const uint4*>(kbase + (size_t)k0_ * a.ld_qkv);                    \
    kreg1 = *reinterpret_cast<const uint4*>(kbase + (size_t)k1_ * a.ld_qkv);                    \
    vreg0 = *reinterpret_cast<const uint4*>(vbase + (size_t)k0_ * a.ld_qkv);                    \
    vreg1 = *reinterpret_cast<const uint4*>(vbase + (size_t)k1_ * a.ld_qkv);                    \
  }
#define ATTN_STORE_TILE(stage_)                                       \
  {                                                                   \
    *reinterpret_cast<uint4*>(KV[stage_][0] + soff0) = kreg0;         \
    *reinterpret_cast<uint4*>(KV[stage_][0] + soff1) = kreg1;         \
    *reinterpret_cast<uint4*>(KV[stage_][1] + voff0) = vreg0;         \
    *reinterpret_cast<uint4*>(KV[stage_][1] + voff1) = vreg1;         \
  }

  const int nkt = (N + 63) / 64;
  ATTN_LOAD_TILE(0);
  ATTN_STORE_TILE(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int key0 = kt * 64;
    const char* Ks = KV[kt & 1][0];
    const char* Vs = KV[kt & 1][1];
    if (kt + 1 < nkt) ATTN_LOAD_TILE(key0 + 64);  // next tile: HBM/L2 -> registers, lands under this tile's MFMAs

    // ---- S^T = K Q^T : sacc[ks][r] = score(query = l31, key = key0 + ks*32 + (r&3) + 8*(r>>2) + 4*kh)
    f32x16 sacc[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[ks][r] = 0.f;
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        bf16x8 kf = read_frag(Ks, ks * 32 + l31, ds * 2 + kh);
        sacc[ks] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], sacc[ks], 0, 0, 0);
      }
    }
    if (key0 + 64 > N) {  // ragged last tile: mask the padded keys
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + ks * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          if (key >= N) sacc[ks][r] = -INFINITY;
        }
    }
    // ---- online softmax (fp32). A query's 64 scores live in lanes l31 and l31+32.
    float mx = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, sacc[0][r]), sacc[1][r]);  // v_max3_f32
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    // the running max stops moving after the first few tiles: skip the rescale of O and l unless some lane needs it
    const bool grow = __any(m_new > m_run);
    float alpha = 1.f;
    if (grow) alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
    m_run = m_new;
    // exp2(s*c - m*c): scalar fp32 VALU on purpose -- v_pk_fma/v_pk_add variants measured 25 % SLOWER here
    // (packed fp32 issues badly beside MFMAs, guide "price of one filler beside MFMAs")
    float psum = 0.f;
    const float mc = m_new * c;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sacc[ks][r], c, -mc));
        sacc[ks][r] = p;
        psum += p;
      }
    if (grow) {
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
    }
    l_run += psum;

    // ---- O^T += V^T P^T over 4 steps of 16 keys
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int r0 = 8 * kk;
        unsigned a0 = pack_bf16x2(sacc[ks][r0 + 0], sacc[ks][r0 + 1]);
        unsigned a1 = pack_bf16x2(sacc[ks][r0 + 2], sacc[ks][r0 + 3]);
        unsigned b0 = pack_bf16x2(sacc[ks][r0 + 4], sacc[ks][r0 + 5]);
        unsigned b1 = pack_bf16x2(sacc[ks][r0 + 6], sacc[ks][r0 + 7]);
        auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        uint4 pw = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
        const int kstep = ks * 2 + kk;  // keys kstep*16 .. +15 of the tile
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const char* vp = Vs + vrd + dt * VTR_DT + kstep * 512;  // 16 keys x 32 B per k-step; + 128 B = 4 keys on
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vp));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vp + 128));
          const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[dt], 0, 0, 0);
        }
      }
    if (kt + 1 < nkt) ATTN_STORE_TILE((kt + 1) & 1);  // the other stage was last read one iteration ago
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  if (q < N) {
    __bf16* o = reinterpret_cast<__bf16*>(a.out) + ((size_t)img * N + q) * a.ld_out + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * kh;
        uint2 pk = make_uint2(pack_bf16x2(oacc[dt][4 * g + 0] * inv, oacc[dt][4 * g + 1] * inv),
                              pack_bf16x2(oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv));
        *reinterpret_cast<uint2*>(o + d) = pk;
      }
  }
}

// ---------------------------------------------------------------- fp32 parity-mode attention
// qkv fp32 [B*N, 3D]; one thread per query row; keys/values of the (image, head) streamed through LDS.
__global__ __launch_bounds__(256) void attn_f32_kernel(AttnArgs a) {
  __shared__ float Ks[64][64];
  __shared__ float Vs[64][64];
  const int tid = threadIdx.x;
  const int qt = blockIdx.x, head = blockIdx.y, img = blockIdx.z;
  const int N = a.n_tok, D = a.dim;
  const float* qkv = reinterpret_cast<const float*>(a.qkv) + (size_t)img * N * a.ld_qkv;
  const int q = qt * 256 + tid;
  const int qc = q < N ? q : N - 1;
  float qv[64], acc[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) {
    qv[d] = qkv[(size_t)qc * a.ld_qkv + head * 64 + d] * 0.125f;  // q * scale, like upstream
    acc[d] = 0.f;
  }
  float m_run = -INFINITY, l_run = 0.f;
  for (int key0 = 0; key0 < N; key0 += 64) {
    __syncthreads();
    for (int e = tid; e < 64 * 64; e += 256) {
      const int r = e >> 6, d = e & 63;
      int key = key0 + r;
      key = key < N ? key : N - 1;
      Ks[r][d] = qkv[(size_t)key * a.ld_qkv + D + head * 64 + d];
      Vs[r][d] = qkv[(size_t)key * a.ld_qkv + 2 * D + head * 64 + d];
    }
    __syncthreads();
    const int cnt = min(64, N - key0);
    for (int r = 0; r < cnt; ++r) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < 64; ++d) s = fmaf(qv[d], Ks[r][d], s);
      const float m_new = fmaxf(m_run, s);
      const float alpha = expf(m_run - m_new);
      const float p = expf(s - m_new);
      l_run = l_run * alpha + p;
#pragma unroll
      for (int d = 0; d < 64; ++d) acc[d] = fmaf(p, Vs[r][d], acc[d] * alpha);
      m_run = m_new;
    }
  }
  if (q < N) {
    float* o = reinterpret_cast<float*>(a.out) + ((size_t)img * N + q) * a.ld_out + head * 64;
    const float inv = 1.f / l_run;
#pragma unroll
    for (int d = 0; d < 64; ++d) o[d] = acc[d] * inv;
  }
}

}  // namespace

int attn_launch(const AttnArgs& a, int dtype, hipStream_t st) {
  FP_REQUIRE(a.dim % 64 == 0 && a.heads * 64 == a.dim, "attention: head_dim must be 64 (dim %d heads %d)", a.dim, a.heads);
  FP_REQUIRE(a.n_tok >= 1 && a.batch >= 1, "attention: empty problem");
  if (dtype == FP_DTYPE_BF16) {
    FP_REQUIRE(a.ld_qkv % 8 == 0 && a.ld_out % 4 == 0, "attention(bf16): leading dims must keep 16-byte alignment");
    dim3 grid(cdiv(a.n_tok, 128), a.heads, a.batch);
    hipLaunchKernelGGL(attn_bf16_kernel, grid, dim3(256), 0, st, a);
  } else if (dtype == FP_DTYPE_F32) {
    dim3 grid(cdiv(a.n_tok, 256), a.heads, a.batch);
    hipLaunchKernelGGL(attn_f32_kernel, grid, dim3(256), 0, st, a);
  } else {
    fp_set_error("attention: unsupported dtype %d", dtype);
    return FP_ERR_UNSUPPORTED;
  }
  FP_CHECK_LAUNCH("attention");
  return FP_OK;
}
