// Exact-fp32 MFMA tile engine: C = A[M,K] * B[N,K]^T with fused epilogues.
//
// Used by the descriptor-matching half (squared-L2 distance tiles for the visual-word
// k-NN and the per-template cyclic matching, cosine scores), the PCA projection and the
// fp32 parity mode of the ViT.  v_mfma_f32_32x32x2_f32 is an exact k-ordered fmaf chain
// (guide section 3), so with one accumulator per output and k ascending the result is
// bit-identical to oracle/csrc/oracle.cpp -- that is what makes index parity exact.
//
// Replaces: faiss IndexFlatL2.search (utils/knn_util.py:83), sklearn PCA.transform
// (utils/projector_util.py:66-69), torch cosine_similarity matmul (utils/template_util.py:167).
//
// Tile 128x128x32, 256 threads = 2x2 waves, each wave 64x64 = 2x2 MFMA tiles.
// LDS image is k-major ([k][row], stride 129 dwords): fragment reads are conflict-free
// ds_read_b32; global->LDS goes through registers so the transpose happens on the write.
#include "common.hpp"
#include "kernels.hpp"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDS_STRIDE = 129;
constexpr int STAGE_FLOATS = BK * LDS_STRIDE;  // per operand per stage

struct Frag {
  float4 v[4];
};

// 128 rows x 32 k of one operand -> 4 float4 per thread (8 lanes cover one 128-B row).
FP_DEVICE void load_tile(Frag& f, const float* __restrict__ base, int ld, int row0, int nrows, int k0, int K, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int idx = tid + i * 256;
    int row = idx >> 3, c = idx & 7;
    int k = k0 + c * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrows - row0 && k < K) {
      const float* p = base + (size_t)(row0 + row) * ld + k;
      if (k + 3 < K) {
        v = *reinterpret_cast<const float4*>(p);
      } else {  // K tail (K % 4 != 0 is rejected on the host, so this is k+3 >= K only for padding)
        v.x = p[0];
        if (k + 1 < K) v.y = p[1];
        if (k + 2 < K) v.z = p[2];
      }
    }
    f.v[i] = v;
  }
}

FP_DEVICE void store_tile(const Frag& f, float* __restrict__ lds, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int idx = tid + i * 256;
    int row = idx >> 3, c = idx & 7;
    float* p = lds + (c * 4) * LDS_STRIDE + row;
    p[0 * LDS_STRIDE] = f.v[i].x;
    p[1 * LDS_STRIDE] = f.v[i].y;
    p[2 * LDS_STRIDE] = f.v[i].z;
    p[3 * LDS_STRIDE] = f.v[i].w;
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void f32_tile_kernel(F32TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                        // [2][BK][129]
  float* Bs = smem + 2 * STAGE_FLOATS;     // [2][BK][129]

  const int pair = blockIdx.z;
  int a_off = 0, a_cnt = a.M, b_off = 0, b_cnt = a.N;
  if (a.a_seg_off) {
    int s = a.pair_a_seg ? a.pair_a_seg[pair] : (a.pair_a_div > 0 ? pair / a.pair_a_div : pair);
    a_off = a.a_seg_off[s];
    a_cnt = a.a_seg_off[s + 1] - a_off;
  }
  if (a.b_seg_off) {
    int s = a.pair_b_seg ? (int)a.pair_b_seg[pair] : pair;
    if (s < 0) return;  // empty slot
    b_off = a.b_seg_off[s];
    b_cnt = a.b_seg_off[s + 1] - b_off;
  }
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if (m0 >= a_cnt || n0 >= b_cnt) return;  // fixed max grid, ragged problems

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const float* Ab = a.A + (size_t)a_off * a.lda;
  const float* Bb = a.B + (size_t)b_off * a.ldb;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (a.K + BK - 1) / BK;
  Frag fa, fb;
  load_tile(fa, Ab, a.lda, m0, a_cnt, 0, a.K, tid);
  load_tile(fb, Bb, a.ldb, n0, b_cnt, 0, a.K, tid);
  store_tile(fa, As, tid);
  store_tile(fb, Bs, tid);
  __syncthreads();

  const int kh = lane >> 5, l31 = lane & 31;
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    if (t + 1 < nk) {
      load_tile(fa, Ab, a.lda, m0, a_cnt, (t + 1) * BK, a.K, tid);
      load_tile(fb, Bb, a.ldb, n0, b_cnt, (t + 1) * BK, a.K, tid);
    }
    const float* as = As + cur * STAGE_FLOATS + wm * 64 + l31;
    const float* bs = Bs + cur * STAGE_FLOATS + wn * 64 + l31;
#pragma unroll 4
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int krow = (kk * 2 + kh) * LDS_STRIDE;
      float a0 = as[krow], a1 = as[krow + 32];
      float b0 = bs[krow], b1 = bs[krow + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (t + 1 < nk) {
      store_tile(fa, As + (cur ^ 1) * STAGE_FLOATS, tid);
      store_tile(fb, Bs + (cur ^ 1) * STAGE_FLOATS, tid);
    }
    __syncthreads();
  }

  // ---- epilogue.  acc[tm][tn][r] is C[i][j] with
  //   i = m0 + wm*64 + tm*32 + (r&3) + 8*(r>>2) + 4*(lane>>5),  j = n0 + wn*64 + tn*32 + (lane&31)
  if constexpr (EPI == F32_EPI_DIST_ARGMIN) {
    unsigned long long* row_best = a.row_best + (size_t)pair * a.row_stride;
    unsigned long long* col_best = a.col_best + (size_t)pair * a.col_stride;
    const float* an = a.a_sqnorm + a_off;
    const float* bn = a.b_sqnorm + b_off;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      unsigned long long rbest[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rbest[r] = ~0ull;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int j = n0 + wn * 64 + tn * 32 + l31;
        const bool jv = j < b_cnt;
        const float bnj = jv ? bn[j] : 0.f;
        unsigned long long cbest = ~0ull;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          const bool iv = i < a_cnt;
          float d2 = fmaf(-2.f, acc[tm][tn][r], (iv ? an[i] : 0.f) + bnj);
          d2 = d2 < 0.f ? 0.f : d2;
          if (iv && jv) {
            unsigned long long kr = pack_dist_idx(d2, (unsigned)j);
            unsigned long long kc = pack_dist_idx(d2, (unsigned)i);
            rbest[r] = kr < rbest[r] ? kr : rbest[r];
            cbest = kc < cbest ? kc : cbest;
          }
        }
        // column j: combine the two half-waves (rows +0 / +4), one atomic per column
        unsigned long long o = __shfl_xor(cbest, 32, 64);
        cbest = o < cbest ? o : cbest;
        if (a.col_best && kh == 0 && jv && cbest != ~0ull) atomicMin(col_best + j, cbest);
      }
      // rows: min over the 32 lanes that hold different columns of the same row
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        unsigned long long v = rbest[r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
          unsigned long long t2 = __shfl_xor(v, o, 64);
          v = t2 < v ? t2 : v;
        }
        const int i = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (a.row_best && l31 == 0 && i < a_cnt && v != ~0ull) atomicMin(row_best + i, v);
      }
    }
  } else if constexpr (EPI == F32_EPI_DIST_TOPK) {
    // k nearest columns of every row INSIDE this tile (k = a.row_stride <= 8): the 128 x 128 distances go to LDS (free after
    // the main loop), two threads scan a row's two halves keeping k sorted (d2, column) keys, one shuffle merges them,
    // and the tile emits k keys per row -- 24 B instead of the 512 B of the row's distances.  A merge over the n-tiles'
    // candidates finishes the row (launch_knn_merge).  Replaces "store [m, n] distances + k selection passes over them".
    float* dt = smem;  // [128][129]
    const float* an = a.a_sqnorm + a_off;
    const float* bn = a.b_sqnorm + b_off;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int jl = wn * 64 + tn * 32 + l31, j = n0 + jl;
        const bool jv = j < b_cnt;
        const float bnj = jv ? bn[j] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int il = wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh, i = m0 + il;
          float d2 = fmaf(-2.f, acc[tm][tn][r], (i < a_cnt ? an[i] : 0.f) + bnj);
          d2 = d2 < 0.f ? 0.f : d2;
          dt[il * LDS_STRIDE + jl] = (jv && i < a_cnt) ? d2 : INFINITY;
        }
      }
    __syncthreads();
    constexpr int KMAX = 8;
    const int k = a.row_stride;
    const int row = tid >> 1, half = tid & 1;
    unsigned long long best[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) best[s] = ~0ull;
    auto insert = [&](unsigned long long key) {
#pragma unroll
      for (int s = 0; s < KMAX; ++s) {
        const unsigned long long lo = key < best[s] ? key : best[s];
        key = key < best[s] ? best[s] : key;
        best[s] = lo;
      }
    };
    for (int c = 0; c < 64; ++c) {
      const int jl = half * 64 + ((c + 32 * half) & 63);  // the halves walk 32 columns apart: different LDS banks
      const float d2 = dt[row * LDS_STRIDE + jl];
      insert(d2 == INFINITY ? ~0ull : pack_dist_idx(d2, (unsigned)(n0 + jl)));
    }
    unsigned long long other[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) other[s] = __shfl_xor(best[s], 1, 64);
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
      if (s < k) insert(other[s]);
    if (half == 0 && m0 + row < a_cnt) {
      unsigned long long* o = a.row_best + ((size_t)(m0 + row) * gridDim.x + blockIdx.x) * k;
#pragma unroll
      for (int s = 0; s < KMAX; ++s)
        if (s < k) o[s] = best[s];
    }
  } else {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int j = n0 + wn * 64 + tn * 32 + l31;
        if (j >= b_cnt) continue;
        float bnj = 0.f, bias = 0.f, gam = 1.f;
        if constexpr (EPI == F32_EPI_DIST_STORE) bnj = a.b_sqnorm[b_off + j];
        if constexpr (EPI == F32_EPI_BIAS || EPI == F32_EPI_BIAS_GELU || EPI == F32_EPI_LS_RESID || EPI == F32_EPI_TOKENS || EPI == F32_EPI_SWIGLU)
          bias = a.bias ? a.bias[j] : 0.f;
        if constexpr (EPI == F32_EPI_SUB_VEC) bias = a.bias[j];
        if constexpr (EPI == F32_EPI_LS_RESID) gam = a.gamma[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          if (i >= a_cnt) continue;
          float v = acc[tm][tn][r];
          float* o = a.out + (size_t)pair * a.out_pair_stride + (size_t)(i + (a.out_row_global ? a_off : 0)) * a.ldo + j;
          if constexpr (EPI == F32_EPI_DIST_STORE) {
            float d2 = fmaf(-2.f, v, a.a_sqnorm[a_off + i] + bnj);
            *o = d2 < 0.f ? 0.f : d2;
          } else if constexpr (EPI == F32_EPI_STORE) {
            *o = v;
          } else if constexpr (EPI == F32_EPI_SUB_VEC) {
            *o = v - bias;
          } else if constexpr (EPI == F32_EPI_BIAS) {
            *o = v + bias;
          } else if constexpr (EPI == F32_EPI_BIAS_GELU) {
            float x = v + bias;
            *o = 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
          } else if constexpr (EPI == F32_EPI_LS_RESID) {
            *o = *o + gam * (v + bias);
          } else if constexpr (EPI == F32_EPI_SWIGLU) {
            // adjacent columns (x1_j, x2_j) sit in adjacent lanes: the even lane writes silu(x1) * x2 to column j
            const float mine = v + bias;
            const float other = __shfl_xor(mine, 1, 64);
            if ((j & 1) == 0 && j + 1 < b_cnt)
              a.out[(size_t)pair * a.out_pair_stride + (size_t)i * a.ldo + (j >> 1)] = mine / (1.f + expf(-mine)) * other;
          } else if constexpr (EPI == F32_EPI_TOKENS) {
            // patch-embed: GEMM row i = b*Np + p  ->  token row b*Ntok + tok_skip + p, plus pos-embed
            const int b = i / a.tok_np, p = i - b * a.tok_np;
            float* t = a.out + ((size_t)b * a.tok_n + a.tok_skip + p) * a.ldo + j;
            *t = v + bias + a.pos[(size_t)p * a.ldo + j];
          }
        }
      }
  }
}

template <int EPI>
int launch(const F32TileArgs& a, int max_m, int max_n, int pairs, hipStream_t st) {
  dim3 grid(cdiv(max_n, BN), cdiv(max_m, BM), pairs);
  size_t lds = 4 * STAGE_FLOATS * sizeof(float);
  static FpDeviceOnce attr;
  fp_allow_dynamic_lds(attr, &f32_tile_kernel<EPI>, (int)lds);
  hipLaunchKernelGGL(f32_tile_kernel<EPI>, grid, dim3(256), lds, st, a);
  FP_CHECK_LAUNCH("f32_tile_kernel");
  return FP_OK;
}

}  // namespace

int f32_tile_launch(int epi, const F32TileArgs& a, int max_m, int max_n, int pairs, hipStream_t st) {
  FP_REQUIRE(a.K % 4 == 0 && a.lda % 4 == 0 && a.ldb % 4 == 0, "f32_tile: K, lda, ldb must be multiples of 4 (got %d %d %d)", a.K, a.lda, a.ldb);
  FP_REQUIRE(max_m > 0 && max_n > 0 && pairs > 0, "f32_tile: empty problem");
  switch (epi) {
    case F32_EPI_STORE: return launch<F32_EPI_STORE>(a, max_m, max_n, pairs, st);
    case F32_EPI_DIST_STORE: return launch<F32_EPI_DIST_STORE>(a, max_m, max_n, pairs, st);
    case F32_EPI_DIST_ARGMIN: return launch<F32_EPI_DIST_ARGMIN>(a, max_m, max_n, pairs, st);
    case F32_EPI_SUB_VEC: return launch<F32_EPI_SUB_VEC>(a, max_m, max_n, pairs, st);
    case F32_EPI_BIAS: return launch<F32_EPI_BIAS>(a, max_m, max_n, pairs, st);
    case F32_EPI_BIAS_GELU: return launch<F32_EPI_BIAS_GELU>(a, max_m, max_n, pairs, st);
    case F32_EPI_LS_RESID: return launch<F32_EPI_LS_RESID>(a, max_m, max_n, pairs, st);
    case F32_EPI_TOKENS: return launch<F32_EPI_TOKENS>(a, max_m, max_n, pairs, st);
    case F32_EPI_SWIGLU: return launch<F32_EPI_SWIGLU>(a, max_m, max_n, pairs, st);
    case F32_EPI_DIST_TOPK: return launch<F32_EPI_DIST_TOPK>(a, max_m, max_n, pairs, st);
  }
  fp_set_error("f32_tile: unknown epilogue %d", epi);
  return FP_ERR_INVALID;
}
