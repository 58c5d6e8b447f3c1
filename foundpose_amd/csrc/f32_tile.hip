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
// Tile 128x128x32, 256 threads = 4 waves, each wave up to 2x2 MFMA blocks of 32x32.  A full tile is cut 2x2 (64x64 per
// wave); a RAGGED EDGE tile (<= 64 live rows, or <= 64 live columns: segments are ragged, e.g. 517 query patches = 4 tiles
// + 5 rows) is cut 1x4 / 4x1 so that all four waves share its live blocks, and blocks with no live row or column issue no
// MFMAs at all -- which output a wave computes never changes the k-ordered chain of that output.
// LDS image is k-major ([k][row], stride 129 dwords): fragment reads are conflict-free
// ds_read_b32; global->LDS goes through registers so the transpose happens on the write.
#include <type_traits>
#include "common.hpp"
#include "kernels.hpp"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDS_STRIDE = 129;
constexpr int STAGE_FLOATS = BK * LDS_STRIDE;  // per operand per stage

struct Frag {
  float4 v[4];
};

// 128 rows x 32 k of one operand -> 4 float4 per thread (8 lanes cover one 128-B row).  Branch-free: a position outside the
// operand (row past the segment, k past K -- K % 4 == 0 is checked on the host) loads a clamped address and is zeroed, so
// the four loads always issue back to back and nothing waits between them.
FP_DEVICE void load_tile(Frag& f, const float* __restrict__ base, int ld, int row0, int nrows, int k0, int K, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + i * 256;
    const int row = row0 + (idx >> 3), k = k0 + (idx & 7) * 4;
    const bool ok = row < nrows && k < K;
    const float4 v = *reinterpret_cast<const float4*>(base + (size_t)(row < nrows ? row : nrows - 1) * ld + (k < K ? k : K - 4));
    f.v[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
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

// Per-tile k-selection of the DIST_TOPK epilogue: two threads scan the two 64-column halves of a row of the tile's distance
// image keeping KMAX sorted (d2, column) keys, one shuffle merges the halves, thread `half == 0` emits k keys.  Columns past
// the live part of the tile are never turned into keys.
template <int KMAX>
FP_DEVICE void topk_insert(unsigned long long (&best)[KMAX], unsigned long long key) {
#pragma unroll
  for (int s = 0; s < KMAX; ++s) {  // one compare per slot: the smaller key stays, the larger moves on
    const bool lt = key < best[s];
    const unsigned long long lo = lt ? key : best[s], hi = lt ? best[s] : key;
    best[s] = lo;
    key = hi;
  }
}

template <int KMAX>
FP_DEVICE void tile_topk_rows(const float* dt, int k, int tid, int n0, int live_n, bool row_live, unsigned long long* out) {
  const int row = tid >> 1, half = tid & 1;
  // two independent sorted lists per thread (even / odd steps of the scan): the insertion is a chain of dependent
  // compare-selects, two chains in flight hide each other's latency; they merge at the end
  unsigned long long best[KMAX], bestb[KMAX];
#pragma unroll
  for (int s = 0; s < KMAX; ++s) { best[s] = ~0ull; bestb[s] = ~0ull; }
  if (half * 64 < live_n) {  // a half with no live column keeps its empty list
    for (int c = 0; c < 64; c += 2) {
      const int ja = half * 64 + ((c + 32 * half) & 63), jb = half * 64 + ((c + 1 + 32 * half) & 63);  // the halves walk 32 columns apart: different LDS banks
      const float da = dt[row * LDS_STRIDE + ja], db = dt[row * LDS_STRIDE + jb];
      topk_insert<KMAX>(best, ja >= live_n ? ~0ull : pack_dist_idx(da, (unsigned)(n0 + ja)));
      topk_insert<KMAX>(bestb, jb >= live_n ? ~0ull : pack_dist_idx(db, (unsigned)(n0 + jb)));
    }
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
      if (s < k) topk_insert<KMAX>(best, bestb[s]);
  }
  unsigned long long other[KMAX];
#pragma unroll
  for (int s = 0; s < KMAX; ++s) other[s] = __shfl_xor(best[s], 1, 64);
#pragma unroll
  for (int s = 0; s < KMAX; ++s)
    if (s < k) topk_insert<KMAX>(best, other[s]);
  if (half == 0 && row_live) {
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
      if (s < k) out[s] = best[s];
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
    if (a.pair_b_base) s += a.pair_b_base[a.pair_a_div > 0 ? pair / a.pair_a_div : pair];  // segment ids local to the A segment's group (object)
    b_off = a.b_seg_off[s];
    b_cnt = a.b_seg_off[s + 1] - b_off;
  }
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if (m0 >= a_cnt || n0 >= b_cnt) return;  // fixed max grid, ragged problems
  // (a 64-column tile pitch for launches of ~1 tile per CU -- PCA: 260 tiles on 256 CUs -- was measured: 196 vs 158 us, the
  //  half tiles' extra staging costs more than the shorter tail returns)

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave -> its (up to) 2x2 blocks of 32x32: rows rb[tm], columns cb[tn] inside the tile; lm / ln = block has live rows / columns
  const int live_m = a_cnt - m0 < BM ? a_cnt - m0 : BM, live_n = b_cnt - n0 < BN ? b_cnt - n0 : BN;
  const int layout = live_m <= 64 ? 1 : (live_n <= 64 ? 2 : 0);
  int rb[2], cb[2];
  bool lm[2], ln[2];
  if (layout == 0) {        // 2 x 2 waves of 64 x 64
    rb[0] = (wave >> 1) * 64; rb[1] = rb[0] + 32; cb[0] = (wave & 1) * 64; cb[1] = cb[0] + 32;
    lm[0] = rb[0] < live_m; lm[1] = rb[1] < live_m; ln[0] = cb[0] < live_n; ln[1] = cb[1] < live_n;
  } else if (layout == 1) { // <= 64 live rows: 1 x 4 waves of 64 rows x 32 columns
    rb[0] = 0; rb[1] = 32; cb[0] = wave * 32; cb[1] = cb[0];
    lm[0] = true; lm[1] = 32 < live_m; ln[0] = cb[0] < live_n; ln[1] = false;
  } else {                  // <= 64 live columns: 4 x 1 waves of 32 rows x 64 columns
    rb[0] = wave * 32; rb[1] = rb[0]; cb[0] = 0; cb[1] = 32;
    lm[0] = rb[0] < live_m; lm[1] = false; ln[0] = true; ln[1] = 32 < live_n;
  }
  const bool l00 = lm[0] && ln[0], l01 = lm[0] && ln[1], l10 = lm[1] && ln[0], l11 = lm[1] && ln[1];
  const float* Ab = a.A + (size_t)a_off * a.lda;
  const float* Bb = a.B + (size_t)b_off * a.ldb;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // squared norms of the tile's rows and columns for the distance epilogues: one coalesced load each, parked behind the
  // stage buffers; the first barrier of the K loop publishes them
  float* nrm = smem + 4 * STAGE_FLOATS;  // [128 rows | 128 columns]
  if constexpr (EPI == F32_EPI_DIST_STORE || EPI == F32_EPI_DIST_ARGMIN || EPI == F32_EPI_DIST_TOPK) {
    const int x = threadIdx.x & 127;
    const float dead = EPI == F32_EPI_DIST_ARGMIN ? INFINITY : 0.f;  // argmin: a dead row / column never wins a comparison
    if (threadIdx.x < 128) nrm[x] = m0 + x < a_cnt ? a.a_sqnorm[a_off + m0 + x] : dead;
    else nrm[128 + x] = n0 + x < b_cnt ? a.b_sqnorm[b_off + n0 + x] : dead;
  }
  const int nk = (a.K + BK - 1) / BK;
  Frag fa, fb;
  load_tile(fa, Ab, a.lda, m0, a_cnt, 0, a.K, tid);
  load_tile(fb, Bb, a.ldb, n0, b_cnt, 0, a.K, tid);
  store_tile(fa, As, tid);
  store_tile(fb, Bs, tid);
  __syncthreads();

  const int kh = lane >> 5, l31 = lane & 31;
  // K loop in two instantiations chosen per WORKGROUP (so every wave meets the same barriers in the same code): FULL = all
  // 128 x 128 positions live (2 x 2 cut, no guards, the next k-pair's fragments are read before this one's MFMAs issue);
  // otherwise the guarded form, where wave-uniform flags skip the dead blocks.
  auto k_loop = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    for (int t = 0; t < nk; ++t) {
      const int cur = t & 1;
#ifndef FP_F32_NO_STAGE
      if (t + 1 < nk) {
        load_tile(fa, Ab, a.lda, m0, a_cnt, (t + 1) * BK, a.K, tid);
        load_tile(fb, Bb, a.ldb, n0, b_cnt, (t + 1) * BK, a.K, tid);
      }
#endif
      const float* as0 = As + cur * STAGE_FLOATS + rb[0] + l31 + kh * LDS_STRIDE;
      const float* bs0 = Bs + cur * STAGE_FLOATS + cb[0] + l31 + kh * LDS_STRIDE;
      if constexpr (FULL) {
        float a0 = as0[0], a1 = as0[32], b0 = bs0[0], b1 = bs0[32];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
          float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
          if (kk + 1 < BK / 2) {
            const int krow = (kk + 1) * 2 * LDS_STRIDE;
            na0 = as0[krow]; na1 = as0[krow + 32];
            nb0 = bs0[krow]; nb1 = bs0[krow + 32];
          }
          __builtin_amdgcn_sched_barrier(0);  // keep the reads above the MFMAs (the scheduler sinks them to their first use)
#ifndef FP_F32_NO_MFMA  // (measurement builds: tools/f32_ablate.sh)
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
#else
          acc[0][0][kk] += a0 * b0 + a1 * b1;
#endif
          __builtin_amdgcn_sched_barrier(0);
          a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
      } else if (l00) {
        const int da = rb[1] - rb[0], db = cb[1] - cb[0];  // 32 or 0 (a dead second block re-reads the first, unused)
        float a0 = as0[0], a1 = as0[da], b0 = bs0[0], b1 = bs0[db];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
          float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
          if (kk + 1 < BK / 2) {
            const int krow = (kk + 1) * 2 * LDS_STRIDE;
            na0 = as0[krow]; na1 = as0[krow + da];
            nb0 = bs0[krow]; nb1 = bs0[krow + db];
          }
          __builtin_amdgcn_sched_barrier(0);
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
          if (l01) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
          if (l10) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
          if (l11) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
      }
#ifndef FP_F32_NO_STAGE
      if (t + 1 < nk) {
        store_tile(fa, As + (cur ^ 1) * STAGE_FLOATS, tid);
        store_tile(fb, Bs + (cur ^ 1) * STAGE_FLOATS, tid);
      }
      __syncthreads();
#endif
    }
  };
  if (live_m == BM && live_n == BN) k_loop(std::true_type{});
  else k_loop(std::false_type{});

#ifdef FP_F32_NO_EPI
  if (a.K > 0) {
    if (acc[0][0][0] + acc[0][1][1] + acc[1][0][2] + acc[1][1][3] == 12345.678f) a.out[0] = 1.f;
    return;
  }
#endif
  // ---- epilogue.  acc[tm][tn][r] is C[i][j] with
  //   i = m0 + rb[tm] + (r&3) + 8*(r>>2) + 4*(lane>>5),  j = n0 + cb[tn] + (lane&31)      (block live iff lm[tm] && ln[tn])
  if constexpr (EPI == F32_EPI_DIST_ARGMIN) {
    // Nearest column of every row and nearest row of every column of this tile, merged into the pair's row_best / col_best
    // tables with ONE coalesced u64 atomicMin per row and per column of the tile: every wave reduces its blocks in registers
    // (rows: a 16-exchange reduce-scatter over the 32 lanes that hold a row's columns), parks the results in its own LDS slot,
    // and after a barrier thread t owns row t (t < 128) or column t - 128.
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(smem);  // [rows | cols][wave][128]; the stages are free now
    for (int x = lane; x < 128; x += 64) {
      slots[wave * 128 + x] = ~0ull;
      slots[(4 + wave) * 128 + x] = ~0ull;
    }
    // In a lane the candidates of a row (columns) and of a column (rows) arrive in ascending index order, so a strict float
    // "<" keeps the lowest index among equal distances -- the u64 (d2, index) keys are only built for the cross-lane merges.
    // Dead rows / columns carry an infinite norm (below): their distances never win.
    float cd[2] = {INFINITY, INFINITY};
    unsigned ci[2] = {~0u, ~0u};
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      if (!lm[tm]) continue;
      float rd[16], an_r[16];
      unsigned ri[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        rd[r] = INFINITY;
        ri[r] = ~0u;
        an_r[r] = nrm[rb[tm] + (r & 3) + 8 * (r >> 2) + 4 * kh];
      }
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        if (!ln[tn]) continue;
        const int jl = cb[tn] + l31;
        const unsigned j = (unsigned)(n0 + jl);
        const float bnj = nrm[128 + jl];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned i = (unsigned)(m0 + rb[tm] + (r & 3) + 8 * (r >> 2) + 4 * kh);
          float d2 = fmaf(-2.f, acc[tm][tn][r], an_r[r] + bnj);
          d2 = d2 < 0.f ? 0.f : d2;
          const bool br = d2 < rd[r], bc = d2 < cd[tn];
          rd[r] = br ? d2 : rd[r];
          ri[r] = br ? j : ri[r];
          cd[tn] = bc ? d2 : cd[tn];
          ci[tn] = bc ? i : ci[tn];
        }
      }
      unsigned long long rbest[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rbest[r] = ri[r] == ~0u ? ~0ull : pack_dist_idx(rd[r], ri[r]);
      // rows: reduce-scatter over lane bits 4..1 (each step a lane keeps half of its rows and sends the other half to its
      // partner), then one exchange over bit 0: lane l ends with row r = bits (4,3,2,1) of l31, min over the 32 columns
#pragma unroll
      for (int step = 0; step < 4; ++step) {
        const int half = 8 >> step;            // rows kept after this step
        // keep / send by a masked xor swap on the VALUES: written as `up ? rbest[x + half] : rbest[x]` the compiler turns the
        // pair of selects into one dynamically indexed read of the register array -- a 16-way v_cmp_eq / v_cndmask chain per
        // access, 2350 v_cndmask in the kernel and 59 of the epilogue's 86 us
        const unsigned long long swap = ((l31 >> (4 - step)) & 1) ? ~0ull : 0ull;
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          if (x < half) {
            const unsigned long long d = (rbest[x] ^ rbest[x + half]) & swap;
            const unsigned long long keep = rbest[x] ^ d, send = rbest[x + half] ^ d;
            const unsigned long long got = __shfl_xor(send, 16 >> step, 64);
            rbest[x] = got < keep ? got : keep;
          }
        }
      }
      {
        const unsigned long long got = __shfl_xor(rbest[0], 1, 64);
        rbest[0] = got < rbest[0] ? got : rbest[0];
      }
      if ((l31 & 1) == 0) {
        const int r = l31 >> 1;
        slots[wave * 128 + rb[tm] + (r & 3) + 8 * (r >> 2) + 4 * kh] = rbest[0];
      }
    }
    // columns: the two half-waves hold rows +0 / +4 of the same column
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      if (!ln[tn]) continue;
      const unsigned long long mine = ci[tn] == ~0u ? ~0ull : pack_dist_idx(cd[tn], ci[tn]);
      const unsigned long long o = __shfl_xor(mine, 32, 64);
      const unsigned long long v = o < mine ? o : mine;
      if (kh == 0) slots[(4 + wave) * 128 + cb[tn] + l31] = v;
    }
    __syncthreads();
    {
      const int x = tid & 127, side = tid >> 7;
      const unsigned long long* sl = slots + side * 512 + x;
      unsigned long long v = sl[0];
#pragma unroll
      for (int w = 1; w < 4; ++w) v = sl[w * 128] < v ? sl[w * 128] : v;
      unsigned long long* dst = side ? a.col_best : a.row_best;
      if (dst && a.best_parts) {  // this tile's slice of the partial tables: plain coalesced stores, nothing to preset, no contention
        const size_t part = side ? (size_t)pair * gridDim.y + blockIdx.y : (size_t)pair * gridDim.x + blockIdx.x;
        const int stride = side ? a.col_stride : a.row_stride, pos = (side ? n0 : m0) + x;
        if (pos < stride) dst[part * stride + pos] = v;
      } else if (dst && v != ~0ull) {
        atomicMin(dst + (size_t)pair * (side ? a.col_stride : a.row_stride) + (side ? n0 : m0) + x, v);
      }
    }
  } else if constexpr (EPI == F32_EPI_DIST_TOPK) {
    // k nearest columns of every row INSIDE this tile (k = a.row_stride <= 8): the 128 x 128 distances go to LDS (free after
    // the main loop), two threads scan a row's two halves keeping k sorted (d2, column) keys, one shuffle merges them,
    // and the tile emits k keys per row -- 24 B instead of the 512 B of the row's distances.  A merge over the n-tiles'
    // candidates finishes the row (launch_knn_merge).  Replaces "store [m, n] distances + k selection passes over them".
    float* dt = smem;  // [128][129]
    // (all norm reads first: dt aliases the stage buffers in the compiler's eyes, a norm read after a dt store would wait for it)
    float an_r[2][16], bn_c[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bn_c[t] = nrm[128 + cb[t] + l31];
#pragma unroll
      for (int r = 0; r < 16; ++r) an_r[t][r] = nrm[rb[t] + (r & 3) + 8 * (r >> 2) + 4 * kh];
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        if (!(lm[tm] && ln[tn])) continue;
        const int jl = cb[tn] + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int il = rb[tm] + (r & 3) + 8 * (r >> 2) + 4 * kh;
          const float d2 = fmaf(-2.f, acc[tm][tn][r], an_r[tm][r] + bn_c[tn]);
          dt[il * LDS_STRIDE + jl] = d2 < 0.f ? 0.f : d2;
        }
      }
    __syncthreads();
    // (positions in rows >= a_cnt or columns >= live_n hold whatever was there: such rows emit nothing and such columns are
    //  never turned into keys)
    const int k = a.row_stride;
    unsigned long long* o = a.row_best + ((size_t)(m0 + (tid >> 1)) * gridDim.x + blockIdx.x) * k;
    const bool row_live = m0 + (tid >> 1) < a_cnt;
    if (k == 3) tile_topk_rows<3>(dt, k, tid, n0, live_n, row_live, o);   // the visual-word assignment of the tf-idf descriptors
    else if (k <= 4) tile_topk_rows<4>(dt, k, tid, n0, live_n, row_live, o);
    else tile_topk_rows<8>(dt, k, tid, n0, live_n, row_live, o);
  } else {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        if (!(lm[tm] && ln[tn])) continue;
        const int j = n0 + cb[tn] + l31;
        if (j >= b_cnt) continue;
        float bnj = 0.f, bias = 0.f, gam = 1.f;
        if constexpr (EPI == F32_EPI_DIST_STORE) bnj = nrm[128 + cb[tn] + l31];
        if constexpr (EPI == F32_EPI_BIAS || EPI == F32_EPI_BIAS_GELU || EPI == F32_EPI_LS_RESID || EPI == F32_EPI_TOKENS || EPI == F32_EPI_SWIGLU)
          bias = a.bias ? a.bias[j] : 0.f;
        if constexpr (EPI == F32_EPI_SUB_VEC) bias = a.bias[j];
        if constexpr (EPI == F32_EPI_LS_RESID) gam = a.gamma[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = m0 + rb[tm] + (r & 3) + 8 * (r >> 2) + 4 * kh;
          if (i >= a_cnt) continue;
          float v = acc[tm][tn][r];
          float* o = a.out + (size_t)pair * a.out_pair_stride + (size_t)(i + (a.out_row_global ? a_off : 0)) * a.ldo + j;
          if constexpr (EPI == F32_EPI_DIST_STORE) {
            float d2 = fmaf(-2.f, v, nrm[i - m0] + bnj);
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
  size_t lds = (4 * STAGE_FLOATS + 2 * BM) * sizeof(float);  // two double-buffered operand stages + the tile's row / column norms
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
