// Exact L2 nearest neighbours in two stages: an fp16-MFMA candidate pass with a DERIVED error bound, then the exact fp32 chain on
// the candidates only -- same outputs, bit for bit, as the all-pairs exact-fp32 tile (f32_tile.hip, DIST_TOPK / DIST_ARGMIN).
//
// Replaces, like that tile: faiss IndexFlatL2.search for the visual words (/root/reference/utils/template_util.py:18 through
// utils/knn_util.py:83; k = tfidf_knn_k = 3) and the two 1-NN searches of cyclic_buddies_matching
// (/root/reference/utils/corresp_util.py:46-47: template index searched with the query patches, query index searched with the
// template patches).  The exact-fp32 MFMA runs at 1/16 of the fp16 rate; the three all-pairs launches were 88 % of the matching
// stage (profiles/r3_final_per_step_kernels.csv).
//
// Arithmetic that defines the result (unchanged): dot(x, y) = k-ascending fp32 fma chain, d2 = max(0, fma(-2, dot, |x|^2 + |y|^2)),
// neighbours ordered by (d2, index).
//
// Stage 1 (knn_cand_kernel): x and y are rounded to fp16 on their way into the MFMA (round to nearest even), fp32 accumulate.
//   score s = dot~ - |y|^2 / 2  (larger = nearer; the unclamped approximate distance is |x|^2 - 2 s).
//   Bound, with X = |x|, Y = the largest |y| of the database segment, K = the dimension:
//     |dot~ - dot| <= eps = [ 2^-10 (1 + 1.25 K 2^-12) X Y     operand rounding (2 x 2^-11, both sides) + MFMA / chain accumulation
//                           + sqrt(K) 2^-24 (X + Y) + K 2^-48    elements in the fp16 subnormal range (absolute error 2^-25 each)
//                           + 2^-21 (X^2 + Y^2) ] x 1.01         the roundings of forming d2 and s themselves
//   A lane keeps the k best scores it has seen (s_k the k-th) and emits every database row with s >= min(s_k, |x|^2 / 2) - 2 eps.
//   Superset: the k rows with the best approximate scores have exact distances <= max(0, T) + 2 eps' (T = their k-th approximate
//   distance, eps' = 2 eps), so the exact k-th distance is <= that; a true neighbour y has d2(y) <= exact k-th, hence an approximate
//   distance <= max(0, T) + 2 eps', i.e. s(y) >= min(s_k, |x|^2 / 2) - 2 eps -- and the running s_k only rises, so the test at the time
//   a row is seen is weaker than the final one.  Operands beyond the fp16 range (|v| > 65504, NaN) and lists that overflow mark the
//   row: stage 2 then computes that row by brute force (exact chains against the whole segment).
// Stage 2 (knn_rescore_kernel): 16 lanes per row, a lane per candidate: the exact chain, then the k smallest (d2, index) keys.
#include "common.hpp"
#include "kernels.hpp"

namespace {

constexpr int KC_ROWS = 128;  // rows per workgroup: four waves x one 32-row MFMA block, all four sharing every staged database tile
constexpr int KC_LISTS = 8;   // partial candidate lists per row: (database split <= 4, k-half of the MFMA output layout)
constexpr int KC_SPLIT = 512; // database rows per split: a launch covers the database with up to 4 workgroups per row block
constexpr int kc_cap(int k) { return k == 1 ? 6 : 10; }  // entries per partial list

struct KcProblem {
  int x_off, x_cnt, y_off, y_cnt;
  const float *X, *Y, *xs, *ys;
  bool live;
};

// pair -> (row segment, database segment), the segment logic of F32TileArgs; swap exchanges the roles of the two sides
FP_DEVICE KcProblem kc_problem(const KnnCandArgs& a, int pair) {
  int a_off = 0, a_cnt = a.M, b_off = 0, b_cnt = a.N;
  bool live = true;
  if (a.a_seg_off) {
    const int s = a.pair_a_div > 0 ? pair / a.pair_a_div : pair;
    a_off = a.a_seg_off[s];
    a_cnt = a.a_seg_off[s + 1] - a_off;
  }
  if (a.b_seg_off) {
    int s = a.pair_b_seg ? a.pair_b_seg[pair] : pair;
    if (s < 0) {
      live = false;
      s = 0;
    } else if (a.pair_b_base) {
      s += a.pair_b_base[a.pair_a_div > 0 ? pair / a.pair_a_div : pair];
    }
    b_off = a.b_seg_off[s];
    b_cnt = a.b_seg_off[s + 1] - b_off;
  }
  KcProblem p;
  p.live = live && a_cnt > 0 && b_cnt > 0;
  if (!a.swap) {
    p.x_off = a_off; p.x_cnt = a_cnt; p.y_off = b_off; p.y_cnt = b_cnt;
    p.X = a.A; p.Y = a.B; p.xs = a.a_sqn; p.ys = a.b_sqn;
  } else {
    p.x_off = b_off; p.x_cnt = b_cnt; p.y_off = a_off; p.y_cnt = a_cnt;
    p.X = a.B; p.Y = a.A; p.xs = a.b_sqn; p.ys = a.a_sqn;
  }
  return p;
}

// database rows [lo, hi) of split `sp` when the segment is cut into `ns` splits of whole 64-row staging steps
FP_DEVICE void kc_split_range(int y_cnt, int ns, int sp, int& lo, int& hi) {
  const int steps = (y_cnt + 63) >> 6, per = (steps + ns - 1) / ns;
  lo = sp * per * 64;
  hi = (sp + 1) * per * 64;
  lo = lo < y_cnt ? lo : y_cnt;
  hi = hi < y_cnt ? hi : y_cnt;
}

FP_DEVICE unsigned pack_f16x2_rne(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, f16x2));
}
FP_DEVICE bool beyond_f16(float4 v) {  // true for |v| > 65504 and for NaN (the comparison is false)
  return !(fabsf(v.x) <= 65504.f && fabsf(v.y) <= 65504.f && fabsf(v.z) <= 65504.f && fabsf(v.w) <= 65504.f);
}

// Stage 1.  grid (row blocks of 128, pairs, database splits).  A workgroup keeps its 128 rows as MFMA B fragments in registers (a wave: 32
// rows) and streams its share of the database through LDS, 64 rows per step (fp32 -> fp16 on the way), two MFMA tiles per wave and step.
template <int KMAX, int K>
__global__ __launch_bounds__(256, 2) void knn_cand_kernel(KnnCandArgs a) {
  constexpr int CAP = kc_cap(KMAX);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ksteps = K >> 4;
  constexpr int RS = K * 2 + 16;                    // bytes per staged database row: + 16 keeps 16 lanes x ds_read_b128 on 64 distinct banks
  char* ybuf = smem;                                // [2 stages][64 rows][RS]
  float* ynh = reinterpret_cast<float*>(smem + 2 * 64 * RS);   // [2][64]  -|y|^2 / 2 of the staged rows (-inf: past the segment)
  float* red = ynh + 128;                           // [4] block reduction, [4] = bad-database flag
  unsigned long long* lds_lists = reinterpret_cast<unsigned long long*>(red + 8);   // [256 threads][CAP]: this thread's candidate list
  const int pair = blockIdx.y, sp = blockIdx.z;
  const KcProblem p = kc_problem(a, pair);
  const int r0 = blockIdx.x * KC_ROWS;
  if (!p.live || r0 >= p.x_cnt) return;
  int y_lo, y_hi;
  kc_split_range(p.y_cnt, gridDim.z, sp, y_lo, y_hi);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  const int row = r0 + wave * 32 + l31;
  const bool row_live = row < p.x_cnt;
  const size_t row_slot = (size_t)pair * a.row_stride + (row_live ? row : 0);
  int2* cnt_out = reinterpret_cast<int2*>(a.counts) + row_slot * KC_LISTS + sp * 2 + kh;
  if (y_lo >= y_hi) {                               // (block-uniform) an empty split of a short segment: empty lists, weakest threshold
    if (row_live) *cnt_out = make_int2(0, __float_as_int(-INFINITY));
    return;
  }

  // ---- the largest squared norm of the database segment (the whole segment: one bound for all splits)
  float ym = 0.f;
  for (int j = tid; j < p.y_cnt; j += 256) ym = fmaxf(ym, p.ys[p.y_off + j]);
  ym = wave_max(ym);
  if (lane == 0) red[wave] = ym;
  if (tid == 0) red[4] = 0.f;
  __syncthreads();
  const float ymax2 = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));

  // ---- this wave's 32 rows as MFMA B fragments, kept in registers for the whole kernel: lane (row l31, kh) holds k = 16 s + 8 kh .. + 7
  f16x8 qf[ksteps];
  bool bad_x = false;
  {
    const float* xr = p.X + (size_t)(p.x_off + (row_live ? row : p.x_cnt - 1)) * a.ld + 8 * kh;
#pragma unroll
    for (int s = 0; s < ksteps; ++s) {
      const float4 v0 = *reinterpret_cast<const float4*>(xr + 16 * s), v1 = *reinterpret_cast<const float4*>(xr + 16 * s + 4);
      bad_x |= beyond_f16(v0) | beyond_f16(v1);
      qf[s] = __builtin_bit_cast(f16x8, make_uint4(pack_f16x2_rne(v0.x, v0.y), pack_f16x2_rne(v0.z, v0.w), pack_f16x2_rne(v1.x, v1.y), pack_f16x2_rne(v1.z, v1.w)));
    }
    bad_x |= (bool)__shfl_xor((int)bad_x, 32, 64);
  }
  const float qn = row_live ? p.xs[p.x_off + row] : 0.f;
  const float xn = sqrtf(qn), yn = sqrtf(ymax2);
  constexpr float fk = (float)K;
  const float eps = (0x1p-10f * (1.f + 1.25f * fk * 0x1p-12f) * xn * yn + sqrtf(fk) * 0x1p-24f * (xn + yn) + fk * 0x1p-48f + 0x1p-21f * (qn + ymax2)) * 1.01f;
  const float win = 2.f * eps, s_cap = 0.5f * qn;

  float best[KMAX];
#pragma unroll
  for (int s = 0; s < KMAX; ++s) best[s] = -INFINITY;
  int cnt = 0;
  bool over = false;
  float thr = -INFINITY;
  unsigned long long* my = lds_lists + tid * CAP;    // (in LDS: a list is re-read when it fills -- from global memory that cost ~5 k cycles a time)

  // ---- staging: 64 database rows per step, fp32 -> fp16 on the way, in two half-steps so that only nld / 2 loads are in flight beside the
  //      query-fragment registers: the first half flies under the MFMAs, the second under the selection epilogue
  const int ntp = (y_hi - y_lo + 63) >> 6;
  constexpr int f4_per_row = K >> 2, nld = K >> 4, HL = nld / 2;   // float4 per thread and step = 64 K / 4 / 256
  float4 pre[HL];
  bool bad_y = false;
  auto load = [&](int tp, int h) {
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      const int idx = tid + (h * HL + i) * 256, r = idx / f4_per_row, c = idx - r * f4_per_row, j = y_lo + tp * 64 + r;
      // a row past the split loads the segment's last row (its score is forced to -inf through ynh and it is never emitted): no select on
      // the loaded value -- `cond ? load : 0` compiled to a branch and a wait per load, i.e. 16 serialized L2 round trips per step
      pre[i] = *reinterpret_cast<const float4*>(p.Y + (size_t)(p.y_off + (j < y_hi ? j : p.y_cnt - 1)) * a.ld + c * 4);
    }
  };
  auto store = [&](int tp, int buf, int h) {
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      const int idx = tid + (h * HL + i) * 256, r = idx / f4_per_row, c = idx - r * f4_per_row;
      bad_y |= beyond_f16(pre[i]);
      *reinterpret_cast<uint2*>(ybuf + (buf * 64 + r) * RS + c * 8) = make_uint2(pack_f16x2_rne(pre[i].x, pre[i].y), pack_f16x2_rne(pre[i].z, pre[i].w));
    }
    if (h == 0 && tid < 64) {
      const int j = y_lo + tp * 64 + tid;
      ynh[buf * 64 + tid] = j < y_hi ? -0.5f * p.ys[p.y_off + j] : -INFINITY;
    }
  };
  load(0, 0);
  store(0, 0, 0);
  load(0, 1);
  store(0, 0, 1);
  __syncthreads();
  for (int tp = 0; tp < ntp; ++tp) {
    const int buf = tp & 1;
    if (tp + 1 < ntp) load(tp + 1, 0);
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {                  // the two 32-row tiles of the step
      const int j0 = y_lo + tp * 64 + tl * 32;
      if (j0 < y_hi) {                                // (block-uniform)
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const char* yrow = ybuf + (buf * 64 + tl * 32 + l31) * RS + 16 * kh;
#pragma unroll
        for (int s = 0; s < ksteps; ++s)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(yrow + 32 * s), qf[s], acc, 0, 0, 0);
        // acc[r] = dot~(database row j0 + m, this lane's row), m = (r & 3) + 8 (r >> 2) + 4 kh
        const float* nh = ynh + buf * 64 + tl * 32 + 4 * kh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float sc = acc[r] + nh[(r & 3) + 8 * (r >> 2)];
          acc[r] = sc;
          float v = sc;
#pragma unroll
          for (int t = 0; t < KMAX; ++t) {            // descending insertion: the larger stays, the smaller moves on
            const float hi = fmaxf(best[t], v), lo = fminf(best[t], v);
            best[t] = hi;
            v = lo;
          }
        }
        // the row's k-th best score so far: this lane's and its k-half partner's (the other 16 rows of every tile)
        const float kb = fmaxf(best[KMAX - 1], __shfl_xor(best[KMAX - 1], 32, 64));
        thr = fminf(kb, s_cap) - win;
        // hits of this tile as a 16-bit mask (branch-free), then ONE short wave-uniform loop over the set bits: on unstructured data some
        // lane of the 64 has a hit for nearly every r, and a branch per r ran the emission body 14 times per tile
        unsigned hits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) hits |= (acc[r] >= thr ? 1u : 0u) << r;
        if (j0 + 32 > y_hi) {                         // (block-uniform: the split's last tile) rows past it never count
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (j0 + (r & 3) + 8 * (r >> 2) + 4 * kh >= y_hi) hits &= ~(1u << r);
        }
        if (!row_live) hits = 0;
        while (__builtin_amdgcn_ballot_w64(hits != 0)) {
          if (hits) {
            const int r = __builtin_ctz(hits);
            hits &= hits - 1;
            float sc = acc[0];
#pragma unroll
            for (int q = 1; q < 16; ++q) sc = r == q ? acc[q] : sc;
            if (cnt == CAP) {                         // full: the threshold has only risen since the entries were written -- drop those it has passed
              int w = 0;
              for (int e = 0; e < CAP; ++e) {
                const unsigned long long v = my[e];
                if (__uint_as_float((unsigned)(v >> 32)) >= thr) my[w++] = v;
              }
              cnt = w;
            }
            if (cnt < CAP) my[cnt++] = ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned)(j0 + (r & 3) + 8 * (r >> 2) + 4 * kh);
            else over = true;                         // more rows inside the window than a list holds: stage 2 takes the row by brute force
          }
        }
      }
      if (tl == 0 && tp + 1 < ntp) {
        store(tp + 1, buf ^ 1, 0);
        load(tp + 1, 1);
      }
    }
    if (tp + 1 < ntp) store(tp + 1, buf ^ 1, 1);
    __syncthreads();
  }
  if (bad_y) red[4] = 1.f;
  __syncthreads();
  if (row_live) {
    const bool brute = over || bad_x || red[4] != 0.f;
    // (count, the list's final threshold): stage 2 re-scores only the entries at or above the LARGEST of the row's thresholds
    *cnt_out = make_int2(brute ? -1 : cnt, __float_as_int(thr));
    unsigned long long* out = a.lists + (row_slot * KC_LISTS + sp * 2 + kh) * CAP;
    if (!brute)
      for (int e = 0; e < cnt; ++e) out[e] = my[e];
  }
}

// The exact key of (x, y): k-ascending fma chain, one accumulator -- the arithmetic of the all-pairs exact tile.  The chain is 256 dependent
// fmas; the next 16 elements of both rows are loaded while the current 16 are consumed.
FP_DEVICE unsigned long long kc_exact_key(const float* __restrict__ x, const float* __restrict__ y, int K, float qn, float yn, int j) {
  float acc = 0.f;
  float4 xa[4], ya[4], xb[4], yb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    xa[i] = *reinterpret_cast<const float4*>(x + 4 * i);
    ya[i] = *reinterpret_cast<const float4*>(y + 4 * i);
  }
  for (int k = 0; k < K; k += 16) {
    const int kn = k + 16 < K ? k + 16 : k;           // (the last step re-loads its own chunk: no branch in the loop)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xb[i] = *reinterpret_cast<const float4*>(x + kn + 4 * i);
      yb[i] = *reinterpret_cast<const float4*>(y + kn + 4 * i);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc = fmaf(xa[i].x, ya[i].x, acc);
      acc = fmaf(xa[i].y, ya[i].y, acc);
      acc = fmaf(xa[i].z, ya[i].z, acc);
      acc = fmaf(xa[i].w, ya[i].w, acc);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xa[i] = xb[i];
      ya[i] = yb[i];
    }
  }
  float d2 = fmaf(-2.f, acc, qn + yn);
  d2 = d2 < 0.f ? 0.f : d2;
  return pack_dist_idx(d2, (unsigned)j);
}

FP_DEVICE unsigned long long group16_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    const unsigned long long t = __shfl_xor(v, o, 16);
    v = t < v ? t : v;
  }
  return v;
}

// Stage 2.  16 lanes per row, 4 rows per wave, 16 rows per workgroup.  Lane `sub` of a row walks the list slots sub, sub + 16, ... (every
// lane reads its own entries: no serialized bookkeeping), re-scores the ones that pass the filter with the exact chain and keeps its k
// best keys; a 16-lane merge finishes the row.  (Measured alternatives, slower: one wave per row with the candidates' rows staged through
// LDS, and four lanes per candidate handing the accumulator from quarter to quarter -- both serialize the collection of the candidates.)
template <int KMAX>
__global__ __launch_bounds__(256) void knn_rescore_kernel(KnnCandArgs a, int k, int nsplit) {
  constexpr int CAP = kc_cap(KMAX);
  const int sub = threadIdx.x & 15;
  const long long rg = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int pair = (int)(rg / a.row_stride), row = (int)(rg % a.row_stride);
  bool live = pair < a.pairs;
  KcProblem p;
  if (live) {
    p = kc_problem(a, pair);
    live = p.live && row < p.x_cnt;
  }
  const size_t row_slot = (size_t)pair * a.row_stride + row;
  unsigned long long best[KMAX];
#pragma unroll
  for (int s = 0; s < KMAX; ++s) best[s] = ~0ull;
  if (live) {
    const int2* cn = reinterpret_cast<const int2*>(a.counts) + row_slot * KC_LISTS;
    const int nl = nsplit * 2;
    // a list's threshold min(k-th best it knew of, |x|^2 / 2) - 2 eps only rose while it was written; every one of them is a valid bound
    // for the whole row (header: superset), so the largest filters all lists
    bool brute = false;
    float thr = -INFINITY;
    for (int l = 0; l < nl; ++l) {
      const int2 c = cn[l];
      brute |= c.x < 0;
      thr = fmaxf(thr, __int_as_float(c.y));
    }
    const float* x = p.X + (size_t)(p.x_off + row) * a.ld;
    const float qn = p.xs[p.x_off + row];
    const int total = brute ? p.y_cnt : nl * CAP;     // brute: the row against the whole segment (rare: see the header); else every list slot
    for (int e = sub; e < total; e += 16) {
      int j = e;
      if (!brute) {
        const int l = e / CAP, pos = e - l * CAP;
        if (pos >= cn[l].x) continue;
        const unsigned long long ent = a.lists[(row_slot * KC_LISTS + l) * CAP + pos];
        if (!(__uint_as_float((unsigned)(ent >> 32)) >= thr)) continue;   // written under an earlier, weaker threshold
        j = (int)(ent & 0xffffffffu);
      }
      unsigned long long key = kc_exact_key(x, p.Y + (size_t)(p.y_off + j) * a.ld, a.K, qn, p.ys[p.y_off + j], j);
#pragma unroll
      for (int s = 0; s < KMAX; ++s) {
        const bool lt = key < best[s];
        const unsigned long long lo = lt ? key : best[s], hi = lt ? best[s] : key;
        best[s] = lo;
        key = hi;
      }
    }
  }
  // the k smallest keys of the row's 16 lanes (every database row sits in exactly one list: keys are distinct)
#pragma unroll
  for (int s = 0; s < KMAX; ++s) {
    if (s >= k) break;
    const unsigned long long m = group16_min_u64(best[0]);
    if (best[0] == m && m != ~0ull) {
#pragma unroll
      for (int t = 0; t + 1 < KMAX; ++t) best[t] = best[t + 1];
      best[KMAX - 1] = ~0ull;
    }
    if (live && sub == 0) {
      if (a.out_keys) a.out_keys[row_slot * k + s] = m;
      if (a.out_idx) a.out_idx[row_slot * k + s] = m == ~0ull ? -1 : (int)(m & 0xffffffffu);
      if (a.out_d2) a.out_d2[row_slot * k + s] = m == ~0ull ? INFINITY : __uint_as_float((unsigned)(m >> 32));
    }
  }
}

template <int KMAX, int K>
int launch_kk(const KnnCandArgs& a, int max_rows, int max_db, hipStream_t st) {
  constexpr int lds = 2 * 64 * (K * 2 + 16) + 128 * 4 + 8 * 4 + 256 * kc_cap(KMAX) * 8;
  static FpDeviceOnce attr;
  fp_allow_dynamic_lds(attr, &knn_cand_kernel<KMAX, K>, lds);
  int nsplit = (max_db + KC_SPLIT - 1) / KC_SPLIT;
  nsplit = nsplit < 1 ? 1 : (nsplit > 4 ? 4 : nsplit);
  hipLaunchKernelGGL((knn_cand_kernel<KMAX, K>), dim3(cdiv(max_rows, KC_ROWS), a.pairs, nsplit), dim3(256), lds, st, a);
  FP_CHECK_LAUNCH("knn_cand");
  const long long waves = (long long)a.pairs * a.row_stride;
  hipLaunchKernelGGL(knn_rescore_kernel<KMAX>, dim3((unsigned)((waves + 15) / 16)), dim3(256), 0, st, a, a.k, nsplit);
  FP_CHECK_LAUNCH("knn_rescore");
  return FP_OK;
}

template <int KMAX>
int launch_k(const KnnCandArgs& a, int max_rows, int max_db, hipStream_t st) {
  switch (a.K) {
    case 64: return launch_kk<KMAX, 64>(a, max_rows, max_db, st);
    case 128: return launch_kk<KMAX, 128>(a, max_rows, max_db, st);
    default: return launch_kk<KMAX, 256>(a, max_rows, max_db, st);
  }
}

}  // namespace

size_t knn_cand_scratch_bytes(int k, long long rows) { return (size_t)rows * ((size_t)KC_LISTS * kc_cap(k) * 8 + KC_LISTS * 8); }   // k = 1: 448 B, else 704 B per row

bool knn_cand_supported(int k, int K) { return k >= 1 && k <= 4 && (K == 64 || K == 128 || K == 256); }

int knn_cand_launch(const KnnCandArgs& a_in, int max_rows, int max_db, void* scratch, hipStream_t st) {
  KnnCandArgs a = a_in;
  FP_REQUIRE(knn_cand_supported(a.k, a.K), "knn_cand: k (%d) must be 1..4 and the dimension (%d) 64, 128 or 256", a.k, a.K);
  FP_REQUIRE(a.ld % 4 == 0 && a.pairs >= 1 && a.row_stride >= 1 && max_rows >= 1 && scratch, "knn_cand: bad arguments");
  const long long rows = (long long)a.pairs * a.row_stride;
  // stage 1 puts the pairs on gridDim.y (<= 65535), stage 2 one wave per row on gridDim.x (16 waves per block)
  FP_REQUIRE(a.pairs <= KNN_CAND_MAX_PAIRS && (rows + 15) / 16 <= 0x7fffffffLL, "knn_cand: %d pairs x %d rows exceed the launch grid (at most %d pairs)",
             a.pairs, a.row_stride, KNN_CAND_MAX_PAIRS);
  a.lists = reinterpret_cast<unsigned long long*>(scratch);
  a.counts = reinterpret_cast<int*>(a.lists + (size_t)rows * KC_LISTS * kc_cap(a.k));
  switch (a.k) {
    case 1: return launch_k<1>(a, max_rows, max_db, st);
    case 2: return launch_k<2>(a, max_rows, max_db, st);
    case 3: return launch_k<3>(a, max_rows, max_db, st);
    default: return launch_k<4>(a, max_rows, max_db, st);
  }
}
