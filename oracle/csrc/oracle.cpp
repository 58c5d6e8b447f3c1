// TEST INFRASTRUCTURE ONLY -- CPU oracle for the descriptor-matching half of the
// FoundPose hot path. Nothing in foundpose_amd/ may link, load or call this file;
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
//
// What it restates (reference file:line, /root/reference):
//   * exact brute-force L2 k-NN              utils/knn_util.py:38-106 (faiss IndexFlatL2,
//     faiss 1.8.0 is NOT in the mount -> arithmetic below is the canonical definition:
//     d2 = max(0, fma(-2, <x,y>, |x|^2 + |y|^2)), every reduction a k-ascending fp32
//     fmaf chain, results ascending by (d2, index)).  "parity unpinned" for faiss's
//     own low-order bits; pinned for everything built on top of it.
//   * torch.topk on CPU                      utils/template_util.py:172, utils/corresp_util.py:61
//     (ATen TopKImpl.h: partial_sort when k*64<=n, else nth_element + sort of k-1;
//     libstdc++ algorithms, so ties resolve exactly as in the reference process).
//   * scatter_add_ tf-idf histogram          utils/template_util.py:66-69 (sequential adds)
//
// Build: g++ -O2 -fPIC -shared -fopenmp -mfma -ffp-contract=off (see oracle/Makefile).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

extern "C" {

// |x_i|^2 as a k-ascending fmaf chain.
void orc_sqnorm(const float* x, int64_t n, int64_t d, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    float acc = 0.f;
    const float* r = x + i * d;
    for (int64_t k = 0; k < d; ++k) acc = fmaf(r[k], r[k], acc);
    out[i] = acc;
  }
}

// dots[i][j] = sum_k q[i][k] * db[j][k]; one fp32 accumulator per (i,j), k ascending.
// dbT is the [d][n] transpose so that the j loop vectorises without touching the
// per-element chain order.
static void dots_row(const float* qrow, const float* dbT, int64_t n, int64_t d, float* acc) {
  for (int64_t j = 0; j < n; ++j) acc[j] = 0.f;
  for (int64_t k = 0; k < d; ++k) {
    const float a = qrow[k];
    const float* b = dbT + k * n;
    for (int64_t j = 0; j < n; ++j) acc[j] = fmaf(a, b[j], acc[j]);
  }
}

// Full matrix of canonical squared distances [m][n].
void orc_l2_matrix(const float* q, int64_t m, const float* db, int64_t n, int64_t d, float* out) {
  std::vector<float> dbT((size_t)n * d), qn(m), dn(n);
  for (int64_t j = 0; j < n; ++j)
    for (int64_t k = 0; k < d; ++k) dbT[k * n + j] = db[j * d + k];
  orc_sqnorm(q, m, d, qn.data());
  orc_sqnorm(db, n, d, dn.data());
#pragma omp parallel
  {
    std::vector<float> acc(n);
#pragma omp for schedule(static)
    for (int64_t i = 0; i < m; ++i) {
      dots_row(q + i * d, dbT.data(), n, d, acc.data());
      float* o = out + i * n;
      for (int64_t j = 0; j < n; ++j) {
        float v = fmaf(-2.f, acc[j], qn[i] + dn[j]);
        o[j] = v < 0.f ? 0.f : v;
      }
    }
  }
}

// Exact k-NN, ascending by (d2, index).
void orc_l2_knn(const float* q, int64_t m, const float* db, int64_t n, int64_t d, int64_t k,
                float* out_d2, int64_t* out_idx) {
  std::vector<float> dbT((size_t)n * d), qn(m), dn(n);
  for (int64_t j = 0; j < n; ++j)
    for (int64_t kk = 0; kk < d; ++kk) dbT[kk * n + j] = db[j * d + kk];
  orc_sqnorm(q, m, d, qn.data());
  orc_sqnorm(db, n, d, dn.data());
#pragma omp parallel
  {
    std::vector<float> acc(n);
    std::vector<std::pair<float, int64_t>> cand(n);
#pragma omp for schedule(static)
    for (int64_t i = 0; i < m; ++i) {
      dots_row(q + i * d, dbT.data(), n, d, acc.data());
      for (int64_t j = 0; j < n; ++j) {
        float v = fmaf(-2.f, acc[j], qn[i] + dn[j]);
        cand[j] = {v < 0.f ? 0.f : v, j};
      }
      int64_t kk = std::min(k, n);
      std::partial_sort(cand.begin(), cand.begin() + kk, cand.end());  // pair order = (value, index)
      for (int64_t t = 0; t < k; ++t) {
        out_d2[i * k + t] = t < kk ? cand[t].first : INFINITY;
        out_idx[i * k + t] = t < kk ? cand[t].second : -1;
      }
    }
  }
}

// Column-wise argmin of the canonical distance matrix (the "object -> query" direction
// of cyclic matching): for each db row j, the query i with the smallest (d2, i).
void orc_l2_argmin_cols(const float* q, int64_t m, const float* db, int64_t n, int64_t d,
                        int64_t* out_idx, float* out_d2) {
  std::vector<float> mat((size_t)m * n);
  orc_l2_matrix(q, m, db, n, d, mat.data());
  for (int64_t j = 0; j < n; ++j) {
    float best = INFINITY;
    int64_t bi = -1;
    for (int64_t i = 0; i < m; ++i) {
      float v = mat[i * n + j];
      if (bi < 0 || v < best) { best = v; bi = i; }
    }
    out_idx[j] = bi;
    out_d2[j] = best;
  }
}

// sims[t] = <bank_n[t,:], qn[:]> as one k-ascending fmaf chain per template.
void orc_dot_rows(const float* bank, int64_t t, int64_t w, const float* q, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < t; ++i) {
    float acc = 0.f;
    const float* r = bank + i * w;
    for (int64_t k = 0; k < w; ++k) acc = fmaf(r[k], q[k], acc);
    out[i] = acc;
  }
}

// Same, but each 16-block of k is visited in the order [0,4,8,12, 1,5,9,13, 2,6,10,14, 3,7,11,15]: the canonical chain
// order of the MI355X bank-streaming cosine kernel (a lane's 16-byte load feeds four consecutive MFMA k-steps).
// torch's own cosine_similarity sum order is unspecified (vectorised cascade sum), so any fixed order is a valid
// restatement; this one is used whenever w % 16 == 0, the ascending one otherwise.
// `slices` contiguous k-slices, each its own chain; the slice results are added in slice order (device: one wave per
// slice so that every SIMD's fp32 matrix pipe works on the bank stream; 8 slices when w % 128 == 0, else 1).
void orc_dot_rows_perm16(const float* bank, int64_t t, int64_t w, const float* q, int64_t slices, float* out) {
  const int64_t ws = w / slices;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < t; ++i) {
    const float* r = bank + i * w;
    float total = 0.f;
    for (int64_t s = 0; s < slices; ++s) {
      float acc = 0.f;
      for (int64_t j = s * ws; j < (s + 1) * ws; j += 16)
        for (int u = 0; u < 4; ++u)
          for (int g = 0; g < 4; ++g) acc = fmaf(r[j + 4 * g + u], q[j + 4 * g + u], acc);
      total = s == 0 ? acc : total + acc;
    }
    out[i] = total;
  }
}

// Sequential scatter-add (torch CPU scatter_add_ walks the index tensor in order).
void orc_scatter_add(const int64_t* idx, const float* src, int64_t n, float* out) {
  for (int64_t i = 0; i < n; ++i) out[idx[i]] += src[i];
}

// torch.topk(values, k, largest, sorted) for a 1-D CPU tensor, tie-for-tie.
void orc_topk_torch(const float* values, int64_t n, int64_t k, int largest, int sorted,
                    float* out_val, int64_t* out_idx) {
  if (k == 0) return;
  using elem_t = std::pair<float, int64_t>;
  std::vector<elem_t> queue(n);
  for (int64_t j = 0; j < n; ++j) queue[j] = {values[j], j};
  auto gt = [](const elem_t& x, const elem_t& y) -> bool {
    return ((std::isnan(x.first) && !std::isnan(y.first)) || (x.first > y.first));
  };
  auto lt = [](const elem_t& x, const elem_t& y) -> bool {
    return ((!std::isnan(x.first) && std::isnan(y.first)) || (x.first < y.first));
  };
  const bool use_partial_sort = k * 64 <= n;
  if (use_partial_sort) {
    if (largest) std::partial_sort(queue.begin(), queue.begin() + k, queue.end(), gt);
    else std::partial_sort(queue.begin(), queue.begin() + k, queue.end(), lt);
  } else {
    if (largest) {
      std::nth_element(queue.begin(), queue.begin() + k - 1, queue.end(), gt);
      if (sorted) std::sort(queue.begin(), queue.begin() + k - 1, gt);
    } else {
      std::nth_element(queue.begin(), queue.begin() + k - 1, queue.end(), lt);
      if (sorted) std::sort(queue.begin(), queue.begin() + k - 1, lt);
    }
  }
  for (int64_t j = 0; j < k; ++j) {
    out_val[j] = queue[j].first;
    out_idx[j] = queue[j].second;
  }
}

// Canonical top-k: best value first, ties -> lowest index (what the MI355X fast path emits).
void orc_topk_canonical(const float* values, int64_t n, int64_t k, int largest,
                        float* out_val, int64_t* out_idx) {
  std::vector<std::pair<float, int64_t>> v(n);
  for (int64_t j = 0; j < n; ++j) v[j] = {largest ? -values[j] : values[j], j};
  std::stable_sort(v.begin(), v.end(),
                   [](const std::pair<float, int64_t>& a, const std::pair<float, int64_t>& b) {
                     return a.first < b.first;
                   });
  for (int64_t j = 0; j < k; ++j) {
    out_val[j] = values[v[j].second];
    out_idx[j] = v[j].second;
  }
}

}  // extern "C"
