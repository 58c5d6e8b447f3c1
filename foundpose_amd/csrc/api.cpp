// C ABI of libfoundpose_amd.so (declared in include/foundpose_amd.h): argument checking, scratch carving
// and kernel sequencing.  No allocation, no synchronisation, no global mutable state.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>

#include "../../include/foundpose_amd.h"
#include "common.hpp"
#include "kernels.hpp"

static thread_local char g_err[512] = "";

void fp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

#define ST(s) reinterpret_cast<hipStream_t>(s)
#define TRY(expr)              \
  do {                         \
    int rc__ = (expr);         \
    if (rc__ != FP_OK) return rc__; \
  } while (0)
#define HIP_TRY(expr, what)                                             \
  do {                                                                  \
    hipError_t e__ = (expr);                                            \
    if (e__ != hipSuccess) {                                            \
      fp_set_error("%s: %s", what, hipGetErrorString(e__));             \
      return FP_ERR_HIP;                                                \
    }                                                                   \
  } while (0)

static F32TileArgs zero_tile_args() {
  F32TileArgs a;
  memset(&a, 0, sizeof(a));
  return a;
}

extern "C" {

int fp_abi_version(void) { return FP_ABI_VERSION; }
int fp_build_experiments(void) {
#ifdef FP_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}
const char* fp_last_error(void) { return g_err; }

// ------------------------------------------------------------------ matching half
int fp_sqnorm_rows(const float* x, int64_t n, int d, float* out, fp_stream_t stream) {
  FP_REQUIRE(x && out, "fp_sqnorm_rows: null pointer");
  return launch_sqnorm_rows(x, n, d, d, out, ST(stream));
}

int fp_normalize_rows(const float* x, int64_t n, int d, float eps, float* out, fp_stream_t stream) {
  FP_REQUIRE(x && out, "fp_normalize_rows: null pointer");
  return launch_normalize_rows(x, n, d, eps, out, ST(stream));
}

// The two-stage search of csrc/knn_cand.hip (fp16-MFMA candidate pass with a derived bound + exact re-scoring; outputs bit-identical to
// the all-pairs exact tile) is built and tested but NOT the default: measured on the bench shapes it is slower than the exact-fp32 tile it
// was meant to replace (word 3-NN 234 vs 221 us, cyclic searches 419 vs 338 us: profiles/EXPERIMENTS.md "Two-stage k-NN").  It is compiled into
// FP_EXPERIMENTS builds only (FP_EXPERIMENTS=1 python -m foundpose_amd.build), where FP_KNN_CAND=1 switches it on (read per call, so a test can compare
// both paths in one process); the shipped library has neither the kernels nor the switch.
#ifdef FP_EXPERIMENTS
static bool knn_cand_enabled() {
  const char* e = getenv("FP_KNN_CAND");
  return e != nullptr && atoi(e) != 0;
}
#endif

int fp_knn_l2(const float* q, const float* q_sqnorm, int m, const float* db, const float* db_sqnorm, int n, int d,
              int k, void* scratch, float* out_d2, int32_t* out_idx, fp_stream_t stream) {
  FP_REQUIRE(q && q_sqnorm && db && db_sqnorm && scratch && out_idx, "fp_knn_l2: null pointer");
  FP_REQUIRE(k >= 1 && n >= 1 && d >= 4, "fp_knn_l2: bad sizes (n=%d d=%d k=%d)", n, d, k);
  if (m == 0) return FP_OK;
  // 2 <= k <= 4 against a database of some size (the visual-word search: k = 3, 2048 words), opt-in (FP_KNN_CAND=1): fp16-MFMA candidate
  // pass + exact fp32 chains on the candidates (knn_cand.hip) -- the same d2 / indices as the all-pairs exact tile below, bit for bit.
#ifdef FP_EXPERIMENTS
  if (k >= 2 && n >= 256 && knn_cand_enabled() && knn_cand_supported(k, d) && out_d2) {
    KnnCandArgs c;
    memset(&c, 0, sizeof(c));
    c.A = q; c.B = db; c.ld = d; c.a_sqn = q_sqnorm; c.b_sqn = db_sqnorm; c.K = d; c.M = m; c.N = n;
    c.k = k; c.pairs = 1; c.row_stride = m; c.out_d2 = out_d2; c.out_idx = out_idx;
    return knn_cand_launch(c, m, n, scratch, ST(stream));
  }
#endif
  F32TileArgs a = zero_tile_args();
  a.A = q; a.lda = d; a.B = db; a.ldb = d; a.K = d; a.M = m; a.N = n;
  a.a_sqnorm = q_sqnorm; a.b_sqnorm = db_sqnorm;
  if (k == 1) {
    unsigned long long* best = reinterpret_cast<unsigned long long*>(scratch);
    HIP_TRY(hipMemsetAsync(best, 0xFF, (size_t)m * 8, ST(stream)), "fp_knn_l2: memset");
    a.row_best = best; a.row_stride = 0; a.col_best = nullptr;
    TRY(f32_tile_launch(F32_EPI_DIST_ARGMIN, a, m, n, 1, ST(stream)));
    return launch_unpack_best(best, m, out_d2, out_idx, ST(stream));
  }
  FP_REQUIRE(out_d2, "fp_knn_l2: out_d2 is required for k > 1");
  if (k <= 8) {  // fused: every distance tile emits its k best per row, a merge over the n-tiles finishes the row
    const int nt = (n + 127) / 128;
    a.row_best = reinterpret_cast<unsigned long long*>(scratch); a.row_stride = k;
    TRY(f32_tile_launch(F32_EPI_DIST_TOPK, a, m, n, 1, ST(stream)));
    return launch_knn_merge(a.row_best, m, nt * k, k, out_d2, out_idx, ST(stream));
  }
  float* dist = reinterpret_cast<float*>(scratch);
  a.out = dist; a.ldo = n;
  TRY(f32_tile_launch(F32_EPI_DIST_STORE, a, m, n, 1, ST(stream)));
  return launch_topk_rows(dist, m, n, n, nullptr, k, 0, out_d2, out_idx, ST(stream));
}

int fp_tfidf_build(const int32_t* word_ids, const float* word_d2, int knn_k, const int32_t* seg_off, int num_segs,
                   const float* idf, int num_words, int soft_assign, float soft_sigma_squared, int sqrt_dists,
                   float* desc, float* desc_n, float eps, fp_stream_t stream) {
  FP_REQUIRE(word_ids && word_d2 && seg_off && idf && desc, "fp_tfidf_build: null pointer");
  return launch_tfidf_build(word_ids, word_d2, knn_k, seg_off, num_segs, idf, num_words, soft_assign,
                            soft_sigma_squared, sqrt_dists, desc, desc_n, eps, ST(stream));
}

int fp_cosine_topk(const float* desc_n, const int32_t* det_seg_off, const int32_t* det_num_templates, int num_det,
                   int max_det_per_obj, const float* bank_n, const int32_t* obj_tpl_off, int num_obj,
                   int max_templates, int num_words, int n_top, float* scratch_sims, float* out_scores,
                   int32_t* out_ids, int tie_mode, fp_stream_t stream) {
  FP_REQUIRE(desc_n && det_seg_off && det_num_templates && bank_n && obj_tpl_off && scratch_sims && out_scores && out_ids,
             "fp_cosine_topk: null pointer");
  FP_REQUIRE(num_obj >= 1 && max_templates >= 1 && n_top >= 1, "fp_cosine_topk: bad sizes");
  if (num_det == 0) return FP_OK;
  FP_REQUIRE(tie_mode == 0 || tie_mode == 1, "fp_cosine_topk: tie_mode must be 0 (canonical) or 1 (torch)");
  if (num_words % 16 == 0) {
    // one arithmetic (8 k-slice chains when num_words % 128 == 0) whatever the batch: a score never depends on how many
    // detections share the launch; > 32 detections of an object are served in 32-detection chunks by the same kernel
    CosineArgs c;
    memset(&c, 0, sizeof(c));
    c.desc_n = desc_n; c.bank_n = bank_n; c.det_seg_off = det_seg_off; c.obj_tpl_off = obj_tpl_off; c.W = num_words;
    c.sims = scratch_sims; c.ld_sims = max_templates;
    c.cand = reinterpret_cast<unsigned long long*>(scratch_sims + (size_t)num_det * max_templates + ((size_t)num_det * max_templates & 1));
    c.need_replay = reinterpret_cast<int*>(scratch_sims + 2 * (size_t)num_det * max_templates + 16 * (size_t)num_det + 2);  // behind the keys
    return launch_cosine_topk(c, num_det, num_obj, max_det_per_obj, max_templates, n_top, det_num_templates, out_scores,
                              out_ids, tie_mode, ST(stream));
  }
  // descriptor sizes that are not a multiple of 16 words: generic fp32 tile, k-ascending chains
  F32TileArgs a = zero_tile_args();
  a.A = desc_n; a.lda = num_words; a.B = bank_n; a.ldb = num_words; a.K = num_words;
  a.a_seg_off = det_seg_off; a.b_seg_off = obj_tpl_off;
  a.out = scratch_sims; a.ldo = max_templates; a.out_row_global = 1;
  TRY(f32_tile_launch(F32_EPI_STORE, a, max_det_per_obj, max_templates, num_obj, ST(stream)));
  return launch_topn_rows(scratch_sims, max_templates, num_det, max_templates, det_num_templates, n_top, out_scores, out_ids, tie_mode, ST(stream));
}

int fp_cosine_topk_prefiltered(const float* desc_n, const int32_t* det_seg_off, const int32_t* det_num_templates, int num_det, int max_det_per_obj,
                               const float* bank_n, const void* bank_n_bf16, const int32_t* obj_tpl_off, int num_obj, int max_templates, int num_words,
                               int n_top, float* scratch, float* out_scores, int32_t* out_ids, int tie_mode, fp_stream_t stream) {
  FP_REQUIRE(desc_n && det_seg_off && det_num_templates && bank_n && bank_n_bf16 && obj_tpl_off && scratch && out_scores && out_ids,
             "fp_cosine_topk_prefiltered: null pointer");
  FP_REQUIRE(num_obj >= 1 && max_templates >= 1 && n_top >= 1, "fp_cosine_topk_prefiltered: bad sizes");
  if (num_det == 0) return FP_OK;
  const int force = (tie_mode >> 8) & 1;  // FP_COSINE_FORCE_PREFILTER: the two-stage form whatever the size (tests, measurements)
  tie_mode &= 0xff;
  FP_REQUIRE(tie_mode == 0 || tie_mode == 1, "fp_cosine_topk_prefiltered: tie_mode must be 0 (canonical) or 1 (torch)");
  if (num_words % 16 != 0)  // the generic exact path
    return fp_cosine_topk(desc_n, det_seg_off, det_num_templates, num_det, max_det_per_obj, bank_n, obj_tpl_off, num_obj, max_templates, num_words, n_top,
                          scratch, out_scores, out_ids, tie_mode, stream);
  CosineArgs c;
  memset(&c, 0, sizeof(c));
  c.force_prefilter = force;
  c.desc_n = desc_n; c.bank_n = bank_n; c.bank_bf16 = bank_n_bf16; c.det_seg_off = det_seg_off; c.obj_tpl_off = obj_tpl_off; c.W = num_words;
  c.sims = scratch; c.ld_sims = max_templates;
  c.cand = reinterpret_cast<unsigned long long*>(scratch + (size_t)num_det * max_templates + ((size_t)num_det * max_templates & 1));
  c.need_replay = reinterpret_cast<int*>(scratch + 2 * (size_t)num_det * max_templates + 16 * (size_t)num_det + 2);
  float* extra = scratch + FP_COSINE_SCRATCH_FLOATS(num_det, max_templates) + (FP_COSINE_SCRATCH_FLOATS(num_det, max_templates) & 1);
  return launch_cosine_topk_prefiltered(c, num_det, num_obj, max_det_per_obj, max_templates, n_top, det_num_templates, out_scores, out_ids, tie_mode, extra,
                                        ST(stream));
}

int fp_cyclic_buddies(const float* query_feats, const float* query_sqnorm, const float* query_points,
                      const int32_t* q_off, int num_det, int q_max, const float* bank_feats,
                      const float* bank_sqnorm, const int32_t* tpl_off, int p_max, const float* vertices,
                      const int32_t* tpl_ids, const int32_t* tpl_base, const int32_t* feat_base, int n_slots, int d, int top_k, int k_max,
                      void* scratch, int32_t* out_count, int32_t* out_q_ids, int32_t* out_feat_ids,
                      float* out_dists, float* out_conf, float* out_coord_2d, float* out_coord_3d, int tie_mode,
                      fp_stream_t stream) {
  FP_REQUIRE(query_feats && query_sqnorm && query_points && q_off && bank_feats && bank_sqnorm && tpl_off && vertices &&
                 tpl_ids && feat_base && scratch && out_count && out_q_ids && out_feat_ids && out_dists && out_conf &&
                 out_coord_2d && out_coord_3d,
             "fp_cyclic_buddies: null pointer");
  FP_REQUIRE(q_max >= 1 && p_max >= 1 && n_slots >= 1, "fp_cyclic_buddies: bad sizes");
  const int pairs = num_det * n_slots;
  if (pairs == 0) return FP_OK;
  CyclicArgs c;
  memset(&c, 0, sizeof(c));
#ifdef FP_EXPERIMENTS
  const bool use_cand = knn_cand_enabled() && knn_cand_supported(1, d) && pairs <= KNN_CAND_MAX_PAIRS;   // beyond the launch's grid limits: the all-pairs tile below, same keys
#else
  constexpr bool use_cand = false;
#endif
  if (use_cand) {
#ifdef FP_EXPERIMENTS
    // the two 1-NN searches of a pair (corresp_util.py:46-47) by candidate pass + exact re-scoring (knn_cand.hip): query patch -> nearest
    // template patch into row_best [pairs, q_max], template patch -> nearest query patch into col_best [pairs, p_max]; the same keys
    // (d2, index; ties -> lowest index) the all-pairs tile below leaves, so everything downstream is unchanged
    unsigned long long* row_best = reinterpret_cast<unsigned long long*>(scratch);
    unsigned long long* col_best = row_best + (size_t)pairs * q_max;
    void* cand = col_best + (size_t)pairs * p_max;
    KnnCandArgs kc;
    memset(&kc, 0, sizeof(kc));
    kc.A = query_feats; kc.B = bank_feats; kc.ld = d; kc.a_sqn = query_sqnorm; kc.b_sqn = bank_sqnorm; kc.K = d;
    kc.a_seg_off = q_off; kc.pair_a_div = n_slots; kc.b_seg_off = tpl_off; kc.pair_b_seg = tpl_ids; kc.pair_b_base = tpl_base;
    kc.k = 1; kc.pairs = pairs;
    kc.swap = 0; kc.row_stride = q_max; kc.out_keys = row_best;
    TRY(knn_cand_launch(kc, q_max, p_max, cand, ST(stream)));
    kc.swap = 1; kc.row_stride = p_max; kc.out_keys = col_best;
    TRY(knn_cand_launch(kc, p_max, q_max, cand, ST(stream)));
    c.row_best = row_best; c.row_stride = q_max; c.col_best = col_best; c.col_stride = p_max; c.row_parts = 1; c.col_parts = 1;
#endif
  } else {
  // partial nearest-neighbour tables, one slice per distance tile (no atomics, no preset): [pairs, col tiles, q_max] + [pairs, row tiles, p_max]
  const int row_parts = (p_max + 127) / 128, col_parts = (q_max + 127) / 128;
  unsigned long long* row_best = reinterpret_cast<unsigned long long*>(scratch);
  unsigned long long* col_best = row_best + (size_t)pairs * row_parts * q_max;
  F32TileArgs a = zero_tile_args();
  a.A = query_feats; a.lda = d; a.B = bank_feats; a.ldb = d; a.K = d;
  a.a_seg_off = q_off; a.pair_a_div = n_slots;
  a.b_seg_off = tpl_off; a.pair_b_seg = tpl_ids; a.pair_b_base = tpl_base;
  a.a_sqnorm = query_sqnorm; a.b_sqnorm = bank_sqnorm;
  a.row_best = row_best; a.row_stride = q_max; a.col_best = col_best; a.col_stride = p_max; a.best_parts = 1;
  TRY(f32_tile_launch(F32_EPI_DIST_ARGMIN, a, q_max, p_max, pairs, ST(stream)));
  c.row_best = row_best; c.row_stride = q_max; c.col_best = col_best; c.col_stride = p_max;
  c.row_parts = row_parts; c.col_parts = col_parts;
  }
  c.q_off = q_off; c.tpl_ids = tpl_ids; c.tpl_base = tpl_base; c.tpl_off = tpl_off; c.feat_base = feat_base;
  c.points = query_points; c.vertices = vertices;
  c.n_slots = n_slots; c.top_k = top_k; c.k_max = k_max; c.q_max = q_max; c.tie_mode = tie_mode;
  c.out_count = out_count; c.out_q_ids = out_q_ids; c.out_feat_ids = out_feat_ids; c.out_dists = out_dists;
  c.out_conf = out_conf; c.out_coord_2d = out_coord_2d; c.out_coord_3d = out_coord_3d;
  return launch_cyclic_select(c, pairs, ST(stream));
}

int fp_pack_records(const int32_t* template_ids, const float* template_scores, const int32_t* counts, const int32_t* q_ids, const int32_t* feat_ids,
                    const float* dists, const float* conf, const float* coord_2d, const float* coord_3d, int num_det, int n_slots, int k_max, float* out,
                    fp_stream_t stream) {
  FP_REQUIRE(template_ids && template_scores && counts && q_ids && feat_ids && dists && conf && coord_2d && coord_3d && out, "fp_pack_records: null pointer");
  FP_REQUIRE(num_det >= 0 && n_slots >= 1 && k_max >= 1, "fp_pack_records: bad sizes");
  return launch_pack_records(template_ids, template_scores, counts, q_ids, feat_ids, dists, conf, coord_2d, coord_3d, num_det, n_slots, k_max, out, ST(stream));
}

int fp_sample_bilinear(const float* fmap, int64_t stride_img, int64_t stride_c, int64_t stride_h, int64_t stride_w,
                       int C, int H, int W, int img_w, int img_h, const float* points, const int32_t* point_img,
                       int num_points, float* out, fp_stream_t stream) {
  FP_REQUIRE(fmap && points && out, "fp_sample_bilinear: null pointer");
  SampleArgs a;
  a.fmap = fmap; a.stride_img = stride_img; a.stride_c = stride_c; a.stride_h = stride_h; a.stride_w = stride_w;
  a.C = C; a.H = H; a.W = W; a.img_w = img_w; a.img_h = img_h;
  a.points = points; a.point_img = point_img; a.num_points = num_points; a.out = out;
  return launch_sample_bilinear(a, ST(stream));
}

int fp_pca_project(const float* x, int n, int D, const float* components, int d, const float* mean_proj, float* out,
                   fp_stream_t stream) {
  FP_REQUIRE(x && components && out, "fp_pca_project: null pointer");
  if (n == 0) return FP_OK;
  F32TileArgs a = zero_tile_args();
  a.A = x; a.lda = D; a.B = components; a.ldb = D; a.K = D; a.M = n; a.N = d;
  a.out = out; a.ldo = d; a.bias = mean_proj;
  return f32_tile_launch(mean_proj ? F32_EPI_SUB_VEC : F32_EPI_STORE, a, n, d, 1, ST(stream));
}

int fp_pnp_ransac(const float* coord_2d, const float* coord_3d, const int32_t* counts, const double* cameras, int num_pairs, int n_slots,
                  int k_max, int ransac_iters, double inlier_thresh, double confidence, int lm_iters, int min_corresp, uint64_t seed,
                  int32_t* out_success, double* out_R, double* out_t, int32_t* out_num_inliers, uint8_t* out_inlier_mask,
                  double* out_ransac_pose, fp_stream_t stream) {
  FP_REQUIRE(coord_2d && coord_3d && counts && cameras && out_success && out_R && out_t && out_num_inliers && out_inlier_mask,
             "fp_pnp_ransac: null pointer");
  FP_REQUIRE(num_pairs >= 0 && n_slots >= 1 && num_pairs % n_slots == 0, "fp_pnp_ransac: num_pairs must be a multiple of n_slots");
  PnpArgs a;
  memset(&a, 0, sizeof(a));
  a.coord_2d = coord_2d; a.coord_3d = coord_3d; a.counts = counts; a.cam = cameras;
  a.n_slots = n_slots; a.k_max = k_max; a.iters = ransac_iters; a.lm_iters = lm_iters; a.min_corresp = min_corresp;
  a.thresh = inlier_thresh; a.conf = confidence; a.seed = seed;
  a.success = out_success; a.R = out_R; a.t = out_t; a.n_inliers = out_num_inliers; a.inlier_mask = out_inlier_mask;
  a.ransac_pose = out_ransac_pose;
  return launch_pnp_ransac(a, num_pairs, ST(stream));
}

// ------------------------------------------------------------------ ViT building blocks
int fp_patchify(const float* images, int B, int H, int W, int patch, void* out, int ld_out, int out_dtype,
                fp_stream_t stream) {
  FP_REQUIRE(images && out, "fp_patchify: null pointer");
  return patchify_launch(images, B, H, W, patch, out, ld_out, out_dtype, ST(stream), (out_dtype == FP_DTYPE_F16X3 || out_dtype == FP_DTYPE_F16F8) ? FP_SPLIT_SCALE_ACT : 1.f);
}

int fp_layernorm(const float* x, int ld_x, const float* weight, const float* bias, float eps, void* out, int ld_out,
                 int out_dtype, int dim, int out_rows, int out_rows_per_img, int in_rows_per_img, int in_skip,
                 fp_stream_t stream) {
  FP_REQUIRE(x && weight && bias && out, "fp_layernorm: null pointer");
  LayerNormArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.ld_x = ld_x; a.weight = weight; a.bias = bias; a.eps = eps; a.out = out; a.ld_out = ld_out;
  a.out_dtype = out_dtype; a.out_scale = 0.f; a.dim = dim; a.out_rows = out_rows;
  a.out_rows_per_img = out_rows_per_img > 0 ? out_rows_per_img : (out_rows > 0 ? out_rows : 1);
  a.in_rows_per_img = in_rows_per_img > 0 ? in_rows_per_img : a.out_rows_per_img;
  a.in_skip = in_skip;
  return layernorm_launch(a, ST(stream));
}

int fp_gemm_bf16(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int M_valid, const float* bias,
                 const float* gamma, void* out, int ldo, int epilogue, fp_stream_t stream) {
  FP_REQUIRE(A && W && out, "fp_gemm_bf16: null pointer");
  const int tile = (epilogue >> 8) & 0xfff;  // tuning bits: force the 128 or 256 block tile
  const bool f16 = (epilogue >> 21) & 1;     // FP_GEMM_F16: IEEE fp16 operands and 16-bit outputs
  epilogue &= 0xff;
  FP_REQUIRE(tile == 0 || tile == 64 || tile == 128 || tile == 256 || tile == 320, "fp_gemm_bf16: bad tile override %d", tile);

  FP_REQUIRE(epilogue == GEMM_EPI_BIAS_BF16 || epilogue == GEMM_EPI_GELU_BF16 || epilogue == GEMM_EPI_LS_RESID_F32 ||
                 epilogue == GEMM_EPI_BIAS_F32 || epilogue == GEMM_EPI_SWIGLU_BF16,
             "fp_gemm_bf16: epilogue %d is not available through this entry point", epilogue);
  FP_REQUIRE(epilogue != GEMM_EPI_LS_RESID_F32 || gamma, "fp_gemm_bf16: gamma required");
  GemmBf16Args a;
  memset(&a, 0, sizeof(a));
  a.A = reinterpret_cast<const __bf16*>(A); a.lda = lda; a.W = reinterpret_cast<const __bf16*>(W); a.ldw = ldw;
  a.M = M; a.N = N; a.K = K; a.M_valid = M_valid; a.bias = bias; a.gamma = gamma; a.out = out; a.ldo = ldo;
  a.tile_override = tile;
  return f16 ? gemm_f16_launch(epilogue, a, ST(stream)) : gemm_bf16_launch(epilogue, a, ST(stream));
}

int fp_gemm_bf16_ln(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int M_valid, const float* bias, void* out, int ldo,
                    int epilogue, const float* colsum, const float* ln_row, void* xb, int ld_xb, float* stats, fp_stream_t stream) {
  FP_REQUIRE(A && W && out && bias, "fp_gemm_bf16_ln: null pointer");
  const int tile = (epilogue >> 8) & 0xfff;
  const bool f16 = (epilogue >> 21) & 1;     // FP_GEMM_F16: IEEE fp16 operands, 16-bit outputs and (hi, lo) stream
  epilogue &= 0xff;
  FP_REQUIRE(tile == 0 || tile == 64 || tile == 128 || tile == 256 || tile == 320, "fp_gemm_bf16_ln: bad tile override %d", tile);
  GemmBf16Args a;
  memset(&a, 0, sizeof(a));
  a.A = reinterpret_cast<const __bf16*>(A); a.lda = lda; a.W = reinterpret_cast<const __bf16*>(W); a.ldw = ldw;
  a.M = M; a.N = N; a.K = K; a.M_valid = M_valid; a.bias = bias; a.out = out; a.ldo = ldo; a.tile_override = tile;
  if (epilogue == GEMM_EPI_RESID_HILO) {  // producer on the (hi, lo) stream: `out` is the LOW-half array (bf16, row stride ld_xb), xb the high halves
    FP_REQUIRE(xb && stats, "fp_gemm_bf16_ln: epilogue 8 needs xb, out (= the low halves) and stats");
    a.xb = reinterpret_cast<__bf16*>(xb); a.ld_xb = ld_xb; a.xl = reinterpret_cast<__bf16*>(out); a.stats_out = reinterpret_cast<float2*>(stats);
    a.out = nullptr;
  } else if (epilogue == GEMM_EPI_RESID_F32) {  // producer: x += acc + bias, plus bf16(x) and the partial row sums
    FP_REQUIRE((xb == nullptr) == (stats == nullptr), "fp_gemm_bf16_ln: xb and stats go together");
    FP_REQUIRE(!xb || (N % 128 == 0 && ld_xb >= N && ld_xb % 4 == 0), "fp_gemm_bf16_ln: N must be a multiple of 128 and ld_xb cover the row");
    a.xb = reinterpret_cast<__bf16*>(xb); a.ld_xb = ld_xb; a.stats_out = reinterpret_cast<float2*>(stats);
  } else {                               // consumer: epi(rstd * (acc - mean * colsum) + bias)
    FP_REQUIRE(epilogue == GEMM_EPI_BIAS_BF16 || epilogue == GEMM_EPI_GELU_BF16 || epilogue == GEMM_EPI_SWIGLU_BF16,
               "fp_gemm_bf16_ln: epilogue %d has no folded-LayerNorm form", epilogue);
    FP_REQUIRE(colsum && ln_row, "fp_gemm_bf16_ln: colsum and ln_row are required");
    a.colsum = colsum; a.ln_stats = reinterpret_cast<const float2*>(ln_row); a.ln_eps = 1e-6f;
  }
  return f16 ? gemm_f16_launch(epilogue, a, ST(stream)) : gemm_bf16_launch(epilogue, a, ST(stream));
}

int fp_ln_finalize(const float* stats, int parts, int stats_stride, int rows, int dim, float eps, float* ln_row, fp_stream_t stream) {
  FP_REQUIRE(stats && ln_row && parts >= 1 && dim >= 1, "fp_ln_finalize: bad arguments");
  return ln_finalize_launch(reinterpret_cast<const float2*>(stats), parts, stats_stride, rows, dim, eps, reinterpret_cast<float2*>(ln_row), ST(stream));
}

#ifdef FP_GEMM_TIMELINE
// Measurement build only (tools/build_variant.sh -DFP_GEMM_TIMELINE, tools/gemm_timeline.py): the same GEMM writing four
// shader-clock stamps per workgroup into a caller-owned buffer of dbg_len >= 4 * grid u64 slots.
int fp_gemm_bf16_timeline(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int M_valid, const float* bias,
                          const float* gamma, void* out, int ldo, int epilogue, unsigned long long* dbg, int64_t dbg_len,
                          fp_stream_t stream) {
  FP_REQUIRE(A && W && out && dbg, "fp_gemm_bf16_timeline: null pointer");
  FP_REQUIRE(dbg_len >= 4ll * ((M + 127) / 128) * ((N + 127) / 128) * 2, "fp_gemm_bf16_timeline: stamp buffer too small");
  GemmBf16Args a;
  memset(&a, 0, sizeof(a));
  a.A = reinterpret_cast<const __bf16*>(A); a.lda = lda; a.W = reinterpret_cast<const __bf16*>(W); a.ldw = ldw;
  a.M = M; a.N = N; a.K = K; a.M_valid = M_valid; a.bias = bias; a.gamma = gamma; a.out = out; a.ldo = ldo;
  a.tile_override = (epilogue >> 8) & 0xfff;
  a.dbg = dbg;
  return gemm_bf16_launch(epilogue & 0xff, a, ST(stream));
}
#endif

static int gemm_fp8_impl(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int M_valid, const float* bias,
                         const float* col_scale, void* out, int ldo, int epilogue, float out_scale, int* sat, fp_stream_t stream, int no_tall = 0) {
  FP_REQUIRE(A && W && out, "fp_gemm_fp8: null pointer");
  FP_REQUIRE(out_scale >= 0.f, "fp_gemm_fp8: out_scale must be >= 0");
  GemmBf16Args a;
  memset(&a, 0, sizeof(a));
  a.A = reinterpret_cast<const __bf16*>(A); a.lda = lda; a.W = reinterpret_cast<const __bf16*>(W); a.ldw = ldw;
  a.M = M; a.N = N; a.K = K; a.M_valid = M_valid; a.bias = bias; a.gamma = col_scale; a.out = out; a.ldo = ldo;
  a.out_scale = out_scale; a.sat = sat; a.no_tall = no_tall;
  a.tile_override = (epilogue >> 8) & 0xfff;  // tuning bits: 256 / 320 force that block tile (benchmarks, tests)
  FP_REQUIRE(a.tile_override == 0 || a.tile_override == 256 || a.tile_override == 320, "fp_gemm_fp8: bad tile override %d", a.tile_override);
  return gemm_fp8_launch(epilogue & 0xff, a, ST(stream));
}

int fp_gemm_fp8(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int M_valid, const float* bias,
                const float* col_scale, void* out, int ldo, int epilogue, float out_scale, fp_stream_t stream) {
  return gemm_fp8_impl(A, lda, W, ldw, M, N, K, M_valid, bias, col_scale, out, ldo, epilogue, out_scale, nullptr, stream);
}

int fp_gemm_split(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int M_valid, const float* bias, const float* gamma,
                  void* out, int ldo, int epilogue, float acc_scale, float out_scale, fp_stream_t stream) {
  FP_REQUIRE(A && W && out, "fp_gemm_split: null pointer");
  const int tile = (epilogue >> 8) & 0xfff;
  const bool f16f8 = (epilogue >> 20) & 1;   // FP_GEMM_SPLIT_F16F8: the operands (and a GELU / SwiGLU output) are f16f8 rows
  epilogue &= 0xff;
  FP_REQUIRE(tile == 0 || tile == 128 || tile == 256, "fp_gemm_split: bad tile override %d", tile);
  FP_REQUIRE(epilogue == GEMM_EPI_BIAS_BF16 || epilogue == GEMM_EPI_GELU_BF16 || epilogue == GEMM_EPI_LS_RESID_F32 ||
                 epilogue == GEMM_EPI_BIAS_F32 || epilogue == GEMM_EPI_SWIGLU_BF16,
             "fp_gemm_split: epilogue %d is not available through this entry point", epilogue);
  GemmBf16Args a;
  memset(&a, 0, sizeof(a));
  a.A = reinterpret_cast<const __bf16*>(A); a.lda = lda; a.W = reinterpret_cast<const __bf16*>(W); a.ldw = ldw;
  a.M = M; a.N = N; a.K = K; a.M_valid = M_valid; a.bias = bias; a.gamma = gamma; a.out = out; a.ldo = ldo;
  a.tile_override = tile; a.acc_scale = acc_scale; a.out_scale = out_scale;
  return f16f8 ? gemm_splitx_launch(epilogue, a, ST(stream)) : gemm_split_launch(epilogue, a, ST(stream));
}

int fp_attention_split(const void* qkv, int ld_qkv, void* out, int ld_out, int B, int n_tok, int dim, int heads, float in_scale, float out_scale,
                       int out_dtype, fp_stream_t stream) {
  FP_REQUIRE(qkv && out, "fp_attention_split: null pointer");
  const int variant = (out_dtype >> 8) & 0xff;  // test bits, as in fp_attention: 0 = the default kernel, 1 = the lock-step kernel, 2 = the role-split kernel (bit-identical)
  out_dtype &= 0xff;
  FP_REQUIRE(out_dtype == FP_DTYPE_F16X3 || out_dtype == FP_DTYPE_F16F8, "fp_attention_split: the output is a split-fp16 row (FP_F16X3) or an f16f8 row (FP_F16F8)");
  FP_REQUIRE(variant <= 2, "fp_attention_split: unknown kernel variant %d", variant);
  // in_scale >= 1: the kernel's lazy-rescale threshold is a constant in raw score units (1 / (0.125 log2 e), attn.hip SPLIT_LAZY_TH) and the scores carry
  // in_scale^2 -- in exponent units the reference trails the maximum by at most 1 / in_scale^2, so p' = 2^14 p <= 2^15 fits fp16 for in_scale >= 1 only
  FP_REQUIRE(in_scale >= 1.f && out_scale > 0.f, "fp_attention_split: in_scale must be >= 1 (the split P is packed unclamped: p' <= 2^15 needs it) and out_scale positive");
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.variant = variant;
  a.qkv = qkv; a.ld_qkv = ld_qkv; a.out = out; a.ld_out = ld_out;
  a.batch = B; a.n_tok = n_tok; a.dim = dim; a.heads = heads; a.in_scale = in_scale; a.out_scale = out_scale;
  a.out_fmt = out_dtype == FP_DTYPE_F16F8 ? 1 : 0;
  return attn_launch(a, FP_DTYPE_F16X3, ST(stream));
}

int fp_layernorm_scaled(const float* x, int ld_x, const float* weight, const float* bias, float eps, void* out, int ld_out, int out_dtype, float out_scale,
                        int dim, int out_rows, fp_stream_t stream) {
  FP_REQUIRE(x && weight && bias && out, "fp_layernorm_scaled: null pointer");
  FP_REQUIRE(out_dtype == FP_DTYPE_FP8 || out_dtype == FP_DTYPE_F16X3 || out_dtype == FP_DTYPE_F16F8, "fp_layernorm_scaled: the scaled outputs are fp8 bytes, split-fp16 rows and f16f8 rows");
  LayerNormArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.ld_x = ld_x; a.weight = weight; a.bias = bias; a.eps = eps; a.out = out; a.ld_out = ld_out;
  a.out_dtype = out_dtype; a.out_scale = out_scale; a.dim = dim; a.out_rows = out_rows;
  a.out_rows_per_img = out_rows > 0 ? out_rows : 1; a.in_rows_per_img = a.out_rows_per_img; a.in_skip = 0;
  return layernorm_launch(a, ST(stream));
}

int fp_quantize_fp8(const void* in, int in_dtype, int64_t n, float scale, void* out, fp_stream_t stream) {
  FP_REQUIRE(in && out, "fp_quantize_fp8: null pointer");
  FP_REQUIRE(in_dtype == FP_F32 || in_dtype == FP_BF16, "fp_quantize_fp8: input must be fp32 or bf16");
  FP_REQUIRE(scale > 0.f, "fp_quantize_fp8: scale must be positive");
  return quantize_fp8_launch(in, in_dtype, n, scale, out, ST(stream));
}

int fp_gemm_f32(const float* A, int lda, const float* W, int ldw, int M, int N, int K, const float* bias,
                const float* gamma, float* out, int ldo, int epilogue, fp_stream_t stream) {
  FP_REQUIRE(A && W && out, "fp_gemm_f32: null pointer");
  FP_REQUIRE(epilogue == F32_EPI_STORE || epilogue == F32_EPI_BIAS || epilogue == F32_EPI_BIAS_GELU ||
                 epilogue == F32_EPI_LS_RESID || epilogue == F32_EPI_SUB_VEC || epilogue == F32_EPI_SWIGLU,
             "fp_gemm_f32: epilogue %d is not available through this entry point", epilogue);
  FP_REQUIRE(epilogue != F32_EPI_LS_RESID || gamma, "fp_gemm_f32: gamma required");
  if (M == 0) return FP_OK;
  F32TileArgs a = zero_tile_args();
  a.A = A; a.lda = lda; a.B = W; a.ldb = ldw; a.K = K; a.M = M; a.N = N;
  a.out = out; a.ldo = ldo; a.bias = bias; a.gamma = gamma;
  return f32_tile_launch(epilogue, a, M, N, 1, ST(stream));
}

int fp_attention(const void* qkv, int ld_qkv, void* out, int ld_out, int B, int n_tok,
                 int dim, int heads, int dtype, fp_stream_t stream) {
  FP_REQUIRE(qkv && out, "fp_attention: null pointer");
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.qkv = qkv; a.ld_qkv = ld_qkv; a.out = out; a.ld_out = ld_out;
  a.batch = B; a.n_tok = n_tok; a.dim = dim; a.heads = heads;
  a.variant = (dtype >> 8) & 0xff;  // tuning / test bits: which bf16 work split runs (FP_ATTN_VARIANT_*)
  return attn_launch(a, dtype & 0xff, ST(stream));
}

int fp_convert_f32_to_bf16(const float* in, void* out, int64_t n, fp_stream_t stream) {
  FP_REQUIRE(in && out, "fp_convert_f32_to_bf16: null pointer");
  return convert_f32_to_bf16_launch(in, out, n, ST(stream));
}

int fp_warp_crops(const void* src, int n_src, int src_h, int src_w, int channels, int mode, const int32_t* src_index,
                  const double* params, int batch, int out_h, int out_w, int depth_check, void* out, float* map_out,
                  fp_stream_t stream) {
  FP_REQUIRE(src && params && out, "fp_warp_crops: null pointer");
  FP_REQUIRE(mode == FP_WARP_LINEAR || mode == FP_WARP_NEAREST, "fp_warp_crops: unknown mode %d", mode);
  FP_REQUIRE(n_src >= 1 && src_h >= 1 && src_w >= 1 && batch >= 1 && out_h >= 1 && out_w >= 1, "fp_warp_crops: empty problem");
  FP_REQUIRE(mode == FP_WARP_LINEAR ? channels >= 1 : channels == 1, "fp_warp_crops: a nearest-mode source is a single-channel u8 mask");
  FP_REQUIRE(src_index || batch <= n_src, "fp_warp_crops: %d crops but %d source images and no src_index", batch, n_src);
  WarpArgs a{src, n_src, src_h, src_w, channels, mode, src_index, params, batch, out_h, out_w, depth_check, out, map_out};
  return launch_warp_crops(a, ST(stream));
}

// ------------------------------------------------------------------ ViT forward (launch sequence in C++)
}  // extern "C"

namespace {
enum { VIT_FULL = 0, VIT_PREFIX = 1, VIT_LAST_SELECTED = 2 };
struct VitSelection { const int32_t* rows; const int32_t* off; int num, max_per_img; };

// VIT_FULL: embedding + blocks 0..layer.  VIT_PREFIX: embedding + blocks 0..layer-1, leaving what block `layer` starts from:
// the fp32 stream ws->x, its bf16 copy and the LayerNorm row sums -- or, with ws->xl set (the extractor's default, resid_hilo=True) and layer > 0,
// the (xb, xl) pair and the row sums ONLY: ws->x is then stale (it holds the token embedding) and must not be read by a caller.  VIT_LAST_SELECTED: block `layer` alone, computed for the selected
// tokens only (queries of the attention, rows of proj / fc1 / fc2) on top of a VIT_PREFIX run -- keys and values are all tokens.
// first_block > 0 (precision schedules, fp_vit_forward_blocks): no embedding -- the fp32 stream of blocks 0..first_block-1 is already in ws->x (another model's
// blocks left it there) and blocks first_block..layer of THIS model continue from it (a folded-LayerNorm model starts its chain -- 16-bit copy, row sums -- from that stream).
int vit_forward_impl(const fp_vit_model* m, const fp_vit_workspace* ws, const float* images, int B, int H, int W, int layer, int mode,
                     const VitSelection* sel, fp_stream_t stream, int first_block = 0) {
  FP_REQUIRE(m && ws && (images || mode == VIT_LAST_SELECTED || first_block > 0) && m->blocks, "fp_vit_forward: null pointer");
  FP_REQUIRE(first_block == 0 || (first_block >= 1 && first_block <= layer), "fp_vit_forward_blocks: first_block %d out of range (layer %d)", first_block, layer);
  FP_REQUIRE(layer >= -1 && layer < m->depth, "fp_vit_forward: layer %d out of range (depth %d)", layer, m->depth);  // -1: token embedding only
  const int pstride = m->patch_stride > 0 ? m->patch_stride : m->patch;  // the conv stride of the patch embedding (dinov2_utils.py:364-389)
  FP_REQUIRE(pstride != m->patch || (H % m->patch == 0 && W % m->patch == 0), "fp_vit_forward: image size must be a multiple of the patch size");
  FP_REQUIRE(H >= m->patch && W >= m->patch && (pstride == m->patch || mode == VIT_FULL), "fp_vit_forward: image smaller than a patch, or token selection with stride != patch size");
  const int D = m->dim, np = (1 + (H - m->patch) / pstride) * (1 + (W - m->patch) / pstride), ntok = 1 + m->registers + np;
  const int Mtok = B * ntok, Mp = B * np;
  FP_REQUIRE(ws->m_pad >= Mtok && ws->m_pad % 128 == 0, "fp_vit_forward: workspace m_pad (%d) too small for %d tokens or not a multiple of 128", ws->m_pad, Mtok);
  FP_REQUIRE(ws->m_patch_pad >= Mp && ws->m_patch_pad % 128 == 0, "fp_vit_forward: workspace m_patch_pad too small");
  FP_REQUIRE(ws->patches && ws->x && ws->y && ws->qkv && ws->h, "fp_vit_forward: workspace buffer missing");
  hipStream_t st = ST(stream);
  const bool f8 = m->weight_dtype == FP_DTYPE_FP8;  // e4m3 block matrices; activations and patch embed stay bf16
  const bool sx = m->weight_dtype == FP_DTYPE_F16F8;  // f16f8 rows (common.hpp): the split mode with the cross terms on the fp8 pipe
  const bool sp = m->weight_dtype == FP_DTYPE_F16X3 || sx;  // split-fp16 operands everywhere (near-exact modes): rows of 2 x halves
  const bool h16 = m->weight_dtype == FP_DTYPE_F16;  // the "f16" mode: the bf16 pipeline (folded LayerNorms, (hi, lo) stream) on IEEE fp16 operands
  const bool bf = m->weight_dtype == FP_DTYPE_BF16 || f8 || h16;
  const int adt = sx ? FP_DTYPE_F16F8 : (sp ? FP_DTYPE_F16X3 : (h16 ? FP_DTYPE_F16 : (bf ? FP_DTYPE_BF16 : FP_DTYPE_F32)));
  const int attn_dt = h16 ? FP_DTYPE_F16 : FP_DTYPE_BF16;
  FP_REQUIRE(!h16 || m->ln_fold, "fp_vit_forward: weight_dtype FP_F16 runs the folded-LayerNorm pipeline only (ln_fold = 1, dim %% 128 == 0)");
  FP_REQUIRE(!sx || (D % 64 == 0 && m->hidden % 64 == 0 && m->patch_k_pad % 64 == 0), "fp_vit_forward: the f16f8 mode needs dim, hidden and patch_k_pad to be multiples of 64");
  const int em = sp ? 2 : 1;                          // stored elements per logical element of an operand row
  FP_REQUIRE(!sp || m->patch_acc_scale > 0.f, "fp_vit_forward: the f16x3 mode needs patch_acc_scale");
  FP_REQUIRE(!f8 || (ws->a8 && ws->m_pad % 256 == 0), "fp_vit_forward: the fp8 mode needs workspace a8 and m_pad %% 256 == 0");
  FP_REQUIRE(!f8 || ((ws->ld_y == 0 || (ws->ld_y >= m->dim && ws->ld_y % 16 == 0)) && (ws->ld_h == 0 || (ws->ld_h >= m->hidden && ws->ld_h % 16 == 0)) &&
                     (m->ld_w_dim == 0 || (m->ld_w_dim >= m->dim && m->ld_w_dim % 16 == 0)) && (m->ld_w_hidden == 0 || (m->ld_w_hidden >= m->hidden && m->ld_w_hidden % 16 == 0))),
             "fp_vit_forward: fp8 row strides (bytes) must cover the row and keep 16-byte alignment");

  // tokens: [cls + pos0 | registers | patch_embed(x) + pos]
  if (mode != VIT_LAST_SELECTED && first_block == 0) {
  if (pstride == m->patch) TRY(patchify_launch(images, B, H, W, m->patch, ws->patches, em * m->patch_k_pad, adt, st, FP_SPLIT_SCALE_ACT));
  else TRY(patchify_strided_launch(images, B, H, W, m->patch, pstride, ws->patches, em * m->patch_k_pad, adt, st, FP_SPLIT_SCALE_ACT));
  TRY(prefix_tokens_launch(m->prefix, 1 + m->registers, D, ws->x, B, ntok, st));
  if (sp) {
    GemmBf16Args g;
    memset(&g, 0, sizeof(g));
    g.A = reinterpret_cast<const __bf16*>(ws->patches); g.lda = 2 * m->patch_k_pad;
    g.W = reinterpret_cast<const __bf16*>(m->patch_w); g.ldw = 2 * m->patch_k_pad;
    g.M = ws->m_patch_pad; g.N = D; g.K = m->patch_k_pad; g.M_valid = Mp; g.bias = m->patch_b;
    g.out = ws->x; g.ldo = D; g.pos = m->pos_patch; g.tok_np = np; g.tok_n = ntok; g.tok_skip = 1 + m->registers;
    g.acc_scale = m->patch_acc_scale;
    TRY(sx ? gemm_splitx_launch(GEMM_EPI_TOKENS_F32, g, st) : gemm_split_launch(GEMM_EPI_TOKENS_F32, g, st));
  } else if (bf) {
    GemmBf16Args g;
    memset(&g, 0, sizeof(g));
    g.A = reinterpret_cast<const __bf16*>(ws->patches); g.lda = m->patch_k_pad;
    g.W = reinterpret_cast<const __bf16*>(m->patch_w); g.ldw = m->patch_k_pad;
    g.M = ws->m_patch_pad; g.N = D; g.K = m->patch_k_pad; g.M_valid = Mp; g.bias = m->patch_b;
    g.out = ws->x; g.ldo = D; g.pos = m->pos_patch; g.tok_np = np; g.tok_n = ntok; g.tok_skip = 1 + m->registers;
    TRY(h16 ? gemm_f16_launch(GEMM_EPI_TOKENS_F32, g, st) : gemm_bf16_launch(GEMM_EPI_TOKENS_F32, g, st));
  } else {
    F32TileArgs a = zero_tile_args();
    a.A = reinterpret_cast<const float*>(ws->patches); a.lda = m->patch_k_pad;
    a.B = reinterpret_cast<const float*>(m->patch_w); a.ldb = m->patch_k_pad; a.K = m->patch_k_pad; a.M = Mp; a.N = D;
    a.out = ws->x; a.ldo = D; a.bias = m->patch_b; a.pos = m->pos_patch; a.tok_np = np; a.tok_n = ntok;
    a.tok_skip = 1 + m->registers;
    TRY(f32_tile_launch(F32_EPI_TOKENS, a, Mp, D, 1, st));
  }
  }

  // row strides of the bf16 / fp32 operands (fp8 mode: dense)
  const int ldy = (!f8 && ws->ld_y) ? ws->ld_y : em * D, ldh = (!f8 && ws->ld_h) ? ws->ld_h : em * m->hidden;
  const int ldq = ws->ld_qkv ? ws->ld_qkv : em * 3 * D;
  const int ldwd = (!f8 && m->ld_w_dim) ? m->ld_w_dim : em * D, ldwh = (!f8 && m->ld_w_hidden) ? m->ld_w_hidden : em * m->hidden;
  FP_REQUIRE(ldy >= em * D && ldh >= em * m->hidden && ldwd >= em * D && ldwh >= em * m->hidden && ldy % 8 == 0 && ldh % 8 == 0 && ldwd % 8 == 0 && ldwh % 8 == 0 &&
                 ldq >= em * 3 * D && ldq % 8 == 0,
             "fp_vit_forward: operand row strides must cover the row and keep 16-byte alignment");
  LayerNormArgs ln;
  memset(&ln, 0, sizeof(ln));
  ln.sat = ws->sat;
  ln.x = ws->x; ln.ld_x = D; ln.eps = 1e-6f; ln.out = ws->y; ln.ld_out = ldy; ln.out_dtype = adt;
  ln.dim = D; ln.out_rows = Mtok; ln.out_rows_per_img = Mtok; ln.in_rows_per_img = Mtok; ln.in_skip = 0;
  AttnArgs at;
  memset(&at, 0, sizeof(at));
  at.qkv = ws->qkv; at.ld_qkv = ldq; at.out = ws->y; at.ld_out = ldy;
  at.batch = B; at.n_tok = ntok; at.dim = D; at.heads = m->heads;
  at.sat = ws->sat;

  // LayerNorm folded into the GEMMs (bf16 blocks): see fp_vit_model.ln_fold.  The chain starts from the token embedding.
  const bool fold = m->ln_fold && bf && !f8;
  int ln_parts = 1;
  float2* stats = reinterpret_cast<float2*>(ws->stats);
  if (fold) {
    FP_REQUIRE(ws->xb && ws->stats, "fp_vit_forward: ln_fold needs workspace xb and stats");
    FP_REQUIRE(D % 128 == 0, "fp_vit_forward: ln_fold needs dim %% 128 == 0");
    ln_parts = D / 128;  // one partial sum per 128-column group of the residual GEMMs, whatever tile they run with
    if (layer >= 0 && mode != VIT_LAST_SELECTED) TRY(rowstats_cast_launch(ws->x, Mtok, D, ws->xb, ldy, stats, ws->m_pad, ln_parts, st, ws->xl, h16));
  }
  // Blocks in FRONT of the hooked one keep the residual stream as (hi, lo) bf16 arrays (ws->xb, ws->xl) instead of fp32 + a bf16 copy: the
  // residual GEMMs then read 4 and write 4 bytes per element instead of 4 + 6 (hi IS the next GEMM's A operand).  16 mantissa bits per update
  // against the 8 of the bf16 operands everything is multiplied in.  The hooked block itself runs on an fp32 stream rebuilt from the pair
  // (all rows, or the selected rows only), so the engine's token-selected form and the full form stay bit-identical.
  const bool hilo = fold && ws->xl != nullptr;
  FP_REQUIRE(mode == VIT_FULL || fold || sp || f8, "fp_vit_forward_prefix / fp_vit_block_selected: bf16 model with ln_fold, an fp8 or an f16x3 model");
  float2* ln_row = stats + (size_t)ln_parts * ws->m_pad;  // (rstd, mean * rstd) per row, behind the partial-sum slots
  int rows_valid = Mtok, rows_pad = ws->m_pad;  // the selected tail of the hooked block narrows these to the compact rows
  // (the residual GEMM files its partial sums with a stride of ITS row count: rows_pad)
  auto finalize = [&]() -> int { return ln_finalize_launch(stats, ln_parts, rows_pad, rows_valid, D, 1e-6f, ln_row, st); };
  auto gemm = [&](const void* A, int lda, const void* Wt, int ldw, int N, int K, const float* bias, const float* gamma, void* out, int ldo, int epi,
                  const float* colsum, bool produce, float w_inv_scale = 0.f) -> int {
    GemmBf16Args g;
    memset(&g, 0, sizeof(g));
    g.A = reinterpret_cast<const __bf16*>(A); g.lda = lda; g.W = reinterpret_cast<const __bf16*>(Wt); g.ldw = ldw;
    g.M = rows_pad; g.N = N; g.K = K; g.M_valid = rows_valid; g.bias = bias; g.gamma = gamma; g.out = out; g.ldo = ldo;
    g.no_tall = (m->flags & FP_VIT_NO_TALL_TILES) ? 1 : 0;
    g.acc_scale = h16 ? w_inv_scale : 0.f;   // FP_F16: 1 / (power-of-two scale of this weight matrix), fp_vit_block.act_scale[j]; 0 = unscaled
    if (colsum) { g.ln_stats = ln_row; g.ln_parts = ln_parts; g.ln_eps = 1e-6f; g.colsum = colsum; }
    if (produce) { g.xb = reinterpret_cast<__bf16*>(ws->xb); g.ld_xb = ldy; g.stats_out = stats; g.xl = reinterpret_cast<__bf16*>(ws->xl); }
    return h16 ? gemm_f16_launch(epi, g, st) : gemm_bf16_launch(epi, g, st);
  };

  const int i_first = mode == VIT_LAST_SELECTED ? layer : first_block, i_last = mode == VIT_PREFIX ? layer - 1 : layer;
  for (int i = i_first; i <= i_last; ++i) {
    const fp_vit_block& b = m->blocks[i];
    if (fold && mode == VIT_LAST_SELECTED) {
      // The hooked block for the selected tokens only.  K and V need every token: LayerNorm constants and the qkv GEMM run on
      // all rows (its Q columns of unselected rows are the only wasted work); attention takes its queries through the index
      // list and writes compact rows; from there on every operand has num_sel rows.  Row r of every compact buffer is
      // token sel->rows[r]; the per-row arithmetic (GEMM chains, 128-column stat groups) does not depend on where a row sits,
      // so the selected rows carry the bits the full block would have given them.
      FP_REQUIRE(b.qkv_colsum && b.fc1_colsum, "fp_vit_forward: ln_fold needs the column sums of qkv_w / fc1_w");
      TRY(finalize());
      TRY(gemm(ws->xb, ldy, b.qkv_w, ldwd, 3 * D, D, b.qkv_b, nullptr, ws->qkv, ldq, GEMM_EPI_BIAS_BF16, b.qkv_colsum, false, b.act_scale[0]));
      AttnArgs as = at;
      as.sel_rows = sel->rows; as.sel_off = sel->off; as.max_sel = sel->max_per_img;
      TRY(attn_launch(as, attn_dt, st));
      float* xs = reinterpret_cast<float*>(ws->qkv);  // qkv is dead after the attention: [num_sel, D] fp32 rows of the stream
      if (hilo && layer > 0) TRY(hilo_rows_launch(ws->xb, ws->xl, ldy, sel->rows, sel->num, D, xs, st, h16));   // the blocks in front left (hi, lo) pairs
      else TRY(gather_rows_launch(ws->x, sel->rows, sel->num, D, xs, st));
      rows_valid = sel->num;
      rows_pad = (sel->num + 255) / 256 * 256 < ws->m_pad ? (sel->num + 255) / 256 * 256 : ws->m_pad;
      TRY(gemm(ws->y, ldy, b.proj_w, ldwd, D, D, b.proj_b, nullptr, xs, D, GEMM_EPI_RESID_F32, nullptr, true, b.act_scale[1]));
      TRY(finalize());
      if (m->ffn_swiglu)
        TRY(gemm(ws->xb, ldy, b.fc1_w, ldwd, 2 * m->hidden, D, b.fc1_b, nullptr, ws->h, ldh, GEMM_EPI_SWIGLU_BF16, b.fc1_colsum, false, b.act_scale[2]));
      else
        TRY(gemm(ws->xb, ldy, b.fc1_w, ldwd, m->hidden, D, b.fc1_b, nullptr, ws->h, ldh, GEMM_EPI_GELU_BF16, b.fc1_colsum, false, b.act_scale[2]));
      TRY(gemm(ws->h, ldh, b.fc2_w, ldwh, D, m->hidden, b.fc2_b, nullptr, xs, D, GEMM_EPI_RESID_F32, nullptr, false, b.act_scale[3]));
      continue;
    }
    if (fold) {
      // x += ls1 * proj(attn(ln1(x))): qkv reads bf16(x) and normalises in its epilogue; proj refreshes bf16(x) + row sums
      FP_REQUIRE(b.qkv_colsum && b.fc1_colsum, "fp_vit_forward: ln_fold needs the column sums of qkv_w / fc1_w");
      const bool pair = hilo && i < layer;            // a block in front of the hooked one: (hi, lo) stream
      const int resid = pair ? GEMM_EPI_RESID_HILO : GEMM_EPI_RESID_F32;
      if (hilo && i == layer && i > 0) TRY(hilo_rows_launch(ws->xb, ws->xl, ldy, nullptr, Mtok, D, ws->x, st, h16));   // VIT_FULL: the hooked block's fp32 stream, all rows
      TRY(finalize());
      TRY(gemm(ws->xb, ldy, b.qkv_w, ldwd, 3 * D, D, b.qkv_b, nullptr, ws->qkv, ldq, GEMM_EPI_BIAS_BF16, b.qkv_colsum, false, b.act_scale[0]));
      TRY(attn_launch(at, attn_dt, st));
      TRY(gemm(ws->y, ldy, b.proj_w, ldwd, D, D, b.proj_b, nullptr, ws->x, D, resid, nullptr, true, b.act_scale[1]));
      // x += ls2 * fc2(act(fc1(ln2(x))))
      TRY(finalize());
      if (m->ffn_swiglu)
        TRY(gemm(ws->xb, ldy, b.fc1_w, ldwd, 2 * m->hidden, D, b.fc1_b, nullptr, ws->h, ldh, GEMM_EPI_SWIGLU_BF16, b.fc1_colsum, false, b.act_scale[2]));
      else
        TRY(gemm(ws->xb, ldy, b.fc1_w, ldwd, m->hidden, D, b.fc1_b, nullptr, ws->h, ldh, GEMM_EPI_GELU_BF16, b.fc1_colsum, false, b.act_scale[2]));
      TRY(gemm(ws->h, ldh, b.fc2_w, ldwh, D, m->hidden, b.fc2_b, nullptr, ws->x, D, resid, nullptr, i < layer, b.act_scale[3]));
      continue;
    }
    // x += ls1 * proj(attn(ln1(x)))
    ln.weight = b.ln1_w; ln.bias = b.ln1_b;
    if (sp) {
      // f16x3 block: every GEMM / attention operand is a split-fp16 row written by the kernel in front of it (LayerNorm, the
      // qkv / GELU / SwiGLU epilogues, the attention kernel) with a fixed power-of-two scale; b.act_scale[j] = 1 / (scale of the
      // input x scale of the matrix) undoes both in the epilogue of GEMM j.  The residual stream, LayerNorm and softmax are fp32.
      // mode VIT_LAST_SELECTED (the hooked block for the selected tokens only, as in the bf16 branch above): LayerNorm 1 and the qkv
      // GEMM on all rows (keys / values need every token), attention takes its queries through the index list and writes compact
      // rows, and from there on every operand has num_sel rows: the residual rows are gathered into the dead qkv buffer, proj /
      // LayerNorm 2 / fc1 / fc2 run on them.  Per-row arithmetic does not depend on where a row sits: the same bits as the full block.
      const bool selected = mode == VIT_LAST_SELECTED;
      auto sgemm = [&](const void* A, int lda, const void* Wt, int ldw, int N, int K, const float* bias, const float* gamma, void* out, int ldo, int epi,
                       float acc_scale, float out_scale) -> int {
        GemmBf16Args g;
        memset(&g, 0, sizeof(g));
        g.A = reinterpret_cast<const __bf16*>(A); g.lda = lda; g.W = reinterpret_cast<const __bf16*>(Wt); g.ldw = ldw;
        g.M = rows_pad; g.N = N; g.K = K; g.M_valid = rows_valid; g.bias = bias; g.gamma = gamma; g.out = out; g.ldo = ldo;
        g.acc_scale = acc_scale; g.out_scale = out_scale; g.sat = ws->sat;
        return sx ? gemm_splitx_launch(epi, g, st) : gemm_split_launch(epi, g, st);
      };
      ln.out_scale = FP_SPLIT_SCALE_ACT;
      TRY(layernorm_launch(ln, st));
      TRY(sgemm(ws->y, ldy, b.qkv_w, ldwd, 3 * D, D, b.qkv_b, nullptr, ws->qkv, ldq, GEMM_EPI_BIAS_BF16, b.act_scale[0], FP_SPLIT_SCALE_QKV));
      AttnArgs as = at;
      as.in_scale = FP_SPLIT_SCALE_QKV; as.out_scale = FP_SPLIT_SCALE_ACT;
      as.out_fmt = sx ? 1 : 0;   // f16f8: q | k | v stay split-fp16 rows (the attention's own three-MFMA products), its output is proj's f16f8 operand
      float* xr = ws->x;  // the residual rows the rest of the block updates
      if (selected) {
        as.sel_rows = sel->rows; as.sel_off = sel->off; as.max_sel = sel->max_per_img;
        TRY(attn_launch(as, FP_DTYPE_F16X3, st));
        xr = reinterpret_cast<float*>(ws->qkv);  // qkv is dead after the attention: [num_sel, D] fp32 rows of the stream
        TRY(gather_rows_launch(ws->x, sel->rows, sel->num, D, xr, st));
        rows_valid = sel->num;
        rows_pad = (sel->num + 255) / 256 * 256 < ws->m_pad ? (sel->num + 255) / 256 * 256 : ws->m_pad;
        ln.x = xr; ln.out_rows = rows_valid; ln.out_rows_per_img = rows_valid; ln.in_rows_per_img = rows_valid;
      } else {
        TRY(attn_launch(as, FP_DTYPE_F16X3, st));
      }
      TRY(sgemm(ws->y, ldy, b.proj_w, ldwd, D, D, b.proj_b, b.ls1, xr, D, GEMM_EPI_LS_RESID_F32, b.act_scale[1], 0.f));
      ln.weight = b.ln2_w; ln.bias = b.ln2_b;
      TRY(layernorm_launch(ln, st));
      if (m->ffn_swiglu)
        TRY(sgemm(ws->y, ldy, b.fc1_w, ldwd, 2 * m->hidden, D, b.fc1_b, nullptr, ws->h, ldh, GEMM_EPI_SWIGLU_BF16, b.act_scale[2], FP_SPLIT_SCALE_HID));
      else
        TRY(sgemm(ws->y, ldy, b.fc1_w, ldwd, m->hidden, D, b.fc1_b, nullptr, ws->h, ldh, GEMM_EPI_GELU_BF16, b.act_scale[2], FP_SPLIT_SCALE_HID));
      TRY(sgemm(ws->h, ldh, b.fc2_w, ldwh, D, m->hidden, b.fc2_b, b.ls2, xr, D, GEMM_EPI_LS_RESID_F32, b.act_scale[3], 0.f));
      continue;
    }
    if (f8) {
      // fp8 block: every GEMM input is produced as e4m3 bytes by the kernel in front of it -- LayerNorm, attention and
      // the GELU / SwiGLU epilogue quantise with the block's static scales on their way out (ws->a8; the hidden
      // activations reuse ws->h as a byte buffer) -- so the four GEMMs run on the fp8 MFMA with no extra pass.
      // row strides in bytes (= fp8 elements): a8 [m_pad, ld8y], hidden bytes [m_pad, ld8h], matrices [N, ld8wd / ld8wh]
      const int ld8y = ws->ld_y ? ws->ld_y : D, ld8h = ws->ld_h ? ws->ld_h : m->hidden;
      const int ld8wd = m->ld_w_dim ? m->ld_w_dim : D, ld8wh = m->ld_w_hidden ? m->ld_w_hidden : m->hidden;
      // mode VIT_LAST_SELECTED (the hooked block for the selected tokens only, as in the bf16 and f16x3 branches): LayerNorm 1 and the qkv GEMM on
      // all rows (keys / values need every token), attention takes its queries through the index list and writes compact e4m3 rows, the residual
      // rows are gathered into the dead qkv buffer, and proj / LayerNorm 2 / fc1 / fc2 run on num_sel rows.  Per-row arithmetic (static scales,
      // k-ordered GEMM chains) does not depend on where a row sits: the selected rows carry the bits the full block would have given them.
      const bool selected = mode == VIT_LAST_SELECTED;
      LayerNormArgs l8 = ln;
      l8.out = ws->a8; l8.ld_out = ld8y; l8.out_dtype = FP_DTYPE_FP8; l8.out_scale = b.act_scale[0];
      TRY(layernorm_launch(l8, st));
      TRY(gemm_fp8_impl(ws->a8, ld8y, b.qkv_w, ld8wd, ws->m_pad, 3 * D, D, Mtok, b.qkv_b, b.qkv_s, ws->qkv, ldq, GEMM_EPI_BIAS_BF16, 0.f, ws->sat, stream, (m->flags & FP_VIT_NO_TALL_TILES) ? 1 : 0));
      AttnArgs a8 = at;
      a8.out = ws->a8; a8.ld_out = ld8y; a8.out_fp8_scale = b.act_scale[1];
      float* xr = ws->x;  // the residual rows the rest of the block updates
      int rv = Mtok, rp = ws->m_pad;
      if (selected) {
        a8.sel_rows = sel->rows; a8.sel_off = sel->off; a8.max_sel = sel->max_per_img;
        TRY(attn_launch(a8, FP_DTYPE_BF16, st));
        xr = reinterpret_cast<float*>(ws->qkv);  // qkv is dead after the attention: [num_sel, D] fp32 rows of the stream
        TRY(gather_rows_launch(ws->x, sel->rows, sel->num, D, xr, st));
        rv = sel->num;
        rp = (sel->num + 255) / 256 * 256 < ws->m_pad ? (sel->num + 255) / 256 * 256 : ws->m_pad;
        l8.x = xr; l8.out_rows = rv; l8.out_rows_per_img = rv; l8.in_rows_per_img = rv;
      } else {
        TRY(attn_launch(a8, FP_DTYPE_BF16, st));
      }
      TRY(gemm_fp8_impl(ws->a8, ld8y, b.proj_w, ld8wd, rp, D, D, rv, b.proj_b, b.proj_s, xr, D, GEMM_EPI_LS_RESID_F32, 0.f, ws->sat, stream, (m->flags & FP_VIT_NO_TALL_TILES) ? 1 : 0));
      l8.weight = b.ln2_w; l8.bias = b.ln2_b; l8.out_scale = b.act_scale[2];
      TRY(layernorm_launch(l8, st));
      if (m->ffn_swiglu)
        TRY(gemm_fp8_impl(ws->a8, ld8y, b.fc1_w, ld8wd, rp, 2 * m->hidden, D, rv, b.fc1_b, b.fc1_s, ws->h, ld8h, GEMM_EPI_SWIGLU_BF16, b.act_scale[3], ws->sat, stream, (m->flags & FP_VIT_NO_TALL_TILES) ? 1 : 0));
      else
        TRY(gemm_fp8_impl(ws->a8, ld8y, b.fc1_w, ld8wd, rp, m->hidden, D, rv, b.fc1_b, b.fc1_s, ws->h, ld8h, GEMM_EPI_GELU_BF16, b.act_scale[3], ws->sat, stream, (m->flags & FP_VIT_NO_TALL_TILES) ? 1 : 0));
      TRY(gemm_fp8_impl(ws->h, ld8h, b.fc2_w, ld8wh, rp, D, m->hidden, rv, b.fc2_b, b.fc2_s, xr, D, GEMM_EPI_LS_RESID_F32, 0.f, ws->sat, stream, (m->flags & FP_VIT_NO_TALL_TILES) ? 1 : 0));
      continue;
    }
    TRY(layernorm_launch(ln, st));
    if (bf) {
      TRY(fp_gemm_bf16(ws->y, ldy, b.qkv_w, ldwd, ws->m_pad, 3 * D, D, Mtok, b.qkv_b, nullptr, ws->qkv, ldq, GEMM_EPI_BIAS_BF16, stream));
      TRY(attn_launch(at, FP_DTYPE_BF16, st));
      TRY(fp_gemm_bf16(ws->y, ldy, b.proj_w, ldwd, ws->m_pad, D, D, Mtok, b.proj_b, b.ls1, ws->x, D, GEMM_EPI_LS_RESID_F32, stream));
    } else {
      TRY(fp_gemm_f32((const float*)ws->y, ldy, (const float*)b.qkv_w, ldwd, Mtok, 3 * D, D, b.qkv_b, nullptr, (float*)ws->qkv, ldq, F32_EPI_BIAS, stream));
      TRY(attn_launch(at, FP_DTYPE_F32, st));
      TRY(fp_gemm_f32((const float*)ws->y, ldy, (const float*)b.proj_w, ldwd, Mtok, D, D, b.proj_b, b.ls1, ws->x, D, F32_EPI_LS_RESID, stream));
    }
    // x += ls2 * fc2(gelu(fc1(ln2(x))))
    ln.weight = b.ln2_w; ln.bias = b.ln2_b;
    TRY(layernorm_launch(ln, st));
    if (bf) {
      if (m->ffn_swiglu)  // fc1_w = w12 with rows interleaved (x1_j, x2_j); h = silu(x1) * x2
        TRY(fp_gemm_bf16(ws->y, ldy, b.fc1_w, ldwd, ws->m_pad, 2 * m->hidden, D, Mtok, b.fc1_b, nullptr, ws->h, ldh, GEMM_EPI_SWIGLU_BF16, stream));
      else
        TRY(fp_gemm_bf16(ws->y, ldy, b.fc1_w, ldwd, ws->m_pad, m->hidden, D, Mtok, b.fc1_b, nullptr, ws->h, ldh, GEMM_EPI_GELU_BF16, stream));
      TRY(fp_gemm_bf16(ws->h, ldh, b.fc2_w, ldwh, ws->m_pad, D, m->hidden, Mtok, b.fc2_b, b.ls2, ws->x, D, GEMM_EPI_LS_RESID_F32, stream));
    } else {
      if (m->ffn_swiglu)
        TRY(fp_gemm_f32((const float*)ws->y, ldy, (const float*)b.fc1_w, ldwd, Mtok, 2 * m->hidden, D, b.fc1_b, nullptr, (float*)ws->h, ldh, F32_EPI_SWIGLU, stream));
      else
        TRY(fp_gemm_f32((const float*)ws->y, ldy, (const float*)b.fc1_w, ldwd, Mtok, m->hidden, D, b.fc1_b, nullptr, (float*)ws->h, ldh, F32_EPI_BIAS_GELU, stream));
      TRY(fp_gemm_f32((const float*)ws->h, ldh, (const float*)b.fc2_w, ldwh, Mtok, D, m->hidden, b.fc2_b, b.ls2, ws->x, D, F32_EPI_LS_RESID, stream));
    }
  }
  return FP_OK;
}
}  // namespace

extern "C" {

int fp_vit_forward(const fp_vit_model* m, const fp_vit_workspace* ws, const float* images, int B, int H, int W,
                   int layer, fp_stream_t stream) {
  return vit_forward_impl(m, ws, images, B, H, W, layer, VIT_FULL, nullptr, stream);
}

int fp_vit_forward_prefix(const fp_vit_model* m, const fp_vit_workspace* ws, const float* images, int B, int H, int W,
                          int layer, fp_stream_t stream) {
  FP_REQUIRE(layer >= 0, "fp_vit_forward_prefix: layer must be >= 0");
  return vit_forward_impl(m, ws, images, B, H, W, layer, VIT_PREFIX, nullptr, stream);
}

int fp_vit_forward_blocks(const fp_vit_model* m, const fp_vit_workspace* ws, int B, int H, int W, int first_block, int layer, int prefix_only, fp_stream_t stream) {
  FP_REQUIRE(first_block >= 1, "fp_vit_forward_blocks: first_block must be >= 1 (fp_vit_forward runs the embedding and every block)");
  return vit_forward_impl(m, ws, nullptr, B, H, W, layer, prefix_only ? VIT_PREFIX : VIT_FULL, nullptr, stream, first_block);
}

int fp_vit_stream_f32(const fp_vit_model* m, const fp_vit_workspace* ws, int B, int H, int W, int layer, float* out, fp_stream_t stream) {
  FP_REQUIRE(m && ws && out && layer >= 0, "fp_vit_stream_f32: null pointer");
  const int pstride = m->patch_stride > 0 ? m->patch_stride : m->patch;
  const int np = (1 + (H - m->patch) / pstride) * (1 + (W - m->patch) / pstride), rows = B * (1 + m->registers + np), D = m->dim;
  const bool h16 = m->weight_dtype == FP_DTYPE_F16;
  const bool pair = m->ln_fold && (m->weight_dtype == FP_DTYPE_BF16 || h16) && ws->xl != nullptr && layer > 0;   // what fp_vit_forward_prefix(layer) left behind
  if (pair) return hilo_rows_launch(ws->xb, ws->xl, ws->ld_y ? ws->ld_y : D, nullptr, rows, D, out, ST(stream), h16);
  if (out != ws->x) HIP_TRY(hipMemcpyAsync(out, ws->x, (size_t)rows * D * 4, hipMemcpyDeviceToDevice, ST(stream)), "fp_vit_stream_f32: copy");
  return FP_OK;
}

int fp_vit_block_selected(const fp_vit_model* m, const fp_vit_workspace* ws, int B, int H, int W, int layer,
                          const int32_t* sel_rows, const int32_t* sel_off, int num_sel, int max_sel_per_img, fp_stream_t stream) {
  FP_REQUIRE(sel_rows && sel_off, "fp_vit_block_selected: null pointer");
  FP_REQUIRE(layer >= 0 && num_sel >= 1 && max_sel_per_img >= 1 && max_sel_per_img <= num_sel, "fp_vit_block_selected: bad sizes (layer %d, %d selected, at most %d per image)",
             layer, num_sel, max_sel_per_img);
  FP_REQUIRE(m && ws && num_sel <= ws->m_pad, "fp_vit_block_selected: more selected tokens than workspace rows");
  const VitSelection sel{sel_rows, sel_off, num_sel, max_sel_per_img};
  return vit_forward_impl(m, ws, nullptr, B, H, W, layer, VIT_LAST_SELECTED, &sel, stream);
}

int fp_vit_features(const fp_vit_model* m, const fp_vit_workspace* ws, int B, int n_patches, int apply_norm,
                    float* fmap, float* cls, fp_stream_t stream) {
  FP_REQUIRE(m && ws && fmap, "fp_vit_features: null pointer");
  const int D = m->dim, ntok = 1 + m->registers + n_patches;
  hipStream_t st = ST(stream);
  if (apply_norm) {
    LayerNormArgs ln;
    memset(&ln, 0, sizeof(ln));
    ln.x = ws->x; ln.ld_x = D; ln.weight = m->norm_w; ln.bias = m->norm_b; ln.eps = 1e-6f;
    ln.out_dtype = FP_DTYPE_F32; ln.dim = D; ln.in_rows_per_img = ntok; ln.ld_out = D;
    ln.sat = m->weight_dtype == FP_DTYPE_F16 ? ws->sat : nullptr;   // the "f16" mode's overflow report: non-finite features (common.hpp)
    ln.out = fmap; ln.out_rows = B * n_patches; ln.out_rows_per_img = n_patches; ln.in_skip = 1 + m->registers;
    TRY(layernorm_launch(ln, st));
    if (cls) {
      ln.out = cls; ln.out_rows = B; ln.out_rows_per_img = 1; ln.in_skip = 0;
      TRY(layernorm_launch(ln, st));
    }
  } else {
    HIP_TRY(hipMemcpy2DAsync(fmap, (size_t)n_patches * D * 4, ws->x + (size_t)(1 + m->registers) * D, (size_t)ntok * D * 4,
                             (size_t)n_patches * D * 4, B, hipMemcpyDeviceToDevice, st), "fp_vit_features: copy");
    if (cls)
      HIP_TRY(hipMemcpy2DAsync(cls, (size_t)D * 4, ws->x, (size_t)ntok * D * 4, (size_t)D * 4, B, hipMemcpyDeviceToDevice, st),
              "fp_vit_features: copy cls");
  }
  return FP_OK;
}

int fp_vit_sample_features(const fp_vit_model* m, const fp_vit_workspace* ws, int B, int grid_h, int grid_w, int apply_norm, int img_w, int img_h,
                           const float* points, const int32_t* point_img, int num_points, float* out, fp_stream_t stream) {
  FP_REQUIRE(m && ws && ws->x && points && out, "fp_vit_sample_features: null pointer");
  FP_REQUIRE(B >= 1 && grid_h >= 1 && grid_w >= 1 && img_w >= 1 && img_h >= 1, "fp_vit_sample_features: bad sizes");
  const int ntok = 1 + m->registers + grid_h * grid_w;
  return ln_sample_launch(ws->x, m->dim, m->norm_w, m->norm_b, 1e-6f, apply_norm, m->dim, ntok, 1 + m->registers, grid_h, grid_w, img_w, img_h,
                          points, point_img, num_points, out, ST(stream), nullptr, m->weight_dtype == FP_DTYPE_F16 ? ws->sat : nullptr);
}

int fp_query_select(const uint8_t* masks, int B, int H, int W, const int32_t* pix_x, const int32_t* pix_y, const float* grid_points, int num_points,
                    const int64_t* point_cells, int num_cells, int n_tok, int32_t* scratch, int32_t* counts, float* out_points, int32_t* out_point_img,
                    int32_t* out_q_off, int32_t* sel_rows, int32_t* sel_off, int32_t* row_map, fp_stream_t stream) {
  FP_REQUIRE(masks && pix_x && pix_y && grid_points && scratch && counts && out_points && out_point_img, "fp_query_select: null pointer");
  FP_REQUIRE(!point_cells || (sel_rows && sel_off && row_map), "fp_query_select: the token selection needs sel_rows, sel_off and row_map");
  return query_select_launch(masks, B, H, W, pix_x, pix_y, grid_points, num_points, reinterpret_cast<const long long*>(point_cells), num_cells, n_tok,
                             scratch, counts, out_points, out_point_img, out_q_off, sel_rows, sel_off, row_map, ST(stream));
}

int fp_vit_sample_features_selected(const fp_vit_model* m, const fp_vit_workspace* ws, int B, int grid_h, int grid_w, int apply_norm, int img_w,
                                    int img_h, const float* points, const int32_t* point_img, int num_points, const int32_t* row_map, float* out,
                                    fp_stream_t stream) {
  FP_REQUIRE(m && ws && ws->qkv && points && out && row_map, "fp_vit_sample_features_selected: null pointer");
  FP_REQUIRE(B >= 1 && grid_h >= 1 && grid_w >= 1 && img_w >= 1 && img_h >= 1, "fp_vit_sample_features_selected: bad sizes");
  const int ntok = 1 + m->registers + grid_h * grid_w;
  // fp_vit_block_selected left the selected tokens' rows of the residual stream, compact, where the qkv projections were
  return ln_sample_launch(reinterpret_cast<const float*>(ws->qkv), m->dim, m->norm_w, m->norm_b, 1e-6f, apply_norm, m->dim, ntok, 1 + m->registers,
                          grid_h, grid_w, img_w, img_h, points, point_img, num_points, out, ST(stream), row_map, m->weight_dtype == FP_DTYPE_F16 ? ws->sat : nullptr);
}

}  // extern "C"
