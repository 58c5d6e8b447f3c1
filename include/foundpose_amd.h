/* foundpose_amd -- C ABI of the MI355X (gfx950) FoundPose hot path.
 *
 * The reference (facebookresearch/foundpose) has NO native/FFI layer: its boundary for this path is the
 * Python call surface used by scripts/infer.py:468-542.  This header is the new native boundary that the
 * drop-in Python modules in foundpose_amd/ (same names and signatures as the reference's utils modules) bind with
 * ctypes.  Each entry point cites the reference computation it replaces (paths relative to the reference
 * root).  Conventions:
 *   - every pointer is a DEVICE pointer unless it says "host"; the caller owns all memory (no allocation
 *     inside the library), including scratch; all work is enqueued on `stream` (a hipStream_t), nothing
 *     synchronises; the library keeps no mutable global state besides the thread-local error string and idempotent
 *     per-device caches (which kernels already had their dynamic-LDS limit raised on a device, its CU count)
 *   - return value: 0 = ok, 1 = invalid argument, 2 = unsupported, 3 = HIP failure; fp_last_error() gives text
 *   - indices on the device are int32; distances/scores fp32; L2 distances are SQUARED (faiss convention)
 *   - canonical ordering of every top-k: best value first, ties broken by the lowest index
 */
#ifndef FOUNDPOSE_AMD_H
#define FOUNDPOSE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fp_stream_t; /* hipStream_t */

enum { FP_F32 = 0, FP_BF16 = 1, FP_FP8 = 2, FP_F16X3 = 3, FP_F16F8 = 4, FP_F16 = 5 }; /* element types of activation / weight buffers (FP_FP8: OCP e4m3
                                                               weights, fp_vit_model only; FP_F16X3: split-fp16 rows, FP_F16F8: f16f8 rows, see below;
                                                               FP_F16: plain IEEE fp16 -- the "f16" mode, see FP_GEMM_F16) */

/* ---- split-fp16 rows (the "f16x3" near-exact mode) --------------------------------------------------------------------
 * The reference computes the backbone in fp32 (scripts/infer.py:468-473).  The fp32-input MFMA runs at 1/16 of the fp16 /
 * bf16 rate on this part, so the near-exact mode carries every GEMM / attention operand as a PAIR of fp16 numbers and builds
 * each product from three fp16 MFMAs with fp32 accumulation: x ~ (hi + lo) / s, hi = f16(s x), lo = f16(s x - hi) (22 mantissa
 * bits; s a power of two that places typical magnitudes well inside the fp16 normal range, saturation at +-65504), and
 * a b = hi_a hi_b + hi_a lo_b + lo_a hi_b (the dropped lo lo term is <= 2^-22 relative).
 * Storage: a logical row of K values (K % 32 == 0) is 2K halves: group g = k / 32 holds hi(x[32g .. 32g + 31]) in halves
 * [64g, 64g + 32) and lo(...) in [64g + 32, 64g + 64).  Fixed scales of the activation rows inside fp_vit_forward: */
#define FP_SPLIT_SCALE_ACT 16.f /* LayerNorm outputs, attention outputs, normalised pixels of the patch rows: saturation at +-4094 */
#define FP_SPLIT_SCALE_QKV 16.f /* q, k, v rows written by the qkv GEMM: saturation at +-4094 */
#define FP_SPLIT_SCALE_HID 4.f  /* hidden activations written by the GELU / SwiGLU epilogue: saturation at +-16376 */
/* (A pair represents x to within max(2^-22 |x|, 2^-25 / s): the lo half of a small value is an fp16 subnormal, which the MFMA
 *  multiplies like any other number, so a modest scale costs nothing on O(1) activations and leaves three decades of head room for
 *  the outlier channels of real checkpoints.) */

/* ---- f16f8 rows (the "f16f8" mode: the split product with its two cross terms on the fp8 pipe) ---------------------------------
 * a b = hi_a hi_b + (hi_a lo_b + lo_a hi_b): the cross terms are ~2^-11 of the product, so they do not need fp16 operands.  An f16f8
 * row keeps the fp16 high halves and carries e4m3 copies of hi and lo for the cross terms, which run as two 64-wide
 * v_mfma_scale_f32_32x32x64_f8f6f4 (twice the fp16 rate): 8 instead of 12 fp16-MFMA units per 64 k at the same 4 bytes per element.
 * Storage: a logical row of K values (K % 64 == 0) is 4K bytes; group g = k / 64 holds  bytes [256g, 256g + 128): hi = f16(s x) (64 halves);
 * [256g + 128, 256g + 192): e4m3(hi 2^-7); [256g + 192, 256g + 256): e4m3((s x - hi) 2^4).  Same scales s and saturation report as the
 * split-fp16 rows.  A product carries ~14 mantissa bits at the worst (measured on fc2, K = 4096: max error 1.3e-5 of the output scale
 * against 1.9e-6 for f16x3); fp_vit_forward with weight_dtype FP_F16F8 uses these rows for every GEMM operand, while q | k | v and the
 * attention's own products stay split-fp16 (three fp16 MFMAs). */
#define FP_GEMM_SPLIT_F16F8 (1 << 20) /* OR-ed into fp_gemm_split's `epilogue`: A, W (and a GELU / SwiGLU output) are f16f8 rows, K % 64 == 0 */

/* ---- plain fp16 rows (the "f16" mode) ------------------------------------------------------------------------------------------
 * The bf16 pipeline of fp_vit_forward -- folded LayerNorms, (hi, lo) residual stream, the same kernels, tiles and bytes -- on IEEE fp16 operands
 * (v_mfma_f32_32x32x16_f16 runs at the bf16 rate): 11 significant bits per operand instead of 8, and a GELU polynomial good to 3.8e-5 (bf16 epilogue: 4e-4).
 * fp16 has bf16's speed but not its range: a 16-bit output beyond +-65504 becomes inf (nothing is clamped).  An inf poisons its row's residual stream and, through
 * the next attention's keys and values, every token of the image, so the pipeline's last kernel (fp_vit_features / fp_vit_sample_features*) counts non-finite
 * features into fp_vit_workspace.sat[0]; the Python extractor raises FoundPoseSaturationError for such a batch (apply_norm = 0: the caller checks the copy).
 * No operand scales: the residual stream and the activations of DINOv2 checkpoints sit orders of magnitude inside the range, values below 6e-5 keep an
 * absolute error <= 3e-8 (fp16 subnormals, which the MFMA honours). */
#define FP_GEMM_F16 (1 << 21) /* OR-ed into fp_gemm_bf16's / fp_gemm_bf16_ln's `epilogue`: A, W, the 16-bit outputs and the (xb, xl) stream are IEEE fp16 */

#define FP_ABI_VERSION 18
int fp_abi_version(void);
/* 1 if the library was compiled with -DFP_EXPERIMENTS (FP_EXPERIMENTS=1 python -m foundpose_amd.build): the measured-slower kernels kept for A/B runs -- the
 * role-split split-fp16 attention (fp_attention_split variant 2), the bf16 attention work splits 2 / 3 / 4, the two-stage k-NN of csrc/knn_cand.hip -- and
 * their environment switches (FP_KNN_CAND, FP_GEMM_RAST, FP_COSINE_MERGE_REPLAY) exist in such builds only.  The shipped library (0) reads NO environment
 * variable and keeps no mutable state beyond its idempotent per-device launch caches. */
int fp_build_experiments(void);
const char* fp_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Descriptor matching half
 * ---------------------------------------------------------------------------------------------- */

/* out[i] = |x_i|^2 as a k-ascending fp32 fma chain.  Part of faiss IndexFlatL2 (utils/knn_util.py:48-50). */
int fp_sqnorm_rows(const float* x, int64_t n, int d, float* out, fp_stream_t stream);

/* out = x / max(|x|, eps) per row: the per-operand normalisation of torch cosine_similarity
 * (utils/template_util.py:167); run once per bank on template_descs. */
int fp_normalize_rows(const float* x, int64_t n, int d, float eps, float* out, fp_stream_t stream);

/* Exact brute-force L2 k-NN: KNN.fit + KNN.search with metric "l2" (utils/knn_util.py:38-106).
 * q [m,d], db [n,d], precomputed squared norms of both.  Scratch: FP_KNN_SCRATCH_BYTES(m, n, k): k == 1: m*8 bytes; 2 <= k <= 8:
 * m * max(ceil(n/128) * k * 8, 704) bytes (candidate keys, no distance matrix); k > 8: m*n*4 bytes.  out_d2 [m,k] (squared),
 * out_idx [m,k] int32, ascending, ties -> lowest index, (inf, -1) past the database size.
 * FP_EXPERIMENTS builds only (fp_build_experiments() == 1, there with FP_KNN_CAND=1 in the environment; measured slower than the all-pairs tile on the benchmark shapes): 2 <= k <= 4
 * with d = 64 / 128 / 256, n >= 256 (the visual-word search: k = 3, d = 256) runs in two stages -- an fp16-MFMA candidate pass whose
 * error bound is derived from the operands' norms (csrc/knn_cand.hip), then the exact fp32 chain on the candidates -- with outputs
 * bit-identical to the all-pairs exact-fp32 tile that serves every other case; rows the bound cannot cover (values beyond the fp16
 * range, more near-ties than a candidate list holds) are computed by exact brute force inside the second stage. */
#define FP_KNN_SCRATCH_BYTES(m, n, k) \
  ((k) == 1 ? (size_t)(m) * 8 : ((k) <= 8 ? (size_t)(m) * ((size_t)(((n) + 127) / 128) * (k) * 8 > 704 ? (size_t)(((n) + 127) / 128) * (k) * 8 : 704) : (size_t)(m) * (n) * 4))
int fp_knn_l2(const float* q, const float* q_sqnorm, int m, const float* db, const float* db_sqnorm, int n,
              int d, int k, void* scratch, float* out_d2, int32_t* out_idx, fp_stream_t stream);

/* tf-idf descriptors for `num_segs` point sets (one detection's query patches each):
 * calc_tfidf (utils/template_util.py:31-71) + the query-side normalisation of cosine_similarity.
 * word_ids/word_d2 [sumQ, knn_k] from fp_knn_l2 against the visual words; seg_off [num_segs+1].
 * sqrt_dists=1 applies the sqrt of find_nearest_object_features (template_util.py:26-27) before the
 * soft-assignment weights (the bank side, template_util.py:112-119, passes 0).
 * desc [num_segs, num_words]; desc_n (may be null) = desc / max(|desc|, eps). */
int fp_tfidf_build(const int32_t* word_ids, const float* word_d2, int knn_k, const int32_t* seg_off, int num_segs,
                   const float* idf, int num_words, int soft_assign, float soft_sigma_squared, int sqrt_dists,
                   float* desc, float* desc_n, float eps, fp_stream_t stream);

/* Template retrieval: cosine similarity of each detection's descriptor against the template descriptors of
 * its object + top-n (tfidf_matching, utils/template_util.py:167-174).  Detections are grouped by object:
 * det_seg_off [num_obj+1] over rows of desc_n, obj_tpl_off [num_obj+1] over rows of bank_n (both
 * normalised), det_num_templates [num_det] = template count of each detection's object.  scratch_sims:
 * FP_COSINE_SCRATCH_FLOATS(num_det, max_templates) floats -- the finished scores [num_det, max_templates] (left there
 * for the caller) followed by the candidate keys of the fused top-n and one replay flag per detection.  Canonical fp32 summation order of a score:
 * num_words % 128 == 0: 8 contiguous k-slices, each an fma chain visiting every 16-block of k as
 * [0,4,8,12,1,5,...,15], slice sums added in slice order; num_words % 16 == 0: one such chain; else k ascending.
 * The order depends on num_words only -- never on the batch: more than 32 detections of one object are served in
 * 32-detection chunks by the same kernel.  out_scores/out_ids [num_det, n_top]; ids are object-local template ids,
 * -1 / -inf past the object's template count.  tie_mode: 0 = canonical (score, then lowest id); 1 = the tie order of
 * torch.topk on a CPU tensor (libstdc++ partial_sort / nth_element+sort replayed on the device; rows of any length,
 * n_top <= 32).  With n_top <= 7 the replay runs only for rows whose best n_top + 1 scores contain a tie (equal values,
 * +-0 or NaN): strictly decreasing scores make the top-n list unique, so the canonical list IS torch's. */
#define FP_COSINE_SCRATCH_FLOATS(num_det, max_templates) (2 * (size_t)(num_det) * (size_t)(max_templates) + 17 * (size_t)(num_det) + 2)
int fp_cosine_topk(const float* desc_n, const int32_t* det_seg_off, const int32_t* det_num_templates, int num_det,
                   int max_det_per_obj,
                   const float* bank_n, const int32_t* obj_tpl_off, int num_obj, int max_templates, int num_words,
                   int n_top, float* scratch_sims, float* out_scores, int32_t* out_ids, int tie_mode, fp_stream_t stream);

/* The same retrieval, same outputs bit for bit, in two stages (what match_batch calls): a first pass over an fp16 copy of the bank
 * (bank_n_f16 [T_total, num_words], round to nearest even; half the bytes, 1/16 of the matrix time) leaves approximate scores,
 * provably within eps = 2^-10 x 1.5625 of the exact ones for L2-normalised rows (bound derived in csrc/match.hip); every template within 2 eps of the (n_top + 1)-th best approximate score
 * -- a superset of the exact top n_top + 1, however large -- is then re-scored with the exact fp32 chain of fp_cosine_topk, and the
 * top n_top of those exact scores is the answer.  tie_mode 1: a row whose best n_top + 1 exact scores contain a tie needs its whole
 * row for the replay of torch.topk's order; such rows release fp_cosine_topk's single-pass kernel and the replay behind a device-side
 * flag (both exit at once otherwise).  The three dependent launches cost ~10 us of latency each, so the two-stage form is taken only
 * when the single pass would stream more than ~250 MB (max_templates x ceil(max_det_per_obj / 32) >= 30 000; tie_mode |
 * FP_COSINE_FORCE_PREFILTER forces it); otherwise, and when num_words is not a multiple of 1024 (<= 4096 with tie_mode 0, <= 2048 with
 * tie_mode 1: the strict order's exact fallback is the <= 2048-word single-pass kernel), for more than 65536
 * templates per object or n_top > 7, the call IS fp_cosine_topk.  scratch: FP_COSINE_PREFILTER_SCRATCH_FLOATS(num_det, max_templates) floats.
 * HARD PRECONDITIONS of the candidate bound (not checked; fp_cosine_topk has none of them and stays exact for any input): every row of
 * desc_n and bank_n has L2 norm <= 1 (what fp_normalize_rows writes), and bank_n_f16 is the round-to-nearest-even fp16 image of bank_n
 * element for element.  Unnormalised rows or any other fp16 copy void the superset guarantee -- true top-n templates can then be dropped
 * silently.  The Python side (foundpose_amd/bank.py) builds both from the same tensor. */
#define FP_COSINE_FORCE_PREFILTER 256
#define FP_COSINE_PREFILTER_SCRATCH_FLOATS(num_det, max_templates) (FP_COSINE_SCRATCH_FLOATS(num_det, max_templates) + 3 * (size_t)(num_det) * (size_t)(max_templates) + 32 * (size_t)(num_det) + 16)
int fp_cosine_topk_prefiltered(const float* desc_n, const int32_t* det_seg_off, const int32_t* det_num_templates, int num_det, int max_det_per_obj,
                               const float* bank_n, const void* bank_n_f16, const int32_t* obj_tpl_off, int num_obj, int max_templates, int num_words,
                               int n_top, float* scratch, float* out_scores, int32_t* out_ids, int tie_mode, fp_stream_t stream);

/* Cyclic best-buddy matching of every detection against its n_slots retrieved templates and assembly of the
 * 2D-3D correspondences (cyclic_buddies_matching + the gather in establish_correspondences,
 * utils/corresp_util.py:34-70,107-155).
 *   query_feats [sumQ,d], query_sqnorm [sumQ], query_points [sumQ,2], q_off [B+1]
 *   bank_feats [N_f,d] sorted by template, bank_sqnorm [N_f], tpl_off [T_total+1], vertices [N_f,3]
 *   tpl_ids [B*n_slots] template ids, <0 = empty slot: object-local (as fp_cosine_topk reports them) when tpl_base [B] = first
 *   template of each detection's object is given, GLOBAL ids when tpl_base is NULL
 *   feat_base [B]: first feature row of the detection's object (reported feature ids are object-local)
 *   scratch: FP_CYCLIC_SCRATCH_BYTES(B * n_slots, q_max, p_max) bytes (nearest-neighbour keys of both directions + the candidate
 *   lists of the two-stage search, or one slice of keys per 128 x 128 distance tile of the all-pairs form; nothing has to be preset)
 *   FP_EXPERIMENTS builds with FP_KNN_CAND=1 (see fp_knn_l2) and d = 64 / 128 / 256: the two 1-NN searches run as fp16-MFMA candidate pass + exact re-scoring
 *   (csrc/knn_cand.hip, same keys bit for bit); otherwise the all-pairs exact-fp32 tile
 * outputs, padded to k_max >= top_k per (detection, slot): count, query ids, object feature ids (= the
 * reference's nn_vertex_ids), cycle distances, confidences, coord_2d, coord_3d.  tie_mode as in fp_cosine_topk: 1 makes
 * the order (and the choice among tied distances at the top_k boundary) identical to the reference's
 * torch.topk(-cycle_dists, k). */
#define FP_CYCLIC_SCRATCH_TILES(pairs, q_max, p_max) \
  (8 * (size_t)(pairs) * ((size_t)(((p_max) + 127) / 128) * (size_t)(q_max) + (size_t)(((q_max) + 127) / 128) * (size_t)(p_max)))
#define FP_CYCLIC_SCRATCH_CAND(pairs, q_max, p_max) \
  (8 * (size_t)(pairs) * ((size_t)(q_max) + (size_t)(p_max)) + 448 * (size_t)(pairs) * (size_t)((q_max) > (p_max) ? (q_max) : (p_max)))
#define FP_CYCLIC_SCRATCH_BYTES(pairs, q_max, p_max) \
  (FP_CYCLIC_SCRATCH_TILES(pairs, q_max, p_max) > FP_CYCLIC_SCRATCH_CAND(pairs, q_max, p_max) ? FP_CYCLIC_SCRATCH_TILES(pairs, q_max, p_max) \
                                                                                             : FP_CYCLIC_SCRATCH_CAND(pairs, q_max, p_max))
int fp_cyclic_buddies(const float* query_feats, const float* query_sqnorm, const float* query_points,
                      const int32_t* q_off, int num_det, int q_max, const float* bank_feats,
                      const float* bank_sqnorm, const int32_t* tpl_off, int p_max, const float* vertices,
                      const int32_t* tpl_ids, const int32_t* tpl_base, const int32_t* feat_base, int n_slots, int d, int top_k, int k_max,
                      void* scratch, int32_t* out_count, int32_t* out_q_ids, int32_t* out_feat_ids,
                      float* out_dists, float* out_conf, float* out_coord_2d, float* out_coord_3d, int tie_mode,
                      fp_stream_t stream);

/* The fixed-size record of each detection for the one exchange step of a multi-GPU run (an RCCL all-gather of these rows):
 * out [num_det, n_slots * (3 + 9 * k_max)] 32-bit words typed fp32 -- per slot (template id, score, count), then per correspondence
 * (query id, feature id = nn_vertex_ids, distance, confidence, x, y, X, Y, Z).  Integer fields keep their bit patterns (ids above
 * 2^24 survive). */
int fp_pack_records(const int32_t* template_ids, const float* template_scores, const int32_t* counts, const int32_t* q_ids, const int32_t* feat_ids,
                    const float* dists, const float* conf, const float* coord_2d, const float* coord_3d, int num_det, int n_slots, int k_max, float* out,
                    fp_stream_t stream);

/* Coarse pose of every (detection, template slot) pair from its 2D-3D correspondences: estimate_pose
 * (utils/pnp_util.py:20-84 = cv2.solvePnPRansac(..., SOLVEPNP_ITERATIVE) + cv2.solvePnPRefineLM on the inliers), the call
 * scripts/infer.py:552-580 makes once per retrieved template.  Inputs are fp_cyclic_buddies' padded outputs: coord_2d
 * [num_pairs, k_max, 2], coord_3d [num_pairs, k_max, 3], counts [num_pairs]; cameras [num_pairs / n_slots, 4] f64 =
 * (fx, fy, cx, cy) of each detection's crop camera.  RANSAC over `ransac_iters` P3P hypotheses (3 points + 1 to choose
 * among the solutions), inlier = reprojection error <= inlier_thresh px, the first model with the most inliers inside the
 * adaptively shortened budget (confidence) wins, then <= lm_iters Levenberg-Marquardt iterations on its inliers (20 for
 * the refinement inside solvePnPRansac, +20 with pnp_refine_lm).  Pairs with fewer than min_corresp (6, infer.py:556)
 * correspondences or without a model fail (success 0).  Outputs: success, R [.,9] row-major and t [.,3] (model ->
 * camera, f64), the RANSAC inlier count (the reference's `quality`), the inlier mask [., k_max], and (may be null) the
 * winning model before refinement [., 12].  cv2's own arithmetic and random stream are not reproduced (cv2 is not
 * available to pin them): same scheme, different minimal solver and sampler -- see csrc/pnp.hip. */
int fp_pnp_ransac(const float* coord_2d, const float* coord_3d, const int32_t* counts, const double* cameras, int num_pairs, int n_slots,
                  int k_max, int ransac_iters, double inlier_thresh, double confidence, int lm_iters, int min_corresp, uint64_t seed,
                  int32_t* out_success, double* out_R, double* out_t, int32_t* out_num_inliers, uint8_t* out_inlier_mask,
                  double* out_ransac_pose, fp_stream_t stream);

/* sample_feature_map_at_points (utils/feature_util.py:100-131): bilinear grid_sample, zeros padding,
 * align_corners=False.  fmap addressed by element strides (image, channel, y, x); point_img (may be null)
 * maps each point to its image.  out [num_points, C]. */
int fp_sample_bilinear(const float* fmap, int64_t stride_img, int64_t stride_c, int64_t stride_h, int64_t stride_w,
                       int C, int H, int W, int img_w, int img_h, const float* points, const int32_t* point_img,
                       int num_points, float* out, fp_stream_t stream);

/* PCAProjector.transform (utils/projector_util.py:66-69): out = x @ C^T - mean_proj, mean_proj = mu @ C^T
 * (computed with the same kernel by passing x = mu, n = 1, mean_proj = null). */
int fp_pca_project(const float* x, int n, int D, const float* components, int d, const float* mean_proj,
                   float* out, fp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Feature extraction half (DINOv2 ViT; the reference reaches it through
 * DinoFeatureExtractor.forward, utils/dinov2_utils.py:115-158)
 * ---------------------------------------------------------------------------------------------- */

typedef struct {
  const float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *ls1, *ls2; /* fp32 [D] */
  const void *qkv_w, *proj_w, *fc1_w, *fc2_w;             /* [N,K] row-major (torch Linear), bf16 or f32 */
  const float *qkv_b, *proj_b, *fc1_b, *fc2_b;            /* fp32 */
  /* weight_dtype == FP_FP8 only: the four matrices hold OCP e4m3 bytes quantised per output channel, *_s [N] are the
   * dequantisation scales of the output columns, 1 / (act_scale x weight scale of the channel) -- proj_s and fc2_s
   * already multiplied by ls1 / ls2 --, the four biases are divided by (1 / (act_scale x weight scale)), and
   * act_scale[0..3] quantise the inputs of qkv, proj, fc1, fc2 (static per-tensor scales from a calibration batch) */
  const float *qkv_s, *proj_s, *fc1_s, *fc2_s;
  float act_scale[4];
  /* weight_dtype == FP_F16X3 (FP_F16F8: the same with f16f8 rows): the four matrices are split-fp16 rows ([N, 2K] halves) of s_w W with a power-of-two s_w per
   * matrix, and act_scale[0..3] = 1 / (scale of the GEMM's input rows x s_w) for qkv, proj, fc1, fc2: the epilogue computes
   * acc * act_scale + bias.  ls1 / ls2 are applied as usual. */
  /* weight_dtype == FP_F16: the four (folded) matrices may be stored times a power of two s_w each -- fp16 keeps its 11 bits only down to 6e-5, which a
   * LayerScale-folded matrix of small gammas would undershoot -- with act_scale[0..3] = 1 / s_w for qkv, proj, fc1, fc2 (0 = unscaled); colsum is then the
   * row sum of the STORED matrix. */
  /* fp_vit_model.ln_fold only: fp32 [N] sums of the rows of the (gain-folded, bf16-rounded) qkv_w / fc1_w */
  const float *qkv_colsum, *fc1_colsum;
} fp_vit_block;

typedef struct {
  int dim, depth, heads, hidden, registers, patch;
  int ffn_swiglu;          /* 1 for ViT-g: fc1_w/fc1_b hold mlp.w12 with rows INTERLEAVED (x1_j, x2_j) [2*hidden, D],
                              fc2_w/fc2_b hold mlp.w3 [D, hidden]; h = silu(x1) * x2 is fused into the first GEMM */
  int weight_dtype;        /* FP_BF16 | FP_F32 | FP_F16: dtype of the matrices and of the activation buffers (FP_F16 needs ln_fold = 1); FP_FP8: e4m3 block
                              matrices (see fp_vit_block), bf16 activation buffers and patch-embed weight */
  const void* patch_w;     /* [D, patch_k_pad]: conv weight flattened (c,py,px), zero padded */
  int patch_k_pad;         /* multiple of 64 */
  const float* patch_b;    /* [D] */
  const float* pos_patch;  /* [Np, D] pos-embed rows of the patch tokens for the current grid */
  const float* prefix;     /* [1+registers, D]: cls_token + pos[0], then the register tokens */
  const float *norm_w, *norm_b;
  const fp_vit_block* blocks; /* HOST array of `depth` entries */
  int ld_w_dim, ld_w_hidden;  /* row strides (elements) of the block matrices with K = dim (qkv, proj, fc1) and with
                                 K = hidden (fc2); 0 = dense (dim / hidden).  A stride that is not a multiple of 2 KiB
                                 keeps the 8 rows of a staging instruction off one L2 channel */
  int patch_stride;        /* conv stride of the patch embedding; 0 = patch (the shipped configs).  Smaller strides (the reference's
                              patch_vit_resolution, utils/dinov2_utils.py:364-389) give overlapping patches: 1 + (size - patch) / stride
                              tokens per axis, pos_patch then holds the reference's strided position encoding; full forward only */
  float patch_acc_scale;   /* FP_F16X3 only: 1 / (FP_SPLIT_SCALE_ACT x scale of the split patch_w [D, 2 * patch_k_pad]) */
  int ln_fold;             /* FP_BF16 / FP_F16 only.  1: the two LayerNorms of a block are folded into the GEMMs around them -- no
                              LayerNorm kernel runs inside the blocks.  qkv_w / fc1_w then hold W * diag(ln weight) (bf16),
                              qkv_b / fc1_b hold b + W ln_bias, *_colsum the row sums of those matrices; proj_w / fc2_w hold
                              diag(LayerScale) W and proj_b / fc2_b hold LayerScale * b (ls1 / ls2 are then unused); the
                              residual GEMMs (proj, fc2) also emit bf16(x) and per-row (sum x, sum x^2), and the qkv / fc1
                              epilogues compute rstd * (acc - mean * colsum) + bias.  Needs workspace xb / stats. */
  int flags;               /* tuning bits of a whole forward (results are bit-identical either way): FP_VIT_NO_TALL_TILES */
} fp_vit_model;
#define FP_VIT_NO_TALL_TILES 1 /* the wide bf16 / f16 GEMMs never take their 320-row block tile (A/B switch; the tiles walk k in the same order) */

typedef struct {
  void* patches; /* [m_patch_pad, patch_k_pad] activation dtype */
  float* x;      /* [m_pad, D] fp32 residual stream */
  void* y;       /* [m_pad, D] activation dtype (LN output / attention output) */
  void* qkv;     /* [m_pad, 3D] */
  void* h;       /* [m_pad, hidden] */
  void* a8;      /* FP_FP8 only: [m_pad, max(D, hidden)] bytes, the quantised input of the GEMM about to run; m_pad must
                    then be a multiple of 256 */
  int ld_y, ld_h, ld_qkv; /* row strides (elements) of y, h and qkv; 0 = dense (D / hidden / 3D) */
  int m_pad;     /* rows allocated, multiple of 128, >= B*(1+R+Np); a multiple of 1280 (whole 256- AND 320-row tiles) lets the wide bf16 / fp8
                    GEMMs of a large batch take their taller tile (DinoFeatureExtractor.padded_rows) */
  int m_patch_pad; /* multiple of 128, >= B*Np */
  void* xb;      /* ln_fold only: [m_pad, D] bf16 copy of the residual stream (row stride ld_y), the A operand of qkv / fc1 */
  float* stats;  /* ln_fold only: [D / 128 + 1, m_pad, 2] fp32: partial row sums (sum x, sum x^2) per 128-column group of the
                    producer, then one slot of (rstd, mean * rstd) per row */
  int32_t* sat;  /* may be NULL; else [2] STICKY saturation counters, only ever incremented (the caller zeroes and reads them):
                    [0] FP_F16X3: a producer of split-fp16 rows (LayerNorm, the qkv / GELU / SwiGLU epilogues) clamped |s x| > 65504,
                        i.e. an activation beyond +-4094 (LayerNorm outputs, q, k, v) or +-16376 (hidden) -- the near-exact mode's
                        features are then NOT the fp32 arithmetic's; the Python extractor raises FoundPoseSaturationError on it;
                        FP_F16: the last kernel of the pipeline (final norm / sampling) produced a non-finite feature -- an fp16 activation
                        beyond +-65504 somewhere in the backbone (see "plain fp16 rows");
                    [1] FP_FP8: a quantising producer (LayerNorm, attention output, GELU / SwiGLU epilogue) clamped |s x| > 448
                        (an input beyond its static calibration scale).
                    Counted per reporting thread, not per element: non-zero means "at least one live output row clamped".
                    The attention output and the softmax probabilities of the f16x3 mode cannot clamp (a convex combination of v rows
                    that fit their scale; p <= 2) and do not report; padding rows never report. */
  void* xl;      /* ln_fold only, may be NULL: [m_pad, D] bf16 (row stride ld_y), the LOW halves of the residual stream.  When given, the blocks in
                    front of the hooked one keep the stream as the pair (xb, xl) -- x = hi + lo, hi' = bf16(x'), lo' = bf16(x' - hi'): 16 mantissa
                    bits per update -- and their residual GEMMs read 4 + write 4 bytes per element instead of 4 + 6 (no fp32 read-modify-write
                    beside a separate bf16 copy); the hooked block runs on an fp32 stream rebuilt from the pair.  NULL: fp32 stream throughout. */
} fp_vit_workspace;

/* images [B,3,H,W] fp32 in [0,1] -> ws->x holds the output of blocks[layer] for every token
 * (what the reference's forward hook captures, dinov2_utils.py:160-211), blocks after `layer` are not run
 * (layer = -1: the token embedding only -- used by the fp8 calibration pass). */
int fp_vit_forward(const fp_vit_model* model, const fp_vit_workspace* ws, const float* images, int B, int H, int W,
                   int layer, fp_stream_t stream);

/* Final LayerNorm on CLS + patch tokens with the register tokens dropped (dinov2_utils.py:138-142,304):
 * fmap [B, Np, D] fp32 token-major (the reference's [B,D,Hp,Wp] is a permuted view of it), cls [B, D].
 * apply_norm = 0 copies the raw tokens. */
int fp_vit_features(const fp_vit_model* model, const fp_vit_workspace* ws, int B, int n_patches, int apply_norm,
                    float* fmap, float* cls, fp_stream_t stream);

/* Final LayerNorm + sample_feature_map_at_points in one pass (SURVEY 8b `fp_ln_gather_pca`, its LayerNorm + gather half;
 * fp_pca_project finishes): out[p, :] = bilinear sample (feature_util.py:100-131) at points[p] (image coordinates, image
 * img_w x img_h, detection point_img[p]) of LayerNorm(tokens) (dinov2_utils.py:138-142; apply_norm = 0: raw tokens),
 * computed from the residual stream fp_vit_forward left in ws->x -- the [B, Np, D] feature map is never written.
 * Bit-identical to fp_vit_features followed by fp_sample_bilinear.  out [num_points, D] fp32. */
int fp_vit_sample_features(const fp_vit_model* model, const fp_vit_workspace* ws, int B, int grid_h, int grid_w, int apply_norm, int img_w,
                           int img_h, const float* points, const int32_t* point_img, int num_points, float* out, fp_stream_t stream);

/* Query-token selection in the hooked block (bf16 model with ln_fold, fp8 model, or f16x3 model).  The reference runs the backbone on every token and
 * then reads the feature map at the query points only (utils/dinov2_utils.py:257,304 -> utils/feature_util.py:100-131 at the
 * points of scripts/infer.py:452-466): the hooked block's OUTPUT is needed for the patch tokens under the sampling taps and
 * for no other token, while its keys and values still come from all tokens.  Three calls replace fp_vit_forward +
 * fp_vit_sample_features with identical sampled features (bit for bit: a token's row never depends on which rows share its
 * GEMM tile or attention block):
 *   fp_vit_forward_prefix   embedding + blocks 0..layer-1 (what block `layer` starts from stays in the workspace).  The workspace
 *                           state between the calls is PRIVATE to them: with ws->xl set (the (hi, lo) residual stream) and layer > 0 the
 *                           stream lives in the (ws->xb, ws->xl) pair only and ws->x is UNDEFINED (it still holds the token
 *                           embedding); do not read ws->x after a prefix run -- fp_vit_block_selected rebuilds the rows it needs;
 *   fp_vit_block_selected   block `layer`: LayerNorm constants and the qkv projection for all tokens, then attention
 *                           queries, proj, fc1 and fc2 for the selected tokens only.  sel_rows [num_sel] = global token
 *                           rows (b * n_tok + token), ascending, grouped by image; sel_off [B + 1] = offsets of the images
 *                           in sel_rows; max_sel_per_img >= the largest per-image count (host value: it sizes the grid).
 *                           The selected rows of the residual stream are left compact ([num_sel, D] fp32) in ws->qkv;
 *   fp_vit_sample_features_selected   as fp_vit_sample_features, reading those rows through row_map [B * grid_h * grid_w]:
 *                           patch cell -> its row in the compact buffer, < 0 if the cell was not selected.  Every tap of
 *                           every point must be selected (a missed tap returns NaN features, never a silently wrong row). */
/* Query points of a batch and, optionally, the token selection above -- on the device, two small launches, no host
 * round trip (generate_grid_points + filter_points_by_mask, utils/feature_util.py:19-41, called per detection at
 * scripts/infer.py:359,478).  masks [B, H, W] u8; grid point g = grid_points[g] (x, y) with pixel (pix_x[g], pix_y[g]) =
 * int(point + 0.5); it is a query point of image b iff the pixel lies strictly inside the canvas and on the mask.
 * -> counts [B] points per image (+ [B, 2B) selected tokens per image when point_cells is given: copy them to the host);
 *    out_points [B * num_points, 2] / out_point_img [B * num_points]: the first sum(counts) rows are the query points,
 *    grouped by image, grid order inside (the order of the reference's boolean indexing); out_q_off [B + 1] (may be null).
 * point_cells (may be null) [num_points, 9] i64 = the patch cells the sampling of grid point g may read (the 3 x 3 cells
 * around the cell its sampling position rounds to; num_cells for "outside the map") -> sel_rows [B * num_cells] (first
 * sum(counts[B:]) entries valid), sel_off [B + 1], row_map [B * num_cells] as fp_vit_block_selected /
 * fp_vit_sample_features_selected take them.  scratch: B * (num_points + num_cells) i32. */
int fp_query_select(const uint8_t* masks, int B, int H, int W, const int32_t* pix_x, const int32_t* pix_y, const float* grid_points, int num_points,
                    const int64_t* point_cells, int num_cells, int n_tok, int32_t* scratch, int32_t* counts, float* out_points, int32_t* out_point_img,
                    int32_t* out_q_off, int32_t* sel_rows, int32_t* sel_off, int32_t* row_map, fp_stream_t stream);
int fp_vit_forward_prefix(const fp_vit_model* model, const fp_vit_workspace* ws, const float* images, int B, int H, int W, int layer,
                          fp_stream_t stream);
int fp_vit_block_selected(const fp_vit_model* model, const fp_vit_workspace* ws, int B, int H, int W, int layer, const int32_t* sel_rows,
                          const int32_t* sel_off, int num_sel, int max_sel_per_img, fp_stream_t stream);
/* Precision schedules: blocks 0..k-1 in one model (e.g. FP_F16), blocks k..layer in another (FP_F16X3 / FP_F16F8 / FP_F32) over the same fp32 stream.
 *   fp_vit_stream_f32      the residual stream fp_vit_forward_prefix(model, ws, .., layer = k) -- or, for a model without folded LayerNorms, fp_vit_forward(.., layer =
 *                          k - 1) -- left in its workspace (the (xb, xl) pair of a folded-LayerNorm model, else ws->x), as fp32 rows out [B * n_tok, D]: the `x` buffer
 *                          of the second model's workspace;
 *   fp_vit_forward_blocks  blocks first_block..layer (prefix_only: ..layer-1) of `model` on the stream already in ws->x: no embedding; 1 <= first_block <= layer.
 *                          fp_vit_block_selected / fp_vit_features / fp_vit_sample_features* follow as after fp_vit_forward_prefix / fp_vit_forward. */
int fp_vit_stream_f32(const fp_vit_model* model, const fp_vit_workspace* ws, int B, int H, int W, int layer, float* out, fp_stream_t stream);
int fp_vit_forward_blocks(const fp_vit_model* model, const fp_vit_workspace* ws, int B, int H, int W, int first_block, int layer, int prefix_only,
                          fp_stream_t stream);
int fp_vit_sample_features_selected(const fp_vit_model* model, const fp_vit_workspace* ws, int B, int grid_h, int grid_w, int apply_norm,
                                    int img_w, int img_h, const float* points, const int32_t* point_img, int num_points,
                                    const int32_t* row_map, float* out, fp_stream_t stream);

/* Building blocks, exported for unit tests and for callers that schedule the layers themselves. */
int fp_patchify(const float* images, int B, int H, int W, int patch, void* out, int ld_out, int out_dtype,
                fp_stream_t stream);
int fp_layernorm(const float* x, int ld_x, const float* weight, const float* bias, float eps, void* out, int ld_out,
                 int out_dtype, int dim, int out_rows, int out_rows_per_img, int in_rows_per_img, int in_skip,
                 fp_stream_t stream);
/* epilogue: 0 bias->bf16, 1 bias+gelu->bf16, 3 LayerScale*(.)+residual (fp32 in place), 5 bias->f32,
 * 6 SwiGLU (interleaved column pairs -> [M, N/2] bf16);
 * tuning bits: epilogue | (128 << 8), | (256 << 8) or | (320 << 8) forces that block tile (default: chosen from the shape; 320 = the 320 x 256
 * tile of the bias / GELU epilogues, M a multiple of 320; every tile gives the same bits).  Row tiles without live rows (>= M_valid) are not launched. */
int fp_gemm_bf16(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int M_valid, const float* bias,
                 const float* gamma, void* out, int ldo, int epilogue, fp_stream_t stream);
/* The GEMMs of a block with the LayerNorm folded in (fp_vit_model.ln_fold), exported for unit tests and benchmarks.
 * epilogue 8 (producer on the (hi, lo) stream, fp_vit_workspace.xl): x = xb + out (both bf16 [M, ld_xb]: high and LOW halves), x' = x + acc + bias,
 *   xb = bf16(x'), out = bf16(x' - xb), stats as for epilogue 7.
 * epilogue 7 (producer, proj / fc2 with LayerScale folded into W and bias): out(f32) += acc + bias; if xb != NULL also
 *   xb[M, ld_xb] = bf16(out) and stats[(col / 128) * M + row] = (sum, sum of squares) of the row over that 128-column
 *   group (float2 per slot, N / 128 slots of M rows).
 * epilogues 0 / 1 / 6 (consumer, qkv / fc1 with the LayerNorm gain folded into W, the shift into bias):
 *   out = epi(ln_row[r].x * acc - ln_row[r].y * colsum[n] + bias[n]), ln_row [M, 2] = (rstd, mean * rstd) per row from
 *   fp_ln_finalize, colsum [N] = row sums of W.  Tuning bits as in fp_gemm_bf16. */
int fp_gemm_bf16_ln(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int M_valid, const float* bias, void* out, int ldo,
                    int epilogue, const float* colsum, const float* ln_row, void* xb, int ld_xb, float* stats, fp_stream_t stream);
/* stats [parts, stats_stride, 2] partial row sums over `dim` columns in total -> ln_row [rows, 2] = (rstd, mean * rstd). */
int fp_ln_finalize(const float* stats, int parts, int stats_stride, int rows, int dim, float eps, float* ln_row, fp_stream_t stream);

/* fp8 GEMM (BASELINE config 5, "ViT-g/14 fp8"): A [M, K] and W [N, K] hold OCP e4m3 bytes (fp_quantize_fp8), products
 * and accumulation in fp32 on the block-scaled MFMA with unit scales (v_mfma_scale_f32_32x32x64_f8f6f4, twice the bf16
 * rate).  out = epilogue((acc + bias[n]) * col_scale[n]): col_scale = 1 / (activation scale x weight scale of channel
 * n) (x LayerScale gamma for epilogue 3), bias pre-divided by col_scale.  Epilogues 0, 1, 3, 6 as fp_gemm_bf16;
 * M, N multiples of 256, K a multiple of 128.
 * out_scale > 0 (epilogues 1 and 6 only): the result feeds the next fp8 GEMM and is written as e4m3(value * out_scale)
 * bytes, ldo in bytes; out_scale = 0: bf16 / fp32 output as fp_gemm_bf16. */
int fp_gemm_fp8(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int M_valid, const float* bias,
                const float* col_scale, void* out, int ldo, int epilogue, float out_scale, fp_stream_t stream);
/* out[i] = e4m3(clamp(in[i] * scale, +-448)), round to nearest even; in fp32 or bf16 (in_dtype FP_F32 / FP_BF16) */
/* Split-fp16 GEMM (f16x3): A [M, 2K] and W [N, 2K] halves (split rows, scales s_a and s_w), K the LOGICAL depth (multiple of 32),
 * lda / ldw in halves.  v = acc * acc_scale + bias with acc_scale = 1 / (s_a s_w), then the epilogue numbered as for
 * fp_gemm_bf16: 0 (bias), 1 (GELU, the exact erf form here) and 6 (SwiGLU) write a split row again ([M, 2N] -- SwiGLU: [M, N] --
 * halves, values x out_scale, ldo in halves); 3 (out += gamma v) and 5 (bias) write fp32. */
int fp_gemm_split(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int M_valid, const float* bias, const float* gamma,
                  void* out, int ldo, int epilogue, float acc_scale, float out_scale, fp_stream_t stream);
/* Attention on split rows: qkv [B*N, 6D] halves (q | k | v, each 2D, split-fp16 rows of scale in_scale) -> out [B*N, 2D] halves (scale out_scale);
 * out_dtype FP_F16X3: a split-fp16 row, FP_F16F8: an f16f8 row (the f16f8 mode's proj operand); optionally OR-ed with FP_ATTN_VARIANT(v), test bits as in
 * fp_attention: 0 = the kernel the pipeline runs (the lock-step kernel), 1 = the lock-step kernel,
 * 2 = the role-split kernel (the two waves of a SIMD half a key tile apart; measured not faster, profiles/EXPERIMENTS.md section 0b: FP_EXPERIMENTS builds only, the shipped
 * library returns FP_ERR_UNSUPPORTED; in_scale >= 1).  All bit-identical. */
int fp_attention_split(const void* qkv, int ld_qkv, void* out, int ld_out, int B, int n_tok, int dim, int heads, float in_scale, float out_scale,
                       int out_dtype, fp_stream_t stream);
/* LayerNorm whose output carries a scale: out_dtype FP_FP8 (e4m3(y * out_scale) bytes), FP_F16X3 (split row of y * out_scale) or FP_F16F8 (f16f8 row) */
int fp_layernorm_scaled(const float* x, int ld_x, const float* weight, const float* bias, float eps, void* out, int ld_out, int out_dtype, float out_scale,
                        int dim, int out_rows, fp_stream_t stream);
int fp_quantize_fp8(const void* in, int in_dtype, int64_t n, float scale, void* out, fp_stream_t stream);
/* exact-fp32 MFMA GEMM; epilogue: 0 store, 4 bias, 5 bias+gelu, 6 LayerScale residual, 8 SwiGLU (as above) */
int fp_gemm_f32(const float* A, int lda, const float* W, int ldw, int M, int N, int K, const float* bias,
                const float* gamma, float* out, int ldo, int epilogue, fp_stream_t stream);
/* qkv [B*N, 3D] (q | k | v column blocks, head-major inside) -> out [B*N, D].
 * dtype: FP_F32 / FP_BF16 / FP_F16 (IEEE fp16 q | k | v and output: the "f16" mode's kernel, variant 0 only), optionally OR-ed with FP_ATTN_VARIANT(v) to pick a
 * bf16 work split (all bit-identical): 0 = 64 queries per wave, K/V by LDS-DMA (default), 1 = 32 queries per wave with register staging (the cross-check);
 * FP_EXPERIMENTS builds only (measured slower): 2 = the DMA kernel with 8 waves per 256-query block, 3 = 8 waves x 64 queries (512-query blocks), 4 = K prefetch.
 * FP_F32: 0 = flash attention on the fp32 MFMA (default), 1 = one thread per query with one fma chain per score (its cross-check;
 * the two agree to fp32 rounding, not bit for bit). */
#define FP_ATTN_VARIANT(v) ((v) << 8)
int fp_attention(const void* qkv, int ld_qkv, void* out, int ld_out, int B, int n_tok,
                 int dim, int heads, int dtype, fp_stream_t stream);
int fp_convert_f32_to_bf16(const float* in, void* out, int64_t n, fp_stream_t stream);

/* ---- crop producer (the step right before the path) ---------------------------------------------------
 * Batched misc.warp_image (utils/misc.py:458-519) as scripts/infer.py:433-450 calls it: every destination pixel of a
 * crop camera is mapped through the source camera in fp64 (window_to_eye -> eye_to_world -> world_to_eye ->
 * eye_to_window, points behind the source camera -> -1 when depth_check), cast to fp32 and resampled with cv2.remap
 * semantics (constant border 0).
 *   mode FP_WARP_LINEAR : src fp32 [n_src, src_h, src_w, channels] (HWC, [0,1]) -> out fp32 [batch, channels, out_h,
 *                         out_w] (CHW: array_to_tensor(...).permute(2,0,1), infer.py:466-468); INTER_LINEAR, which is
 *                         also what cv2.remap does for INTER_AREA
 *   mode FP_WARP_NEAREST: src u8 [n_src, src_h, src_w] -> out u8 [batch, out_h, out_w]   (the modal mask)
 * params [batch, 32] doubles per crop: crop camera f[2], c[2], R[9] (row-major rotation of T_world_from_eye), t[3],
 * then the same 16 for the source camera.  src_index [batch] picks the source image of each crop (null: crop b reads
 * image b).  map_out (may be null) receives the fp32 maps [batch, 2, out_h, out_w]. */
enum { FP_WARP_LINEAR = 0, FP_WARP_NEAREST = 1 };
int fp_warp_crops(const void* src, int n_src, int src_h, int src_w, int channels, int mode, const int32_t* src_index,
                  const double* params, int batch, int out_h, int out_w, int depth_check, void* out, float* map_out,
                  fp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FOUNDPOSE_AMD_H */
