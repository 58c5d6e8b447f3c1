// Internal launch interfaces between the C ABI (api.cpp) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---------------------------------------------------------------- f32_tile.hip
enum : int {
  F32_EPI_STORE = 0,        // out = acc
  F32_EPI_DIST_STORE = 1,   // out = max(0, fma(-2, acc, |a|^2 + |b|^2))
  F32_EPI_DIST_ARGMIN = 2,  // atomicMin of (d2,idx) keys per row and per column
  F32_EPI_SUB_VEC = 3,      // out = acc - vec[j]          (PCA: X C^T - mu C^T)
  F32_EPI_BIAS = 4,         // out = acc + bias[j]
  F32_EPI_BIAS_GELU = 5,    // out = gelu_erf(acc + bias[j])
  F32_EPI_LS_RESID = 6,     // out += gamma[j] * (acc + bias[j])
  F32_EPI_TOKENS = 7,       // patch-embed scatter into the token sequence (+ pos-embed)
  F32_EPI_SWIGLU = 8,       // columns interleaved (x1_j, x2_j): out[:, j] = silu(acc_2j + b_2j) * (acc_2j+1 + b_2j+1)
  F32_EPI_DIST_TOPK = 9,    // per row the k smallest (d2, column) keys of this tile -> row_best[row, n_tile, k] (k = row_stride <= 8)
};

struct F32TileArgs {
  const float* A; int lda;
  const float* B; int ldb;
  int K, M, N;
  // ragged grouping (device tables, may be null): pair -> segment of A rows / B rows
  const int* a_seg_off; const int* pair_a_seg; int pair_a_div;  // pair_a_div > 0: segment = pair / pair_a_div
  const int* b_seg_off; const int* pair_b_seg;
  const int* pair_b_base;  // may be null; else pair_b_seg values are relative: + pair_b_base[pair / pair_a_div]
  // epilogue operands
  float* out; int ldo; long long out_pair_stride; int out_row_global;  // out_row_global: row index = a_off + i
  const float* a_sqnorm; const float* b_sqnorm;
  unsigned long long* row_best; int row_stride;
  unsigned long long* col_best; int col_stride;
  // DIST_ARGMIN: 0 = atomicMin into row_best[pair * row_stride + row] / col_best[pair * col_stride + col] (tables preset to ~0);
  // 1 = every tile stores its own slice, no atomics, no preset: row_best[(pair * grid.x + tile_n) * row_stride + row] and
  // col_best[(pair * grid.y + tile_m) * col_stride + col]; the reader takes the min over the live tiles
  int best_parts;
  const float* bias; const float* gamma;
  const float* pos; int tok_np, tok_n, tok_skip;
};

int f32_tile_launch(int epi, const F32TileArgs& a, int max_m, int max_n, int pairs, hipStream_t st);

// ---------------------------------------------------------------- knn_cand.hip
// Exact L2 k-NN (k <= 4, dimension 64 / 128 / 256) in two stages: fp16-MFMA candidate pass with a derived bound + exact fp32 chains
// on the candidates -- the outputs of the all-pairs exact tile (DIST_TOPK / DIST_ARGMIN) bit for bit.  Segment tables as in F32TileArgs.
struct KnnCandArgs {
  const float* A; const float* B; int ld;   // [*, K] fp32, one row stride for both sides
  const float* a_sqn; const float* b_sqn;   // squared norms (k-ascending fma chains) of the rows of A / B
  int K, M, N;                              // unsegmented: rows of A, rows of B
  const int* a_seg_off; int pair_a_div;     // pair -> A segment pair / pair_a_div (null: the whole of A)
  const int* b_seg_off; const int* pair_b_seg; const int* pair_b_base;   // pair -> B segment pair_b_seg[pair] (+ base of the pair's group); < 0: empty pair
  int swap;                                 // 0: rows = the A segment, database = the B segment; 1: rows = the B segment, database = the A segment
  int k, pairs, row_stride;                 // neighbours per row; problems; rows reserved per pair in the per-row tables (>= the longest row segment)
  unsigned long long* lists; int* counts;   // (set by the launcher from the scratch block)
  unsigned long long* out_keys;             // [pairs * row_stride, k] (d2 bits << 32 | index) ascending, ~0 past the database; may be null
  float* out_d2; int* out_idx;              // [pairs * row_stride, k]; may be null
};
size_t knn_cand_scratch_bytes(int k, long long rows);   // rows = pairs * row_stride
constexpr int KNN_CAND_MAX_PAIRS = 65535;   // gridDim.y of the candidate pass; callers with more (detection, slot) pairs use the all-pairs tile
bool knn_cand_supported(int k, int K);
int knn_cand_launch(const KnnCandArgs& a, int max_rows, int max_db, void* scratch, hipStream_t st);   // max_db: the longest database segment (sizes the split)

// ---------------------------------------------------------------- match.hip
struct CyclicArgs {
  const int* q_off;        // [B+1] query-point segment per detection
  const int* tpl_ids;      // [B*n_slots] template id per (detection, slot); <0 = empty slot; global, or object-local with tpl_base
  const int* tpl_base;     // [B] first template of the detection's object (null: tpl_ids are global)
  const int* tpl_off;      // [T_total+1] feature segment per template
  const int* feat_base;    // [B] first feature row of the detection's object (ids are reported object-local)
  const float* points;     // [sumQ, 2]
  const float* vertices;   // [N_f, 3]
  const unsigned long long* row_best; int row_stride;  // [pairs, row_parts, row_stride] (d2, template patch): per column tile of the template
  const unsigned long long* col_best; int col_stride;  // [pairs, col_parts, col_stride] (d2, query patch): per row tile of the queries
  int row_parts, col_parts;
  int n_slots, top_k, k_max, q_max;
  int tie_mode;            // 0 canonical (value, index); 1 torch.topk's CPU tie order (stl_order.hpp)
  int* out_count;          // [pairs]
  int* out_q_ids;          // [pairs, k_max]
  int* out_feat_ids;       // [pairs, k_max]
  float* out_dists;        // [pairs, k_max]
  float* out_conf;         // [pairs, k_max]
  float* out_coord_2d;     // [pairs, k_max, 2]
  float* out_coord_3d;     // [pairs, k_max, 3]
};

struct SampleArgs {
  const float* fmap; long long stride_img, stride_c, stride_h, stride_w;
  int C, H, W, img_w, img_h;
  const float* points; const int* point_img; int num_points;
  float* out;  // [num_points, C]
};

struct WarpArgs {
  const void* src; int n_src, src_h, src_w, channels, mode;
  const int* src_index; const double* params; int batch, out_h, out_w, depth_check;
  void* out; float* map_out;
};
int launch_warp_crops(const WarpArgs& a, hipStream_t st);

int launch_sqnorm_rows(const float* x, long long n, int d, int ld, float* out, hipStream_t st);
int launch_normalize_rows(const float* x, long long n, int d, float eps, float* out, hipStream_t st);
int launch_topk_rows(const float* vals, int rows, int n, int ld, const int* row_len, int k, int largest,
                     float* out_val, int* out_idx, hipStream_t st);
int launch_tfidf_build(const int* word_ids, const float* word_d2, int knn_k, const int* seg_off, int num_segs,
                       const float* idf, int num_words, int soft, float sigma_sq, int sqrt_dists,
                       float* desc, float* desc_n, float eps, hipStream_t st);
int launch_cyclic_select(const CyclicArgs& a, int num_pairs, hipStream_t st);
int launch_sample_bilinear(const SampleArgs& a, hipStream_t st);
int launch_sqrt_inplace(float* x, long long n, hipStream_t st);

// ---------------------------------------------------------------- gemm_bf16.hip
enum : int {
  GEMM_EPI_BIAS_BF16 = 0,     // out(bf16) = acc + bias
  GEMM_EPI_GELU_BF16 = 1,     // out(bf16) = gelu_erf(acc + bias)
  GEMM_EPI_LS_RESID_F32 = 3,  // out(f32) += gamma * (acc + bias)      (LayerScale + residual)
  GEMM_EPI_TOKENS_F32 = 4,    // patch-embed rows scattered into the token sequence (+ bias + pos-embed)
  GEMM_EPI_BIAS_F32 = 5,      // out(f32) = acc + bias
  GEMM_EPI_RESID_F32 = 7,     // out(f32) += acc + bias (LayerScale already folded into W and bias) + the LayerNorm-fold outputs (xb, stats)
  GEMM_EPI_SWIGLU_BF16 = 6,   // SwiGLU FFN: columns interleaved (x1_j, x2_j) -> out(bf16)[:, j] = silu(x1_j) * x2_j, [M, N/2]
  GEMM_EPI_RESID_HILO = 8,    // as RESID_F32 with the residual stream held as TWO bf16 arrays: x = xb (hi) + xl (lo); reads both, adds acc + bias,
                              // writes hi' = bf16(x'), lo' = bf16(x' - hi') and the LayerNorm row sums -- no fp32 stream, no separate bf16 copy
};

struct GemmBf16Args {
  const __bf16* A; int lda;   // [M, K] activations (M padded to 128)
  const __bf16* W; int ldw;   // [N, K] weights (torch Linear layout)
  int M, N, K, M_valid;
  const float* bias;          // [N] or null
  const float* gamma;         // [N] LayerScale
  void* out; int ldo;
  const float* pos;           // [Np, ldo] fp32 pos-embed rows of the patch tokens
  int tok_np, tok_n, tok_skip;  // patches / tokens per image, first patch token index (1 + registers)
  int tile_override;          // 0 = auto, 64 (x 128) / 128 / 256 / 320 = force that block tile (benchmarks, tests)
  int no_tall;                // 1 = the automatic choice never takes the 320-row tile (fp_vit_model.flags & FP_VIT_NO_TALL_TILES: the A/B switch of a whole forward)
  int m_tiles;                // set by the launcher: row tiles that hold live rows (tiles of padding rows only are not launched)
  unsigned rast_r, rast_gn;
  float out_scale;            // fp8 kernels: > 0 -> the GELU / SwiGLU result leaves as e4m3(value * out_scale) bytes; f16x3: scale of a split-fp16 output
  float acc_scale;            // f16x3 kernels: out = epi(acc * acc_scale + bias), acc_scale = 1 / (scale of A x scale of W)
  unsigned long long* dbg;    // optional [grid, 4] shader-clock stamps: start, prologue done, main loop done, epilogue drained
  // ---- LayerNorm folded into the GEMMs around it (bf16 ViT blocks, vit forward only):
  // producer (LS_RESID): besides the fp32 residual stream it writes xb = bf16(x) -- the next GEMM's A operand -- and per
  // row the partial sums (sum x, sum x^2) of every 128-column group into stats[(column / 128) * M + row]
  __bf16* xb; int ld_xb;
  __bf16* xl;                 // RESID_HILO: the low halves of the stream (row stride ld_xb), read and written next to xb
  float2* stats_out;
  // consumer (BIAS / GELU / SwiGLU epilogues; W carries the LayerNorm gain, bias the LayerNorm shift):
  //   out = epi(rstd_r * (acc - mean_r * colsum_n) + bias_n);  ln_stats [M] = (rstd_r, mean_r * rstd_r) from ln_finalize_launch
  const float2* ln_stats; int ln_parts; float ln_eps;
  const float* colsum;        // [N] fp32 sums of the rows of W
  int* sat;                   // may be null; else [2] sticky saturation counters (common.hpp report_saturation): the split-fp16 / e4m3 epilogues report clamped outputs
};

int gemm_bf16_launch(int epi, const GemmBf16Args& a, hipStream_t st);
// f16x3 mode: A [M, 2K] and W [N, 2K] split-fp16 rows (common.hpp) behind the __bf16 pointers, a.K = the logical K (multiple of 32),
// lda / ldw in halves; out = epi(acc * a.acc_scale + bias); the BIAS / GELU (exact erf) / SwiGLU epilogues write split-fp16 rows
// again ([M, 2N] halves, values scaled by a.out_scale), LS_RESID / TOKENS / BIAS_F32 write fp32
int gemm_split_launch(int epi, const GemmBf16Args& a, hipStream_t st);
// f16f8 mode: the same with A and W as f16f8 rows (common.hpp; K a multiple of 64): hi*hi on the fp16 MFMA, the two cross terms on the fp8 MFMA;
// the GELU / SwiGLU outputs are f16f8 rows, the BIAS output (q | k | v) stays a split-fp16 row for the attention kernel
int gemm_splitx_launch(int epi, const GemmBf16Args& a, hipStream_t st);
int gemm_fp8_launch(int epi, const GemmBf16Args& a, hipStream_t st);  // A, W: OCP fp8 e4m3 bytes behind the __bf16 pointers
// "f16" mode: gemm_bf16_launch's kernels with IEEE fp16 operands and fp16 outputs behind the __bf16 pointers (A, W, out of the 16-bit epilogues, xb / xl
// of the residual epilogues); v_mfma_f32_32x32x16_f16, the nine-coefficient GELU polynomial; an fp16 output beyond +-65504 becomes inf (no report here: see common.hpp)
int gemm_f16_launch(int epi, const GemmBf16Args& a, hipStream_t st);

// ---------------------------------------------------------------- dtypes of the C ABI
enum : int { FP_DTYPE_F32 = 0, FP_DTYPE_BF16 = 1, FP_DTYPE_FP8 = 2, FP_DTYPE_F16X3 = 3, FP_DTYPE_F16F8 = 4,   // F16F8: common.hpp "f16f8 rows"
             FP_DTYPE_F16 = 5 };   // plain IEEE fp16 rows: the "f16" mode = the bf16 pipeline (same kernels, same bytes) on fp16 operands

// ---------------------------------------------------------------- attn.hip
struct AttnArgs {
  const void* qkv; int ld_qkv;   // [B*N, 3D] (bf16 or f32): q | k | v column blocks, head-major inside
  void* out; int ld_out;         // [B*N, D]
  int batch, n_tok, dim, heads;
  float out_fp8_scale;           // bf16 kernel: > 0 -> the output leaves as e4m3(o * scale) bytes (ld_out in bytes)
  // query selection (bf16, 64-queries-per-wave kernel; null = every token is a query): image b attends with the tokens
  // sel_rows[sel_off[b] .. sel_off[b+1]) (global rows b * n_tok + token, ascending) over ALL its keys, and output row r of
  // the compact [num_sel, D] result belongs to sel_rows[r]
  const int* sel_rows; const int* sel_off; int max_sel;  // max_sel >= the largest per-image count (sizes the grid)
  // bf16 work split (bit-identical outputs): 0 = 64 queries per wave, K/V by LDS-DMA (default); 1 = 32 queries per wave, register
  // staging (the cross-check); 2 = the DMA kernel with one 32-query block per wave, 8 waves per 256-query block
  int variant;
  int tail_last;                 // set by the launcher (bf16 w64 kernel): the last (short) query tile of every (image, head) pair goes to the END of its XCD's block sequence
  float in_scale, out_scale;     // f16x3 kernel: power-of-two scale the split-fp16 q / k / v rows carry, and the one the output row gets
  int out_fmt;                   // f16x3 kernel: 0 = the output is a split-fp16 row, 1 = an f16f8 row (common.hpp; the f16f8 mode's proj operand)
  int* sat;                      // may be null; else [2] sticky saturation counters: the e4m3 output of the bf16 kernel reports clamps; the split-fp16
                                 // output (a convex combination of v rows that already fit their scale) cannot clamp and reports non-finite values only
};
int attn_launch(const AttnArgs& a, int dtype, hipStream_t st);

// ---------------------------------------------------------------- vit.hip
struct LayerNormArgs {
  const float* x; int ld_x;      // fp32 residual stream
  const float* weight; const float* bias; float eps;
  void* out; int ld_out; int out_dtype;   // FP_DTYPE_FP8: e4m3(y * out_scale) bytes, ld_out in bytes (dim % 256 == 0)
  float out_scale;
  int dim;
  int out_rows;                  // rows to produce
  int out_rows_per_img, in_rows_per_img, in_skip;  // out row r -> in row (r / orpi) * irpi + in_skip + r % orpi
  int* sat;                      // may be null; else [2] sticky saturation counters: split-fp16 / e4m3 outputs report clamps, an fp32 output reports
                                 // non-finite values (the final norm of the "f16" mode)
};
int layernorm_launch(const LayerNormArgs& a, hipStream_t st);
int ln_sample_launch(const float* x, int ld_x, const float* weight, const float* bias, float eps, int apply_norm, int dim, int ntok, int skip,
                     int gh, int gw, int img_w, int img_h, const float* points, const int* point_img, int num_points, float* out, hipStream_t st,
                     const int* row_map = nullptr, int* sat = nullptr);   // sat: non-finite features are counted into sat[0] (the "f16" mode's overflow report)
int gather_rows_launch(const float* x, const int* rows, int n, int dim, float* out, hipStream_t st);
int query_select_launch(const unsigned char* masks, int B, int H, int W, const int* pix_x, const int* pix_y, const float* grid_pts, int G,
                        const long long* cells9, int C, int n_tok, int* scratch, int* counts, float* out_pts, int* out_img, int* q_off, int* sel_rows,
                        int* sel_off, int* row_map, hipStream_t st);
// x [rows, dim] fp32 -> xb = bf16(x) [rows, ld_xb] and stats[0 * stats_stride + row] = (sum x, sum x^2), slots 1..parts-1 zero
// partial sums [parts][rows] (sum x, sum x^2) over `dim` columns -> out[row] = (rstd, mean * rstd)
int ln_finalize_launch(const float2* partial, int parts, int stride, int rows, int dim, float eps, float2* out, hipStream_t st);
int rowstats_cast_launch(const float* x, int rows, int dim, void* xb, int ld_xb, float2* stats, int stats_stride, int parts, hipStream_t st,
                         void* xl = nullptr, bool h16 = false);   // xl: also write lo = bf16(x - bf16(x)) (row stride ld_xb); h16: fp16 instead of bf16
// out[r] = float(xb[row]) + float(xl[row]) (fp32 [n, dim]); row = rows[r], or r when rows is null: the (hi, lo) stream back as fp32
int hilo_rows_launch(const void* xb, const void* xl, int ld, const int* rows, int n, int dim, float* out, hipStream_t st, bool h16 = false);

int patchify_launch(const float* images, int batch, int height, int width, int patch, void* out, int ld_out,
                    int out_dtype, hipStream_t st, float out_scale = 1.f);  // out_scale: FP_DTYPE_F16X3 rows only
int patchify_strided_launch(const float* images, int batch, int height, int width, int patch, int stride, void* out, int ld_out, int out_dtype,
                            hipStream_t st, float out_scale = 1.f);
int prefix_tokens_launch(const float* prefix, int n_prefix, int dim, float* tokens, int batch, int n_tok, hipStream_t st);
int convert_f32_to_bf16_launch(const float* in, void* out, long long n, hipStream_t st);
int quantize_fp8_launch(const void* in, int in_dtype, long long n, float scale, void* out, hipStream_t st);
int launch_knn_merge(const unsigned long long* cand, int rows, int ncand, int k, float* out_d2, int* out_idx, hipStream_t st);
int launch_pack_records(const int* tpl_ids, const float* scores, const int* counts, const int* q_ids, const int* feat_ids, const float* dists, const float* conf,
                        const float* c2d, const float* c3d, int num_det, int n, int K, float* out, hipStream_t st);
int launch_unpack_best(const unsigned long long* best, long long n, float* d2, int* idx, hipStream_t st);

struct CosineArgs {
  const float* desc_n;  // [num_det, W] normalised query descriptors, grouped by object
  const float* bank_n;  // [T_total, W] normalised template descriptors
  const int* det_seg_off;  // [num_obj + 1]
  const int* obj_tpl_off;  // [num_obj + 1]
  int W;
  float* sims; int ld_sims;  // [num_det, ld_sims] finished scores
  int k_slices;              // set by the launcher: 8 when W % 128 == 0, else 1 (canonical chain split)
  unsigned long long* cand;  // [num_det, grid.x, n_top] candidate keys of the fused kernel (null: not wanted)
  int n_top;                 // candidates per (workgroup, detection): the caller's n_top, + 1 in the torch tie order
  int* need_replay;          // [num_det] torch tie order: 1 = the row has a tie among its best n_top + 1 scores
  const void* bank_bf16;     // [T_total, W] bf16 copy of bank_n (prefiltered retrieval only)
  int force_prefilter;       // prefiltered retrieval: take the two-stage form whatever the size (tests, measurements)
  const int* run_flag;       // may be null; else the kernel runs only if *run_flag != 0 (exact fallback of the prefiltered retrieval)
};
int launch_cosine_topk(const CosineArgs& a, int num_det, int num_obj, int max_det_per_obj, int max_templates, int n_top,
                       const int* det_num_templates, float* out_scores, int* out_ids, int tie_mode, hipStream_t st);
int launch_cosine_topk_prefiltered(const CosineArgs& a, int num_det, int num_obj, int max_det_per_obj, int max_templates, int n_top,
                                   const int* det_num_templates, float* out_scores, int* out_ids, int tie_mode, float* extra_scratch, hipStream_t st);
int launch_topn_rows(const float* sims, int ld, int rows, int max_len, const int* row_len, int n_top, float* out_scores,
                     int* out_ids, int tie_mode, hipStream_t st, const int* need_replay = nullptr);

// ---------------------------------------------------------------- pnp.hip
struct PnpArgs {
  const float* coord_2d;   // [pairs, k_max, 2] pixels
  const float* coord_3d;   // [pairs, k_max, 3] model space
  const int* counts;       // [pairs] valid correspondences per pair
  const double* cam;       // [pairs / n_slots, 4] fx, fy, cx, cy of each detection's (crop) camera
  int n_slots, k_max, iters, lm_iters, min_corresp;
  double thresh, conf;
  unsigned long long seed;
  int* success;            // [pairs]
  double* R;               // [pairs, 9] row-major model -> camera
  double* t;               // [pairs, 3]
  int* n_inliers;          // [pairs] RANSAC inliers of the winning model (= the reference's `quality`)
  unsigned char* inlier_mask;  // [pairs, k_max]
  double* ransac_pose;     // [pairs, 12] the winning model before refinement (R | t), may be null
};
int launch_pnp_ransac(const PnpArgs& a, int num_pairs, hipStream_t st);
