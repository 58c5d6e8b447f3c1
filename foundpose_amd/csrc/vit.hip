// Memory-bound ViT glue kernels: LayerNorm (+cast, + token re-indexing), patch extraction with the
// ImageNet normalisation folded in, CLS/register token rows.
//
// Reference behaviour restated: T.Normalize(mean,std) (/root/reference/utils/dinov2_utils.py:111-123),
// the backbone's patch_embed unfold, nn.LayerNorm(eps=1e-6) inside the blocks and the final
// `self.model.norm(tokens)` on CLS+patch tokens with the register tokens dropped (dinov2_utils.py:138-142,304).
#include "common.hpp"
#include "kernels.hpp"

namespace {

// One wave per row at a time; the row lives in registers, two-pass mean/var.  Waves are persistent (grid-stride over
// the rows) and fetch their next row before reducing the current one, so the 12 dependent shuffle steps of the two
// reductions overlap with HBM latency instead of following it.
// VEC = floats per lane and access: 4 (16-B loads, 8-B bf16 stores; needs D % 256 == 0: ViT-B/L/g) or 2 (ViT-S).
// MAXI = vectors a lane may hold of one row (dim <= 64 * VEC * MAXI), a compile-time bound on the two row images a wave keeps in registers (the current row
// and the next one, in flight): sized for dim 2048 the kernel needed 171 VGPRs = 2 waves per SIMD, and with 8 waves per CU x one prefetched row each the
// launch kept 12 MB in flight -- 3.9 TB/s at the ~3 us loaded latency (config 5: 80 launches x 349 us = 9 % of the step).  Instantiated per row length, the
// ViT-L / ViT-g rows take 4 / 6 vectors: ~70 / ~95 VGPRs, 5-7 waves per SIMD.  Same loops, same order: the same bits.
template <int VEC, int MAXI>
__global__ __launch_bounds__(256) void layernorm_kernel(LayerNormArgs a) {
  typedef __attribute__((ext_vector_type(VEC))) float vec_t;
  const int lane = threadIdx.x & 63;
  const int wstride = gridDim.x * 4;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.out_rows) return;
  const int nv = a.dim / (64 * VEC);  // vectors per lane
  auto row_ptr = [&](int r) {
    const int img = r / a.out_rows_per_img, p = r - img * a.out_rows_per_img;
    return a.x + ((size_t)img * a.in_rows_per_img + a.in_skip + p) * a.ld_x;
  };
  vec_t v[MAXI], nx[MAXI];
  float amax = 0.f;  // largest |scale * y| packed into a split-fp16 / e4m3 row (saturation report)
  {
    const float* x = row_ptr(row);
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
      if (i < nv) nx[i] = *reinterpret_cast<const vec_t*>(x + (i * 64 + lane) * VEC);
  }
  for (; row < a.out_rows; row += wstride) {
#pragma unroll
    for (int i = 0; i < MAXI; ++i) v[i] = nx[i];
    if (row + wstride < a.out_rows) {
      const float* x = row_ptr(row + wstride);
#pragma unroll
      for (int i = 0; i < MAXI; ++i)
        if (i < nv) nx[i] = *reinterpret_cast<const vec_t*>(x + (i * 64 + lane) * VEC);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
      if (i < nv) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) s += v[i][e];
      }
    const float mean = wave_sum(s) / (float)a.dim;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
      if (i < nv) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float d = v[i][e] - mean;
          ss += d * d;
        }
      }
    const float rstd = rsqrtf(wave_sum(ss) / (float)a.dim + a.eps);
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
      if (i < nv) {
        const int c = (i * 64 + lane) * VEC;
        const vec_t w = *reinterpret_cast<const vec_t*>(a.weight + c);
        const vec_t b = *reinterpret_cast<const vec_t*>(a.bias + c);
        vec_t y;
#pragma unroll
        for (int e = 0; e < VEC; ++e) y[e] = (v[i][e] - mean) * rstd * w[e] + b[e];
        if (a.out_dtype == FP_DTYPE_FP8) {
          if constexpr (VEC == 4)
            *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(a.out) + (size_t)row * a.ld_out + c) =
                pack_fp8x4(y[0] * a.out_scale, y[1] * a.out_scale, y[2] * a.out_scale, y[3] * a.out_scale, amax);
        } else if (a.out_dtype == FP_DTYPE_F16X3) {  // split-fp16 row (common.hpp): the next GEMM's A operand in the f16x3 mode
          _Float16* o = reinterpret_cast<_Float16*>(a.out) + (size_t)row * a.ld_out + split16_pos(c);
          unsigned h01, l01;
          split16_pack2(y[0], y[1], a.out_scale, h01, l01, amax);
          if constexpr (VEC == 4) {
            unsigned h23, l23;
            split16_pack2(y[2], y[3], a.out_scale, h23, l23, amax);
            *reinterpret_cast<uint2*>(o) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(o + 32) = make_uint2(l01, l23);
          } else {
            *reinterpret_cast<unsigned*>(o) = h01;
            *reinterpret_cast<unsigned*>(o + 32) = l01;
          }
        } else if (a.out_dtype == FP_DTYPE_F16F8) {  // f16f8 row (common.hpp): fp16 high halves + the e4m3 copies of hi and lo
          char* orow = reinterpret_cast<char*>(a.out) + (size_t)row * a.ld_out * 2;   // ld_out in halves
          unsigned h01, p01;
          splitx_pack2(y[0], y[1], a.out_scale, h01, p01, amax);
          if constexpr (VEC == 4) {
            unsigned h23, p23;
            splitx_pack2(y[2], y[3], a.out_scale, h23, p23, amax);
            splitx_store4(orow, c, h01, p01, h23, p23);
          } else {
            splitx_store2(orow, c, h01, p01);
          }
        } else if (a.out_dtype == FP_DTYPE_BF16) {
          __bf16* o = reinterpret_cast<__bf16*>(a.out) + (size_t)row * a.ld_out + c;
          if constexpr (VEC == 4) *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]));
          else *reinterpret_cast<unsigned*>(o) = pack_bf16x2(y[0], y[1]);
        } else if (a.out_dtype == FP_DTYPE_F16) {   // plain fp16 row (the "f16" mode's operand format)
          _Float16* o = reinterpret_cast<_Float16*>(a.out) + (size_t)row * a.ld_out + c;
          if constexpr (VEC == 4) *reinterpret_cast<uint2*>(o) = make_uint2(pack_h2<true>(y[0], y[1]), pack_h2<true>(y[2], y[3]));
          else *reinterpret_cast<unsigned*>(o) = pack_h2<true>(y[0], y[1]);
        } else {
          *reinterpret_cast<vec_t*>(reinterpret_cast<float*>(a.out) + (size_t)row * a.ld_out + c) = y;
          if (a.sat) {   // fp32 output with a counter (the final norm of the "f16" mode): non-finite features are reported (common.hpp, "16-bit operand formats")
#pragma unroll
            for (int e = 0; e < VEC; ++e) amax = nanmax3(amax, fabsf(y[e]), 0.f);
          }
        }
      }
  }
  if (a.out_dtype == FP_DTYPE_F16X3 || a.out_dtype == FP_DTYPE_F16F8) report_saturation(a.sat, 0, amax, a.out_dtype == FP_DTYPE_F16F8 ? FP_SX_MAX : FP_F16_MAX);
  else if (a.out_dtype == FP_DTYPE_F32) report_saturation(a.sat, 0, amax, 3.0e38f);
  else if (a.out_dtype == FP_DTYPE_FP8) report_saturation(a.sat, 1, amax, FP_E4M3_MAX);
}

// Final LayerNorm + bilinear sampling in one pass (SURVEY 8b `fp_ln_gather_pca`, the LN + gather half): the reference
// normalises the whole token map (dinov2_utils.py:138-142) and then samples it at the query points inside the mask
// (feature_util.py:100-131) -- 38 % of the cells at the metric's disc mask.  Here one wave per query point normalises the
// (up to) four tokens its bilinear footprint touches, straight from the residual stream, and combines them: the
// [B, Np, D] fp32 map (180 MB at batch 32) is never written or re-read.  Arithmetic identical to layernorm_kernel followed
// by sample_bilinear_kernel (same per-lane channel layout, same reduction order, same fma chain): bit-identical output.
struct LnSampleArgs {
  const float* x; int ld_x;            // residual stream [B * ntok, D]
  const float* weight; const float* bias; float eps; int apply_norm;
  int dim, ntok, skip, gh, gw, img_w, img_h;
  const float* points; const int* point_img; int num_points;
  float* out;                           // [num_points, dim]
  const int* row_map;                   // null: x holds every token.  Else x holds the SELECTED tokens only, compact:
                                        // row_map[img * gh * gw + cell] = the patch token's row in x (< 0: not selected)
  int* sat;                             // may be null; else [2] sticky counters: slot 0 counts threads that produced a non-finite feature (the "f16" mode's
                                        // overflow report: an fp16 activation beyond +-65504 anywhere in the backbone ends up here as inf / NaN)
};

// NVT = vectors of VEC floats a lane may hold of one row (dim <= 64 * VEC * NVT); small NVT keeps two rows in registers
// (the current tap's and the next tap's, in flight under the current one's two reductions) at 4+ waves per SIMD.
template <int VEC, int NVT>
__global__ __launch_bounds__(256) void ln_sample_kernel(LnSampleArgs a) {
  typedef __attribute__((ext_vector_type(VEC))) float vec_t;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= a.num_points) return;
  const int nv = a.dim / (64 * VEC);
  const int img = a.point_img ? a.point_img[p] : 0;
  const float px = a.points[2 * p], py = a.points[2 * p + 1];
  // coordinate and weight arithmetic of torch's CPU grid_sampler, as in sample_bilinear_kernel (match.hip)
  float ix, iy;
  {
#pragma clang fp contract(off)
    const float sx = 2.0f / (float)a.img_w, sy = 2.0f / (float)a.img_h;
    const float ux = sx * px, uy = sy * py;
    const float u1 = (ux - 1.0f) + 1.0f, v1 = (uy - 1.0f) + 1.0f;
    ix = fmaf(u1, (float)a.gw / 2.f, -0.5f);
    iy = fmaf(v1, (float)a.gh / 2.f, -0.5f);
  }
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  float wx1, wx0, wy1, wy0;
  {
#pragma clang fp contract(off)
    wx1 = ix - fx; wx0 = 1.f - wx1; wy1 = iy - fy; wy0 = 1.f - wy1;
  }
  // tap t of torch's order nw, ne, sw, se: grid cell (x0 + (t & 1), y0 + (t >> 1)); a tap outside the grid contributes zero
  auto tap_valid = [&](int t) {
    const int tx = x0 + (t & 1), ty = y0 + (t >> 1);
    return tx >= 0 && tx < a.gw && ty >= 0 && ty < a.gh;  // wave-uniform
  };
  auto tap_load = [&](int t, vec_t (&v)[NVT]) {
    int tx = x0 + (t & 1), ty = y0 + (t >> 1);
    tx = tx < 0 ? 0 : (tx >= a.gw ? a.gw - 1 : tx);   // an invalid tap reads a valid row and is zeroed below
    ty = ty < 0 ? 0 : (ty >= a.gh ? a.gh - 1 : ty);
    size_t row = (size_t)img * a.ntok + a.skip + ty * a.gw + tx;
    bool have = true;
    if (a.row_map) {
      const int r = a.row_map[(size_t)img * a.gh * a.gw + ty * a.gw + tx];
      have = r >= 0;  // a tap the selection missed would be a caller bug: it is made loud (NaN), never silently wrong
      row = have ? (size_t)r : 0;
    }
    const float* x = a.x + row * a.ld_x;
#pragma unroll
    for (int i = 0; i < NVT; ++i)
      if (i < nv) {
        v[i] = *reinterpret_cast<const vec_t*>(x + (i * 64 + lane) * VEC);
        if (!have) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[i][e] = __builtin_nanf("");
        }
      }
  };
  vec_t acc[NVT], v[NVT], vn[NVT];
#pragma unroll
  for (int i = 0; i < NVT; ++i)
#pragma unroll
    for (int e = 0; e < VEC; ++e) { acc[i][e] = 0.f; vn[i][e] = 0.f; }
  tap_load(0, v);
  // NOT unrolled: unrolled, the four taps' rows are all hoisted to the top (1792 spilled VGPRs at dim 2048)
#pragma unroll 1
  for (int t = 0; t < 4; ++t) {
    if (t < 3) tap_load(t + 1, vn);
    float wtt;
    {
#pragma clang fp contract(off)
      wtt = ((t >> 1) ? wy1 : wy0) * ((t & 1) ? wx1 : wx0);   // s*e, s*w, n*e, n*w
    }
    const bool valid = tap_valid(t);
    if (a.apply_norm) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NVT; ++i)
        if (i < nv) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) s += v[i][e];
        }
      const float mean = wave_sum(s) / (float)a.dim;
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < NVT; ++i)
        if (i < nv) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float d = v[i][e] - mean;
            ss += d * d;
          }
        }
      const float rstd = rsqrtf(wave_sum(ss) / (float)a.dim + a.eps);
#pragma unroll
      for (int i = 0; i < NVT; ++i)
        if (i < nv) {
          const int c = (i * 64 + lane) * VEC;
          const vec_t w = *reinterpret_cast<const vec_t*>(a.weight + c);
          const vec_t b = *reinterpret_cast<const vec_t*>(a.bias + c);
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[i][e] = (v[i][e] - mean) * rstd * w[e] + b[e];
        }
    }
#pragma unroll
    for (int i = 0; i < NVT; ++i)
      if (i < nv) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float val = valid ? v[i][e] : 0.f;
          if (t == 0) {
#pragma clang fp contract(off)
            acc[i][e] = val * wtt;
          } else {
            acc[i][e] = fmaf(val, wtt, acc[i][e]);
          }
        }
      }
#pragma unroll
    for (int i = 0; i < NVT; ++i) v[i] = vn[i];
  }
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NVT; ++i)
    if (i < nv) {
      *reinterpret_cast<vec_t*>(a.out + (size_t)p * a.dim + (i * 64 + lane) * VEC) = acc[i];
#pragma unroll
      for (int e = 0; e < VEC; ++e) amax = nanmax3(amax, fabsf(acc[i][e]), 0.f);
    }
  report_saturation(a.sat, 0, amax, 3.0e38f);   // (sat null: no report)
}

// images [B,3,H,W] in [0,1] -> rows of the patch-embed GEMM: row = b*Np + gy*gw + gx,
// column = c*P*P + py*P + px (the flattening of the conv weight [D,3,P,P]); columns >= 3*P*P are zero.
// A patch row is 14 pixels wide: whichever side a thread walks, the other side is touched in 28/56-byte pieces.  So one
// workgroup takes one ROW OF PATCHES (image b, patch row gy) and builds it in LDS in OUTPUT layout: a wave reads whole image
// rows (coalesced, two rows = 18 loads in flight), normalises, and drops each pixel at [patch gx][c*P*P + py*P + px] through
// a per-workgroup x -> (gx, px) table (no per-pixel division); the gw output rows then leave as whole 16-byte chunks.
// (History: one thread per element 120 us, one per 14-pixel run 117 us, a strip in input layout with index divisions in
//  both phases 92 us.)
// SPLIT (f16x3 mode): T = _Float16 and a row is the split-fp16 image (common.hpp) of the normalised pixels times `scale`; ld = 2 x
// the padded column count.
template <typename T, int SPLIT = 0>   // SPLIT: 0 plain rows of T, 1 split-fp16 rows, 2 f16f8 rows (common.hpp)
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, int B, int H, int W, int P, T* __restrict__ out, int ld, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int V = 16 / sizeof(T);  // elements per 16-byte chunk
  const int gw = W / P, gh = H / P, lds_ld = ld + V;  // + one chunk per row: the patches' rows start in different banks
  T* tile = reinterpret_cast<T*>(smem_raw);                                 // [gw][lds_ld]
  unsigned short* xtab = reinterpret_cast<unsigned short*>(tile + (size_t)gw * lds_ld);  // [W]: gx * lds_ld is too wide for 16 bits: (gx << 8) | px
  const int b = blockIdx.x / gh, gy = blockIdx.x - b * gh, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncols = 3 * P * P;
  for (int x = tid; x < W; x += 256) {
    const int gx = x / P;
    xtab[x] = (unsigned short)((gx << 8) | (x - gx * P));
  }
  if constexpr (SPLIT) {  // hi and lo halves of the padding columns are scattered over the row: clear all of it
    for (int id = tid; id < gw * ld; id += 256) {
      const int gx = id / ld;
      tile[(size_t)gx * lds_ld + (id - gx * ld)] = (T)0.f;
    }
  } else {
    const int pad = ld - ncols;  // zero columns behind the pixels
    for (int id = tid; id < gw * pad; id += 256) {
      const int gx = id / pad;
      tile[(size_t)gx * lds_ld + ncols + (id - gx * pad)] = (T)0.f;
    }
  }
  __syncthreads();
  const float* img_b = img + (size_t)b * 3 * H * W;
  constexpr int XI = 9;  // 64-lane pieces of an image row held at once (W <= 576; wider rows loop)
  for (int r0 = wave; r0 < 3 * P; r0 += 8) {
    float v[2][XI];
    for (int xb = 0; xb < W; xb += 64 * XI) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = r0 + 4 * h;
        const int c = r / P, py = r - c * P;  // wave-uniform
        const float* src = img_b + ((size_t)c * H + gy * P + py) * W;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
          const int x = xb + lane + 64 * i;
          v[h][i] = (r < 3 * P && x < W) ? src[x] : 0.f;
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = r0 + 4 * h;
        if (r >= 3 * P) continue;
        const int c = r / P;
        const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f);
        const float stdv = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
#pragma unroll
        for (int i = 0; i < XI; ++i) {
          const int x = xb + lane + 64 * i;
          if (x < W) {
            const unsigned t = xtab[x];
            const float nv = (v[h][i] - mean) / stdv;  // T.Normalize: sub then div
            if constexpr (SPLIT == 2) {
              unsigned h, p8;
              splitx_pack2(nv, 0.f, scale, h, p8);
              const int col = r * P + (int)(t & 255u);
              char* rowb = reinterpret_cast<char*>(tile + (size_t)(t >> 8) * lds_ld);
              *reinterpret_cast<unsigned short*>(rowb + splitx_pos(col) * 2) = (unsigned short)h;
              rowb[splitx_hi8(col)] = (char)p8;
              rowb[splitx_hi8(col) + 64] = (char)(p8 >> 16);
            } else if constexpr (SPLIT == 1) {
              const float sv = nv * scale;
              const _Float16 hi = (_Float16)sv;
              T* d = tile + (size_t)(t >> 8) * lds_ld + split16_pos(r * P + (int)(t & 255u));
              d[0] = hi;
              d[32] = (_Float16)(sv - (float)hi);
            } else {
              tile[(size_t)(t >> 8) * lds_ld + r * P + (t & 255u)] = (T)nv;
            }
          }
        }
      }
    }
  }
  __syncthreads();
  const int chunks = ld / V;
  T* out_row = out + (size_t)(b * gh + gy) * gw * ld;
  for (int gx = wave; gx < gw; gx += 4)
    for (int ch = lane; ch < chunks; ch += 64)
      *reinterpret_cast<uint4*>(out_row + (size_t)gx * ld + ch * V) = *reinterpret_cast<const uint4*>(tile + (size_t)gx * lds_ld + ch * V);
}

// stride != patch size (the reference's patch_vit_resolution, dinov2_utils.py:364-389: the conv of the patch embedding runs with a
// smaller stride, patches overlap): row = b * gh * gw + gy * gw + gx with gh = 1 + (H - P) / stride, same column order.  One thread per
// output element -- a rarely used configuration, not a hot kernel.  OUT: 0 fp32, 1 bf16, 2 split-fp16, 3 f16f8 rows (ld in halves).
template <int OUT>
__global__ __launch_bounds__(256) void patchify_strided_kernel(const float* __restrict__ img, int B, int H, int W, int P, int stride, int gh, int gw,
                                                               void* __restrict__ out, int ld, int cols_pad, float scale) {
  const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)B * gh * gw * cols_pad;
  if (id >= total) return;
  const int col = (int)(id % cols_pad);
  const long long row = id / cols_pad;
  const int gx = (int)(row % gw), gy = (int)((row / gw) % gh), b = (int)(row / ((long long)gw * gh));
  float v = 0.f;
  if (col < 3 * P * P) {
    const int c = col / (P * P), py = (col - c * P * P) / P, px = col - c * P * P - py * P;
    const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f);
    const float stdv = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
    v = (img[(((size_t)b * 3 + c) * H + gy * stride + py) * W + gx * stride + px] - mean) / stdv;
  }
  if constexpr (OUT == 0) reinterpret_cast<float*>(out)[(size_t)row * ld + col] = v;
  else if constexpr (OUT == 1) reinterpret_cast<__bf16*>(out)[(size_t)row * ld + col] = (__bf16)v;
  else if constexpr (OUT == 4) reinterpret_cast<_Float16*>(out)[(size_t)row * ld + col] = (_Float16)v;   // plain fp16 rows (the "f16" mode)
  else if constexpr (OUT == 3) {
    unsigned h, p8;
    splitx_pack2(v, 0.f, scale, h, p8);
    char* rowb = reinterpret_cast<char*>(out) + (size_t)row * ld * 2;
    *reinterpret_cast<unsigned short*>(rowb + splitx_pos(col) * 2) = (unsigned short)h;
    rowb[splitx_hi8(col)] = (char)p8;
    rowb[splitx_hi8(col) + 64] = (char)(p8 >> 16);
  } else {
    const float sv = v * scale;
    const _Float16 hi = (_Float16)sv;
    _Float16* d = reinterpret_cast<_Float16*>(out) + (size_t)row * ld + split16_pos(col);
    d[0] = hi;
    d[32] = (_Float16)(sv - (float)hi);
  }
}

// Entry of the folded-LayerNorm block chain: xb = bf16(x) and the row sums (sum x, sum x^2) of the token embedding, which no
// LayerScale GEMM has produced yet.  One wave per row; slot 0 of the partial-sum table gets the whole row, the others zero.
// H16: the 16-bit arrays are IEEE fp16 (the "f16" mode) instead of bf16.
template <bool H16>
__global__ __launch_bounds__(256) void rowstats_cast_kernel(const float* __restrict__ x, int rows, int dim, __bf16* __restrict__ xb, int ld_xb,
                                                            float2* __restrict__ stats, int stats_stride, int parts, __bf16* __restrict__ xl) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * dim;
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane * 4; c < dim; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + c);
    const uint2 h = make_uint2(pack_h2<H16>(v.x, v.y), pack_h2<H16>(v.z, v.w));
    *reinterpret_cast<uint2*>(xb + (size_t)row * ld_xb + c) = h;
    if (xl) {  // the (hi, lo) residual stream: lo = 16-bit(x - hi)
      const f32x2 h01 = unpack_h2<H16>(h.x), h23 = unpack_h2<H16>(h.y);
      *reinterpret_cast<uint2*>(xl + (size_t)row * ld_xb + c) = make_uint2(pack_h2<H16>(v.x - h01[0], v.y - h01[1]), pack_h2<H16>(v.z - h23[0], v.w - h23[1]));
    }
    s1 += (v.x + v.y) + (v.z + v.w);
    s2 += fmaf(v.x, v.x, v.y * v.y) + fmaf(v.z, v.z, v.w * v.w);
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane < parts) stats[(size_t)lane * stats_stride + row] = lane == 0 ? make_float2(s1, s2) : make_float2(0.f, 0.f);
}

// Folded LayerNorm: the residual GEMM's column tiles leave `parts` partial sums per row; one thread per row adds them in
// slot order and leaves (rstd, mean * rstd), the two numbers the next GEMM's epilogue needs.
__global__ void ln_finalize_kernel(const float2* __restrict__ partial, int parts, int stride, int rows, float inv_dim, float eps, float2* __restrict__ out) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  float s1 = 0.f, s2 = 0.f;
  for (int p = 0; p < parts; ++p) {
    const float2 v = partial[(size_t)p * stride + row];
    s1 += v.x;
    s2 += v.y;
  }
  const float mu = s1 * inv_dim;
  const float rs = rsqrtf(fmaxf(s2 * inv_dim - mu * mu, 0.f) + eps);
  out[row] = make_float2(rs, mu * rs);
}

// ---- query points and token selection of a batch (fp_query_select): the mask test of every grid point
// (filter_points_by_mask, feature_util.py:36-41 in the reference: pixel = int(point + 0.5) strictly inside the canvas and on
// the mask), the per-image point lists in grid order, and -- for the hooked block -- which patch tokens the sampling of those
// points will read.  Kernel 1, one workgroup per image: ordered ballot scans rank the live points; the cells they name
// (cells9: for every grid point the 3 x 3 cells around its sampling position, a superset of the four bilinear taps) are
// flagged in LDS and ranked the same way.  Kernel 2 adds the counts of the images in front and writes the lists.
FP_DEVICE int block_rank256(bool f, int& base, int* wave_tot, int lane, int wave) {
  const unsigned long long m = __ballot(f);
  const int before = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) wave_tot[wave] = __popcll(m);
  __syncthreads();
  int woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wave) woff += wave_tot[w];
    tot += wave_tot[w];
  }
  const int r = f ? base + woff + before : -1;
  base += tot;
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void query_flags_kernel(const unsigned char* __restrict__ masks, int H, int W, const int* __restrict__ pix_x,
                                                          const int* __restrict__ pix_y, int G, const long long* __restrict__ cells9, int C,
                                                          int* __restrict__ point_rank, int* __restrict__ cell_rank, int* __restrict__ counts) {
  extern __shared__ int sel_smem[];
  int* flag = sel_smem;            // [C + 1] (slot C collects the cells outside the map)
  __shared__ int wave_tot[4];
  const int b = blockIdx.x, B = gridDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (cells9) {
    for (int c = tid; c <= C; c += 256) flag[c] = 0;
    __syncthreads();
  }
  const unsigned char* mk = masks + (size_t)b * H * W;
  int base = 0;
  for (int g0 = 0; g0 < G; g0 += 256) {
    const int g = g0 + tid;
    bool on = false;
    if (g < G) {
      const int x = pix_x[g], y = pix_y[g];
      on = x > 0 && x < W && y > 0 && y < H && mk[(size_t)y * W + x] != 0;
    }
    if (on && cells9) {
#pragma unroll
      for (int k = 0; k < 9; ++k) flag[(int)cells9[(size_t)g * 9 + k]] = 1;  // benign race: every writer stores 1
    }
    const int r = block_rank256(on, base, wave_tot, lane, wave);
    if (g < G) point_rank[(size_t)b * G + g] = r;
  }
  if (tid == 0) counts[b] = base;
  if (!cells9) return;
  __syncthreads();
  base = 0;
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int c = c0 + tid;
    const int r = block_rank256(c < C && flag[c] != 0, base, wave_tot, lane, wave);
    if (c < C) cell_rank[(size_t)b * C + c] = r;
  }
  if (tid == 0) counts[B + b] = base;
}

__global__ __launch_bounds__(256) void query_lists_kernel(const int* __restrict__ point_rank, const int* __restrict__ cell_rank,
                                                          const int* __restrict__ counts, int G, int C, int n_tok, const float* __restrict__ grid_pts,
                                                          float* __restrict__ out_pts, int* __restrict__ out_img, int* __restrict__ q_off,
                                                          int* __restrict__ sel_rows, int* __restrict__ sel_off, int* __restrict__ row_map) {
  __shared__ int s_off[2];
  const int b = blockIdx.x, B = gridDim.x, tid = threadIdx.x;
  if (tid < 128) {  // wave 0: points in front of this image, wave 1: selected cells in front of it
    const int which = tid >> 6, lane = tid & 63;
    int part = 0;
    if (which == 0 || cell_rank)
      for (int i = lane; i < b; i += 64) part += counts[which * B + i];
    part = (int)wave_sum((float)part);  // < 2^24: exact in fp32
    if (lane == 0) {
      s_off[which] = part;
      int* off = which ? sel_off : q_off;
      if (off && (which == 0 || cell_rank)) {
        off[b] = part;
        if (b == B - 1) off[B] = part + counts[which * B + b];
      }
    }
  }
  __syncthreads();
  const int poff = s_off[0], coff = s_off[1], skip = n_tok - C;
  for (int g = tid; g < G; g += 256) {
    const int r = point_rank[(size_t)b * G + g];
    if (r >= 0) {
      out_pts[2 * (size_t)(poff + r)] = grid_pts[2 * g];
      out_pts[2 * (size_t)(poff + r) + 1] = grid_pts[2 * g + 1];
      out_img[poff + r] = b;
    }
  }
  if (!cell_rank) return;
  for (int c = tid; c < C; c += 256) {
    const int r = cell_rank[(size_t)b * C + c];
    row_map[(size_t)b * C + c] = r < 0 ? -1 : coff + r;
    if (r >= 0) sel_rows[coff + r] = b * n_tok + skip + c;
  }
}

// rows of the fp32 residual stream picked by index: out[r] = x[rows[r]] (the selected tokens of the hooked block)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, const int* __restrict__ rows, int n, int dim, float* __restrict__ out) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= n) return;
  const float4* src = reinterpret_cast<const float4*>(x + (size_t)rows[r] * dim);
  float4* dst = reinterpret_cast<float4*>(out + (size_t)r * dim);
  for (int c = lane; c < dim / 4; c += 64) dst[c] = src[c];
}

// the (hi, lo) bf16 residual stream back as fp32 rows: out[r] = hi[row] + lo[row], row = rows[r] (the selected tokens of the hooked block) or r
template <bool H16>   // H16: the pair is IEEE fp16 (the "f16" mode)
__global__ __launch_bounds__(256) void hilo_rows_kernel(const __bf16* __restrict__ xb, const __bf16* __restrict__ xl, int ld, const int* __restrict__ rows,
                                                        int n, int dim, float* __restrict__ out) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= n) return;
  const size_t row = rows ? (size_t)rows[r] : (size_t)r;
  for (int c = lane * 4; c < dim; c += 256) {
    const uint2 h = *reinterpret_cast<const uint2*>(xb + row * ld + c), l = *reinterpret_cast<const uint2*>(xl + row * ld + c);
    const f32x2 h01 = unpack_h2<H16>(h.x), h23 = unpack_h2<H16>(h.y), l01 = unpack_h2<H16>(l.x), l23 = unpack_h2<H16>(l.y);
    *reinterpret_cast<float4*>(out + (size_t)r * dim + c) = make_float4(h01[0] + l01[0], h01[1] + l01[1], h23[0] + l23[0], h23[1] + l23[1]);
  }
}

__global__ void prefix_tokens_kernel(const float* __restrict__ prefix, int n_prefix, int dim, float* __restrict__ tokens, int batch, int n_tok) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)batch * n_prefix * dim;
  if (e >= total) return;
  const int c = (int)(e % dim);
  const int r = (int)((e / dim) % n_prefix);
  const int b = (int)(e / ((long long)dim * n_prefix));
  tokens[((size_t)b * n_tok + r) * dim + c] = prefix[(size_t)r * dim + c];
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ in, __bf16* __restrict__ out, long long n) {
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i + 1 < n) {
    *reinterpret_cast<unsigned*>(out + i) = pack_bf16x2(in[i], in[i + 1]);
  } else if (i < n) {
    out[i] = (__bf16)in[i];
  }
}

}  // namespace

int layernorm_launch(const LayerNormArgs& a, hipStream_t st) {
  FP_REQUIRE(a.dim % 128 == 0 && a.dim <= 2048, "layernorm: dim must be a multiple of 128 and <= 2048 (got %d)", a.dim);
  FP_REQUIRE(a.ld_x % 2 == 0 && a.ld_out % 2 == 0, "layernorm: leading dims must be even");
  if (a.out_rows == 0) return FP_OK;
  const int wgs = cdiv(a.out_rows, 4);
  const int grid = wgs < 2048 ? wgs : 2048;  // up to 8 workgroups (32 waves) per CU, each wave walks out_rows / 8192 rows
  FP_REQUIRE(a.out_dtype != FP_DTYPE_FP8 || (a.dim % 256 == 0 && a.ld_x % 4 == 0 && a.ld_out % 4 == 0 && a.out_scale > 0.f),
             "layernorm: fp8 output needs dim %% 256 == 0 and a positive scale");
  FP_REQUIRE((a.out_dtype != FP_DTYPE_F16X3 && a.out_dtype != FP_DTYPE_F16F8) || (a.ld_out >= 2 * a.dim && a.ld_out % 4 == 0 && a.out_scale > 0.f),
             "layernorm: a split-fp16 / f16f8 output row is 2 * dim halves and needs a positive scale");
  if (a.dim % 256 == 0 && a.ld_x % 4 == 0 && a.ld_out % 4 == 0) {
    if (a.dim <= 512) hipLaunchKernelGGL((layernorm_kernel<4, 2>), dim3(grid), dim3(256), 0, st, a);
    else if (a.dim <= 1024) hipLaunchKernelGGL((layernorm_kernel<4, 4>), dim3(grid), dim3(256), 0, st, a);
    else if (a.dim <= 1536) hipLaunchKernelGGL((layernorm_kernel<4, 6>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((layernorm_kernel<4, 8>), dim3(grid), dim3(256), 0, st, a);
  } else {
    if (a.dim <= 384) hipLaunchKernelGGL((layernorm_kernel<2, 3>), dim3(grid), dim3(256), 0, st, a);
    else if (a.dim <= 768) hipLaunchKernelGGL((layernorm_kernel<2, 6>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((layernorm_kernel<2, 16>), dim3(grid), dim3(256), 0, st, a);
  }
  FP_CHECK_LAUNCH("layernorm");
  return FP_OK;
}

int query_select_launch(const unsigned char* masks, int B, int H, int W, const int* pix_x, const int* pix_y, const float* grid_pts, int G,
                        const long long* cells9, int C, int n_tok, int* scratch, int* counts, float* out_pts, int* out_img, int* q_off, int* sel_rows,
                        int* sel_off, int* row_map, hipStream_t st) {
  FP_REQUIRE(B >= 1 && G >= 1 && H >= 1 && W >= 1, "query_select: bad sizes");
  FP_REQUIRE(!cells9 || (C >= 1 && n_tok >= C && (size_t)(C + 1) * 4 <= 64 * 1024), "query_select: bad token map");
  FP_REQUIRE((long long)B * (G > n_tok ? G : n_tok) < (1ll << 24), "query_select: batch too large");
  int* point_rank = scratch;
  int* cell_rank = cells9 ? scratch + (size_t)B * G : nullptr;
  hipLaunchKernelGGL(query_flags_kernel, dim3(B), dim3(256), cells9 ? (size_t)(C + 1) * 4 : 0, st, masks, H, W, pix_x, pix_y, G, cells9, C, point_rank,
                     cell_rank, counts);
  FP_CHECK_LAUNCH("query_flags");
  hipLaunchKernelGGL(query_lists_kernel, dim3(B), dim3(256), 0, st, point_rank, cell_rank, counts, G, C, n_tok, grid_pts, out_pts, out_img, q_off, sel_rows,
                     sel_off, row_map);
  FP_CHECK_LAUNCH("query_lists");
  return FP_OK;
}

int gather_rows_launch(const float* x, const int* rows, int n, int dim, float* out, hipStream_t st) {
  FP_REQUIRE(dim % 4 == 0, "gather_rows: dim %% 4 != 0");
  if (n == 0) return FP_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, x, rows, n, dim, out);
  FP_CHECK_LAUNCH("gather_rows");
  return FP_OK;
}

int ln_sample_launch(const float* x, int ld_x, const float* weight, const float* bias, float eps, int apply_norm, int dim, int ntok, int skip,
                     int gh, int gw, int img_w, int img_h, const float* points, const int* point_img, int num_points, float* out, hipStream_t st,
                     const int* row_map, int* sat) {
  FP_REQUIRE(dim % 128 == 0 && dim <= 2048, "ln_sample: dim (%d) must be a multiple of 128, at most 2048", dim);
  if (num_points == 0) return FP_OK;
  LnSampleArgs a{x, ld_x, weight, bias, eps, apply_norm, dim, ntok, skip, gh, gw, img_w, img_h, points, point_img, num_points, out, row_map, sat};
  const dim3 grid(cdiv(num_points, 4));
  if (dim % 256 == 0 && dim <= 1024) hipLaunchKernelGGL((ln_sample_kernel<4, 4>), grid, dim3(256), 0, st, a);
  else if (dim % 256 == 0) hipLaunchKernelGGL((ln_sample_kernel<4, 8>), grid, dim3(256), 0, st, a);
  else if (dim <= 512) hipLaunchKernelGGL((ln_sample_kernel<2, 4>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((ln_sample_kernel<2, 16>), grid, dim3(256), 0, st, a);
  FP_CHECK_LAUNCH("ln_sample");
  return FP_OK;
}

int patchify_strided_launch(const float* images, int batch, int height, int width, int patch, int stride, void* out, int ld_out, int out_dtype,
                            hipStream_t st, float out_scale) {
  FP_REQUIRE(stride >= 1 && height >= patch && width >= patch, "patchify: stride %d / image %dx%d / patch %d", stride, height, width, patch);
  const int gh = 1 + (height - patch) / stride, gw = 1 + (width - patch) / stride;
  const bool split = out_dtype == FP_DTYPE_F16X3 || out_dtype == FP_DTYPE_F16F8;
  const int cols_pad = split ? ld_out / 2 : ld_out;   // logical columns of a row (zero beyond 3 P^2)
  FP_REQUIRE(cols_pad >= 3 * patch * patch && (!split || ld_out % 64 == 0) && (out_dtype != FP_DTYPE_F16F8 || ld_out % 128 == 0), "patchify: ld_out too small");
  const long long total = (long long)batch * gh * gw * cols_pad;
  if (total == 0) return FP_OK;
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (out_dtype == FP_DTYPE_F16F8) hipLaunchKernelGGL(patchify_strided_kernel<3>, dim3(grid), dim3(256), 0, st, images, batch, height, width, patch, stride, gh, gw, out, ld_out, cols_pad, out_scale);
  else if (split) hipLaunchKernelGGL(patchify_strided_kernel<2>, dim3(grid), dim3(256), 0, st, images, batch, height, width, patch, stride, gh, gw, out, ld_out, cols_pad, out_scale);
  else if (out_dtype == FP_DTYPE_BF16) hipLaunchKernelGGL(patchify_strided_kernel<1>, dim3(grid), dim3(256), 0, st, images, batch, height, width, patch, stride, gh, gw, out, ld_out, cols_pad, 1.f);
  else if (out_dtype == FP_DTYPE_F16) hipLaunchKernelGGL(patchify_strided_kernel<4>, dim3(grid), dim3(256), 0, st, images, batch, height, width, patch, stride, gh, gw, out, ld_out, cols_pad, 1.f);
  else hipLaunchKernelGGL(patchify_strided_kernel<0>, dim3(grid), dim3(256), 0, st, images, batch, height, width, patch, stride, gh, gw, out, ld_out, cols_pad, 1.f);
  FP_CHECK_LAUNCH("patchify_strided");
  return FP_OK;
}

int patchify_launch(const float* images, int batch, int height, int width, int patch, void* out, int ld_out,
                    int out_dtype, hipStream_t st, float out_scale) {
  FP_REQUIRE(height % patch == 0 && width % patch == 0, "patchify: image %dx%d is not a multiple of the patch size %d", height, width, patch);
  FP_REQUIRE(ld_out >= 3 * patch * patch, "patchify: ld_out too small");
  const unsigned grid = (unsigned)(batch * (height / patch));  // one workgroup per row of patches
  if (grid == 0) return FP_OK;
  const bool split = out_dtype == FP_DTYPE_F16X3 || out_dtype == FP_DTYPE_F16F8;  // ld_out = halves per row = 2 x the padded column count (a multiple of 32; f16f8: of 64)
  FP_REQUIRE(!split || (ld_out % 64 == 0 && ld_out >= 2 * ((3 * patch * patch + 31) / 32 * 32)), "patchify: a split-fp16 row is 2 x the columns padded to 32");
  FP_REQUIRE(out_dtype != FP_DTYPE_F16F8 || (ld_out % 128 == 0 && ld_out >= 2 * ((3 * patch * patch + 63) / 64 * 64)), "patchify: an f16f8 row is 2 x the columns padded to 64");
  const size_t esz = out_dtype == FP_DTYPE_F32 ? 4 : 2;
  FP_REQUIRE(ld_out % (16 / esz) == 0, "patchify: ld_out must keep 16-byte rows");
  const size_t lds = (size_t)(width / patch) * (ld_out + 16 / esz) * esz + (size_t)width * 2;  // the row of patches in output layout + the x table
  FP_REQUIRE(lds <= 160 * 1024 && patch <= 255 && width / patch <= 255, "patchify: a row of patches (%d x %d columns) does not fit LDS", width / patch, ld_out);
  static FpDeviceOnce attr_b, attr_f, attr_s, attr_x, attr_h;
  fp_allow_dynamic_lds(attr_h, &patchify_kernel<_Float16, 0>, 160 * 1024);
  fp_allow_dynamic_lds(attr_x, &patchify_kernel<_Float16, 2>, 160 * 1024);
  fp_allow_dynamic_lds(attr_b, &patchify_kernel<__bf16>, 160 * 1024);
  fp_allow_dynamic_lds(attr_f, &patchify_kernel<float>, 160 * 1024);
  fp_allow_dynamic_lds(attr_s, &patchify_kernel<_Float16, 1>, 160 * 1024);
  if (out_dtype == FP_DTYPE_F16F8)
    hipLaunchKernelGGL((patchify_kernel<_Float16, 2>), dim3(grid), dim3(256), lds, st, images, batch, height, width, patch, reinterpret_cast<_Float16*>(out), ld_out, out_scale);
  else if (split)
    hipLaunchKernelGGL((patchify_kernel<_Float16, 1>), dim3(grid), dim3(256), lds, st, images, batch, height, width, patch, reinterpret_cast<_Float16*>(out), ld_out, out_scale);
  else if (out_dtype == FP_DTYPE_BF16)
    hipLaunchKernelGGL(patchify_kernel<__bf16>, dim3(grid), dim3(256), lds, st, images, batch, height, width, patch, reinterpret_cast<__bf16*>(out), ld_out, 1.f);
  else if (out_dtype == FP_DTYPE_F16)   // plain fp16 rows (normalised pixels: |v| < 3)
    hipLaunchKernelGGL((patchify_kernel<_Float16, 0>), dim3(grid), dim3(256), lds, st, images, batch, height, width, patch, reinterpret_cast<_Float16*>(out), ld_out, 1.f);
  else
    hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid), dim3(256), lds, st, images, batch, height, width, patch, reinterpret_cast<float*>(out), ld_out, 1.f);
  FP_CHECK_LAUNCH("patchify");
  return FP_OK;
}

int ln_finalize_launch(const float2* partial, int parts, int stride, int rows, int dim, float eps, float2* out, hipStream_t st) {
  if (rows == 0) return FP_OK;
  hipLaunchKernelGGL(ln_finalize_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, st, partial, parts, stride, rows, 1.f / (float)dim, eps, out);
  FP_CHECK_LAUNCH("ln_finalize");
  return FP_OK;
}

int rowstats_cast_launch(const float* x, int rows, int dim, void* xb, int ld_xb, float2* stats, int stats_stride, int parts, hipStream_t st, void* xl, bool h16) {
  FP_REQUIRE(dim % 4 == 0 && ld_xb % 4 == 0 && parts >= 1 && parts <= 64, "rowstats_cast: dim / ld_xb must be multiples of 4, parts in [1, 64]");
  if (rows == 0) return FP_OK;
  if (h16)
    hipLaunchKernelGGL(rowstats_cast_kernel<true>, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, rows, dim, reinterpret_cast<__bf16*>(xb), ld_xb, stats, stats_stride, parts,
                       reinterpret_cast<__bf16*>(xl));
  else
    hipLaunchKernelGGL(rowstats_cast_kernel<false>, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, rows, dim, reinterpret_cast<__bf16*>(xb), ld_xb, stats, stats_stride, parts,
                       reinterpret_cast<__bf16*>(xl));
  FP_CHECK_LAUNCH("rowstats_cast");
  return FP_OK;
}

int hilo_rows_launch(const void* xb, const void* xl, int ld, const int* rows, int n, int dim, float* out, hipStream_t st, bool h16) {
  FP_REQUIRE(xb && xl && out && dim % 4 == 0 && ld % 4 == 0, "hilo_rows: bad arguments");
  if (n == 0) return FP_OK;
  if (h16) hipLaunchKernelGGL(hilo_rows_kernel<true>, dim3(cdiv(n, 4)), dim3(256), 0, st, reinterpret_cast<const __bf16*>(xb), reinterpret_cast<const __bf16*>(xl), ld, rows, n, dim, out);
  else hipLaunchKernelGGL(hilo_rows_kernel<false>, dim3(cdiv(n, 4)), dim3(256), 0, st, reinterpret_cast<const __bf16*>(xb), reinterpret_cast<const __bf16*>(xl), ld, rows, n, dim, out);
  FP_CHECK_LAUNCH("hilo_rows");
  return FP_OK;
}

int prefix_tokens_launch(const float* prefix, int n_prefix, int dim, float* tokens, int batch, int n_tok, hipStream_t st) {
  const long long total = (long long)batch * n_prefix * dim;
  if (total == 0) return FP_OK;
  hipLaunchKernelGGL(prefix_tokens_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, prefix, n_prefix, dim, tokens, batch, n_tok);
  FP_CHECK_LAUNCH("prefix_tokens");
  return FP_OK;
}

// ---------------------------------------------------------------- fp8 (OCP e4m3) quantisation of a GEMM operand
// out[i] = e4m3(clamp(in[i] * scale, +-448)), round to nearest even (v_cvt_pk_fp8_f32; gfx950 converts to the OCP format).
// 16 elements per thread: 16 output bytes, one 16-B store.
template <typename T>
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const T* __restrict__ in, unsigned char* __restrict__ out, long long n, float scale) {
  const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 16;
  if (i0 >= n) return;
  float v[16];
  if (i0 + 16 <= n) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 x = *reinterpret_cast<const float4*>(in + i0 + 4 * q);
        v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const uint4 x = *reinterpret_cast<const uint4*>(in + i0 + 8 * q);
        const unsigned w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[8 * q + 2 * j] = __uint_as_float(w[j] << 16);
          v[8 * q + 2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
        }
      }
    }
  } else {
    for (int j = 0; j < 16; ++j) v[j] = i0 + j < n ? (float)in[i0 + j] : 0.f;
  }
  unsigned o[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = fminf(fmaxf(v[4 * q + j] * scale, -448.f), 448.f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], w, true);
    o[q] = (unsigned)w;
  }
  if (i0 + 16 <= n) *reinterpret_cast<uint4*>(out + i0) = make_uint4(o[0], o[1], o[2], o[3]);
  else for (int j = 0; j < 16 && i0 + j < n; ++j) out[i0 + j] = (unsigned char)(o[j >> 2] >> (8 * (j & 3)));
}

int quantize_fp8_launch(const void* in, int in_dtype, long long n, float scale, void* out, hipStream_t st) {
  if (n == 0) return FP_OK;
  const unsigned grid = (unsigned)((n + 4095) / 4096);
  if (in_dtype == FP_DTYPE_BF16)
    hipLaunchKernelGGL(quantize_fp8_kernel<__bf16>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const __bf16*>(in), reinterpret_cast<unsigned char*>(out), n, scale);
  else
    hipLaunchKernelGGL(quantize_fp8_kernel<float>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const float*>(in), reinterpret_cast<unsigned char*>(out), n, scale);
  FP_CHECK_LAUNCH("quantize_fp8");
  return FP_OK;
}

int convert_f32_to_bf16_launch(const float* in, void* out, long long n, hipStream_t st) {
  if (n == 0) return FP_OK;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n / 2 + 256) / 256)), dim3(256), 0, st, in, reinterpret_cast<__bf16*>(out), n);
  FP_CHECK_LAUNCH("f32_to_bf16");
  return FP_OK;
}
