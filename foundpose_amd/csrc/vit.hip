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
template <int VEC>
__global__ __launch_bounds__(256) void layernorm_kernel(LayerNormArgs a) {
  constexpr int MAXI = 2048 / (64 * VEC);
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
                pack_fp8x4(y[0] * a.out_scale, y[1] * a.out_scale, y[2] * a.out_scale, y[3] * a.out_scale);
        } else if (a.out_dtype == FP_DTYPE_BF16) {
          __bf16* o = reinterpret_cast<__bf16*>(a.out) + (size_t)row * a.ld_out + c;
          if constexpr (VEC == 4) *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]));
          else *reinterpret_cast<unsigned*>(o) = pack_bf16x2(y[0], y[1]);
        } else {
          *reinterpret_cast<vec_t*>(reinterpret_cast<float*>(a.out) + (size_t)row * a.ld_out + c) = y;
        }
      }
  }
}

// images [B,3,H,W] in [0,1] -> rows of the patch-embed GEMM: row = b*Np + gy*gw + gx,
// column = c*P*P + py*P + px (the flattening of the conv weight [D,3,P,P]); columns >= 3*P*P are zero.
// One thread moves one P-pixel run of a patch (a row of the patch in one channel): P contiguous floats in, P contiguous
// outputs, so the div/mod address arithmetic is paid once per run instead of once per element (120 -> ~40 us at 32 x 518^2).
template <typename T>
__global__ void patchify_kernel(const float* __restrict__ img, int B, int H, int W, int P, T* __restrict__ out, int ld) {
  const int nseg = 3 * P + 1;  // 3*P pixel runs + one run of zero padding per output row
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int gw = W / P, gh = H / P, np = gw * gh;
  if (e >= (long long)B * np * nseg) return;
  const int seg = (int)(e % nseg);
  const long long row = e / nseg;
  T* orow = out + row * ld;
  if (seg == 3 * P) {
    for (int col = 3 * P * P; col < ld; ++col) orow[col] = (T)0.f;
    return;
  }
  const int b = (int)(row / np), pi = (int)(row % np);
  const int gy = pi / gw, gx = pi - gy * gw;
  const int c = seg / P, py = seg - c * P;
  const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f);
  const float stdv = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
  const float* src = img + (((size_t)b * 3 + c) * H + gy * P + py) * W + gx * P;
  T* dst = orow + c * P * P + py * P;
  if constexpr (sizeof(T) == 2) {
    if ((P & 1) == 0 && (W & 1) == 0 && (ld & 1) == 0) {  // float2 in, packed bf16 pairs out (alignment: P, W, ld even)
      for (int px = 0; px < P; px += 2) {
        const float2 x = *reinterpret_cast<const float2*>(src + px);
        *reinterpret_cast<unsigned*>(dst + px) = pack_bf16x2((x.x - mean) / stdv, (x.y - mean) / stdv);  // T.Normalize: sub then div
      }
      return;
    }
  }
  for (int px = 0; px < P; ++px) dst[px] = (T)((src[px] - mean) / stdv);
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
  const int grid = wgs < 2048 ? wgs : 2048;  // 8 workgroups (32 waves) per CU, each wave walks out_rows / 8192 rows
  FP_REQUIRE(a.out_dtype != FP_DTYPE_FP8 || (a.dim % 256 == 0 && a.ld_x % 4 == 0 && a.ld_out % 4 == 0 && a.out_scale > 0.f),
             "layernorm: fp8 output needs dim %% 256 == 0 and a positive scale");
  if (a.dim % 256 == 0 && a.ld_x % 4 == 0 && a.ld_out % 4 == 0)
    hipLaunchKernelGGL(layernorm_kernel<4>, dim3(grid), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(layernorm_kernel<2>, dim3(grid), dim3(256), 0, st, a);
  FP_CHECK_LAUNCH("layernorm");
  return FP_OK;
}

int patchify_launch(const float* images, int batch, int height, int width, int patch, void* out, int ld_out,
                    int out_dtype, hipStream_t st) {
  FP_REQUIRE(height % patch == 0 && width % patch == 0, "patchify: image %dx%d is not a multiple of the patch size %d", height, width, patch);
  FP_REQUIRE(ld_out >= 3 * patch * patch, "patchify: ld_out too small");
  const long long total = (long long)batch * (height / patch) * (width / patch) * (3 * patch + 1);  // one thread per pixel run
  if (total == 0) return FP_OK;
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (out_dtype == FP_DTYPE_BF16)
    hipLaunchKernelGGL(patchify_kernel<__bf16>, dim3(grid), dim3(256), 0, st, images, batch, height, width, patch, reinterpret_cast<__bf16*>(out), ld_out);
  else
    hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid), dim3(256), 0, st, images, batch, height, width, patch, reinterpret_cast<float*>(out), ld_out);
  FP_CHECK_LAUNCH("patchify");
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
