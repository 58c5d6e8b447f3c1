"""Tensor-level wrappers over the C ABI: allocate outputs/scratch with torch, pass raw pointers.

torch is plumbing here (device memory + the current HIP stream); every computation is a
hand-written gfx950 kernel behind include/foundpose_amd.h.
"""

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import call, knn_scratch_bytes, ptr, stream, require_cuda


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def sqnorm_rows(x: torch.Tensor) -> torch.Tensor:
    require_cuda(x)
    x = _f32c(x)
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    call("fp_sqnorm_rows", ptr(x), x.shape[0], x.shape[1], ptr(out), stream())
    return out


def normalize_rows(x: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    require_cuda(x)
    x = _f32c(x)
    out = torch.empty_like(x)
    call("fp_normalize_rows", ptr(x), x.shape[0], x.shape[1], eps, ptr(out), stream())
    return out


_KNN_SCRATCH_BYTES = 1 << 29  # rows are processed in chunks so the [m, n] distance scratch stays <= 512 MiB


def knn_l2(q: torch.Tensor, db: torch.Tensor, k: int, q_sqnorm: Optional[torch.Tensor] = None,
           db_sqnorm: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (squared distances [m,k] f32, indices [m,k] int32), ascending, ties -> lowest index."""
    require_cuda(q, db)
    q, db = _f32c(q), _f32c(db)
    m, d = q.shape
    n = db.shape[0]
    if db.shape[1] != d:
        raise ValueError(f"dimension mismatch: queries {d} vs database {db.shape[1]}")
    if n == 0:
        raise ValueError("empty database")
    k_eff = min(k, n)
    qn = sqnorm_rows(q) if q_sqnorm is None else q_sqnorm
    dn = sqnorm_rows(db) if db_sqnorm is None else db_sqnorm
    d2 = torch.empty(m, k_eff, dtype=torch.float32, device=q.device)
    idx = torch.empty(m, k_eff, dtype=torch.int32, device=q.device)
    if m > 0:
        per_row = knn_scratch_bytes(1, n, k_eff)
        chunk = max(1, min(m, _KNN_SCRATCH_BYTES // per_row))
        scratch = torch.empty(chunk * per_row, dtype=torch.uint8, device=q.device)
        for r0 in range(0, m, chunk):
            r1 = min(m, r0 + chunk)
            call("fp_knn_l2", ptr(q[r0:r1]), ptr(qn[r0:r1]), r1 - r0, ptr(db), ptr(dn), n, d, k_eff,
                 ptr(scratch), ptr(d2[r0:r1]), ptr(idx[r0:r1]), stream())
    if k_eff < k:  # faiss pads missing neighbours with (inf, -1)
        pad_d = torch.full((m, k - k_eff), float("inf"), dtype=torch.float32, device=q.device)
        pad_i = torch.full((m, k - k_eff), -1, dtype=torch.int32, device=q.device)
        d2, idx = torch.cat([d2, pad_d], 1), torch.cat([idx, pad_i], 1)
    return d2, idx


def tfidf_build(word_ids: torch.Tensor, word_d2: torch.Tensor, seg_off: torch.Tensor, idf: torch.Tensor,
                soft_assign: bool, soft_sigma_squared: float, sqrt_dists: bool, eps: float = 1e-8, out=None):
    """-> (desc [S, W], desc_n [S, W]) for S = len(seg_off) - 1 point sets (written into `out` = (desc, desc_n) if given:
    contiguous row slices of larger buffers)."""
    require_cuda(word_ids, word_d2, seg_off, idf)
    num_segs = seg_off.shape[0] - 1
    W = idf.shape[0]
    if out is not None:
        desc, desc_n = out
        if desc.shape != (num_segs, W) or desc_n.shape != (num_segs, W) or not desc.is_contiguous() or not desc_n.is_contiguous():
            raise ValueError("tfidf_build: `out` must be two contiguous [num_segs, num_words] fp32 tensors")
    else:
        desc = torch.empty(num_segs, W, dtype=torch.float32, device=idf.device)
        desc_n = torch.empty_like(desc)
    call("fp_tfidf_build", ptr(word_ids), ptr(word_d2), word_ids.shape[1], ptr(seg_off), num_segs, ptr(idf), W,
         int(soft_assign), float(soft_sigma_squared), int(sqrt_dists), ptr(desc), ptr(desc_n), eps, stream())
    return desc, desc_n


def sample_bilinear(fmap_bchw: torch.Tensor, points: torch.Tensor, point_img: Optional[torch.Tensor],
                    image_size: Tuple[int, int]) -> torch.Tensor:
    """fmap_bchw: [B,C,H,W] fp32 with arbitrary strides (no copy); points [P,2] in image coords."""
    require_cuda(fmap_bchw, points)
    if fmap_bchw.dtype != torch.float32:
        fmap_bchw = fmap_bchw.float()
    points = _f32c(points)
    B, Cc, H, W = fmap_bchw.shape
    sb, sc, sh, sw = fmap_bchw.stride()
    out = torch.empty(points.shape[0], Cc, dtype=torch.float32, device=points.device)
    call("fp_sample_bilinear", ptr(fmap_bchw), sb, sc, sh, sw, Cc, H, W, int(image_size[0]), int(image_size[1]),
         ptr(points), ptr(point_img), points.shape[0], ptr(out), stream())
    return out


def pca_project(x: torch.Tensor, components: torch.Tensor, mean_proj: Optional[torch.Tensor]) -> torch.Tensor:
    require_cuda(x, components)
    x, components = _f32c(x), _f32c(components)
    out = torch.empty(x.shape[0], components.shape[0], dtype=torch.float32, device=x.device)
    call("fp_pca_project", ptr(x), x.shape[0], x.shape[1], ptr(components), components.shape[0], ptr(mean_proj),
         ptr(out), stream())
    return out


def gemm_f32(a: torch.Tensor, w: torch.Tensor, bias=None, gamma=None, out=None, epilogue: int = 0) -> torch.Tensor:
    require_cuda(a, w)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    call("fp_gemm_f32", ptr(a), a.stride(0), ptr(w), w.stride(0), M, N, K, ptr(bias), ptr(gamma), ptr(out),
         out.stride(0), epilogue, stream())
    return out


def _h16_bit(a: torch.Tensor, w: torch.Tensor) -> int:
    """FP_GEMM_F16 when the operands are IEEE fp16 (the "f16" mode's kernels), 0 for bf16; both operands must agree."""
    if a.dtype != w.dtype or a.dtype not in (torch.bfloat16, torch.float16):
        raise ValueError("the 16-bit GEMMs take two bf16 or two fp16 operands")
    return _lib.GEMM_F16 if a.dtype == torch.float16 else 0


def gemm_bf16(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, gamma=None, out=None, epilogue: int = 0,
              m_valid: Optional[int] = None) -> torch.Tensor:
    """bf16 operands, or fp16 ones (the "f16" mode: same kernels on v_mfma_f32_32x32x16_f16; 16-bit outputs are then fp16)."""
    require_cuda(a, w, bias)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        dt = torch.float32 if (epilogue & 0xff) in (3, 5) else a.dtype
        out = torch.zeros(M, N // 2 if (epilogue & 0xff) == 6 else N, dtype=dt, device=a.device)
    call("fp_gemm_bf16", ptr(a), a.stride(0), ptr(w), w.stride(0), M, N, K, M if m_valid is None else m_valid,
         ptr(bias), ptr(gamma), ptr(out), out.stride(0), epilogue | _h16_bit(a, w), stream())
    return out


def gemm_bf16_resid_ln(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, x: torch.Tensor, emit: bool = True, tile: int = 0, m_valid: Optional[int] = None):
    """Producer of the folded-LayerNorm chain (epilogue 7): x (fp32, in place) += a @ w.T + bias; with emit also
    -> (xb = bf16(x) [M, N], stats [N / 128, M, 2] partial (sum, sum of squares) per 128-column group)."""
    require_cuda(a, w, bias, x)
    M, K = a.shape
    N = w.shape[0]
    xb = torch.zeros(M, N, dtype=a.dtype, device=a.device) if emit else None
    stats = torch.zeros(N // 128, M, 2, dtype=torch.float32, device=a.device) if emit else None
    call("fp_gemm_bf16_ln", ptr(a), a.stride(0), ptr(w), w.stride(0), M, N, K, M if m_valid is None else m_valid, ptr(bias), ptr(x), x.stride(0),
         7 | (tile << 8) | _h16_bit(a, w), None, None, ptr(xb), N, ptr(stats), stream())
    return xb, stats


def gemm_bf16_resid_hilo(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, xb: torch.Tensor, xl: torch.Tensor, tile: int = 0, m_valid: Optional[int] = None):
    """Producer of the folded-LayerNorm chain on the (hi, lo) residual stream (epilogue 8): x = xb + xl (bf16 [M, N] each, updated in place) += a @ w.T + bias,
    xb = bf16(x), xl = bf16(x - xb).  -> stats [N / 128, M, 2]."""
    require_cuda(a, w, bias, xb, xl)
    M, K = a.shape
    N = w.shape[0]
    if xb.dtype != a.dtype or xl.dtype != a.dtype or xb.stride(0) != xl.stride(0):
        raise ValueError("xb / xl are arrays of the operands' 16-bit type with one row stride")
    if xb.stride(0) % 8 or xb.data_ptr() % 16 or xl.data_ptr() % 16 or xb.stride(1) != 1 or xl.stride(1) != 1:
        raise ValueError("xb / xl: 16-byte aligned rows (row stride a multiple of 8 elements, unit column stride) -- the epilogue moves 8 bf16 per access")
    stats = torch.zeros(N // 128, M, 2, dtype=torch.float32, device=a.device)
    call("fp_gemm_bf16_ln", ptr(a), a.stride(0), ptr(w), w.stride(0), M, N, K, M if m_valid is None else m_valid, ptr(bias), ptr(xl), xl.stride(0),
         8 | (tile << 8) | _h16_bit(a, w), None, None, ptr(xb), xb.stride(0), ptr(stats), stream())
    return stats


def ln_finalize(stats: torch.Tensor, dim: int, eps: float = 1e-6) -> torch.Tensor:
    """stats [parts, M, 2] -> ln_row [M, 2] = (rstd, mean * rstd)."""
    require_cuda(stats)
    parts, M = stats.shape[0], stats.shape[1]
    out = torch.empty(M, 2, dtype=torch.float32, device=stats.device)
    call("fp_ln_finalize", ptr(stats), parts, M, M, dim, eps, ptr(out), stream())
    return out


def gemm_bf16_ln(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, colsum: torch.Tensor, ln_row: torch.Tensor, epilogue: int = 0, out=None,
                 tile: int = 0, m_valid: Optional[int] = None) -> torch.Tensor:
    """Consumer of the folded-LayerNorm chain: out(bf16) = epi(rstd * (a @ w.T) - mean * rstd * colsum + bias), epilogue 0 / 1 (GELU) / 6 (SwiGLU)."""
    require_cuda(a, w, bias, colsum, ln_row)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.zeros(M, N // 2 if epilogue == 6 else N, dtype=a.dtype, device=a.device)
    call("fp_gemm_bf16_ln", ptr(a), a.stride(0), ptr(w), w.stride(0), M, N, K, M if m_valid is None else m_valid, ptr(bias), ptr(out), out.stride(0),
         epilogue | (tile << 8) | _h16_bit(a, w), ptr(colsum), ptr(ln_row), None, 0, None, stream())
    return out


def quantize_fp8(x: torch.Tensor, scale: float) -> torch.Tensor:
    """e4m3(clamp(x * scale, +-448)) -> torch.float8_e4m3fn tensor of x's shape (x fp32 or bf16, contiguous)."""
    require_cuda(x)
    if x.dtype not in (torch.float32, torch.bfloat16) or not x.is_contiguous():
        raise ValueError("quantize_fp8 takes a contiguous fp32 or bf16 tensor")
    out = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device=x.device)
    call("fp_quantize_fp8", ptr(x), _lib.FP_BF16 if x.dtype == torch.bfloat16 else _lib.FP_F32, x.numel(), float(scale), ptr(out), stream())
    return out


def gemm_fp8(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, col_scale: torch.Tensor, out=None, epilogue: int = 0,
             m_valid: Optional[int] = None, out_scale: float = 0.0) -> torch.Tensor:
    """fp8 x fp8 -> fp32-accumulated GEMM: out = epilogue((a @ w.T + bias) * col_scale), a [M, K] and w [N, K] in
    torch.float8_e4m3fn; bias is the true bias divided by col_scale (see include/foundpose_amd.h)."""
    require_cuda(a, w, bias, col_scale)
    if a.dtype != torch.float8_e4m3fn or w.dtype != torch.float8_e4m3fn:
        raise ValueError("gemm_fp8 operands must be torch.float8_e4m3fn")
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        epi = epilogue & 0xff   # (bits 8+: the block-tile override of benchmarks and tests)
        odt = torch.float8_e4m3fn if out_scale > 0 else (torch.float32 if epi == 3 else torch.bfloat16)
        out = torch.zeros(M, N // 2 if epi == 6 else N, dtype=odt, device=a.device)  # out_scale > 0: e4m3(result * out_scale)
    call("fp_gemm_fp8", ptr(a), a.stride(0), ptr(w), w.stride(0), M, N, K, M if m_valid is None else m_valid,
         ptr(bias), ptr(col_scale), ptr(out), out.stride(0), epilogue, float(out_scale), stream())
    return out


def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, out_dtype: torch.dtype, eps: float = 1e-6):
    require_cuda(x, weight, bias)
    rows, D = x.shape
    out = torch.empty(rows, D, dtype=out_dtype, device=x.device)
    call("fp_layernorm", ptr(x), x.stride(0), ptr(weight), ptr(bias), eps, ptr(out), D,
         {torch.bfloat16: _lib.FP_BF16, torch.float16: _lib.FP_F16, torch.float32: _lib.FP_F32}[out_dtype], D, rows, rows, rows, 0, stream())
    return out


def attention(qkv: torch.Tensor, batch: int, n_tok: int, dim: int, heads: int, variant: int = 0):
    """variant: bf16 work split (FP_ATTN_VARIANT in the header; all bit-identical) -- 0 is what the pipeline runs."""
    require_cuda(qkv)
    dt = {torch.bfloat16: _lib.FP_BF16, torch.float16: _lib.FP_F16, torch.float32: _lib.FP_F32}[qkv.dtype]   # fp16: the "f16" mode's kernel (variant 0 only)
    out = torch.zeros(qkv.shape[0], dim, dtype=qkv.dtype, device=qkv.device)
    call("fp_attention", ptr(qkv), qkv.stride(0), ptr(out), dim,
         batch, n_tok, dim, heads, dt | (int(variant) << 8), stream())
    return out


# ---------------------------------------------------------------- split-fp16 rows (the f16x3 near-exact mode, include/foundpose_amd.h)
def pow2_scale(t: torch.Tensor, target: float = 16384.0) -> float:
    """Largest power of two s with max|t| * s <= target (16384 leaves a factor 4 of head room below the fp16 maximum)."""
    import math
    amax = float(t.abs().max())
    if not math.isfinite(amax) or amax <= 0.0:
        return 1.0
    return float(2.0 ** math.floor(math.log2(target / amax)))


def split16_pack(x: torch.Tensor, scale: float = 1.0, pad: int = 0) -> torch.Tensor:
    """fp32 [rows, K] (K % 32 == 0) -> fp16 [rows, 2K] split rows (groups of 32: [hi 32 | lo 32]) of scale * x; `pad` extra halves of row
    stride (the returned tensor is then a view of a wider buffer).  Host-side preparation of weights and test operands; inside the
    pipeline the producing kernels write this layout themselves."""
    rows, K = x.shape
    if K % 32:
        raise ValueError("split16_pack: the row length must be a multiple of 32")
    xs = (x.float() * scale).clamp(-65504.0, 65504.0)
    hi = xs.half()
    lo = (xs - hi.float()).half()
    out = torch.stack([hi.reshape(rows, K // 32, 32), lo.reshape(rows, K // 32, 32)], 2).reshape(rows, 2 * K)
    if pad == 0:
        return out.contiguous()
    buf = torch.zeros(rows, 2 * K + pad, dtype=torch.float16, device=x.device)
    buf[:, :2 * K] = out
    return buf[:, :2 * K]


def splitx_pack(x: torch.Tensor, scale: float = 1.0, pad: int = 0) -> torch.Tensor:
    """fp32 [rows, K] (K % 64 == 0) -> fp16-typed [rows, 2K] f16f8 rows of scale * x (include/foundpose_amd.h "f16f8 rows"): per 64 columns the
    fp16 high halves (128 B), e4m3(hi 2^-7) (64 B) and e4m3((s x - hi) 2^4) (64 B).  Host-side preparation of weights and test operands, the
    same arithmetic as the device's splitx_pack2 (common.hpp)."""
    rows, K = x.shape
    if K % 64:
        raise ValueError("splitx_pack: the row length must be a multiple of 64")
    xs = (x.float() * scale).clamp(-65504.0, 65504.0)
    hi = xs.half()
    hi8 = (hi.float() * 2.0 ** -7).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
    lo8 = ((xs - hi.float()) * 2.0 ** 4).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
    buf = torch.zeros(rows, 2 * K + pad, dtype=torch.float16, device=x.device)
    by = buf.view(torch.uint8)[:, :4 * K].unflatten(1, (K // 64, 256))
    by[:, :, :128] = hi.contiguous().view(torch.uint8).unflatten(1, (K // 64, 128))
    by[:, :, 128:192] = hi8.unflatten(1, (K // 64, 64))
    by[:, :, 192:256] = lo8.unflatten(1, (K // 64, 64))
    return buf[:, :2 * K] if pad else buf


def splitx_unpack(xs: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """What an f16f8 row represents in the products it enters: fp16 [rows, 2K] -> fp32 [rows, K] = (hi + e4m3-lo 2^-4) / scale (summed in fp64)."""
    rows, K2 = xs.shape
    by = xs.contiguous().view(torch.uint8).unflatten(1, (K2 // 128, 256))
    hi = by[:, :, :128].contiguous().view(torch.float16).double()
    lo = by[:, :, 192:256].contiguous().view(torch.float8_e4m3fn).double() * 2.0 ** -4
    return ((hi + lo) / scale).reshape(rows, K2 // 2).float()


def split16_unpack(xs: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """Inverse of split16_pack (up to its 2^-22 rounding): fp16 [rows, 2K] -> fp32 [rows, K] = (hi + lo) / scale (summed in fp64)."""
    rows, K2 = xs.shape
    g = xs.reshape(rows, K2 // 64, 2, 32).double()
    return ((g[:, :, 0] + g[:, :, 1]) / scale).reshape(rows, K2 // 2).float()


def gemm_split(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, acc_scale: float, gamma=None, out=None, epilogue: int = 0,
               out_scale: float = 1.0, m_valid: Optional[int] = None, tile: int = 0, f16f8: bool = False) -> torch.Tensor:
    """a [M, 2K], w [N, 2K] split rows (fp16); -> epilogue 0 / 1 / 6: split rows [M, 2N] (6: [M, N]) scaled by out_scale; 3 / 5: fp32 [M, N].
    f16f8: a and w (and the output of epilogues 1 / 6) are f16f8 rows (splitx_pack); the output of epilogue 0 stays a split-fp16 row."""
    require_cuda(a, w, bias)
    if a.dtype != torch.float16 or w.dtype != torch.float16:
        raise ValueError("gemm_split operands are fp16 split rows")
    M, K = a.shape[0], a.shape[1] // 2
    N = w.shape[0]
    if out is None:
        if epilogue in (3, 5):
            out = torch.zeros(M, N, dtype=torch.float32, device=a.device)
        else:
            out = torch.zeros(M, N if epilogue == 6 else 2 * N, dtype=torch.float16, device=a.device)
    call("fp_gemm_split", ptr(a), a.stride(0), ptr(w), w.stride(0), M, N, K, M if m_valid is None else m_valid, ptr(bias), ptr(gamma),
         ptr(out), out.stride(0), epilogue | (tile << 8) | (_lib.GEMM_SPLIT_F16F8 if f16f8 else 0), float(acc_scale), float(out_scale), stream())
    return out


def attention_split(qkv: torch.Tensor, batch: int, n_tok: int, dim: int, heads: int, in_scale: float, out_scale: float, f16f8_out: bool = False,
                    variant: int = 0) -> torch.Tensor:
    """qkv [B*N, 6D] split rows (q | k | v) -> [B*N, 2D] split rows (f16f8_out: f16f8 rows).  variant 0 = the kernel the pipeline runs, 1 = the lock-step
    kernel, 2 = the role-split kernel (bit-identical)."""
    require_cuda(qkv)
    out = torch.zeros(qkv.shape[0], 2 * dim, dtype=torch.float16, device=qkv.device)
    call("fp_attention_split", ptr(qkv), qkv.stride(0), ptr(out), 2 * dim, batch, n_tok, dim, heads, float(in_scale), float(out_scale),
         (_lib.FP_F16F8 if f16f8_out else _lib.FP_F16X3) | (int(variant) << 8), stream())
    return out


def layernorm_split(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, out_scale: float, eps: float = 1e-6, f16f8: bool = False) -> torch.Tensor:
    require_cuda(x, weight, bias)
    rows, D = x.shape
    out = torch.empty(rows, 2 * D, dtype=torch.float16, device=x.device)
    call("fp_layernorm_scaled", ptr(x), x.stride(0), ptr(weight), ptr(bias), eps, ptr(out), 2 * D, _lib.FP_F16F8 if f16f8 else _lib.FP_F16X3, float(out_scale), D, rows, stream())
    return out
