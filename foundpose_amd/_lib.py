"""ctypes binding of libfoundpose_amd.so (the C ABI in include/foundpose_amd.h).

There is NO fallback: if the library is missing or a call fails, this raises. The product
path never computes on the CPU and never touches oracle/.
"""

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# FOUNDPOSE_AMD_LIB: another build of the SAME library (a measurement variant from tools/build_variant.sh, for same-box A/B runs); it must
# export the same ABI version.  Never a different implementation: there is one product path.
LIB_PATH = os.environ.get("FOUNDPOSE_AMD_LIB") or os.path.join(_HERE, "lib", "libfoundpose_amd.so")

FP_F32, FP_BF16, FP_FP8, FP_F16X3, FP_F16F8, FP_F16 = 0, 1, 2, 3, 4, 5
GEMM_SPLIT_F16F8 = 1 << 20   # FP_GEMM_SPLIT_F16F8 of the header
VIT_NO_TALL_TILES = 1        # FP_VIT_NO_TALL_TILES of the header (fp_vit_model.flags)
GEMM_F16 = 1 << 21           # FP_GEMM_F16 of the header: IEEE fp16 operands / outputs in fp_gemm_bf16, fp_gemm_bf16_ln
SPLIT_SCALE_ACT, SPLIT_SCALE_QKV, SPLIT_SCALE_HID = 16.0, 16.0, 4.0  # FP_SPLIT_SCALE_* of the header
ABI_VERSION = 18

vp, i32, i64, f32, f64, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_uint64


class VitBlock(C.Structure):
    _fields_ = [(n, vp) for n in (
        "ln1_w", "ln1_b", "ln2_w", "ln2_b", "ls1", "ls2",
        "qkv_w", "proj_w", "fc1_w", "fc2_w",
        "qkv_b", "proj_b", "fc1_b", "fc2_b", "qkv_s", "proj_s", "fc1_s", "fc2_s")] + [("act_scale", f32 * 4), ("qkv_colsum", vp), ("fc1_colsum", vp)]


class VitModel(C.Structure):
    _fields_ = [
        ("dim", i32), ("depth", i32), ("heads", i32), ("hidden", i32), ("registers", i32), ("patch", i32),
        ("ffn_swiglu", i32), ("weight_dtype", i32),
        ("patch_w", vp), ("patch_k_pad", i32), ("patch_b", vp), ("pos_patch", vp), ("prefix", vp),
        ("norm_w", vp), ("norm_b", vp), ("blocks", C.POINTER(VitBlock)), ("ld_w_dim", i32), ("ld_w_hidden", i32), ("patch_stride", i32), ("patch_acc_scale", f32), ("ln_fold", i32), ("flags", i32),
    ]


class VitWorkspace(C.Structure):
    _fields_ = [
        ("patches", vp), ("x", vp), ("y", vp), ("qkv", vp), ("h", vp), ("a8", vp),
        ("ld_y", i32), ("ld_h", i32), ("ld_qkv", i32), ("m_pad", i32), ("m_patch_pad", i32), ("xb", vp), ("stats", vp), ("sat", vp), ("xl", vp),
    ]


_PROTOS = {
    "fp_abi_version": [],
    "fp_build_experiments": [],
    "fp_sqnorm_rows": [vp, i64, i32, vp, vp],
    "fp_normalize_rows": [vp, i64, i32, f32, vp, vp],
    "fp_knn_l2": [vp, vp, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp],
    "fp_tfidf_build": [vp, vp, i32, vp, i32, vp, i32, i32, f32, i32, vp, vp, f32, vp],
    "fp_cosine_topk": [vp, vp, vp, i32, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, vp],
    "fp_cosine_topk_prefiltered": [vp, vp, vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, vp],
    "fp_cyclic_buddies": [vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32,
                          vp, vp, vp, vp, vp, vp, vp, vp, i32, vp],
    "fp_pack_records": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp],
    "fp_pnp_ransac": [vp, vp, vp, vp, i32, i32, i32, i32, f64, f64, i32, i32, u64, vp, vp, vp, vp, vp, vp, vp],
    "fp_sample_bilinear": [vp, i64, i64, i64, i64, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp],
    "fp_pca_project": [vp, i32, i32, vp, i32, vp, vp, vp],
    "fp_vit_forward": [C.POINTER(VitModel), C.POINTER(VitWorkspace), vp, i32, i32, i32, i32, vp],
    "fp_vit_sample_features": [C.POINTER(VitModel), C.POINTER(VitWorkspace), i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp],
    "fp_query_select": [vp, i32, i32, i32, vp, vp, vp, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "fp_vit_forward_prefix": [C.POINTER(VitModel), C.POINTER(VitWorkspace), vp, i32, i32, i32, i32, vp],
    "fp_vit_stream_f32": [C.POINTER(VitModel), C.POINTER(VitWorkspace), i32, i32, i32, i32, vp, vp],
    "fp_vit_forward_blocks": [C.POINTER(VitModel), C.POINTER(VitWorkspace), i32, i32, i32, i32, i32, i32, vp],
    "fp_vit_block_selected": [C.POINTER(VitModel), C.POINTER(VitWorkspace), i32, i32, i32, i32, vp, vp, i32, i32, vp],
    "fp_vit_sample_features_selected": [C.POINTER(VitModel), C.POINTER(VitWorkspace), i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp],
    "fp_vit_features": [C.POINTER(VitModel), C.POINTER(VitWorkspace), i32, i32, i32, vp, vp, vp],
    "fp_patchify": [vp, i32, i32, i32, i32, vp, i32, i32, vp],
    "fp_layernorm": [vp, i32, vp, vp, f32, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "fp_gemm_bf16": [vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp, i32, i32, vp],
    "fp_gemm_bf16_ln": [vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp],
    "fp_ln_finalize": [vp, i32, i32, i32, i32, f32, vp, vp],
    "fp_gemm_fp8": [vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp, i32, i32, f32, vp],
    "fp_quantize_fp8": [vp, i32, i64, f32, vp, vp],
    "fp_gemm_split": [vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp, i32, i32, f32, f32, vp],
    "fp_attention_split": [vp, i32, vp, i32, i32, i32, i32, i32, f32, f32, i32, vp],
    "fp_layernorm_scaled": [vp, i32, vp, vp, f32, vp, i32, i32, f32, i32, i32, vp],
    "fp_gemm_f32": [vp, i32, vp, i32, i32, i32, i32, vp, vp, vp, i32, i32, vp],
    "fp_attention": [vp, i32, vp, i32, i32, i32, i32, i32, i32, vp],
    "fp_convert_f32_to_bf16": [vp, vp, i64, vp],
    "fp_warp_crops": [vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp],
}

_lib = None


class FoundPoseNativeError(RuntimeError):
    pass


class FoundPoseSaturationError(FoundPoseNativeError):
    """The f16x3 (near-exact) mode clamped an activation to the range of its split-fp16 row (+-4094 for LayerNorm outputs / q / k / v,
    +-16376 for hidden activations; include/foundpose_amd.h): the features are then no longer the fp32 arithmetic's, and the mode whose
    purpose is index-exact matching must not return a plausible wrong neighbour silently."""


def lib() -> C.CDLL:
    """Loads the HIP library (after torch, so both share one libamdhip64 runtime)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FoundPoseNativeError(
                f"{LIB_PATH} is missing: build it with `python -m foundpose_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        handle.fp_last_error.restype = C.c_char_p
        handle.fp_last_error.argtypes = []
        for name, args in _PROTOS.items():
            fn = getattr(handle, name)  # AttributeError here = header/library mismatch
            fn.argtypes = args
            fn.restype = i32
        if handle.fp_abi_version() != ABI_VERSION:
            raise FoundPoseNativeError("libfoundpose_amd.so ABI version mismatch; rebuild")
        _lib = handle
    return _lib


def cosine_scratch_floats(num_det: int, max_templates: int) -> int:
    """FP_COSINE_SCRATCH_FLOATS of include/foundpose_amd.h."""
    return 2 * num_det * max_templates + 17 * num_det + 2


def cosine_prefilter_scratch_floats(num_det: int, max_templates: int) -> int:
    """FP_COSINE_PREFILTER_SCRATCH_FLOATS of include/foundpose_amd.h."""
    return cosine_scratch_floats(num_det, max_templates) + 3 * num_det * max_templates + 32 * num_det + 16


def cyclic_scratch_bytes(pairs: int, q_max: int, p_max: int) -> int:
    """FP_CYCLIC_SCRATCH_BYTES of include/foundpose_amd.h."""
    tiles = 8 * pairs * (((p_max + 127) // 128) * q_max + ((q_max + 127) // 128) * p_max)
    cand = 8 * pairs * (q_max + p_max) + 448 * pairs * max(q_max, p_max)
    return max(tiles, cand)


def knn_scratch_bytes(m: int, n: int, k: int) -> int:
    """FP_KNN_SCRATCH_BYTES of include/foundpose_amd.h."""
    if k == 1:
        return m * 8
    if k <= 8:
        return m * max(((n + 127) // 128) * k * 8, 704)
    return m * n * 4


def exported_symbols():
    return sorted(list(_PROTOS) + ["fp_last_error"])


def call(name: str, *args) -> None:
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise FoundPoseNativeError(f"{name} failed (code {rc}): {lib().fp_last_error().decode()}")


def ptr(t) -> vp:
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return vp(0)
    return vp(t.data_ptr())


def stream() -> vp:
    return vp(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise FoundPoseNativeError(
                "foundpose_amd runs on the MI355X only: got a CPU tensor (move inputs to 'cuda'; no CPU fallback exists)")


def upload_async(host: torch.Tensor, device) -> torch.Tensor:
    """A small host table -> device through pinned staging, asynchronously on the current stream.  A pageable upload (torch.tensor(..., device=),
    .to(device) of a pageable tensor) blocks the host until everything queued on the stream before it has drained -- a whole batch of the backbone --
    and the launches behind it then reach an idle GPU one launch latency at a time; the caching host allocator keeps the pinned block alive
    until the copy has run."""
    host = host.contiguous()
    staged = torch.empty(host.shape, dtype=host.dtype, pin_memory=True)
    staged.copy_(host)
    return staged.to(device, non_blocking=True)
