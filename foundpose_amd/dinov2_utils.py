"""DINOv2 patch-feature extractor with the reference's interface
(/root/reference/utils/dinov2_utils.py:25-158), executed by hand-written gfx950 kernels.

Differences that do not change results:
  * blocks after `layer` are not executed (the reference runs the whole backbone and keeps only the hooked
    block's output, dinov2_utils.py:257) and no autograd graph is built;
  * weights: the reference downloads the pretrained hub checkpoint (`pretrained=True`, dinov2_utils.py:82);
    there is no network here, so the checkpoint is read from disk (`weights=`, $FOUNDPOSE_DINOV2_WEIGHTS, the torch hub cache) or
    passed as `state_dict=`; without one the constructor RAISES (weights.py) -- random weights only with `random_init_seed=`.
Not implemented: the "attn" facet (the reference's extract_descriptors asserts it away, dinov2_utils.py:285-290).  stride != 14 follows
what the reference's patch_vit_resolution / _fix_pos_enc state (its own branch cannot run: see __init__).
"""

import ctypes as C
import os
import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from . import _lib, weights as _weights
from ._lib import call, ptr, stream
from .vit_config import ARCHS, ExtractorSpec, VitArch, parse_extractor_name


def _interpolate_pos_embed(pos_embed: torch.Tensor, arch: VitArch, gh: int, gw: int) -> torch.Tensor:
    """Pos-embed table for a gh x gw patch grid (host-side weight preparation, once per resolution)."""
    n = pos_embed.shape[1] - 1
    m = int(math.sqrt(n))
    if gh * gw == n and gh == gw:
        return pos_embed
    dim = pos_embed.shape[-1]
    grid = pos_embed[:, 1:].reshape(1, m, m, dim).permute(0, 3, 1, 2)
    if arch.interp_offset:
        kw = {"scale_factor": (float(gh + arch.interp_offset) / m, float(gw + arch.interp_offset) / m)}
    else:
        kw = {"size": (gh, gw)}
    grid = F.interpolate(grid, mode="bicubic", antialias=arch.interp_antialias, **kw)
    if tuple(grid.shape[-2:]) != (gh, gw):
        raise RuntimeError("pos-embed interpolation produced an unexpected grid")
    return torch.cat([pos_embed[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, dim)], dim=1)


def _interpolate_pos_embed_strided(pos_embed: torch.Tensor, patch: int, stride: int, H: int, W: int) -> torch.Tensor:
    """Position encoding for stride != patch size, as the reference's _fix_pos_enc states it (dinov2_utils.py:325-360; upstream calls it with
    (tokens, image height, image width)): bicubic, scale-factor mode, +0.1 fudge, no antialias.  Host-side weight preparation."""
    n = pos_embed.shape[1] - 1
    m = int(math.sqrt(n))
    w0, h0 = 1 + (H - patch) // stride, 1 + (W - patch) // stride
    if w0 * h0 == n and H == W:
        return pos_embed
    dim = pos_embed.shape[-1]
    grid = F.interpolate(pos_embed[:, 1:].reshape(1, m, m, dim).permute(0, 3, 1, 2), scale_factor=((w0 + 0.1) / m, (h0 + 0.1) / m),
                         mode="bicubic", align_corners=False, recompute_scale_factor=False)
    if tuple(grid.shape[-2:]) != (w0, h0):
        raise RuntimeError("pos-embed interpolation produced an unexpected grid")
    return torch.cat([pos_embed[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, dim)], dim=1)


class DinoFeatureExtractor(torch.nn.Module):
    def __init__(self, model_name: str, state_dict: Optional[Dict[str, torch.Tensor]] = None, weights: Optional[str] = None,
                 random_init_seed: Optional[int] = None, precision: str = "bf16", arch: Optional[VitArch] = None, use_graph: bool = False,
                 act_scales: Optional[torch.Tensor] = None, fold_layernorm: bool = True, head_blocks: int = 0, head_precision: str = "f16",
                 resid_hilo: bool = True, ld_pad: int = 64, ld_pad_qkv: int = 0, ld_pad8: int = 0, tall_tiles: bool = True, sat_check: bool = True) -> None:
        """Weights (the reference: hub model with pretrained=True, dinov2_utils.py:81-84): `state_dict=` (upstream key names), `weights=` (checkpoint
        file, or directory holding the upstream file name), else $FOUNDPOSE_DINOV2_WEIGHTS, else the torch hub cache the reference's own call fills;
        none of them -> FoundPoseWeightsError.  Random weights only on an explicit `random_init_seed=` (tests, benchmarks).  Every dict is checked
        like load_state_dict(strict=True) (weights.validate_state_dict).
        Tuning arguments (A/B switches of measurements; none changes what is computed beyond rounding points, the defaults are what the benchmarks run):
        fold_layernorm (bf16: the block LayerNorms folded into the GEMMs), resid_hilo (the residual stream in front of the hooked block as a (hi, lo) 16-bit
        pair), ld_pad / ld_pad_qkv / ld_pad8 (row-stride padding of the operands, elements / bytes), tall_tiles (320-row GEMM tiles), sat_check (forward()
        raises on a saturation report: one host sync per call).  No environment variable is read."""
        super().__init__()
        self.use_graph = use_graph  # replay the forward's launch sequence as one hipGraph (static buffers per batch shape)
        if arch is not None:  # non-hub architecture (unit tests use a tiny one)
            ARCHS[arch.name] = arch
        spec: ExtractorSpec = parse_extractor_name(model_name)
        self.version, self.stride, self.facet = spec.version, spec.stride, spec.facet
        self.layer, self.apply_norm = spec.layer, spec.apply_norm
        self.arch: VitArch = arch if arch is not None else spec.arch
        self.model_base_name = f"dinov2_{self.version}".replace("-", "_")
        self.patch_size = self.arch.patch
        # stride != patch size: the reference's patch_vit_resolution (dinov2_utils.py:364-389) -- the patch-embedding conv runs with
        # the smaller stride (overlapping patches, 1 + (size - patch) // stride tokens per axis) and the position encoding comes from
        # _fix_pos_enc (scale-factor bicubic with the +0.1 fudge).  The reference's own branch cannot run (the function is declared
        # without `self`, closes over the wrapper's non-existent pos_embed and is bound with types.MethodType: its first forward raises);
        # what is implemented is what that code states, pinned by calling the reference's function directly (tests/golden/extractor_tiny_stride7.npz).
        if self.stride != self.patch_size and (self.stride < 1 or (self.patch_size // self.stride) * self.stride != self.patch_size):
            raise AssertionError(f"stride {self.stride} should divide patch_size {self.patch_size}")   # dinov2_utils.py:378-380
        if self.facet not in ("token", "key", "query", "value"):
            raise NotImplementedError(f"facet '{self.facet}' is not implemented on the MI355X path ('token', 'key', 'query', 'value' are)")
        if self.facet != "token" and use_graph:
            raise NotImplementedError("use_graph replays the token path only")
        if not 0 <= self.layer < self.arch.depth:
            raise ValueError(f"layer {self.layer} out of range for {self.version}")
        if precision not in ("bf16", "f16", "fp32", "fp8", "f16x3", "f16f8"):
            raise ValueError("precision must be 'bf16', 'f16', 'fp32', 'f16x3', 'f16f8' or 'fp8'")
        # "f16": the bf16 pipeline -- same kernels, tiles, bytes and speed -- on IEEE fp16 operands (11 significant bits instead of 8; include/foundpose_amd.h
        # "plain fp16 rows"): closer to the reference's fp32 arithmetic at no cost.  What it gives up is bf16's range: an activation beyond +-65504 is
        # reported (FoundPoseSaturationError), never silently wrong.
        if precision == "f16" and self.arch.dim % 128:
            raise NotImplementedError(f"precision='f16' runs the folded-LayerNorm pipeline: dim must be a multiple of 128 ({self.version}: {self.arch.dim})")
        # "f16x3": the near-exact mode.  The reference computes in fp32 (scripts/infer.py:468-473); the fp32-input MFMA runs at 1/16
        # of the fp16 rate, so this mode carries every GEMM / attention operand as a (hi, lo) pair of fp16 numbers (22 mantissa
        # bits) and builds each product from three fp16 MFMAs with fp32 accumulation (include/foundpose_amd.h "split-fp16 rows").
        # Residual stream, LayerNorm, softmax, GELU (exact erf) stay fp32: features at the fp32 path's own noise level, ~3.5x its speed.
        # "f16f8": the same split products with the two cross terms (hi lo, lo hi: ~2^-11 of a product) on the fp8 pipe -- rows carry the fp16 high
        # halves plus e4m3 copies of hi and lo (include/foundpose_amd.h "f16f8 rows"), 8 instead of 12 fp16-MFMA units per 64 k: GEMMs 1.3x
        # faster, a product good to ~14 bits at the worst (fc2: 1.3e-5 of the output scale against f16x3's 1.9e-6); q / k / v and the
        # attention's own products stay three-fp16-MFMA.  Same scales, same saturation report.
        if precision in ("f16x3", "f16f8") and (self.arch.dim % 128 or self.arch.hidden % 128):
            raise NotImplementedError(f"precision='{precision}' needs dim and hidden to be multiples of 128 ({self.version}: {self.arch.dim}, {self.arch.hidden})")
        if precision == "fp8" and (self.arch.dim % 256 or self.arch.hidden % 256):
            raise NotImplementedError(f"precision='fp8' needs dim and hidden to be multiples of 256 ({self.version}: {self.arch.dim}, {self.arch.hidden})")
        # fp8 mode (BASELINE config 5): e4m3 block matrices quantised per output channel, GEMM inputs quantised per tensor
        # with STATIC scales.  They are part of the model: set them with act_scales= (e.g. the scales stored with the bank,
        # repre.extractor_fp8_act_scales) or with an explicit calibrate_fp8(images) call.  A forward without them raises --
        # scales picked up from whichever batch happens to come first would make the features depend on call order and
        # would quantise bank and query descriptors inconsistently.
        self.act_scales: Optional[torch.Tensor] = None if act_scales is None else torch.as_tensor(act_scales, dtype=torch.float32).cpu().clone()  # [depth, 4]: inputs of qkv, proj, fc1, fc2
        if self.act_scales is not None and (precision != "fp8" or tuple(self.act_scales.shape) != (self.arch.depth, 4)):
            raise ValueError(f"act_scales is the [depth, 4] scale table of precision='fp8' (got {tuple(self.act_scales.shape)}, precision {precision})")
        self.precision = precision
        # bf16 mode: the two LayerNorms of every block folded into the GEMMs around them (fp_vit_model.ln_fold): gain into the
        # qkv / fc1 matrices, shift into their biases, LayerScale into the proj / fc2 matrices -- no LayerNorm kernel runs
        # inside the blocks (they were 6.5 % of a step).  fold_layernorm=False keeps the kernel-per-LayerNorm sequence.
        self.fold_layernorm = (bool(fold_layernorm) and precision == "bf16") or precision == "f16"
        self.resid_hilo = bool(resid_hilo) or precision == "f16"
        self.tall_tiles, self.sat_check = bool(tall_tiles), bool(sat_check)
        self._ld_pad_arg, self._ld_pad_qkv_arg, self._ld_pad8_arg = int(ld_pad), int(ld_pad_qkv), int(ld_pad8)
        self._sd, self.weights_source = _weights.resolve(self.model_base_name, self.arch, state_dict, weights, random_init_seed)
        # Precision schedule: blocks 0 .. head_blocks-1 run in `head_precision`, blocks head_blocks .. layer in this extractor's own precision, over one
        # fp32 stream (fp_vit_stream_f32 / fp_vit_forward_blocks): the fast "f16" pipeline in front of a near-exact tail, or the other way round.
        # Measured and NOT a shipped default: see profiles/EXPERIMENTS.md "precision schedules" (tools/schedule_sweep.py).
        self.head_blocks = int(head_blocks)
        self._head: Optional["DinoFeatureExtractor"] = None
        if self.head_blocks:
            modes = ("f16", "f16x3", "f16f8", "fp32")
            if precision not in modes or head_precision not in modes or not 0 < self.head_blocks <= self.layer or self.facet != "token" or use_graph:
                raise ValueError("head_blocks: 1 .. layer blocks in one of 'f16' / 'f16x3' / 'f16f8' / 'fp32' in front of an extractor of another of them (token facet, no graph replay)")
            # a folded-LayerNorm head stops BEFORE block k (fp_vit_forward_prefix, layer = k); the others run blocks 0 .. k-1 in full (fp_vit_forward, layer = k - 1)
            self._head_fold = head_precision == "f16"
            head_name = f"dinov2_version={self.version}_stride={self.stride}_facet=token_layer={self.head_blocks if self._head_fold else self.head_blocks - 1}_norm=1"
            self._head = DinoFeatureExtractor(head_name, state_dict=self._sd, precision=head_precision, arch=arch)
        self._device: Optional[torch.device] = None
        self._w: Dict[str, torch.Tensor] = {}
        self._model = None
        self._blocks = None
        self._grids: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor]] = {}
        self._ws: Dict[Tuple[int, int, int], Tuple[_lib.VitWorkspace, list]] = {}
        self._graphs: Dict[Tuple[int, int, int], tuple] = {}
        self.num_patches: Optional[Tuple[int, int]] = None
        # sticky device-side saturation counters (fp_vit_workspace.sat): [0] split-fp16 clamps (f16x3 mode), [1] e4m3 clamps (fp8 mode);
        # one pair per extractor, shared by all its workspaces; only ever incremented by the kernels, zeroed by reset_saturation()
        self._sat: Optional[torch.Tensor] = None
        self._fp8_sat_warned = False

    # ---- device placement (same call pattern as the reference: extractor.to(device))
    def to(self, device=None, *args, **kwargs):  # type: ignore[override]
        dev = torch.device(device if device is not None else "cuda")
        if dev.type != "cuda":
            raise _lib.FoundPoseNativeError("DinoFeatureExtractor runs on the MI355X only (device must be 'cuda'); no CPU path exists")
        self._prepare(dev)
        if self._head is not None:
            self._head.to(dev)
        return self

    def cuda(self, device=None):  # type: ignore[override]
        return self.to("cuda" if device is None else device)

    def _prepare(self, dev: torch.device) -> None:
        a, sd = self.arch, self._sd
        if self.precision in ("f16x3", "f16f8"):
            return self._prepare_split(dev)
        wdt = torch.float32 if self.precision == "fp32" else (torch.float16 if self.precision == "f16" else torch.bfloat16)
        w: Dict[str, torch.Tensor] = {}

        # Row strides of the block matrices and of the y / h activation buffers are padded by `pad` elements so that a
        # stride is never a multiple of 2 KiB: the 8 rows one staging instruction of the GEMM fetches then spread over
        # the L2 channels instead of queueing on one (ld_pad; 0 = dense; fp8 mode stays dense).
        pad = self._ld_pad_arg if self.precision != "fp8" else 0
        self._ld_pad8 = self._ld_pad8_arg if self.precision == "fp8" else 0  # bytes, fp8 operands (measured: no effect at 128, 1425 detections/s either way)
        if pad % 8:
            raise ValueError("ld_pad must be a multiple of 8")
        self._ld_pad = pad
        # the qkv buffer's stride can be padded too (ld_pad_qkv): in isolation the attention kernel and the qkv GEMM gain
        # 5-9 % from +64..128 elements, inside the pipeline nothing (998 detections/s at 0 / 64 / 128 / 256) -> dense
        self._ld_pad_qkv = self._ld_pad_qkv_arg

        def padded(t2d):  # [N, K] -> view of an [N, K + pad] buffer
            if pad == 0:
                return t2d.contiguous()
            buf = torch.zeros(t2d.shape[0], t2d.shape[1] + pad, dtype=t2d.dtype, device=t2d.device)
            buf[:, :t2d.shape[1]] = t2d
            return buf[:, :t2d.shape[1]]

        def mat(key):
            w[key] = padded(sd[key].to(dev, torch.float32).to(wdt))
            return w[key]

        def vec(key):
            w[key] = sd[key].to(dev, torch.float32).reshape(-1).contiguous()
            return w[key]

        kp = 3 * a.patch * a.patch
        kpad = (kp + 63) // 64 * 64
        pw = torch.zeros(a.dim, kpad, dtype=torch.float32, device=dev)
        pw[:, :kp] = sd["patch_embed.proj.weight"].to(dev, torch.float32).reshape(a.dim, kp)
        w["patch_w"] = pw.to(wdt).contiguous()
        vec("patch_embed.proj.bias")
        vec("norm.weight")
        vec("norm.bias")
        blocks = (_lib.VitBlock * a.depth)()
        fold = self.fold_layernorm

        def f32(key):
            return sd[key].to(dev, torch.float32)

        # f16 mode: every folded matrix carries a power-of-two scale s_w that brings its largest entry to ~2^14 (exact; undone in the epilogue through
        # fp_vit_block.act_scale[j] = 1 / s_w): fp16 keeps 11 significant bits only down to 6e-5, and a LayerScale-folded matrix diag(gamma) W of a checkpoint
        # whose gammas are small (DINOv2 initialises them at 1e-5) would otherwise sit in the subnormal range, where fp16 is WORSE than bf16
        h16 = self.precision == "f16"
        scales = {}

        def scaled(Wf32, name):
            from . import ops
            sw = ops.pow2_scale(Wf32) if h16 else 1.0
            scales[name] = sw
            return padded((Wf32 * sw).to(wdt)) if h16 else padded(Wf32.to(wdt))

        def folded_in(wkey, bkey, nkey, wmat=None, bvec=None, name=None):
            """LayerNorm (gain g, shift s) in front of a Linear (W, b): W' = W diag(g) in bf16 (f16: s_w W'), b' = b + W s, colsum of the stored W'."""
            W = f32(wkey) if wmat is None else wmat
            bb = f32(bkey) if bvec is None else bvec
            g, sh = f32(nkey + ".weight"), f32(nkey + ".bias")
            Wf = scaled(W * g[None, :], name)
            return Wf, (bb + W @ sh).contiguous(), Wf.float().sum(dim=1).contiguous()

        def folded_out(wkey, bkey, gkey, name=None):
            """LayerScale gamma behind a Linear: W'' = diag(gamma) W in bf16 (f16: s_w W''), b'' = gamma * b."""
            gm = f32(gkey)
            return scaled(f32(wkey) * gm[:, None], name), (gm * f32(bkey)).contiguous()

        for i in range(a.depth):
            p = f"blocks.{i}."
            b = blocks[i]
            if fold:
                b.ln1_w, b.ln1_b = ptr(vec(p + "norm1.weight")), ptr(vec(p + "norm1.bias"))
                b.ln2_w, b.ln2_b = ptr(vec(p + "norm2.weight")), ptr(vec(p + "norm2.bias"))
                b.ls1, b.ls2 = ptr(vec(p + "ls1.gamma")), ptr(vec(p + "ls2.gamma"))
                w[p + "qkv.wf"], w[p + "qkv.bf"], w[p + "qkv.cs"] = folded_in(p + "attn.qkv.weight", p + "attn.qkv.bias", p + "norm1", name=0)
                w[p + "proj.wf"], w[p + "proj.bf"] = folded_out(p + "attn.proj.weight", p + "attn.proj.bias", p + "ls1.gamma", name=1)
                if a.ffn == "mlp":
                    w[p + "fc1.wf"], w[p + "fc1.bf"], w[p + "fc1.cs"] = folded_in(p + "mlp.fc1.weight", p + "mlp.fc1.bias", p + "norm2", name=2)
                    w[p + "fc2.wf"], w[p + "fc2.bf"] = folded_out(p + "mlp.fc2.weight", p + "mlp.fc2.bias", p + "ls2.gamma", name=3)
                else:  # SwiGLU: rows of w12 interleaved (x1_j, x2_j), see below
                    w12, b12 = f32(p + "mlp.w12.weight"), f32(p + "mlp.w12.bias")
                    hdn = w12.shape[0] // 2
                    w12i = torch.stack([w12[:hdn], w12[hdn:]], 1).reshape(2 * hdn, -1)
                    b12i = torch.stack([b12[:hdn], b12[hdn:]], 1).reshape(-1)
                    w[p + "fc1.wf"], w[p + "fc1.bf"], w[p + "fc1.cs"] = folded_in(None, None, p + "norm2", w12i, b12i, name=2)
                    w[p + "fc2.wf"], w[p + "fc2.bf"] = folded_out(p + "mlp.w3.weight", p + "mlp.w3.bias", p + "ls2.gamma", name=3)
                b.qkv_w, b.qkv_b, b.qkv_colsum = ptr(w[p + "qkv.wf"]), ptr(w[p + "qkv.bf"]), ptr(w[p + "qkv.cs"])
                b.proj_w, b.proj_b = ptr(w[p + "proj.wf"]), ptr(w[p + "proj.bf"])
                b.fc1_w, b.fc1_b, b.fc1_colsum = ptr(w[p + "fc1.wf"]), ptr(w[p + "fc1.bf"]), ptr(w[p + "fc1.cs"])
                b.fc2_w, b.fc2_b = ptr(w[p + "fc2.wf"]), ptr(w[p + "fc2.bf"])
                for j in range(4):
                    b.act_scale[j] = 1.0 / scales[j] if h16 else 0.0
                continue
            b.ln1_w, b.ln1_b = ptr(vec(p + "norm1.weight")), ptr(vec(p + "norm1.bias"))
            b.ln2_w, b.ln2_b = ptr(vec(p + "norm2.weight")), ptr(vec(p + "norm2.bias"))
            b.ls1, b.ls2 = ptr(vec(p + "ls1.gamma")), ptr(vec(p + "ls2.gamma"))
            b.qkv_w, b.qkv_b = ptr(mat(p + "attn.qkv.weight")), ptr(vec(p + "attn.qkv.bias"))
            b.proj_w, b.proj_b = ptr(mat(p + "attn.proj.weight")), ptr(vec(p + "attn.proj.bias"))
            if a.ffn == "mlp":
                b.fc1_w, b.fc1_b = ptr(mat(p + "mlp.fc1.weight")), ptr(vec(p + "mlp.fc1.bias"))
                b.fc2_w, b.fc2_b = ptr(mat(p + "mlp.fc2.weight")), ptr(vec(p + "mlp.fc2.bias"))
            else:
                # SwiGLU (ViT-g): interleave the rows of w12 as (x1_j, x2_j) so the gate and the value of a hidden unit
                # land in adjacent GEMM columns and silu(x1) * x2 is a per-lane epilogue
                w12 = sd[p + "mlp.w12.weight"].to(dev, torch.float32)
                b12 = sd[p + "mlp.w12.bias"].to(dev, torch.float32)
                hdn = w12.shape[0] // 2
                w[p + "w12i"] = padded(torch.stack([w12[:hdn], w12[hdn:]], 1).reshape(2 * hdn, -1).to(wdt))
                w[p + "b12i"] = torch.stack([b12[:hdn], b12[hdn:]], 1).reshape(-1).contiguous()
                b.fc1_w, b.fc1_b = ptr(w[p + "w12i"]), ptr(w[p + "b12i"])
                b.fc2_w, b.fc2_b = ptr(mat(p + "mlp.w3.weight")), ptr(vec(p + "mlp.w3.bias"))
        m = _lib.VitModel()
        m.dim, m.depth, m.heads, m.hidden, m.registers, m.patch = a.dim, a.depth, a.heads, a.hidden, a.registers, a.patch
        m.ffn_swiglu = int(a.ffn != "mlp")
        m.patch_stride = 0 if self.stride == self.patch_size else self.stride
        m.weight_dtype = _lib.FP_F32 if self.precision == "fp32" else (_lib.FP_F16 if self.precision == "f16" else _lib.FP_BF16)  # "fp8": bf16 until calibrated (_to_fp8)
        m.patch_w, m.patch_k_pad, m.patch_b = ptr(w["patch_w"]), kpad, ptr(w["patch_embed.proj.bias"])
        m.norm_w, m.norm_b = ptr(w["norm.weight"]), ptr(w["norm.bias"])
        m.blocks = C.cast(blocks, C.POINTER(_lib.VitBlock))
        m.ld_w_dim, m.ld_w_hidden = (a.dim + pad, a.hidden + pad) if pad else (0, 0)  # fp8: set by _to_fp8
        m.ln_fold = int(fold)
        m.flags = 0 if self.tall_tiles else _lib.VIT_NO_TALL_TILES
        self._w, self._model, self._blocks, self._device = w, m, blocks, dev
        self._grids.clear()
        self._ws.clear()
        self._graphs.clear()
        if self.precision == "fp8" and self.act_scales is not None:
            self._to_fp8()

    def _prepare_split(self, dev: torch.device) -> None:
        """Weights of the f16x3 mode: every matrix as split-fp16 rows of s_w W (s_w a power of two that brings the largest weight
        to ~2^14: typical weights then sit ~2^11, their lo halves far above the fp16 subnormal range), act_scale = 1 / (s_in s_w)."""
        from . import ops
        a, sd = self.arch, self._sd
        w: Dict[str, torch.Tensor] = {}
        pad = self._ld_pad_arg
        if pad % 8:
            raise ValueError("ld_pad must be a multiple of 8")
        self._ld_pad, self._ld_pad8, self._ld_pad_qkv = pad, 0, 0

        def f32(key):
            return sd[key].to(dev, torch.float32)

        def vec(key):
            w[key] = f32(key).reshape(-1).contiguous()
            return w[key]

        sx = self.precision == "f16f8"
        pack = ops.splitx_pack if sx else ops.split16_pack

        def mat(name, W):  # -> (split rows, scale)
            sw = ops.pow2_scale(W)
            w[name] = pack(W, sw, pad)
            return w[name], sw

        kp = 3 * a.patch * a.patch
        kpad = (kp + 63) // 64 * 64
        pw = torch.zeros(a.dim, kpad, dtype=torch.float32, device=dev)
        pw[:, :kp] = f32("patch_embed.proj.weight").reshape(a.dim, kp)
        spw = ops.pow2_scale(pw)
        w["patch_w"] = pack(pw, spw)
        vec("patch_embed.proj.bias")
        vec("norm.weight")
        vec("norm.bias")
        blocks = (_lib.VitBlock * a.depth)()
        S_ACT, S_HID = _lib.SPLIT_SCALE_ACT, _lib.SPLIT_SCALE_HID
        for i in range(a.depth):
            p = f"blocks.{i}."
            b = blocks[i]
            b.ln1_w, b.ln1_b = ptr(vec(p + "norm1.weight")), ptr(vec(p + "norm1.bias"))
            b.ln2_w, b.ln2_b = ptr(vec(p + "norm2.weight")), ptr(vec(p + "norm2.bias"))
            b.ls1, b.ls2 = ptr(vec(p + "ls1.gamma")), ptr(vec(p + "ls2.gamma"))
            if a.ffn == "mlp":
                fc1_w, fc1_b = f32(p + "mlp.fc1.weight"), vec(p + "mlp.fc1.bias")
                fc2_w, fc2_b = f32(p + "mlp.fc2.weight"), vec(p + "mlp.fc2.bias")
            else:  # SwiGLU: rows of w12 interleaved (x1_j, x2_j) like the other modes
                w12, b12 = f32(p + "mlp.w12.weight"), f32(p + "mlp.w12.bias")
                hdn = w12.shape[0] // 2
                fc1_w = torch.stack([w12[:hdn], w12[hdn:]], 1).reshape(2 * hdn, -1)
                w[p + "b12i"] = fc1_b = torch.stack([b12[:hdn], b12[hdn:]], 1).reshape(-1).contiguous()
                fc2_w, fc2_b = f32(p + "mlp.w3.weight"), vec(p + "mlp.w3.bias")
            mats = [("qkv", f32(p + "attn.qkv.weight"), vec(p + "attn.qkv.bias"), S_ACT), ("proj", f32(p + "attn.proj.weight"), vec(p + "attn.proj.bias"), S_ACT),
                    ("fc1", fc1_w, fc1_b, S_ACT), ("fc2", fc2_w, fc2_b, S_HID)]
            for j, (field, W, bias, s_in) in enumerate(mats):
                ws_, sw = mat(p + field + ".split", W)
                setattr(b, field + "_w", ptr(ws_))
                setattr(b, field + "_b", ptr(bias))
                b.act_scale[j] = 1.0 / (s_in * sw)
        m = _lib.VitModel()
        m.dim, m.depth, m.heads, m.hidden, m.registers, m.patch = a.dim, a.depth, a.heads, a.hidden, a.registers, a.patch
        m.ffn_swiglu = int(a.ffn != "mlp")
        m.patch_stride = 0 if self.stride == self.patch_size else self.stride
        m.weight_dtype = _lib.FP_F16F8 if sx else _lib.FP_F16X3
        m.patch_w, m.patch_k_pad, m.patch_b = ptr(w["patch_w"]), kpad, ptr(w["patch_embed.proj.bias"])
        m.patch_acc_scale = 1.0 / (S_ACT * spw)
        m.norm_w, m.norm_b = ptr(w["norm.weight"]), ptr(w["norm.bias"])
        m.blocks = C.cast(blocks, C.POINTER(_lib.VitBlock))
        m.ld_w_dim, m.ld_w_hidden = (2 * a.dim + pad, 2 * a.hidden + pad) if pad else (0, 0)
        m.ln_fold = 0
        self._w, self._model, self._blocks, self._device = w, m, blocks, dev
        self._grids.clear()
        self._ws.clear()
        self._graphs.clear()

    def _run_backbone(self, images, ws, B: int, H: int, W: int, prefix_only: bool) -> None:
        """fp_vit_forward / fp_vit_forward_prefix, or -- with a precision schedule -- the head's blocks in the f16 mode, its stream as fp32 into this
        workspace, and this model's blocks behind it."""
        if self._head is None:
            call("fp_vit_forward_prefix" if prefix_only else "fp_vit_forward", C.byref(self._model), C.byref(ws), ptr(images), B, H, W, self.layer, stream())
            return
        hd, k = self._head, self.head_blocks
        gh, gw = hd._grid(H, W)
        pos_patch, prefix = hd._grid_tables(gh, gw, H, W)
        hd._model.pos_patch, hd._model.prefix = ptr(pos_patch), ptr(prefix)
        hws, _ = hd._workspace(B, gh, gw)
        if self._head_fold:
            call("fp_vit_forward_prefix", C.byref(hd._model), C.byref(hws), ptr(images), B, H, W, k, stream())      # embedding + blocks 0 .. k-1, the stream as the (hi, lo) pair
        else:
            call("fp_vit_forward", C.byref(hd._model), C.byref(hws), ptr(images), B, H, W, k - 1, stream())         # embedding + blocks 0 .. k-1, the stream in its ws.x
        call("fp_vit_stream_f32", C.byref(hd._model), C.byref(hws), B, H, W, k, ws.x, stream())                      # -> this workspace's fp32 stream
        call("fp_vit_forward_blocks", C.byref(self._model), C.byref(ws), B, H, W, k, self.layer, int(prefix_only), stream())

    def _grid(self, H: int, W: int) -> Tuple[int, int]:
        """Patch tokens per axis (dinov2_utils.py:266-269): 1 + (size - patch) // stride; at stride == patch size the image must tile."""
        if self.stride == self.patch_size:
            if H % self.patch_size or W % self.patch_size:
                raise ValueError(f"image size {H}x{W} is not a multiple of the patch size {self.patch_size}")
            return H // self.patch_size, W // self.patch_size
        if H < self.patch_size or W < self.patch_size:
            raise ValueError(f"image size {H}x{W} is smaller than a patch")
        return 1 + (H - self.patch_size) // self.stride, 1 + (W - self.patch_size) // self.stride

    def _grid_tables(self, gh: int, gw: int, H: int = 0, W: int = 0):
        key = (gh, gw) if self.stride == self.patch_size else (gh, gw, H, W)
        if key not in self._grids:
            a, sd = self.arch, self._sd
            if self.stride == self.patch_size:
                pos = _interpolate_pos_embed(sd["pos_embed"].float(), a, gh, gw)[0]  # [1+Np, D]
            else:
                pos = _interpolate_pos_embed_strided(sd["pos_embed"].float(), a.patch, self.stride, H, W)[0]
            rows = [sd["cls_token"].float().reshape(1, -1) + pos[:1]]
            if a.registers:
                rows.append(sd["register_tokens"].float().reshape(a.registers, -1))
            prefix = torch.cat(rows, 0).to(self._device).contiguous()
            self._grids[key] = (pos[1:].to(self._device).contiguous(), prefix)
        return self._grids[key]

    def padded_rows(self, rows: int) -> int:
        """Rows of the activation buffers for `rows` tokens: whole 256-row GEMM tiles, and whole 320-row tiles as well (the taller tile of the
        wide bf16 / fp8 GEMMs, csrc/gemm_bf16.hip) when that costs < 3 % more rows.  Row tiles without live rows are never launched."""
        m_pad = (rows + 255) // 256 * 256
        m1280 = (rows + 1279) // 1280 * 1280
        if (self.fold_layernorm or self.precision == "fp8") and m1280 * 100 <= m_pad * 103:
            m_pad = m1280
        return m_pad

    def _workspace(self, B: int, gh: int, gw: int):
        key = (B, gh, gw, torch.cuda.current_stream().cuda_stream)  # one workspace per stream: concurrent sub-batches
        if key not in self._ws:
            a, dev = self.arch, self._device
            sp = self.precision in ("f16x3", "f16f8")
            adt = torch.float32 if self.precision == "fp32" else (torch.float16 if (sp or self.precision == "f16") else torch.bfloat16)
            em = 2 if sp else 1  # stored elements per logical element of an operand row (split rows: hi + lo halves)
            np_, ntok = gh * gw, 1 + a.registers + gh * gw
            m_pad = self.padded_rows(B * ntok)
            mp_pad = (B * np_ + 255) // 256 * 256
            bufs = [
                torch.zeros(mp_pad, em * self._model.patch_k_pad, dtype=adt, device=dev),
                torch.zeros(m_pad, a.dim, dtype=torch.float32, device=dev),
                torch.zeros(m_pad, em * a.dim + self._ld_pad, dtype=adt, device=dev),
                torch.zeros(m_pad, em * 3 * a.dim + self._ld_pad_qkv, dtype=adt, device=dev),
                torch.zeros(m_pad, em * a.hidden + self._ld_pad, dtype=adt, device=dev),
            ]
            if self.precision == "fp8":
                p8 = self._ld_pad8
                bufs.append(torch.zeros(m_pad, max(a.dim, a.hidden) + p8, dtype=torch.uint8, device=dev))
            ws = _lib.VitWorkspace()
            if self.fold_layernorm:  # bf16 copy of the residual stream + partial row sums per 128-column tile
                bufs.append(torch.zeros(m_pad, a.dim + self._ld_pad, dtype=adt, device=dev))
                bufs.append(torch.zeros(a.dim // 128 + 1, m_pad, 2, dtype=torch.float32, device=dev))
                ws.xb, ws.stats = ptr(bufs[-2]), ptr(bufs[-1])
                if self.resid_hilo:   # low halves of the (hi, lo) residual stream of the blocks in front of the hooked one (A/B switch)
                    bufs.append(torch.zeros(m_pad, a.dim + self._ld_pad, dtype=adt, device=dev))
                    ws.xl = ptr(bufs[-1])
            ws.patches, ws.x, ws.y, ws.qkv, ws.h = (ptr(t) for t in bufs[:5])
            ws.a8 = ptr(bufs[5]) if self.precision == "fp8" else None
            ws.ld_y, ws.ld_h = (em * a.dim + self._ld_pad, em * a.hidden + self._ld_pad) if self._ld_pad else (0, 0)
            if self.precision == "fp8" and self._ld_pad8:  # byte strides of a8 and of the hidden bytes kept in h
                ws.ld_y, ws.ld_h = a.dim + self._ld_pad8, a.hidden + self._ld_pad8
            ws.ld_qkv = em * 3 * a.dim + self._ld_pad_qkv
            ws.m_pad, ws.m_patch_pad = m_pad, mp_pad
            if self._sat is None:
                self._sat = torch.zeros(2, dtype=torch.int32, device=dev)
            ws.sat = ptr(self._sat)
            self._ws[key] = (ws, bufs)
        return self._ws[key]

    # ---- saturation report (f16x3 / fp8 modes)
    def saturation_counts(self) -> Tuple[int, int]:
        """(split-fp16 clamps, e4m3 clamps) reported by the kernels since the last reset_saturation(); synchronises.  Counted per
        reporting thread: non-zero means at least one live activation left the representable range of its operand row."""
        if self._sat is None:
            return 0, 0
        a, b = self._sat.tolist()
        return int(a), int(b)

    def reset_saturation(self) -> None:
        if self._sat is not None:
            self._sat.zero_()

    def check_saturation(self) -> None:
        """Raises FoundPoseSaturationError if the f16x3 mode clamped an activation since the last reset (sticky: it keeps raising until
        reset_saturation()); warns once if the fp8 mode clamped beyond its calibration scales (expected now and then with static
        scales, but never silent).  Synchronises.  forward() calls it; the batched engine attributes clamps to the batch that caused them
        instead (saturation_snapshot / report_saturation: a result raises for its own batch only)."""
        self.report_saturation(*self.saturation_counts())

    def saturation_snapshot(self) -> Optional[torch.Tensor]:
        """Device-side copy of the two counters, enqueued on the current stream (no sync); None before the first workspace exists (= zeros)."""
        return None if self._sat is None else self._sat.clone()

    def report_saturation(self, n16: int, n8: int) -> None:
        """The verdict for a pair of counts (split-fp16 clamps, e4m3 clamps): raises in the f16x3 mode, warns once in the fp8 mode."""
        if n16 and self.precision == "f16":
            raise _lib.FoundPoseSaturationError(
                f"precision='f16': {n16} kernel thread(s) of the final norm produced non-finite features -- an activation beyond the fp16 range (|x| > 65504 -> inf) "
                "somewhere in the backbone, or a NaN: the features of this batch are not usable.  Use precision='bf16' or 'fp32' for this checkpoint "
                "(or reset_saturation() to acknowledge).")
        if n16 and self.precision in ("f16x3", "f16f8"):
            raise _lib.FoundPoseSaturationError(
                f"precision='{self.precision}': {n16} kernel thread(s) clamped an activation to the split-fp16 range (|x| > {65504 / _lib.SPLIT_SCALE_ACT:.0f} for "
                f"LayerNorm outputs / q / k / v, > {65504 / _lib.SPLIT_SCALE_HID:.0f} for hidden activations) or met a NaN: the features are not the fp32 "
                "arithmetic's.  Use precision='fp32' for this checkpoint (or reset_saturation() to acknowledge).")
        if n8 and self.precision == "fp8" and not self._fp8_sat_warned:
            import warnings
            self._fp8_sat_warned = True
            warnings.warn(f"precision='fp8': {n8} kernel thread(s) clamped an activation at +-448 (inputs beyond the static calibration scales)")

    # ---- forward
    def forward_tokens(self, images: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (fmap [B, Np, D] fp32 token-major, cls [B, D] fp32); the batched fast-path entry."""
        if self._model is None:
            raise _lib.FoundPoseNativeError("call extractor.to('cuda') before running it")
        _lib.require_cuda(images)
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError("images must be [B, 3, H, W]")
        images = images.float().contiguous()
        B, _, H, W = images.shape
        gh, gw = self._grid(H, W)
        pos_patch, prefix = self._grid_tables(gh, gw, H, W)
        self._model.pos_patch, self._model.prefix = ptr(pos_patch), ptr(prefix)
        ws, _ = self._workspace(B, gh, gw)
        if self.precision == "fp8" and self._model.weight_dtype != _lib.FP_FP8:
            raise _lib.FoundPoseNativeError(
                "precision='fp8' needs its static activation scales before the first forward: construct the extractor with "
                "act_scales= (the table stored with the bank) or call calibrate_fp8(calibration_images) once")
        if self.facet != "token":
            fmap, cls = self._forward_facet(images, B, H, W, gh, gw)
        elif self.use_graph:
            fmap, cls = self._forward_graph(images, ws, B, H, W, gh, gw)
        else:
            fmap = torch.empty(B, gh * gw, self.arch.dim, dtype=torch.float32, device=images.device)
            cls = torch.empty(B, self.arch.dim, dtype=torch.float32, device=images.device)
            self._launch(images, ws, B, H, W, gh, gw, fmap, cls)
        self.num_patches = (gh, gw)
        return fmap, cls

    @property
    def supports_token_selection(self) -> bool:
        """The hooked block can be computed for a subset of the tokens (fp_vit_block_selected): bf16 with folded LayerNorms, fp8, or f16x3."""
        mode_ok = (self.precision in ("bf16", "f16") and self.fold_layernorm) or self.precision in ("f16x3", "f16f8", "fp8")
        return (self.facet == "token" and not self.use_graph and mode_ok and self.layer >= 0
                and self.stride == self.patch_size)   # the selection maps query points to 14-px cells

    def forward_hidden(self, images: torch.Tensor, prefix_only: bool = False) -> Tuple[int, int, int]:
        """Runs the backbone up to the hooked block and leaves its output in the workspace (no final norm, no feature map):
        the first half of the engine's fused path, followed by sample_patch_features.  -> (B, gh, gw).  Token facet only.
        prefix_only: stop BEFORE the hooked block (fp_vit_forward_prefix); forward_selected_block then runs it for the tokens
        the sampling needs."""
        if self._model is None:
            raise _lib.FoundPoseNativeError("call extractor.to('cuda') before running it")
        if self.facet != "token" or self.use_graph:
            raise NotImplementedError("forward_hidden serves the eager token path")
        if prefix_only and not self.supports_token_selection:
            raise NotImplementedError("token selection needs the bf16 mode with folded LayerNorms, the fp8 or the f16x3 mode")
        _lib.require_cuda(images)
        images = images.float().contiguous()
        B, _, H, W = images.shape
        gh, gw = self._grid(H, W)
        pos_patch, prefix = self._grid_tables(gh, gw, H, W)
        self._model.pos_patch, self._model.prefix = ptr(pos_patch), ptr(prefix)
        ws, _ = self._workspace(B, gh, gw)
        if self.precision == "fp8" and self._model.weight_dtype != _lib.FP_FP8:
            raise _lib.FoundPoseNativeError("precision='fp8' needs its static activation scales before the first forward (act_scales= / calibrate_fp8)")
        self._run_backbone(images, ws, B, H, W, prefix_only)
        self.num_patches = (gh, gw)
        self._hidden = (B, gh, gw, H, W)
        return B, gh, gw

    def forward_selected_block(self, sel_rows: torch.Tensor, sel_off: torch.Tensor, num_sel: int, max_sel_per_img: int) -> None:
        """The hooked block for the selected tokens only, after forward_hidden(prefix_only=True).  sel_rows: i32 global token
        rows (b * n_tok + token; ascending, grouped by image; entries past num_sel are ignored), sel_off [B+1] i32."""
        B, gh, gw, H, W = self._hidden
        ws, _ = self._workspace(B, gh, gw)
        call("fp_vit_block_selected", C.byref(self._model), C.byref(ws), B, H, W, self.layer, ptr(sel_rows), ptr(sel_off), int(num_sel),
             int(max_sel_per_img), stream())

    def sample_patch_features(self, points: torch.Tensor, point_img: torch.Tensor, row_map: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Final norm + sample_feature_map_at_points for the batch forward_hidden just ran: points [P, 2] in image coordinates,
        point_img [P] i32 -> [P, D] fp32, bit-identical to forward()["feature_maps"] sampled with ops.sample_bilinear.
        row_map (after forward_selected_block): i32 [B * gh * gw], patch cell -> row of the compact selected-token buffer."""
        B, gh, gw, H, W = self._hidden
        ws, _ = self._workspace(B, gh, gw)
        points = points.float().contiguous()
        out = torch.empty(points.shape[0], self.arch.dim, dtype=torch.float32, device=points.device)
        if row_map is None:
            call("fp_vit_sample_features", C.byref(self._model), C.byref(ws), B, gh, gw, int(self.apply_norm), W, H, ptr(points), ptr(point_img),
                 points.shape[0], ptr(out), stream())
        else:
            call("fp_vit_sample_features_selected", C.byref(self._model), C.byref(ws), B, gh, gw, int(self.apply_norm), W, H, ptr(points),
                 ptr(point_img), points.shape[0], ptr(row_map), ptr(out), stream())
        return out

    def _forward_facet(self, images, B, H, W, gh, gw):
        """key / query / value facet (dinov2_utils.py:176-194, 294-311 in the reference): the qkv projection of
        blocks[layer] is what the forward leaves in the workspace; per token the reference orders the vector (d, head)."""
        from . import ops
        a = self.arch
        ws, bufs = self._workspace(B, gh, gw)
        call("fp_vit_forward", C.byref(self._model), C.byref(ws), ptr(images), B, H, W, self.layer, stream())
        ntok, fi = 1 + a.registers + gh * gw, {"query": 0, "key": 1, "value": 2}[self.facet]
        if self.precision in ("f16x3", "f16f8"):  # q | k | v are split-fp16 rows in both modes (hi + lo halves, scale FP_SPLIT_SCALE_QKV) -> fp32
            f = ops.split16_unpack(bufs[3][:B * ntok, fi * 2 * a.dim:(fi + 1) * 2 * a.dim].contiguous(), _lib.SPLIT_SCALE_QKV)
        else:
            f = bufs[3][:B * ntok, fi * a.dim:(fi + 1) * a.dim].float()
        f = f.reshape(B, ntok, a.heads, a.head_dim).permute(0, 1, 3, 2).reshape(B, ntok, a.dim)
        tok = torch.cat([f[:, :1], f[:, 1 + a.registers:]], dim=1)
        if self.apply_norm:
            tok = ops.layernorm(tok.reshape(-1, a.dim).contiguous(), self._w["norm.weight"], self._w["norm.bias"], torch.float32).reshape(B, -1, a.dim)
        return tok[:, 1:].contiguous(), tok[:, 0].contiguous()

    # ---- fp8 mode
    FP8_CALIBRATION_HEADROOM = 1.5

    def calibrate_fp8(self, images: Optional[torch.Tensor] = None, act_scales: Optional[torch.Tensor] = None,
                      headroom: Optional[float] = None) -> torch.Tensor:
        """Fixes the static activation scales of the fp8 mode and quantises the block matrices.

        Either pass `act_scales` [depth, 4] (scale = 448 / amax of the inputs of qkv, proj, fc1, fc2 of each block) or a
        calibration batch `images`: the blocks are then run once in bf16, op by op, and the largest magnitude of each
        GEMM input is recorded; the scales are 448 / (headroom x that maximum).  The maximum over a calibration sample
        underestimates the maximum over everything the model will see, and e4m3 is a FLOATING format: room above the sample
        maximum costs no relative precision (only the smallest binade), so the default keeps 1.5x (a sample-maximum scale,
        headroom=1.0, clamped tens of thousands of values when 32 calibration crops served 128).  -> the scales in use."""
        headroom = self.FP8_CALIBRATION_HEADROOM if headroom is None else float(headroom)
        if headroom < 1.0:
            raise ValueError("headroom < 1 would clamp the calibration batch itself")
        from . import ops
        if self.precision != "fp8" or self._model is None:
            raise _lib.FoundPoseNativeError("calibrate_fp8 needs precision='fp8' and extractor.to('cuda')")
        a, dev, w = self.arch, self._device, self._w
        if act_scales is None:
            if images is None:
                raise ValueError("calibrate_fp8 needs a calibration batch or explicit act_scales")
            images = images.float().contiguous()
            B, _, H, W = images.shape
            gh, gw = self._grid(H, W)
            pos_patch, prefix = self._grid_tables(gh, gw, H, W)
            self._model.pos_patch, self._model.prefix = ptr(pos_patch), ptr(prefix)
            self._model.weight_dtype = _lib.FP_BF16
            ws, bufs = self._workspace(B, gh, gw)
            call("fp_vit_forward", C.byref(self._model), C.byref(ws), ptr(images), B, H, W, -1, stream())  # embedding only
            x, ntok = bufs[1], 1 + a.registers + gh * gw
            mv = B * ntok
            amax = torch.zeros(a.depth, 4)
            for i in range(self.layer + 1):
                p = f"blocks.{i}."
                fc1_w, fc1_b, fc2 = (w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"], "mlp.fc2") if a.ffn == "mlp" else (w[p + "w12i"], w[p + "b12i"], "mlp.w3")
                y = ops.layernorm(x, w[p + "norm1.weight"], w[p + "norm1.bias"], torch.bfloat16)
                qkv = ops.gemm_bf16(y, w[p + "attn.qkv.weight"], w[p + "attn.qkv.bias"], epilogue=0, m_valid=mv)
                o = ops.attention(qkv, B, ntok, a.dim, a.heads)
                ops.gemm_bf16(o, w[p + "attn.proj.weight"], w[p + "attn.proj.bias"], gamma=w[p + "ls1.gamma"], out=x, epilogue=3, m_valid=mv)
                y2 = ops.layernorm(x, w[p + "norm2.weight"], w[p + "norm2.bias"], torch.bfloat16)
                h = ops.gemm_bf16(y2, fc1_w, fc1_b, epilogue=1 if a.ffn == "mlp" else 6, m_valid=mv)
                ops.gemm_bf16(h, w[p + fc2 + ".weight"], w[p + fc2 + ".bias"], gamma=w[p + "ls2.gamma"], out=x, epilogue=3, m_valid=mv)
                amax[i] = torch.stack([t[:mv].float().abs().max() for t in (y, o, y2, h)]).cpu()
            amax[self.layer + 1:] = 1.0
            act_scales = 448.0 / (headroom * amax).clamp_min(1e-12)
        self.act_scales = act_scales.float().cpu().clone()
        self._to_fp8()
        return self.act_scales

    def _to_fp8(self) -> None:
        """Per-output-channel e4m3 block matrices, column scales and pre-divided biases for fp_vit_forward's fp8 path."""
        from . import ops
        a, w = self.arch, self._w
        for i in range(a.depth):
            p, b = f"blocks.{i}.", self._blocks[i]
            names = [("attn.qkv.weight", "attn.qkv.bias", None), ("attn.proj.weight", "attn.proj.bias", "ls1.gamma"),
                     ("mlp.fc1.weight", "mlp.fc1.bias", None) if a.ffn == "mlp" else ("w12i", "b12i", None),
                     ("mlp.fc2.weight", "mlp.fc2.bias", "ls2.gamma") if a.ffn == "mlp" else ("mlp.w3.weight", "mlp.w3.bias", "ls2.gamma")]
            for j, ((wk, bk, gk), field) in enumerate(zip(names, ("qkv", "proj", "fc1", "fc2"))):
                wt = self._fp32_matrix(p, wk)  # quantised from the fp32 checkpoint values, not from their bf16 rounding
                sw = 448.0 / wt.abs().amax(dim=1).clamp_min(1e-12)
                deq = 1.0 / (float(self.act_scales[i, j]) * sw)
                q8 = ops.quantize_fp8((wt * sw[:, None]).contiguous(), 1.0)
                if self._ld_pad8:  # rows K + pad bytes apart (L2 channel spread, as for the bf16 operands)
                    buf = torch.zeros(q8.shape[0], q8.shape[1] + self._ld_pad8, dtype=torch.uint8, device=q8.device)
                    buf[:, :q8.shape[1]] = q8.view(torch.uint8)
                    q8 = buf
                w[p + wk + ".f8"] = q8
                w[p + wk + ".b8"] = (w[p + bk] / deq).contiguous()
                w[p + wk + ".s8"] = (deq * w[p + gk] if gk else deq).contiguous()
                setattr(b, field + "_w", ptr(w[p + wk + ".f8"]))
                setattr(b, field + "_b", ptr(w[p + wk + ".b8"]))
                setattr(b, field + "_s", ptr(w[p + wk + ".s8"]))
                b.act_scale[j] = float(self.act_scales[i, j])
        self._model.weight_dtype = _lib.FP_FP8
        p8 = self._ld_pad8
        self._model.ld_w_dim, self._model.ld_w_hidden = (a.dim + p8, a.hidden + p8) if p8 else (0, 0)
        self._graphs.clear()

    def _fp32_matrix(self, p: str, wk: str) -> torch.Tensor:
        sd, dev = self._sd, self._device
        if wk == "w12i":  # SwiGLU: rows interleaved (x1_j, x2_j) like the bf16 operand
            w12 = sd[p + "mlp.w12.weight"].to(dev, torch.float32)
            hdn = w12.shape[0] // 2
            return torch.stack([w12[:hdn], w12[hdn:]], 1).reshape(2 * hdn, -1).contiguous()
        return sd[p + wk].to(dev, torch.float32).contiguous()

    def _launch(self, images, ws, B, H, W, gh, gw, fmap, cls) -> None:
        """The ~125 kernel launches of one forward (C++ launch sequence) on the current stream."""
        self._run_backbone(images, ws, B, H, W, False)
        call("fp_vit_features", C.byref(self._model), C.byref(ws), B, gh * gw, int(self.apply_norm), ptr(fmap), ptr(cls), stream())

    def _forward_graph(self, images, ws, B, H, W, gh, gw):
        """hipGraph replay of the launch sequence: one graph launch per batch instead of ~125 kernel launches.
        Captured once per (B, grid) workspace over static input / output buffers; the returned tensors are those
        static buffers and are overwritten by the next call with the same batch shape."""
        key = (B, gh, gw)
        entry = self._graphs.get(key)
        if entry is None:
            dev = images.device
            img_s = torch.empty_like(images)
            fmap = torch.empty(B, gh * gw, self.arch.dim, dtype=torch.float32, device=dev)
            cls = torch.empty(B, self.arch.dim, dtype=torch.float32, device=dev)
            img_s.copy_(images)
            self._launch(img_s, ws, B, H, W, gh, gw, fmap, cls)  # warm-up outside capture (function attributes, lazy module load)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._launch(img_s, ws, B, H, W, gh, gw, fmap, cls)
            entry = self._graphs[key] = (g, img_s, fmap, cls)
        g, img_s, fmap, cls = entry
        img_s.copy_(images)
        g.replay()
        return fmap, cls

    def forward(self, images: torch.Tensor) -> Dict[str, torch.Tensor]:
        B, _, H, W = images.shape
        fmap, cls = self.forward_tokens(images)
        if self.precision in ("f16", "f16x3", "f16f8", "fp8") and self.sat_check:
            self.check_saturation()   # one host sync; the reference's forward is synchronous too (CPU tensors)
        if self.precision == "f16" and (not self.apply_norm or self.facet != "token") and not bool(torch.isfinite(fmap).all()):
            # (no final-norm kernel ran on this output: raw hidden states or a q / k / v facet -- the same verdict, checked here)
            raise _lib.FoundPoseSaturationError("precision='f16': non-finite features -- an activation beyond the fp16 range (|x| > 65504) in the backbone, or a NaN")
        gh, gw = self._grid(H, W)
        # [B, D, Hp, Wp] as a permuted VIEW of the token-major buffer, exactly like the reference's output
        return {"cls_tokens": cls, "feature_maps": fmap.reshape(B, gh, gw, self.arch.dim).permute(0, 3, 1, 2)}
