"""CPU oracle for the feature-extraction half (TEST INFRASTRUCTURE ONLY).

fp32 torch-CPU restatement of what the reference's extractor computes
(/root/reference/utils/dinov2_utils.py:115-158 over the DINOv2 backbone called at
dinov2_utils.py:82,257).  The backbone lives in the un-vendored submodule
external/dinov2 (empty in the mount, SHA unrecorded) -> PARITY UNPINNED for the
backbone arithmetic; it is restated from the published DINOv2 architecture
(vision_transformer.py / layers/{patch_embed,attention,block,mlp,swiglu_ffn,layer_scale}.py)
and cross-checked in tests against transformers' independent Dinov2WithRegisters model.
The wrapper logic around it (hook on blocks[layer], CLS/register slicing, final norm,
reshape/permute) IS pinned against the reference wrapper (tests/golden/extractor_*.npz).

`quant="bf16"` is "oracle B": every GEMM operand is rounded to bf16 at the points where
the MI355X bf16 path casts (LN outputs, qkv, softmax probabilities, attention output,
GELU output, weights), accumulation in fp32.
"""

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from foundpose_amd.vit_config import VitArch

IMNET_MEAN = (0.485, 0.456, 0.406)
IMNET_STD = (0.229, 0.224, 0.225)


def _q(x: torch.Tensor, quant: Optional[str]) -> torch.Tensor:
    if quant == "bf16":
        return x.to(torch.bfloat16).to(torch.float32)
    return x


def _fq_act(x: torch.Tensor, scale: float) -> torch.Tensor:
    """fp8 mode: a GEMM input as the device sees it -- times its static scale, clamped to +-448, rounded to OCP e4m3 --
    and back to fp32 (de-scaled)."""
    return (x * scale).clamp(-448, 448).to(torch.float8_e4m3fn).to(torch.float32) / scale


def _fq_weight(w: torch.Tensor) -> torch.Tensor:
    """fp8 mode: a block matrix quantised per output channel (scale = 448 / row amax) from its fp32 checkpoint values."""
    sw = 448.0 / w.abs().amax(dim=1, keepdim=True).clamp_min(1e-12)
    return (w * sw).clamp(-448, 448).to(torch.float8_e4m3fn).to(torch.float32) / sw


def normalize_images(images: torch.Tensor) -> torch.Tensor:
    mean = torch.tensor(IMNET_MEAN, dtype=images.dtype).view(1, 3, 1, 1)
    std = torch.tensor(IMNET_STD, dtype=images.dtype).view(1, 3, 1, 1)
    return (images - mean) / std


def interpolate_pos_embed(pos_embed: torch.Tensor, arch: VitArch, gh: int, gw: int) -> torch.Tensor:
    """Upstream `interpolate_pos_encoding` ([upstream] dinov2/models/vision_transformer.py)."""
    n = pos_embed.shape[1] - 1
    m = int(math.sqrt(n))
    if gh * gw == n and gh == gw:
        return pos_embed
    cls_pos, patch_pos = pos_embed[:, :1], pos_embed[:, 1:]
    dim = pos_embed.shape[-1]
    kwargs = {}
    if arch.interp_offset:
        kwargs["scale_factor"] = (float(gh + arch.interp_offset) / m, float(gw + arch.interp_offset) / m)
    else:
        kwargs["size"] = (gh, gw)
    patch_pos = F.interpolate(
        patch_pos.reshape(1, m, m, dim).permute(0, 3, 1, 2), mode="bicubic",
        antialias=arch.interp_antialias, **kwargs,
    )
    assert patch_pos.shape[-2:] == (gh, gw)
    patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat([cls_pos, patch_pos], dim=1)


def interpolate_pos_embed_strided(pos_embed: torch.Tensor, patch: int, stride: int, H: int, W: int) -> torch.Tensor:
    """The position-encoding function the reference installs when stride != patch size
    (/root/reference/utils/dinov2_utils.py:325-360, `_fix_pos_enc`; upstream calls it as (x, w, h) = (tokens, image HEIGHT, image WIDTH)):
    token counts 1 + (size - patch) // stride per axis, bicubic, scale-factor mode with the +0.1 fudge, no antialias."""
    n = pos_embed.shape[1] - 1
    m = int(math.sqrt(n))
    w0, h0 = 1 + (H - patch) // stride, 1 + (W - patch) // stride     # the reference's names: w = first spatial size
    if w0 * h0 == n and H == W:
        return pos_embed
    dim = pos_embed.shape[-1]
    grid = F.interpolate(pos_embed[:, 1:].reshape(1, m, m, dim).permute(0, 3, 1, 2), scale_factor=((w0 + 0.1) / m, (h0 + 0.1) / m),
                         mode="bicubic", align_corners=False, recompute_scale_factor=False)
    assert grid.shape[-2] == w0 and grid.shape[-1] == h0
    return torch.cat([pos_embed[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, dim)], dim=1)


def embed_tokens(sd: Dict[str, torch.Tensor], arch: VitArch, x: torch.Tensor, quant=None, stride: Optional[int] = None) -> torch.Tensor:
    """Normalised image -> [B, 1+R+Np, D] token sequence (cls, registers, patches).  stride (default: the patch size): the conv stride
    the reference's patch_vit_resolution sets (dinov2_utils.py:364-389) -- overlapping patches, its own position-encoding function."""
    B, _, H, W = x.shape
    gh, gw = H // arch.patch, W // arch.patch
    w = _q(sd["patch_embed.proj.weight"], quant)
    st = arch.patch if stride is None else stride
    t = F.conv2d(_q(x, quant), w, sd["patch_embed.proj.bias"], stride=st)
    t = t.flatten(2).transpose(1, 2)  # [B, Np, D], row-major over (gy, gx)
    t = torch.cat([sd["cls_token"].expand(B, -1, -1), t], dim=1)
    if st != arch.patch:
        t = t + interpolate_pos_embed_strided(sd["pos_embed"], arch.patch, st, H, W)
        if arch.registers:
            t = torch.cat([t[:, :1], sd["register_tokens"].expand(B, -1, -1), t[:, 1:]], dim=1)
        return t
    t = t + interpolate_pos_embed(sd["pos_embed"], arch, gh, gw)
    if arch.registers:
        t = torch.cat([t[:, :1], sd["register_tokens"].expand(B, -1, -1), t[:, 1:]], dim=1)
    return t


def block_forward(sd, arch: VitArch, i: int, x: torch.Tensor, quant=None, fp8_act=None) -> torch.Tensor:
    """fp8_act: the four static activation scales of this block (inputs of qkv, proj, fc1, fc2) -> "oracle C", the fp8
    mode: bf16 everywhere like quant="bf16", plus e4m3 fake quantisation of every block-GEMM input and weight."""
    if fp8_act is not None:
        return _block_forward_fp8(sd, arch, i, x, [float(v) for v in fp8_act])
    p = f"blocks.{i}."
    B, N, D = x.shape
    h, hd = arch.heads, arch.head_dim
    y = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps=1e-6)
    qkv = F.linear(_q(y, quant), _q(sd[p + "attn.qkv.weight"], quant), sd[p + "attn.qkv.bias"])
    qkv = _q(qkv, quant).reshape(B, N, 3, h, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    attn = attn.softmax(dim=-1)
    if quant == "bf16":
        # the kernel rounds un-normalised probabilities exp(s - max) to bf16 before P@V and
        # divides by the fp32 row sum afterwards
        s = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
        e = torch.exp(s - s.amax(dim=-1, keepdim=True))
        o = (_q(e, quant) @ v) / e.sum(dim=-1, keepdim=True)
    else:
        o = attn @ v
    o = o.transpose(1, 2).reshape(B, N, D)
    o = F.linear(_q(o, quant), _q(sd[p + "attn.proj.weight"], quant), sd[p + "attn.proj.bias"])
    x = x + sd[p + "ls1.gamma"] * o
    y = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps=1e-6)
    if arch.ffn == "mlp":
        hdn = F.linear(_q(y, quant), _q(sd[p + "mlp.fc1.weight"], quant), sd[p + "mlp.fc1.bias"])
        hdn = F.gelu(hdn)  # exact erf GELU
        o = F.linear(_q(hdn, quant), _q(sd[p + "mlp.fc2.weight"], quant), sd[p + "mlp.fc2.bias"])
    else:
        x12 = F.linear(_q(y, quant), _q(sd[p + "mlp.w12.weight"], quant), sd[p + "mlp.w12.bias"])
        x1, x2 = x12.chunk(2, dim=-1)
        hdn = F.silu(x1) * x2
        o = F.linear(_q(hdn, quant), _q(sd[p + "mlp.w3.weight"], quant), sd[p + "mlp.w3.bias"])
    return x + sd[p + "ls2.gamma"] * o


def _block_forward_fp8(sd, arch: VitArch, i: int, x: torch.Tensor, s) -> torch.Tensor:
    """The device's fp8 block: the four GEMM inputs are quantised to e4m3 straight from the fp32 values the producing
    kernel holds (LayerNorm output, attention output o / l, GELU or SwiGLU output); qkv and the residual stream are
    bf16 / fp32 as in the bf16 mode."""
    p, q = f"blocks.{i}.", "bf16"
    B, N, D = x.shape
    h, hd = arch.heads, arch.head_dim
    y = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps=1e-6)
    qkv = _q(F.linear(_fq_act(y, s[0]), _fq_weight(sd[p + "attn.qkv.weight"]), sd[p + "attn.qkv.bias"]), q)
    qkv = qkv.reshape(B, N, 3, h, hd).permute(2, 0, 3, 1, 4)
    qq, k, v = qkv[0], qkv[1], qkv[2]
    sc = (qq @ k.transpose(-2, -1)) * (hd ** -0.5)
    e = torch.exp(sc - sc.amax(dim=-1, keepdim=True))
    o = ((_q(e, q) @ v) / e.sum(dim=-1, keepdim=True)).transpose(1, 2).reshape(B, N, D)
    x = x + sd[p + "ls1.gamma"] * F.linear(_fq_act(o, s[1]), _fq_weight(sd[p + "attn.proj.weight"]), sd[p + "attn.proj.bias"])
    y = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps=1e-6)
    if arch.ffn == "mlp":
        hdn = F.gelu(F.linear(_fq_act(y, s[2]), _fq_weight(sd[p + "mlp.fc1.weight"]), sd[p + "mlp.fc1.bias"]))
        o = F.linear(_fq_act(hdn, s[3]), _fq_weight(sd[p + "mlp.fc2.weight"]), sd[p + "mlp.fc2.bias"])
    else:
        x12 = F.linear(_fq_act(y, s[2]), _fq_weight(sd[p + "mlp.w12.weight"]), sd[p + "mlp.w12.bias"])
        x1, x2 = x12.chunk(2, dim=-1)
        o = F.linear(_fq_act(F.silu(x1) * x2, s[3]), _fq_weight(sd[p + "mlp.w3.weight"]), sd[p + "mlp.w3.bias"])
    return x + sd[p + "ls2.gamma"] * o


@torch.no_grad()
def hidden_after_block(sd, arch: VitArch, images: torch.Tensor, layer: int, quant=None, all_blocks=False, fp8_act=None, stride=None) -> torch.Tensor:
    """Output of blocks[layer] for [0,1] images (what the reference's forward hook captures).

    `all_blocks=True` keeps running to the last block like the reference does
    (dinov2_utils.py:257 runs the entire model) -- only used by the cpu_baseline timing.
    """
    x = embed_tokens(sd, arch, normalize_images(images), "bf16" if fp8_act is not None else quant, stride)
    out = None
    last = arch.depth - 1 if all_blocks else layer
    for i in range(last + 1):
        x = block_forward(sd, arch, i, x, quant, None if fp8_act is None else fp8_act[i])
        if i == layer:
            out = x
    return out


@torch.no_grad()
def facet_tokens(sd, arch: VitArch, images: torch.Tensor, layer: int, facet: str, quant=None) -> torch.Tensor:
    """key / query / value facet of blocks[layer] as the reference's attention hook delivers it (dinov2_utils.py:176-194,
    294-311): qkv(norm1(x)) -> [B, h, t, d] -> per token the vector indexed (d, head), i.e. [B, t, d * h + head]."""
    x = embed_tokens(sd, arch, normalize_images(images), quant)
    for i in range(layer):
        x = block_forward(sd, arch, i, x, quant)
    p = f"blocks.{layer}."
    B, N, D = x.shape
    y = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps=1e-6)
    qkv = _q(F.linear(_q(y, quant), _q(sd[p + "attn.qkv.weight"], quant), sd[p + "attn.qkv.bias"]), quant)
    f = qkv.reshape(B, N, 3, arch.heads, arch.head_dim)[:, :, {"query": 0, "key": 1, "value": 2}[facet]]  # [B, t, h, d]
    return f.permute(0, 1, 3, 2).reshape(B, N, D)


@torch.no_grad()
def extractor_forward(sd, arch: VitArch, images: torch.Tensor, layer: int, apply_norm: bool = True, quant=None, all_blocks=False, fp8_act=None,
                      facet: str = "token", stride=None):
    """-> {"cls_tokens": [B,D], "feature_maps": [B,D,Hp,Wp]} exactly like the reference wrapper."""
    B, _, H, W = images.shape
    if facet == "token":
        hs = hidden_after_block(sd, arch, images, layer, quant, all_blocks, fp8_act, stride)
    else:
        hs = facet_tokens(sd, arch, images, layer, facet, quant)
    cls, patch = hs[:, :1], hs[:, 1 + arch.registers:]
    if apply_norm:
        tok = F.layer_norm(torch.cat([cls, patch], 1), (arch.dim,), sd["norm.weight"], sd["norm.bias"], eps=1e-6)
        cls, patch = tok[:, :1], tok[:, 1:]
    st = arch.patch if stride is None else stride
    gh, gw = 1 + (H - arch.patch) // st, 1 + (W - arch.patch) // st   # dinov2_utils.py:266-269 (= H // patch at stride == patch size)
    fmap = patch.reshape(B, gh, gw, arch.dim).permute(0, 3, 1, 2)
    return {"cls_tokens": cls[:, 0], "feature_maps": fmap}


# ---- point utilities + sampling + PCA (utils/feature_util.py:25-131, projector_util.py:66-69)

def generate_grid_points(grid_size, cell_size: float = 1.0) -> torch.Tensor:
    cols, rows = int(grid_size[0] / cell_size), int(grid_size[1] / cell_size)
    half = cell_size / 2.0
    x = torch.linspace(half, grid_size[0] - half, cols, dtype=torch.float)
    y = torch.linspace(half, grid_size[1] - half, rows, dtype=torch.float)
    gx, gy = torch.meshgrid(x, y, indexing="xy")
    return torch.vstack((gx.flatten(), gy.flatten())).T


def filter_points_by_mask(points: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    pi = (points + 0.5).int()
    valid = (pi[:, 0] > 0) & (pi[:, 0] < mask.shape[1]) & (pi[:, 1] > 0) & (pi[:, 1] < mask.shape[0])
    pi = pi[valid]
    return points[valid][mask[pi[:, 1], pi[:, 0]].bool()]


def sample_feature_map_at_points(feature_map_chw: torch.Tensor, points: torch.Tensor, image_size) -> torch.Tensor:
    uv = torch.div(2.0, torch.as_tensor(image_size)) * points - 1.0
    feats = F.grid_sample(feature_map_chw.unsqueeze(0), uv.unsqueeze(0).unsqueeze(2), align_corners=False)
    return feats[0, :, :, 0].permute(1, 0)


def pca_transform(x: torch.Tensor, components: torch.Tensor, mean: torch.Tensor) -> torch.Tensor:
    """sklearn PCA.transform without whitening: X @ C^T - (mu @ C^T)."""
    return x @ components.T - (mean.reshape(1, -1) @ components.T)
