"""Where the DINOv2 backbone's weights come from, and the strict check every checkpoint goes through.

The reference builds its backbone with `dinov2.hub.backbones.<model_base_name>(pretrained=True)`
(/root/reference/utils/dinov2_utils.py:81-84), i.e. it loads the upstream hub checkpoint
`<model_base_name with _reg -> _reg4>_pretrain.pth` into the upstream module with `load_state_dict(strict=True)` semantics
(torch.hub caches the file under `<torch hub dir>/checkpoints/`).  There is no network on the MI355X boxes, so the checkpoint has to be
on disk -- and the extractor must never run on anything else without being told to:

    state_dict=<dict>                      an upstream-layout state dict, already in memory
    weights=<file | directory>             a checkpoint file, or a directory holding the upstream file name
    $FOUNDPOSE_DINOV2_WEIGHTS              the same (file or directory), from the environment
    <torch hub dir>/checkpoints/           where the reference's own `pretrained=True` call leaves the file
    random_init_seed=<int>                 seeded random weights -- tests and benchmarks only, always explicit

Anything else raises FoundPoseWeightsError.  Host-side Python only: nothing here touches the device.
"""

import os
from typing import Dict, List, Optional, Tuple

import torch

from .vit_config import ARCHS, VitArch

ENV_VAR = "FOUNDPOSE_DINOV2_WEIGHTS"
# keys of the upstream checkpoints that the forward pass never reads (the mask token only enters masked-image training)
IGNORED_KEYS = ("mask_token",)


class FoundPoseWeightsError(RuntimeError):
    """No checkpoint, an unreadable one, or one that does not fit the architecture the extractor name asks for."""


def checkpoint_file_names(model_base_name: str) -> List[str]:
    """File names a checkpoint of `model_base_name` (e.g. dinov2_vitl14, dinov2_vitl14_reg) is looked up under, the upstream hub name first
    (dinov2_vitl14_pretrain.pth, dinov2_vitl14_reg4_pretrain.pth)."""
    names = []
    if model_base_name.endswith("_reg"):
        stem = model_base_name[: -len("_reg")]
        names += [f"{stem}_reg4_pretrain.pth", f"{stem}_reg4.pth"]
    else:
        names += [f"{model_base_name}_pretrain.pth"]
    names += [f"{model_base_name}.pth"]
    return names


def expected_shapes(arch: VitArch) -> Dict[str, Tuple[int, ...]]:
    """Every parameter the forward pass reads, with the shape the upstream module of this architecture declares."""
    D, R, H = arch.dim, arch.registers, arch.hidden
    s: Dict[str, Tuple[int, ...]] = {
        "cls_token": (1, 1, D),
        "pos_embed": (1, 1 + arch.pretrain_grid ** 2, D),
        "patch_embed.proj.weight": (D, 3, arch.patch, arch.patch),
        "patch_embed.proj.bias": (D,),
        "norm.weight": (D,),
        "norm.bias": (D,),
    }
    if R:
        s["register_tokens"] = (1, R, D)
    for i in range(arch.depth):
        p = f"blocks.{i}."
        s.update({p + "norm1.weight": (D,), p + "norm1.bias": (D,), p + "attn.qkv.weight": (3 * D, D), p + "attn.qkv.bias": (3 * D,),
                  p + "attn.proj.weight": (D, D), p + "attn.proj.bias": (D,), p + "ls1.gamma": (D,),
                  p + "norm2.weight": (D,), p + "norm2.bias": (D,), p + "ls2.gamma": (D,)})
        if arch.ffn == "mlp":
            s.update({p + "mlp.fc1.weight": (H, D), p + "mlp.fc1.bias": (H,), p + "mlp.fc2.weight": (D, H), p + "mlp.fc2.bias": (D,)})
        else:
            s.update({p + "mlp.w12.weight": (2 * H, D), p + "mlp.w12.bias": (2 * H,), p + "mlp.w3.weight": (D, H), p + "mlp.w3.bias": (D,)})
    return s


def describe_checkpoint(sd: Dict[str, torch.Tensor]) -> str:
    """What architecture a state dict looks like (for the error message of a mismatch)."""
    try:
        dim = int(sd["cls_token"].shape[-1])
        blocks = {int(k.split(".")[1]) for k in sd if k.startswith("blocks.") and k.split(".")[1].isdigit()}
        depth = max(blocks) + 1 if blocks else 0
        regs = int(sd["register_tokens"].shape[1]) if "register_tokens" in sd else 0
        ffn = "swiglu" if any(".mlp.w12." in k for k in sd) else "mlp"
        patch = int(sd["patch_embed.proj.weight"].shape[-1]) if "patch_embed.proj.weight" in sd else -1
        match = [n for n, a in ARCHS.items() if (a.dim, a.depth, a.registers, a.ffn, a.patch) == (dim, depth, regs, ffn, patch)]
        looks = f"dim {dim}, {depth} blocks, {regs} register tokens, {ffn} ffn, patch {patch}"
        return looks + (f" = {match[0]}" if match else " (no DINOv2 hub architecture)")
    except Exception:  # not even a ViT-shaped dict
        return "not a DINOv2 state dict (no cls_token)"


def validate_state_dict(sd, arch: VitArch, source: str = "state_dict") -> Dict[str, torch.Tensor]:
    """`load_state_dict(strict=True)` as a function: every expected key present with the expected shape, no unexpected key except the
    ones the forward never reads, floating-point tensors without NaN / Inf.  -> the dict restricted to the expected keys."""
    if not isinstance(sd, dict) or not all(isinstance(k, str) for k in sd):
        raise FoundPoseWeightsError(f"{source}: expected a state dict (str -> tensor), got {type(sd).__name__}")
    want = expected_shapes(arch)
    missing = [k for k in want if k not in sd]
    unexpected = [k for k in sd if k not in want and k not in IGNORED_KEYS]
    wrong = []
    for k, shape in want.items():
        if k in sd:
            t = sd[k]
            if not isinstance(t, torch.Tensor):
                wrong.append(f"{k}: {type(t).__name__} is not a tensor")
            elif tuple(t.shape) != shape:
                wrong.append(f"size mismatch for {k}: checkpoint {tuple(t.shape)}, {arch.name} expects {shape}")
            elif not t.dtype.is_floating_point:
                wrong.append(f"{k}: dtype {t.dtype} is not floating point")
    if missing or unexpected or wrong:
        def some(keys):
            return ", ".join(keys[:6]) + (f", ... ({len(keys)} in all)" if len(keys) > 6 else "")
        parts = [f"{source} does not fit dinov2_{arch.name.replace('-', '_')} ({arch.dim} channels, {arch.depth} blocks, "
                 f"{arch.registers} register tokens, {arch.ffn} ffn); the checkpoint looks like: {describe_checkpoint(sd)}"]
        if missing:
            parts.append("Missing key(s): " + some(missing))
        if unexpected:
            parts.append("Unexpected key(s): " + some(unexpected))
        if wrong:
            parts.append("; ".join(wrong[:6]) + (f"; ... ({len(wrong)} in all)" if len(wrong) > 6 else ""))
        raise FoundPoseWeightsError(". ".join(parts))
    out = {k: sd[k] for k in want}
    bad = [k for k, t in out.items() if not bool(torch.isfinite(t.float()).all())]
    if bad:
        raise FoundPoseWeightsError(f"{source}: non-finite values in {', '.join(bad[:6])}")
    return out


def load_checkpoint_file(path: str) -> Dict[str, torch.Tensor]:
    """A `.pth` as upstream publishes it (a flat state dict); one level of {"state_dict" | "model": ...} wrapping is unwrapped.
    Tensors only (weights_only=True): a checkpoint is data, never code."""
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:
        raise FoundPoseWeightsError(f"cannot read checkpoint {path}: {type(e).__name__}: {e}") from e
    if isinstance(obj, dict):
        for k in ("state_dict", "model"):
            if k in obj and isinstance(obj[k], dict) and "cls_token" not in obj:
                return obj[k]
    return obj


def find_checkpoint(model_base_name: str, weights: Optional[str] = None) -> Tuple[Optional[str], List[str]]:
    """-> (path or None, the places looked at, in order).  An explicit `weights` that does not resolve raises instead of falling through
    to the next source: a typo in a path must not silently select another file."""
    names = checkpoint_file_names(model_base_name)
    looked: List[str] = []

    def in_location(loc: str, what: str) -> Optional[str]:
        if os.path.isfile(loc):
            return loc
        if os.path.isdir(loc):
            for n in names:
                looked.append(os.path.join(loc, n))
                if os.path.isfile(looked[-1]):
                    return looked[-1]
            raise FoundPoseWeightsError(f"{what}={loc}: the directory holds none of {names}")
        raise FoundPoseWeightsError(f"{what}={loc}: no such file or directory")

    if weights is not None:
        return in_location(os.fspath(weights), "weights"), looked
    env = os.environ.get(ENV_VAR)
    if env:
        return in_location(env, "$" + ENV_VAR), looked
    hub = os.path.join(torch.hub.get_dir(), "checkpoints")
    for n in names[:1]:  # only the upstream file name: that is what the reference's pretrained=True call caches
        looked.append(os.path.join(hub, n))
        if os.path.isfile(looked[-1]):
            return looked[-1], looked
    return None, looked


def resolve(model_base_name: str, arch: VitArch, state_dict=None, weights: Optional[str] = None,
            random_init_seed: Optional[int] = None) -> Tuple[Dict[str, torch.Tensor], str]:
    """-> (validated state dict, where it came from).  Exactly one source; see the module docstring for the order."""
    given = [n for n, v in (("state_dict", state_dict), ("weights", weights), ("random_init_seed", random_init_seed)) if v is not None]
    if len(given) > 1:
        raise ValueError(f"give one of state_dict=, weights=, random_init_seed= (got {', '.join(given)})")
    if state_dict is not None:
        return validate_state_dict(state_dict, arch, "state_dict"), "state_dict"
    if random_init_seed is not None:
        from . import synthetic
        if isinstance(random_init_seed, bool) or not isinstance(random_init_seed, int):
            raise TypeError("random_init_seed must be an int")
        return validate_state_dict(synthetic.make_vit_state_dict(arch, random_init_seed), arch, "random init"), f"random_init_seed={random_init_seed}"
    path, looked = find_checkpoint(model_base_name, weights)
    if path is None:
        raise FoundPoseWeightsError(
            f"no DINOv2 checkpoint for {model_base_name}: the reference loads the pretrained hub model (pretrained=True, utils/dinov2_utils.py:81-84); "
            f"this build reads it from disk and found none of: {', '.join(looked)}. Pass weights=<file or directory> (CLI: --weights), a state_dict=, "
            f"or set ${ENV_VAR}; random weights only with an explicit random_init_seed=<int>.")
    return validate_state_dict(load_checkpoint_file(path), arch, path), path


def expected_subset(sd: Dict[str, torch.Tensor], arch: VitArch) -> Dict[str, torch.Tensor]:
    """The entries of `sd` the forward pass reads (drops mask_token)."""
    return {k: sd[k] for k in expected_shapes(arch)}
