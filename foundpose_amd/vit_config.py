"""DINOv2 ViT architecture table + extractor-name grammar.

The reference builds its backbone through `dinov2.hub.backbones.<name>` (an
un-vendored submodule; call site /root/reference/utils/dinov2_utils.py:81-84)
and parses the extractor name at dinov2_utils.py:60-78. This module restates
the architecture constants of the published DINOv2 backbones and the name
grammar, so the MI355X path can be configured from the same strings.
"""

from dataclasses import dataclass
from typing import Dict


@dataclass(frozen=True)
class VitArch:
    name: str  # e.g. "vitl14-reg"
    dim: int
    depth: int
    heads: int
    ffn: str  # "mlp" (GELU) | "swiglu"
    hidden: int  # MLP hidden width (fc1 out; for swiglu the gated width)
    registers: int
    patch: int = 14
    pretrain_grid: int = 37  # pos_embed is a 37x37 table (518 / 14)
    # pos-embed interpolation flavour of the hub entry point (only used when the
    # crop is not 518x518): the *-reg hub models use antialias + offset 0.
    interp_antialias: bool = False
    interp_offset: float = 0.1

    @property
    def head_dim(self) -> int:
        return self.dim // self.heads


def _swiglu_hidden(dim: int) -> int:
    # upstream SwiGLUFFNFused: hidden = (int(4*dim*2/3) + 7) // 8 * 8
    return (int(4 * dim * 2 / 3) + 7) // 8 * 8


def _mk(name: str, dim: int, depth: int, heads: int, ffn: str) -> Dict[str, VitArch]:
    hidden = 4 * dim if ffn == "mlp" else _swiglu_hidden(dim)
    return {
        name: VitArch(name, dim, depth, heads, ffn, hidden, 0),
        name + "-reg": VitArch(
            name + "-reg", dim, depth, heads, ffn, hidden, 4,
            interp_antialias=True, interp_offset=0.0,
        ),
    }


ARCHS: Dict[str, VitArch] = {}
ARCHS.update(_mk("vits14", 384, 12, 6, "mlp"))
ARCHS.update(_mk("vitb14", 768, 12, 12, "mlp"))
ARCHS.update(_mk("vitl14", 1024, 24, 16, "mlp"))
ARCHS.update(_mk("vitg14", 1536, 40, 24, "swiglu"))


@dataclass
class ExtractorSpec:
    """Parsed extractor name (same grammar and defaults as the reference)."""

    version: str = "vits14-reg"
    stride: int = 14
    facet: str = "token"
    layer: int = 9
    apply_norm: bool = True

    @property
    def arch(self) -> VitArch:
        return ARCHS[self.version]


def parse_extractor_name(model_name: str) -> ExtractorSpec:
    """`dinov2_<version>` or `dinov2_version=..._stride=..._facet=..._layer=..._norm=...`.

    Unknown keys (e.g. `logbin`) are ignored, exactly as the reference does
    (dinov2_utils.py:67-78); the short form keeps layer=9 (dinov2_utils.py:62-64).
    """
    items = model_name.split("_")
    if items[0] != "dinov2":
        raise AssertionError(f"not a dinov2 extractor name: {model_name}")
    spec = ExtractorSpec()
    if len(items) == 2:
        spec.version = items[1]
    else:
        for item in items[1:]:
            key, value = item.split("=")
            if key == "version":
                spec.version = value
            elif key == "stride":
                spec.stride = int(value)
            elif key == "facet":
                spec.facet = value
            elif key == "layer":
                spec.layer = int(value)
            elif key == "norm":
                spec.apply_norm = bool(int(value))
    if spec.version not in ARCHS:
        raise KeyError(f"unknown DINOv2 version '{spec.version}'")
    return spec
