"""Imports the reference's own Python modules (TEST INFRASTRUCTURE, BUILD CONTAINER ONLY).

/root/reference is read-only public content that never travels to the GPU box, and
its native third-party deps are absent here (faiss, cv2, torchvision, kornia,
torchinfo, dinov2). This shim registers minimal stand-ins in `sys.modules`, then
imports `utils.{knn_util,template_util,corresp_util,repre_util,projector_util,
feature_util,dinov2_utils}` unmodified, so `oracle/make_golden.py` can run the
reference's logic on synthetic inputs and freeze the results as fixtures.

Stand-ins and what they do NOT pin:
  faiss ........ brute-force IndexFlatL2/IndexFlatIP in torch (fp32 `|x|^2+|y|^2-2xy`,
                 ties -> lowest index). faiss's own low-order bits are unpinned.
  dinov2 ....... backbone adapter over transformers' Dinov2WithRegistersModel carrying the
                 same weights (an independent implementation of the architecture).
  cv2 .......... three interpolation constants (default args in utils/misc.py).
  torchvision .. transforms.Normalize as (x-mean)/std.
  kornia, torchinfo: empty modules (imported, unused on this path).
"""

import os
import sys
import types

import torch

REF_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "utils"))


class _IndexFlat:
    def __init__(self, d: int, metric: str):
        self.d, self.metric, self.data = d, metric, torch.zeros(0, d)

    def train(self, x):
        pass

    def add(self, x):
        self.data = torch.cat([self.data, x.detach().cpu().float()], 0)

    def search(self, q, k):
        q = q.detach().cpu().float()
        if self.metric == "l2":
            d = (q * q).sum(1, keepdim=True) + (self.data * self.data).sum(1)[None] - 2.0 * (q @ self.data.T)
            d = d.clamp_min(0)
            order = torch.argsort(d, dim=1, stable=True)[:, :k]
            return torch.gather(d, 1, order), order
        s = q @ self.data.T
        order = torch.argsort(-s, dim=1, stable=True)[:, :k]
        return torch.gather(s, 1, order), order


def _install_standins(backbone_factory=None) -> None:
    sys.dont_write_bytecode = True
    import transformers  # noqa: F401  (must be imported before a fake torchvision appears)

    faiss = types.ModuleType("faiss")
    faiss.IndexFlatL2 = lambda d: _IndexFlat(d, "l2")
    faiss.IndexFlatIP = lambda d: _IndexFlat(d, "ip")
    contrib = types.ModuleType("faiss.contrib")
    tu = types.ModuleType("faiss.contrib.torch_utils")
    faiss.contrib, contrib.torch_utils = contrib, tu
    sys.modules.update({"faiss": faiss, "faiss.contrib": contrib, "faiss.contrib.torch_utils": tu})

    cv2 = types.ModuleType("cv2")
    cv2.INTER_NEAREST, cv2.INTER_LINEAR, cv2.INTER_AREA = 0, 1, 3
    sys.modules["cv2"] = cv2

    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class Normalize(torch.nn.Module):
        def __init__(self, mean, std):
            super().__init__()
            self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

        def forward(self, x):
            return (x - self.mean.to(x.device)) / self.std.to(x.device)

    tvt.Normalize = Normalize
    tv.transforms = tvt
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt})

    for name in ("kornia", "torchinfo"):
        mod = types.ModuleType(name)
        if name == "torchinfo":
            mod.summary = lambda *a, **k: None
        sys.modules[name] = mod

    dinov2 = types.ModuleType("dinov2")
    hub = types.ModuleType("dinov2.hub")
    backbones = types.ModuleType("dinov2.hub.backbones")
    dinov2.hub, hub.backbones = hub, backbones
    sys.modules.update({"dinov2": dinov2, "dinov2.hub": hub, "dinov2.hub.backbones": backbones})


def set_backbone(model_base_name: str, factory) -> None:
    """Register `factory(pretrained=True) -> nn.Module` as dinov2.hub.backbones.<name>."""
    sys.modules["dinov2.hub.backbones"].__dict__[model_base_name] = factory


_imported = None


def import_reference():
    """-> namespace with the reference modules (knn_util, template_util, ...)."""
    global _imported
    if _imported is not None:
        return _imported
    if not reference_available():
        raise RuntimeError("/root/reference is not present (GPU box?) -- fixtures are generated in the build container only")
    _install_standins()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import importlib

    ns = types.SimpleNamespace()
    for name in ("knn_util", "template_util", "corresp_util", "repre_util", "projector_util", "feature_util", "dinov2_utils", "misc"):
        setattr(ns, name, importlib.import_module(f"utils.{name}"))
    _imported = ns
    return ns


class HFBackboneAdapter(torch.nn.Module):
    """Exposes the attribute surface the reference wrapper touches (dinov2_utils.py:97-98,
    140, 206-211, 257, 304) on top of transformers' Dinov2WithRegistersModel."""

    def __init__(self, sd, arch, image_size: int, pos_fn=None):
        """pos_fn(tokens, height, width): replaces the stand-in's position-encoding interpolation.  The non-register hub entries
        (`dinov2_vit{s,b,l,g}14`) interpolate in scale-factor mode with the +0.1 offset and no antialias -- transformers' Dinov2Model
        interpolates by size -- so for those the fixtures install the REFERENCE's own `_fix_pos_enc(patch, (patch, patch))`
        (dinov2_utils.py:325-360), which is that call argument for argument."""
        super().__init__()
        from transformers import Dinov2Config, Dinov2Model, Dinov2WithRegistersConfig, Dinov2WithRegistersModel

        common = dict(hidden_size=arch.dim, num_hidden_layers=arch.depth, num_attention_heads=arch.heads,
                      mlp_ratio=4, image_size=image_size, patch_size=arch.patch, layer_norm_eps=1e-6,
                      use_swiglu_ffn=(arch.ffn == "swiglu"), hidden_act="gelu")
        if arch.registers > 0:
            hf = Dinov2WithRegistersModel(Dinov2WithRegistersConfig(num_register_tokens=arch.registers, **common)).eval()
        else:
            hf = Dinov2Model(Dinov2Config(**common)).eval()
        D = arch.dim
        m = {
            "embeddings.cls_token": sd["cls_token"],
            "embeddings.mask_token": sd["mask_token"],
            "embeddings.position_embeddings": sd["pos_embed"],
            "embeddings.patch_embeddings.projection.weight": sd["patch_embed.proj.weight"],
            "embeddings.patch_embeddings.projection.bias": sd["patch_embed.proj.bias"],
            "layernorm.weight": sd["norm.weight"],
            "layernorm.bias": sd["norm.bias"],
        }
        if arch.registers > 0:
            m["embeddings.register_tokens"] = sd["register_tokens"]
        for i in range(arch.depth):
            s, t = f"blocks.{i}.", f"encoder.layer.{i}."
            qw, qb = sd[s + "attn.qkv.weight"], sd[s + "attn.qkv.bias"]
            for j, nm in enumerate(("query", "key", "value")):
                m[t + f"attention.attention.{nm}.weight"] = qw[j * D:(j + 1) * D]
                m[t + f"attention.attention.{nm}.bias"] = qb[j * D:(j + 1) * D]
            m[t + "attention.output.dense.weight"] = sd[s + "attn.proj.weight"]
            m[t + "attention.output.dense.bias"] = sd[s + "attn.proj.bias"]
            m[t + "layer_scale1.lambda1"] = sd[s + "ls1.gamma"]
            m[t + "layer_scale2.lambda1"] = sd[s + "ls2.gamma"]
            for nm in ("norm1", "norm2"):
                m[t + nm + ".weight"] = sd[s + nm + ".weight"]
                m[t + nm + ".bias"] = sd[s + nm + ".bias"]
            if arch.ffn == "mlp":
                for nm in ("fc1", "fc2"):
                    m[t + f"mlp.{nm}.weight"] = sd[s + f"mlp.{nm}.weight"]
                    m[t + f"mlp.{nm}.bias"] = sd[s + f"mlp.{nm}.bias"]
            else:
                m[t + "mlp.weights_in.weight"] = sd[s + "mlp.w12.weight"]
                m[t + "mlp.weights_in.bias"] = sd[s + "mlp.w12.bias"]
                m[t + "mlp.weights_out.weight"] = sd[s + "mlp.w3.weight"]
                m[t + "mlp.weights_out.bias"] = sd[s + "mlp.w3.bias"]
        missing, unexpected = hf.load_state_dict(m, strict=True)
        # attribute surface of upstream blocks that the key / query / value facet hooks touch (dinov2_utils.py:184-194,
        # 206-214): block.attn is called with norm1(x), has .qkv (one fused projection) and .num_heads
        for layer in hf.encoder.layer:
            att = layer.attention
            inner = att.attention
            att.qkv = (lambda x, a=inner: torch.cat([a.query(x), a.key(x), a.value(x)], dim=-1))
            att.num_heads = arch.heads
            layer.attn = att
        if pos_fn is not None:
            hf.embeddings.interpolate_pos_encoding = lambda emb, height, width: pos_fn(emb, height, width)
        self.hf = hf
        self.blocks = hf.encoder.layer
        self.norm = hf.layernorm
        self.num_register_tokens = arch.registers
        pe = types.SimpleNamespace()
        pe.patch_size = (arch.patch, arch.patch)
        pe.proj = hf.embeddings.patch_embeddings.projection
        self.patch_embed = pe

    def forward(self, x):
        return self.hf(x).last_hidden_state
