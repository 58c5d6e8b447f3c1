"""Seeded synthetic inputs (no network here: no checkpoints, no BOP data).

Shapes follow SURVEY.md section 8(d): random-init ViT weights in the upstream
DINOv2 state_dict key layout, [0,1] crops, disc masks, and a planted-positive
template bank laid out like the reference's `repre.pth`
(/root/reference/utils/repre_util.py:34-83, gen_repre.py:187-214).

Everything here is host-side torch-CPU data generation; nothing computes the
hot path.
"""

import math
from typing import Dict, Optional, Tuple

import torch

from .vit_config import VitArch


def make_vit_state_dict(arch: VitArch, seed: int = 1234, ls_gamma: float = 1.0) -> Dict[str, torch.Tensor]:
    """Random DINOv2 weights with the upstream key names (fp32, CPU)."""
    g = torch.Generator().manual_seed(seed)

    def tn(*shape, std=0.02):
        t = torch.empty(*shape, dtype=torch.float32)
        torch.nn.init.trunc_normal_(t, std=std, a=-2 * std, b=2 * std, generator=g)
        return t

    D, R = arch.dim, arch.registers
    sd: Dict[str, torch.Tensor] = {}
    sd["cls_token"] = tn(1, 1, D)
    sd["pos_embed"] = tn(1, 1 + arch.pretrain_grid ** 2, D)
    if R:
        sd["register_tokens"] = tn(1, R, D)
    sd["mask_token"] = torch.zeros(1, D)
    sd["patch_embed.proj.weight"] = tn(D, 3, arch.patch, arch.patch)
    sd["patch_embed.proj.bias"] = tn(D)
    for i in range(arch.depth):
        p = f"blocks.{i}."
        # LayerNorm affine slightly off identity so that a missing gamma/beta shows up.
        sd[p + "norm1.weight"] = 1.0 + tn(D, std=0.05)
        sd[p + "norm1.bias"] = tn(D, std=0.05)
        sd[p + "attn.qkv.weight"] = tn(3 * D, D)
        sd[p + "attn.qkv.bias"] = tn(3 * D)
        sd[p + "attn.proj.weight"] = tn(D, D)
        sd[p + "attn.proj.bias"] = tn(D)
        sd[p + "ls1.gamma"] = ls_gamma * (1.0 + tn(D, std=0.05))
        sd[p + "norm2.weight"] = 1.0 + tn(D, std=0.05)
        sd[p + "norm2.bias"] = tn(D, std=0.05)
        if arch.ffn == "mlp":
            sd[p + "mlp.fc1.weight"] = tn(arch.hidden, D)
            sd[p + "mlp.fc1.bias"] = tn(arch.hidden)
            sd[p + "mlp.fc2.weight"] = tn(D, arch.hidden)
            sd[p + "mlp.fc2.bias"] = tn(D)
        else:
            sd[p + "mlp.w12.weight"] = tn(2 * arch.hidden, D)
            sd[p + "mlp.w12.bias"] = tn(2 * arch.hidden)
            sd[p + "mlp.w3.weight"] = tn(D, arch.hidden)
            sd[p + "mlp.w3.bias"] = tn(D)
        sd[p + "ls2.gamma"] = ls_gamma * (1.0 + tn(D, std=0.05))
    sd["norm.weight"] = 1.0 + tn(D, std=0.05)
    sd["norm.bias"] = tn(D, std=0.05)
    return sd


def make_crops(batch: int, size: int, seed: int = 0) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.rand(batch, 3, size, size, generator=g, dtype=torch.float32)


def make_dictionary_crops(batch: int, size: int, mask: torch.Tensor, dict_size: int = 682, patch: int = 14, seed: int = 0,
                          pixel_noise: float = 0.02, dict_seed: int = 4242) -> Tuple[torch.Tensor, torch.Tensor]:
    """Crops assembled from a dictionary of `dict_size` noise patch textures: the patches whose centre lies inside `mask`
    are DISTINCT textures (a random draw without replacement), the others random ones, plus a little pixel noise.
    Real object surfaces repeat local structure across views -- that repetition is what gives descriptors clusters for
    the visual words to sit on; iid noise crops have none (every patch is equidistant from every other, and the 3
    nearest of 2048 words are decided by rounding).
    -> (crops [B, 3, S, S] f32 in [0, 1], texture id of every patch [B, S/patch, S/patch] i64)."""
    g = torch.Generator().manual_seed(dict_seed)
    n = size // patch
    textures = torch.rand(dict_size, 3, patch, patch, generator=g, dtype=torch.float32)
    inside = mask[patch // 2::patch, patch // 2::patch][:n, :n].bool().reshape(-1)
    n_in = int(inside.sum())
    if n_in > dict_size:
        raise ValueError(f"{n_in} patches inside the mask need a dictionary of at least that many textures (got {dict_size})")
    g = torch.Generator().manual_seed(seed)
    crops = torch.full((batch, 3, size, size), 0.5, dtype=torch.float32)
    ids = torch.empty(batch, n * n, dtype=torch.int64)
    for b in range(batch):
        ids[b] = torch.randint(0, dict_size, (n * n,), generator=g)
        ids[b, inside] = torch.randperm(dict_size, generator=g)[:n_in]
        crops[b, :, : n * patch, : n * patch] = textures[ids[b]].reshape(n, n, 3, patch, patch).permute(2, 0, 3, 1, 4).reshape(3, n * patch, n * patch)
    crops += pixel_noise * torch.randn(crops.shape, generator=g)
    return crops.clamp_(0.0, 1.0), ids.reshape(batch, n, n)


def make_disc_mask(size: int, rel_radius: float = 0.35) -> torch.Tensor:
    """uint8 [S,S]: centred disc of radius rel_radius*S (Q ~ 0.385*Np grid points)."""
    ys, xs = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    c = (size - 1) / 2.0
    return (((xs - c) ** 2 + (ys - c) ** 2) <= (rel_radius * size) ** 2).to(torch.uint8)


def make_bank_features(
    num_templates: int,
    feat_dim: int = 256,
    min_patches: int = 300,
    max_patches: int = 450,
    seed: int = 7,
) -> Dict[str, torch.Tensor]:
    """Raw bank content: CSR-sorted features, vertices and the template id run per feature."""
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(min_patches, max_patches + 1, (num_templates,), generator=g)
    n_f = int(counts.sum())
    # PCA-like decaying spectrum, sigma_j ~ j^-0.5
    sigma = (torch.arange(1, feat_dim + 1, dtype=torch.float32)) ** -0.5
    feat_vectors = torch.randn(n_f, feat_dim, generator=g) * sigma
    vertices = torch.randn(n_f, 3, generator=g) * 50.0
    feat_to_template_ids = torch.repeat_interleave(
        torch.arange(num_templates, dtype=torch.int32), counts
    )
    return {
        "feat_vectors": feat_vectors.contiguous(),
        "vertices": vertices.contiguous(),
        "feat_to_template_ids": feat_to_template_ids,
        "feat_to_vertex_ids": torch.arange(n_f, dtype=torch.int32),
        "template_counts": counts,
    }


def pick_centroids(feat_vectors: torch.Tensor, num_words: int, seed: int = 11) -> torch.Tensor:
    """Visual words = random bank rows (stand-in for the offline k-means)."""
    g = torch.Generator().manual_seed(seed)
    n = feat_vectors.shape[0]
    if n >= num_words:
        ids = torch.randperm(n, generator=g)[:num_words]
        return feat_vectors[ids].clone().contiguous()
    extra = torch.randn(num_words - n, feat_vectors.shape[1], generator=g) * feat_vectors.std(0)
    return torch.cat([feat_vectors, extra], 0).contiguous()


def make_planted_query(
    bank: Dict[str, torch.Tensor],
    template_id: int,
    num_grid: int,
    seed: int,
    noise: float = 0.05,
    cell: float = 14.0,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Query = noisy copy of one template's patch features at distinct grid cells.

    Returns (query_points [Q,2] f32 at cell centres, query_features [Q,d] f32).
    """
    g = torch.Generator().manual_seed(seed)
    ids = torch.nonzero(bank["feat_to_template_ids"] == template_id).flatten()
    q = min(len(ids), num_grid * num_grid)
    feats = bank["feat_vectors"][ids[:q]]
    feats = feats + noise * torch.randn(feats.shape, generator=g)
    cells = torch.randperm(num_grid * num_grid, generator=g)[:q].sort().values
    xs = (cells % num_grid).float() * cell + cell / 2
    ys = (cells // num_grid).float() * cell + cell / 2
    return torch.stack([xs, ys], 1).contiguous(), feats.contiguous()
