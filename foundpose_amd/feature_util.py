"""Extractor factory, grid points, mask filtering and feature sampling with the reference's signatures
(/root/reference/utils/feature_util.py:18-131). Point generation/filtering is index arithmetic on tiny
tensors (torch ops on whatever device the inputs live on); sampling is a HIP kernel.
"""

from typing import Tuple

import torch

from . import dinov2_utils, ops


def make_feature_extractor(model_name: str, **kwargs) -> torch.nn.Module:
    if model_name.startswith("dinov2_"):
        return dinov2_utils.DinoFeatureExtractor(model_name=model_name, **kwargs)
    raise NotImplementedError(model_name)


def generate_grid_points(grid_size: Tuple[int, int], cell_size: float = 1.0) -> torch.Tensor:
    """Centres of the cells of a regular grid, row-major with x fastest -> [num_points, 2]."""
    cols = int(grid_size[0] / cell_size)
    rows = int(grid_size[1] / cell_size)
    half = cell_size / 2.0
    xs = torch.linspace(half, grid_size[0] - half, cols, dtype=torch.float)
    ys = torch.linspace(half, grid_size[1] - half, rows, dtype=torch.float)
    gx, gy = torch.meshgrid(xs, ys, indexing="xy")
    return torch.stack((gx.reshape(-1), gy.reshape(-1)), dim=1)


def filter_points_by_box(points: torch.Tensor, box: Tuple[float, float, float, float]) -> Tuple[torch.Tensor, torch.Tensor]:
    x1, y1, x2, y2 = box
    valid = (points[:, 0] > x1) & (points[:, 0] < x2) & (points[:, 1] > y1) & (points[:, 1] < y2)
    return points[valid], valid


def filter_points_by_mask(points: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """Keeps the points whose pixel (after +0.5 and truncation) lies strictly inside the canvas and on the mask."""
    pix = (points + 0.5).int()
    pix, valid = filter_points_by_box(pix, (0, 0, mask.shape[1], mask.shape[0]))
    on_mask = mask[pix[:, 1].long(), pix[:, 0].long()].bool()
    return points[valid][on_mask]


def sample_feature_map_at_points(feature_map_chw: torch.Tensor, points: torch.Tensor, image_size: Tuple[int, int]) -> torch.Tensor:
    """Bilinear sampling (grid_sample semantics: zeros padding, align_corners=False) -> [num_points, C].

    `feature_map_chw` may be any strided view (the extractor returns a permuted view of its token-major
    output, like the reference does); it is read in place.
    """
    fmap = feature_map_chw.to("cuda") if not feature_map_chw.is_cuda else feature_map_chw
    pts = points.to("cuda") if not points.is_cuda else points
    out = ops.sample_bilinear(fmap.unsqueeze(0), pts, None, image_size)
    return out if points.is_cuda else out.to(points.device)
