"""Extractor factory, grid points, mask filtering, feature sampling and 3D registration of template features with the
reference's signatures (/root/reference/utils/feature_util.py:18-237). Point generation/filtering is index arithmetic on tiny
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
    """Keeps the points whose pixel (after +0.5 and truncation) lies strictly inside the canvas and on the mask (feature_util.py:29-41 in the reference).
    Device tensors take fp_query_select (two small launches + ONE host wait for the count -- the tensor-indexing form below costs ~12 launches and two
    host waits, a quarter of a millisecond of the per-detection loop, scripts/infer.py:478); same points, same order."""
    if points.is_cuda and mask.is_cuda and points.dim() == 2 and points.shape[1] == 2 and mask.dim() == 2 and points.dtype == torch.float32 and points.shape[0] > 0:
        from ._lib import call, ptr, stream
        G, (H, W) = points.shape[0], mask.shape
        pts = points.contiguous()
        pix = (pts + 0.5).int()
        pix_x, pix_y = pix[:, 0].contiguous(), pix[:, 1].contiguous()
        m8 = mask if mask.dtype == torch.uint8 else (mask.view(torch.uint8) if mask.dtype == torch.bool else (mask != 0).to(torch.uint8))
        cnt = torch.empty(1, dtype=torch.int32, device=pts.device)
        out_pts = torch.empty(G, 2, dtype=torch.float32, device=pts.device)
        out_img = torch.empty(G, dtype=torch.int32, device=pts.device)
        scratch = torch.empty(G, dtype=torch.int32, device=pts.device)
        call("fp_query_select", ptr(m8.contiguous()), 1, H, W, ptr(pix_x), ptr(pix_y), ptr(pts), G, None, 0, 0, ptr(scratch), ptr(cnt), ptr(out_pts), ptr(out_img),
             None, None, None, None, stream())
        return out_pts[:int(cnt.item())]
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


def lift_2d_points_to_3d(points: torch.Tensor, depth_image: torch.Tensor, camera_model) -> torch.Tensor:
    """Pixel -> camera-space 3D point from a depth image (feature_util.py:134-159): the ray ((p - c), f_mean) scaled so
    that its z equals the depth at the pixel containing p.  [n, 2], [H, W] -> [n, 3] float32."""
    device = points.device
    focal = 0.5 * (camera_model.f[0] + camera_model.f[1])
    c = torch.as_tensor(camera_model.c).to(torch.float32).to(device)
    rays = torch.hstack([points - c, focal * torch.ones(points.shape[0], 1).to(torch.float32).to(device)])
    depths = depth_image[torch.floor(points[:, 1]).to(torch.int32), torch.floor(points[:, 0]).to(torch.int32)].reshape(-1, 1)
    rays *= depths / rays[:, 2].reshape(-1, 1)
    return rays


def erode_mask(mask: torch.Tensor, size: int = 5) -> torch.Tensor:
    """Binary erosion with a size x size box, pixels outside the image not counting against a border pixel -- what
    kornia.morphology.erosion(mask, ones(5, 5)) with its default geodesic border does at feature_util.py:183-191
    (kornia is absent from the image: restated, unpinned)."""
    m = mask.reshape(1, 1, *mask.shape[-2:]).to(torch.float32)
    pad = size // 2
    m = torch.nn.functional.pad(m, (pad, pad, pad, pad), value=1.0e4)
    return (-torch.nn.functional.max_pool2d(-m, size, stride=1)).squeeze(0).squeeze(0).to(mask.dtype)


def transform_3d_points_torch(trans: torch.Tensor, points: torch.Tensor) -> torch.Tensor:
    """4x4 transform of [n, 3] points through homogeneous coordinates (utils/geometry.py:31-50)."""
    points_h = torch.hstack([points, torch.ones((points.shape[0], 1), dtype=points.dtype, device=points.device)])
    return torch.matmul(trans.to(points.dtype), points_h.T)[:3, :].T


def get_visual_features_registered_in_3d(image_chw: torch.Tensor, depth_image_hw: torch.Tensor, object_mask: torch.Tensor,
                                         camera, T_model_from_camera: torch.Tensor, extractor: torch.nn.Module,
                                         grid_cell_size: float, debug: bool = False):
    """One rendered template -> (feat_vectors [n, D], vertex_ids [n] i32, vertices_in_model [n, 3]): grid points inside
    the eroded mask, lifted through the depth image into model space, with the patch features sampled there
    (feature_util.py:162-237; the batched form over all templates is bank_builder.register_templates_in_3d)."""
    device = image_chw.device
    grid_points = generate_grid_points((image_chw.shape[2], image_chw.shape[1]), grid_cell_size).to(device)
    query_points = filter_points_by_mask(grid_points, erode_mask(object_mask))
    vertices_in_cam = lift_2d_points_to_3d(query_points, depth_image_hw, camera)
    vertices_in_model = transform_3d_points_torch(T_model_from_camera.to(device), vertices_in_cam)
    vertex_ids = torch.arange(vertices_in_model.shape[0], dtype=torch.int32)
    feature_map_chw = extractor(image_chw.unsqueeze(0))["feature_maps"][0]
    feat_vectors = sample_feature_map_at_points(feature_map_chw, query_points, (image_chw.shape[-1], image_chw.shape[-2])).detach()
    return feat_vectors, vertex_ids, vertices_in_model
