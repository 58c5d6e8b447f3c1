"""Pose from 2D-3D correspondences with the reference's interface (/root/reference/utils/pnp_util.py:20-84: cv2.solvePnPRansac +
optional cv2.solvePnPRefineLM), run on the MI355X for a whole batch of (detection, template) pairs at once, and the
best-coarse-pose selection of /root/reference/scripts/infer.py:552-602.

cv2 is not a dependency of this path (and is absent from the image): the scheme is OpenCV's -- RANSAC over minimal
samples, inlier = reprojection error <= threshold, best = first model with the most inliers within the adaptively
shortened budget, Levenberg-Marquardt on the inliers -- with a P3P minimal solver and a counter-based sampler
(csrc/pnp.hip).  cv2's random stream is not reproduced, so individual hypotheses differ; the estimate agrees wherever the
inlier set is unambiguous (parity of cv2's own arithmetic: unpinned).
"""

from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import call, ptr, stream, require_cuda, upload_async
from .matching import MatchResult

MIN_CORRESP = 6  # scripts/infer.py:555-559


def _intrinsics(camera) -> Tuple[float, float, float, float]:
    """(fx, fy, cx, cy) of a pinhole camera: an object or dict with f / c (the reference's PinholePlaneCameraModel), a 3x3 K,
    or the four numbers themselves."""
    if isinstance(camera, dict):
        f, c = camera["f"], camera["c"]
    elif hasattr(camera, "f") and hasattr(camera, "c"):
        f, c = camera.f, camera.c
    else:
        K = np.asarray(camera, np.float64)
        if K.shape == (4,):
            return float(K[0]), float(K[1]), float(K[2]), float(K[3])
        return float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    f, c = np.asarray(f, np.float64).reshape(-1), np.asarray(c, np.float64).reshape(-1)
    return float(f[0]), float(f[1]), float(c[0]), float(c[1])


def solve_pnp_ransac_batch(coord_2d: torch.Tensor, coord_3d: torch.Tensor, counts: torch.Tensor, cameras: Sequence[Any],
                           pnp_ransac_iter: int = 1000, pnp_inlier_thresh: float = 3.0, pnp_required_ransac_conf: float = 0.99,
                           pnp_refine_lm: bool = True, seed: int = 0, return_ransac_pose: bool = False,
                           min_corresp: int = 6) -> Dict[str, torch.Tensor]:
    """coord_2d [B, n, K, 2], coord_3d [B, n, K, 3], counts [B, n] (MatchResult's padded layout); cameras: one per detection.
    -> dict of device tensors: success [B, n] bool, R [B, n, 3, 3] f64, t [B, n, 3] f64, quality [B, n] (RANSAC inliers),
    inliers [B, n, K] bool (+ ransac_pose [B, n, 12]).  min_corresp: sets with fewer correspondences fail without being tried --
    6 in the driver's loop (scripts/infer.py:555-559), 4 for a bare estimate_pose call (what cv2.solvePnPRansac needs)."""
    require_cuda(coord_2d, coord_3d, counts)
    B, n, K = coord_2d.shape[:3]
    dev = coord_2d.device
    if len(cameras) != B:
        raise ValueError(f"{len(cameras)} cameras for {B} detections")
    cam = upload_async(torch.tensor([_intrinsics(c) for c in cameras], dtype=torch.float64).reshape(B, 4), dev)   # (a pageable upload would block until the batch has drained)
    c2, c3 = coord_2d.float().contiguous(), coord_3d.float().contiguous()
    cnt = counts.to(torch.int32).contiguous()
    P = B * n
    success = torch.zeros(P, dtype=torch.int32, device=dev)
    R = torch.zeros(P, 9, dtype=torch.float64, device=dev)
    t = torch.zeros(P, 3, dtype=torch.float64, device=dev)
    ninl = torch.zeros(P, dtype=torch.int32, device=dev)
    mask = torch.zeros(P, K, dtype=torch.uint8, device=dev)
    rp = torch.zeros(P, 12, dtype=torch.float64, device=dev) if return_ransac_pose else None
    lm_iters = 20 + (20 if pnp_refine_lm else 0)
    call("fp_pnp_ransac", ptr(c2), ptr(c3), ptr(cnt), ptr(cam), P, n, K, int(pnp_ransac_iter), float(pnp_inlier_thresh),
         float(pnp_required_ransac_conf), lm_iters, int(min_corresp), int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(success), ptr(R), ptr(t), ptr(ninl), ptr(mask), ptr(rp), stream())
    out = {"success": success.reshape(B, n).bool(), "R": R.reshape(B, n, 3, 3), "t": t.reshape(B, n, 3),
           "quality": ninl.reshape(B, n).to(torch.float64), "inliers": mask.reshape(B, n, K).bool()}
    if rp is not None:
        out["ransac_pose"] = rp.reshape(B, n, 12)
    return out


def estimate_poses(res: MatchResult, cameras: Sequence[Any], pnp_type: str = "opencv", pnp_ransac_iter: int = 1000,
                   pnp_inlier_thresh: float = 3.0, pnp_required_ransac_conf: float = 0.99, pnp_refine_lm: bool = True, seed: int = 0):
    """All coarse poses of a batch (the loop of infer.py:552-580 for every detection at once)."""
    if pnp_type != "opencv":
        raise ValueError("Unsupported PnP type")
    counts = torch.where(res.template_ids >= 0, res.counts, torch.zeros_like(res.counts))
    return solve_pnp_ransac_batch(res.coord_2d, res.coord_3d, counts, cameras, pnp_ransac_iter, pnp_inlier_thresh,
                                  pnp_required_ransac_conf, pnp_refine_lm, seed)


def select_best_coarse(poses: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """infer.py:582-602: among a detection's successful coarse poses the one of the highest quality, the first on ties.
    -> found [B] bool, corresp_id [B], R [B, 3, 3], t [B, 3], quality [B]."""
    q = torch.where(poses["success"], poses["quality"], torch.full_like(poses["quality"], -1.0))
    best_q, _ = q.max(dim=1)
    n = q.shape[1]
    first = torch.where(q == best_q[:, None], torch.arange(n, device=q.device)[None, :], torch.full_like(q, n, dtype=torch.int64)).min(dim=1).values
    first = first.clamp_max(n - 1)
    idx = first[:, None, None]
    return {"found": best_q >= 0, "corresp_id": first,
            "R": poses["R"].gather(1, idx[..., None].expand(-1, 1, 3, 3))[:, 0], "t": poses["t"].gather(1, idx.expand(-1, 1, 3))[:, 0],
            "quality": best_q}


def estimate_pose(corresp: Dict[str, Any], camera_c2w: Any, pnp_type: str, pnp_ransac_iter: int, pnp_inlier_thresh: float,
                  pnp_required_ransac_conf: float, pnp_refine_lm: bool, seed: int = 0) -> Tuple[bool, Optional[np.ndarray], Optional[np.ndarray], Optional[np.ndarray], Optional[float]]:
    """The reference's per-correspondence-set call (utils/pnp_util.py:20-84): -> (success, R_m2c [3,3], t_m2c [3,1], inlier ids
    [num_inliers, 1], quality); (False, None, None, None, None) where cv2 would have raised or failed."""
    if pnp_type != "opencv":
        raise ValueError("Unsupported PnP type")
    c2 = torch.as_tensor(corresp["coord_2d"]).to("cuda", torch.float32)
    c3 = torch.as_tensor(corresp["coord_3d"]).to("cuda", torch.float32)
    k = int(c2.shape[0])
    if k < 4:
        return False, None, None, None, None
    out = solve_pnp_ransac_batch(c2.reshape(1, 1, k, 2), c3.reshape(1, 1, k, 3), torch.tensor([[k]], dtype=torch.int32, device="cuda"),
                                 [camera_c2w], pnp_ransac_iter, pnp_inlier_thresh, pnp_required_ransac_conf, pnp_refine_lm, seed, min_corresp=4)
    if not bool(out["success"][0, 0]):
        return False, None, None, None, None
    inl = torch.nonzero(out["inliers"][0, 0]).to(torch.int32).cpu().numpy()
    return True, out["R"][0, 0].cpu().numpy(), out["t"][0, 0].cpu().numpy().reshape(3, 1), inl, float(out["quality"][0, 0])
