"""Crop producer on the GPU (the step right before the hot path).

Mirrors the reference's per-detection preparation, scripts/infer.py:411-462:
  calc_crop_box            utils/misc.py:171-209   amodal box -> (square) crop box
  construct_crop_camera    utils/misc.py:212-277   virtual pinhole camera looking at the box (host, fp64 numpy)
  warp_image               utils/misc.py:458-519   destination pixel -> source pixel map + cv2.remap
The camera construction is a handful of 3x3 operations and stays on the host in fp64; the per-pixel map and the
resampling -- the part that is O(pixels) and that the reference runs through numpy + cv2 on one CPU thread per
detection -- is the HIP kernel fp_warp_crops, batched over all detections of an image and writing the [B,3,S,S] tensor
the extractor consumes.  There is no CPU fallback.

cv2 is not needed: INTER_* are the cv2 constants' values.  cv2.remap's arithmetic is restated (OpenCV 4.5 semantics,
see csrc/crop.hip); it cannot be pinned in this image (cv2 absent), the map chain in front of it is pinned against
the reference (tests/golden/crop_*.npz).
"""

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import call, ptr, require_cuda, stream, upload_async

INTER_NEAREST, INTER_LINEAR, INTER_AREA = 0, 1, 3


class AlignedBox2f:
    """Axis-aligned 2D box (the fields of utils/structs.py:115-170 that the crop path reads)."""

    def __init__(self, left: float, top: float, right: float, bottom: float):
        self.left, self.top, self.right, self.bottom = left, top, right, bottom

    @property
    def width(self) -> float:
        return self.right - self.left

    @property
    def height(self) -> float:
        return self.bottom - self.top

    def __repr__(self):
        return f"AlignedBox2f(left={self.left}, top={self.top}, right={self.right}, bottom={self.bottom})"


class PinholePlaneCameraModel:
    """Pinhole camera with the attribute names of utils/structs.py:255-352,672 (width, height, f, c,
    T_world_from_eye as 4x4 fp64); only what the crop path touches."""

    def __init__(self, width: int, height: int, f, c, T_world_from_eye: Optional[np.ndarray] = None):
        self.width, self.height = int(width), int(height)
        self.f = tuple(np.broadcast_to(f, 2))
        self.c = tuple(c)
        T = np.eye(4) if T_world_from_eye is None else np.array(T_world_from_eye, dtype=np.float64)
        if T.shape == (3, 4):
            T = np.vstack([T, [0.0, 0.0, 0.0, 1.0]])
        if np.abs((T.T @ T)[:3, :3] - np.eye(3)).max() >= 1.0e-5:
            raise ValueError("camera T_world_from_eye must be a rigid transform")
        self.T_world_from_eye = T


def calc_crop_box(box: AlignedBox2f, box_scaling_factor: float = 1.0, make_square: bool = False) -> AlignedBox2f:
    """Scales the box about its centre and optionally pads the short side to a square (utils/misc.py:171-209)."""
    w, h = box.width * box_scaling_factor, box.height * box_scaling_factor
    if make_square:
        w = h = max(w, h)
    x_pad, y_pad = 0.5 * (w - box.width), 0.5 * (h - box.height)
    return AlignedBox2f(box.left - x_pad, box.top - y_pad, box.right + x_pad, box.bottom + y_pad)


def _unit(v: np.ndarray, eps: float = 5.43e-20) -> np.ndarray:
    return v / np.maximum(eps, (v * v).sum(axis=-1, keepdims=True) ** 0.5)  # utils/geometry.py:213-229


def _rotation_between(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Rotation taking direction a onto direction b (Rodrigues form of utils/geometry.py:135-150)."""
    a, b = _unit(a), _unit(b)
    v = np.cross(a, b)
    s, c = np.linalg.norm(v), np.dot(a, b)
    vx = np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])
    return np.eye(3, 3, dtype=a.dtype) + vx + np.matmul(vx, vx) * (1 - c) / (max(s * s, 1e-15))


def gen_look_at_matrix(orig_camera_from_world: np.ndarray, center: np.ndarray) -> np.ndarray:
    """camera_from_world of the camera at the same position whose +z passes through `center` (world coordinates);
    utils/geometry.py:52-88 with camera_angle = 0."""
    center_local = center.reshape(-1, 3) @ orig_camera_from_world[:3, :3].T
    center_local = center_local.reshape(center.shape) + orig_camera_from_world[:3, 3]
    z_dir_local = center_local / np.linalg.norm(center_local)
    delta_r_local = _rotation_between(np.array([0, 0, 1], dtype=center.dtype), z_dir_local)
    world_from_aligned = np.linalg.inv(orig_camera_from_world).copy()
    world_from_aligned[0:3, 0:3] = world_from_aligned[0:3, 0:3] @ delta_r_local
    return np.linalg.inv(world_from_aligned)


def construct_crop_camera(box: AlignedBox2f, camera_model_c2w: PinholePlaneCameraModel,
                          viewport_size: Tuple[int, int], viewport_rel_pad: float) -> PinholePlaneCameraModel:
    """Virtual pinhole camera whose optical axis passes through the box centre and whose focal length makes the
    sphere around the box (+ padding) fill the viewport (utils/misc.py:212-277)."""
    T = camera_model_c2w.T_world_from_eye
    f = 0.5 * (camera_model_c2w.f[0] + camera_model_c2w.f[1])
    cx, cy = camera_model_c2w.c
    corners = np.array([[box.left - cx, box.top - cy, f], [box.right - cx, box.top - cy, f],
                        [box.left - cx, box.bottom - cy, f], [box.right - cx, box.bottom - cy, f]])
    corners /= np.linalg.norm(corners, axis=1, keepdims=True)
    centroid_c = np.mean(corners, axis=0)
    centroid_w = T.dot(np.hstack([centroid_c, 1]).reshape((4, 1)))[:3, 0]
    radius = np.linalg.norm(corners - centroid_c, axis=1).max()
    trans_w2c = np.linalg.inv(T)
    trans_w2vc = gen_look_at_matrix(trans_w2c, centroid_w)
    centroid_h = np.hstack((np.expand_dims(centroid_w, axis=0), np.ones((1, 1))))
    centroid_vc = trans_w2vc.dot(centroid_h.T)[:3, :].T.squeeze()
    # The reference computes the next four lines on float32 arrays with float64 scalars mixed in; under its pinned
    # numpy (1.26.4, value-based casting: conda_foundpose_gpu.yaml:23) every step is a float32 operation.  The casts
    # make that explicit, so the result does not depend on the numpy generation running this file.
    fx_fy_orig = np.array(camera_model_c2w.f, dtype=np.float32)
    radius_2d = fx_fy_orig * np.float32(radius) / np.float32(centroid_vc[2])
    extent_2d = np.float32(1.0 + viewport_rel_pad) * radius_2d
    cx_cy = np.array(viewport_size, dtype=np.float32) / np.float32(2.0) - np.float32(0.5)
    fx_fy = fx_fy_orig * cx_cy / extent_2d
    return PinholePlaneCameraModel(width=viewport_size[0], height=viewport_size[1], f=tuple(fx_fy), c=tuple(cx_cy),
                                   T_world_from_eye=np.linalg.inv(trans_w2vc))


def camera_pair_params(src_camera: PinholePlaneCameraModel, dst_camera: PinholePlaneCameraModel) -> np.ndarray:
    """The 32 doubles fp_warp_crops takes per crop: (f, c, R row-major, t) of the crop camera, then of the source."""
    out = []
    for cam in (dst_camera, src_camera):
        T = cam.T_world_from_eye
        out += [np.float64(cam.f[0]), np.float64(cam.f[1]), np.float64(cam.c[0]), np.float64(cam.c[1])]
        out += list(T[:3, :3].reshape(-1)) + list(T[:3, 3])
    return np.asarray(out, dtype=np.float64)


def _warp(src: torch.Tensor, mode: int, src_index: Optional[torch.Tensor], params: np.ndarray, out_hw, depth_check: bool,
          want_maps: bool = False):
    require_cuda(src)
    B = params.shape[0]
    H, W = out_hw
    p = upload_async(torch.from_numpy(np.ascontiguousarray(params, dtype=np.float64)), src.device)   # pinned + asynchronous: the host goes on to the next image while this batch runs
    if mode == INTER_NEAREST:
        if src.dtype != torch.uint8 or src.dim() != 3:
            raise ValueError("nearest-mode source must be uint8 [n, H, W]")
        n, sh, sw, ch = src.shape[0], src.shape[1], src.shape[2], 1
        out = torch.empty(B, H, W, dtype=torch.uint8, device=src.device)
    else:
        if src.dtype != torch.float32 or src.dim() != 4:
            raise ValueError("linear-mode source must be float32 [n, H, W, C]")
        n, sh, sw, ch = src.shape
        out = torch.empty(B, ch, H, W, dtype=torch.float32, device=src.device)
    src = src.contiguous()
    maps = torch.empty(B, 2, H, W, dtype=torch.float32, device=src.device) if want_maps else None
    idx = None if src_index is None else (src_index.to(torch.int32).contiguous() if src_index.is_cuda else upload_async(src_index.to(torch.int32), src.device))
    call("fp_warp_crops", ptr(src), n, sh, sw, ch, 1 if mode == INTER_NEAREST else 0, ptr(idx), ptr(p), B, H, W,
         1 if depth_check else 0, ptr(out), ptr(maps), stream())
    return (out, maps) if want_maps else out


def warp_image(src_camera: PinholePlaneCameraModel, dst_camera: PinholePlaneCameraModel, src_image: torch.Tensor,
               interpolation: int = INTER_LINEAR, depth_check: bool = True) -> torch.Tensor:
    """One image through one camera pair, same result layout as the reference (HxWxC float32, or HxW uint8 for a
    mask warped with INTER_NEAREST); `src_image` is a CUDA tensor.  INTER_AREA is resampled as INTER_LINEAR, which is
    what cv2.remap does with it."""
    params = camera_pair_params(src_camera, dst_camera)[None]
    hw = (dst_camera.height, dst_camera.width)
    if interpolation == INTER_NEAREST:
        if src_image.dim() != 2:
            raise ValueError("INTER_NEAREST is implemented for single-channel uint8 masks")
        return _warp(src_image[None].to(torch.uint8), INTER_NEAREST, None, params, hw, depth_check)[0]
    if interpolation not in (INTER_LINEAR, INTER_AREA):
        raise ValueError(f"unsupported interpolation {interpolation}")
    img = src_image if src_image.dim() == 3 else src_image[..., None]
    out = _warp(img[None].to(torch.float32), INTER_LINEAR, None, params, hw, depth_check)[0].permute(1, 2, 0)
    return out if src_image.dim() == 3 else out[..., 0]


def warp_crops(images_hwc: torch.Tensor, masks: torch.Tensor, src_cameras: Sequence[PinholePlaneCameraModel],
               crop_cameras: Sequence[PinholePlaneCameraModel], image_index: Optional[Sequence[int]] = None,
               depth_check: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """Batched form for the extractor: images [n,H,W,3] float32 in [0,1], per-detection masks [B,H,W] uint8 ->
    (crops [B,3,S,S] float32, crop masks [B,S,S] uint8); detection b reads image image_index[b] (default b)."""
    B = len(crop_cameras)
    params = np.stack([camera_pair_params(s, d) for s, d in zip(src_cameras, crop_cameras)])
    hw = (crop_cameras[0].height, crop_cameras[0].width)
    if any((c.height, c.width) != hw for c in crop_cameras):
        raise ValueError("all crop cameras of a batch must share one viewport size")
    idx = None if image_index is None else torch.as_tensor(list(image_index), dtype=torch.int32)
    crops = _warp(images_hwc, INTER_LINEAR, idx, params, hw, depth_check)
    crop_masks = _warp(masks, INTER_NEAREST, None, params, hw, depth_check)
    assert crops.shape[0] == B
    return crops, crop_masks


def crop_detections(image_hwc: torch.Tensor, masks_modal: torch.Tensor, boxes_amodal: Sequence[Sequence[float]],
                    camera_c2w: PinholePlaneCameraModel, crop_size: Tuple[int, int], crop_rel_pad: float):
    """All detections of one image, the way infer.py:411-450 prepares each of them: square crop box, virtual camera,
    warped RGB crop and modal mask.  -> (crops [B,3,S,S], masks [B,S,S], crop cameras)."""
    cams: List[PinholePlaneCameraModel] = []
    for b in boxes_amodal:
        box = calc_crop_box(AlignedBox2f(b[0], b[1], b[2], b[3]), make_square=True)
        cams.append(construct_crop_camera(box, camera_c2w, crop_size, crop_rel_pad))
    crops, crop_masks = warp_crops(image_hwc[None], masks_modal, [camera_c2w] * len(cams), cams, [0] * len(cams))
    return crops, crop_masks, cams
