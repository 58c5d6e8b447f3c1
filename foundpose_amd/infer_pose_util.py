"""Detections in, object instances out: the input side of the driver, with the reference's function names
(/root/reference/utils/infer_pose_util.py:24-151).  CNOS/BOP detection files are JSON lists of
{scene_id, image_id, category_id, bbox [x, y, w, h], segmentation (COCO RLE), score, time}.

The reference decodes the RLE with bop_toolkit's pycoco_utils and opens the mask with cv2.morphologyEx; neither package
is a dependency here: the RLE codec is restated (uncompressed counts lists as CNOS writes them, and COCO's compressed
strings), the 3x3 opening is two pooling passes with cv2's border rules (erosion ignores the outside, dilation too)."""

import json
from collections import defaultdict
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch


def load_detections_in_bop_format(path: str) -> Dict[Tuple[int, int, int], List[Dict[str, Any]]]:
    with open(path) as f:
        pred_mask_list = json.load(f)
    detections = defaultdict(list)
    for pred in pred_mask_list:
        key = (pred["scene_id"], pred["image_id"], pred["category_id"])
        detections[key].append({"bbox": pred["bbox"], "segmentation": pred["segmentation"], "score": pred["score"], "time": pred["time"]})
    return detections


def _decode_compressed_counts(s: str) -> List[int]:
    """COCO's LEB128-like string form of the run lengths (pycocotools rleFrString)."""
    counts, p, m = [], 0, 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if m > 2:
            x += counts[m - 2]
        counts.append(x)
        m += 1
    return counts


def rle_to_binary_mask(rle: Dict[str, Any]) -> np.ndarray:
    """COCO RLE {"counts", "size": [h, w]} -> bool [h, w]; runs alternate 0/1 starting with 0, column-major."""
    h, w = rle["size"]
    counts = rle["counts"]
    if isinstance(counts, (str, bytes)):
        counts = _decode_compressed_counts(counts.decode() if isinstance(counts, bytes) else counts)
    flat = np.zeros(h * w, dtype=bool)
    pos, val = 0, False
    for c in counts:
        if val:
            flat[pos:pos + c] = True
        pos += c
        val = not val
    return flat.reshape((h, w), order="F")


def binary_mask_to_rle(mask: np.ndarray) -> Dict[str, Any]:
    """Inverse of rle_to_binary_mask (uncompressed counts), the form CNOS result files use."""
    flat = np.asarray(mask).astype(bool).ravel(order="F")
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    edges = np.concatenate([[0], change, [flat.size]])
    counts = np.diff(edges).tolist()
    if flat.size and flat[0]:
        counts = [0] + counts
    return {"counts": counts, "size": [int(mask.shape[0]), int(mask.shape[1])]}


def open_mask_3x3(mask: np.ndarray) -> np.ndarray:
    """cv2.morphologyEx(mask, MORPH_OPEN, 3x3 rect): erosion then dilation; pixels outside the image never win (cv2's
    default border value for each of the two passes)."""
    m = torch.as_tensor(np.asarray(mask) != 0, dtype=torch.float32)[None, None]
    er = 1.0 - torch.nn.functional.max_pool2d(torch.nn.functional.pad(1.0 - m, (1, 1, 1, 1), value=0.0), 3, 1)
    di = torch.nn.functional.max_pool2d(torch.nn.functional.pad(er, (1, 1, 1, 1), value=0.0), 3, 1)
    return di[0, 0].numpy().astype(np.uint8)


def mask_iou(a: np.ndarray, b: np.ndarray) -> float:
    a, b = np.asarray(a).astype(bool), np.asarray(b).astype(bool)
    union = np.logical_or(a, b).sum()
    return float(np.logical_and(a, b).sum() / union) if union else 0.0


def _centre_crop_offsets(canvas_wh: Tuple[int, int], image_wh: Tuple[int, int]) -> np.ndarray:
    """(dx, dy) of an image that was centre-cropped out of the detector's canvas (e.g. to a multiple of the ViT patch size); the image can
    never be larger than the canvas the masks were predicted on."""
    canvas, image = np.asarray(canvas_wh, np.int64), np.asarray(image_wh, np.int64)
    if np.any(image > canvas):
        raise ValueError("Image is larger than mask.")
    return (canvas - image) // 2


def _best_overlap(mask: np.ndarray, annos: Sequence[Any]) -> Tuple[Optional[Any], float]:
    """The annotation whose modal mask overlaps `mask` most (first one on ties, annotation 0 when nothing overlaps) and that IoU; all IoUs in
    one vectorised pass over the stacked annotation masks."""
    if len(annos) == 0:
        return None, 0.0
    m = np.asarray(mask).astype(bool)
    stack = np.stack([np.asarray(a.masks_modal).astype(bool) for a in annos])
    inter = np.logical_and(stack, m[None]).reshape(len(annos), -1).sum(1)
    union = np.logical_or(stack, m[None]).reshape(len(annos), -1).sum(1)
    ious = np.where(union > 0, inter / np.maximum(union, 1), 0.0)
    best = int(np.argmax(ious)) if float(ious.max()) > 0.0 else 0    # argmax returns the first maximum, like a strict `>` scan
    return annos[best], float(ious[best]) if float(ious.max()) > 0.0 else 0.0


def _instance_from_detection(pred: Dict[str, Any], image_size: Tuple[int, int], gt_object_annos: Sequence[Any]) -> Dict[str, Any]:
    """One CNOS detection -> the instance record of the driver: opened modal mask and amodal box, both moved into the (possibly
    centre-cropped) image, plus the best-overlapping annotation when ground truth is given."""
    mask = open_mask_3x3(rle_to_binary_mask(pred["segmentation"]).astype(np.uint8))
    dx, dy = _centre_crop_offsets((mask.shape[1], mask.shape[0]), image_size)
    # (the reference slices [shift:-shift], which empties the mask when the shift is 0; a zero shift is a no-op here)
    mask = mask[dy:mask.shape[0] - dy, dx:mask.shape[1] - dx]
    x, y, w, h = np.array(pred["bbox"])                       # CNOS boxes are (x, y, w, h) on the detector's canvas
    box = np.array(pred["bbox"])
    box[:] = (x - dx, y - dy, x - dx + w, y - dy + h)        # -> (x1, y1, x2, y2) in the image, dtype of the input kept
    gt_anno, gt_iou = _best_overlap(mask, gt_object_annos)
    return {"input_box_amodal": box, "input_mask_modal": mask, "gt_anno": gt_anno, "gt_iou": gt_iou, "time": pred["time"]}


def get_instances_for_pose_estimation(bop_chunk_id: int, bop_im_id: int, obj_id: int, use_detections: bool, detections: Dict[Any, Any],
                                      max_num_preds: int, gt_object_annos: Sequence[Any], image_size: Tuple[int, int]) -> List[Dict[str, Any]]:
    """Per-instance dicts {input_box_amodal (x1, y1, x2, y2), input_mask_modal uint8 [H, W], gt_anno, gt_iou, time}: the interface of
    /root/reference/utils/infer_pose_util.py:44-151.  With detections: the max_num_preds best-scoring ones of (scene, image, object) -- a
    single detection is kept whatever max_num_preds says, as in the reference -- else one instance per ground-truth annotation."""
    if not use_detections:
        return [{"input_box_amodal": np.array(a.boxes_amodal).copy(), "input_mask_modal": np.array(a.masks_modal).copy(), "gt_anno": a}
                for a in gt_object_annos]
    preds = detections.get((bop_chunk_id, bop_im_id, obj_id))
    if preds is None:
        return []
    if len(preds) > 1:
        order = sorted(range(len(preds)), key=lambda i: preds[i]["score"], reverse=True)   # stable: equal scores keep their file order
        preds = [preds[i] for i in order[:max_num_preds]]
    return [_instance_from_detection(p_, image_size, gt_object_annos) for p_ in preds]
