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


def get_instances_for_pose_estimation(bop_chunk_id: int, bop_im_id: int, obj_id: int, use_detections: bool, detections: Dict[Any, Any],
                                      max_num_preds: int, gt_object_annos: Sequence[Any], image_size: Tuple[int, int]) -> List[Dict[str, Any]]:
    """Per-instance dicts {input_box_amodal (x1, y1, x2, y2), input_mask_modal uint8 [H, W], gt_anno, gt_iou, time}."""
    instance_infos: List[Dict[str, Any]] = []
    if not use_detections:
        for anno in gt_object_annos:
            instance_infos.append({"input_box_amodal": np.array(anno.boxes_amodal).copy(), "input_mask_modal": np.array(anno.masks_modal).copy(), "gt_anno": anno})
        return instance_infos
    key = (bop_chunk_id, bop_im_id, obj_id)
    if key not in detections:
        return []
    preds = detections[key]
    if len(preds) > 1:
        preds = sorted(preds, key=lambda x: x["score"], reverse=True)[:max_num_preds]
    for pred in preds:
        box_amodal = np.array(pred["bbox"])  # (x, y, w, h)
        mask_modal = open_mask_3x3(rle_to_binary_mask(pred["segmentation"]).astype(np.uint8))
        mask_size = (mask_modal.shape[1], mask_modal.shape[0])
        # the input image may have been centre-cropped to a multiple of the ViT patch size
        shift_x = shift_y = 0
        if image_size[0] < mask_size[0]:
            shift_x = (mask_size[0] - image_size[0]) // 2
        elif image_size[0] > mask_size[0]:
            raise ValueError("Image is larger than mask.")
        if image_size[1] < mask_size[1]:
            shift_y = (mask_size[1] - image_size[1]) // 2
        elif image_size[1] > mask_size[1]:
            raise ValueError("Image is larger than mask.")
        # (the reference slices [shift:-shift], which empties the mask when the shift is 0; a zero shift is a no-op here)
        mask_modal = mask_modal[shift_y:mask_modal.shape[0] - shift_y, shift_x:mask_modal.shape[1] - shift_x]
        box_amodal[0] -= shift_x
        box_amodal[1] -= shift_y
        box_amodal[2] += box_amodal[0]
        box_amodal[3] += box_amodal[1]
        best_anno_id, best_anno_iou, gt_anno = 0, 0.0, None
        if len(gt_object_annos) != 0:
            for anno_id, anno in enumerate(gt_object_annos):
                iou = mask_iou(mask_modal, anno.masks_modal)
                if iou > best_anno_iou:
                    best_anno_iou, best_anno_id = iou, anno_id
            gt_anno = gt_object_annos[best_anno_id]
        instance_infos.append({"input_box_amodal": box_amodal, "input_mask_modal": mask_modal, "gt_anno": gt_anno, "gt_iou": best_anno_iou, "time": pred["time"]})
    return instance_infos
