"""CPU: detections -> instance records (foundpose_amd/infer_pose_util.py; interface of /root/reference/utils/infer_pose_util.py:44-151)."""
import types

import numpy as np
import pytest

from foundpose_amd import infer_pose_util as ipu


def _det(mask, bbox, score, t=0.5):
    return {"bbox": bbox, "score": score, "time": t, "segmentation": ipu.binary_mask_to_rle(mask)}


def test_rle_round_trip_and_opening():
    rng = np.random.default_rng(0)
    m = (rng.random((37, 53)) > 0.6).astype(np.uint8)
    assert np.array_equal(ipu.rle_to_binary_mask(ipu.binary_mask_to_rle(m)), m)
    blob = np.zeros((20, 20), np.uint8)
    blob[5:12, 4:15] = 1
    blob[1, 1] = 1                                   # an isolated pixel: removed by the 3 x 3 opening, the block survives
    opened = ipu.open_mask_3x3(blob)
    assert opened[1, 1] == 0 and np.array_equal(opened[5:12, 4:15], np.ones((7, 11), np.uint8)) and opened.sum() == 77


def test_instances_sorted_cut_shifted_and_matched_to_ground_truth():
    H, W = 60, 80                                     # detector canvas; the image was centre-cropped to 56 x 70 (h x w)
    m1 = np.zeros((H, W), np.uint8); m1[10:30, 20:50] = 1
    m2 = np.zeros((H, W), np.uint8); m2[35:55, 10:40] = 1
    m3 = np.zeros((H, W), np.uint8); m3[5:15, 60:75] = 1
    dets = {(1, 7, 3): [_det(m2, [10, 35, 30, 20], 0.4), _det(m1, [20, 10, 30, 20], 0.9), _det(m3, [60, 5, 15, 10], 0.7)]}
    dx, dy = (W - 70) // 2, (H - 56) // 2
    g1 = types.SimpleNamespace(masks_modal=m1[dy:H - dy, dx:W - dx].copy(), boxes_amodal=np.array([0, 0, 1, 1]))
    g2 = types.SimpleNamespace(masks_modal=m2[dy:H - dy, dx:W - dx].copy(), boxes_amodal=np.array([0, 0, 1, 1]))
    inst = ipu.get_instances_for_pose_estimation(1, 7, 3, True, dets, 2, [g2, g1], (70, 56))
    assert len(inst) == 2                                                    # the two best-scoring detections, best first
    assert [i["time"] for i in inst] == [0.5, 0.5]
    a, b = inst
    assert a["input_box_amodal"].tolist() == [20 - dx, 10 - dy, 50 - dx, 30 - dy]      # (x, y, w, h) on the canvas -> (x1, y1, x2, y2) in the image
    assert b["input_box_amodal"].tolist() == [60 - dx, 5 - dy, 75 - dx, 15 - dy]
    assert a["input_mask_modal"].shape == (56, 70) and np.array_equal(a["input_mask_modal"], m1[dy:H - dy, dx:W - dx])
    assert a["gt_anno"] is g1 and a["gt_iou"] == pytest.approx(1.0)
    assert b["gt_anno"] is g2 and b["gt_iou"] == 0.0                         # nothing overlaps: annotation 0, IoU 0 (the reference's initial values)
    # one detection only: kept whatever max_num_preds says; same-size image: no shift (the reference's [0:-0] slice would empty the mask)
    one = ipu.get_instances_for_pose_estimation(1, 7, 3, True, {(1, 7, 3): [_det(m1, [20, 10, 30, 20], 0.9)]}, 0, [], (W, H))
    assert len(one) == 1 and one[0]["input_mask_modal"].sum() == m1.sum() and one[0]["gt_anno"] is None
    assert ipu.get_instances_for_pose_estimation(1, 8, 3, True, dets, 2, [], (70, 56)) == []
    with pytest.raises(ValueError, match="larger than mask"):
        ipu.get_instances_for_pose_estimation(1, 7, 3, True, dets, 2, [], (W + 2, H))
    gt = ipu.get_instances_for_pose_estimation(1, 7, 3, False, dets, 2, [g1, g2], (70, 56))
    assert len(gt) == 2 and gt[0]["gt_anno"] is g1 and "gt_iou" not in gt[0]
