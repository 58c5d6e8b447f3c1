"""The driver end to end on a synthetic scene (/root/reference/scripts/infer.py:55-100, 290-816,
utils/infer_pose_util.py:24-151, utils/eval_util.py:231-355, scripts/prepare_bop_submission.py:63-99): options JSON in the
reference's format, CNOS-format detections with RLE masks, a `repre.pth` bank on disk, frames -> `estimated-poses.json` and
the BOP19 csv, with the pose of each instance checked against the pose its bank was planted with."""
import json
import os

import numpy as np
import pytest
import torch

from foundpose_amd import bank_builder, crop_util, feature_util, infer, infer_pose_util, repre_util, workload

pytestmark = pytest.mark.gpu
NAME = "dinov2_version=vits14-reg_stride=14_facet=token_layer=9_logbin=0_norm=1"


def _scene(tmp_path, ex):
    """Options JSON, CNOS detections, one frame and a repre.pth whose templates 3 and 7 are the two instances' own crops -> the pieces of a run."""
    g = torch.Generator().manual_seed(0)
    H, W = 480, 640
    image = (torch.rand(H, W, 3, generator=g) * 255).to(torch.uint8).numpy()
    cam = crop_util.PinholePlaneCameraModel(W, H, (600.0, 600.0), (320.0, 240.0), np.eye(4))
    boxes_xywh = [[100, 80, 180, 160], [380, 220, 150, 190]]
    masks = np.zeros((2, H, W), np.uint8)
    for b, (x, y, w, h) in enumerate(boxes_xywh):
        masks[b, y + 10:y + h - 10, x + 10:x + w - 10] = 1
    dets = [{"scene_id": 1, "image_id": 3, "category_id": 1, "bbox": boxes_xywh[b], "score": 0.9 - 0.1 * b, "time": 0.25,
             "segmentation": infer_pose_util.binary_mask_to_rle(masks[b])} for b in range(2)]
    dets.append({"scene_id": 1, "image_id": 3, "category_id": 2, "bbox": [0, 0, 10, 10], "score": 0.5, "time": 0.25,
                 "segmentation": infer_pose_util.binary_mask_to_rle(np.zeros((H, W), np.uint8))})  # another object: ignored
    det_path = tmp_path / "cnos.json"
    det_path.write_text(json.dumps(dets))
    opts_path = tmp_path / "opts.json"
    opts_path.write_text(json.dumps({"infer_opts": {
        "version": "v1", "object_dataset": "synth", "repre_version": "v1", "object_lids": [1], "crop_rel_pad": 0.2, "crop_size": [224, 224],
        "use_detections": True, "extractor_name": NAME, "grid_cell_size": 14.0, "match_template_type": "tfidf", "match_top_n_templates": 5,
        "match_feat_matching_type": "cyclic_buddies", "match_top_k_buddies": 300, "pnp_type": "opencv", "pnp_ransac_iter": 400,
        "pnp_inlier_thresh": 10.0, "final_pose_type": "best_coarse", "num_preds_factor": 2, "vis_results": False}}))
    opts = infer.load_opts(str(opts_path))
    assert opts.crop_size == (224, 224) and opts.pnp_refine_lm is True and opts.pnp_required_ransac_conf == 0.99

    # ---- the bank: the two instances' own crops as templates 3 and 7 (vertices from planted poses in their crop cameras)
    img_f = torch.from_numpy(image).cuda().float() / 255.0
    boxes_xyxy = [[x, y, x + w, y + h] for x, y, w, h in boxes_xywh]
    crops, crop_masks, cams = crop_util.crop_detections(img_f, torch.from_numpy(masks).cuda(), boxes_xyxy, cam, (224, 224), 0.2)
    T = 12
    tpl = torch.rand(T, 3, 224, 224, generator=g).cuda()
    tmask = torch.zeros(T, 224, 224, dtype=torch.uint8).cuda()
    tmask[:, 40:190, 30:200] = 1
    slots = [3, 7]
    for b, s in enumerate(slots):
        tpl[s], tmask[s] = crops[b], crop_masks[b]
    feats, f2t, pts = bank_builder.extract_template_features(ex, tpl, tmask)
    verts = torch.randn(feats.shape[0], 3, generator=g).cuda() * 50.0
    R = workload._random_rotations(2, g)
    t = torch.tensor([[10.0, -20.0, 900.0], [-30.0, 15.0, 1100.0]], dtype=torch.float64)
    for b, s in enumerate(slots):
        rows = f2t == s
        K = torch.tensor([[cams[b].f[0], 0, cams[b].c[0]], [0, cams[b].f[1], cams[b].c[1]], [0, 0, 1.0]], dtype=torch.float64)
        verts[rows] = workload.planted_vertices(pts[rows], K, R[b], t[b]).cuda()
    repre = bank_builder.build_object_repre(feats, f2t, verts, T, pca_components=128, cluster_num=64, cluster_iters=10)
    repre.feat_opts = repre_util.FeatureOpts(extractor_name=NAME)
    rdir = repre_util.get_object_repre_dir_path(str(tmp_path / "object_repre"), opts.repre_version, opts.object_dataset, 1)
    repre_util.save_object_repre(repre, rdir)

    return dict(image=image, cam=cam, cams=cams, masks=masks, boxes_xyxy=boxes_xyxy, det_path=det_path, opts_path=opts_path, opts=opts, rdir=rdir, R=R, t=t)


def test_infer_driver_synthetic_scene(tmp_path):
    ex = feature_util.make_feature_extractor(NAME, random_init_seed=1234, precision="fp32").to("cuda")
    sc = _scene(tmp_path, ex)
    image, cam, cams, masks, boxes_xyxy, det_path, opts, rdir, R, t = (sc[k] for k in ("image", "cam", "cams", "masks", "boxes_xyxy", "det_path", "opts", "rdir", "R", "t"))
    # ---- the driver
    frames = lambda lid: iter([{"scene_id": 1, "im_id": 3, "image": image, "camera": cam}])
    out_dir = str(tmp_path / "inference")
    # (one target instance x num_preds_factor 2 = both detections; without targets AND without annotations the reference skips the image,
    #  scripts/infer.py:317-321 -- checked at the end)
    paths = infer.infer(opts, frames, infer_pose_util.load_detections_in_bop_format(str(det_path)), {1: repre_util.load_object_repre(rdir)}, out_dir, extractor=ex,
                        num_target_insts={1: {(1, 3): 1}})
    est = json.load(open(os.path.join(out_dir, "1", "estimated-poses.json")))
    assert len(est) == 2
    for e in est:
        assert set(e) == {"scene_id", "img_id", "obj_id", "inst_id", "hypothesis_id", "score", "R", "t", "time", "cnos_time"}
        assert all(isinstance(e[k], str) for k in ("scene_id", "img_id", "obj_id", "inst_id", "hypothesis_id", "score"))
        assert (e["scene_id"], e["img_id"], e["obj_id"], e["hypothesis_id"]) == ("1", "3", "1", "0") and e["cnos_time"] == 0.25
        assert np.array(e["R"]).shape == (3, 3) and np.array(e["t"]).shape == (3, 1) and float(e["score"]) > 0.9
        # the reference's per-stage keys (scripts/infer.py:464-633, persisted by utils/eval_util.py:327)
        assert set(e["time"]) == {"prep", "feat_extract", "grid_sample", "proj", "corresp", "pose_coarse", "final_select"}
        assert all(v >= 0.0 for v in e["time"].values()) and e["time"]["feat_extract"] > 0 and e["time"]["corresp"] > 0
    for e in est:  # instances are ordered by detection score -> inst_id b is box b
        b = int(e["inst_id"])
        T_m2c = np.eye(4)
        T_m2c[:3, :3], T_m2c[:3, 3] = R[b].numpy(), t[b].numpy()
        want = np.linalg.inv(cam.T_world_from_eye) @ cams[b].T_world_from_eye @ T_m2c   # planted pose in the ORIGINAL camera
        assert np.abs(np.array(e["R"]) - want[:3, :3]).max() < 1e-4
        assert np.linalg.norm(np.array(e["t"]).ravel() - want[:3, 3]) / np.linalg.norm(want[:3, 3]) < 1e-4
    csv = open(paths[-1]).read().splitlines()
    assert paths[-1].endswith("coarse_synth-estimated-poses.csv") and csv[0] == "scene_id,im_id,obj_id,score,R,t,time" and len(csv) == 3
    row = csv[1].split(",")
    assert row[:3] == ["1", "3", "1"] and len(row[4].split(" ")) == 9 and len(row[5].split(" ")) == 3 and float(row[6]) > 0.25

    # ---- number of poses per (image, object) = num_preds_factor x inst_count of the test targets (infer.py:308-346)
    dets_l = infer_pose_util.load_detections_in_bop_format(str(det_path))
    rep = {1: repre_util.load_object_repre(rdir)}
    o1 = opts._replace(num_preds_factor=1.0)

    def run(tag, targets):
        d = str(tmp_path / tag)
        infer.infer(o1, frames, dets_l, rep, d, extractor=ex, num_target_insts=targets)
        f = os.path.join(d, "1", "estimated-poses.json")
        return json.load(open(f)) if os.path.exists(f) else []
    two = run("t2", {1: {(1, 3): 2}})          # a multi-instance target: both detections get a pose
    assert sorted(int(e["inst_id"]) for e in two) == [0, 1]
    one = run("t1", {1: {(1, 3): 1}})          # one target instance: only the top-scoring detection
    assert [int(e["inst_id"]) for e in one] == [0]
    assert run("t0", {1: {(1, 3): 0}}) == [] and run("tmiss", {1: {(2, 9): 3}}) == [] and run("tobj", {5: {(1, 3): 2}}) == []
    assert run("tnone", None) == []            # no targets and no ground-truth annotations: num_target_insts = 0, the image is skipped (infer.py:317-321)

    # ---- ground-truth annotations instead of targets (infer.py:286-305, 317): only annotations of THIS object that are visible enough count
    class Anno:
        def __init__(self, lid, vis, b):
            self.lid, self.visibilities, self.masks_modal, self.boxes_amodal = lid, vis, masks[b], np.array(boxes_xyxy[b], np.float32)
    def run_gt(tag, annos):
        d = str(tmp_path / tag)
        fr = lambda lid: iter([{"scene_id": 1, "im_id": 3, "image": image, "camera": cam, "gt_annos": annos}])
        infer.infer(o1, fr, dets_l, rep, d, extractor=ex)
        f = os.path.join(d, "1", "estimated-poses.json")
        return json.load(open(f)) if os.path.exists(f) else []
    assert len(run_gt("g2", [Anno(1, 0.9, 0), Anno(1, 0.5, 1)])) == 2          # two visible annotations of object 1 -> two predictions
    assert len(run_gt("g1", [Anno(1, 0.9, 0), Anno(1, 0.05, 1), Anno(2, 0.9, 1)])) == 1   # visibility 0.05 <= min_visibility 0.1, and another object's annotation
    assert run_gt("g0", [Anno(1, float("nan"), 0), Anno(2, 0.9, 1)]) == []     # annotations present but none qualifies: the frame is skipped


def test_cli_loads_an_upstream_layout_checkpoint_file(tmp_path):
    """`python -m foundpose_amd.infer --weights <dir>`: the checkpoint file (upstream hub name, upstream key names, mask_token included) is what the
    run computes with -- poses equal to those of an extractor built from the same state dict in memory -- and without a checkpoint the CLI raises
    (/root/reference/utils/dinov2_utils.py:81-84: pretrained=True; scripts/infer.py:125-128)."""
    from PIL import Image
    from foundpose_amd import synthetic, weights
    from foundpose_amd.vit_config import ARCHS
    sd = synthetic.make_vit_state_dict(ARCHS["vits14-reg"], seed=77)
    assert "mask_token" in sd
    ckdir = tmp_path / "ckpt"
    ckdir.mkdir()
    torch.save(sd, ckdir / "dinov2_vits14_reg4_pretrain.pth")
    ex = feature_util.make_feature_extractor(NAME, state_dict=sd, precision="fp32").to("cuda")
    sc = _scene(tmp_path, ex)
    # the BOP split on disk: <split>/<scene:06d>/rgb/<im:06d>.png + scene_camera.json, test_targets_bop19.json beside the split
    split = tmp_path / "synth" / "test"
    (split / "000001" / "rgb").mkdir(parents=True)
    Image.fromarray(sc["image"]).save(split / "000001" / "rgb" / "000003.png")
    cam = sc["cam"]
    (split / "000001" / "scene_camera.json").write_text(json.dumps({"3": {"cam_K": [cam.f[0], 0, cam.c[0], 0, cam.f[1], cam.c[1], 0, 0, 1], "depth_scale": 1.0}}))
    (tmp_path / "synth" / "test_targets_bop19.json").write_text(json.dumps([{"scene_id": 1, "im_id": 3, "obj_id": 1, "inst_count": 1}]))
    argv = ["--opts", str(sc["opts_path"]), "--dataset-dir", str(split), "--detections", str(sc["det_path"]), "--repre-dir", str(tmp_path / "object_repre"),
            "--precision", "fp32"]
    old_hub = torch.hub.get_dir()
    torch.hub.set_dir(str(tmp_path / "empty_hub"))
    env = os.environ.pop(weights.ENV_VAR, None)
    try:
        with pytest.raises(weights.FoundPoseWeightsError, match="dinov2_vits14_reg4_pretrain.pth"):
            infer.main(argv + ["--output-dir", str(tmp_path / "out_none")])
        infer.main(argv + ["--output-dir", str(tmp_path / "out_cli"), "--weights", str(ckdir)])
        os.environ[weights.ENV_VAR] = str(ckdir / "dinov2_vits14_reg4_pretrain.pth")
        infer.main(argv + ["--output-dir", str(tmp_path / "out_env")])
    finally:
        torch.hub.set_dir(old_hub)
        os.environ.pop(weights.ENV_VAR, None)
        if env is not None:
            os.environ[weights.ENV_VAR] = env
    frames = lambda lid: iter([{"scene_id": 1, "im_id": 3, "image": sc["image"], "camera": cam}])
    infer.infer(sc["opts"], frames, infer_pose_util.load_detections_in_bop_format(str(sc["det_path"])), {1: repre_util.load_object_repre(sc["rdir"])},
                str(tmp_path / "out_mem"), extractor=ex, num_target_insts={1: {(1, 3): 1}})
    want = json.load(open(tmp_path / "out_mem" / "1" / "estimated-poses.json"))
    assert len(want) == 2
    for tag in ("out_cli", "out_env"):
        got = json.load(open(tmp_path / tag / "1" / "estimated-poses.json"))
        assert [(e["inst_id"], e["R"], e["t"], e["score"]) for e in got] == [(e["inst_id"], e["R"], e["t"], e["score"]) for e in want], tag
    # another checkpoint gives other poses scores: the file is really what is read (not a seed default)
    torch.save(synthetic.make_vit_state_dict(ARCHS["vits14-reg"], seed=78), ckdir / "other.pth")
    infer.main(argv + ["--output-dir", str(tmp_path / "out_other"), "--weights", str(ckdir / "other.pth")])
    p = tmp_path / "out_other" / "1" / "estimated-poses.json"
    other = json.load(open(p)) if p.exists() else []
    assert [(e["R"], e["score"]) for e in other] != [(e["R"], e["score"]) for e in want]


def test_infer_driver_without_cropping(tmp_path):
    """crop=False (scripts/infer.py:355-357, 411-416): the whole image and each instance's modal mask go to the extractor as they are, the grid
    covers the image, the original camera is the one the poses are solved in.  Planted like the cropped scene: the bank holds the image's own
    features under each instance's mask as templates 2 and 5, their vertices come from known poses in the ORIGINAL camera."""
    g = torch.Generator().manual_seed(1)
    H, W = 224, 336          # multiples of the patch size, as the backbone's patch embedding demands of an uncropped input
    image = (torch.rand(H, W, 3, generator=g) * 255).to(torch.uint8).numpy()
    cam = crop_util.PinholePlaneCameraModel(W, H, (400.0, 410.0), (168.0, 110.0), np.eye(4))
    boxes_xywh = [[20, 30, 150, 160], [180, 20, 140, 180]]
    masks = np.zeros((2, H, W), np.uint8)
    for b, (x, y, w, h) in enumerate(boxes_xywh):
        masks[b, y + 6:y + h - 6, x + 6:x + w - 6] = 1
    dets = [{"scene_id": 2, "image_id": 7, "category_id": 1, "bbox": boxes_xywh[b], "score": 0.9 - 0.1 * b, "time": 0.1,
             "segmentation": infer_pose_util.binary_mask_to_rle(masks[b])} for b in range(2)]
    det_path = tmp_path / "cnos.json"
    det_path.write_text(json.dumps(dets))
    opts = infer.load_opts({"infer_opts": {
        "version": "v1", "object_dataset": "synth", "repre_version": "v1", "object_lids": [1], "crop": False, "use_detections": True,
        "extractor_name": NAME, "grid_cell_size": 14.0, "match_top_n_templates": 5, "match_top_k_buddies": 300, "pnp_ransac_iter": 400,
        "pnp_inlier_thresh": 10.0, "num_preds_factor": 2, "vis_results": False}})
    ex = feature_util.make_feature_extractor(NAME, random_init_seed=1234, precision="fp32").to("cuda")
    img_f = torch.from_numpy(image).cuda().float() / 255.0
    T = 9
    tpl = torch.rand(T, 3, H, W, generator=g).cuda()
    tmask = torch.zeros(T, H, W, dtype=torch.uint8).cuda()
    tmask[:, 40:180, 60:300] = 1
    slots = [2, 5]
    for b, s in enumerate(slots):
        tpl[s], tmask[s] = img_f.permute(2, 0, 1), torch.from_numpy(masks[b]).cuda()
    feats, f2t, pts = bank_builder.extract_template_features(ex, tpl, tmask)
    verts = torch.randn(feats.shape[0], 3, generator=g).cuda() * 50.0
    R = workload._random_rotations(2, g)
    t = torch.tensor([[5.0, -12.0, 800.0], [-20.0, 9.0, 950.0]], dtype=torch.float64)
    K = torch.tensor([[cam.f[0], 0, cam.c[0]], [0, cam.f[1], cam.c[1]], [0, 0, 1.0]], dtype=torch.float64)
    for b, s in enumerate(slots):
        rows = f2t == s
        verts[rows] = workload.planted_vertices(pts[rows], K, R[b], t[b]).cuda()
    repre = bank_builder.build_object_repre(feats, f2t, verts, T, pca_components=128, cluster_num=64, cluster_iters=10)
    frames = lambda lid: iter([{"scene_id": 2, "im_id": 7, "image": image, "camera": cam}])
    out_dir = str(tmp_path / "inference")
    infer.infer(opts, frames, infer_pose_util.load_detections_in_bop_format(str(det_path)), {1: repre}, out_dir, extractor=ex, num_target_insts={1: {(2, 7): 1}})
    est = json.load(open(os.path.join(out_dir, "1", "estimated-poses.json")))
    assert len(est) == 2
    for e in est:
        b = int(e["inst_id"])
        assert np.abs(np.array(e["R"]) - R[b].numpy()).max() < 1e-4
        assert np.linalg.norm(np.array(e["t"]).ravel() - t[b].numpy()) / np.linalg.norm(t[b].numpy()) < 1e-4
    # an image that does not tile into patches is refused, like the backbone's patch embedding refuses it in the reference
    H2 = 230
    masks2 = np.zeros((2, H2, W), np.uint8)
    masks2[:, 40:200, 30:150] = 1
    dets2 = [{"scene_id": 2, "image_id": 7, "category_id": 1, "bbox": [30, 40, 120, 160], "score": 0.9, "time": 0.1,
              "segmentation": infer_pose_util.binary_mask_to_rle(masks2[0])}]
    (tmp_path / "cnos2.json").write_text(json.dumps(dets2))
    cam2 = crop_util.PinholePlaneCameraModel(W, H2, (400.0, 410.0), (168.0, 110.0), np.eye(4))
    frames2 = lambda lid: iter([{"scene_id": 2, "im_id": 7, "image": np.zeros((H2, W, 3), np.uint8), "camera": cam2}])
    with pytest.raises(AssertionError, match="not a multiple of patch size"):
        infer.infer(opts, frames2, infer_pose_util.load_detections_in_bop_format(str(tmp_path / "cnos2.json")), {1: repre}, str(tmp_path / "bad"), extractor=ex,
                    num_target_insts={1: {(2, 7): 1}})
