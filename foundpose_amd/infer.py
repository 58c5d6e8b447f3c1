"""The inference driver (/root/reference/scripts/infer.py:55-100, 290-816) over the MI355X path: options with the
reference's names and defaults, CNOS detections in, `estimated-poses.json` per object out.

What differs from the reference loop, by design: the instances of an image go through the crop producer, the extractor,
the matching and the PnP tail as ONE batch on the device (the reference handles them one at a time on the CPU), and the
evaluation / rendering / visualisation branches (ground-truth errors, HTML) are not part of this path.

  python -m foundpose_amd.infer --opts configs/infer/lmo.json --dataset-dir <bop split dir> --detections <cnos json> \\
         --repre-dir <output>/object_repre --output-dir <output>/inference
"""

import argparse
import json
import os
import time
from typing import Any, Dict, Iterable, List, NamedTuple, Optional, Sequence, Tuple

import numpy as np
import torch

from . import crop_util, engine as fe, eval_util, feature_util, infer_pose_util, pnp_util, repre_util
from .bank import DeviceBank


class InferOpts(NamedTuple):
    """Options of scripts/infer.py:55-100 (same names, same defaults)."""
    version: str
    repre_version: str
    object_dataset: str
    object_lids: Optional[List[int]] = None
    max_sym_disc_step: float = 0.01
    crop: bool = True
    crop_rel_pad: float = 0.2
    crop_size: Tuple[int, int] = (420, 420)
    use_detections: bool = True
    num_preds_factor: float = 1.0
    min_visibility: float = 0.1
    extractor_name: str = "dinov2_vitl14"
    grid_cell_size: float = 1.0
    max_num_queries: int = 1000000
    match_template_type: str = "tfidf"
    match_top_n_templates: int = 5
    match_feat_matching_type: str = "cyclic_buddies"
    match_top_k_buddies: int = 300
    pnp_type: str = "opencv"
    pnp_ransac_iter: int = 1000
    pnp_required_ransac_conf: float = 0.99
    pnp_inlier_thresh: float = 10.0
    pnp_refine_lm: bool = True
    final_pose_type: str = "best_coarse"
    save_estimates: bool = True
    vis_results: bool = True
    vis_corresp_top_n: int = 100
    vis_feat_map: bool = True
    vis_for_paper: bool = True
    debug: bool = True


def load_opts(path_or_dict) -> InferOpts:
    """configs/infer/*.json: {"infer_opts": {...}}; unknown keys are an error, like NamedTuple construction in the reference."""
    d = path_or_dict
    if not isinstance(d, dict):
        with open(path_or_dict) as f:
            d = json.load(f)
    d = dict(d.get("infer_opts", d))
    if "crop_size" in d:
        d["crop_size"] = tuple(d["crop_size"])
    return InferOpts(**d)


def infer_object(opts: InferOpts, object_lid: int, repre: repre_util.FeatureBasedObjectRepre, frames: Iterable[Dict[str, Any]],
                 detections: Dict[Any, Any], extractor=None, num_target_insts: Optional[Dict[Tuple[int, int], int]] = None,
                 precision: str = "bf16", seed: int = 0, weights: Optional[str] = None) -> eval_util.PoseEvaluator:
    """One object over a stream of frames (the body of infer.py's per-object loop).  A frame is
    {"scene_id", "im_id", "image": HWC uint8 or float [0,1] (numpy or tensor), "camera": PinholePlaneCameraModel (c2w)}."""
    if opts.match_template_type != "tfidf":
        raise ValueError(f"Unknown matching type '{opts.match_template_type}'.")
    if opts.match_feat_matching_type != "cyclic_buddies":
        raise ValueError(f"Unknown feature matching type ({opts.match_feat_matching_type}).")
    if opts.final_pose_type != "best_coarse":
        raise ValueError(f"Unknown final pose type {opts.final_pose_type}")
    # scripts/infer.py:482-485 subsamples the query points with torch.randperm when a mask yields more than max_num_queries of them
    # (default 1 000 000: never for a crop).  The batched path keeps every point; an option value that could trigger the subsampling
    # is refused instead of being ignored (crop=False: checked per frame against the image's own grid).
    def check_max_queries(size_wh):
        max_points = int(size_wh[0] // opts.grid_cell_size) * int(size_wh[1] // opts.grid_cell_size)
        if opts.max_num_queries < max_points:
            raise NotImplementedError(f"max_num_queries={opts.max_num_queries} could subsample the {max_points} grid points of a {size_wh[0]}x{size_wh[1]} input: not on the batched path")
    if opts.crop:
        check_max_queries(opts.crop_size)
    if extractor is None:  # infer.py:125-128; the checkpoint: weights=, $FOUNDPOSE_DINOV2_WEIGHTS or the torch hub cache, else this raises
        extractor = feature_util.make_feature_extractor(opts.extractor_name, precision=precision, weights=weights).to("cuda")
    bank = DeviceBank([repre])
    eng = fe.FoundPoseEngine(extractor, bank, opts.grid_cell_size, opts.match_top_n_templates, opts.match_top_k_buddies, tie_order="torch")
    eng.record_stage_times = True
    evaluator = eval_util.PoseEvaluator()
    vertices = repre.vertices.cpu().numpy()
    for frame in frames:
        scene_id, im_id, cam = frame["scene_id"], frame["im_id"], frame["camera"]
        # number of target instances (infer.py:308-321): from the test targets when given -- frames that are not a target of
        # this object, or whose count is 0, are skipped -- otherwise the number of ground-truth annotations of the frame
        # ground-truth annotations of this object that are sufficiently visible (infer.py:286-305): a frame that HAS annotations but
        # none of them qualifies is skipped; a frame without annotations (sample.objects_anno is None) goes on with an empty list
        object_annos = []
        if frame.get("gt_annos") is not None:
            object_annos = [a for a in frame["gt_annos"]
                            if getattr(a, "lid", object_lid) == object_lid and not np.isnan(getattr(a, "visibilities", 1.0))
                            and getattr(a, "visibilities", 1.0) > opts.min_visibility]
            if len(object_annos) == 0:
                continue
        if num_target_insts is not None:
            if (scene_id, im_id) not in num_target_insts:
                continue
            n_target = int(num_target_insts[(scene_id, im_id)])
        else:
            n_target = len(object_annos)     # infer.py:317: no targets and no annotations -> 0 -> the frame is skipped
        if n_target == 0:
            continue
        instances = infer_pose_util.get_instances_for_pose_estimation(
            scene_id, im_id, object_lid, opts.use_detections, detections, int(opts.num_preds_factor * n_target), object_annos,
            (cam.width, cam.height))
        kept = []
        for inst_j, inst in enumerate(instances):
            evaluator.detection_times[(scene_id, im_id)] = inst.get("time", 0) if opts.use_detections else 0
            if inst["input_mask_modal"].sum() > cam.width * cam.height:  # infer.py:388-392
                continue
            if inst["input_mask_modal"].sum() == 0:
                continue
            kept.append((inst_j, inst))
        if not kept:
            continue
        t0 = time.perf_counter()
        img = frame["image"]
        if not isinstance(img, torch.Tensor):
            arr = np.asarray(img)
            img = torch.from_numpy(arr if arr.flags.writeable else arr.copy())   # (PIL hands out read-only arrays)
        img = (img.to("cuda", torch.float32) / 255.0) if img.dtype == torch.uint8 else img.to("cuda", torch.float32)
        masks = torch.from_numpy(np.stack([i["input_mask_modal"] for _, i in kept]).astype(np.uint8)).cuda()
        boxes = [i["input_box_amodal"].tolist() for _, i in kept]
        if opts.crop:
            crops, crop_masks, cams = crop_util.crop_detections(img, masks, boxes, cam, tuple(opts.crop_size), opts.crop_rel_pad)
        else:
            # crop=False (infer.py:355-357, 411-416): the whole image and the modal mask of every instance go to the extractor unchanged, the
            # original camera stays the camera the poses are solved in.  The image must tile into patches -- the backbone's patch embedding
            # asserts it in the reference as well (every instance then shares one feature map; it is computed per instance here, like there).
            ps = extractor.patch_size
            if img.shape[0] % ps or img.shape[1] % ps:
                raise AssertionError(f"Input image height {img.shape[0]} / width {img.shape[1]} is not a multiple of patch size {ps} (crop=False)")
            check_max_queries((cam.width, cam.height))
            crops = img.permute(2, 0, 1)[None].expand(len(kept), -1, -1, -1).contiguous()
            crop_masks, cams = masks, [cam] * len(kept)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        res = eng.infer_batch(crops, crop_masks, [0] * len(kept))
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        poses = pnp_util.estimate_poses(res, cams, opts.pnp_type, opts.pnp_ransac_iter, opts.pnp_inlier_thresh, opts.pnp_required_ransac_conf,
                                        opts.pnp_refine_lm, seed=seed)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        best = pnp_util.select_best_coarse(poses)
        found, cid = best["found"].cpu().tolist(), best["corresp_id"].cpu().tolist()
        Rb, tb = best["R"].cpu().numpy(), best["t"].cpu().numpy()
        t4 = time.perf_counter()
        n = len(kept)
        # The reference's per-detection `times` keys (infer.py:464-633, persisted by eval_util.py:327).  The instances of a frame
        # run as ONE batch here, so every instance is charged its share of the batch; the four stages inside infer_batch come
        # from HIP events on the stream (their sum is the device time of t1..t2, the host-side remainder is in feat_extract).
        st = eng.stage_times()
        dev_sum = sum(st.values())
        times = {"prep": (t1 - t0) / n,
                 "feat_extract": (st.get("feat_extract", 0.0) + max(0.0, (t2 - t1) - dev_sum)) / n,
                 "grid_sample": st.get("grid_sample", 0.0) / n, "proj": st.get("proj", 0.0) / n, "corresp": st.get("corresp", 0.0) / n,
                 "pose_coarse": (t3 - t2) / n, "final_select": (t4 - t3) / n}
        for b, (inst_j, inst) in enumerate(kept):
            if not found[b]:
                continue
            c = res.corresp_list(b)[cid[b]]
            corresp_np = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in c.items()}
            T_m2c = np.eye(4)
            T_m2c[:3, :3], T_m2c[:3, 3] = Rb[b], tb[b]
            T_m2w = cams[b].T_world_from_eye @ T_m2c  # infer.py:661-666
            evaluator.update_without_anno(scene_id, im_id, inst_j, 0, vertices, object_lid, T_m2w[:3, :3], T_m2w[:3, 3], cam, cams[b], times, corresp_np,
                                          inlier_radius=10)
    return evaluator


def infer(opts: InferOpts, frames_by_object, detections, repres: Dict[int, repre_util.FeatureBasedObjectRepre], output_dir: str, extractor=None,
          precision: str = "bf16", num_target_insts: Optional[Dict[int, Dict[Tuple[int, int], int]]] = None, weights: Optional[str] = None) -> List[str]:
    """All objects: `frames_by_object(lid)` yields the frames that show object `lid`; one estimated-poses.json per object
    under <output_dir>/<lid>/ (infer.py:813-816), then the BOP19 csv.
    num_target_insts: {object lid: {(scene_id, im_id): inst_count}} from test_targets_bop19.json -- the number of poses to
    estimate per (image, object) is num_preds_factor x inst_count (infer.py:308-346); frames without an entry are skipped."""
    lids = list(opts.object_lids) if opts.object_lids is not None else sorted(repres)
    if extractor is None:
        extractor = feature_util.make_feature_extractor(opts.extractor_name, precision=precision, weights=weights).to("cuda")
    paths = []
    for lid in lids:
        ev = infer_object(opts, lid, repres[lid], frames_by_object(lid), detections, extractor,
                          num_target_insts=None if num_target_insts is None else num_target_insts.get(lid, {}))
        if opts.save_estimates:
            p = os.path.join(output_dir, str(lid), "estimated-poses.json")
            ev.save_results_json(p)
            paths.append(p)
    if opts.save_estimates:
        paths.append(eval_util.prepare_bop_submission(output_dir, opts.object_dataset, lids))
    return paths


# ---------------------------------------------------------------------------------------------------- BOP split on disk
def load_bop_frames(split_dir: str, targets: Sequence[Dict[str, int]], object_lid: int):
    """Frames of a BOP split that show `object_lid` according to test_targets_bop19.json entries
    ({"scene_id", "im_id", "obj_id", "inst_count"}): <split>/<scene:06d>/rgb/<im:06d>.{png,jpg} + scene_camera.json (cam_K)."""
    from PIL import Image
    cams: Dict[int, Dict[str, Any]] = {}
    for tgt in targets:
        if tgt["obj_id"] != object_lid:
            continue
        sid, iid = tgt["scene_id"], tgt["im_id"]
        sdir = os.path.join(split_dir, f"{sid:06d}")
        if sid not in cams:
            with open(os.path.join(sdir, "scene_camera.json")) as f:
                cams[sid] = json.load(f)
        K = np.array(cams[sid][str(iid)]["cam_K"], np.float64).reshape(3, 3)
        path = next(p for p in (os.path.join(sdir, "rgb", f"{iid:06d}.png"), os.path.join(sdir, "rgb", f"{iid:06d}.jpg"),
                                os.path.join(sdir, "gray", f"{iid:06d}.tif")) if os.path.exists(p))
        image = np.asarray(Image.open(path).convert("RGB"))
        camera = crop_util.PinholePlaneCameraModel(image.shape[1], image.shape[0], (K[0, 0], K[1, 1]), (K[0, 2], K[1, 2]), np.eye(4))
        yield {"scene_id": sid, "im_id": iid, "image": image, "camera": camera}


def main(argv: Optional[Sequence[str]] = None) -> None:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--opts", required=True, help="options JSON ({'infer_opts': {...}}, e.g. the reference's configs/infer/lmo.json)")
    ap.add_argument("--dataset-dir", required=True, help="BOP split directory (<datasets>/<dataset>/<split>)")
    ap.add_argument("--targets", default=None, help="test_targets_bop19.json (default: <dataset-dir>/../test_targets_bop19.json)")
    ap.add_argument("--detections", required=True, help="CNOS detections in the BOP format")
    ap.add_argument("--repre-dir", required=True, help="<output>/object_repre (repre.pth under <version>/<dataset>/<lid>/)")
    ap.add_argument("--output-dir", required=True)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f16", "f16x3", "f16f8", "fp32"])
    ap.add_argument("--weights", default=None, help="DINOv2 checkpoint: a .pth with the upstream key names, or a directory holding the upstream file "
                    "(dinov2_vitl14_pretrain.pth, dinov2_vits14_reg4_pretrain.pth, ...); default $FOUNDPOSE_DINOV2_WEIGHTS, then the torch hub cache. "
                    "Without a checkpoint the run fails: there is no random-weight fallback")
    args = ap.parse_args(argv)
    opts = load_opts(args.opts)
    # the checkpoint is resolved before anything else is read: a missing one must fail in seconds, not after the banks are loaded
    extractor = feature_util.make_feature_extractor(opts.extractor_name, precision=args.precision, weights=args.weights)
    with open(args.targets or os.path.join(os.path.dirname(os.path.abspath(args.dataset_dir)), "test_targets_bop19.json")) as f:
        targets = json.load(f)
    detections = infer_pose_util.load_detections_in_bop_format(args.detections)
    lids = opts.object_lids or sorted({t["obj_id"] for t in targets})
    repres = {lid: repre_util.load_object_repre(repre_util.get_object_repre_dir_path(args.repre_dir, opts.repre_version, opts.object_dataset, lid)) for lid in lids}
    n_inst: Dict[int, Dict[Tuple[int, int], int]] = {}
    for t in targets:
        n_inst.setdefault(t["obj_id"], {})[(t["scene_id"], t["im_id"])] = t["inst_count"]
    out = infer(opts._replace(object_lids=list(lids)), lambda lid: load_bop_frames(args.dataset_dir, targets, lid), detections, repres, args.output_dir,
                extractor=extractor.to("cuda"), precision=args.precision, num_target_insts=n_inst)
    print("\n".join(out))


if __name__ == "__main__":
    main()
