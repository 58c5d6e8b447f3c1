"""Object representation from rendered templates (/root/reference/scripts/gen_repre.py:37-400), built on the MI355X:
template metadata + RGB / depth / mask images in, `repre.pth` out, with the reference's option names and file layout.

Per object: batched extractor forwards over the templates with the patch features registered in 3D through the rendered
depth (bank_builder.register_templates_in_3d), PCA fitted on the device (projector_util.PCAProjector.fit), k-means visual
words (cluster_util.kmeans), tf-idf template descriptors (bank_builder.calc_tfidf_descriptors), a 3-component PCA for
visualisation, saved with repre_util.save_object_repre.  Rendering the templates themselves (scripts/gen_templates.py)
is upstream of this step and outside the path.

  python -m foundpose_amd.gen_repre --opts configs/gen_repre/lmo.json --output-path <output>
"""

import argparse
import json
import os
from typing import Any, Dict, List, NamedTuple, Optional, Sequence

import numpy as np
import torch

from . import bank_builder, cluster_util, crop_util, feature_util, projector_util, repre_util


class GenRepreOpts(NamedTuple):
    """Options of scripts/gen_repre.py:37-64 (same names, same defaults)."""
    version: str
    templates_version: str
    object_dataset: str
    object_lids: Optional[List[int]] = None
    extractor_name: str = "dinov2_vits14_reg"
    grid_cell_size: float = 14.0
    apply_pca: bool = True
    pca_components: int = 256
    pca_whiten: bool = False
    pca_max_samples_for_fitting: int = 100000
    cluster_features: bool = True
    cluster_num: int = 2048
    template_desc_opts: Optional[repre_util.TemplateDescOpts] = None
    overwrite: bool = True
    debug: bool = True


def load_opts(path_or_dict) -> GenRepreOpts:
    d = path_or_dict
    if not isinstance(d, dict):
        with open(path_or_dict) as f:
            d = json.load(f)
    d = dict(d.get("gen_repre_opts", d))
    if d.get("template_desc_opts") is not None and not isinstance(d["template_desc_opts"], repre_util.TemplateDescOpts):
        d["template_desc_opts"] = repre_util.TemplateDescOpts(**d["template_desc_opts"])
    return GenRepreOpts(**d)


def _load_image(path: str) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.open(path))


def load_template_metadata(output_path: str, opts: GenRepreOpts, object_lid: int) -> List[Dict[str, Any]]:
    """<output>/templates/<templates_version>/<dataset>/<lid>/metadata.json (gen_repre.py:84-93)."""
    with open(os.path.join(output_path, "templates", opts.templates_version, opts.object_dataset, str(object_lid), "metadata.json")) as f:
        return json.load(f)


def generate_raw_repre(opts: GenRepreOpts, object_dataset: str, object_lid: int, extractor, metadata: List[Dict[str, Any]],
                       batch_size: int = 32) -> repre_util.FeatureBasedObjectRepre:
    """gen_repre.py:67-214 for all templates of one object, batched: features registered in 3D + the template images and cameras."""
    templates, depths, masks, cams, T_mfc = [], [], [], [], []
    for data_id, s in enumerate(metadata):
        assert s["dataset"] == object_dataset and s["lid"] == object_lid and s["template_id"] == data_id
        c = s["cameras"]
        cam = crop_util.PinholePlaneCameraModel(c["ImageSizeX"], c["ImageSizeY"], (c["fx"], c["fy"]), (c["cx"], c["cy"]), np.array(c["T_WorldFromCamera"]))
        img = _load_image(s["rgb_image_path"])
        if img.ndim == 2:
            img = np.stack([img] * 3, -1)
        templates.append(torch.from_numpy(np.array(img[..., :3])).permute(2, 0, 1))
        depths.append(torch.from_numpy(_load_image(s["depth_map_path"]).astype(np.float32)))
        m = _load_image(s["binary_mask_path"])
        masks.append(torch.from_numpy((m if m.ndim == 2 else m[..., 0]).astype(np.float32)))
        T_wfm = np.eye(4)
        T_wfm[:3, :3], T_wfm[:3, 3:] = np.array(s["pose"]["R"], np.float64).reshape(3, 3), np.array(s["pose"]["t"], np.float64).reshape(3, 1)
        # float32 like the reference (gen_repre.py:150-161): T_model_from_camera = inv(T_world_from_model) @ T_world_from_camera
        T_mfw = torch.linalg.inv(torch.from_numpy(T_wfm).to(torch.float32))
        T_mfc.append(T_mfw @ torch.from_numpy(cam.T_world_from_eye).to(torch.float32))
        cams.append(cam)
    tpl_u8 = torch.stack(templates)
    feats, f2t, verts, f2v = bank_builder.register_templates_in_3d(
        extractor, tpl_u8.to(torch.float32) / 255.0, torch.stack(depths), torch.stack(masks), cams, torch.stack(T_mfc), opts.grid_cell_size, batch_size)
    return repre_util.FeatureBasedObjectRepre(
        vertices=verts, feat_vectors=feats, feat_opts=repre_util.FeatureOpts(extractor_name=opts.extractor_name), feat_to_vertex_ids=f2v,
        feat_to_template_ids=f2t, templates=tpl_u8,
        template_cameras_cam_from_model=[{"f": torch.tensor(c.f), "c": torch.tensor(c.c), "width": c.width, "height": c.height,
                                          "T_world_from_eye": torch.tensor(c.T_world_from_eye)} for c in cams])


def finish_repre(opts: GenRepreOpts, repre: repre_util.FeatureBasedObjectRepre) -> repre_util.FeatureBasedObjectRepre:
    """gen_repre.py:271-365: PCA, visual words, template descriptors, the visualisation projector."""
    feats = repre.feat_vectors.float().cuda()
    if opts.apply_pca:
        proj = projector_util.PCAProjector(n_components=opts.pca_components, whiten=opts.pca_whiten)
        proj.fit(feats, max_samples=opts.pca_max_samples_for_fitting)
        repre.feat_raw_projectors.append(proj)
        feats = proj.transform(feats)
        repre.feat_vectors = feats
    if opts.cluster_features:
        centroids, cluster_ids, _ = cluster_util.kmeans(feats, opts.cluster_num, verbose=False)
        repre.feat_cluster_centroids, repre.feat_to_cluster_ids = centroids, cluster_ids
    if opts.template_desc_opts is not None:
        repre.template_desc_opts = opts.template_desc_opts
        if opts.template_desc_opts.desc_type == "tfidf":
            assert repre.feat_cluster_centroids is not None and repre.feat_to_cluster_ids is not None and repre.templates is not None
            descs, idfs, _ = bank_builder.calc_tfidf_descriptors(feats, repre.feat_to_template_ids.cuda(), repre.feat_cluster_centroids,
                                                                 len(repre.templates), opts.template_desc_opts, feat_to_word_ids=repre.feat_to_cluster_ids)
            repre.template_descs, repre.feat_cluster_idfs = descs, idfs
        else:
            raise ValueError(f"Unknown template descriptor type {opts.template_desc_opts.desc_type}.")
    # visualisation projector (gen_repre.py:349-363): the raw-feature PCA is reused when there is one -- consumers apply
    # feat_vis_projectors to RAW extractor features -- and a 3-component PCA is fitted only when PCA was not applied
    if len(repre.feat_raw_projectors) and isinstance(repre.feat_raw_projectors[0], projector_util.PCAProjector):
        repre.feat_vis_projectors = [repre.feat_raw_projectors[0]]
    else:
        vis = projector_util.PCAProjector(n_components=3, whiten=False)
        vis.fit(feats, max_samples=opts.pca_max_samples_for_fitting)
        repre.feat_vis_projectors = [vis]
    return repre


def generate_repre(opts: GenRepreOpts, dataset: str, lid: int, output_path: str, extractor=None, metadata: Optional[List[Dict[str, Any]]] = None,
                   precision: str = "bf16", weights: Optional[str] = None) -> str:
    """-> the directory holding repre.pth and config.json (<output>/object_repre/<dataset>/<version>/<lid>)."""
    out_dir = repre_util.get_object_repre_dir_path(os.path.join(output_path, "object_repre"), opts.version, dataset, lid)
    if os.path.exists(out_dir) and not opts.overwrite:
        raise ValueError(f"Output directory already exists: {out_dir}")
    os.makedirs(out_dir, exist_ok=True)
    cfg = opts._asdict()
    if opts.template_desc_opts is not None:
        cfg["template_desc_opts"] = opts.template_desc_opts._asdict()
    with open(os.path.join(out_dir, "config.json"), "w") as f:
        json.dump(cfg, f, indent=2)
    if extractor is None:  # gen_repre.py:250-251; raises without a checkpoint (weights.py)
        extractor = feature_util.make_feature_extractor(opts.extractor_name, precision=precision, weights=weights).to("cuda")
    if metadata is None:
        metadata = load_template_metadata(output_path, opts, lid)
    repre = finish_repre(opts, generate_raw_repre(opts, dataset, lid, extractor, metadata))
    if getattr(extractor, "precision", None) == "fp8" and extractor.act_scales is not None:
        repre.extractor_fp8_act_scales = extractor.act_scales.tolist()  # the bank carries the scales it was extracted with
    repre_util.save_object_repre(repre, out_dir)
    return out_dir


def main(argv: Optional[Sequence[str]] = None) -> None:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--opts", required=True)
    ap.add_argument("--output-path", required=True, help="root holding templates/ (input) and object_repre/ (output)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f16", "f16x3", "f16f8", "fp32"])
    ap.add_argument("--weights", default=None, help="DINOv2 checkpoint file (upstream key names) or a directory holding the upstream file name; "
                    "default $FOUNDPOSE_DINOV2_WEIGHTS, then the torch hub cache; without one the run fails (no random-weight fallback)")
    args = ap.parse_args(argv)
    opts = load_opts(args.opts)
    ex = feature_util.make_feature_extractor(opts.extractor_name, precision=args.precision, weights=args.weights).to("cuda")
    for lid in opts.object_lids or []:
        print(generate_repre(opts, opts.object_dataset, lid, args.output_path, ex))


if __name__ == "__main__":
    main()
