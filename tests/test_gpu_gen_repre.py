"""gen_repre end to end (/root/reference/scripts/gen_repre.py:67-377): a templates directory in the layout
gen_templates writes (metadata.json + RGB / 16-bit depth / mask PNGs) -> repre.pth, then the inference path on it."""
import json
import os

import numpy as np
import pytest
import torch

from foundpose_amd import engine as fe
from foundpose_amd import feature_util, gen_repre, repre_util
from foundpose_amd.bank import DeviceBank

pytestmark = pytest.mark.gpu
NAME = "dinov2_version=vits14-reg_stride=14_facet=token_layer=9_logbin=0_norm=1"


def _write_templates(root, T, S, rng):
    from PIL import Image
    tdir = os.path.join(root, "templates", "v1", "synth", "4")
    os.makedirs(tdir, exist_ok=True)
    meta = []
    yy, xx = np.mgrid[0:S, 0:S]
    for t in range(T):
        rgb = (rng.random((S, S, 3)) * 255).astype(np.uint8)
        mask = ((((xx - S / 2) / (0.38 * S)) ** 2 + ((yy - S / 2) / (0.30 * S)) ** 2) <= 1.0).astype(np.uint8) * 255
        depth = (600.0 + 40.0 * np.sin(xx / 30.0 + t) + 25.0 * np.cos(yy / 25.0)).astype(np.uint16)   # mm
        paths = {k: os.path.join(tdir, f"{k}_{t:04d}.png") for k in ("rgb", "depth", "mask")}
        Image.fromarray(rgb).save(paths["rgb"])
        Image.fromarray(depth).save(paths["depth"])
        Image.fromarray(mask).save(paths["mask"])
        Rw = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        Rw *= np.sign(np.linalg.det(Rw))
        Twc = np.eye(4)
        Twc[:3, :3], Twc[:3, 3] = Rw, rng.normal(size=3) * 100
        Rm = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        Rm *= np.sign(np.linalg.det(Rm))
        meta.append({"dataset": "synth", "lid": 4, "template_id": t, "rgb_image_path": paths["rgb"], "depth_map_path": paths["depth"],
                     "binary_mask_path": paths["mask"], "pose": {"R": Rm.tolist(), "t": (rng.normal(size=(3, 1)) * 50).tolist()},
                     "cameras": {"ImageSizeX": S, "ImageSizeY": S, "fx": 500.0, "fy": 505.0, "cx": S / 2.0, "cy": S / 2.0 - 1.0, "T_WorldFromCamera": Twc.tolist()}})
    with open(os.path.join(tdir, "metadata.json"), "w") as f:
        json.dump(meta, f)
    return meta


def test_gen_repre_from_templates_dir_then_inference(tmp_path):
    rng = np.random.default_rng(0)
    T, S = 20, 224
    meta = _write_templates(str(tmp_path), T, S, rng)
    opts = gen_repre.load_opts({"gen_repre_opts": {"version": "v1", "templates_version": "v1", "object_dataset": "synth", "object_lids": [4],
                                                   "extractor_name": NAME, "grid_cell_size": 14.0, "apply_pca": True, "pca_components": 64,
                                                   "cluster_features": True, "cluster_num": 48, "template_desc_opts": {"desc_type": "tfidf"}}})
    ex = feature_util.make_feature_extractor(NAME, random_init_seed=1234, precision="fp32").to("cuda")
    out_dir = gen_repre.generate_repre(opts, "synth", 4, str(tmp_path), extractor=ex)
    assert out_dir == os.path.join(str(tmp_path), "object_repre", "synth", "v1", "4") and os.path.exists(os.path.join(out_dir, "config.json"))
    r = repre_util.load_object_repre(out_dir)
    n = r.feat_vectors.shape[0]
    assert r.feat_vectors.shape == (n, 64) and r.vertices.shape == (n, 3) and r.feat_to_template_ids.shape == (n,) and n > T * 50
    assert r.templates.shape == (T, 3, S, S) and r.templates.dtype == torch.uint8 and len(r.template_cameras_cam_from_model) == T
    assert r.template_descs.shape == (T, 48) and r.feat_cluster_centroids.shape == (48, 64) and r.feat_cluster_idfs.shape == (48,)
    assert r.feat_opts.extractor_name == NAME and r.template_desc_opts.desc_type == "tfidf"
    assert len(r.feat_raw_projectors) == 1 and r.feat_raw_projectors[0].components.shape == (64, 384) and r.feat_vis_projectors[0].components.shape == (64, 384)   # the raw-feature PCA reused, like the reference (gen_repre.py:349-353)
    f2t = r.feat_to_template_ids.cpu().long()
    assert bool((f2t[1:] >= f2t[:-1]).all()) and int(f2t.max()) == T - 1
    # 3D registration: a template's vertices, moved back into its camera, project onto patch centres with the rendered depth
    for t in (0, 7):
        m = meta[t]
        T_wfm = np.eye(4)
        T_wfm[:3, :3], T_wfm[:3, 3:] = np.array(m["pose"]["R"]), np.array(m["pose"]["t"])
        T_cfm = np.linalg.inv(np.array(m["cameras"]["T_WorldFromCamera"])) @ T_wfm
        v = r.vertices[f2t == t].cpu().numpy().astype(np.float64)
        vc = v @ T_cfm[:3, :3].T + T_cfm[:3, 3]
        f_mean = 0.5 * (500.0 + 505.0)  # the reference lifts with the average focal length (feature_util.py:141-142)
        u = f_mean * vc[:, 0] / vc[:, 2] + S / 2.0
        w = f_mean * vc[:, 1] / vc[:, 2] + S / 2.0 - 1.0
        assert np.abs((u - 7) / 14 - np.round((u - 7) / 14)).max() < 2e-3 and np.abs((w - 7) / 14 - np.round((w - 7) / 14)).max() < 2e-3
        assert 500.0 < vc[:, 2].min() and vc[:, 2].max() < 700.0
    # the bank serves the inference path: every template, used as a query crop, retrieves itself first
    eng = fe.FoundPoseEngine(ex, DeviceBank([r]), 14.0, 5, 300, tie_order="torch")
    crops = r.templates[:8].cuda().float() / 255.0
    from PIL import Image
    masks = torch.from_numpy(np.stack([np.asarray(Image.open(meta[t]["binary_mask_path"])) for t in range(8)])).cuda()
    res = eng.infer_batch(crops, masks)
    assert res.template_ids[:, 0].tolist() == list(range(8))
    assert bool((res.template_scores[:, 0] > 0.8).all())
