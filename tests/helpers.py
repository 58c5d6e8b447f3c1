"""Shared test helpers: rebuild fixture inputs, compare correspondence lists."""

import os

import numpy as np
import torch

from oracle.make_golden import MATCH_CASES, TINY, TINY0, build_match_inputs, checksum  # noqa: F401

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def match_case_inputs(name):
    """-> (case dict, golden dict, repre dict (numpy, reference-built descs), pts, feats)."""
    c = MATCH_CASES[name]
    g = load_golden(name)
    bank, centroids, pts, feats = build_match_inputs(c)
    assert np.isclose(checksum(bank["feat_vectors"], centroids, pts, feats), g["input_checksum"], rtol=0, atol=1e-6), \
        "seeded inputs differ from the ones the fixture was generated with"
    repre = {
        "vertices": bank["vertices"].numpy(),
        "feat_vectors": bank["feat_vectors"].numpy(),
        "feat_to_vertex_ids": bank["feat_to_vertex_ids"].numpy(),
        "feat_to_template_ids": bank["feat_to_template_ids"].numpy(),
        "feat_to_cluster_ids": g["feat_to_cluster_ids"],
        "feat_cluster_centroids": centroids.numpy(),
        "feat_cluster_idfs": g["word_idfs"],
        "template_descs": g["template_descs"],
        "template_desc_opts": {"desc_type": "tfidf", "tfidf_knn_metric": "l2", "tfidf_knn_k": 3,
                               "tfidf_soft_assign": bool(c["soft"]), "tfidf_soft_sigma_squared": 10.0},
    }
    return c, g, repre, pts.numpy(), feats.numpy()


NOREG_CASES = (("vitl14", 518), ("vitl14", 420), ("vits14", 420), ("vitb14", 518), ("vitb14", 420), ("vitg14", 224))


def noreg_case(version, S):
    """Inputs of tests/golden/extractor_<version>_<S>.npz (the non-register hub entries, scripts/infer.py:75) -> (g, name, spec, sd, imgs)."""
    from foundpose_amd import synthetic
    from foundpose_amd.vit_config import parse_extractor_name
    g = load_golden(f"extractor_{version}_{S}")
    name = str(g["name"])
    spec = parse_extractor_name(name)
    assert spec.version == version and spec.layer == int(g["layer"]) and spec.arch.registers == 0
    sd = synthetic.make_vit_state_dict(spec.arch, seed=int(g["weights_seed"]))
    imgs = synthetic.make_crops(1, S, seed=int(g["image_seed"]))
    assert np.isclose(checksum(imgs, sd[f"blocks.{spec.layer}.attn.qkv.weight"], sd["pos_embed"]), g["input_checksum"], atol=1e-6)
    return g, name, spec, sd, imgs
