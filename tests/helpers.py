"""Shared test helpers: rebuild fixture inputs, compare correspondence lists."""

import os

import numpy as np
import torch

from oracle.make_golden import MATCH_CASES, TINY, build_match_inputs, checksum  # noqa: F401

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
