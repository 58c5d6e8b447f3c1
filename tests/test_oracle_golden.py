"""CPU: the oracle reproduces the reference's outputs on every committed fixture."""

import numpy as np
import pytest
import torch

from oracle import clib, match as om
from tests.helpers import MATCH_CASES, load_golden, match_case_inputs


@pytest.mark.parametrize("name", sorted(MATCH_CASES))
def test_bank_side_descriptors(name):
    c, g, repre, pts, feats = match_case_inputs(name)
    T = c["T"]
    word_ids = clib.l2_knn(repre["feat_vectors"], repre["feat_cluster_centroids"], 1)[1][:, 0]
    assert np.array_equal(word_ids, g["feat_to_cluster_ids"])
    descs, idfs = om.calc_tfidf_descriptors(
        repre["feat_vectors"], word_ids, repre["feat_to_template_ids"], repre["feat_cluster_centroids"], T,
        3, bool(c["soft"]), 10.0)
    np.testing.assert_allclose(idfs, g["word_idfs"], rtol=3e-7, atol=0)  # numpy vs torch log: 1 ulp
    np.testing.assert_allclose(descs, g["template_descs"], rtol=5e-5, atol=1e-8)  # soft-assign weights amplify the 1e-6 distance differences of the faiss stand-in


@pytest.mark.parametrize("name", sorted(MATCH_CASES))
def test_establish_correspondences_matches_reference(name):
    c, g, repre, pts, feats = match_case_inputs(name)
    out = om.establish_correspondences(pts, feats, repre, c["top_n"], c["top_k"], topk_mode="torch")
    assert [o["template_id"] for o in out] == list(g["template_ids"])
    np.testing.assert_allclose([o["template_score"] for o in out], g["template_scores"], rtol=0, atol=2e-6)
    for i, o in enumerate(out):
        # bit-exact indices, order included (torch.topk tie behaviour emulated)
        assert np.array_equal(o["coord_2d_ids"], g[f"coord_2d_ids_{i}"]), f"template {i}"
        assert np.array_equal(o["nn_vertex_ids"], g[f"nn_vertex_ids_{i}"])
        np.testing.assert_array_equal(o["coord_2d"], g[f"coord_2d_{i}"])
        np.testing.assert_array_equal(o["coord_3d"], g[f"coord_3d_{i}"])
        np.testing.assert_array_equal(o["nn_dists"], g[f"nn_dists_{i}"])
        np.testing.assert_allclose(o["coord_conf"], g[f"coord_conf_{i}"], rtol=0, atol=1e-7, equal_nan=True)


@pytest.mark.parametrize("name", sorted(MATCH_CASES))
def test_canonical_mode_same_sets_modulo_ties(name):
    """Canonical (value, index) order selects the same multiset of cycle distances."""
    c, g, repre, pts, feats = match_case_inputs(name)
    out = om.establish_correspondences(pts, feats, repre, c["top_n"], c["top_k"], topk_mode="canonical")
    assert sorted(o["template_id"] for o in out) == sorted(g["template_ids"])
    by_t = {int(t): i for i, t in enumerate(g["template_ids"])}
    for o in out:
        i = by_t[o["template_id"]]
        np.testing.assert_array_equal(np.sort(o["nn_dists"]), np.sort(g[f"nn_dists_{i}"]))


def test_topk_torch_emulation_on_ties():
    rng = np.random.default_rng(0)
    for trial in range(200):
        n = int(rng.integers(2, 1500))
        k = int(rng.integers(1, n + 1))
        v = (rng.integers(0, 6, n) * 14).astype(np.float32)
        tv, ti = torch.topk(torch.from_numpy(-v), k, sorted=True)
        ov, oi = clib.topk_torch(-v, k, True)
        assert np.array_equal(ti.numpy(), oi)


def test_cosine_word_metric_matches_reference():
    """tfidf_knn_metric = "cosine" (scripts/infer.py:218-222 -> knn_util.py:52-57, 91-100): the oracle's cosine word search, tf-idf and
    the whole establish_correspondences vs the reference run over a cosine word index (tests/golden/wrappers_cosine.npz)."""
    import os
    from foundpose_amd import repre_util
    from oracle.make_golden import build_wrapper_inputs
    from tests.helpers import GOLDEN
    g = load_golden("wrappers_cosine")
    r = repre_util.load_object_repre(os.path.join(GOLDEN, "repre_ref"), tensor_device="cpu")
    _, _, pts, feats, _, _ = build_wrapper_inputs()
    ids, dists = om.nearest_words(feats.numpy(), r.feat_cluster_centroids.numpy(), 3, "cosine")
    assert np.array_equal(ids, g["word_ids"]) and (g["word_ids"] != g["l2_word_ids"]).any()
    np.testing.assert_allclose(dists, g["word_dists"], rtol=0, atol=2e-4)   # sqrt(1 - a.b) near 0.3: the 1-ulp difference of 1 - sim is amplified by the root
    np.testing.assert_allclose(om.calc_tfidf(ids, dists, r.feat_cluster_idfs.numpy(), False, 10.0), g["tfidf_hard"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(om.calc_tfidf(ids, dists, r.feat_cluster_idfs.numpy(), True, 10.0), g["tfidf_soft"], rtol=0, atol=1e-6)
    repre = {"vertices": r.vertices.numpy(), "feat_vectors": r.feat_vectors.numpy(), "feat_to_template_ids": r.feat_to_template_ids.numpy(),
             "feat_cluster_centroids": r.feat_cluster_centroids.numpy(), "feat_cluster_idfs": r.feat_cluster_idfs.numpy(),
             "template_descs": r.template_descs.numpy(),
             "template_desc_opts": {"tfidf_knn_metric": "cosine", "tfidf_knn_k": 3, "tfidf_soft_assign": False, "tfidf_soft_sigma_squared": 10.0}}
    out = om.establish_correspondences(pts.numpy(), feats.numpy(), repre, 5, 300, "torch")
    assert [o["template_id"] for o in out] == g["template_ids"].tolist() == g["tm_ids"].tolist()
    np.testing.assert_allclose([o["template_score"] for o in out], g["template_scores"], rtol=0, atol=2e-6)
    for i, o in enumerate(out):
        assert np.array_equal(o["coord_2d_ids"], g[f"coord_2d_ids_{i}"]) and np.array_equal(o["nn_vertex_ids"], g[f"nn_vertex_ids_{i}"])
    with pytest.raises(ValueError, match="not supported"):
        om.nearest_words(feats.numpy(), r.feat_cluster_centroids.numpy(), 3, "dot")
