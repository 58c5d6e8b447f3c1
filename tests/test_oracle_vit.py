"""CPU: the ViT oracle vs (a) the reference wrapper over the HF stand-in backbone (fixtures),
(b) point utilities / sampling / PCA fixtures, (c) the composite hot-section fixture."""

import numpy as np
import pytest
import torch

from foundpose_amd import synthetic
from foundpose_amd.vit_config import ARCHS, parse_extractor_name
from oracle import match as om
from oracle import vit as ov
from tests.helpers import NOREG_CASES, TINY, TINY0, checksum, load_golden, noreg_case


def test_extractor_tiny_matches_reference_wrapper():
    g = load_golden("extractor_tiny")
    sd = synthetic.make_vit_state_dict(TINY, seed=int(g["weights_seed"]))
    imgs = synthetic.make_crops(2, 56, seed=int(g["image_seed"]))
    assert np.isclose(checksum(imgs, sd["blocks.1.attn.qkv.weight"], sd["pos_embed"]), g["input_checksum"], atol=1e-6)
    for layer, norm in ((1, 1), (2, 1), (0, 0)):
        o = ov.extractor_forward(sd, TINY, imgs, layer, bool(norm))
        np.testing.assert_allclose(o["feature_maps"].numpy(), g[f"fmap_l{layer}_n{norm}"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(o["cls_tokens"].numpy(), g[f"cls_l{layer}_n{norm}"], rtol=0, atol=2e-5)


def test_extractor_vits14reg_518_matches_reference_wrapper():
    g = load_golden("extractor_vits14reg_518")
    spec = parse_extractor_name("dinov2_version=vits14-reg_stride=14_facet=token_layer=9_logbin=0_norm=1")
    assert (spec.version, spec.layer, spec.apply_norm) == ("vits14-reg", 9, True)
    sd = synthetic.make_vit_state_dict(spec.arch, seed=int(g["weights_seed"]))
    imgs = synthetic.make_crops(1, 518, seed=int(g["image_seed"]))
    assert np.isclose(checksum(imgs, sd["blocks.9.attn.qkv.weight"]), g["input_checksum"], atol=1e-6)
    o = ov.extractor_forward(sd, spec.arch, imgs, spec.layer, True)
    fm = o["feature_maps"].numpy()
    assert fm.shape == (1, 384, 37, 37)
    np.testing.assert_allclose(fm[:, ::8, ::3, ::3], g["fmap_sub"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(o["cls_tokens"].numpy(), g["cls"], rtol=0, atol=5e-5)


def test_extractor_vits14reg_420_matches_reference_wrapper():
    """The shipped LM-O geometry (configs/infer/lmo.json:6-12): 420 x 420 crops, pos-embed interpolated 37 x 37 -> 30 x 30 with the
    `-reg` hub flags (size mode, bicubic, antialias).  Pins the oracle's interpolation to the reference wrapper's fixture."""
    g = load_golden("extractor_vits14reg_420")
    spec = parse_extractor_name("dinov2_version=vits14-reg_stride=14_facet=token_layer=9_logbin=0_norm=1")
    sd = synthetic.make_vit_state_dict(spec.arch, seed=int(g["weights_seed"]))
    imgs = synthetic.make_crops(1, 420, seed=int(g["image_seed"]))
    assert np.isclose(checksum(imgs, sd["blocks.9.attn.qkv.weight"], sd["pos_embed"]), g["input_checksum"], atol=1e-6)
    o = ov.extractor_forward(sd, spec.arch, imgs, spec.layer, True)
    fm = o["feature_maps"].numpy()
    assert fm.shape == (1, 384, 30, 30)
    np.testing.assert_allclose(fm[:, ::4, ::2, ::2], g["fmap_sub"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(o["cls_tokens"].numpy(), g["cls"], rtol=0, atol=5e-5)
    assert abs(float(fm.mean()) - float(g["fmap_mean"])) < 1e-5


def test_extractor_stride7_matches_reference_pieces():
    """stride != patch size: the oracle's strided position encoding equals the reference's `_fix_pos_enc` function output
    (square, non-square and the identity case), and the whole forward equals the reference wrapper run with that function
    (tests/golden/extractor_tiny_stride7.npz; why pieces: oracle/make_golden.py::gen_extractor_stride)."""
    g = load_golden("extractor_tiny_stride7")
    sd = synthetic.make_vit_state_dict(TINY, seed=int(g["weights_seed"]))
    for (H, W) in ((56, 56), (70, 56), (28, 28)):
        np.testing.assert_allclose(ov.interpolate_pos_embed_strided(sd["pos_embed"], 14, 7, H, W).numpy(), g[f"pos_{H}x{W}"], rtol=0, atol=1e-6)
    imgs = synthetic.make_crops(2, 56, seed=int(g["image_seed"]))
    for layer, norm in ((2, 1), (0, 0)):
        o = ov.extractor_forward(sd, TINY, imgs, layer, bool(norm), stride=7)
        assert o["feature_maps"].shape == (2, TINY.dim, 7, 7)
        np.testing.assert_allclose(o["feature_maps"].numpy(), g[f"fmap_l{layer}_n{norm}"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(o["cls_tokens"].numpy(), g[f"cls_l{layer}_n{norm}"], rtol=0, atol=2e-5)


def test_extractor_tiny_noreg_matches_reference_wrapper():
    """0 register tokens + the scale-factor / offset-0.1 / no-antialias pos-embed interpolation of the non-register hub entries:
    the oracle vs the reference wrapper over the non-register stand-in (oracle/make_golden.py::gen_extractor_noreg)."""
    g = load_golden("extractor_tiny_noreg")
    sd = synthetic.make_vit_state_dict(TINY0, seed=int(g["weights_seed"]))
    assert "register_tokens" not in sd
    for (H, W) in ((56, 56), (84, 84), (70, 42)):
        np.testing.assert_allclose(ov.interpolate_pos_embed(sd["pos_embed"], TINY0, H // 14, W // 14).numpy(), g[f"pos_{H}x{W}"], rtol=0, atol=1e-6)
        imgs = torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(H * 1000 + W))
        for layer, norm in ((2, 1), (0, 0)):
            o = ov.extractor_forward(sd, TINY0, imgs, layer, bool(norm))
            np.testing.assert_allclose(o["feature_maps"].numpy(), g[f"fmap_{H}x{W}_l{layer}_n{norm}"], rtol=0, atol=2e-5)
            np.testing.assert_allclose(o["cls_tokens"].numpy(), g[f"cls_{H}x{W}_l{layer}_n{norm}"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("version,S", NOREG_CASES)
def test_extractor_noreg_hub_archs_match_reference_wrapper(version, S):
    """`dinov2_vitl14` (InferOpts' default, scripts/infer.py:75; short form -> layer 9), `dinov2_vits14`, `dinov2_vitb14` (short and long
    form) at 518 and 420: the oracle vs the reference wrapper's fixture."""
    g, name, spec, sd, imgs = noreg_case(version, S)
    o = ov.extractor_forward(sd, spec.arch, imgs, spec.layer, spec.apply_norm)
    fm = o["feature_maps"].numpy()
    cs, ss = (int(v) for v in g["sub"])
    assert fm.shape == (1, spec.arch.dim, S // 14, S // 14)
    np.testing.assert_allclose(fm[:, ::cs, ::ss, ::ss], g["fmap_sub"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(o["cls_tokens"].numpy(), g["cls"], rtol=0, atol=5e-5)
    assert abs(float(fm.mean()) - float(g["fmap_mean"])) < 1e-5


def test_name_grammar_defaults():
    s = parse_extractor_name("dinov2_vitl14")
    assert (s.version, s.layer, s.stride, s.facet, s.apply_norm) == ("vitl14", 9, 14, "token", True)
    s = parse_extractor_name("dinov2_version=vitl14_stride=14_facet=key_layer=18_norm=0")
    assert (s.version, s.layer, s.facet, s.apply_norm) == ("vitl14", 18, "key", False)
    assert ARCHS["vitg14-reg"].hidden == 4096 and ARCHS["vitl14"].heads == 16


def test_points_sampling_pca():
    g = load_golden("points_sample_pca")
    for s in (518, 420):
        np.testing.assert_array_equal(ov.generate_grid_points((s, s), 14.0).numpy(), g[f"grid_{s}"])
    gen = torch.Generator().manual_seed(int(g["mask_seed"]))
    mask = (torch.rand(518, 518, generator=gen) > 0.6).to(torch.uint8)
    pts = ov.generate_grid_points((518, 518), 14.0)
    np.testing.assert_array_equal(ov.filter_points_by_mask(pts, mask).numpy(), g["filtered_random"])
    qp = ov.filter_points_by_mask(pts, synthetic.make_disc_mask(518))
    np.testing.assert_array_equal(qp.numpy(), g["filtered_disc"])
    fmap = torch.from_numpy(g["fmap"])
    np.testing.assert_allclose(ov.sample_feature_map_at_points(fmap, qp, (518, 518)).numpy(), g["sampled_grid"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(
        ov.sample_feature_map_at_points(fmap, torch.from_numpy(g["offgrid_points"]), (518, 518)).numpy(),
        g["sampled_offgrid"], rtol=0, atol=1e-6)
    y = ov.pca_transform(torch.from_numpy(g["pca_x"]), torch.from_numpy(g["pca_components"]), torch.from_numpy(g["pca_mean"]))
    np.testing.assert_allclose(y.numpy(), g["pca_y"], rtol=0, atol=2e-5)


def test_hot_section_composite():
    """infer.py:468-542 end to end on the oracle vs the reference-run fixture."""
    g = load_golden("hot_section_tiny")
    S = int(g["image_size"])
    sd = synthetic.make_vit_state_dict(TINY, seed=int(g["weights_seed"]))
    q_img = torch.from_numpy(g["q_img"]).unsqueeze(0)
    fmap = ov.extractor_forward(sd, TINY, q_img, 2, True)["feature_maps"][0]
    np.testing.assert_allclose(fmap.numpy(), g["fmap"], rtol=0, atol=5e-5)
    grid = ov.generate_grid_points((S, S), 14.0)
    qp = ov.filter_points_by_mask(grid, torch.from_numpy(g["tpl_masks"][4]))
    np.testing.assert_array_equal(qp.numpy(), g["query_points"])
    qf = ov.sample_feature_map_at_points(fmap, qp, (S, S))
    np.testing.assert_allclose(qf.numpy(), g["query_features"], rtol=0, atol=5e-5)
    qfp = ov.pca_transform(qf, torch.from_numpy(g["pca_components"]), torch.from_numpy(g["pca_mean"]))
    np.testing.assert_allclose(qfp.numpy(), g["query_features_proj"], rtol=0, atol=1e-4)
    repre = {
        "vertices": g["vertices"], "feat_vectors": g["bank_feats"], "feat_to_template_ids": g["f2t"],
        "feat_cluster_centroids": g["centroids"], "feat_cluster_idfs": g["idfs"], "template_descs": g["template_descs"],
        "template_desc_opts": {"tfidf_knn_k": 3, "tfidf_soft_assign": False, "tfidf_soft_sigma_squared": 10.0},
    }
    # matching on the fixture's own projected features: bit-exact indices
    out = om.establish_correspondences(g["query_points"], g["query_features_proj"], repre, 5, 300, "torch")
    assert [o["template_id"] for o in out] == list(g["template_ids"])
    for i, o in enumerate(out):
        assert np.array_equal(o["coord_2d_ids"], g[f"coord_2d_ids_{i}"])
        assert np.array_equal(o["nn_vertex_ids"], g[f"nn_vertex_ids_{i}"])
    # and through the oracle's own features (fp32 noise 1e-5): same retrieved templates
    out2 = om.establish_correspondences(qp.numpy(), qfp.numpy(), repre, 5, 300, "torch")
    assert [o["template_id"] for o in out2] == list(g["template_ids"])


def test_oracle_facets_match_reference_wrapper_fixture():
    """key / query / value facets: oracle vs the reference wrapper's attention hooks (tests/golden/extractor_tiny_facets.npz)."""
    import numpy as np
    from foundpose_amd import synthetic
    from oracle import vit as ov
    from tests.helpers import TINY, load_golden
    g = load_golden("extractor_tiny_facets")
    sd = synthetic.make_vit_state_dict(TINY, int(g["weights_seed"]))
    imgs = synthetic.make_crops(2, 56, seed=int(g["image_seed"]))
    for facet in ("key", "query", "value"):
        for layer, norm in ((1, 1), (2, 0)):
            o = ov.extractor_forward(sd, TINY, imgs, layer, bool(norm), facet=facet)
            np.testing.assert_allclose(o["feature_maps"].numpy(), g[f"fmap_{facet}_l{layer}_n{norm}"], rtol=0, atol=5e-6)
            np.testing.assert_allclose(o["cls_tokens"].numpy(), g[f"cls_{facet}_l{layer}_n{norm}"], rtol=0, atol=5e-6)
