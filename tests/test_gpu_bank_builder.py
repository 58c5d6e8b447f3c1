"""Bank-builder tier on the MI355X: PCA fit and k-means, the two offline steps in front of the tf-idf
descriptors (scripts/gen_repre.py:271-306 in the reference).  sklearn is the checker for the PCA; faiss (the
reference's k-means) is absent from the image, so k-means is checked through its defining properties."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _decaying_data(n, d, seed):
    g = torch.Generator().manual_seed(seed)
    basis = torch.linalg.qr(torch.randn(d, d, generator=g))[0]
    scale = torch.arange(1, d + 1, dtype=torch.float32) ** -0.7
    return (torch.randn(n, d, generator=g) * scale) @ basis.T + torch.randn(d, generator=g) * 0.3


@pytest.mark.parametrize("n,d,k", [(5000, 96, 16), (20000, 384, 64)])
def test_pca_fit_matches_sklearn_full_solver(n, d, k):
    from sklearn.decomposition import PCA
    from foundpose_amd import projector_util
    x = _decaying_data(n, d, seed=n)
    ref = PCA(n_components=k, svd_solver="full").fit(x.numpy())
    p = projector_util.PCAProjector(n_components=k)
    p.fit(x.cuda())
    comps = p.components.numpy()
    # same sign convention, same directions (well-separated spectrum): |cos| ~ 1 and positive
    cos = np.sum(comps * ref.components_, axis=1)
    assert cos.min() > 0.9995, cos.min()
    np.testing.assert_allclose(p.extra["explained_variance"].numpy(), ref.explained_variance_, rtol=2e-4)
    np.testing.assert_allclose(p.extra["explained_variance_ratio"].numpy(), ref.explained_variance_ratio_, rtol=2e-4)
    np.testing.assert_allclose(p.extra["singular_values"].numpy(), ref.singular_values_, rtol=2e-4)
    np.testing.assert_allclose(float(p.extra["noise_variance"]), ref.noise_variance_, rtol=2e-3)
    np.testing.assert_allclose(p.mean.numpy(), ref.mean_, atol=2e-5)  # fp32 column means, different summation trees
    got = p.transform(x[:512].cuda()).cpu().numpy()
    want = ref.transform(x[:512].numpy())
    assert np.abs(got - want).max() < 2e-3 * np.abs(want).max()
    # round trip through the repre.pth tensordict (projector_util.py:91-145 in the reference)
    q = projector_util.projector_from_tensordict(projector_util.projector_to_tensordict(p))
    assert torch.equal(q.transform(x[:64].cuda()), p.transform(x[:64].cuda()))


def test_pca_fit_max_samples_and_errors():
    from foundpose_amd import projector_util
    x = _decaying_data(3000, 64, seed=1).cuda()
    p = projector_util.PCAProjector(n_components=8)
    torch.manual_seed(0)
    p.fit(x, max_samples=1000)
    assert p.components.shape == (8, 64)
    with pytest.raises(ValueError):
        projector_util.PCAProjector(n_components=80).fit(x)


def test_kmeans_properties():
    from foundpose_amd import cluster_util, ops
    g = torch.Generator().manual_seed(3)
    centers = torch.randn(32, 48, generator=g) * 6
    lab = torch.randint(0, 32, (20000,), generator=g)
    x = (centers[lab] + torch.randn(20000, 48, generator=g) * 0.5).cuda()
    c1, ids1, d1 = cluster_util.kmeans(x, 32, num_iter=25, verbose=False)
    c2, ids2, d2 = cluster_util.kmeans(x, 32, num_iter=25, verbose=False)
    assert torch.equal(c1, c2) and torch.equal(ids1, ids2) and torch.equal(d1, d2)  # deterministic for a seed
    assert ids1.dtype == torch.int32 and c1.shape == (32, 48)
    # outputs are consistent: ids / distances are the exact 1-NN of the returned centroids (cluster_util.py:59)
    dd, ii = ops.knn_l2(x, c1, 1)
    assert torch.equal(ii[:, 0], ids1.to(ii.dtype)) and torch.equal(dd[:, 0], d1)
    # Lloyd's objective does not increase
    objs = []
    for it in (1, 2, 4, 8, 25):
        _, _, d = cluster_util.kmeans(x, 32, num_iter=it, verbose=False)
        objs.append(float(d.double().sum()))
    assert all(b <= a * (1 + 1e-6) for a, b in zip(objs, objs[1:])), objs
    # well separated blobs: the clustering is pure for almost every cluster (a random init may merge / split a few)
    purity = 0
    for c in range(32):
        m = lab[(ids1 == c).cpu()]
        if len(m):
            purity += int(torch.bincount(m, minlength=32).max())
    assert purity > 0.7 * 20000


def test_kmeans_refills_empty_clusters():
    from foundpose_amd import cluster_util
    g = torch.Generator().manual_seed(4)
    x = torch.cat([torch.randn(4000, 8, generator=g) * 0.01, torch.randn(4, 8, generator=g) * 0.01 + 50.0]).cuda()
    c, ids, _ = cluster_util.kmeans(x, 16, num_iter=10, verbose=False)
    assert len(torch.unique(ids)) == 16  # no cluster stays empty
    with pytest.raises(ValueError):
        cluster_util.kmeans(x[:8], 16, verbose=False)


def test_build_object_repre_end_to_end_retrieval():
    """Templates -> ViT features -> PCA -> k-means words -> tf-idf descriptors -> bank; a query crop that IS template 7
    then retrieves template 7 first and its correspondences point back at template 7's own patches (distance 0)."""
    from foundpose_amd import bank_builder, corresp_util, feature_util, synthetic
    ex = feature_util.make_feature_extractor("dinov2_version=vits14-reg_stride=14_facet=token_layer=9_norm=1", random_init_seed=3).to("cuda")
    T, S = 24, 224
    templates = synthetic.make_crops(T, S, seed=11)
    masks = synthetic.make_disc_mask(S).unsqueeze(0).repeat(T, 1, 1)
    feats, f2t, pts = bank_builder.extract_template_features(ex, templates, masks, batch_size=8)
    assert feats.shape[1] == 384 and int(f2t.max()) == T - 1 and bool((f2t[1:] >= f2t[:-1]).all())
    verts = torch.randn(feats.shape[0], 3, generator=torch.Generator().manual_seed(0))
    repre = bank_builder.build_object_repre(feats, f2t, verts, T, pca_components=64, cluster_num=128, cluster_iters=10)
    assert repre.template_descs.shape == (T, 128) and repre.feat_vectors.shape[1] == 64
    sel = (f2t == 7)
    q_raw, q_pts = feats[sel], pts[sel]
    q = repre.feat_raw_projectors[0].transform(q_raw)
    assert torch.equal(q, repre.feat_vectors[sel])  # same projection, same bits
    corresp = corresp_util.establish_correspondences(q_pts, q, repre, "tfidf", "cyclic_buddies", 5, 300)
    assert int(corresp[0]["template_id"]) == 7 and float(corresp[0]["template_score"]) > 0.999
    first = int(torch.nonzero(sel)[0])
    assert torch.equal(corresp[0]["nn_vertex_ids"].cpu() - first, corresp[0]["coord_2d_ids"].cpu())  # patch i <-> its own bank row


def test_register_templates_in_3d_batched_equals_per_template():
    """bank_builder.register_templates_in_3d (batched) == feature_util.get_visual_features_registered_in_3d per template
    (the reference's per-template routine, feature_util.py:162-237), and the lifted vertices re-project onto their pixels."""
    from foundpose_amd import bank_builder, crop_util, feature_util, synthetic
    ex = feature_util.make_feature_extractor("dinov2_version=vits14-reg_stride=14_facet=token_layer=9_norm=1", random_init_seed=3).to("cuda")
    T, S = 5, 224
    templates = synthetic.make_crops(T, S, seed=4).cuda()
    masks = synthetic.make_disc_mask(S).unsqueeze(0).repeat(T, 1, 1).cuda()
    masks[1, :, :60] = 0
    g = torch.Generator().manual_seed(8)
    depths = (500.0 + 80.0 * torch.rand(T, S, S, generator=g)).cuda()
    cams = [crop_util.PinholePlaneCameraModel(S, S, (300.0 + 5 * t, 310.0), (S / 2 - 0.5, S / 2 + 1.5), np.eye(4)) for t in range(T)]
    Ts = torch.eye(4).repeat(T, 1, 1)
    for t in range(T):
        Ts[t, :3, :3] = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
        Ts[t, :3, 3] = torch.randn(3, generator=g) * 50
    feats, f2t, verts, f2v = bank_builder.register_templates_in_3d(ex, templates, depths, masks, cams, Ts, batch_size=2)
    assert torch.equal(f2v.cpu(), torch.arange(feats.shape[0], dtype=torch.int32))
    r0 = 0
    for t in range(T):
        fv, vid, vm = feature_util.get_visual_features_registered_in_3d(templates[t], depths[t], masks[t], cams[t], Ts[t], ex, 14.0)
        n = fv.shape[0]
        assert n > 20 and bool((f2t[r0:r0 + n] == t).all())
        assert torch.equal(vm, verts[r0:r0 + n])
        assert float((fv - feats[r0:r0 + n]).abs().max()) <= 2e-2 * float(fv.abs().max())  # bf16 ViT, batch of 1 vs batch of 2
        # back to the camera and onto the image plane: the pixel the vertex was lifted from
        v_cam = (vm - Ts[t, :3, 3].cuda()) @ Ts[t, :3, :3].cuda()
        f = 0.5 * (cams[t].f[0] + cams[t].f[1])
        uv = v_cam[:, :2] / v_cam[:, 2:3] * f + torch.tensor(cams[t].c, dtype=torch.float32).cuda()
        grid = feature_util.generate_grid_points((S, S), 14.0).cuda()
        q = feature_util.filter_points_by_mask(grid, feature_util.erode_mask(masks[t]))
        assert float((uv - q).abs().max()) < 1e-2
        r0 += n
    assert r0 == feats.shape[0]
