"""GPU parity: descriptor-matching kernels (through the C ABI) vs the CPU oracle and the reference fixtures.

Bar: bit-exact indices and distances (integer/index work and exact-fp32 fma chains), tolerances stated
inline only where the reference itself is order-dependent (torch sums).
"""

import numpy as np
import pytest
import torch

from oracle import clib, match as om
from tests.helpers import MATCH_CASES, load_golden, match_case_inputs, experiments_build

pytestmark = pytest.mark.gpu


def cu(x, dtype=None):
    t = torch.as_tensor(x)
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def test_sqnorm_bitexact():
    from foundpose_amd import ops
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1000, 256)).astype(np.float32)
    got = ops.sqnorm_rows(cu(x)).cpu().numpy()
    assert np.array_equal(got, clib.sqnorm(x))


@pytest.mark.parametrize("m,n,d,k", [(300, 500, 256, 1), (300, 500, 256, 3), (77, 2048, 256, 3), (1, 5, 32, 1),
                                     (513, 129, 64, 5), (40, 40, 256, 40), (200, 300, 36, 2),
                                     # ragged edge tiles of the 128 x 128 engine (1x4 / 4x1 wave cuts, dead 32-blocks): every epilogue
                                     (133, 133, 64, 1), (133, 133, 64, 3), (37, 300, 32, 3), (300, 37, 32, 3), (300, 37, 32, 1),
                                     (64, 64, 32, 5), (65, 200, 32, 1), (197, 70, 32, 3), (160, 193, 32, 12), (33, 161, 32, 8),
                                     (5, 517, 32, 1), (517, 5, 32, 1), (517, 5, 32, 4), (129, 97, 32, 1), (97, 129, 32, 2)])
def test_knn_l2_bitexact(m, n, d, k):
    from foundpose_amd import ops
    rng = np.random.default_rng(m * 7 + n)
    q = rng.standard_normal((m, d)).astype(np.float32)
    db = rng.standard_normal((n, d)).astype(np.float32)
    db[n // 2] = db[0]          # duplicated rows -> exact distance ties, must resolve to the lowest index
    if n > 10:
        db[n - 1] = db[3]
    q[0] = db[0]                # zero distance
    d2, idx = ops.knn_l2(cu(q), cu(db), k)
    o_d2, o_idx = clib.l2_knn(q, db, k)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), o_idx)
    assert np.array_equal(d2.cpu().numpy(), o_d2)  # bitwise: MFMA fp32 == k-ordered fmaf chain


@pytest.mark.parametrize("Q,P", [(5, 5), (5, 140), (140, 5), (33, 70), (70, 33), (64, 129), (129, 64), (133, 133), (517, 389), (389, 517),
                                 (97, 200), (200, 97), (1, 300), (300, 1)])
def test_cyclic_buddies_pair_edge_tiles(Q, P):
    """One (query crop, template) pair at sizes that put every wave cut of the distance tile (2x2, 1x4, 4x1, dead blocks) on
    both the row side (query -> nearest template patch) and the column side (template patch -> nearest query patch):
    ids, cycle distances and scores equal the oracle's, in the reference's torch.topk order."""
    from foundpose_amd import corresp_util
    rng = np.random.default_rng(Q * 1000 + P)
    obj = rng.standard_normal((P, 64)).astype(np.float32)
    qf = rng.standard_normal((Q, 64)).astype(np.float32)
    n_copy = min(Q, P) // 2
    qf[:n_copy] = obj[rng.permutation(P)[:n_copy]]            # exact buddies (zero feature distance)
    if Q > 3:
        qf[Q - 1] = qf[0]                                      # duplicated query patch: exact ties on the column side
    if P > 3:
        obj[P - 1] = obj[1]                                    # duplicated template patch: exact ties on the row side
    pts = (rng.integers(0, 37, (Q, 2)) * 14 + 7).astype(np.float32)
    for top_k in (300, 7):
        q_ids, o_ids, dists, scores = corresp_util.cyclic_buddies_matching(cu(pts), cu(qf), None, cu(obj), None, top_k)
        o_q, o_o, o_d, o_s, _ = om.cyclic_buddies(pts, qf, obj, top_k, topk_mode="torch")
        assert np.array_equal(q_ids.cpu().numpy(), o_q)
        assert np.array_equal(o_ids.cpu().numpy(), o_o)
        assert np.array_equal(dists.cpu().numpy(), o_d)
        assert np.array_equal(scores.cpu().numpy(), o_s, equal_nan=True)


def test_gemm_f32_edge_tiles_exact_chain():
    """Plain-store epilogue on ragged tiles: every output equals the k-ascending fmaf chain of the oracle, bit for bit."""
    from foundpose_amd import ops
    rng = np.random.default_rng(11)
    for m, n, K in [(5, 133, 36), (133, 5, 36), (70, 70, 64), (33, 200, 32), (200, 33, 32), (129, 129, 40), (64, 65, 32), (260, 256, 128)]:
        a = rng.standard_normal((m, K)).astype(np.float32)
        w = rng.standard_normal((n, K)).astype(np.float32)
        got = ops.gemm_f32(cu(a), cu(w)).cpu().numpy()
        ref = np.stack([clib.dot_rows(w, a[i]) for i in range(m)])
        assert np.array_equal(got, ref), (m, n, K)


def test_knn_interface_matches_reference_conventions():
    from foundpose_amd.knn_util import KNN
    rng = np.random.default_rng(3)
    db = torch.from_numpy(rng.standard_normal((200, 256)).astype(np.float32))
    q = torch.from_numpy(rng.standard_normal((50, 256)).astype(np.float32))
    index = KNN(k=3, metric="l2")
    index.fit(db)
    d, i = index.search(q.cuda())
    assert d.is_cuda and i.dtype == torch.int64 and d.shape == (50, 3)
    d_cpu, i_cpu = index.search(q)  # CPU in -> CPU out, like the reference moves results back
    assert not d_cpu.is_cuda
    o_d2, o_idx = clib.l2_knn(q.numpy(), db.numpy(), 3)
    assert np.array_equal(i_cpu.numpy(), o_idx) and np.array_equal(d_cpu.numpy(), o_d2)
    with pytest.raises(ValueError):
        KNN(metric="manhattan").fit(db)
    cos = KNN(k=2, metric="cosine")
    cos.fit(db)
    dc, ic = cos.search(q)
    qn = q / q.norm(dim=1, keepdim=True)
    dn = db / db.norm(dim=1, keepdim=True)
    ref = 1.0 - qn @ dn.T
    rv, ri = torch.topk(ref, 2, largest=False)
    assert torch.equal(ic, ri)
    torch.testing.assert_close(dc, rv, atol=2e-6, rtol=0)


@pytest.mark.parametrize("soft", [False, True])
def test_tfidf_build(soft):
    from foundpose_amd import ops
    rng = np.random.default_rng(5)
    W, k = 256, 3
    counts = [37, 1, 512, 90]
    Q = sum(counts)
    ids = rng.integers(0, W, (Q, k)).astype(np.int32)
    d2 = (rng.random((Q, k)) * 30).astype(np.float32)
    idf = (rng.random(W) * 3).astype(np.float32)
    seg = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    desc, desc_n = ops.tfidf_build(cu(ids), cu(d2), cu(seg), cu(idf), soft, 10.0, sqrt_dists=True)
    for s in range(len(counts)):
        sl = slice(seg[s], seg[s + 1])
        ref = om.calc_tfidf(ids[sl].astype(np.int64), np.sqrt(d2[sl]), idf, soft, 10.0)
        if soft:  # expf vs numpy exp: 1-2 ulp on the weights
            np.testing.assert_allclose(desc[s].cpu().numpy(), ref, rtol=2e-6, atol=1e-9)
        else:
            assert np.array_equal(desc[s].cpu().numpy(), ref)
            assert np.array_equal(desc_n[s].cpu().numpy(), om.l2_normalize_rows(ref[None])[0])


def _gpu_corresp(repre_np, pts, feats, top_n, top_k, tie_order="canonical"):
    from foundpose_amd import corresp_util, repre_util
    o = repre_np["template_desc_opts"]
    repre = repre_util.FeatureBasedObjectRepre(
        vertices=torch.from_numpy(repre_np["vertices"]), feat_vectors=torch.from_numpy(repre_np["feat_vectors"]),
        feat_to_template_ids=torch.from_numpy(repre_np["feat_to_template_ids"]),
        feat_cluster_centroids=torch.from_numpy(repre_np["feat_cluster_centroids"]),
        feat_cluster_idfs=torch.from_numpy(repre_np["feat_cluster_idfs"]),
        template_descs=torch.from_numpy(repre_np["template_descs"]),
        template_desc_opts=repre_util.TemplateDescOpts(tfidf_knn_k=o["tfidf_knn_k"], tfidf_soft_assign=o["tfidf_soft_assign"],
                                                       tfidf_soft_sigma_squared=o["tfidf_soft_sigma_squared"]))
    return corresp_util.establish_correspondences(
        query_points=torch.from_numpy(pts).cuda(), query_features=torch.from_numpy(feats).cuda(), object_repre=repre,
        template_matching_type="tfidf", feat_matching_type="cyclic_buddies", top_n_templates=top_n,
        top_k_buddies=top_k, debug=True, tie_order=tie_order)


@pytest.mark.parametrize("name", sorted(MATCH_CASES))
def test_establish_correspondences_strict_order_equals_reference(name):
    """Default drop-in behaviour (tie_order="torch"): identical to the reference's own run, index for index,
    INCLUDING the order among tied cycle distances and the choice at the top-k boundary."""
    c, g, repre, pts, feats = match_case_inputs(name)
    got = _gpu_corresp(repre, pts, feats, c["top_n"], c["top_k"], tie_order="torch")
    assert [int(a["template_id"]) for a in got] == list(g["template_ids"])
    np.testing.assert_allclose([float(a["template_score"]) for a in got], g["template_scores"], rtol=0, atol=2e-6)
    for i, a in enumerate(got):
        assert np.array_equal(a["coord_2d_ids"].cpu().numpy(), g[f"coord_2d_ids_{i}"]), f"template slot {i}"
        assert np.array_equal(a["nn_vertex_ids"].cpu().numpy(), g[f"nn_vertex_ids_{i}"])
        assert np.array_equal(a["coord_2d"].cpu().numpy(), g[f"coord_2d_{i}"])
        assert np.array_equal(a["coord_3d"].cpu().numpy(), g[f"coord_3d_{i}"])
        assert np.array_equal(a["nn_dists"].cpu().numpy(), g[f"nn_dists_{i}"])
        np.testing.assert_allclose(a["coord_conf"].cpu().numpy(), g[f"coord_conf_{i}"], rtol=0, atol=1e-7, equal_nan=True)


def test_strict_template_order_with_duplicate_templates():
    """Exact score ties (duplicated templates): tie_order="torch" reproduces torch.topk's pick, canonical picks the lowest id."""
    from foundpose_amd import ops
    from foundpose_amd._lib import call, ptr, stream
    rng = np.random.default_rng(4)
    for T in (100, 800):  # nth_element branch (5*64 > 100) and partial_sort branch (5*64 <= 800)
        W, Bq = 64, 3
        bank = rng.random((T, W)).astype(np.float32)
        bank[T // 2:] = bank[: T - T // 2]  # every descriptor appears twice
        q = rng.random((Bq, W)).astype(np.float32)
        bank_n, q_n = ops.normalize_rows(cu(bank)), ops.normalize_rows(cu(q))
        seg, tpl = cu(np.array([0, Bq], np.int32)), cu(np.array([0, T], np.int32))
        nt = cu(np.full(Bq, T, np.int32))
        from foundpose_amd._lib import cosine_scratch_floats
        sims = torch.empty(cosine_scratch_floats(Bq, T), device="cuda")
        for mode in (0, 1):
            sc = torch.empty(Bq, 5, device="cuda")
            ids = torch.empty(Bq, 5, dtype=torch.int32, device="cuda")
            call("fp_cosine_topk", ptr(q_n), ptr(seg), ptr(nt), Bq, Bq, ptr(bank_n), ptr(tpl), 1, T, W, 5, ptr(sims), ptr(sc), ptr(ids), mode, stream())
            s_cpu = sims[:Bq * T].reshape(Bq, T).cpu()  # the finished scores stay at the head of the scratch
            for b in range(Bq):
                if mode == 1:
                    ref = torch.topk(s_cpu[b], 5, sorted=True)[1]
                else:
                    ref = torch.from_numpy(clib.topk_canonical(s_cpu[b].numpy(), 5, True)[1])
                assert ids[b].cpu().tolist() == ref.tolist(), (T, mode, b)


@pytest.mark.parametrize("W", [128, 2048])   # 2048: the fused streaming kernel + tie flags; 128 words: the generic fallback + full replay
@pytest.mark.parametrize("T", [64, 500, 3000])
def test_torch_order_replay_only_where_ties_exist(T, W):
    """tie_order="torch" replays torch.topk only for rows whose best n + 1 scores contain a tie; a row without one takes the
    canonical list.  Rows built to put a tie (a) inside the top 5, (b) across the cut (ranks 5/6), (c) only below rank 6
    (no replay needed), (d) nowhere, (e) everywhere (all templates identical), (f) zero scores (disjoint words): each
    must equal torch.topk on the CPU copy of the device scores, and the canonical mode the lowest-id rule."""
    from foundpose_amd import ops
    from foundpose_amd._lib import call, ptr, stream, cosine_scratch_floats
    rng = np.random.default_rng(T + W)
    half = W // 2
    bank = np.zeros((T, W), np.float32)
    bank[:, :half] = rng.random((T, half)).astype(np.float32)            # templates live on the first half of the words
    q = np.zeros((6, W), np.float32)
    q[:, :half] = rng.random((6, half)).astype(np.float32)
    order = np.argsort(-(bank / np.linalg.norm(bank, axis=1, keepdims=True)) @ (q[0] / np.linalg.norm(q[0])))
    bank_a = bank.copy(); bank_a[order[2]] = bank_a[order[1]]            # (a) ranks 2 and 3 identical
    bank_b = bank.copy(); bank_b[order[5]] = bank_b[order[4]]            # (b) ranks 5 and 6 identical
    bank_c = bank.copy(); bank_c[order[8]] = bank_c[order[7]]            # (c) ranks 8 and 9 identical
    cases = [("a", bank_a, q[0]), ("b", bank_b, q[0]), ("c", bank_c, q[0]), ("d", bank, q[1]),
             ("e", np.repeat(bank[:1], T, 0), q[2])]
    qz = np.zeros(W, np.float32); qz[half:] = rng.random(W - half).astype(np.float32)
    cases.append(("f", bank, qz))                                        # (f) every score exactly +0
    for name, bk, qq in cases:
        bank_n, q_n = ops.normalize_rows(cu(bk)), ops.normalize_rows(cu(qq[None]))
        seg, tpl, nt = cu(np.array([0, 1], np.int32)), cu(np.array([0, T], np.int32)), cu(np.full(1, T, np.int32))
        sims = torch.empty(cosine_scratch_floats(1, T), device="cuda")
        for mode in (1, 0):
            sc = torch.empty(1, 5, device="cuda")
            ids = torch.empty(1, 5, dtype=torch.int32, device="cuda")
            call("fp_cosine_topk", ptr(q_n), ptr(seg), ptr(nt), 1, 1, ptr(bank_n), ptr(tpl), 1, T, W, 5, ptr(sims), ptr(sc), ptr(ids), mode, stream())
            s_cpu = sims[:T].cpu()
            if mode == 1:
                rv, ri = torch.topk(s_cpu, 5, sorted=True)
            else:
                v, i = clib.topk_canonical(s_cpu.numpy(), 5, True)
                rv, ri = torch.from_numpy(v), torch.from_numpy(i)
            assert ids[0].cpu().tolist() == ri.tolist(), (name, T, W, mode)
            assert torch.equal(sc[0].cpu(), rv), (name, T, W, mode)


@pytest.mark.parametrize("n_top", [1, 6, 7, 8, 12])
def test_torch_order_top_n_sizes_vs_torch_topk(n_top):
    """n_top up to 7 takes the candidate path with n_top + 1 keys per workgroup (8 = the candidate list's width), 8 and above
    the full replay from the score matrix: both must equal torch.topk on the CPU copy of the scores, with duplicated
    templates (ties) in the bank."""
    from foundpose_amd import ops
    from foundpose_amd._lib import call, ptr, stream, cosine_scratch_floats
    rng = np.random.default_rng(n_top)
    T, W, Bq = 1500, 2048, 5
    bank = rng.random((T, W)).astype(np.float32)
    bank[700:760] = bank[:60]                       # sixty duplicated templates: exact score ties
    q = rng.random((Bq, W)).astype(np.float32)
    q[1] = bank[10]                                 # its best score is shared by templates 10 and 710
    bank_n, q_n = ops.normalize_rows(cu(bank)), ops.normalize_rows(cu(q))
    seg, tpl, nt = cu(np.array([0, Bq], np.int32)), cu(np.array([0, T], np.int32)), cu(np.full(Bq, T, np.int32))
    sims = torch.empty(cosine_scratch_floats(Bq, T), device="cuda")
    sc = torch.empty(Bq, n_top, device="cuda")
    ids = torch.empty(Bq, n_top, dtype=torch.int32, device="cuda")
    call("fp_cosine_topk", ptr(q_n), ptr(seg), ptr(nt), Bq, Bq, ptr(bank_n), ptr(tpl), 1, T, W, n_top, ptr(sims), ptr(sc), ptr(ids), 1, stream())
    s_cpu = sims[:Bq * T].reshape(Bq, T).cpu()
    for b in range(Bq):
        rv, ri = torch.topk(s_cpu[b], n_top, sorted=True)
        assert ids[b].cpu().tolist() == ri.tolist(), (n_top, b)
        assert torch.equal(sc[b].cpu(), rv), (n_top, b)


@pytest.mark.parametrize("name", sorted(MATCH_CASES))
def test_establish_correspondences_vs_oracle_and_reference(name):
    c, g, repre, pts, feats = match_case_inputs(name)
    got = _gpu_corresp(repre, pts, feats, c["top_n"], c["top_k"])
    ora = om.establish_correspondences(pts, feats, repre, c["top_n"], c["top_k"], topk_mode="canonical")
    assert len(got) == len(ora) == len(g["template_ids"])
    # (1) vs the oracle in canonical tie order: everything bit-exact
    for a, b in zip(got, ora):
        assert int(a["template_id"]) == b["template_id"]
        np.testing.assert_allclose(float(a["template_score"]), b["template_score"], rtol=0, atol=1e-6)
        assert np.array_equal(a["coord_2d_ids"].cpu().numpy(), b["coord_2d_ids"])
        assert np.array_equal(a["nn_vertex_ids"].cpu().numpy(), b["nn_vertex_ids"])
        assert np.array_equal(a["coord_2d"].cpu().numpy(), b["coord_2d"])
        assert np.array_equal(a["coord_3d"].cpu().numpy(), b["coord_3d"])
        assert np.array_equal(a["nn_dists"].cpu().numpy(), b["nn_dists"])
        np.testing.assert_array_equal(a["coord_conf"].cpu().numpy(), b["coord_conf"])
    # (2) vs the reference's own run (fixture): same retrieved templates in the same order, scores within
    # 2e-6 (torch sums in a different order), same multiset of cycle distances per template (the reference's
    # torch.topk orders ties by libstdc++ internals; the oracle's "torch" mode reproduces that on the CPU)
    assert [int(a["template_id"]) for a in got] == list(g["template_ids"])
    np.testing.assert_allclose([float(a["template_score"]) for a in got], g["template_scores"], rtol=0, atol=2e-6)
    for i, a in enumerate(got):
        np.testing.assert_array_equal(np.sort(a["nn_dists"].cpu().numpy()), np.sort(g[f"nn_dists_{i}"]))
        if len(g[f"nn_dists_{i}"]) == len(pts):  # everything selected: identical index sets
            assert set(a["coord_2d_ids"].cpu().numpy().tolist()) == set(g[f"coord_2d_ids_{i}"].tolist())


def test_match_batch_multi_detection_multi_object():
    """B detections over 2 objects in one call == each detection alone (and == the oracle)."""
    from foundpose_amd import repre_util
    from foundpose_amd.bank import DeviceBank
    from foundpose_amd.matching import match_batch
    names = ["match_planted", "match_ties"]
    repres, queries = [], []
    for nm in names:
        c, g, r, pts, feats = match_case_inputs(nm)
        # equalise the number of words across objects (bank requirement): pad words far away, idf 1
        W = 128
        cent = r["feat_cluster_centroids"]
        if cent.shape[0] < W:
            pad = W - cent.shape[0]
            r["feat_cluster_centroids"] = np.concatenate([cent, np.full((pad, 256), 1e3, np.float32)], 0)
            r["feat_cluster_idfs"] = np.concatenate([r["feat_cluster_idfs"], np.ones(pad, np.float32)])
            r["template_descs"] = np.concatenate([r["template_descs"], np.zeros((r["template_descs"].shape[0], pad), np.float32)], 1)
        repres.append(r)
        queries.append((pts, feats, c))
    def to_repre(r):
        return repre_util.FeatureBasedObjectRepre(
            vertices=torch.from_numpy(r["vertices"]), feat_vectors=torch.from_numpy(r["feat_vectors"]),
            feat_to_template_ids=torch.from_numpy(r["feat_to_template_ids"]),
            feat_cluster_centroids=torch.from_numpy(r["feat_cluster_centroids"]),
            feat_cluster_idfs=torch.from_numpy(r["feat_cluster_idfs"]), template_descs=torch.from_numpy(r["template_descs"]),
            template_desc_opts=repre_util.TemplateDescOpts())
    bank = DeviceBank([to_repre(r) for r in repres])
    # detections: obj0, obj0 (a subset of the query), obj1
    p0, f0, c0 = queries[0]
    p1, f1, c1 = queries[1]
    dets = [(0, p0, f0), (0, p0[:20], f0[:20]), (1, p1, f1)]
    qf = torch.from_numpy(np.concatenate([d[2] for d in dets])).cuda()
    qp = torch.from_numpy(np.concatenate([d[1] for d in dets])).cuda()
    res = match_batch(bank, qf, qp, [len(d[1]) for d in dets], [d[0] for d in dets], 5, 300)
    for b, (obj, p, f) in enumerate(dets):
        ora = om.establish_correspondences(p, f, repres[obj], 5, 300, topk_mode="canonical")
        got = res.corresp_list(b, debug=True)
        assert [int(x["template_id"]) for x in got] == [o["template_id"] for o in ora]
        for a, o in zip(got, ora):
            assert np.array_equal(a["coord_2d_ids"].cpu().numpy(), o["coord_2d_ids"])
            assert np.array_equal(a["nn_vertex_ids"].cpu().numpy(), o["nn_vertex_ids"])
            assert np.array_equal(a["coord_3d"].cpu().numpy(), o["coord_3d"])


def test_sampling_and_pca_vs_reference_fixture():
    from foundpose_amd import feature_util, projector_util
    g = load_golden("points_sample_pca")
    fmap = torch.from_numpy(g["fmap"]).cuda()
    qp = torch.from_numpy(g["filtered_disc"]).cuda()
    s = feature_util.sample_feature_map_at_points(fmap, qp, (518, 518))
    assert np.array_equal(s.cpu().numpy(), g["sampled_grid"])  # bit-exact vs torch's CPU grid_sample
    # a permuted (non-contiguous) view must be read in place with identical results
    fmap_view = fmap.permute(1, 2, 0).contiguous().permute(2, 0, 1)
    assert not fmap_view.is_contiguous()
    s2 = feature_util.sample_feature_map_at_points(fmap_view, qp, (518, 518))
    assert torch.equal(s, s2)
    so = feature_util.sample_feature_map_at_points(fmap, torch.from_numpy(g["offgrid_points"]).cuda(), (518, 518))
    assert np.array_equal(so.cpu().numpy(), g["sampled_offgrid"])
    proj = projector_util.projector_from_tensordict({"pca_projector": {
        "components": torch.from_numpy(g["pca_components"]), "mean": torch.from_numpy(g["pca_mean"]), "whiten": torch.tensor(False)}})
    y = projector_util.project_features(torch.from_numpy(g["pca_x"]).cuda(), [proj])
    np.testing.assert_allclose(y.cpu().numpy(), g["pca_y"], rtol=0, atol=2e-5)  # fp32 GEMM order vs MKL


def test_errors_are_loud():
    from foundpose_amd import _lib, corresp_util, ops
    with pytest.raises(_lib.FoundPoseNativeError):
        ops.sqnorm_rows(torch.zeros(4, 8))  # CPU tensor: no fallback
    with pytest.raises(ValueError):
        corresp_util.establish_correspondences(torch.zeros(1, 2), torch.zeros(1, 4), None, "bow", "cyclic_buddies", 5, 300)


@pytest.mark.parametrize("name", ["match_planted", "match_soft", "match_partialsort"])
def test_bank_builder_vs_reference_fixture(name):
    """Device bank builder (calc_tfidf_descriptors) vs the reference's own output stored in the fixture."""
    from foundpose_amd import bank_builder, repre_util
    c, g, repre, pts, feats = match_case_inputs(name)
    opts = repre_util.TemplateDescOpts(tfidf_soft_assign=bool(c["soft"]))
    descs, idfs, f2c = bank_builder.calc_tfidf_descriptors(
        cu(repre["feat_vectors"]), cu(repre["feat_to_template_ids"]), cu(repre["feat_cluster_centroids"]), c["T"], opts)
    assert np.array_equal(f2c.cpu().numpy(), g["feat_to_cluster_ids"])
    np.testing.assert_allclose(idfs.cpu().numpy(), g["word_idfs"], rtol=3e-7, atol=0)
    np.testing.assert_allclose(descs.cpu().numpy(), g["template_descs"], rtol=5e-5, atol=1e-8)
    # the drop-in with the reference's signature (template_util.py:74-83) gives the same tensors
    from foundpose_amd import template_util
    d2, i2 = template_util.calc_tfidf_descriptors(
        cu(repre["feat_vectors"]), cu(g["feat_to_cluster_ids"]), cu(repre["feat_to_template_ids"]), cu(repre["feat_cluster_centroids"]),
        int(c["T"]), 3, bool(c["soft"]), 10.0)
    assert torch.equal(d2, descs) and torch.equal(i2, idfs)


def test_engine_batch_equals_per_detection():
    """FoundPoseEngine on a batch == the drop-in per-detection call sequence of infer.py:468-542."""
    from foundpose_amd import corresp_util, engine, feature_util, projector_util, repre_util, synthetic
    from foundpose_amd.bank import DeviceBank
    from tests.helpers import TINY
    g = load_golden("hot_section_tiny")
    S = int(g["image_size"])
    ex = feature_util.make_feature_extractor("dinov2_version=tiny-reg_stride=14_facet=token_layer=2_logbin=0_norm=1",
                                             random_init_seed=int(g["weights_seed"]), precision="fp32", arch=TINY).to("cuda")
    proj = projector_util.projector_from_tensordict({"pca_projector": {
        "components": torch.from_numpy(g["pca_components"]), "mean": torch.from_numpy(g["pca_mean"]), "whiten": torch.tensor(False)}})
    repre = repre_util.FeatureBasedObjectRepre(
        vertices=torch.from_numpy(g["vertices"]), feat_vectors=torch.from_numpy(g["bank_feats"]),
        feat_to_template_ids=torch.from_numpy(g["f2t"]), feat_cluster_centroids=torch.from_numpy(g["centroids"]),
        feat_cluster_idfs=torch.from_numpy(g["idfs"]), template_descs=torch.from_numpy(g["template_descs"]),
        template_desc_opts=repre_util.TemplateDescOpts(), feat_raw_projectors=[proj])
    eng = engine.FoundPoseEngine(ex, DeviceBank([repre]), 14.0, 5, 300, tie_order="torch")
    imgs = torch.from_numpy(g["tpl_imgs"][[4, 7, 1]].astype(np.float32)).cuda()
    imgs[0] = torch.from_numpy(g["q_img"]).cuda()
    masks = torch.from_numpy(g["tpl_masks"][[4, 7, 1]]).cuda()
    res = eng.infer_batch(imgs, masks)
    assert res.template_ids[0].tolist() == list(g["template_ids"])  # the fixture's query is detection 0
    grid = feature_util.generate_grid_points((S, S), 14.0).cuda()
    for b in range(3):
        fmap = ex(imgs[b:b + 1])["feature_maps"][0]
        qp = feature_util.filter_points_by_mask(grid, masks[b])
        qf = feature_util.sample_feature_map_at_points(fmap, qp, (S, S)).contiguous()
        qfp = projector_util.project_features(qf, repre.feat_raw_projectors).contiguous()
        single = corresp_util.establish_correspondences(qp, qfp, repre, "tfidf", "cyclic_buddies", 5, 300)
        batch = res.corresp_list(b)
        assert len(single) == len(batch)
        for s_, b_ in zip(single, batch):
            assert int(s_["template_id"]) == int(b_["template_id"])
            assert torch.equal(s_["coord_2d_ids"], b_["coord_2d_ids"]) and torch.equal(s_["nn_vertex_ids"], b_["nn_vertex_ids"])
            assert torch.equal(s_["coord_3d"], b_["coord_3d"]) and torch.equal(s_["coord_2d"], b_["coord_2d"])
    rec = engine.pack_result(res)
    assert rec.shape == (3, 5 * (3 + 300 * 9))


# ------------------------------------------------------------------ full-size properties (BASELINE sizes: T = 10 000, W = 2048)
def _full_size_bank(T=10000, W=2048, seed=3):
    g = torch.Generator(device="cuda").manual_seed(seed)
    from foundpose_amd import ops
    bank = torch.rand(T, W, generator=g, device="cuda") * (torch.rand(T, W, generator=g, device="cuda") < 0.02)  # sparse, like tf-idf
    bank[:, 0] += 1e-3  # no all-zero rows
    return ops.normalize_rows(bank)


def _cosine_topk(desc_n, bank_n, n_top, tie_mode=0):
    from foundpose_amd._lib import call, ptr, stream
    B, W = desc_n.shape
    T = bank_n.shape[0]
    seg = torch.tensor([0, B], dtype=torch.int32, device="cuda")
    off = torch.tensor([0, T], dtype=torch.int32, device="cuda")
    nt = torch.full((B,), T, dtype=torch.int32, device="cuda")
    from foundpose_amd._lib import cosine_scratch_floats
    sims = torch.zeros(cosine_scratch_floats(B, T), device="cuda")
    sc, ids = torch.empty(B, n_top, device="cuda"), torch.empty(B, n_top, dtype=torch.int32, device="cuda")
    call("fp_cosine_topk", ptr(desc_n), ptr(seg), ptr(nt), B, B, ptr(bank_n), ptr(off), 1, T, W, n_top, ptr(sims), ptr(sc), ptr(ids),
         tie_mode, stream())
    return sc, ids, sims


def test_full_size_retrieval_planted_and_kernel_agreement():
    """T = 10 000 templates x 2048 words, 32 detections:
    * a query that IS a bank row comes back first with score 1 (size-independent planted property),
    * scores and ids do not depend on the batch: the same 32 queries inside a 64-detection call (two 32-detection
      chunks of the fused kernel) come back bit-identical,
    * the top-5 equals a sort of the oracle's canonical chain scores on a sample of rows."""
    bank_n = _full_size_bank()
    T, W = bank_n.shape
    g = torch.Generator().manual_seed(5)
    planted = torch.randint(0, T, (32,), generator=g)
    desc_n = bank_n[planted.cuda()].clone()
    desc_n[16:] = torch.nn.functional.normalize(desc_n[16:] + 0.02 * torch.rand(16, W, device="cuda"), dim=1)  # half of them perturbed
    from foundpose_amd import ops
    desc_n = ops.normalize_rows(desc_n)
    sc, ids, _ = _cosine_topk(desc_n, bank_n, 5)
    assert torch.equal(ids[:, 0].cpu(), planted.to(torch.int32))
    assert float((sc[:16, 0] - 1).abs().max()) < 1e-6
    # same 32 queries inside a 64-detection call
    sc2, ids2, _ = _cosine_topk(torch.cat([desc_n, desc_n.flip(0)]), bank_n, 5)
    assert torch.equal(ids2[:32], ids) and torch.equal(sc2[:32], sc)
    assert torch.equal(ids2[32:], ids.flip(0)) and torch.equal(sc2[32:], sc.flip(0))
    # oracle chain (8 k-slices x permuted 16-blocks) on 3 detections
    bn = bank_n.cpu().numpy()
    for b in (0, 17, 31):
        ref = clib.dot_rows(bn, desc_n[b].cpu().numpy(), perm16=True)
        vals, idx = clib.topk_canonical(ref, 5, True)
        assert np.array_equal(idx.astype(np.int32), ids[b].cpu().numpy())
        assert np.array_equal(vals, sc[b].cpu().numpy())


@pytest.mark.parametrize("T,B", [(10000, 128), (50000, 128), (50000, 37)])
def test_config5_sized_retrieval_batch_invariant_and_strict(T, B):
    """BASELINE config 5 sizes (50k templates, 128 detections of one object per GPU): every score equals the score the
    same query gets alone (one arithmetic whatever the batch), the canonical top-5 equals the oracle's sort of the
    device's own scores, and the strict order equals torch.topk run on those scores on the CPU -- with duplicated
    templates, so exact score ties exist at every rank."""
    W = 2048
    g = torch.Generator(device="cuda").manual_seed(T + B)
    from foundpose_amd import ops
    bank = torch.rand(T, W, generator=g, device="cuda") * (torch.rand(T, W, generator=g, device="cuda") < 0.02)
    bank[:, 0] += 1e-3
    bank[T // 2:] = bank[: T - T // 2]  # every descriptor appears twice: ties between ids t and t + T/2
    bank_n = ops.normalize_rows(bank)
    del bank
    desc_n = ops.normalize_rows(torch.rand(B, W, generator=g, device="cuda") * (torch.rand(B, W, generator=g, device="cuda") < 0.3))
    sc0, ids0, sims0 = _cosine_topk(desc_n, bank_n, 5, tie_mode=0)
    sc1, ids1, sims1 = _cosine_topk(desc_n, bank_n, 5, tie_mode=1)
    S = sims0[:B * T].reshape(B, T)
    assert torch.equal(S, sims1[:B * T].reshape(B, T))
    for b in (0, 31, 32, B - 1):  # alone == inside the batch
        _, _, s_one = _cosine_topk(desc_n[b:b + 1].contiguous(), bank_n, 5)
        assert torch.equal(s_one[:T], S[b]), b
    ref = clib.dot_rows(bank_n.cpu().numpy(), desc_n[B // 2].cpu().numpy(), perm16=True)  # the oracle's chain
    assert np.array_equal(ref, S[B // 2].cpu().numpy())
    S_cpu = S.cpu()
    for b in range(B):
        vals, idx = clib.topk_canonical(S_cpu[b].numpy(), 5, True)
        assert np.array_equal(idx.astype(np.int32), ids0[b].cpu().numpy()), b
        assert np.array_equal(vals, sc0[b].cpu().numpy()), b
        tv, ti = torch.topk(S_cpu[b], 5, sorted=True)  # the reference's own call (template_util.py:172)
        assert ids1[b].cpu().tolist() == ti.tolist(), b
        assert torch.equal(sc1[b].cpu(), tv), b


def test_retrieval_more_than_8_templates_canonical():
    """n_top > 8 leaves the candidate lists of the fused kernel: same scores, canonical order by selection passes."""
    bank_n = _full_size_bank(T=3000, seed=3)
    from foundpose_amd import ops
    desc_n = ops.normalize_rows(torch.rand(5, bank_n.shape[1], generator=torch.Generator(device="cuda").manual_seed(2), device="cuda"))
    sc, ids, sims = _cosine_topk(desc_n, bank_n, 12)
    S = sims[:5 * 3000].reshape(5, 3000).cpu()
    for b in range(5):
        vals, idx = clib.topk_canonical(S[b].numpy(), 12, True)
        assert np.array_equal(idx.astype(np.int32), ids[b].cpu().numpy())
        assert np.array_equal(vals, sc[b].cpu().numpy())
    sc5, ids5, _ = _cosine_topk(desc_n, bank_n, 5)
    assert torch.equal(ids5, ids[:, :5]) and torch.equal(sc5, sc[:, :5])


def test_full_size_strict_and_canonical_agree_without_ties():
    """On tie-free scores the reference's torch.topk order and the canonical order are the same list."""
    bank_n = _full_size_bank(T=10000, seed=9)
    from foundpose_amd import ops
    desc_n = ops.normalize_rows(torch.rand(8, bank_n.shape[1], generator=torch.Generator(device="cuda").manual_seed(1), device="cuda"))
    sc0, ids0, _ = _cosine_topk(desc_n, bank_n, 5, tie_mode=0)
    sc1, ids1, _ = _cosine_topk(desc_n, bank_n, 5, tie_mode=1)
    assert torch.equal(ids0, ids1) and torch.equal(sc0, sc1)


@pytest.mark.parametrize("T,W,B", [(1, 128, 1), (3, 128, 2), (17, 256, 5), (800, 2048, 33), (129, 192, 7), (16, 2048, 32)])
def test_retrieval_ragged_shapes_vs_oracle(T, W, B):
    """Fewer templates than top-n (ids -1 / score -inf past the end), template counts that are not multiples of the
    16-row block, word counts that take the 1-slice chain, and a detection count that takes the fallback kernel."""
    g = torch.Generator().manual_seed(T * 7 + W + B)
    from foundpose_amd import ops
    bank_n = ops.normalize_rows(torch.rand(T, W, generator=g).cuda())
    desc_n = ops.normalize_rows(torch.rand(B, W, generator=g).cuda())
    sc, ids, _ = _cosine_topk(desc_n, bank_n, 5)
    bn = bank_n.cpu().numpy()
    for b in range(B):
        ref = clib.dot_rows(bn, desc_n[b].cpu().numpy(), perm16=(W % 16 == 0))
        k = min(5, T)
        vals, idx = clib.topk_canonical(ref, k, True)
        assert np.array_equal(idx.astype(np.int32), ids[b, :k].cpu().numpy())
        assert np.array_equal(vals, sc[b, :k].cpu().numpy())
        assert bool((ids[b, k:] == -1).all()) and bool(torch.isinf(sc[b, k:]).all())


def test_match_batch_degenerate_detections():
    """A detection with a single query patch and one with fewer patches than top-k buddies, next to a normal one:
    each equals the oracle run alone (the reference would process them one by one, corresp_util.py:73-169)."""
    from foundpose_amd import repre_util
    from foundpose_amd.bank import DeviceBank
    from foundpose_amd.matching import match_batch
    c, g, r, pts, feats = match_case_inputs("match_planted")
    repre = repre_util.FeatureBasedObjectRepre(
        vertices=torch.from_numpy(r["vertices"]), feat_vectors=torch.from_numpy(r["feat_vectors"]),
        feat_to_template_ids=torch.from_numpy(r["feat_to_template_ids"]),
        feat_cluster_centroids=torch.from_numpy(r["feat_cluster_centroids"]),
        feat_cluster_idfs=torch.from_numpy(r["feat_cluster_idfs"]), template_descs=torch.from_numpy(r["template_descs"]),
        template_desc_opts=repre_util.TemplateDescOpts())
    bank = DeviceBank([repre])
    dets = [(pts[:1], feats[:1]), (pts, feats), (pts[5:12], feats[5:12])]
    qf = torch.from_numpy(np.concatenate([d[1] for d in dets])).cuda()
    qp = torch.from_numpy(np.concatenate([d[0] for d in dets])).cuda()
    res = match_batch(bank, qf, qp, [len(d[0]) for d in dets], None, 5, 300)
    for b, (p, f) in enumerate(dets):
        ora = om.establish_correspondences(p, f, r, 5, 300, topk_mode="canonical")
        got = res.corresp_list(b, debug=True)
        assert [int(x["template_id"]) for x in got] == [o["template_id"] for o in ora]
        for a, o in zip(got, ora):
            assert np.array_equal(a["coord_2d_ids"].cpu().numpy(), o["coord_2d_ids"])
            assert np.array_equal(a["nn_vertex_ids"].cpu().numpy(), o["nn_vertex_ids"])
            # all cycle distances of a 1-patch detection are 0 -> conf = 1 - 0/0 = NaN, like the reference (corresp_util.py:64)
            assert np.array_equal(a["coord_conf"].cpu().numpy(), o["coord_conf"], equal_nan=True)


# ------------------------------------------------------------------ randomized sweep: GPU == oracle on many seeded configurations
def _random_case(seed):
    rng = np.random.RandomState(seed)
    T = int(rng.choice([1, 2, 5, 6, 17, 40, 130]))
    pmin = int(rng.choice([1, 4, 20, 60]))
    pmax = pmin + int(rng.choice([0, 3, 25, 80]))
    W = int(rng.choice([16, 48, 128, 256]))
    noise = float(rng.choice([0.0, 0.0, 0.05, 0.5]))  # 0.0: exact copies of bank rows -> zero distances, heavy ties
    dup = min(int(rng.choice([0, 0, 3, 9])), pmin)  # duplicated query patches (exact ties); a template has >= pmin patches
    return dict(T=T, pmin=pmin, pmax=pmax, W=W, tpl=int(rng.randint(T)), noise=noise, top_n=int(rng.choice([1, 3, 5])),
                top_k=int(rng.choice([1, 7, 50, 300])), soft=bool(rng.rand() < 0.3), bank_seed=1000 + seed, q_seed=2000 + seed, dup=dup)


@pytest.mark.parametrize("seed", range(24))
def test_random_sweep_vs_oracle_both_tie_orders(seed):
    """Seeded random banks / queries (tiny to medium, ties, soft and hard assignment, fewer templates than top-n, fewer
    patches than top-k): the device result equals the oracle index for index in BOTH tie orders -- the canonical one and
    the reference's torch.topk CPU order."""
    from oracle.make_golden import build_match_inputs
    from foundpose_amd import corresp_util, repre_util
    c = _random_case(seed)
    bank, centroids, pts, feats = build_match_inputs(c)
    W = min(c["W"], bank["feat_vectors"].shape[0] // 4 * 4)  # the device path wants num_words % 4 == 0 (checked below)
    centroids = centroids[:W]
    opts = {"desc_type": "tfidf", "tfidf_knn_metric": "l2", "tfidf_knn_k": min(3, W), "tfidf_soft_assign": c["soft"],
            "tfidf_soft_sigma_squared": 10.0}
    r = om.build_synthetic_repre({k: v.numpy() for k, v in bank.items()}, centroids.numpy(), opts)
    repre = repre_util.FeatureBasedObjectRepre(
        vertices=torch.from_numpy(r["vertices"]), feat_vectors=torch.from_numpy(r["feat_vectors"]),
        feat_to_template_ids=torch.from_numpy(r["feat_to_template_ids"]),
        feat_cluster_centroids=torch.from_numpy(r["feat_cluster_centroids"]), feat_cluster_idfs=torch.from_numpy(r["feat_cluster_idfs"]),
        template_descs=torch.from_numpy(r["template_descs"]),
        template_desc_opts=repre_util.TemplateDescOpts(tfidf_knn_k=opts["tfidf_knn_k"], tfidf_soft_assign=c["soft"]))
    for order in ("canonical", "torch"):
        if c["top_n"] > c["T"]:
            # fewer templates than top_n: the reference's torch.topk raises (template_util.py:172) and so do the drop-in functions; the batched
            # form (match_batch / the engine, which serves objects with different template counts in one batch) returns what exists
            from foundpose_amd import template_util
            from foundpose_amd.matching import match_batch
            with pytest.raises(RuntimeError, match="selected index k out of range"):
                corresp_util.establish_correspondences(pts.cuda(), feats.cuda(), repre, "tfidf", "cyclic_buddies", c["top_n"], c["top_k"], tie_order=order)
            with pytest.raises(RuntimeError, match="selected index k out of range"):
                template_util.tfidf_matching(feats.cuda(), repre, c["top_n"])
            got = match_batch(template_util.get_device_bank(repre), feats.cuda(), pts.cuda(), [pts.shape[0]], None, c["top_n"], c["top_k"],
                              keep_debug=True, tie_order=order).corresp_list(0, debug=True)
        else:
            got = corresp_util.establish_correspondences(pts.cuda(), feats.cuda(), repre, "tfidf", "cyclic_buddies", c["top_n"], c["top_k"],
                                                         debug=True, tie_order=order)
        ora = om.establish_correspondences(pts.numpy(), feats.numpy(), r, c["top_n"], c["top_k"], topk_mode=order)
        assert [int(x["template_id"]) for x in got] == [o["template_id"] for o in ora], (order, c)
        for a, o in zip(got, ora):
            assert np.array_equal(a["coord_2d_ids"].cpu().numpy(), o["coord_2d_ids"]), (order, c)
            assert np.array_equal(a["nn_vertex_ids"].cpu().numpy(), o["nn_vertex_ids"]), (order, c)
            assert np.array_equal(a["nn_dists"].cpu().numpy(), o["nn_dists"]), (order, c)
            assert np.array_equal(a["coord_conf"].cpu().numpy(), o["coord_conf"], equal_nan=True), (order, c)
            # scores: the idf terms go through logf on the device and numpy's log in the oracle (1 ulp apart at times)
            assert abs(float(a["template_score"]) - float(o["template_score"])) <= 2e-6, (order, c)


def test_num_words_not_multiple_of_4_fails_loudly():
    from foundpose_amd._lib import FoundPoseNativeError
    with pytest.raises(FoundPoseNativeError, match="multiples of 4"):
        _cosine_topk(torch.rand(2, 14).cuda(), torch.rand(5, 14).cuda(), 3)


@pytest.mark.parametrize("T,order", [(10000, "ascending"), (10000, "descending"), (30000, "ascending"), (5000, "plateaus"), (2048, "ascending"), (2500, "plateaus")])
def test_strict_topn_adversarial_rows(T, order):
    """Rows on which torch.topk's partial_sort does the most work -- scores ascending along the row (every element replaces the
    heap's root; the candidate lists of the filter phase overflow and the row is replayed chunk by chunk), descending (nothing
    after the first five acts), long exact plateaus (ties everywhere) -- against torch.topk on the device's own scores."""
    from foundpose_amd import ops
    W, B = 16, 3
    t = torch.arange(T, dtype=torch.float64)
    if order == "ascending":
        theta = (1.0 - t / T) * 1.5
    elif order == "descending":
        theta = (t / T) * 1.5
    else:
        theta = torch.floor(t / 37.0) % 11 * 0.1   # 11 score levels in runs of 37
    bank = torch.zeros(T, W, dtype=torch.float64)
    bank[:, 0], bank[:, 1] = torch.cos(theta), torch.sin(theta)
    q = torch.zeros(B, W, dtype=torch.float64)
    q[:, 0] = 1.0
    q[1, 1] = 0.3
    q[2, 1] = -0.2
    sc, ids, sims = _cosine_topk(ops.normalize_rows(q.float().cuda()), ops.normalize_rows(bank.float().cuda()), 5, tie_mode=1)
    S = sims[:B * T].reshape(B, T).cpu()
    for b in range(B):
        tv, ti = torch.topk(S[b], 5, sorted=True)
        assert ids[b].cpu().tolist() == ti.tolist(), (order, b)
        assert torch.equal(sc[b].cpu(), tv)


# ------------------------------------------------------------------ prefiltered retrieval == single-pass retrieval, bit for bit
def _both_retrievals(desc_n, bank_n, seg, tpl_off, nt, n_top, tie_mode, max_det):
    """-> ((scores, ids) of fp_cosine_topk, (scores, ids) of fp_cosine_topk_prefiltered) on the same inputs."""
    from foundpose_amd._lib import call, cosine_prefilter_scratch_floats, cosine_scratch_floats, ptr, stream
    B, W = desc_n.shape
    T = int((tpl_off[1:] - tpl_off[:-1]).max())
    bank_bf = bank_n.to(torch.float16).contiguous()
    out = []
    for pre in (False, True):
        sims = torch.full((cosine_prefilter_scratch_floats(B, T) if pre else cosine_scratch_floats(B, T),), float("nan"), device="cuda")  # poisoned scratch
        sc, ids = torch.empty(B, n_top, device="cuda"), torch.empty(B, n_top, dtype=torch.int32, device="cuda")
        if pre:
            call("fp_cosine_topk_prefiltered", ptr(desc_n), ptr(seg), ptr(nt), B, max_det, ptr(bank_n), ptr(bank_bf), ptr(tpl_off), tpl_off.shape[0] - 1, T, W,
                 n_top, ptr(sims), ptr(sc), ptr(ids), tie_mode | 256, stream())   # | FP_COSINE_FORCE_PREFILTER: the two-stage form at every size
        else:
            call("fp_cosine_topk", ptr(desc_n), ptr(seg), ptr(nt), B, max_det, ptr(bank_n), ptr(tpl_off), tpl_off.shape[0] - 1, T, W, n_top, ptr(sims), ptr(sc),
                 ptr(ids), tie_mode, stream())
        torch.cuda.synchronize()
        out.append((sc.clone(), ids.clone()))
    return out


def _assert_same(a, b, what):
    assert torch.equal(a[1], b[1]), f"{what}: ids differ\n{a[1]}\n{b[1]}"
    assert torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)), f"{what}: scores differ"


@pytest.mark.parametrize("tie_mode", [0, 1])
@pytest.mark.parametrize("T,W,B", [(10000, 2048, 32), (3000, 1024, 7), (800, 4096, 40), (16, 2048, 3), (3, 2048, 2), (50000, 2048, 32)])
def test_prefiltered_retrieval_equals_single_pass(T, W, B, tie_mode):
    """The two-stage retrieval (fp16 candidate pass + exact re-scoring) returns the single-pass kernel's scores and ids bit for bit: tf-idf-like
    sparse banks at the metric's size and config 5's, more than 32 detections of an object (two chunks), fewer templates than n_top + 1."""
    from foundpose_amd import ops
    bank_n = _full_size_bank(T, W, seed=T + W)
    g = torch.Generator(device="cuda").manual_seed(B)
    desc_n = ops.normalize_rows(torch.rand(B, W, generator=g, device="cuda") * (torch.rand(B, W, generator=g, device="cuda") < 0.05) + 1e-4)
    desc_n[0] = bank_n[T // 2]                                  # one query is a bank row: score 1 at the top
    seg = torch.tensor([0, B], dtype=torch.int32, device="cuda")
    off = torch.tensor([0, T], dtype=torch.int32, device="cuda")
    nt = torch.full((B,), T, dtype=torch.int32, device="cuda")
    a, b = _both_retrievals(desc_n, bank_n, seg, off, nt, 5, tie_mode, B)
    _assert_same(a, b, f"T={T} W={W} B={B} tie_mode={tie_mode}")
    if T >= 5:
        assert int(b[1][0, 0]) == T // 2


@pytest.mark.parametrize("tie_mode", [0, 1])
def test_prefiltered_retrieval_adversarial_rows(tie_mode):
    """Rows built against the candidate logic: (a) duplicated templates at the top (exact ties -> the strict order releases the fallback),
    (b) a tie across the cut, (c) every template identical (everything is a candidate and everything ties), (d) a dense band of
    near-equal scores around rank 6 (hundreds of candidates inside 2 eps), (e) all scores exactly zero, (f) a NaN in a descriptor."""
    from foundpose_amd import ops
    rng = np.random.default_rng(7)
    T, W = 2000, 2048
    base = (rng.random((T, W)) * (rng.random((T, W)) < 0.03)).astype(np.float32)
    base[:, 0] += 1e-3
    q = (rng.random(W) * (rng.random(W) < 0.1)).astype(np.float32) + 1e-4
    order = np.argsort(-(base / np.linalg.norm(base, axis=1, keepdims=True)) @ (q / np.linalg.norm(q)))
    bank_a = base.copy(); bank_a[order[1]] = bank_a[order[0]]; bank_a[order[3]] = bank_a[order[2]]
    bank_b = base.copy(); bank_b[order[5]] = bank_b[order[4]]
    bank_c = np.repeat(base[:1], T, 0)
    bank_d = base.copy()
    for r in range(5, 405):                                      # 400 templates = rank-6 template + a tiny perturbation
        bank_d[order[r]] = base[order[5]] * (1.0 + 1e-4 * rng.standard_normal(W).astype(np.float32))
    qz = np.zeros(W, np.float32); qz[W // 2:] = 1.0
    bank_e = base.copy(); bank_e[:, W // 2:] = 0.0
    cases = [("a", bank_a, q), ("b", bank_b, q), ("c", bank_c, q), ("d", bank_d, q), ("e", bank_e, qz)]
    seg, off, nt = cu(np.array([0, 1], np.int32)), cu(np.array([0, T], np.int32)), cu(np.full(1, T, np.int32))
    for name, bk, qq in cases:
        bank_n, q_n = ops.normalize_rows(cu(bk)), ops.normalize_rows(cu(qq[None]))
        a, b = _both_retrievals(q_n, bank_n, seg, off, nt, 5, tie_mode, 1)
        _assert_same(a, b, f"case {name} tie_mode={tie_mode}")
    bank_n = ops.normalize_rows(cu(base))
    bank_n[order[2], 7] = float("nan")                           # (f) a NaN score ranks first in both paths
    a, b = _both_retrievals(ops.normalize_rows(cu(q[None])), bank_n, seg, off, nt, 5, tie_mode, 1)
    assert torch.equal(a[1], b[1]) and torch.equal(a[0].isnan(), b[0].isnan()) and torch.equal(a[0].nan_to_num(7.0), b[0].nan_to_num(7.0))


@pytest.mark.parametrize("W", [3072, 4096])
def test_prefiltered_retrieval_wide_banks_with_ties_strict_order(W):
    """with 3072 / 4096 words the strict order's exact fallback used to be the <= 2048-word single-pass kernel (garbage
    scores for any row with a tie among its best n + 1).  Rows WITH ties -- duplicated templates at the top, a tie across the cut, all
    templates identical -- in the torch order and in the canonical one, forced two-stage call vs fp_cosine_topk, and vs torch.topk."""
    from foundpose_amd import ops
    rng = np.random.default_rng(W)
    T = 1500
    base = (rng.random((T, W)) * (rng.random((T, W)) < 0.03)).astype(np.float32)
    base[:, 0] += 1e-3
    q = (rng.random(W) * (rng.random(W) < 0.1)).astype(np.float32) + 1e-4
    order = np.argsort(-(base / np.linalg.norm(base, axis=1, keepdims=True)) @ (q / np.linalg.norm(q)))
    bank_a = base.copy(); bank_a[order[1]] = bank_a[order[0]]; bank_a[order[3]] = bank_a[order[2]]
    bank_b = base.copy(); bank_b[order[5]] = bank_b[order[4]]
    bank_c = np.repeat(base[:1], T, 0)
    seg, off, nt = cu(np.array([0, 1], np.int32)), cu(np.array([0, T], np.int32)), cu(np.full(1, T, np.int32))
    for name, bk in (("a", bank_a), ("b", bank_b), ("c", bank_c)):
        bank_n, q_n = ops.normalize_rows(cu(bk)), ops.normalize_rows(cu(q[None]))
        for tie_mode in (0, 1):
            a, b = _both_retrievals(q_n, bank_n, seg, off, nt, 5, tie_mode, 1)
            _assert_same(a, b, f"W={W} case {name} tie_mode={tie_mode}")
        sc, ids, sims = _cosine_topk(q_n, bank_n, 5, tie_mode=1)
        tv, ti = torch.topk(sims[:T].cpu(), 5, sorted=True)
        assert b[1][0].cpu().tolist() == ti.tolist() and torch.equal(b[0][0].cpu(), tv), (W, name)


def test_prefiltered_retrieval_multi_object_groups():
    """Three objects with different template counts, detections grouped by object (one of them with no detection): the prefiltered call
    serves every (object, chunk) pair like the single-pass call."""
    from foundpose_amd import ops
    Ts, W = [700, 2500, 64], 2048
    bank_n = torch.cat([_full_size_bank(t, W, seed=11 + i) for i, t in enumerate(Ts)])
    off = torch.tensor([0, 700, 3200, 3264], dtype=torch.int32, device="cuda")
    dets = [35, 0, 4]                                           # 35 detections of object 0 (two chunks), none of object 1, four of object 2
    B = sum(dets)
    g = torch.Generator(device="cuda").manual_seed(2)
    desc_n = ops.normalize_rows(torch.rand(B, W, generator=g, device="cuda") * (torch.rand(B, W, generator=g, device="cuda") < 0.05) + 1e-4)
    seg = torch.tensor([0, 35, 35, 39], dtype=torch.int32, device="cuda")
    nt = torch.tensor([700] * 35 + [64] * 4, dtype=torch.int32, device="cuda")
    for tie_mode in (0, 1):
        a, b = _both_retrievals(desc_n, bank_n, seg, off, nt, 5, tie_mode, 35)
        _assert_same(a, b, f"multi-object tie_mode={tie_mode}")


# ------------------------------------------------------------------ the opt-in two-stage k-NN (csrc/knn_cand.hip) == the all-pairs exact tile, bit for bit
def _knn_both(monkeypatch, q, db, k):
    from foundpose_amd import ops
    out = []
    for on in ("0", "1"):
        monkeypatch.setenv("FP_KNN_CAND", on)
        d2, idx = ops.knn_l2(cu(q), cu(db), k)
        torch.cuda.synchronize()
        out.append((d2.cpu().numpy().copy(), idx.cpu().numpy().copy()))
    return out


@pytest.mark.skipif(not experiments_build(), reason="csrc/knn_cand.hip (measured slower than the exact tile) and its FP_KNN_CAND switch are compiled into FP_EXPERIMENTS builds only")
@pytest.mark.parametrize("m,n,K,k", [(5, 300, 64, 3), (70, 300, 64, 2), (200, 700, 128, 3), (300, 2048, 256, 3), (1000, 2048, 256, 3), (37, 1000, 256, 4), (129, 257, 256, 3)])
def test_two_stage_knn_equals_exact_tile(monkeypatch, m, n, K, k):
    """fp16-MFMA candidate pass + exact fp32 chains on the candidates (FP_KNN_CAND=1): distances and indices equal the all-pairs exact-fp32
    tile's and the oracle's, bit for bit, on unstructured data (every distance within a few percent of every other: the hardest case for a
    candidate window) at several shapes incl. ragged row blocks and database tails."""
    rng = np.random.default_rng(m + n + K)
    q, db = rng.standard_normal((m, K)).astype(np.float32), rng.standard_normal((n, K)).astype(np.float32)
    (d0, i0), (d1, i1) = _knn_both(monkeypatch, q, db, k)
    assert np.array_equal(i0, i1) and np.array_equal(d0.view(np.int32), d1.view(np.int32))
    o_d2, o_idx = clib.l2_knn(q, db, k)
    assert np.array_equal(i1, o_idx) and np.array_equal(d1, o_d2)


@pytest.mark.skipif(not experiments_build(), reason="csrc/knn_cand.hip (measured slower than the exact tile) and its FP_KNN_CAND switch are compiled into FP_EXPERIMENTS builds only")
def test_two_stage_knn_adversarial_rows(monkeypatch):
    """Rows built against the candidate logic: exact duplicates in the database (ties -> lowest index), a query that IS a database row
    (distance 0: the clamp), every database row identical (more candidates than a list holds -> the brute-force rows of stage 2), values
    beyond the fp16 range and a NaN-free but huge dynamic range (-> brute force), tiny values in the fp16 subnormal range, zero vectors."""
    rng = np.random.default_rng(5)
    n, K, k = 1024, 256, 3
    db = rng.standard_normal((n, K)).astype(np.float32)
    db[700] = db[3]; db[701] = db[3]; db[20] = db[900]
    q = rng.standard_normal((40, K)).astype(np.float32)
    q[0] = db[3]                      # three exact ties at distance 0
    q[1] = db[900] + 1e-4             # near-ties between rows 20 and 900
    q[2] = 0.0
    q[3] *= 1e-6                      # fp16-subnormal elements
    q[4] *= 3e4                       # elements beyond 65504 -> the row is taken by brute force
    (d0, i0), (d1, i1) = _knn_both(monkeypatch, q, db, k)
    assert np.array_equal(i0, i1) and np.array_equal(d0.view(np.int32), d1.view(np.int32))
    assert i1[0].tolist() == [3, 700, 701] and d1[0].tolist() == [0.0, 0.0, 0.0]
    same = np.repeat(db[:1], n, 0)    # every row inside every window
    (d0, i0), (d1, i1) = _knn_both(monkeypatch, q[:8], same, k)
    assert np.array_equal(i0, i1) and np.array_equal(d0.view(np.int32), d1.view(np.int32)) and i1[5].tolist() == [0, 1, 2]
    big = db.copy(); big[77] *= 1e5   # a database row beyond the fp16 range: every row of the launch falls back
    (d0, i0), (d1, i1) = _knn_both(monkeypatch, q, big, k)
    assert np.array_equal(i0, i1) and np.array_equal(d0.view(np.int32), d1.view(np.int32))
    tiny = (db * 1e-7).astype(np.float32)
    (d0, i0), (d1, i1) = _knn_both(monkeypatch, (q * 1e-7).astype(np.float32), tiny, k)
    assert np.array_equal(i0, i1) and np.array_equal(d0.view(np.int32), d1.view(np.int32))


@pytest.mark.skipif(not experiments_build(), reason="csrc/knn_cand.hip (measured slower than the exact tile) and its FP_KNN_CAND switch are compiled into FP_EXPERIMENTS builds only")
@pytest.mark.parametrize("Q,P", [(5, 140), (140, 5), (133, 133), (517, 389), (389, 517), (1, 300), (300, 1)])
def test_two_stage_cyclic_equals_exact_tile(monkeypatch, Q, P):
    """The two 1-NN searches of cyclic_buddies_matching through the two-stage path (segmented, both directions): ids, cycle distances and
    scores equal the all-pairs path's and the oracle's, ties (duplicated query / template patches) included."""
    from foundpose_amd import corresp_util
    rng = np.random.default_rng(Q * 1000 + P + 1)
    obj = rng.standard_normal((P, 64)).astype(np.float32)
    qf = rng.standard_normal((Q, 64)).astype(np.float32)
    n_copy = min(Q, P) // 2
    qf[:n_copy] = obj[rng.permutation(P)[:n_copy]]
    if Q > 3:
        qf[Q - 1] = qf[0]
    if P > 3:
        obj[P - 1] = obj[1]
    pts = (rng.integers(0, 37, (Q, 2)) * 14 + 7).astype(np.float32)
    got = []
    for on in ("0", "1"):
        monkeypatch.setenv("FP_KNN_CAND", on)
        got.append([t.cpu().numpy().copy() for t in corresp_util.cyclic_buddies_matching(cu(pts), cu(qf), None, cu(obj), None, 300)])
    for a_, b_ in zip(*got):
        assert np.array_equal(a_, b_, equal_nan=True)
    o_q, o_o, o_d, o_s, _ = om.cyclic_buddies(pts, qf, obj, 300, topk_mode="torch")
    assert np.array_equal(got[1][0], o_q) and np.array_equal(got[1][1], o_o) and np.array_equal(got[1][2], o_d)
