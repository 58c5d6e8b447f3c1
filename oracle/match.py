"""CPU oracle for the descriptor-matching half (TEST INFRASTRUCTURE ONLY).

Restates, function by function (reference paths relative to /root/reference):
  knn_l2 ............... utils/knn_util.py:38-106 (faiss IndexFlatL2: squared dists, i64 ids)
  nearest_words ........ utils/template_util.py:13-29 (search + sqrt, returns ids first)
  calc_tfidf ........... utils/template_util.py:31-71
  calc_tfidf_descriptors utils/template_util.py:74-123 (bank side; squared dists there)
  tfidf_matching ....... utils/template_util.py:126-176
  cyclic_buddies ....... utils/corresp_util.py:34-70
  establish_correspondences utils/corresp_util.py:73-169

Arithmetic conventions (the canonical definition the HIP kernels are held to,
bit for bit): every distance / dot product is a k-ascending fp32 fmaf chain
(oracle/csrc/oracle.cpp). `topk_mode`:
  "torch"     -> tie-for-tie what torch.topk does on CPU (the reference's behaviour)
  "canonical" -> best value first, ties broken by lowest index (GPU fast path)
"""

from typing import Dict, List, Optional, Tuple

import numpy as np

from . import clib

EPS_COS = 1e-8  # torch.nn.functional.cosine_similarity default eps


def _topk(values: np.ndarray, k: int, largest: bool, mode: str):
    if mode == "torch":
        return clib.topk_torch(values, k, largest, True)
    if mode == "canonical":
        return clib.topk_canonical(values, k, largest)
    raise ValueError(mode)


def knn_l2(queries: np.ndarray, db: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """-> (squared distances [m,k] f32 ascending, indices [m,k] i64)."""
    return clib.l2_knn(queries, db, k)


def unit_rows(x: np.ndarray) -> np.ndarray:
    """x / ||x|| without an eps (KNN's cosine metric, knn_util.py:55, 94), ||x||^2 as the k-ascending fmaf chain."""
    with np.errstate(invalid="ignore", divide="ignore"):
        return (x / np.sqrt(clib.sqnorm(x)).astype(np.float32)[:, None]).astype(np.float32)


def nearest_words(query_features: np.ndarray, centroids: np.ndarray, k: int, metric: str = "l2"):
    """find_nearest_object_features over KNN(metric).  "cosine" (knn_util.py:52-57, 91-100): unit rows on both sides, inner-product
    search, distance 1 - similarity -- restated as the L2 search on the unit rows (same ranking; 1 - a.b = |a - b|^2 / 2),
    the form the device computes."""
    if metric == "cosine":
        d2, ids = knn_l2(unit_rows(query_features), unit_rows(centroids), k)
        return ids, np.sqrt(d2 * np.float32(0.5))
    if metric != "l2":
        raise ValueError(f"Metric {metric} is not supported.")
    d2, ids = knn_l2(query_features, centroids, k)
    return ids, np.sqrt(d2)


def calc_tfidf(
    word_ids: np.ndarray,
    word_dists: np.ndarray,
    word_idfs: np.ndarray,
    soft_assignment: bool = True,
    soft_sigma_squared: float = 100.0,
) -> np.ndarray:
    f32 = np.float32
    if soft_assignment:
        w = np.exp(-np.square(word_dists.astype(f32)) / f32(2.0 * soft_sigma_squared)).astype(f32)
    else:
        w = np.ones_like(word_dists, dtype=f32)
    # torch.nn.functional.normalize(p=2, dim=1): x / max(||x||_2, 1e-12)
    nrm = np.sqrt((w * w).sum(axis=1, dtype=f32)).astype(f32)
    w = (w / np.maximum(nrm, f32(1e-12))[:, None]).astype(f32).reshape(-1)
    tf = (w / f32(word_ids.shape[0])).astype(f32)
    ids_flat = word_ids.reshape(-1).astype(np.int64)
    tfidf = (tf * word_idfs.astype(f32)[ids_flat]).astype(f32)
    return clib.scatter_add(ids_flat, tfidf, word_idfs.shape[0])


def calc_word_idfs(feat_to_word_ids: np.ndarray, feat_to_template_ids: np.ndarray, num_words: int, num_templates: int):
    occ = np.zeros(num_words, np.int64)
    for t in range(num_templates):
        occ[np.unique(feat_to_word_ids[feat_to_template_ids == t])] += 1
    with np.errstate(divide="ignore"):
        return np.log(np.float32(num_templates) / occ.astype(np.float32)).astype(np.float32)


def calc_tfidf_descriptors(
    feat_vectors, feat_to_word_ids, feat_to_template_ids, feat_words, num_templates,
    tfidf_knn_k, tfidf_soft_assign, tfidf_soft_sigma_squared,
):
    idfs = calc_word_idfs(feat_to_word_ids, feat_to_template_ids, len(feat_words), num_templates)
    descs = []
    for t in range(num_templates):
        feats = feat_vectors[feat_to_template_ids == t]
        d2, ids = knn_l2(feats, feat_words, tfidf_knn_k)
        # NB: the bank side passes *squared* distances (template_util.py:112-119).
        descs.append(calc_tfidf(ids, d2, idfs, tfidf_soft_assign, tfidf_soft_sigma_squared))
    return np.stack(descs, 0), idfs


def l2_normalize_rows(x: np.ndarray, eps: float = EPS_COS) -> np.ndarray:
    """x / max(||x||, eps) with ||x||^2 a k-ascending fmaf chain (bank prep + query prep)."""
    n = np.sqrt(clib.sqnorm(x)).astype(np.float32)
    return (x / np.maximum(n, np.float32(eps))[:, None]).astype(np.float32)


def cosine_scores(template_descs: np.ndarray, query_tfidf: np.ndarray) -> np.ndarray:
    """cosine_similarity(template_descs, tile(query)): normalise each side, then dot."""
    bank_n = l2_normalize_rows(template_descs)
    q_n = l2_normalize_rows(query_tfidf[None, :])[0]
    # chain order of the device kernel: 16-block permuted order when W % 16 == 0 (8 k-slices when W % 128 == 0),
    # k ascending otherwise
    return clib.dot_rows(bank_n, q_n, perm16=(bank_n.shape[1] % 16 == 0))


def tfidf_matching(query_features, repre: Dict, top_n: int, topk_mode: str = "torch"):
    opts = repre["template_desc_opts"]
    ids, dists = nearest_words(query_features, repre["feat_cluster_centroids"], opts["tfidf_knn_k"], opts.get("tfidf_knn_metric", "l2"))
    q_tfidf = calc_tfidf(ids, dists, repre["feat_cluster_idfs"], opts["tfidf_soft_assign"], opts["tfidf_soft_sigma_squared"])
    sims = cosine_scores(repre["template_descs"], q_tfidf)
    # (torch.topk raises when there are fewer templates than top_n, template_util.py:172; the device path returns the
    #  templates there are, padded with -1 -- the oracle follows the device here so the case stays comparable)
    scores, tids = _topk(sims, min(top_n, sims.shape[0]), True, topk_mode)
    return tids, scores, {"word_ids": ids, "word_dists": dists, "query_tfidf": q_tfidf, "sims": sims}


def cyclic_buddies(query_points, query_features, object_features, top_k: int, topk_mode: str = "torch"):
    q2o = clib.l2_knn(query_features, object_features, 1)[1][:, 0]
    o2q = clib.l2_argmin_cols(query_features, object_features)[0]
    cycle_ids = o2q[q2o]
    diff = (query_points - query_points[cycle_ids]).astype(np.float32)
    # torch.linalg.norm(axis=1) on [Q,2]: sqrt(x^2 + y^2) in fp32
    cycle_dists = np.sqrt((diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]).astype(np.float32)).astype(np.float32)
    k = min(top_k, query_points.shape[0])
    _, q_ids = _topk(-cycle_dists, k, True, topk_mode)
    bb_dists = cycle_dists[q_ids]
    with np.errstate(invalid="ignore", divide="ignore"):
        bb_scores = (np.float32(1.0) - bb_dists / bb_dists.max()).astype(np.float32)
    return q_ids, q2o[q_ids], bb_dists, bb_scores, {"q2o": q2o, "o2q": o2q, "cycle_dists": cycle_dists}


def template_offsets(feat_to_template_ids: np.ndarray, num_templates: int) -> np.ndarray:
    counts = np.bincount(feat_to_template_ids.astype(np.int64), minlength=num_templates)
    return np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)


def establish_correspondences(
    query_points: np.ndarray,
    query_features: np.ndarray,
    repre: Dict,
    top_n_templates: int = 5,
    top_k_buddies: int = 300,
    topk_mode: str = "torch",
) -> List[Dict]:
    tids, tscores, _ = tfidf_matching(query_features, repre, top_n_templates, topk_mode)
    f2t = repre["feat_to_template_ids"]
    out = []
    for c, tid in enumerate(tids):
        feat_ids = np.nonzero(f2t == tid)[0]
        q_ids, o_ids, dists, scores, _ = cyclic_buddies(
            query_points, query_features, repre["feat_vectors"][feat_ids], top_k_buddies, topk_mode
        )
        obj_feat_ids = feat_ids[o_ids]
        out.append({
            "template_id": int(tid),
            "template_score": np.float32(tscores[c]),
            "coord_2d": query_points[q_ids],
            "coord_2d_ids": q_ids,
            "coord_3d": repre["vertices"][obj_feat_ids],
            "coord_conf": scores,
            "nn_vertex_ids": obj_feat_ids,
            "nn_dists": dists,
        })
    return out


def build_synthetic_repre(bank: Dict, centroids: np.ndarray, opts: Optional[Dict] = None) -> Dict:
    """Bank builder on the oracle (small banks only): word assignment, idf, template descs."""
    opts = opts or {"desc_type": "tfidf", "tfidf_knn_metric": "l2", "tfidf_knn_k": 3,
                    "tfidf_soft_assign": False, "tfidf_soft_sigma_squared": 10.0}
    fv = np.asarray(bank["feat_vectors"], np.float32)
    f2t = np.asarray(bank["feat_to_template_ids"], np.int32)
    T = int(f2t.max()) + 1
    word_ids = clib.l2_knn(fv, centroids, 1)[1][:, 0].astype(np.int32)
    descs, idfs = calc_tfidf_descriptors(
        fv, word_ids, f2t, centroids, T, opts["tfidf_knn_k"], opts["tfidf_soft_assign"], opts["tfidf_soft_sigma_squared"]
    )
    return {
        "vertices": np.asarray(bank["vertices"], np.float32),
        "feat_vectors": fv,
        "feat_to_vertex_ids": np.asarray(bank["feat_to_vertex_ids"], np.int32),
        "feat_to_template_ids": f2t,
        "feat_to_cluster_ids": word_ids,
        "feat_cluster_centroids": np.asarray(centroids, np.float32),
        "feat_cluster_idfs": idfs,
        "template_descs": descs,
        "template_desc_opts": opts,
    }
