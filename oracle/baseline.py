"""Reference-equivalent CPU path, timed as bench.py's `cpu_baseline` (TEST INFRASTRUCTURE ONLY).

Executes the per-detection hot section the way the reference does (/root/reference/scripts/infer.py:468-542):
batch of one, fp32 torch on the host cores, the WHOLE backbone run even past the hooked layer
(utils/dinov2_utils.py:257), PCA as X@C^T - mu@C^T, brute-force L2 searches (BLAS formulation, like faiss for
>= 20 queries) for the visual words and for each of the retrieved templates in both directions, torch.topk.
The reference itself cannot run on the GPU box (its files never ship; faiss/dinov2/cv2 are absent).
"""

import time
from typing import Dict, List

import torch

import numpy as np

from . import match as om
from . import vit as ov


def _l2(q: torch.Tensor, db: torch.Tensor) -> torch.Tensor:
    return ((q * q).sum(1, keepdim=True) + (db * db).sum(1)[None, :] - 2.0 * (q @ db.T)).clamp_min_(0)


@torch.no_grad()
def run_detection(sd, arch, layer: int, image: torch.Tensor, mask: torch.Tensor, bank: Dict, top_n: int = 5, top_k: int = 300,
                  return_features: bool = False):
    """One detection end to end; returns (times dict with the reference's keys, list of corresp dicts)
    [+ (query_points, projected query_features) with return_features]."""
    t: Dict[str, float] = {}
    S = image.shape[-1]
    t0 = time.perf_counter()
    fmap = ov.extractor_forward(sd, arch, image.unsqueeze(0), layer, True, all_blocks=True)["feature_maps"][0]
    t["feat_extract"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    grid = ov.generate_grid_points((S, S), 14.0)
    qp = ov.filter_points_by_mask(grid, mask)
    qf = ov.sample_feature_map_at_points(fmap, qp, (S, S)).contiguous()
    t["grid_sample"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    if "pca_components" in bank:
        qf = ov.pca_transform(qf, bank["pca_components"], bank["pca_mean"]).contiguous()
    t["proj"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    d2 = _l2(qf, bank["feat_cluster_centroids"])
    wd2, wid = torch.topk(d2, 3, dim=1, largest=False)
    w = torch.nn.functional.normalize(torch.ones_like(wd2), p=2, dim=1).reshape(-1)
    tfidf = torch.zeros(bank["feat_cluster_idfs"].shape[0]).scatter_add_(0, wid.reshape(-1), (w / qf.shape[0]) * bank["feat_cluster_idfs"][wid.reshape(-1)])
    sims = torch.nn.functional.cosine_similarity(bank["template_descs"], tfidf.tile(bank["template_descs"].shape[0], 1))
    scores, tids = torch.topk(sims, top_n, sorted=True)
    out: List[Dict] = []
    f2t = bank["feat_to_template_ids"]
    for tid in tids:
        ids = torch.nonzero(f2t == tid).flatten()  # the reference's O(N_f) mask scan per template
        tf = bank["feat_vectors"][ids]
        dm = _l2(qf, tf)
        q2o, o2q = dm.argmin(1), dm.argmin(0)
        cyc = o2q[q2o]
        cd = torch.linalg.norm(qp - qp[cyc], axis=1)
        k = min(top_k, qp.shape[0])
        _, sel = torch.topk(-cd, k=k, sorted=True)
        out.append({"template_id": int(tid), "coord_2d_ids": sel, "nn_vertex_ids": ids[q2o[sel]], "nn_dists": cd[sel]})
    t["corresp"] = time.perf_counter() - t0
    if return_features:
        return t, out, (qp, qf)
    return t, out


def exact_matching(query_points, query_features, small: Dict, fetch_template, top_n: int = 5, top_k: int = 300, topk_mode: str = "torch",
                   return_words: bool = False):
    """establish_correspondences (utils/corresp_util.py:73-169) in the oracle's pinned arithmetic (oracle/match.py:
    fixed-order fp32 chains, the reference's torch.topk tie order) on a bank that is too large to copy to the host as a
    whole: `small` holds feat_cluster_centroids / feat_cluster_idfs / template_descs / template_desc_opts, and
    fetch_template(tid) -> (features [P, d] of that template, index of its first feature row in the object)."""
    qp = np.ascontiguousarray(np.asarray(query_points, np.float32))
    qf = np.ascontiguousarray(np.asarray(query_features, np.float32))
    tids, tscores, dbg = om.tfidf_matching(qf, small, top_n, topk_mode)
    out: List[Dict] = []
    for c, tid in enumerate(tids):
        feats, first = fetch_template(int(tid))
        q_ids, o_ids, dists, scores, _ = om.cyclic_buddies(qp, qf, np.ascontiguousarray(np.asarray(feats, np.float32)), top_k, topk_mode)
        out.append({"template_id": int(tid), "template_score": np.float32(tscores[c]), "coord_2d_ids": q_ids,
                    "nn_vertex_ids": first + o_ids, "nn_dists": dists, "coord_conf": scores})
    if return_words:
        return out, dbg["word_ids"]
    return out


@torch.no_grad()
def oracle_a_features(sd, arch, layer: int, image: torch.Tensor, mask: torch.Tensor, pca_components=None, pca_mean=None, quant=None):
    """Oracle A's query side of one detection: fp32 extractor (blocks 0..layer only -- the later blocks the reference
    also runs do not feed the hooked output), mask-filtered grid points, bilinear samples, PCA.  -> (points, features).
    quant="bf16": ORACLE B (hard part 2) -- the same computation with every GEMM / attention operand rounded to bf16 at the
    device's cast points (oracle/vit.py), fp32 accumulation; everything behind the backbone stays the fp32 arithmetic."""
    S = image.shape[-1]
    fmap = ov.extractor_forward(sd, arch, image.unsqueeze(0), layer, True, quant=quant)["feature_maps"][0]
    qp = ov.filter_points_by_mask(ov.generate_grid_points((S, S), 14.0), mask)
    qf = ov.sample_feature_map_at_points(fmap, qp, (S, S)).contiguous()
    if pca_components is not None:
        qf = ov.pca_transform(qf, pca_components, pca_mean).contiguous()
    return qp, qf
