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

from . import vit as ov


def _l2(q: torch.Tensor, db: torch.Tensor) -> torch.Tensor:
    return ((q * q).sum(1, keepdim=True) + (db * db).sum(1)[None, :] - 2.0 * (q @ db.T)).clamp_min_(0)


@torch.no_grad()
def run_detection(sd, arch, layer: int, image: torch.Tensor, mask: torch.Tensor, bank: Dict, top_n: int = 5, top_k: int = 300):
    """One detection end to end; returns (times dict with the reference's keys, list of corresp dicts)."""
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
    return t, out
