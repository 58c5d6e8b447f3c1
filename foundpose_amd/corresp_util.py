"""2D-3D correspondence establishment with the reference's signatures
(/root/reference/utils/corresp_util.py:34-169), executed on the MI355X.

Tie order: the drop-in functions default to `tie_order="torch"` -- among equal cycle distances (and equal template
scores) the selection and order are exactly those of the reference's `torch.topk` on CPU tensors, replayed on the
device (csrc/stl_order.hpp). The batched engine defaults to the cheaper canonical order (value, then lowest index).

`visual_words_knn_index` / `template_knn_indices` (faiss indices the reference builds per object and per
template, scripts/infer.py:216-239) are accepted for call compatibility and not needed: the HBM bank is
CSR-indexed by template, so no per-template index or mask scan exists.  What IS read from a passed
`visual_words_knn_index` is its `.metric` ("l2" | "cosine"): the reference searches the words with whatever index the
caller built (infer.py:218-222 builds it from template_desc_opts.tfidf_knn_metric, the default here).
"""

from typing import Any, Dict, List, Optional, Tuple

import torch

from . import knn_util, ops, repre_util, template_util
from .matching import match_batch


def convert_px_indices_to_im_coords(px_indices: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    return scale * (px_indices.float() + 0.5)


def cyclic_buddies_matching(query_points: torch.Tensor, query_features: torch.Tensor, query_knn_index: Optional[knn_util.KNN],
                            object_features: torch.Tensor, object_knn_index: Optional[knn_util.KNN], top_k: int,
                            debug: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Best buddies of one (query, template) pair -> (query ids, object ids, cycle dists, scores)."""
    repre = repre_util.FeatureBasedObjectRepre(
        vertices=torch.zeros(object_features.shape[0], 3), feat_vectors=object_features,
        feat_to_template_ids=torch.zeros(object_features.shape[0], dtype=torch.int32),
        feat_cluster_centroids=object_features[:1].repeat(4, 1), feat_cluster_idfs=torch.ones(4),
        template_descs=torch.ones(1, 4), template_desc_opts=repre_util.TemplateDescOpts(tfidf_knn_k=1))
    bank = template_util.get_device_bank(repre)
    res = match_batch(bank, query_features.to("cuda"), query_points.to("cuda"), [query_points.shape[0]], None, 1, top_k, tie_order="torch")
    c = int(res.counts[0, 0])
    dev = query_points.device
    return (res.q_ids[0, 0, :c].to(torch.int64).to(dev), res.feat_ids[0, 0, :c].to(torch.int64).to(dev),
            res.dists[0, 0, :c].to(dev), res.conf[0, 0, :c].to(dev))


def establish_correspondences(
    query_points: torch.Tensor,
    query_features: torch.Tensor,
    object_repre: repre_util.FeatureBasedObjectRepre,
    template_matching_type: str,
    feat_matching_type: str,
    top_n_templates: int,
    top_k_buddies: int,
    visual_words_knn_index: Optional[knn_util.KNN] = None,
    template_knn_indices: Optional[List[knn_util.KNN]] = None,
    debug: bool = False,
    tie_order: str = "torch",
) -> List[Dict]:
    if template_matching_type != "tfidf":
        raise ValueError(f"Unknown matching type '{template_matching_type}'.")
    if feat_matching_type != "cyclic_buddies":
        raise ValueError(f"Unknown feature matching type ({feat_matching_type}).")
    assert object_repre.feat_vectors is not None
    assert object_repre.vertices is not None
    template_util.check_top_n(top_n_templates, object_repre)   # torch.topk's error for an object with fewer templates (template_util.py:172)
    bank = template_util.get_device_bank(object_repre)
    res = match_batch(bank, query_features.to("cuda"), query_points.to("cuda"), [query_points.shape[0]], None,
                      top_n_templates, top_k_buddies, keep_debug=debug, tie_order=tie_order,
                      word_metric=getattr(visual_words_knn_index, "metric", None))
    return res.corresp_list(0, debug=debug)
