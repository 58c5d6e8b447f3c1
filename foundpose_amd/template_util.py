"""tf-idf bag-of-visual-words template retrieval with the reference's function names
(/root/reference/utils/template_util.py), executed by the MI355X kernels."""

from typing import Optional, Tuple

import torch

from . import knn_util, ops, repre_util
from .bank import DeviceBank


def get_device_bank(object_repre: repre_util.FeatureBasedObjectRepre) -> DeviceBank:
    """One HBM-resident bank per object representation, built on first use."""
    bank = getattr(object_repre, "_device_bank", None)
    if bank is None:
        bank = DeviceBank([object_repre])
        object_repre._device_bank = bank
    return bank


def check_top_n(top_n_templates: int, object_repre: repre_util.FeatureBasedObjectRepre) -> None:
    """The reference selects with torch.topk(scores, k=top_n_templates) (template_util.py:172), which raises when the object has fewer
    templates than that; the drop-in functions raise the same error with the same message.  (The batched engine clamps instead: it
    serves objects with different template counts in one batch and marks missing slots with template id -1.)"""
    num_templates = int(object_repre.template_descs.shape[0])
    if top_n_templates > num_templates:
        raise RuntimeError(f"selected index k out of range (top_n_templates = {top_n_templates}, the object has {num_templates} templates)")


def find_nearest_object_features(query_features: torch.Tensor, knn_index: knn_util.KNN) -> Tuple[torch.Tensor, torch.Tensor]:
    nn_dists, nn_ids = knn_index.search(query_features)
    return nn_ids, torch.sqrt(nn_dists)  # faiss-style squared distances -> L2


def calc_tfidf(feature_word_ids: torch.Tensor, feature_word_dists: torch.Tensor, word_idfs: torch.Tensor,
               soft_assignment: bool = True, soft_sigma_squared: float = 100.0) -> torch.Tensor:
    """tf-idf vector of ONE point set; `feature_word_dists` are used as given (no sqrt)."""
    ids = feature_word_ids.to("cuda", torch.int32).contiguous()
    dists = feature_word_dists.to("cuda", torch.float32).contiguous()
    seg = torch.tensor([0, ids.shape[0]], dtype=torch.int32, device="cuda")
    desc, _ = ops.tfidf_build(ids, dists, seg, word_idfs.to("cuda", torch.float32).contiguous(),
                              soft_assignment, soft_sigma_squared, sqrt_dists=False)
    return desc[0].to(feature_word_ids.device)


def calc_tfidf_descriptors(
    feat_vectors: torch.Tensor,
    feat_to_word_ids: torch.Tensor,
    feat_to_template_ids: torch.Tensor,
    feat_words: torch.Tensor,
    num_templates: int,
    tfidf_knn_k: int,
    tfidf_soft_assign: bool,
    tfidf_soft_sigma_squared: float,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Bank side of the descriptors (reference: template_util.py:74-123), all templates in one pass on the device:
    -> (template_descs [T, W], word_idfs [W]).  Features must be grouped by template (ascending ids), as gen_repre
    produces them."""
    from . import bank_builder
    opts = repre_util.TemplateDescOpts(tfidf_knn_k=tfidf_knn_k, tfidf_soft_assign=tfidf_soft_assign,
                                       tfidf_soft_sigma_squared=tfidf_soft_sigma_squared)
    descs, idfs, _ = bank_builder.calc_tfidf_descriptors(feat_vectors.cuda(), feat_to_template_ids.cuda(), feat_words.cuda(),
                                                         num_templates, opts, feat_to_word_ids=feat_to_word_ids)
    return descs, idfs


def tfidf_matching(query_features: torch.Tensor, object_repre: repre_util.FeatureBasedObjectRepre, top_n_templates: int,
                   visual_words_knn_index: Optional[knn_util.KNN] = None, debug: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    if object_repre.template_desc_opts is None or object_repre.template_desc_opts.desc_type != "tfidf":
        raise ValueError("Template descriptors need to be tfidf.")
    from .matching import match_batch  # local import: matching imports this module's bank helper

    check_top_n(top_n_templates, object_repre)
    bank = get_device_bank(object_repre)
    qf = query_features.to("cuda", torch.float32)
    pts = torch.zeros(qf.shape[0], 2, dtype=torch.float32, device="cuda")
    res = match_batch(bank, qf, pts, [qf.shape[0]], None, top_n_templates, 1, tie_order="torch",
                      word_metric=getattr(visual_words_knn_index, "metric", None))
    return res.template_ids[0].to(torch.int64).to(query_features.device), res.template_scores[0].to(query_features.device)


def template_matching(query_features: torch.Tensor, object_repre: repre_util.FeatureBasedObjectRepre, top_n_templates: int,
                      matching_type: str, visual_words_knn_index: Optional[knn_util.KNN] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    if matching_type == "tfidf":
        return tfidf_matching(query_features, object_repre, top_n_templates, visual_words_knn_index)
    raise ValueError(f"Unknown matching type '{matching_type}'.")
