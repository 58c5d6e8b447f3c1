"""Template-descriptor construction on the MI355X (bank-builder tier).

Device restatement of `calc_tfidf_descriptors` (/root/reference/utils/template_util.py:74-123): word
assignment of every bank feature (k-means 1-NN assignment), idf = log(T / #templates containing the word),
and one tf-idf descriptor per template from the k nearest words of its patches. Reuses the inference
kernels (exact-fp32 k-NN tile, tf-idf histogram) on much larger inputs; used to synthesise the 10k-template
bank of the benchmark and by the parity tests.
"""

from typing import Tuple

import torch

from . import ops
from .repre_util import TemplateDescOpts


def template_offsets(feat_to_template_ids: torch.Tensor, num_templates: int) -> torch.Tensor:
    counts = torch.bincount(feat_to_template_ids.to(torch.int64), minlength=num_templates)
    return torch.cat([torch.zeros(1, dtype=torch.int64, device=counts.device), torch.cumsum(counts, 0)]).to(torch.int32)


def calc_tfidf_descriptors(
    feat_vectors: torch.Tensor,          # [N_f, d] cuda, sorted by template
    feat_to_template_ids: torch.Tensor,  # [N_f] i32 cuda
    feat_words: torch.Tensor,            # [W, d] cuda
    num_templates: int,
    opts: TemplateDescOpts = TemplateDescOpts(),
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (template_descs [T, W], word_idfs [W], feat_to_cluster_ids [N_f] i32)."""
    fv = feat_vectors.float().contiguous()
    words = feat_words.float().contiguous()
    W = words.shape[0]
    f2t = feat_to_template_ids.to(torch.int64)
    words_sqn = ops.sqnorm_rows(words)
    fv_sqn = ops.sqnorm_rows(fv)
    # 1-NN word of every feature (the k-means assignment of cluster_util.py:59)
    _, wid1 = ops.knn_l2(fv, words, 1, fv_sqn, words_sqn)
    feat_to_cluster_ids = wid1[:, 0].contiguous()
    # idf: number of templates in which each word occurs (template_util.py:94-102)
    pairs = torch.unique(f2t * W + feat_to_cluster_ids.to(torch.int64))
    occ = torch.bincount(pairs % W, minlength=W)
    idfs = torch.log(torch.as_tensor(float(num_templates)) .to(fv.device) / occ.to(torch.float32))
    # k nearest words of every feature, then one histogram per template (squared distances on this side)
    d2, wid = ops.knn_l2(fv, words, opts.tfidf_knn_k, fv_sqn, words_sqn)
    seg = template_offsets(feat_to_template_ids, num_templates).to(fv.device)
    descs, _ = ops.tfidf_build(wid, d2, seg, idfs.contiguous(), opts.tfidf_soft_assign, opts.tfidf_soft_sigma_squared,
                               sqrt_dists=False)
    return descs, idfs, feat_to_cluster_ids
