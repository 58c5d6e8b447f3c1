"""HBM-resident template bank (the device layout the matching kernels read).

The reference keeps the bank as CPU tensors plus one faiss index per template (built with an O(N_f) mask
scan per template, /root/reference/scripts/infer.py:216-239) and re-scans `feat_to_template_ids` for every
retrieved template (utils/corresp_util.py:110-113). Here the bank of ALL objects lives in HBM once:

  feats      [N_f, d]  f32   patch descriptors, sorted by (object, template)   (repre.feat_vectors)
  feat_sqn   [N_f]     f32   |x|^2 as fma chains (what IndexFlatL2 precomputes)
  vertices   [N_f, 3]  f32
  tpl_off    [T+1]     i32   CSR offsets of each template's run of features (global template ids)
  descs_n    [T, W]    f32   tf-idf descriptors, pre-normalised for cosine similarity
  per object: words [W, d] + |w|^2, idf [W], first template / first feature, tf-idf options, PCA projector
"""

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch

from . import ops
from .repre_util import FeatureBasedObjectRepre, TemplateDescOpts


@dataclass
class ObjectEntry:
    tpl_base: int
    num_templates: int
    feat_base: int
    num_feats: int
    words: torch.Tensor
    words_sqn: torch.Tensor
    idf: torch.Tensor
    opts: TemplateDescOpts
    projectors: list
    _unit: Optional[tuple] = None

    def unit_words(self):
        """Words scaled to unit length + their squared norms (tfidf_knn_metric="cosine": KNN.fit normalises the database,
        knn_util.py:52-57); built on first use -- the shipped options search with "l2"."""
        if self._unit is None:
            wn = ops.normalize_rows(self.words, 0.0)
            self._unit = (wn, ops.sqnorm_rows(wn))
        return self._unit


class DeviceBank:
    def __init__(self, repres: Sequence[FeatureBasedObjectRepre], device: str = "cuda") -> None:
        if len(repres) == 0:
            raise ValueError("empty bank")
        dev = torch.device(device)
        feats, verts, descs, offs = [], [], [], [0]
        self.objects: List[ObjectEntry] = []
        tpl_base = feat_base = 0
        self.p_max = 1
        for r in repres:
            for name in ("vertices", "feat_vectors", "feat_to_template_ids", "feat_cluster_centroids",
                         "feat_cluster_idfs", "template_descs"):
                assert getattr(r, name) is not None, f"object representation lacks `{name}`"
            if r.template_desc_opts is None or r.template_desc_opts.desc_type != "tfidf":
                raise ValueError("Template descriptors need to be tfidf.")
            f2t = r.feat_to_template_ids.to("cpu", torch.int64)
            T = int(r.template_descs.shape[0])
            if f2t.numel() and bool((f2t[1:] < f2t[:-1]).any()):
                raise ValueError("feat_to_template_ids must be sorted (contiguous runs per template, gen_repre.py:187-214)")
            counts = torch.bincount(f2t, minlength=T)
            if counts.numel() != T:
                raise ValueError("feat_to_template_ids refers to templates beyond template_descs")
            self.p_max = max(self.p_max, int(counts.max()))
            run = torch.cumsum(counts, 0) + feat_base
            offs.extend(run.tolist())
            fv = r.feat_vectors.to(dev, torch.float32).contiguous()
            feats.append(fv)
            verts.append(r.vertices.to(dev, torch.float32).contiguous())
            descs.append(r.template_descs.to(dev, torch.float32).contiguous())
            words = r.feat_cluster_centroids.to(dev, torch.float32).contiguous()
            self.objects.append(ObjectEntry(
                tpl_base=tpl_base, num_templates=T, feat_base=feat_base, num_feats=int(fv.shape[0]), words=words,
                words_sqn=ops.sqnorm_rows(words), idf=r.feat_cluster_idfs.to(dev, torch.float32).contiguous(),
                opts=r.template_desc_opts, projectors=list(r.feat_raw_projectors)))
            tpl_base += T
            feat_base += int(fv.shape[0])
        self.device = dev
        self.feats = torch.cat(feats, 0) if len(feats) > 1 else feats[0]
        self.vertices = torch.cat(verts, 0) if len(verts) > 1 else verts[0]
        self.feat_sqn = ops.sqnorm_rows(self.feats)
        self.feat_dim = int(self.feats.shape[1])
        self.num_words = int(self.objects[0].words.shape[0])
        if any(o.words.shape[0] != self.num_words for o in self.objects):
            raise ValueError("all objects of one bank must use the same number of visual words")
        d_all = torch.cat(descs, 0) if len(descs) > 1 else descs[0]
        self.descs_n = ops.normalize_rows(d_all, 1e-8)  # cosine_similarity's per-operand normalisation, once
        self._descs_f16: Optional[torch.Tensor] = None   # built on the first retrieval that takes the two-stage path (descs_f16())
        self.tpl_off = torch.tensor(offs, dtype=torch.int32, device=dev)
        self.obj_tpl_off = torch.tensor([o.tpl_base for o in self.objects] + [tpl_base], dtype=torch.int32, device=dev)
        self.num_templates_total = tpl_base
        self.max_templates = max(o.num_templates for o in self.objects)

    def descs_f16(self) -> torch.Tensor:
        """fp16 image (round to nearest even) of `descs_n` for the candidate pass of the two-stage retrieval
        (fp_cosine_topk_prefiltered): candidates come from it, scores never do.  +50 % descriptor memory, so it is built lazily --
        `prefilter_applies` is false for typical banks (e.g. 800 templates per object)."""
        if self._descs_f16 is None:
            self._descs_f16 = self.descs_n.to(torch.float16).contiguous()
        return self._descs_f16

    def prefilter_applies(self, max_det_per_obj: int, tie_mode: int) -> bool:
        """Mirrors launch_cosine_topk_prefiltered's own gate (csrc/match.hip): the two-stage form pays off once the single-pass kernel
        would stream more than ~250 MB of fp32 bank; word counts the fp16 pass (and, in the torch order, the exact fallback) handles."""
        w = self.num_words
        w_ok = w % 1024 == 0 and (w <= 2048 if tie_mode == 1 else w <= 4096)
        return w_ok and self.max_templates * ((max_det_per_obj + 31) // 32) >= 30000

    @property
    def num_objects(self) -> int:
        return len(self.objects)

    def hbm_bytes(self) -> int:
        n = self.feats.numel() + self.vertices.numel() + self.feat_sqn.numel() + self.descs_n.numel()
        return 4 * n
