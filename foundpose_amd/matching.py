"""Batched descriptor matching on the MI355X: visual-word k-NN -> tf-idf -> template retrieval ->
cyclic best buddies -> 2D-3D correspondences, for B detections at once.

This is the batched form of `establish_correspondences` (/root/reference/utils/corresp_util.py:73-169 with
utils/template_util.py:126-176); the reference processes one detection at a time on the CPU through faiss.
Detections must be grouped by object (ascending object index) so the bank is streamed once per object group.
"""

import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch

from . import ops
from ._lib import call, cosine_prefilter_scratch_floats, cosine_scratch_floats, cyclic_scratch_bytes, ptr, stream, require_cuda
from .bank import DeviceBank


@dataclass
class MatchResult:
    """Padded device tensors; slot j of detection b is valid for the first counts[b, j] entries."""

    template_ids: torch.Tensor     # [B, n] i32 object-local template ids (-1 = no template)
    template_scores: torch.Tensor  # [B, n] f32 cosine similarity
    counts: torch.Tensor           # [B, n] i32
    q_ids: torch.Tensor            # [B, n, K] i32  (coord_2d_ids)
    feat_ids: torch.Tensor         # [B, n, K] i32  (nn_vertex_ids)
    dists: torch.Tensor            # [B, n, K] f32  cycle distances in px
    conf: torch.Tensor             # [B, n, K] f32
    coord_2d: torch.Tensor         # [B, n, K, 2] f32
    coord_3d: torch.Tensor         # [B, n, K, 3] f32
    query_tfidf: Optional[torch.Tensor] = None  # [B, W] (debug)
    word_ids: Optional[torch.Tensor] = None     # [sumQ, k]
    extractor: Optional[object] = None          # set by the engine in the f16x3 / fp8 modes: corresp_list() asks it whether an activation was
                                                # clamped on the way (sticky device-side counters, DinoFeatureExtractor.check_saturation)
    sat_delta: Optional[torch.Tensor] = None    # [2] i32 on the device: clamps counted during this batch's backbone (f16x3, fp8)
    _sat_error: Optional[Exception] = None      # the saturation verdict of this result, once read (see corresp_list)
    ready: Optional["torch.cuda.Event"] = None  # set when the matching ran on the engine's side stream (overlap_matching): the tensors are
                                                # complete once this event has fired; wait() makes the current stream wait for it

    def wait(self) -> "MatchResult":
        """Makes the current stream wait for the matching stream, and tells the caching allocator that the result tensors
        (allocated on the matching stream) are now in use on the current one: without record_stream their blocks would go
        back to the matching stream's pool the moment the result is dropped, and the next batch's matching could overwrite
        them under a consumer kernel still reading on this stream."""
        if self.ready is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(self.ready)
            for t in (self.template_ids, self.template_scores, self.counts, self.q_ids, self.feat_ids, self.dists, self.conf, self.coord_2d,
                      self.coord_3d, self.query_tfidf, self.word_ids):
                if t is not None and t.is_cuda:
                    t.record_stream(cur)
        return self

    def corresp_list(self, b: int, debug: bool = False) -> List[Dict]:
        """The reference's List[Dict] for detection b (keys as in corresp_util.py:142-163)."""
        self.wait()
        if self.extractor is not None:   # f16x3: raises FoundPoseSaturationError if the backbone clamped an activation
            # the counters are read once per result; the verdict is kept, so EVERY access of a clamped result raises (any detection index,
            # any number of times), not only the first one
            ex, self.extractor = self.extractor, None
            try:
                if self.sat_delta is not None:       # what THIS batch clamped (engine: counters snapshotted around its backbone)
                    ex.report_saturation(*(int(v) for v in self.sat_delta.tolist()))
                else:
                    torch.cuda.current_stream().synchronize()
                    ex.check_saturation()
            except Exception as e:
                self._sat_error = e
        if self._sat_error is not None:
            raise self._sat_error
        # one host copy for the slot table (counts | template ids) and one widening per index array -- the per-slot form (two .tolist() waits, three
        # casts per slot) was ~60 us of the per-detection loop; the dict entries are views, as the reference's are
        n = self.counts.shape[1]
        tab = torch.cat([self.counts[b], self.template_ids[b]]).tolist()
        counts, tids = tab[:n], tab[n:]
        tid64, q64, f64 = self.template_ids[b].to(torch.int64), self.q_ids[b].to(torch.int64), self.feat_ids[b].to(torch.int64)
        out = []
        for j, (c, tid) in enumerate(zip(counts, tids)):
            if tid < 0:
                continue
            d = {
                "template_id": tid64[j],
                "template_score": self.template_scores[b, j],
                "coord_2d": self.coord_2d[b, j, :c],
                "coord_2d_ids": q64[j, :c],
                "coord_3d": self.coord_3d[b, j, :c],
                "coord_conf": self.conf[b, j, :c],
                "nn_vertex_ids": f64[j, :c],
            }
            if debug:
                d["nn_dists"] = self.dists[b, j, :c]
                d["nn_indices"] = d["nn_vertex_ids"]
            out.append(d)
        return out


def match_batch(
    bank: DeviceBank,
    query_features: torch.Tensor,   # [sumQ, d] f32, detections concatenated
    query_points: torch.Tensor,     # [sumQ, 2] f32
    q_counts: Sequence[int],        # Q_b per detection (host)
    det_obj: Optional[Sequence[int]] = None,  # object index per detection, ascending (host); default all 0
    top_n_templates: int = 5,
    top_k_buddies: int = 300,
    keep_debug: bool = False,
    tie_order: str = "canonical",  # "canonical": (value, lowest index);  "torch": the reference's torch.topk CPU tie order
    word_metric: Optional[str] = None,  # metric of the visual-word search; default: each object's template_desc_opts.tfidf_knn_metric
    prefilter: bool = True,  # the two-stage template retrieval where it applies (same outputs bit for bit; False: always the single-pass exact kernel)
    mark=None,  # optional callable(name): called with "retrieval_begin" / "retrieval_end" around the template retrieval (the engine records HIP events there)
) -> MatchResult:
    require_cuda(query_features, query_points)
    if tie_order not in ("canonical", "torch"):
        raise ValueError(f"unknown tie_order '{tie_order}'")
    tie_mode = 1 if tie_order == "torch" else 0
    dev = query_features.device
    B = len(q_counts)
    det_obj = [0] * B if det_obj is None else list(det_obj)
    if any(det_obj[i] > det_obj[i + 1] for i in range(B - 1)):
        raise ValueError("detections must be grouped by object (ascending object index)")
    if query_features.shape[1] != bank.feat_dim:
        raise ValueError(f"query features have {query_features.shape[1]} dims, bank has {bank.feat_dim}")
    qf = query_features.float().contiguous()
    qp = query_points.float().contiguous()
    q_off_h = [0]
    for c in q_counts:
        q_off_h.append(q_off_h[-1] + int(c))
    sumQ = q_off_h[-1]
    if qf.shape[0] != sumQ:
        raise ValueError("query_features rows do not match sum(q_counts)")
    q_max = max(1, max(q_counts) if B else 1)
    n, K = top_n_templates, top_k_buddies
    groups = []  # (obj, first_det, end_det)
    i = 0
    while i < B:
        j = i
        while j < B and det_obj[j] == det_obj[i]:
            j += 1
        groups.append((det_obj[i], i, j))
        i = j
    det_seg_h = [0] * (bank.num_objects + 1)
    for obj, d0, d1 in groups:
        det_seg_h[obj + 1] = d1 - d0
    for k_ in range(bank.num_objects):
        det_seg_h[k_ + 1] += det_seg_h[k_]
    # ---- every small host table of the batch in ONE upload: q_off [B+1] | det_seg [num_obj+1] | det_nt [B] | tpl_base [B] | feat_base [B]
    tabs_h = q_off_h + det_seg_h + [bank.objects[o].num_templates for o in det_obj] + [bank.objects[o].tpl_base for o in det_obj] \
        + [bank.objects[o].feat_base for o in det_obj]
    # (pinned staging + asynchronous copy: a pageable upload blocks the host until the stream has drained -- the whole backbone of this batch --
    #  and the launches behind it would then reach an idle GPU one launch latency at a time; the caching host allocator keeps the pinned
    #  block alive until the copy has run)
    tabs_host = torch.empty(len(tabs_h), dtype=torch.int32, pin_memory=True)
    tabs_host.copy_(torch.tensor(tabs_h, dtype=torch.int32))
    tabs = tabs_host.to(dev, non_blocking=True)
    o1 = B + 1
    o2 = o1 + bank.num_objects + 1
    q_off, det_seg, det_nt, tpl_base, feat_base = tabs[:o1], tabs[o1:o2], tabs[o2:o2 + B], tabs[o2 + B:o2 + 2 * B], tabs[o2 + 2 * B:o2 + 3 * B]

    q_sqn = ops.sqnorm_rows(qf)

    # ---- per object group: nearest visual words (k-NN, k = tfidf_knn_k) and tf-idf descriptors, written in place
    W = bank.num_words
    desc = torch.empty(B, W, dtype=torch.float32, device=dev)
    desc_n = torch.empty(B, W, dtype=torch.float32, device=dev)
    word_ids_all = []
    for obj, d0, d1 in groups:
        o = bank.objects[obj]
        r0, r1 = q_off_h[d0], q_off_h[d1]
        metric = o.opts.tfidf_knn_metric if word_metric is None else word_metric
        if metric == "l2":
            w_d2, w_ids = ops.knn_l2(qf[r0:r1], o.words, o.opts.tfidf_knn_k, q_sqn[r0:r1], o.words_sqn)
        elif metric == "cosine":
            # KNN(metric="cosine") (knn_util.py:52-57, 91-100): words and queries scaled to unit length (no eps: a zero row is NaN there
            # too), inner-product search, distance 1 - similarity.  On unit vectors |a - b|^2 = 2 - 2 a.b, so the exact-fp32 L2 tile
            # kernel ranks identically (d^2 ascending = similarity descending) and d^2 / 2 is the cosine distance; the tf-idf kernel
            # takes its square root like find_nearest_object_features does (template_util.py:26-27; it only matters with soft assignment).
            words_n, words_n_sqn = o.unit_words()
            qn = ops.normalize_rows(qf[r0:r1], 0.0)
            w_d2, w_ids = ops.knn_l2(qn, words_n, o.opts.tfidf_knn_k, None, words_n_sqn)
            w_d2.mul_(0.5)
        else:
            raise ValueError(f"Metric {metric} is not supported.")   # knn_util.py:62-63
        seg = q_off[d0:d1 + 1] if r0 == 0 else (q_off[d0:d1 + 1] - r0)
        ops.tfidf_build(w_ids, w_d2, seg, o.idf, o.opts.tfidf_soft_assign, o.opts.tfidf_soft_sigma_squared, sqrt_dists=True,
                        out=(desc[d0:d1], desc_n[d0:d1]))
        if keep_debug:
            word_ids_all.append(w_ids)

    # ---- template retrieval: cosine vs the object's template descriptors, top-n
    # detections are sorted by object, so object o's detections are rows det_seg_h[o]:det_seg_h[o+1]
    max_det = max(d1 - d0 for _, d0, d1 in groups) if groups else 1
    t_scores = torch.empty(B, n, dtype=torch.float32, device=dev)
    t_ids = torch.empty(B, n, dtype=torch.int32, device=dev)
    if mark is not None:
        mark("retrieval_begin")
    if prefilter and bank.prefilter_applies(max_det, tie_mode):
        sims = torch.empty(cosine_prefilter_scratch_floats(B, bank.max_templates), dtype=torch.float32, device=dev)
        call("fp_cosine_topk_prefiltered", ptr(desc_n), ptr(det_seg), ptr(det_nt), B, max_det, ptr(bank.descs_n), ptr(bank.descs_f16()),
             ptr(bank.obj_tpl_off), bank.num_objects, bank.max_templates, W, n, ptr(sims), ptr(t_scores), ptr(t_ids), tie_mode, stream())
    else:
        sims = torch.empty(cosine_scratch_floats(B, bank.max_templates), dtype=torch.float32, device=dev)  # finished scores [B, T] + candidate keys
        call("fp_cosine_topk", ptr(desc_n), ptr(det_seg), ptr(det_nt), B, max_det, ptr(bank.descs_n),
             ptr(bank.obj_tpl_off), bank.num_objects, bank.max_templates, W, n, ptr(sims), ptr(t_scores), ptr(t_ids), tie_mode, stream())

    if mark is not None:
        mark("retrieval_end")

    # ---- cyclic best buddies against the retrieved templates + correspondence assembly (the kernel pads its records itself)
    pairs = B * n   # (the kernels add the object's first template to the object-local ids themselves)
    scratch = torch.empty(cyclic_scratch_bytes(pairs, q_max, bank.p_max) // 8, dtype=torch.int64, device=dev)
    counts = torch.empty(B, n, dtype=torch.int32, device=dev)
    q_ids = torch.empty(B, n, K, dtype=torch.int32, device=dev)
    feat_ids = torch.empty(B, n, K, dtype=torch.int32, device=dev)
    dists = torch.empty(B, n, K, dtype=torch.float32, device=dev)
    conf = torch.empty(B, n, K, dtype=torch.float32, device=dev)
    c2d = torch.empty(B, n, K, 2, dtype=torch.float32, device=dev)
    c3d = torch.empty(B, n, K, 3, dtype=torch.float32, device=dev)
    call("fp_cyclic_buddies", ptr(qf), ptr(q_sqn), ptr(qp), ptr(q_off), B, q_max, ptr(bank.feats), ptr(bank.feat_sqn),
         ptr(bank.tpl_off), bank.p_max, ptr(bank.vertices), ptr(t_ids), ptr(tpl_base), ptr(feat_base), n, bank.feat_dim, K, K,
         ptr(scratch), ptr(counts), ptr(q_ids), ptr(feat_ids), ptr(dists), ptr(conf), ptr(c2d), ptr(c3d), tie_mode, stream())
    return MatchResult(t_ids, t_scores, counts, q_ids, feat_ids, dists, conf, c2d, c3d,
                       query_tfidf=desc if keep_debug else None,
                       word_ids=torch.cat(word_ids_all, 0) if keep_debug and word_ids_all else None)
