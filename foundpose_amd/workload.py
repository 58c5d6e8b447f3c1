"""The planted synthetic workload of bench.py and of the end-to-end parity tests ("planted
positives: each query crop's features = a known template's patch features + noise, so retrieval has a verifiable
answer and decision margins"), plus the index-agreement statistics both report.

With random-init ViT weights, iid noise crops and a random bank nothing has a margin: every patch feature is equidistant
from every other (measured: nearest other patch at 30.5, mean pairwise 34.3 -- tools/feature_stats.py), so the three
nearest of 2048 visual words, the tf-idf histograms built on them and the top-5 template list are decided by rounding,
and no two arithmetic modes can be compared.  Two things give the workload the structure real data has:
  * crops are assembled from a dictionary of 682 patch textures (synthetic.make_dictionary_crops; distinct inside the
    mask): a texture's feature is nearly the same wherever it appears, so features cluster, and the 2048 visual words are
    three instances of every texture -- the tfidf_knn_k = 3 nearest words of a patch are its own texture's, with a margin;
  * the bank is made FROM the crops: the projected fp32 features of detection b are written, with graded noise and as
    nested, shrinking subsets, into five consecutive templates t_b .. t_b+4 of its object, their vertices come from a
    known pose (R_b, t_b) through a smooth depth surface; every other template is a random set of textures.
The expected answer of the pipeline is then known -- templates t_b .. t_b+4 in that order, correspondences that satisfy the
planted pose -- and the margins are set by the construction, not by chance.

Everything here is data generation and bookkeeping around the product path (it runs the extractor it is handed and the
device bank builder); nothing imports oracle/.
"""

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import bank_builder, feature_util, ops, projector_util, repre_util, synthetic

PLANT_NOISE = (0.05, 0.15, 0.25, 0.35, 0.45)  # feature noise of templates t_b + r, in units of the feature std
PLANT_PATCHES = (0.87, 0.81, 0.75, 0.69, 0.63)  # their patch counts as fractions of the detection's query patches (nested subsets)
WORDS_PER_TEXTURE = 3   # = tfidf_knn_k of the shipped options: a patch's k nearest words are the instances of its texture
# The HARD variant (bench.py `parity.hard`): only the best view is planted -- template t_b holds the detection's features at the noise / patch
# fraction below -- and the other four retrieved templates are whatever the tf-idf retrieval finds among the UNRELATED ones (random texture
# sets: "wrong views", as the templates 2..5 of a real detection mostly are).  Nothing behind slot 1 has an engineered margin: a query patch
# whose texture a wrong view does not contain has no counterpart there, its nearest neighbour is decided among unrelated features, and the
# top-k cut falls among cycle distances of such patches -- the regime in which reduced-precision features move indices.
HARD_NOISE = (0.35,)
HARD_PATCHES = (0.69,)


@dataclass
class PlantedWorkload:
    crops: torch.Tensor            # [B, 3, S, S] f32 in [0, 1], cuda
    masks: torch.Tensor            # [B, S, S] u8, cuda
    det_obj: List[int]             # object index per detection, ascending
    repres: List[repre_util.FeatureBasedObjectRepre]
    targets: torch.Tensor          # [B] planted template id (object-local) of each detection; t .. t+4 are its graded copies
    K: torch.Tensor                # [3, 3] crop camera intrinsics (f64)
    R: torch.Tensor                # [B, 3, 3] planted model->camera rotations (f64)
    t: torch.Tensor                # [B, 3] planted translations (f64)
    planted_rows: Optional[List[Tuple[int, int, int, torch.Tensor]]] = None   # (object, first feature row (object-local), detection, query points [P, 2]) of every planted template


def _random_rotations(n: int, g: torch.Generator) -> torch.Tensor:
    q = torch.randn(n, 4, generator=g, dtype=torch.float64)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).reshape(n, 3, 3)


def planted_vertices(points: torch.Tensor, K: torch.Tensor, R: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """Model-space 3D point seen at each pixel: depth from a smooth surface, X_model = R^T (z K^-1 [u, v, 1] - t)."""
    p = points.to(torch.float64).cpu()
    u, v = p[:, 0], p[:, 1]
    S = 2.0 * float(K[0, 2])
    z = float(t[2]) + 40.0 * torch.sin(u * (2.0 * math.pi / S)) * torch.cos(v * (2.0 * math.pi / S)) + 15.0 * torch.cos(u * (5.0 / S) + v * (3.0 / S))
    xc = torch.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], 1)
    return ((xc - t[None, :]) @ R).to(torch.float32)  # rows: R^T (xc - t)


def query_features(extractor, crops: torch.Tensor, masks: torch.Tensor, cell: float = 14.0):
    """Raw (un-projected) patch features of every detection inside its mask, through the product kernels.
    -> (features [sumQ, D], points [sumQ, 2], counts per detection)."""
    B, _, H, W = crops.shape
    grid = feature_util.generate_grid_points((W, H), cell).to(crops.device)
    pts, img, counts = [], [], []
    for b in range(B):
        p = feature_util.filter_points_by_mask(grid, masks[b])
        pts.append(p)
        img.append(torch.full((p.shape[0],), b, dtype=torch.int32, device=p.device))
        counts.append(int(p.shape[0]))
    pts, img = torch.cat(pts).contiguous(), torch.cat(img).contiguous()
    feats = []
    for b0 in range(0, B, 32):  # the extractor's workspace is per batch shape: keep it at <= 32 crops
        b1 = min(B, b0 + 32)
        fmap, _ = extractor.forward_tokens(crops[b0:b1])
        gh, gw = extractor.num_patches
        sel = (img >= b0) & (img < b1)
        feats.append(ops.sample_bilinear(fmap.reshape(b1 - b0, gh, gw, fmap.shape[-1]).permute(0, 3, 1, 2), pts[sel].contiguous(),
                                         (img[sel] - b0).contiguous(), (W, H)))
    return torch.cat(feats), pts, counts


def build_planted_workload(extractor, batch: int, size: int, num_objects: int, templates_per_object: int, feat_dim: int = 256,
                           num_words: int = 2048, seed: int = 0, crop_seed: int = 0, noise: Sequence[float] = PLANT_NOISE,
                           patch_frac: Sequence[float] = PLANT_PATCHES, mask: Optional[torch.Tensor] = None,
                           words_per_texture: int = WORDS_PER_TEXTURE, hard: bool = False) -> PlantedWorkload:
    """`extractor`: the extractor whose features are planted (use precision="fp32": the reference's arithmetic).
    hard=True: the margin-free variant (HARD_NOISE / HARD_PATCHES above) -- same crops, masks, words, projector and poses as the planted
    workload of the same seeds, only template t_b planted; `targets` then names the expected BEST template only."""
    if hard:
        noise, patch_frac = HARD_NOISE, HARD_PATCHES
    if len(noise) != len(patch_frac):
        raise ValueError("noise and patch_frac describe the same planted templates")
    dev = torch.device("cuda", torch.cuda.current_device())
    m = synthetic.make_disc_mask(size) if mask is None else mask
    n_tex = num_words // words_per_texture   # (a mask with more patches than num_words / 3 needs fewer instance words per texture)
    crops_h, tex = synthetic.make_dictionary_crops(batch, size, m, n_tex, extractor.patch_size, seed=crop_seed)
    crops = crops_h.to(dev)
    masks = m.unsqueeze(0).repeat(batch, 1, 1).to(dev)
    det_obj = sorted(i % num_objects for i in range(batch))
    g = torch.Generator().manual_seed(seed)
    gd = torch.Generator(device=dev).manual_seed(seed)
    raw, pts, counts = query_features(extractor, crops, masks)
    D = raw.shape[1]
    # PCA stand-in: random orthonormal rows (the projector the engine applies to the query side)
    comps = torch.linalg.qr(torch.randn(D, feat_dim, generator=g))[0].T.contiguous()
    proj = projector_util.projector_from_tensordict({"pca_projector": {
        "components": comps, "mean": torch.randn(D, generator=g) * 0.1, "whiten": torch.tensor(False)}})
    qf = proj.transform(raw)                       # [sumQ, feat_dim] on the device
    sigma = float(qf.std())
    q_off = [0]
    for c in counts:
        q_off.append(q_off[-1] + c)
    q_min = min(counts)
    if int(patch_frac[-1] * q_min) < 1:
        raise ValueError(f"a detection has only {q_min} query patches")
    # texture of every query patch (points are cell centres) and one instance of every texture seen: the visual words
    ps = extractor.patch_size
    cell = (pts[:, 1] / ps).long() * tex.shape[2] + (pts[:, 0] / ps).long()
    det_of = torch.repeat_interleave(torch.arange(batch, device=dev), torch.tensor(counts, device=dev))
    q_tex = tex.reshape(batch, -1).to(dev)[det_of, cell]                      # [sumQ]
    order = torch.argsort(q_tex, stable=True)
    st = q_tex[order]
    start = torch.ones_like(st, dtype=torch.bool)
    start[1:] = st[1:] != st[:-1]
    grp_first = torch.cummax(torch.where(start, torch.arange(st.shape[0], device=dev), torch.zeros_like(st)), 0).values
    rank = torch.arange(st.shape[0], device=dev) - grp_first                  # instance number of a row within its texture
    seen_rows = order[start]                                                  # one instance of each texture present
    # words: WORDS_PER_TEXTURE instances of every texture (from different crops); a texture seen fewer times gets jittered
    # copies of its first instance, one never seen gets filler words far from every feature
    words = 4.0 * sigma * torch.randn(num_words, feat_dim, generator=gd, device=dev)
    first_of = torch.full((n_tex,), -1, dtype=torch.int64, device=dev)
    first_of[st[start]] = seen_rows
    have = first_of >= 0
    for r in range(words_per_texture):
        rows_r = order[rank == r]
        w_r = words[r * n_tex:(r + 1) * n_tex]
        w_r[have] = qf[first_of[have]] + 0.02 * sigma * torch.randn(int(have.sum()), feat_dim, generator=gd, device=dev)
        w_r[q_tex[rows_r]] = qf[rows_r]
    words = words.contiguous()
    real_words = torch.cat([words[r * n_tex:(r + 1) * n_tex][have] for r in range(words_per_texture)])  # without the filler

    K = torch.tensor([[1.2 * size, 0.0, size / 2.0], [0.0, 1.2 * size, size / 2.0], [0.0, 0.0, 1.0]], dtype=torch.float64)
    R = _random_rotations(batch, g)
    t = torch.stack([torch.randn(batch, generator=g, dtype=torch.float64) * 20.0, torch.randn(batch, generator=g, dtype=torch.float64) * 20.0,
                     900.0 + 200.0 * torch.rand(batch, generator=g, dtype=torch.float64)], 1)

    T = templates_per_object
    n_obj_det = [det_obj.count(o) for o in range(num_objects)]
    if any(T < 8 * (n + 1) for n in n_obj_det):  # noqa
        raise ValueError("too few templates per object to plant five per detection")
    lo, hi = max(1, int(0.58 * q_min)), max(1, int(0.87 * q_min))   # 300..450 patches per template at the 518 px disc mask
    targets = torch.zeros(batch, dtype=torch.int64)
    repres = []
    planted_rows = []
    b_first = 0
    for o in range(num_objects):
        n_det = n_obj_det[o]
        pcounts = torch.randint(lo, hi + 1, (T,), generator=g)
        stride = T // (n_det + 1)
        for j in range(n_det):  # patch counts of the planted templates: graded fractions of the detection's patches
            for r, fr in enumerate(patch_frac):
                pcounts[4 + j * stride + r] = max(1, int(fr * counts[b_first + j]))
        off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(pcounts, 0)])
        n_f = int(off[-1])
        # every other template: a random set of textures, each as a noisy copy of one of its words (so every real word is
        # some feature's nearest word and its idf = log(T / #templates) stays finite, template_util.py:94-102)
        src = torch.randint(0, real_words.shape[0], (n_f,), generator=gd, device=dev)
        feats = real_words[src] + 0.3 * sigma * torch.randn(n_f, feat_dim, generator=gd, device=dev)
        verts = torch.randn(n_f, 3, generator=gd, device=dev) * 50.0
        for j in range(n_det):
            b = b_first + j
            t_b = 4 + j * stride
            targets[b] = t_b
            qb, pb = qf[q_off[b]:q_off[b + 1]], pts[q_off[b]:q_off[b + 1]]
            perm = torch.randperm(counts[b], generator=g)
            for r, nz in enumerate(noise):
                tpl = t_b + r
                P = int(pcounts[tpl])
                sub = perm[:P].sort().values.to(dev)   # nested: template r+1 holds a subset of template r's patches
                rows = slice(int(off[tpl]), int(off[tpl]) + P)
                feats[rows] = qb[sub] + nz * sigma * torch.randn(P, feat_dim, generator=gd, device=dev)
                verts[rows] = planted_vertices(pb[sub], K, R[b], t[b]).to(dev)
                planted_rows.append((o, int(off[tpl]), b, pb[sub].clone()))
        f2t = torch.repeat_interleave(torch.arange(T, dtype=torch.int32), pcounts).to(dev)
        opts = repre_util.TemplateDescOpts()
        descs, idfs, f2c = bank_builder.calc_tfidf_descriptors(feats, f2t, words, T, opts)
        repres.append(repre_util.FeatureBasedObjectRepre(
            vertices=verts, feat_vectors=feats, feat_to_template_ids=f2t, feat_to_cluster_ids=f2c,
            feat_to_vertex_ids=torch.arange(n_f, dtype=torch.int32, device=dev), feat_cluster_centroids=words,
            feat_cluster_idfs=idfs, template_descs=descs, template_desc_opts=opts, feat_raw_projectors=[proj]))
        b_first += n_det
    return PlantedWorkload(crops, masks, det_obj, repres, targets, K, R, t, planted_rows)


# ---------------------------------------------------------------------------------------------------- pose-level agreement under noise
def noisy_vertices(wl: PlantedWorkload, sigma_px: float, seed: int = 0) -> torch.Tensor:
    """The bank's vertices [N_f, 3] (objects concatenated, DeviceBank order) with every PLANTED 3D point re-derived from its query pixel displaced by
    N(0, sigma_px^2) per axis: the 2D-3D pairs of the planted templates then carry sigma_px of reprojection noise, as on real data, and two different
    inlier sets give two different poses (on the noise-free workload any subset of a planted template's correspondences yields the planted pose)."""
    g = torch.Generator().manual_seed(seed)
    verts = [r.vertices.clone() for r in wl.repres]
    for obj, row0, b, pts in wl.planted_rows:
        p = pts.cpu().to(torch.float64) + sigma_px * torch.randn(pts.shape[0], 2, generator=g, dtype=torch.float64)
        verts[obj][row0:row0 + pts.shape[0]] = planted_vertices(p, wl.K, wl.R[b], wl.t[b]).to(verts[obj].device)
    return torch.cat(verts, 0)


def with_vertices(res, bank, vertices: torch.Tensor, det_obj: Sequence[int]):
    """`res` (a MatchResult of `bank`) with coord_3d re-read from another vertex table through its feature ids (which do not depend on the vertices)."""
    import dataclasses
    base = torch.tensor([bank.objects[o].feat_base for o in det_obj], dtype=torch.int64, device=res.feat_ids.device)
    rows = (res.feat_ids.to(torch.int64) + base[:, None, None]).clamp_(0, vertices.shape[0] - 1)    # (entries past counts[b, j] are padding: any row)
    return dataclasses.replace(res, coord_3d=vertices.to(res.feat_ids.device)[rows].contiguous(), extractor=None, sat_delta=None)


def pose_agreement(best: Dict[str, torch.Tensor], ref: Dict[str, torch.Tensor]) -> Dict:
    """Best coarse poses (pnp_util.select_best_coarse) of two runs over the same detections: north_star's tolerance is 1e-4 relative on R, t."""
    both = (best["found"] & ref["found"]).cpu()
    dR = (best["R"] - ref["R"]).abs().amax(dim=(1, 2)).cpu()
    dt = ((best["t"] - ref["t"]).norm(dim=1) / ref["t"].norm(dim=1)).cpu()
    ok = both & (dR < 1e-4) & (dt < 1e-4)
    same = both & (dR == 0) & (dt == 0)
    return {"detections": int(both.numel()), "found_both": int(both.sum()), "max_abs_dR": float(dR[both].max()) if bool(both.any()) else None,
            "max_rel_dt": float(dt[both].max()) if bool(both.any()) else None, "within_1e-4": int(ok.sum()), "identical": int(same.sum())}


# ---------------------------------------------------------------------------------------------------- agreement statistics
def _np(x):
    import numpy as np
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def parity_stats(got: List[List[Dict]], ref: List[List[Dict]]) -> Dict:
    """Index agreement of two runs of the path over the same detections (each a list of the reference's per-template
    dicts, corresp_util.py:142-163).  Counts are over detections / (detection, template slot) pairs:
      templates_equal ........ the top-n template id lists are identical, in order
      top1_equal ............. the best template is the same
      corresp_equal .......... slots (of detections with identical template lists) whose coord_2d_ids AND nn_vertex_ids
                               are identical index for index, in order
      corresp_overlap ........ mean Jaccard overlap of the (query patch, object feature) pair sets of those slots"""
    import numpy as np
    n = len(got)
    tpl_eq = top1 = slots = slots_eq = 0
    overlap = []
    for a, b in zip(got, ref):
        ia, ib = [int(x["template_id"]) for x in a], [int(x["template_id"]) for x in b]
        top1 += int(len(ia) > 0 and len(ib) > 0 and ia[0] == ib[0])
        if ia != ib:
            continue
        tpl_eq += 1
        for x, y in zip(a, b):
            qa, qb = _np(x["coord_2d_ids"]).astype(np.int64), _np(y["coord_2d_ids"]).astype(np.int64)
            va, vb = _np(x["nn_vertex_ids"]).astype(np.int64), _np(y["nn_vertex_ids"]).astype(np.int64)
            slots += 1
            slots_eq += int(np.array_equal(qa, qb) and np.array_equal(va, vb))
            sa, sb = set(zip(qa.tolist(), va.tolist())), set(zip(qb.tolist(), vb.tolist()))
            overlap.append(len(sa & sb) / max(1, len(sa | sb)))
    return {"detections": n, "templates_equal": tpl_eq, "top1_equal": top1, "slots_compared": slots, "corresp_equal": slots_eq,
            "corresp_overlap": round(float(np.mean(overlap)), 4) if overlap else None}


def stage_flips(got: List[List[Dict]], ref: List[List[Dict]], got_words=None, ref_words=None) -> Dict:
    """Where two runs of the path part ways, attributed to the stage that decided it (each run: per detection the reference's
    per-template dicts; *_words: per detection the [Q, k] nearest visual words, optional):
      word_rows_differ ....... query patches whose k nearest words differ (visual-word k-NN, template_util.py:13-29)
      template_lists_differ .. detections whose top-n template lists differ (tf-idf retrieval, template_util.py:167-174)
      and over the (detection, slot) pairs of detections with identical template lists:
      nn_flips ............... query patches selected by BOTH runs that map to different object features: a flipped nearest-neighbour
                               argmin (corresp_util.py:46); slots_with_nn_flip counts the slots that have one
      slots_selection_differs  slots whose selected query-patch SETS differ: some cycle distance changed (a flipped argmin in either
                               direction) and moved a patch across the top-k cut (corresp_util.py:49-61)
      slots_order_only ....... slots with the same (patch, feature) pairs in a different ORDER: torch.topk's introselect is a
                               function of the whole distance array, so one changed distance elsewhere reorders ties."""
    import numpy as np
    out = {"detections": len(got), "word_rows_differ": None, "word_rows": None, "template_lists_differ": 0, "slots_compared": 0, "nn_flips": 0,
           "slots_with_nn_flip": 0, "slots_selection_differs": 0, "slots_order_only": 0, "slots_identical": 0}
    if got_words is not None and ref_words is not None:
        rows = diff = 0
        for a, b in zip(got_words, ref_words):
            a, b = _np(a).astype(np.int64), _np(b).astype(np.int64)
            rows += a.shape[0]
            diff += int((a != b).any(axis=1).sum())
        out["word_rows"], out["word_rows_differ"] = rows, diff
    for a, b in zip(got, ref):
        if [int(x["template_id"]) for x in a] != [int(x["template_id"]) for x in b]:
            out["template_lists_differ"] += 1
            continue
        for x, y in zip(a, b):
            qa, qb = _np(x["coord_2d_ids"]).astype(np.int64), _np(y["coord_2d_ids"]).astype(np.int64)
            va, vb = _np(x["nn_vertex_ids"]).astype(np.int64), _np(y["nn_vertex_ids"]).astype(np.int64)
            out["slots_compared"] += 1
            if np.array_equal(qa, qb) and np.array_equal(va, vb):
                out["slots_identical"] += 1
                continue
            ma, mb = dict(zip(qa.tolist(), va.tolist())), dict(zip(qb.tolist(), vb.tolist()))
            flips = sum(1 for q, v in ma.items() if q in mb and mb[q] != v)
            out["nn_flips"] += flips
            out["slots_with_nn_flip"] += int(flips > 0)
            if set(ma) != set(mb):
                out["slots_selection_differs"] += 1
            elif flips == 0:
                out["slots_order_only"] += 1
    return out


def planted_stats(got: List[List[Dict]], targets: Sequence[int], n_planted: int = 5) -> Dict:
    """Against the planted answer: detections whose top-n list is exactly t_b .. t_b+n-1, and whose best is t_b."""
    exact = top1 = 0
    for a, t in zip(got, targets):
        ids = [int(x["template_id"]) for x in a]
        top1 += int(len(ids) > 0 and ids[0] == int(t))
        exact += int(ids == [int(t) + r for r in range(n_planted)])
    return {"detections": len(got), "planted_top1": top1, "planted_top5_in_order": exact}
