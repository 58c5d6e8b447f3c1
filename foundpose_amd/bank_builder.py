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
    feat_to_word_ids: torch.Tensor = None,  # [N_f] word of every feature if already known (k-means assignment)
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (template_descs [T, W], word_idfs [W], feat_to_cluster_ids [N_f] i32)."""
    fv = feat_vectors.float().contiguous()
    words = feat_words.float().contiguous()
    W = words.shape[0]
    f2t = feat_to_template_ids.to(torch.int64)
    words_sqn = ops.sqnorm_rows(words)
    fv_sqn = ops.sqnorm_rows(fv)
    # 1-NN word of every feature (the k-means assignment of cluster_util.py:59)
    if feat_to_word_ids is None:
        _, wid1 = ops.knn_l2(fv, words, 1, fv_sqn, words_sqn)
        feat_to_cluster_ids = wid1[:, 0].contiguous()
    else:
        feat_to_cluster_ids = feat_to_word_ids.to(device=fv.device, dtype=torch.int32).contiguous()
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


def extract_template_features(extractor, templates: torch.Tensor, masks: torch.Tensor, grid_cell: float = 14.0,
                              batch_size: int = 32) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Patch features of the rendered templates inside their masks (scripts/gen_repre.py:136-214 in the reference:
    extractor on each template, grid points filtered by the mask, bilinear samples of the feature map), batched.

    templates [T, 3, S, S] f32 in [0, 1], masks [T, S, S] -> (features [N_f, D] f32, feat_to_template_ids [N_f] i32,
    points [N_f, 2] f32 pixel coordinates), contiguous runs per template in template order."""
    from . import feature_util
    T, _, H, W = templates.shape
    grid = feature_util.generate_grid_points((W, H), grid_cell).cuda()
    feats, ids, pts = [], [], []
    for t0 in range(0, T, batch_size):
        imgs = templates[t0:t0 + batch_size].cuda()
        msk = masks[t0:t0 + batch_size].cuda()
        fmap, _ = extractor.forward_tokens(imgs)
        gh, gw = extractor.num_patches
        D = fmap.shape[-1]
        q_pts, q_img = [], []
        for b in range(imgs.shape[0]):
            p = feature_util.filter_points_by_mask(grid, msk[b])
            q_pts.append(p)
            q_img.append(torch.full((p.shape[0],), b, dtype=torch.int32, device=p.device))
            ids.append(torch.full((p.shape[0],), t0 + b, dtype=torch.int32, device=p.device))
        q_pts, q_img = torch.cat(q_pts).contiguous(), torch.cat(q_img).contiguous()
        feats.append(ops.sample_bilinear(fmap.reshape(imgs.shape[0], gh, gw, D).permute(0, 3, 1, 2), q_pts, q_img, (W, H)))
        pts.append(q_pts)
    return torch.cat(feats), torch.cat(ids), torch.cat(pts)


def register_templates_in_3d(extractor, templates: torch.Tensor, depths: torch.Tensor, masks: torch.Tensor, cameras,
                             Ts_model_from_camera: torch.Tensor, grid_cell: float = 14.0, batch_size: int = 32):
    """Batched feature_util.get_visual_features_registered_in_3d over all rendered templates of an object
    (scripts/gen_repre.py:136-214): masks eroded by 5x5, grid points inside them lifted through the depth images and
    moved to model space, patch features sampled at the same points from batched extractor forwards.

    templates [T,3,S,S] f32, depths [T,S,S] f32, masks [T,S,S], cameras: T pinhole models (f, c), Ts_model_from_camera
    [T,4,4] -> (features [N_f, D], feat_to_template_ids [N_f] i32, vertices [N_f, 3] in model space,
    feat_to_vertex_ids [N_f] i32)."""
    from . import feature_util
    eroded = torch.stack([feature_util.erode_mask(masks[t].cuda()) for t in range(masks.shape[0])])
    feats, f2t, pts = extract_template_features(extractor, templates, eroded, grid_cell, batch_size)
    verts = torch.empty(pts.shape[0], 3, dtype=torch.float32, device=pts.device)
    off = template_offsets(f2t, templates.shape[0]).tolist()
    for t in range(templates.shape[0]):
        a, b = off[t], off[t + 1]
        if b > a:
            v = feature_util.lift_2d_points_to_3d(pts[a:b], depths[t].cuda(), cameras[t])
            verts[a:b] = feature_util.transform_3d_points_torch(Ts_model_from_camera[t].cuda(), v)
    return feats, f2t, verts, torch.arange(pts.shape[0], dtype=torch.int32, device=pts.device)


def build_object_repre(raw_features: torch.Tensor, feat_to_template_ids: torch.Tensor, vertices: torch.Tensor,
                       num_templates: int, pca_components: int = 256, pca_max_samples: int = 100000,
                       cluster_num: int = 2048, cluster_iters: int = 50,
                       desc_opts: TemplateDescOpts = TemplateDescOpts(), verbose: bool = False):
    """PCA -> k-means visual words -> tf-idf template descriptors, all on the MI355X
    (scripts/gen_repre.py:271-345 in the reference), returned as the bank object the inference path loads.

    raw_features [N_f, D] (template order), vertices [N_f, 3] = the 3D point of every feature (their registration
    from rendered depth is upstream of this function)."""
    from . import cluster_util, projector_util, repre_util
    feats = raw_features.float().cuda()
    projectors = []
    if pca_components and pca_components < feats.shape[1]:
        proj = projector_util.PCAProjector(n_components=pca_components)
        proj.fit(feats, max_samples=pca_max_samples)
        feats = proj.transform(feats)
        projectors.append(proj)
    centroids, cluster_ids, _ = cluster_util.kmeans(feats, cluster_num, num_iter=cluster_iters, verbose=verbose)
    f2t = feat_to_template_ids.to(torch.int32).cuda()
    descs, idfs, f2c = calc_tfidf_descriptors(feats, f2t, centroids, num_templates, desc_opts)
    return repre_util.FeatureBasedObjectRepre(
        vertices=vertices.float().cuda(), feat_vectors=feats, feat_to_template_ids=f2t, feat_to_cluster_ids=f2c,
        feat_to_vertex_ids=torch.arange(feats.shape[0], dtype=torch.int32, device=feats.device),
        feat_cluster_centroids=centroids, feat_cluster_idfs=idfs, template_descs=descs, template_desc_opts=desc_opts,
        feat_raw_projectors=projectors)
