"""K-means on the MI355X: drop-in for /root/reference/utils/cluster_util.py (bank-builder tier).

The reference trains `faiss.Kmeans(d, k, niter=50, seed=0, spherical=False)` (cluster_util.py:38-50) and assigns every
sample to its nearest centroid with the trained index (:59).  faiss is not in the build image and its sampling /
empty-cluster handling are not pinned by any reference test ("parity unpinned"), so this is Lloyd's algorithm with the
published faiss behaviour restated: centroids initialised from a seeded random subset, assignment by exact squared-L2
arg-min (the same exact-fp32 MFMA distance tile as the inference k-NN, ties -> lowest centroid index), centroid = mean of
its samples, an empty cluster re-seeded by splitting the currently largest cluster (faiss: `split_clusters`, centroid
times 1 +- 1/1024 on alternating dims).  Deterministic for a given seed: the centroid sums are sample-ordered fp32 fma
chains (a one-hot GEMM on the exact-fp32 MFMA tile), not atomics.
"""

from typing import Tuple

import torch

from . import ops

EPS_SPLIT = 1.0 / 1024.0


def _assign(samples: torch.Tensor, s_sqn: torch.Tensor, centroids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    d2, ids = ops.knn_l2(samples, centroids, 1, s_sqn, ops.sqnorm_rows(centroids))
    return d2[:, 0], ids[:, 0]


def _update(samples_t: torch.Tensor, ids: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-cluster means with a deterministic summation order: the sums are ONE exact-fp32 MFMA GEMM of the one-hot
    assignment matrix [k, n] with the transposed samples [d, n] -- every (cluster, dim) sum is a sample-ascending fp32
    fma chain -- instead of atomics (order-dependent) or a sort + prefix-sum pass (9x slower at 3e5 x 256)."""
    n = ids.shape[0]
    onehot = torch.zeros(k, samples_t.shape[1], dtype=torch.float32, device=ids.device)  # (columns padded to 4: zeros)
    onehot[ids.to(torch.int64), torch.arange(n, device=ids.device)] = 1.0
    counts = torch.bincount(ids.to(torch.int64), minlength=k)
    sums = ops.gemm_f32(onehot, samples_t)  # [k, d]
    return sums / counts.clamp_min(1).unsqueeze(1).float(), counts


def kmeans(samples: torch.Tensor, num_centroids: int, num_iter: int = 50, verbose: bool = True,
           seed: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (centroids [k, d] f32, cluster_ids [n] i32, centroid_distances [n] f32 squared L2), on the samples' device."""
    x = samples.float().contiguous()
    if not x.is_cuda:
        raise RuntimeError("cluster_util.kmeans runs on the MI355X: pass a CUDA tensor (there is no CPU fallback)")
    n, d = x.shape
    k = int(num_centroids)
    if n < k:
        raise ValueError(f"{n} samples cannot form {k} clusters")
    g = torch.Generator(device="cpu").manual_seed(seed)
    centroids = x[torch.randperm(n, generator=g)[:k].to(x.device)].clone()
    x_sqn = ops.sqnorm_rows(x)
    x_t = torch.zeros(d, (n + 3) // 4 * 4, dtype=torch.float32, device=x.device)  # [d, n (+pad)]: the update GEMM's K
    x_t[:, :n] = x.t()                                                             # dimension runs over the samples
    for it in range(num_iter):
        d2, ids = _assign(x, x_sqn, centroids)
        new_c, counts = _update(x_t, ids, k)
        empty = torch.nonzero(counts == 0).flatten().tolist()
        if empty:  # split the largest clusters, one per empty slot (faiss split_clusters)
            sizes = counts.clone()
            alt = torch.where(torch.arange(d, device=x.device) % 2 == 0, 1.0 + EPS_SPLIT, 1.0 - EPS_SPLIT)
            for e in empty:
                big = int(torch.argmax(sizes))
                new_c[e] = new_c[big] * alt
                new_c[big] = new_c[big] * (2.0 - alt)
                sizes[e] = sizes[big] // 2
                sizes[big] -= sizes[e]
        centroids = new_c
        if verbose and (it == 0 or it + 1 == num_iter):
            print(f"kmeans iter {it + 1}/{num_iter}: objective {float(d2.double().sum()):.6g}, {len(empty)} empty", flush=True)
    d2, ids = _assign(x, x_sqn, centroids)
    return centroids, ids.to(torch.int32), d2
