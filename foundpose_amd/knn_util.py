"""Exact k-NN with the reference's `KNN` interface (/root/reference/utils/knn_util.py:10-112).

The reference wraps faiss IndexFlatL2 / IndexFlatIP and always searches on the CPU. Here `fit` keeps the
database rows (and their squared norms) in HBM and `search` is an exact-fp32 MFMA distance tile + top-k on
the MI355X. Same conventions: L2 distances are SQUARED, ascending; cosine returns 1 - similarity; indices
are int64 on the query's device. Ties are broken by the lowest index.
"""

from typing import Any, Optional, Tuple

import torch

from . import ops
from ._lib import require_cuda


class KNN:
    def __init__(self, k: int = 1, metric: str = "l2", radius: Optional[float] = None, res: Optional[Any] = None) -> None:
        self.index: Any = None
        self.k: int = k
        self.metric: str = metric
        self.radius: Optional[float] = radius
        self.res: Optional[Any] = res
        self._sqn: Optional[torch.Tensor] = None

    def fit(self, data: torch.Tensor) -> None:
        if self.metric not in ("l2", "cosine"):
            raise ValueError(f"Metric {self.metric} is not supported.")
        data = data.to("cuda", torch.float32)
        if self.metric == "cosine":
            data = data / torch.linalg.norm(data, dim=1, keepdim=True)
        self.index = data.contiguous()
        self._sqn = ops.sqnorm_rows(self.index)

    def search(self, data: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.metric not in ("l2", "cosine"):
            raise ValueError(f"Metric {self.metric} is not supported.")
        if self.index is None:
            raise RuntimeError("KNN.search called before KNN.fit")
        if self.radius is not None:
            raise NotImplementedError(
                "range search (radius) is not implemented on the MI355X path.  (No shipped config sets a radius, and the reference's "
                "own branch calls index.range_search_with_radius (knn_util.py:86-89), a method faiss indices do not have -- the "
                "faiss API is range_search(x, thresh) -> (lims, D, I) -- so there is no behaviour to match.)")
        src_device = data.device
        q = data.to("cuda", torch.float32)
        if self.metric == "cosine":
            q = q / torch.linalg.norm(q, dim=1, keepdim=True)
        d2, idx = ops.knn_l2(q.contiguous(), self.index, self.k, None, self._sqn)
        if self.metric == "cosine":
            # unit vectors: |a-b|^2 = 2 - 2cos  ->  cosine distance 1 - cos = d2 / 2 (same ranking as IndexFlatIP)
            d2 = d2 * 0.5
        return d2.to(src_device), idx.to(torch.int64).to(src_device)

    def serialize_index(self) -> None:
        self.index = self.index.cpu()

    def deserialize_index(self) -> None:
        self.index = self.index.cuda()
