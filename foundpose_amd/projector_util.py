"""PCA projection of raw ViT features (mirror of /root/reference/utils/projector_util.py).

`transform` = X @ C^T - (mu @ C^T) (sklearn PCA.transform without whitening, projector_util.py:66-69)
as one exact-fp32 MFMA GEMM with the subtraction fused; no host round trip. `fit` is the offline
bank-builder step (sklearn SVD in the reference) and is not part of the inference path.
"""

from typing import Any, Dict, List, Optional

import torch

from . import ops


class Projector:
    def fit(self, data_x: torch.Tensor, data_y: Optional[torch.Tensor] = None, **kwargs: Any) -> None:
        raise NotImplementedError

    def transform(self, data_x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError


class PCAProjector(Projector):
    def __init__(self, n_components: int, whiten: bool = False, **kwargs: Dict[str, Any]) -> None:
        self.n_components = n_components
        self.whiten = whiten
        self.components: Optional[torch.Tensor] = None  # [n_components, D]
        self.mean: Optional[torch.Tensor] = None        # [D]
        self.extra: Dict[str, torch.Tensor] = {}        # explained_variance etc., carried through save/load
        self._dev: Dict[Any, Any] = {}

    def fit(self, data_x: torch.Tensor, data_y: Optional[torch.Tensor] = None, **kwargs: Any) -> None:
        """Fits the PCA on the MI355X (bank-builder tier; reference: projector_util.py:51-64 -> sklearn PCA.fit).

        Same estimator as sklearn's full solver: mean, covariance (X-mu)^T (X-mu) / (n-1) -- the [D, D] Gram matrix is
        one exact-fp32 MFMA GEMM over the samples -- eigen-decomposition (fp64, D x D: host-sized), components = leading
        eigenvectors with sklearn's sign convention (svd_flip on V: the largest-magnitude entry of every component is
        positive).  `max_samples` caps the fitting set with a random permutation like the reference.  sklearn picks a
        RANDOMIZED solver for 256 of 1024 dims, so its own output is seed-dependent; the subspace and variances agree
        to solver tolerance (tests/test_gpu_bank_builder.py checks against the full solver).
        """
        x = data_x
        if "max_samples" in kwargs and x.shape[0] > kwargs["max_samples"]:
            x = x[torch.randperm(x.shape[0])[: kwargs["max_samples"]].to(x.device)]
        x = x.float().cuda().contiguous() if not x.is_cuda else x.float().contiguous()
        n, D = x.shape
        if n < 2 or self.n_components > min(n, D):
            raise ValueError(f"cannot fit {self.n_components} components on {n} samples of {D} dims")
        mean = x.mean(dim=0)
        xc = x - mean
        if n % 4:  # the tile kernel wants K % 4 == 0: all-zero (centred) samples add nothing to the Gram matrix
            xc = torch.cat([xc, torch.zeros(4 - n % 4, D, dtype=xc.dtype, device=xc.device)])
        xc_t = xc.t().contiguous()                              # [D, n]: rows = dims, the GEMM's K runs over samples
        cov = ops.gemm_f32(xc_t, xc_t) / float(n - 1)           # exact-fp32 MFMA chains, k ascending
        evals, evecs = torch.linalg.eigh(cov.double().cpu())    # ascending
        evals = evals.flip(0).clamp_min(0.0)
        comps = evecs.flip(1).t().contiguous()                  # [D, D] rows = components, variance descending
        sign = torch.sign(comps.gather(1, comps.abs().argmax(dim=1, keepdim=True)))
        sign[sign == 0] = 1.0
        comps = comps * sign
        k = self.n_components
        total = float(evals.sum())
        self.components = comps[:k].float()
        self.mean = mean.cpu()
        self.extra = {
            "explained_variance": evals[:k].float(),
            "explained_variance_ratio": (evals[:k] / total).float(),
            "singular_values": torch.sqrt(evals[:k] * (n - 1)).float(),
            "noise_variance": (evals[k:].mean() if k < min(n, D) else torch.tensor(0.0, dtype=torch.float64)).float(),
        }
        self._dev = {}

    def _device_state(self, device):
        key = str(device)
        if key not in self._dev:
            comps = self.components.to(device=device, dtype=torch.float32).contiguous()
            mean = self.mean.to(device=device, dtype=torch.float32).reshape(1, -1).contiguous()
            mean_proj = ops.pca_project(mean, comps, None).reshape(-1).contiguous()  # mu @ C^T, same kernel
            self._dev[key] = (comps, mean_proj)
        return self._dev[key]

    def transform(self, data_x: torch.Tensor) -> torch.Tensor:
        if self.components is None:
            raise RuntimeError("PCAProjector has no components (load it from a tensordict)")
        comps, mean_proj = self._device_state(data_x.device)
        return ops.pca_project(data_x, comps, mean_proj)


def project_features(feat_vectors: torch.Tensor, projectors: List[Projector], batch_size: int = 4096) -> torch.Tensor:
    for projector in projectors:
        feat_vectors = projector.transform(feat_vectors)
    return feat_vectors


def projector_to_tensordict(projector: Projector) -> Dict[str, Any]:
    if isinstance(projector, PCAProjector):
        d = {"components": projector.components, "mean": projector.mean, "whiten": torch.tensor(projector.whiten)}
        d.update(projector.extra)
        return {"pca_projector": d}
    raise ValueError(f"Unknown projector type: {type(projector)}")


def projector_from_tensordict(projector_dict: Dict[str, Any]) -> Projector:
    if "pca_projector" in projector_dict:
        p = projector_dict["pca_projector"]
        comps = torch.as_tensor(p["components"])
        # like the reference, the stored `whiten` flag is not applied at transform time (projector_util.py:128)
        proj = PCAProjector(n_components=comps.shape[0], whiten=bool(torch.as_tensor(p.get("whiten", False))))
        proj.components = comps.float()
        proj.mean = torch.as_tensor(p["mean"]).float()
        proj.extra = {k: v for k, v in p.items() if k not in ("components", "mean", "whiten")}
        return proj
    raise ValueError("Unknown projector type.")
