"""Template-bank container and `repre.pth` I/O with the reference's field names and file layout
(/root/reference/utils/repre_util.py:24-223), so banks written by the reference's gen_repre.py load here
and vice versa. Tensors stay torch tensors; `to_device_bank()` builds the HBM-resident CSR layout the
MI355X kernels read (foundpose_amd/bank.py).
"""

import os
from dataclasses import dataclass, field
from typing import Any, Dict, List, NamedTuple, Optional

import torch

from . import projector_util


class FeatureOpts(NamedTuple):
    extractor_name: str


class TemplateDescOpts(NamedTuple):
    desc_type: str = "tfidf"
    tfidf_knn_metric: str = "l2"
    tfidf_knn_k: int = 3
    tfidf_soft_assign: bool = False
    tfidf_soft_sigma_squared: float = 10.0


@dataclass
class FeatureBasedObjectRepre:
    """Visual object features registered in 3D (same fields as the reference)."""

    vertices: Optional[torch.Tensor] = None              # [N_f, 3]
    vertex_normals: Optional[torch.Tensor] = None
    feat_vectors: Optional[torch.Tensor] = None          # [N_f, d]
    feat_opts: Optional[FeatureOpts] = None
    feat_to_vertex_ids: Optional[torch.Tensor] = None    # [N_f] i32
    feat_to_template_ids: Optional[torch.Tensor] = None  # [N_f] i32, contiguous runs per template
    feat_to_cluster_ids: Optional[torch.Tensor] = None   # [N_f] i32
    feat_cluster_centroids: Optional[torch.Tensor] = None  # [W, d]
    feat_cluster_idfs: Optional[torch.Tensor] = None     # [W]
    feat_raw_projectors: List[Any] = field(default_factory=list)
    feat_vis_projectors: List[Any] = field(default_factory=list)
    templates: Optional[torch.Tensor] = None             # [T, 3, H, W] u8 (visualisation only)
    template_cameras_cam_from_model: List[Any] = field(default_factory=list)
    template_descs: Optional[torch.Tensor] = None        # [T, W]
    template_desc_opts: Optional[TemplateDescOpts] = None
    # not in the reference: the static activation scales of the fp8 extractor the bank was built with ([depth][4] nested
    # list).  Stored as a plain list, which the reference's loader skips (it copies tensors and its own named entries only).
    extractor_fp8_act_scales: Optional[Any] = None


def get_object_repre_dir_path(base_dir: str, repre_type: str, dataset: str, lid: int) -> str:
    return os.path.join(base_dir, dataset, repre_type, str(lid))


def save_object_repre(repre: FeatureBasedObjectRepre, repre_dir: str) -> None:
    obj: Dict[str, Any] = {k: v for k, v in repre.__dict__.items() if v is not None and torch.is_tensor(v)}
    obj["template_cameras_cam_from_model"] = [
        c if isinstance(c, dict) else {
            "f": torch.as_tensor(c.f), "c": torch.as_tensor(c.c), "width": c.width, "height": c.height,
            "T_world_from_eye": torch.as_tensor(c.T_world_from_eye),
        } for c in repre.template_cameras_cam_from_model
    ]
    obj["feat_opts"] = repre.feat_opts._asdict() if repre.feat_opts is not None else None
    obj["template_desc_opts"] = repre.template_desc_opts._asdict() if repre.template_desc_opts is not None else None
    obj["feat_raw_projectors"] = [projector_util.projector_to_tensordict(p) for p in repre.feat_raw_projectors]
    obj["feat_vis_projectors"] = [projector_util.projector_to_tensordict(p) for p in repre.feat_vis_projectors]
    if repre.extractor_fp8_act_scales is not None:
        obj["extractor_fp8_act_scales"] = torch.as_tensor(repre.extractor_fp8_act_scales, dtype=torch.float32).tolist()
    os.makedirs(repre_dir, exist_ok=True)
    torch.save(obj, os.path.join(repre_dir, "repre.pth"))


def load_object_repre(repre_dir: str, tensor_device: str = "cuda", load_fields: Optional[List[str]] = None) -> FeatureBasedObjectRepre:
    """Reads `<repre_dir>/repre.pth`. Like the reference, tensors are returned where torch.load puts them
    (`tensor_device` is accepted and ignored, repre_util.py:202-208); DeviceBank moves them to HBM."""
    obj = torch.load(os.path.join(repre_dir, "repre.pth"), weights_only=False)
    out: Dict[str, Any] = {k: v for k, v in obj.items() if isinstance(v, torch.Tensor)}

    def want(name):
        return load_fields is None or name in load_fields

    if obj.get("feat_opts") is not None and want("feat_opts"):
        out["feat_opts"] = FeatureOpts(**dict(obj["feat_opts"]))
    out["feat_raw_projectors"] = [projector_util.projector_from_tensordict(p) for p in obj.get("feat_raw_projectors", [])] if want("feat_raw_projectors") else []
    out["feat_vis_projectors"] = [projector_util.projector_from_tensordict(p) for p in obj.get("feat_vis_projectors", [])] if want("feat_vis_projectors") else []
    # cameras stay plain dicts (f, c, width, height, T_world_from_eye): the pinhole model class is geometry outside this path
    out["template_cameras_cam_from_model"] = list(obj.get("template_cameras_cam_from_model", [])) if want("template_cameras_cam_from_model") else []
    if want("template_desc_opts") and obj.get("template_desc_opts") is not None:
        out["template_desc_opts"] = TemplateDescOpts(**dict(obj["template_desc_opts"]))
    if obj.get("extractor_fp8_act_scales") is not None:
        out["extractor_fp8_act_scales"] = obj["extractor_fp8_act_scales"]
    return FeatureBasedObjectRepre(**out)


def convert_object_repre_to_numpy(repre: FeatureBasedObjectRepre) -> FeatureBasedObjectRepre:
    out = FeatureBasedObjectRepre()
    for name, value in repre.__dict__.items():
        if isinstance(value, torch.Tensor):
            value = value.detach().cpu().numpy()
        setattr(out, name, value)
    return out
