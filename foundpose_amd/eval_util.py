"""Result collection and the output formats of the driver (/root/reference/utils/eval_util.py:231-355 -- the part that
runs without ground truth -- and /root/reference/scripts/prepare_bop_submission.py:30-99): `estimated-poses.json` per
object and the BOP19 csv built from them."""

import json
import os
from collections import defaultdict
from typing import Any, Dict, List, Sequence

import numpy as np


def _jsonable(x):
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    if isinstance(x, dict):
        return {k: _jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    return x


class PoseEvaluator:
    """The bookkeeping half of the reference's PoseEvaluator: pose estimates in the ORIGINAL camera frame, the
    many-to-many aware inlier ratio as the score, per-stage run times, CNOS detection times."""

    def __init__(self) -> None:
        self.result_ids: List[tuple] = []
        self.R: List[np.ndarray] = []
        self.t: List[np.ndarray] = []
        self.score: List[float] = []
        self.time: List[Dict[str, float]] = []
        self.inliers_est_err: List[Dict[str, float]] = []
        self.detection_times: Dict[tuple, float] = {}
        self.mssd: List[float] = []  # stays empty: errors against ground truth are evaluation, outside this path

    def update_without_anno(self, scene_id: int, im_id: int, inst_id: int, hypothesis_id: int, object_repre_vertices: np.ndarray, obj_lid: int,
                            R_m2w: np.ndarray, t_m2w: np.ndarray, orig_camera_c2w, camera_c2w, time_per_inst: Dict[str, float],
                            corresp: Dict[str, np.ndarray], inlier_radius: float = 10) -> Dict[str, Any]:
        T_m2w = np.eye(4)
        T_m2w[:3, :3], T_m2w[:3, 3] = np.asarray(R_m2w, np.float64), np.asarray(t_m2w, np.float64).reshape(3)
        T_m2c = np.linalg.inv(camera_c2w.T_world_from_eye) @ T_m2w
        T_m2oc = np.linalg.inv(orig_camera_c2w.T_world_from_eye) @ T_m2w
        v = np.asarray(object_repre_vertices, np.float64)[np.asarray(corresp["nn_vertex_ids"], np.int64)]
        vc = v @ T_m2c[:3, :3].T + T_m2c[:3, 3]
        proj = np.stack([camera_c2w.f[0] * vc[:, 0] / vc[:, 2] + camera_c2w.c[0], camera_c2w.f[1] * vc[:, 1] / vc[:, 2] + camera_c2w.c[1]], 1)
        corr_dist_est = np.linalg.norm(np.asarray(corresp["coord_2d"], np.float64) - proj, axis=1)
        inliers_est = np.where(corr_dist_est <= inlier_radius)[0]
        ids = np.asarray(corresp["coord_2d_ids"], np.int64)
        unique_2d_ids = list(dict.fromkeys(ids.tolist()))
        est_err = np.zeros(len(unique_2d_ids), dtype=float)
        for i, q in enumerate(unique_2d_ids):  # a query patch counts once, however many of its matches are inliers
            if np.sum(corr_dist_est[ids == q] <= inlier_radius) > 0:
                est_err[i] = 1
        inliers_est_err = {str(int(inlier_radius)): float(np.mean(est_err)) if len(est_err) else 0.0}
        self.R.append(T_m2oc[:3, :3])
        self.t.append(T_m2oc[:3, 3:])
        self.time.append(dict(time_per_inst))
        self.score.append(inliers_est_err[str(int(inlier_radius))])
        self.result_ids.append((scene_id, im_id, obj_lid, inst_id, hypothesis_id))
        self.inliers_est_err.append(inliers_est_err)
        return {"inliers_est": inliers_est, "inliers_est_err": inliers_est_err, "corr_dist_est": corr_dist_est}

    def save_results_json(self, path: str) -> None:
        """estimated-poses.json: ids and score as strings, R [3][3], t [3][1], the run-time dict, the detector's time."""
        out = []
        for i, (scene_id, img_id, obj_id, inst_id, hypothesis_id) in enumerate(self.result_ids):
            out.append({"scene_id": str(scene_id), "img_id": str(img_id), "obj_id": str(obj_id), "inst_id": str(inst_id),
                        "hypothesis_id": str(hypothesis_id), "score": str(self.score[i]), "R": _jsonable(self.R[i]), "t": _jsonable(self.t[i]),
                        "time": _jsonable(self.time[i]), "cnos_time": self.detection_times[(scene_id, img_id)]})
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=2)


def prepare_bop_submission(output_dir: str, object_dataset: str, object_lids: Sequence[int]) -> str:
    """<output_dir>/<lid>/estimated-poses.json of every object -> coarse_<dataset>-estimated-poses.csv (BOP19 format).
    `time` of a row = run time of the whole image: the instances' stage times summed over all objects + the detector's."""
    per_image, det_time = defaultdict(float), {}
    loaded = {}
    for lid in object_lids:
        with open(os.path.join(output_dir, str(lid), "estimated-poses.json")) as f:
            loaded[lid] = json.load(f)
        for e in loaded[lid]:
            key = (e["scene_id"], e["img_id"])
            det_time[key] = e["cnos_time"]
            per_image[key] += sum(e["time"].values())
    lines = ["scene_id,im_id,obj_id,score,R,t,time"]
    for lid in object_lids:
        for e in loaded[lid]:
            key = (e["scene_id"], e["img_id"])
            lines.append("{},{},{},{},{},{},{}".format(
                e["scene_id"], e["img_id"], e["obj_id"], e["score"], " ".join(map(str, np.array(e["R"]).flatten().tolist())),
                " ".join(map(str, np.array(e["t"]).flatten().tolist())), per_image[key] + det_time[key]))
    path = os.path.join(output_dir, f"coarse_{object_dataset}-estimated-poses.csv")
    with open(path, "wb") as f:
        f.write("\n".join(lines).encode("utf-8"))
    return path
