"""The reference loop's shape -- ONE detection at a time through the drop-in calls (scripts/infer.py:468-542) -- for profiling:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b1 -- python tools/latency_b1.py [n] [precision]
prints host-synchronous ms per stage; the kernel table of the run shows where a B = 1 forward spends its time."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundpose_amd import corresp_util, feature_util, projector_util, workload  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
name = "dinov2_version=vitl14-reg_stride=14_facet=token_layer=18_norm=1"
ex32 = feature_util.make_feature_extractor(name, random_init_seed=1234, precision="fp32").to("cuda")
wl = workload.build_planted_workload(ex32, 8, 518, 1, 800, 256, 2048, seed=7, crop_seed=0)
del ex32
ex = feature_util.make_feature_extractor(name, random_init_seed=1234, precision=prec, use_graph=os.environ.get("FP_B1_GRAPH") == "1").to("cuda")
repre = wl.repres[0]
grid = feature_util.generate_grid_points((518, 518), 14.0).cuda()
keys = ("feat_extract", "grid_sample", "proj", "corresp")
tot = {k: [] for k in keys}


def one(b, sync=True):
    t = {}
    s = torch.cuda.synchronize if sync else (lambda: None)
    s(); t0 = time.perf_counter()
    fmap = ex(wl.crops[b:b + 1])["feature_maps"][0]
    s(); t["feat_extract"] = time.perf_counter() - t0; t0 = time.perf_counter()
    qp = feature_util.filter_points_by_mask(grid, wl.masks[b])
    qf = feature_util.sample_feature_map_at_points(fmap, qp, (518, 518)).contiguous()
    s(); t["grid_sample"] = time.perf_counter() - t0; t0 = time.perf_counter()
    qf = projector_util.project_features(qf, repre.feat_raw_projectors).contiguous()
    s(); t["proj"] = time.perf_counter() - t0; t0 = time.perf_counter()
    c = corresp_util.establish_correspondences(qp, qf, repre, "tfidf", "cyclic_buddies", 5, 300)
    s(); t["corresp"] = time.perf_counter() - t0
    return t


one(0); one(1)
for i in range(n):
    t = one(i % 8)
    for k in keys:
        tot[k].append(t[k])
print({k: round(1e3 * float(np.median(v)), 3) for k, v in tot.items()}, "ms per stage (median), total", round(1e3 * sum(float(np.median(v)) for v in tot.values()), 3))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(n):
    one(i % 8, sync=False)
torch.cuda.synchronize()
print("without per-stage syncs:", round(1e3 * (time.perf_counter() - t0) / n, 3), "ms per detection")
