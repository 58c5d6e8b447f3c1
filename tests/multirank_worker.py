"""Worker of tests/test_gpu_bench_multirank.py::test_uneven_shards_through_the_real_engine (launched by torch.distributed.run, one process
per rank, all on cuda:0, records through host memory over gloo): 37 detections over 2 ranks -> shards of 19 and 18 -> the REAL
FoundPoseEngine on each shard -> pack_result -> pad_records -> gather_records -> unpack; rank 0 compares every gathered detection
with a single-process run over all 37, field by field, bit for bit."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    num_det = int(sys.argv[1])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from foundpose_amd import engine as fe
    from foundpose_amd import feature_util, workload
    from foundpose_amd.bank import DeviceBank
    name = "dinov2_version=vits14-reg_stride=14_facet=token_layer=9_norm=1"
    ex32 = feature_util.make_feature_extractor(name, random_init_seed=1234, precision="fp32").to("cuda")
    wl = workload.build_planted_workload(ex32, num_det, 224, 3, 8 * (num_det + 2), seed=3, crop_seed=9)   # three objects (13 + 12 + 12 detections): the shard boundary at 19 falls inside object 1
    bank = DeviceBank(wl.repres)
    ex = feature_util.make_feature_extractor(name, random_init_seed=1234, precision="bf16").to("cuda")
    eng = fe.FoundPoseEngine(ex, bank, 14.0, 5, 300, tie_order="torch")
    lo, hi = fe.shard_detections(num_det, world, rank)
    per = fe.shard_rows(num_det, world)
    res = eng.infer_batch(wl.crops[lo:hi], wl.masks[lo:hi], wl.det_obj[lo:hi])
    rec = fe.pad_records(fe.pack_result(res), per)
    allrec = fe.gather_records(rec, world)
    out = {"rank": rank, "shard": [lo, hi], "rows": int(allrec.shape[0])}
    if rank == 0:
        got = fe.unpack_result(allrec[fe.gathered_valid_index(num_det, world).to(allrec.device)], 5, 300)
        want = eng.infer_batch(wl.crops, wl.masks, wl.det_obj)
        bad = []
        for f in ("template_ids", "template_scores", "counts", "q_ids", "feat_ids", "dists", "conf", "coord_2d", "coord_3d"):
            a, b = getattr(got, f).contiguous(), getattr(want, f).contiguous()
            if not torch.equal(a.view(torch.int32), b.view(torch.int32)):   # bit patterns: NaN confidences compare equal
                bad.append(f)
        pad_ids = allrec.view(torch.int32)[:, 0]
        out.update({"mismatched_fields": bad, "padding_rows": int((pad_ids == -1).sum()),
                    "planted": workload.planted_stats([got.corresp_list(b) for b in range(num_det)], wl.targets.tolist())})
    dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
