"""CPU, world_size 2, gloo: detection sharding + the single gather step (engine.shard_detections /
engine.gather_records).  Records are produced by the oracle here (no GPU in this container); on the GPU the
same two functions wrap FoundPoseEngine.infer_batch (bench.py)."""

import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from foundpose_amd import engine


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _records(det_ids):
    """Deterministic fixed-size record per detection (stands in for pack_result(infer_batch(shard)))."""
    out = torch.zeros(len(det_ids), 16)
    for i, d in enumerate(det_ids):
        g = torch.Generator().manual_seed(1000 + int(d))
        out[i] = torch.rand(16, generator=g)
        out[i, 0] = float(d)
    return out


def _worker(rank, world, port, num_det, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = engine.shard_detections(num_det, world, rank)
    per_rank = (num_det + world - 1) // world
    ids = list(range(lo, hi)) + [-1] * (per_rank - (hi - lo))  # tail shard padded to a fixed size
    local = _records(ids)
    allrec = engine.gather_records(local, world)
    ret[rank] = allrec.numpy()
    dist.barrier()
    dist.destroy_process_group()


def test_shards_cover_everything_once():
    for n in (0, 1, 7, 8, 31, 32, 1000):
        for w in (1, 2, 3, 8):
            spans = [engine.shard_detections(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_gather_world2_gloo():
    world, num_det = 2, 7
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, num_det, ret), nprocs=world, join=True)
    assert np.array_equal(ret[0], ret[1])  # every rank holds the full result
    got = ret[0]
    valid = got[got[:, 0] >= 0]
    assert sorted(valid[:, 0].astype(int).tolist()) == list(range(num_det))
    ref = _records(list(range(num_det))).numpy()
    order = np.argsort(valid[:, 0])
    assert np.array_equal(valid[order], ref)


def test_pack_result_layout():
    from foundpose_amd.matching import MatchResult
    B, n, K = 2, 5, 4
    r = MatchResult(
        template_ids=torch.arange(B * n).reshape(B, n).int(), template_scores=torch.rand(B, n), counts=torch.full((B, n), K).int(),
        q_ids=torch.arange(B * n * K).reshape(B, n, K).int(), feat_ids=torch.arange(B * n * K).reshape(B, n, K).int() * 2,
        dists=torch.rand(B, n, K), conf=torch.rand(B, n, K), coord_2d=torch.rand(B, n, K, 2), coord_3d=torch.rand(B, n, K, 3))
    rec = engine.pack_result(r)
    assert rec.shape == (B, n * (3 + K * engine.RECORD_FLOATS_PER_CORRESP))
    per = rec.reshape(B, n, 3 + K * 9)
    assert torch.equal(per[..., 0], r.template_ids.float()) and torch.equal(per[..., 2], r.counts.float())
    body = per[..., 3:].reshape(B, n, K, 9)
    assert torch.equal(body[..., 0], r.q_ids.float()) and torch.equal(body[..., 6:9], r.coord_3d)
