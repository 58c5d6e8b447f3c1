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


def _match_result(B, n, K, id_base=0):
    from foundpose_amd.matching import MatchResult
    g = torch.Generator().manual_seed(5 + id_base % 97)
    return MatchResult(
        template_ids=torch.arange(B * n).reshape(B, n).int(), template_scores=torch.rand(B, n, generator=g), counts=torch.full((B, n), K).int(),
        q_ids=torch.arange(B * n * K).reshape(B, n, K).int(), feat_ids=(torch.arange(B * n * K).reshape(B, n, K) * 2 + id_base).int(),
        dists=torch.rand(B, n, K, generator=g), conf=torch.rand(B, n, K, generator=g), coord_2d=torch.rand(B, n, K, 2, generator=g),
        coord_3d=torch.rand(B, n, K, 3, generator=g))


def test_pack_result_layout():
    B, n, K = 2, 5, 4
    r = _match_result(B, n, K)
    rec = engine.pack_result(r)
    assert rec.shape == (B, n * (3 + K * engine.RECORD_FLOATS_PER_CORRESP)) and rec.dtype == torch.float32
    per = rec.reshape(B, n, 3 + K * 9)
    # integer fields travel bit-cast (exact for any int32), float fields as they are
    assert torch.equal(per[..., 0].contiguous().view(torch.int32), r.template_ids) and torch.equal(per[..., 2].contiguous().view(torch.int32), r.counts)
    body = per[..., 3:].reshape(B, n, K, 9)
    assert torch.equal(body[..., 0].contiguous().view(torch.int32), r.q_ids) and torch.equal(body[..., 6:9], r.coord_3d)


def test_pack_unpack_round_trip_large_ids():
    """BASELINE config 5's bank has N_f = 18.7 M > 2^24 features: nn_vertex_ids above 2^24 (and negative 'no template' ids)
    must survive the record exactly -- a float conversion would round them to even."""
    B, n, K = 3, 5, 7
    r = _match_result(B, n, K, id_base=(1 << 24) + 1)   # odd ids above 2^24: not representable in fp32
    r.template_ids[1, 3] = -1
    r.feat_ids[2, 4, 6] = 2 ** 31 - 1
    assert not torch.equal(r.feat_ids.float().to(torch.int32), r.feat_ids)   # the old conversion would have lost them
    u = engine.unpack_result(engine.pack_result(r), n, K)
    for f in ("template_ids", "template_scores", "counts", "q_ids", "feat_ids", "dists", "conf", "coord_2d", "coord_3d"):
        assert torch.equal(getattr(u, f), getattr(r, f)), f
        assert getattr(u, f).dtype == getattr(r, f).dtype


def _worker_large_ids(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    local = engine.pack_result(_match_result(2, 5, 6, id_base=18_707_508 - 200 + rank))   # config 5's N_f, ids around 18.7 M
    u = engine.unpack_result(engine.gather_records(local, world), 5, 6)
    ret[rank] = (u.feat_ids.numpy(), u.q_ids.numpy(), u.template_ids.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world2_gloo_ids_above_2_24_exact():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_large_ids, args=(world, port, ret), nprocs=world, join=True)
    want = torch.cat([_match_result(2, 5, 6, id_base=18_707_508 - 200 + r).feat_ids for r in range(world)]).numpy()
    assert want.max() > (1 << 24)
    for rank in range(world):
        assert np.array_equal(ret[rank][0], want)   # every rank holds every rank's ids, bit for bit


def test_bench_self_launch_builds_the_drivers_launch_line(monkeypatch):
    """`python bench.py --gpus N` with no launcher re-executes itself under torch.distributed.run (bench.self_launch): the command is the
    driver's own launch line on 127.0.0.1 with this process's arguments, and its exit code is handed back."""
    import subprocess
    import sys
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen.update(cmd=cmd, env=env)
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert bench.self_launch(4) == 7
    cmd, env = seen["cmd"], seen["env"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
