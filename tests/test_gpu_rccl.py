"""The real exchange step of the path on RCCL: engine.gather_records over the `nccl` backend (= RCCL on ROCm), one process
per GPU.  Needs >= 2 visible devices -- skipped on the single-GPU test boxes (there the same code path is covered with
gloo through host memory, tests/test_gpu_bench_multirank.py); it runs as soon as a multi-GPU node is visible."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from foundpose_amd import engine

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    per_rank, width = 4, 5 * (3 + 300 * engine.RECORD_FLOATS_PER_CORRESP)   # the record of one detection: 54 KB of fp32
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    local = torch.rand(per_rank, width, generator=g, device=dev)
    local[:, 0] = torch.arange(per_rank, device=dev) + rank * per_rank
    out = engine.gather_records(local, world)
    assert out.is_cuda and out.shape == (world * per_rank, width)
    assert torch.equal(out[rank * per_rank:(rank + 1) * per_rank], local)
    ret[rank] = (out[:, 0].cpu().tolist(), float(out.double().sum()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL refuses two ranks on one device)")
def test_gather_records_over_rccl():
    world = min(8, torch.cuda.device_count())
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret[r][0] == [float(i) for i in range(world * 4)] for r in range(world))   # every rank holds every record, in rank order
    assert len({ret[r][1] for r in range(world)}) == 1


def _worker_single(rank, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    width = 5 * (3 + 300 * engine.RECORD_FLOATS_PER_CORRESP)
    local = torch.rand(4, width, device=dev)
    out = torch.empty(4, width, device=dev)
    dist.all_gather_into_tensor(out, local)                     # the collective gather_records issues for N > 1
    t = torch.tensor([3.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                    # bench.py's max-over-ranks clock
    dist.barrier()
    torch.cuda.synchronize()
    ret[0] = (bool(torch.equal(out, local)), float(t.item()))
    dist.destroy_process_group()


def test_rccl_backend_initialises_and_runs_the_collectives_on_one_gpu():
    """What a 1-GPU box can check of the RCCL leg: the `nccl` backend (= RCCL) initialises against the device and the three collectives
    the N > 1 path uses (all_gather_into_tensor of the records, all_reduce of the clocks, barrier) run on device tensors.  The
    multi-rank semantics are covered by the gloo tests; the >= 2 GPU test above runs where the hardware exists."""
    ret = mp.Manager().dict()
    mp.spawn(_worker_single, args=(_free_port(), ret), nprocs=1, join=True)
    assert ret[0] == (True, 3.5)
