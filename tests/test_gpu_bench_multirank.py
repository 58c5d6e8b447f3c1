"""bench.py under torch.distributed.run with two ranks, dry-run mode (both ranks on cuda:0, records through host memory
over gloo): the launch contract, sharding, the gather step, the max-over-ranks timing and the one JSON line from rank 0.
The real multi-GPU runs use RCCL; nothing else differs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_dry_run():
    env = dict(os.environ, FP_BENCH_ONE_DEVICE="1", FP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--templates", "600", "--batch", "8", "--version", "vits14-reg", "--layer", "9", "--size", "224"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["value"] > 0 and abs(d["value"] - 2 * 8 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 0.02 * d["value"]  # whole-job rate
    assert "cpu_baseline" not in d and "roofline" in d      # the CPU baseline is timed at N = 1 only
    assert d["ranks_seen"] == 2                              # both ranks' records arrived in the gather
    # what makes a real N > 1 scaling run diagnosable: every rank's own step time and the exchange step alone
    mg = d["multi_gpu"]
    assert len(mg["per_rank_ms_per_step"]["all"]) == 2 and 0 < mg["per_rank_ms_per_step"]["min"] <= mg["per_rank_ms_per_step"]["max"]
    assert abs(mg["per_rank_ms_per_step"]["max"] - d["ms_per_step"]) < 1e-2 * d["ms_per_step"] + 1e-3    # `value` is built on the MAX over ranks
    assert mg["gather_ms"]["max_over_ranks"] >= mg["gather_ms"]["mean_over_ranks"] > 0 and mg["gather_ms"]["record_bytes_per_rank"] == 8 * 5 * (3 + 300 * 9) * 4
    assert d["config"]["tie_order"] == "torch" and d["parity"]["planted"]["planted_top1"] == 8


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher (WORLD_SIZE unset): bench.py starts torch.distributed.run itself, rank 0 prints the one JSON line and
    both ranks' records arrive in the gather -- a driver that calls the N > 1 run the way it calls N = 1 still gets a measurement."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(FP_BENCH_ONE_DEVICE="1", FP_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--templates", "600", "--batch", "8",
           "--version", "vits14-reg", "--layer", "9", "--size", "224", "--skip-probes"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 2 and d["value"] > 0


def test_uneven_shards_through_the_real_engine():
    """37 detections over 2 ranks (shards of 19 and 18, cut inside an object group): the real FoundPoseEngine on each shard, then
    pack_result -> pad_records -> gather_records -> unpack_result; every gathered detection equals the single-process result bit for
    bit and the one padding row is dropped by position (tests/multirank_worker.py)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29537", os.path.join(ROOT, "tests", "multirank_worker.py"), "37"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["shard"] == [0, 19] and d["rows"] == 38 and d["padding_rows"] == 1
    assert d["mismatched_fields"] == [], d
    assert d["planted"]["planted_top5_in_order"] == 37
