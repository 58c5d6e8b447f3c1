"""Candidate statistics of the opt-in two-stage k-NN (csrc/knn_cand.hip): brute-force rows, emitted and re-scored candidates per row,
equality with the CPU oracle.   FP_KNN_CAND=1 python tools/knn_cand_stats.py   (on the GPU box)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundpose_amd import ops
from foundpose_amd._lib import call, ptr, stream, knn_scratch_bytes
from oracle import clib
rng = np.random.default_rng(0)
for (m, n, K, k) in [(70, 300, 64, 3), (1000, 2048, 256, 3), (16544, 2048, 256, 3)]:
    q = torch.from_numpy(rng.standard_normal((m, K)).astype(np.float32)).cuda()
    db = torch.from_numpy(rng.standard_normal((n, K)).astype(np.float32)).cuda()
    qn, dn = ops.sqnorm_rows(q), ops.sqnorm_rows(db)
    scratch = torch.zeros(knn_scratch_bytes(m, n, k), dtype=torch.uint8, device="cuda")
    d2 = torch.empty(m, k, device="cuda"); idx = torch.empty(m, k, dtype=torch.int32, device="cuda")
    call("fp_knn_l2", ptr(q), ptr(qn), m, ptr(db), ptr(dn), n, K, k, ptr(scratch), ptr(d2), ptr(idx), stream())
    torch.cuda.synchronize()
    cap = 10
    raw = scratch[m * 8 * cap * 8: m * 8 * cap * 8 + m * 64].view(torch.int32).reshape(m, 8, 2).cpu()[:, :2 * min(4, max(1, (n + 511) // 512))]
    counts = raw[:, :, 0].numpy()
    thr = raw[:, :, 1].contiguous().view(torch.float32).max(dim=1).values
    ents = scratch[:m * 8 * cap * 8].view(torch.int64).reshape(m, 8, cap).cpu()[:, :raw.shape[1]]
    sc = (ents >> 32).to(torch.int32).view(torch.float32) if False else torch.from_numpy((ents.numpy() >> 32).astype(np.int32).view(np.float32))
    valid = torch.arange(cap)[None, None, :] < torch.from_numpy(counts)[:, :, None]
    kept = ((sc >= thr[:, None, None]) & valid).sum(dim=(1, 2)).float()
    brute = (counts < 0).any(1)
    print(f"m={m} n={n} K={K} k={k}: brute rows {brute.sum()} ({100.0 * brute.mean():.2f} %), candidates per non-brute row: mean {counts[~brute].sum(1).mean():.1f} max {counts[~brute].sum(1).max()}; re-scored after the stage-2 filter: mean {kept.mean():.1f} max {int(kept.max())}")
    if m <= 1000:
        o_d2, o_idx = clib.l2_knn(q.cpu().numpy(), db.cpu().numpy(), k)
        print("   equal:", np.array_equal(idx.cpu().numpy(), o_idx), np.array_equal(d2.cpu().numpy(), o_d2))
