"""PnP-RANSAC tail on the MI355X (/root/reference/utils/pnp_util.py:20-84, scripts/infer.py:552-602).
cv2 is absent, so parity with its arithmetic is unpinned; what is checked:
  * planted poses: noise-free inliers + gross outliers -> R, t to 1e-6 relative and exactly the planted inlier set;
  * noisy inliers: the returned inlier set recounts identically in numpy from the returned RANSAC model, and the refined pose
    is the minimiser scipy's Levenberg-Marquardt reaches from the same start on the same inliers;
  * degenerate inputs fail loudly-but-safely (fewer than 6 correspondences, coincident points): success False, no crash;
  * the reference's call surface (estimate_pose on one correspondence dict, best-of-n selection);
  * end to end: engine -> correspondences -> poses on the planted workload, fp32 and bf16 extractors: the planted pose of
    every detection to north_star's 1e-4 relative on R, t."""
import numpy as np
import pytest
import torch

from foundpose_amd import pnp_util
from oracle import pnp as opnp

pytestmark = pytest.mark.gpu
CAM = (620.0, 615.0, 259.0, 255.5)


def _scene(rng, n, n_out, noise):
    R = opnp.rodrigues(rng.normal(size=3))
    t = np.array([rng.normal() * 30, rng.normal() * 30, 800 + rng.random() * 400])
    X = (rng.normal(size=(n, 3)) * 60).astype(np.float32)
    uv, _ = opnp.project(R, t, X, CAM)
    uv = uv + rng.normal(size=uv.shape) * noise
    out = rng.choice(n, n_out, replace=False)
    uv[out] = rng.random((n_out, 2)) * 518
    truth = np.ones(n, bool)
    truth[out] = False
    return R, t, X, uv.astype(np.float32), truth


def _batch(scenes, K):
    B = len(scenes)
    c2, c3 = torch.zeros(B, 1, K, 2), torch.zeros(B, 1, K, 3)
    cnt = torch.zeros(B, 1, dtype=torch.int32)
    for b, (_, _, X, uv, _) in enumerate(scenes):
        n = X.shape[0]
        c2[b, 0, :n], c3[b, 0, :n], cnt[b, 0] = torch.from_numpy(uv), torch.from_numpy(X), n
    return c2.cuda(), c3.cuda(), cnt.cuda()


def test_planted_poses_with_outliers():
    rng = np.random.default_rng(0)
    scenes = [_scene(rng, n, int(n * frac), 0.0) for n, frac in [(300, 0.3), (300, 0.6), (40, 0.25), (6, 0.0), (120, 0.5), (300, 0.0)]]
    c2, c3, cnt = _batch(scenes, 300)
    out = pnp_util.solve_pnp_ransac_batch(c2, c3, cnt, [CAM] * len(scenes), 400, 10.0, 0.99, True, seed=1)
    assert bool(out["success"].all())
    for b, (R, t, X, uv, truth) in enumerate(scenes):
        Rg, tg = out["R"][b, 0].cpu().numpy(), out["t"][b, 0].cpu().numpy()
        assert np.abs(Rg - R).max() < 1e-6 and np.abs(tg - t).max() / np.linalg.norm(t) < 1e-6, b
        assert np.abs(Rg @ Rg.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(Rg) - 1) < 1e-12
        mask = out["inliers"][b, 0, :len(truth)].cpu().numpy()
        # a gross outlier lands within 10 px of its true projection with probability ~1e-3; the planted set otherwise
        assert (mask & ~truth).sum() <= 1 and (mask | ~truth).all(), b
        assert int(out["quality"][b, 0]) == int(mask.sum())
    # deterministic in the seed
    again = pnp_util.solve_pnp_ransac_batch(c2, c3, cnt, [CAM] * len(scenes), 400, 10.0, 0.99, True, seed=1)
    assert torch.equal(again["R"], out["R"]) and torch.equal(again["inliers"], out["inliers"])


def test_noisy_inliers_vs_numpy_recount_and_scipy_lm():
    rng = np.random.default_rng(5)
    scenes = [_scene(rng, 300, 90, 1.0) for _ in range(8)]
    c2, c3, cnt = _batch(scenes, 300)
    out = pnp_util.solve_pnp_ransac_batch(c2, c3, cnt, [CAM] * 8, 400, 10.0, 0.99, True, seed=3, return_ransac_pose=True)
    for b, (R, t, X, uv, truth) in enumerate(scenes):
        rp = out["ransac_pose"][b, 0].cpu().numpy()
        mask = out["inliers"][b, 0].cpu().numpy()
        recount = opnp.inlier_mask(rp[:9].reshape(3, 3), rp[9:], X, uv, CAM, 10.0)
        assert np.array_equal(mask, recount), b               # the mask IS the reprojection test of the winning model
        assert (mask & truth).sum() >= 0.95 * truth.sum()     # and it found the planted structure
        Rs, ts, rms = opnp.refine_lm(rp[:9].reshape(3, 3), rp[9:], X[mask], uv[mask], CAM)
        Rg, tg = out["R"][b, 0].cpu().numpy(), out["t"][b, 0].cpu().numpy()
        assert np.abs(Rg - Rs).max() < 1e-7 and np.abs(tg - ts).max() / np.linalg.norm(ts) < 1e-7, b   # same minimiser
        assert np.abs(Rg - R).max() < 5e-3                    # and close to the truth (1 px noise)


def test_degenerate_inputs_fail_safely():
    rng = np.random.default_rng(2)
    R, t, X, uv, _ = _scene(rng, 50, 0, 0.0)
    c2, c3 = torch.zeros(4, 1, 64, 2), torch.zeros(4, 1, 64, 3)
    cnt = torch.tensor([[5], [0], [50], [50]], dtype=torch.int32)     # < 6 correspondences: skipped like infer.py:556
    c2[0, 0, :5], c3[0, 0, :5] = torch.from_numpy(uv[:5]), torch.from_numpy(X[:5])
    c2[2, 0, :50] = torch.from_numpy(uv)                               # all 3D points coincide: no triangle
    c3[2, 0, :50] = torch.from_numpy(X[:1]).repeat(50, 1)
    c2[3, 0, :50], c3[3, 0, :50] = torch.from_numpy(uv), torch.from_numpy(X)
    out = pnp_util.solve_pnp_ransac_batch(c2.cuda(), c3.cuda(), cnt.cuda(), [CAM] * 4, 100, 10.0, 0.99, True)
    assert out["success"].flatten().tolist() == [False, False, False, True]
    assert int(out["inliers"][:3].sum()) == 0 and bool(torch.isfinite(out["R"]).all())


def test_reference_call_surface_and_best_coarse_selection():
    rng = np.random.default_rng(9)
    R, t, X, uv, truth = _scene(rng, 200, 60, 0.5)
    corresp = {"coord_2d": torch.from_numpy(uv), "coord_3d": torch.from_numpy(X)}
    cam = {"f": (CAM[0], CAM[1]), "c": (CAM[2], CAM[3])}
    ok, R_est, t_est, inliers, quality = pnp_util.estimate_pose(corresp, cam, "opencv", 400, 10.0, 0.99, True)
    assert ok and R_est.shape == (3, 3) and t_est.shape == (3, 1) and inliers.ndim == 2 and inliers.shape[1] == 1
    assert quality == float(len(inliers)) and np.abs(R_est - R).max() < 5e-3
    ok2, *_ = pnp_util.estimate_pose({"coord_2d": torch.from_numpy(uv[:3]), "coord_3d": torch.from_numpy(X[:3])}, cam, "opencv", 400, 10.0, 0.99, True)
    assert not ok2
    with pytest.raises(ValueError, match="Unsupported PnP type"):
        pnp_util.estimate_pose(corresp, cam, None, 400, 10.0, 0.99, True)
    poses = {"success": torch.tensor([[True, True, False], [False, False, False], [True, True, True]]),
             "quality": torch.tensor([[10.0, 10.0, 99.0], [5.0, 6.0, 7.0], [3.0, 8.0, 8.0]], dtype=torch.float64),
             "R": torch.arange(3 * 3 * 9, dtype=torch.float64).reshape(3, 3, 3, 3), "t": torch.arange(27, dtype=torch.float64).reshape(3, 3, 3)}
    best = pnp_util.select_best_coarse(poses)
    assert best["found"].tolist() == [True, False, True] and best["corresp_id"].tolist()[0] == 0 and best["corresp_id"].tolist()[2] == 1
    assert torch.equal(best["R"][2], poses["R"][2, 1]) and torch.equal(best["t"][0], poses["t"][0, 0])


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_engine_to_pose_on_planted_workload(precision):
    """crops -> extractor -> matching -> PnP on the planted workload (ViT-S/14-reg, 224 px, so it runs in seconds): the planted
    pose of every detection comes back within 1e-4 relative on R and t -- north_star's pose tolerance -- from the benchmarked
    bf16 mode as from the fp32 mode."""
    from foundpose_amd import engine as fe
    from foundpose_amd import feature_util, workload
    from foundpose_amd.bank import DeviceBank
    name = "dinov2_version=vits14-reg_stride=14_facet=token_layer=9_norm=1"
    ex32 = feature_util.make_feature_extractor(name, random_init_seed=1234, precision="fp32").to("cuda")
    wl = workload.build_planted_workload(ex32, 16, 224, 2, 400, seed=3, crop_seed=1)
    bank = DeviceBank(wl.repres)
    ex = ex32 if precision == "fp32" else feature_util.make_feature_extractor(name, random_init_seed=1234, precision="bf16").to("cuda")
    res = fe.FoundPoseEngine(ex, bank, 14.0, 5, 300, tie_order="torch").infer_batch(wl.crops, wl.masks, wl.det_obj)
    poses = pnp_util.estimate_poses(res, [wl.K.numpy()] * 16, "opencv", 400, 10.0, 0.99, True)
    best = pnp_util.select_best_coarse(poses)
    assert bool(best["found"].all())
    R, t = best["R"].cpu(), best["t"].cpu()
    err_R = (R - wl.R).abs().amax(dim=(1, 2))
    err_t = (t - wl.t).norm(dim=1) / wl.t.norm(dim=1)
    print(f"\n[{precision}] pose vs planted: max |dR| {float(err_R.max()):.2e}, max |dt|/|t| {float(err_t.max()):.2e}, quality {best['quality'].tolist()}")
    assert float(err_R.max()) < 1e-4 and float(err_t.max()) < 1e-4
