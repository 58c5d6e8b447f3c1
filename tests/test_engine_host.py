"""Host-side logic of the batched engine (no GPU): the token selection's cell table must cover every bilinear tap the
sampling kernel can touch -- fp_vit_sample_features_selected returns NaN for a tap outside the selection, so the cover
property is what makes the selected path equal to the full one for ANY grid geometry, not only the tested ones."""
import numpy as np
import pytest
import torch

from foundpose_amd.engine import FoundPoseEngine


class _Ext:  # the two attributes _cells9 reads
    def __init__(self, patch_size):
        self.patch_size = patch_size


def _taps(px, py, W, H, gw, gh):
    """The four taps of ln_sample_kernel / sample_bilinear_kernel for a point, in its float32 arithmetic (vit.hip)."""
    f = np.float32
    sx, sy = f(2.0) / f(W), f(2.0) / f(H)
    u1 = (sx * f(px) - f(1.0)) + f(1.0)
    v1 = (sy * f(py) - f(1.0)) + f(1.0)
    ix = f(np.float64(u1) * np.float64(f(gw) / f(2.0)) - 0.5)   # fma: one rounding
    iy = f(np.float64(v1) * np.float64(f(gh) / f(2.0)) - 0.5)
    x0, y0 = int(np.floor(ix)), int(np.floor(iy))
    return [(x, y) for y in (y0, y0 + 1) for x in (x0, x0 + 1) if 0 <= x < gw and 0 <= y < gh]


@pytest.mark.parametrize("size,cell,patch", [(518, 14.0, 14), (224, 14.0, 14), (420, 14.0, 14), (210, 10.0, 14), (126, 7.0, 14), (518, 9.25, 14), (224, 16.0, 16)])
def test_cells9_cover_every_sampling_tap(size, cell, patch):
    eng = FoundPoseEngine(_Ext(patch), None, grid_cell_size=cell)
    W = H = size
    gw = gh = size // patch
    pts = eng._grid(W, H, torch.device("cpu"))[0].numpy()
    cells = eng._cells9(W, H, torch.device("cpu")).view(-1, 9).numpy()
    assert cells.shape[0] == pts.shape[0]
    assert cells.min() >= 0 and cells.max() <= gw * gh          # gw * gh = "outside the map"
    for g in range(pts.shape[0]):
        have = set(int(c) for c in cells[g] if c < gw * gh)
        for (x, y) in _taps(pts[g, 0], pts[g, 1], W, H, gw, gh):
            assert y * gw + x in have, (size, cell, g, pts[g], x, y)


def test_activation_rows_are_padded_to_whole_tiles_of_both_heights():
    """DinoFeatureExtractor.padded_rows: whole 256-row GEMM tiles always; whole 320-row tiles as well (a multiple of 1280) when the bf16
    (folded LayerNorm) or fp8 path can take the taller tile and the padding costs < 3 % more rows."""
    from foundpose_amd import feature_util
    name = "dinov2_version=vits14-reg_stride=14_facet=token_layer=2_norm=1"
    bf = feature_util.make_feature_extractor(name, random_init_seed=1, precision="bf16")
    assert bf.padded_rows(32 * 1374) == 44800 and bf.padded_rows(24 * 1374) == 33280      # the bench batch, and a batch of 24
    assert bf.padded_rows(1374) == 1536 and bf.padded_rows(5 * 261) == 1536                # small batches: 256-row tiles only
    assert bf.padded_rows(44800) == 44800 and bf.padded_rows(44801) == 46080
    for rows in (1, 255, 256, 257, 43968, 100000):
        assert bf.padded_rows(rows) % 256 == 0 and rows <= bf.padded_rows(rows) <= max(256, int(rows * 1.03) + 255)
    f32 = feature_util.make_feature_extractor(name, random_init_seed=1, precision="fp32")
    assert f32.padded_rows(32 * 1374) == 44032                                              # no taller tile in the fp32 mode
    plain = feature_util.make_feature_extractor(name, random_init_seed=1, precision="bf16", fold_layernorm=False)
    assert plain.padded_rows(32 * 1374) == 44032


def test_f16f8_row_packing_host_side():
    """ops.splitx_pack / splitx_unpack (host-side preparation of the f16f8 mode's weights, include/foundpose_amd.h "f16f8 rows"): layout and what a row
    represents, on CPU tensors -- the device producers write the same bytes (tests/test_gpu_f16f8.py)."""
    from foundpose_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(6, 192, generator=g) * torch.logspace(-1, 2, 192)[None, :]
    s = 8.0
    p = ops.splitx_pack(x, s)
    assert p.shape == (6, 384) and p.dtype == torch.float16
    by = p.view(torch.uint8).unflatten(1, (3, 256))
    hi = (x * s).half()
    assert torch.equal(by[:, :, :128].contiguous().view(torch.float16).reshape(6, 192), hi)
    assert torch.equal(by[:, :, 128:192].reshape(6, 192), (hi.float() * 2.0 ** -7).to(torch.float8_e4m3fn).view(torch.uint8))
    assert torch.equal(by[:, :, 192:256].reshape(6, 192), ((x * s - hi.float()) * 16.0).to(torch.float8_e4m3fn).view(torch.uint8))
    back = ops.splitx_unpack(p, s)
    big = x.abs() * s > 1.0
    assert float(((back - x).abs() / x.abs())[big].max()) < 2.0 ** -14
    padded = ops.splitx_pack(x, s, pad=64)
    assert padded.stride(0) == 384 + 64 and torch.equal(padded.contiguous(), p)
    with pytest.raises(ValueError):
        ops.splitx_pack(x[:, :96], s)
    # values beyond the fp16 range saturate in the high half and in its e4m3 copy (the device reports them: saturation counters)
    sat = ops.splitx_pack(torch.full((1, 64), 1.0e6), 1.0)
    assert float(sat[0, 0]) == 65504.0 and int(sat.view(torch.uint8)[0, 128]) == int(torch.tensor(448.0).to(torch.float8_e4m3fn).view(torch.uint8))


def test_pose_agreement_counts_identical_and_within_tolerance():
    """workload.pose_agreement (bench.py parity.hard.pose_under_noise): detections found by both runs, identical poses, poses within north_star's 1e-4 on R AND t."""
    import torch
    from foundpose_amd import workload
    R = torch.eye(3, dtype=torch.float64).repeat(4, 1, 1)
    t = torch.tensor([[0.0, 0.0, 1000.0]] * 4, dtype=torch.float64)
    ref = {"found": torch.tensor([True, True, True, False]), "R": R, "t": t}
    R2, t2 = R.clone(), t.clone()
    R2[1, 0, 1] += 5e-5            # within 1e-4, not identical
    t2[2, 2] += 1.0                # 1e-3 relative: outside
    got = {"found": torch.tensor([True, True, True, True]), "R": R2, "t": t2}
    st = workload.pose_agreement(got, ref)
    assert st["detections"] == 4 and st["found_both"] == 3 and st["identical"] == 1 and st["within_1e-4"] == 2
    assert abs(st["max_abs_dR"] - 5e-5) < 1e-12 and abs(st["max_rel_dt"] - 1e-3) < 1e-9


def test_update_bars_merges_and_keeps_untouched_keys(tmp_path, monkeypatch):
    """tools/update_bars.py folds a (possibly partial) run into measured_bars.json: keys the run did not touch keep their bar, changes are printed old -> new,
    --only-new leaves every recorded bar alone."""
    import json
    import os
    import runpy
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = tmp_path / "repo"
    (fake / "tests" / "golden").mkdir(parents=True)
    (fake / "tools").mkdir()
    (fake / "tools" / "update_bars.py").write_text(open(os.path.join(root, "tools", "update_bars.py")).read())
    bars = fake / "tests" / "golden" / "measured_bars.json"
    bars.write_text(json.dumps({"a": 1.0, "b": 2.0}))
    run = tmp_path / "run.jsonl"
    run.write_text("\n".join(json.dumps(r) for r in [{"key": "b", "measured": 3.0}, {"key": "b", "measured": 2.5}, {"key": "c", "measured": 0.5}]))
    monkeypatch.setattr(sys, "argv", ["update_bars.py", str(run), "--only-new"])
    runpy.run_path(str(fake / "tools" / "update_bars.py"), run_name="__main__")
    assert json.loads(bars.read_text()) == {"a": 1.0, "b": 2.0, "c": 0.5}
    monkeypatch.setattr(sys, "argv", ["update_bars.py", str(run)])
    runpy.run_path(str(fake / "tools" / "update_bars.py"), run_name="__main__")
    assert json.loads(bars.read_text()) == {"a": 1.0, "b": 3.0, "c": 0.5}
