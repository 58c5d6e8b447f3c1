"""Shared test helpers: rebuild fixture inputs, compare correspondence lists."""

import os

import numpy as np
import torch

from oracle.make_golden import MATCH_CASES, TINY, TINY0, build_match_inputs, checksum  # noqa: F401

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def match_case_inputs(name):
    """-> (case dict, golden dict, repre dict (numpy, reference-built descs), pts, feats)."""
    c = MATCH_CASES[name]
    g = load_golden(name)
    bank, centroids, pts, feats = build_match_inputs(c)
    assert np.isclose(checksum(bank["feat_vectors"], centroids, pts, feats), g["input_checksum"], rtol=0, atol=1e-6), \
        "seeded inputs differ from the ones the fixture was generated with"
    repre = {
        "vertices": bank["vertices"].numpy(),
        "feat_vectors": bank["feat_vectors"].numpy(),
        "feat_to_vertex_ids": bank["feat_to_vertex_ids"].numpy(),
        "feat_to_template_ids": bank["feat_to_template_ids"].numpy(),
        "feat_to_cluster_ids": g["feat_to_cluster_ids"],
        "feat_cluster_centroids": centroids.numpy(),
        "feat_cluster_idfs": g["word_idfs"],
        "template_descs": g["template_descs"],
        "template_desc_opts": {"desc_type": "tfidf", "tfidf_knn_metric": "l2", "tfidf_knn_k": 3,
                               "tfidf_soft_assign": bool(c["soft"]), "tfidf_soft_sigma_squared": 10.0},
    }
    return c, g, repre, pts.numpy(), feats.numpy()


NOREG_CASES = (("vitl14", 518), ("vitl14", 420), ("vits14", 420), ("vitb14", 518), ("vitb14", 420), ("vitg14", 224))


def noreg_case(version, S):
    """Inputs of tests/golden/extractor_<version>_<S>.npz (the non-register hub entries, scripts/infer.py:75) -> (g, name, spec, sd, imgs)."""
    from foundpose_amd import synthetic
    from foundpose_amd.vit_config import parse_extractor_name
    g = load_golden(f"extractor_{version}_{S}")
    name = str(g["name"])
    spec = parse_extractor_name(name)
    assert spec.version == version and spec.layer == int(g["layer"]) and spec.arch.registers == 0
    sd = synthetic.make_vit_state_dict(spec.arch, seed=int(g["weights_seed"]))
    imgs = synthetic.make_crops(1, S, seed=int(g["image_seed"]))
    assert np.isclose(checksum(imgs, sd[f"blocks.{spec.layer}.attn.qkv.weight"], sd["pos_embed"]), g["input_checksum"], atol=1e-6)
    return g, name, spec, sd, imgs


# ---------------------------------------------------------------------------------------------------- bars of the lossy modes (bf16, fp8)
# The lossy extractor modes are held to 2.5 x the error MEASURED on the MI355X with the committed kernels (tests/golden/measured_bars.json:
# key -> error in units of the reference's feature scale), not to a generic "bf16-ish" constant: a regression that doubled or tripled the
# error of a mode must fail.  The kernels are deterministic, so the measured value is a property of the code, not of the box.
# Regenerate after a deliberate arithmetic change:  FP_RECORD_BARS=gpurun_out/bars.jsonl python -m pytest tests -m gpu -q ; python tools/update_bars.py
BARS_FILE = os.path.join(GOLDEN, "measured_bars.json")
BAR_FACTOR = 2.5
_BARS = None


def check_bar(key, measured, fallback, floor=0.0):
    """Asserts measured <= max(BAR_FACTOR x the recorded value of `key`, floor) (or <= fallback while a key has no record yet); the message
    carries the measured value.  `floor`: for quantities whose recorded value can be exactly 0 (1 - overlap of index sets).
    FP_RECORD_BARS=<file>: also appends {key, measured} to that file (tools/update_bars.py folds it into measured_bars.json)."""
    import json
    global _BARS
    if _BARS is None:
        _BARS = json.load(open(BARS_FILE)) if os.path.exists(BARS_FILE) else {}
    measured = float(measured)
    rec = os.environ.get("FP_RECORD_BARS")
    if rec:
        with open(rec, "a") as f:
            f.write(json.dumps({"key": key, "measured": measured}) + "\n")
    if key in _BARS:
        bar = max(BAR_FACTOR * float(_BARS[key]), floor)
        assert measured <= bar, f"{key}: measured {measured:.3e} > {BAR_FACTOR} x recorded {float(_BARS[key]):.3e} = {bar:.3e}"
    else:
        assert measured <= fallback, f"{key}: measured {measured:.3e} > fallback bar {fallback:.3e} (no recorded value yet)"
    return measured


def assert_features_close(key, got, ref, scale, tol, lossy):
    """max |got - ref| <= tol x scale for the exact modes (fp32, f16x3); for the lossy ones (bf16, fp8) the measured-bar rule above."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    err = float(np.abs(got - ref).max() / scale)
    if lossy:
        check_bar(key, err, tol)
    else:
        assert err <= tol, f"{key}: {err:.3e} > {tol:.1e} of the feature scale"
    return err


def experiments_build() -> bool:
    """True when the library under test was compiled with -DFP_EXPERIMENTS (FP_EXPERIMENTS=1 python -m foundpose_amd.build --force): the measured-slower kernels
    kept for A/B runs (role-split split-fp16 attention, bf16 attention work splits 2 / 3, the two-stage k-NN) exist in such builds only."""
    try:
        from foundpose_amd import _lib
        return bool(_lib.lib().fp_build_experiments())
    except Exception:
        return False
