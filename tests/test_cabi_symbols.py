"""CPU: the C-ABI library loads and exports every symbol include/foundpose_amd.h declares."""

import os
import re

import pytest


def test_library_exports_every_declared_symbol():
    import torch  # noqa: F401  (loads the HIP runtime the library links against)
    from foundpose_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from foundpose_amd import build
        build.build(verbose=False)
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "foundpose_amd.h")).read()
    declared = set(re.findall(r"\b(fp_[a-z0-9_]+)\s*\(", header)) - {"fp_stream_t"}
    handle = _lib.lib()
    for name in sorted(declared):
        assert hasattr(handle, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.exported_symbols()), "ctypes prototypes out of sync with the header"
    assert handle.fp_abi_version() == _lib.ABI_VERSION == 18


def test_product_never_imports_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "foundpose_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"
                assert "liboracle" not in src


def test_cpu_tensor_is_rejected_loudly():
    import torch
    from foundpose_amd import _lib, ops
    with pytest.raises(_lib.FoundPoseNativeError):
        ops.sqnorm_rows(torch.zeros(4, 8))


def test_bank_builder_entry_points_refuse_cpu_tensors_and_bad_names():
    """Host logic that needs no GPU: loud failures instead of silent CPU work."""
    import torch
    from foundpose_amd import cluster_util, feature_util, knn_util
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cluster_util.kmeans(torch.zeros(64, 8), 4, verbose=False)
    with pytest.raises(NotImplementedError):
        feature_util.make_feature_extractor("resnet50")                      # feature_util.py:18-23 in the reference
    with pytest.raises(AssertionError, match="should divide patch_size"):                    # dinov2_utils.py:378-380 in the reference
        feature_util.make_feature_extractor("dinov2_version=vits14-reg_stride=5_facet=token_layer=9_norm=1")
    assert feature_util.make_feature_extractor("dinov2_version=vits14-reg_stride=7_facet=token_layer=9_norm=1", random_init_seed=0).stride == 7
    with pytest.raises(NotImplementedError):
        feature_util.make_feature_extractor("dinov2_version=vits14-reg_stride=14_facet=attn_layer=9_norm=1")
    with pytest.raises(ValueError):
        knn_util.KNN(k=1, metric="manhattan").fit(torch.zeros(4, 4))         # knn_util.py:61-63 in the reference


def test_infer_driver_refuses_options_it_would_otherwise_ignore():
    """scripts/infer.py options the batched path does not implement are refused before any GPU work: a max_num_queries that could trigger the
    reference's random subsampling (infer.py:482-485), unknown matching / pose types.  (crop=False runs: tests/test_gpu_infer_driver.py.)"""
    from foundpose_amd import infer
    base = dict(version="v", repre_version="r", object_dataset="lmo")
    for bad, exc in ((dict(max_num_queries=500), NotImplementedError),
                     (dict(match_template_type="sift"), ValueError), (dict(match_feat_matching_type="1nn"), ValueError),
                     (dict(final_pose_type="refined"), ValueError)):
        with pytest.raises(exc):
            infer.infer_object(infer.InferOpts(**base, **bad), 1, None, [], {})
    assert infer.load_opts({"infer_opts": dict(base, crop_size=[420, 420])}).max_num_queries == 1000000


def test_oracle_topn_is_clamped_to_the_template_count():
    """torch.topk raises when there are fewer templates than top_n -- and so do the drop-in establish_correspondences / tfidf_matching
    (tests/test_gpu_matching.py::test_random_sweep_vs_oracle_both_tie_orders); the oracle, like the batched match_batch / engine, returns what exists."""
    import numpy as np
    from oracle import clib, match as om
    from oracle.make_golden import build_match_inputs
    c = dict(T=2, pmin=20, pmax=20, W=16, tpl=1, noise=0.05, top_n=5, top_k=10, soft=False, bank_seed=5, q_seed=6, dup=0)
    bank, centroids, pts, feats = build_match_inputs(c)
    r = om.build_synthetic_repre({k: v.numpy() for k, v in bank.items()}, centroids.numpy())
    out = om.establish_correspondences(pts.numpy(), feats.numpy(), r, 5, 10)
    assert [o["template_id"] for o in out][0] == 1 and len(out) == 2
    with pytest.raises(ValueError):
        clib.topk_torch(np.zeros(3, np.float32), 5)


def test_graft_entry_build_runs():
    """The driver's "does it build" hook: compiles what is stale (nothing, normally), loads the library, checks the ABI."""
    import __graft_entry__ as entry
    entry.build()


def test_shipped_library_reads_no_environment_variable():
    """the library keeps no global mutable state and has no hidden switches.  The shipped build does not even IMPORT getenv (the A/B
    switches of measurements are explicit arguments -- fp_vit_model.flags, the variant bits of fp_attention* -- or live in FP_EXPERIMENTS builds,
    fp_build_experiments() == 1), and the Python host mirror reads no FP_* variable either (extractor / engine constructor arguments instead)."""
    import subprocess
    import torch  # noqa: F401
    from foundpose_amd import _lib
    handle = _lib.lib()
    if handle.fp_build_experiments():
        pytest.skip("an FP_EXPERIMENTS build is under test")
    syms = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in syms and "knn_cand" not in syms
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "foundpose_amd")
    for fn in sorted(os.listdir(pkg)):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            for m in re.findall(r"environ(?:\.get)?\W+[\"'](FP_[A-Z0-9_]+)[\"']", src):
                assert fn == "build.py", f"{fn} reads ${m}"    # (the build tool's FP_EXPERIMENTS selects what is compiled; the library never sees it)
