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
    assert handle.fp_abi_version() == 3


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
