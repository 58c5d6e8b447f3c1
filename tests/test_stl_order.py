"""CPU: the product's restatement of libstdc++'s partial_sort / nth_element / sort (stl_order.hpp, used on the
device for the strict torch.topk tie order) vs the real std:: algorithms and vs torch.topk itself."""

import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import clib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("stl") / "libstl_order_check.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests", "csrc", "stl_order_check.cpp")])
    return ctypes.CDLL(so)


def run(lib, v, k):
    v = np.ascontiguousarray(v, np.float32)
    out = np.empty(k, np.int64)
    lib.stl_order_topk(v.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(len(v)), ctypes.c_int64(k), out.ctypes.data_as(ctypes.c_void_p))
    return out


def test_matches_std_and_torch_on_tied_inputs(lib):
    rng = np.random.default_rng(0)
    for trial in range(1500):
        n = int(rng.integers(1, 2049))
        k = int(rng.integers(1, n + 1))
        kind = trial % 5
        if kind == 0:
            v = rng.standard_normal(n)
        elif kind == 1:
            v = -(rng.integers(0, 4, n) * 14.0)          # heavy ties, like cycle distances
        elif kind == 2:
            v = -np.sqrt((rng.integers(0, 6, n) * 14.0) ** 2 + (rng.integers(0, 6, n) * 14.0) ** 2)
        elif kind == 3:
            v = np.zeros(n)                               # all tied
        else:
            v = -np.sort(rng.integers(0, 50, n)).astype(np.float64)  # sorted input, ties
        v = v.astype(np.float32)
        got = run(lib, v, k)
        ref_std = clib.topk_torch(v, k, True, True)[1]
        assert np.array_equal(got, ref_std), (trial, n, k)
        if trial % 10 == 0:
            ref_torch = torch.topk(torch.from_numpy(v), k, sorted=True)[1].numpy()
            assert np.array_equal(got, ref_torch)


def test_partial_sort_branch_large_n(lib):
    rng = np.random.default_rng(1)
    for n, k in ((10000, 5), (800, 5), (320, 5), (319, 5), (50000, 20)):
        v = np.round(rng.random(n), 2).astype(np.float32)  # ties at 0.01 resolution
        assert np.array_equal(run(lib, v, k), torch.topk(torch.from_numpy(v), k, sorted=True)[1].numpy())
