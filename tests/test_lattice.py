"""The closed-form lattice walk (nerfacc_amd/csrc/lattice.hpp, used by the traversal kernels)
must reproduce the sequential fp32 chain t += dt of the reference (grid.cu:157-161,199-216)
bit for bit.  Host-compiled check over random and adversarial cases.  CPU only."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "lattice_check.c")
LIB = os.path.join(HERE, "native", "liblattice_check.so")


@pytest.fixture(scope="module")
def lib():
    hdr = os.path.join(HERE, "..", "nerfacc_amd", "csrc", "lattice.hpp")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        subprocess.check_call(["gcc", "-O2", "-x", "c", "-std=gnu11", "-ffp-contract=off", "-fPIC", "-shared", "-fvisibility=hidden",
                               SRC, "-o", LIB])
    L = ctypes.CDLL(LIB)
    for f in (L.check_advance, L.check_until):
        f.restype = ctypes.c_int64
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _cases(rng, n):
    # step sizes around the NeRF regime plus exact powers of two / half-way patterns (ties)
    d = np.exp(rng.uniform(np.log(1e-4), np.log(0.3), n)).astype(np.float32)
    sel = rng.random(n)
    d[sel < 0.15] = np.float32(2.0) ** rng.integers(-12, -1, (sel < 0.15).sum())          # power of two
    tie = (sel >= 0.15) & (sel < 0.35)
    bits = d[tie].view(np.uint32)
    k = rng.integers(1, 12, tie.sum()).astype(np.uint32)
    bits = (bits >> k << k) | (np.uint32(1) << (k - 1))                                    # ...1000 tail => ties in higher binades
    d[tie] = bits.view(np.float32)
    t = (rng.random(n) * rng.choice([0.0, 1e-3, 1.0, 8.0, 100.0], n)).astype(np.float32)
    t[rng.random(n) < 0.2] = 0.0
    return t, d


def test_advance_matches_sequential(lib):
    rng = np.random.default_rng(0)
    n = 400000
    t, d = _cases(rng, n)
    j = rng.integers(0, 3000, n).astype(np.int64)
    first = ctypes.c_int64(-1)
    bad = lib.check_advance(ctypes.c_int64(n), _p(t), _p(d), _p(j), ctypes.byref(first))
    assert bad == 0, (bad, first.value, t[first.value], d[first.value], j[first.value])


def test_until_matches_sequential(lib):
    rng = np.random.default_rng(1)
    n = 400000
    t, d = _cases(rng, n)
    target = (t + rng.random(n).astype(np.float32) * rng.choice([0.0, 0.01, 1.0, 6.0, 30.0], n)).astype(np.float32)
    target[rng.random(n) < 0.05] -= 1.0                       # already past the target
    first = ctypes.c_int64(-1)
    bad = lib.check_until(ctypes.c_int64(n), _p(t), _p(d), _p(target), ctypes.byref(first))
    assert bad == 0, (bad, first.value, t[first.value], d[first.value], target[first.value])


def test_degenerate_inputs(lib):
    # stuck walks (d below half an ulp of t), huge / tiny / non-finite values: same answer, no hang
    t = np.array([1e8, 16777216.0, 1.0, 0.0, 3.0, np.inf, 2.0, 1.0], np.float32)
    d = np.array([1.0, 1.0, 1e-9, 1e-30, np.inf, 1.0, np.nan, 5e-3], np.float32)
    j = np.full(8, 100, np.int64)
    first = ctypes.c_int64(-1)
    assert lib.check_advance(ctypes.c_int64(8), _p(t), _p(d), _p(j), ctypes.byref(first)) == 0
    target = np.array([2e8, 16777300.0, 2.0, 1e-28, 9.0, 9.0, 9.0, np.nan], np.float32)
    assert lib.check_until(ctypes.c_int64(8), _p(t), _p(d), _p(target), ctypes.byref(first)) == 0, first.value
