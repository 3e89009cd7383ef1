"""Kernel-level GPU unit tests: the device-side fp64 MFMA workgroup GEMM against numpy (exact on integer
data, so a wrong fragment map cannot hide behind a tolerance)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import ttn_amd
    ttn_amd.ensure_init(0)
    return ttn_amd


def _gemm(T, A, B, C0, alpha, beta, ta, tb):
    m, k = (A.shape[1], A.shape[0]) if ta else A.shape
    n = B.shape[0] if tb else B.shape[1]
    A = np.ascontiguousarray(A, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64)
    Cc = np.ascontiguousarray(C0, dtype=np.float64).copy()
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    T._lib.check(T._lib.lib().ttn_selftest_gemm(m, n, k, p(A), p(B), p(Cc), alpha, beta, int(ta), int(tb)))
    return Cc


@pytest.mark.parametrize("m,n,k", [(16, 16, 4), (32, 32, 16), (128, 128, 16), (128, 384, 192), (1, 1, 1), (17, 33, 5),
                                   (130, 70, 37), (64, 200, 129), (3, 300, 2), (64, 384, 128), (40, 520, 800), (192, 192, 1000)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_wg_gemm_exact_on_integers(T, m, n, k, ta, tb):
    rng = np.random.default_rng(m * 1000 + n * 10 + k)
    A = rng.integers(-8, 9, size=(m, k)).astype(np.float64)          # asymmetric integer data: results are exact
    B = rng.integers(-8, 9, size=(k, n)).astype(np.float64)
    C0 = rng.integers(-8, 9, size=(m, n)).astype(np.float64)
    ref = 2.0 * (A @ B) - 3.0 * C0
    got = _gemm(T, A.T if ta else A, B.T if tb else B, C0, 2.0, -3.0, ta, tb)
    assert np.array_equal(got, ref)
    got0 = _gemm(T, A.T if ta else A, B.T if tb else B, np.full((m, n), np.nan), 1.0, 0.0, ta, tb)   # beta = 0 must not read C
    assert np.array_equal(got0, A @ B)


def test_wg_gemm_random_fp64(T):
    rng = np.random.default_rng(5)
    A = rng.standard_normal((128, 192))
    B = rng.standard_normal((192, 384))
    got = _gemm(T, A, B, np.zeros((128, 384)), 1.0, 0.0, 0, 0)
    ref = A @ B
    assert np.max(np.abs(got - ref)) <= 1e-13 * np.max(np.abs(ref)) * 10
