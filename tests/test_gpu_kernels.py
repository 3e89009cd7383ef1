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
    m, k = (A.shape[1], A.shape[0]) if (ta & 1) else A.shape
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


@pytest.mark.parametrize("p,q", [(128, 384), (64, 384), (96, 200), (16, 192), (64, 1280), (33, 50), (128, 48), (1, 7), (128, 3000), (130, 64)])
@pytest.mark.parametrize("ta", [0, 1])
@pytest.mark.parametrize("wg512", [0, 1])
def test_wg_syrk_exact_on_integers(T, monkeypatch, p, q, ta, wg512):
    """G = alpha A A^T by the Gram-product routine (one staging per K chunk, lower-triangle tiles, both triangles written),
    both builds; shapes beyond its limits (p > 128, q above the offset table) take the general GEMM."""
    monkeypatch.setenv("TTN_WG512_SELFTEST", str(wg512))
    rng = np.random.default_rng(p * 100 + q)
    A = rng.integers(-8, 9, size=(p, q)).astype(np.float64)
    got = _gemm(T, A.T if ta else A, np.zeros((q, p)), np.full((p, p), np.nan), 0.5, 0.0, 2 | ta, 0)
    assert np.array_equal(got, 0.5 * (A @ A.T))


@pytest.mark.parametrize("m,n,k", [(64, 384, 128), (64, 128, 64), (64, 96, 128), (33, 200, 70), (16, 64, 4), (64, 1088, 128), (48, 130, 127), (64, 2000, 128), (70, 100, 50)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("wg512", [0, 1])
def test_wg_gemm_ra_exact_on_integers(T, monkeypatch, m, n, k, ta, tb, wg512):
    """C = alpha A B by the register-A form (A fragments of all of k in registers, B streamed in column chunks), both builds; shapes
    beyond its limits take the general GEMM."""
    monkeypatch.setenv("TTN_WG512_SELFTEST", str(wg512))
    rng = np.random.default_rng(m * 1000 + n * 10 + k)
    A = rng.integers(-8, 9, size=(m, k)).astype(np.float64)
    B = rng.integers(-8, 9, size=(k, n)).astype(np.float64)
    got = _gemm(T, A.T if ta else A, B.T if tb else B, np.full((m, n), np.nan), -1.5, 0.0, 4 | ta, tb)
    assert np.array_equal(got, -1.5 * (A @ B))


def test_wg_gemm_random_fp64(T):
    rng = np.random.default_rng(5)
    A = rng.standard_normal((128, 192))
    B = rng.standard_normal((192, 384))
    got = _gemm(T, A, B, np.zeros((128, 384)), 1.0, 0.0, 0, 0)
    ref = A @ B
    assert np.max(np.abs(got - ref)) <= 1e-13 * np.max(np.abs(ref)) * 10


# ---- the symmetric eigensolver of the Gram routes (csrc/ttn_eig_kernels.h) ----------------------------------------------------
@pytest.mark.parametrize("n,r,nev,decades,seed", [(128, 64, 64, 1.5, 0), (128, 64, 128, 2.0, 1), (128, 17, 40, 1.0, 2), (128, 64, 64, 0.0, 3),
                                                  (64, 64, 64, 1.5, 4), (64, 20, 64, 2.0, 5), (64, 64, 64, 0.0, 6)])
def test_eig_selftest(n, r, nev, decades, seed):
    """Tridiagonalisation + bisection + twisted factorisations + back-transformation against numpy.linalg.eigh on Gram matrices
    shaped like the workload's (smooth spectra over `decades` decades; decades = 0: a plain random Gram matrix).
    Tolerances: eigenvalues 1e-13 relative to the largest (i.e. singular values to ~1e-13 of the smallest at kappa^2 = 1e4);
    orthonormality and residual of the returned vectors 1e-11 (the route's a-posteriori limit is 2e-11)."""
    import ctypes as C
    import ttn_amd as T
    T.ensure_init(0)
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, 3 * n))
    if decades > 0:
        U, s, Vt = np.linalg.svd(M, full_matrices=False)
        M = (U * (s[0] * 10.0 ** (-decades * np.arange(n) / (n - 1)))) @ Vt
    M /= np.max(np.abs(M))
    G = np.asfortranarray(M @ M.T)
    sig = np.zeros(128)
    X = np.zeros((128, 64), order="F")
    tk = (C.c_int64 * 6)()
    T._lib.check(T._lib.lib().ttn_selftest_eig128(G.ctypes.data_as(C.c_void_p), n, r, nev, sig.ctypes.data_as(C.c_void_p),
                                                  X.ctypes.data_as(C.c_void_p), tk))
    assert tk[1] == 0
    w = np.linalg.eigvalsh(G)[::-1]
    assert np.max(np.abs(sig[:nev] ** 2 - w[:nev])) <= 1e-13 * w[0]
    Ux = X[:n, :r] / sig[:r]
    assert np.max(np.abs(Ux.T @ Ux - np.eye(r))) <= 1e-11
    assert np.max(np.abs(G @ Ux - Ux * w[:r])) <= 1e-11 * w[0]
