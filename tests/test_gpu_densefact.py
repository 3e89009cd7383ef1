"""GPU tests of the dense QR / SVD kernels behind the TDVP sweeps (csrc/ttn_densefact_kernels.h: ttn_dense_qr, ttn_dense_svd) through the
host mirror's helpers (tdvp._qr_j / _svd_j: Julia column-major matrices as reversed-shape device arrays), Float64 and ComplexF64,
tall / wide / square / rank-deficient, against NumPy's LAPACK: factors reproduce the matrix and are orthonormal to 1e-13, R is upper
triangular with the diagonal LAPACK's zlarfg gives (real; |.| equal to NumPy's), singular values to 1e-13 relative to the largest
(src/solvers/tdvp.jl:76-80, :120-126, :252, :276 are the call sites these replace)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import ttn_amd
    ttn_amd.ensure_init(0)
    return ttn_amd


def _rnd(rng, cplx, m, n):
    x = rng.standard_normal((m, n))
    return x + 1j * rng.standard_normal((m, n)) if cplx else x


SHAPES = [(1, 1), (2, 1), (1, 3), (4, 4), (16, 7), (7, 16), (128, 64), (64, 128), (96, 96), (33, 130), (257, 12)]


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("m,n", SHAPES)
def test_dense_qr(T, cplx, m, n):
    D = T.tdvp
    torch, stream = D._dev()
    rng = np.random.default_rng(1000 * m + n + cplx)
    X = _rnd(rng, cplx, m, n)
    if m >= 4 and n >= 4:
        X[:, 2] = X[:, 1]                                                    # a dependent column
    with torch.cuda.stream(stream):
        Qt, Rt = D._qr_j(D._up(X, X.dtype))
        Q, R = D._down(Qt), D._down(Rt)
    r = min(m, n)
    assert Q.shape == (m, r) and R.shape == (r, n)
    sc = max(np.max(np.abs(X)), 1e-300)
    assert np.max(np.abs(Q @ R - X)) <= 1e-13 * sc * max(m, n)
    assert np.max(np.abs(Q.conj().T @ Q - np.eye(r))) <= 1e-13 * max(m, n)
    assert np.max(np.abs(np.tril(R, -1))) == 0.0
    assert np.max(np.abs(np.imag(np.diag(R)))) == 0.0                        # zlarfg: beta is real
    if not (m >= 4 and n >= 4):                                              # (behind a numerically zero pivot the reflectors are not unique)
        Rn = np.linalg.qr(X, mode="reduced")[1]
        assert np.allclose(np.abs(np.diag(R)), np.abs(np.diag(Rn)), rtol=1e-10, atol=1e-12 * sc)
    else:
        Rn = np.linalg.qr(X, mode="reduced")[1]
        assert np.allclose(np.abs(np.diag(R))[:2], np.abs(np.diag(Rn))[:2], rtol=1e-10)


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("m,n", SHAPES)
def test_dense_svd(T, cplx, m, n):
    D = T.tdvp
    torch, stream = D._dev()
    rng = np.random.default_rng(2000 * m + n + cplx)
    X = _rnd(rng, cplx, m, n)
    if min(m, n) >= 8:                                                       # decaying spectrum + an exactly dependent pair
        X = X @ np.diag(0.5 ** np.arange(n))
        X[:, 3] = X[:, 5]
    with torch.cuda.stream(stream):
        Ut, sd, Vtt = D._svd_j(D._up(X, X.dtype))
        U, s, Vt = D._down(Ut), sd.cpu().numpy(), D._down(Vtt)
    k = min(m, n)
    assert U.shape == (m, k) and Vt.shape == (k, n) and s.shape == (k,)
    sn = np.linalg.svd(X, compute_uv=False)
    assert np.all(np.diff(s) <= 0) and np.max(np.abs(s - sn)) <= 1e-13 * sn[0]
    assert np.max(np.abs((U * s[None, :]) @ Vt - X)) <= 1e-13 * sn[0] * max(m, n)
    keep = s > 1e-10 * sn[0]                                                 # the vectors of zero singular values are not defined
    Uk, Vk = U[:, keep], Vt[keep, :]
    assert np.max(np.abs(Uk.conj().T @ Uk - np.eye(Uk.shape[1]))) <= 1e-12 * max(m, n)
    assert np.max(np.abs(Vk @ Vk.conj().T - np.eye(Vk.shape[0]))) <= 1e-12 * max(m, n)


@pytest.mark.parametrize("cplx", [False, True])
def test_dense_svd_rank_one_blocks(T, cplx):
    """The two-site tensors of a TDVP2 sweep over a product-like state are rank-one blocks: after the first rotations the other columns
    carry norms of 1e-17, 1e-33, 1e-160 ... of the first (the case on which (be - al) / 2|ga| overflowed when squared and the pair was
    never orthogonalised).  Singular values, reconstruction, descending order; orthonormality of the vectors that carry weight."""
    D = T.tdvp
    torch, stream = D._dev()
    rng = np.random.default_rng(7 + cplx)
    for m, n in ((4, 4), (16, 16), (32, 8), (8, 32)):
        u, v = _rnd(rng, cplx, m, 1), _rnd(rng, cplx, 1, n)
        X = u @ v
        with torch.cuda.stream(stream):
            Ut, sd, Vtt = D._svd_j(D._up(X, X.dtype))
            U, s, Vt = D._down(Ut), sd.cpu().numpy(), D._down(Vtt)
        sn = np.linalg.svd(X, compute_uv=False)
        assert abs(s[0] - sn[0]) <= 1e-13 * sn[0] and np.all(s[1:] <= 1e-14 * sn[0]) and np.all(np.diff(s) <= 0)
        assert np.max(np.abs((U * s[None, :]) @ Vt - X)) <= 1e-13 * sn[0] * max(m, n)
        assert abs(np.linalg.norm(U[:, 0]) - 1) < 1e-13 and abs(np.linalg.norm(Vt[0, :]) - 1) < 1e-13
