"""GPU parity of mals_linsolve (src/solvers/mals.jl:240-312): HIP path (ttn_mals_linsolve) vs the CPU oracle.
Tolerances: adapted ranks exact where the local spectra have a gap around the sv_trunc threshold (tol = 1e-10 ... 1e-6 on
these inputs; with tol at rounding level the decisions are noise, compared by value only); iterate as a tensor 1e-8 relative
(the two-site solve amplifies rounding by cond(K), the SVD split by the gaps); the reference's own assertions
(test/test_mals.jl:19-77) on its shapes."""
import numpy as np
import pytest

from oracle import tt_oracle as O
from tests.helpers import to_oracle, to_product, tt_norm_stable, tt_rel_diff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import ttn_amd
    ttn_amd.ensure_init(0)
    return ttn_amd


def _spd(d, shift):
    return O.tto_add(O.Delta(d), O.tto_scale(shift, O.id_tto(d)))


def _resid(A, x, b):
    return tt_norm_stable(O.sub(O.apply(A, x), b)) / max(tt_norm_stable(b), np.finfo(float).eps)


def test_mals_reference_cases(T):
    """test/test_mals.jl:19-77 — same shapes / operators / assertions (inputs from NumPy's generator)."""
    rng = np.random.default_rng(5678)
    d = 4
    b, x0 = O.rand_tt((2,) * d, [1, 2, 2, 2, 1], rng), O.rand_tt((2,) * d, [1, 2, 2, 2, 1], rng)
    x = T.solvers.mals_linsolve(to_product(_spd(d, 3.0)), to_product(b), to_product(x0))
    assert x.N == d and tuple(x.ttv_dims) == (2,) * d and all(np.isfinite(x.ttv_rks))
    A = _spd(d, 10.0)
    x = T.solvers.mals_linsolve(to_product(A), to_product(b), to_product(x0), tol=1e-10, rmax=8)
    assert _resid(A, to_oracle(x), b) < 0.5
    b1, x1 = O.rand_tt((2,) * d, [1] * 5, rng), O.rand_tt((2,) * d, [1] * 5, rng)
    x = T.solvers.mals_linsolve(to_product(O.id_tto(d)), to_product(b1), to_product(x1), tol=1e-12, rmax=4)
    assert _resid(O.id_tto(d), to_oracle(x), b1) < 0.05
    x = T.solvers.mals_linsolve(to_product(_spd(d, 5.0)), to_product(b), to_product(x1), tol=1e-10, rmax=4)
    assert max(x.ttv_rks) <= 4
    A = _spd(d, 3.0)
    xl = T.solvers.mals_linsolve(to_product(A), to_product(b), to_product(x0), tol=1e-2, rmax=8)
    xt = T.solvers.mals_linsolve(to_product(A), to_product(b), to_product(x0), tol=0.0, rmax=8)
    assert max(xl.ttv_rks) <= max(xt.ttv_rks) + 2


@pytest.mark.parametrize("d,r0,rb,shift,tol,rmax,seed", [(6, 2, 2, 2.0, 1e-10, 64, 0), (8, 2, 2, 3.0, 1e-8, 8, 1), (8, 3, 2, 0.0, 1e-10, 6, 2),
                                                         (10, 2, 3, 1.0, 1e-6, 8, 3), (5, 1, 2, 2.0, 1e-10, 4, 4), (2, 1, 2, 1.0, 1e-10, 2, 5)])
def test_mals_vs_oracle(T, d, r0, rb, shift, tol, rmax, seed):
    rng = np.random.default_rng(seed)
    A = _spd(d, shift) if shift else O.Delta(d)
    b = O.rand_tt((2,) * d, rb, rng)
    x0 = O.rand_tt((2,) * d, r0, rng)
    ref = O.mals_linsolve(A, b, x0, tol=tol, rmax=rmax)
    got = T.solvers.mals_linsolve(to_product(A), to_product(b), to_product(x0), tol=tol, rmax=rmax)
    assert list(got.ttv_rks) == list(ref.ttv_rks)
    assert list(got.ttv_ot) == list(ref.ttv_ot) == [0] + [1] * (d - 1)
    assert tt_rel_diff(to_oracle(got), ref) <= 1e-8
    assert abs(_resid(A, to_oracle(got), b) - _resid(A, ref, b)) <= 1e-8


def test_mals_solves_exactly_when_ranks_allow(T):
    rng = np.random.default_rng(8)
    d = 6
    A = _spd(d, 2.0)
    b = O.rand_tt((2,) * d, 2, rng)
    x0 = O.rand_tt((2,) * d, 2, rng)
    got = T.solvers.mals_linsolve(to_product(A), to_product(b), to_product(x0), tol=1e-14, rmax=64)
    dense = np.linalg.solve(O.qtto_to_matrix(A), O.qtt_to_vector(b))
    assert np.max(np.abs(O.qtt_to_vector(to_oracle(got)) - dense)) <= 1e-10 * np.max(np.abs(dense))


def test_mals_batch(T):
    rng = np.random.default_rng(21)
    d, B, rmax = 8, 6, 6
    A = _spd(d, 1.5)
    bs = [O.rand_tt((2,) * d, 2, rng) for _ in range(B)]
    x0s = [O.rand_tt((2,) * d, 2, rng) for _ in range(B)]
    dA = T.DeviceTTO(to_product(A))
    db = T.DeviceTT((2,) * d, bs[0].ttv_rks, batch=B)
    dx0 = T.DeviceTT((2,) * d, x0s[0].ttv_rks, batch=B)
    for i in range(B):
        db.upload(i, to_product(bs[i]))
        dx0.upload(i, to_product(x0s[i]))
    dx = T.DeviceTT((2,) * d, T.solvers.mals_capacity((2,) * d, x0s[0].ttv_rks, rmax), batch=B)
    T.solvers.mals_linsolve_(dA, db, dx0, dx, 1e-9, rmax)
    T.device.compress_status(dx)
    for i in range(B):
        ref = O.mals_linsolve(A, bs[i], x0s[i], tol=1e-9, rmax=rmax)
        got = dx.download(i)
        assert list(got.ttv_rks) == list(ref.ttv_rks)
        assert tt_rel_diff(to_oracle(got), ref) <= 1e-8


def test_mals_errors(T):
    rng = np.random.default_rng(1)
    d = 4
    b = to_product(O.rand_tt((2,) * d, 2, rng))
    x0 = to_product(O.rand_tt((2,) * d, 2, rng))
    with pytest.raises(T.TTNError):                            # singular two-site system
        T.solvers.mals_linsolve(to_product(O.tto_scale(0.0, O.id_tto(d))), b, x0)
    dA = T.DeviceTTO(to_product(O.id_tto(12)))
    big = T.DeviceTT((2,) * 12, [1] + [40] * 11 + [1])
    x12 = T.DeviceTT.from_host(to_product(O.rand_tt((2,) * 12, 2, rng)))
    with pytest.raises(T.TTNError):                            # two-site systems above the device limit
        T.solvers.mals_linsolve_(dA, x12, x12, big, 1e-10, 40)
