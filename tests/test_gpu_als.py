"""GPU parity of als_linsolve (src/solvers/als.jl:161-222): HIP path (ttn_als_linsolve) vs the CPU oracle.
Both run the SAME deterministic algorithm (dense local systems, LU with partial pivoting, thin QR core moves); QR sign
conventions differ (a gauge), so iterates are compared as tensors: ||x_gpu - x_cpu|| / ||x_cpu|| <= 1e-9 (typically 1e-13;
the bound scales with cond(K)), ranks and ot flags exact, plus the reference's own residual assertions (test/test_als.jl)."""
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
    # ||A x - b|| / ||b|| through orthogonalization (the dot-product form of norm() floors at sqrt(eps))
    return tt_norm_stable(O.sub(O.apply(A, x), b)) / max(tt_norm_stable(b), np.finfo(float).eps)


def _ot_after(d, sweep_count):
    ot = [0] + [-1] * (d - 1)          # orthogonalize(x; i = 1)
    done = 0
    while done < sweep_count:
        done += 1
        for i in range(d - 1):
            ot[i], ot[i + 1] = -1, 0
        if done == sweep_count:
            break
        done += 1
        for i in range(d - 1, 0, -1):
            ot[i], ot[i - 1] = 1, 0
    return ot


def test_als_reference_cases(T):
    """test/test_als.jl:30-77 — same shapes, operators and assertions (inputs from NumPy's generator)."""
    rng = np.random.default_rng(9999)
    # return type and structure
    dims, rks = (2, 2, 2), [1, 2, 2, 1]
    A, b, x0 = O.rand_tto(dims, 3, rng), O.rand_tt(dims, rks, rng), O.rand_tt(dims, rks, rng)
    x = T.solvers.als_linsolve(to_product(A), to_product(b), to_product(x0))
    assert x.N == 3 and tuple(x.ttv_dims) == dims and list(x.ttv_rks) == rks
    assert tt_rel_diff(to_oracle(x), O.als_linsolve(A, b, x0)) <= 1e-9
    # residual decreases for a well-conditioned system
    d = 4
    A = _spd(d, 10.0)
    b, x0 = O.rand_tt((2,) * d, [1, 2, 2, 2, 1], rng), O.rand_tt((2,) * d, [1, 2, 2, 2, 1], rng)
    x = T.solvers.als_linsolve(to_product(A), to_product(b), to_product(x0), sweep_count=4)
    assert _resid(A, to_oracle(x), b) < 0.5
    assert tt_rel_diff(to_oracle(x), O.als_linsolve(A, b, x0, sweep_count=4)) <= 1e-9
    # identity operator gives x ~ b
    A = O.id_tto(d)
    b, x0 = O.rand_tt((2,) * d, [1] * 5, rng), O.rand_tt((2,) * d, [1] * 5, rng)
    x = T.solvers.als_linsolve(to_product(A), to_product(b), to_product(x0), sweep_count=4)
    assert _resid(A, to_oracle(x), b) < 0.05
    # single forward half sweep
    d = 3
    A = _spd(d, 5.0)
    b, x0 = O.rand_tt((2,) * d, [1, 2, 2, 1], rng), O.rand_tt((2,) * d, [1, 2, 2, 1], rng)
    x = T.solvers.als_linsolve(to_product(A), to_product(b), to_product(x0), sweep_count=1)
    assert tuple(x.ttv_dims) == tuple(b.ttv_dims)
    assert tt_rel_diff(to_oracle(x), O.als_linsolve(A, b, x0, sweep_count=1)) <= 1e-9


@pytest.mark.parametrize("d,r,rb,shift,sweeps,seed", [(6, 2, 2, 3.0, 2, 0), (8, 4, 3, 3.0, 2, 1), (8, 4, 3, 0.0, 3, 2), (10, 6, 2, 1.0, 2, 3),
                                                      (12, 8, 2, 0.5, 2, 4), (5, 4, 4, 2.0, 4, 5), (2, 2, 2, 1.0, 2, 6)])
def test_als_vs_oracle(T, d, r, rb, shift, sweeps, seed):
    rng = np.random.default_rng(seed)
    A = _spd(d, shift) if shift else O.Delta(d)
    b = O.rand_tt((2,) * d, rb, rng)
    x0 = O.rand_tt((2,) * d, r, rng)
    ref = O.als_linsolve(A, b, x0, sweep_count=sweeps)
    got = T.solvers.als_linsolve(to_product(A), to_product(b), to_product(x0), sweep_count=sweeps)
    assert list(got.ttv_rks) == list(ref.ttv_rks) == list(x0.ttv_rks)
    assert list(got.ttv_ot) == list(ref.ttv_ot) == _ot_after(d, sweeps)
    assert tt_rel_diff(to_oracle(got), ref) <= 1e-9
    assert abs(_resid(A, to_oracle(got), b) - _resid(A, ref, b)) <= 1e-9


def test_als_full_rank_is_exact(T):
    """With full ranks the ALS local problem at the last site is the whole system: the answer is A \\ b."""
    rng = np.random.default_rng(3)
    d = 5
    A = _spd(d, 3.0)
    b = O.rand_tt((2,) * d, 3, rng)
    x0 = O.rand_tt((2,) * d, [1, 2, 4, 4, 2, 1], rng)
    got = T.solvers.als_linsolve(to_product(A), to_product(b), to_product(x0), sweep_count=2)
    dense = np.linalg.solve(O.qtto_to_matrix(A), O.qtt_to_vector(b))
    assert np.max(np.abs(O.qtt_to_vector(to_oracle(got)) - dense)) <= 1e-10 * np.max(np.abs(dense))


def test_als_batch_mixed_dims_and_general_operator(T):
    rng = np.random.default_rng(12)
    dims, B = (2, 3, 2, 3), 5
    rks = [1, 2, 3, 2, 1]
    A = O.rand_tto(dims, 2, rng)
    Aop = O.tto_add(A, O.tto_scale(4.0, _id_general(dims)))          # safely non-singular: A + 4 I
    bs = [O.rand_tt(dims, [1, 2, 2, 2, 1], rng) for _ in range(B)]
    x0s = [O.rand_tt(dims, rks, rng) for _ in range(B)]
    dA = T.DeviceTTO(to_product(Aop))
    db = T.DeviceTT(dims, [1, 2, 2, 2, 1], batch=B)
    dx0 = T.DeviceTT(dims, rks, batch=B)
    dx = T.DeviceTT(dims, rks, batch=B)
    for i in range(B):
        db.upload(i, to_product(bs[i]))
        dx0.upload(i, to_product(x0s[i]))
    T.solvers.als_linsolve_(dA, db, dx0, dx, 3)
    T.device.compress_status(dx)
    for i in range(B):
        ref = O.als_linsolve(Aop, bs[i], x0s[i], sweep_count=3)
        assert tt_rel_diff(to_oracle(dx.download(i)), ref) <= 1e-9


def _id_general(dims):
    cores = [np.eye(n).reshape(n, n, 1, 1) for n in dims]
    return O.TToperator(len(dims), cores, tuple(dims), [1] * (len(dims) + 1), [0] * len(dims))


def test_als_errors(T):
    rng = np.random.default_rng(1)
    d = 4
    zero = O.tto_scale(0.0, O.id_tto(d))
    b = to_product(O.rand_tt((2,) * d, 2, rng))
    x0 = to_product(O.rand_tt((2,) * d, 2, rng))
    with pytest.raises(T.TTNError):                           # singular local system (LAPACK: SingularException)
        T.solvers.als_linsolve(to_product(zero), b, x0)
    fat = to_product(O.rand_tt((2,) * d, [1, 4, 4, 4, 1], rng))        # ranks orthogonalize would cut
    with pytest.raises(T.TTNError):
        T.solvers.als_linsolve(to_product(O.id_tto(d)), b, fat)
    with pytest.raises(AssertionError):
        T.solvers.als_linsolve(to_product(O.id_tto(d)), b, x0, sweep_count=0)


# ------------------------------------------------------------------------------------------------------------------------
# the GRID form (csrc/ttn_als_grid.h): local systems beyond the 2048 unknowns of the one-workgroup LU — assembly and blocked LU on
# the whole chip, the half sweeps walked on the host
# ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d,r,shift,sweeps", [(8, 4, 0.5, 3), (10, 6, 1.0, 2), (6, 8, 0.0, 4)])
def test_als_grid_form_equals_one_workgroup_form(T, monkeypatch, d, r, shift, sweeps):
    """The same problems through both forms (TTN_ALS_GRID=1 forces the grid form on systems the one-workgroup form can take): same
    operations in the same order, so the iterates agree to rounding; and both agree with the oracle."""
    rng = np.random.default_rng(300 + d)
    A = _spd(d, shift)
    b, x0 = O.rand_tt((2,) * d, 3, rng), O.rand_tt((2,) * d, r, rng)
    one = T.solvers.als_linsolve(to_product(A), to_product(b), to_product(x0), sweep_count=sweeps)
    monkeypatch.setenv("TTN_ALS_GRID", "1")
    grid = T.solvers.als_linsolve(to_product(A), to_product(b), to_product(x0), sweep_count=sweeps)
    assert list(grid.ttv_rks) == list(one.ttv_rks) and list(grid.ttv_ot) == list(one.ttv_ot) == _ot_after(d, sweeps)
    assert tt_rel_diff(to_oracle(grid), to_oracle(one)) <= 1e-12
    assert tt_rel_diff(to_oracle(grid), O.als_linsolve(A, b, x0, sweep_count=sweeps)) <= 1e-9


def test_als_grid_form_batch_and_singular_system(T, monkeypatch):
    """A batch through the grid form (one train after the other) and LAPACK's SingularException: an exactly singular local system
    (the zero operator) is reported as TTN_ERR_SINGULAR through the handle's status, like the one-workgroup form."""
    monkeypatch.setenv("TTN_ALS_GRID", "1")
    rng = np.random.default_rng(77)
    d = 6
    A = _spd(d, 2.0)
    dA = T.DeviceTTO(to_product(A))
    bs = [O.rand_tt((2,) * d, 2, rng) for _ in range(3)]
    xs = [O.rand_tt((2,) * d, 4, rng) for _ in range(3)]
    db, dx0 = T.DeviceTT((2,) * d, bs[0].ttv_rks, batch=3), T.DeviceTT((2,) * d, xs[0].ttv_rks, batch=3)
    for k in range(3):
        db.upload(k, to_product(bs[k])); dx0.upload(k, to_product(xs[k]))
    dx = T.DeviceTT((2,) * d, xs[0].ttv_rks, batch=3)
    T.solvers.als_linsolve_(dA, db, dx0, dx, 2)
    T.device.compress_status(dx)
    for k in range(3):
        assert tt_rel_diff(to_oracle(dx.download(k)), O.als_linsolve(A, bs[k], xs[k], sweep_count=2)) <= 1e-9
    Z = O.tto_scale(0.0, O.id_tto(d))
    with pytest.raises(T._lib.TTNError):
        T.solvers.als_linsolve(to_product(Z), to_product(bs[0]), to_product(xs[0]))


def test_als_rank40_beyond_the_one_workgroup_limit(T):
    """Rank-40 start train on 2 x 7 bits: one-site systems of 2 * 40 * 40 = 3200 unknowns (> 2048: only the grid form can take them;
    round 2 refused the call).  Operator: the 2D Laplacian of examples/Laplace_pde.jl scaled by h^2 plus the identity (cond ~ 9), so
    the iterate is pinned: ranks and gauge flags exact, tensor 1e-9, residual equal."""
    rng = np.random.default_rng(11)
    dd = 7
    L1, I1 = O.toeplitz_to_qtto(2.0, -1.0, -1.0, dd), O.id_tto(dd)
    kron = lambda P_, Q_: O.TToperator(P_.N + Q_.N, list(P_.tto_vec) + list(Q_.tto_vec), tuple(P_.tto_dims) + tuple(Q_.tto_dims),   # noqa: E731
                                       list(P_.tto_rks[:-1]) + list(Q_.tto_rks), [0] * (P_.N + Q_.N))
    A = O.tto_add(O.tto_add(kron(L1, I1), kron(I1, L1)), kron(I1, I1))
    d = 2 * dd
    b = O.rand_tt((2,) * d, 3, rng)
    x0 = O.rand_tt((2,) * d, 40, rng)
    assert max(2 * x0.ttv_rks[i] * x0.ttv_rks[i + 1] for i in range(d)) == 3200
    ref = O.als_linsolve(A, b, x0, sweep_count=2)
    got = T.solvers.als_linsolve(to_product(A), to_product(b), to_product(x0), sweep_count=2)
    assert list(got.ttv_rks) == list(ref.ttv_rks) and list(got.ttv_ot) == list(ref.ttv_ot)
    assert tt_rel_diff(to_oracle(got), ref) <= 1e-9
    assert abs(_resid(A, to_oracle(got), b) - _resid(A, ref, b)) <= 1e-9
