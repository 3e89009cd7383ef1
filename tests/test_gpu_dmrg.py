"""GPU parity of dmrg_linsolve, N = 2 (src/solvers/dmrg.jl:388-472): HIP path (ttn_dmrg_linsolve) vs the CPU oracle.
Both sides solve every local system densely (the reference's it_solver = false branch); the reference's default KrylovKit
branch reaches the same local solutions to linsolv_tol only, so agreement with IT is at that level by construction.
Tolerances: adapted ranks exact where the local spectra have a gap around the cut_off_index threshold (tol = 1e-10 ... 1e-6
here); iterate as a tensor 1e-8 relative (cond(K) and the gaps of the SVD split amplify rounding); the reference's own
assertions (test/test_dmrg.jl:29-75) on its shapes."""
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


def test_dmrg_reference_cases(T):
    """test/test_dmrg.jl:29-75 — same shapes / operators / schedules / assertions (inputs from NumPy's generator)."""
    rng = np.random.default_rng(1234)
    d = 4
    b, x0 = O.rand_tt((2,) * d, [1, 2, 2, 2, 1], rng), O.rand_tt((2,) * d, [1, 2, 2, 2, 1], rng)
    x = T.solvers.dmrg_linsolve(to_product(_spd(d, 3.0)), to_product(b), to_product(x0), sweep_schedule=[2], rmax_schedule=[4])
    assert x.N == d and tuple(x.ttv_dims) == (2,) * d and all(np.isfinite(x.ttv_rks))
    A = _spd(d, 10.0)
    x = T.solvers.dmrg_linsolve(to_product(A), to_product(b), to_product(x0), sweep_schedule=[4], rmax_schedule=[8])
    assert _resid(A, to_oracle(x), b) < 0.5
    x1 = O.rand_tt((2,) * d, [1] * 5, rng)
    x = T.solvers.dmrg_linsolve(to_product(_spd(d, 5.0)), to_product(b), to_product(x1), sweep_schedule=[2, 4], rmax_schedule=[2, 8])
    assert tuple(x.ttv_dims) == (2,) * d
    b1 = O.rand_tt((2,) * d, [1] * 5, rng)
    x = T.solvers.dmrg_linsolve(to_product(O.id_tto(d)), to_product(b1), to_product(x1), sweep_schedule=[4], rmax_schedule=[4])
    assert _resid(O.id_tto(d), to_oracle(x), b1) < 0.05


@pytest.mark.parametrize("d,r0,rb,shift,tol,sched,rmaxs,seed", [
    (6, 2, 2, 2.0, 1e-10, [2], [64], 0), (8, 2, 2, 3.0, 1e-8, [3], [8], 1), (8, 3, 2, 0.0, 1e-10, [2, 4], [3, 6], 2),
    (10, 2, 3, 1.0, 1e-6, [2], [8], 3), (5, 1, 2, 2.0, 1e-10, [1, 2, 3], [2, 3, 4], 4), (2, 1, 2, 1.0, 1e-10, [2], [2], 5),
    (3, 2, 2, 1.0, 1e-10, [2], [4], 6), (6, 2, 2, 2.0, 1e-10, [1], [4], 7)])
def test_dmrg_vs_oracle(T, d, r0, rb, shift, tol, sched, rmaxs, seed):
    rng = np.random.default_rng(seed)
    A = _spd(d, shift) if shift else O.Delta(d)
    b = O.rand_tt((2,) * d, rb, rng)
    x0 = O.rand_tt((2,) * d, r0, rng)
    ref = O.dmrg_linsolve(A, b, x0, tol=tol, sweep_schedule=sched, rmax_schedule=rmaxs)
    got = T.solvers.dmrg_linsolve(to_product(A), to_product(b), to_product(x0), tol=tol, sweep_schedule=sched, rmax_schedule=rmaxs)
    assert list(got.ttv_rks) == list(ref.ttv_rks)
    assert list(got.ttv_ot) == list(ref.ttv_ot) == [0] + [-1] * (d - 1)
    assert tt_rel_diff(to_oracle(got), ref) <= 1e-8
    assert abs(_resid(A, to_oracle(got), b) - _resid(A, ref, b)) <= 1e-8


def test_dmrg_solves_exactly_when_ranks_allow(T):
    rng = np.random.default_rng(8)
    d = 6
    A = _spd(d, 2.0)
    b = O.rand_tt((2,) * d, 2, rng)
    x0 = O.rand_tt((2,) * d, 2, rng)
    got = T.solvers.dmrg_linsolve(to_product(A), to_product(b), to_product(x0), tol=1e-14, sweep_schedule=[3], rmax_schedule=[64])
    dense = np.linalg.solve(O.qtto_to_matrix(A), O.qtt_to_vector(b))
    assert np.max(np.abs(O.qtt_to_vector(to_oracle(got)) - dense)) <= 1e-10 * np.max(np.abs(dense))


def test_dmrg_batch(T):
    rng = np.random.default_rng(21)
    d, B = 8, 6
    sched, rmaxs = [2, 3], [4, 6]
    A = _spd(d, 1.5)
    bs = [O.rand_tt((2,) * d, 2, rng) for _ in range(B)]
    x0s = [O.rand_tt((2,) * d, 2, rng) for _ in range(B)]
    dA = T.DeviceTTO(to_product(A))
    db = T.DeviceTT((2,) * d, bs[0].ttv_rks, batch=B)
    dx0 = T.DeviceTT((2,) * d, x0s[0].ttv_rks, batch=B)
    for i in range(B):
        db.upload(i, to_product(bs[i]))
        dx0.upload(i, to_product(x0s[i]))
    dx = T.DeviceTT((2,) * d, T.solvers.mals_capacity((2,) * d, x0s[0].ttv_rks, max(rmaxs)), batch=B)
    T.solvers.dmrg_linsolve_(dA, db, dx0, dx, 1e-9, sched, rmaxs)
    T.device.compress_status(dx)
    for i in range(B):
        ref = O.dmrg_linsolve(A, bs[i], x0s[i], tol=1e-9, sweep_schedule=sched, rmax_schedule=rmaxs)
        got = dx.download(i)
        assert list(got.ttv_rks) == list(ref.ttv_rks)
        assert tt_rel_diff(to_oracle(got), ref) <= 1e-8


def test_dmrg_errors(T):
    rng = np.random.default_rng(1)
    d = 4
    A = to_product(_spd(d, 2.0))
    b = to_product(O.rand_tt((2,) * d, 2, rng))
    x0 = to_product(O.rand_tt((2,) * d, 2, rng))
    with pytest.raises(T.TTNError):                            # a schedule the reference's while-loop never leaves
        T.solvers.dmrg_linsolve(A, b, x0, sweep_schedule=[2, 2], rmax_schedule=[2, 4])
    with pytest.raises(T.TTNError):
        T.solvers.dmrg_linsolve(A, b, x0, sweep_schedule=[0], rmax_schedule=[4])
    with pytest.raises(T.TTNError):                            # rmax_schedule[i_schedule] out of bounds in the reference
        T.solvers.dmrg_linsolve(A, b, x0, sweep_schedule=[2, 4], rmax_schedule=[4])
    with pytest.raises(T.TTNError):                            # single-site scheme: als_linsolve
        T.solvers.dmrg_linsolve(A, b, x0, N=1)
    with pytest.raises(T.TTNError):                            # singular two-site system
        T.solvers.dmrg_linsolve(to_product(O.tto_scale(0.0, O.id_tto(d))), b, x0)


# ---- the matrix-free local solver (dmrg.jl:92-171: `it_solver || N > itslv_thresh` -> conjugate gradients) --------------------------
# The CG iteration itself lives in KrylovKit (third party, not in the reference tree): the oracle restates its published recurrence
# (O.cg_solve) on the reference's symmetrised operator and start vectors.  A converged CG solve agrees with the dense solve to
# linsolv_tol * cond(K) whatever the rounding path, so the bar here is: ranks exact, iterate as a tensor 1e-7 relative against the
# oracle's CG run with the same keywords, and against the DENSE device path to the solver tolerance.
@pytest.mark.parametrize("d,r0,rb,shift,tol,sched,rmaxs,lt,seed", [
    (6, 2, 2, 2.0, 1e-10, [2], [8], 1e-12, 0), (8, 2, 2, 3.0, 1e-8, [3], [8], 1e-12, 1), (8, 3, 2, 1.0, 1e-10, [2, 4], [3, 6], 1e-12, 2),
    (10, 2, 3, 1.0, 1e-6, [2], [8], 1e-10, 3), (5, 1, 2, 2.0, 1e-10, [1, 2, 3], [2, 3, 4], 1e-13, 4), (9, 4, 2, 2.0, 1e-10, [2], [16], 1e-12, 5)])
def test_dmrg_matrix_free_cg_vs_oracle_and_dense(T, d, r0, rb, shift, tol, sched, rmaxs, lt, seed):
    rng = np.random.default_rng(100 + seed)
    A = _spd(d, shift)
    b = O.rand_tt((2,) * d, rb, rng)
    x0 = O.rand_tt((2,) * d, r0, rng)
    st = {}
    ref = O.dmrg_linsolve(A, b, x0, tol=tol, sweep_schedule=sched, rmax_schedule=rmaxs, it_solver=True, linsolv_tol=lt, stats=st)
    got = T.solvers.dmrg_linsolve(to_product(A), to_product(b), to_product(x0), tol=tol, sweep_schedule=sched, rmax_schedule=rmaxs,
                                  it_solver=True, linsolv_tol=lt)
    iters = T.solvers.dmrg_cg_iterations(1)[0]
    assert iters > 0 and abs(iters - st["cg_iterations"]) <= max(4, st["cg_iterations"] // 10), (iters, st)
    assert list(got.ttv_rks) == list(ref.ttv_rks)
    assert list(got.ttv_ot) == list(ref.ttv_ot)
    assert tt_rel_diff(to_oracle(got), ref) <= 1e-7
    dense = T.solvers.dmrg_linsolve(to_product(A), to_product(b), to_product(x0), tol=tol, sweep_schedule=sched, rmax_schedule=rmaxs)
    assert T.solvers.dmrg_cg_iterations(1)[0] == 0                     # (every system below 2048 unknowns: solved by LU)
    assert list(dense.ttv_rks) == list(got.ttv_rks)
    assert tt_rel_diff(to_oracle(got), to_oracle(dense)) <= 1e-7


def test_dmrg_itslv_thresh_switches_per_system(T):
    """itslv_thresh = 20: the windows at the ends of the chain (<= 16 unknowns) are solved densely, the inner ones by CG — the
    reference's per-system switch (dmrg.jl:96).  Oracle with the same keywords."""
    rng = np.random.default_rng(77)
    d = 8
    A = _spd(d, 2.0)
    b = O.rand_tt((2,) * d, 2, rng)
    x0 = O.rand_tt((2,) * d, 3, rng)
    kw = dict(tol=1e-10, sweep_schedule=[2], rmax_schedule=[6], linsolv_tol=1e-12, itslv_thresh=20)
    st = {}
    ref = O.dmrg_linsolve(A, b, x0, stats=st, **kw)
    got = T.solvers.dmrg_linsolve(to_product(A), to_product(b), to_product(x0), **kw)
    assert 0 < st["cg_solves"] < 2 * (d - 2) + 1
    assert list(got.ttv_rks) == list(ref.ttv_rks)
    assert tt_rel_diff(to_oracle(got), ref) <= 1e-7


def test_dmrg_cg_hits_maxiter_like_the_reference(T):
    """linsolv_maxiter = 3: KrylovKit returns the current iterate (with a warning) and the sweep goes on; so does the device."""
    rng = np.random.default_rng(78)
    d = 7
    A = _spd(d, 0.5)
    b = O.rand_tt((2,) * d, 2, rng)
    x0 = O.rand_tt((2,) * d, 2, rng)
    kw = dict(tol=1e-10, sweep_schedule=[2], rmax_schedule=[4], it_solver=True, linsolv_tol=1e-14, linsolv_maxiter=3)
    ref = O.dmrg_linsolve(A, b, x0, **kw)
    got = T.solvers.dmrg_linsolve(to_product(A), to_product(b), to_product(x0), **kw)
    assert list(got.ttv_rks) == list(ref.ttv_rks)
    assert tt_rel_diff(to_oracle(got), ref) <= 1e-7


def test_dmrg_large_local_systems_beyond_the_dense_limit(T):
    """Rank-40 start train on 2 x 7 bits: two-site systems of 4 * 40 * 40 = 6400 unknowns (> 2048: only the matrix-free path can take
    them; the dense path would need a 328 MB matrix per system).  Operator: the 2D Laplacian of examples/Laplace_pde.jl scaled by h^2
    plus the identity (cond ~ 9: CG converges to linsolv_tol well inside maxiter, so the iterate is pinned)."""
    rng = np.random.default_rng(79)
    dd = 7
    L1 = O.toeplitz_to_qtto(2.0, -1.0, -1.0, dd)
    I1 = O.id_tto(dd)
    kron = lambda P_, Q_: O.TToperator(P_.N + Q_.N, list(P_.tto_vec) + list(Q_.tto_vec), tuple(P_.tto_dims) + tuple(Q_.tto_dims),   # noqa: E731
                                       list(P_.tto_rks[:-1]) + list(Q_.tto_rks), [0] * (P_.N + Q_.N))
    A = O.tto_add(O.tto_add(kron(L1, I1), kron(I1, L1)), kron(I1, I1))
    d = 2 * dd
    b = O.rand_tt((2,) * d, 3, rng)
    x0 = O.rand_tt((2,) * d, 40, rng)
    assert max(4 * x0.ttv_rks[i] * x0.ttv_rks[i + 2] for i in range(d - 1)) == 6400
    kw = dict(tol=1e-8, sweep_schedule=[2], rmax_schedule=[40], it_solver=True, linsolv_tol=1e-9)
    st = {}
    ref = O.dmrg_linsolve(A, b, x0, stats=st, **kw)
    got = T.solvers.dmrg_linsolve(to_product(A), to_product(b), to_product(x0), **kw)
    assert list(got.ttv_rks) == list(ref.ttv_rks), (got.ttv_rks, ref.ttv_rks)
    assert tt_rel_diff(to_oracle(got), ref) <= 1e-7
    assert abs(_resid(A, to_oracle(got), b) - _resid(A, ref, b)) <= 1e-7
