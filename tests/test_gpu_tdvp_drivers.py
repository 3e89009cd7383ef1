"""GPU parity of the TDVP drivers (src/solvers/tdvp.jl:45-357) on the device path of tensortrainnumerics.jl_amd/tdvp.py — site tensors,
operator cores and environments resident in HBM, the five contractions by this library's kernels, Lanczos `exponentiate` around
them — against (a) the oracle's restatement on the same inputs (dense tensors to 1e-9: both sides exponentiate to KrylovKit's
tolerance 1e-12 with their own step schedules; the gauge of the cores is free, the tensor is not) and (b) the known answers of the
reference's own tests (test/test_tdvp.jl:147-375), which tests/test_oracle_reference_pins.py also holds the oracle to."""
import math

import numpy as np
import pytest

from oracle import tt_oracle as O
from helpers import to_oracle, to_product
from test_oracle_reference_pins import heat_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import ttn_amd
    ttn_amd.ensure_init(0)
    return ttn_amd


def _rel(a, b):
    ta, tb = O.ttv_to_tensor(to_oracle(a)), O.ttv_to_tensor(to_oracle(b))
    return float(np.linalg.norm(ta - tb) / max(np.linalg.norm(tb), 1e-300))


def _zero_id(d, cplx):
    H = O.tto_scale(0.0, O.id_tto(d))
    return O._tdvp_complex_op(H) if cplx else H


def _rand_complex_tt(d, r, seed):
    rng = np.random.default_rng(seed)
    x = O.rand_tt((2,) * d, r, rng)
    y = O.rand_tt((2,) * d, r, rng)
    z = O.TTvector(d, [a + 1j * b for a, b in zip(x.ttv_vec, y.ttv_vec)], x.ttv_dims, list(x.ttv_rks), [0] * d)
    return O.scale(1.0 / O.norm(z), O.orthogonalize(z))


@pytest.mark.parametrize("d,r,dt", [(5, 3, 0.05 + 0j), (6, 4, 0.02j), (4, 2, 0.1 + 0j)])
def test_tdvp1sweep_vs_oracle(T, d, r, dt):
    H = O.tto_scale(0.3, O.Delta(d))                                       # real symmetric: ishermitian = true is the reference's default
    psi = _rand_complex_tt(d, r, 100 + d)
    ref, Fref = O.tdvp1sweep_(dt, O.copy_tt(psi), O._tdvp_complex_op(H), None)
    got, F = T.tdvp.tdvp1sweep_(dt, to_product(O.copy_tt(psi)), to_product(O._tdvp_complex_op(H)), None)
    assert got.ttv_rks == ref.ttv_rks and got.ttv_ot == [0] * d and len(F) == d + 2
    assert _rel(got, ref) < 1e-9
    # the environments are gauge dependent; their boundary entries are not
    Fh = T.tdvp.envs_to_host(F)
    assert Fh[0].shape == (1, 1, 1) and Fh[-1].shape == (1, 1, 1)
    # a second sweep with the carried environments against the oracle's
    ref2, _ = O.tdvp1sweep_(dt, ref, O._tdvp_complex_op(H), Fref)
    got2, _ = T.tdvp.tdvp1sweep_(dt, got, to_product(O._tdvp_complex_op(H)), F)
    assert _rel(got2, ref2) < 1e-9


@pytest.mark.parametrize("d,r,dt,mb", [(5, 3, 0.05 + 0j, 2 ** 62), (6, 4, 0.04j, 3), (6, 2, 0.05, 8)])
def test_tdvp2sweep_vs_oracle(T, d, r, dt, mb):
    H = O.tto_scale(0.3, O.Delta(d))
    psi = _rand_complex_tt(d, r, 200 + d)
    # (truncerr 1e-8: a cut at 1e-12 of the norm sits in the rounding noise of the singular values, where LAPACK and hipSOLVER may
    #  keep different ranks; the discarded weight bounds the distance of the two results)
    ref, _ = O.tdvp2sweep_(dt, O.copy_tt(psi), O._tdvp_complex_op(H), None, max_bond=mb, truncerr=1e-8)
    got, F = T.tdvp.tdvp2sweep_(dt, to_product(O.copy_tt(psi)), to_product(O._tdvp_complex_op(H)), None, max_bond=mb, truncerr=1e-8)
    assert got.ttv_rks == ref.ttv_rks and max(got.ttv_rks) <= mb and len(F) == d + 2
    assert _rel(got, ref) < 1e-7


def test_zero_hamiltonian_is_the_identity(T):
    d = 4                                                                   # test/test_tdvp.jl:147-160, :236-258
    psi0 = O._tdvp_complex(O.orthogonalize(O.qtt_sin(d, lam=math.pi)))
    H0 = to_product(_zero_id(d, True))
    p1, F = T.tdvp.tdvp1sweep_(complex(0.1), to_product(O.copy_tt(psi0)), H0, None)
    assert _rel(p1, psi0) < 1e-12 and len(F) == d + 2
    for dt in (0.1j, 0.05, 0.05j):
        p2, F2 = T.tdvp.tdvp2sweep_(dt, to_product(O.copy_tt(psi0)), H0, None)
        assert np.allclose(O.ttv_to_tensor(to_oracle(p2)), O.ttv_to_tensor(psi0), atol=1e-10, rtol=1e-10)
    d = 6                                                                   # :260-268: max_bond is respected
    psi0 = O._tdvp_complex(O.orthogonalize(O.add(O.qtt_sin(d, lam=math.pi), O.qtt_sin(d, lam=2 * math.pi))))
    p3, _ = T.tdvp.tdvp2sweep_(0.1j, to_product(psi0), to_product(_zero_id(d, True)), None, max_bond=2, truncerr=0.0)
    assert max(p3.ttv_rks) <= 2


@pytest.mark.parametrize("which,d,tol_id", [("tdvp", 4, 1e-10), ("tdvp2", 6, 1e-7)])
def test_driver_basic_behaviour(T, which, d, tol_id):
    run = getattr(T.tdvp, which)                                            # test/test_tdvp.jl:164-206, :270-317
    u0 = O.qtt_sin(d, lam=math.pi)
    kw = dict(normalize=False, sweeps=1, carry_env=False)
    H0c, H0r = to_product(_zero_id(d, True)), to_product(_zero_id(d, False))
    assert np.iscomplexobj(run(H0c, to_product(O._tdvp_complex(u0)), [0.1], imaginary_time=False, **kw).ttv_vec[0])
    assert not np.iscomplexobj(run(H0r, to_product(u0), [0.1], imaginary_time=True, **kw).ttv_vec[0])
    _, err = run(H0c, to_product(O._tdvp_complex(u0)), [0.1], imaginary_time=False, return_error=True, **kw)
    assert abs(err) <= 1e-6
    psi0 = O._tdvp_complex(O.orthogonalize(u0))
    assert _rel(run(H0c, to_product(psi0), [0.1], imaginary_time=False, **kw), psi0) <= tol_id
    a = run(H0c, to_product(O._tdvp_complex(u0)), [0.1, 0.1], normalize=False, sweeps=2, carry_env=True, imaginary_time=False)
    b = run(H0c, to_product(O._tdvp_complex(u0)), [0.1, 0.1], normalize=False, sweeps=2, carry_env=False, imaginary_time=False)
    assert _rel(a, b) <= 1e-10


def test_heat_eigenmode_on_the_device(T):
    A, u0, lam = heat_problem()                                             # test/test_tdvp.jl:329-356: exp(lambda t) u0 is the exact solution
    steps = [1e-3] * 5
    target = math.exp(lam * sum(steps)) * O.ttv_to_tensor(u0)
    sol = T.tdvp.tdvp(to_product(A), to_product(u0), steps, imaginary_time=True, normalize=False)
    assert not np.iscomplexobj(sol.ttv_vec[0]) and sol.ttv_ot == [0] + [-1] * (u0.N - 1)
    assert np.linalg.norm(O.ttv_to_tensor(to_oracle(sol)) - target) / np.linalg.norm(target) < 1e-8
    sol2 = T.tdvp.tdvp2(to_product(A), to_product(u0), steps, imaginary_time=True, normalize=False, max_bond=8, truncerr=1e-12)
    assert np.linalg.norm(O.ttv_to_tensor(to_oracle(sol2)) - target) / np.linalg.norm(target) < 1e-8
    # and the oracle's own run of the same driver: same tensor
    ref = O.tdvp(A, u0, steps, imaginary_time=True, normalize=False)
    assert _rel(sol, ref) < 1e-9


def test_return_error_residual_and_real_time_against_oracle(T):
    d = 4                                                                   # test/test_tdvp.jl:358-375: A = I/2, every state evolves exactly
    A = O.tto_scale(0.5, O.id_tto(d))
    u0 = O.qtt_sin(d, lam=math.pi)
    for it in (False, True):
        _, e1 = T.tdvp.tdvp(to_product(A), to_product(u0), [1e-3] * 5, imaginary_time=it, return_error=True, normalize=False)
        _, e2 = T.tdvp.tdvp2(to_product(A), to_product(u0), [1e-3] * 5, imaginary_time=it, return_error=True, normalize=False, max_bond=8, truncerr=1e-12)
        _, r1 = O.tdvp(A, u0, [1e-3] * 5, imaginary_time=it, return_error=True, normalize=False)
        assert e1 < 1e-3 and e2 < 1e-3 and abs(e1 - r1) < 1e-5        # (a norm of a difference through dot products: sqrt-of-rounding level)
    # real time with the Laplacian, normalised, two steps of two sweeps: the tensor of the oracle's run
    d = 6
    H = O.tto_scale(0.2, O.Delta(d))
    x = O.rand_tt((2,) * d, 3, np.random.default_rng(9))
    ref = O.tdvp(H, x, [0.05, 0.05], sweeps=2, normalize=True, imaginary_time=False)
    got = T.tdvp.tdvp(to_product(H), to_product(x), [0.05, 0.05], sweeps=2, normalize=True, imaginary_time=False)
    assert got.ttv_rks == ref.ttv_rks and _rel(got, ref) < 1e-9
    assert abs(np.linalg.norm(O.ttv_to_tensor(to_oracle(got))) - 1.0) < 1e-12            # norm conservation under normalize = true
    ref2 = O.tdvp2(H, x, [0.05, 0.05], sweeps=1, normalize=True, imaginary_time=False, max_bond=4, truncerr=1e-12)
    got2 = T.tdvp.tdvp2(to_product(H), to_product(x), [0.05, 0.05], sweeps=1, normalize=True, imaginary_time=False, max_bond=4, truncerr=1e-12)
    assert got2.ttv_rks == ref2.ttv_rks and _rel(got2, ref2) < 1e-8
