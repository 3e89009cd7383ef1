"""Pins the CPU oracle (oracle/tt_oracle.py) to the reference's own known-answer tests.

Each test restates the INPUTS and EXPECTED PROPERTIES of a test in /root/reference/test
(cited per test); expected values are closed forms or dense linear algebra, so no Julia
run is needed.  CPU only.
"""
import math

import numpy as np
import pytest

from oracle import tt_oracle as O

PI = math.pi


def _rng(seed):
    return np.random.default_rng(seed)


# test/test_tt_tools.jl:929-947
def test_r_and_d_to_rks_reference_vectors():
    dims = (2, 3, 4)
    out = O.r_and_d_to_rks([1, 100, 100, 1], dims)
    assert out[0] == 1 and out[-1] == 1 and out[1] <= 2 and out[2] <= 4
    assert out == [1, 2, 4, 1]
    assert O.r_and_d_to_rks([1, 1, 1, 1], dims) == [1, 1, 1, 1]
    assert O.r_and_d_to_rks([5, 5, 5], (0, 2), rmax=4) == [1, 2, 1]
    assert O.r_and_d_to_rks([5, 5, 5], (0, 0), rmax=4) == [1, 4, 1]


# C3 rank profile of rand_tt(dims, 64) (src/tt_tools.jl:134-139)
def test_rank_profile_c3():
    rks = O.r_and_d_to_rks([64] * 31, (2,) * 30, rmax=64)
    assert rks == [1, 2, 4, 8, 16, 32] + [64] * 19 + [32, 16, 8, 4, 2, 1]


# test/test_qtt_tools.jl:111-126 + closed form sin(lam*pi*x) on the reference grid
@pytest.mark.parametrize("d", [3, 6, 8])
def test_qtt_sin_closed_form(d):
    tt = O.qtt_sin(d, lam=PI)
    assert tt.ttv_vec[0].shape == (2, 1, 2) and tt.ttv_vec[d - 1].shape == (2, 2, 1)
    assert tt.ttv_vec[0][0, 0, 0] == math.sin(0.0) and tt.ttv_vec[0][0, 0, 1] == math.cos(0.0)
    x = np.linspace(0, 1, 2 ** d)
    assert np.allclose(O.qtt_to_vector(tt), np.sin(PI * PI * x), atol=1e-12)


# test/test_qtt_multidim.jl:240-242 and test/test_tt_operators.jl (Δ vs tridiagonal)
@pytest.mark.parametrize("d", [2, 3, 4, 6])
def test_delta_is_tridiagonal(d):
    A = O.Delta(d)
    assert A.tto_rks == [1] + [3] * (d - 1) + [1]
    n = 2 ** d
    ref = 2 * np.eye(n) - np.eye(n, k=1) - np.eye(n, k=-1)
    assert np.array_equal(O.qtto_to_matrix(A), ref)


def test_toeplitz_general_bands():
    d = 4
    n = 2 ** d
    M = O.qtto_to_matrix(O.toeplitz_to_qtto(0.5, 3.0, -7.0, d))
    ref = 0.5 * np.eye(n) + 3.0 * np.eye(n, k=1) - 7.0 * np.eye(n, k=-1)  # beta super, gamma sub
    assert np.array_equal(M, ref)


# test/test_tt_operations.jl:41-71 (hadamard closed forms, atol 1e-12)
def test_hadamard_closed_forms():
    d = 8
    x = np.linspace(0, 1, 2 ** d)
    A1 = O.qtt_exp(d)
    A2 = O.qtt_sin(d, lam=PI)
    A3 = O.qtt_cos(d, lam=PI)
    A4 = O.qtt_polynom([0.0, 2.0, 3.0, -8.0, -5.0], d, a=0.0, b=1.0)
    poly = 2 * x + 3 * x ** 2 - 8 * x ** 3 - 5 * x ** 4
    assert np.allclose(O.qtt_to_vector(O.hadamard(A2, A3)), np.cos(PI ** 2 * x) * np.sin(PI ** 2 * x), atol=1e-12, rtol=0)
    assert np.allclose(O.qtt_to_vector(O.hadamard(A1, A2)), np.exp(x) * np.sin(PI ** 2 * x), atol=1e-12, rtol=0)
    assert np.allclose(O.qtt_to_vector(O.hadamard(A4, A2)), poly * np.sin(PI ** 2 * x), atol=1e-12, rtol=0)
    assert np.allclose(O.qtt_to_vector(O.hadamard(A4, A3)), poly * np.cos(PI ** 2 * x), atol=1e-12, rtol=0)


# test/test_tt_operations.jl:106-114
def test_add_inplace_ranks():
    rng = _rng(1)
    x = O.rand_tt((2, 3), [1, 2, 1], rng)
    y = O.rand_tt((2, 3), [1, 3, 1], rng)
    expected = O.ttv_to_tensor(O.add(x, y))
    assert np.allclose(expected, O.ttv_to_tensor(x) + O.ttv_to_tensor(y), atol=1e-12)
    r = O.add_(x, y)
    assert r is x
    assert np.allclose(O.ttv_to_tensor(x), expected, atol=1e-12)
    assert x.ttv_rks == [1, 5, 1] and all(o == 0 for o in x.ttv_ot)


# test/test_tt_operations.jl:116-122 + dense check of the contraction
def test_apply_vs_dense():
    rng = _rng(2)
    dims = (2, 3)
    A = O.rand_tto(dims, 2, rng)
    v = O.rand_tt(dims, [1, 2, 1], rng)
    y = O.apply(A, v)
    assert y.ttv_rks == [a * b for a, b in zip(A.tto_rks, v.ttv_rks)] and all(o == 0 for o in y.ttv_ot)
    TA = O.tto_to_tensor(A)  # [x1,x2,y1,y2]
    ref = np.einsum("abcd,cd->ab", TA, O.ttv_to_tensor(v))
    assert np.allclose(O.ttv_to_tensor(y), ref, atol=1e-12)


# test/test_qtt_multidim.jl:658-682 restated in 1-D: Δ action vs dense matvec, < 1e-8
def test_laplacian_action():
    d = 6
    v = O.qtt_sin(d, lam=1.0)
    Av = O.apply(O.Delta(d), v)
    ref = O.qtto_to_matrix(O.Delta(d)) @ O.qtt_to_vector(v)
    assert np.max(np.abs(O.qtt_to_vector(Av) - ref)) < 1e-8


# test/test_tt_operations.jl:303-320 and test/test_qtt_multidim.jl:464-473
def test_dot_norm_vs_dense():
    d = 8
    A1 = O.qtt_exp(d)
    A2 = O.qtt_sin(d, lam=PI)
    S1, S2 = O.qtt_to_vector(A1), O.qtt_to_vector(A2)
    assert O.euclidean_distance(A1, A1) == 0.0
    assert math.isclose(O.dot(A1, A2), float(S1 @ S2), rel_tol=1e-10)
    assert math.isclose(O.norm(A2), float(np.linalg.norm(S2)), rel_tol=1e-10)
    assert abs(math.sqrt(S1 @ S1 - 2 * (S1 @ S2) + S2 @ S2) - O.euclidean_distance(A1, A2)) < 1e-10


# test/test_tt_tools.jl:981-1017
def test_orthogonalize_properties():
    rng = _rng(3)
    dims = (2, 3, 4)
    tt = O.rand_tt(dims, [1, 2, 3, 1], rng)
    T0 = O.ttv_to_tensor(tt)
    for center in (1, 2, 3):
        orth = O.orthogonalize(tt, i=center)
        assert np.allclose(O.ttv_to_tensor(orth), T0, atol=1e-12)
        assert orth.ttv_ot[center - 1] == 0
        assert all(orth.ttv_ot[j] == 1 for j in range(center - 1))
        assert all(orth.ttv_ot[j] == -1 for j in range(center, 3))
        for j in range(center - 1):
            G = orth.ttv_vec[j]
            n, rl, rr = G.shape
            A = G.transpose(1, 0, 2).reshape(rl * n, rr, order="F")  # reshape(permutedims(G,(2,1,3)), rl*n, rr)
            assert np.allclose(A.T @ A, np.eye(rr), atol=1e-12)
        for j in range(center, 3):
            G = orth.ttv_vec[j]
            n, rl, rr = G.shape
            A = G.transpose(1, 2, 0).reshape(rl, rr * n, order="F")
            assert np.allclose(A @ A.T, np.eye(rl), atol=1e-12)


# test/test_tdvp.jl:28-44
def test_svdtrunc_truncerr_zero():
    rng = _rng(4)
    A = rng.standard_normal((6, 4))
    U, S, Vt = O.svdtrunc(A, max_bond=100, truncerr=0.0)
    assert U.shape[0] == 6 and Vt.shape[1] == 4 and len(S) == U.shape[1] == Vt.shape[0] == 4
    U2, S2, Vt2 = O.svdtrunc(A, max_bond=2, truncerr=0.0)
    assert len(S2) == 2
    assert np.allclose(S2, np.linalg.svd(A, compute_uv=False)[:2], rtol=1e-12, atol=1e-12)
    assert len(O.svdtrunc(rng.standard_normal((5, 5)), max_bond=1)[1]) == 1


def test_svdtrunc_relative_tail_rule():
    # src/tt_cross_interpolation.jl:153-163: drop the longest tail with 2-norm <= truncerr*||s||
    s = np.array([1.0, 1e-3, 1e-7, 1e-9])
    A = np.diag(s)
    assert len(O.svdtrunc(A, truncerr=1e-6)[1]) == 2
    assert len(O.svdtrunc(A, truncerr=1e-8)[1]) == 3
    assert len(O.svdtrunc(A, truncerr=0.0)[1]) == 4
    # all-zero matrix: loop never breaks, r stays len(s)
    assert len(O.svdtrunc(np.zeros((3, 3)), truncerr=1e-3)[1]) == 3


# test/test_tt_tools.jl:433-498
def test_bond_truncate_shapes_rank1_and_assert():
    rng = _rng(5)
    tt = O.TTvector(3, [rng.standard_normal((2, 1, 4)), rng.standard_normal((2, 4, 4)), rng.standard_normal((2, 4, 1))],
                    (2, 2, 2), [1, 4, 4, 1], [0, 0, 0])
    y = O.tt_bond_truncate_(tt, 1, max_bond=2, truncerr=0.0, faithful=True)
    assert tt.ttv_rks[1] <= 2
    r = tt.ttv_rks[1]
    assert tt.ttv_vec[0].shape == (2, 1, r) and tt.ttv_vec[1].shape == (2, r, 4)
    assert y.ttv_rks[1] == tt.ttv_rks[1] and y.ttv_vec[0].shape == tt.ttv_vec[0].shape

    u, v, p, q = [1.2, -0.5], [0.7, 0.3], [2.0, 3.0], [4.0, 5.0]
    c1 = np.zeros((2, 1, 2))
    c2 = np.zeros((2, 2, 1))
    for s in range(2):
        for g in range(2):
            c1[s, 0, g] = p[g] * u[s]
            c2[s, g, 0] = q[g] * v[s]
    tt2 = O.TTvector(2, [c1, c2], (2, 2), [1, 2, 1], [0, 0])
    T0 = O.ttv_to_tensor(tt2)
    O.tt_bond_truncate_(tt2, 1, max_bond=1)
    assert tt2.ttv_rks[1] == 1 and tt2.ttv_vec[0].shape == (2, 1, 1) and tt2.ttv_vec[1].shape == (2, 1, 1)
    assert np.allclose(O.ttv_to_tensor(tt2), T0, atol=1e-12)

    tt3 = O.rand_tt((2, 2, 2), [1, 2, 2, 1], rng)
    with pytest.raises(AssertionError):
        O.tt_bond_truncate_(tt3, 0)
    with pytest.raises(AssertionError):
        O.tt_bond_truncate_(tt3, tt3.N)


# test/test_tt_tools.jl:500-574
def test_tt_compress_behaviour():
    rng = _rng(6)
    tt = O.rand_tt((2, 2, 2), [1, 2, 2, 1], rng)
    before = list(tt.ttv_rks)
    T0 = O.ttv_to_tensor(tt)
    y = O.tt_compress_(tt, 10, sweeps=1)
    assert y is tt and tt.ttv_rks == before
    assert np.allclose(O.ttv_to_tensor(tt), T0, atol=1e-12)

    tt = O.rand_tt((2, 2, 2, 2), [1, 4, 4, 4, 1], rng)
    y = O.tt_compress_(tt, 2, sweeps=1)
    assert y is tt and max(tt.ttv_rks) <= 2
    for i in range(4):
        assert tt.ttv_vec[i].shape == (2, tt.ttv_rks[i], tt.ttv_rks[i + 1])
    with pytest.raises(AssertionError):
        O.tt_compress_(O.rand_tt((2, 2, 2), [1, 2, 2, 1], rng), 2, sweeps=0)
    tt = O.rand_tt((2, 2, 2), [1, 3, 3, 1], rng)
    assert O.tt_compress_(tt, 3, sweeps=2, truncerr=0.0) is tt


# test/test_qtt_multidim.jl:577-597 restated with closed-form inputs: a separable exponential
# (rank-1 in QTT) padded to rank 4 compresses back to rank 1 with values < 1e-10.
def test_compress_separable_exp_to_rank1():
    d = 12
    e = O.qtt_exp(d, alpha=-1.0)
    padded = O.add(O.add(e, O.scale(0.5, e)), O.add(O.scale(-0.25, e), e))  # rank 4, same function * 2.25
    assert max(padded.ttv_rks) == 4
    O.tt_compress_(padded, 10, truncerr=1e-12)
    assert max(padded.ttv_rks) == 1
    x = np.linspace(0, 1, 2 ** d)
    assert np.max(np.abs(O.qtt_to_vector(padded) - 2.25 * np.exp(-x))) < 1e-10


# test/test_qtt_multidim.jl:599-614 restated: sin*sin, max_bond 8, truncerr 1e-12 -> < 1e-8
def test_compress_sinsin_accuracy():
    d = 10
    s = O.qtt_sin(d, lam=2.0)
    h = O.hadamard(s, O.qtt_cos(d, lam=3.0))           # rank 4, exactly representable with rank <= 4
    big = O.add(h, O.scale(1e-3, O.hadamard(s, s)))       # rank 8
    ref = O.qtt_to_vector(big)
    O.tt_compress_(big, 8, truncerr=1e-12)
    assert max(big.ttv_rks) <= 8
    assert np.max(np.abs(O.qtt_to_vector(big) - ref)) < 1e-8


# BASELINE.json config 1 + README.md:84-103 inputs: tt_compress!(id_tto(6)*qtt_sin(6, λ=π), 2)
def test_config1_plumbing():
    A, x = O.id_tto(6), O.qtt_sin(6, lam=PI)
    y = O.apply(A, x)
    assert y.ttv_rks == [1, 2, 2, 2, 2, 2, 1]
    O.tt_compress_(y, 2)
    assert y.ttv_rks == [1, 2, 2, 2, 2, 2, 1]
    grid = np.linspace(0, 1, 64)
    assert np.allclose(O.qtt_to_vector(y), np.sin(PI * PI * grid), atol=1e-12)


# test/test_euler.jl:269-298: one RK4 step built from apply + '+' + scalar* + tt_compress!
def test_rk4_step_vs_dense():
    rng = _rng(7)
    d = 4
    hh = 1 / d ** 2
    A = O.toeplitz_to_qtto(-2.0, 1.0, 1.0, d)
    A = O.TToperator(A.N, [c.copy() for c in A.tto_vec], A.tto_dims, A.tto_rks, A.tto_ot)
    A.tto_vec[0] = (-hh ** 2) * A.tto_vec[0]   # scalar * TToperator scales the first core with ot == 0
    u0 = O.rand_tt((2,) * d, [1, 2, 2, 2, 1], rng)
    h, mb = 0.05, 8

    def cmp(t):
        return O.tt_compress_(t, mb)
    k1 = O.apply(A, u0)
    k2 = O.apply(A, cmp(O.add(u0, O.scale(h / 2, k1))))
    k3 = O.apply(A, cmp(O.add(u0, O.scale(h / 2, k2))))
    k4 = O.apply(A, cmp(O.add(u0, O.scale(h, k3))))
    incr = O.scale(h / 6, cmp(O.add(O.add(O.add(k1, O.scale(2, k2)), O.scale(2, k3)), k4)))
    sol = cmp(O.add(u0, incr))

    Ad, ud = O.qtto_to_matrix(A), O.qtt_to_vector(u0)
    K1 = Ad @ ud
    K2 = Ad @ (ud + h / 2 * K1)
    K3 = Ad @ (ud + h / 2 * K2)
    K4 = Ad @ (ud + h * K3)
    ref = ud + h / 6 * (K1 + 2 * K2 + 2 * K3 + K4)
    assert np.linalg.norm(O.qtt_to_vector(sol) - ref) / np.linalg.norm(ref) < 1e-6


# ---- hadamard_ttm / reorder ------------------------------------------------------------------------------
def test_hadamard_ttm_known_answers():
    """test/test_tt_operations.jl:72-99 — "Hadamard TTM algorithm vs naive", same inputs and tolerances."""
    d = 8
    xp = np.linspace(0, 1, 2 ** d)
    A1, A2, A3 = O.qtt_exp(d), O.qtt_sin(d, lam=math.pi), O.qtt_cos(d, lam=math.pi)
    A4 = O.qtt_polynom([0.0, 2.0, 3.0, -8.0, -5.0], d, a=0.0, b=1.0)
    pol = 2 * xp + 3 * xp ** 2 - 8 * xp ** 3 - 5 * xp ** 4
    assert np.allclose(O.qtt_to_vector(O.hadamard_ttm(A2, A3)), np.cos(math.pi ** 2 * xp) * np.sin(math.pi ** 2 * xp), atol=1e-10, rtol=0)
    assert np.allclose(O.qtt_to_vector(O.hadamard_ttm(A1, A2)), np.exp(xp) * np.sin(math.pi ** 2 * xp), atol=1e-10, rtol=0)
    assert np.allclose(O.qtt_to_vector(O.hadamard_ttm(A4, A2)), pol * np.sin(math.pi ** 2 * xp), atol=1e-4, rtol=0)
    assert np.allclose(O.qtt_to_vector(O.hadamard_ttm(A4, A3)), pol * np.cos(math.pi ** 2 * xp), atol=1e-4, rtol=0)
    for a, b in ((A2, A3), (A1, A2), (A4, A2), (A4, A3)):
        h = O.hadamard(a, b)
        assert O.euclidean_distance(O.hadamard_ttm(a, b), h) / O.norm(h) < 1e-5


def test_reorder_round_trip_and_axis_permutation():
    """test/test_qtt_multidim.jl:182-199, 488-518: reorder preserves the function values (here: the dense tensor with
    permuted axes), round-trips, and preserves the norm."""
    rng = np.random.default_rng(3)
    for n_dims, bits in ((2, 3), (3, 3)):
        N = n_dims * bits
        x = O.rand_tt((2,) * N, 3, rng)
        dense = O.ttv_to_tensor(x)
        for thr in (0.0, 1e-14):
            il = O.reorder(x, n_dims, bits, True, threshold=thr)
            perm = O.reorder_perm(n_dims, bits, True)
            assert np.max(np.abs(O.ttv_to_tensor(il) - np.transpose(dense, np.argsort(perm)))) < 1e-10
            back = O.reorder(il, n_dims, bits, False, threshold=thr)
            assert np.max(np.abs(O.ttv_to_tensor(back) - dense)) < 1e-10
            assert abs(O.norm(il) - O.norm(x)) <= 1e-10 * O.norm(x)


def test_reorder_op_is_the_axis_permutation():
    """test/test_qtt_multidim.jl:694-722 checks reorder(A::QTToperator) through A*v on function values; the same statement
    on the dense operator: both index groups are permuted by reorder's site map, and the round trip is the identity."""
    rng = np.random.default_rng(4)
    n_dims, bits = 2, 3
    N = n_dims * bits
    A = O.rand_tto((2,) * N, 3, rng)
    dense = O.tto_to_tensor(A)
    inv = list(np.argsort(O.reorder_perm(n_dims, bits, True)))
    for thr in (0.0, 1e-14):
        B = O.reorder_op(A, n_dims, bits, True, threshold=thr)
        assert np.max(np.abs(O.tto_to_tensor(B) - np.transpose(dense, inv + [N + a for a in inv]))) < 1e-10 * np.max(np.abs(dense))
        C = O.reorder_op(B, n_dims, bits, False, threshold=thr)
        assert np.max(np.abs(O.tto_to_tensor(C) - dense)) < 1e-10 * np.max(np.abs(dense))


# ---- ttv_decomp ----------------------------------------------------------------------------------------------
def test_ttv_decomp_reference_cases():
    """test/test_tt_tools.jl:319-322 (index = 2: ot == [-1, 0, 1], reconstruction 1e-10), :1023-1028 (Bell state: the
    entanglement spectrum of the one bond is (1/2, 1/2))."""
    rng = np.random.default_rng(0)
    t = rng.standard_normal((2, 3, 2))
    tt = O.ttv_decomp(t, index=2)
    assert tt.ttv_ot == [-1, 0, 1]
    assert np.allclose(O.ttv_to_tensor(tt), t, atol=1e-10, rtol=0)
    bell = np.zeros((2, 2))
    bell[0, 0] = bell[1, 1] = 1 / math.sqrt(2)
    bt = O.ttv_decomp(bell)
    assert bt.ttv_rks == [1, 2, 1]
    sv = np.linalg.svd(bt.ttv_vec[0][:, 0, :], compute_uv=False)          # root core carries the Schmidt values
    assert np.allclose(sv ** 2, [0.5, 0.5])
    # exact-rank input: ranks are recovered, the gauge is as the docstring of the reference states
    x = O.rand_tt((2,) * 9, 3, rng)
    dense = O.ttv_to_tensor(x)
    for index in (1, 5, 9):
        tt = O.ttv_decomp(dense, index=index, tol=1e-10 * np.max(np.abs(dense)))
        assert tt.ttv_rks == x.ttv_rks
        assert np.max(np.abs(O.ttv_to_tensor(tt) - dense)) < 1e-12 * np.max(np.abs(dense))


# ---- als_linsolve ----------------------------------------------------------------------------------------------
def test_als_linsolve_reference_assertions():
    """test/test_als.jl:30-77: structure, residual < 0.5 for Δ + 10 I after 4 half sweeps, identity gives x ~ b (< 0.05),
    a single forward half sweep; plus: with full ranks two half sweeps solve the system exactly."""
    rng = np.random.default_rng(9999)

    def spd(d, shift):
        return O.tto_add(O.Delta(d), O.tto_scale(shift, O.id_tto(d)))

    def resid(A, x, b):
        return O.norm(O.sub(O.apply(A, x), b)) / max(O.norm(b), np.finfo(float).eps)

    dims, rks = (2, 2, 2), [1, 2, 2, 1]
    x = O.als_linsolve(O.rand_tto(dims, 3, rng), O.rand_tt(dims, rks, rng), O.rand_tt(dims, rks, rng))
    assert x.N == 3 and x.ttv_dims == dims and x.ttv_rks == rks
    d = 4
    A, b, x0 = spd(d, 10.0), O.rand_tt((2,) * d, [1, 2, 2, 2, 1], rng), O.rand_tt((2,) * d, [1, 2, 2, 2, 1], rng)
    assert resid(A, O.als_linsolve(A, b, x0, sweep_count=4), b) < 0.5
    A, b, x0 = O.id_tto(d), O.rand_tt((2,) * d, [1] * 5, rng), O.rand_tt((2,) * d, [1] * 5, rng)
    assert resid(A, O.als_linsolve(A, b, x0, sweep_count=4), b) < 0.05
    d = 3
    A, b, x0 = spd(d, 5.0), O.rand_tt((2,) * d, [1, 2, 2, 1], rng), O.rand_tt((2,) * d, [1, 2, 2, 1], rng)
    assert O.als_linsolve(A, b, x0, sweep_count=1).ttv_dims == b.ttv_dims
    d = 5
    A, b, x0 = spd(d, 3.0), O.rand_tt((2,) * d, 3, rng), O.rand_tt((2,) * d, [1, 2, 4, 4, 2, 1], rng)
    dense = np.linalg.solve(O.qtto_to_matrix(A), O.qtt_to_vector(b))
    assert np.max(np.abs(O.qtt_to_vector(O.als_linsolve(A, b, x0, sweep_count=2)) - dense)) <= 1e-11 * np.max(np.abs(dense))


def test_mals_linsolve_reference_assertions():
    """test/test_mals.jl:19-77 with NumPy inputs, plus exactness against the dense solve when the ranks allow it."""
    rng = np.random.default_rng(5678)

    def spd(d, shift):
        return O.tto_add(O.Delta(d), O.tto_scale(shift, O.id_tto(d)))

    def resid(A, x, b):
        return O.norm(O.sub(O.apply(A, x), b)) / max(O.norm(b), np.finfo(float).eps)

    d = 4
    b, x0 = O.rand_tt((2,) * d, [1, 2, 2, 2, 1], rng), O.rand_tt((2,) * d, [1, 2, 2, 2, 1], rng)
    x = O.mals_linsolve(spd(d, 3.0), b, x0)
    assert x.N == d and x.ttv_dims == (2,) * d
    assert resid(spd(d, 10.0), O.mals_linsolve(spd(d, 10.0), b, x0, tol=1e-10, rmax=8), b) < 0.5
    b1, x1 = O.rand_tt((2,) * d, [1] * 5, rng), O.rand_tt((2,) * d, [1] * 5, rng)
    assert resid(O.id_tto(d), O.mals_linsolve(O.id_tto(d), b1, x1, tol=1e-12, rmax=4), b1) < 0.05
    assert max(O.mals_linsolve(spd(d, 5.0), b, x1, tol=1e-10, rmax=4).ttv_rks) <= 4
    xl, xt = O.mals_linsolve(spd(d, 3.0), b, x0, tol=1e-2, rmax=8), O.mals_linsolve(spd(d, 3.0), b, x0, tol=0.0, rmax=8)
    assert max(xl.ttv_rks) <= max(xt.ttv_rks) + 2
    # hand evaluation of mals.jl:47-54: the tail is summed until its weight reaches tol * ||s||^2; the value that crossed is kept
    assert list(O.sv_trunc(np.array([3.0, 2.0, 1e-9, 1e-10]), 1e-12)) == [3.0, 2.0]
    assert list(O.sv_trunc(np.array([3.0, 2.0, 1.0]), 0.2)) == [3.0, 2.0]          # 1 < 2.8, 1 + 4 >= 2.8 -> i = 2 -> s[1:2]
    assert list(O.sv_trunc(np.array([3.0, 2.0, 1.0]), 0.0)) == [3.0, 2.0, 1.0]
    d = 6
    A, b, x0 = spd(d, 2.0), O.rand_tt((2,) * d, 2, rng), O.rand_tt((2,) * d, 2, rng)
    dense = np.linalg.solve(O.qtto_to_matrix(A), O.qtt_to_vector(b))
    assert np.max(np.abs(O.qtt_to_vector(O.mals_linsolve(A, b, x0, tol=1e-14, rmax=64)) - dense)) <= 1e-11 * np.max(np.abs(dense))


def test_dmrg_linsolve_reference_assertions():
    """test/test_dmrg.jl:20-75: the cut_off_index known answer and the N = 2 dmrg_linsolve cases (NumPy inputs, dense local
    solves), plus the sweep plan the reference's while-loop walks through and exactness against the dense solve."""
    s = np.array([1.0, 1.0 - 5.0e-11, 0.1])
    assert O.cut_off_index(s, (1.0 - 2.0e-11) / np.linalg.norm(s)) == 2          # test_dmrg.jl:20-25
    assert O.cut_off_index(np.array([3.0, 2.0, 1e-3, 1e-9]), 1e-6) == 3
    assert O.cut_off_index(np.array([1.0, 1e-11, 1e-12, 1e-13]), 1e-11 * 0.5) == 4   # the absolute 1e-10 window swallows the tail
    assert O.dmrg_sweep_plan([2], [4]) == ([4], 4)                               # default schedule: one sweep + the closing step
    assert O.dmrg_sweep_plan([2, 4], [2, 8]) == ([2, 8, 8], 8)
    assert O.dmrg_sweep_plan([1], [3]) == ([], 3)
    rng = np.random.default_rng(1234)

    def spd(d, shift):
        return O.tto_add(O.Delta(d), O.tto_scale(shift, O.id_tto(d)))

    def resid(A, x, b):
        return O.norm(O.sub(O.apply(A, x), b)) / max(O.norm(b), np.finfo(float).eps)

    d = 4
    b, x0 = O.rand_tt((2,) * d, [1, 2, 2, 2, 1], rng), O.rand_tt((2,) * d, [1, 2, 2, 2, 1], rng)
    x = O.dmrg_linsolve(spd(d, 3.0), b, x0, sweep_schedule=[2], rmax_schedule=[4])
    assert x.N == d and x.ttv_dims == (2,) * d and x.ttv_ot == [0, -1, -1, -1]
    assert resid(spd(d, 10.0), O.dmrg_linsolve(spd(d, 10.0), b, x0, sweep_schedule=[4], rmax_schedule=[8]), b) < 0.5
    x1 = O.rand_tt((2,) * d, [1] * 5, rng)
    x = O.dmrg_linsolve(spd(d, 5.0), b, x1, sweep_schedule=[2, 4], rmax_schedule=[2, 8])
    assert x.ttv_dims == b.ttv_dims
    b1 = O.rand_tt((2,) * d, [1] * 5, rng)
    assert resid(O.id_tto(d), O.dmrg_linsolve(O.id_tto(d), b1, x1, sweep_schedule=[4], rmax_schedule=[4]), b1) < 0.05
    d = 6
    A, b, x0 = spd(d, 2.0), O.rand_tt((2,) * d, 2, rng), O.rand_tt((2,) * d, 2, rng)
    dense = np.linalg.solve(O.qtto_to_matrix(A), O.qtt_to_vector(b))
    assert np.max(np.abs(O.qtt_to_vector(O.dmrg_linsolve(A, b, x0, tol=1e-14, sweep_schedule=[3], rmax_schedule=[64])) - dense)) <= 1e-11 * np.max(np.abs(dense))
    # the two-site schemes agree where both converge: mals after its sweep, dmrg after one sweep + closing step
    xm = O.mals_linsolve(A, b, x0, tol=1e-14, rmax=64)
    assert np.max(np.abs(O.qtt_to_vector(xm) - dense)) <= 1e-11 * np.max(np.abs(dense))


# ---- TDVP local contractions: the reference's own known-answer tests (explicit loop nests), test/test_tdvp.jl:58-135, :206-226 ----------
def _crandn(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def test_tdvp_layout_helpers_like_reference():
    rng = np.random.default_rng(42)
    A = rng.standard_normal((3, 4, 5))                                   # test_tdvp.jl:47-60
    assert O.tdvp_to_lsr(A).shape == (4, 3, 5)
    assert np.array_equal(O.tdvp_to_lsr(O.tdvp_to_lsr(A)), A)
    M = rng.standard_normal((2, 3, 4, 5))                                # (s_out, s_in, a, b)   :63-68
    M2 = O.tdvp_mpo_to_asbs(M)
    assert M2.shape == (4, 2, 5, 3) and np.array_equal(np.transpose(M2, (1, 3, 0, 2)), M)
    X, Y = _crandn(rng, 2, 3, 4), _crandn(rng, 2, 3, 4)                  # :70-76
    assert np.isclose(O.tdvp_dot3(X, Y), np.sum(np.conj(X.ravel(order="F")) * Y.ravel(order="F")), rtol=1e-12, atol=1e-12)


def test_tdvp_applyH1_known_answer():
    """test_tdvp.jl:78-97: explicit five-deep loop."""
    rng = np.random.default_rng(43)
    Dl, d_in, d_out, Dr, a, b = 2, 3, 3, 2, 2, 2
    AC, FL, FR, M = _crandn(rng, Dl, d_in, Dr), _crandn(rng, Dl, a, Dl), _crandn(rng, Dr, b, Dr), _crandn(rng, a, d_out, b, d_in)
    ref = np.zeros((Dl, d_out, Dr), dtype=complex)
    for al in range(Dl):
        for s in range(d_out):
            for be in range(Dr):
                z = 0.0
                for ap in range(Dl):
                    for sp in range(d_in):
                        for bp in range(Dr):
                            for ai in range(a):
                                for bi in range(b):
                                    z += FL[al, ai, ap] * AC[ap, sp, bp] * M[ai, s, bi, sp] * FR[bp, bi, be]
                ref[al, s, be] = z
    assert np.allclose(O.tdvp_applyH1_lsr(AC, FL, FR, M), ref, rtol=1e-12, atol=1e-12)


def test_tdvp_applyH0_known_answer():
    """test_tdvp.jl:99-117."""
    rng = np.random.default_rng(44)
    Dl, Dr, a = 3, 2, 4
    C, FL, FR = _crandn(rng, Dl, Dr), _crandn(rng, Dl, a, Dl), _crandn(rng, Dr, a, Dr)
    ref = np.zeros((Dl, Dr), dtype=complex)
    for al in range(Dl):
        for be in range(Dr):
            ref[al, be] = sum(FL[al, ai, ap] * C[ap, bp] * FR[bp, ai, be] for ap in range(Dl) for ai in range(a) for bp in range(Dr))
    assert np.allclose(O.tdvp_applyH0(C, FL, FR), ref, rtol=1e-12, atol=1e-12)


def test_tdvp_env_updates_shapes_and_loops():
    """test_tdvp.jl:119-135 asserts the shapes; the values are checked here against the index strings of tdvp.jl:37-43 written as loops."""
    rng = np.random.default_rng(45)
    Dl, d, Dr, a_in, a_out = 2, 3, 4, 2, 5
    A, FL, FR = _crandn(rng, Dl, d, Dr), _crandn(rng, Dl, a_in, Dl), _crandn(rng, Dr, a_in, Dr)
    M_L, M_R = _crandn(rng, a_in, d, a_out, d), _crandn(rng, a_out, d, a_in, d)
    FLn, FRp = O.tdvp_update_left_env(A, M_L, FL), O.tdvp_update_right_env(A, M_R, FR)
    assert FLn.shape == (Dr, a_out, Dr) and FRp.shape == (Dl, a_out, Dl)
    ref = np.zeros_like(FLn)
    for al in range(Dr):
        for ao in range(a_out):
            for be in range(Dr):
                ref[al, ao, be] = sum(FL[x, ap, y] * A[y, sp, be] * M_L[ap, s, ao, sp] * np.conj(A[x, s, al])
                                      for x in range(Dl) for ap in range(a_in) for y in range(Dl) for sp in range(d) for s in range(d))
    assert np.allclose(FLn, ref, rtol=1e-12, atol=1e-12)
    ref = np.zeros_like(FRp)
    for al in range(Dl):
        for ao in range(a_out):
            for be in range(Dl):
                ref[al, ao, be] = sum(A[al, sp, x] * FR[x, ap, y] * M_R[ao, s, ap, sp] * np.conj(A[be, s, y])
                                      for x in range(Dr) for ap in range(a_in) for y in range(Dr) for sp in range(d) for s in range(d))
    assert np.allclose(FRp, ref, rtol=1e-12, atol=1e-12)


def test_tdvp_applyH2_equals_two_site_loop():
    rng = np.random.default_rng(46)
    Dl, d1, d2, Dr, a, b, c = 2, 2, 3, 2, 2, 3, 2
    AAC, FL, FR = _crandn(rng, Dl, d1, d2, Dr), _crandn(rng, Dl, a, Dl), _crandn(rng, Dr, c, Dr)
    M1, M2 = _crandn(rng, a, d1, b, d1), _crandn(rng, b, d2, c, d2)
    ref = np.zeros((Dl, d1, d2, Dr), dtype=complex)
    for al in range(Dl):
        for s1 in range(d1):
            for s2 in range(d2):
                for be in range(Dr):
                    ref[al, s1, s2, be] = sum(FL[al, ai, ap] * AAC[ap, t1, t2, bp] * M1[ai, s1, bi, t1] * M2[bi, s2, ci, t2] * FR[bp, ci, be]
                                              for ai in range(a) for ap in range(Dl) for t1 in range(d1) for t2 in range(d2)
                                              for bp in range(Dr) for bi in range(b) for ci in range(c))
    assert np.allclose(O.tdvp_applyH2_lsr(AAC, FL, FR, M1, M2), ref, rtol=1e-12, atol=1e-12)


# --------------------------------------------------------------------------------------
# TDVP drivers (test/test_tdvp.jl:147-375): the reference's known answers, restated on the oracle's tdvp1sweep_ / tdvp2sweep_ / tdvp / tdvp2
# --------------------------------------------------------------------------------------
def _absnorm(x):
    return math.sqrt(max(float(np.real(O.dot(x, x))), 0.0))


def _tt_rel(a, b):
    """The reference measures absnorm(a - b) / absnorm(b) through the TT dot product of the difference train; that number is
    sqrt(rounding of ||a||^2 - 2 <a, b> + ||b||^2) ~ 1e-8 ||a|| whenever it does not round to exactly zero, so its bars (1e-12, 1e-10)
    are restated on the dense tensors (d <= 6 here), where they mean what they say."""
    ta, tb = O.ttv_to_tensor(a), O.ttv_to_tensor(b)
    return float(np.linalg.norm(ta - tb) / max(np.linalg.norm(tb), np.finfo(float).eps))


def _zero_id(d, cplx):
    H = O.tto_scale(0.0, O.id_tto(d))
    return O._tdvp_complex_op(H) if cplx else H


def test_tdvp1sweep_zero_hamiltonian_is_the_identity():
    d = 4                                                               # test/test_tdvp.jl:147-160
    psi = O._tdvp_complex(O.orthogonalize(O.qtt_sin(d, lam=math.pi)))
    psi2, F = O.tdvp1sweep_(complex(0.1), O.copy_tt(psi), _zero_id(d, True), None)
    assert _tt_rel(psi2, psi) < 1e-12
    assert len(F) == psi.N + 2


def test_tdvp_basic_behaviour():
    d = 4                                                               # test/test_tdvp.jl:164-206
    u0 = O.qtt_sin(d, lam=math.pi)
    kw = dict(normalize=False, sweeps=1, carry_env=False)
    psi_rt = O.tdvp(_zero_id(d, True), O._tdvp_complex(u0), [0.1], imaginary_time=False, **kw)
    assert np.iscomplexobj(psi_rt.ttv_vec[0])
    psi_it = O.tdvp(_zero_id(d, False), u0, [0.1], imaginary_time=True, **kw)
    assert not np.iscomplexobj(psi_it.ttv_vec[0])
    _, err = O.tdvp(_zero_id(d, True), O._tdvp_complex(u0), [0.1], imaginary_time=False, return_error=True, **kw)
    assert abs(err) <= 1e-6
    psi0 = O._tdvp_complex(O.orthogonalize(u0))
    psi_id = O.tdvp(_zero_id(d, True), psi0, [0.1], imaginary_time=False, **kw)
    assert _tt_rel(psi_id, psi0) <= 1e-10
    a = O.tdvp(_zero_id(d, True), O._tdvp_complex(u0), [0.1, 0.1], normalize=False, sweeps=2, carry_env=True, imaginary_time=False)
    b = O.tdvp(_zero_id(d, True), O._tdvp_complex(u0), [0.1, 0.1], normalize=False, sweeps=2, carry_env=False, imaginary_time=False)
    assert _tt_rel(a, b) <= 1e-10


def test_tdvp2sweep_zero_hamiltonian_and_max_bond():
    d = 4                                                               # test/test_tdvp.jl:236-268
    psi0 = O._tdvp_complex(O.orthogonalize(O.qtt_sin(d, lam=math.pi)))
    for dt in (0.1j, 0.05, 0.05j):
        psi1, F1 = O.tdvp2sweep_(dt, O.copy_tt(psi0), _zero_id(d, True), None)
        assert len(F1) == psi0.N + 2 and F1[0].shape == (1, 1, 1) and F1[-1].shape == (1, 1, 1)
        assert np.allclose(O.ttv_to_tensor(psi1), O.ttv_to_tensor(psi0), atol=1e-10, rtol=1e-10)
    d = 6
    psi0 = O._tdvp_complex(O.orthogonalize(O.add(O.qtt_sin(d, lam=math.pi), O.qtt_sin(d, lam=2 * math.pi))))
    psi2, _ = O.tdvp2sweep_(0.1j, O.copy_tt(psi0), _zero_id(d, True), None, max_bond=2, truncerr=0.0)
    assert max(psi2.ttv_rks) <= 2


def test_tdvp2_basic_behaviour():
    d = 6                                                               # test/test_tdvp.jl:270-317
    u0 = O.qtt_sin(d, lam=math.pi)
    kw = dict(normalize=False, sweeps=1, carry_env=False)
    assert np.iscomplexobj(O.tdvp2(_zero_id(d, True), O._tdvp_complex(u0), [0.1], imaginary_time=False, **kw).ttv_vec[0])
    assert not np.iscomplexobj(O.tdvp2(_zero_id(d, False), u0, [0.1], imaginary_time=True, **kw).ttv_vec[0])
    _, err = O.tdvp2(_zero_id(d, True), O._tdvp_complex(u0), [0.1], imaginary_time=False, return_error=True, **kw)
    assert abs(err) <= 1e-6
    psi0 = O._tdvp_complex(O.orthogonalize(u0))
    assert _tt_rel(O.tdvp2(_zero_id(d, True), psi0, [0.1], imaginary_time=False, **kw), psi0) <= 1e-7
    a = O.tdvp2(_zero_id(d, True), O._tdvp_complex(u0), [0.1, 0.1], normalize=False, sweeps=2, carry_env=True, imaginary_time=False)
    b = O.tdvp2(_zero_id(d, True), O._tdvp_complex(u0), [0.1, 0.1], normalize=False, sweeps=2, carry_env=False, imaginary_time=False)
    assert _tt_rel(a, b) <= 1e-10
    d = 4                                                               # test/test_tdvp.jl:319-327: the imaginary-time branch on a complex train
    psi0 = O._tdvp_complex(O.orthogonalize(O.qtt_sin(d, lam=math.pi)))
    psi_it = O.tdvp2(_zero_id(d, True), psi0, [0.02, 0.02], normalize=False, sweeps=2, carry_env=True, imaginary_time=True)
    assert _tt_rel(psi_it, psi0) < 1e-12


def heat_problem(d=4, kappa=0.1):
    """test/test_tdvp.jl:329-356: A = (kappa / h^2) (Delta ⊗ I + I ⊗ Delta) on 2 d bits in serial order, u0 = sin ⊗ sin on the
    interior grid, an eigenvector of A: exp(lambda t) u0 is the exact solution."""
    N = 2 ** d
    h = 1.0 / (N + 1)
    D1, I1 = O.toeplitz_to_qtto(-2.0, 1.0, 1.0, d), O.id_tto(d)
    kron = lambda X, Y: O.TToperator(2 * d, [c.copy() for c in X.tto_vec] + [c.copy() for c in Y.tto_vec], tuple(X.tto_dims) + tuple(Y.tto_dims),     # noqa: E731
                                     list(X.tto_rks) + list(Y.tto_rks[1:]), [0] * (2 * d))
    A = O.tto_scale(kappa / h ** 2, O.tto_add(kron(D1, I1), kron(I1, D1)))
    s1 = O.qtt_sin(d, a=h, b=1 - h)
    u0 = O.TTvector(2 * d, [c.copy() for c in s1.ttv_vec] * 2, tuple(s1.ttv_dims) * 2, list(s1.ttv_rks) + list(s1.ttv_rks[1:]), [0] * (2 * d))
    lam = float(np.real(O.dot(u0, O.apply(A, u0)) / O.dot(u0, u0)))
    return A, u0, lam


def test_tdvp_heat_eigenmode():
    A, u0, lam = heat_problem()
    steps = [1e-3] * 5
    target = math.exp(lam * sum(steps)) * O.ttv_to_tensor(u0)
    sol = O.tdvp(A, u0, steps, imaginary_time=True, normalize=False)
    assert np.linalg.norm(O.ttv_to_tensor(sol) - target) / np.linalg.norm(target) < 1e-8
    sol2 = O.tdvp2(A, u0, steps, imaginary_time=True, normalize=False, max_bond=8, truncerr=1e-12)
    assert np.linalg.norm(O.ttv_to_tensor(sol2) - target) / np.linalg.norm(target) < 1e-8


def test_tdvp_return_error_residual_both_time_directions():
    d = 4                                                               # test/test_tdvp.jl:358-375: A = I/2 -> every state evolves exactly
    A = O.tto_scale(0.5, O.id_tto(d))
    u0 = O.qtt_sin(d, lam=math.pi)
    for it in (False, True):
        _, e1 = O.tdvp(A, u0, [1e-3] * 5, imaginary_time=it, return_error=True, normalize=False)
        _, e2 = O.tdvp2(A, u0, [1e-3] * 5, imaginary_time=it, return_error=True, normalize=False, max_bond=8, truncerr=1e-12)
        assert e1 < 1e-3 and e2 < 1e-3
