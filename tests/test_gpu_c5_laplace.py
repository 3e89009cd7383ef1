"""BASELINE config C5 on the GPU: als_linsolve / mals_linsolve / two-site dmrg_linsolve on the 2D Laplace problem of
examples/Laplace_pde.jl:12-27 at its real size — d = 12 bits per dimension, serial QTT ordering, 24 sites — HIP path vs the
CPU oracle, at the largest ranks the dense local solver takes (local systems <= 2048 unknowns: ALS rank 32, two-site rank 22).

Operator: A = (1/h^2) (Δ1d ⊗ I + I ⊗ Δ1d) with Δ1d = toeplitz_to_qtto(-2, 1, 1, d) (ranks 4 inside each half, 2 at the junction);
right-hand side b = -(1/h^2) qtt_sin(d; a = h, b = 1 - h) ⊗ e_1; start trains from NumPy's generator (Julia's `rand_tt` stream is
not reproducible here).  cond(A) ~ (2^12 + 1)^2 = 1.7e7, so the local systems amplify rounding by up to that factor: iterates are
compared as tensors to 1e-8 wherever the local solutions are full rank; where the ranks over-parametrise the solution (ALS at rank
32: the local solution is numerically rank deficient and the QR core move completes it arbitrarily — DESIGN.md §4.7) the iterate
is gauge-path dependent in the reference itself and only ranks, gauge flags and the residual level are asserted (parity unpinned).
"""
import math

import numpy as np
import pytest

from oracle import tt_oracle as O
from tests.helpers import to_oracle, to_product, tt_norm_stable, tt_rel_diff

pytestmark = pytest.mark.gpu

BITS = 12


@pytest.fixture(scope="module")
def T():
    import ttn_amd
    ttn_amd.ensure_init(0)
    return ttn_amd


def _kron_op(A, B):       # kron of TT operators: the cores side by side (src/tt_operations.jl:427-433)
    return O.TToperator(A.N + B.N, list(A.tto_vec) + list(B.tto_vec), tuple(A.tto_dims) + tuple(B.tto_dims),
                        list(A.tto_rks[:-1]) + list(B.tto_rks), [0] * (A.N + B.N))


def _kron_vec(a, b):
    return O.TTvector(a.N + b.N, list(a.ttv_vec) + list(b.ttv_vec), tuple(a.ttv_dims) + tuple(b.ttv_dims),
                      list(a.ttv_rks[:-1]) + list(b.ttv_rks), [0] * (a.N + b.N))


@pytest.fixture(scope="module")
def problem():
    d = BITS
    h = 1.0 / (2 ** d + 1)
    L1 = O.toeplitz_to_qtto(-2.0, 1.0, 1.0, d)
    A = O.tto_scale(1 / h ** 2, O.tto_add(_kron_op(L1, O.id_tto(d)), _kron_op(O.id_tto(d), L1)))
    e1 = O.TTvector(d, [np.array([[[1.0]], [[0.0]]]) for _ in range(d)], (2,) * d, [1] * (d + 1), [0] * d)     # qtt_basis_vector(d, 1)
    b = O.scale(-1 / h ** 2, _kron_vec(O.qtt_sin(d, a=h, b=1 - h, lam=1.0 / math.pi), e1))
    assert A.N == 2 * d and max(A.tto_rks) == 4 and A.tto_rks[d] == 2
    return A, b


def _resid(A, x, b):
    return tt_norm_stable(O.sub(O.apply(A, x), b)) / tt_norm_stable(b)


def test_c5_als_at_rhs_ranks(T, problem):
    """als_linsolve with the start ranks of examples/Laplace_pde.jl:24 (those of b): the ranks can carry the solution, the local
    solutions are full rank, the iterate is well defined."""
    A, b = problem
    rng = np.random.default_rng(5)
    x0 = O.rand_tt((2,) * A.N, b.ttv_rks, rng)
    ref = O.als_linsolve(A, b, x0, sweep_count=4)
    got = T.solvers.als_linsolve(to_product(A), to_product(b), to_product(x0), sweep_count=4)
    assert list(got.ttv_rks) == list(ref.ttv_rks) and list(got.ttv_ot) == list(ref.ttv_ot)
    rg, rr = _resid(A, to_oracle(got), b), _resid(A, ref, b)
    assert tt_rel_diff(to_oracle(got), ref) <= 1e-8
    assert abs(rg - rr) <= 1e-6 * max(rr, 1e-12) + 1e-9, (rg, rr)


def test_c5_als_rank32_largest_dense_system(T, problem):
    """Rank 32: local systems of 2 * 32 * 32 = 2048 unknowns, the device limit of the dense LU.  Over-parametrised (the solution
    has rank 2): iterate parity is unpinned (module docstring); ranks, gauge flags and the residual LEVEL are asserted."""
    A, b = problem
    rng = np.random.default_rng(6)
    x0 = O.rand_tt((2,) * A.N, 32, rng)
    assert max(2 * x0.ttv_rks[i] * x0.ttv_rks[i + 1] for i in range(A.N)) == 2048
    ref = O.als_linsolve(A, b, x0, sweep_count=2)
    got = T.solvers.als_linsolve(to_product(A), to_product(b), to_product(x0), sweep_count=2)
    assert list(got.ttv_rks) == list(ref.ttv_rks) and list(got.ttv_ot) == list(ref.ttv_ot)
    rg, rr = _resid(A, to_oracle(got), b), _resid(A, ref, b)
    assert np.isfinite(rg) and rg <= 10.0 * rr + 1e-6, (rg, rr)


def test_c5_als_rank64_grid_form(T, problem):
    """Rank 64 on the C5 problem: one-site systems of 2 * 64 * 64 = 8192 unknowns (a 0.5 GB K per site), four times the limit of the
    one-workgroup LU — the grid form (csrc/ttn_als_grid.h: assembly and blocked LU with partial pivoting on the whole chip).  The
    oracle needs ~10 min of CPU for this run, so only oracle-independent facts are asserted: the ranks stay those of the start train
    (als.jl:177), the gauge flags are the sweep's, and the residual after two half sweeps is at the level the rank-32 run reaches
    against its oracle (3e-3; the start train's residual is O(1)).  The iterate itself is pinned at a size the oracle can take:
    tests/test_gpu_als.py::test_als_rank40_beyond_the_one_workgroup_limit (3200 unknowns, tensor 1e-9)."""
    A, b = problem
    rng = np.random.default_rng(6)
    x0 = O.rand_tt((2,) * A.N, 64, rng)
    assert max(2 * x0.ttv_rks[i] * x0.ttv_rks[i + 1] for i in range(A.N)) == 8192
    got = T.solvers.als_linsolve(to_product(A), to_product(b), to_product(x0), sweep_count=2)
    assert list(got.ttv_rks) == list(x0.ttv_rks)
    assert list(got.ttv_ot) == [0] + [1] * (A.N - 1)                                    # forward, then backward half sweep: the centre is site 1 (als.jl:112-118)
    rg = _resid(A, to_oracle(got), b)
    print(f"C5 als rank 64 (grid form): residual {rg:.3e}")
    assert np.isfinite(rg) and rg <= 2e-2


def test_c5_mals(T, problem):
    """mals_linsolve(A, b, x0) as examples/Laplace_pde.jl:26 calls it (one sweep, tol 1e-12), rmax 22: two-site systems up to
    4 * 22 * 22 = 1936 unknowns."""
    A, b = problem
    rng = np.random.default_rng(7)
    x0 = O.rand_tt((2,) * A.N, b.ttv_rks, rng)
    ref = O.mals_linsolve(A, b, x0, tol=1e-12, rmax=22)
    got = T.solvers.mals_linsolve(to_product(A), to_product(b), to_product(x0), tol=1e-12, rmax=22)
    assert list(got.ttv_rks) == list(ref.ttv_rks), (got.ttv_rks, ref.ttv_rks)
    rg, rr = _resid(A, to_oracle(got), b), _resid(A, ref, b)
    assert tt_rel_diff(to_oracle(got), ref) <= 1e-8
    assert abs(rg - rr) <= 1e-6 * max(rr, 1e-12) + 1e-9, (rg, rr)


@pytest.mark.parametrize("sched,rmaxs", [([2], [16]), ([4], [22]), ([8], [22])])
def test_c5_dmrg_two_site(T, problem, sched, rmaxs):
    """dmrg_linsolve(A, b, x0; N = 2, tol) as examples/Laplace_pde.jl:27, local systems solved densely on both sides (the reference's
    it_solver = false branch, dmrg.jl:173-175)."""
    A, b = problem
    rng = np.random.default_rng(8)
    x0 = O.rand_tt((2,) * A.N, b.ttv_rks, rng)
    ref = O.dmrg_linsolve(A, b, x0, tol=1e-10, sweep_schedule=sched, rmax_schedule=rmaxs)
    got = T.solvers.dmrg_linsolve(to_product(A), to_product(b), to_product(x0), tol=1e-10, sweep_schedule=sched, rmax_schedule=rmaxs)
    assert list(got.ttv_rks) == list(ref.ttv_rks), (got.ttv_rks, ref.ttv_rks)
    assert list(got.ttv_ot) == list(ref.ttv_ot)
    rg, rr = _resid(A, to_oracle(got), b), _resid(A, ref, b)
    assert tt_rel_diff(to_oracle(got), ref) <= 1e-8
    assert abs(rg - rr) <= 1e-6 * max(rr, 1e-12) + 1e-9, (rg, rr)


def _golden_c5(rank):
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"c5_matrix_free_rank{rank}.json")
    with open(path) as fh:
        return json.load(fh)


@pytest.mark.parametrize("rank", [32, 64, 128])
def test_c5_dmrg_matrix_free_reference_defaults(T, problem, rank):
    """dmrg_linsolve with the REFERENCE's default local solver (it_solver = true, linsolv_maxiter = 200, linsolv_tol =
    max(sqrt(tol), 1e-8); dmrg.jl:392-395) from a random rank-`rank` start train: two-site systems of 4 * rank^2 = 4096 / 16384 /
    65536 unknowns (rank 128 = BASELINE config C5's stated rank bound) — beyond the dense path — solved matrix-free by conjugate
    gradients on the device (wg_cg_two_site).  KrylovKit's algorithm selector runs CG(maxiter = krylovdim * linsolv_maxiter =
    6000) for this call (the convention src/solvers/euler.jl:29 spells out; round 2 capped the iteration at 200 on both sides and
    every local solve ended unconverged): with the real cap the local solves CONVERGE to linsolv_tol, and device and oracle
    agree — same ranks, same residual to four digits, iterate 3e-11 at rank 32 (measured, round 3).
    Oracle: run live at rank 32 (5 s); at ranks 64 / 128 its result comes from tests/golden/c5_matrix_free_rank<R>.json
    (tests/golden/make_c5_golden.py: 1 min / 40 min of CPU), same seed.  Asserted: final ranks (exact at ranks 32 / 64, within one per bond at rank 128) and gauge flags exact, total CG
    iterations within 10 % (the iteration count of a cond-1.7e7 solve is sensitive to the summation order), residual after the
    sweep equal to 1e-3 relative; at rank 32 also the iterate as a tensor (1e-8)."""
    A, b = problem
    rng = np.random.default_rng(9)
    x0 = O.rand_tt((2,) * A.N, rank, rng)
    kw = dict(tol=1e-10, sweep_schedule=[2], rmax_schedule=[rank], it_solver=True)
    if rank == 32:
        st = {}
        ref = O.dmrg_linsolve(A, b, x0, stats=st, **kw)
        gold = {"cg_iterations": st["cg_iterations"], "residual": _resid(A, ref, b), "ranks": list(ref.ttv_rks), "ot": list(ref.ttv_ot)}
        live = _golden_c5(32)                      # the committed fixture must be what the oracle produces today
        assert live["ranks"] == gold["ranks"] and abs(live["cg_iterations"] - gold["cg_iterations"]) <= 0.02 * gold["cg_iterations"]
    else:
        ref, gold = None, _golden_c5(rank)
    got = T.solvers.dmrg_linsolve(to_product(A), to_product(b), to_product(x0), **kw)
    iters = T.solvers.dmrg_cg_iterations(1)[0]
    rg, rr = _resid(A, to_oracle(got), b), gold["residual"]
    print(f"C5 matrix-free rank {rank}: CG iterations device {iters} / oracle {gold['cg_iterations']}, residual device {rg:.3e} / oracle {rr:.3e}, "
          f"ranks device {list(got.ttv_rks)} / oracle {gold['ranks']}")
    assert abs(iters - gold["cg_iterations"]) <= 0.1 * gold["cg_iterations"]
    assert np.isfinite(rg) and abs(rg - rr) <= 1e-3 * rr, (rg, rr)
    assert list(got.ttv_ot) == gold["ot"]
    if rank <= 64:
        assert list(got.ttv_rks) == gold["ranks"], (list(got.ttv_rks), gold["ranks"])
    else:
        # rank 128: cut_off_index cuts at tol * ||s|| = 1e-10 ||s||, inside the rounding noise of the cond-1.7e7 local solves of the
        # 65 536-unknown windows — measured: two bonds keep one singular value more than the oracle (6 against 5), everything
        # else equal, residual equal to four digits.  A rank decision inside the noise is parity-unpinned: bound it by one per bond.
        assert all(abs(a - c) <= 1 for a, c in zip(got.ttv_rks, gold["ranks"])), (list(got.ttv_rks), gold["ranks"])
    if ref is not None:
        err = tt_rel_diff(to_oracle(got), ref)
        print(f"C5 matrix-free rank 32: iterate rel. difference device vs oracle {err:.2e}")
        assert err <= 1e-8


def test_dmrg_default_rmax_schedule_is_clamped_not_refused(T, problem):
    """dmrg_linsolve called with the reference's DEFAULT rmax_schedule = isqrt(prod(dims)) (4096 for the 24-site C5 problem,
    dmrg.jl:391): the device bound n * rank <= 256 clamps the capacity (ranks saturate at 128) instead of refusing the call
    (round-2 regression: every 2^d problem with d >= 16 threw).  The solution has rank ~10, so nothing saturates here and
    the result equals the run with an explicit rmax of 128."""
    A, b = problem
    cap = T.solvers.dmrg_capacity((2,) * A.N, b.ttv_rks, 4096)
    assert max(cap) == 128 and all(2 * c <= 256 for c in cap)
    with pytest.raises(T._lib.TTNError):
        T.solvers.dmrg_capacity((2,) * A.N, [1] + [200] * (A.N - 1) + [1], 4096)      # START ranks beyond the bound are refused
    rng = np.random.default_rng(8)
    x0 = O.rand_tt((2,) * A.N, b.ttv_rks, rng)
    got = T.solvers.dmrg_linsolve(to_product(A), to_product(b), to_product(x0), tol=1e-10, it_solver=True)        # default schedules
    ref = T.solvers.dmrg_linsolve(to_product(A), to_product(b), to_product(x0), tol=1e-10, it_solver=True, rmax_schedule=[128])
    assert list(got.ttv_rks) == list(ref.ttv_rks)
    assert tt_rel_diff(to_oracle(got), to_oracle(ref)) <= 1e-12
    assert _resid(A, to_oracle(got), b) <= 5e-2
