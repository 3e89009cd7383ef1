"""GPU parity of the TDVP local contractions (src/solvers/tdvp.jl:29-43, :205-208): the HIP GEMM chains (k_tdvp, through
ttn_tdvp_contract_f64) against the oracle's einsum restatements, which tests/test_oracle_reference_pins.py pins to the reference's own
explicit-loop known answers (test/test_tdvp.jl:78-135).  Float64 and ComplexF64, single systems and batches, shared and per-system
operator cores, the reference's test shapes and TDVP-sized ones (bond 64..96, QTT Laplacian operator cores).
Tolerance: these are sums of at most Dl*a*d*Dr*b products with no cancellation structure: max |Δ| <= 1e-12 * max |ref| (the
reference's own tests use rtol = atol = 1e-12)."""
import numpy as np
import pytest

from oracle import tt_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import ttn_amd
    ttn_amd.ensure_init(0)
    return ttn_amd


def _rnd(rng, cplx, *shape):
    x = rng.standard_normal(shape)
    return x + 1j * rng.standard_normal(shape) if cplx else x


def _close(got, ref):
    assert got.shape == ref.shape and got.dtype == ref.dtype
    assert np.max(np.abs(got - ref)) <= 1e-12 * max(np.max(np.abs(ref)), 1e-300), np.max(np.abs(got - ref)) / np.max(np.abs(ref))


SHAPES = [(2, 3, 2, 2, 2), (3, 2, 4, 3, 2), (17, 2, 23, 3, 4), (64, 2, 64, 4, 4), (96, 3, 40, 2, 5), (1, 2, 8, 1, 3), (8, 2, 1, 3, 1)]


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("Dl,d,Dr,a,b", SHAPES)
def test_applyH1_vs_oracle(T, cplx, Dl, d, Dr, a, b):
    rng = np.random.default_rng(Dl * 100 + Dr + 7 * cplx)
    AC, FL, FR, M = _rnd(rng, cplx, Dl, d, Dr), _rnd(rng, cplx, Dl, a, Dl), _rnd(rng, cplx, Dr, b, Dr), _rnd(rng, cplx, a, d, b, d)
    _close(T.tdvp._applyH1_lsr(AC, FL, FR, M), O.tdvp_applyH1_lsr(AC, FL, FR, M))


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("Dl,Dr,a", [(3, 2, 4), (2, 3, 1), (40, 64, 4), (128, 96, 3), (1, 5, 2)])
def test_applyH0_vs_oracle(T, cplx, Dl, Dr, a):
    rng = np.random.default_rng(Dl * 10 + Dr + cplx)
    C, FL, FR = _rnd(rng, cplx, Dl, Dr), _rnd(rng, cplx, Dl, a, Dl), _rnd(rng, cplx, Dr, a, Dr)
    _close(T.tdvp._applyH0(C, FL, FR), O.tdvp_applyH0(C, FL, FR))


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("Dl,d,Dr,a_in,a_out", [(2, 3, 4, 2, 5), (4, 2, 3, 3, 2), (48, 2, 64, 4, 4), (64, 2, 33, 3, 3), (1, 2, 6, 1, 3)])
def test_env_updates_vs_oracle(T, cplx, Dl, d, Dr, a_in, a_out):
    rng = np.random.default_rng(Dl + 31 * Dr + cplx)
    A, FL, FR = _rnd(rng, cplx, Dl, d, Dr), _rnd(rng, cplx, Dl, a_in, Dl), _rnd(rng, cplx, Dr, a_in, Dr)
    M_L, M_R = _rnd(rng, cplx, a_in, d, a_out, d), _rnd(rng, cplx, a_out, d, a_in, d)
    FLn = T.tdvp._update_left_env(A, M_L, FL)
    FRp = T.tdvp._update_right_env(A, M_R, FR)
    assert FLn.shape == (Dr, a_out, Dr) and FRp.shape == (Dl, a_out, Dl)            # test/test_tdvp.jl:133-134
    _close(FLn, O.tdvp_update_left_env(A, M_L, FL))
    _close(FRp, O.tdvp_update_right_env(A, M_R, FR))


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("Dl,d1,d2,Dr,a,b,c", [(2, 2, 3, 2, 2, 3, 2), (16, 2, 2, 24, 3, 3, 3), (64, 2, 2, 64, 4, 4, 4), (5, 3, 2, 1, 1, 2, 1)])
def test_applyH2_vs_oracle(T, cplx, Dl, d1, d2, Dr, a, b, c):
    rng = np.random.default_rng(Dl + Dr + 5 * cplx)
    AAC, FL, FR = _rnd(rng, cplx, Dl, d1, d2, Dr), _rnd(rng, cplx, Dl, a, Dl), _rnd(rng, cplx, Dr, c, Dr)
    M1, M2 = _rnd(rng, cplx, a, d1, b, d1), _rnd(rng, cplx, b, d2, c, d2)
    _close(T.tdvp._applyH2_lsr(AAC, FL, FR, M1, M2), O.tdvp_applyH2_lsr(AAC, FL, FR, M1, M2))


@pytest.mark.parametrize("cplx", [False, True])
def test_batched_contractions_shared_and_per_system_operator(T, cplx):
    rng = np.random.default_rng(11 + cplx)
    B, Dl, d, Dr, a, b = 5, 24, 2, 32, 3, 3
    AC, FL, FR = _rnd(rng, cplx, B, Dl, d, Dr), _rnd(rng, cplx, B, Dl, a, Dl), _rnd(rng, cplx, B, Dr, b, Dr)
    Msh, Mb = _rnd(rng, cplx, a, d, b, d), _rnd(rng, cplx, B, a, d, b, d)
    got_sh, got_b = T.tdvp._applyH1_lsr(AC, FL, FR, Msh), T.tdvp._applyH1_lsr(AC, FL, FR, Mb)
    for i in range(B):
        _close(got_sh[i], O.tdvp_applyH1_lsr(AC[i], FL[i], FR[i], Msh))
        _close(got_b[i], O.tdvp_applyH1_lsr(AC[i], FL[i], FR[i], Mb[i]))
    C = _rnd(rng, cplx, B, Dl, Dr)
    FRa = _rnd(rng, cplx, B, Dr, a, Dr)
    got = T.tdvp._applyH0(C, FL, FRa)
    for i in range(B):
        _close(got[i], O.tdvp_applyH0(C[i], FL[i], FRa[i]))


def test_energy_expectation_through_the_environments(T):
    """The use the reference makes of these pieces (tdvp.jl:52-66, :74-78): with left-orthonormal sites to the left and
    right-orthonormal ones to the right, <AC| H1 |AC> computed through the environments equals <ψ|H|ψ> — here for the QTT Laplacian
    (operator cores in (a, s, b, s') layout via _mpo_to_asbs) and a rank-24 random train, environments built by the device updates."""
    d, r = 10, 24
    rng = np.random.default_rng(3)
    psi = O.orthogonalize(O.rand_tt((2,) * d, r, rng), i=1)                      # site 1 is the centre, the rest right-orthonormal
    H = O.Delta(d)
    A = [O.tdvp_to_lsr(c) for c in psi.ttv_vec]
    M = [O.tdvp_mpo_to_asbs(c) for c in H.tto_vec]
    F = [None] * (d + 2)
    F[0] = np.ones((1, 1, 1))
    F[d + 1] = np.ones((1, 1, 1))
    for k in range(d - 1, -1, -1):                                               # F[k+1] = _update_right_env(A[k], M[k], F[k+2])
        F[k + 1] = T.tdvp._update_right_env(A[k], M[k], F[k + 2])
    HAC = T.tdvp._applyH1_lsr(A[0], F[0], F[2], M[0])
    e_env = float(np.real(O.tdvp_dot3(A[0], HAC)))
    e_ref = O.dot(psi, O.apply(H, psi))
    assert abs(e_env - e_ref) <= 1e-11 * abs(e_ref)
