"""GPU parity of the site-swap chains: hadamard_ttm (src/tt_operations.jl:363-422) and QTT reorder
(src/qtt_tools.jl:660-775), HIP path through the C ABI vs the CPU oracle and the reference's own known answers.

Tolerances (fp64): swap lists / op lists / rank bookkeeping bit-exact; values: the cores of a swap SVD are gauge dependent
(and the reference keeps numerically-zero singular directions when threshold == 0), so trains are compared as tensors:
max |T_gpu - T_cpu| <= 1e-11 * max |T_cpu| (random trains), and the reference's own atol for its known-answer cases.
"""
import math

import numpy as np
import pytest

from oracle import tt_oracle as O
from tests.helpers import to_oracle, to_product

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import ttn_amd
    ttn_amd.ensure_init(0)
    return ttn_amd


def _dense(x):
    return O.ttv_to_tensor(to_oracle(x) if not isinstance(x, O.TTvector) else x)


def _close(a, b, rel):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    assert np.max(np.abs(a - b)) <= rel * np.max(np.abs(b)), (np.max(np.abs(a - b)), np.max(np.abs(b)))


# ---- hadamard_ttm ----------------------------------------------------------------------------------------------------
def test_hadamard_ttm_reference_known_answers(T):
    """test/test_tt_operations.jl:72-99 ("Hadamard TTM algorithm vs naive"), same inputs, same tolerances."""
    d = 8
    xp = np.linspace(0, 1, 2 ** d)
    A1, A2, A3 = T.qtt_exp(d), T.qtt_sin(d, lam=math.pi), T.qtt_cos(d, lam=math.pi)
    A4 = to_product(O.qtt_polynom([0.0, 2.0, 3.0, -8.0, -5.0], d, a=0.0, b=1.0))
    pol = 2 * xp + 3 * xp ** 2 - 8 * xp ** 3 - 5 * xp ** 4
    cases = [(A2, A3, np.cos(math.pi ** 2 * xp) * np.sin(math.pi ** 2 * xp), 1e-10),
             (A1, A2, np.exp(xp) * np.sin(math.pi ** 2 * xp), 1e-10),
             (A4, A2, pol * np.sin(math.pi ** 2 * xp), 1e-4),
             (A4, A3, pol * np.cos(math.pi ** 2 * xp), 1e-4)]
    for a, b, expected, atol in cases:
        z = T.hadamard_ttm(a, b)
        assert np.allclose(T.qtt_to_vector(z), expected, atol=atol, rtol=0)
        h = T.hadamard(a, b)
        assert T.euclidean_distance(z, h) / T.norm(h) < 1e-5
        # and against the oracle's hadamard_ttm on the same inputs
        zo = O.hadamard_ttm(to_oracle(a), to_oracle(b))
        assert np.max(np.abs(T.qtt_to_vector(z) - O.qtt_to_vector(zo))) <= 1e-12 * np.max(np.abs(expected))


@pytest.mark.parametrize("d,rx,ry,seed", [(6, 2, 3, 0), (8, 3, 3, 1), (10, 4, 2, 2), (5, 1, 4, 3), (7, 4, 4, 4), (2, 2, 2, 5), (1, 1, 1, 6)])
def test_hadamard_ttm_random_vs_oracle(T, d, rx, ry, seed):
    rng = np.random.default_rng(seed)
    x = O.rand_tt((2,) * d, rx, rng)
    y = O.rand_tt((2,) * d, ry, rng)
    # tol = 1e-10: the spectra have a gap (true singular values >~ 1e-3 sigma_1, rounding noise <~ 1e-12 sigma_1) -> exact ranks
    ref = O.hadamard_ttm(x, y, tol=1e-10)
    got = T.qtt.hadamard_ttm(to_product(x), to_product(y), tol=1e-10)
    assert list(got.ttv_rks) == list(ref.ttv_rks)
    assert list(got.ttv_ot) == [0] * d
    _close(_dense(got), _dense(ref), 1e-9)
    _close(_dense(got), _dense(x) * _dense(y), 1e-9)
    # default tol = 1e-14 sits INSIDE the rounding noise the earlier swaps leave behind (the reference itself keeps directions
    # at 1e-13 sigma_1 and drops others at 1e-14): its ranks are noise dependent — parity unpinned — so only values are compared
    got = T.hadamard_ttm(to_product(x), to_product(y))
    _close(_dense(got), _dense(x) * _dense(y), 1e-11)
    _close(_dense(got), _dense(O.hadamard_ttm(x, y)), 1e-11)


def test_hadamard_ttm_rmax_and_tol(T):
    rng = np.random.default_rng(11)
    d = 8
    x = O.rand_tt((2,) * d, 3, rng)
    y = O.rand_tt((2,) * d, 3, rng)
    for tol, rmax in ((1e-10, 4), (1e-3, 2 ** 62), (0.0, 6)):
        ref = O.hadamard_ttm(x, y, tol=tol, rmax=rmax)
        got = T.qtt.hadamard_ttm(to_product(x), to_product(y), tol=tol, rmax=rmax)
        assert list(got.ttv_rks) == list(ref.ttv_rks)
        _close(_dense(got), _dense(ref), 1e-10)


def test_hadamard_ttm_three_level_dims_and_batch(T):
    """n = 3 sites, and a batch of different train pairs in one launch."""
    rng = np.random.default_rng(5)
    d, B = 5, 6
    dims = (3,) * d
    xs = [O.rand_tt(dims, 2 + (b % 2), rng) for b in range(B)]
    ys = [O.rand_tt(dims, 2, rng) for b in range(B)]
    capx = [max(x.ttv_rks[k] for x in xs) for k in range(d + 1)]
    capy = [max(y.ttv_rks[k] for y in ys) for k in range(d + 1)]
    dx, dy = T.DeviceTT(dims, capx, batch=B), T.DeviceTT(dims, capy, batch=B)
    for b in range(B):
        dx.upload(b, to_product(xs[b]))
        dy.upload(b, to_product(ys[b]))
    dz = T.DeviceTT(dims, [1] + [64] * (d - 1) + [1], batch=B)
    T.qtt.hadamard_ttm_(dx, dy, dz, tol=1e-10, work_cap=64)
    T.device.compress_status(dz)
    for b in range(B):
        ref = O.hadamard_ttm(xs[b], ys[b], tol=1e-10)
        got = dz.download(b)
        assert list(got.ttv_rks) == list(ref.ttv_rks)
        _close(_dense(got), _dense(ref), 1e-9)


def test_hadamard_ttm_errors(T):
    rng = np.random.default_rng(2)
    x = to_product(O.rand_tt((2,) * 6, 3, rng))
    y = to_product(O.rand_tt((2,) * 6, 3, rng))
    dx, dy = T.DeviceTT.from_host(x), T.DeviceTT.from_host(y)
    dz = T.DeviceTT((2,) * 6, [1, 2, 2, 2, 2, 2, 1])               # too small for ranks up to 8
    T.qtt.hadamard_ttm_(dx, dy, dz, work_cap=16)
    with pytest.raises(T.TTNError):
        T.device.compress_status(dz)
    dz2 = T.DeviceTT((2,) * 6, [1, 16, 16, 16, 16, 16, 1])
    with pytest.raises(T.TTNError):                                 # work_cap below an input rank
        T.qtt.hadamard_ttm_(dx, dy, dz2, work_cap=2)
    with pytest.raises(T.TTNError):                                 # n * work_cap > 256
        T.qtt.hadamard_ttm_(dx, dy, dz2, work_cap=256)
    w = T.DeviceTT((2, 3, 2, 2, 2, 2), [1, 2, 2, 2, 2, 2, 1])
    with pytest.raises(T.TTNError):
        T.qtt.hadamard_ttm_(w, w, T.DeviceTT((2, 3, 2, 2, 2, 2), [1, 2, 2, 2, 2, 2, 1]))


# ---- reorder ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_dims,bits,r,seed", [(2, 3, 3, 0), (3, 3, 2, 1), (2, 5, 4, 2), (4, 2, 3, 3), (2, 6, 8, 4)])
def test_reorder_is_the_axis_permutation(T, n_dims, bits, r, seed):
    """Size-independent property (and what test/test_qtt_multidim.jl:182-199, 488-518 check through function values): the
    reordered train is the source tensor with its axes permuted by reorder's site map; the round trip is the identity;
    the norm is preserved."""
    rng = np.random.default_rng(seed)
    N = n_dims * bits
    x = O.rand_tt((2,) * N, r, rng)
    dense = O.ttv_to_tensor(x)
    for threshold in (0.0, 1e-14):
        il = T.reorder(to_product(x), n_dims, bits, "interleaved", threshold=threshold)
        perm = O.reorder_perm(n_dims, bits, True)
        _close(_dense(il), np.transpose(dense, np.argsort(perm)), 1e-11)
        ref = O.reorder(x, n_dims, bits, True, threshold=threshold)
        assert list(il.ttv_rks) == list(ref.ttv_rks)
        _close(_dense(il), _dense(ref), 1e-11)
        back = T.reorder(il, n_dims, bits, "serial", threshold=threshold)
        _close(_dense(back), dense, 1e-11)
        assert abs(T.norm(il) - O.norm(x)) <= 1e-10 * O.norm(x)


def test_reorder_reference_function_case(T):
    """test/test_qtt_multidim.jl:182-199: f(x) = sin(pi x1) cos(pi x2) on 2 x 3 bits, serial <-> interleaved (the reference
    builds the QTT by ttv_decomp of the sampled tensor; here the sampled tensor is compressed by the oracle's TT-SVD)."""
    bits = 3
    g = np.arange(2 ** bits) / 2 ** bits
    F = np.sin(np.pi * g)[:, None] * np.cos(np.pi * g)[None, :]
    # serial QTT: sites = bits of x1 (most significant first), then bits of x2; cores from exact rank-1 x rank-1 structure
    Tser = F.reshape((2,) * (2 * bits))
    cores, rks = [], [1]
    cur = Tser.reshape(1, -1)
    for k in range(2 * bits - 1):
        cur = cur.reshape(rks[-1] * 2, -1)
        U, s, Vt = np.linalg.svd(cur, full_matrices=False)
        rr = int(np.count_nonzero(s > 1e-13 * s[0]))
        cores.append(U[:, :rr].reshape(rks[-1], 2, rr).transpose(1, 0, 2).copy())
        cur = s[:rr, None] * Vt[:rr]
        rks.append(rr)
    cores.append(cur.reshape(rks[-1], 2, 1).transpose(1, 0, 2).copy())
    rks.append(1)
    x = O.TTvector(2 * bits, cores, (2,) * (2 * bits), rks, [0] * (2 * bits))
    assert np.max(np.abs(O.ttv_to_tensor(x) - Tser)) < 1e-13
    il = T.reorder(to_product(x), 2, bits, "interleaved")
    back = T.reorder(il, 2, bits, "serial")
    perm = O.reorder_perm(2, bits, True)
    assert np.max(np.abs(_dense(il) - np.transpose(Tser, np.argsort(perm)))) < 1e-10
    assert np.max(np.abs(_dense(back) - Tser)) < 1e-10


def test_swap_sites_batch_and_errors(T):
    rng = np.random.default_rng(9)
    d, B = 8, 5
    xs = [O.rand_tt((2,) * d, 3, rng) for _ in range(B)]
    cap = [1, 2, 4, 8, 16, 8, 4, 2, 1]
    dx = T.DeviceTT((2,) * d, cap, batch=B)
    for b in range(B):
        dx.upload(b, to_product(xs[b]))
    swaps = [3, 4, 2, 5, 3, 1, 7]
    T.qtt.swap_sites_(dx, swaps, 0.0)
    T.device.compress_status(dx)
    for b in range(B):
        ref = O.swap_sites_(O.copy_tt(xs[b]), swaps, 0.0)
        got = dx.download(b)
        assert list(got.ttv_rks) == list(ref.ttv_rks)
        _close(_dense(got), _dense(ref), 1e-11)
    with pytest.raises(AssertionError):
        T.qtt.swap_sites_(dx, [8], 0.0)                     # k must be in 1:(N-1)
    small = T.DeviceTT.from_host(to_product(xs[0]))          # capacity = current ranks: threshold 0 must grow a bond
    with pytest.raises(T.TTNError):
        T.qtt.swap_sites_(small, [4], 0.0)


@pytest.mark.parametrize("n_dims,bits,R,seed", [(2, 3, 3, 0), (3, 2, 2, 1), (2, 4, 2, 2)])
def test_reorder_op_vs_oracle(T, n_dims, bits, R, seed):
    """reorder(A::QTToperator) (src/qtt_tools.jl:852-932): the operator rides the vector kernel with physical dimension 4."""
    rng = np.random.default_rng(seed)
    N = n_dims * bits
    A = O.rand_tto((2,) * N, R, rng)
    dense = O.tto_to_tensor(A)
    inv = list(np.argsort(O.reorder_perm(n_dims, bits, True)))
    for threshold in (0.0, 1e-13):
        ref = O.reorder_op(A, n_dims, bits, True, threshold=threshold)
        got = T.reorder_op(to_product(A), n_dims, bits, "interleaved", threshold=threshold)
        assert list(got.tto_rks) == list(ref.tto_rks)
        _close(O.tto_to_tensor(to_oracle(got)), np.transpose(dense, inv + [N + a for a in inv]), 1e-11)
        if threshold > 0 or N <= 6:      # threshold 0 keeps every direction: at N = 8 the way back outgrows the rank capacity (64)
            back = T.reorder_op(got, n_dims, bits, "serial", threshold=threshold)
            _close(O.tto_to_tensor(to_oracle(back)), dense, 1e-11)
