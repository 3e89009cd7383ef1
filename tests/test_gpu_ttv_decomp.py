"""GPU parity of ttv_decomp (src/tt_tools.jl:186-252): the HIP path through ttn_ttv_decomp vs the CPU oracle.
Tolerances (fp64): ranks / ot flags exact (spectra with a gap around tol); reconstruction 1e-12 relative; cores equal to the
oracle's up to one sign per bond, 1e-9 relative (non-degenerate singular values); orthogonality of the gauged cores 1e-12."""
import math

import numpy as np
import pytest

from oracle import tt_oracle as O
from tests.helpers import sign_fix_compare, to_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import ttn_amd
    ttn_amd.ensure_init(0)
    return ttn_amd


def _gauge_err(tt):
    worst = 0.0
    for k, c in enumerate(tt.ttv_vec):
        c = np.asarray(c)
        n, rl, rr = c.shape
        if tt.ttv_ot[k] == -1:
            m = c.transpose(1, 0, 2).reshape(rl * n, rr)
            worst = max(worst, float(np.max(np.abs(m.T @ m - np.eye(rr)))))
        elif tt.ttv_ot[k] == 1:
            m = c.transpose(1, 0, 2).reshape(rl, n * rr)
            worst = max(worst, float(np.max(np.abs(m @ m.T - np.eye(rl)))))
    return worst


def test_ttv_decomp_reference_cases(T):
    """test/test_tt_tools.jl:319-322 and :1023-1028, same inputs / assertions (see the oracle pin of the same name)."""
    rng = np.random.default_rng(0)
    t = rng.standard_normal((2, 3, 2))
    tt = T.ttv_decomp(t, index=2)
    assert list(tt.ttv_ot) == [-1, 0, 1]
    assert np.allclose(O.ttv_to_tensor(to_oracle(tt)), t, atol=1e-10, rtol=0)
    bell = np.zeros((2, 2))
    bell[0, 0] = bell[1, 1] = 1 / math.sqrt(2)
    bt = T.ttv_decomp(bell)
    assert list(bt.ttv_rks) == [1, 2, 1]
    assert np.allclose(np.linalg.svd(np.asarray(bt.ttv_vec[0])[:, 0, :], compute_uv=False) ** 2, [0.5, 0.5])
    assert np.allclose(O.ttv_to_tensor(to_oracle(bt)), bell, atol=1e-14)


@pytest.mark.parametrize("dims,index,seed", [((2,) * 10, 1, 0), ((2,) * 10, 4, 1), ((2,) * 10, 10, 2), ((3, 2, 4, 2, 3), 1, 3),
                                             ((3, 2, 4, 2, 3), 3, 4), ((2,) * 14, 1, 5), ((4, 4, 4, 4), 2, 6), ((5,), 1, 7), ((2, 7), 2, 8)])
def test_ttv_decomp_full_rank_vs_oracle(T, dims, index, seed):
    rng = np.random.default_rng(seed)
    t = rng.standard_normal(dims)
    ref = O.ttv_decomp(t, index=index)
    got = T.ttv_decomp(t, index=index)
    assert list(got.ttv_rks) == list(ref.ttv_rks)
    assert list(got.ttv_ot) == list(ref.ttv_ot)
    sc = np.max(np.abs(t))
    assert np.max(np.abs(O.ttv_to_tensor(to_oracle(got)) - t)) <= 1e-12 * sc
    assert _gauge_err(got) <= 1e-12
    assert sign_fix_compare(to_oracle(got), ref) <= 1e-9


@pytest.mark.parametrize("d,r,index", [(10, 3, 1), (12, 4, 6), (9, 2, 9), (16, 5, 1)])
def test_ttv_decomp_recovers_exact_ranks(T, d, r, index):
    rng = np.random.default_rng(d + r)
    x = O.rand_tt((2,) * d, r, rng)
    dense = O.ttv_to_tensor(x)
    tol = 1e-10 * np.max(np.abs(dense))
    ref = O.ttv_decomp(dense, index=index, tol=tol)
    got = T.ttv_decomp(dense, index=index, tol=tol)
    assert list(got.ttv_rks) == list(ref.ttv_rks) == list(x.ttv_rks)
    assert np.max(np.abs(O.ttv_to_tensor(to_oracle(got)) - dense)) <= 1e-12 * np.max(np.abs(dense))
    assert _gauge_err(got) <= 1e-12


def test_ttv_decomp_batch_and_capacity(T):
    rng = np.random.default_rng(3)
    dims, B = (2, 3, 2, 3, 2), 7
    ts = rng.standard_normal((B,) + dims)
    cap = [1, 2, 6, 6, 2, 1]
    z = T.DeviceTT(dims, cap, batch=B)
    T.qtt.ttv_decomp_(z, ts, index=3)
    T.device.compress_status(z)
    for b in range(B):
        got = z.download(b)
        ref = O.ttv_decomp(ts[b], index=3)
        assert list(got.ttv_rks) == list(ref.ttv_rks)
        assert list(got.ttv_ot) == [-1, -1, 0, 1, 1]
        assert np.max(np.abs(O.ttv_to_tensor(to_oracle(got)) - ts[b])) <= 1e-12 * np.max(np.abs(ts[b]))
    small = T.DeviceTT(dims, [1, 2, 3, 3, 2, 1], batch=B)          # full rank is 6 at the middle bonds
    T.qtt.ttv_decomp_(small, ts, index=1)
    with pytest.raises(T.TTNError):
        T.device.compress_status(small)
    with pytest.raises(T.TTNError):
        T.qtt.ttv_decomp_(z, ts, index=6)
