"""The 512-THREAD build of k_compress (csrc/ttn_wg512.hip: two workgroups per CU, LDS image 64 x 128, persistent train loop) against the
same oracle comparisons as the 1024-thread build: every compress / bond-truncate / apply+compress parity test of
tests/test_gpu_parity.py is re-run with TTN_WG512=1 (the library then launches the 512-thread build whatever the batch size;
by default it takes it for batches of more than 256 trains — i.e. for what bench.py times, which test_wg512_bench_batch_parity
pins at the acceptance limit of its routes), and the GEMM / eigensolver self-tests of tests/test_gpu_kernels.py with
TTN_WG512_SELFTEST=1.  Same tolerances, stated in those files."""
import pytest

import tests.test_gpu_kernels as K
import tests.test_gpu_parity as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import ttn_amd
    ttn_amd.ensure_init(0)
    return ttn_amd


@pytest.fixture(autouse=True)
def _wg512(monkeypatch):
    monkeypatch.setenv("TTN_WG512", "1")
    monkeypatch.setenv("TTN_WG512_SELFTEST", "1")
    yield


def test_default_build_selection(T, monkeypatch):
    """Without TTN_WG512 the library picks the build by batch size: both must give the oracle's answer on the same trains (the
    selection itself is host logic: batches of more than 256 trains -> 512-thread build)."""
    monkeypatch.delenv("TTN_WG512")
    P.test_apply_compress_vs_oracle(T, 12, 16, 2)


@pytest.mark.parametrize("m,n,k", [(16, 16, 4), (128, 128, 16), (128, 384, 192), (1, 1, 1), (17, 33, 5), (130, 70, 37), (64, 200, 129),
                                   (3, 300, 2), (64, 384, 128), (40, 520, 800), (192, 192, 1000), (64, 64, 128), (128, 64, 64)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_wg512_gemm_exact_on_integers(T, m, n, k, ta, tb):
    K.test_wg_gemm_exact_on_integers(T, m, n, k, ta, tb)


def test_wg512_gemm_random_fp64(T):
    K.test_wg_gemm_random_fp64(T)


@pytest.mark.parametrize("n,r,nev,decades,seed", [(128, 64, 64, 1.5, 0), (128, 64, 128, 2.0, 1), (128, 17, 40, 1.0, 2), (128, 64, 64, 0.0, 3),
                                                  (64, 64, 64, 1.5, 4), (64, 20, 64, 2.0, 5), (64, 64, 64, 0.0, 6)])
def test_wg512_eig_selftest(n, r, nev, decades, seed):
    K.test_eig_selftest(n, r, nev, decades, seed)


@pytest.mark.parametrize("name", list("abce"))
def test_wg512_golden_random_small(T, name):
    P.test_golden_random_small(T, name)


def test_wg512_golden_config1(T):
    P.test_golden_config1(T)


def test_wg512_bond_truncate_reference_cases(T):
    P.test_bond_truncate_reference_cases(T)


def test_wg512_tt_compress_reference_behaviour(T, caplog):
    P.test_tt_compress_reference_behaviour(T, caplog)


def test_wg512_compress_rank_growth_like_reference(T):
    P.test_compress_rank_growth_like_reference(T)


def test_wg512_compress_compressible_inputs(T):
    P.test_compress_compressible_inputs(T)


@pytest.mark.parametrize("d,r,seed", [(8, 6, 1), (12, 16, 2), (20, 32, 20)])
def test_wg512_apply_compress_vs_oracle(T, d, r, seed):
    P.test_apply_compress_vs_oracle(T, d, r, seed)


def test_wg512_batched_handles_independent_trains(T):
    P.test_batched_handles_independent_trains(T)


def test_wg512_headline_config_properties(T):
    P.test_headline_config_properties(T)


@pytest.mark.parametrize("seed", [30, 192, 210])
def test_wg512_headline_config_singular_values(T, seed):
    P.test_headline_config_singular_values(T, seed)


def test_wg512_bench_batch_parity(T):
    P.test_bench_batch_parity(T)


def test_wg512_compress_general_dims_and_odd_sizes(T):
    P.test_compress_general_dims_and_odd_sizes(T)


def test_wg512_compress_short_side_above_128_uses_fallback(T):
    P.test_compress_short_side_above_128_uses_fallback(T)


def test_wg512_compress_long_panel_householder(T):
    P.test_compress_long_panel_householder(T)


@pytest.mark.parametrize("d,r,mb", [(14, 100, 100), (16, 128, 128)])
def test_wg512_compress_ranks_up_to_128_blocked_jacobi(T, d, r, mb):
    P.test_compress_ranks_up_to_128_blocked_jacobi(T, d, r, mb)


def test_wg512_compress_truncerr_on_incompressible_input(T):
    P.test_compress_truncerr_on_incompressible_input(T)


def test_wg512_bond_truncate_every_bond(T):
    P.test_bond_truncate_every_bond(T)


def test_wg512_ragged_batch_different_ranks_per_train(T):
    P.test_ragged_batch_different_ranks_per_train(T)


def test_wg512_fast_routes_agree_with_robust_route(T, monkeypatch):
    P.test_fast_routes_agree_with_robust_route(T, monkeypatch)


@pytest.mark.parametrize("seed", [3, 192])
def test_wg512_rank64_route_variants_agree(T, monkeypatch, seed):
    P.test_rank64_route_variants_agree(T, monkeypatch, seed)


@pytest.mark.parametrize("dims,oprks,xr,mb", [((2,) * 30, None, 64, 64), ((2, 3, 2, 2, 3, 2, 2), [1, 2, 3, 2, 4, 2, 3, 1], 5, 6),
                                              ((4, 2, 3, 4, 2), [1, 3, 2, 2, 3, 1], 7, 9)])
def test_wg512_fused_apply_compress_equals_apply_then_compress(T, monkeypatch, dims, oprks, xr, mb):
    P.test_fused_apply_compress_equals_apply_then_compress(T, monkeypatch, dims, oprks, xr, mb)


def test_wg512_sweep_ranges_and_core_handoff_reproduce_compress(T):
    P.test_sweep_ranges_and_core_handoff_reproduce_compress(T)


def test_wg512_randomized_apply_compress_parity(T):
    P.test_randomized_apply_compress_parity(T)


@pytest.mark.parametrize("d,kind,xr,max_bond,truncerr,seed", [(12, 0, 64, 64, 0.0, 0), (13, 0, 64, 50, 0.0, 1), (12, 1, 64, 33, 0.0, 2),
                                                              (12, 2, 64, 64, 1e-8, 3), (14, 2, 48, 64, 0.0, 4), (12, 0, 64, 20, 1e-12, 5),
                                                              (12, 0, 32, 32, 0.0, 6), (13, 2, 32, 20, 1e-8, 7), (12, 1, 32, 31, 0.0, 8)])
def test_wg512_apply_compress_eigen_routes_vs_oracle(T, d, kind, xr, max_bond, truncerr, seed):
    P.test_apply_compress_eigen_routes_vs_oracle(T, d, kind, xr, max_bond, truncerr, seed)
