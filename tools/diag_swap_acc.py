"""Diagnostic: accuracy of hadamard_ttm (device and CPU oracle) against the dense product, and the rank profiles."""
import sys
import numpy as np
sys.path.insert(0, ".")
import ttn_amd as T
from oracle import tt_oracle as O
from tests.helpers import to_product, to_oracle
T.ensure_init(0)
for d, r, tol in ((16, 3, 1e-10), (14, 4, 1e-10), (10, 4, 1e-14), (8, 3, 1e-14)):
    rng = np.random.default_rng(1)
    x = O.rand_tt((2,) * d, r, rng); y = O.rand_tt((2,) * d, r, rng)
    exact = O.ttv_to_tensor(x) * O.ttv_to_tensor(y)
    ref = O.hadamard_ttm(x, y, tol=tol)
    sc = np.max(np.abs(exact))
    try:
        got = T.qtt.hadamard_ttm(to_product(x), to_product(y), tol=tol)
    except T.TTNError as e:
        print(f"d={d} r={r} tol={tol}: {e}; cpu ranks {ref.ttv_rks}")
        continue
    print(f"d={d} r={r} tol={tol}: err gpu {np.max(np.abs(O.ttv_to_tensor(to_oracle(got)) - exact)) / sc:.2e}  cpu {np.max(np.abs(O.ttv_to_tensor(ref) - exact)) / sc:.2e}")
    print("   gpu ranks", got.ttv_rks)
    print("   cpu ranks", ref.ttv_rks)
