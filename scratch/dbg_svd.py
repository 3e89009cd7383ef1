import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, math
import torch; torch.cuda.is_available()
import ttn_amd as T
from ttn_amd import tdvp as D
from oracle import tt_oracle as O
from helpers import to_product
from test_oracle_reference_pins import heat_problem
T.ensure_init(0)
orig = D._svd_j
def wrapped(Mt):
    try:
        return orig(Mt)
    except Exception as e:
        X = D._down(Mt)
        np.save("gpurun_out/svd_fail.npy", X)
        print("FAILED on", X.shape, X.dtype, "svals", np.linalg.svd(X, compute_uv=False))
        raise
D._svd_j = wrapped
A, u0, lam = heat_problem()
sol2 = D.tdvp2(to_product(A), to_product(u0), [1e-3] * 5, imaginary_time=True, normalize=False, max_bond=8, truncerr=1e-12)
print("ok")
