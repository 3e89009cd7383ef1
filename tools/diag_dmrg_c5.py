"""Diagnostic (not a test): two-site DMRG with the matrix-free CG local solver on the 2D Laplace problem of examples/Laplace_pde.jl at
BASELINE config C5 size (2 x 12 bits, rank up to 128 -> local systems of 65 536 unknowns).   python tools/diag_dmrg_c5.py [rank] [batch] [maxiter]"""
import math
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import ttn_amd as T
from oracle import tt_oracle as O
from tests.helpers import to_product, to_oracle, tt_norm_stable
import tests.test_gpu_c5_laplace as C5

rank = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
maxiter = int(sys.argv[3]) if len(sys.argv) > 3 else 200
T.ensure_init(0)
d = 12
h = 1.0 / (2 ** d + 1)
L1 = O.toeplitz_to_qtto(-2.0, 1.0, 1.0, d)
A = O.tto_scale(1 / h ** 2, O.tto_add(C5._kron_op(L1, O.id_tto(d)), C5._kron_op(O.id_tto(d), L1)))
e1 = O.TTvector(d, [np.array([[[1.0]], [[0.0]]]) for _ in range(d)], (2,) * d, [1] * (d + 1), [0] * d)
b = O.scale(-1 / h ** 2, C5._kron_vec(O.qtt_sin(d, a=h, b=1 - h, lam=1.0 / math.pi), e1))
N = 2 * d
rng = np.random.default_rng(9)
x0 = O.rand_tt((2,) * N, rank, rng)
print("start ranks", x0.ttv_rks, "largest two-site system", max(4 * x0.ttv_rks[i] * x0.ttv_rks[i + 2] for i in range(N - 1)))
dA = T.DeviceTTO(to_product(A))
db = T.DeviceTT.from_host(to_product(b), batch=B)
dx0 = T.DeviceTT.from_host(to_product(x0), batch=B)
cap = T.solvers.dmrg_capacity((2,) * N, x0.ttv_rks, rank)
dx = T.DeviceTT((2,) * N, cap, batch=B)
for rep in range(2):
    T.device.sync()
    t0 = time.perf_counter()
    T.solvers.dmrg_linsolve_(dA, db, dx0, dx, 1e-10, [2], [rank], it_solver=True, linsolv_maxiter=maxiter)
    T.device.compress_status(dx)
    dt = time.perf_counter() - t0
    it = T.solvers.dmrg_cg_iterations(B)
    got = to_oracle(dx.download(0))
    res = tt_norm_stable(O.sub(O.apply(A, got), b)) / tt_norm_stable(b)
    # flops of one symmetrised matvec at the start ranks (upper bound for the sweep: the ranks only shrink)
    print(f"run {rep}: {dt:.2f} s for {B} system(s), one full sweep + closing solve (45 local solves); CG iterations per system {it[0]}; "
          f"residual {res:.3e}; ranks {got.ttv_rks}")
