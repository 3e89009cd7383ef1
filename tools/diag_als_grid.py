"""Diagnostic (not a test): als_linsolve through the grid form at large ranks on the 2D Laplace problem of examples/Laplace_pde.jl
(BASELINE config C5: d = 2 x 12 bits).   python tools/diag_als_grid.py [rank] [bits] [sweeps]"""
import math
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import ttn_amd as T
from oracle import tt_oracle as O
from tests.helpers import to_product, to_oracle, tt_norm_stable
import tests.test_gpu_c5_laplace as C5

rank = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 12
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
T.ensure_init(0)
d = bits
h = 1.0 / (2 ** d + 1)
L1 = O.toeplitz_to_qtto(-2.0, 1.0, 1.0, d)
A = O.tto_scale(1 / h ** 2, O.tto_add(C5._kron_op(L1, O.id_tto(d)), C5._kron_op(O.id_tto(d), L1)))
e1 = O.TTvector(d, [np.array([[[1.0]], [[0.0]]]) for _ in range(d)], (2,) * d, [1] * (d + 1), [0] * d)
b = O.scale(-1 / h ** 2, C5._kron_vec(O.qtt_sin(d, a=h, b=1 - h, lam=1.0 / math.pi), e1))
N = 2 * d
rng = np.random.default_rng(6)
x0 = O.rand_tt((2,) * N, rank, rng)
print("start ranks", x0.ttv_rks, "largest one-site system", max(2 * x0.ttv_rks[i] * x0.ttv_rks[i + 1] for i in range(N)))
t0 = time.perf_counter()
got = T.solvers.als_linsolve(to_product(A), to_product(b), to_product(x0), sweep_count=sweeps)
dt = time.perf_counter() - t0
res = tt_norm_stable(O.sub(O.apply(A, to_oracle(got)), b)) / tt_norm_stable(b)
print(f"als_linsolve rank {rank}, {sweeps} half sweeps: {dt:.2f} s, residual {res:.3e}, ranks {list(got.ttv_rks)}")
