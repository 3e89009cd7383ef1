"""Generates tests/golden/c5_matrix_free_rank<R>.json from the CPU oracle: two-site dmrg_linsolve with the REFERENCE's default local
solver (it_solver = true, linsolv_maxiter = 200 -> KrylovKit CG(maxiter = 30 * 200), linsolv_tol = max(sqrt(tol), 1e-8);
src/solvers/dmrg.jl:392-395, :170) on the 2D Laplace problem of examples/Laplace_pde.jl:12-27 at BASELINE config C5's size
(d = 2 x 12 bits) from a random rank-R start train (numpy default_rng(9), the seed of tests/test_gpu_c5_laplace.py).

The oracle needs minutes (rank 32) to hours (rank 128) for these runs — up to 6000 CG iterations on 4 R^2 unknowns per local solve, 45
local solves — far too slow for the GPU box's test budget, so its result is stored: relative residual after the sweep, total CG
iterations, number of local solves, final ranks and gauge flags.  The start train is NOT stored: it is regenerated from the seed.

Run from the repo root:  python tests/golden/make_c5_golden.py 32 [64 128]
"""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tt_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
BITS = 12


def kron_op(A, B):
    return O.TToperator(A.N + B.N, list(A.tto_vec) + list(B.tto_vec), tuple(A.tto_dims) + tuple(B.tto_dims),
                        list(A.tto_rks[:-1]) + list(B.tto_rks), [0] * (A.N + B.N))


def kron_vec(a, b):
    return O.TTvector(a.N + b.N, list(a.ttv_vec) + list(b.ttv_vec), tuple(a.ttv_dims) + tuple(b.ttv_dims),
                      list(a.ttv_rks[:-1]) + list(b.ttv_rks), [0] * (a.N + b.N))


def problem():
    d = BITS
    h = 1.0 / (2 ** d + 1)
    L1 = O.toeplitz_to_qtto(-2.0, 1.0, 1.0, d)
    A = O.tto_scale(1 / h ** 2, O.tto_add(kron_op(L1, O.id_tto(d)), kron_op(O.id_tto(d), L1)))
    e1 = O.TTvector(d, [np.array([[[1.0]], [[0.0]]]) for _ in range(d)], (2,) * d, [1] * (d + 1), [0] * d)
    b = O.scale(-1 / h ** 2, kron_vec(O.qtt_sin(d, a=h, b=1 - h, lam=1.0 / math.pi), e1))
    return A, b


def norm_stable(x):
    return float(np.linalg.norm(O.orthogonalize(x, i=1).ttv_vec[0]))


def main():
    A, b = problem()
    for rank in [int(v) for v in sys.argv[1:]] or [32]:
        rng = np.random.default_rng(9)
        x0 = O.rand_tt((2,) * A.N, rank, rng)
        st = {}
        t0 = time.perf_counter()
        ref = O.dmrg_linsolve(A, b, x0, tol=1e-10, sweep_schedule=[2], rmax_schedule=[rank], it_solver=True, stats=st)
        dt = time.perf_counter() - t0
        res = norm_stable(O.sub(O.apply(A, ref), b)) / norm_stable(b)
        rec = {"rank": rank, "bits": BITS, "seed": 9, "kw": {"tol": 1e-10, "sweep_schedule": [2], "rmax_schedule": [rank], "it_solver": True},
               "cg_maxiter": O.KRYLOVDIM_DEFAULT * 200, "cg_iterations": int(st["cg_iterations"]), "cg_solves": int(st["cg_solves"]),
               "residual": res, "ranks": [int(v) for v in ref.ttv_rks], "ot": [int(v) for v in ref.ttv_ot], "oracle_seconds": round(dt, 1)}
        with open(os.path.join(OUT, f"c5_matrix_free_rank{rank}.json"), "w") as fh:
            json.dump(rec, fh, indent=1)
        print(rec, flush=True)


if __name__ == "__main__":
    main()
