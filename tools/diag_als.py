"""Diagnostic (not a test): als_linsolve on the 2D Laplace problem of examples/Laplace_pde.jl (serial QTT ordering,
d bits per dimension), device vs the CPU oracle.   python tools/diag_als.py [bits] [rank] [batch] [sweeps]"""
import math
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import ttn_amd as T
from oracle import tt_oracle as O
from tests.helpers import to_product, to_oracle, tt_rel_diff, tt_norm_stable

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
sweeps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
T.ensure_init(0)


def kron_op(A, Bq):       # kron(A, B) of TT operators: the cores side by side (src/tt_operations.jl:427-433)
    return O.TToperator(A.N + Bq.N, list(A.tto_vec) + list(Bq.tto_vec), tuple(A.tto_dims) + tuple(Bq.tto_dims),
                        list(A.tto_rks[:-1]) + list(Bq.tto_rks), [0] * (A.N + Bq.N))


def kron_vec(a, b):
    return O.TTvector(a.N + b.N, list(a.ttv_vec) + list(b.ttv_vec), tuple(a.ttv_dims) + tuple(b.ttv_dims),
                      list(a.ttv_rks[:-1]) + list(b.ttv_rks), [0] * (a.N + b.N))


d = bits
Nn = 2 ** d
h = 1.0 / (Nn + 1)
L1 = O.toeplitz_to_qtto(-2.0, 1.0, 1.0, d)
A = O.tto_scale(1 / h ** 2, O.tto_add(kron_op(L1, O.id_tto(d)), kron_op(O.id_tto(d), L1)))
e1 = O.TTvector(d, [np.array([[[1.0]], [[0.0]]]) for _ in range(d)], (2,) * d, [1] * (d + 1), [0] * d)     # qtt_basis_vector(d, 1)
b = O.scale(-1 / h ** 2, kron_vec(O.qtt_sin(d, a=h, b=1 - h, lam=1.0 / math.pi), e1))
print("operator ranks", A.tto_rks, " rhs ranks", b.ttv_rks)
rng = np.random.default_rng(0)
N = 2 * d
x0 = O.rand_tt((2,) * N, rank, rng)
print("start ranks", x0.ttv_rks, " largest local system", max(2 * x0.ttv_rks[i] * x0.ttv_rks[i + 1] for i in range(N)))
dA = T.DeviceTTO(to_product(A))
db = T.DeviceTT.from_host(to_product(b), batch=B)
dx0 = T.DeviceTT((2,) * N, x0.ttv_rks, batch=B)
for i in range(B):
    dx0.upload(i, to_product(x0 if i == 0 else O.rand_tt((2,) * N, rank, rng)))
dx = T.DeviceTT((2,) * N, x0.ttv_rks, batch=B)
T.solvers.als_linsolve_(dA, db, dx0, dx, sweeps)
T.device.compress_status(dx)
with T.StreamTimer() as tm:
    T.solvers.als_linsolve_(dA, db, dx0, dx, sweeps)
T.device.compress_status(dx)
t0 = time.time()
ref = O.als_linsolve(A, b, x0, sweep_count=sweeps)
t_cpu = time.time() - t0
got = to_oracle(dx.download(0))
res = tt_norm_stable(O.sub(O.apply(A, got), b)) / tt_norm_stable(b)
print(f"als_linsolve 2D Laplace {bits}+{bits} bits rank {rank}, {sweeps} half sweeps: device {tm.ms:.1f} ms for {B} systems ({tm.ms / B:.2f} ms each);"
      f" cpu oracle {t_cpu * 1e3:.1f} ms each; rel diff {tt_rel_diff(got, ref):.1e}; residual {res:.2e}")

# mals_linsolve on the same problem, start ranks of the right-hand side like examples/Laplace_pde.jl:24-27
x0m = O.rand_tt((2,) * N, b.ttv_rks, rng)
rmax = 16
cap = T.solvers.mals_capacity((2,) * N, x0m.ttv_rks, rmax)
dbm = T.DeviceTT.from_host(to_product(b), batch=B)
dx0m = T.DeviceTT.from_host(to_product(x0m), batch=B)
dxm = T.DeviceTT((2,) * N, cap, batch=B)
T.solvers.mals_linsolve_(dA, dbm, dx0m, dxm, 1e-12, rmax)
T.device.compress_status(dxm)
with T.StreamTimer() as tm:
    T.solvers.mals_linsolve_(dA, dbm, dx0m, dxm, 1e-12, rmax)
T.device.compress_status(dxm)
t0 = time.time()
refm = O.mals_linsolve(A, b, x0m, tol=1e-12, rmax=rmax)
t_cpu = time.time() - t0
gotm = to_oracle(dxm.download(0))
res = tt_norm_stable(O.sub(O.apply(A, gotm), b)) / tt_norm_stable(b)
resc = tt_norm_stable(O.sub(O.apply(A, refm), b)) / tt_norm_stable(b)
print(f"mals_linsolve (tol 1e-12, rmax {rmax}): ranks {gotm.ttv_rks} (cpu {refm.ttv_rks}); device {tm.ms:.1f} ms for {B} systems; cpu oracle {t_cpu * 1e3:.1f} ms each;"
      f" residual {res:.2e} (cpu {resc:.2e})")

# dmrg_linsolve (N = 2) on the same problem: examples/Laplace_pde.jl:27 runs the default schedule; a longer one shows convergence
for sched, rmaxs in (([2], [rmax]), ([6], [rmax])):
    T.solvers.dmrg_linsolve_(dA, dbm, dx0m, dxm, 1e-10, sched, rmaxs)
    T.device.compress_status(dxm)
    with T.StreamTimer() as tm:
        T.solvers.dmrg_linsolve_(dA, dbm, dx0m, dxm, 1e-10, sched, rmaxs)
    T.device.compress_status(dxm)
    t0 = time.time()
    refd = O.dmrg_linsolve(A, b, x0m, tol=1e-10, sweep_schedule=sched, rmax_schedule=rmaxs)
    t_cpu = time.time() - t0
    gotd = to_oracle(dxm.download(0))
    res = tt_norm_stable(O.sub(O.apply(A, gotd), b)) / tt_norm_stable(b)
    resc = tt_norm_stable(O.sub(O.apply(A, refd), b)) / tt_norm_stable(b)
    print(f"dmrg_linsolve (tol 1e-10, schedule {sched} / {rmaxs}): ranks {gotd.ttv_rks} (cpu {refd.ttv_rks}); device {tm.ms:.1f} ms for {B} systems;"
          f" cpu oracle {t_cpu * 1e3:.1f} ms each; residual {res:.2e} (cpu {resc:.2e}); rel diff {tt_rel_diff(gotd, refd):.1e}")
