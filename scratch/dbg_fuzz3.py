import sys, os; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import ttn_amd as T
from oracle import tt_oracle as O
from helpers import to_oracle, to_product, tt_rel_diff
T.ensure_init(0)
rng = np.random.default_rng(5)
d = 10
x = O.rand_tt((2,) * d, 16, rng)
z = O.zeros_tt(x.ttv_dims, x.ttv_rks)
for name, tt in (("zero train", z),):
    for c in (1, 5, d):
        got = T.orthogonalize(to_product(tt), i=c); ref = O.orthogonalize(tt, i=c)
        print(name, c, "ranks equal", list(got.ttv_rks) == ref.ttv_rks, "ot equal", list(got.ttv_ot) == ref.ttv_ot, "max|core|", max(float(np.max(np.abs(np.asarray(a)))) for a in got.ttv_vec[c-1:c]), "nan", any(np.isnan(np.asarray(a)).any() for a in got.ttv_vec))
# a train with one zero core and one with duplicated columns (rank deficient everywhere)
y = O.copy_tt(x); y.ttv_vec[6][:] = 0.0
w = O.copy_tt(x)
for k in range(1, d - 1):
    w.ttv_vec[k][:, :, 1::2] = w.ttv_vec[k][:, :, 0::2][:, :, : w.ttv_vec[k][:, :, 1::2].shape[2]]
for name, tt in (("zero core", y), ("duplicated columns", w)):
    for c in (1, 4):
        got = T.orthogonalize(to_product(tt), i=c); ref = O.orthogonalize(tt, i=c)
        dn = np.linalg.norm(O.ttv_to_tensor(to_oracle(got)) - O.ttv_to_tensor(tt)) / max(np.linalg.norm(O.ttv_to_tensor(tt)), 1e-300)
        print(name, c, "ranks equal", list(got.ttv_rks) == ref.ttv_rks, "tensor", dn, "nan", any(np.isnan(np.asarray(a)).any() for a in got.ttv_vec))
