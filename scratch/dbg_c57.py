import sys, os; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import ttn_amd as T
from oracle import tt_oracle as O
from helpers import to_oracle, to_product, tt_rel_diff
T.ensure_init(0)
rng = np.random.default_rng(2024)
for trial in range(58):
    d = int(rng.integers(2, 13)); n = 2 if trial % 4 else 3; dims = (n,) * d
    rks = [1] + [int(rng.integers(1, 41)) for _ in range(d - 1)] + [1]
    x = O.rand_tt(dims, rks, rng)
    mb = int(rng.integers(1, 30)); te = [0.0, 0.0, 1e-8, 1e-4][trial % 4]; sw = 1 if trial % 7 else 2
    A = O.rand_tto(dims, int(rng.integers(1, 4)), rng)
    if trial == 57: break
print("trial", trial, "d", d, "ranks", x.ttv_rks, "mb", mb, "te", te, "sw", sw)
ref = O.tt_compress_(O.copy_tt(x), mb); got = T.tt_compress_(to_product(O.copy_tt(x)), mb)
print("final", got.ttv_rks, ref.ttv_rks, tt_rel_diff(to_oracle(got), ref), "vs input: dev", tt_rel_diff(to_oracle(got), x), "oracle", tt_rel_diff(ref, x))
# step by step: bond truncations L->R then R->L
xo = O.copy_tt(x); xd = to_product(O.copy_tt(x))
order = list(range(1, d)) + list(range(d - 1, 0, -1))
for k in order:
    so = []
    O.tt_bond_truncate_(xo, k, max_bond=mb, svals_out=so)
    T._tt_bond_truncate_(xd, k, max_bond=mb)
    print("bond", k, "ranks dev", xd.ttv_rks, "oracle", xo.ttv_rks, "diff", tt_rel_diff(to_oracle(xd), xo), "svals", np.array2string(np.asarray(so[-1])[:6], precision=3) if so else "")
