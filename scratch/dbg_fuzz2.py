import sys, os; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import ttn_amd as T
from oracle import tt_oracle as O
from helpers import to_oracle, to_product, tt_rel_diff
T.ensure_init(0)
rng = np.random.default_rng(99)
bad = 0
for trial in range(60):
    d = int(rng.integers(2, 15))
    n = 2 if trial % 5 else int(rng.integers(2, 4))
    dims = (n,) * d
    rx = [1] + [int(rng.integers(1, 40)) for _ in range(d - 1)] + [1]
    ry = [1] + [int(rng.integers(1, 40)) for _ in range(d - 1)] + [1]
    x, y = O.rand_tt(dims, rx, rng), O.rand_tt(dims, ry, rng)
    xp, yp = to_product(x), to_product(y)
    # dot
    ref = O.dot(x, y); got = T.dot(xp, yp)
    sc = np.sqrt(abs(O.dot(x, x)) * abs(O.dot(y, y)))
    if abs(got - ref) > 1e-12 * sc: bad += 1; print("dot", trial, d, n, got, ref)
    # add
    z = T.add(xp, yp) if hasattr(T, "add") else xp + yp
    if tt_rel_diff(to_oracle(z), O.add(x, y)) > 1e-13 or list(z.ttv_rks) != O.add(x, y).ttv_rks: bad += 1; print("add", trial)
    # hadamard (small ranks to keep the product ranks moderate)
    xs, ys = O.rand_tt(dims, [1] + [int(rng.integers(1, 9)) for _ in range(d - 1)] + [1], rng), O.rand_tt(dims, [1] + [int(rng.integers(1, 9)) for _ in range(d - 1)] + [1], rng)
    h = T.hadamard(to_product(xs), to_product(ys))
    href = O.hadamard(xs, ys)
    if list(h.ttv_rks) != href.ttv_rks or max(np.max(np.abs(np.asarray(a) - b)) for a, b in zip(h.ttv_vec, href.ttv_vec)) > 1e-13 * max(np.max(np.abs(b)) for b in href.ttv_vec):
        bad += 1; print("hadamard", trial)
    # apply with a random operator
    A = O.rand_tto(dims, int(rng.integers(1, 5)), rng)
    ya = T.apply(to_product(A), xp) if hasattr(T, "apply") else to_product(A) * xp
    yref = O.apply(A, x)
    if list(ya.ttv_rks) != yref.ttv_rks or max(np.max(np.abs(np.asarray(a) - b)) for a, b in zip(ya.ttv_vec, yref.ttv_vec)) > 1e-12 * max(np.max(np.abs(b)) for b in yref.ttv_vec):
        bad += 1; print("apply", trial)
    # scale
    s = T.scale(-1.75, xp) if hasattr(T, "scale") else (-1.75) * xp
    if tt_rel_diff(to_oracle(s), O.scale(-1.75, x)) > 1e-13: bad += 1; print("scale", trial)
print("bad", bad)
