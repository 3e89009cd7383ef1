import sys, os; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import ttn_amd as T
from ttn_amd import device as D
from oracle import tt_oracle as O
from helpers import to_oracle, to_product, tt_rel_diff
T.ensure_init(0)
rng = np.random.default_rng(123)
bad = 0
# apply with operator cores that do not fit the LDS staging (rank 40: 6400 doubles per core) and mixed sizes
for trial, (d, n, ro, rx) in enumerate([(5, 2, 40, 9), (4, 2, 33, 64), (4, 3, 20, 7), (6, 2, 24, 16)]):
    dims = (n,) * d
    A = O.rand_tto(dims, ro, rng); x = O.rand_tt(dims, rx, rng)
    ya, yref = T.apply(to_product(A), to_product(x)), O.apply(A, x)
    e = max(np.max(np.abs(np.asarray(a) - b)) for a, b in zip(ya.ttv_vec, yref.ttv_vec)) / max(np.max(np.abs(b)) for b in yref.ttv_vec)
    print("apply big operator", d, n, ro, rx, "ranks ok", list(ya.ttv_rks) == yref.ttv_rks, "err", e)
    bad += (e > 1e-12) or list(ya.ttv_rks) != yref.ttv_rks
# dot: ranks up to 64 (LDS-resident path), above (generic), unequal, long chains, ragged batch
for trial in range(30):
    d = int(rng.integers(2, 31))
    hi = 65 if trial % 3 else 100
    rx = [1] + [int(rng.integers(1, hi)) for _ in range(d - 1)] + [1]
    ry = [1] + [int(rng.integers(1, hi)) for _ in range(d - 1)] + [1]
    x, y = O.rand_tt((2,) * d, rx, rng), O.rand_tt((2,) * d, ry, rng)
    x = O.scale(1 / O.norm(x), x); y = O.scale(1 / O.norm(y), y)
    got, ref = T.dot(to_product(x), to_product(y)), O.dot(x, y)
    if abs(got - ref) > 1e-12: bad += 1; print("dot", trial, d, got, ref)
B, d = 40, 12
cap = [1] + [64] * (d - 1) + [1]
xs = [O.rand_tt((2,) * d, [1] + [int(rng.integers(1, 65)) for _ in range(d - 1)] + [1], rng) for _ in range(B)]
ys = [O.rand_tt((2,) * d, [1] + [int(rng.integers(1, 65)) for _ in range(d - 1)] + [1], rng) for _ in range(B)]
dx, dy = T.DeviceTT((2,) * d, cap, batch=B), T.DeviceTT((2,) * d, cap, batch=B)
for b in range(B):
    dx.upload(b, to_product(xs[b])); dy.upload(b, to_product(ys[b]))
got = D.dot(dx, dy)
for b in range(B):
    ref = O.dot(xs[b], ys[b]); sc = O.norm(xs[b]) * O.norm(ys[b])
    if abs(got[b] - ref) > 1e-12 * sc: bad += 1; print("batch dot", b, got[b], ref)
print("bad", bad)
