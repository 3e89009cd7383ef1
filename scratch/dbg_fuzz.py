import sys, os; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import ttn_amd as T
from oracle import tt_oracle as O
from helpers import to_oracle, to_product
T.ensure_init(0)
def worst_of(got, center):
    worst = 0.0; where = None
    for j, G in enumerate(got.ttv_vec):
        G = np.asarray(G); n, rl, rr = G.shape
        if j < center - 1:
            A = G.transpose(1, 0, 2).reshape(rl * n, rr, order="F"); w = float(np.max(np.abs(A.T @ A - np.eye(rr))))
        elif j > center - 1:
            A = G.transpose(1, 2, 0).reshape(rl, rr * n, order="F"); w = float(np.max(np.abs(A @ A.T - np.eye(rl))))
        else:
            continue
        if w > worst: worst, where = w, (j, G.shape)
    return worst, where
rng = np.random.default_rng(4242)
for trial in range(40):
    d = int(rng.integers(3, 17))
    rks = [1] + [int(rng.integers(1, 65)) for _ in range(d - 1)] + [1]
    x = O.rand_tt((2,) * d, rks, rng)
    center = int(rng.integers(1, d + 1))
    xp = to_product(x)
    res = {}
    for form in ("default", "0", "noramp"):
        os.environ.pop("TTN_ORTHO512", None); os.environ.pop("TTN_ORTHO_RAMP", None)
        if form == "0": os.environ["TTN_ORTHO512"] = "0"
        if form == "noramp": os.environ["TTN_ORTHO_RAMP"] = "0"
        res[form] = worst_of(T.orthogonalize(xp, i=center), center)
    os.environ.pop("TTN_ORTHO512", None); os.environ.pop("TTN_ORTHO_RAMP", None)
    ro = worst_of(to_product(O.orthogonalize(x, i=center)), center)
    if max(res["default"][0], res["0"][0]) > 3e-13:
        print(trial, "d", d, "centre", center, "ranks", x.ttv_rks, "default", res["default"], "single", res["0"], "noramp", res["noramp"], "oracle", ro[0], "actual ranks", to_product(O.orthogonalize(x, i=center)).ttv_rks)
