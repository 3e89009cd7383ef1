import sys, os; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import ctypes as C
import numpy as np
import ttn_amd as T
from oracle import tt_oracle as O
from helpers import to_oracle, to_product, tt_rel_diff
T.ensure_init(0)
def worst_of(got, center):
    worst = 0.0
    for j, G in enumerate(got.ttv_vec):
        G = np.asarray(G); n, rl, rr = G.shape
        if j > center - 1:
            A = G.transpose(1, 2, 0).reshape(rl, rr * n, order="F"); worst = max(worst, float(np.max(np.abs(A @ A.T - np.eye(rl)))))
    return worst
rng = np.random.default_rng(11)
d, r = 14, 48
for eps in (1e-1, 3e-2, 1e-2, 3e-3, 1e-3):
    x = O.rand_tt((2,) * d, r, rng)
    for k in (6, 7, 8):
        c = x.ttv_vec[k]
        c[:, 1::2, :] = c[:, 0::2, :][:, : c[:, 1::2, :].shape[1], :] + eps * c[:, 1::2, :]
    xp = to_product(x)
    os.environ["TTN_ORTHO512"] = "0"
    g0 = T.orthogonalize(xp, i=1)
    print("   single launch: tensor", tt_rel_diff(to_oracle(g0), x), "orth", worst_of(g0, 1))
    os.environ.pop("TTN_ORTHO512")
    got = T.orthogonalize(xp, i=1)
    st = (C.c_int64 * 4)(); T._lib.check(T._lib.lib().ttn_debug_ortho_state(0, st))
    ref = O.orthogonalize(x, i=1)
    print("eps", eps, "finished by k_ortho512", int(st[3]), "next site", int(st[0]), "ranks ok", list(got.ttv_rks) == ref.ttv_rks, "tensor", tt_rel_diff(to_oracle(got), x), "orth", worst_of(got, 1))
