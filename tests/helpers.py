"""Shared test helpers: oracle <-> product conversions and golden-fixture loading."""
import os

import numpy as np

from oracle import tt_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def tt_from_golden(g, prefix):
    dims = tuple(int(v) for v in g[f"{prefix}_dims"])
    rks = [int(v) for v in g[f"{prefix}_rks"]]
    ot = [int(v) for v in g[f"{prefix}_ot"]] if f"{prefix}_ot" in g else [0] * len(dims)
    cores = [np.array(g[f"{prefix}_core{k}"]) for k in range(len(dims))]
    return O.TTvector(len(dims), cores, dims, rks, ot)


def tto_from_golden(g, prefix):
    dims = tuple(int(v) for v in g[f"{prefix}_dims"])
    rks = [int(v) for v in g[f"{prefix}_rks"]]
    cores = [np.array(g[f"{prefix}_core{k}"]) for k in range(len(dims))]
    return O.TToperator(len(dims), cores, dims, rks, [0] * len(dims))


def to_product(x):
    """oracle TTvector/TToperator -> product (ttn_amd) TTvector/TToperator"""
    import ttn_amd as T
    if isinstance(x, O.TTvector):
        return T.TTvector(x.N, [np.asfortranarray(c) for c in x.ttv_vec], x.ttv_dims, x.ttv_rks, x.ttv_ot)
    return T.TToperator(x.N, [np.asfortranarray(c) for c in x.tto_vec], x.tto_dims, x.tto_rks, x.tto_ot)


def to_oracle(x):
    if hasattr(x, "ttv_vec"):
        return O.TTvector(x.N, [np.array(c) for c in x.ttv_vec], tuple(x.ttv_dims), list(x.ttv_rks), list(x.ttv_ot))
    return O.TToperator(x.N, [np.array(c) for c in x.tto_vec], tuple(x.tto_dims), list(x.tto_rks), list(x.tto_ot))


def tt_norm_stable(x):
    """||x|| without the cancellation of dot(x,x) on differences: orthogonalize to site 1 (oracle) and take
    the Frobenius norm of the centre core."""
    y = O.orthogonalize(x, i=1)
    return float(np.linalg.norm(y.ttv_vec[0]))


def tt_rel_diff(a, b):
    """||a - b|| / ||b|| for trains too long to densify (d=30).  The difference is formed as a TT
    (ranks add) and its norm taken after orthogonalization, so the result is accurate far below
    sqrt(eps) (aa - 2ab + bb would cancel catastrophically at ~1e-8)."""
    diff = O.sub(a, b)
    return tt_norm_stable(diff) / tt_norm_stable(b)


def sign_fix_compare(a, b):
    """tt_compress! output cores are unique up to one sign per bond (non-degenerate singular
    values): fix signs bond by bond and return the max relative core difference."""
    worst = 0.0
    carry = None
    for k in range(a.N):
        ca, cb = np.array(a.ttv_vec[k]), np.array(b.ttv_vec[k])
        if carry is not None:
            ca = ca * carry[None, :, None]
        if k < a.N - 1:
            # sign of each right-index slice
            sa = np.sign(np.einsum("iab,iab->b", ca, cb))
            sa[sa == 0] = 1.0
            ca = ca * sa[None, None, :]
            carry = sa
        worst = max(worst, float(np.max(np.abs(ca - cb)) / max(np.max(np.abs(cb)), 1e-300)))
    return worst
