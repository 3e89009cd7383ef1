"""Generates tests/golden/*.npz from the CPU oracle (oracle/tt_oracle.py).

The reference (Julia) cannot run in the build container, so these vectors come from the oracle,
which is itself pinned to the reference's known-answer tests (tests/test_oracle_reference_pins.py).
Inputs of the closed-form cases follow the reference constructors exactly; random cases use
numpy's default_rng with the seeds below.  Outputs of gauge-dependent ops are stored as
gauge-invariant quantities (ranks, ot flags, per-bond singular values, dense reconstruction).

Run from the repo root:  python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tt_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def pack_tt(prefix, x, out):
    out[f"{prefix}_dims"] = np.array(x.ttv_dims, dtype=np.int64)
    out[f"{prefix}_rks"] = np.array(x.ttv_rks, dtype=np.int64)
    out[f"{prefix}_ot"] = np.array(x.ttv_ot, dtype=np.int64)
    for k, c in enumerate(x.ttv_vec):
        out[f"{prefix}_core{k}"] = np.asarray(c, dtype=np.float64)


def pack_tto(prefix, A, out):
    out[f"{prefix}_dims"] = np.array(A.tto_dims, dtype=np.int64)
    out[f"{prefix}_rks"] = np.array(A.tto_rks, dtype=np.int64)
    for k, c in enumerate(A.tto_vec):
        out[f"{prefix}_core{k}"] = np.asarray(c, dtype=np.float64)


def closed_forms():
    out = {}
    for d in (4, 6):
        A = O.Delta(d)
        pack_tto(f"delta{d}", A, out)
        out[f"delta{d}_dense"] = O.qtto_to_matrix(A)
    s = O.qtt_sin(6, lam=math.pi)
    pack_tt("sin6", s, out)
    out["sin6_dense"] = O.qtt_to_vector(s)
    # config 1: tt_compress!(id_tto(6) * qtt_sin(6, λ=π), 2)
    y = O.apply(O.id_tto(6), s)
    pack_tt("c1_applied", y, out)
    sv = []
    O.tt_compress_(y, 2, svals_out=sv)
    pack_tt("c1_compressed", y, out)
    out["c1_dense"] = O.qtt_to_vector(y)
    for i, v in enumerate(sv):
        out[f"c1_sv{i}"] = v
    np.savez_compressed(os.path.join(OUT, "closed_forms.npz"), **out)


def random_small():
    out = {}
    cases = [
        # name, dims, ranks x, ranks y, operator rmax, max_bond, truncerr, seed
        ("a", (2, 2, 2, 2), [1, 2, 3, 2, 1], [1, 3, 2, 2, 1], 2, 2, 0.0, 11),
        ("b", (2, 3, 4, 2, 2), [1, 2, 5, 6, 2, 1], [1, 2, 4, 3, 2, 1], 3, 3, 0.0, 12),
        ("c", (2,) * 8, [1, 2, 4, 6, 6, 6, 4, 2, 1], [1, 2, 3, 3, 3, 3, 3, 2, 1], 2, 4, 0.0, 13),
        ("e", (3, 2, 3, 2), [1, 3, 4, 3, 1], [1, 2, 2, 2, 1], 2, 5, 1e-3, 14),
    ]
    for name, dims, rx, ry, rA, mb, te, seed in cases:
        rng = np.random.default_rng(seed)
        x = O.rand_tt(dims, rx, rng)
        y = O.rand_tt(dims, ry, rng)
        A = O.rand_tto(dims, rA, rng)
        p = f"{name}_"
        out[p + "max_bond"] = np.int64(mb)
        out[p + "truncerr"] = np.float64(te)
        pack_tt(p + "x", x, out)
        pack_tt(p + "y", y, out)
        pack_tto(p + "A", A, out)
        ax = O.apply(A, x)
        pack_tt(p + "apply", ax, out)
        out[p + "dot"] = np.float64(O.dot(x, y))
        out[p + "norm_x"] = np.float64(O.norm(x))
        pack_tt(p + "hadamard", O.hadamard(x, y), out)
        pack_tt(p + "add", O.add(x, y), out)
        pack_tt(p + "scale", O.scale(-2.5, x), out)
        for c in range(1, len(dims) + 1):
            o = O.orthogonalize(x, i=c)
            out[p + f"orth{c}_rks"] = np.array(o.ttv_rks, dtype=np.int64)
            out[p + f"orth{c}_ot"] = np.array(o.ttv_ot, dtype=np.int64)
        out[p + "x_dense"] = O.ttv_to_tensor(x)
        z = O.copy_tt(ax)
        sv = []
        O.tt_compress_(z, mb, truncerr=te, svals_out=sv)
        out[p + "compress_rks"] = np.array(z.ttv_rks, dtype=np.int64)
        out[p + "compress_dense"] = O.ttv_to_tensor(z)
        out[p + "compress_nsv"] = np.int64(len(sv))
        for i, v in enumerate(sv):
            out[p + f"compress_sv{i}"] = v
    np.savez_compressed(os.path.join(OUT, "random_small.npz"), **out)


if __name__ == "__main__":
    closed_forms()
    random_small()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")
