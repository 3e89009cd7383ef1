"""Diagnostic (not a test): the 128x128 symmetric eigensolver of the Gram route against NumPy."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import ttn_amd as T

T.ensure_init(0)
L = T._lib.lib()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for trial in range(4):
    if trial < 3:
        M = rng.standard_normal((128, 384))
        # shape the spectrum like the workload: smooth decay over ~1.5 decades
        U, s, Vt = np.linalg.svd(M, full_matrices=False)
        s = s[0] * 10.0 ** (-1.5 * np.arange(128) / 127 * (trial + 1) / 3)
        M = (U * s) @ Vt
    else:
        M = rng.standard_normal((128, 384))
    M /= np.max(np.abs(M))
    for n, r, nev in ((128, 64, 64), (128, 64, 128), (128, 17, 40), (64, 64, 64), (64, 20, 64), (64, 32, 32), (64, 5, 9)):
        G = np.asfortranarray(M[:n] @ M[:n].T)
        sig = np.zeros(128)
        X = np.zeros((128, 64), order="F")
        tk = (C.c_int64 * 6)()
        T._lib.check(L.ttn_selftest_eig128(G.ctypes.data_as(C.c_void_p), n, r, nev, sig.ctypes.data_as(C.c_void_p), X.ctypes.data_as(C.c_void_p), tk))
        w, V = np.linalg.eigh(G)
        w, V = w[::-1], V[:, ::-1]
        sref = np.sqrt(w)
        es = np.max(np.abs(sig[:nev] - sref[:nev]) / sref[:nev])
        Ux = X[:n, :r] / sig[:r]
        orth = np.max(np.abs(Ux.T @ Ux - np.eye(r)))
        resid = np.max(np.abs(G @ Ux - Ux * w[:r]) / w[0])
        sgn = np.sign(np.sum(Ux * V[:, :r], axis=0))
        ev = np.max(np.abs(Ux * sgn - V[:, :r]))
        print(f"trial {trial} n={n} r={r} nev={nev}: rc={tk[1]} ticks={tk[0]} [tridiag {tk[2]} bisect {tk[3]} twisted {tk[4]} back {tk[5]}]  sig rel err {es:.1e}  |U'U-I| {orth:.1e}  resid {resid:.1e}  vec diff {ev:.1e}")
