"""Diagnostic (not a test): fine-grained cycle counters of ONE bond step (TTN_PROF_STEP=k) of the fused apply+round, median over the
trains of the batch.  Usage: TTN_PROF_STEP=40 [TTN_WG512=1] python tools/diag_fine.py [batch]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TTN_PROF", "1")
import numpy as np
import ttn_amd as T
from ttn_amd import device as D
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
d, r = int(os.environ.get("TTN_D", 30)), int(os.environ.get("TTN_R", 64))
T.ensure_init(0)
A = T.Delta(d); dA = T.DeviceTTO(A)
x0 = T.rand_tt((2,) * d, r, seed=30)
dx = T.DeviceTT((2,) * d, x0.ttv_rks, batch=B)
for b in range(B):
    dx.upload(b, T.rand_tt((2,) * d, r, seed=30 + b))
dy = T.DeviceTT((2,) * d, [a * c for a, c in zip(A.tto_rks, x0.ttv_rks)], batch=B)
for it in range(2):
    D.apply_compress(dA, dx, dy, r); D.sync()
D.compress_status(dy)
out = (C.c_int64 * 64)()
rows = []
for tb in range(min(B, 64)):
    T._lib.check(T._lib.lib().ttn_prof_fine(tb, out))
    rows.append(list(out))
med = np.median(np.array(rows, dtype=np.float64), axis=0)
F = ["absmax x2", "syrk A'^T A'", "syrk B' B'^T", "diag test", "T3 = D^1/2 Gb D^1/2", "eig64", "kappa test + rank rule", "T1/T2 scaling", "Lf^T GEMM", "Rf GEMM",
     "check syrk Lf", "check syrk Rf", "check_diag x2", "copy Lf -> core", "copy Rf -> core"]
G = {26: "step prologue (before the merge)", 27: "fused merge (whole call)", 20: "(merge, Gram, eig ... up to the rank rule)", 21: "rank rule", 22: "kept-block test + Us scaling", 23: "Ro = Us^T M GEMM", 24: "check syrk Ro", 25: "check_diag"}
print(f"step {os.environ.get('TTN_PROF_STEP')}  batch {B}  build {'512' if os.environ.get('TTN_WG512') == '1' or (B > 256 and os.environ.get('TTN_WG512') != '0') else '1024'}: median ticks over {len(rows)} trains")
for i, nme in enumerate(F):
    if med[i] > 0: print(f"  F {i:2d} {nme:28s} {med[i]:10.0f}")
for i, nme in G.items():
    if med[i] > 0: print(f"  G {i:2d} {nme:28s} {med[i]:10.0f}")
e = med[32:48]
if e[2] > 0:
    names = ["tridiagonalisation", "bisection", "eigenvalues out", "twisted factorisations", "back-transformation", ""]
    ks = [k for k in range(2, 12) if e[k] > 0]
    for a, b_ in zip(ks[:-1], ks[1:]):
        print(f"  eig mark {a}->{b_} {e[b_] - e[a]:10.0f}")

m = med[48:56]
if m[5] > 0:
    for a, b_, nme in [(0, 1, "operator core + x staging + barrier"), (1, 2, "first pass: main loop (A loads + MFMAs)"), (2, 3, "first pass: epilogue"), (3, 4, "remaining passes"), (4, 5, "max reduction + barrier")]:
        print(f"  merge {nme:42s} {m[b_] - m[a]:10.0f}")
