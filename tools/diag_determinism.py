"""Diagnostic (not a test): run the fused apply+round twice on the same batch and report every (train, step) whose route or Jacobi
sweep count differs between the two launches — the same inputs must take the same routes."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TTN_PROF", "1")
import ttn_amd as T
from ttn_amd import device as D
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
d, r = 30, 64
T.ensure_init(0)
A = T.Delta(d); dA = T.DeviceTTO(A)
x0 = T.rand_tt((2,) * d, r, seed=30)
dx = T.DeviceTT((2,) * d, x0.ttv_rks, batch=B)
for b in range(B):
    dx.upload(b, T.rand_tt((2,) * d, r, seed=30 + b))
dy = T.DeviceTT((2,) * d, [a * c for a, c in zip(A.tto_rks, x0.ttv_rks)], batch=B)
runs = []
for it in range(3):
    D.apply_compress(dA, dx, dy, r); D.sync()
    D.compress_status(dy)
    st = (C.c_int64 * 120)()
    rows = []
    for tb in range(B):
        T._lib.check(T._lib.lib().ttn_prof_steps(tb, st))
        rows.append([(int(v >> 48) & 3, int(v >> 32) & 0xffff, int(v) & 0xfff) for v in st[:2 * (d - 1)]])
    runs.append(rows)
ndiff = 0
for it in (1, 2):
    for tb in range(B):
        for k, (a, b) in enumerate(zip(runs[0][tb], runs[it][tb])):
            if a != b:
                ndiff += 1
                if ndiff <= 40: print(f"run {it} train {tb} step {k}: {'FGHD'[a[0]]}{a[1]}:{a[2]} vs {'FGHD'[b[0]]}{b[1]}:{b[2]}")
print("differences:", ndiff)
from collections import Counter
cnt = Counter()
for tb in range(B):
    for k, a in enumerate(runs[0][tb]):
        cnt[(k, 'FGHD'[a[0]], a[1])] += 1
print("routes by step (run 0):", " ".join(f"{k}:{rt}{p}x{n}" for (k, rt, p), n in sorted(cnt.items()) if p <= 96))
