"""Diagnostic: distribution over a batch of the conditioning s1/s_kept of one bond step of the C3 sweep.
   python tools/diag_kappa_batch.py [B] [step]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ttn_amd as T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
step = int(sys.argv[2]) if len(sys.argv) > 2 else 24
d, r = 30, 64
T.ensure_init(0)
A = T.Delta(d); dA = T.DeviceTTO(A)
x0 = T.rand_tt((2,) * d, r, seed=30)
dx = T.DeviceTT((2,) * d, x0.ttv_rks, batch=B)
for b in range(B):
    dx.upload(b, T.rand_tt((2,) * d, r, seed=30 + b))
dy = T.DeviceTT((2,) * d, [a * c for a, c in zip(A.tto_rks, x0.ttv_rks)], batch=B)
dy.capture_singular_values(True)
T.device.apply_compress(dA, dx, dy, r, 0.0, 1)
T.device.compress_status(dy)
ks = []
for b in range(B):
    s = dy.singular_values(b, step)
    s = s[s >= 0]
    k = min(len(s), r)
    ks.append(s[0] / max(s[k - 1], 1e-300))
order = np.argsort(np.array(ks))
print("worst seeds (seed:kappa):", " ".join(f"{30 + int(b)}:{ks[b]:.3g}" for b in order[-16:]))
ks = np.sort(np.array(ks))
print(f"step {step}: kappa of the kept block over {B} trains: min {ks[0]:.3g} median {ks[B // 2]:.3g} 90% {ks[int(0.9 * B)]:.3g} 99% {ks[int(0.99 * B)]:.3g} max {ks[-1]:.3g}")
print("largest:", " ".join(f"{v:.3g}" for v in ks[-8:]))
