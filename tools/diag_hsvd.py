"""Diagnostic (not a test): ttv_decomp timing, device vs the CPU oracle."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import ttn_amd as T
from oracle import tt_oracle as O
from tests.helpers import to_oracle

T.ensure_init(0)
rng = np.random.default_rng(0)
for d, B, rcap, lowrank in ((12, 256, 64, 0), (16, 64, 128, 0), (20, 4, 64, 6), (16, 256, 16, 4)):
    dims = (2,) * d
    if lowrank:
        base = [O.ttv_to_tensor(O.rand_tt(dims, lowrank, rng)) for _ in range(min(B, 4))]
        ts = np.stack([base[b % len(base)] for b in range(B)])
        tol = 1e-10 * np.max(np.abs(ts))
    else:
        ts = rng.standard_normal((B,) + dims)
        tol = 1e-12
    cap = [1] + [min(2 ** k, 2 ** (d - k), rcap) for k in range(1, d)] + [1]
    z = T.DeviceTT(dims, cap, batch=B)
    try:
        T.qtt.ttv_decomp_(z, ts, 1, tol)
        T.device.compress_status(z)
        t0 = time.time()
        T.qtt.ttv_decomp_(z, ts, 1, tol)         # synchronous (includes the host->device copy of the tensors)
        t_gpu = time.time() - t0
        t0 = time.time()
        ref = O.ttv_decomp(ts[0], 1, tol)
        t_cpu = time.time() - t0
        got = z.download(0)
        err = np.max(np.abs(O.ttv_to_tensor(to_oracle(got)) - ts[0])) / np.max(np.abs(ts[0]))
        print(f"d={d} batch={B} ranks max {max(got.ttv_rks)} (cpu {max(ref.ttv_rks)}): device {t_gpu * 1e3:.1f} ms total = {t_gpu * 1e3 / B:.3f} ms/tensor;"
              f" cpu oracle {t_cpu * 1e3:.1f} ms/tensor; rel err {err:.1e}")
    except T.TTNError as e:
        print(f"d={d} batch={B}: {e}")
