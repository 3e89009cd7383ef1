"""Diagnostic (not a test): timing of the site-swap chains on the device vs the CPU oracle.
   python tools/diag_swap.py [d] [rx] [ry] [batch]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import ttn_amd as T
from oracle import tt_oracle as O
from tests.helpers import to_product

d = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rx = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ry = int(sys.argv[3]) if len(sys.argv) > 3 else 4
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
T.ensure_init(0)
rng = np.random.default_rng(0)
x = O.rand_tt((2,) * d, rx, rng)
y = O.rand_tt((2,) * d, ry, rng)
dx = T.DeviceTT.from_host(to_product(x), batch=B)
dy = T.DeviceTT.from_host(to_product(y), batch=B)
cap = 128
dz = T.DeviceTT((2,) * d, [1] + [cap] * (d - 1) + [1], batch=B)
for tol in (1e-10,):
    T.qtt.hadamard_ttm_(dx, dy, dz, tol=tol, work_cap=cap)
    T.device.compress_status(dz)
    with T.StreamTimer() as tm:
        T.qtt.hadamard_ttm_(dx, dy, dz, tol=tol, work_cap=cap)
    sweeps = T.device.compress_status(dz)
    t0 = time.time()
    ref = O.hadamard_ttm(x, y, tol=tol)
    t_cpu = time.time() - t0
    got = dz.download(0)
    nsw = d * (d - 1) // 2
    print(f"hadamard_ttm d={d} rx={rx} ry={ry} tol={tol}: ranks max {max(got.ttv_rks)} (cpu {max(ref.ttv_rks)}), {nsw} swaps + {d} contractions")
    print(f"  device: {tm.ms:.1f} ms for {B} trains = {tm.ms / B:.3f} ms/train, {B * nsw / tm.ms * 1e3:.0f} swap SVDs/s, Jacobi sweeps/train {sweeps[0]}")
    print(f"  cpu oracle (1 core): {t_cpu * 1e3:.1f} ms/train -> x{t_cpu * 1e3 / (tm.ms / B):.1f} per train-equivalent")
    hr = [a * b for a, b in zip(x.ttv_rks, y.ttv_rks)]
    need, _ = T.device.compress_rank_bound((2,) * d, hr, max(hr))
    h = T.DeviceTT((2,) * d, need, batch=B)
    with T.StreamTimer() as tm2:
        T.device.hadamard(dx, dy, h)
        T.device.tt_compress_(h, max(hr))
    print(f"  for comparison hadamard + tt_compress!(., {max(hr)}): {tm2.ms:.1f} ms for {B} trains")

# reorder
nd, bits, r = 2, d // 2, 8
N = nd * bits
rk = [1] + [min(r, 2 ** min(k % bits, bits - k % bits)) if k % bits else 1 for k in range(1, N)] + [1]   # separable in the two dimensions
q = O.rand_tt((2,) * N, rk, rng)
sw = T.bubble_sort_swaps(T.reorder_perm(nd, bits, "interleaved"))
dq = T.DeviceTT.from_host(to_product(q), batch=B, cap_rks=[1] + [cap] * (N - 1) + [1])
try:
    with T.StreamTimer() as tm:
        T.qtt.swap_sites_(dq, sw, 1e-12)
    T.device.compress_status(dq)
    t0 = time.time()
    ref = O.swap_sites_(O.copy_tt(q), sw, 1e-12)
    t_cpu = time.time() - t0
    print(f"reorder serial->interleaved {nd}x{bits} bits rank {r}: {len(sw)} swaps, ranks max {max(dq.download(0).ttv_rks)} (cpu {max(ref.ttv_rks)})")
    print(f"  device: {tm.ms:.1f} ms for {B} trains ({tm.ms / B:.3f} ms/train); cpu oracle {t_cpu * 1e3:.1f} ms/train")
except T.TTNError as e:
    print("reorder:", e)
