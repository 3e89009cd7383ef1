"""Diagnostic (not a test): run apply+round on a batch of distinct trains, print Jacobi status/sweeps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ttn_amd as T
from ttn_amd import device as D
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d, r = int(os.environ.get("TTN_D", 30)), int(os.environ.get("TTN_R", 64))
T.ensure_init(0)
A = T.Delta(d); dA = T.DeviceTTO(A)
x0 = T.rand_tt((2,) * d, r, seed=30)
dx = T.DeviceTT((2,) * d, x0.ttv_rks, batch=B)
for b in range(B):
    dx.upload(b, T.rand_tt((2,) * d, r, seed=30 + b))
dy = T.DeviceTT((2,) * d, [a * c for a, c in zip(A.tto_rks, x0.ttv_rks)], batch=B)
for it in range(2):
    D.sync()
    t0 = time.perf_counter()
    D.apply_compress(dA, dx, dy, r); D.sync()        # fused (TTN_NOFUSE=1: apply, then compress)
    t1 = time.perf_counter()
    try:
        sw = D.compress_status(dy); ok = True
    except Exception as e:
        ok = False; print("status:", e)
        import ctypes as C
        out = (C.c_int64 * B)(); T._lib.lib().ttn_compress_status(dy.h, out); sw = list(out)
    print(f"iter {it}: compress {1e3*(t1-t0):.1f} ms  ok={ok} sweeps min/max/mean {min(sw)}/{max(sw)}/{sum(sw)/len(sw):.1f}  -> {B*d/(t1-t0):.0f} cores/s")

if os.environ.get("TTN_PROF"):
    import ctypes as C
    out = (C.c_int64 * 16)()
    T._lib.check(T._lib.lib().ttn_prof_get(0, out))
    names = ["merge", "scale", "LQ/chol", "jacobi", "sort", "split", "F:rest", "gramG", "F:grams", "F:chols", "F:jacobi", "G:chol"]
    tot = sum(out[:12])
    print("phase ticks (100MHz):", {n: int(v) for n, v in zip(names, out)}, "total ms", tot / 1e5)
    st = (C.c_int64 * 120)()
    T._lib.check(T._lib.lib().ttn_prof_steps(0, st))
    print("per step (route p:sweeps@kclk):", " ".join(f"{'FGHD'[(v >> 48) & 3]}{(v >> 32) & 0xffff}:{v & 0xfff}@{(v >> 12) & 0xfffff}" for v in st[:2 * (d - 1)]))
    for tb in range(min(B, 8)):
        T._lib.check(T._lib.lib().ttn_prof_get(tb, out))
        print(f"train {tb}: route-F steps tested for a diagonal left Gram {out[13]}, taken {out[14]}, largest off-diagonal level {out[12] * 1e-18:.2e}; Gram steps finished by the Jacobi polish {out[15]}")
    if os.environ.get("TTN_STEP"):
        k = int(os.environ["TTN_STEP"])
        rows = []
        for tb in range(min(B, 64)):
            T._lib.check(T._lib.lib().ttn_prof_steps(tb, st))
            v = st[k]
            tot = sum(((w >> 12) & 0xfffff) for w in st[:2 * (d - 1)])
            rows.append(f"{'FGHD'[(v >> 48) & 3]}{(v >> 32) & 0xffff}@{(v >> 12) & 0xfffff}/{tot}")
        print(f"step {k} per train (route p@kclk/total kclk):", " ".join(rows))
    if os.environ.get("TTN_OUTLIERS"):
        tots = []
        for tb in range(B):
            T._lib.check(T._lib.lib().ttn_prof_steps(tb, st))
            steps = list(st[:2 * (d - 1)])
            tots.append((sum(((w >> 12) & 0xfffff) for w in steps), tb, steps))
        tots.sort()
        print("per-train total kclk: min", tots[0][0], "median", tots[len(tots) // 2][0], "max", tots[-1][0])
        med = tots[len(tots) // 2][2]
        for tot, tb, steps in tots[-4:]:
            diff = [f"{k}:{'FGHD'[(v >> 48) & 3]}{(v >> 32) & 0xffff}@{(v >> 12) & 0xfffff}(median {(m >> 12) & 0xfffff})" for k, (v, m) in enumerate(zip(steps, med))
                    if abs(((v >> 12) & 0xfffff) - ((m >> 12) & 0xfffff)) > 150]
            print(f"train {tb}: total {tot}: " + " ".join(diff))
