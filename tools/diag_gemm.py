"""Diagnostic (not a test): cycles of the workgroup GEMM on one CU for the shapes of the C3 sweep, against the
fp64 MFMA peak of one CU (128 flop/clk)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ttn_amd as T
T.ensure_init(0)
L = T._lib.lib()
print("build:", "512-thread" if os.environ.get("TTN_WG512_SELFTEST") else "1024-thread", " grid:", os.environ.get("TTN_BENCH_GRID", "1"))
shapes = [  # (m, n, k, ta, tb, what)
    (64, 64, 128, 0, 1, "F: B' B'^T"), (128, 64, 64, 0, 0, "F: Lf = A' T1"), (64, 128, 64, 1, 0, "F: Rf = T2^T B'"), (64, 64, 384, 0, 1, "G: check Ro Ro^T"),
    (128, 384, 192, 0, 0, "merge L->R"), (128, 128, 384, 0, 1, "gram M M^T"), (64, 384, 128, 1, 0, "split Vt = U^T M"),
    (128, 128, 64, 0, 0, "merge R->L"), (64, 64, 128, 1, 0, "gram A^T A (F)"), (64, 64, 64, 0, 0, "small 64^3"),
    (128, 128, 128, 0, 0, "128^3"), (128, 64, 128, 0, 0, "U = X W"), (192, 192, 128, 0, 0, "dot-like"),
    # ta & 2: the Gram-product routine wg_syrk (flop counted as the full 2 m m k of the GEMM it replaces)
    (64, 384, 128, 5, 0, "ra: split Vt = U^T M"), (64, 128, 64, 5, 0, "ra: F Rf = T2^T B'"), (64, 128, 64, 4, 1, "ra: F Lf^T"),
    (128, 128, 384, 2, 0, "syrk M M^T"), (64, 64, 384, 2, 0, "syrk check / ramp"), (64, 64, 128, 2, 0, "syrk B' B'^T (F)"), (64, 64, 128, 3, 0, "syrk A'^T A' (F)"),
]
if os.environ.get("TTN_DIAG_SYRK_ONLY"): shapes = [sh for sh in shapes if (sh[3] & 6) or "split" in sh[5] or "F: Lf" in sh[5] or "F: Rf" in sh[5] or "gram" in sh[5] or "check" in sh[5] or "B' B'^T" in sh[5]]
for m, n, k, ta, tb, what in shapes:
    reps = 20
    cy = C.c_int64()
    T._lib.check(L.ttn_bench_gemm(m, n, k, ta, tb, reps, C.byref(cy)))
    per = cy.value / reps
    flop = 2.0 * m * n * k
    print(f"{what:20s} {m:4d}x{n:4d}x{k:4d} ta={ta} tb={tb}: {per:10.0f} clk  {flop/per:6.1f} flop/clk  ({100*flop/per/128:5.1f}% of CU peak)")

for n in (64, 128):
    reps = 10
    c0, c1, c2, sw = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    T._lib.check(L.ttn_bench_lds(0, n, reps, C.byref(c0), None))
    T._lib.check(L.ttn_bench_lds(1, n, reps, C.byref(c1), None))
    T._lib.check(L.ttn_bench_lds(2, n, reps, C.byref(c2), C.byref(sw)))
    print(f"LDS n={n}: setup {c0.value/reps:.0f} clk, Cholesky {(c1.value-c0.value)/reps:.0f} clk, "
          f"Jacobi {(c2.value-c0.value)/reps:.0f} clk in {sw.value} sweeps = {(c2.value-c0.value)/reps/max(sw.value,1):.0f} clk/sweep")
