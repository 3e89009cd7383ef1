import torch, time
n = 3 * 2**30 // 8
a = torch.empty(n, dtype=torch.float64, device="cuda"); b = torch.empty(n, dtype=torch.float64, device="cuda")
def t(f, reps=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
tw = t(lambda: a.fill_(1.0)); tc = t(lambda: b.copy_(a)); tr = t(lambda: a.sum())
print(f"fill (write only) {n*8/tw/1e12:.2f} TB/s; copy (read+write) {2*n*8/tc/1e12:.2f} TB/s; sum (read only) {n*8/tr/1e12:.2f} TB/s")
