#!/usr/bin/env python3
"""bench.py — "QTT Laplacian apply + round" throughput in TT cores per second on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--d 30] [--rank 64]
                    [--shard trains|cores] [--op compress|apply|hadamard|add|scale|dot|orthogonalize]

One STEP = one pass of the hot path over one batch of B independent synthetic trains resident in HBM:
    y_b = tt_compress!(Δ(d) * x_b, rank)       for b = 1..B      (src/solvers/euler.jl:55's operator)
as ONE launch of ttn_apply_compress: the apply (src/tt_operations.jl:101-111) is fused into the first L->R
sweep of the persistent k_compress kernel (src/tt_tools.jl:743-789), y = Δx never exists in HBM.
value = (#ranks * B * d) / (max-over-ranks wall time per step)   [TT cores / s], inputs already in HBM.

Multi-GPU: one process per GPU.  Launched by `python -m torch.distributed.run ... bench.py --gpus N` (the driver) the
ranks come from RANK / WORLD_SIZE; launched as plain `python bench.py --gpus N` this process starts the N workers ITSELF
(a torchrun child, before anything here touches a GPU) and relays their output.  A line whose n_gpus would differ from
--gpus is refused.  Trains are independent, so they shard across ranks with no data-path collective (weak scaling:
B trains per GPU); torch.distributed (RCCL) carries only barriers and the max-over-ranks timing reduction.
`--shard cores` measures the core-wise sharded pipeline (BASELINE config C4) instead.

The JSON line also carries
  roofline     for the dominant kernel (k_compress): algorithmic fp64 flops of the sweep (merge GEMMs +
               Golub–Van-Loan thin-SVD count, evaluated on the actual rank profile) x B per launch / average launch
               duration measured with HIP events on the library's stream; `traffic` = HBM bytes per launch from the
               committed rocprofv3 PMC passes of this command (profiles/k_compress_traffic.json, written by
               scratch/collect_profiles.sh — counters cannot be read from inside the process);
  cpu_baseline the CPU oracle ("port": NumPy + LAPACK gesdd, the reference algorithm WITHOUT the discarded per-bond
               orthogonalize) timed on this box's host cores: one train per core on all cores, and one train with
               all BLAS threads; value = the best;
  batch_sweep  the same step at B = 1, 8, 64, 256 (B = 1: latency of one train);
  verified     number of trains of the timed batch downloaded after the timed region and checked against the oracle
               (ranks exact, tensor difference <= 1e-9).
`--op X` prints one roofline line for another hot-path kernel instead (HBM-bound: apply / hadamard / add / scale;
fp64-MFMA-bound: dot / orthogonalize) on C3-shaped batches.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6     # AMD MI355X public fp64 vector = matrix peak; scratch/mfma_peak.hip measures 128 flop/clk/CU = the same
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E ~8 TB/s (about 6.3 TB/s achievable by a read stream)


def sweep_algorithmic_flops(d, rks_in, rks_out_after_lr, max_bond, n=2):
    """Algorithmic fp64 flops of one tt_compress! (sweeps=1, truncerr=0) on input ranks `rks_in`:
    per bond step 2*mr*mc*rm (merge GEMM, src/tt_tools.jl:749) + thin-SVD Golub–Van-Loan count
    6*M*N^2 + 20*N^3 with M >= N."""
    def svd_flops(a, b):
        M, N = max(a, b), min(a, b)
        return 6.0 * M * N * N + 20.0 * N ** 3
    total = 0.0
    r = list(rks_in)
    for k in range(d - 1):                       # L -> R
        mr, mc, rm = n * r[k], n * r[k + 2], r[k + 1]
        total += 2.0 * mr * mc * rm + svd_flops(mr, mc)
        r[k + 1] = min(mr, mc, max_bond)
    for k in range(d - 2, -1, -1):               # R -> L
        mr, mc, rm = n * r[k], n * r[k + 2], r[k + 1]
        total += 2.0 * mr * mc * rm + svd_flops(mr, mc)
        r[k + 1] = min(mr, mc, max_bond)
    return total


# ------------------------------------------------------------------------------------------------------------------
# cpu_baseline leg (the ONLY place besides `verify` where bench.py touches oracle/, and only as the thing timed beside
# the GPU number / the checker — never in the product path)
# ------------------------------------------------------------------------------------------------------------------
def _cpu_worker(args):
    d, rank, seed, reps, faithful, threads = args
    from threadpoolctl import threadpool_limits
    import ttn_amd as T
    from oracle import tt_oracle as O
    from tests.helpers import to_oracle
    with threadpool_limits(limits=threads):
        x = to_oracle(T.rand_tt((2,) * d, rank, seed=seed))
        A = O.Delta(d)
        t0 = time.perf_counter()
        for _ in range(reps):
            O.tt_compress_(O.apply(A, x), rank, faithful=faithful)
        return time.perf_counter() - t0


def cgroup_cpu_quota():
    """Effective CPU quota of this process's cgroup in cores (cpu.max = "<quota> <period>" / "max <period>"), or None."""
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            with open(path) as fh:
                q, per = fh.read().split()[:2]
            return None if q == "max" else float(q) / float(per)
        except (OSError, ValueError):
            pass
    try:                                            # cgroup v1
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
            q = float(fh.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
            per = float(fh.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def host_info():
    info = {"nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "cgroup_cpu_quota_cores": cgroup_cpu_quota()}
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    info["cpu_model"] = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        from threadpoolctl import threadpool_info
        import numpy  # noqa: F401  (loads the BLAS the oracle uses)
        import scipy.linalg  # noqa: F401
        info["blas"] = sorted({"%s %s (%s, %s threads)" % (p.get("internal_api"), p.get("version"), p.get("architecture"), p.get("num_threads"))
                               for p in threadpool_info() if p.get("user_api") == "blas"})
    except Exception:
        pass
    return info


def cpu_baseline(d, rank, budget_s=20.0):
    """Oracle (reference algorithm, lean: without the discarded orthogonalize) on ALL host cores this process may use:
    (i) one independent train per process, 1 BLAS thread each; (ii) one train with all BLAS threads.  value = best."""
    import multiprocessing as mp
    cores = max(1, len(os.sched_getaffinity(0)))
    quota = cgroup_cpu_quota()
    if quota:                                       # more processes than twice the quota only add scheduling noise
        cores = max(1, min(cores, 2 * int(quota + 0.999)))
    t1 = _cpu_worker((d, rank, 30, 1, False, 1))                  # calibrate on one train, one thread
    reps = max(1, min(8, int(budget_s / max(t1, 1e-3))))
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_worker, [(d, rank, 30, 1, False, 1)] * cores)           # warm the workers (imports, page-in)
        t0 = time.perf_counter()
        pool.map(_cpu_worker, [(d, rank, 30 + c, reps, False, 1) for c in range(cores)])
        wall = time.perf_counter() - t0
    multi = cores * reps * d / wall
    single = d / t1
    tall = _cpu_worker((d, rank, 30, max(1, min(3, reps)), False, cores)) / max(1, min(3, reps))
    one_train_all_threads = d / tall
    # `cores` in the record = the EFFECTIVE parallelism: the box's CPU share is a cgroup quota far below the visible hardware
    # threads, so N processes deliver multi / single cores' worth of work, not N (round 2 printed the process count: 256)
    eff = max(1.0, multi / single)
    cands = [(multi, round(eff, 1), "%d procs x 1 BLAS thread" % cores), (single, 1, "1 proc x 1 BLAS thread"),
             (one_train_all_threads, round(max(1.0, one_train_all_threads / single), 1), "1 proc x %d BLAS threads" % cores)]
    value, used, how = max(cands)
    # the reference-FAITHFUL flavour (with the orthogonalize that _tt_bond_truncate! computes and tt_compress! discards,
    # src/tt_tools.jl:769,779) on one train and one core, if the lean timing says it fits ~15 s
    faithful = None
    if t1 * 12 < 15.0:
        try:
            faithful = d / _cpu_worker((d, rank, 30, 1, True, 1))
        except Exception:
            faithful = None
    out = {"value": round(value, 2), "unit": "TT cores/s", "cores": used, "kind": "port",
           "sample": f"{cores} procs x {reps} trains of d={d} rank={rank} (lean oracle = reference algorithm without the discarded "
                     f"orthogonalize, NumPy+LAPACK gesdd); best of: {how}",
           "trains_per_core_all_cores": round(multi, 2), "one_train_one_thread": round(single, 2),
           "one_train_all_threads": round(one_train_all_threads, 2), "processes": cores,
           "parallel_efficiency": round(multi / single / cores, 4), "host": host_info()}
    if faithful:
        out["faithful_single_core"] = round(faithful, 2)
    return out


def verify_against_oracle(T, dy, seeds_by_slot, d, r, slots):
    """Download `slots` of the timed batch and compare with the oracle on the same seeds: ranks exact, ||y_gpu - y_cpu|| / ||y_cpu||
    <= 1e-9 (the parity bar of tests/test_gpu_parity.py).  Raises on a mismatch; returns the number of trains checked."""
    from oracle import tt_oracle as O
    from tests.helpers import to_oracle, tt_rel_diff
    A = O.Delta(d)
    worst = 0.0
    for b in slots:
        got = to_oracle(dy.download(b))
        ref = O.tt_compress_(O.apply(A, to_oracle(T.rand_tt((2,) * d, r, seed=seeds_by_slot[b]))), r)
        if got.ttv_rks != ref.ttv_rks:
            raise SystemExit(f"bench.py: train {b} (seed {seeds_by_slot[b]}): ranks {got.ttv_rks} != oracle {ref.ttv_rks}")
        err = tt_rel_diff(got, ref)
        worst = max(worst, err)
        if not err <= 1e-9:
            raise SystemExit(f"bench.py: train {b} (seed {seeds_by_slot[b]}): rel. difference to the oracle {err:.3e} > 1e-9")
    return len(slots), worst


# ------------------------------------------------------------------------------------------------------------------
# the compress step on one configuration (headline C3, sub-record C2)
# ------------------------------------------------------------------------------------------------------------------
def run_compress_config(T, D, d, r, B, steps, warmup, rank, world, barrier, dist, red_device, verify):
    """W untimed + K timed launches of ttn_apply_compress over B synthetic trains per rank; returns the measured record of
    this rank's view (value = whole job: max-over-ranks wall time)."""
    A = T.Delta(d)
    dA = T.DeviceTTO(A)
    x0 = T.rand_tt((2,) * d, r, seed=30)
    dx = T.DeviceTT((2,) * d, x0.ttv_rks, batch=B)
    seeds = [30 + g for g in T.shard.weak_train_ids(rank, world, B)]     # distinct synthetic trains, seeds 30 + global index
    for b, sd in enumerate(seeds):
        dx.upload(b, T.rand_tt((2,) * d, r, seed=sd))
    ycap = [a * c for a, c in zip(A.tto_rks, x0.ttv_rks)]
    dy = T.DeviceTT((2,) * d, ycap, batch=B)
    for _ in range(warmup):
        D.apply_compress(dA, dx, dy, r)
    D.compress_status(dy)                                      # raises if any Jacobi SVD failed to converge
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        # ttn_apply_compress: apply fused into the first L->R sweep of k_compress (y = A*x never exists in HBM)
        D.event_record(2 * i)
        D.apply_compress(dA, dx, dy, r)
        D.event_record(2 * i + 1)
    barrier_free_sync(D)
    elapsed = T.shard.max_over_ranks(time.perf_counter() - t0, dist, device=red_device)
    if dist is not None:
        dist.barrier()
    sweeps = D.compress_status(dy)
    kms = [D.event_elapsed_ms(2 * i, 2 * i + 1) for i in range(steps)]
    k_avg_s = sum(kms) / len(kms) / 1e3
    out_rks, _ = dy.ranks(0)
    flops = sweep_algorithmic_flops(d, ycap, None, r) * B
    rec = {"d": d, "rank": r, "batch": B, "steps": steps, "warmup": warmup, "elapsed": elapsed, "ms_per_step": round(elapsed / steps * 1e3, 3),
           "value": round(T.shard.cores_per_second(world, B, d, elapsed / steps), 1), "k_avg_s": k_avg_s, "flops": flops,
           "achieved": flops / k_avg_s / 1e12, "frac": round(flops / k_avg_s / 1e12 / FP64_PEAK_TFLOPS, 4), "out_rks": out_rks,
           "jacobi_sweeps": sweeps[0]}
    if verify:
        slots = sorted({0, B // 2, B - 1})
        n, worst = verify_against_oracle(T, dy, seeds, d, r, slots)
        rec["verified"], rec["verified_max_rel_diff"] = n, float("%.3e" % worst)
    dx.free(); dy.free(); dA.free()
    return rec


def barrier_free_sync(D):
    import torch
    D.sync()
    torch.cuda.synchronize()


def batch_sweep(T, D, d, r):
    """The same step at B = 1, 8, 64, 128, 256 (B = 1: latency of one train).  At 128 and 256 trains both kernel builds are timed
    (TTN_WG512 is read per launch): `build` names the one the library picks by itself."""
    A = T.Delta(d)
    dA = T.DeviceTTO(A)
    x0 = T.rand_tt((2,) * d, r, seed=30)
    ycap = [a * c for a, c in zip(A.tto_rks, x0.ttv_rks)]
    out = []
    for Bs in (1, 8, 64, 128, 256):
        sx = T.DeviceTT((2,) * d, x0.ttv_rks, batch=Bs)
        for b in range(Bs):
            sx.upload(b, T.rand_tt((2,) * d, r, seed=30 + b))
        sy = T.DeviceTT((2,) * d, ycap, batch=Bs)

        def timed():
            for _ in range(2):
                D.apply_compress(dA, sx, sy, r)
            D.sync()
            nrep = 5
            ts = time.perf_counter()
            for _ in range(nrep):
                D.apply_compress(dA, sx, sy, r)
            D.sync()
            tb = (time.perf_counter() - ts) / nrep
            D.compress_status(sy)
            return tb
        tb = timed()
        rec = {"batch": Bs, "value": round(Bs * d / tb, 1), "unit": "TT cores/s", "ms_per_step": round(tb * 1e3, 3)}
        if Bs >= 128 and "TTN_WG512" not in os.environ:
            alt = {}
            for flag in ("0", "1"):
                os.environ["TTN_WG512"] = flag
                alt["wg1024" if flag == "0" else "wg512"] = round(timed() * 1e3, 3)
            del os.environ["TTN_WG512"]
            rec["ms_per_step_by_build"] = alt
        out.append(rec)
        sx.free()
        sy.free()
    dA.free()
    return out


def drop_in_latency(T, d, r):
    """What a drop-in user of the STATELESS entry points gets for ONE train (julia/TTNBackend.jl binds `*` and `tt_compress!` to
    them: src/tt_operations.jl:101, src/tt_tools.jl:772): host buffers in, host buffers out — ttn_apply_f64 (upload x and A,
    download the 11.4 MB y) then ttn_compress_f64 (upload y, download the result).  PCIe-inclusive wall time, never `value`."""
    A = T.Delta(d)
    x = T.rand_tt((2,) * d, r, seed=30)
    t_apply, t_comp = [], []
    for _ in range(4):
        t0 = time.perf_counter()
        y = A * x                                   # ttn_apply_f64
        t1 = time.perf_counter()
        T.tt_compress_(y, r)                        # ttn_compress_f64
        t2 = time.perf_counter()
        t_apply.append(t1 - t0)
        t_comp.append(t2 - t1)
    ta, tc = min(t_apply[1:]), min(t_comp[1:])
    t_fused = []
    for _ in range(4):                              # ttn_apply_compress_f64: the same op in one call, A*x never crosses PCIe
        t0 = time.perf_counter()
        T.apply_compress(A, x, r)
        t_fused.append(time.perf_counter() - t0)
    tf = min(t_fused[1:])
    return {"what": "ttn_apply_f64 + ttn_compress_f64 on one train, host buffers in and out (PCIe-inclusive; best of 3 after one warm-up)",
            "ms_apply": round(ta * 1e3, 3), "ms_compress": round(tc * 1e3, 3), "ms_total": round((ta + tc) * 1e3, 3),
            "value": round(d / (ta + tc), 1), "unit": "TT cores/s",
            "fused_call": {"what": "ttn_apply_compress_f64: tt_compress!(A * x, max_bond) as one stateless call", "ms_total": round(tf * 1e3, 3),
                           "value": round(d / tf, 1)}}


def pmc_busy_fractions():
    """MFMA-pipe and VALU busy fractions of k_compress from the newest committed PMC summary (profiles/r*_pmc.json, separate
    rocprofv3 --pmc passes of the default bench command): what the chip actually does, beside the algorithmic `frac`."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json"))):
        try:
            with open(path) as fh:
                j = json.load(fh)
            dv = j.get("derived") or {}
            if "mfma_pipe_busy_frac" in dv:
                best = {"file": os.path.relpath(path, ROOT), "mfma_pipe_busy_frac": round(dv["mfma_pipe_busy_frac"], 4),
                        "valu_busy_frac": round(dv.get("valu_busy_frac", 0.0), 4),
                        "mfma_f64_mops_per_launch": j.get("k_compress", {}).get("SQ_INSTS_VALU_MFMA_MOPS_F64_per_launch")}
        except (OSError, ValueError):
            pass
    return best


def headline_record(args, head, d, r, B, world):
    traffic, traffic_upper, traffic_src = None, None, None
    tpath = os.path.join(ROOT, "profiles", "k_compress_traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as fh:
            tj = json.load(fh)
        if (tj.get("d"), tj.get("rank"), tj.get("batch"), 1) == (d, r, B, world):
            traffic = tj.get("traffic_bytes_per_launch")
            traffic_upper = tj.get("traffic_bytes_per_launch_upper")
            traffic_src = {"file": "profiles/k_compress_traffic.json", "tag": tj.get("tag"), "commit": tj.get("commit"),
                           "note": "profile-derived (separate --pmc passes), not measured by this run; traffic = WRITE + raw FETCH_SIZE "
                                   "(lower figure), traffic_upper = WRITE + 2 x FETCH_SIZE (gfx950 counts wide reads at half)"}
    res = {
        "metric": "TT cores/sec for QTT Laplacian apply+round, d=%d rank-%d" % (d, r),
        "value": head["value"], "unit": "TT cores/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C3: tt_compress!(Delta(%d)*x, %d), x = rand_tt(dims=2^%d, rank %d), batch of %d independent "
                               "trains per GPU resident in HBM (seeds 30+i), truncerr=0, sweeps=1" % (d, r, d, r, B),
                   "d": d, "rank": r, "batch_per_gpu": B, "parallelism": "trains sharded over %d GPU(s), no collective" % world,
                   "out_ranks": head["out_rks"]},
        "roofline": {"bound": "mfma", "kernel": "k_compress", "achieved": round(head["achieved"], 3), "peak": FP64_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": head["frac"], "traffic": traffic, "traffic_upper": traffic_upper, "traffic_source": traffic_src,
                     "achieved_is": "ALGORITHMIC-equivalent TFLOP/s: SURVEY 8d's flop count (merge GEMMs + Golub-Van-Loan dense thin-SVD count) / "
                                    "measured launch time; the kernel executes far fewer flops than that count (Gram / eigensolver routes): "
                                    "see pipe_utilisation for what the matrix pipe really does",
                     "pipe_utilisation": pmc_busy_fractions(),
                     "algorithmic_flops_per_launch": head["flops"], "avg_launch_ms": round(head["k_avg_s"] * 1e3, 3),
                     "bond_steps_per_s": round(2 * (d - 1) * B / head["k_avg_s"], 1),
                     "jacobi_sweeps_per_train": head["jacobi_sweeps"]},
    }
    if "verified" in head:
        res["verified"] = head["verified"]
        res["verified_max_rel_diff"] = head["verified_max_rel_diff"]
    return res


def guarded_core_sharded(args, T, D, torch, dist, rank, world, red_device, head, extras, c2):
    """BASELINE config C4 inside the N > 1 line: the core-wise sharded pipeline over the same ranks, after the headline
    measurement.  Its RCCL point-to-point transport has never run on real multi-GPU hardware, so a WATCHDOG bounds it: if the
    section has not finished after --core-sharded-timeout seconds, rank 0 prints the headline line with the failure recorded and
    every rank leaves the process — a hang here must not cost the run its headline number."""
    import threading
    done = threading.Event()

    def watchdog():
        if done.wait(args.core_sharded_timeout):
            return
        if rank == 0:
            res = headline_record(args, head, args.d, args.rank, args.batch, world)
            res.update(extras)
            if c2 is not None:
                res["c2"] = {"value": c2["value"], "ms_per_step": c2["ms_per_step"], "frac": c2["frac"], "verified": c2.get("verified")}
            res["core_sharded"] = {"error": "core-wise sharded section did not finish within %d s (watchdog)" % args.core_sharded_timeout}
            print(json.dumps(res), flush=True)
        os._exit(0)

    th = threading.Thread(target=watchdog, daemon=True)
    th.start()
    try:
        rec = core_sharded_record(args, T, D, torch, dist, rank, world, red_device, B=min(args.batch, 256), steps=3, warmup=1)
    except Exception as exc:                        # a failure of the sub-record is recorded, not fatal
        rec = {"error": "%s: %s" % (type(exc).__name__, exc)}
    done.set()
    return rec



# ------------------------------------------------------------------------------------------------------------------
# launcher
# ------------------------------------------------------------------------------------------------------------------
def launch_workers(n, argv):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks as a FRESH child (torch.distributed.run)
    — this process has not touched a GPU and never will — relay its output and exit with its code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def core_sharded_record(args, T, D, torch, dist, rank, world, red_device, B=None, steps=None, warmup=None):
    """Every chain cut core-wise into `world` segments (pipeline.py), micro-batches of B trains pipelined through the ranks with
    one boundary-core hand-off per segment boundary, direction and micro-batch.  One STEP = tt_compress!(Delta*x, r) of ALL
    micro-batches; value = microbatches * B * d / max-over-ranks step time (strong scaling of a fixed set of chains: the work
    per GPU shrinks with N).  Returns the record on rank 0, None elsewhere.  No roofline/cpu legs: the kernels are the same."""
    import numpy as np
    from ttn_amd import pipeline as PL
    d, r = args.d, args.rank
    B = args.batch if B is None else B
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    M = args.microbatches if args.microbatches > 0 else 2 * world
    A = T.Delta(d)
    lo, hi = PL.extended_range(d, rank, world)
    backend = PL.DeviceBackend()
    comm_dev = "cuda" if args.backend == "nccl" else "cpu"
    transport = PL.DistTransport(dist, comm_dev) if dist is not None else None
    A_cores = [np.asfortranarray(c) for c in A.tto_vec[lo:hi]]
    prepared = []
    for m in range(M):
        trains = [[np.asfortranarray(c) for c in T.rand_tt((2,) * d, r, seed=30 + m * B + b).ttv_vec[lo:hi]] for b in range(B)]
        prepared.append(backend.prepare(A_cores, A.tto_rks[lo:hi + 1], trains, (2,) * (hi - lo)))

    def barrier():
        D.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def step():
        return PL.sharded_apply_compress(backend, transport, rank, world, prepared, hi - lo, r)

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        segs = step()
    D.sync()
    torch.cuda.synchronize()
    elapsed = T.shard.max_over_ranks(time.perf_counter() - t0, dist, device=red_device)
    if dist is not None:
        dist.barrier()
    rks_local = segs[0].ranks(0)[0]
    for prep in prepared:
        for h in prep:
            h.free()
    if rank != 0:
        return None
    return {
        "metric": "TT cores/sec for QTT Laplacian apply+round, d=%d rank-%d" % (d, r),
        "value": round(M * B * d / (elapsed / steps), 1), "unit": "TT cores/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C4: tt_compress!(Delta(%d)*x, %d), every chain cut core-wise into %d segments, %d micro-batches "
                               "of %d trains pipelined through the ranks, boundary-core hand-offs over %s" %
                               (d, r, world, M, B, "RCCL (xGMI)" if args.backend == "nccl" else args.backend),
                   "d": d, "rank": r, "batch_per_microbatch": B, "microbatches": M,
                   "parallelism": "core-wise pipeline over %d GPU(s)" % world, "rank0_segment_out_ranks": rks_local},
    }


def bench_core_sharded(args, T, D, torch, dist, rank, world, red_device):
    """--shard cores: the core-wise sharded pipeline (BASELINE config C4) as the line of its own."""
    res = core_sharded_record(args, T, D, torch, dist, rank, world, red_device)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------
# --op: roofline lines of the other hot-path kernels
# ------------------------------------------------------------------------------------------------------------------
def bench_op(args, T, D, emit=True):
    """One C3-shaped batch, one kernel: bytes (or flops) / HIP-event time / peak.  Algorithmic bytes = the cores read + the cores
    written, once each; algorithmic flops of dot = sum_k 2 n (rA rB rB' + rA rA' rB'), of orthogonalize = the
    Householder QR/LQ count 2 m n^2 - 2/3 n^3 per core plus forming Q (the same again) plus the R / L carry GEMM."""
    import numpy as np
    d, r, B, op = args.d, args.rank, args.batch, args.op
    dims = (2,) * d
    A = T.Delta(d)
    dA = T.DeviceTTO(A)
    x0 = T.rand_tt(dims, r, seed=30)
    xr = list(x0.ttv_rks)
    dx = T.DeviceTT(dims, xr, batch=B)
    for b in range(B):
        dx.upload(b, T.rand_tt(dims, r, seed=30 + b))
    core_bytes = lambda rk: 8.0 * sum(2 * rk[k] * rk[k + 1] for k in range(d))          # noqa: E731
    # orthogonalize of rank <= 64 QTT trains is three kernels (csrc/ttn_ortho_ramp.h, ttn_ortho512.h, the 1024-thread k_orthogonalize for
    # the left sweep and for trains the other two refuse): avg_launch_ms spans all of them, first launch to last (ttn_last_launch_ms)
    kernel_names = {"dot": "k_dot_fused", "orthogonalize": "k_ortho_ramp + k_ortho512 (+ k_orthogonalize)"}
    bound, flops, nbytes, kernel = "hbm", 0.0, 0.0, kernel_names.get(op, "k_" + op)
    if op == "apply":
        yr = [a * c for a, c in zip(A.tto_rks, xr)]
        dy = T.DeviceTT(dims, yr, batch=B)
        run = lambda: D.apply(dA, dx, dy)                                                 # noqa: E731
        nbytes = B * (core_bytes(xr) + core_bytes(yr)) + 8.0 * sum(c.size for c in A.tto_vec)
    elif op == "hadamard":
        # the Hadamard square of a rank-r train has rank r^2: C3-shaped INPUT ranks would need 4096-rank outputs (2 GB per train);
        # use the rank profile min(8, ...) -> output ranks 64, the size class of the headline trains
        h0 = T.rand_tt(dims, 8, seed=30)
        hr = list(h0.ttv_rks)
        da = T.DeviceTT.from_host(h0, batch=B)
        db = T.DeviceTT.from_host(T.rand_tt(dims, 8, seed=31), batch=B)
        zr = [p * q for p, q in zip(hr, hr)]
        dz = T.DeviceTT(dims, zr, batch=B)
        run = lambda: D.hadamard(da, db, dz)                                              # noqa: E731
        nbytes = B * (2 * core_bytes(hr) + core_bytes(zr))
    elif op == "add":
        dx2 = T.DeviceTT(dims, xr, batch=B)
        D.scale(1.0, dx, dx2)
        zr = [p + q for p, q in zip(xr, xr)]
        zr[0] = zr[-1] = 1
        dz = T.DeviceTT(dims, zr, batch=B)
        run = lambda: D.add(dx, dx2, dz)                                                  # noqa: E731
        nbytes = B * (2 * core_bytes(xr) + core_bytes(zr))
    elif op == "scale":
        dz = T.DeviceTT(dims, xr, batch=B)
        run = lambda: D.scale(1.5, dx, dz)                                                # noqa: E731
        nbytes = B * 2 * core_bytes(xr)
    elif op == "dot":
        bound = "mfma"
        run = lambda: D.dot(dx, dx)                                                       # noqa: E731
        flops = B * sum(2.0 * 2 * (xr[k] * xr[k] * xr[k + 1] + xr[k] * xr[k + 1] * xr[k + 1]) for k in range(d))
        nbytes = B * 2 * core_bytes(xr)
    elif op == "orthogonalize":
        bound = "mfma"
        dz = T.DeviceTT(dims, xr, batch=B)
        run = lambda: D.orthogonalize(dx, 1, dz)                                          # noqa: E731
        for k in range(1, d):           # right-to-left LQ sweep to site 1: core k as r_{k} x (n r_{k+1}) -> m = n r_{k+1}, n_ = r_k
            m_, n_ = 2 * xr[k + 1], xr[k]
            if m_ < n_:
                m_, n_ = n_, m_
            flops += 2.0 * (2.0 * m_ * n_ * n_ - 2.0 / 3.0 * n_ ** 3) + 2.0 * 2 * xr[k - 1] * xr[k] * n_
        flops *= B
        nbytes = B * 2 * core_bytes(xr)
    else:
        raise SystemExit("unknown --op " + op)
    for _ in range(max(1, args.warmup)):
        run()
    D.sync()
    ms = []
    for i in range(args.steps):
        D.event_record(2 * i)
        run()
        D.event_record(2 * i + 1)
    D.sync()
    ms = [D.event_elapsed_ms(2 * i, 2 * i + 1) for i in range(args.steps)]
    call_ms = float(np.median(ms))
    if op in ("dot", "orthogonalize"):
        # ttn_dot is a SYNCHRONOUS call (it returns host values: launch, device-to-host copy, stream sync), so an event pair around
        # it also times the host's wake-up; the library brackets the kernel itself (ttn_last_launch_ms) — that is the launch duration
        ms = []
        for _ in range(args.steps):
            run()
            ms.append(D.last_launch_ms())
    t = float(np.median(ms)) / 1e3
    if bound == "hbm":
        achieved, peak, unit = nbytes / t / 1e9, HBM_PEAK_GBS, "GB/s"
    else:
        achieved, peak, unit = flops / t / 1e12, FP64_PEAK_TFLOPS, "TFLOP/s"
    res = {"metric": "TT cores/sec, %s on C3-shaped trains (d=%d rank-%d)" % (op, d, r), "value": round(B * d / t, 1), "unit": "TT cores/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(t * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "%s over a batch of %d trains resident in HBM" % (op, B), "d": d, "rank": r, "batch_per_gpu": B},
           "roofline": {"bound": bound, "kernel": kernel, "achieved": round(achieved, 3), "peak": peak, "unit": unit,
                        "frac": round(achieved / peak, 4), "traffic": None, "algorithmic_bytes_per_launch": nbytes,
                        "algorithmic_flops_per_launch": flops, "avg_launch_ms": round(t * 1e3, 4),
                        "call_ms_incl_host_sync": round(call_ms, 4)}}
    if emit:
        print(json.dumps(res), flush=True)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024, help="independent trains per GPU per step")
    ap.add_argument("--d", type=int, default=30)
    ap.add_argument("--rank", type=int, default=64)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-single", action="store_true", help="skip the batch sweep B = 1, 8, 64, 256")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-run check of downloaded trains against the oracle")
    ap.add_argument("--no-c2", action="store_true", help="skip the C2 (d=20, rank 32) sub-record")
    ap.add_argument("--no-ops", action="store_true", help="skip the other_kernels sub-record (apply / add / scale / hadamard / dot / orthogonalize)")
    ap.add_argument("--no-core-sharded", action="store_true", help="N>1: skip the core-wise sharded (C4) sub-record")
    ap.add_argument("--core-sharded-timeout", type=int, default=240, help="N>1: watchdog of the C4 sub-record in seconds")
    ap.add_argument("--shard", default="trains", choices=["trains", "cores"],
                    help="N>1: 'trains' (default) = independent trains per GPU, no data-path collective; 'cores' = every chain cut "
                         "core-wise into N segments with boundary-core hand-offs (pipeline.py; micro-batches of --batch trains)")
    ap.add_argument("--microbatches", type=int, default=0, help="--shard cores: micro-batches in flight per step (default 2*N)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse "
                                                        "the multi-process path on a box with fewer GPUs than ranks)")
    ap.add_argument("--op", default="compress", choices=["compress", "apply", "hadamard", "add", "scale", "dot", "orthogonalize"],
                    help="compress (default) = the headline apply+round step; the others print the roofline line of that kernel alone")
    ap.add_argument("--launch-check", action="store_true",
                    help="only start the ranks, form the process group, all-reduce the world size and print it (no GPU needed: the "
                         "launcher test of tests/test_distributed_gloo.py)")
    args = ap.parse_args()

    # ---- launcher: N > 1 without a torchrun environment -> start the ranks as a child BEFORE anything touches a GPU ----
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_workers(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a line whose n_gpus differs from --gpus")

    import torch
    dist = None
    if args.launch_check:
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo" if args.backend != "nccl" or not torch.cuda.is_available() else "nccl",
                                    rank=rank, world_size=world)
            t = torch.ones(1, dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t)
            n = int(t.item())
            dist.destroy_process_group()
        else:
            n = 1
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": n, "requested": args.gpus}), flush=True)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    red_device = "cuda" if args.backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    import ttn_amd as T
    from ttn_amd import device as D
    T.ensure_init(dev_index)

    d, r, B = args.d, args.rank, args.batch
    if args.op != "compress":
        if world != 1:
            raise SystemExit("--op lines are single-GPU measurements")
        return bench_op(args, T, D)
    if args.shard == "cores":
        return bench_core_sharded(args, T, D, torch, dist, rank, world, red_device)
    # ---- the headline step: C3 unless --d / --rank say otherwise -------------------------------------------------------
    def barrier():
        D.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    head = run_compress_config(T, D, d, r, B, args.steps, args.warmup, rank, world, barrier, dist, red_device,
                               verify=(rank == 0 and not args.no_verify))
    extras = {}
    if rank == 0 and not args.no_single:
        extras["batch_sweep"] = batch_sweep(T, D, d, r)
        extras["single_train"] = extras["batch_sweep"][0]
        extras["drop_in"] = drop_in_latency(T, d, r)
    # ---- the other hot-path kernels as driver-observed sub-records: the same measurement as `--op X`, on a batch of
    #      256 (streaming kernels: 3 GB of output per launch at most) or 1024 (dot / orthogonalize) C3-shaped trains; < 3 s in all ----
    if rank == 0 and world == 1 and not args.no_ops and (d, r) == (30, 64):
        import copy
        ops = {}
        for op, ob in (("apply", 256), ("add", 256), ("scale", 256), ("hadamard", 256), ("dot", 1024), ("orthogonalize", 1024)):
            a2 = copy.copy(args)
            a2.op, a2.batch, a2.steps, a2.warmup = op, ob, 5, 2
            try:
                rec = bench_op(a2, T, D, emit=False)
                ops[op] = {"batch": ob, "ms_per_launch": rec["ms_per_step"], "bound": rec["roofline"]["bound"], "achieved": rec["roofline"]["achieved"],
                           "unit": rec["roofline"]["unit"], "frac": rec["roofline"]["frac"], "kernel": rec["roofline"]["kernel"]}
            except Exception as e:                                   # a sub-record must never cost the headline line
                ops[op] = {"error": repr(e)[:200]}
        extras["other_kernels"] = ops
    # ---- BASELINE config C2 (d = 20, rank 32) as a driver-observed sub-record, same op, same batch size (12 ms per step) ----
    c2 = None
    if not args.no_c2 and (d, r) == (30, 64):
        c2 = run_compress_config(T, D, 20, 32, B, max(3, args.steps), 1, rank, world, barrier, dist, red_device,
                                 verify=(rank == 0 and not args.no_verify))
    # ---- BASELINE config C4 (cores sharded over the GPUs) as a sub-record of the N > 1 line, guarded by a watchdog ----
    core_sharded = None
    if world > 1 and not args.no_core_sharded:
        core_sharded = guarded_core_sharded(args, T, D, torch, dist, rank, world, red_device, head if rank == 0 else None, extras, c2)

    if rank == 0:
        res = headline_record(args, head, d, r, B, world)
        res.update(extras)
        if c2 is not None:
            res["c2"] = {"workload": "C2: tt_compress!(Delta(20)*x, 32), batch of %d trains per GPU" % B, "value": c2["value"], "unit": "TT cores/s",
                         "ms_per_step": c2["ms_per_step"], "frac": c2["frac"], "verified": c2.get("verified"),
                         "verified_max_rel_diff": c2.get("verified_max_rel_diff")}
        if core_sharded is not None:
            res["core_sharded"] = core_sharded
        if not args.no_cpu and world == 1:
            res["cpu_baseline"] = cpu_baseline(d, r)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
