"""Core-wise sharded tt_compress!(A*x, r) (tensortrainnumerics.jl_amd/pipeline.py): the chain is cut into
`world` segments, one process per segment, boundary cores handed between neighbours over torch.distributed.

* CPU (not gpu): the orchestration runs under gloo with the CPU oracle as compute backend
  (tests/pipeline_oracle_backend.py); the union of the segments must equal the oracle's unsharded result.
* GPU: the same with the HIP backend, two ranks sharing cuda:0 (gloo, boundary cores staged through the host), against
  the single-process device result.  The nccl/xGMI transport differs only in where the tensors live."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sizes(nmb, per_mb):
    """per_mb: one size for every micro-batch, or a list of sizes (ragged micro-batches)."""
    return list(per_mb) if isinstance(per_mb, (list, tuple)) else [per_mb] * nmb


def _inputs(d, r, nmb, per_mb):
    """Operator cores/ranks and nmb micro-batches of per_mb trains (seeds 30 + global index), as numpy."""
    import ttn_amd as T
    A = T.Delta(d)
    mbs = []
    g = 0
    for sz in _sizes(nmb, per_mb):
        trains = []
        for _ in range(sz):
            trains.append([np.asfortranarray(c) for c in T.rand_tt((2,) * d, r, seed=30 + g).ttv_vec])
            g += 1
        mbs.append(trains)
    return [np.asfortranarray(c) for c in A.tto_vec], list(A.tto_rks), mbs


def _worker(rank, world, port, use_gpu, d, r, nmb, per_mb, max_bond, q):
    try:
        _worker_body(rank, world, port, use_gpu, d, r, nmb, per_mb, max_bond, q)
    except BaseException as exc:                          # report instead of leaving the parent to time out on the queue
        import traceback
        q.put((rank, "ERROR: " + "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))))
        raise


def _worker_body(rank, world, port, use_gpu, d, r, nmb, per_mb, max_bond, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ttn_amd as T
    from ttn_amd import pipeline as PL
    if use_gpu:
        backend = PL.DeviceBackend()
    else:
        from oracle import tt_oracle as O
        from tests.pipeline_oracle_backend import OracleBackend
        backend = OracleBackend(O)
    A_cores, A_rks, mbs = _inputs(d, r, nmb, per_mb)
    lo, hi = PL.extended_range(d, rank, world)
    prepared = [backend.prepare(A_cores[lo:hi], A_rks[lo:hi + 1], [t[lo:hi] for t in trains], (2,) * (hi - lo)) for trains in mbs]
    segs = PL.sharded_apply_compress(backend, PL.DistTransport(dist, "cpu"), rank, world, prepared, hi - lo, max_bond)
    own0 = 0 if rank == 0 else 1                           # slot 0 of rank > 0 mirrors the neighbour's last core
    out = [[backend.download(seg, b)[own0:] for b in range(sz)] for seg, sz in zip(segs, _sizes(nmb, per_mb))]
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, use_gpu, d, r, nmb, per_mb, max_bond):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(rk, world, port, use_gpu, d, r, nmb, per_mb, max_bond, q)) for rk in range(world)]
    for p in procs:
        p.start()
    res = []
    for _ in range(world):
        item = q.get(timeout=600)
        if isinstance(item[1], str):
            for p in procs:
                p.terminate()
            raise AssertionError(f"rank {item[0]}: {item[1]}")
        res.append(item)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    res.sort(key=lambda t: t[0])
    # glue the segments: result[mb][train] = list of d cores
    glued = [[sum((res[rk][1][m][b] for rk in range(world)), []) for b in range(sz)] for m, sz in enumerate(_sizes(nmb, per_mb))]
    return glued


# world 2 / 3 on a short chain (incl. RAGGED micro-batches: the last one smaller, so every hand-off's message size is derived per
# micro-batch), and BASELINE config C4's chain length d = 30 cut into 4 and 8 segments (segment_bounds(30, 8): 4 + 4 + ... cores)
# with three micro-batches in flight — more micro-batches than ring buffers, so the persistent receive / send rings wrap around.
@pytest.mark.parametrize("world,d,r,nmb,per_mb", [(2, 9, 4, 2, 2), (3, 9, 4, 2, 2), (3, 9, 4, 3, [2, 2, 1]), (4, 30, 4, 3, 2), (8, 30, 3, 3, [2, 1, 2])])
def test_core_sharded_compress_equals_unsharded_oracle(world, d, r, nmb, per_mb):
    from oracle import tt_oracle as O
    glued = _run(world, False, d, r, nmb, per_mb, r)
    A_cores, A_rks, mbs = _inputs(d, r, nmb, per_mb)
    A = O.TToperator(d, A_cores, (2,) * d, A_rks, [0] * d)
    for m in range(nmb):
        for b in range(len(mbs[m])):
            x = O.TTvector(d, mbs[m][b], (2,) * d, [1] + [int(c.shape[2]) for c in mbs[m][b]], [0] * d)
            y = O.tt_compress_(O.apply(A, x), r)
            got = glued[m][b]
            assert len(got) == d
            assert [int(c.shape[2]) for c in got] == list(y.ttv_rks[1:])
            for k in range(d):                              # same bond steps in the same order: identical cores
                np.testing.assert_allclose(got[k], y.ttv_vec[k], rtol=0, atol=1e-13)


def test_segment_bounds():
    from ttn_amd import pipeline as PL
    for d, w in ((30, 8), (30, 4), (9, 3), (5, 5), (7, 1)):
        b = PL.segment_bounds(d, w)
        assert b[0][0] == 0 and b[-1][1] == d and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert all(hi > lo for lo, hi in b)
        for p in range(w):
            lo, hi = PL.extended_range(d, p, w)
            assert lo == b[p][0] - (1 if p else 0) and hi == b[p][1]


@pytest.mark.gpu
@pytest.mark.parametrize("world,d,r,nmb,per_mb", [(2, 16, 16, 2, 3), (3, 13, 8, 3, [2, 2, 1]), (2, 30, 64, 2, 2)])
def test_core_sharded_compress_on_gpu_equals_single_process(world, d, r, nmb, per_mb):
    """Ranks sharing one GPU (gloo transport, boundary cores staged through the host): the sharded device result — fused apply inside
    every segment's L->R pass, one packed message per hand-off, event-ordered streams — equals the unsharded device result.  The last
    case is BASELINE config C4's chain (d = 30, rank 64) cut in two."""
    import ttn_amd as T
    from ttn_amd import device as D
    glued = _run(world, True, d, r, nmb, per_mb, r)
    A = T.Delta(d)
    dA = T.DeviceTTO(A)
    g = 0
    for m in range(nmb):
        for b in range(_sizes(nmb, per_mb)[m]):
            x = T.rand_tt((2,) * d, r, seed=30 + g)
            g += 1
            dx = T.DeviceTT.from_host(x)
            dy = T.DeviceTT((2,) * d, [a * c for a, c in zip(A.tto_rks, x.ttv_rks)])
            D.apply_compress(dA, dx, dy, r)
            y = dy.download(0)
            got = glued[m][b]
            assert [int(c.shape[2]) for c in got] == list(y.ttv_rks[1:])
            for k in range(d):
                np.testing.assert_allclose(got[k], y.ttv_vec[k], rtol=0, atol=1e-12)
