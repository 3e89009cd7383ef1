"""Core-wise sharding of long chains over the GPUs of one node (BASELINE config 4).

The chain of d cores is cut into `world` contiguous segments, one per rank (= one per GPU).  `*`
(apply) is per-core independent, so every rank applies its own segment with no communication.
`tt_compress!` (src/tt_tools.jl:772-789) is a strict left<->right recurrence over the bonds; the
only state that crosses a segment boundary is ONE core per train and direction:

  L->R   rank p finishes its last interior bond and sends core c_{p+1}-1 (already truncated on its
         left bond: n * r_new * r_old doubles, 196 608 B per C3 train) to rank p+1, which owns the
         straddling bond (c_{p+1}-1, c_{p+1});
  R->L   rank p+1 finishes with the straddling bond and sends that core (now n * r * r) back.

Rank p > 0 therefore works on an EXTENDED segment [c_p - 1, c_{p+1}): slot 0 mirrors the neighbour's
last core.  The bond steps themselves are the same kernels in the same order as on one GPU
(`ttn_sweep`), so the sharded result is the unsharded result.  One train keeps only one GPU busy at
a time; throughput comes from pipelining micro-batches of trains through the ranks (fill/drain
(world-1) stages per direction).  Independent trains shard with NO communication (shard.py) and that
is what bench.py measures; this module is for chains that should not be replicated.

Transport: torch.distributed point-to-point (RCCL over xGMI with backend "nccl": the exported core
buffer is a CUDA tensor, no host staging; "gloo" in the CPU/one-GPU tests: staged through host
memory).  The compute backend is an argument so that the orchestration (who sends what when) is
testable without a GPU: tests/pipeline_oracle_backend.py plugs the CPU oracle in for the "not gpu"
gloo test; this module itself only knows the HIP backend.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from .shard import partition


def segment_bounds(d: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous balanced cut of cores 0..d-1 into `world` segments [lo, hi) (0-based), each non-empty."""
    assert 1 <= world <= d
    return [partition(d, p, world) for p in range(world)]


def extended_range(d: int, rank: int, world: int) -> Tuple[int, int]:
    """Cores [lo_ext, hi) a rank holds during a sweep: its segment plus the left neighbour's last core."""
    lo, hi = segment_bounds(d, world)[rank]
    return (lo - 1 if rank > 0 else lo), hi


# --------------------------------------------------------------------------------------------------------------------
# compute backends
# --------------------------------------------------------------------------------------------------------------------
class DeviceBackend:
    """Segments are DeviceTT handles (libttn_hip).  Exported cores are torch CUDA tensors."""

    name = "hip"

    def __init__(self):
        import torch
        from . import _lib, device
        self.torch, self._lib, self.D = torch, _lib, device

    def prepare(self, A_cores, A_rks, x_trains, dims):
        """Upload the extended segment of the operator and of a micro-batch of trains (list of per-train core lists);
        returns the resident (dA, dx, dy) triple that apply_prepared() works on."""
        from .tt import TToperator, TTvector
        N = len(dims)
        A = TToperator(N, A_cores, tuple(dims), list(A_rks), [0] * N)
        dA = self.D.DeviceTTO(A)
        B = len(x_trains)
        xr = [int(x_trains[0][0].shape[1])] + [int(c.shape[2]) for c in x_trains[0]]
        dx = self.D.DeviceTT(dims, xr, batch=B)
        for b, cores in enumerate(x_trains):
            rks = [int(cores[0].shape[1])] + [int(c.shape[2]) for c in cores]
            dx.upload(b, TTvector(N, cores, tuple(dims), rks, [0] * N))
        dy = self.D.DeviceTT(dims, [a * r for a, r in zip(A_rks, xr)], batch=B)
        return dA, dx, dy

    def apply_prepared(self, prep):
        """y_ext = A_ext * x_ext on the device; returns the DeviceTT the sweeps then work on."""
        dA, dx, dy = prep
        self.D.apply(dA, dx, dy)
        return dy

    def sweep(self, seg, k_first: int, k_last: int, max_bond: int, truncerr: float):
        self._lib.check(self._lib.lib().ttn_sweep(seg.h, int(k_first) + 1, int(k_last) + 1, int(max_bond), float(truncerr)))

    def export_core(self, seg, k: int):
        import ctypes as C
        n, bl, br = C.c_int64(), C.c_int64(), C.c_int64()
        self._lib.check(self._lib.lib().ttn_tt_core_extent(seg.h, k + 1, C.byref(n), C.byref(bl), C.byref(br)))
        dev = self.torch.device("cuda", self.torch.cuda.current_device())
        data = self.torch.empty((seg.batch, n.value), dtype=self.torch.float64, device=dev)
        rks = self.torch.empty((seg.batch, 2), dtype=self.torch.int64, device=dev)
        self._lib.check(self._lib.lib().ttn_tt_core_export(seg.h, k + 1, data.data_ptr(), rks.data_ptr()))
        self.D.sync()                              # the copy ran on the library's stream; the transport uses torch's
        return data, rks, int(bl.value), int(br.value)

    def import_core(self, seg, k: int, data, rks, bl: int, br: int):
        dev = self.torch.device("cuda", self.torch.cuda.current_device())
        data = data.to(dev).contiguous()
        rks = rks.to(dev).contiguous()
        self.torch.cuda.synchronize()
        self._lib.check(self._lib.lib().ttn_tt_core_import(seg.h, k + 1, data.data_ptr(), rks.data_ptr(), int(bl), int(br)))
        self.D.sync()                              # `data` may be freed by the caller right after

    def download(self, seg, b: int):
        """Cores of train b as numpy arrays (verification only)."""
        return [np.array(c) for c in seg.download(b).ttv_vec]

    def ncores(self, seg) -> int:
        return seg.N


# --------------------------------------------------------------------------------------------------------------------
# transport
# --------------------------------------------------------------------------------------------------------------------
class DistTransport:
    """Point-to-point hand-off of one boundary core (header, ranks, data) over torch.distributed."""

    def __init__(self, dist, device="cpu"):
        import torch
        self.dist, self.torch, self.device = dist, torch, torch.device(device)
        self._pending = []                         # (request, tensor) of sends in flight: the tensors must stay alive

    def send(self, payload, dst: int):
        """Non-blocking: the sender goes on with its next micro-batch while the neighbour is still busy."""
        data, rks, bl, br = payload
        hdr = self.torch.tensor([data.shape[0], data.shape[1], bl, br], dtype=self.torch.int64, device=self.device)
        for t in (hdr, rks.to(self.device).contiguous(), data.to(self.device).contiguous()):
            self._pending.append((self.dist.isend(t, dst), t))

    def flush(self):
        for req, _ in self._pending:
            req.wait()
        self._pending = []

    def recv(self, src: int):
        hdr = self.torch.empty(4, dtype=self.torch.int64, device=self.device)
        self.dist.recv(hdr, src)
        B, n, bl, br = (int(v) for v in hdr.tolist())
        rks = self.torch.empty((B, 2), dtype=self.torch.int64, device=self.device)
        data = self.torch.empty((B, n), dtype=self.torch.float64, device=self.device)
        self.dist.recv(rks, src)
        self.dist.recv(data, src)
        return data, rks, bl, br


# --------------------------------------------------------------------------------------------------------------------
# the sharded op
# --------------------------------------------------------------------------------------------------------------------
def sharded_apply_compress(backend, transport, rank: int, world: int, prepared: Sequence, n_ext: int, max_bond: int,
                           truncerr: float = 0.0):
    """tt_compress!(A * x, max_bond) (sweeps = 1) on this rank's segment of every micro-batch of trains.

    `prepared` = one backend.prepare(A_cores, A_rks, x_trains, dims) per micro-batch, all for the EXTENDED range of this
    rank (extended_range; n_ext cores).  Returns the list of extended segments (backend objects); cores [1:] (all of them
    on rank 0) are this rank's part of the result.

    Schedule: every rank runs the L->R stage of micro-batch 0, 1, 2, ... in order, then the R->L stage in the same
    order; a stage starts when the neighbour's hand-off for that micro-batch arrives, so the ranks work on different
    micro-batches at the same time (pipeline, (world-1) stages of fill and drain per direction)."""
    segs = [backend.apply_prepared(p) for p in prepared]
    first, last = rank == 0, rank == world - 1
    # ---- L -> R ----
    for seg in segs:
        if not first:
            backend.import_core(seg, 0, *transport.recv(rank - 1))
        if n_ext >= 2:
            backend.sweep(seg, 0, n_ext - 2, max_bond, truncerr)
        if not last:
            transport.send(backend.export_core(seg, n_ext - 1), rank + 1)
    # ---- R -> L ----
    for seg in segs:
        if not last:
            backend.import_core(seg, n_ext - 1, *transport.recv(rank + 1))
        if n_ext >= 2:
            backend.sweep(seg, n_ext - 2, 0, max_bond, truncerr)
        if not first:
            transport.send(backend.export_core(seg, 0), rank - 1)
    if transport is not None:
        transport.flush()
    return segs
