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
        import ctypes as C
        import torch
        from . import _lib, device
        self.torch, self._lib, self.D = torch, _lib, device
        _lib.ensure_init()
        h = C.c_void_p()
        _lib.check(_lib.lib().ttn_stream_handle(C.byref(h)))
        # the library's HIP stream as a torch stream: the hand-offs are ordered against it with EVENTS (wait_stream), the host
        # never blocks between a sweep and the send that follows it
        self.lib_stream = torch.cuda.ExternalStream(h.value)
        self._keep = []                            # tensors the library stream still reads (released at flush)
        self._export_bufs = {}                     # (batch, doubles per train) -> persistent (core data, ranks) device buffers

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
        """y_ext = A_ext * x_ext, VIRTUALLY: y only receives the product's ranks (ttn_apply_begin); the cores are built inside
        the bond steps of lr_sweep (fused apply, as in ttn_apply_compress).  Returns the DeviceTT the sweeps work on."""
        dA, dx, dy = prep
        self._lib.check(self._lib.lib().ttn_apply_begin(dA.h, dx.h, dy.h))
        dy._fuse = (dA, dx)
        return dy

    def lr_sweep(self, seg, n_ext: int, max_bond: int, truncerr: float, first_real: bool):
        """The L->R pass over all bonds of the extended segment with the fused apply; first_real: core 0 was imported."""
        dA, dx = seg._fuse
        if n_ext >= 2:
            self._lib.check(self._lib.lib().ttn_apply_sweep(dA.h, dx.h, seg.h, 1, n_ext - 1, int(max_bond), float(truncerr), 1 if first_real else 0))
        elif not first_real:
            self.D.apply(dA, dx, seg)              # a one-core segment with nothing to its left: the core itself

    def sweep(self, seg, k_first: int, k_last: int, max_bond: int, truncerr: float):
        self._lib.check(self._lib.lib().ttn_sweep(seg.h, int(k_first) + 1, int(k_last) + 1, int(max_bond), float(truncerr)))

    def export_core(self, seg, k: int):
        import ctypes as C
        n, bl, br = C.c_int64(), C.c_int64(), C.c_int64()
        self._lib.check(self._lib.lib().ttn_tt_core_extent(seg.h, k + 1, C.byref(n), C.byref(bl), C.byref(br)))
        # persistent export buffers, one pair per (batch, row length): no allocation per hand-off
        key = (seg.batch, int(n.value))
        bufs = self._export_bufs.get(key)
        if bufs is None:
            dev = self.torch.device("cuda", self.torch.cuda.current_device())
            bufs = (self.torch.empty((seg.batch, n.value), dtype=self.torch.float64, device=dev),
                    self.torch.empty((seg.batch, 2), dtype=self.torch.int64, device=dev))
            self._export_bufs[key] = bufs
        data, rks = bufs
        # the library's stream writes these buffers: whatever torch's stream still does with them (the copy of the PREVIOUS
        # export into its message) must be done first — and whatever the transport does next waits for the export.  Both are
        # device-side event waits; the host goes on.
        self.lib_stream.wait_stream(self.torch.cuda.current_stream())
        self._lib.check(self._lib.lib().ttn_tt_core_export(seg.h, k + 1, data.data_ptr(), rks.data_ptr()))
        self.torch.cuda.current_stream().wait_stream(self.lib_stream)
        return data, rks, int(bl.value), int(br.value)

    def import_core(self, seg, k: int, data, rks, bl: int, br: int):
        dev = self.torch.device("cuda", self.torch.cuda.current_device())
        data = data.to(dev).contiguous()
        rks = rks.to(dev).contiguous()
        self.lib_stream.wait_stream(self.torch.cuda.current_stream())        # the received / copied tensors are ready before the import reads them
        self._lib.check(self._lib.lib().ttn_tt_core_import(seg.h, k + 1, data.data_ptr(), rks.data_ptr(), int(bl), int(br)))
        self._keep.append((data, rks))             # alive until the library stream has consumed them (release())

    def fence(self):
        """torch's current stream waits (on the device) for everything enqueued on the library's stream so far: called before a
        persistent receive buffer that an earlier import read is handed to the transport again."""
        self.torch.cuda.current_stream().wait_stream(self.lib_stream)

    def release(self):
        """Host sync point at the end of a sharded op: the tensors handed to the library may be freed afterwards."""
        self.D.sync()
        self._keep = []

    def download(self, seg, b: int):
        """Cores of train b as numpy arrays (verification only)."""
        return [np.array(c) for c in seg.download(b).ttv_vec]

    def ncores(self, seg) -> int:
        return seg.N

    def batch(self, seg) -> int:
        return seg.batch

    def phys_dim(self, seg, k: int) -> int:
        return int(seg.dims[k])

    def core_capacity(self, seg, k: int) -> Tuple[int, int]:
        """Rank capacities of core k's slot = the product ranks A.rks .* x.rks of that core."""
        return int(seg.cap[k]), int(seg.cap[k + 1])


# --------------------------------------------------------------------------------------------------------------------
# transport
# --------------------------------------------------------------------------------------------------------------------
class DistTransport:
    """Point-to-point hand-off of one boundary core over torch.distributed: ONE message per hand-off.

    Sender and receiver agree on the message size without a header: the extents (BL, BR) of the core are upper bounds both
    sides derive from what they hold — the rank capacities of the core's slot and max_bond (handoff_extents).  Message =
    [batch x n*BL*BR doubles: every train's core, compact with its current ranks at the start of its row | batch x 2: the
    two ranks, as float64 (exact below 2^53)].

    Buffers are PERSISTENT: per (peer, message length) a ring of RING message buffers on each side, reused round-robin — a send
    buffer is reused once the send that last used it has completed, a receive buffer once its contents have been imported.
    Receives can be PRE-POSTED (post_recv) so that the hand-off of micro-batch m+1 is already in flight while micro-batch m is
    being swept; recv() posts the receive itself if nobody did.  Per pair of neighbours the order of operations is the same on
    both sides (all L->R hand-offs in micro-batch order, then all R->L hand-offs), which RCCL's in-order point-to-point
    matching requires: receives are therefore never posted across the direction change."""

    RING = 2

    def __init__(self, dist, device="cpu"):
        import torch
        self.dist, self.torch, self.device = dist, torch, torch.device(device)
        self._send_ring = {}                       # (dst, numel) -> {"bufs": [...], "reqs": [...], "next": i}
        self._recv_ring = {}                       # (src, numel) -> {"bufs": [...], "next": i}
        self._stage = {}                           # (numel, device) -> staging buffer on the data's device (host transports)
        self._posted = {}                          # (src, tag) -> (request, buffer, B, W)

    def _ring(self, table, key, numel, with_reqs):
        ring = table.get(key)
        if ring is None:
            ring = {"bufs": [self.torch.zeros(numel, dtype=self.torch.float64, device=self.device) for _ in range(self.RING)], "next": 0}
            if with_reqs:
                ring["reqs"] = [None] * self.RING
            table[key] = ring
        return ring

    def send(self, payload, dst: int, n: int, BL: int, BR: int):
        """Non-blocking: the sender goes on with its next micro-batch while the neighbour is still busy."""
        data, rks, bl, br = payload
        assert bl <= BL and br <= BR, (bl, br, BL, BR)
        B, W = data.shape[0], n * BL * BR
        numel = B * (W + 2)
        ring = self._ring(self._send_ring, (dst, numel), numel, True)
        i = ring["next"]
        ring["next"] = (i + 1) % self.RING
        if ring["reqs"][i] is not None:
            ring["reqs"][i].wait()                 # the send that last used this buffer (RING hand-offs ago) has completed
        out = ring["bufs"][i]
        if data.device == out.device:
            msg = out
        else:                                      # host transport (gloo): pack on the data's device, one copy to the host buffer
            skey = (numel, str(data.device))
            msg = self._stage.get(skey)
            if msg is None:
                msg = self._stage[skey] = self.torch.zeros(numel, dtype=self.torch.float64, device=data.device)
        msg[: B * W].view(B, W)[:, : data.shape[1]] = data
        msg[B * W:].view(B, 2).copy_(rks.to(self.torch.float64))
        if msg is not out:
            out.copy_(msg)
        ring["reqs"][i] = self.dist.isend(out, dst)

    def flush(self):
        for ring in self._send_ring.values():
            for i, req in enumerate(ring["reqs"]):
                if req is not None:
                    req.wait()
                    ring["reqs"][i] = None
        assert not self._posted, "a pre-posted receive was never consumed"

    def post_recv(self, src: int, B: int, n: int, BL: int, BR: int, tag):
        """Start receiving the hand-off `tag` (any hashable: direction and micro-batch index) from `src` into the next buffer of
        the ring.  The caller guarantees that the buffer's previous contents have been consumed (DeviceBackend.fence)."""
        W = n * BL * BR
        numel = B * (W + 2)
        ring = self._ring(self._recv_ring, (src, numel), numel, False)
        i = ring["next"]
        ring["next"] = (i + 1) % self.RING
        buf = ring["bufs"][i]
        self._posted[(src, tag)] = (self.dist.irecv(buf, src), buf, B, W)

    def recv(self, src: int, B: int, n: int, BL: int, BR: int, tag=None):
        if (src, tag) not in self._posted:
            self.post_recv(src, B, n, BL, BR, tag)
        req, msg, B_, W = self._posted.pop((src, tag))
        assert (B_, W) == (B, n * BL * BR), "hand-off size mismatch between the pre-posted receive and its use"
        req.wait()
        return msg[: B * W].view(B, W), msg[B * W:].view(B, 2).to(self.torch.int64), BL, BR


def handoff_extents(cap_left: int, cap_right: int, max_bond: int, direction: int) -> Tuple[int, int]:
    """Upper bounds (BL, BR) of the ranks of a boundary core when it is handed over, from what BOTH neighbours know: the rank
    capacities of its slot (the product ranks A.rks .* x.rks of the mirrored core) and max_bond.  L->R (direction 0): the left
    bond has been truncated, the right one not yet; R->L (1): both have."""
    bl = min(int(cap_left), int(max_bond))
    br = min(int(cap_right), int(max_bond)) if direction else int(cap_right)
    return bl, br


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
    fence = getattr(backend, "fence", lambda: None)

    def shape(seg, k, direction):
        """(batch, physical dimension, BL, BR) of the hand-off of core k of THIS micro-batch (micro-batches may differ in size)."""
        return (backend.batch(seg), backend.phys_dim(seg, k)) + handoff_extents(*backend.core_capacity(seg, k), max_bond, direction)

    M = len(segs)
    # ---- L -> R ----
    if not first and M:
        transport.post_recv(rank - 1, *shape(segs[0], 0, 0), tag=("lr", 0))
    for m, seg in enumerate(segs):
        if not first:
            got = transport.recv(rank - 1, *shape(seg, 0, 0), tag=("lr", m))
            if m + 1 < M:                          # the next micro-batch's core travels while this one is swept
                fence()
                transport.post_recv(rank - 1, *shape(segs[m + 1], 0, 0), tag=("lr", m + 1))
            backend.import_core(seg, 0, *got)
        backend.lr_sweep(seg, n_ext, max_bond, truncerr, not first)
        if not last:
            transport.send(backend.export_core(seg, n_ext - 1), rank + 1, *shape(seg, n_ext - 1, 0)[1:])
    # ---- R -> L ----
    if not last and M:
        fence()
        transport.post_recv(rank + 1, *shape(segs[0], n_ext - 1, 1), tag=("rl", 0))
    for m, seg in enumerate(segs):
        if not last:
            got = transport.recv(rank + 1, *shape(seg, n_ext - 1, 1), tag=("rl", m))
            if m + 1 < M:
                fence()
                transport.post_recv(rank + 1, *shape(segs[m + 1], n_ext - 1, 1), tag=("rl", m + 1))
            backend.import_core(seg, n_ext - 1, *got)
        if n_ext >= 2:
            backend.sweep(seg, n_ext - 2, 0, max_bond, truncerr)
        if not first:
            transport.send(backend.export_core(seg, 0), rank - 1, *shape(seg, 0, 1)[1:])
    if transport is not None:
        transport.flush()
    backend.release()
    return segs
