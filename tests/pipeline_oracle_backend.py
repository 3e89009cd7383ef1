"""CPU-oracle compute backend for tensortrainnumerics.jl_amd/pipeline.py — TEST INFRASTRUCTURE.

Lets the core-wise sharding orchestration (segments, boundary-core hand-offs, schedule) run under gloo on
the CPU: segments are lists of per-train numpy core lists and the bond steps are the oracle's
tt_bond_truncate_.  The product path has no such backend (no CPU fallback)."""
import numpy as np


class OracleBackend:
    """Segments are lists of per-train numpy core lists; bond steps by the CPU oracle.  TEST INFRASTRUCTURE: it lets
    the orchestration (who sends what when) run under gloo without a GPU; never used by the product path."""

    name = "oracle"

    def __init__(self, oracle_module):
        import torch
        self.O, self.torch = oracle_module, torch

    def prepare(self, A_cores, A_rks, x_trains, dims):
        return A_cores, x_trains

    def apply_prepared(self, prep):
        A_cores, x_trains = prep
        out = []
        for cores in x_trains:
            y = []
            for k, (a, x) in enumerate(zip(A_cores, cores)):
                n, _, Rl, Rr = a.shape
                _, rl, rr = x.shape
                t = np.einsum("ijab,jcd->iacbd", a, x)              # (i, a', nu', a, nu): operator index fastest
                y.append(np.asfortranarray(t.reshape(n, Rl * rl, Rr * rr, order="F")))
            out.append(y)
        # rank capacities of every core's slot = the product ranks (what the device handle is created with)
        self._cap = [(int(c.shape[1]), int(c.shape[2])) for c in out[0]]
        return out

    def lr_sweep(self, seg, n_ext, max_bond, truncerr, first_real):
        if n_ext >= 2:                              # (the product was formed by apply_prepared; an imported core 0 replaced its slot)
            self.sweep(seg, 0, n_ext - 2, max_bond, truncerr)

    def release(self):
        pass

    def batch(self, seg):
        return len(seg)

    def phys_dim(self, seg, k):
        return int(seg[0][k].shape[0])

    def core_capacity(self, seg, k):
        return self._cap[k]

    def _train(self, cores):
        dims = tuple(int(c.shape[0]) for c in cores)
        rks = [int(cores[0].shape[1])] + [int(c.shape[2]) for c in cores]
        return self.O.TTvector(len(cores), [np.asfortranarray(c) for c in cores], dims, rks, [0] * len(cores))

    def sweep(self, seg, k_first: int, k_last: int, max_bond: int, truncerr: float):
        step = 1 if k_first <= k_last else -1
        for b, cores in enumerate(seg):
            t = self._train(cores)
            for k in range(k_first, k_last + step, step):
                self.O.tt_bond_truncate_(t, k + 1, max_bond=max_bond, truncerr=truncerr)
            seg[b] = list(t.ttv_vec)

    def export_core(self, seg, k: int):
        bl = max(int(c[k].shape[1]) for c in seg)
        br = max(int(c[k].shape[2]) for c in seg)
        n = int(seg[0][k].shape[0])
        data = np.zeros((len(seg), max(int(c[k].size) for c in seg)))
        rks = np.zeros((len(seg), 2), dtype=np.int64)
        for b, cores in enumerate(seg):
            c = cores[k]
            data[b, : c.size] = c.reshape(-1, order="F")
            rks[b] = (c.shape[1], c.shape[2])
        return self.torch.from_numpy(data), self.torch.from_numpy(rks), bl, br

    def import_core(self, seg, k: int, data, rks, bl: int, br: int):
        data, rks = data.cpu().numpy(), rks.cpu().numpy()
        for b, cores in enumerate(seg):
            n = int(cores[k].shape[0])
            rl, rr = int(rks[b, 0]), int(rks[b, 1])
            # a COPY: the transport's receive buffers are persistent and will be overwritten by a later hand-off
            cores[k] = np.array(data[b, : n * rl * rr].reshape((n, rl, rr), order="F"), order="F", copy=True)

    def download(self, seg, b: int):
        return [np.array(c) for c in seg[b]]

    def ncores(self, seg) -> int:
        return len(seg[0])
