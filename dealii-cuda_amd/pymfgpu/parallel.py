"""Single-node multi-GPU mode: cells sharded by z-slab, one process per GPU, shared-dof
contributions summed over RCCL (torch.distributed backend "nccl") after each vmult.

The reference is single-GPU (no MPI/NCCL anywhere); this is new work mapped to
xGMI's point-to-point topology: a slab shares dofs only with its two z-neighbours,
so the exchange is one 2-rank all-reduce per interface plane ((p*n+1)^2 dofs) over the direct
xGMI link, not an 8-rank collective.  Modes:
  "p2p"       (default) both neighbours in ONE grouped RCCL send/recv (batch_isend_irecv); the interface
              planes of a z-slab are contiguous slices of the lexicographic dof vector, so nothing is
              packed: send the slice, add the received plane (masked on constrained rows)
  "pair"      one 2-rank all-reduce per interface plane, even interfaces then odd ones
  "allreduce" the literal dense all-reduce over all interface planes (simple validated fallback)

Everything here works on CPU tensors with the gloo backend as well (tests/test_distributed.py).
"""
from __future__ import annotations

import numpy as np


def slab_ranges(n_cells_z: int, world: int):
    """balanced contiguous split of the last mesh direction"""
    base, rem = divmod(n_cells_z, world)
    out, z = [], 0
    for r in range(world):
        c = base + (1 if r < rem else 0)
        out.append((z, z + c))
        z += c
    return out


class SlabExchange:
    """Sum of shared-dof contributions between neighbouring slabs.

    After the local cell loop each rank holds, on an interface plane, only its own cells'
    contributions; the exchange leaves the full sum on both sharers, so the next iterate's ghost
    values are consistent without a second exchange.  Constrained (Dirichlet) interface dofs are
    identity rows on both sides and are not summed."""

    def __init__(self, mesh, rank: int, world: int, device, dtype, mode: str = "p2p"):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.rank, self.world, self.mode = rank, world, mode
        con = mesh.arrays()["constrained_dofs"]
        self.idx, self.free, self.peer = {}, {}, {}
        for which, peer in ((0, rank - 1), (1, rank + 1)):
            ids = mesh.interface_dofs(which)
            if len(ids) == 0:
                continue
            assert 0 <= peer < world
            self.idx[which] = torch.as_tensor(ids.astype(np.int64), device=device)
            self.free[which] = torch.as_tensor((~np.isin(ids, con)).astype(np.float64), device=device).to(dtype)
            self.peer[which] = peer
        self.plane = next(iter(self.idx.values())).numel() if self.idx else 0
        # contiguous fast path: lower plane = first `plane` entries, upper plane = last ones
        n_dofs = mesh.n_dofs
        self.slices = {}
        for which, ids in self.idx.items():
            lo = int(ids[0].item())
            if bool((ids == torch.arange(lo, lo + self.plane, device=ids.device)).all()):
                self.slices[which] = slice(lo, lo + self.plane)
        self.recv = {w: torch.empty(self.plane, device=device, dtype=dtype) for w in self.idx}
        self.pair_groups = {}
        if world > 1 and mode == "pair":
            # every rank must create every group, in the same order
            for i in range(world - 1):
                g = dist.new_group(ranks=[i, i + 1])
                if rank == i:
                    self.pair_groups[1] = g
                elif rank == i + 1:
                    self.pair_groups[0] = g
        if world > 1 and mode == "allreduce":
            self.buf = torch.zeros((world - 1) * self.plane, device=device, dtype=dtype)

    def exchange_add(self, dst):
        """dst: 1-D tensor of the slab's dofs (partial sums on interface planes) -> full sums"""
        if self.world == 1 or not self.idx:
            return
        torch, dist = self.torch, self.dist
        if self.mode == "p2p":
            ops, views = [], {}
            for which in self.idx:
                v = dst[self.slices[which]] if which in self.slices else dst[self.idx[which]]
                views[which] = v
                ops.append(dist.P2POp(dist.isend, v, self.peer[which]))
                ops.append(dist.P2POp(dist.irecv, self.recv[which], self.peer[which]))
            for r in dist.batch_isend_irecv(ops):
                r.wait()
            for which in self.idx:
                if which in self.slices:
                    views[which].addcmul_(self.recv[which], self.free[which])
                else:
                    dst[self.idx[which]] = views[which] + self.recv[which] * self.free[which]
        elif self.mode == "pair":
            # even interfaces (0-1, 2-3, ...) first, then odd ones: every rank is in at most one
            # collective per phase, so the two phases cannot deadlock
            for phase in (0, 1):
                for which in (0, 1):
                    if which not in self.idx:
                        continue
                    iface = self.rank if which == 1 else self.rank - 1  # interface index = lower rank
                    if iface % 2 != phase:
                        continue
                    mine = dst[self.idx[which]]
                    tot = mine.clone()
                    dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.pair_groups[which])
                    dst[self.idx[which]] = mine + (tot - mine) * self.free[which]
        elif self.mode == "allreduce":
            self.buf.zero_()
            for which in self.idx:
                iface = self.rank if which == 1 else self.rank - 1
                self.buf[iface * self.plane:(iface + 1) * self.plane] = dst[self.idx[which]]
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM)
            for which in self.idx:
                iface = self.rank if which == 1 else self.rank - 1
                mine = dst[self.idx[which]]
                tot = self.buf[iface * self.plane:(iface + 1) * self.plane]
                dst[self.idx[which]] = mine + (tot - mine) * self.free[which]
        else:
            raise ValueError(self.mode)


class DistributedLaplace:
    """LaplaceOperatorGpu over a z-slab partition: local vmult (HIP, through the C-ABI) + exchange.
    `local_vmult(dst, src)` is injectable so the CPU tests can drive the same exchange code with the
    oracle as the local operator."""

    def __init__(self, mesh, rank, world, device, dtype, local_vmult, mode="p2p"):
        self.local_vmult = local_vmult
        self.exchange = SlabExchange(mesh, rank, world, device, dtype, mode)

    def vmult(self, dst, src):
        self.local_vmult(dst, src)
        self.exchange.exchange_add(dst)
