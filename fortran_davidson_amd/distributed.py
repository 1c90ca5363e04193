"""Launch-side helpers for the row-slab sharded engine (one process per GPU).

The data path (all-gather of the new basis block, all-reduce of the small Gram blocks) is RCCL inside
libdavidson_hip.so; this module only mirrors the partition arithmetic of dav_create (csrc/engine.hip)
for the launcher/tests and distributes the RCCL unique id over an existing torch.distributed group.
"""
from __future__ import annotations

from dataclasses import dataclass


def _roundup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass(frozen=True)
class RowPartition:
    """Rows [row0, row0 + nloc) of rank `rank`; same formulas as dav_create."""
    n: int
    nranks: int
    rank: int

    @property
    def nslab(self) -> int:
        return _roundup((self.n + self.nranks - 1) // self.nranks, 16)

    @property
    def row0(self) -> int:
        return self.rank * self.nslab

    @property
    def nloc(self) -> int:
        return max(0, min(self.nslab, self.n - self.row0))

    @property
    def nloc_pad(self) -> int:
        return _roundup(self.nslab, 256)

    @property
    def ncols_pad(self) -> int:
        return _roundup(self.nranks * self.nslab, 64)

    def rows(self):
        return self.row0, self.row0 + self.nloc


def exchange_unique_id(dist, rank: int) -> bytes:
    """Rank 0 creates the RCCL unique id, every rank receives it (any torch.distributed backend)."""
    from .engine_c import CEngine
    ident = [CEngine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ident, src=0)
    return ident[0]
