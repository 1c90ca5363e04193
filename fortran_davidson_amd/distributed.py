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
        return _roundup(self.nranks * self.nslab, 256)      # whole 256-row blocks: the symmetric sweeps index by tile

    def rows(self):
        return self.row0, self.row0 + self.nloc


SYM_TB = 256          # tile edge of the symmetric-tiled storage
SYM_GROUP = 4         # block rows per ownership group (what the 2- and 4-block-row schedules nest in)
_OWNER_CACHE: dict = {}


def sym_group_owners(nb: int, nranks: int) -> list:
    """Owners of the groups of 4 block rows: longest group first, each to the rank that holds the fewest tiles so far
    (ties: lowest rank) - same table as sym_group_owners in csrc/engine.hip."""
    ng = (nb + SYM_GROUP - 1) // SYM_GROUP
    owner, load = [0] * ng, [0] * nranks
    for q in range(ng - 1, -1, -1):
        tiles = sum(i + 1 for i in range(SYM_GROUP * q, min(nb, SYM_GROUP * q + SYM_GROUP)))
        best = min(range(nranks), key=lambda r: (load[r], r))
        owner[q] = best
        load[best] += tiles
    return owner


@dataclass(frozen=True)
class SymmetricTileOwnership:
    """Which block rows of the lower block triangle rank `rank` stores, and where (same arithmetic as sym_setup)."""
    n: int
    nranks: int
    rank: int

    @property
    def nb(self) -> int:
        return RowPartition(self.n, self.nranks, self.rank).ncols_pad // SYM_TB

    def owner(self, block_row: int) -> int:
        return self._owners()[block_row // SYM_GROUP]

    def _owners(self):
        key = (self.nb, self.nranks)
        if key not in _OWNER_CACHE:
            _OWNER_CACHE[key] = sym_group_owners(*key)
        return _OWNER_CACHE[key]

    def row_off(self):
        """first tile of every block row in this rank's storage (-1: another rank's); local tile count"""
        off, count = [], 0
        for i in range(self.nb):
            if self.owner(i) == self.rank:
                off.append(count)
                count += i + 1
            else:
                off.append(-1)
        return off, count


def exchange_unique_id(dist, rank: int) -> bytes:
    """Rank 0 creates the RCCL unique id, every rank receives it (any torch.distributed backend)."""
    from .engine_c import CEngine
    ident = [CEngine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ident, src=0)
    return ident[0]
