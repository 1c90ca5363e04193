"""TEST INFRASTRUCTURE: a numpy model of the engine's row-slab protocol, one instance per rank, with
torch.distributed (gloo) standing in for RCCL.  It follows the driver loop of
fortran_davidson_amd/fortran/davidson.f90 phase by phase (what is local, what is all-gathered, what is
all-reduced) so that the multi-rank formulation can be checked on CPU against the single-process oracle."""
import numpy as np
import torch
import torch.distributed as dist

from fortran_davidson_amd.distributed import RowPartition
from oracle import davidson_oracle as O


def allreduce(x):
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy()


def allgather_rows(local, part):
    """all-gather of equal nslab-row slabs (zero padded), like the packed Xt exchange"""
    k = local.shape[1]
    send = np.zeros((part.nslab, k))
    send[: part.nloc] = local
    bufs = [torch.zeros(part.nslab, k, dtype=torch.float64) for _ in range(part.nranks)]
    dist.all_gather(bufs, torch.from_numpy(send))
    return np.vstack([b.numpy() for b in bufs])[: part.n]


def gram(P, Q):
    return allreduce(P.T @ Q)


def orthonormalise(V, T):
    """block_orthonormalise: SVQB passes with all-reduced Gram blocks"""
    for p in range(6):
        C = gram(V, T) if V.shape[1] else np.zeros((0, T.shape[1]))
        G = gram(T, T)
        Gp = G - C.T @ C
        d = 1.0 / np.sqrt(np.diag(Gp))
        w, U = np.linalg.eigh(Gp * d[:, None] * d[None, :])
        wmin, wmax = w.min(), w.max()
        w = np.maximum(w, 1e-14 * wmax)
        M = d[:, None] * U / np.sqrt(w)[None, :]
        T = (T - V @ C) @ M
        if p >= 1 and wmin > 0.5 and wmax < 2.0:
            break
    return T


class SymmetricTiles:
    """The symmetric-tiled storage dealt out over the ranks: this rank keeps the tiles (I, J <= I) of the block rows
    it owns (groups of 4, longest first to the least loaded rank).  apply(): the exchange steps of the engine's multi-rank symmetric sweep -
    all-gather of the block, this rank's partial of the WHOLE product from its tiles (direct and transposed product of
    every off-diagonal tile), reduce-scatter of the partials (gloo has no reduce-scatter: all-reduce + own chunk is
    the same sum), rows of this rank's slab."""

    def __init__(self, n, sparsity, diag_val, seed, part):
        from fortran_davidson_amd.distributed import SymmetricTileOwnership, SYM_TB
        self.part, self.tb = part, SYM_TB
        own = SymmetricTileOwnership(n, part.nranks, part.rank)
        self.tiles = {}
        for I in range(own.nb):
            r0, r1 = I * SYM_TB, min(n, (I + 1) * SYM_TB)
            if own.owner(I) != part.rank or r0 >= n:
                continue
            rows = O.generate_diagonal_dominant(n, sparsity, diag_val, seed=seed, rows=(r0, r1))   # block row I, all columns
            for J in range(I + 1):
                self.tiles[(I, J)] = rows[:, J * SYM_TB:min(n, (J + 1) * SYM_TB)]

    def diagonal(self, n):
        d = np.zeros(n)
        for (I, J), t in self.tiles.items():
            if I == J:
                d[I * self.tb:I * self.tb + t.shape[0]] = np.diag(t)
        return allreduce(d)

    def apply(self, X):
        part = self.part
        Xg = allgather_rows(X, part)
        Wp = np.zeros((part.nranks * part.nslab, X.shape[1]))
        for (I, J), t in self.tiles.items():
            ri, rj = slice(I * self.tb, I * self.tb + t.shape[0]), slice(J * self.tb, J * self.tb + t.shape[1])
            Wp[ri] += t @ Xg[rj]
            if I != J:
                Wp[rj] += t.T @ Xg[ri]
        Ws = allreduce(Wp)
        return Ws[part.row0:part.row0 + part.nloc]


def sharded_dense_dpr(n, lowest, sparsity, seed, max_it, tol, max_dim=None, seed_b=None, storage="full"):
    rank, world = dist.get_rank(), dist.get_world_size()
    part = RowPartition(n, world, rank)
    r0, r1 = part.rows()
    gev = seed_b is not None
    if storage == "symmetric":
        return _sharded_dense_dpr_symmetric(n, lowest, sparsity, seed, max_it, tol, max_dim, seed_b, part)
    A = O.generate_diagonal_dominant(n, sparsity, seed=seed, rows=(r0, r1))          # local slab, all columns
    B = O.generate_diagonal_dominant(n, sparsity, 1.0, seed=seed_b, rows=(r0, r1)) if gev else None
    dA = allgather_rows(A[np.arange(r1 - r0), np.arange(r0, r1)][:, None], part)[:, 0]
    dB = allgather_rows(B[np.arange(r1 - r0), np.arange(r0, r1)][:, None], part)[:, 0] if gev else np.ones(n)
    max_dim = 10 * lowest if max_dim is None else max_dim
    idx = O.lowest_diagonal_indices(dA, 2 * lowest)
    m = 2 * lowest
    V = np.zeros((r1 - r0, m))
    for c, g in enumerate(idx):
        if r0 <= g < r1:
            V[g - r0, c] = 1.0
    apply = lambda Mx, X: Mx @ allgather_rows(X, part)        # the one exchange step of the apply
    W = apply(A, V)
    BV = apply(B, V) if gev else V
    H = gram(V, W)
    S = gram(V, BV) if gev else None
    conv = np.zeros(lowest, bool)
    widths = []
    for it in range(1, max_it + 1):
        theta, Y = O.lapack_generalized_eigensolver(H, S)
        widths.append(m)
        X = V @ Y[:, :lowest]
        R = W @ Y - (BV @ Y) * theta[None, :]
        err = np.sqrt(allreduce(np.sum(R[:, :lowest] ** 2, axis=0)))
        conv |= err < tol
        if conv.all():
            return theta[:lowest], allgather_rows(X, part), it, widths
        if m <= max_dim:
            T = R / (theta[None, :] * dB[r0:r1, None] - dA[r0:r1, None])
            T = orthonormalise(V, T)
            V = np.hstack([V, T])
            W = np.hstack([W, apply(A, T)])
            BV = np.hstack([BV, apply(B, T)]) if gev else V
            m *= 2
        else:
            V = V @ Y[:, : 2 * lowest]
            if gev:
                V = orthonormalise(V[:, :0], V)
            m = 2 * lowest
            W = apply(A, V)
            BV = apply(B, V) if gev else V
        H = gram(V, W)
        S = gram(V, BV) if gev else None
    return theta[:lowest], allgather_rows(X, part), max_it + 1, widths


def _sharded_dense_dpr_symmetric(n, lowest, sparsity, seed, max_it, tol, max_dim, seed_b, part):
    """the same loop with the operators as symmetric tiles dealt out over the ranks (SymmetricTiles.apply)"""
    r0, r1 = part.rows()
    gev = seed_b is not None
    At = SymmetricTiles(n, sparsity, None, seed, part)
    Bt = SymmetricTiles(n, sparsity, 1.0, seed_b, part) if gev else None
    dA = At.diagonal(n)
    dB = Bt.diagonal(n) if gev else np.ones(n)
    max_dim = 10 * lowest if max_dim is None else max_dim
    idx = O.lowest_diagonal_indices(dA, 2 * lowest)
    m = 2 * lowest
    V = np.zeros((r1 - r0, m))
    for c, g in enumerate(idx):
        if r0 <= g < r1:
            V[g - r0, c] = 1.0
    W = At.apply(V)
    BV = Bt.apply(V) if gev else V
    H = gram(V, W)
    S = gram(V, BV) if gev else None
    conv = np.zeros(lowest, bool)
    widths = []
    for it in range(1, max_it + 1):
        theta, Y = O.lapack_generalized_eigensolver(H, S)
        widths.append(m)
        X = V @ Y[:, :lowest]
        R = W @ Y - (BV @ Y) * theta[None, :]
        err = np.sqrt(allreduce(np.sum(R[:, :lowest] ** 2, axis=0)))
        conv |= err < tol
        if conv.all():
            return theta[:lowest], allgather_rows(X, part), it, widths
        if m <= max_dim:
            T = R / (theta[None, :] * dB[r0:r1, None] - dA[r0:r1, None])
            T = orthonormalise(V, T)
            V = np.hstack([V, T])
            W = np.hstack([W, At.apply(T)])
            BV = np.hstack([BV, Bt.apply(T)]) if gev else V
            m *= 2
        else:
            V = V @ Y[:, : 2 * lowest]
            if gev:
                V = orthonormalise(V[:, :0], V)
            m = 2 * lowest
            W = At.apply(V)
            BV = Bt.apply(V) if gev else V
        H = gram(V, W)
        S = gram(V, BV) if gev else None
    return theta[:lowest], allgather_rows(X, part), max_it + 1, widths
