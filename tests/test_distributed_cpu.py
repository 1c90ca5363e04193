"""CPU, world_size 2 (gloo): the row-slab formulation of the engine (what is all-gathered, what is
all-reduced, how rows are partitioned) reproduces the single-process oracle and the golden values."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from fortran_davidson_amd.distributed import RowPartition
from oracle import davidson_oracle as O


def test_row_partition_arithmetic():
    for n, p in [(20000, 1), (20000, 8), (1000, 3), (50, 2), (17, 4), (200000, 8), (1000000, 8)]:
        parts = [RowPartition(n, p, r) for r in range(p)]
        assert sum(q.nloc for q in parts) == n
        assert all(q.nslab % 16 == 0 and q.nloc_pad % 256 == 0 and q.ncols_pad % 64 == 0 for q in parts)
        assert all(q.ncols_pad >= p * q.nslab >= n for q in parts)
        covered = []
        for q in parts:
            covered += list(range(*q.rows()))[:3] + list(range(*q.rows()))[-3:]
            assert q.row0 == q.rank * q.nslab          # gathered index == global index
        assert max(covered) == n - 1


def test_symmetric_tile_ownership_arithmetic():
    """the lower block triangle dealt out by groups of 4 block rows, longest first to the least loaded rank: one owner per
    block row, contiguous storage offsets, super rows of 2 and 4 block rows never straddle two owners, tiles balanced
    within half a per cent on large problems"""
    from fortran_davidson_amd.distributed import SymmetricTileOwnership, sym_group_owners
    assert sym_group_owners(20, 2) == [0, 0, 1, 1, 0]            # groups of 10, 26, 42, 58, 74 tiles: 74 -> 0, 58 -> 1, 42 -> 1, 26 -> 0, 10 -> 0 (tie)
    for n, p in [(200000, 8), (1000000, 8), (20000, 2), (2305, 5), (300, 3), (50, 2)]:
        owns = [SymmetricTileOwnership(n, p, r) for r in range(p)]
        nb = owns[0].nb
        assert nb * 256 >= n
        counts = []
        for o in owns:
            off, count = o.row_off()
            counts.append(count)
            mine = [i for i in range(nb) if off[i] >= 0]
            assert all(o.owner(i) == o.rank for i in mine)
            assert [off[i] for i in mine] == list(np.cumsum([0] + [i + 1 for i in mine[:-1]]))[:len(mine)]   # contiguous, in block-row order (a rank may own nothing)
            assert all(o.owner(i) == o.owner(i - i % 4) for i in range(nb))                           # groups of 4 have one owner
        assert sum(counts) == nb * (nb + 1) // 2
        if nb >= 64 * p:
            assert max(counts) <= 1.005 * (sum(counts) / p)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sharded_model import sharded_dense_dpr
    lam, vec, iters, widths = sharded_dense_dpr(**case)
    if rank == 0:
        np.savez(out, lam=lam, vec=vec, iters=iters, widths=np.array(widths))
    dist.destroy_process_group()


def _run(case, tmp_path, world=2):
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(world, _free_port(), case, out), nprocs=world, join=True)
    return np.load(out)


@pytest.mark.parametrize("storage", ["full", "symmetric"])
@pytest.mark.parametrize("name", ["c1_n50_std_dpr", "n1000_restart_dpr", "n1000_gev_restart_dpr"])
def test_sharded_formulation_matches_reference_golden(golden, tmp_path, name, storage):
    """row slabs of the operator (all-gather only) and symmetric tiles dealt out over the ranks (all-gather, per-rank
    partial of the whole product, reduce-scatter)"""
    manifest, arrays = golden
    c = manifest["dense"][name]
    case = dict(n=c["n"], lowest=c["lowest"], sparsity=c["sparsity"], seed=c["seed_a"], max_it=c["max_it"],
                tol=c["tol"], max_dim=c["max_dim"], seed_b=c["seed_b"], storage=storage)
    res = _run(case, tmp_path)
    assert np.abs(res["lam"] - arrays[f"{name}__evals"]).max() < 1e-8
    assert int(res["iters"]) == c["iters"]
    assert list(res["widths"]) == c["widths"]


@pytest.mark.parametrize("storage", ["full", "symmetric"])
def test_sharded_three_ranks_uneven_rows(tmp_path, storage):
    case = dict(n=333, lowest=3, sparsity=1e-2, seed=5, max_it=100, tol=1e-8, storage=storage)
    res = _run(case, tmp_path, world=3)
    A = O.generate_diagonal_dominant(333, 1e-2, seed=5)
    lam_o, vec_o, it_o = O.generalized_eigensolver_dense(A, 3, "DPR", 100, 1e-8)
    assert np.abs(res["lam"] - lam_o).max() < 1e-8
    assert int(res["iters"]) == it_o
    r = np.linalg.norm(A @ res["vec"] - res["vec"] * res["lam"][None, :], axis=0)
    assert (r < 1e-8).all()


def test_bench_launch_plumbing_world_size_2():
    """bench.py under torch.distributed.run with 2 ranks on CPU: rendezvous on 127.0.0.1, gloo group,
    unique-id broadcast, barrier, max over ranks (the control plane around the RCCL engine)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--control-plane-only"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["control_plane"] == "ok" and out["world"] == 2 and out["rows_rank0"] == [0, 100000]     # default order 200000 (configs[2]) over 2 ranks


def test_bench_starts_its_own_launcher_for_several_gpus():
    """`python bench.py --gpus 2` WITHOUT an outer launcher: the parent starts torch.distributed.run as a child process
    (it never touches a GPU itself), rank 0's JSON line comes through and the exit code is the child's; --n is
    forwarded as --order (the launcher's parser rejects --n)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--control-plane-only", "--n", "1000"],
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["control_plane"] == "ok" and out["world"] == 2 and out["rows_rank0"] == [0, 512]
    # a failing child propagates its exit code
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--control-plane-only", "--no-such-flag"],
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert bad.returncode != 0
