#!/usr/bin/env python3
"""bench.py - Davidson iterations/s + achieved A*V HBM GB/s vs roofline (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one complete `generalized_eigensolver` solve (Fortran driver loop on the HIP engine) of
the workload BASELINE.json quotes the metric on that fits one GPU: configs[1] = N=20000 dense fp64,
lowest=8, DPR, tol=1e-8, generate_diagonal_dominant(N, 1e-3) - the matrix is generated in HBM before
the timed region and stays resident.  value = Davidson iterations per second over the K timed solves.
N > 1: the same problem row-partitioned over the ranks (strong scaling), new-basis block exchanged with
an RCCL all-gather inside libdavidson_hip.so; torch.distributed (gloo) is only the control plane
(unique-id broadcast, barriers, max-over-ranks of the time).

Extra objects in the JSON line: `roofline` (dominant kernel = dense block matvec, HIP-event timed on
the engine's stream inside the timed region), `cpu_baseline` (the reference itself, oracle/_ref, on
the host cores in a child process), `apply_k8` (the north-star microbenchmark: A*V at k=8) and
`large` (configs[2]: N=200000, lowest=16, restart at 80 - symmetric-tiled storage on one GPU, full row
slabs on several).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s measured achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", "--order", dest="n", type=int, default=20000,
                    help="matrix order (use --order under torch.distributed.run, whose parser rejects the prefix --n)")
    ap.add_argument("--lowest", type=int, default=8)
    ap.add_argument("--sparsity", type=float, default=1e-3)
    ap.add_argument("--tol", type=float, default=1e-8)
    ap.add_argument("--large-n", type=int, default=200000, help="order of the configs[2] solve (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--control-plane-only", action="store_true",
                    help="exercise the launch plumbing (rendezvous, id broadcast, barrier, max over ranks) without a GPU")
    ap.add_argument("--cpu-n", type=int, default=0, help="order for the CPU baseline (0 = same as --n)")
    return ap.parse_args()


CPU_CHILD = r"""
import ctypes, json, os, sys, time
sys.path.insert(0, {root!r})
import numpy as np
lowest, tol = {lowest}, {tol}
A = np.load({path!r}, mmap_mode="r")
A = np.asfortranarray(A)
from oracle import ref, davidson_oracle as O        # never imports the product, never touches the GPU
out = dict(n=int(A.shape[0]))
if ref.available():
    ref.lib()
    try:
        threads = int(ctypes.CDLL("/opt/conda/lib/libmkl_rt.so").mkl_get_max_threads())
    except Exception:
        threads = os.cpu_count()
    t = time.perf_counter(); lam, vec, it = ref.dense_solve(A, lowest, "DPR", 1000, tol); dt = time.perf_counter() - t
    out.update(kind="reference", iters=int(it), seconds=dt, cores=threads, evals=[float(x) for x in lam])
else:
    t = time.perf_counter(); lam, vec, it = O.generalized_eigensolver_dense(A, lowest, "DPR", 1000, tol); dt = time.perf_counter() - t
    out.update(kind="port", iters=int(it), seconds=dt, cores=os.cpu_count(), evals=[float(x) for x in lam])
print("CPU_BASELINE " + json.dumps(out))
"""


def cpu_baseline(n, lowest, sparsity, tol):
    """The reference's own CPU+LAPACK path (oracle/_ref = the reference compiled with flang + MKL) on
    the same generate_diagonal_dominant input, timed in a child process that never touches the GPU,
    never imports torch (its libgomp breaks threaded MKL) and never loads the product libraries
    (they pin MKL to its sequential layer).  The input is written by the host-side Fortran generator
    of the product (bit-identical to the device generator) to a scratch file."""
    import tempfile
    import numpy as np
    import fortran_davidson_amd as fd
    path = os.path.join(tempfile.gettempdir(), f"davidson_cpu_baseline_{os.getpid()}.npy")
    try:
        np.save(path, fd.generate_diagonal_dominant(n, sparsity, None, 1))
        code = CPU_CHILD.format(root=ROOT, lowest=lowest, tol=tol, path=path)
        env = dict(os.environ)
        env["HIP_VISIBLE_DEVICES"] = ""
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
        for line in res.stdout.splitlines():
            if line.startswith("CPU_BASELINE "):
                return json.loads(line[len("CPU_BASELINE "):])
        return {"error": (res.stderr or res.stdout)[-400:]}
    except Exception as exc:       # noqa: BLE001
        return {"error": repr(exc)}
    finally:
        if os.path.exists(path):
            os.remove(path)


def pmc_traffic(n):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (profiles/), which
    was collected with rocprofv3 on this same command; None when no summary matches the workload."""
    path = os.path.join(ROOT, "profiles", f"r01_pmc_traffic_n{n}.json")
    try:
        with open(path) as f:
            k = json.load(f)["kernels"]
        vals = [k[name]["hbm_bytes_per_launch_corrected"] for name in ("matvec_dense_kernel<1>", "matvec_dense_kernel<2>")
                if name in k]
        return round(sum(vals) / len(vals), 0) if vals else None
    except Exception:      # noqa: BLE001
        return None


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if args.control_plane_only:
        # CPU-testable part of the multi-GPU launch: what bench.py does around the engine
        ident = [bytes(range(128)) if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(ident, src=0)
            dist.barrier()
            t = torch.tensor([float(rank + 1)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            assert float(t.item()) == float(world)
        assert ident[0] == bytes(range(128))
        from fortran_davidson_amd.distributed import RowPartition
        part = RowPartition(args.n, world, rank)
        if rank == 0:
            print(json.dumps({"control_plane": "ok", "world": world, "rows_rank0": list(part.rows())}))
        if world > 1:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # DAVIDSON_TRANSPORT=shm: all ranks share GPU 0 and exchange through shared memory - a test transport
    # that runs this whole multi-process flow on a single-GPU box (tests/test_bench_multiprocess_gpu.py).
    # The multi-GPU data path is RCCL, one GPU per rank.
    transport = os.environ.get("DAVIDSON_TRANSPORT", "rccl")
    device = local_rank if transport == "rccl" else 0
    torch.cuda.set_device(device)

    import fortran_davidson_amd as fd
    engines_made = [0]

    def make_engine(n, lowest, max_dim=None, storage="full"):
        eng = fd.DavidsonEngine(n, lowest, max_dim, gev=False, device=device, rank=rank, nranks=world, storage=storage)
        if world > 1:
            engines_made[0] += 1
            if transport == "shm":
                ident = [f"/dav_bench_{os.getpid()}_{engines_made[0]}" if rank == 0 else None]
                dist.broadcast_object_list(ident, src=0)
                eng.c.comm_init_shm(ident[0])
            else:
                ident = [fd.CEngine.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ident, src=0)
                eng.comm_init(ident[0])
        return eng

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- the timed workload: configs[1] -----------------------------------------------------------
    n, lowest = args.n, args.lowest
    eng = make_engine(n, lowest)
    eng.generate_diagonal_dominant(1, args.sparsity, seed=1)       # resident in HBM before timing
    for _ in range(args.warmup):
        lam, _, iters = eng.solve("DPR", 1000, args.tol, want_vectors=False)
    eng.c.synchronize()
    eng.c.reset_stats()
    barrier()
    t0 = time.perf_counter()
    total_iters = 0
    for _ in range(args.steps):
        lam, _, iters = eng.solve("DPR", 1000, args.tol, want_vectors=False)
        total_iters += iters
    eng.c.synchronize()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    st = eng.c.stats()
    value = total_iters / elapsed

    # roofline of the dominant kernel over the timed region (HIP events on the engine's stream)
    ach = st.apply_bytes / (st.apply_ms * 1e-3) / 1e9 if st.apply_ms > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": "matvec_dense_kernel<NT> (A*V block matvec, full storage)",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                "traffic": pmc_traffic(n) if world == 1 else None, "launches": int(st.applies), "avg_launch_ms": round(st.apply_ms / max(st.applies, 1), 4),
                "algorithmic_bytes_per_launch": round(st.apply_bytes / max(st.applies, 1), 0),
                "note": "per-rank; bytes = 8*nloc*N + 16*N*k per launch (SURVEY 8d)"}
    # device time by phase: a separate, untimed pass with every phase bracketed by events (the timed region
    # above only brackets the block matvec - an event pair costs ~5 us of host time per launch group)
    eng.c.set_timing(2)
    eng.c.reset_stats()
    nph = min(args.steps, 5)
    for _ in range(nph):
        eng.solve("DPR", 1000, args.tol, want_vectors=False)
    eng.c.synchronize()
    sp = eng.c.stats()
    phase = {"apply_ms": round(sp.apply_ms / nph, 4), "gram_ms": round(sp.gram_ms / nph, 4),
             "panel_ms": round(sp.panel_ms / nph, 4), "comm_ms": round(sp.comm_ms / nph, 4)}
    eng.c.set_timing(1)

    # north-star microbenchmark: A*V at k=8 (and 16, 32) on the resident matrix
    apply_k = {}
    for k in (8, 16, 32):
        ms, nbytes = eng.c.bench_apply(k, 20)
        apply_k[f"k{k}"] = {"ms": round(ms, 4), "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1),
                            "frac_of_8TBps": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
    # opt-in correction policy (not the reference's, so never part of `value`): same workload,
    # corrections only for the wanted pairs that have not converged
    eng.set_correction_policy("unconverged")
    eng.solve("DPR", 1000, args.tol, want_vectors=False)
    eng.c.synchronize(); barrier()
    t2 = time.perf_counter()
    it_pol = 0
    for _ in range(args.steps):
        lam_pol, _, it = eng.solve("DPR", 1000, args.tol, want_vectors=False)
        it_pol += it
    eng.c.synchronize(); barrier()
    dt_pol = max_over_ranks(time.perf_counter() - t2)
    opt_in = {"policy": "unconverged (engine_set_correction_policy / DAVIDSON_CORRECTION_POLICY)",
              "ms_per_solve": round(dt_pol / args.steps * 1e3, 4), "iters_per_solve": it_pol // args.steps,
              "solves_per_s": round(args.steps / dt_pol, 2),
              "reference_policy_solves_per_s": round(args.steps / elapsed, 2),
              "max_abs_eigenvalue_diff_vs_reference_policy": float(np.abs(lam_pol - lam).max())}
    eng.close()

    # same workload with the matrix kept as its lower block triangle (engine option, single GPU): every
    # off-diagonal tile is read once and used twice, so a sweep moves half the bytes
    sym = None
    if world == 1:
        try:
            es = make_engine(n, lowest, None, "symmetric")
            es.generate_diagonal_dominant(1, args.sparsity, seed=1)
            es.solve("DPR", 1000, args.tol, want_vectors=False)
            es.c.synchronize()
            t4 = time.perf_counter()
            it_s = 0
            for _ in range(args.steps):
                lam_s, _, it = es.solve("DPR", 1000, args.tol, want_vectors=False)
                it_s += it
            es.c.synchronize()
            dt_s = time.perf_counter() - t4
            sym = {"storage": "symmetric-tiled (dav_set_storage / engine_set_storage)", "ms_per_solve": round(dt_s / args.steps * 1e3, 4),
                   "iterations_per_s": round(it_s / dt_s, 2), "iters_per_solve": it_s // args.steps,
                   "max_abs_eigenvalue_diff_vs_full_storage": float(np.abs(lam_s - lam).max())}
            es.close()
        except Exception as exc:       # noqa: BLE001
            sym = {"error": repr(exc)[:300]}

    # ---- configs[2]: N=200000 dense fp64, lowest=16, DPR, subspace restart at 80 ------------------------
    # one GPU: symmetric-tiled storage (160 GB, K1s sweep); >= 2 GPUs: full row slabs + RCCL all-gather
    large = None
    if args.large_n > 0:
        try:
            storage = "symmetric" if world == 1 else "full"
            big = make_engine(args.large_n, 16, 80, storage)
            big.generate_diagonal_dominant(1, args.sparsity, seed=1)
            big.solve("DPR", 1000, args.tol, want_vectors=False)
            big.c.synchronize(); big.c.reset_stats(); barrier()
            t1 = time.perf_counter()
            reps = 2
            it_big = 0
            for _ in range(reps):
                lam_big, _, it = big.solve("DPR", 1000, args.tol, want_vectors=False)
                it_big += it
            big.c.synchronize(); barrier()
            dt = max_over_ranks(time.perf_counter() - t1)
            sb = big.c.stats()
            ms8, b8 = big.c.bench_apply(8, 5)
            large = {"workload": f"N={args.large_n} dense fp64, lowest=16, DPR, max_dim_sub=80, storage={storage}, "
                                 f"{world} GPU(s)",
                     "iterations_per_s": round(it_big / dt, 3), "iters_per_solve": it_big // reps,
                     "ms_per_solve": round(dt / reps * 1e3, 2),
                     "apply_GBps_per_rank": round(sb.apply_bytes / (sb.apply_ms * 1e-3) / 1e9, 1),
                     "apply_GBps_note": "stored bytes counted once per launch; a launch covers up to 32 columns "
                                        "(symmetric storage: two paired 16-column groups, MFMA/issue bound)",
                     "apply_k8": {"ms": round(ms8, 3), "GBps_per_rank": round(b8 / (ms8 * 1e-3) / 1e9, 1),
                                  "frac_of_8TBps": round(b8 / (ms8 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                  "algorithmic_bytes": b8,
                                  "note": "bytes = 8*S + 16*N*k, S = N(N+1)/2 (symmetric-tiled) or nloc*N (full slab)"},
                     "eigenvalues": [float(x) for x in lam_big[:3]]}
            big.set_correction_policy("unconverged")
            big.solve("DPR", 1000, args.tol, want_vectors=False)
            big.c.synchronize(); barrier()
            t3 = time.perf_counter()
            lam_p, _, it_p = big.solve("DPR", 1000, args.tol, want_vectors=False)
            big.c.synchronize(); barrier()
            dt_p = max_over_ranks(time.perf_counter() - t3)
            large["opt_in_policy_unconverged"] = {"ms_per_solve": round(dt_p * 1e3, 2), "iters_per_solve": int(it_p),
                                                  "max_abs_eigenvalue_diff": float(np.abs(lam_p - lam_big).max())}
            big.close()
        except Exception as exc:       # noqa: BLE001
            large = {"error": repr(exc)[:300]}

    # ---- CPU baseline: rank 0, N=1 only -----------------------------------------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cn = args.cpu_n or n
        raw = cpu_baseline(cn, lowest, args.sparsity, args.tol)
        if "seconds" in raw:
            cpu = {"value": round(raw["iters"] / raw["seconds"], 4), "unit": "Davidson iterations/s", "cores": raw["cores"],
                   "kind": raw["kind"], "seconds": round(raw["seconds"], 3), "iters": raw["iters"],
                   "sample": f"one full solve of the same workload (N={cn}, lowest={lowest}, DPR, tol={args.tol}) by the "
                             "reference built with flang+MKL (oracle/_ref), all host threads",
                   "max_abs_eigenvalue_diff_vs_gpu": float(np.abs(np.array(raw["evals"]) - lam).max()) if cn == n else None}
        else:
            cpu = raw

    if rank == 0:
        line = {"metric": "Davidson iterations/sec (dense DPR solve, matrix resident in HBM) + A*V HBM GB/s vs roofline",
                "value": round(value, 3), "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"N={n} dense fp64 full storage, lowest={lowest}, DPR, tol={args.tol}, "
                                       f"generate_diagonal_dominant(N,{args.sparsity}) seed 1, max_dim={10 * lowest}",
                           "N": n, "lowest": lowest, "iters_per_solve": total_iters // args.steps,
                           "basis_widths": "2L,4L,8L", "parallelism": f"row-slab x{world}"},
                "eigenvalues": [float(x) for x in lam[:3]],
                "roofline": roofline, "phase_ms_per_step": phase, "apply": apply_k, "opt_in_policy": opt_in, "symmetric_storage": sym, "large": large, "cpu_baseline": cpu}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
