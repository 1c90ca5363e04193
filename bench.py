#!/usr/bin/env python3
"""bench.py - Davidson iterations/s + achieved A*V HBM GB/s vs roofline (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Several GPUs: the timed workload only, unless --all-legs (see its help).

A "step" is one complete `generalized_eigensolver` solve (Fortran driver loop on the HIP engine) of the
largest BASELINE.json configuration that fits one GPU: configs[2] = N=200000 dense fp64, lowest=16, DPR,
subspace restart at 80 (max_dim_sub=80), tol=1e-8, generate_diagonal_dominant(N, 1e-3).  The matrix is
generated in HBM before the timed region and stays resident as its lower block triangle (symmetric-tiled
storage, 160 GB); several GPUs deal the block rows out among themselves (N*N/2P entries each), all-gather the
new basis block and reduce-scatter the partial products with RCCL inside libdavidson_hip.so; torch.distributed (gloo) is only the
control plane (unique-id broadcast, barriers, max-over-ranks of the time).  value = Davidson iterations
per second over the K timed solves (strong scaling: the problem is the same for every N).

Objects in the JSON line besides the contract's fields:
  roofline       the dominant kernel of the timed region (the block matvec as the solve launches it: 32 / 64
                 columns per launch - bound by the fp64 matrix pipe), HIP-event timed inside the solves; its hbm_* keys
                 (and the nested `hbm` object) carry the NORTH-STAR measurement - A*V at N=200000, k=8 on the same resident
                 matrix, END TO END (operand packing + sweep kernel + fixed-order reduction) and the sweep kernel alone,
                 against 8 TB/s, against the copy / triad rate and against the read-only rate measured in the same run - and k=16 beside it
  scaling_model  one GPU: predicted ms per solve on 2 / 4 / 8 GPUs from this run's measured phases and the one-GPU rehearsal of P ranks
  configs4_free_harness  the reference's own benchmark program (src/benchmark_free.f90) (matrix-free test operator, B = I): its N=1000 configuration beside the
                 reference on the host cores, the same operator at N=10^5 / 10^6 against the measured fp64 transcendental rate
  comm           several GPUs: ranks RCCL reports, storage mode, per-solve all-gather / reduce-scatter / all-reduce ms and bytes
  apply          the same for k = 8, 16, 32
  hbm_measured   device copy / triad / read-only rate of this box (what 8 TB/s amount to in practice); HBM fractions are quoted against them too
  configs2_restart  configs[2] with a denser coupling: the solve goes through collapse restarts at full size
  configs3_gjd   BASELINE configs[3]: N=200000 generalized (A, B), GJD correction, lowest=8
  configs4_free  BASELINE configs[4]: matrix-free hashed diagonal-dominant operator, lowest=8, DPR
  small          BASELINE configs[1]: N=20000 dense, lowest=8, DPR (full storage)
  dropin         the reference-signature dense call (host matrix in, upload through PCIe included)
  cpu_baseline   the reference itself (oracle/_ref: flang + MKL build) on the host cores, N=20000 sample
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s measured achievable
# Generated (matrix-free) sweeps: fp64 MFMA and VALU instructions of a wave do not overlap (profiles/ubench/valucost.hip,
# r03_valucost.log: n VALU instructions behind an MFMA cost 12.5 + 4 n cycles, v_mul_lo_u32 and v_mad_u64_u32 included - none of
# them is quarter rate on gfx950), so one wave-evaluation (64 entries) costs the generator's 26 VALU instructions x 4 cycles
# (the interior-tile path of k_matvec_sym9.hip compiled on its own: per entry 5 v_xor, 4 v_mul_lo_u32, 2 v_mad_u64_u32, 2
# v_lshrrev_b64, 2 v_add3, 2 v_lshrrev_b32, 2 v_cvt_f64_u32, 1 each v_alignbit / v_add_f64 / v_fmac_f64 / v_mul_f64, ~1.3
# 64-bit adds) PLUS the two 64-cycle MFMAs that consume it in a 16-column sweep (direct + transposed product).
GEN_CYCLES_PER_WAVE_EVALUATION = 26 * 4 + 2 * 64
# the reference's matrix-free test operator in its one-variable form (csrc/common.h: dav_harness_poly): 2 additions + 17 FMAs per entry
HARNESS_CYCLES_PER_WAVE_EVALUATION = 19 * 4 + 2 * 64
HARNESS_CYCLES_32 = (19 + 1.5) * 4 + 4 * 64       # the wide generating kernel: + 12 v_accvgpr_read per 8 entries, 4 MFMAs per 64 entries
HARNESS_MODEL = ("1024 SIMDs x 2.4 GHz x 64 lanes / 204 cycles per wave-evaluation of 64 entries and 16 columns: 19 fp64 VALU instructions x 4 cycles "
                 "(x = 1 - |l_i - l_j|, degree-17 Horner) + 2 fp64 MFMAs x 64 cycles (direct + transposed product) on one issue port; "
                 "32 columns per launch (matvec_symw_kernel<2, ., ., 2>): 20.5 x 4 + 4 x 64 = 338")
GEN_MODEL = ("1024 SIMDs x 2.4 GHz x 64 lanes / 232 cycles per wave-evaluation: 26 VALU instructions x 4 cycles (splitmix64 + key + "
             "conversion; no quarter-rate instruction among them, profiles/ubench/r03_valucost.log) + 2 fp64 MFMAs x 64 cycles (16 columns, "
             "direct + transposed product) - the two kinds of instruction do not overlap within a SIMD (same log: n VALU instructions "
             "behind an MFMA cost 12.5 + 4 n cycles); the LDS transposition, X_I reads and the exchange of the sweep are not in the model")
FP64_MFMA_PEAK_TFLOPS = 78.6    # 256 CUs x 4 SIMDs x (16x16x4 MACs / 64 cycles) x 2 x 2.4 GHz (measured: 64.0 cycles per
                                # v_mfma_f64_16x16x4_f64, 16.3 per v_mfma_f64_4x4x4_4b_f64 - profiles/ubench)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", "--order", dest="n", type=int, default=200000,
                    help="order of the timed workload (use --order under torch.distributed.run, whose parser rejects --n)")
    ap.add_argument("--lowest", type=int, default=16)
    ap.add_argument("--max-dim", type=int, default=80)
    ap.add_argument("--sparsity", type=float, default=1e-3)
    ap.add_argument("--tol", type=float, default=1e-8)
    ap.add_argument("--storage", default="auto", help="auto = symmetric tiles (one GPU, or dealt out over the ranks); full = full row slabs")
    ap.add_argument("--restart-sparsity", type=float, default=2e-2, help="coupling of the restart-forcing configs[2] leg (0 = skip)")
    ap.add_argument("--small-n", type=int, default=20000, help="order of the configs[1] leg (0 = skip)")
    ap.add_argument("--gjd-n", type=int, default=-1, help="order of the configs[3] leg (-1 = same as --order, 0 = skip)")
    ap.add_argument("--free-n", type=int, default=1000000, help="order of the configs[4] leg (0 = skip)")
    ap.add_argument("--harness-n", type=int, default=100000, help="order of the benchmark_free leg's large solve (0 = skip the leg)")
    ap.add_argument("--harness-n2", type=int, default=1000000, help="order of that leg's full solve at configs[4]'s order, lowest = 8 (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="only the timed workload and its roofline objects")
    ap.add_argument("--all-legs", action="store_true",
                    help="several GPUs: also run the configs[1] / [3] / [4] legs (default there: the timed workload only - a leg that fails on ONE "
                         "rank would leave its peers in a collective, and the watchdog's exit would take the headline measurement with it)")
    ap.add_argument("--control-plane-only", action="store_true",
                    help="exercise the launch plumbing (rendezvous, id broadcast, barrier, max over ranks) without a GPU")
    ap.add_argument("--cpu-n", type=int, default=0, help="order for the CPU baseline (0 = same as --small-n)")
    ap.add_argument("--cpu-n2", type=int, default=40000, help="second, larger order for the CPU baseline (0 = none)")
    return ap.parse_args()


CPU_CHILD = r"""
import ctypes, json, os, sys, time
sys.path.insert(0, {root!r})
import numpy as np
lowest, tol, sparsity, n_list = {lowest}, {tol}, {sparsity}, {n_list}
from oracle import ref, davidson_oracle as O        # never imports the product, never touches the GPU
out = dict(runs=[])
def numa_nodes():
    try:
        return len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
    except OSError:
        return None
out["numa_nodes"] = numa_nodes()
out["logical_cpus"] = os.cpu_count()
try:
    out["sockets"] = len(set(l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("physical id")))
    out["cpu_model"] = next(l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
except Exception:
    pass
if ref.available():
    ref.lib()
    mkl = ctypes.CDLL("/opt/conda/lib/libmkl_rt.so")
    threads = int(mkl.mkl_get_max_threads())
    out.update(kind="reference", cores=threads)
    dp = ctypes.POINTER(ctypes.c_double)
    for n in n_list:
        # The matrix is produced INSIDE this process by the reference-side driver under OpenMP (oracle/ref_driver.f90:
        # ref_generate - the product's counter-based generator restated there), so every page is first touched by the thread
        # that owns those columns: the pages are spread over the NUMA nodes of the host, as a parallel producer would leave them
        t = time.perf_counter(); A = ref.generate(n, sparsity, 1); tg = time.perf_counter() - t
        run = dict(n=n, generate_seconds=round(tg, 2))
        dt0 = None
        if n == n_list[0]:
            # the second solve is the one reported: the first carries MKL's one-time initialisation (threads, code paths)
            t = time.perf_counter(); lam, vec, it = ref.dense_solve(A, lowest, "DPR", 1000, tol); dt0 = time.perf_counter() - t
        t = time.perf_counter(); lam, vec, it = ref.dense_solve(A, lowest, "DPR", 1000, tol); dt = time.perf_counter() - t
        if dt0 is None:
            dt0 = dt
        run.update(iters=int(it), seconds=dt, seconds_first_call=dt0, evals=[float(x) for x in lam])
        # the primitive behind the reference's residual loop (lapack_matrix_vector -> DGEMV, src/lapack_wrapper.f90:330-364,
        # called m times per iteration at src/davidson.f90:163-170): one sweep of A through MKL, all threads
        x = np.ones(n); y = np.zeros(n)
        def sweep():
            mkl.cblas_dgemv(102, 111, n, n, ctypes.c_double(1.0), A.ctypes.data_as(dp), n, x.ctypes.data_as(dp), 1, ctypes.c_double(0.0),
                            y.ctypes.data_as(dp), 1)
        sweep()
        t = time.perf_counter()
        for _ in range(5):
            sweep()
        run["dgemv_sweep_GBps"] = 8.0 * n * n * 5 / (time.perf_counter() - t) / 1e9
        # ... and what explains that rate: the transposed form of the same sweep (one dot product per column: what threads
        # well; A is symmetric, so it is the same product - the reference calls 'N') and a plain OpenMP read of the matrix
        def sweep_t():
            mkl.cblas_dgemv(102, 112, n, n, ctypes.c_double(1.0), A.ctypes.data_as(dp), n, x.ctypes.data_as(dp), 1, ctypes.c_double(0.0),
                            y.ctypes.data_as(dp), 1)
        sweep_t()
        t = time.perf_counter()
        for _ in range(5):
            sweep_t()
        run["dgemv_transposed_GBps"] = 8.0 * n * n * 5 / (time.perf_counter() - t) / 1e9
        ref.stream_sum(A)
        t = time.perf_counter()
        for _ in range(5):
            ref.stream_sum(A)
        run["openmp_read_GBps"] = 8.0 * n * n * 5 / (time.perf_counter() - t) / 1e9
        yo = ref.gemv_omp(A, x)
        t = time.perf_counter()
        for _ in range(5):
            yo = ref.gemv_omp(A, x)
        run["openmp_gemv_GBps"] = 8.0 * n * n * 5 / (time.perf_counter() - t) / 1e9
        run["openmp_gemv_max_rel_diff_vs_mkl"] = float(np.abs(yo - y).max() / np.abs(y).max())
        out["runs"].append(run)
        print("CPU_BASELINE_PROGRESS", n, round(dt, 2), flush=True)
        del A
else:
    n = n_list[0]
    A = O.generate_diagonal_dominant(n, sparsity, seed=1)
    t = time.perf_counter(); lam, vec, it = O.generalized_eigensolver_dense(A, lowest, "DPR", 1000, tol); dt = time.perf_counter() - t
    out.update(kind="port", cores=os.cpu_count())
    out["runs"].append(dict(n=n, iters=int(it), seconds=dt, evals=[float(x) for x in lam]))
if {bench_free} and ref.available():
    # the reference's own benchmark program (src/benchmark_free.f90:80-111: N=1000, lowest=3, max_dim_sub=20, A = its cos row generator
    # applied through free_matmul under OpenMP, B = I, DPR) as a call on the same host cores
    t = time.perf_counter(); lam, vec, it = ref.free_solve_benchmark(1000, 3, 1000, 1e-8, 20); dt = time.perf_counter() - t
    out["benchmark_free"] = dict(n=1000, lowest=3, max_dim_sub=20, seconds=dt, iters=int(it), evals=[float(x) for x in lam])
print("CPU_BASELINE " + json.dumps(out))
"""


def cpu_share():
    """CPUs this process may really use: the affinity mask, cut down by the cgroup's CPU quota (the GPU boxes hand a job a share of
    the host - 16 CPUs per GPU - while os.cpu_count() and MKL still see all 256 hardware threads: 128 MKL threads on a 16-CPU
    quota is what made the round-3 baseline crawl at 30-40 GB/s)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f1, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                q, per = float(f1.read()), float(f2.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_baseline(n_list, lowest, tol, sparsity, bench_free=True):
    """The reference's own CPU+LAPACK path (oracle/_ref = the reference compiled with flang + MKL) on generate_diagonal_dominant
    inputs, timed in a child process that never touches the GPU, never imports torch (its libgomp breaks threaded MKL) and never
    loads the product libraries (they pin MKL to its sequential layer).  The matrix is generated inside that child, in parallel
    (first touch by the threads that later read it), bit-identical to the device generator."""
    try:
        code = CPU_CHILD.format(root=ROOT, lowest=lowest, tol=tol, sparsity=sparsity, n_list=list(n_list), bench_free=bool(bench_free))
        env = dict(os.environ)
        env["HIP_VISIBLE_DEVICES"] = ""
        share, quota = cpu_share()
        for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):              # as many threads as CPUs the job really has
            env[var] = str(share)
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1200, env=env)
        for line in res.stdout.splitlines():
            if line.startswith("CPU_BASELINE "):
                out = json.loads(line[len("CPU_BASELINE "):])
                out["cpu_share"] = share
                out["cgroup_cpu_quota"] = quota
                return out
        return {"error": (res.stderr or res.stdout)[-400:]}
    except Exception as exc:       # noqa: BLE001
        return {"error": repr(exc)}


def pmc_traffic(n, storage, kernel, rank_by_grid):
    """(HBM bytes per launch, provenance) of a roofline kernel from the rocprofv3 PMC summary under profiles/ - an OFFLINE figure
    (counters cannot be read from inside the run): collected with `rocprofv3 --pmc` on this command at the commit the
    file names (profiles/summarize.py).  `kernel` = name prefix; launches of one kernel are grouped by grid size
    (column groups per launch).  rank_by_grid = i: the i-th largest grid; None: the mean over the grid sizes (a solve
    launches each of them once: 32 and 64 columns).  None when no summary matches the workload."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_pmc_traffic_n{n}_{storage}.json")))
    path = found[-1] if found else os.path.join(ROOT, "profiles", f"r04_pmc_traffic_n{n}_{storage}.json")
    try:
        with open(path) as f:
            doc = json.load(f)
        rows = sorted((k for k in doc["kernels"] if k["kernel"].startswith(kernel)), key=lambda k: -k["grid_size"])
        prov = {"source": os.path.relpath(path, ROOT), "collected_at_commit": doc.get("commit"),
                "note": "offline rocprofv3 --pmc pass (2*FETCH_SIZE + WRITE_SIZE per MI355X_MICROARCH.md), not measured in this run"}
        if rank_by_grid is None:
            prov.update(kernel=rows[0]["kernel"], per_grid_size={str(r["grid_size"]): r["hbm_bytes_per_launch_corrected"] for r in rows})
            return round(sum(r["hbm_bytes_per_launch_corrected"] for r in rows) / len(rows)), prov
        row = rows[rank_by_grid]
        prov.update(kernel=row["kernel"], grid_size=row["grid_size"])
        return row["hbm_bytes_per_launch_corrected"], prov
    except Exception:      # noqa: BLE001
        return None, None


REHEARSAL = os.path.join(ROOT, "profiles", "experiments", "r06_ranks_rehearsal_n200000.jsonl")
LINK_GBPS = 70.0            # assumed payload rate of one xGMI link in one direction (153.6 GB/s bidirectional on the data sheet)


def rehearsal_inputs():
    """Per-rank times of a configs[2] solve on P = 1, 2, 4, 8 ranks MEASURED ON ONE GPU: P engines over the loopback transport of
    the test build taking turns on the device (profiles/tools/ranks_rehearsal.py, DAV_TEST_SERIALIZE=1) - each rank's HIP-event
    times are those of a rank that owns a GPU.  {P: {apply_local_ms, small_ms, collectives}} (median over the ranks, the rank that
    goes first after the transport's host-staged collective kept apart), or {} when the log is not there."""
    out = {}
    try:
        with open(REHEARSAL) as f:
            for line in f:
                line = line.strip()
                if not line.startswith("{"):
                    continue
                d = json.loads(line)
                if d.get("n") != 200000 or d.get("DAV_SYM_RUN9", "0") not in ("0", ""):
                    continue
                out[int(d["ranks"])] = {"apply_local_ms": d["apply_local_ms"]["median"], "sweep_kernel_ms": d["sweep_kernel_ms"]["median"],
                                        "small_ms": d["gram_ms"]["median"] + d["panel_ms"]["median"],
                                        "collectives": d["collectives_per_solve"]}
    except (OSError, ValueError, KeyError):
        return {}
    return out


def scaling_model(n, lowest, world, measured=None):
    """Predicted milliseconds per configs[2] solve on `world` GPUs (nothing in it has run on more than one GPU).  Inputs, in this
    order of preference: THIS run (one GPU: end-to-end sweeps, Gram / panel phases, host share of a solve), the one-GPU rehearsal
    of P ranks taking turns on the device (per-rank sweep + packing + reduction time at P = 2, 4, 8: REHEARSAL), and two assumptions
    that only a multi-GPU run can replace: LINK_GBPS per xGMI link and direction, 30 us per collective.  Collective volume per
    solve and rank: all-gather and reduce-scatter of the 32- and the 64-column block (8 N 96 bytes (P-1)/P each) + the
    reduce-scatter of W0 (8 N 32).  symmetric_ms prices them as RINGS bound by ONE link (the pessimistic end); symmetric_all_links_ms
    as exchanges that use the P-1 direct links of the xGMI mesh at once (what RCCL's multi-ring schedules aim at) - the first
    multi-GPU run's `comm` object (measured ms and bytes per collective) says which."""
    P = max(int(world), 1)
    reh = rehearsal_inputs()
    m = dict(measured or {})
    scale = (n / 200000.0) ** 2
    base = reh.get(1, {"apply_local_ms": 123.0, "small_ms": 1.06})
    box = (m["apply_ms"] / (base["apply_local_ms"] * scale)) if m.get("apply_ms") else 1.0      # this box against the rehearsal's
    if P in reh:
        local = reh[P]["apply_local_ms"] * scale * box
        small = reh[P]["small_ms"] * scale ** 0.5 * box
        ncoll = reh[P]["collectives"] if P > 1 else 0
        src = "rehearsal"
    else:
        local = base["apply_local_ms"] * scale * box / P / (1.0 - 0.014 * (P - 1))             # fitted to the rehearsal: 0.90 at P = 8
        small = (0.18 + 0.88 / P) * scale ** 0.5 * box
        ncoll = 11 if P > 1 else 0
        src = "fit"
    host = m.get("host_ms", 1.4)
    lat = 0.03 * ncoll
    blk = 8.0 * n * (2 * lowest + 4 * lowest)          # the two expansion blocks of a solve: 2 lowest and 4 lowest columns
    w0 = 8.0 * n * 2 * lowest
    frac = (P - 1) / P
    ring_blk = blk * frac / (LINK_GBPS * 1e9) * 1e3
    ring_w0 = w0 * frac / (LINK_GBPS * 1e9) * 1e3
    links_blk = 0.0 if P == 1 else blk / P / (LINK_GBPS * 1e9) * 1e3
    links_w0 = 0.0 if P == 1 else w0 / P / (LINK_GBPS * 1e9) * 1e3
    sym = local + small + host + lat + 2 * ring_blk + ring_w0
    sym_links = local + small + host + lat + 2 * links_blk + links_w0
    # opt-in (DAV_SYM_OVERLAP=1): the collectives of the 64-column block (two thirds of the block volume) run under its sweeps
    sym_ovl = local + small + host + lat + 2 * ring_blk * (1.0 - 2.0 / 3.0 * 2.0 / 3.0) + ring_w0
    # north_star's literal partition (row slabs, all-gather only, 8 N^2 / P bytes per sweep) at the K1 rates of this run's N=20000 leg
    f32 = m.get("k1_frac_k32", 0.624)
    t64 = m.get("k1_tflops_k64", 57.8)
    full32 = 8.0 * n * n / P / (f32 * 8.0e12) * 1e3
    full64 = max(8.0 * n * n / P / (0.66 * 8.0e12), 2.0 * n * n * 64 / P / (t64 * 1e12)) * 1e3
    full = full32 + full64 + small + host + lat + ring_blk
    one = m.get("ms_per_solve") or (base["apply_local_ms"] * scale * box + base["small_ms"] * scale ** 0.5 * box + host)
    r2 = lambda x: round(x, 2)      # noqa: E731
    return {"symmetric_ms": r2(sym), "symmetric_all_links_ms": r2(sym_links), "symmetric_overlapped_ms": r2(sym_ovl), "full_ms": r2(full),
            "speedup_symmetric": r2(one / sym), "speedup_symmetric_all_links": r2(one / sym_links),
            "speedup_symmetric_overlapped": r2(one / sym_ovl), "one_gpu_ms": r2(one),
            "inputs": {"per_rank_sweeps_end_to_end_ms": r2(local), "per_rank_gram_panel_ms": r2(small), "host_ms": r2(host),
                       "collectives_per_solve": ncoll, "latency_ms": r2(lat), "ring_ms_per_block_pass": r2(ring_blk),
                       "all_links_ms_per_block_pass": r2(links_blk), "link_GBps_assumed": LINK_GBPS, "per_rank_times_from": src,
                       "rehearsal_log": os.path.relpath(REHEARSAL, ROOT) if reh else None, "box_speed_vs_rehearsal": round(box, 3)},
            "note": "model, not a measurement: symmetric = dealt-out tiles, collectives in program order as one-link rings; all_links = the "
                    "same volume over the P-1 direct links at once; overlapped = opt-in second stream; full = row slabs"}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run ... bench.py ...` as a CHILD process,
    let rank 0's JSON line through and return the child's exit code.  This (parent) process never imports torch and never
    touches the GPU - a process that has initialised the GPU must not be replaced by another program, and it is not."""
    import socket
    preload = " ".join(os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD"))
    if "rocprof" in preload.lower() or any(k.startswith("ROCPROF") for k in os.environ):
        raise SystemExit("bench.py --gpus N will not start its own launcher under a profiler: the profiler's preloaded library has "
                         "already initialised the GPU in this process, and launching torch.distributed.run from here is the "
                         "launcher hop that is not allowed on this pool.  Profile one rank's program directly: "
                         "rocprofv3 ... -- python3 bench.py --gpus 1 ..., or put the per-rank program after `--` under your own launcher.")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    fwd = ["--order" if a == "--n" else ("--order=" + a[4:] if a.startswith("--n=") else a) for a in sys.argv[1:]]   # the launcher's parser rejects --n
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + fwd
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    # the pool's driver only supports dmabuf IPC (RCCL's peer-to-peer buffers): in place before the HIP runtime loads, whoever launched us
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: launch one rank per GPU "
                         "(python bench.py --gpus N starts torch.distributed.run by itself)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a collective that never completes (a peer died, a link hangs) ends its rank with a message after this many seconds
        # (the engine's watchdog; its own default is 600): every collective of this bench completes in milliseconds, the longest
        # legitimate wait is for the slowest rank to finish generating its tiles (seconds)
        os.environ.setdefault("DAVIDSON_COLLECTIVE_TIMEOUT", "180")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if world > 1 and not args.all_legs:
        args.headline_only = True
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
    if transport == "rccl-one-gpu":
        # Rehearsal of the REAL RCCL path on a one-GPU box (tests/test_rccl_one_gpu.py): every rank uses GPU 0 and poses as a host of
        # its own (NCCL_HOSTID), so RCCL builds a genuine multi-rank communicator - its all-gather / reduce-scatter / all-reduce /
        # send / receive kernels and proxy threads, over the loopback socket transport instead of xGMI.  Everything above the wire
        # is what a multi-GPU run executes; the times mean nothing.
        os.environ["NCCL_HOSTID"] = f"davidson-rehearsal-host-{rank}"
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
    device = local_rank if transport == "rccl" else 0
    torch.cuda.set_device(device)

    import fortran_davidson_amd as fd
    engines_made = [0]

    def make_engine(n, lowest, max_dim=None, storage="full", gev=False):
        eng = fd.DavidsonEngine(n, lowest, max_dim, gev=gev, device=device, rank=rank, nranks=world, storage=storage)
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

    def timed_solves(eng, method, reps, tol):
        """reps solves bracketed by barrier + synchronize on both sides; max over ranks of the wall time"""
        eng.c.synchronize(); barrier()
        t0 = time.perf_counter()
        its = 0
        lam = None
        for _ in range(reps):
            lam, _, it = eng.solve(method, 1000, tol, want_vectors=False)
            its += it
        eng.c.synchronize(); barrier()
        return max_over_ranks(time.perf_counter() - t0), its, lam

    def apply_rooflines(eng, ks, reps):
        out = {}
        for k in ks:
            ms, kms, nbytes, flops = eng.c.bench_apply2(k, reps)
            out[f"k{k}"] = {"ms_end_to_end": round(ms, 4), "ms_kernel_only": round(kms, 4),
                            "GBps_end_to_end": round(nbytes / (ms * 1e-3) / 1e9, 1),
                            "GBps_kernel_only": round(nbytes / (kms * 1e-3) / 1e9, 1),
                            "frac_of_8TBps_end_to_end": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                            "frac_of_8TBps_kernel_only": round(nbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                            "frac_of_measured_stream_end_to_end": round(nbytes / (ms * 1e-3) / 1e9 / stream_gbps, 4) if stream_gbps > 0 else None,
                            "TFLOPs_kernel_only": round(flops / (kms * 1e-3) / 1e12, 2),
                            "algorithmic_bytes": nbytes}
        return out

    def measured_stream():
        """Achievable HBM rate on THIS box: device copy a = b and triad a = b + s c of 2 GiB arrays by the engine's own streaming
        kernels (16 B per lane; dav_bench_stream, HIP events), read + written bytes per second - what the 8 TB/s of the data
        sheet amount to in practice."""
        try:
            with fd.CEngine(n=1024, max_cols=16, device=device) as e0:
                cp, tr, rd = e0.bench_stream3(0, 5)
            return {"copy_GBps": round(cp, 1), "triad_GBps": round(tr, 1), "read_GBps": round(rd, 1), "bytes_per_array": 8 * (1 << 28),
                    "note": "dav_bench_stream: plain streaming kernels (16 B per lane, 8 accesses in flight, contiguous 32 KiB pieces, 2048 workgroups) on 2 GiB float64 arrays, "
                            "read + written bytes, HIP events, this run; read_GBps: the same pattern reading two of the arrays and writing one partial sum "
                            "per workgroup - the practical roof of a sweep that reads 8*S bytes and writes 8*N*k"}
        except Exception as exc:       # noqa: BLE001
            return {"error": repr(exc)[:200]}

    hbm_measured = measured_stream() if rank == 0 or world > 1 else None
    stream_gbps = max(hbm_measured.get("copy_GBps", 0.0), hbm_measured.get("triad_GBps", 0.0)) if hbm_measured else 0.0
    read_gbps = hbm_measured.get("read_GBps", 0.0) if hbm_measured else 0.0

    # auto: the storage the model of DESIGN section 6 predicts to be the faster one for this rank count (symmetric tiles for every
    # P up to 8 at the timed size; the full row slabs of north_star's partition are `--storage full`)
    if args.storage != "auto":
        storage = args.storage
    else:
        # symmetric tiles unless the model puts the row slabs clearly ahead (5 %): with one-link rings the two are within 2 % of each
        # other at 8 GPUs, with faster exchanges the symmetric storage leads - and it needs half the memory
        mdl = scaling_model(args.n, args.lowest, world)
        storage = "symmetric" if world == 1 or mdl["full_ms"] >= 0.95 * mdl["symmetric_ms"] else "full"
        if storage == "full" and 8.0 * args.n * args.n / world > 200e9:
            storage = "symmetric"          # a full row slab of this size does not fit the GPU
    storage_words = {"symmetric": "symmetric-tiled (lower block triangle, N(N+1)/2 entries" +
                                  (f", dealt out over the {world} ranks by groups of 4 block rows" if world > 1 else "") + ")",
                     "full": "full row slabs"}[storage]

    # ---- the timed workload: configs[2] ---------------------------------------------------------------
    n, lowest, max_dim = args.n, args.lowest, args.max_dim
    eng = make_engine(n, lowest, max_dim, storage)
    t_gen = time.perf_counter()
    eng.generate_diagonal_dominant(1, args.sparsity, seed=1)       # resident in HBM before timing
    eng.c.synchronize()
    t_gen = time.perf_counter() - t_gen
    if world > 1:
        # several ranks: one untimed priming solve in front of the W warm-up solves - the engine's first wide block over a real
        # communicator runs the trial of the collective paths (dav_comm_path: six extra block sweeps and RCCL's lazy connection
        # set-up), which must not land in the timed solves even with --warmup 0
        eng.solve("DPR", 1000, args.tol, want_vectors=False)
    for _ in range(args.warmup):
        eng.solve("DPR", 1000, args.tol, want_vectors=False)
    eng.c.synchronize()
    eng.c.reset_stats()
    elapsed, total_iters, lam = timed_solves(eng, "DPR", args.steps, args.tol)
    st = eng.c.stats()
    value = total_iters / elapsed

    # roofline of the dominant kernel over the timed region (HIP events on the engine's stream, inside the solves)
    launches = max(int(st.apply_launches), 1)
    kms = st.apply_kernel_ms / launches
    tflops = st.apply_flops / (st.apply_kernel_ms * 1e-3) / 1e12 if st.apply_kernel_ms > 0 else 0.0
    cols_per_launch = st.apply_cols / launches
    kernel_name = "matvec_symw_kernel<2> (symmetric tiles, 32 / 64 columns per launch)" if storage == "symmetric" else "matvec_dense_kernel<NT> (row slab)"
    hbm_in_solve = st.apply_bytes / (st.apply_ms * 1e-3) / 1e9 if st.apply_ms > 0 else 0.0
    mfma_bound = cols_per_launch > 16
    tr_solve = pmc_traffic(n, storage, "matvec_symw_kernel<2" if storage == "symmetric" else "matvec_dense_kernel", None) if world == 1 else (None, None)
    if tr_solve[0] is None and world == 1 and storage == "symmetric":       # small orders: the k <= 8 / one-block-row kernels carry the solve
        tr_solve = pmc_traffic(n, storage, "matvec_sym", None)
    tr_k8 = pmc_traffic(n, storage, "matvec_sym9_kernel<4", 0) if (world == 1 and storage == "symmetric") else (None, None)
    roofline = {"bound": "mfma" if mfma_bound else "hbm", "kernel": kernel_name,
                "achieved": round(tflops, 2) if mfma_bound else round(st.apply_bytes / (st.apply_kernel_ms * 1e-3) / 1e9, 1),
                "peak": FP64_MFMA_PEAK_TFLOPS if mfma_bound else HBM_PEAK_GBPS, "unit": "TFLOP/s" if mfma_bound else "GB/s",
                "frac": round(tflops / FP64_MFMA_PEAK_TFLOPS, 4) if mfma_bound
                        else round(st.apply_bytes / (st.apply_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "traffic": tr_solve[0],
                # where `traffic` comes from, as a short string among the first keys (the driver's record keeps those): an OFFLINE figure
                "traffic_source": (f"{tr_solve[1]['source']} (offline rocprofv3 --pmc passes at commit {tr_solve[1].get('collected_at_commit')})"
                                   if tr_solve[1] else None),
                "traffic_provenance": tr_solve[1],
                "launches": launches, "avg_launch_ms": round(kms, 4), "columns_per_launch": round(cols_per_launch, 1),
                "flops_per_launch": round(st.apply_flops / launches, 0),
                "algorithmic_bytes_per_launch": round(st.apply_bytes / max(st.applies, 1) , 0),
                "apply_ms_end_to_end_per_solve": round(st.apply_ms / args.steps, 3),
                "apply_ms_kernel_only_per_solve": round(st.apply_kernel_ms / args.steps, 3),
                "in_solve_GBps_end_to_end": round(hbm_in_solve, 1),
                "note": "per rank, HIP events in the timed solves; 32 / 64 columns per launch: MFMA roof; hbm_*: A*V at N x 8"}

    # north-star microbenchmark: A*V at k = 8 (16, 32) on the same resident matrix, end to end and kernel only
    apply_k = apply_rooflines(eng, (8, 16, 32), 10 if n >= 100000 else 20)
    a8, a16 = apply_k["k8"], apply_k["k16"]
    tr_k16 = pmc_traffic(n, storage, "matvec_symw_kernel<1", 0) if (world == 1 and storage == "symmetric") else (None, None)
    # INSIDE `roofline` (flat scalar keys: the driver's record keeps the scalars of this object), so that the record alone lets a
    # reader recompute both the MFMA fraction of the in-solve launches and the HBM fraction of BASELINE's metric (N x k stated);
    # bytes = 8*S + 16*N*k per rank, S = N(N+1)/2 / n_gpus (symmetric-tiled) or nloc*N (row slab); end to end = pack_xt
    # (+ all-gather) + sweep kernel + fixed-order reduction of the partial sums (+ reduce-scatter); HIP events on the engine's stream
    roofline.update({
        "hbm_N": n, "hbm_k": 8, "hbm_peak_GBps": HBM_PEAK_GBPS,
        "hbm_algorithmic_bytes": a8["algorithmic_bytes"],
        "hbm_ms_end_to_end": a8["ms_end_to_end"], "hbm_ms_kernel_only": a8["ms_kernel_only"],
        "hbm_GBps_end_to_end": a8["GBps_end_to_end"], "hbm_GBps_kernel_only": a8["GBps_kernel_only"],
        "hbm_frac": a8["frac_of_8TBps_end_to_end"], "hbm_frac_kernel_only": a8["frac_of_8TBps_kernel_only"],
        "hbm_measured_stream_GBps": stream_gbps or None, "hbm_frac_of_measured_stream": a8["frac_of_measured_stream_end_to_end"],
        "hbm_measured_read_GBps": read_gbps or None,
        "hbm_frac_of_measured_read": round(a8["GBps_end_to_end"] / read_gbps, 4) if read_gbps > 0 else None,
        "hbm_frac_of_measured_read_kernel_only": round(a8["GBps_kernel_only"] / read_gbps, 4) if read_gbps > 0 else None,
        "hbm_traffic": tr_k8[0],
        "hbm_k16_ms_end_to_end": a16["ms_end_to_end"], "hbm_k16_ms_kernel_only": a16["ms_kernel_only"],
        "hbm_k16_algorithmic_bytes": a16["algorithmic_bytes"], "hbm_k16_GBps_end_to_end": a16["GBps_end_to_end"],
        "hbm_k16_frac": a16["frac_of_8TBps_end_to_end"], "hbm_k16_traffic": tr_k16[0],
        # what of a solve is not the roofline kernel: the rest of the applies (pack + fixed-order reduction) and the rest of the solve
        "ms_per_solve": round(elapsed / args.steps * 1e3, 3),
        "non_kernel_ms_per_solve": round(elapsed / args.steps * 1e3 - st.apply_kernel_ms / args.steps, 3),
        "apply_non_kernel_ms_per_solve": round((st.apply_ms - st.apply_kernel_ms) / args.steps, 3)})

    extras = {}
    # one solve at timing level 2: Gram / panel phases (and, several ranks, every collective) by HIP events - inputs of the scaling model
    eng.c.set_timing(2)
    eng.c.synchronize(); eng.c.reset_stats()
    t_l2 = time.perf_counter()
    eng.solve("DPR", 1000, args.tol, want_vectors=False)
    eng.c.synchronize()
    t_l2 = (time.perf_counter() - t_l2) * 1e3
    sc = eng.c.stats()
    eng.c.set_timing(1)
    measured = None
    if world == 1:
        measured = {"apply_ms": st.apply_ms / args.steps, "ms_per_solve": elapsed / args.steps * 1e3,
                    "host_ms": max(elapsed / args.steps * 1e3 - st.apply_ms / args.steps - sc.gram_ms - sc.panel_ms, 0.0)}
        roofline["gram_ms_per_solve"] = round(sc.gram_ms, 3)
        roofline["panel_ms_per_solve"] = round(sc.panel_ms, 3)
        roofline["host_and_latency_ms_per_solve"] = round(measured["host_ms"], 3)
        extras["scaling_model"] = {f"P{p_}": scaling_model(n, lowest, p_, measured) for p_ in (2, 4, 8)}
        roofline.update({f"model_P8_{k_}": extras["scaling_model"]["P8"][k_] for k_ in ("symmetric_ms", "symmetric_all_links_ms", "speedup_symmetric", "speedup_symmetric_all_links")})
    if world > 1:
        # collectives of one solve (the level-2 solve above: an event pair around every collective or group); sweep times over the ranks
        def over_ranks(x, op):
            t = torch.tensor([x], dtype=torch.float64)
            dist.all_reduce(t, op=op)
            return float(t.item())
        local_ms = sc.apply_ms - sc.apply_comm_ms
        # which way the collectives of wide blocks go: decided by the engine at its first wide block (dav_comm_path; the warm-up solves)
        cpath = eng.c.comm_path() if transport.startswith("rccl") else {"selected": "program order", "trial_ran": False, "columns": 0,
                                                                        "trial_ms_max_over_ranks": {}, "validated": {}}
        sweep_min, sweep_max = over_ranks(sc.apply_kernel_ms, dist.ReduceOp.MIN), over_ranks(sc.apply_kernel_ms, dist.ReduceOp.MAX)
        local_min, local_max = over_ranks(local_ms, dist.ReduceOp.MIN), over_ranks(local_ms, dist.ReduceOp.MAX)
        extras["comm"] = {
            "transport": transport, "ranks_reported_by_rccl": int(sc.comm_ranks), "world_size": world, "storage": storage,
            "collectives_overlapped_with_sweeps": bool(sc.comm_overlap),
            "path_selected": cpath["selected"], "path_trial_ran": cpath["trial_ran"], "path_trial_columns": cpath["columns"],
            "path_trial_ms_max_over_ranks": cpath["trial_ms_max_over_ranks"], "path_validated": cpath["validated"],
            "collectives_per_solve": int(sc.collectives),
            "allgather_ms_per_solve": round(sc.allgather_ms, 3), "allgather_MB_per_solve": round(sc.allgather_bytes / 1e6, 2),
            "reduce_scatter_ms_per_solve": round(sc.reduce_scatter_ms, 3), "reduce_scatter_MB_per_solve": round(sc.reduce_scatter_bytes / 1e6, 2),
            "allreduce_ms_per_solve": round(sc.allreduce_ms, 3), "allreduce_MB_per_solve": round(sc.allreduce_bytes / 1e6, 3),
            "sweep_kernel_ms_per_solve": round(sc.apply_kernel_ms, 3), "apply_ms_end_to_end_per_solve": round(sc.apply_ms, 3),
            "sweep_kernel_ms_min_over_ranks": round(sweep_min, 3), "sweep_kernel_ms_max_over_ranks": round(sweep_max, 3),
            "apply_local_ms_min_over_ranks": round(local_min, 3), "apply_local_ms_max_over_ranks": round(local_max, 3),
            "allgather_busbw_GBps": round(sc.allgather_bytes * (world - 1) / world / (sc.allgather_ms * 1e-3) / 1e9, 1) if sc.allgather_ms > 0 else None,
            "reduce_scatter_busbw_GBps": round(sc.reduce_scatter_bytes * (world - 1) / world / (sc.reduce_scatter_ms * 1e-3) / 1e9, 1) if sc.reduce_scatter_ms > 0 else None,
            "gram_ms_per_solve": round(sc.gram_ms, 3), "panel_ms_per_solve": round(sc.panel_ms, 3),
            "model_ms_per_solve": scaling_model(n, lowest, world),
            "note": "rank 0's view of one solve; payload per rank (all-gather: bytes received, reduce-scatter: bytes contributed); with "
                    "overlapped collectives their time runs under the sweeps and is not additive to apply_ms"}
        # the scalars also inside `roofline` (the driver's record keeps the scalars of that object)
        roofline.update({"comm_" + k_: v_ for k_, v_ in extras["comm"].items() if isinstance(v_, (int, float, bool, str)) and k_ != "note"})
        roofline.update({"comm_path_trial_ms_" + k_.replace(" ", "_"): v_ for k_, v_ in cpath["trial_ms_max_over_ranks"].items()})
        roofline.update({"comm_model_" + k_: v_ for k_, v_ in extras["comm"]["model_ms_per_solve"].items() if isinstance(v_, (int, float))})
    if not args.headline_only:
        # opt-in correction policies (not the reference's, so never part of `value`): the same problem, same engine
        extras["opt_in_policy"] = {"reference_policy_all": {"ms_per_solve": round(elapsed / args.steps * 1e3, 3), "iters_per_solve": total_iters // args.steps}}
        for pol_ in ("unconverged", "locking"):
            try:
                eng.set_correction_policy(pol_)
                eng.solve("DPR", 1000, args.tol, want_vectors=False)
                eng.c.synchronize(); eng.c.reset_stats()
                dt_pol, it_pol, lam_pol = timed_solves(eng, "DPR", 2, args.tol)
                sp_ = eng.c.stats()
                extras["opt_in_policy"][pol_] = {"ms_per_solve": round(dt_pol / 2 * 1e3, 3), "iters_per_solve": it_pol // 2,
                                                 "columns_swept_per_solve": int(sp_.apply_cols) // 2,
                                                 "max_abs_eigenvalue_diff_vs_reference_policy": float(np.abs(lam_pol - lam).max())}
            except Exception as exc:       # noqa: BLE001
                extras["opt_in_policy"][pol_] = {"error": repr(exc)[:200]}
        eng.set_correction_policy("all")
        extras["opt_in_policy"]["note"] = ("engine_set_correction_policy / DAVIDSON_CORRECTION_POLICY: unconverged = corrections only for wanted pairs above the "
                                           "tolerance; locking = converged wanted pairs leave the active basis, which stays orthogonal to them (standard problems)")
        extras["opt_in_policy"]["max_abs_eigenvalue_diff_vs_reference_policy"] = max(
            (v.get("max_abs_eigenvalue_diff_vs_reference_policy", 0.0) for v in extras["opt_in_policy"].values() if isinstance(v, dict)), default=0.0)
    eng.close()

    # Order of the legs: the small (launch-bound) problems right behind the timed workload, the other 100+ GB problems after
    # them - measured: the N=20000 symmetric-storage leg runs at 1.64 ms per solve in a fresh process and behind the headline,
    # but at 2.4 ms behind the configs[3] / configs[4] legs (their 120-133 GB allocations leave the allocator a fragmented pool)
    if not args.headline_only:
        # ---- configs[1]: N=20000 dense, lowest=8, DPR (full storage; latency-bound at this size) ---------------
        if args.small_n > 0:
            try:
                sn = args.small_n
                s = make_engine(sn, 8, None, "full")
                s.generate_diagonal_dominant(1, args.sparsity, seed=1)
                for _ in range(5):
                    s.solve("DPR", 1000, args.tol, want_vectors=False)
                s.c.synchronize(); s.c.reset_stats()
                dt_s, it_s, lam_s = timed_solves(s, "DPR", 50, args.tol)
                ss = s.c.stats()
                small = {"workload": f"N={sn} dense fp64 full storage, lowest=8, DPR, tol={args.tol}, {world} GPU(s)",
                         "iterations_per_s": round(it_s / dt_s, 2), "ms_per_solve": round(dt_s / 50 * 1e3, 4), "iters_per_solve": it_s // 50,
                         "apply_GBps_end_to_end": round(ss.apply_bytes / (ss.apply_ms * 1e-3) / 1e9, 1),
                         "apply_GBps_kernel_only": round(ss.apply_bytes / (ss.apply_kernel_ms * 1e-3) / 1e9, 1),
                         "eigenvalues": [float(x) for x in lam_s[:3]]}
                s.c.set_timing(2); s.c.reset_stats()
                for _ in range(5):
                    s.solve("DPR", 1000, args.tol, want_vectors=False)
                s.c.synchronize()
                sp = s.c.stats()
                small["phase_ms_per_solve"] = {"apply_ms": round(sp.apply_ms / 5, 4), "gram_ms": round(sp.gram_ms / 5, 4),
                                               "panel_ms": round(sp.panel_ms / 5, 4), "comm_ms": round(sp.comm_ms / 5, 4)}
                s.c.set_timing(1)
                small["apply"] = apply_rooflines(s, (8, 16, 32, 64), 20)
                s.close()
                # the same problem with only the lower block triangle resident (engine_set_storage(eng, "symmetric")): half the bytes per sweep
                y = make_engine(sn, 8, None, "symmetric")
                y.generate_diagonal_dominant(1, args.sparsity, seed=1)
                for _ in range(5):
                    y.solve("DPR", 1000, args.tol, want_vectors=False)
                dt_y, it_y, lam_y = timed_solves(y, "DPR", 50, args.tol)
                small["symmetric_storage"] = {"ms_per_solve": round(dt_y / 50 * 1e3, 4), "iterations_per_s": round(it_y / dt_y, 2),
                                              "iters_per_solve": it_y // 50,
                                              "max_abs_eigenvalue_diff_vs_full_storage": float(np.abs(lam_y - lam_s).max())}
                y.close()
                extras["small"] = small
            except Exception as exc:       # noqa: BLE001
                extras["small"] = {"error": repr(exc)[:300]}

    if not args.headline_only and args.restart_sparsity > 0:
        # ---- configs[2], restart-forcing variant: a denser coupling so that the basis passes max_dim_sub before the pairs
        # converge and the solve goes through collapse restarts (src/davidson.f90:215-220) at full size ----------------------
        try:
            r = make_engine(n, lowest, max_dim, storage)
            r.generate_diagonal_dominant(1, args.restart_sparsity, seed=3)
            r.solve("DPR", 1000, args.tol, want_vectors=False)
            r.c.synchronize(); r.c.reset_stats()
            dt_r, it_r, lam_r = timed_solves(r, "DPR", 1, args.tol)
            sr = r.c.stats()
            extras["configs2_restart"] = {
                "workload": f"N={n} dense fp64, lowest={lowest}, DPR, max_dim_sub={max_dim}, tol={args.tol}, "
                            f"generate_diagonal_dominant(N,{args.restart_sparsity}) seed 3, storage: {storage}, {world} GPU(s)",
                "iters": it_r, "restarts": int(sr.restarts), "seconds": round(dt_r, 4), "iterations_per_s": round(it_r / dt_r, 3),
                "sweeps_of_A": int(sr.applies), "columns_swept": int(sr.apply_cols),
                "columns_swept_per_restart_cycle": round(sr.apply_cols / (int(sr.restarts) + 1), 1),
                "ms_in_sweeps_end_to_end": round(sr.apply_ms, 2), "ms_in_panel_products_incl_restart": round(sr.panel_ms, 2),
                "ms_in_gram": round(sr.gram_ms, 2),
                "TFLOPs_in_sweep_kernels": round(sr.apply_flops / (sr.apply_kernel_ms * 1e-3) / 1e12, 2) if sr.apply_kernel_ms > 0 else None,
                "eigenvalues": [float(x) for x in lam_r[:3]],
                "note": "a restart contracts V, W = A*V (and B*V) with the kept Ritz vectors: no sweep of A follows it "
                        "(2 lowest + 4 lowest columns per cycle at restart width 80 = 96 at lowest=16; the reference re-applies A to the whole basis every iteration)"}
            r.close()
        except Exception as exc:       # noqa: BLE001
            extras["configs2_restart"] = {"error": repr(exc)[:300]}

    if not args.headline_only:
        # ---- configs[3]: generalized (A, B), GJD correction, lowest=8 ---------------------------------------
        # A stored (symmetric tiles on one GPU), B = the same generator with unit diagonal evaluated on the fly:
        # two stored 160 GB matrices do not fit one GPU
        gn = n if args.gjd_n < 0 else args.gjd_n
        if gn > 0:
            try:
                # (the resident part of B follows the free memory of the moment unless bounded: 80 % of its tiles - what fits next to
                # A's 160.5 GB at N=200000 - makes the split, and with it the order of the sums, the same on every box)
                mine = "DAV_B_RESIDENT" not in os.environ
                if mine:
                    os.environ["DAV_B_RESIDENT"] = "80"
                try:
                    g = make_engine(gn, 8, 80, storage, gev=True)          # the knobs are read at dav_create
                finally:
                    if mine:
                        del os.environ["DAV_B_RESIDENT"]
                g.generate_diagonal_dominant(1, args.sparsity, seed=1)
                g.set_hashed_operator(2, args.sparsity, 1.0, seed=2)
                g.solve("GJD", 1000, args.tol, want_vectors=False)        # warm-up (lazy workspace)
                g.c.set_timing(2)
                g.c.synchronize(); g.c.reset_stats()
                dt_g, it_g, lam_g = timed_solves(g, "GJD", 1, args.tol)
                sg = g.c.stats()
                b_resident = g.c.resident_fraction(1)
                dev_other = sg.gram_ms + sg.panel_ms + sg.comm_ms       # panel_ms includes the sweeps of B
                extras["configs3_gjd"] = {
                    "workload": f"N={gn} generalized (A stored {storage}, B = hashed unit-diagonal operator: the tiles of its longest block rows "
                                f"resident next to A - {b_resident:.3f} of them - the others generated in the sweep), "
                                f"GJD, lowest=8, max_dim_sub=80, tol={args.tol}, {world} GPU(s)",
                    "B_resident_fraction_of_tiles": round(b_resident, 4),
                    "iters": it_g, "seconds": round(dt_g, 4), "iterations_per_s": round(it_g / dt_g, 4),
                    "sweeps_of_A": int(sg.applies), "columns_swept": int(sg.apply_cols),
                    "ms_per_sweep_of_A_end_to_end": round(sg.apply_ms / max(sg.applies, 1), 3),
                    "ms_in_sweeps_of_A": round(sg.apply_ms, 2),
                    "ms_in_other_device_phases_incl_sweeps_of_B": round(dev_other, 2),
                    "ms_latency_remainder": round(dt_g * 1e3 - sg.apply_ms - dev_other, 2),
                    "eigenvalues": [float(x) for x in lam_g[:3]],
                    "note": "sweeps_of_A x ms_per_sweep + the same number of generated-B sweeps is the device floor; the "
                            "remainder is host latency of the inner MINRES (dot-product round trips, small uploads)"}
                # rooflines of the two sweeps (per 16-column group; B is generated once per group): A against HBM, B against the
                # integer-VALU bound of its generator (same model as configs4_free)
                peak_evals = 1024 * 2.4e9 * 64 / GEN_CYCLES_PER_WAVE_EVALUATION
                extras["configs3_gjd"]["roofline"] = {
                    "A_sweeps": {"bound": "hbm (16 columns) / mfma (32, 64)", "ms_per_sweep_end_to_end": round(sg.apply_ms / max(sg.applies, 1), 3),
                                 "GBps_end_to_end": round(sg.apply_bytes / (sg.apply_ms * 1e-3) / 1e9, 1) if sg.apply_ms > 0 else None,
                                 "frac_of_8TBps": round(sg.apply_bytes / (sg.apply_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if sg.apply_ms > 0 else None,
                                 "TFLOPs_kernel_only": round(sg.apply_flops / (sg.apply_kernel_ms * 1e-3) / 1e12, 2) if sg.apply_kernel_ms > 0 else None},
                    "B_sweeps": {
                        "resident_fraction_of_tiles": round(b_resident, 4),
                        # the sweep kernels of B by what they read (HIP events around every launch, timing level 2)
                        "stored_tiles": {"bound": "hbm (8 / 16 columns) / mfma (32 / 64)", "launches": int(sg.b_stored_launches), "ms": round(sg.b_stored_kernel_ms, 2),
                                         "GBps": round(sg.b_stored_bytes / (sg.b_stored_kernel_ms * 1e-3) / 1e9, 1) if sg.b_stored_kernel_ms > 0 else None,
                                         "frac_of_8TBps": round(sg.b_stored_bytes / (sg.b_stored_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if sg.b_stored_kernel_ms > 0 else None,
                                         "TFLOPs": round(sg.b_stored_flops / (sg.b_stored_kernel_ms * 1e-3) / 1e12, 2) if sg.b_stored_kernel_ms > 0 else None},
                        "generated_block_rows": {"bound": "valu-int + mfma on one issue port", "launches": int(sg.b_generated_launches), "ms": round(sg.b_generated_kernel_ms, 2),
                                                 "evaluations_per_s": round(sg.b_generated_entries / (sg.b_generated_kernel_ms * 1e-3), 0) if sg.b_generated_kernel_ms > 0 else None,
                                                 "peak_16_columns": round(peak_evals, 0),
                                                 "frac_of_16_column_issue_bound": round(sg.b_generated_entries / (sg.b_generated_kernel_ms * 1e-3) / peak_evals, 4) if sg.b_generated_kernel_ms > 0 else None},
                        "note": "per-kernel: stored_tiles = launches of the stored-tile kernels on B's resident block rows (bytes = 8 x stored entries + 16 N k), "
                                "generated_block_rows = launches that evaluate the hash (entries once per 16 columns; once per 32 in the wide kernel's generating variant)"},
                    "device_floor_seconds": round((sg.apply_ms + dev_other) * 1e-3, 3),
                    "note": "fp64 MFMA and the generator's integer VALU work share the SIMD's issue port (docs/history/DESIGN_rounds_1_to_5.md section 0, item 3; DESIGN section 10): a fused A + B pass "
                            "could hide generation only under the HBM stalls of the 16-column A sweeps"}
                # opt-in correction policies on the generalized problem ("locking" covers A x = lambda B x since round 6: the
                # guard vectors B x of the locked pairs keep the search space B-orthogonal to them) - never part of the headline
                pol3 = {"reference_policy_all": {"seconds": round(dt_g, 4), "iters": it_g, "sweeps_of_A": int(sg.applies)}}
                for pol_ in ("unconverged", "locking"):
                    try:
                        g.set_correction_policy(pol_)
                        g.c.synchronize(); g.c.reset_stats()
                        dt_p3, it_p3, lam_p3 = timed_solves(g, "GJD", 1, args.tol)
                        sp3 = g.c.stats()
                        pol3[pol_] = {"seconds": round(dt_p3, 4), "iters": it_p3, "sweeps_of_A": int(sp3.applies), "columns_swept": int(sp3.apply_cols),
                                      "max_abs_eigenvalue_diff_vs_reference_policy": float(np.abs(lam_p3 - lam_g).max())}
                    except Exception as exc:   # noqa: BLE001
                        pol3[pol_] = {"error": repr(exc)[:300]}
                g.set_correction_policy("all")
                extras["configs3_gjd"]["opt_in_policy"] = pol3
                # opt-in mixed-precision correction path: the inner sweeps of A read an fp32 copy of its tiles
                try:
                    g.set_inner_precision(32)
                    g.solve("GJD", 1000, args.tol, want_vectors=False)    # warm-up: builds the fp32 copy (80 GB at N=200000)
                    g.c.synchronize(); g.c.reset_stats()
                    dt_m, it_m, lam_m = timed_solves(g, "GJD", 1, args.tol)
                    sm_ = g.c.stats()
                    extras["configs3_gjd"]["inner_fp32"] = {
                        "seconds": round(dt_m, 4), "iters": it_m, "sweeps_of_A": int(sm_.applies),
                        "ms_per_sweep_of_A_end_to_end": round(sm_.apply_ms / max(sm_.applies, 1), 3),
                        "max_abs_eigenvalue_diff_vs_fp64_inner": float(np.abs(lam_m - lam_g).max()),
                        "note": "engine_set_inner_precision(eng, 32): inner MINRES sweeps of A of up to 16 columns on fp32 tiles (fp64 accumulation); kept as an option for orders where "
                                "the fp32 copy fits next to B's resident tiles - at N=200000 it competes with them for the same memory and buys nothing (docs/history/DESIGN_rounds_1_to_5.md section 11); not the default"}
                except Exception as exc:   # noqa: BLE001  (e.g. no room for the fp32 copy)
                    extras["configs3_gjd"]["inner_fp32"] = {"error": repr(exc)[:300]}
                g.close()
            except Exception as exc:       # noqa: BLE001
                extras["configs3_gjd"] = {"error": repr(exc)[:300]}

        # ---- configs[4]: matrix-free hashed diagonal-dominant operator, lowest=8, DPR -------------------------
        if args.free_n > 0:
            try:
                fn = args.free_n
                fstorage = storage      # one GPU: every symmetric pair generated once (partial-sum slabs at N=10^6: 133 GB)
                f = make_engine(fn, 8, 80, fstorage, gev=True)
                f.set_hashed_operator(1, args.sparsity, seed=1)
                f.set_identity(2)                                        # B = I as src/benchmark_free.f90:65-76
                f.c.bench_apply2(32, 1)                                  # untimed: lazy workspace (partial-sum slabs of a 32-column launch) allocated
                f.c.synchronize(); f.c.reset_stats()
                dt_f, it_f, lam_f = timed_solves(f, "DPR", 1, args.tol)
                sf = f.c.stats()
                entries = float(fn) * float(fn) / world                  # entries of A one rank's sweep stands for
                per_launch_ms = sf.apply_kernel_ms / max(int(sf.apply_launches), 1)
                extras["configs4_free"] = {
                    "workload": f"N={fn} matrix-free hashed diagonal-dominant operator (entries generated in registers, "
                                f"{'each symmetric pair once' if fstorage == 'symmetric' else 'row slab per rank'}), B = I, "
                                f"lowest=8, DPR, tol={args.tol}, {world} GPU(s)",
                    "iters": it_f, "seconds": round(dt_f, 3), "iterations_per_s": round(it_f / dt_f, 4),
                    "sweeps": int(sf.applies), "launches": int(sf.apply_launches), "ms_per_launch": round(per_launch_ms, 2),
                    "entries_of_A_per_s_per_rank": round(entries / (per_launch_ms * 1e-3), 0) if per_launch_ms > 0 else None,
                    "eigenvalues": [float(x) for x in lam_f[:3]]}
                # issue-port roofline of the generated sweeps (GEN_CYCLES_PER_WAVE_EVALUATION above): a launch generates every entry of
                # the rank's part once and feeds it to 2 MFMAs per 16 columns - 16 columns per launch (232 cycles per wave-evaluation)
                # or 32 (the generating variant of the wide kernel: 26 x 4 + 4 x 64 = 360); frac = model time / measured time
                evals = entries * (0.5 if fstorage == "symmetric" else 1.0)
                launches = max(int(sf.apply_launches), 1)
                n32 = min(max((int(sf.apply_cols) - 16 * launches) // 16, 0), launches) if fstorage == "symmetric" else 0
                n16 = launches - n32
                rate = 1024 * 2.4e9 * 64
                model_ms = evals * (n16 * GEN_CYCLES_PER_WAVE_EVALUATION + n32 * (26 * 4 + 4 * 64)) / rate * 1e3
                peak_evals = rate / GEN_CYCLES_PER_WAVE_EVALUATION
                if per_launch_ms > 0:
                    extras["configs4_free"]["roofline"] = {
                        "bound": "valu-int + mfma on one issue port", "unit": "hash evaluations/s", "achieved": round(evals / (per_launch_ms * 1e-3), 0),
                        "peak": round(evals * launches / (model_ms * 1e-3), 0), "frac": round(model_ms / sf.apply_kernel_ms, 4),
                        "launches_of_16_columns": n16, "launches_of_32_columns": n32,
                        "peak_16_columns": round(peak_evals, 0), "peak_32_columns": round(rate / (26 * 4 + 4 * 64), 0),
                        "model": GEN_MODEL + "; a 32-column launch (matvec_symw_kernel<2, GEN>) pays the 26 VALU instructions once for 4 MFMAs: 360 cycles"}
                f.close()
            except Exception as exc:       # noqa: BLE001
                extras["configs4_free"] = {"error": repr(exc)[:300]}

    if not args.headline_only and args.harness_n > 0 and world == 1:
        # ---- the reference's own benchmark program (src/benchmark_free.f90): its matrix-free test operator, B = I, DPR --------------
        # A_ij = cos(log(sqrt(atan2(e_lo, e_hi)))) * 1e-4 (+ i on the diagonal), e = exp(real(i) / real(N)).  Round 6: with 0 < e_lo <= e_hi
        # the entry is a function of ONE variable, x = 1 - |2 log e_i - 2 log e_j|, evaluated as a degree-17 polynomial (csrc/common.h:
        # dav_harness_poly; 19 fp64 VALU instructions per entry instead of four library calls, ~395).  Every symmetric pair is evaluated
        # once.  Roofline: the issue-port model of the generated sweeps (configs4_free) - VALU work and the fp64 MFMAs it feeds share the
        # SIMD's issue port: 19 x 4 + 2 x 64 = 204 cycles per 64 entries and 16 columns.
        try:
            hb = {}
            h1 = make_engine(1000, 3, 20, "symmetric", gev=True)
            h1.set_harness_operator(1); h1.set_identity(2)
            for _ in range(3):
                h1.solve("DPR", 1000, 1e-8, want_vectors=False)
            dt_h, it_h, lam_h = timed_solves(h1, "DPR", 20, 1e-8)
            h1.close()
            hb["reference_configuration"] = {"workload": "benchmark_free.f90:80-111: N=1000, lowest=3, max_dim_sub=20, tol=1e-8, A = cos row generator, B = I, DPR",
                                             "ms_per_solve": round(dt_h / 20 * 1e3, 4), "iters_per_solve": it_h // 20,
                                             "iterations_per_s": round(it_h / dt_h, 1), "eigenvalues": [float(x) for x in lam_h]}
            rate = 1024 * 2.4e9 * 64 / HARNESS_CYCLES_PER_WAVE_EVALUATION          # entries / s of the model, 16 columns per launch

            def harness_solve(n_h, lowest_h, max_dim_h):
                h = make_engine(n_h, lowest_h, max_dim_h, "symmetric", gev=True)
                h.set_harness_operator(1); h.set_identity(2)
                if n_h <= 200000:
                    h.solve("DPR", 1000, 1e-8, want_vectors=False)             # warm-up (lazy workspace)
                else:
                    h.c.bench_apply2(32, 1)                                    # untimed: the partial-sum slabs of a 32-column launch allocated
                h.c.synchronize(); h.c.reset_stats()
                dt, it, lam = timed_solves(h, "DPR", 1, 1e-8)
                st = h.c.stats()
                h.close()
                entries = 0.5 * float(n_h) * (float(n_h) + 1.0)
                launches = max(int(st.apply_launches), 1)
                per_launch = st.apply_kernel_ms / launches
                # a launch generates every entry once: for 16 columns (204 cycles per 64 entries) or - blocks wider than 16 columns, the
                # generating variant of the wide kernel - for 32 (HARNESS_CYCLES_32)
                n32 = min(max((int(st.apply_cols) - 16 * launches) // 16, 0), launches)
                n16 = launches - n32
                unit_rate = 1024 * 2.4e9 * 64
                model_ms = entries * (n16 * HARNESS_CYCLES_PER_WAVE_EVALUATION + n32 * HARNESS_CYCLES_32) / unit_rate * 1e3
                return {"workload": f"N={n_h}, lowest={lowest_h}, max_dim_sub={max_dim_h}, tol=1e-8, the reference's test operator (entries generated in the sweep, each symmetric pair once), B = I, DPR",
                        "iters": it, "seconds": round(dt, 4), "iterations_per_s": round(it / dt, 3), "sweeps": int(st.applies),
                        "launches_of_16_columns": n16, "launches_of_32_columns": n32, "columns_swept": int(st.apply_cols),
                        "ms_per_launch": round(per_launch, 3), "ms_in_sweep_kernels": round(st.apply_kernel_ms, 2), "eigenvalues": [float(x) for x in lam[:3]],
                        "entries_evaluated_per_s": round(entries * launches / (st.apply_kernel_ms * 1e-3), 0) if st.apply_kernel_ms > 0 else None,
                        "roofline": {"bound": "valu-fp64 + mfma on one issue port", "unit": "entries/s",
                                     "achieved": round(entries * launches / (st.apply_kernel_ms * 1e-3), 0) if st.apply_kernel_ms > 0 else None,
                                     "peak": round(entries * launches / (model_ms * 1e-3), 0) if model_ms > 0 else None,
                                     "frac": round(model_ms / st.apply_kernel_ms, 4) if st.apply_kernel_ms > 0 else None,
                                     "peak_16_columns": round(rate, 0), "peak_32_columns": round(unit_rate / HARNESS_CYCLES_32, 0),
                                     "entries_per_launch": entries, "model": HARNESS_MODEL}}
            hb["large"] = harness_solve(args.harness_n, 3, 20)
            hb["roofline"] = dict(hb["large"]["roofline"], N=args.harness_n)
            # A/B: the same sweep with the formula as written (four library calls per entry: DAV_HARNESS_LIBM=1 at dav_create) - round 5's path
            try:
                os.environ["DAV_HARNESS_LIBM"] = "1"
                hl = fd.CEngine(n=args.harness_n, max_cols=16, device=device)
                try:
                    hl.set_storage(1)
                    il = np.arange(1, args.harness_n + 1, dtype=np.float32)
                    hl.set_operator_harness(0, np.exp(il / np.float32(args.harness_n), dtype=np.float32).astype(np.float64))
                    hl.apply(0, 0, 0, 16, 1, 0); hl.synchronize(); hl.reset_stats()
                    hl.apply(0, 0, 0, 16, 1, 0); hl.synchronize()
                    hb["library_call_chain_sweep_ms"] = round(hl.stats().apply_kernel_ms, 3)
                    hb["library_call_chain_rate_entries_per_s"] = round(hl.bench_harness_rate(3000), 0)
                finally:
                    hl.close()
            finally:
                os.environ.pop("DAV_HARNESS_LIBM", None)
            if args.harness_n2 > 0:
                # configs[4]'s order on the reference's own operator: a FULL solve (lowest = 8, the doubling policy's 16 / 32 / 64-column blocks)
                hb["solve_at_configs4_order"] = harness_solve(args.harness_n2, 8, 80)
            extras["configs4_free_harness"] = hb
        except Exception as exc:       # noqa: BLE001
            extras["configs4_free_harness"] = {"error": repr(exc)[:300]}

    if not args.headline_only:
        # ---- drop-in entry + CPU baseline: rank 0, one GPU only (both need the matrix in host memory) ---------
        if rank == 0 and world == 1 and args.small_n > 0 and not (args.no_dropin and args.no_cpu_baseline):
            cn = args.cpu_n or args.small_n
            if not args.no_dropin:
                # A Fortran program that calls the reference-signature generic three times on a host matrix, in a FRESH process
                # (fortran_davidson_amd/fortran/dropin_timing.f90: no Python, no PyTorch): first call = what a process pays once,
                # the others = the steady state of a call (engine created, matrix uploaded over PCIe, solved, eigenvectors
                # downloaded, everything released).  Default storage: symmetric tiles when the symmetry probe of the host matrix
                # passes; DAVIDSON_STORAGE=full = the whole matrix as until round 4; an asymmetric input takes that path by itself.
                def timing_child(asym, env_extra):
                    exe = os.path.join(ROOT, "fortran_davidson_amd", "lib", "dropin_timing")
                    env = dict(os.environ, DAVIDSON_VERBOSE="1", **env_extra)
                    env.pop("DAVIDSON_HIP_LIB", None)
                    res = subprocess.run([exe, str(cn), "8", str(asym)], capture_output=True, text=True, timeout=900, env=env)
                    out = {}
                    for ln in res.stdout.splitlines():
                        if "DROPIN_TIMING" in ln:
                            ln = ln[ln.index("DROPIN_TIMING"):]
                            f = ln.split()
                            secs = [float(x) for x in f[f.index("seconds=") + 1:f.index("seconds=") + 4]]
                            evs = [float(x) for x in f[f.index("eigenvalues=") + 1:f.index("eigenvalues=") + 4]]
                            out.update({"first_call_seconds": round(secs[0], 4), "seconds": round(min(secs[1:]), 4), "iters": int(f[3].split("=")[1]),
                                        "iterations_per_s": round(int(f[3].split("=")[1]) / min(secs[1:]), 2), "eigenvalues": evs})
                        if "davidson dense call:" in ln:
                            out["symmetric_tiles"] = "symmetric tiles=T" in ln
                            out["phases_ms_create_upload_solve_destroy"] = [float(x) for x in ln.split("]=")[1].split()]
                    if "seconds" not in out:
                        raise RuntimeError((res.stdout + res.stderr)[-300:])
                    return out
                try:
                    d0 = timing_child(0, {})
                    extras["dropin"] = {"call": "generalized_eigensolver(matrix, eigenvalues, eigenvectors, lowest, method, max_iterations, tolerance, iters) - "
                                                "src/davidson.f90:51-52, from a Fortran program in a fresh process", "N": cn, **d0,
                                        "upload_GB": round((4.0 if d0.get("symmetric_tiles") else 8.0) * cn * cn / 1e9, 3)}
                    d1 = timing_child(0, {"DAVIDSON_STORAGE": "full"})
                    extras["dropin"]["full_storage"] = {**d1, "upload_GB": round(8.0 * cn * cn / 1e9, 3),
                                                        "max_abs_eigenvalue_diff": float(np.abs(np.array(d1["eigenvalues"]) - np.array(d0["eigenvalues"])).max())}
                    d2 = timing_child(1, {})
                    extras["dropin"]["asymmetric_input"] = {**d2, "note": "one entry of the matrix changed by 1e-9: the symmetry probe fails and the call uploads the full matrix",
                                                            "max_abs_eigenvalue_diff_vs_full_storage": float(np.abs(np.array(d2["eigenvalues"]) - np.array(d1["eigenvalues"])).max())}
                except Exception as exc:       # noqa: BLE001
                    extras["dropin"] = {"error": repr(exc)[:300]}
            if not args.no_cpu_baseline:
                # two orders: the configs[1] problem and a second, larger one that supports the extrapolation to the timed workload
                cn2 = args.cpu_n2 if args.cpu_n2 > cn else 0
                raw = cpu_baseline([cn] + ([cn2] if cn2 else []), 8, args.tol, args.sparsity, bench_free="reference_configuration" in extras.get("configs4_free_harness", {}))
                runs = raw.get("runs") or []
                if "benchmark_free" in raw and "reference_configuration" in extras.get("configs4_free_harness", {}):
                    bf, gpu = raw["benchmark_free"], extras["configs4_free_harness"]["reference_configuration"]
                    extras["configs4_free_harness"]["cpu_baseline"] = {
                        "value": round(bf["iters"] / bf["seconds"], 4), "unit": "iterations/s", "cores": raw.get("cores"), "kind": raw.get("kind"),
                        "seconds": round(bf["seconds"], 3), "iters": bf["iters"], "eigenvalues": bf["evals"],
                        "max_abs_eigenvalue_diff_vs_gpu": float(np.abs(np.array(bf["evals"]) - np.array(gpu["eigenvalues"])).max()),
                        "sample": "the reference's benchmark program as ONE call (oracle/ref_driver.f90: ref_free_solve_benchmark = src/benchmark_free.f90:80-111: "
                                  "N=1000, lowest=3, max_dim_sub=20, DPR; its free_matmul under OpenMP), whole solve incl. its N unit-vector diagonal probes"}
                if runs and "seconds" in runs[0]:
                    r0 = runs[0]
                    extras["cpu_baseline"] = {
                        "value": round(r0["iters"] / r0["seconds"], 4), "unit": "iterations/s", "cores": raw["cores"],
                        "kind": raw["kind"], "seconds": round(r0["seconds"], 3), "iters": r0["iters"],
                        "seconds_first_call": round(r0.get("seconds_first_call", r0["seconds"]), 3),
                        "sockets": raw.get("sockets"), "numa_nodes": raw.get("numa_nodes"), "logical_cpus": raw.get("logical_cpus"),
                        "cpus_available_to_this_job": raw.get("cpu_share"), "cgroup_cpu_quota": raw.get("cgroup_cpu_quota"),
                        "cpu_model": raw.get("cpu_model"),
                        "sample": f"one full solve (the second of two: the first warms MKL up) at N={cn}, lowest=8, DPR, tol={args.tol} (the configs[1] problem - the timed "
                                  "N=200000 matrix needs 320 GB in the reference's full storage and (m+1) sweeps of it per iteration) "
                                  "by the reference built with flang+MKL (oracle/_ref) on as many MKL / OpenMP threads as this job has CPUs (cores: the "
                                  "affinity mask cut down by the cgroup quota - the box shows 256 hardware threads, a one-GPU job gets a share of "
                                  "them); the matrix is generated in the child under OpenMP (parallel first touch)",
                        "max_abs_eigenvalue_diff_vs_gpu": float(np.abs(np.array(r0["evals"][:3]) - np.array(extras["small"]["eigenvalues"])).max())
                        if cn == args.small_n and "eigenvalues" in extras.get("small", {}) else None}
                    cb = extras["cpu_baseline"]
                    # sweeps of A the reference makes and what they cost on these host cores
                    per_n = []
                    for r in runs:
                        widths = [2 * 8 * 2 ** i for i in range(r["iters"])]
                        sweeps = sum(m + 1 for m in widths)
                        per_n.append({"N": r["n"], "iters": r["iters"], "seconds": round(r["seconds"], 3),
                                      "iterations_per_s": round(r["iters"] / r["seconds"], 4), "basis_widths": widths, "sweeps_of_A": sweeps,
                                      "GB_swept": round(sweeps * 8.0 * r["n"] * r["n"] / 1e9, 1),
                                      "GBps_if_all_time_were_sweeps": round(sweeps * 8.0 * r["n"] * r["n"] / r["seconds"] / 1e9, 1),
                                      "dgemv_sweep_GBps": round(r.get("dgemv_sweep_GBps", 0.0), 1),
                                      "dgemv_transposed_GBps": round(r.get("dgemv_transposed_GBps", 0.0), 1),
                                      "openmp_read_GBps": round(r.get("openmp_read_GBps", 0.0), 1),
                                      "openmp_gemv_GBps": round(r.get("openmp_gemv_GBps", 0.0), 1), "generate_seconds": r.get("generate_seconds")})
                    cb["by_order"] = per_n
                    if "dgemv_sweep_GBps" in r0:
                        cb["dgemv_sweep_GBps"] = round(r0["dgemv_sweep_GBps"], 1)
                        cb["dgemv_note"] = ("MKL DGEMV 'N' (what lapack_matrix_vector calls, src/lapack_wrapper.f90:362) on a matrix whose pages were first "
                                            "touched in parallel, on the job's CPUs; by_order also carries the transposed form, a plain OpenMP read and a "
                                            "plain OpenMP matrix-vector product of the same matrix: where the last two are several times the DGEMV "
                                            "rate, what holds the reference back on this host is MKL's DGEMV itself, not page placement or the thread count")
                        bw = max(r.get("dgemv_sweep_GBps", 0.0) for r in runs)
                        w2 = [2 * lowest * 2 ** i for i in range(total_iters // args.steps)]
                        s2 = sum(m + 1 for m in w2)
                        est = s2 * 8.0 * float(n) * float(n) / (bw * 1e9)
                        bw_omp = max(r.get("openmp_gemv_GBps", 0.0) for r in runs)
                        cb["extrapolation_to_timed_workload"] = {
                            "N": n, "lowest": lowest, "basis_widths": w2, "sweeps_of_A": s2, "bytes_per_sweep": 8.0 * float(n) * float(n),
                            "dgemv_rate_used_GBps": round(bw, 1),
                            "seconds_per_solve_at_measured_dgemv_rate": round(est, 1),
                            "iterations_per_s": round((total_iters // args.steps) / est, 5),
                            "seconds_per_solve_if_the_sweeps_ran_at_the_openmp_gemv_rate": round(s2 * 8.0 * float(n) * float(n) / (bw_omp * 1e9), 1) if bw_omp > 0 else None,
                            "assumption": "the reference's (m+1) sweeps of A per iteration (m DGEMVs for the residues + 1 DGEMM, src/davidson.f90:163-170,223) "
                                          "at the best DGEMV rate measured above on these cores, full storage (320 GB - would have to fit host memory); "
                                          "QR and the small eigenproblem not counted; by_order shows how the measured solves compare with that sweep model"}
                else:
                    extras["cpu_baseline"] = raw

    if rank == 0:
        # the driver's record keeps the first ~20 scalar keys of `roofline` and cuts strings at ~120 characters: what a reader needs to
        # recompute both fractions comes first, the long texts last
        first = ["bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "avg_launch_ms", "launches", "ms_per_solve", "non_kernel_ms_per_solve",
                 "hbm_N", "hbm_k", "hbm_algorithmic_bytes", "hbm_ms_end_to_end", "hbm_GBps_end_to_end", "hbm_frac", "hbm_frac_kernel_only",
                 "hbm_traffic", "hbm_frac_of_measured_read", "hbm_k16_frac", "apply_non_kernel_ms_per_solve", "flops_per_launch",
                 "algorithmic_bytes_per_launch", "columns_per_launch", "kernel"]
        last = ["note", "traffic_provenance"]
        roofline = {**{k_: roofline[k_] for k_ in first if k_ in roofline},
                    **{k_: v_ for k_, v_ in roofline.items() if k_ not in first and k_ not in last},
                    **{k_: roofline[k_] for k_ in last if k_ in roofline}}
        line = {"metric": "Davidson iterations/sec (dense DPR solve, matrix resident in HBM) + A*V HBM GB/s vs roofline",
                "value": round(value, 4), "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"N={n} dense fp64 lowest={lowest} DPR max_dim_sub={max_dim} tol={args.tol} sparsity={args.sparsity} seed=1 storage={storage}",
                           "N": n, "lowest": lowest, "max_dim_sub": max_dim, "storage": storage, "sparsity": args.sparsity, "seed": 1,
                           "iters_per_solve": total_iters // args.steps, "parallelism": (f"block rows of the lower triangle over {world} GPUs, row slabs of the panels" if storage == "symmetric" else f"row-slab x{world}") if world > 1 else "single GPU",
                           "generate_seconds": round(t_gen, 2), "storage_detail": storage_words},
                "eigenvalues": [float(x) for x in lam[:3]],
                "roofline": roofline, "apply": apply_k, "hbm_measured": hbm_measured}
        line.update(extras)
        if "cpu_baseline" not in line:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
