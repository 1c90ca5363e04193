"""What one rank of a P-rank configs[2] run does, measured on ONE GPU: P engines (rank r of P) as threads over the loopback
transport of the TEST build, taking turns on the device (DAV_TEST_SERIALIZE=1: between two collectives one rank runs at a time),
so that a rank's HIP-event times are those of a rank that owns a GPU.  Prints per setting the min / median / max over the ranks of
the sweep-kernel time, the local apply time (packing + kernel + fixed-order reduction, collectives taken out), Gram / panel phases
and the collectives per solve - the inputs of bench.py's scaling model.

    python profiles/tools/ranks_rehearsal.py [--ranks 8] [--n 200000] [--lowest 16] [--env DAV_SYM_RUN9=0,8,12,16,32]
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("DAVIDSON_HIP_LIB", os.path.join(ROOT, "fortran_davidson_amd", "lib", "test", "libdavidson_hip.so"))
os.environ["LD_LIBRARY_PATH"] = os.path.dirname(os.environ["DAVIDSON_HIP_LIB"]) + ":" + os.environ.get("LD_LIBRARY_PATH", "")
os.environ["DAV_TEST_SERIALIZE"] = "1"
import numpy as np          # noqa: E402
import fortran_davidson_amd as fd          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--n", type=int, default=200000)
ap.add_argument("--lowest", type=int, default=16)
ap.add_argument("--max-dim", type=int, default=80)
ap.add_argument("--solves", type=int, default=3)
ap.add_argument("--env", default="", help="NAME=v1,v2,...: one run per value (read at dav_create)")
args = ap.parse_args()
name, values = (args.env.split("=")[0], args.env.split("=")[1].split(",")) if args.env else ("", [""])


def run(nranks):
    engs = [fd.DavidsonEngine(args.n, args.lowest, args.max_dim, rank=r, nranks=nranks, storage="symmetric") for r in range(nranks)]
    if nranks > 1:
        handles = (C.c_void_p * nranks)(*[e.c.h for e in engs])
        assert fd.hip_lib().dav_local_group_join(handles, nranks) == 0
    out, err = [None] * nranks, [None] * nranks

    def work(r):
        try:
            eng = engs[r]
            eng.generate_diagonal_dominant(1, 1e-3, seed=1)
            eng.solve("DPR", 1000, 1e-8, want_vectors=False)
            eng.c.set_timing(2)
            eng.c.synchronize(); eng.c.reset_stats()
            for _ in range(args.solves):
                lam, _, iters = eng.solve("DPR", 1000, 1e-8, want_vectors=False)
            eng.c.synchronize()
            st = eng.c.stats()
            k = float(args.solves)
            out[r] = dict(iters=iters, sweep_kernel_ms=st.apply_kernel_ms / k, apply_local_ms=(st.apply_ms - st.apply_comm_ms) / k,
                          gram_ms=st.gram_ms / k, panel_ms=st.panel_ms / k, collectives=st.collectives / k, lam0=float(lam[0]))
        except Exception as exc:      # noqa: BLE001
            err[r] = exc
        finally:
            if nranks > 1:
                fd.hip_lib().dav_local_group_yield(engs[r].c.h)

    th = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in th]
    [t.join() for t in th]
    for e in engs:
        e.close()
    assert all(x is None for x in err), err
    return out


for v in values:
    if name:
        os.environ[name] = v
    res = run(args.ranks)
    line = {"ranks": args.ranks, "n": args.n, name or "env": v, "iters": res[0]["iters"], "collectives_per_solve": res[0]["collectives"]}
    first = int(os.environ.get("DAV_TEST_SERIALIZE_FIRST", "0"))
    line["first_rank_in_turn"] = first
    for key in ("sweep_kernel_ms", "apply_local_ms", "gram_ms", "panel_ms"):
        line[key + "_by_rank"] = [round(r[key], 3) for r in res]
        # the rank that goes first in every turn starts on a GPU that has idled through the loopback transport's host-staged
        # collective (tens of ms): its figures carry a clock ramp that a rank owning a GPU does not see - kept apart
        rest = sorted(r[key] for i, r in enumerate(res) if i != first or len(res) == 1)
        line[key] = {"min": round(rest[0], 3), "median": round(rest[len(rest) // 2], 3), "max": round(rest[-1], 3),
                     "first_in_turn": round(res[first][key], 3)}
    print(json.dumps(line), flush=True)
