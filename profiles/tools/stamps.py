"""Diagnostic build (scratch/lib/libdav_stamp.so): where the cycles of a unit go in matvec_symw_kernel."""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch  # noqa
import numpy as np
import fortran_davidson_amd as fd
from fortran_davidson_amd.engine_c import OP_A, PANEL_V, PANEL_W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
ks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "32,64,16").split(",")]
lib = fd.hip_lib()
with fd.CEngine(n=n, max_cols=64) as e:
    e.set_storage(1)
    e.set_dense_generated(OP_A, 1, 1e-3)
    e.panel_put(PANEL_V, 0, np.random.default_rng(0).standard_normal((n, 64)))
    e.synchronize()
    out = (C.c_ulonglong * 12)()
    for k in ks:
        e.bench_apply2(k, 1)
        lib.dav_symw_stamps(out)
        ms, kms, nbytes, flops = e.bench_apply2(k, 3)
        lib.dav_symw_stamps(out)
        hs, bar, sm, units, tk, tr, vm, f0, f1, f2, un, nw = [float(x) for x in out]
        print(json.dumps({"k": k, "kernel_ms": round(kms, 3), "cycles_per_unit": {"half_steps": round(hs / units, 1), "z_write_and_barrier": round(bar / units, 1),
                          "sum_and_flush": round(sm / units, 1)}, "mfma_cycles_per_unit": 64 * 64 * (2 if k > 16 else 1), "clock_GHz": round(tk / tr * 0.1, 3), "vmcnt_wait_per_unit": round(vm / units, 1), "unit_cycles": round(un / units, 1), "loop_share_of_wave_time": round(un / tk, 4), "wave_time_share_of_kernel": round(tr / 1e8 / (1024 * 3 * kms * 1e-3), 4), "per_half_step": {"loads": round(f0 / units / 4, 1), "transposition": round(f1 / units / 4, 1), "mfmas": round(f2 / units / 4, 1)}}), flush=True)
