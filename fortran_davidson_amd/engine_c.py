"""Thin ctypes view of the C ABI (include/davidson_hip.h) - used by the kernel-level parity tests
and by bench.py for the roofline measurement.  One method per C entry point, numpy in/out."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import hip_lib

OP_A, OP_B = 0, 1
PANEL_V, PANEL_W, PANEL_BV, PANEL_X, PANEL_R, PANEL_S = range(6)
METHOD_DPR, METHOD_GJD = 0, 1


# int fn(ctx, hip_stream, n, row0, nloc, k, x_dev, ldx, y_dev, ldy) - include/davidson_hip.h: dav_device_apply_fn
DEVICE_APPLY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64)


class DavidsonHipError(RuntimeError):
    pass


class Stats(C.Structure):
    _fields_ = [("n", C.c_int64), ("nloc", C.c_int64), ("nranks", C.c_int32), ("rank", C.c_int32),
                ("m", C.c_int32), ("applies", C.c_int32), ("apply_cols", C.c_int64),
                ("apply_ms", C.c_double), ("apply_bytes", C.c_double), ("last_apply_ms", C.c_double),
                ("last_apply_bytes", C.c_double), ("gram_ms", C.c_double), ("panel_ms", C.c_double),
                ("comm_ms", C.c_double), ("apply_kernel_ms", C.c_double), ("apply_flops", C.c_double),
                ("apply_launches", C.c_int64), ("restarts", C.c_int64),
                ("allgather_ms", C.c_double), ("reduce_scatter_ms", C.c_double), ("allreduce_ms", C.c_double),
                ("allgather_bytes", C.c_double), ("reduce_scatter_bytes", C.c_double), ("allreduce_bytes", C.c_double),
                ("collectives", C.c_int64), ("comm_ranks", C.c_int32), ("comm_overlap", C.c_int32), ("apply_comm_ms", C.c_double),
                ("b_stored_kernel_ms", C.c_double), ("b_stored_bytes", C.c_double), ("b_stored_flops", C.c_double),
                ("b_generated_kernel_ms", C.c_double), ("b_generated_entries", C.c_double), ("b_generated_flops", C.c_double),
                ("b_stored_launches", C.c_int64), ("b_generated_launches", C.c_int64)]


ABI_VERSION = 108      # DAV_HIP_ABI_VERSION of include/davidson_hip.h this module mirrors


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


class CEngine:
    """RAII wrapper over dav_create/dav_destroy.  `handle` may be borrowed from the Fortran side."""

    def __init__(self, n=None, max_cols=None, gev=False, device=0, rank=0, nranks=1, handle=None):
        self.lib = hip_lib()
        if self.lib.dav_version() != ABI_VERSION:
            raise DavidsonHipError(f"libdavidson_hip.so reports ABI version {self.lib.dav_version()}, engine_c.py mirrors {ABI_VERSION}")
        self.owned = handle is None
        if handle is None:
            h = C.c_void_p()
            self._chk(self.lib.dav_create(C.byref(h), C.c_int(device), C.c_int64(n), C.c_int(max_cols),
                                          C.c_int(1 if gev else 0), C.c_int(rank), C.c_int(nranks)))
            self.h = h
        else:
            self.h = C.c_void_p(handle)
        st = self.stats()
        self.n = st.n

    def _chk(self, rc):
        if rc != 0:
            raise DavidsonHipError(self.lib.dav_last_error().decode())

    def close(self):
        if self.owned and self.h:
            self.lib.dav_destroy(self.h)
        self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- bookkeeping
    def stats(self) -> Stats:
        st = Stats()
        self._chk(self.lib.dav_get_stats_n(self.h, C.byref(st), C.c_size_t(C.sizeof(st))))
        return st

    def reset_stats(self):
        self._chk(self.lib.dav_reset_stats(self.h))

    def set_timing(self, level):
        """0 = no events, 1 = block matvec only (default), 2 = every phase (gram_ms, panel_ms, comm_ms)."""
        self._chk(self.lib.dav_set_timing(self.h, C.c_int(level)))

    def synchronize(self):
        self._chk(self.lib.dav_synchronize(self.h))

    def local_rows(self):
        r0, nl = C.c_int64(), C.c_int64()
        self._chk(self.lib.dav_local_rows(self.h, C.byref(r0), C.byref(nl)))
        return r0.value, nl.value

    def comm_init(self, unique_id: bytes):
        buf = C.create_string_buffer(unique_id, 128)
        self._chk(self.lib.dav_comm_init(self.h, buf))

    def comm_path(self):
        """What the trial of the collective paths decided (dav_comm_path): dict(selected, trial_ran, columns, ms, valid)"""
        sel, ran, cols = C.c_int(), C.c_int(), C.c_int()
        ms = (C.c_double * 3)()
        valid = (C.c_int * 3)()
        self._chk(self.lib.dav_comm_path(self.h, C.byref(sel), C.byref(ran), C.byref(cols), ms, valid))
        names = {-1: "undecided", 0: "program order", 1: "direct exchange", 2: "second stream"}
        return {"selected": names[sel.value], "trial_ran": bool(ran.value), "columns": cols.value,
                "trial_ms_max_over_ranks": {names[i]: round(ms[i], 4) for i in range(3)},
                "validated": {names[i]: bool(valid[i]) for i in range(3)}}

    def comm_init_shm(self, name: str):
        """Test transport for ranks that are processes sharing one GPU (POSIX shared memory `name`)."""
        self._chk(self.lib.dav_comm_init_shm(self.h, name.encode()))

    @staticmethod
    def comm_unique_id() -> bytes:
        lib = hip_lib()
        buf = C.create_string_buffer(128)
        if lib.dav_comm_unique_id(buf) != 0:
            raise DavidsonHipError(lib.dav_last_error().decode())
        return buf.raw

    # -- operators
    def device_memory(self):
        """(free, total) bytes of the engine's device"""
        f, t = C.c_int64(0), C.c_int64(0)
        self._chk(self.lib.dav_device_memory(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def set_storage(self, mode):
        """0 = full storage, 1 = symmetric-tiled (lower block triangle only)."""
        self._chk(self.lib.dav_set_storage(self.h, C.c_int(mode)))

    def set_dense_host(self, which, a):
        a = _f(a)
        self._chk(self.lib.dav_set_dense_host(self.h, C.c_int(which), _dp(a), C.c_int64(a.shape[0])))

    def set_dense_dev(self, which, dev_ptr, lda):
        """a(lda, n) column-major in device memory (e.g. a torch tensor's data_ptr())."""
        self._chk(self.lib.dav_set_dense_dev(self.h, C.c_int(which), C.c_void_p(dev_ptr), C.c_int64(lda)))

    def set_dense_generated(self, which, seed, sparsity, diag_val=None):
        self._chk(self.lib.dav_set_dense_generated(self.h, C.c_int(which), C.c_uint64(seed), C.c_double(sparsity),
                                                   C.c_int(0 if diag_val is None else 1),
                                                   C.c_double(0.0 if diag_val is None else diag_val)))

    # -- streaming ingest (rows in the reference's on-disk order, row-major)
    def dense_begin(self, which):
        self._chk(self.lib.dav_dense_begin(self.h, C.c_int(which)))

    def dense_put_rows(self, which, row0, rows):
        """rows: (nrows, n) C-ordered block of complete rows, global rows row0..row0+nrows-1."""
        r = np.ascontiguousarray(rows, dtype=np.float64)
        self._chk(self.lib.dav_dense_put_rows(self.h, C.c_int(which), C.c_int64(row0), C.c_int64(r.shape[0]), _dp(r),
                                              C.c_int64(r.shape[1])))

    def dense_end(self, which):
        self._chk(self.lib.dav_dense_end(self.h, C.c_int(which)))

    def set_dense_file(self, which, path, fmt="text"):
        """fmt: "text" = the reference's write_matrix/read_matrix format, "f64" = raw row-major float64."""
        code = {"text": 0, "f64": 1}[fmt]
        self._chk(self.lib.dav_set_dense_file(self.h, C.c_int(which), str(path).encode(), C.c_int(code)))

    def set_operator_hashed(self, which, seed, sparsity, diag_val=None):
        self._chk(self.lib.dav_set_operator_hashed(self.h, C.c_int(which), C.c_uint64(seed), C.c_double(sparsity),
                                                   C.c_int(0 if diag_val is None else 1),
                                                   C.c_double(0.0 if diag_val is None else diag_val)))

    def set_operator_harness(self, which, e_table):
        e = np.ascontiguousarray(e_table, dtype=np.float64)
        self._chk(self.lib.dav_set_operator_harness(self.h, C.c_int(which), _dp(e)))

    def set_operator_identity(self, which):
        self._chk(self.lib.dav_set_operator_identity(self.h, C.c_int(which)))

    def set_operator_host(self, which, diag):
        d = np.ascontiguousarray(diag, dtype=np.float64)
        self._chk(self.lib.dav_set_operator_host(self.h, C.c_int(which), _dp(d)))

    def set_operator_device(self, which, fn, ctx, diag):
        """The caller's own block apply on device memory (dav_device_apply_fn): `fn` a ctypes function pointer (DEVICE_APPLY_FN or a
        symbol of a loaded library), `ctx` an integer / c_void_p passed through, `diag` the operator's diagonal (n)."""
        d = np.ascontiguousarray(diag, dtype=np.float64)
        self._device_ops = getattr(self, "_device_ops", {})
        self._device_ops[which] = (fn, ctx)                # keep the callback alive as long as the engine
        self._chk(self.lib.dav_set_operator_device(self.h, C.c_int(which), C.cast(fn, C.c_void_p), C.c_void_p(ctx if isinstance(ctx, int) else C.cast(ctx, C.c_void_p).value), _dp(d)))

    def get_diagonal(self, which):
        d = np.zeros(self.n)
        self._chk(self.lib.dav_get_diagonal(self.h, C.c_int(which), _dp(d)))
        return d

    # -- hot path
    def init_basis(self, ncols):
        idx = np.zeros(ncols, dtype=np.int64)
        self._chk(self.lib.dav_init_basis(self.h, C.c_int(ncols), idx.ctypes.data_as(C.POINTER(C.c_int64))))
        return idx

    def apply(self, which, src_panel, c0, k, dst_panel, d0):
        self._chk(self.lib.dav_apply(self.h, C.c_int(which), C.c_int(src_panel), C.c_int(c0), C.c_int(k),
                                     C.c_int(dst_panel), C.c_int(d0)))

    def apply_inner(self, which, src_panel, c0, k, dst_panel, d0):
        """dav_apply as the GJD correction solve issues it (may read the fp32 copy of the tiles; csrc/davidson_hip_private.h)"""
        self._chk(self.lib.dav_apply_inner(self.h, C.c_int(which), C.c_int(src_panel), C.c_int(c0), C.c_int(k),
                                           C.c_int(dst_panel), C.c_int(d0)))

    def resident_fraction(self, which):
        f = C.c_double()
        self._chk(self.lib.dav_resident_fraction(self.h, C.c_int(which), C.byref(f)))
        return f.value

    def gram(self, panel_p, p0, p, panel_q, q0, q):
        out = np.zeros((p, q), order="F")
        self._chk(self.lib.dav_gram(self.h, C.c_int(panel_p), C.c_int(p0), C.c_int(p), C.c_int(panel_q), C.c_int(q0),
                                    C.c_int(q), _dp(out), C.c_int64(p)))
        return out

    def project(self, c0, k, H, S=None):
        ld = H.shape[0]
        sp = _dp(S) if S is not None else C.POINTER(C.c_double)()
        self._chk(self.lib.dav_project(self.h, C.c_int(c0), C.c_int(k), _dp(H), C.c_int64(ld), sp, C.c_int64(ld)))

    def ritz_residual_correction(self, m, lowest, Y, theta, method=METHOD_DPR):
        Y = _f(Y)
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        res = np.zeros(lowest)
        self._chk(self.lib.dav_ritz_residual_correction(self.h, C.c_int(m), C.c_int(lowest), _dp(Y),
                                                        C.c_int64(Y.shape[0]), _dp(theta), C.c_int(method), _dp(res)))
        return res

    def gjd_correction(self, m, theta, max_inner=500, inner_tol=1e-12):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        it = C.c_int(0)
        self._chk(self.lib.dav_gjd_correction(self.h, C.c_int(m), _dp(theta), C.c_int(max_inner),
                                              C.c_double(inner_tol), C.byref(it)))
        return it.value

    def gjd_correction_n(self, m, ncols, theta, tol_per_col, max_inner=500, inner_tol=1e-12):
        """dav_gjd_correction_n: per-column relative tolerances; a negative entry marks a follower (stops at |tol| or when every
        column with a positive entry has stopped)"""
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        tols = np.ascontiguousarray(tol_per_col, dtype=np.float64)
        assert theta.size >= ncols and tols.size >= ncols
        it = C.c_int(0)
        self._chk(self.lib.dav_gjd_correction_n(self.h, C.c_int(m), C.c_int(ncols), _dp(theta), C.c_int(max_inner),
                                                C.c_double(inner_tol), _dp(tols), C.byref(it)))
        return it.value

    def ortho_gram(self, m, kt):
        Cm = np.zeros((max(m, 1), kt), order="F")
        G = np.zeros((kt, kt), order="F")
        self._chk(self.lib.dav_ortho_gram(self.h, C.c_int(m), C.c_int(kt), _dp(Cm), C.c_int64(max(m, 1)), _dp(G),
                                          C.c_int64(kt)))
        return Cm[:m], G

    def ortho_apply(self, m, kt, Cm, M):
        Cm = _f(Cm) if m > 0 else np.zeros((1, kt), order="F")
        M = _f(M)
        self._chk(self.lib.dav_ortho_apply(self.h, C.c_int(m), C.c_int(kt), _dp(Cm), C.c_int64(Cm.shape[0]), _dp(M),
                                           C.c_int64(M.shape[0])))

    def ortho_apply_all(self, m, kt, Cm, M):
        """dav_ortho_apply on T and on its images in the W (and BV) panels"""
        Cm = _f(Cm) if m > 0 else np.zeros((1, kt), order="F")
        M = _f(M)
        self._chk(self.lib.dav_ortho_apply_all(self.h, C.c_int(m), C.c_int(kt), _dp(Cm), C.c_int64(Cm.shape[0]), _dp(M),
                                               C.c_int64(M.shape[0])))

    def project_ortho(self, m, k, gev=False):
        """dav_project_ortho: ([V T]^T (A T), [V T]^T (B T) | None, V^T T, T^T T) in one fetch"""
        p = m + k
        H = np.zeros((p, k), order="F")
        S = np.zeros((p, k), order="F") if gev else None
        Cm = np.zeros((max(m, 1), k), order="F")
        G = np.zeros((k, k), order="F")
        sp = _dp(S) if gev else C.POINTER(C.c_double)()
        self._chk(self.lib.dav_project_ortho(self.h, C.c_int(m), C.c_int(k), _dp(H), C.c_int64(p), sp, C.c_int64(p), _dp(Cm),
                                             C.c_int64(max(m, 1)), _dp(G), C.c_int64(k)))
        return H, S, Cm[:m], G

    def expand(self, m, kt):
        self._chk(self.lib.dav_expand(self.h, C.c_int(m), C.c_int(kt)))

    def restart(self, m, keep, Yk):
        Yk = _f(Yk)
        self._chk(self.lib.dav_restart(self.h, C.c_int(m), C.c_int(keep), _dp(Yk), C.c_int64(Yk.shape[0])))

    def panel_transform(self, src_panel, s0, p, M, dst_panel, d0):
        M = _f(M)
        self._chk(self.lib.dav_panel_transform(self.h, C.c_int(src_panel), C.c_int(s0), C.c_int(p), _dp(M),
                                               C.c_int64(M.shape[0]), C.c_int(M.shape[1]), C.c_int(dst_panel),
                                               C.c_int(d0)))

    def panel_get(self, panel, c0, k):
        out = np.zeros((self.n, k), order="F")
        self._chk(self.lib.dav_panel_get(self.h, C.c_int(panel), C.c_int(c0), C.c_int(k), _dp(out), C.c_int64(self.n)))
        return out

    def panel_put(self, panel, c0, data):
        data = _f(data)
        self._chk(self.lib.dav_panel_put(self.h, C.c_int(panel), C.c_int(c0), C.c_int(data.shape[1]), _dp(data),
                                         C.c_int64(data.shape[0])))

    def set_width(self, m):
        self._chk(self.lib.dav_set_width(self.h, C.c_int(m)))

    def ranks_agree(self, words):
        w = np.ascontiguousarray(words, dtype=np.float64)
        self._chk(self.lib.dav_ranks_agree(self.h, _dp(w), C.c_int(w.size)))

    def agree_inputs(self, words):
        """the inputs of a solve, verified across the ranks in a fixed-size collective whenever they differ from the last verified ones"""
        w = np.ascontiguousarray(words, dtype=np.float64)
        self._chk(self.lib.dav_agree_inputs(self.h, _dp(w), C.c_int(w.size)))

    def agree_next(self, words):
        w = np.ascontiguousarray(words, dtype=np.float64)
        self._chk(self.lib.dav_agree_next(self.h, _dp(w), C.c_int(w.size)))

    def set_inner_precision(self, bits):
        self._chk(self.lib.dav_set_inner_precision(self.h, C.c_int(bits)))

    def rr_enable(self, on=True):
        self._chk(self.lib.dav_rr_enable(self.h, C.c_int(1 if on else 0)))

    def project_dev(self, c0, k):
        self._chk(self.lib.dav_project_dev(self.h, C.c_int(c0), C.c_int(k)))

    def rr_ritz(self, m, ncorr, lowest, method=METHOD_DPR, want_gram=False):
        """(theta[m], resnorm[lowest], sweeps[, C, G]) from the device-resident projected matrices"""
        theta, res, sweeps = np.zeros(m), np.zeros(lowest), C.c_int(0)
        if want_gram:
            Cm, G = np.zeros((m, ncorr), order="F"), np.zeros((ncorr, ncorr), order="F")
            self._chk(self.lib.dav_rr_ritz(self.h, C.c_int(m), C.c_int(ncorr), C.c_int(lowest), C.c_int(method), _dp(theta), _dp(res),
                                           _dp(Cm), C.c_int64(m), _dp(G), C.c_int64(ncorr), C.byref(sweeps)))
            return theta, res, sweeps.value, Cm, G
        self._chk(self.lib.dav_rr_ritz(self.h, C.c_int(m), C.c_int(ncorr), C.c_int(lowest), C.c_int(method), _dp(theta), _dp(res),
                                       None, C.c_int64(0), None, C.c_int64(0), C.byref(sweeps)))
        return theta, res, sweeps.value

    def rr_get(self, m, ncols):
        theta, Y = np.zeros(m), np.zeros((m, ncols), order="F")
        self._chk(self.lib.dav_rr_get(self.h, C.c_int(m), C.c_int(ncols), _dp(theta), _dp(Y), C.c_int64(m)))
        return theta, Y

    def bench_apply(self, k, reps, which=OP_A):
        """(ms per apply END TO END: pack + kernel + reduction, algorithmic bytes per apply)"""
        ms, nbytes = C.c_double(), C.c_double()
        self._chk(self.lib.dav_bench_apply(self.h, C.c_int(which), C.c_int(k), C.c_int(reps), C.byref(ms),
                                           C.byref(nbytes)))
        return ms.value, nbytes.value

    def bench_stream(self, doubles=0, reps=5):
        """(copy GB/s, triad GB/s) of plain streaming kernels on this box (read + written bytes)"""
        cp, tr = C.c_double(), C.c_double()
        self._chk(self.lib.dav_bench_stream(self.h, C.c_int64(doubles), C.c_int(reps), C.byref(cp), C.byref(tr)))
        return cp.value, tr.value

    def bench_stream3(self, doubles=0, reps=5):
        """(copy, triad, read-only GB/s): bench_stream plus a kernel that only reads (two arrays, one partial sum per workgroup written)"""
        cp, tr, rd = C.c_double(), C.c_double(), C.c_double()
        self._chk(self.lib.dav_bench_stream3(self.h, C.c_int64(doubles), C.c_int(reps), C.byref(cp), C.byref(tr), C.byref(rd)))
        return cp.value, tr.value, rd.value

    def bench_harness_rate(self, iters=2000):
        """entries per second of the matrix-free test operator's arithmetic (atan2 + sqrt + log + cos, fp64) on registers"""
        r = C.c_double(0.0)
        self._chk(self.lib.dav_bench_harness_rate(self.h, C.c_int(iters), C.byref(r)))
        return r.value

    def bench_apply2(self, k, reps, which=OP_A):
        """(ms per apply end to end, ms of the block-matvec kernel alone, algorithmic bytes, flops) per apply"""
        ms, kms, nbytes, flops = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        self._chk(self.lib.dav_bench_apply2(self.h, C.c_int(which), C.c_int(k), C.c_int(reps), C.byref(ms), C.byref(kms),
                                            C.byref(nbytes), C.byref(flops)))
        return ms.value, kms.value, nbytes.value, flops.value


def parse_text_f64(data: bytes) -> np.ndarray:
    """The engine's parser of the reference's text dumps (host only): all numbers in `data`."""
    lib = hip_lib()
    n = C.c_size_t(0)
    if lib.dav_parse_text_f64(data, C.c_size_t(len(data)), None, C.c_size_t(0), C.byref(n)) != 0:
        raise DavidsonHipError(lib.dav_last_error().decode())
    out = np.empty(n.value)
    if lib.dav_parse_text_f64(data, C.c_size_t(len(data)), _dp(out), C.c_size_t(out.size), C.byref(n)) != 0:
        raise DavidsonHipError(lib.dav_last_error().decode())
    return out


def buffer_cache_held():
    """(idle device bytes, idle pinned host bytes) the buffer cache holds right now (measurement door, csrc/davidson_hip_private.h)"""
    lib = hip_lib()
    d, h = C.c_int64(), C.c_int64()
    lib.dav_buffer_cache_held(C.byref(d), C.byref(h))
    return d.value, h.value


def free_buffers() -> None:
    """Return the buffer cache's idle device / pinned blocks (kept from the engine destroyed last) to the device."""
    hip_lib().dav_free_buffers()
