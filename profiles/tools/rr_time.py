import ctypes as C, numpy as np, time, os, sys
os.environ.setdefault("MKL_THREADING_LAYER", "SEQUENTIAL")
mkl = C.CDLL("/opt/conda/lib/libmkl_rt.so", mode=C.RTLD_GLOBAL)
def I(x): return C.byref(C.c_int(x))
def D(x): return C.byref(C.c_double(x))
def ptr(a): return a.ctypes.data_as(C.c_void_p)
def dsyev(A):
    n = A.shape[0]; a = np.asfortranarray(A.copy()); w = np.zeros(n); lwork = 64 * n + 100; work = np.zeros(lwork); info = C.c_int()
    mkl.dsyev_(b"V", b"U", I(n), ptr(a), I(n), ptr(w), ptr(work), I(lwork), C.byref(info)); return w, a
def dsyevd(A):
    n = A.shape[0]; a = np.asfortranarray(A.copy()); w = np.zeros(n); lwork = 1 + 6 * n + 2 * n * n; work = np.zeros(lwork); liwork = 3 + 5 * n; iwork = np.zeros(liwork, dtype=np.int32); info = C.c_int()
    mkl.dsyevd_(b"V", b"U", I(n), ptr(a), I(n), ptr(w), ptr(work), I(lwork), ptr(iwork), I(liwork), C.byref(info)); return w, a
def dsyevr(A, k):
    n = A.shape[0]; a = np.asfortranarray(A.copy()); w = np.zeros(n); z = np.zeros((n, k), order="F"); m = C.c_int(); isuppz = np.zeros(2 * n, dtype=np.int32)
    lwork = 64 * n + 100; work = np.zeros(lwork); liwork = 10 * n + 10; iwork = np.zeros(liwork, dtype=np.int32); info = C.c_int()
    mkl.dsyevr_(b"V", b"I", b"U", I(n), ptr(a), I(n), D(0.0), D(0.0), I(1), I(k), D(0.0), C.byref(m), ptr(w), ptr(z), I(n), ptr(isuppz), ptr(work), I(lwork), ptr(iwork), I(liwork), C.byref(info))
    return w[:k], z
def dsyevx(A, k):
    n = A.shape[0]; a = np.asfortranarray(A.copy()); w = np.zeros(n); z = np.zeros((n, k), order="F"); m = C.c_int()
    lwork = 64 * n + 100; work = np.zeros(lwork); iwork = np.zeros(5 * n, dtype=np.int32); ifail = np.zeros(n, dtype=np.int32); info = C.c_int()
    mkl.dsyevx_(b"V", b"I", b"U", I(n), ptr(a), I(n), D(0.0), D(0.0), I(1), I(k), D(0.0), C.byref(m), ptr(w), ptr(z), I(n), ptr(work), I(lwork), ptr(iwork), ptr(ifail), C.byref(info))
    return w[:k], z
rng = np.random.default_rng(0)
for n in (32, 64, 128):
    # a projected matrix like the solver's: diag ~ small integers + small couplings
    Q = np.linalg.qr(rng.standard_normal((n, n)))[0]
    A = Q @ np.diag(np.arange(1, n + 1) + 1e-3 * rng.standard_normal(n)) @ Q.T; A = (A + A.T) / 2
    for name, f in (("dsyev", lambda: dsyev(A)), ("dsyevd", lambda: dsyevd(A)), ("dsyevr32", lambda: dsyevr(A, min(32, n))), ("dsyevx32", lambda: dsyevx(A, min(32, n))), ("dsyevr_all", lambda: dsyevr(A, n))):
        f(); t0 = time.perf_counter()
        for _ in range(50): w, z = f()
        dt = (time.perf_counter() - t0) / 50
        k = z.shape[1]
        res = np.abs(A @ z[:, :k] - z[:, :k] * w[:k]).max(); orth = np.abs(z[:, :k].T @ z[:, :k] - np.eye(k)).max()
        print(f"n={n:4d} {name:10s} {dt*1e6:8.1f} us  resid {res:.1e} orth {orth:.1e}")


# the pieces of a subset solve called one by one: DSYTRD (tridiagonal reduction), DSTEMR (MRRR on the leading k pairs), DORMTR (back-transformation)
def pieces(A, k):
    n = A.shape[0]; a = np.asfortranarray(A.copy()); d = np.zeros(n); e = np.zeros(n); tau = np.zeros(n)
    lwork = 64 * n + 100; work = np.zeros(lwork); info = C.c_int()
    mkl.dsytrd_(b"U", I(n), ptr(a), I(n), ptr(d), ptr(e), ptr(tau), ptr(work), I(lwork), C.byref(info))
    w = np.zeros(n); z = np.zeros((n, k), order="F"); m = C.c_int(); isuppz = np.zeros(2 * n, dtype=np.int32); tryrac = C.c_int(1)
    lw2 = 18 * n + 100; work2 = np.zeros(lw2); liw = 10 * n + 100; iwork = np.zeros(liw, dtype=np.int32)
    mkl.dstemr_(b"V", b"I", I(n), ptr(d), ptr(e), D(0.0), D(0.0), I(1), I(k), C.byref(m), ptr(w), ptr(z), I(n), I(k), ptr(isuppz), C.byref(tryrac),
                ptr(work2), I(lw2), ptr(iwork), I(liw), C.byref(info))
    mkl.dormtr_(b"L", b"U", b"N", I(n), I(k), ptr(a), I(n), ptr(tau), ptr(z), I(n), ptr(work), I(lwork), C.byref(info))
    return w[:k], z


for n in (64, 128, 256):
    Q = np.linalg.qr(rng.standard_normal((n, n)))[0]
    A = Q @ np.diag(np.arange(1, n + 1) + 1e-3 * rng.standard_normal(n)) @ Q.T; A = (A + A.T) / 2
    for k in (16, 32):
        pieces(A, k); t0 = time.perf_counter()
        for _ in range(50): w, z = pieces(A, k)
        dt = (time.perf_counter() - t0) / 50
        res = np.abs(A @ z - z * w).max(); orth = np.abs(z.T @ z - np.eye(k)).max()
        print(f"n={n:4d} dsytrd+dstemr({k:2d})+dormtr {dt*1e6:8.1f} us  resid {res:.1e} orth {orth:.1e}")
    for name, f in (("dsyev", lambda: dsyev(A)), ("dsyevd", lambda: dsyevd(A)), ("dsyevr_all", lambda: dsyevr(A, n))):
        f(); t0 = time.perf_counter()
        for _ in range(20): f()
        print(f"n={n:4d} {name:10s} {(time.perf_counter() - t0) / 20 * 1e6:8.1f} us")
