"""CPU: the drop-in boundary loads and exports what include/davidson_hip.h declares; the Fortran host
helper modules (lapack_wrapper / array_utils mirrors) agree with the oracle.  No compute on a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import fortran_davidson_amd as fd
from oracle import davidson_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(path):
    return sorted(set(re.findall(r"\b(dav_[a-z0-9_]+)\s*\(", open(path).read())))


def test_c_abi_exports_every_declared_symbol():
    """Both builds export every entry point of the public header; the private header (measurement doors of bench.py, doors of
    the TEST build) is not part of include/: its test-transport doors exist in the test build only."""
    names = _declared(os.path.join(ROOT, "include", "davidson_hip.h"))
    assert len(names) >= 30
    assert not [n for n in names if n.startswith("dav_bench_") or n in ("dav_local_group_join", "dav_comm_init_shm")]
    private = _declared(os.path.join(ROOT, "fortran_davidson_amd", "csrc", "davidson_hip_private.h"))
    test_only = ["dav_local_group_join", "dav_local_group_yield", "dav_comm_init_shm"]
    assert all(n in private for n in test_only + ["dav_bench_apply2", "dav_bench_stream", "dav_apply_inner"])
    libdir = os.path.join(ROOT, "fortran_davidson_amd", "lib")
    fd.hip_lib()                 # first: it brings PyTorch's HIP runtime in before any other copy (see _lib.py)
    for path, is_test in ((os.path.join(libdir, "libdavidson_hip.so"), False), (os.path.join(libdir, "test", "libdavidson_hip.so"), True)):
        lib = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
        want = names + [n for n in private if is_test or n not in test_only]
        missing = [n for n in want if not hasattr(lib, n)]
        assert not missing, (path, missing)
        hdr = open(os.path.join(ROOT, "include", "davidson_hip.h")).read()
        assert lib.dav_version() == int(re.search(r"#define DAV_HIP_ABI_VERSION (\d+)", hdr).group(1))
    assert fd.hip_lib().dav_version() == fd.engine_c.ABI_VERSION


def test_stats_mirror_has_the_size_of_the_c_structure():
    """ctypes mirror of dav_stats against the C header, compiled here (a layout change without a version bump is an overrun
    in a caller built against the old header: ADVICE round 3)."""
    import subprocess
    import tempfile
    src = '#include <stdio.h>\n#include "davidson_hip.h"\nint main(void){printf("%zu %d\\n", sizeof(dav_stats), DAV_HIP_ABI_VERSION);return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")], check=True)
        size, ver = subprocess.run([os.path.join(d, "s")], capture_output=True, text=True, check=True).stdout.split()
    assert int(size) == ctypes.sizeof(fd.engine_c.Stats) and int(ver) == fd.engine_c.ABI_VERSION


@pytest.mark.parametrize("p,q", [(1, 1), (3, 5), (6, 16), (17, 33), (64, 64), (130, 100), (128, 64)])
def test_operand_image_of_a_small_matrix(p, q):
    """The layout the panel kernel reads its small-matrix operand in (round 4): for step s = i / 4 and tile t = j / 16 the 64 values
    lane c + 16 g <-> M[4 s + g][16 t + c] are contiguous; zero outside p x q; tiles per step a multiple of 4."""
    lib = fd.hip_lib()
    rng = np.random.default_rng(p * 1000 + q)
    ld = p + 3
    M = np.asfortranarray(rng.standard_normal((ld, q)))
    n, tp = ctypes.c_int64(), ctypes.c_int64()
    dp = ctypes.POINTER(ctypes.c_double)
    assert lib.dav_pack_operand_image(M.ctypes.data_as(dp), ctypes.c_int64(ld), p, q, None, ctypes.byref(n), ctypes.byref(tp)) == 0
    assert tp.value % 4 == 0 and tp.value * 16 >= q and n.value == (p + 3) // 4 * tp.value * 64
    out = np.full(n.value, np.nan)
    assert lib.dav_pack_operand_image(M.ctypes.data_as(dp), ctypes.c_int64(ld), p, q, out.ctypes.data_as(dp), None, None) == 0
    ref = np.zeros(n.value)
    for i in range(p):
        for j in range(q):
            ref[((i // 4) * tp.value + j // 16) * 64 + (j % 16) + 16 * (i % 4)] = M[i, j]
    assert np.array_equal(out, ref)


def test_fortran_host_exports_api_doors():
    lib = fd.fortran_lib()
    for name in ["fd_dense_solve", "fd_free_solve", "fd_engine_create", "fd_engine_solve", "fd_engine_destroy",
                 "fd_engine_set_operator", "fd_engine_set_dense", "fd_lapack_qr", "fd_lapack_eigensolver"]:
        assert hasattr(lib, name), name
    # the Fortran module procedures themselves are in the library (drop-in link target)
    out = os.popen(f"nm -D {os.path.join(ROOT, 'fortran_davidson_amd', 'lib', 'libfortran_davidson_amd.so')}").read()
    for mod_proc in ["davidson_dense", "davidson_free", "generalized_eigensolver_dense",
                     "generalized_eigensolver_free", "free_matmul", "lapack_qr", "generate_preconditioner"]:
        assert mod_proc in out.lower(), mod_proc


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(fd.DavidsonHipError):
        fd.CEngine(n=64, max_cols=16)


def test_host_generator_matches_oracle_bitwise():
    for n, sp, dv, seed in [(64, 1e-3, None, 1), (50, 1e-3, 1.0, 2), (33, 5e-2, None, 9)]:
        assert np.array_equal(fd.generate_diagonal_dominant(n, sp, dv, seed),
                              O.generate_diagonal_dominant(n, sp, dv, seed))


def test_lapack_wrapper_mirrors(golden):
    _, arrays = golden
    A = O.generate_diagonal_dominant(50, 1e-3, seed=7)
    B = O.generate_diagonal_dominant(50, 1e-3, 1.0, seed=8)
    w, v = fd.lapack_generalized_eigensolver(A)
    assert np.allclose(w, arrays["lapack__dsyev_w"], atol=1e-12)
    assert np.allclose(np.abs(v), np.abs(arrays["lapack__dsyev_v"]), atol=1e-8)
    w, v = fd.lapack_generalized_eigensolver(A, B)
    assert np.allclose(w, arrays["lapack__dsygv_w"], atol=1e-12)
    q = fd.lapack_qr(A[:, :20])
    assert np.allclose(q, arrays["lapack__qr_q"], atol=1e-12)
    x = np.random.default_rng(0).standard_normal(50)
    assert np.allclose(fd.lapack_solver(A, A @ x), x, atol=1e-10)
    assert np.allclose(fd.lapack_matmul("T", "N", A[:, :7], B[:, :5]), A[:, :7].T @ B[:, :5], atol=1e-13)
    assert np.allclose(fd.lapack_matmul("N", "T", A[:, :7], B[:, :7]), A[:, :7] @ B[:, :7].T, atol=1e-13)


def test_sort_and_preconditioner():
    rng = np.random.default_rng(3)
    d = rng.standard_normal(200)
    keys, s = fd.lapack_sort("I", d)
    assert np.array_equal(s, np.sort(d))
    assert np.array_equal(s[keys - 1], d)                     # keys(i) = rank of the original entry i
    keys, s = fd.lapack_sort("D", d)
    assert np.array_equal(s, np.sort(d)[::-1])
    # duplicates: stable
    keys, s = fd.lapack_sort("I", np.array([2.0, 1.0, 2.0, 1.0]))
    assert list(keys) == [3, 1, 4, 2]
    pre = fd.generate_preconditioner(d, 6)
    assert np.array_equal(pre, O.generate_preconditioner(d, 6))
    assert fd.norm(np.array([3.0, 4.0])) == 5.0


@pytest.mark.parametrize("n,nvec", [(40, 40), (40, 6), (48, 48), (64, 8), (150, 150), (150, 20), (300, 64)])
@pytest.mark.parametrize("gev", [False, True])
def test_rayleigh_ritz_solver_all_routes_against_scipy(n, nvec, gev):
    """lapack_rayleigh_ritz: DSYEV/DSYGV below order 48 (the reference's route), divide and conquer above,
    MRRR on a leading subset (Cholesky-reduced when generalized) - same eigenpairs, DSYGV normalisation."""
    import scipy.linalg
    from fortran_davidson_amd.solver import lapack_rayleigh_ritz
    rng = np.random.default_rng(n + nvec)
    H = rng.standard_normal((n, n)); H = (H + H.T) / 2 + np.diag(np.arange(n))
    S = None
    if gev:
        S = rng.standard_normal((n, n)) * 0.02; S = np.eye(n) + (S + S.T) / 2
    w, Y = lapack_rayleigh_ritz(H, nvec, S)
    ref = scipy.linalg.eigh(H, S, eigvals_only=True)[:nvec]
    assert np.abs(w - ref).max() < 1e-10
    SY = Y if S is None else S @ Y
    assert np.abs(H @ Y - SY * w[None, :]).max() < 1e-9
    assert np.abs(Y.T @ SY - np.eye(nvec)).max() < 1e-10


def test_product_library_is_built_without_the_test_transports():
    """lib/libdavidson_hip.so (what a user links) carries no loopback / shared-memory transport - their entry points do not
    exist in it; lib/test/libdavidson_hip.so (same sources, -DDAV_TEST_TRANSPORTS=1, same soname) is what pytest loads."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prod = os.path.join(root, "fortran_davidson_amd", "lib", "libdavidson_hip.so")
    test = os.path.join(root, "fortran_davidson_amd", "lib", "test", "libdavidson_hip.so")
    for path in (prod, test):
        assert os.path.exists(path), path
        dyn = subprocess.run(["readelf", "-d", path], capture_output=True, text=True).stdout
        assert "soname: [libdavidson_hip.so]" in dyn
    syms = {path: subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout for path in (prod, test)}
    for name in ("dav_local_group_join", "dav_comm_init_shm"):
        assert name not in syms[prod] and name in syms[test]
    assert b"DAV_TEST_STALL_MS" not in open(prod, "rb").read() and b"DAV_TEST_STALL_MS" in open(test, "rb").read()
    # conftest.py points pytest at the test build unless the caller chose a library itself
    assert os.path.samefile(fd._lib.HIP_LIB, test) or os.environ.get("DAVIDSON_HIP_LIB") not in (None, test)
    # the Fortran host library links against the product build alone (no test door among its undefined symbols)
    und = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(root, "fortran_davidson_amd", "lib", "libfortran_davidson_amd.so")],
                         capture_output=True, text=True).stdout
    assert "dav_create" in und and "dav_comm_init_shm" not in und and "dav_local_group_join" not in und
