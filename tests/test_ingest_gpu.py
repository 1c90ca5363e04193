"""On-disk ingest on the GPU: files in the reference's text format (write_matrix / read_matrix,
src/tests/test_utils.f90:118-166) and raw float64, streamed by row blocks into the resident layouts
(full row slab, symmetric tiles, several ranks).  The resident matrix must equal the file bit for bit:
checked through the diagonal, through A*X against numpy and through the golden eigenpairs."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import fortran_davidson_amd as fd
from fortran_davidson_amd.engine_c import OP_A, OP_B, PANEL_V, PANEL_W, DavidsonHipError
from oracle import davidson_oracle as O

pytestmark = pytest.mark.gpu
EV_TOL = 1e-8


def write_text(path, A):
    # write_matrix: one list-directed value per line, row i outer, column j inner
    with open(path, "w") as f:
        f.write("".join("   %.16E     \n" % v for v in np.asarray(A).reshape(-1)))


def write_f64(path, A):
    np.ascontiguousarray(A, dtype="<f8").tofile(path)


def resident_matches(eng, A, which=OP_A, k=5):
    """diag and A*X of the matrix held by the engine against numpy on the file's matrix."""
    n = A.shape[0]
    assert np.array_equal(eng.get_diagonal(which), np.diag(A))
    rng = np.random.default_rng(3)
    X = rng.standard_normal((n, k))
    eng.panel_put(PANEL_V, 0, X)
    eng.apply(which, PANEL_V, 0, k, PANEL_W, 0)
    W = eng.panel_get(PANEL_W, 0, k)
    ref = A @ X
    assert np.abs(W - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("storage", [0, 1])
@pytest.mark.parametrize("fmt", ["text", "f64"])
def test_reference_test_matrix_from_file(golden, tmp_path, storage, fmt):
    """The reference's own 100 x 100 matrix.txt content, dumped the way write_matrix would, solved from the file."""
    manifest, arrays = golden
    A = arrays["matrix_txt__A"]
    path = tmp_path / ("m." + fmt)
    (write_text if fmt == "text" else write_f64)(path, A)
    with fd.CEngine(n=100, max_cols=64) as e:
        e.set_storage(storage)
        e.set_dense_file(OP_A, path, fmt)
        resident_matches(e, A)
    case = manifest["dense"]["matrix_txt_dpr"]
    with fd.DavidsonEngine(100, case["lowest"], case["max_dim"], storage="symmetric" if storage else "full") as eng:
        eng.read_matrix(1, path, fmt)
        lam, vec, iters = eng.solve("DPR", case["max_it"], case["tol"])
    assert np.abs(lam - arrays["matrix_txt_dpr__evals"]).max() < EV_TOL
    assert iters == case["iters"]
    assert (np.linalg.norm(A @ vec - vec * lam[None, :], axis=0) < case["tol"]).all()


@pytest.mark.parametrize("n", [1, 31, 257, 777])
@pytest.mark.parametrize("storage", [0, 1])
def test_ragged_orders_and_small_read_chunks(tmp_path, monkeypatch, n, storage):
    """Orders that are not multiples of the 32 x 32 copy tile or the 256 tile edge; the text is read in
    997-byte pieces so that numbers are cut at every possible place."""
    monkeypatch.setenv("DAV_INGEST_CHUNK", "997")
    monkeypatch.setenv("DAV_INGEST_THREADS", "3")
    A = O.generate_diagonal_dominant(n, 1e-2, seed=77)
    path = tmp_path / "a.txt"
    write_text(path, A)
    with fd.CEngine(n=n, max_cols=16) as e:
        e.set_storage(storage)
        e.set_dense_file(OP_A, path, "text")
        resident_matches(e, A, k=min(5, 16))


def test_generalized_pair_from_files_matches_in_memory_solve(tmp_path, monkeypatch):
    monkeypatch.setenv("DAVIDSON_STORAGE", "full")         # the dense front end's default is symmetric tiles since round 5; the file engine below keeps full rows
    n, L = 400, 3
    A = O.generate_diagonal_dominant(n, 1e-3, seed=1)
    B = O.generate_diagonal_dominant(n, 1e-3, 1.0, seed=2)
    write_text(tmp_path / "a.txt", A)
    write_f64(tmp_path / "b.f64", B)
    lam0, vec0, it0 = fd.generalized_eigensolver(A, L, "GJD", 100, 1e-8, None, B)
    with fd.DavidsonEngine(n, L, gev=True) as eng:
        eng.read_matrix(1, tmp_path / "a.txt")
        eng.read_matrix(2, tmp_path / "b.f64", "f64")
        lam, vec, it = eng.solve("GJD", 100, 1e-8)
    assert it == it0 and np.array_equal(lam, lam0)        # same bits in HBM -> same run


def test_row_blocks_in_any_order_and_size(tmp_path):
    n = 1100
    A = O.generate_diagonal_dominant(n, 1e-2, seed=5)
    bounds = [0, 1, 40, 41, 300, 811, 1100]                  # ragged blocks, one larger than a 256-row tile
    blocks = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1)]
    for storage in (0, 1):
        with fd.CEngine(n=n, max_cols=16) as e:
            e.set_storage(storage)
            e.dense_begin(OP_A)
            for r0, r1 in reversed(blocks):
                padded = np.full((r1 - r0, n + 3), np.nan)   # leading dimension larger than n
                padded[:, :n] = A[r0:r1]
                lib = fd.hip_lib()
                rc = lib.dav_dense_put_rows(e.h, C.c_int(OP_A), C.c_int64(r0), C.c_int64(r1 - r0),
                                            padded.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(n + 3))
                assert rc == 0, lib.dav_last_error()
            e.dense_end(OP_A)
            resident_matches(e, A)


def test_staging_buffers_wrap_on_a_large_matrix(tmp_path):
    """N = 6000: 288 MB of float64 = three 128 MiB staging buffers worth of rows (double buffering wraps)."""
    n = 6000
    with fd.CEngine(n=n, max_cols=16) as g:
        g.set_dense_generated(OP_A, 11, 1e-3)
        d = g.get_diagonal(OP_A)
    A = O.generate_diagonal_dominant(n, 1e-3, seed=11)
    assert np.array_equal(np.diag(A), d)
    path = tmp_path / "big.f64"
    write_f64(path, A)
    for storage in (0, 1):
        with fd.CEngine(n=n, max_cols=16) as e:
            e.set_storage(storage)
            e.set_dense_file(OP_A, path, "f64")
            resident_matches(e, A, k=8)


@pytest.mark.parametrize("nranks", [2, 3])
def test_each_rank_ingests_only_its_row_slab(golden, tmp_path, nranks):
    """Row-slab ranks (loopback transport, one GPU): every rank opens the same files, keeps its rows."""
    manifest, arrays = golden
    name = "n1000_gev_restart_dpr"
    case = manifest["dense"][name]
    n = case["n"]
    A = O.generate_diagonal_dominant(n, case["sparsity"], seed=case["seed_a"])
    B = O.generate_diagonal_dominant(n, case["sparsity"], 1.0, seed=case["seed_b"])
    write_text(tmp_path / "a.txt", A)
    write_f64(tmp_path / "b.f64", B)
    engs = [fd.DavidsonEngine(n, case["lowest"], case["max_dim"], gev=True, rank=r, nranks=nranks) for r in range(nranks)]
    handles = (C.c_void_p * nranks)(*[e.c.h for e in engs])
    assert fd.hip_lib().dav_local_group_join(handles, nranks) == 0
    out = [None] * nranks

    def work(r):
        engs[r].read_matrix(1, tmp_path / "a.txt", "text")
        engs[r].read_matrix(2, tmp_path / "b.f64", "f64")
        out[r] = engs[r].solve("DPR", case["max_it"], case["tol"])

    threads = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in threads]
    [t.join(timeout=300) for t in threads]
    assert all(o is not None for o in out), "a rank did not finish"
    for lam, vec, iters in out:
        assert np.abs(lam - arrays[f"{name}__evals"]).max() < EV_TOL
        assert iters == case["iters"]
    for e in engs:
        e.close()


def test_bad_files_fail_loudly_and_leave_the_engine_usable(tmp_path):
    n = 64
    A = O.generate_diagonal_dominant(n, 1e-2, seed=3)
    with fd.CEngine(n=n, max_cols=16) as e:
        with pytest.raises(DavidsonHipError, match="cannot open"):
            e.set_dense_file(OP_A, tmp_path / "missing.txt", "text")
        write_text(tmp_path / "short.txt", A[: n - 1])
        with pytest.raises(DavidsonHipError, match="file ends in row 64"):
            e.set_dense_file(OP_A, tmp_path / "short.txt", "text")
        write_text(tmp_path / "long.txt", np.vstack([A, A[:1]]))
        with pytest.raises(DavidsonHipError, match="more than 64 x 64"):
            e.set_dense_file(OP_A, tmp_path / "long.txt", "text")
        with open(tmp_path / "junk.txt", "w") as f:
            f.write("1.0\n2.0\nhello\n")
        with pytest.raises(DavidsonHipError, match="not a number: 'hello'"):
            e.set_dense_file(OP_A, tmp_path / "junk.txt", "text")
        write_f64(tmp_path / "short.f64", A[:, : n - 1])
        with pytest.raises(DavidsonHipError, match="size is not 8 n\\^2"):
            e.set_dense_file(OP_A, tmp_path / "short.f64", "f64")
        with pytest.raises(DavidsonHipError, match="no streaming upload open"):
            e.dense_put_rows(OP_A, 0, A[:4])
        e.dense_begin(OP_A)
        with pytest.raises(DavidsonHipError, match="another streaming upload is open"):
            e.dense_begin(OP_B)
        with pytest.raises(DavidsonHipError, match="bad arguments"):
            e.dense_put_rows(OP_A, n - 2, A[:4])
        e.dense_put_rows(OP_A, 0, A)
        e.dense_end(OP_A)
        resident_matches(e, A)
        write_text(tmp_path / "ok.txt", A)
        e.set_dense_file(OP_A, tmp_path / "ok.txt", "text")     # and again through the file path
        resident_matches(e, A)
