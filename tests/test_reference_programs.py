"""The reference's OWN test programs (src/tests/*.f90, compiled unchanged by oracle/build_ref.sh against
our modules and libraries; binaries only, under oracle/_ref/ref_tests/) executed on this engine, then
judged with the criteria of the reference's Python checkers (src/tests/test_davidson.py:15-79,
src/tests/test_lapack.py:30-52): eigenvalues `allclose` to scipy `eigh` of the dumped matrices, DPR and
GJD agreeing, and every property line of the Fortran programs printing T.

Dump format: one value per line, row-major (src/tests/test_utils.f90:139-166)."""
import os
import re
import subprocess

import numpy as np
import pytest
import scipy.linalg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_tests")


def run(prog, cwd):
    exe = os.path.join(BIN, prog)
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not built (oracle/build_ref.sh needs /root/reference)")
    env = dict(os.environ, OMP_NUM_THREADS="8")       # the callbacks' OpenMP loop (free_matmul) on a 256-thread host
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600, cwd=cwd, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    return res.stdout


def load(cwd, name, shape=None):
    a = np.loadtxt(os.path.join(cwd, name))
    return a if shape is None else a.reshape(shape)


def test_reference_lapack_wrapper_program(tmp_path):
    """test_call_lapack.f90 + test_lapack.py:30-52 - host wrappers only, runs without a GPU."""
    run("test_call_lapack", tmp_path)
    mtx = load(tmp_path, "test_lapack_matrix.txt", (50, 50))
    stx = load(tmp_path, "test_lapack_stx.txt", (50, 50))
    for tag, b in (("", None), ("_gen", stx)):
        es = load(tmp_path, f"test_lapack_eigenvalues{tag}.txt")
        vs = load(tmp_path, f"test_lapack_eigenvectors{tag}.txt", (50, 50))
        w, v = scipy.linalg.eigh(mtx, b=b)
        assert np.allclose(es, w)
        assert np.allclose(np.abs(vs), np.abs(v))
    q = load(tmp_path, "test_lapack_qr.txt", (50, 50))
    assert np.allclose(q.T @ q, np.eye(50), atol=1e-12)


@pytest.mark.gpu
def test_reference_dense_properties_program(tmp_path):
    out = run("test_dense_properties", tmp_path)
    flags = re.findall(r":\s+([TF])\s*$", out, flags=re.M)
    assert len(flags) >= 9 and all(f == "T" for f in flags), out


@pytest.mark.gpu
def test_reference_free_properties_program(tmp_path):
    out = run("test_free_properties", tmp_path)
    flags = re.findall(r"succeeded:\s*([TF])", out)
    assert len(flags) == 3 and all(f == "T" for f in flags), out
    assert re.search(r"DPR method:\s*T", out), out


@pytest.mark.gpu
def test_reference_dense_numpy_program(tmp_path):
    """test_dense_numpy.f90 judged as test_davidson.py:15-51 does."""
    run("test_dense_numpy", tmp_path)
    for tag, gen in (("spec", False), ("gen", True)):
        mtx = load(tmp_path, f"test_dense_{tag}_matrix.txt", (50, 50))
        stx = load(tmp_path, f"test_dense_{tag}_stx.txt", (50, 50)) if gen else None
        es_dpr = load(tmp_path, f"test_dense_{tag}_eigenvalues_DPR.txt")
        es_gjd = load(tmp_path, f"test_dense_{tag}_eigenvalues_GJD.txt")
        ref = scipy.linalg.eigh(mtx, b=stx)[0][:3]
        assert np.allclose(es_dpr, es_gjd)
        assert np.allclose(ref, es_dpr)
        for meth, es in (("DPR", es_dpr), ("GJD", es_gjd)):
            vs = load(tmp_path, f"test_dense_{tag}_eigenvectors_{meth}.txt", (50, 3))
            bx = vs if stx is None else stx @ vs
            assert (np.linalg.norm(mtx @ vs - bx * es[None, :], axis=0) < 1e-8).all()


@pytest.mark.gpu
def test_reference_free_numpy_program(tmp_path):
    """test_free_numpy.f90 judged as test_davidson.py:54-79 does."""
    run("test_free_numpy", tmp_path)
    mtx = load(tmp_path, "matrix_free.txt", (50, 50))
    stx = load(tmp_path, "stx_free.txt", (50, 50))
    es = load(tmp_path, "eigenvalues_DPR_free.txt")
    vs = load(tmp_path, "eigenvectors_DPR_free.txt", (50, 3))
    assert np.allclose(es, scipy.linalg.eigh(mtx, b=stx)[0][:3])
    assert (np.linalg.norm(mtx @ vs - (stx @ vs) * es[None, :], axis=0) < 1e-8).all()


@pytest.mark.gpu
def test_reference_demo_program_main(tmp_path):
    """src/main.f90: N=100 generalized problem, GJD vs DPR, tol 1e-5, max_dim 10."""
    out = run("main", tmp_path)
    assert re.search(r"computed by different methods are the same:\s+T", out), out
    errs = [float(x) for x in re.findall(r"\|\|Error\|\|:\s+([0-9.Ee+-]+)", out)]
    assert len(errs) == 6 and max(errs) < 1e-5, out          # tolerance of the demo (main.f90:52-54)
    its = [int(x) for x in re.findall(r"converged in:\s+(\d+)", out)]
    assert len(its) == 2 and all(0 < i <= 1000 for i in its)


@pytest.mark.gpu
def test_reference_benchmark_free_program(tmp_path):
    """src/benchmark_free.f90: N=1000 matrix-free through host callbacks (free_matmul), B = I."""
    out = run("benchmark_free", tmp_path)
    flags = re.findall(r"succeeded:\s*([TF])", out)
    assert len(flags) == 3 and all(f == "T" for f in flags), out
