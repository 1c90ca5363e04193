"""The Fortran boundary as a Fortran user sees it: programs that `use davidson` compile and link
against our modules (CPU), run on the GPU and agree with the golden values of the reference.
Where /root/reference is present (build container only) the reference's OWN test programs are
compiled, unchanged, against our modules - the source-compatibility proof of the drop-in API."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FC = "/opt/rocm/lib/llvm/bin/flang"
MODDIR = os.path.join(ROOT, "fortran_davidson_amd", "fortran", "build")
LIBDIR = os.path.join(ROOT, "fortran_davidson_amd", "lib")
SRC = os.path.join(ROOT, "tests", "fortran")
REF_TESTS = "/root/reference/src/tests"

needs_flang = pytest.mark.skipif(not os.path.exists(FC), reason="flang not available")


def compile_link(sources, exe, workdir, test_build=False):
    """test_build: link against the TEST build of the engine (lib/test/libdavidson_hip.so, the one with the shared-memory
    transport) - for programs that bind a door of csrc/davidson_hip_private.h themselves"""
    hip_dirs = [f"-L{os.path.join(LIBDIR, 'test')}"] if test_build else []
    cmd = [FC, "-O1", "-fopenmp=libiomp5", f"-I{MODDIR}", "-module-dir", str(workdir), *sources, *hip_dirs,
           f"-L{LIBDIR}", "-lfortran_davidson_amd", "-ldavidson_hip", f"-Wl,-rpath,{LIBDIR}",
           "-L/opt/conda/lib", "-Wl,-rpath,/opt/conda/lib", "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=workdir)
    assert res.returncode == 0, res.stderr[-3000:]
    return exe


def build_programs(tmp_path):
    # the build travels: binaries are created under tests/fortran/_bin (git-ignored)
    bindir = os.path.join(SRC, "_bin")
    os.makedirs(bindir, exist_ok=True)
    dense = compile_link([os.path.join(SRC, "prog_dense.f90")], os.path.join(bindir, "prog_dense"), tmp_path)
    free = compile_link([os.path.join(SRC, "harness_ops.f90"), os.path.join(SRC, "prog_free.f90")],
                        os.path.join(bindir, "prog_free"), tmp_path)
    return dense, free


def build_ranks_program(tmp_path):
    bindir = os.path.join(SRC, "_bin")
    os.makedirs(bindir, exist_ok=True)
    return compile_link([os.path.join(SRC, "prog_ranks.f90")], os.path.join(bindir, "prog_ranks"), tmp_path, test_build=True)


def build_options_program(tmp_path):
    bindir = os.path.join(SRC, "_bin")
    os.makedirs(bindir, exist_ok=True)
    return compile_link([os.path.join(SRC, "prog_options.f90")], os.path.join(bindir, "prog_options"), tmp_path)


def build_device_operator_program(tmp_path):
    """links the caller's own library too (lib/test/libuser_operator.so: tests/helpers/user_operator.hip)"""
    bindir = os.path.join(SRC, "_bin")
    os.makedirs(bindir, exist_ok=True)
    tdir = os.path.join(LIBDIR, "test")
    cmd = [FC, "-O1", "-fopenmp=libiomp5", f"-I{MODDIR}", "-module-dir", str(tmp_path), os.path.join(SRC, "prog_device_operator.f90"),
           f"-L{tdir}", "-luser_operator", f"-L{LIBDIR}", "-lfortran_davidson_amd", "-ldavidson_hip", f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{tdir}",
           "-L/opt/conda/lib", "-Wl,-rpath,/opt/conda/lib", "-o", os.path.join(bindir, "prog_device_operator")]
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp_path)
    assert res.returncode == 0, res.stderr[-3000:]
    return os.path.join(bindir, "prog_device_operator")


def build_ingest_program(tmp_path):
    bindir = os.path.join(SRC, "_bin")
    os.makedirs(bindir, exist_ok=True)
    return compile_link([os.path.join(SRC, "prog_ingest.f90")], os.path.join(bindir, "prog_ingest"), tmp_path)


@needs_flang
def test_user_programs_compile_and_link(tmp_path):
    if not os.path.isdir(MODDIR):
        pytest.skip("module files not built")
    dense, free = build_programs(tmp_path)
    assert os.path.exists(dense) and os.path.exists(free)
    assert os.path.exists(build_ingest_program(tmp_path))
    assert os.path.exists(build_ranks_program(tmp_path))
    assert os.path.exists(build_options_program(tmp_path))
    assert os.path.exists(build_device_operator_program(tmp_path))


@needs_flang
@pytest.mark.skipif(not os.path.isdir(REF_TESTS), reason="reference sources only exist in the build container")
@pytest.mark.parametrize("prog", ["test_dense_properties", "test_free_properties", "test_dense_numpy",
                                  "test_free_numpy", "test_call_lapack"])
def test_reference_test_programs_compile_against_our_modules(tmp_path, prog):
    """Unmodified reference test sources + OUR modules: must compile and link (they run on a GPU box)."""
    srcs = [os.path.join(REF_TESTS, "test_utils.f90"), os.path.join(REF_TESTS, prog + ".f90")]
    compile_link(srcs, str(tmp_path / prog), tmp_path)


def _run(exe):
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    return res.returncode, res.stdout + res.stderr


@needs_flang
@pytest.mark.gpu
def test_dense_program_runs_on_gpu(golden, tmp_path):
    manifest, arrays = golden
    dense, _ = build_programs(tmp_path)
    rc, out = _run(dense)
    assert rc == 0, out
    checks = re.findall(r"CHECK (\S+) ([TF])", out)
    assert len(checks) >= 10 and all(v == "T" for _, v in checks), out
    iters = [int(x) for x in re.search(r"ITERS\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)", out).groups()]
    gold = manifest["dense"]
    assert iters == [gold["c1_n50_std_dpr"]["iters"], gold["c1_n50_std_gjd"]["iters"],
                     gold["c1_n50_gev_dpr"]["iters"], gold["c1_n50_gev_gjd"]["iters"]]
    ev = np.array([float(x) for x in re.search(r"EVALS_DPR(.*)", out).group(1).split()])
    assert np.abs(ev - arrays["c1_n50_std_dpr__evals"]).max() < 1e-8
    ev = np.array([float(x) for x in re.search(r"EVALS_GEN(.*)", out).group(1).split()])
    assert np.abs(ev - arrays["c1_n50_gev_dpr__evals"]).max() < 1e-8


@needs_flang
@pytest.mark.gpu
def test_free_program_runs_on_gpu(golden, tmp_path):
    manifest, arrays = golden
    _, free = build_programs(tmp_path)
    rc, out = _run(free)
    assert rc == 0, out
    checks = re.findall(r"CHECK (\S+) ([TF])", out)
    assert len(checks) >= 4 and all(v == "T" for _, v in checks), out
    ev = np.array([float(x) for x in re.search(r"EVALS_FREE(.*)", out).group(1).split()])
    assert np.abs(ev - arrays["free_n50__evals"]).max() < 1e-8
    iters = [int(x) for x in re.search(r"ITERS\s+(\d+)\s+(\d+)", out).groups()]
    assert iters[0] == manifest["free"]["free_n50"]["iters"]


@needs_flang
@pytest.mark.gpu
def test_ingest_program_runs_on_gpu(tmp_path):
    """Text dump in the reference's write_matrix format, raw float64 and row-block streaming, from Fortran."""
    exe = build_ingest_program(tmp_path)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300, cwd=tmp_path)
    out = res.stdout + res.stderr
    assert res.returncode == 0, out
    checks = re.findall(r"CHECK (\S+) ([TF])", out)
    assert len(checks) == 4 * 6 and all(v == "T" for _, v in checks), out


@needs_flang
@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 3])
def test_multi_rank_fortran_program(tmp_path, nranks):
    """One Fortran process per rank (engine_create(rank, nranks) + engine_comm_init_shm + the generic), the ranks
    sharing this box's GPU: every rank returns the full eigenvectors, rank 0 also checks them against its own
    single-rank solve."""
    exe = build_ranks_program(tmp_path)
    tag = f"dav_prog_ranks_{os.getpid()}_{nranks}"
    procs = [subprocess.Popen([exe, str(r), str(nranks), tag], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              cwd=tmp_path) for r in range(nranks)]
    outs = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    evals = []
    for r, (pr, out) in enumerate(zip(procs, outs)):
        assert pr.returncode == 0, out[-2000:]
        checks = re.findall(r"CHECK (\S+) ([TF])", out)
        assert len(checks) == (6 if r == 0 else 4) and all(v == "T" for _, v in checks), out
        evals.append([float(x) for x in re.search(r"EVALS(.*)", out).group(1).split()])
    assert all(e == evals[0] for e in evals)            # same bits on every rank


@needs_flang
@pytest.mark.gpu
def test_options_program_runs_on_gpu(tmp_path):
    """engine_set_storage / engine_set_inner_precision / engine_set_device_rr from a Fortran program, against the
    drop-in calls on the same matrices."""
    exe = build_options_program(tmp_path)
    rc, out = _run(exe)
    assert rc == 0, out
    checks = re.findall(r"CHECK (\S+) ([TF])", out)
    assert len(checks) == 14 and all(v == "T" for _, v in checks), out


@needs_flang
@pytest.mark.gpu
def test_device_operator_program_runs_on_gpu(tmp_path):
    """engine_set_device_operator (c_funloc of the caller's bind(C) launcher) + the generic on the engine, from a Fortran program,
    against the drop-in dense call on the same banded matrix: DPR and GJD."""
    exe = build_device_operator_program(tmp_path)
    rc, out = _run(exe)
    assert rc == 0, out
    checks = re.findall(r"CHECK (\S+) ([TF])", out)
    assert len(checks) == 12 and all(v == "T" for _, v in checks), out
