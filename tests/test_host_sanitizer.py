"""CPU-side sanitizer build of the host-only code (the reference's Debug build with run-time checking,
src/CMakeLists.txt:13-17, is its only "sanitizer").  No GPU: AddressSanitizer / UBSan are not available for device code on
this pool, and none of what runs here touches the HIP runtime.
  * fortran_davidson_amd/csrc/ingest.hip - text parser (dav_parse_text_f64's engine) and the two file readers: plain C++,
    compiled with g++ -fsanitize=address,undefined and driven by tests/host_sanitizer/ingest_driver.cpp;
  * the Fortran host units numeric_kinds / lapack_wrapper / array_utils: flang -fsanitize=address,
    tests/host_sanitizer/fortran_units.f90;
  * the rank decisions of the block orthonormalisation (fortran/davidson_ortho.f90: what replaced the reference's Householder QR of
    the whole basis, src/lapack_wrapper.f90:176-236) on host arrays, against the column choices of DGEQRF with the column order
    preserved, for DAV_ORTHO_EARLY = 0 and 1: tests/host_sanitizer/ortho_driver.f90."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fortran_davidson_amd", "csrc")
FSRC = os.path.join(ROOT, "fortran_davidson_amd", "fortran")
HERE = os.path.join(ROOT, "tests", "host_sanitizer")
FC = "/opt/rocm/lib/llvm/bin/flang"
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
ENV.pop("LD_PRELOAD", None)


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_ingest_host_code_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "ingest_asan")
    res = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I" + CSRC, "-x", "c++",
                          os.path.join(CSRC, "ingest.hip"), os.path.join(HERE, "ingest_driver.cpp"), "-o", exe, "-lpthread"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    run = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=600, env=ENV)
    assert run.returncode == 0 and "host sanitizer driver: ok" in run.stdout, (run.stdout + run.stderr)[-4000:]
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-4000:]


@pytest.mark.skipif(not os.path.exists(FC), reason="flang not available")
def test_fortran_host_units_under_asan(tmp_path):
    objs = []
    for unit in ("numeric_kinds", "lapack_wrapper", "array_utils"):
        obj = str(tmp_path / (unit + ".o"))
        res = subprocess.run([FC, "-g", "-O1", "-fsanitize=address", "-module-dir", str(tmp_path), "-c", os.path.join(FSRC, unit + ".f90"), "-o", obj],
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-3000:]
        objs.append(obj)
    exe = str(tmp_path / "fortran_units")
    res = subprocess.run([FC, "-g", "-O1", "-fsanitize=address", "-module-dir", str(tmp_path), os.path.join(HERE, "fortran_units.f90"), *objs,
                          "-L/opt/conda/lib", "-Wl,--no-as-needed", "-lmkl_intel_lp64", "-lmkl_sequential", "-lmkl_core",
                          "-Wl,-rpath,/opt/conda/lib", "-o", exe], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(ENV, ASAN_OPTIONS="detect_leaks=0"))
    assert run.returncode == 0 and "fortran units under the sanitizer: ok" in run.stdout, (run.stdout + run.stderr)[-4000:]
    assert "ERROR: AddressSanitizer" not in run.stderr, run.stderr[-4000:]


MKL = ["-L/opt/conda/lib", "-Wl,--no-as-needed", "-lmkl_intel_lp64", "-lmkl_sequential", "-lmkl_core", "-Wl,-rpath,/opt/conda/lib"]


@pytest.fixture(scope="module")
def ortho_driver(tmp_path_factory):
    if not os.path.exists(FC):
        pytest.skip("flang not available")
    tmp = tmp_path_factory.mktemp("ortho")
    objs = []
    for unit in ("numeric_kinds", "lapack_wrapper", "davidson_knobs", "davidson_ortho"):
        obj = str(tmp / (unit + ".o"))
        res = subprocess.run([FC, "-g", "-O1", "-fsanitize=address", "-module-dir", str(tmp), "-c", os.path.join(FSRC, unit + ".f90"), "-o", obj],
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-3000:]
        objs.append(obj)
    exe = str(tmp / "ortho_driver")
    res = subprocess.run([FC, "-g", "-O1", "-fsanitize=address", "-module-dir", str(tmp), os.path.join(HERE, "ortho_driver.f90"), *objs, *MKL, "-o", exe],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    return exe


@pytest.mark.parametrize("early", ["0", "1"])
def test_rank_decisions_of_the_block_orthonormalisation_against_householder_qr(ortho_driver, early):
    """Banded, block-diagonal, duplicated-column, zero-column, near-dependent (1e-3 .. 1e-14) and inside-the-basis correction blocks
    through the PRODUCT's passes (block_orthonormalise over a host backend): the result is orthonormal, spans every column
    Householder QR keeps, and the columns declared dependent are the ones QR declares (see the driver's header for the sense in
    which DAV_ORTHO_EARLY = 0 matches); restart_transform and dependent_columns on inputs with known answers."""
    run = subprocess.run([ortho_driver], capture_output=True, text=True, timeout=600, env=dict(ENV, ASAN_OPTIONS="detect_leaks=0", DAV_ORTHO_EARLY=early))
    assert run.returncode == 0 and "ortho driver: ok" in run.stdout, (run.stdout + run.stderr)[-4000:]
    assert f"DAV_ORTHO_EARLY on: {'T' if early == '1' else 'F'}" in run.stdout
    assert "ERROR: AddressSanitizer" not in run.stderr, run.stderr[-4000:]
    assert run.stdout.count("ok=T") >= 22 and "ok=F" not in run.stdout
