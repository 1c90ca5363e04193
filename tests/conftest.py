import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# pytest runs against the TEST build of the engine (fortran_davidson_amd/lib/test/libdavidson_hip.so: the product's sources plus
# the loopback / shared-memory transports that let several ranks share the one GPU of the test box); the product library
# (lib/libdavidson_hip.so) is built without them.  Same soname: the Fortran host library binds to the copy already loaded, and
# the Fortran test programs (linked with a RUNPATH to lib/) find the test copy first through LD_LIBRARY_PATH.
_TEST_LIB_DIR = os.path.join(ROOT, "fortran_davidson_amd", "lib", "test")
if os.path.exists(os.path.join(_TEST_LIB_DIR, "libdavidson_hip.so")):
    os.environ.setdefault("DAVIDSON_HIP_LIB", os.path.join(_TEST_LIB_DIR, "libdavidson_hip.so"))
    os.environ["LD_LIBRARY_PATH"] = _TEST_LIB_DIR + (":" + os.environ["LD_LIBRARY_PATH"] if os.environ.get("LD_LIBRARY_PATH") else "")


# device code of the tests themselves (tests/helpers/Makefile: the caller's own operator kernel) - __graft_entry__.build() makes it; a
# tree built with `make -C fortran_davidson_amd` alone gets it here (hipcc cross-compiles without a GPU)
if not os.path.exists(os.path.join(_TEST_LIB_DIR, "libuser_operator.so")) and os.path.exists("/opt/rocm/bin/hipcc"):
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "helpers")], capture_output=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Outputs of the reference itself (oracle/gen_golden.py ran it in the build container)."""
    with open(os.path.join(GOLDEN, "reference_cases.json")) as f:
        manifest = json.load(f)
    arrays = dict(np.load(os.path.join(GOLDEN, "reference_cases.npz")))
    return manifest, arrays


def case_matrices(case, arrays):
    """Rebuild the input matrices of a golden dense case (oracle generator or matrix.txt data)."""
    from oracle import davidson_oracle as O
    if case["matrix"] == "matrix_txt":
        return arrays["matrix_txt__A"], None
    A = O.generate_diagonal_dominant(case["n"], case["sparsity"], seed=case["seed_a"])
    B = None
    if case["gev"]:
        B = O.generate_diagonal_dominant(case["n"], case["sparsity"], 1.0, seed=case["seed_b"])
    return A, B
