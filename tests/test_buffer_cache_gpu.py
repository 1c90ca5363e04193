"""The buffer cache of include/davidson_hip.h (dav_free_buffers): the reference's interface is one call per eigenproblem
(src/davidson.f90:51-52), so the drop-in creates and destroys an engine per call; the blocks of the engine destroyed last are kept
for the next one of the same sizes."""
import numpy as np
import pytest

import fortran_davidson_amd as fd
from oracle import davidson_oracle as O

pytestmark = pytest.mark.gpu


def raw_free():
    import torch
    return torch.cuda.mem_get_info(0)[0]


def test_blocks_are_kept_reused_and_given_back():
    n, lowest = 6000, 4
    a = O.generate_diagonal_dominant(n, 1e-3, seed=7)
    fd.generalized_eigensolver(a[:400, :400].copy(), lowest, "DPR", 100, 1e-8)     # (the runtime loads the kernels: memory of its own)
    fd.free_buffers()
    before = raw_free()
    lam1, vec1, it1 = fd.generalized_eigensolver(a, lowest, "DPR", 100, 1e-8)
    held = before - raw_free()
    assert held > 8 * n * n // 2                           # the tiles (at least the lower triangle) and the panels are still allocated
    lam2, vec2, it2 = fd.generalized_eigensolver(a, lowest, "DPR", 100, 1e-8)
    assert before - raw_free() <= held + (4 << 20)        # the second call took the first one's blocks, it did not allocate its own
    assert it2 == it1 and np.array_equal(lam1, lam2) and np.array_equal(vec1, vec2)   # recycled memory, the same bits
    # another size: the blocks of the first size sit idle through one create-destroy cycle and are then let go
    b = O.generate_diagonal_dominant(3000, 1e-3, seed=8)
    fd.generalized_eigensolver(b, lowest, "DPR", 100, 1e-8)
    fd.generalized_eigensolver(b, lowest, "DPR", 100, 1e-8)
    assert before - raw_free() < held // 2                # (9 MB of tiles instead of 150)
    fd.free_buffers()
    assert before - raw_free() < (64 << 20)               # everything is back (the runtime keeps a little of its own)


def test_engine_reports_idle_blocks_as_free_memory():
    fd.free_buffers()
    with fd.DavidsonEngine(5000, 4, None, storage="symmetric") as eng:
        eng.generate_diagonal_dominant(1, 1e-3, seed=1)
        eng.solve("DPR", 100, 1e-8)
    idle = raw_free()
    with fd.DavidsonEngine(64, 2, None) as eng:            # a small engine: asks the device, the big one's blocks are idle
        free_seen, total = eng.c.device_memory()
    fd.free_buffers()
    assert free_seen > idle and free_seen <= total


def test_the_cache_never_holds_more_than_its_cap():
    """DAVIDSON_BUFFER_CACHE_MB (default 4096; read once per process, hence a child): with a cap of 16 MB the 150 MB of tiles of an
    order-6000 call go straight back to the device when the call returns - a drop-in call on a large matrix must not leave the
    device full behind it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, torch\n"
        "import fortran_davidson_amd as fd\n"
        "from oracle import davidson_oracle as O\n"
        "a = O.generate_diagonal_dominant(6000, 1e-3, seed=7)\n"
        "fd.generalized_eigensolver(a[:400, :400].copy(), 4, 'DPR', 100, 1e-8)\n"
        "fd.free_buffers(); before = torch.cuda.mem_get_info(0)[0]\n"
        "fd.generalized_eigensolver(a, 4, 'DPR', 100, 1e-8)\n"
        "print('HELD', before - torch.cuda.mem_get_info(0)[0])\n" % root)
    held = {}
    for cap in ("16", "4096"):
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, DAVIDSON_BUFFER_CACHE_MB=cap))
        assert res.returncode == 0, (res.stdout + res.stderr)[-2000:]
        held[cap] = int([ln for ln in res.stdout.splitlines() if ln.startswith("HELD")][0].split()[1])
    assert held["16"] < (64 << 20) and held["4096"] > 8 * 6000 * 6000 // 2


def test_pinned_blocks_have_their_own_small_cap():
    """Round 6 (round-5 advisor): page-locked host blocks are capped separately (DAVIDSON_BUFFER_CACHE_PINNED_MB, default 256) from
    the device blocks - a streaming upload's two pinned staging buffers (here 2 x 96 MB) must not stay locked behind a call when the
    cap says 64, and stay for the next upload under the default."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, torch\n"
        "import fortran_davidson_amd as fd\n"
        "from fortran_davidson_amd.engine_c import buffer_cache_held, OP_A\n"
        "n = 3000\n"
        "rows = np.random.default_rng(0).standard_normal((n, n)); rows = rows + rows.T\n"
        "with fd.CEngine(n=n, max_cols=16) as e:\n"
        "    e.dense_begin(OP_A); e.dense_put_rows(OP_A, 0, rows); e.dense_end(OP_A)\n"
        "print('HELD', *buffer_cache_held())\n" % root)
    held = {}
    for cap in ("0", "256"):
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, DAVIDSON_BUFFER_CACHE_PINNED_MB=cap))
        assert res.returncode == 0, (res.stdout + res.stderr)[-2000:]
        held[cap] = [int(x) for x in [ln for ln in res.stdout.splitlines() if ln.startswith("HELD")][0].split()[1:]]
    assert held["0"][1] == 0 and held["0"][0] > 0            # device blocks stay, pinned ones went back
    assert 0 < held["256"][1] <= (256 << 20)
