"""fortran_davidson_amd - MI355X-native block Davidson eigensolver behind the Fortran API of
NLESC-JCER/Fortran_Davidson.

The product is two shared libraries under ``fortran_davidson_amd/lib``:

* ``libdavidson_hip.so``          hand-written HIP kernels for gfx950 + the C ABI (include/davidson_hip.h)
* ``libfortran_davidson_amd.so``  the Fortran host: modules ``davidson``, ``davidson_dense``,
  ``davidson_free``, ``lapack_wrapper``, ``array_utils``, ``numeric_kinds`` (drop-in for the
  reference's modules) and bind(C) doors used by this Python layer.

This Python package is only the harness side (tests, bench, multi-GPU launch plumbing): it calls
the Fortran API through ctypes.  There is no CPU fallback anywhere: importing works without a GPU,
any compute call without one raises.
"""
from .solver import (DavidsonEngine, generalized_eigensolver, generate_diagonal_dominant,  # noqa: F401
                     lapack_generalized_eigensolver, lapack_qr, lapack_sort, generate_preconditioner,
                     lapack_matmul, lapack_solver, norm)
from .engine_c import CEngine, DavidsonHipError, free_buffers  # noqa: F401
from ._lib import hip_lib, fortran_lib, build  # noqa: F401
