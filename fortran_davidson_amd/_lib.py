"""Locating, building and loading the native libraries.  Fails loudly when they are missing."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
HIP_LIB = os.environ.get("DAVIDSON_HIP_LIB", os.path.join(LIB_DIR, "libdavidson_hip.so"))   # override: A/B builds
FORTRAN_LIB = os.path.join(LIB_DIR, "libfortran_davidson_amd.so")

_hip = None
_fortran = None


class NativeLibraryMissing(RuntimeError):
    pass


def build(verbose: bool = False) -> None:
    """Compile every HIP kernel for gfx950 (hipcc cross-compiles without a GPU) and the Fortran host."""
    res = subprocess.run(["make", "-C", _HERE, "-j4"], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise RuntimeError("building fortran_davidson_amd failed")


def hip_lib() -> C.CDLL:
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise NativeLibraryMissing(f"{HIP_LIB} not built: run `make -C fortran_davidson_amd` "
                                       "(there is no CPU fallback)")
        # One HIP runtime per process.  PyTorch ships its own libamdhip64.so.7 and asks the loader for it by
        # file name, so if the system copy is already in the process (brought in by this library) a later
        # `import torch` loads a second runtime, which then reports "No HIP GPUs are available".  With torch
        # imported first both share torch's copy (same soname).  Pure Fortran hosts never meet this.
        try:
            import torch  # noqa: F401
        except ImportError:    # harness without PyTorch: the system runtime is the only one
            pass
        _hip = C.CDLL(HIP_LIB, mode=C.RTLD_LOCAL)
        _hip.dav_last_error.restype = C.c_char_p
        runtimes = hip_runtimes_mapped()
        if len(runtimes) > 1:
            raise RuntimeError("two HIP runtimes are mapped into this process (" + ", ".join(runtimes) + "): the "
                               "engine and PyTorch would each see their own devices - import torch before "
                               "fortran_davidson_amd, or unset LD_LIBRARY_PATH overrides")
        if os.environ.get("DAVIDSON_VERBOSE"):
            print("fortran_davidson_amd: HIP runtime bound:", runtimes[0] if runtimes else "(not found in /proc/self/maps)")
    return _hip


def hip_runtimes_mapped() -> list:
    """Distinct libamdhip64 files mapped into this process (one is the only healthy answer)."""
    found = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(" ", 1)[-1].strip()
                if "libamdhip64" in os.path.basename(path) and os.path.realpath(path) not in found:
                    found.append(os.path.realpath(path))
    except OSError:
        pass
    return found


def fortran_lib() -> C.CDLL:
    global _fortran
    if _fortran is None:
        hip_lib()
        if not os.path.exists(FORTRAN_LIB):
            raise NativeLibraryMissing(f"{FORTRAN_LIB} not built: run `make -C fortran_davidson_amd`")
        _fortran = C.CDLL(FORTRAN_LIB, mode=C.RTLD_GLOBAL)  # MKL dlopens its kernels and needs libmkl_core global
        _fortran.fd_engine_create.restype = C.c_void_p
        _fortran.fd_engine_handle.restype = C.c_void_p
    return _fortran
