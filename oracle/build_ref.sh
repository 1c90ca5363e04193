#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.
#
# Builds the reference itself (NLESC-JCER/Fortran_Davidson, read from /root/reference/src where it
# lies) into oracle/_ref/libref_davidson.so together with OUR bind(C) driver oracle/ref_driver.f90.
# Nothing of the reference is copied into the repository: sources are compiled in place, module
# files and objects go to a mktemp scratch dir, only the .so lands in oracle/_ref/ (git-ignored).
#
# Toolchain: AMD flang (ROCm 7.2) + MKL (libmkl_rt) - both part of this image, nothing is stubbed.
#
# One deviation, stated openly: flang enforces the F2008 rule that a separate
# module procedure's characteristics match its interface body; davidson.f90:43 declares the result
# extent as size(ritz_vectors, 2), davidson.f90:649 as size(residues, 2).  Both are size(V,2) at the
# only call site (davidson.f90:199-205).  gfortran accepts this, flang rejects it, so line 649 is
# rewritten ON THE FLY in a pipe (sed | flang -x f95 -); no patched copy of the file is ever stored.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${REFERENCE_ROOT:-/root/reference}/src"
OUT="$HERE/_ref"
FC="${FC:-/opt/rocm/lib/llvm/bin/flang}"
MKL_DIR="${MKL_DIR:-/opt/conda/lib}"

if [ ! -d "$REF" ]; then
  echo "build_ref: $REF not present (expected on the GPU box) - keeping prebuilt oracle/_ref" >&2
  exit 0
fi
if [ ! -x "$FC" ] || [ ! -e "$MKL_DIR/libmkl_rt.so" ]; then
  echo "build_ref: flang or MKL missing - reference is unbuildable here" >&2
  exit 3
fi

TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$OUT"
FFLAGS="-O2 -fPIC -fopenmp -module-dir $TMP -I$TMP"

$FC $FFLAGS -c "$REF/numeric_kinds.f90"  -o "$TMP/numeric_kinds.o"
$FC $FFLAGS -c "$REF/lapack_wrapper.f90" -o "$TMP/lapack_wrapper.o"
$FC $FFLAGS -c "$REF/array_utils.f90"    -o "$TMP/array_utils.o"
sed '649s/size(residues, 2)/size(ritz_vectors, 2)/' "$REF/davidson.f90" \
  | $FC $FFLAGS -c -x f95 - -o "$TMP/davidson.o"
$FC $FFLAGS -c "$REF/tests/test_utils.f90" -o "$TMP/test_utils.o"
$FC $FFLAGS -c "$HERE/ref_driver.f90"      -o "$TMP/ref_driver.o"

$FC -shared -Wl,-Bsymbolic -fopenmp=libiomp5 -o "$OUT/libref_davidson.so" \
  "$TMP/ref_driver.o" "$TMP/test_utils.o" "$TMP/davidson.o" "$TMP/array_utils.o" \
  "$TMP/lapack_wrapper.o" "$TMP/numeric_kinds.o" \
  -L"$MKL_DIR" -lmkl_rt -Wl,-rpath,"$MKL_DIR"
echo "build_ref: wrote $OUT/libref_davidson.so"

# ---- the reference's own test programs, UNCHANGED, linked against OUR modules/libraries ---------------
# (drop-in proof: they run on the MI355X engine in `pytest -m gpu`, see tests/test_reference_programs_gpu.py)
OURS="$HERE/../fortran_davidson_amd"
if [ -e "$OURS/lib/libfortran_davidson_amd.so" ] && [ -d "$OURS/fortran/build" ]; then
  mkdir -p "$OUT/ref_tests"
  T2="$(mktemp -d)"
  for prog in test_dense_properties test_free_properties test_dense_numpy test_free_numpy test_call_lapack; do
    $FC -O1 -fopenmp=libiomp5 -I"$OURS/fortran/build" -module-dir "$T2" "$REF/tests/test_utils.f90" "$REF/tests/$prog.f90" \
      -L"$OURS/lib" -lfortran_davidson_amd -ldavidson_hip -Wl,-rpath,'$ORIGIN/../../../fortran_davidson_amd/lib' \
      -L"$MKL_DIR" -Wl,-rpath,"$MKL_DIR" -o "$OUT/ref_tests/$prog"
  done
  # the two demo drivers (each carries its own copy of the helper module)
  for prog in main benchmark_free; do
    T3="$(mktemp -d)"
    $FC -O1 -fopenmp=libiomp5 -I"$OURS/fortran/build" -module-dir "$T3" "$REF/$prog.f90" \
      -L"$OURS/lib" -lfortran_davidson_amd -ldavidson_hip -Wl,-rpath,'$ORIGIN/../../../fortran_davidson_amd/lib' \
      -L"$MKL_DIR" -Wl,-rpath,"$MKL_DIR" -o "$OUT/ref_tests/$prog"
    rm -rf "$T3"
  done
  rm -rf "$T2"
  echo "build_ref: wrote $OUT/ref_tests/ (reference test programs on our libraries)"
fi
