// Shared device helpers for the gfx950 (CDNA4) Davidson kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// v_mfma_f64_16x16x4_f64: D(16x16) += A(16x4) * B(4x16), one f64 of A and of B per lane.
//   A operand: lane l holds A[row = l & 15][k = l >> 4]
//   B operand: lane l holds B[k = l >> 4][col = l & 15]
//   C/D:       lane l, reg r (0..3) holds D[row = (l >> 4) + 4 r][col = l & 15]
__device__ __forceinline__ f64x4 mfma_f64(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// "Last workgroup" pattern: every workgroup of a launch writes its partial result to global memory and calls this (all threads,
// convergent); exactly ONE workgroup - the one that arrives last - gets true and may then read ALL partials and finish the
// reduction in whatever fixed order it likes (the result does not depend on who arrives last).  Release / acquire at device
// scope around the ticket: on gfx950 every XCD has its own L2, so the fence before the ticket writes this workgroup's partials
// back and the fence after it invalidates what the last workgroup's CU and L2 may hold of the others'.  The counter must be 0
// before the launch; the last workgroup resets it, so one zeroed word serves every later launch on the same stream.
__device__ __forceinline__ bool dav_last_workgroup(unsigned* counter, unsigned total) {
  __shared__ unsigned dav_ticket;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) dav_ticket = atomicAdd(counter, 1u);
  __syncthreads();
  const bool last = dav_ticket == total - 1;
  if (last) {
    __threadfence();
    if (threadIdx.x == 0) *counter = 0u;
  }
  return last;
}

// splitmix64 counter-based stream shared (bit for bit) with oracle/davidson_oracle.py:uniform01.
__host__ __device__ __forceinline__ uint64_t dav_splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ double dav_uniform01(uint64_t seed, uint64_t lo, uint64_t hi) {
  uint64_t key = (lo << 32) + hi + seed * 0x9E3779B97F4A7C15ull;
  return (double)(dav_splitmix64(key) >> 11) * (1.0 / 9007199254740992.0);
}

// Entry (gi, gj) (0-based global indices) of generate_diagonal_dominant (array_utils.f90:86-113).
__host__ __device__ __forceinline__ double dav_hashed_entry(uint64_t seed, double sparsity, int use_diag,
                                                            double diag_val, int64_t gi, int64_t gj) {
  if (gi == gj) return use_diag ? diag_val : (double)(gi + 1);
  uint64_t lo = (uint64_t)(gi < gj ? gi : gj), hi = (uint64_t)(gi < gj ? gj : gi);
  return dav_uniform01(seed, lo, hi) * sparsity;
}

// Harness operator entry (src/tests/test_utils.f90:38-116): trig(log(sqrt(atan2(e_min, e_max)))) * 1e-4f,
// e_min/e_max = table entries at min(i,j)/max(i,j); A (trig=0, cos): + real(i) on the diagonal;
// B (trig=1, sin): exactly 1 on the diagonal.
// dav_harness_entry_libm: the formula as written - four library calls per entry (DAV_HARNESS_LIBM=1; the A/B path of round 5).
__device__ __forceinline__ double dav_harness_entry_libm(const double* __restrict__ e, int trig, int64_t gi, int64_t gj) {
  const double scale = (double)1e-4f;
  int64_t lo = gi < gj ? gi : gj, hi = gi < gj ? gj : gi;
  double t = log(sqrt(atan2(e[lo], e[hi])));
  if (trig == 0) {
    double v = cos(t) * scale;
    return gi == gj ? v + (double)(float)(gi + 1) : v;
  }
  return gi == gj ? 1.0 : sin(t) * scale;
}
// The same entries as a function of ONE variable (round 6).  0 < e_lo <= e_hi, so atan2(e_lo, e_hi) = atan(exp(log e_lo - log e_hi)):
// with the table l2_i = 2 log e_i (made once, on the host, from the e_i the caller hands over) the off-diagonal entry is
//   scale * trig(0.5 log atan exp((x - 1) / 2)),   x = 1 - |l2_i - l2_j|  in [-1, 1]
// - a function analytic far beyond the interval, evaluated as a polynomial in x by Horner's rule: one subtraction, one addition and
// 17 (cos) / 19 (sin) FMAs instead of atan2 + sqrt + log + cos.  Coefficients: harness_poly.h, generated (and checked against 60-digit
// arithmetic: 0.6 / 1.2 ulp) by tools/gen_harness_poly.py; the literal 1e-4f is folded into them.
#include "harness_poly.h"
template <bool SIN>
__device__ __forceinline__ double dav_harness_poly(double li, double lj) {
  const double x = 1.0 - __builtin_fabs(li - lj);
  if constexpr (!SIN) {
    constexpr double cf[DAV_HARNESS_COS_DEGREE + 1] = {DAV_HARNESS_COS_COEFFS};
    double acc = cf[DAV_HARNESS_COS_DEGREE];
#pragma unroll
    for (int k = DAV_HARNESS_COS_DEGREE - 1; k >= 0; --k) acc = __builtin_fma(acc, x, cf[k]);
    return acc;
  } else {
    constexpr double cf[DAV_HARNESS_SIN_DEGREE + 1] = {DAV_HARNESS_SIN_COEFFS};
    double acc = cf[DAV_HARNESS_SIN_DEGREE];
#pragma unroll
    for (int k = DAV_HARNESS_SIN_DEGREE - 1; k >= 0; --k) acc = __builtin_fma(acc, x, cf[k]);
    return acc;
  }
}
// full entry (diagonal included) from the table of 2 log e_i
__device__ __forceinline__ double dav_harness_entry_poly(const double* __restrict__ l2, int trig, int64_t gi, int64_t gj) {
  if (trig == 0) {
    const double v = dav_harness_poly<false>(l2[gi], l2[gj]);
    return gi == gj ? v + (double)(float)(gi + 1) : v;
  }
  return gi == gj ? 1.0 : dav_harness_poly<true>(l2[gi], l2[gj]);
}

enum { DAV_KIND_NONE = 0, DAV_KIND_DENSE = 1, DAV_KIND_HASHED = 2, DAV_KIND_HARNESS = 3,
       DAV_KIND_IDENTITY = 4, DAV_KIND_HOST = 5, DAV_KIND_DEVICE = 6 };

struct OpParams {          // passed by value to the matrix-free kernels
  int kind;
  uint64_t seed;
  double sparsity;
  int use_diag;
  double diag_val;
  int trig;
  const double* e_table;   // device, n entries: exp(real(i) / real(n)) as the caller evaluated it (single precision)
  const double* l2_table;  // device, roundup(n, 256) + 256 entries: 2 log e_i, zero behind n (harness operator, polynomial form)
  const double* dadd_table; // device, like l2_table: the diagonal ENTRIES of operator A: poly(1) + (double)(float)(i + 1) (src/tests/test_utils.f90:49)
  int libm;                // harness operator: 1 = the four library calls per entry (DAV_HARNESS_LIBM=1), 0 = the polynomial form
};
// entry of the harness operator in whichever form the engine was created with
__device__ __forceinline__ double dav_harness_entry(const OpParams& op, int64_t gi, int64_t gj) {
  return op.libm ? dav_harness_entry_libm(op.e_table, op.trig, gi, gj) : dav_harness_entry_poly(op.l2_table, op.trig, gi, gj);
}
