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

// Harness operator entry (tests/test_utils.f90:38-116): trig(log(sqrt(atan2(e_min, e_max)))) * 1e-4f,
// e_min/e_max = table entries at min(i,j)/max(i,j); A (trig=0, cos): + real(i) on the diagonal;
// B (trig=1, sin): exactly 1 on the diagonal.
__device__ __forceinline__ double dav_harness_entry(const double* __restrict__ e, int trig, int64_t gi, int64_t gj) {
  const double scale = (double)1e-4f;
  int64_t lo = gi < gj ? gi : gj, hi = gi < gj ? gj : gi;
  double t = log(sqrt(atan2(e[lo], e[hi])));
  if (trig == 0) {
    double v = cos(t) * scale;
    return gi == gj ? v + (double)(float)(gi + 1) : v;
  }
  return gi == gj ? 1.0 : sin(t) * scale;
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
  const double* e_table;   // device, n entries
};
