// K1 - block matvec W = A * X for a row slab of A (replaces the DGEMM 'N','N' of
// src/davidson.f90:131,223 / lapack_wrapper.f90:279-328, and the callbacks of :378-379).
//
// HBM-bound for k <= ~32 columns: A (column-major, rows contiguous) is streamed exactly once with
// 16-byte-per-lane coalesced loads (one wave-load = 32 rows x 4 columns = four 256-byte row runs);
// the block X is pre-packed as Xt[group][row j][16] so that the MFMA B operand of a 4-column step is
// one coalesced 512-byte load that stays in L2; the contraction runs on v_mfma_f64_16x16x4_f64 with
// the accumulators (64 rows x 16*NT columns per wave) in registers for the whole column chunk.
// The column range is split over blockIdx.y (>>256 workgroups); partial tiles go to a slab that a
// second, tiny kernel sums in a fixed order (bitwise reproducible, no fp64 atomics).
#include "kernels.h"
#include <type_traits>
#include <cstdlib>

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_xt_kernel(const double* __restrict__ src, int64_t ld, int64_t nloc,
                                                      int64_t nslab, int k, double* __restrict__ xt,
                                                      int64_t group_stride, int64_t row_off) {
  int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int g = blockIdx.y;
  if (j >= nslab) return;
  double v[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    int col = 16 * g + c;
    v[c] = (col < k && j < nloc) ? src[j + (int64_t)col * ld] : 0.0;
  }
  f64x2* dst = reinterpret_cast<f64x2*>(xt + g * group_stride + (row_off + j) * 16);
#pragma unroll
  for (int c = 0; c < 8; ++c) dst[c] = f64x2{v[2 * c], v[2 * c + 1]};
}

void launch_pack_xt(hipStream_t st, const double* src, int64_t ld, int64_t nloc, int64_t nslab, int k,
                    double* xt, int64_t xt_group_stride, int64_t row_off) {
  int groups = (k + 15) / 16;
  dim3 grid((unsigned)((nslab + 255) / 256), groups);
  hipLaunchKernelGGL(pack_xt_kernel, grid, dim3(256), 0, st, src, ld, nloc, nslab, k, xt, xt_group_stride, row_off);
}

// ---------------------------------------------------------------------------------------------
// One j-iteration = 16 columns of A = 4 MFMA k-steps.  Lane (c = lane & 15, g = lane >> 4):
//   A loads:  rows r0 + 2c, r0 + 2c + 1 (+32 for the second load) of column j + 4u + g
//   B loads:  Xt[group t][j + 4u + g][c]
//   acc[rt][t]: rt = 2*half + parity -> rows r0 + 32*half + 2*(g + 4*reg) + parity, column 16 t + c
template <int NT>
__global__ __launch_bounds__(256) void matvec_dense_kernel(const double* __restrict__ A, int64_t lda,
                                                           int64_t ncols_pad, const double* __restrict__ xt,
                                                           int64_t group_stride, double* __restrict__ slab,
                                                           int64_t nrows_pad, int jc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * MV_ROWS + wave * 64;
  const int s = blockIdx.y;
  const int64_t j0 = (int64_t)s * jc;
  int64_t j1 = j0 + jc;
  if (j1 > ncols_pad) j1 = ncols_pad;

  const double* ap = A + r0 + 2 * c + (j0 + g) * lda;
  const double* xp = xt + (j0 + g) * 16 + c;

  f64x4 acc[4][NT];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[rt][t] = f64x4{0.0, 0.0, 0.0, 0.0};

  for (int64_t j = j0; j < j1; j += 16) {
    f64x2 a[4][2];
    double b[4][NT];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u][0] = *reinterpret_cast<const f64x2*>(ap + (int64_t)(4 * u) * lda);
      a[u][1] = *reinterpret_cast<const f64x2*>(ap + (int64_t)(4 * u) * lda + 32);
#pragma unroll
      for (int t = 0; t < NT; ++t) b[u][t] = xp[t * group_stride + 4 * u * 16];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[0][t] = mfma_f64(a[u][0].x, b[u][t], acc[0][t]);
        acc[1][t] = mfma_f64(a[u][0].y, b[u][t], acc[1][t]);
        acc[2][t] = mfma_f64(a[u][1].x, b[u][t], acc[2][t]);
        acc[3][t] = mfma_f64(a[u][1].y, b[u][t], acc[3][t]);
      }
    }
    ap += 16 * lda;
    xp += 16 * 16;
  }

  double* out = slab + (int64_t)s * (NT * 16) * nrows_pad;
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const int half = rt >> 1, par = rt & 1;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        int64_t row = r0 + 32 * half + 2 * (g + 4 * reg) + par;
        out[(int64_t)(16 * t + c) * nrows_pad + row] = acc[rt][t][reg];
      }
    }
  }
}

// Matrix-free variant: the A entries of the same register tile are generated in place.
template <int NT, int KIND>
__global__ __launch_bounds__(256) void matvec_free_kernel(OpParams op, int64_t row0, int64_t nloc, int64_t n,
                                                          int64_t ncols_pad, const double* __restrict__ xt,
                                                          int64_t group_stride, double* __restrict__ slab,
                                                          int64_t nrows_pad, int jc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * MV_ROWS + wave * 64;
  const int s = blockIdx.y;
  const int64_t j0 = (int64_t)s * jc;
  int64_t j1 = j0 + jc;
  if (j1 > ncols_pad) j1 = ncols_pad;

  f64x4 acc[4][NT];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[rt][t] = f64x4{0.0, 0.0, 0.0, 0.0};

  int64_t li[4] = {r0 + 2 * c, r0 + 2 * c + 1, r0 + 32 + 2 * c, r0 + 32 + 2 * c + 1};
  // MODE 0: every test per entry (bounds, diagonal, which index is the smaller).  MODE 1 / 2 (hashed
  // operator only): the whole column range lies strictly below / above the rows of this workgroup and
  // inside the matrix, so lo/hi are known and the key of uniform01 is one 64-bit add away from a
  // per-row constant - the generator is VALU bound, every instruction removed counts.
  const uint64_t seedmix = op.seed * 0x9E3779B97F4A7C15ull;
  uint64_t kb_below[4], kb_above[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const uint64_t gi = (uint64_t)(row0 + li[rt]);
    kb_below[rt] = gi + seedmix;                 // key = (gj << 32) + gi + seedmix
    kb_above[rt] = (gi << 32) + seedmix;         // key = (gi << 32) + gj + seedmix
  }
  // (m * 2^-53) * sparsity with one rounding, as dav_uniform01 * sparsity gives: m (53 bits) is exactly
  // hi * 2^32 + lo, and 2^-53 * sparsity is an exact scaling of sparsity
  const double scale = op.sparsity * (1.0 / 9007199254740992.0);
  auto u53 = [](uint64_t m) { return __builtin_fma((double)(uint32_t)(m >> 32), 4294967296.0, (double)(uint32_t)m); };
  auto sweep = [&](auto mode_tag, int64_t ja, int64_t jb) {
    constexpr int MODE = decltype(mode_tag)::value;
    const double* xq = xt + (ja + g) * 16 + c;
    for (int64_t j = ja; j < jb; j += 4) {
      const int64_t gj = j + g;
      double a[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        double v = 0.0;
        if (MODE == 1) {
          v = u53(dav_splitmix64(((uint64_t)gj << 32) + kb_below[rt]) >> 11) * scale;
        } else if (MODE == 2) {
          v = u53(dav_splitmix64(kb_above[rt] + (uint64_t)gj) >> 11) * scale;
        } else if (li[rt] < nloc && gj < n) {
          int64_t gi = row0 + li[rt];
          if (KIND == DAV_KIND_HASHED) v = dav_hashed_entry(op.seed, op.sparsity, op.use_diag, op.diag_val, gi, gj);
          else v = dav_harness_entry(op, gi, gj);
        }
        a[rt] = v;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        double b = xq[t * group_stride];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt][t] = mfma_f64(a[rt], b, acc[rt][t]);
      }
      xq += 4 * 16;
    }
  };
  // global rows of this workgroup: [gr0, gr0 + MV_ROWS); fast segments need all of them inside the slab
  const int64_t gr0 = row0 + (int64_t)blockIdx.x * MV_ROWS;
  const bool rows_inside = (int64_t)(blockIdx.x + 1) * MV_ROWS <= nloc;
  if (KIND == DAV_KIND_HASHED && rows_inside) {
    int64_t jn = j1 < n ? j1 : (n / 4 * 4);                             // columns [jn, j1) may cross the matrix edge
    if (jn < j0) jn = j0;                                               // a column chunk that lies wholly in the padding
    int64_t jlo = gr0 / 4 * 4;                                          // first column step that can touch the diagonal band
    int64_t jhi = (gr0 + MV_ROWS + 3) / 4 * 4;                          // first column step strictly above it
    jlo = jlo < j0 ? j0 : (jlo > jn ? jn : jlo);
    jhi = jhi < jlo ? jlo : (jhi > jn ? jn : jhi);
    sweep(std::integral_constant<int, 1>{}, j0, jlo);                   // gj < every gi
    sweep(std::integral_constant<int, 0>{}, jlo, jhi);                  // diagonal band
    sweep(std::integral_constant<int, 2>{}, jhi, jn);                   // gj > every gi
    sweep(std::integral_constant<int, 0>{}, jn, j1);                    // ragged edge
  } else {
    sweep(std::integral_constant<int, 0>{}, j0, j1);
  }

  double* out = slab + (int64_t)s * (NT * 16) * nrows_pad;
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const int half = rt >> 1, par = rt & 1;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        int64_t row = r0 + 32 * half + 2 * (g + 4 * reg) + par;
        out[(int64_t)(16 * t + c) * nrows_pad + row] = acc[rt][t][reg];
      }
    }
  }
}

// target > 0: aim at that many workgroups; forced_nsplit > 0: exact split count (Tune::mv_target / mv_nsplit, tuning runs)
void matvec_plan(int64_t nrows_pad, int64_t ncols_pad, int ngroups, int* nsplit, int* jc, int64_t target, int64_t forced_nsplit) {
  int64_t rowblocks = nrows_pad / MV_ROWS;
  // Measured on MI355X (N=20000, 79 row blocks, split count swept 12..64): the sweep is fastest when the
  // grid just fills the resident capacity once (256 CUs x 5 workgroups = 1280): 16 splits = 1264
  // workgroups gives 5.6 TB/s, 17 splits (1343, a second sparsely filled round) 5.37, 26 splits 5.35.
  // So: the largest split count whose grid does not exceed one full round; for tall matrices (few
  // splits possible) fall back to ~1024 workgroups.
  int64_t s;
  if (target > 0) {
    s = (target + rowblocks - 1) / rowblocks;
  } else {
    s = 1280 / rowblocks;
    if (s < 1 || rowblocks * s < 900) s = (1024 + rowblocks - 1) / rowblocks;
  }
  if (forced_nsplit > 0) s = forced_nsplit;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  int64_t chunk = (ncols_pad + s - 1) / s;
  chunk = (chunk + 15) / 16 * 16;
  if (chunk < 64) chunk = 64;
  *jc = (int)chunk;
  *nsplit = (int)((ncols_pad + chunk - 1) / chunk);
  (void)ngroups;
}

size_t matvec_slab_doubles(int64_t nrows_pad, int ngroups, int nsplit) {
  return (size_t)nsplit * (size_t)ngroups * 16 * (size_t)nrows_pad;
}

void launch_matvec_dense(hipStream_t st, const double* A, int64_t lda, int64_t nrows_pad, int64_t ncols_pad,
                         const double* xt, int64_t xt_group_stride, int ngroups, double* slab, int nsplit, int jc) {
  dim3 grid((unsigned)(nrows_pad / MV_ROWS), nsplit);
  switch (ngroups) {
    case 1: hipLaunchKernelGGL(matvec_dense_kernel<1>, grid, dim3(256), 0, st, A, lda, ncols_pad, xt, xt_group_stride, slab, nrows_pad, jc); break;
    case 2: hipLaunchKernelGGL(matvec_dense_kernel<2>, grid, dim3(256), 0, st, A, lda, ncols_pad, xt, xt_group_stride, slab, nrows_pad, jc); break;
    default: hipLaunchKernelGGL(matvec_dense_kernel<4>, grid, dim3(256), 0, st, A, lda, ncols_pad, xt, xt_group_stride, slab, nrows_pad, jc); break;
  }
}

template <int KIND>
static void launch_free_kind(hipStream_t st, OpParams op, int64_t row0, int64_t nloc, int64_t n, int64_t nrows_pad,
                             int64_t ncols_pad, const double* xt, int64_t xt_group_stride, int ngroups, double* slab,
                             int nsplit, int jc) {
  dim3 grid((unsigned)(nrows_pad / MV_ROWS), nsplit);
  switch (ngroups) {
    case 1: hipLaunchKernelGGL((matvec_free_kernel<1, KIND>), grid, dim3(256), 0, st, op, row0, nloc, n, ncols_pad, xt, xt_group_stride, slab, nrows_pad, jc); break;
    case 2: hipLaunchKernelGGL((matvec_free_kernel<2, KIND>), grid, dim3(256), 0, st, op, row0, nloc, n, ncols_pad, xt, xt_group_stride, slab, nrows_pad, jc); break;
    default: hipLaunchKernelGGL((matvec_free_kernel<4, KIND>), grid, dim3(256), 0, st, op, row0, nloc, n, ncols_pad, xt, xt_group_stride, slab, nrows_pad, jc); break;
  }
}

void launch_matvec_free(hipStream_t st, OpParams op, int64_t row0, int64_t nloc, int64_t n, int64_t nrows_pad,
                        int64_t ncols_pad, const double* xt, int64_t xt_group_stride, int ngroups, double* slab,
                        int nsplit, int jc) {
  if (op.kind == DAV_KIND_HASHED)
    launch_free_kind<DAV_KIND_HASHED>(st, op, row0, nloc, n, nrows_pad, ncols_pad, xt, xt_group_stride, ngroups, slab, nsplit, jc);
  else
    launch_free_kind<DAV_KIND_HARNESS>(st, op, row0, nloc, n, nrows_pad, ncols_pad, xt, xt_group_stride, ngroups, slab, nsplit, jc);
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void slab_reduce_kernel(const double* __restrict__ slab, int nsplit,
                                                          int64_t nrows_pad, int ncol16, int64_t nloc, int k,
                                                          double* __restrict__ dst, int64_t ldd) {
  int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  int col = blockIdx.y;
  if (i >= nrows_pad || col >= k) return;
  f64x2 sum = {0.0, 0.0};
  const double* p = slab + (int64_t)col * nrows_pad + i;
  const int64_t stride = (int64_t)ncol16 * nrows_pad;
  // four interleaved partial sums in a fixed order: four 16-byte loads in flight per thread
  f64x2 part[4] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
  int s = 0;
  for (; s + 4 <= nsplit; s += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      f64x2 v = *reinterpret_cast<const f64x2*>(p + (s + u) * stride);
      part[u].x += v.x;
      part[u].y += v.y;
    }
  }
  for (int u = 0; s < nsplit; ++s, ++u) {
    f64x2 v = *reinterpret_cast<const f64x2*>(p + s * stride);
    part[u].x += v.x;
    part[u].y += v.y;
  }
  sum.x = (part[0].x + part[1].x) + (part[2].x + part[3].x);
  sum.y = (part[0].y + part[1].y) + (part[2].y + part[3].y);
  if (i >= nloc) sum.x = 0.0;
  if (i + 1 >= nloc) sum.y = 0.0;
  *reinterpret_cast<f64x2*>(dst + (int64_t)col * ldd + i) = sum;
}

void launch_slab_reduce(hipStream_t st, const double* slab, int nsplit, int64_t nrows_pad, int ngroups,
                        int64_t nloc, int k, double* dst, int64_t ldd) {
  dim3 grid((unsigned)((nrows_pad / 2 + 255) / 256), k);
  hipLaunchKernelGGL(slab_reduce_kernel, grid, dim3(256), 0, st, slab, nsplit, nrows_pad, ngroups * 16, nloc, k, dst, ldd);
}
