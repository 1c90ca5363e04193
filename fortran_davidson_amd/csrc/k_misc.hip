// K6 setup kernels and small utilities: input generator (semantics of generate_diagonal_dominant,
// src/array_utils.f90:86-113), diagonal extraction (array_utils.f90:115-134 without the O(N^2) loop),
// unit-vector basis (array_utils.f90:136-160) and its image A*V0 as a column gather.
#include "kernels.h"

__global__ __launch_bounds__(256) void generate_dense_tile_kernel(double* __restrict__ A, int64_t lda, int64_t nrows_pad,
                                                                  int64_t row0, int64_t nloc, int64_t n, int64_t col0,
                                                                  uint64_t seed, double sparsity, int use_diag,
                                                                  double diag_val) {
  int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  int64_t j = col0 + blockIdx.y;
  if (i >= nrows_pad) return;
  f64x2 v = {0.0, 0.0};
  if (j < n) {
    if (i < nloc) v.x = dav_hashed_entry(seed, sparsity, use_diag, diag_val, row0 + i, j);
    if (i + 1 < nloc) v.y = dav_hashed_entry(seed, sparsity, use_diag, diag_val, row0 + i + 1, j);
  }
  *reinterpret_cast<f64x2*>(A + j * lda + i) = v;
}

void launch_generate_dense(hipStream_t st, double* A, int64_t lda, int64_t nrows_pad, int64_t ncols_pad, int64_t row0,
                           int64_t nloc, int64_t n, uint64_t seed, double sparsity, int use_diag, double diag_val) {
  const int64_t ymax = 32768;   // grid.y limit: tile the columns
  for (int64_t jb = 0; jb < ncols_pad; jb += ymax) {
    int64_t ny = ncols_pad - jb < ymax ? ncols_pad - jb : ymax;
    dim3 grid((unsigned)((nrows_pad / 2 + 255) / 256), (unsigned)ny);
    hipLaunchKernelGGL(generate_dense_tile_kernel, grid, dim3(256), 0, st, A, lda, nrows_pad, row0, nloc, n, jb, seed,
                       sparsity, use_diag, diag_val);
  }
}

__global__ void diag_dense_kernel(const double* __restrict__ A, int64_t lda, int64_t row0, int64_t nloc,
                                  double* __restrict__ diag) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < nloc) diag[i] = A[(row0 + i) * lda + i];
}
void launch_diag_dense(hipStream_t st, const double* A, int64_t lda, int64_t row0, int64_t nloc, double* diag) {
  if (nloc <= 0) return;
  hipLaunchKernelGGL(diag_dense_kernel, dim3((unsigned)((nloc + 255) / 256)), dim3(256), 0, st, A, lda, row0, nloc, diag);
}

__global__ void diag_free_kernel(OpParams op, int64_t row0, int64_t nloc, double* __restrict__ diag) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nloc) return;
  int64_t gi = row0 + i;
  double v;
  if (op.kind == DAV_KIND_HASHED) v = dav_hashed_entry(op.seed, op.sparsity, op.use_diag, op.diag_val, gi, gi);
  else if (op.kind == DAV_KIND_HARNESS) v = dav_harness_entry(op, gi, gi);
  else v = 1.0;
  diag[i] = v;
}
void launch_diag_free(hipStream_t st, OpParams op, int64_t row0, int64_t nloc, double* diag) {
  if (nloc <= 0) return;
  hipLaunchKernelGGL(diag_free_kernel, dim3((unsigned)((nloc + 255) / 256)), dim3(256), 0, st, op, row0, nloc, diag);
}

__global__ void gather_columns_kernel(const double* __restrict__ A, int64_t lda, int64_t nrows_pad,
                                      const int64_t* __restrict__ idx, double* __restrict__ dst, int64_t ldd) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int c = blockIdx.y;
  if (i < nrows_pad) dst[(int64_t)c * ldd + i] = A[idx[c] * lda + i];
}
void launch_gather_columns(hipStream_t st, const double* A, int64_t lda, int64_t nrows_pad, const int64_t* idx_dev,
                           int k, double* dst, int64_t ldd) {
  hipLaunchKernelGGL(gather_columns_kernel, dim3((unsigned)((nrows_pad + 255) / 256), k), dim3(256), 0, st, A, lda,
                     nrows_pad, idx_dev, dst, ldd);
}

// columns idx[0:k] of a generated operator (rows of this rank's slab): what Op * e_p is, without a sweep
__global__ void gather_columns_free_kernel(OpParams op, int64_t row0, int64_t nloc, int64_t nrows_pad, const int64_t* __restrict__ idx,
                                           double* __restrict__ dst, int64_t ldd) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int c = blockIdx.y;
  if (i >= nrows_pad) return;
  double v = 0.0;
  if (i < nloc) {
    const int64_t gi = row0 + i, gj = idx[c];
    if (op.kind == DAV_KIND_HASHED) v = dav_hashed_entry(op.seed, op.sparsity, op.use_diag, op.diag_val, gi, gj);
    else if (op.kind == DAV_KIND_HARNESS) v = dav_harness_entry(op, gi, gj);
    else v = gi == gj ? 1.0 : 0.0;
  }
  dst[(int64_t)c * ldd + i] = v;
}
void launch_gather_columns_free(hipStream_t st, OpParams op, int64_t row0, int64_t nloc, int64_t nrows_pad, const int64_t* idx_dev, int k,
                                double* dst, int64_t ldd) {
  hipLaunchKernelGGL(gather_columns_free_kernel, dim3((unsigned)((nrows_pad + 255) / 256), k), dim3(256), 0, st, op, row0, nloc, nrows_pad,
                     idx_dev, dst, ldd);
}

// V0^T (Op V0) for unit columns V0 = e_idx is the operator's entries: h0[i + j * k] = Op(idx[i], idx[j]).  Generated operator: every
// rank generates all of them; full rows of ONE rank: read from the matrix.
__global__ void entries_free_kernel(OpParams op, const int64_t* __restrict__ idx, int k, double* __restrict__ h0) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= k * k) return;
  const int64_t gi = idx[t % k], gj = idx[t / k];
  double v;
  if (op.kind == DAV_KIND_HASHED) v = dav_hashed_entry(op.seed, op.sparsity, op.use_diag, op.diag_val, gi, gj);
  else if (op.kind == DAV_KIND_HARNESS) v = dav_harness_entry(op, gi, gj);
  else v = gi == gj ? 1.0 : 0.0;
  h0[t] = v;
}
void launch_entries_free(hipStream_t st, OpParams op, const int64_t* idx_dev, int k, double* h0) {
  hipLaunchKernelGGL(entries_free_kernel, dim3((unsigned)((k * k + 255) / 256)), dim3(256), 0, st, op, idx_dev, k, h0);
}
__global__ void entries_dense_kernel(const double* __restrict__ A, int64_t lda, const int64_t* __restrict__ idx, int k, double* __restrict__ h0) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < k * k) h0[t] = A[idx[t / k] * lda + idx[t % k]];
}
void launch_entries_dense(hipStream_t st, const double* A, int64_t lda, const int64_t* idx_dev, int k, double* h0) {
  hipLaunchKernelGGL(entries_dense_kernel, dim3((unsigned)((k * k + 255) / 256)), dim3(256), 0, st, A, lda, idx_dev, k, h0);
}

__global__ void zero_pad_rows_kernel(double* __restrict__ dst, int64_t ldd, int64_t nloc, int64_t nrows_pad) {
  const int64_t i = nloc + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < nrows_pad) dst[(int64_t)blockIdx.y * ldd + i] = 0.0;
}
void launch_zero_pad_rows(hipStream_t st, double* dst, int64_t ldd, int64_t nloc, int64_t nrows_pad, int k) {
  if (nrows_pad <= nloc || k <= 0) return;
  hipLaunchKernelGGL(zero_pad_rows_kernel, dim3((unsigned)((nrows_pad - nloc + 255) / 256), k), dim3(256), 0, st, dst, ldd, nloc, nrows_pad);
}

__global__ void unit_columns_kernel(const int64_t* __restrict__ idx, int64_t row0, int64_t nloc, int64_t nrows_pad,
                                    double* __restrict__ dst, int64_t ldd) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int c = blockIdx.y;
  if (i < nrows_pad) dst[(int64_t)c * ldd + i] = (i < nloc && row0 + i == idx[c]) ? 1.0 : 0.0;
}
void launch_unit_columns(hipStream_t st, const int64_t* idx_dev, int k, int64_t row0, int64_t nloc, int64_t nrows_pad,
                         double* dst, int64_t ldd) {
  hipLaunchKernelGGL(unit_columns_kernel, dim3((unsigned)((nrows_pad + 255) / 256), k), dim3(256), 0, st, idx_dev, row0,
                     nloc, nrows_pad, dst, ldd);
}

__global__ void copy_columns_kernel(const double* __restrict__ src, int64_t lds, double* __restrict__ dst, int64_t ldd,
                                    int64_t nrows_pad) {
  int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  int c = blockIdx.y;
  if (i < nrows_pad)
    *reinterpret_cast<f64x2*>(dst + (int64_t)c * ldd + i) = *reinterpret_cast<const f64x2*>(src + (int64_t)c * lds + i);
}
void launch_copy_columns(hipStream_t st, const double* src, int64_t lds, double* dst, int64_t ldd, int64_t nrows_pad, int k) {
  if (k <= 0) return;
  hipLaunchKernelGGL(copy_columns_kernel, dim3((unsigned)((nrows_pad / 2 + 255) / 256), k), dim3(256), 0, st, src, lds,
                     dst, ldd, nrows_pad);
}

// dst[i, c] = i < nloc ? src[c * lds + i] : 0 for i < nrows_pad: the chunk a reduce-scatter delivered (columns of nslab
// contiguous rows) becomes panel columns with their zero padding
// (accumulate: added to what dst holds - the second part of an operator that is swept in two parts)
__global__ __launch_bounds__(256) void chunk_to_panel_kernel(const double* __restrict__ src, int64_t lds, int64_t nloc, int64_t nrows_pad,
                                                             double* __restrict__ dst, int64_t ldd, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (i < nrows_pad) dst[(int64_t)c * ldd + i] = i < nloc ? (accumulate ? dst[(int64_t)c * ldd + i] : 0.0) + src[(int64_t)c * lds + i] : 0.0;
}
void launch_chunk_to_panel(hipStream_t st, const double* src, int64_t lds, int64_t nloc, int64_t nrows_pad, int k, double* dst, int64_t ldd,
                           bool accumulate) {
  if (k <= 0) return;
  hipLaunchKernelGGL(chunk_to_panel_kernel, dim3((unsigned)((nrows_pad + 255) / 256), k), dim3(256), 0, st, src, lds, nloc, nrows_pad,
                     dst, ldd, accumulate ? 1 : 0);
}

// ---- stream microbenchmark (dav_bench_stream): what the HBM delivers to plain streaming kernels on this box ------------------
// 16 bytes per lane, four independent accesses in flight per lane, grid-stride over 2048 workgroups (Guideline 11 / 13 of the
// CDNA4 guide); MODE 0: copy a = b (one read, one write), MODE 1: triad a = b + s c (two reads, one write)
template <int MODE>
__global__ __launch_bounds__(256) void stream_kernel(double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ c,
                                                     double s, int64_t n2) {
  // a workgroup moves contiguous 32 KiB pieces (256 lanes x 16 B x 8 accesses in flight per lane), grid-stride over the pieces
  constexpr int U = 8;
  const int64_t piece = 256 * U, stride = (int64_t)gridDim.x * piece;
  int64_t i = (int64_t)blockIdx.x * piece + threadIdx.x;
  for (; i + (U - 1) * 256 < n2; i += stride) {
    f64x2 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(b) + i + u * 256);
    if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < U; ++u) y[u] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(c) + i + u * 256);
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = x[u] + s * y[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(x[u], reinterpret_cast<f64x2*>(a) + i + u * 256);
  }
  for (int u = 0; u < U; ++u) {
    const int64_t j = i + u * 256;
    if (j < n2 && j >= (n2 / piece) * piece) {                               // the last, partial piece
      f64x2 x = reinterpret_cast<const f64x2*>(b)[j];
      if (MODE == 1) x = x + s * reinterpret_cast<const f64x2*>(c)[j];
      reinterpret_cast<f64x2*>(a)[j] = x;
    }
  }
}
// MODE 2 of the microbenchmark: reads only (b and c, 16 B per lane, the same access pattern), one partial sum per workgroup written
// to a[blockIdx.x] - the roof of a kernel that, like the operator sweeps, reads far more than it writes
__global__ __launch_bounds__(256) void stream_read_kernel(double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ c,
                                                          int64_t n2) {
  constexpr int U = 8;
  const int64_t piece = 256 * U, stride = (int64_t)gridDim.x * piece;
  f64x2 acc = {0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * piece + threadIdx.x; i + (U - 1) * 256 < n2; i += stride) {
    f64x2 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(b) + i + u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u) y[u] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(c) + i + u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u) acc = acc + x[u] + y[u];
  }
  __shared__ double part[256];
  part[threadIdx.x] = acc[0] + acc[1];
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int l = 0; l < 256; ++l) t += part[l];
    a[blockIdx.x] = t;
  }
}
void launch_stream(hipStream_t st, int mode, double* a, const double* b, const double* c, double s, int64_t n) {
  if (mode == 2) { hipLaunchKernelGGL(stream_read_kernel, dim3(2048), dim3(256), 0, st, a, b, c, n / 2); return; }
  if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(2048), dim3(256), 0, st, a, b, c, s, n / 2);
  else hipLaunchKernelGGL(stream_kernel<1>, dim3(2048), dim3(256), 0, st, a, b, c, s, n / 2);
}

// What the chip delivers of the matrix-free TEST operator's arithmetic (src/tests/test_utils.f90:72-116: one fp64 atan2, sqrt, log
// and cos / sin per matrix entry): every lane evaluates `iters` entries of dav_harness_entry's chain on register operands in
// [1, e] (the range of the exp table), no memory traffic - the roof of the generated sweeps of that operator.  8 waves per
// workgroup, `wgs` workgroups (2 x 256 CUs: two waves per SIMD, like the sweep kernel).
__global__ __launch_bounds__(512) void harness_rate_kernel(double* __restrict__ out, int iters, double x0, double dx) {
  const int lane = threadIdx.x + blockIdx.x * 512;
  double x = x0 + 1e-9 * (lane & 1023), y = 2.0 - 1e-9 * (lane & 511), acc = 0.0;
  for (int i = 0; i < iters; ++i) {
    acc += cos(log(sqrt(atan2(x, y)))) * (double)1e-4f;
    x += dx;
    y += 0.5 * dx;
  }
  out[lane] = acc;
}
void launch_harness_rate(hipStream_t st, double* out, int wgs, int iters) {
  hipLaunchKernelGGL(harness_rate_kernel, dim3(wgs), dim3(512), 0, st, out, iters, 1.0, 1e-7);
}

// Fixed-order sum of the chunks of a direct reduce-scatter (DAV_COLL_DIRECT): rank order, the rank's own chunk read where it lies.
__global__ __launch_bounds__(256) void sum_parts_kernel(const double* __restrict__ own, const double* __restrict__ stage, int nparts, int self,
                                                        size_t count, double* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    double s = 0.0;
    for (int p = 0; p < nparts; ++p) s += p == self ? own[i] : stage[(size_t)(p < self ? p : p - 1) * count + i];
    out[i] = s;
  }
}
void launch_sum_parts(hipStream_t st, const double* own, const double* stage, int nparts, int self, size_t count, double* out) {
  const unsigned grid = (unsigned)std::min<size_t>(4096, (count + 255) / 256);
  hipLaunchKernelGGL(sum_parts_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, own, stage, nparts, self, count, out);
}

// Trial of the collective paths (engine_apply.hip: coll_path_trial): column j of two blocks compared - out[3 j] = entries whose bits
// differ, out[3 j + 1] = max |a - b|, out[3 j + 2] = max |a| - one workgroup per column, fixed-order tree in LDS.
__global__ __launch_bounds__(256) void compare_blocks_kernel(const double* __restrict__ a, const double* __restrict__ b, int64_t ld, int64_t rows,
                                                             double* __restrict__ out) {
  __shared__ double sh[3][256];
  const double* ac = a + (int64_t)blockIdx.x * ld;
  const double* bc = b + (int64_t)blockIdx.x * ld;
  double differ = 0.0, maxdiff = 0.0, maxabs = 0.0;
  for (int64_t i = threadIdx.x; i < rows; i += 256) {
    const double x = ac[i], y = bc[i];
    if (__double_as_longlong(x) != __double_as_longlong(y)) differ += 1.0;
    const double d = fabs(x - y);
    maxdiff = !(d <= maxdiff) ? d : maxdiff;         // a NaN difference wins: the comparison fails
    maxabs = fmax(maxabs, fabs(x));
  }
  sh[0][threadIdx.x] = differ; sh[1][threadIdx.x] = maxdiff; sh[2][threadIdx.x] = maxabs;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
      const double d = sh[1][threadIdx.x + s];
      sh[1][threadIdx.x] = !(d <= sh[1][threadIdx.x]) ? d : sh[1][threadIdx.x];
      sh[2][threadIdx.x] = fmax(sh[2][threadIdx.x], sh[2][threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[3 * blockIdx.x] = sh[0][0]; out[3 * blockIdx.x + 1] = sh[1][0]; out[3 * blockIdx.x + 2] = sh[2][0]; }
}
void launch_compare_blocks(hipStream_t st, const double* a, const double* b, int64_t ld, int64_t rows, int k, double* out) {
  hipLaunchKernelGGL(compare_blocks_kernel, dim3(k), dim3(256), 0, st, a, b, ld, rows, out);
}
// test hook of that trial (DAV_COLL_TRIAL_CORRUPT): p[0] += delta
__global__ void poke_kernel(double* p, double delta) { p[0] += delta; }
void launch_poke(hipStream_t st, double* p, double delta) { hipLaunchKernelGGL(poke_kernel, dim3(1), dim3(1), 0, st, p, delta); }
