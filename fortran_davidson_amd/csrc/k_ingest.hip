// Ingest: a block of complete matrix rows, staged ROW-major in device memory (the order of the
// reference's on-disk format, read_matrix / write_matrix in src/tests/test_utils.f90:118-135,150-166),
// scattered into the engine's resident layout - the padded column-major row slab of this rank, or the
// lower block triangle of 256 x 256 tiles.  A 32 x 32 transposing copy through LDS keeps both the reads
// (along a row) and the writes (along a column) coalesced.  HBM-bound byte work: 16 B per element.
#include "kernels.h"

template <int SYM>
__global__ __launch_bounds__(256) void rows_scatter_kernel(const double* __restrict__ stage, int64_t ldr, int64_t grow0,
                                                           int64_t nrows, int64_t n, double* __restrict__ dst, int64_t lda,
                                                           int64_t slab_row0, int64_t slab_rows, const int64_t* __restrict__ row_off) {
  __shared__ double t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int64_t r = r0 + ty + 8 * s, c = c0 + tx;
    t[ty + 8 * s][tx] = (r < nrows && c < n) ? stage[r * ldr + c] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int64_t r = r0 + tx, c = c0 + ty + 8 * s;      // tx now runs along the rows
    if (r >= nrows || c >= n) continue;
    const int64_t gi = grow0 + r;                         // global row
    const double v = t[tx][ty + 8 * s];
    if (SYM) {
      const int64_t I = gi / SYM_TB, J = c / SYM_TB;
      if (J > I || row_off[I] < 0) continue;              // upper block triangle / block row of another rank: not stored here
      dst[(row_off[I] + J) * (int64_t)(SYM_TB * SYM_TB) + (c % SYM_TB) * SYM_TB + (gi % SYM_TB)] = v;
    } else {
      const int64_t li = gi - slab_row0;
      if (li < 0 || li >= slab_rows) continue;            // row of another rank
      dst[li + c * lda] = v;
    }
  }
}

void launch_rows_scatter(hipStream_t st, const double* stage, int64_t ldr, int64_t grow0, int64_t nrows, int64_t n,
                         double* dst, int64_t lda, int64_t slab_row0, int64_t slab_rows, int sym, const int64_t* row_off) {
  if (nrows <= 0 || n <= 0) return;
  dim3 grid((unsigned)((n + 31) / 32), (unsigned)((nrows + 31) / 32));
  if (sym)
    hipLaunchKernelGGL(rows_scatter_kernel<1>, grid, dim3(256), 0, st, stage, ldr, grow0, nrows, n, dst, lda, slab_row0, slab_rows, row_off);
  else
    hipLaunchKernelGGL(rows_scatter_kernel<0>, grid, dim3(256), 0, st, stage, ldr, grow0, nrows, n, dst, lda, slab_row0, slab_rows, row_off);
}
