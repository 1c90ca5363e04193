// K1s - block matvec on SYMMETRIC-TILED storage: only the lower block triangle of A (tiles of
// 256 x 256, each contiguous, diagonal tiles stored in full) lives in HBM - N(N+1)/2 instead of N^2
// entries, which is what makes N = 200000 (160 GB) fit one MI355X - and every off-diagonal tile is
// used twice per sweep:   W_I += A_IJ X_J   (direct)   and   W_J += A_IJ^T X_I   (transposed).
//
// Workgroup = 4 waves, one run of tiles (I, J0..J1) of block row I.  Wave w owns rows 64w..64w+63 of
// the block row.  Per 16-column step it loads its 64 x 16 sub-block twice, microseconds apart (second
// read is an L1/L2 hit): once in the "direct" lane layout (16 B = 2 rows of one column per lane; MFMA
// contraction over columns, accumulators 64 rows x 16 NT stay in registers for the whole run) and once
// in the "Gram" lane layout (4 consecutive rows of one column per lane; MFMA contraction over rows).
// The transposed partials of the four waves are summed through LDS once per 64 columns and written
// to a per-tile slab; direct partials go to a per-run slab; a second kernel adds, in fixed order, the
// slabs that belong to each output block (bitwise reproducible, no fp64 atomics).
#include "kernels.h"

// tile (I, J), J <= I, at tiles + (I (I+1)/2 + J) * TB*TB, column-major with leading dimension TB
__device__ __forceinline__ const double* sym_tile(const double* tiles, int I, int J) {
  return tiles + ((int64_t)I * (I + 1) / 2 + J) * (int64_t)(SYM_TB * SYM_TB);
}

template <int NT>
__global__ __launch_bounds__(256, 2) void matvec_sym_kernel(const double* __restrict__ tiles, const int* __restrict__ items,
                                                         const double* __restrict__ xt, int64_t group_stride,
                                                         double* __restrict__ slabD, double* __restrict__ slabT) {
  constexpr int RS = 65;                        // padded stride: block columns land on different LDS banks
  constexpr int TRS = 66;                       // padded column stride of the transposition scratch (528 B)
  // one LDS region, two uses that never overlap in time (barriers below): per-wave transposition scratch
  // tr[wave][16 columns][TRS] during the steps, Z partials red[wave][16 NT block columns][RS] at batch end
  constexpr int REGION = (RS * 16 * NT > 16 * TRS) ? RS * 16 * NT : 16 * TRS;
  __shared__ __attribute__((aligned(16))) double scratch[4][REGION];
  double (*red)[REGION] = scratch;
  double (*tr)[REGION] = scratch;
  constexpr int XS = 17;                        // padded row stride of the X_I copy
  __shared__ double xs[NT][SYM_TB * XS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int I = items[3 * blockIdx.x], J0 = items[3 * blockIdx.x + 1], J1 = items[3 * blockIdx.x + 2];
  const int64_t ibase = (int64_t)I * SYM_TB + wave * 64;      // first global row of this wave

  // B operand of the transposed product: the X_I rows of this block row, kept in LDS (padded rows) -
  // in registers they would cost 32 NT VGPRs that the two-step-deep load pipeline needs.
  for (int e = threadIdx.x; e < SYM_TB * 16 * NT; e += 256) {
    int r = e >> 4, cc = e & 15, t = r / SYM_TB;
    r -= t * SYM_TB;
    xs[t][r * XS + cc] = xt[t * group_stride + ((int64_t)I * SYM_TB + r) * 16 + cc];
  }
  __syncthreads();

  f64x4 acc[4][NT];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[rt][t] = f64x4{0.0, 0.0, 0.0, 0.0};

  // Per 16-column step each wave holds its 64 x 16 sub-block in registers in the direct layout; the
  // Gram-layout copy the transposed product needs is made through a wave-private LDS scratch
  // (8 x ds_write_b128, 8 x ds_read_b128 - no barrier: a wave's DS operations complete in order).  The
  // HBM loads of step s+1 are issued before the 32 NT MFMAs of step s (register double buffer), so
  // HBM latency hides under the matrix pipe even at 2 waves per SIMD.  (Re-reading the sub-block from
  // global memory instead was measured: issued late it misses L2 and doubles the HBM traffic; issued
  // together with the first read it costs as many registers as this scheme and loads the TA twice.)
  const int nbatch = (J1 - J0) * 4;                 // 64-column batches
  const int64_t dlane = wave * 64 + 2 * c + (int64_t)g * SYM_TB;
  double* tw = tr[wave];                            // [16 columns][TRS doubles]
  const int nsteps = nbatch * 4;
  auto step_ptr = [&](int q) {                       // direct-layout base of 16-column step q of the run (clamped)
    q = q < nsteps ? q : nsteps - 1;
    return sym_tile(tiles, I, J0 + (q >> 4)) + (int64_t)((q & 15) * 16) * SYM_TB + dlane;
  };
  f64x2 a[4][2], a1[4][2];
  {
    const double* ad = step_ptr(0);
    const double* ad1 = step_ptr(1);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u][0] = *reinterpret_cast<const f64x2*>(ad + (int64_t)(4 * u) * SYM_TB);
      a[u][1] = *reinterpret_cast<const f64x2*>(ad + (int64_t)(4 * u) * SYM_TB + 32);
      a1[u][0] = *reinterpret_cast<const f64x2*>(ad1 + (int64_t)(4 * u) * SYM_TB);
      a1[u][1] = *reinterpret_cast<const f64x2*>(ad1 + (int64_t)(4 * u) * SYM_TB + 32);
    }
  }
  for (int bt = 0; bt < nbatch; ++bt) {
    const int J = J0 + (bt >> 2), cb = bt & 3;
    const double* tile = sym_tile(tiles, I, J);
    const bool offdiag = (J != I);
    const double* xj = xt + ((int64_t)J * SYM_TB + g) * 16 + c;
    f64x4 z[4][NT];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int t = 0; t < NT; ++t) z[jt][t] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const int col = cb * 64 + jt * 16;
      // prefetch the sub-block two steps ahead
      const double* adn = step_ptr(bt * 4 + jt + 2);
      f64x2 an[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        an[u][0] = *reinterpret_cast<const f64x2*>(adn + (int64_t)(4 * u) * SYM_TB);
        an[u][1] = *reinterpret_cast<const f64x2*>(adn + (int64_t)(4 * u) * SYM_TB + 32);
      }
      double b[4][NT];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t) b[u][t] = xj[t * group_stride + (col + 4 * u) * 16];
      f64x2 p[4][2];
      if (offdiag) {
        // direct layout -> LDS: lane (c, g) owns rows 2c, 2c+1 (+32) of column 4u + g
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          *reinterpret_cast<f64x2*>(tw + (4 * u + g) * TRS + 2 * c) = a[u][0];
          *reinterpret_cast<f64x2*>(tw + (4 * u + g) * TRS + 2 * c + 32) = a[u][1];
        }
        // LDS -> Gram layout: lane (c, g) owns rows 16 ib + 4g .. +3 of column c
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
          p[ib][0] = *reinterpret_cast<const f64x2*>(tw + c * TRS + 16 * ib + 4 * g);
          p[ib][1] = *reinterpret_cast<const f64x2*>(tw + c * TRS + 16 * ib + 4 * g + 2);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[0][t] = mfma_f64(a[u][0].x, b[u][t], acc[0][t]);
          acc[1][t] = mfma_f64(a[u][0].y, b[u][t], acc[1][t]);
          acc[2][t] = mfma_f64(a[u][1].x, b[u][t], acc[2][t]);
          acc[3][t] = mfma_f64(a[u][1].y, b[u][t], acc[3][t]);
        }
      if (offdiag) {
        // (interleaving these dependent MFMAs with the direct ones was measured 9 % slower)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int xr = (wave * 64 + 16 * ib + 4 * g) * XS + c;
            z[jt][t] = mfma_f64(p[ib][0].x, xs[t][xr], z[jt][t]);
            z[jt][t] = mfma_f64(p[ib][0].y, xs[t][xr + XS], z[jt][t]);
            z[jt][t] = mfma_f64(p[ib][1].x, xs[t][xr + 2 * XS], z[jt][t]);
            z[jt][t] = mfma_f64(p[ib][1].y, xs[t][xr + 3 * XS], z[jt][t]);
          }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u][0] = a1[u][0];
        a[u][1] = a1[u][1];
        a1[u][0] = an[u][0];
        a1[u][1] = an[u][1];
      }
    }
    if (offdiag) {
      // z[jt][t][reg]: tile column cb*64 + jt*16 + g + 4 reg, block column 16 t + c.  Sum the 4 waves.
      __syncthreads();
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg)
            red[wave][(16 * t + c) * RS + jt * 16 + g + 4 * reg] = z[jt][t][reg];
      __syncthreads();
      // slabT tile (I, J): [block column][256 tile columns]
      double* outT = slabT + (((int64_t)I * (I - 1) / 2 + J) * (16 * NT)) * SYM_TB + cb * 64;
      for (int e = threadIdx.x; e < 64 * 16 * NT; e += 256) {
        const int le = (e >> 6) * RS + (e & 63);
        double v = red[0][le] + red[1][le] + red[2][le] + red[3][le];
        outT[(int64_t)(e >> 6) * SYM_TB + (e & 63)] = v;
      }
      __syncthreads();          // the region goes back to being transposition scratch
    }
  }

  double* outD = slabD + (int64_t)blockIdx.x * (16 * NT) * SYM_TB + wave * 64;
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const int half = rt >> 1, par = rt & 1;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        outD[(int64_t)(16 * t + c) * SYM_TB + 32 * half + 2 * (g + 4 * reg) + par] = acc[rt][t][reg];
  }
}

void launch_matvec_sym(hipStream_t st, const double* tiles, const int* items_dev, int nitems, const double* xt,
                       int64_t xt_group_stride, int ngroups, double* slabD, double* slabT) {
  // one 16-column group per pass: two groups per pass need more registers/LDS than 2 waves per SIMD
  // allow and measured slower than two passes
  (void)ngroups;
  hipLaunchKernelGGL(matvec_sym_kernel<1>, dim3(nitems), dim3(256), 0, st, tiles, items_dev, xt, xt_group_stride, slabD, slabT);
}

// W[J*256 + r, col] = sum over runs of block row J of slabD + sum over I > J of slabT(I, J), fixed order.
__global__ __launch_bounds__(256) void sym_reduce_kernel(const double* __restrict__ slabD, const double* __restrict__ slabT,
                                                         const int* __restrict__ row_item_begin, int nb, int ncol16,
                                                         int64_t nloc, int k, double* __restrict__ dst, int64_t ldd) {
  const int J = blockIdx.x, col = blockIdx.y, r = threadIdx.x;
  if (col >= k) return;
  double sum = 0.0;
  for (int it = row_item_begin[J]; it < row_item_begin[J + 1]; ++it)
    sum += slabD[((int64_t)it * ncol16 + col) * SYM_TB + r];
  for (int I = J + 1; I < nb; ++I)
    sum += slabT[((((int64_t)I * (I - 1) / 2 + J) * ncol16) + col) * SYM_TB + r];
  int64_t row = (int64_t)J * SYM_TB + r;
  dst[(int64_t)col * ldd + row] = row < nloc ? sum : 0.0;
}

void launch_sym_reduce(hipStream_t st, const double* slabD, const double* slabT, const int* row_item_begin_dev, int nb,
                       int ngroups, int64_t nloc, int k, double* dst, int64_t ldd) {
  hipLaunchKernelGGL(sym_reduce_kernel, dim3(nb, k), dim3(256), 0, st, slabD, slabT, row_item_begin_dev, nb, ngroups * 16,
                     nloc, k, dst, ldd);
}

// ---- storage helpers ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void generate_sym_tiles_kernel(double* __restrict__ tiles, int64_t tile0, int64_t ntiles,
                                                                 int64_t n, uint64_t seed, double sparsity, int use_diag,
                                                                 double diag_val) {
  // blockIdx.x = tile (relative to tile0), blockIdx.y = 16 columns of the tile; thread = row
  int64_t t = tile0 + blockIdx.x;
  if (t >= ntiles) return;
  // invert t = I (I+1)/2 + J
  int64_t I = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (I * (I + 1) / 2 > t) --I;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  int64_t J = t - I * (I + 1) / 2;
  double* tile = tiles + t * (int64_t)(SYM_TB * SYM_TB);
  int64_t gi = I * SYM_TB + threadIdx.x;
  for (int cc = 0; cc < 16; ++cc) {
    int64_t lc = blockIdx.y * 16 + cc, gj = J * SYM_TB + lc;
    double v = 0.0;
    if (gi < n && gj < n) v = dav_hashed_entry(seed, sparsity, use_diag, diag_val, gi, gj);
    tile[lc * SYM_TB + threadIdx.x] = v;
  }
}

void launch_generate_sym_tiles(hipStream_t st, double* tiles, int64_t ntiles, int64_t n, uint64_t seed, double sparsity,
                               int use_diag, double diag_val) {
  const int64_t batch = 32768;
  for (int64_t t0 = 0; t0 < ntiles; t0 += batch) {
    int64_t nt = ntiles - t0 < batch ? ntiles - t0 : batch;
    hipLaunchKernelGGL(generate_sym_tiles_kernel, dim3((unsigned)nt, SYM_TB / 16), dim3(SYM_TB), 0, st, tiles, t0, ntiles, n,
                       seed, sparsity, use_diag, diag_val);
  }
}

__device__ __forceinline__ double sym_entry(const double* tiles, int64_t i, int64_t j) {
  if (i < j) { int64_t t = i; i = j; j = t; }
  int I = (int)(i / SYM_TB), J = (int)(j / SYM_TB);
  return sym_tile(tiles, I, J)[(j % SYM_TB) * SYM_TB + (i % SYM_TB)];
}

__global__ void diag_sym_kernel(const double* __restrict__ tiles, int64_t n, double* __restrict__ diag) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) diag[i] = sym_entry(tiles, i, i);
}
void launch_diag_sym(hipStream_t st, const double* tiles, int64_t n, double* diag) {
  hipLaunchKernelGGL(diag_sym_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tiles, n, diag);
}

__global__ void gather_columns_sym_kernel(const double* __restrict__ tiles, int64_t n, int64_t nrows_pad,
                                          const int64_t* __restrict__ idx, double* __restrict__ dst, int64_t ldd) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int c = blockIdx.y;
  if (i < nrows_pad) dst[(int64_t)c * ldd + i] = i < n ? sym_entry(tiles, i, idx[c]) : 0.0;
}
void launch_gather_columns_sym(hipStream_t st, const double* tiles, int64_t n, int64_t nrows_pad, const int64_t* idx_dev,
                               int k, double* dst, int64_t ldd) {
  hipLaunchKernelGGL(gather_columns_sym_kernel, dim3((unsigned)((nrows_pad + 255) / 256), k), dim3(256), 0, st, tiles, n,
                     nrows_pad, idx_dev, dst, ldd);
}
