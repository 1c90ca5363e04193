// K1s - block matvec on SYMMETRIC-TILED storage: only the lower block triangle of A (tiles of
// 256 x 256, each contiguous, diagonal tiles stored in full) lives in HBM - N(N+1)/2 instead of N^2
// entries, which is what makes N = 200000 (160 GB) fit one MI355X - and every off-diagonal tile is
// used twice per sweep:   W_I += A_IJ X_J   (direct)   and   W_J += A_IJ^T X_I   (transposed).
//
// Workgroup = 4 waves, one run of tiles (I, J0..J1) of block row I.  Wave w owns rows 64w..64w+63 of
// the block row.  Per 16-column step it holds its 64 x 16 sub-block in registers in the "direct" lane
// layout (16 B = 2 rows of one column per lane; MFMA contraction over columns, accumulators 64 rows x
// 16 stay in registers for the whole run); the "Gram" lane layout the transposed product needs
// (4 consecutive rows of one column per lane; MFMA contraction over rows) is made through a
// wave-private LDS scratch (8 x ds_write_b128 + 8 x ds_read_b128, no barrier: a wave's DS operations
// complete in order).  The transposed partials of the four waves are summed through LDS once per 64
// columns and written to a per-tile slab; direct partials go to a per-run slab; a second kernel adds,
// in fixed order, the slabs that belong to each output block (bitwise reproducible, no fp64 atomics).
//
// Load pipeline: a 4-slot register ring indexed by the (compile-time) step number inside a 64-column
// batch; the loads of step q+3 are issued before the MFMAs of step q, with no register moves (a rotating
// buffer with moves makes the compiler wait for the newest loads at the end of every step, which is what
// the first versions of this kernel did).  One wave per SIMD, up to 24 KB in flight per wave.
//
// Measured dead ends, kept here so they are not retried blindly: re-reading the sub-block from global
// memory in the Gram layout (late: misses L2 and doubles HBM traffic; early: as many registers as the
// LDS scheme and twice the TA work); interleaving the dependent transposed MFMAs with the direct ones
// (9 % slower); two 16-column groups per pass (registers/LDS exceed 2 waves per SIMD, slower than two
// passes); producing the Gram-layout operand one step ahead with sched_group_barrier interleaving (same
// speed: hipcc keeps the DS writes in one block).  Counters (profiles/r01_pmc_mfma_clock_n40000.json,
// N=40000): matrix pipe 43 % busy at 2.34 GHz, wave time = 51 % MFMA issue-blocked + 26 % s_waitcnt/barrier
// + 20 % issuing the ~80 VALU / 30 DS / 13 VMEM instructions per step; HBM fetch = tile bytes.  What is
// left is the per-64-column cross-wave reduction and the un-overlapped non-MFMA issue at one wave per SIMD.
#include "kernels.h"

// tile (I, J), J <= I, at tiles + (I (I+1)/2 + J) * TB*TB, column-major with leading dimension TB
__device__ __forceinline__ const double* sym_tile(const double* tiles, int I, int J) {
  return tiles + ((int64_t)I * (I + 1) / 2 + J) * (int64_t)(SYM_TB * SYM_TB);
}

constexpr int SYM_DEPTH = 3;      // steps of load lookahead (ring of 4 slots)

__global__ __launch_bounds__(256, 1) void matvec_sym_kernel(const double* __restrict__ tiles, const int* __restrict__ items,
                                                            const double* __restrict__ xt, double* __restrict__ slabD,
                                                            double* __restrict__ slabT) {
  constexpr int RS = 65;          // padded stride of the Z-partial exchange: block columns on different banks
  constexpr int TRS = 66;         // padded column stride of the transposition scratch (528 B)
  constexpr int XS = 17;          // padded row stride of the X_I copy
  __shared__ double red[4][RS * 16];
  __shared__ __attribute__((aligned(16))) double tr[4][16 * TRS];
  __shared__ double xs[SYM_TB * XS];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int I = items[3 * blockIdx.x], J0 = items[3 * blockIdx.x + 1], J1 = items[3 * blockIdx.x + 2];

  // B operand of the transposed product: the X_I rows of this block row (LDS, padded rows)
  for (int e = threadIdx.x; e < SYM_TB * 16; e += 256)
    xs[(e >> 4) * XS + (e & 15)] = xt[((int64_t)I * SYM_TB + (e >> 4)) * 16 + (e & 15)];
  __syncthreads();

  f64x4 acc[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) acc[rt] = f64x4{0.0, 0.0, 0.0, 0.0};

  const int nbatch = (J1 - J0) * 4;                 // 64-column batches
  const int nsteps = nbatch * 4;                    // 16-column steps
  const int64_t dlane = wave * 64 + 2 * c + (int64_t)g * SYM_TB;
  double* tw = tr[wave];

  // ring slot of step q is q & 3 (= jt inside a batch): A sub-block in the direct layout + B operand
  f64x2 ra[4][4][2];
  double rb[4][4];
  auto load_step = [&](int q, f64x2 (&a)[4][2], double (&b)[4]) {
    q = q < nsteps ? q : nsteps - 1;                // clamped at the end of the run: a harmless re-read
    const int J = J0 + (q >> 4), col = (q & 15) * 16;
    const double* ad = sym_tile(tiles, I, J) + (int64_t)col * SYM_TB + dlane;
    const double* xj = xt + ((int64_t)J * SYM_TB + col + g) * 16 + c;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u][0] = *reinterpret_cast<const f64x2*>(ad + (int64_t)(4 * u) * SYM_TB);
      a[u][1] = *reinterpret_cast<const f64x2*>(ad + (int64_t)(4 * u) * SYM_TB + 32);
      b[u] = xj[(4 * u) * 16];
    }
  };
#pragma unroll
  for (int d = 0; d < SYM_DEPTH; ++d) load_step(d, ra[d], rb[d]);

  for (int bt = 0; bt < nbatch; ++bt) {
    const int J = J0 + (bt >> 2), cb = bt & 3;
    const bool offdiag = (J != I);
    f64x4 z[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) z[jt] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      f64x2 (&a)[4][2] = ra[jt];
      double (&b)[4] = rb[jt];
      load_step(bt * 4 + jt + SYM_DEPTH, ra[(jt + SYM_DEPTH) & 3], rb[(jt + SYM_DEPTH) & 3]);
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
      for (int u = 0; u < 4; ++u) {
        acc[0] = mfma_f64(a[u][0].x, b[u], acc[0]);
        acc[1] = mfma_f64(a[u][0].y, b[u], acc[1]);
        acc[2] = mfma_f64(a[u][1].x, b[u], acc[2]);
        acc[3] = mfma_f64(a[u][1].y, b[u], acc[3]);
      }
      if (offdiag) {
        // four independent accumulator chains (as in the direct product): a dependent f64 MFMA issued
        // fewer than ~4 slots behind its producer stalls the pipe (SQ_WAIT_INST_ANY was 51 % of wave time
        // with one or two chains)
        f64x4 zc[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) zc[s4] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
          const int xr = (wave * 64 + 16 * ib + 4 * g) * XS + c;
          zc[0] = mfma_f64(p[ib][0].x, xs[xr], zc[0]);
          zc[1] = mfma_f64(p[ib][0].y, xs[xr + XS], zc[1]);
          zc[2] = mfma_f64(p[ib][1].x, xs[xr + 2 * XS], zc[2]);
          zc[3] = mfma_f64(p[ib][1].y, xs[xr + 3 * XS], zc[3]);
        }
        z[jt] = (zc[0] + zc[1]) + (zc[2] + zc[3]);
      }
    }
    if (offdiag) {
      // z[jt][reg]: tile column cb*64 + jt*16 + g + 4 reg, block column c.  Sum the 4 waves.
      __syncthreads();
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) red[wave][c * RS + jt * 16 + g + 4 * reg] = z[jt][reg];
      __syncthreads();
      // slabT tile (I, J): [16 block columns][256 tile columns]
      double* outT = slabT + (((int64_t)I * (I - 1) / 2 + J) * 16) * SYM_TB + cb * 64;
      for (int e = threadIdx.x; e < 64 * 16; e += 256) {
        const int le = (e >> 6) * RS + (e & 63);
        outT[(int64_t)(e >> 6) * SYM_TB + (e & 63)] = red[0][le] + red[1][le] + red[2][le] + red[3][le];
      }
    }
  }

  double* outD = slabD + (int64_t)blockIdx.x * 16 * SYM_TB + wave * 64;
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const int half = rt >> 1, par = rt & 1;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
      outD[(int64_t)c * SYM_TB + 32 * half + 2 * (g + 4 * reg) + par] = acc[rt][reg];
  }
}

void launch_matvec_sym(hipStream_t st, const double* tiles, const int* items_dev, int nitems, const double* xt,
                       int64_t xt_group_stride, int ngroups, double* slabD, double* slabT) {
  // one 16-column group per pass
  (void)ngroups;
  (void)xt_group_stride;
  hipLaunchKernelGGL(matvec_sym_kernel, dim3(nitems), dim3(256), 0, st, tiles, items_dev, xt, slabD, slabT);
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
  (void)ngroups;
  hipLaunchKernelGGL(sym_reduce_kernel, dim3(nb, k), dim3(256), 0, st, slabD, slabT, row_item_begin_dev, nb, 16, nloc, k,
                     dst, ldd);
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
