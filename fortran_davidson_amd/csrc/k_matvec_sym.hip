// K1s - block matvec on SYMMETRIC-TILED storage: only the lower block triangle of A (tiles of
// 256 x 256, each contiguous, diagonal tiles stored in full) lives in HBM - N(N+1)/2 instead of N^2
// entries, which is what makes N = 200000 (160 GB) fit one MI355X - and every off-diagonal tile is
// used twice per sweep:   W_I += A_IJ X_J   (direct)   and   W_J += A_IJ^T X_I   (transposed).
//
// One-block-row kernel matvec_sym8_kernel (8 waves, two per SIMD; small matrices and the harness operator - larger ones run the
// super-row schedules of k_matvec_sym9.hip / k_matvec_symw.hip, same slabs, same kind of reduction): workgroup = one run of
// tiles (I, J0..J1) of block row I.  Per half-step the 32 x 16 sub-block sits in registers in the "direct" lane layout (16 B =
// 2 rows of one column per lane; MFMA contraction over columns); the "Gram" lane layout the transposed product needs (4
// consecutive rows of one column per lane; MFMA contraction over rows) is made through a wave-private LDS scratch (no barrier:
// a wave's DS operations complete in order).  A second kernel adds, in fixed order, the slabs that belong to each output block
// (bitwise reproducible, no fp64 atomics).
//
// History, so that measured dead ends are not retried blindly.  The first version ran ONE wave per SIMD (4 waves, each owning 16
// tile columns over all 256 rows, transposed partials complete inside a wave, a 4-slot load ring three steps ahead): its ~100
// non-MFMA instructions per step are serialised with the 32 MFMAs (N=60000 / N=200000, k=16: 3.34 / 32.9 ms); it was kept behind
// an A/B knob until round 4 and then deleted - the two-waves-per-SIMD kernel below replaced it everywhere in round 1, and the
// one-wave-per-SIMD idea came back in round 3 as k_matvec_symw.hip with the memory operations placed between the MFMAs.  Also
// tried on that version: re-reading the sub-block from global memory in the Gram layout (late: misses L2 and doubles HBM
// traffic; early: as many registers as the LDS scheme and twice the TA work); interleaving the dependent transposed MFMAs with
// the direct ones (9 % slower); two 16-column groups per pass (slower than two passes); producing the Gram operand one step
// ahead with sched_group_barrier (same speed); summing the waves once per 128 columns (1 %); runs of 4..14 tiles per workgroup
// (within 2 %; a workgroup costs ~7 us to start and drain); on the 8-wave kernel: fp64 atomics into W instead of the slabs and
// the reduction kernel (not reproducible; N=60000, k=16: 3.07 -> 2.80 ms, but N=200000, k=8: 29.9 -> 29.5 ms and the paired
// 32-column launches 4 % slower).  Traffic per sweep: the stored half matrix once, plus 1/16 of it written as per-tile Z slabs
// and read back by the reduction kernel (8 % of the sweep time) - the price of a deterministic sum.
#include "kernels.h"
#include <cstdlib>
#include <type_traits>

// tile (I, J), J <= I, at tiles + (row_off[I] + J) * TB*TB, column-major with leading dimension TB.  row_off[I] = first
// tile of block row I in THIS rank's storage: I (I+1)/2 on a single rank; with several ranks only the block rows a rank
// owns are stored (the others carry -1 and are never addressed)
__device__ __forceinline__ const double* sym_tile(const double* tiles, const int64_t* __restrict__ row_off, int I, int J) {
  return tiles + (row_off[I] + J) * (int64_t)(SYM_TB * SYM_TB);
}

// ---- two waves per SIMD -----------------------------------------------------------------------------------
// Measured on gfx950 (profiles/ubench/r01_mfma_f64_overlap.log): while a wave has a v_mfma_f64_16x16x4 in
// flight it issues nothing else - every VALU / DS / VMEM instruction of that wave adds its own issue time
// (4-5 cycles, 18+ for a 16-byte global load) on top of the 64 cycles per MFMA; a SECOND wave on the same
// SIMD, however, issues in the shadow of those MFMAs at full MFMA rate for the first.  A workgroup has 8 waves: wave (w, h) owns tile columns
// 16w..16w+15 of every 64-column batch over the row half h (128 rows, four 32-row half-steps), which fits
// 256 registers.  The transposed partial of a unit is now split over the two waves of a pair: wave h=1
// hands its 16 x 16 partial to wave h=0 through LDS (one workgroup barrier per unit, double buffered).
constexpr int SYM8_DEPTH = 3;     // half-steps of load lookahead (ring of 4 slots)

// GEN = true: the matrix-free variant - the entries of the hashed diagonal-dominant operator (same values as
// the dense generator) are produced in registers instead of being loaded, ONCE per symmetric pair: half the
// hash evaluations of the row-slab kernel (matvec_free_kernel), which is what that VALU-bound path is made of.
template <bool GEN>
__global__ __launch_bounds__(512, 1) void matvec_sym8_kernel(const double* __restrict__ tiles, const int64_t* __restrict__ row_off,
                                                             const int* __restrict__ items,
                                                             const double* __restrict__ xt, double* __restrict__ slabD,
                                                             double* __restrict__ slabT, int kcols, int npair,
                                                             int64_t xt_gstride, int64_t slabD_gstride, int64_t slabT_gstride,
                                                             OpParams op, int64_t n) {
  // npair = 2: two 16-column groups in ONE launch - workgroups 2i and 2i+1 run the same work item on group 0
  // and group 1, are dispatched back to back and stream the same tiles at the same pace, so the second read
  // of a tile is served by the memory-side cache instead of HBM
  // consecutive workgroups go to consecutive XCDs (8 of them, each with its own L2): the two members of a
  // pair are 8 apart in the grid, so they land on the same XCD and can share its L2 as well
  int item, grp;
  if (npair == 2) {
    const int nfull = (int)(gridDim.x / 16) * 16;            // whole blocks of 8 pairs
    if ((int)blockIdx.x < nfull) {
      item = (blockIdx.x / 16) * 8 + (blockIdx.x % 8);
      grp = (blockIdx.x / 8) % 2;
    } else {                                                   // ragged tail: plain interleaving
      item = nfull / 2 + (blockIdx.x - nfull) / 2;
      grp = (blockIdx.x - nfull) % 2;
    }
  } else {
    item = blockIdx.x;
    grp = 0;
  }
  xt += grp * xt_gstride;
  slabD += grp * slabD_gstride;
  slabT += grp * slabT_gstride;
  kcols = kcols - 16 * grp < 16 ? kcols - 16 * grp : 16;
  constexpr int TRS = 34;         // padded column stride of the 32-row transposition scratch (272 B)
  constexpr int TRW = 16 * TRS;   // doubles per wave
  constexpr int XT = 258;         // padded column stride of the transposed X_I copy
  constexpr int RS = 33;          // padded stride of the end-of-run exchange (32 rows per block column)
  __shared__ __attribute__((aligned(16))) double tr[8 * TRW];
  __shared__ __attribute__((aligned(16))) double xsT[16 * XT];
  __shared__ __attribute__((aligned(16))) double zred[2][4][2][128];   // [unit parity][column group][half of f64x4][lane x 2]
  // Stage of the transposed partials: 32 rows of 256 tile columns = 4 tiles x 8 block columns (k <= 8) or 2 tiles
  // x 16.  The units deposit their 16 x 16 pieces here; every 4 (2) tiles the stage leaves the chip as full 2 KB
  // rows.  Why: global stores and loads share one in-order counter (vmcnt), so a wave that has stored cannot
  // retire its NEXT tile loads before that store is acknowledged, and the per-unit barrier hands the stall to the
  // whole workgroup.  Measured at N=200000, k=8 on one box: no Z stores at all 25.8 ms, stores into an 8 MB
  // (cache-resident) window 27.8 ms, 32-byte fragments per unit straight to the slab 28.8 ms, one staged flush per
  // tile 28.8 ms (N=100000: 7.9 -> 7.0 ms), non-temporal stores 29.8 ms: what costs is the number of store
  // EVENTS a wave waits behind, hence as few flushes as the LDS allows.
  constexpr int ZS = 258;         // padded row stride: the unit writes (c * ZS + g + 4 reg) are conflict free
  constexpr int ZROWS = 32;
  __shared__ __attribute__((aligned(16))) double zst[ZROWS * ZS];
  static_assert(TRW >= 16 * RS, "the end-of-run exchange reuses the transposition scratch");
  const int zrows = kcols <= 8 ? 8 : 16;            // rows of one tile in the stage
  const int zslots = ZROWS / zrows;                  // tiles the stage holds

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int w = wave & 3, h = wave >> 2;
  const int c = lane & 15, g = lane >> 4;
  // work item: (block row, first tile, end tile, slab slot); items are dispatched longest first
  const int I = items[4 * item], J0 = items[4 * item + 1], J1 = items[4 * item + 2];

  for (int e = threadIdx.x; e < SYM_TB * 16; e += 512)
    xsT[(e & 15) * XT + (e >> 4)] = xt[((int64_t)I * SYM_TB + (e >> 4)) * 16 + (e & 15)];
  __syncthreads();

  // direct partials: 128 rows (this wave's half) x 16 block columns; [half-step][row parity]
  f64x4 acc[4][2];
#pragma unroll
  for (int hs = 0; hs < 4; ++hs) { acc[hs][0] = f64x4{0.0, 0.0, 0.0, 0.0}; acc[hs][1] = f64x4{0.0, 0.0, 0.0, 0.0}; }

  const int nunits = (J1 - J0) * 4;
  const int nsteps = nunits * 4;                    // half-steps
  const int64_t dlane = 128 * h + 2 * c + (int64_t)g * SYM_TB;
  double* tw = tr + wave * TRW;

  f64x2 ra[4][4];                                   // ring: slot = half-step inside the unit
  const uint64_t seedmix = op.seed * 0x9E3779B97F4A7C15ull;
  const double gscale = op.sparsity * (1.0 / 9007199254740992.0);
  const bool rows_inside = ((int64_t)I + 1) * SYM_TB <= n;       // no ragged rows in this block row
  auto load_hs = [&](int s, f64x2 (&a)[4]) {
    s = s < nsteps ? s : nsteps - 1;
    const int q = s >> 2, hs = s & 3;
    const int J = J0 + (q >> 2), col = (q & 3) * 64 + w * 16;
    if constexpr (GEN) {
      // lane (c, g): rows gi, gi + 1 of column gj + 4u
      const int64_t gi = (int64_t)I * SYM_TB + 128 * h + 32 * hs + 2 * c;
      const int64_t gj = (int64_t)J * SYM_TB + col + g;
      if (op.kind == DAV_KIND_HARNESS) {
        // the reference's test operator (transcendental entries, symmetric by construction)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t cj = gj + 4 * u;
          a[u].x = (gi < n && cj < n) ? dav_harness_entry(op, gi, cj) : 0.0;
          a[u].y = (gi + 1 < n && cj < n) ? dav_harness_entry(op, gi + 1, cj) : 0.0;
        }
      } else if (J < I && rows_inside) {
        // strictly below the diagonal and inside the matrix: lo = column, hi = row, no tests per entry
        const uint64_t k0 = (uint64_t)gi + seedmix;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint64_t kc = ((uint64_t)(gj + 4 * u) << 32) + k0;
          const uint64_t m0 = dav_splitmix64(kc) >> 11, m1 = dav_splitmix64(kc + 1) >> 11;
          a[u].x = __builtin_fma((double)(uint32_t)(m0 >> 32), 4294967296.0, (double)(uint32_t)m0) * gscale;
          a[u].y = __builtin_fma((double)(uint32_t)(m1 >> 32), 4294967296.0, (double)(uint32_t)m1) * gscale;
        }
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t cj = gj + 4 * u;
          a[u].x = (gi < n && cj < n) ? dav_hashed_entry(op.seed, op.sparsity, op.use_diag, op.diag_val, gi, cj) : 0.0;
          a[u].y = (gi + 1 < n && cj < n) ? dav_hashed_entry(op.seed, op.sparsity, op.use_diag, op.diag_val, gi + 1, cj) : 0.0;
        }
      }
    } else {
      const double* ad = sym_tile(tiles, row_off, I, J) + (int64_t)col * SYM_TB + 32 * hs + dlane;
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const f64x2*>(ad + (int64_t)(4 * u) * SYM_TB);
    }
  };
  auto load_b = [&](int q, double (&b)[4]) {
    q = q < nunits ? q : nunits - 1;
    const int J = J0 + (q >> 2), col = (q & 3) * 64 + w * 16;
    const double* xj = xt + ((int64_t)J * SYM_TB + col + g) * 16 + c;
#pragma unroll
    for (int u = 0; u < 4; ++u) b[u] = xj[(4 * u) * 16];
  };
  double b[4], bn[4];
  load_b(0, b);
#pragma unroll
  for (int d = 0; d < SYM8_DEPTH; ++d) load_hs(d, ra[d]);

  const int nq_off = ((J1 - 1 == I ? J1 - 1 : J1) - J0) * 4;
  // tiles [t0, t0 + cnt) of the run (J = J0 + t): stage -> slabT tile (I, J) = [16 block columns][256 tile columns]
  auto flush_tiles = [&](int t0, int cnt) {
    for (int sidx = 0; sidx < cnt; ++sidx) {
      const double* zs = zst + (size_t)((t0 + sidx) % zslots) * zrows * ZS;
      double* outT = slabT + ((int64_t)I * (I - 1) / 2 + J0 + t0 + sidx) * 16 * SYM_TB;
      for (int e = threadIdx.x; e < kcols * (SYM_TB / 2); e += 512) {   // block columns beyond the k in use: nobody reads them
        const int bc = e >> 7, pr = e & 127;
        *reinterpret_cast<f64x2*>(outT + bc * SYM_TB + 2 * pr) = *reinterpret_cast<const f64x2*>(zs + bc * ZS + 2 * pr);
      }
    }
  };
  auto unit = [&](auto off_tag, int q) {
    constexpr bool OFF = decltype(off_tag)::value;
    load_b(q + 1, bn);
    int xoff = c * XT + 128 * h + 4 * g;            // opaque: keeps the X_I reads inside the loop
    asm volatile("" : "+v"(xoff));
    const double* xw = xsT + xoff;
    f64x4 zc[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) zc[s4] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int hs = 0; hs < 4; ++hs) {
      f64x2 (&a)[4] = ra[hs];
      load_hs(q * 4 + hs + SYM8_DEPTH, ra[(hs + SYM8_DEPTH) & 3]);
      if constexpr (OFF) {
        f64x2 p[2][2], xb[2][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<f64x2*>(tw + (4 * u + g) * TRS + 2 * c) = a[u];
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
          p[ib][0] = *reinterpret_cast<const f64x2*>(tw + c * TRS + 16 * ib + 4 * g);
          p[ib][1] = *reinterpret_cast<const f64x2*>(tw + c * TRS + 16 * ib + 4 * g + 2);
          xb[ib][0] = *reinterpret_cast<const f64x2*>(xw + 32 * hs + 16 * ib);
          xb[ib][1] = *reinterpret_cast<const f64x2*>(xw + 32 * hs + 16 * ib + 2);
        }
        // direct and transposed MFMAs alternate: every accumulator chain is touched once per four MFMAs
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
          acc[hs][0] = mfma_f64(a[2 * ib].x, b[2 * ib], acc[hs][0]);
          zc[0] = mfma_f64(p[ib][0].x, xb[ib][0].x, zc[0]);
          acc[hs][1] = mfma_f64(a[2 * ib].y, b[2 * ib], acc[hs][1]);
          zc[1] = mfma_f64(p[ib][0].y, xb[ib][0].y, zc[1]);
          acc[hs][0] = mfma_f64(a[2 * ib + 1].x, b[2 * ib + 1], acc[hs][0]);
          zc[2] = mfma_f64(p[ib][1].x, xb[ib][1].x, zc[2]);
          acc[hs][1] = mfma_f64(a[2 * ib + 1].y, b[2 * ib + 1], acc[hs][1]);
          zc[3] = mfma_f64(p[ib][1].y, xb[ib][1].y, zc[3]);
        }
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc[hs][0] = mfma_f64(a[u].x, b[u], acc[hs][0]);
          acc[hs][1] = mfma_f64(a[u].y, b[u], acc[hs][1]);
        }
      }
    }
    if constexpr (OFF) {
      // z[reg]: tile column col + g + 4 reg, block column c, summed over this wave's 128 rows; the pair's
      // other half arrives through LDS.  One barrier per unit; the buffer alternates with the unit parity.
      f64x4 z = (zc[0] + zc[1]) + (zc[2] + zc[3]);
      double* zr = &zred[q & 1][w][0][0];
      if (h == 1) {
        *reinterpret_cast<f64x2*>(zr + 2 * lane) = f64x2{z[0], z[1]};
        *reinterpret_cast<f64x2*>(zr + 128 + 2 * lane) = f64x2{z[2], z[3]};
      }
      __syncthreads();
      if (h == 0) {
        const f64x2 z01 = *reinterpret_cast<const f64x2*>(zr + 2 * lane);
        const f64x2 z23 = *reinterpret_cast<const f64x2*>(zr + 128 + 2 * lane);
        z[0] += z01.x; z[1] += z01.y; z[2] += z23.x; z[3] += z23.y;
        if (c < zrows) {          // block columns beyond the k in use carry zeros nobody reads
          double* zs = zst + (size_t)(((q >> 2) % zslots) * zrows + c) * ZS + (q & 3) * 64 + w * 16 + g;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) zs[4 * reg] = z[reg];
        }
      }
      if ((q & 3) == 3 && ((q >> 2) + 1) % zslots == 0) {    // the stage is full
        __syncthreads();
        flush_tiles((q >> 2) + 1 - zslots, zslots);
        __syncthreads();
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) b[u] = bn[u];
  };
  int q = 0;
  for (; q < nq_off; ++q) unit(std::true_type{}, q);
  if ((nq_off >> 2) % zslots != 0) {   // what the stage still holds
    __syncthreads();
    flush_tiles((nq_off >> 2) - (nq_off >> 2) % zslots, (nq_off >> 2) % zslots);
  }
  for (; q < nunits; ++q) unit(std::false_type{}, q);

  // end of the run: sum the direct partials over the four column groups, one 32-row half-step at a time
  double* outD = slabD + (int64_t)items[4 * item + 3] * 16 * SYM_TB;
#pragma unroll
  for (int hs = 0; hs < 4; ++hs) {
    __syncthreads();
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) tw[c * RS + 2 * (g + 4 * reg) + par] = acc[hs][par][reg];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int e = threadIdx.x + 512 * t;
      const int hh = e >> 9, bc = (e >> 5) & 15, r = e & 31;
      const double* rp = tr + (hh * 4) * TRW + bc * RS + r;
      if (bc < kcols) outD[(int64_t)bc * SYM_TB + 128 * hh + 32 * hs + r] = (rp[0] + rp[TRW]) + (rp[2 * TRW] + rp[3 * TRW]);
    }
  }
}

void launch_matvec_sym(hipStream_t st, const double* tiles, const int64_t* row_off, const int* items_dev, int nitems, const double* xt, int kcols,
                       double* slabD, double* slabT, int npair, int64_t xt_gstride, int64_t slabD_gstride, int64_t slabT_gstride) {
  // kcols <= 16 * npair block columns are in use; npair = 2 runs two 16-column groups as paired workgroups
  hipLaunchKernelGGL(matvec_sym8_kernel<false>, dim3(nitems * npair), dim3(512), 0, st, tiles, row_off, items_dev, xt, slabD, slabT, kcols,
                     npair, xt_gstride, slabD_gstride, slabT_gstride, OpParams{}, (int64_t)0);
}
void launch_matvec_sym_generated(hipStream_t st, OpParams op, int64_t n, const int* items_dev, int nitems, const double* xt, int kcols,
                                 double* slabD, double* slabT, int npair, int64_t xt_gstride, int64_t slabD_gstride,
                                 int64_t slabT_gstride) {
  hipLaunchKernelGGL(matvec_sym8_kernel<true>, dim3(nitems * npair), dim3(512), 0, st, (const double*)nullptr, (const int64_t*)nullptr, items_dev, xt, slabD,
                     slabT, kcols, npair, xt_gstride, slabD_gstride, slabT_gstride, op, n);
}

// W[J*256 + r, col] = sum over runs of block row J of slabD + sum over I > J of slabT(I, J), fixed order.  With several
// ranks (owned != nullptr) only the block rows this rank owns contribute and the result is this rank's PARTIAL of the
// whole product, laid out for the reduce-scatter that follows: [rank p][column][row of p's slab] (chunk_rows = nslab).
__global__ __launch_bounds__(256) void sym_reduce_kernel(const double* __restrict__ slabD, const double* __restrict__ slabT,
                                                         const int* __restrict__ row_item_begin, const int64_t* __restrict__ owned,
                                                         int nb, int ncol16, int64_t nloc, int k, double* __restrict__ dst, int64_t ldd,
                                                         int64_t chunk_rows, int64_t total_rows, int accumulate) {
  const int J = blockIdx.x, col = blockIdx.y, r = threadIdx.x;
  if (col >= k) return;
  double sum = 0.0;
  for (int it = row_item_begin[J]; it < row_item_begin[J + 1]; ++it)
    sum += slabD[((int64_t)it * ncol16 + col) * SYM_TB + r];
  // four interleaved partial sums (fixed order, so still reproducible): four loads in flight per thread
  auto zt = [&](int I) { return (!owned || owned[I] >= 0) ? slabT[((((int64_t)I * (I - 1) / 2 + J) * ncol16) + col) * SYM_TB + r] : 0.0; };
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int I = J + 1;
  for (; I + 3 < nb; I += 4) { s0 += zt(I); s1 += zt(I + 1); s2 += zt(I + 2); s3 += zt(I + 3); }
  for (; I < nb; ++I) s0 += zt(I);
  sum += (s0 + s1) + (s2 + s3);
  int64_t row = (int64_t)J * SYM_TB + r;
  if (chunk_rows > 0) {
    if (row < total_rows) dst[(row / chunk_rows) * (chunk_rows * k) + (int64_t)col * chunk_rows + row % chunk_rows] = sum;
  } else {
    dst[(int64_t)col * ldd + row] = row < nloc ? (accumulate ? dst[(int64_t)col * ldd + row] + sum : sum) : 0.0;
  }
}

void launch_sym_reduce(hipStream_t st, const double* slabD, const double* slabT, const int* row_item_begin_dev, const int64_t* owned,
                       int nb, int64_t nloc, int k, double* dst, int64_t ldd, int64_t chunk_rows, int64_t total_rows, bool accumulate) {
  hipLaunchKernelGGL(sym_reduce_kernel, dim3(nb, k), dim3(256), 0, st, slabD, slabT, row_item_begin_dev, owned, nb, 16, nloc, k,
                     dst, ldd, chunk_rows, total_rows, accumulate && chunk_rows == 0 ? 1 : 0);
}

// ---- storage helpers ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void generate_sym_tiles_kernel(double* __restrict__ row_tiles, int I, int64_t n, uint64_t seed,
                                                                 double sparsity, int use_diag, double diag_val) {
  // blockIdx.x = tile column J of block row I, blockIdx.y = 16 columns of the tile; thread = row
  const int64_t J = blockIdx.x;
  double* tile = row_tiles + J * (int64_t)(SYM_TB * SYM_TB);
  int64_t gi = (int64_t)I * SYM_TB + threadIdx.x;
  for (int cc = 0; cc < 16; ++cc) {
    int64_t lc = blockIdx.y * 16 + cc, gj = J * SYM_TB + lc;
    double v = 0.0;
    if (gi < n && gj < n) v = dav_hashed_entry(seed, sparsity, use_diag, diag_val, gi, gj);
    tile[lc * SYM_TB + threadIdx.x] = v;
  }
}

// tiles (I, 0..I) of every block row this rank stores (row_off_host[I] >= 0)
void launch_generate_sym_tiles(hipStream_t st, double* tiles, const int64_t* row_off_host, int nb, int64_t n, uint64_t seed,
                               double sparsity, int use_diag, double diag_val) {
  for (int I = 0; I < nb; ++I)
    if (row_off_host[I] >= 0)
      hipLaunchKernelGGL(generate_sym_tiles_kernel, dim3((unsigned)(I + 1), SYM_TB / 16), dim3(SYM_TB), 0, st,
                         tiles + row_off_host[I] * (int64_t)(SYM_TB * SYM_TB), I, n, seed, sparsity, use_diag, diag_val);
}

__device__ __forceinline__ double sym_entry(const double* tiles, const int64_t* __restrict__ row_off, int64_t i, int64_t j) {
  if (i < j) { int64_t t = i; i = j; j = t; }
  int I = (int)(i / SYM_TB), J = (int)(j / SYM_TB);
  if (row_off[I] < 0) return 0.0;                       // block row of another rank
  return sym_tile(tiles, row_off, I, J)[(j % SYM_TB) * SYM_TB + (i % SYM_TB)];
}

// diag[i] for the global rows [0, nrows): entries of block rows this rank does not store are written as 0 (several
// ranks: the caller sums the pieces)
__global__ void diag_sym_kernel(const double* __restrict__ tiles, const int64_t* __restrict__ row_off, int64_t n, int64_t nrows,
                                double* __restrict__ diag) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < nrows) diag[i] = i < n ? sym_entry(tiles, row_off, i, i) : 0.0;
}
void launch_diag_sym(hipStream_t st, const double* tiles, const int64_t* row_off, int64_t n, int64_t nrows, double* diag) {
  hipLaunchKernelGGL(diag_sym_kernel, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, st, tiles, row_off, n, nrows, diag);
}

__global__ void gather_columns_sym_kernel(const double* __restrict__ tiles, const int64_t* __restrict__ row_off, int64_t n,
                                          int64_t nrows_pad, const int64_t* __restrict__ idx, double* __restrict__ dst, int64_t ldd) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int c = blockIdx.y;
  if (i < nrows_pad) dst[(int64_t)c * ldd + i] = i < n ? sym_entry(tiles, row_off, i, idx[c]) : 0.0;
}
void launch_gather_columns_sym(hipStream_t st, const double* tiles, const int64_t* row_off, int64_t n, int64_t nrows_pad,
                               const int64_t* idx_dev, int k, double* dst, int64_t ldd) {
  hipLaunchKernelGGL(gather_columns_sym_kernel, dim3((unsigned)((nrows_pad + 255) / 256), k), dim3(256), 0, st, tiles, row_off, n,
                     nrows_pad, idx_dev, dst, ldd);
}

// The same columns from the tiles of SEVERAL ranks: every rank writes what its block rows hold of columns idx[0:k] (zero where
// another rank stores the entry) over all `total_rows` rows, in the reduce-scatter layout of the sweeps
// ([rank][column][row of the rank's slab], chunk_rows = rows per slab); the sum over the ranks is exact - one owner per entry
__global__ void gather_columns_sym_rs_kernel(const double* __restrict__ tiles, const int64_t* __restrict__ row_off, int64_t n,
                                             int64_t chunk_rows, int64_t total_rows, const int64_t* __restrict__ idx, int k,
                                             double* __restrict__ dst) {
  int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int c = blockIdx.y;
  if (row < total_rows)
    dst[(row / chunk_rows) * (chunk_rows * k) + (int64_t)c * chunk_rows + row % chunk_rows] = row < n ? sym_entry(tiles, row_off, row, idx[c]) : 0.0;
}
void launch_gather_columns_sym_rs(hipStream_t st, const double* tiles, const int64_t* row_off, int64_t n, int64_t chunk_rows,
                                  int64_t total_rows, const int64_t* idx_dev, int k, double* dst) {
  hipLaunchKernelGGL(gather_columns_sym_rs_kernel, dim3((unsigned)((total_rows + 255) / 256), k), dim3(256), 0, st, tiles, row_off, n,
                     chunk_rows, total_rows, idx_dev, k, dst);
}

__global__ void entries_sym_kernel(const double* __restrict__ tiles, const int64_t* __restrict__ row_off, const int64_t* __restrict__ idx, int k,
                                   double* __restrict__ h0) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < k * k) h0[t] = sym_entry(tiles, row_off, idx[t % k], idx[t / k]);
}
void launch_entries_sym(hipStream_t st, const double* tiles, const int64_t* row_off, const int64_t* idx_dev, int k, double* h0) {
  hipLaunchKernelGGL(entries_sym_kernel, dim3((unsigned)((k * k + 255) / 256)), dim3(256), 0, st, tiles, row_off, idx_dev, k, h0);
}

// Upload path of a host matrix into symmetric tiles: block column J arrives as ONE panel (rows J*256 .. n of its <= 256
// columns, column-major with leading dimension ldp - a 2-D copy with long rows, which is what PCIe moves at full rate; a
// tile-by-tile copy has 2 KB rows and runs at a third of it) and is cut into the tiles (I >= J, J) this rank stores.
__global__ __launch_bounds__(256) void retile_panel_kernel(const double* __restrict__ panel, int64_t ldp, int64_t nrows, int ncols, int J,
                                                           const int64_t* __restrict__ row_off, double* __restrict__ tiles) {
  const int I = J + blockIdx.x;
  if (row_off[I] < 0) return;                                  // block row of another rank
  double* tile = tiles + (row_off[I] + J) * (int64_t)(SYM_TB * SYM_TB);
  const int64_t r = (int64_t)blockIdx.x * SYM_TB + threadIdx.x;  // row inside the panel
  for (int c = blockIdx.y * 16; c < blockIdx.y * 16 + 16; ++c)
    tile[(int64_t)c * SYM_TB + threadIdx.x] = (r < nrows && c < ncols) ? panel[r + (int64_t)c * ldp] : 0.0;
}
void launch_retile_panel(hipStream_t st, const double* panel, int64_t ldp, int64_t nrows, int ncols, int J, int nb,
                         const int64_t* row_off, double* tiles) {
  hipLaunchKernelGGL(retile_panel_kernel, dim3((unsigned)(nb - J), SYM_TB / 16), dim3(SYM_TB), 0, st, panel, ldp, nrows, ncols, J, row_off, tiles);
}
