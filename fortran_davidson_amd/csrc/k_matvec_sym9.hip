// K1s, super-row schedule - the symmetric-tiled sweep with R block rows per workgroup (R = 2 or 4).
//
// What the schedule is for.  The sweep of k_matvec_sym.hip (one block row per workgroup) writes one transposed
// partial Z (256 tile columns x k) per TILE and reads it back in the reduction kernel.  Measured on MI355X,
// N=200000, k=8 (same box): sweep without those writes 26.1-26.4 ms, with them 28.8 ms, with the same store
// instructions carrying one lane each 26.9 ms, with the stores hitting an 8 MB window 27.9 ms - it is the written
// BYTES that cost, about 2.5x what a read byte costs, and neither staging, non-temporal stores nor fewer store
// events change that.  So the volume has to go: here a workgroup owns R vertically adjacent block rows, its eight
// waves always work on the SAME 16 NCG tile columns (NCG = 4 / R column groups) of R tiles at once - wave (w, hh)
// keeps its fixed 128-row slice hh of the 256 R rows for the whole run - and the transposed partials of the R
// tiles are summed on chip before they leave it: Z volume / R, and with it the bytes the reduction kernel reads.
//
//   R = 4: k <= 8  (X_I of 1024 rows x 8 block columns is what fits the LDS next to the rest)
//   R = 2: k <= 16 (and the paired 32-column launches)
//
// Per unit (128 rows x 16 columns per wave, four 32-row half-steps; same register ring, same LDS transposition
// and the same MFMA interleave as matvec_sym8_kernel): every wave writes its 16 x 16 transposed partial to LDS,
// one workgroup barrier, then the 2R waves of a column group sum disjoint parts of it in fixed order (bitwise
// reproducible) into a 64-column stage that leaves the chip as 512-byte rows.  The direct partials need no
// cross-wave sum over row slices at all (R = 4: none whatsoever; R = 2: two column groups, at the end of the run).
//
// Measured dead ends of this kernel (N=200000, random X, same box): without the per-unit barrier 28.9 against 28.3-28.6 ms
// at k = 8 and no change at k = 16 / 32 - the barrier is not what the waves wait for; the LDS transposition of half-step
// hs + 1 issued before the MFMAs of half-step hs (software pipeline, second set of Gram registers, 256 VGPRs) 54.0
// against 50.7 ms at k = 32 and 32.5 against 30.1 ms at k = 16 - it costs a half-step of global-load lookahead.
// 32 columns as four groups of EIGHT on the R = 4 / 4x4x4 path (tile reads shared by four workgroups) 70.6 against 53.0 ms: every
// group repeats the LDS transposition of the tile, and that, not the MFMA count, is what the extra groups multiply.
//
// Tiles that do not exist for a wave (above the diagonal inside the R x R diagonal super block, or block rows past
// the end of the matrix) are replaced by a stored tile of the same super row and masked: the B operand of the direct
// product and the transposed partial are multiplied by 0.
#include "kernels.h"
#include <cstdlib>
#include <type_traits>

// row_off[I] = first tile of block row I in this rank's storage (k_matvec_sym.hip)
__device__ __forceinline__ const double* sym9_tile(const double* tiles, const int64_t* __restrict__ row_off, int I, int J) {
  return tiles + (row_off[I] + J) * (int64_t)(SYM_TB * SYM_TB);
}

// F32: the tiles are an fp32 copy of the stored operator (mixed-precision inner sweeps of the GJD correction):
// half the bytes per sweep, entries widened to fp64 in registers, products and sums in fp64 as before.
// M4 (k <= 8 only, i.e. R = 4): the products run on v_mfma_f64_4x4x4_4b_f64 - four independent 4x4x4 blocks per
// instruction, 16.3 cycles (profiles/ubench/r02_mfma4x4.log) - instead of the 16-wide v_mfma_f64_16x16x4_f64, whose
// block-column half 8..15 multiplies zeros at k <= 8: the same 64 matrix entries per instruction, half the pipe time,
// half the accumulator registers.  Lane layout (measured): A lane = i + 4 blk + 16 k, B lane = j + 4 blk + 16 k,
// D lane = j + 4 blk + 16 i; with lane = c + 16 g that is i / j = c & 3, blk = c >> 2, k (or D's i) = g - the natural
// (direct) and the Gram (transposed) register layouts of this kernel fit as they are, the four blocks being four groups
// of rows (direct) or of tile columns (transposed); the B operand is X restricted to block columns (c & 3) + 4 half,
// the same for every blk.  Why it pays although the sweep is HBM-bound: on real data the matrix pipe's power sets the
// clock - same box, N=200000, k=8, random X: 28.5 ms with every second MFMA removed 27.1 ms.
__device__ __forceinline__ double mfma4_f64(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

// HARN (with GEN): the generated operator is the reference's matrix-free test operator (src/tests/test_utils.f90:72-116,
// src/benchmark_free.f90:38-63: cos / sin (log (sqrt (atan2 (e_lo, e_hi)))) * 1e-4 from a table of exp(real(i) / real(n))), every
// symmetric pair evaluated ONCE and used for both products.  HARN = 1 / 2: the cos / sin operator.  Round 6: the entries are a function
// of ONE variable (common.h: dav_harness_poly - a table of 2 log e_i, one subtraction, one addition and 17 / 19 FMAs per entry instead of
// atan2 + sqrt + log + cos: ~20 VALU instructions against ~395); the wave keeps the table values of its 128 rows in registers for the
// whole work item and fetches the 16 column values of a unit one unit ahead.  Tiles strictly below the diagonal inside the matrix take
// that path, the diagonal tile and the ragged last block row the same polynomial behind bounds and diagonal tests.  HARN = 3: the formula
// as written, four library calls per entry (an engine created with DAV_HARNESS_LIBM=1, or a table that is not exp(real(i) / real(n))).
template <int R, bool GEN, bool F32, bool M4, int HARN = 0>
__global__ __launch_bounds__(512, 1) void matvec_sym9_kernel(const void* __restrict__ tiles_v, const int64_t* __restrict__ row_off,
                                                             const int* __restrict__ items,
                                                             const int* __restrict__ zslot_begin, const double* __restrict__ xt,
                                                             double* __restrict__ slabD, double* __restrict__ slabT, int kcols,
                                                             int npair, int64_t xt_gstride, int64_t slabD_gstride,
                                                             int64_t slabT_gstride, int nb, OpParams op, int64_t n) {
  const double* tiles = static_cast<const double*>(tiles_v);
  const float* tiles32 = static_cast<const float*>(tiles_v);
  (void)tiles; (void)tiles32;
  constexpr int NCG = 4 / R;            // column groups of 16 tile columns per batch
  constexpr int NRS = 2 * R;            // 128-row slices
  constexpr int BW = 16 * NCG;          // tile columns per batch
  constexpr int UPJ = SYM_TB / BW;      // units per tile column
  constexpr int UPS = 64 / BW;          // units per 64-column strip of the stage
  constexpr int DEPTH = 3;              // half-steps of load lookahead (ring of 4 slots)
  constexpr int TRS = 34, TRW = 16 * TRS, RS = 33;
  constexpr int XROWS = R == 4 ? 8 : 16;             // block columns of X_I kept in LDS
  constexpr int XT = R * SYM_TB + 2;                 // padded row stride of the transposed X_I copy
  constexpr int ZW = 64, ZS = ZW + 2;                // stage strip: 64 tile columns, padded
  constexpr int EPW = 256 / NRS;                     // entries of a 16 x 16 partial each wave of a column group sums
  __shared__ __attribute__((aligned(16))) double tr[8 * TRW];
  __shared__ __attribute__((aligned(16))) double xsT[(XROWS + (XROWS < 16)) * XT];   // R = 4: a row of zeros for lanes c >= 8
  __shared__ __attribute__((aligned(16))) double zred[2][8][256];   // [unit parity][wave][f64x4 per lane]
  __shared__ __attribute__((aligned(16))) double zst[2][16 * ZS];   // [strip parity][block column][tile column]
  static_assert(TRW >= 16 * RS, "the end-of-run exchange reuses the transposition scratch");

  // npair (2 or 4) 16-column groups in one launch: the workgroups of one work item are 8 apart in the grid - consecutive
  // workgroups go to consecutive XCDs, so they share an XCD and its L2 - and stream the same tiles at the same pace
  int item, grp;
  if (npair > 1) {
    const int span = 8 * npair;
    const int nfull = (int)(gridDim.x / span) * span;
    if ((int)blockIdx.x < nfull) {
      item = (blockIdx.x / span) * 8 + (blockIdx.x % 8);
      grp = (blockIdx.x / 8) % npair;
    } else {
      item = nfull / npair + (blockIdx.x - nfull) / npair;
      grp = (blockIdx.x - nfull) % npair;
    }
  } else {
    item = blockIdx.x;
    grp = 0;
  }
  xt += grp * xt_gstride;
  slabD += grp * slabD_gstride;
  slabT += grp * slabT_gstride;
  kcols = kcols - 16 * grp < 16 ? kcols - 16 * grp : 16;

  // the wave index decides the block row and with it every tile address and every branch of the generated variant:
  // made wave-uniform for the compiler (scalar registers, real branches)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w = wave % NCG, hh = wave / NCG;
  const int c = lane & 15, g = lane >> 4;
  // work item: (super row, first tile column, end tile column, slab slot)
  const int S = items[4 * item], J0 = items[4 * item + 1], J1 = items[4 * item + 2];
  const int I0 = S * R;
  const int Imax = I0 + R - 1 < nb - 1 ? I0 + R - 1 : nb - 1;     // last block row of the super row that exists
  const int I = I0 + (hh >> 1);                                   // this wave's block row (may lie past the end)
  const int rhalf = hh & 1;

  // X_I of the whole super row, transposed ([block column][row]); rows past the matrix are zero
  for (int e = threadIdx.x; e < R * SYM_TB * XROWS; e += 512) {
    const int bc = e % XROWS, r = e / XROWS;
    const int64_t grow = (int64_t)I0 * SYM_TB + r;
    xsT[bc * XT + r] = grow < (int64_t)nb * SYM_TB ? xt[grow * 16 + bc] : 0.0;
  }
  // block columns the LDS copy does not hold (R = 4: 8..15) read zeros, as they would from Xt: operands that carry
  // data cost matrix-pipe power - with the wanted columns mirrored into the unused half of the 16-wide MFMA the k = 8
  // sweep ran 4 % slower on random X (28.6 against 27.5 ms at N=200000) although nothing reads those results
  if constexpr (XROWS < 16)
    for (int e = threadIdx.x; e < XT; e += 512) xsT[XROWS * XT + e] = 0.0;
  __syncthreads();

  static_assert(!M4 || R == 4, "the 4x4x4 path serves k <= 8, which is what R = 4 means");
  f64x4 acc[4][2];                 // 16-wide MFMA: [half-step][row parity] x (4 row groups in the f64x4)
  double acc4[4][2][2];            // 4x4x4 MFMA:  [half-step][row parity][block-column half]
#pragma unroll
  for (int hs = 0; hs < 4; ++hs) {
    acc[hs][0] = f64x4{0.0, 0.0, 0.0, 0.0}; acc[hs][1] = f64x4{0.0, 0.0, 0.0, 0.0};
    acc4[hs][0][0] = acc4[hs][0][1] = acc4[hs][1][0] = acc4[hs][1][1] = 0.0;
  }

  const int nunits = (J1 - J0) * UPJ;
  const int nsteps = nunits * 4;
  double* tw = tr + wave * TRW;

  // register ring of the tile loads: raw fp32 pairs when the tiles are fp32 (widened where they are consumed, so that
  // the loads stay three half-steps ahead), fp64 pairs otherwise
  using RingT = std::conditional_t<F32, float2, f64x2>;
  RingT ra[4][4];
  const uint64_t seedmix = op.seed * 0x9E3779B97F4A7C15ull;
  const double gscale = op.sparsity * (1.0 / 9007199254740992.0);
  // tile this wave works on in tile column J: its own if it is stored, else the (stored) tile of block row Imax
  auto tile_row = [&](int J) { return (I <= Imax && J <= I) ? I : Imax; };
  auto load_hs = [&](int s, RingT (&a)[4]) {
    s = s < nsteps ? s : nsteps - 1;
    const int q = s >> 2, hs = s & 3;
    const int J = J0 + q / UPJ, col = (q % UPJ) * BW + w * 16;
    const int Ie = tile_row(J);
    if constexpr (GEN) {
      const int64_t gi = (int64_t)Ie * SYM_TB + 128 * rhalf + 32 * hs + 2 * c;
      const int64_t gj = (int64_t)J * SYM_TB + col + g;
      if constexpr (HARN != 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t cj = gj + 4 * u;
          if constexpr (HARN == 3) {
            a[u].x = (gi < n && cj < n) ? dav_harness_entry_libm(op.e_table, op.trig, gi, cj) : 0.0;
            a[u].y = (gi + 1 < n && cj < n) ? dav_harness_entry_libm(op.e_table, op.trig, gi + 1, cj) : 0.0;
          } else {
            a[u].x = (gi < n && cj < n) ? dav_harness_entry_poly(op.l2_table, HARN - 1, gi, cj) : 0.0;
            a[u].y = (gi + 1 < n && cj < n) ? dav_harness_entry_poly(op.l2_table, HARN - 1, gi + 1, cj) : 0.0;
          }
        }
      } else if (J < Ie && ((int64_t)Ie + 1) * SYM_TB <= n) {
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
    } else if constexpr (F32) {
      const float* ad = tiles32 + (row_off[Ie] + J) * (int64_t)(SYM_TB * SYM_TB) + (int64_t)(col + g) * SYM_TB + 128 * rhalf + 32 * hs + 2 * c;
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const float2*>(ad + (int64_t)(4 * u) * SYM_TB);
    } else {
      const double* ad = sym9_tile(tiles, row_off, Ie, J) + (int64_t)(col + g) * SYM_TB + 128 * rhalf + 32 * hs + 2 * c;
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const f64x2*>(ad + (int64_t)(4 * u) * SYM_TB);
    }
  };
  // B operand of the direct product for unit q (X_J rows of the unit's 16 tile columns), zero where this wave has no tile
  // b[u][0]: 16-wide MFMA, block column c.  b[u][0..1]: 4x4x4 MFMA, block columns (c & 3) and (c & 3) + 4.
  auto load_b = [&](int q, double (&b)[4][2]) {
    q = q < nunits ? q : nunits - 1;
    const int J = J0 + q / UPJ, col = (q % UPJ) * BW + w * 16;
    const double dm = (I <= Imax && J <= I) ? 1.0 : 0.0;
    const double* xj = xt + ((int64_t)J * SYM_TB + col + g) * 16 + (M4 ? (c & 3) : c);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      b[u][0] = xj[(4 * u) * 16] * dm;
      if constexpr (M4) b[u][1] = xj[(4 * u) * 16 + 4] * dm;
    }
  };
  // strip st of the run (64 tile columns; four per tile column): stage -> slabT slot (S, J) = [16 block columns][256 tile columns]
  const int64_t zbase = zslot_begin[S];
  auto flush_strip = [&](int st) {
    const int J = J0 + (st >> 2);
    if (J >= Imax) return;                           // no block row of the super row lies below tile column J
    const double* zs = zst[st & 1];
    double* outT = slabT + (zbase + J) * 16 * SYM_TB + (st & 3) * ZW;
    for (int e = threadIdx.x; e < kcols * (ZW / 2); e += 512) {
      const int bc = e >> 5, pr = e & 31;
      *reinterpret_cast<f64x2*>(outT + bc * SYM_TB + 2 * pr) = *reinterpret_cast<const f64x2*>(zs + bc * ZS + 2 * pr);
    }
  };

  // harness operator, polynomial form: 2 log e of this wave's rows (rows 2c, 2c + 1 of the four half-steps; a wave without a block row
  // generates - and masks - the tile of block row Imax, as the stored kernels read it) and of a unit's tile columns g + 4u
  const int Irow = I <= Imax ? I : Imax;
  f64x2 lrow[4];
  double lcol[4], lcoln[4];
  auto load_lcol = [&](int q, double (&lc)[4]) {
    if constexpr (HARN == 1 || HARN == 2) {
      q = q < nunits ? q : nunits - 1;
      const int J = J0 + q / UPJ, col = (q % UPJ) * BW + w * 16;
      const double* lp = op.l2_table + (int64_t)J * SYM_TB + col + g;
#pragma unroll
      for (int u = 0; u < 4; ++u) lc[u] = lp[4 * u];
    }
  };
  if constexpr (HARN == 1 || HARN == 2) {
#pragma unroll
    for (int hs = 0; hs < 4; ++hs) lrow[hs] = *reinterpret_cast<const f64x2*>(op.l2_table + (int64_t)Irow * SYM_TB + 128 * rhalf + 32 * hs + 2 * c);
    load_lcol(0, lcol);
  }
  // half-step hs of unit q of the harness operator -> rows 2c, 2c + 1 of tile columns 4u + g
  auto harness_hs = [&](int q, int hs, f64x2 (&a)[4]) {
    if constexpr (HARN != 0) {
      const int J = J0 + q / UPJ;
      if (HARN != 3 && J < Irow && ((int64_t)Irow + 1) * SYM_TB <= n) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a[u].x = dav_harness_poly<HARN == 2>(lrow[hs].x, lcol[u]);
          a[u].y = dav_harness_poly<HARN == 2>(lrow[hs].y, lcol[u]);
        }
      } else {
        RingT now[4];
        load_hs(q * 4 + hs, now);
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = f64x2{(double)now[u].x, (double)now[u].y};
      }
    }
  };

  double b[4][2], bn[4][2];
  load_b(0, b);
  // (the transcendental test operator is evaluated where it is consumed: there is no load latency to run ahead of, and the ring's
  // 64 registers are what its four library calls per entry need - with the ring the kernel spilled 150-230 bytes per lane)
  if constexpr (HARN == 0) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load_hs(d, ra[d]);
  }

  for (int q = 0; q < nunits; ++q) {
    load_b(q + 1, bn);
    load_lcol(q + 1, lcoln);
    const int J = J0 + q / UPJ;
    const double zm = (I <= Imax && J < I) ? 1.0 : 0.0;
    int xoff = (M4 ? (c & 3) : (c < XROWS ? c : XROWS)) * XT + 128 * hh + 4 * g;   // opaque: keeps the X_I reads inside the loop
    asm volatile("" : "+v"(xoff));
    const double* xw = xsT + xoff;
    f64x4 zc[4];
    double zc4[2][4];                 // [block-column half][row within the group of four]
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) { zc[s4] = f64x4{0.0, 0.0, 0.0, 0.0}; zc4[0][s4] = zc4[1][s4] = 0.0; }
#pragma unroll
    for (int hs = 0; hs < 4; ++hs) {
      f64x2 a[4];
      if constexpr (HARN != 0) {
        harness_hs(q, hs, a);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = f64x2{(double)ra[hs][u].x, (double)ra[hs][u].y};
        load_hs(q * 4 + hs + DEPTH, ra[(hs + DEPTH) & 3]);
      }
      f64x2 p[2][2], xb[2][2], xb1[2][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) *reinterpret_cast<f64x2*>(tw + (4 * u + g) * TRS + 2 * c) = a[u];
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) {
        p[ib][0] = *reinterpret_cast<const f64x2*>(tw + c * TRS + 16 * ib + 4 * g);
        p[ib][1] = *reinterpret_cast<const f64x2*>(tw + c * TRS + 16 * ib + 4 * g + 2);
        xb[ib][0] = *reinterpret_cast<const f64x2*>(xw + 32 * hs + 16 * ib);
        xb[ib][1] = *reinterpret_cast<const f64x2*>(xw + 32 * hs + 16 * ib + 2);
        if constexpr (M4) {           // block columns (c & 3) + 4
          xb1[ib][0] = *reinterpret_cast<const f64x2*>(xw + 4 * XT + 32 * hs + 16 * ib);
          xb1[ib][1] = *reinterpret_cast<const f64x2*>(xw + 4 * XT + 32 * hs + 16 * ib + 2);
        }
      }
      if constexpr (M4) {
        // direct: D[row 2 (4 blk + i) + parity, block column j + 4 half] += sum_k A[row, tile column 4 u + k] X[.., ..]
        // transposed: D[tile column 4 blk + i, block column j + 4 half] += sum_k P[row 16 ib + 4 k + r, tile column] X_I[row, ..]
        // twelve independent accumulator chains, direct and transposed alternating
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
          acc4[hs][0][0] = mfma4_f64(a[2 * ib].x, b[2 * ib][0], acc4[hs][0][0]);
          zc4[0][0] = mfma4_f64(p[ib][0].x, xb[ib][0].x, zc4[0][0]);
          acc4[hs][0][1] = mfma4_f64(a[2 * ib].x, b[2 * ib][1], acc4[hs][0][1]);
          zc4[1][0] = mfma4_f64(p[ib][0].x, xb1[ib][0].x, zc4[1][0]);
          acc4[hs][1][0] = mfma4_f64(a[2 * ib].y, b[2 * ib][0], acc4[hs][1][0]);
          zc4[0][1] = mfma4_f64(p[ib][0].y, xb[ib][0].y, zc4[0][1]);
          acc4[hs][1][1] = mfma4_f64(a[2 * ib].y, b[2 * ib][1], acc4[hs][1][1]);
          zc4[1][1] = mfma4_f64(p[ib][0].y, xb1[ib][0].y, zc4[1][1]);
          acc4[hs][0][0] = mfma4_f64(a[2 * ib + 1].x, b[2 * ib + 1][0], acc4[hs][0][0]);
          zc4[0][2] = mfma4_f64(p[ib][1].x, xb[ib][1].x, zc4[0][2]);
          acc4[hs][0][1] = mfma4_f64(a[2 * ib + 1].x, b[2 * ib + 1][1], acc4[hs][0][1]);
          zc4[1][2] = mfma4_f64(p[ib][1].x, xb1[ib][1].x, zc4[1][2]);
          acc4[hs][1][0] = mfma4_f64(a[2 * ib + 1].y, b[2 * ib + 1][0], acc4[hs][1][0]);
          zc4[0][3] = mfma4_f64(p[ib][1].y, xb[ib][1].y, zc4[0][3]);
          acc4[hs][1][1] = mfma4_f64(a[2 * ib + 1].y, b[2 * ib + 1][1], acc4[hs][1][1]);
          zc4[1][3] = mfma4_f64(p[ib][1].y, xb1[ib][1].y, zc4[1][3]);
        }
      } else {
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
          acc[hs][0] = mfma_f64(a[2 * ib].x, b[2 * ib][0], acc[hs][0]);
          zc[0] = mfma_f64(p[ib][0].x, xb[ib][0].x, zc[0]);
          acc[hs][1] = mfma_f64(a[2 * ib].y, b[2 * ib][0], acc[hs][1]);
          zc[1] = mfma_f64(p[ib][0].y, xb[ib][0].y, zc[1]);
          acc[hs][0] = mfma_f64(a[2 * ib + 1].x, b[2 * ib + 1][0], acc[hs][0]);
          zc[2] = mfma_f64(p[ib][1].x, xb[ib][1].x, zc[2]);
          acc[hs][1] = mfma_f64(a[2 * ib + 1].y, b[2 * ib + 1][0], acc[hs][1]);
          zc[3] = mfma_f64(p[ib][1].y, xb[ib][1].y, zc[3]);
        }
      }
    }
    if constexpr (M4) {
      // z4[half]: tile column col + 4 (c >> 2) + g, block column (c & 3) + 4 half, summed over this wave's 128 rows
      double* zr = &zred[q & 1][wave][0];
      zr[lane] = ((zc4[0][0] + zc4[0][1]) + (zc4[0][2] + zc4[0][3])) * zm;
      zr[64 + lane] = ((zc4[1][0] + zc4[1][1]) + (zc4[1][2] + zc4[1][3])) * zm;
    } else {
      // z[reg]: tile column col + g + 4 reg, block column c, summed over this wave's 128 rows
      const f64x4 z = ((zc[0] + zc[1]) + (zc[2] + zc[3])) * zm;
      double* zr = &zred[q & 1][wave][0];
      *reinterpret_cast<f64x2*>(zr + 2 * lane) = f64x2{z[0], z[1]};
      *reinterpret_cast<f64x2*>(zr + 128 + 2 * lane) = f64x2{z[2], z[3]};
    }
    __syncthreads();
    // this barrier also publishes the stage writes of unit q - 1: the previous strip is complete
    if (q % UPS == 0 && q > 0) flush_strip(q / UPS - 1);
    // the 2R waves of column group w sum disjoint parts of the 16 x 16 partial, slices in fixed order
    constexpr int EPWU = M4 ? 128 / NRS : EPW;          // the 4x4x4 path exchanges 16 x 8 partials
    for (int e = hh * EPWU + lane; e < (hh + 1) * EPWU; e += 64) {
      double s = zred[q & 1][w][e];
#pragma unroll
      for (int sl = 1; sl < NRS; ++sl) s += zred[q & 1][sl * NCG + w][e];
      int bc, tcol;
      if constexpr (M4) {
        const int half = e >> 6, ln = e & 63, cc = ln & 15, gg = ln >> 4;
        bc = (cc & 3) + 4 * half;
        tcol = (q % UPJ) * BW + w * 16 + 4 * (cc >> 2) + gg;
      } else {
        const int half = e >> 7, ln = (e & 127) >> 1, j = e & 1;
        const int gg = ln >> 4, reg = 2 * half + j;
        bc = ln & 15;
        tcol = (q % UPJ) * BW + w * 16 + gg + 4 * reg;
      }
      zst[(q / UPS) & 1][bc * ZS + (tcol & (ZW - 1))] = s;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { b[u][0] = bn[u][0]; b[u][1] = bn[u][1]; }
    if constexpr (HARN == 1 || HARN == 2) {
#pragma unroll
      for (int u = 0; u < 4; ++u) lcol[u] = lcoln[u];
    }
  }
  __syncthreads();
  flush_strip(nunits / UPS - 1);

  // end of the run: the direct partials of every row slice, summed over the column groups, one 32-row half-step at a time
  double* outD = slabD + (int64_t)items[4 * item + 3] * R * 16 * SYM_TB;
#pragma unroll
  for (int hs = 0; hs < 4; ++hs) {
    __syncthreads();
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      if constexpr (M4) {
        tw[(c & 3) * RS + 2 * (4 * (c >> 2) + g) + par] = acc4[hs][par][0];
        tw[((c & 3) + 4) * RS + 2 * (4 * (c >> 2) + g) + par] = acc4[hs][par][1];
      } else {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) tw[c * RS + 2 * (g + 4 * reg) + par] = acc[hs][par][reg];
      }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NRS; ++t) {
      const int e = threadIdx.x + 512 * t;              // (slice, block column, row of the half-step)
      const int sl = e >> 9, bc = (e >> 5) & 15, r = e & 31;
      const double* rp = tr + (sl * NCG) * TRW + bc * RS + r;
      double sum = rp[0];
#pragma unroll
      for (int ww = 1; ww < NCG; ++ww) sum += rp[ww * TRW];
      if (bc < kcols) outD[((int64_t)(sl >> 1) * 16 + bc) * SYM_TB + 128 * (sl & 1) + 32 * hs + r] = sum;
    }
  }
}

void launch_matvec_sym9(hipStream_t st, int R, bool gen, const void* tiles, bool tiles_f32, const int64_t* row_off, OpParams op, int64_t n,
                        int nb, const int* items_dev, int nitems, const int* zslot_begin_dev, const double* xt, int kcols, double* slabD,
                        double* slabT, int npair, int64_t xt_gstride, int64_t slabD_gstride, int64_t slabT_gstride, bool m4) {
  dim3 grid(nitems * npair), block(512);
#define DAV_SYM9_LAUNCH(RR, GG, FF, MM, ...)                                                                                     \
  hipLaunchKernelGGL((matvec_sym9_kernel<RR, GG, FF, MM, ##__VA_ARGS__>), grid, block, 0, st, tiles, row_off, items_dev, zslot_begin_dev, xt, slabD, slabT, \
                     kcols, npair, xt_gstride, slabD_gstride, slabT_gstride, nb, op, n)
  // m4 = false (Tune::sym_mfma4 = 0): the k <= 8 sweep of a stored fp64 matrix on the 16-wide MFMA (A/B runs)
  const bool harness = gen && op.kind == DAV_KIND_HARNESS;
  const bool hsin = harness && op.trig != 0;
  const bool hlibm = harness && op.libm != 0;
  if (R == 4) {
    if (hlibm) DAV_SYM9_LAUNCH(4, true, false, true, 3);
    else if (hsin) DAV_SYM9_LAUNCH(4, true, false, true, 2);
    else if (harness) DAV_SYM9_LAUNCH(4, true, false, true, 1);
    else if (gen) DAV_SYM9_LAUNCH(4, true, false, true);
    else if (tiles_f32) DAV_SYM9_LAUNCH(4, false, true, true);
    else if (m4) DAV_SYM9_LAUNCH(4, false, false, true);
    else DAV_SYM9_LAUNCH(4, false, false, false);
  } else {
    if (hlibm) DAV_SYM9_LAUNCH(2, true, false, false, 3);
    else if (hsin) DAV_SYM9_LAUNCH(2, true, false, false, 2);
    else if (harness) DAV_SYM9_LAUNCH(2, true, false, false, 1);
    else if (gen) DAV_SYM9_LAUNCH(2, true, false, false);
    else if (tiles_f32) DAV_SYM9_LAUNCH(2, false, true, false);
    else DAV_SYM9_LAUNCH(2, false, false, false);
  }
#undef DAV_SYM9_LAUNCH
}

// fp32 copy of stored tiles (same layout): the operand of the mixed-precision inner sweeps
__global__ __launch_bounds__(256) void tiles_to_f32_kernel(const double* __restrict__ src, float* __restrict__ dst, int64_t count4) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < count4; e += (int64_t)gridDim.x * 256) {
    const f64x2 a = reinterpret_cast<const f64x2*>(src)[2 * e], b = reinterpret_cast<const f64x2*>(src)[2 * e + 1];
    reinterpret_cast<float4*>(dst)[e] = float4{(float)a.x, (float)a.y, (float)b.x, (float)b.y};
  }
}
void launch_tiles_to_f32(hipStream_t st, const double* src, float* dst, int64_t count) {
  hipLaunchKernelGGL(tiles_to_f32_kernel, dim3(256 * 32), dim3(256), 0, st, src, dst, count / 4);
}

// W[J*256 + r, col] = sum over the items of super row J / R of slabD (block row J % R of the item)
//                   + sum over the super rows S that reach below block row J of slabT(S, J), fixed order.
// Several ranks (owned != nullptr): only the super rows this rank owns contribute; the partial product goes out in the
// layout of the reduce-scatter that follows, [rank p][column][row of p's slab] (chunk_rows = nslab).
__global__ __launch_bounds__(256) void sym9_reduce_kernel(const double* __restrict__ slabD, const double* __restrict__ slabT,
                                                          const int* __restrict__ row_item_begin, const int* __restrict__ zslot_begin,
                                                          const int64_t* __restrict__ owned, const int* __restrict__ next_owned, int R, int nb,
                                                          int nsuper, int64_t nloc, int k, double* __restrict__ dst, int64_t ldd,
                                                          int64_t chunk_rows, int64_t total_rows, int accumulate) {
  const int J = blockIdx.x, col = blockIdx.y, r = threadIdx.x;
  if (col >= k) return;
  double sum = 0.0;
  const int Sown = J / R, sub = J % R;
  for (int it = row_item_begin[Sown]; it < row_item_begin[Sown + 1]; ++it)
    sum += slabD[(((int64_t)it * R + sub) * 16 + col) * SYM_TB + r];
  auto zt = [&](int S) { return slabT[(((int64_t)zslot_begin[S] + J) * 16 + col) * SYM_TB + r]; };
  // super row S holds a partial for tile column J iff its last existing block row lies below J
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  const int Send = (nb - 1 > J) ? nsuper : 0;       // the last super row is clamped to nb - 1
  if (!owned) {
    int S = (J + 1) / R;
    for (; S + 3 < Send; S += 4) { s0 += zt(S); s1 += zt(S + 1); s2 += zt(S + 2); s3 += zt(S + 3); }
    for (; S < Send; ++S) s0 += zt(S);
  } else {
    // only the super rows of this set (several ranks / a part of an operator), in ascending order, four interleaved partial sums
    int S = (J + 1) / R < nsuper ? next_owned[(J + 1) / R] : nsuper;
    while (S < Send) {
      s0 += zt(S); S = next_owned[S + 1];
      if (S >= Send) break;
      s1 += zt(S); S = next_owned[S + 1];
      if (S >= Send) break;
      s2 += zt(S); S = next_owned[S + 1];
      if (S >= Send) break;
      s3 += zt(S); S = next_owned[S + 1];
    }
  }
  sum += (s0 + s1) + (s2 + s3);
  const int64_t row = (int64_t)J * SYM_TB + r;
  if (chunk_rows > 0) {
    if (row < total_rows) dst[(row / chunk_rows) * (chunk_rows * k) + (int64_t)col * chunk_rows + row % chunk_rows] = sum;
  } else {
    dst[(int64_t)col * ldd + row] = row < nloc ? (accumulate ? dst[(int64_t)col * ldd + row] + sum : sum) : 0.0;
  }
}

void launch_sym9_reduce(hipStream_t st, const double* slabD, const double* slabT, const int* row_item_begin_dev,
                        const int* zslot_begin_dev, const int64_t* owned, const int* next_owned, int R, int nb, int64_t nloc, int k, double* dst,
                        int64_t ldd, int64_t chunk_rows, int64_t total_rows, bool accumulate) {
  const int nsuper = (nb + R - 1) / R;
  hipLaunchKernelGGL(sym9_reduce_kernel, dim3(nb, k), dim3(256), 0, st, slabD, slabT, row_item_begin_dev, zslot_begin_dev, owned, next_owned, R,
                     nb, nsuper, nloc, k, dst, ldd, chunk_rows, total_rows, accumulate && chunk_rows == 0 ? 1 : 0);
}
