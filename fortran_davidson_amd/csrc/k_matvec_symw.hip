// K1s, wide blocks - the symmetric-tiled sweep with ONE wave per SIMD and 16 NB block columns per workgroup (NB = 1, 2).
//
// Why.  v_mfma_f64_16x16x4_f64 holds the SIMD's vector issue for its whole 64 cycles (profiles/ubench/r01_mfma_f64_overlap.log:
// a wave with one in flight issues nothing else, a second wave on the SIMD one instruction per ~21 cycles), so in the
// MFMA-bound launches of a solve (32 / 64 columns) every instruction that is not an MFMA is paid for in matrix-pipe time.
// matvec_sym9_kernel<2> (16 columns per workgroup, two waves per SIMD, 216 VGPRs) issues 2.6 vector instructions besides
// each MFMA - tile loads, the LDS transposition of the tile, the X_I operand reads, 64-bit address arithmetic - and
// repeats the transposition of every tile in each of the 2 / 4 column groups of a launch: pipe 73 % busy (round 2).
// Here a workgroup is 4 waves x 512 registers.  Wave v keeps, for the whole work item,
//   - its 128-row slice of the super row (R = 2 block rows x 2 halves) x 16 NB block columns of direct partials,
//   - the X_I operand of the transposed product for those rows (it never changes within an item: no LDS, no re-reads),
// so one tile load and one LDS transposition feed 16 NB MFMAs per 32 x 16 sub-block instead of 16, the transposition of
// half-step s + 1 is issued before the MFMAs of half-step s (its latency hides behind them - the registers for that are
// what the two-wave kernel did not have), and addresses are a scalar base plus constant per-lane offsets.
// Same work items, slab layout, masking rules and fixed-order sums as matvec_sym9_kernel<2> (k_matvec_sym9.hip), so the
// same reduction kernel follows; results are bitwise reproducible run to run.
//
// Registers.  The compiler splits the 512 registers of a wave into 256 VGPRs + 256 accumulation registers and, left to
// itself, parks MFMA operands in the second half and copies them back before every use (160 v_accvgpr moves per unit).
// The MFMAs are therefore inline assembly with the register file of every operand stated: direct partials and X_I in
// accumulation registers (an MFMA reads A / B operands from either half), everything a VALU or DS instruction touches
// in VGPRs.  What the compiler does not know about an asm MFMA is handled structurally: every MFMA operand is
// produced by a load (never by a VALU instruction: the B operand of a tile that is not there is loaded from a page of
// zeros instead of multiplied by zero; a chain starts with the literal-zero form; no branches around the MFMAs, whose
// copies the compiler would put between them), and results are read by other instructions only behind
// the 18 wait states a 16-pass MFMA needs (MFMA_DRAIN).  tests/test_isa_lint.py checks the emitted code for both rules.
#include "kernels.h"
#include <cstdlib>
#include <type_traits>

#define MFMA_DRAIN "s_nop 15\n\ts_nop 3"

// D (accumulation registers) += A (VGPR) B (VGPR)
__device__ __forceinline__ void mfma_acc_vv(f64x4& d, double a, double b) {
  asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}
// the same with D in VGPRs (what does not fit the accumulation half)
__device__ __forceinline__ void mfma_v_vv(f64x4& d, double a, double b) {
  asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
// D (VGPRs) = or += A (VGPR) B (accumulation register)
__device__ __forceinline__ void mfma_v_va_first(f64x4& d, double a, double b) {
  asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, 0" : "=v"(d) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_v_va(f64x4& d, double a, double b) {
  asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
}

__device__ double symw_zero_page[16 * 16];     // B operand of a tile that is not there

template <int NB>
__global__ __launch_bounds__(256, 1) void matvec_symw_kernel(const double* __restrict__ tiles, const int64_t* __restrict__ row_off,
                                                             const int* __restrict__ items, const int* __restrict__ zslot_begin,
                                                             const double* __restrict__ xt, double* __restrict__ slabD,
                                                             double* __restrict__ slabT, int kcols, int nwg, int64_t xt_gstride,
                                                             int64_t slabD_gstride, int64_t slabT_gstride, int nb) {
  constexpr int R = 2;                  // block rows per workgroup
  constexpr int NRS = 4;                // 128-row slices = waves
  constexpr unsigned UPJ = SYM_TB / 16; // units (16 tile columns) per tile column
  constexpr unsigned UPS = 4;           // units per 64-column strip of the stage
  constexpr int DEPTH = 3;              // half-steps of load lookahead (ring of 4 slots)
  constexpr int ACC_A = NB == 2 ? 3 : 4; // half-steps whose direct partials live in accumulation registers (X_I: all of it)
  constexpr int TRS = 34, TRW = 16 * TRS, RS = 33;
  constexpr int ZW = 64, ZS = ZW + 2;   // stage strip: 64 tile columns, padded
  __shared__ __attribute__((aligned(16))) double tr[NRS * TRW];
  __shared__ __attribute__((aligned(16))) double zred[2][NRS][NB * 256];   // [unit parity][wave][group][f64x4 per lane]
  __shared__ __attribute__((aligned(16))) double zst[2][NB * 16 * ZS];      // [strip parity][block column][tile column]
  static_assert(TRW >= 16 * RS, "the end-of-run exchange reuses the transposition scratch");

  // nwg workgroups of one work item (each 16 NB columns of the block) are 8 apart in the grid: same XCD, shared L2
  int item, grp;
  if (nwg > 1) {
    const int span = 8 * nwg;
    const int nfull = (int)(gridDim.x / span) * span;
    if ((int)blockIdx.x < nfull) {
      item = (blockIdx.x / span) * 8 + (blockIdx.x % 8);
      grp = (blockIdx.x / 8) % nwg;
    } else {
      item = nfull / nwg + (blockIdx.x - nfull) / nwg;
      grp = (blockIdx.x - nfull) % nwg;
    }
  } else {
    item = blockIdx.x;
    grp = 0;
  }
  xt += (int64_t)grp * NB * xt_gstride;
  slabD += (int64_t)grp * NB * slabD_gstride;
  slabT += (int64_t)grp * NB * slabT_gstride;
  kcols -= 16 * NB * grp;               // columns of this workgroup's groups that exist: group bcb holds min(16, kcols - 16 bcb)

  const unsigned lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned c = lane & 15, g = lane >> 4;
  const int S = items[4 * item], J0 = items[4 * item + 1], J1 = items[4 * item + 2];
  const int I0 = S * R;
  const int Imax = I0 + R - 1 < nb - 1 ? I0 + R - 1 : nb - 1;     // last block row of the super row that exists
  const int I = I0 + (wave >> 1);                                 // this wave's block row (may lie past the end)
  const int rhalf = wave & 1;
  const bool have_row = I <= Imax;

  // X_I of this wave's 128 rows, in the lane layout of the transposed product's B operand: for half-step hs, 16-row block
  // ib, row pair j and parity xy the lane (k = g, column c) holds X[row 32 hs + 16 ib + 4 g + 2 j + xy, block column c].
  // A wave without a block row never uses it (its MFMAs are skipped).
  double xI[4][2][2][2][NB];
  {
    const double* xr = xt + ((int64_t)(have_row ? I : Imax) * SYM_TB + 128 * rhalf) * 16;
    const unsigned xo = (4 * g) * 16 + c;
#pragma unroll
    for (int hs = 0; hs < 4; ++hs)
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int xy = 0; xy < 2; ++xy)
#pragma unroll
            for (int bcb = 0; bcb < NB; ++bcb)
              xI[hs][ib][j][xy][bcb] = xr[bcb * xt_gstride + xo + (32 * hs + 16 * ib + 2 * j + xy) * 16];
  }

  f64x4 acc[4][2][NB];              // direct partials: [half-step][row parity][group] x (4 row groups in the f64x4)
#pragma unroll
  for (int hs = 0; hs < 4; ++hs)
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
      for (int bcb = 0; bcb < NB; ++bcb) acc[hs][par][bcb] = f64x4{0.0, 0.0, 0.0, 0.0};

  const unsigned nunits = (unsigned)(J1 - J0) * UPJ;
  double* tw = tr + wave * TRW;

  // tile loads: buffer loads - a scalar descriptor on the unit's 128 x 16 sub-block, constant per-lane offsets (rows 2c, 2c+1
  // of column 4u + g), the half-step as an immediate: no vector address arithmetic in the loop
  using u32x4 = unsigned __attribute__((ext_vector_type(4)));
  using u32x2 = unsigned __attribute__((ext_vector_type(2)));
  f64x2 ra[4][4];
  unsigned voff[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) voff[u] = ((4 * u + g) * SYM_TB + 2 * c) * (unsigned)sizeof(double);
  // tile this wave works on in tile column J: its own if it is stored, else the (stored) tile of block row Imax
  auto unit_rsrc = [&](unsigned q) {
    q = q < nunits ? q : nunits - 1;
    const int J = J0 + (int)(q / UPJ);
    const unsigned col = (q % UPJ) * 16;
    const int Ie = (have_row && J <= I) ? I : Imax;
    const double* ub = tiles + (row_off[Ie] + J) * (int64_t)(SYM_TB * SYM_TB) + col * SYM_TB + 128 * rhalf;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(ub), 0, 16 * SYM_TB * (int)sizeof(double), 0x00020000);
  };
  auto load_hs = [&](__amdgpu_buffer_rsrc_t ur, int hs, f64x2 (&a)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      a[u] = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(ur, voff[u] + hs * 32 * (unsigned)sizeof(double), 0, 0));
  };
  // B operand of the direct product for unit q: X_J rows of the unit's 16 tile columns - or, where this wave has no
  // stored tile (above the diagonal inside the diagonal super block, block row past the end), a page of zeros
  const unsigned boff = (g * 16 + c) * (unsigned)sizeof(double);
  auto load_b = [&](unsigned q, double (&b)[4][NB]) {
    q = q < nunits ? q : nunits - 1;
    const int J = J0 + (int)(q / UPJ);
    const bool stored = have_row && J <= I;
    const double* xj = stored ? xt + ((int64_t)J * SYM_TB + (q % UPJ) * 16) * 16 : symw_zero_page;
    const int gs = stored ? (int)(xt_gstride * (int64_t)sizeof(double)) : 0;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(xj), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int bcb = 0; bcb < NB; ++bcb)
        b[u][bcb] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(xr, boff + (4 * u) * 16 * (unsigned)sizeof(double), bcb * gs, 0));
  };
  // strip st of the run (64 tile columns; four per tile column): stage -> slabT slot (S, J) = [16 block columns][256 tile columns]
  const int64_t zbase = zslot_begin[S];
  auto flush_strip = [&](unsigned st) {
    const int J = J0 + (int)(st >> 2);
    if (J >= Imax) return;                           // no block row of the super row lies below tile column J
    const double* zs = zst[st & 1];
#pragma unroll
    for (int bcb = 0; bcb < NB; ++bcb) {
      const int kc = kcols - 16 * bcb < 16 ? kcols - 16 * bcb : 16;
      double* outT = slabT + bcb * slabT_gstride + (zbase + J) * 16 * SYM_TB + (st & 3) * ZW;
      for (int e = threadIdx.x; e < kc * (ZW / 2); e += 256) {
        const int bc = e >> 5, pr = e & 31;
        *reinterpret_cast<f64x2*>(outT + bc * SYM_TB + 2 * pr) = *reinterpret_cast<const f64x2*>(zs + (16 * bcb + bc) * ZS + 2 * pr);
      }
    }
  };
  // LDS transposition of a 32 x 16 sub-block: direct layout (rows 2c, 2c+1 of column 4u + g) -> Gram layout (p[ib][j] =
  // rows 16 ib + 4 g + 2 j, + 1 of column c); wave-private, DS operations of a wave execute in order: no barrier
  auto transpose = [&](const f64x2 (&a)[4], f64x2 (&p)[2][2]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<f64x2*>(tw + (4 * u + g) * TRS + 2 * c) = a[u];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      p[ib][0] = *reinterpret_cast<const f64x2*>(tw + c * TRS + 16 * ib + 4 * g);
      p[ib][1] = *reinterpret_cast<const f64x2*>(tw + c * TRS + 16 * ib + 4 * g + 2);
    }
  };

  double b0[4][NB], b1[4][NB];       // B operands of the direct product: even / odd units
  f64x2 p[2][2][2];                 // [half-step parity]: Gram-layout operands of the current and of the next half-step
  load_b(0, b0);
  __amdgpu_buffer_rsrc_t ur = unit_rsrc(0);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load_hs(ur, d, ra[d]);
  transpose(ra[0], p[0]);

  auto unit = [&](unsigned q, const double (&b)[4][NB], double (&bn)[4][NB]) {
    load_b(q + 1, bn);
    const __amdgpu_buffer_rsrc_t urn = unit_rsrc(q + 1);
    const int J = J0 + (int)(q / UPJ);
    const double zm = (have_row && J < I) ? 1.0 : 0.0;   // the tile lies below the diagonal: it feeds the transposed product
    f64x4 zc[NB][2];                           // transposed partials of the unit: [group][parity chain]
#pragma unroll
    for (int hs = 0; hs < 4; ++hs) {
      // the ring slot being refilled held half-step hs - 1, whose MFMAs have been issued
      if (hs == 0) load_hs(ur, DEPTH, ra[DEPTH & 3]);
      else load_hs(urn, hs - 1, ra[(hs + DEPTH) & 3]);
      transpose(ra[(hs + 1) & 3], p[(hs + 1) & 1]);
      const f64x2(&a)[4] = ra[hs];
      const f64x2(&pc)[2][2] = p[hs & 1];
      // direct: D[row 2 (g + 4 reg) + par, block column c] += sum_k A[row, tile column 4 u + k] X_J[tile column, c]
      // transposed: Z[tile column g + 4 reg, block column c] += sum_k P[row 16 ib + 4 k + 2 j + xy, tile column] X_I[row, c]
      // alternating; every accumulator chain is touched once per 4 NB MFMAs
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int u = 2 * ib + j;
#pragma unroll
          for (int bcb = 0; bcb < NB; ++bcb) {
            if (hs < ACC_A) mfma_acc_vv(acc[hs][0][bcb], a[u].x, b[u][bcb]);
            else mfma_v_vv(acc[hs][0][bcb], a[u].x, b[u][bcb]);
            if (hs == 0 && u == 0) mfma_v_va_first(zc[bcb][0], pc[ib][j].x, xI[hs][ib][j][0][bcb]);
            else mfma_v_va(zc[bcb][0], pc[ib][j].x, xI[hs][ib][j][0][bcb]);
            if (hs < ACC_A) mfma_acc_vv(acc[hs][1][bcb], a[u].y, b[u][bcb]);
            else mfma_v_vv(acc[hs][1][bcb], a[u].y, b[u][bcb]);
            if (hs == 0 && u == 0) mfma_v_va_first(zc[bcb][1], pc[ib][j].y, xI[hs][ib][j][1][bcb]);
            else mfma_v_va(zc[bcb][1], pc[ib][j].y, xI[hs][ib][j][1][bcb]);
          }
        }
    }
    // z[reg]: tile column col + g + 4 reg, block column c of group bcb, summed over this wave's 128 rows
    {
      double* zr = &zred[q & 1][wave][0];
      if constexpr (NB == 2) asm volatile(MFMA_DRAIN : "+v"(zc[0][0]), "+v"(zc[0][1]), "+v"(zc[1][0]), "+v"(zc[1][1]));
      else asm volatile(MFMA_DRAIN : "+v"(zc[0][0]), "+v"(zc[0][1]));
#pragma unroll
      for (int bcb = 0; bcb < NB; ++bcb) {
        const f64x4 z = (zc[bcb][0] + zc[bcb][1]) * zm;
        *reinterpret_cast<f64x2*>(zr + 256 * bcb + 2 * lane) = f64x2{z[0], z[1]};
        *reinterpret_cast<f64x2*>(zr + 256 * bcb + 128 + 2 * lane) = f64x2{z[2], z[3]};
      }
    }
    __syncthreads();
    // this barrier also publishes the stage writes of unit q - 1: the previous strip is complete
    if (q % UPS == 0 && q > 0) flush_strip(q / UPS - 1);
    // the four waves sum disjoint parts of the 16 x 16 NB partial, slices in fixed order
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const unsigned e = wave * (NB * 64) + 64 * t + lane;
      double s = zred[q & 1][0][e];
#pragma unroll
      for (int sl = 1; sl < NRS; ++sl) s += zred[q & 1][sl][e];
      const unsigned bcb = e >> 8, half = (e >> 7) & 1, ln = (e & 127) >> 1, jj = e & 1;
      const unsigned gg = ln >> 4, reg = 2 * half + jj;
      const unsigned bc = 16 * bcb + (ln & 15);
      const unsigned tcol = (q % UPJ) * 16 + gg + 4 * reg;
      zst[(q / UPS) & 1][bc * ZS + (tcol & (ZW - 1))] = s;
    }
    ur = urn;
  };
  // nunits is a multiple of 16: two units per trip, the B operand sets alternate (no register copies)
  for (unsigned q = 0; q < nunits; q += 2) {
    unit(q, b0, b1);
    unit(q + 1, b1, b0);
  }
  __syncthreads();
  flush_strip(nunits / UPS - 1);

  // end of the run: the direct partials of every row slice (complete: one column group per workgroup), one 32-row
  // half-step and one group at a time through the transposition scratch so that they leave as 256-byte rows
  double* outD = slabD + (int64_t)items[4 * item + 3] * R * 16 * SYM_TB;
#pragma unroll
  for (int bcb = 0; bcb < NB; ++bcb) {
    const int kc = kcols - 16 * bcb < 16 ? kcols - 16 * bcb : 16;
#pragma unroll
    for (int hs = 0; hs < 4; ++hs) {
      if (hs < ACC_A) asm volatile(MFMA_DRAIN : "+a"(acc[hs][0][bcb]), "+a"(acc[hs][1][bcb]));
      else asm volatile(MFMA_DRAIN : "+v"(acc[hs][0][bcb]), "+v"(acc[hs][1][bcb]));
#pragma unroll
      for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) tw[c * RS + 2 * (g + 4 * reg) + par] = acc[hs][par][bcb][reg];
      // wave-private scratch, in-order DS: the reads below see the writes above
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int e = lane + 64 * t;                      // (block column, row of the half-step)
        const int bc = e >> 5, r = e & 31;
        const double v = tw[bc * RS + r];
        if (bc < kc && have_row)
          outD[bcb * slabD_gstride + ((int64_t)(wave >> 1) * 16 + bc) * SYM_TB + 128 * rhalf + 32 * hs + r] = v;
      }
    }
  }
  // block rows of the super row past the end of the matrix: their slab rows are read by nobody
}

void launch_matvec_symw(hipStream_t st, int nbw, const double* tiles, const int64_t* row_off, int nb, const int* items_dev, int nitems,
                        const int* zslot_begin_dev, const double* xt, int kcols, double* slabD, double* slabT, int nwg,
                        int64_t xt_gstride, int64_t slabD_gstride, int64_t slabT_gstride) {
  dim3 grid(nitems * nwg), block(256);
  if (nbw == 2)
    hipLaunchKernelGGL((matvec_symw_kernel<2>), grid, block, 0, st, tiles, row_off, items_dev, zslot_begin_dev, xt, slabD, slabT, kcols, nwg,
                       xt_gstride, slabD_gstride, slabT_gstride, nb);
  else
    hipLaunchKernelGGL((matvec_symw_kernel<1>), grid, block, 0, st, tiles, row_off, items_dev, zslot_begin_dev, xt, slabD, slabT, kcols, nwg,
                       xt_gstride, slabD_gstride, slabT_gstride, nb);
}
