// K1s, wide blocks - the symmetric-tiled sweep with ONE wave per SIMD and 16 NB block columns per workgroup (NB = 1, 2).
//
// Why.  v_mfma_f64_16x16x4_f64 runs on the SIMD's fp64 vector units for 64 cycles (profiles/ubench/r01_mfma_f64_overlap.log: a
// second wave on the SIMD gets one VALU instruction per ~21 cycles), so in the MFMA-bound launches of a solve (32 / 64
// columns) the instructions that are not MFMAs - how many, and where they stand - decide how busy the matrix pipe is.
// matvec_sym9_kernel<2> (16 columns per workgroup, two waves per SIMD, 216 VGPRs) issues 2.6 vector instructions besides
// each MFMA - tile loads, the LDS transposition of the tile, the X_I operand reads, 64-bit address arithmetic - and
// repeats the transposition of every tile in each of the 2 / 4 column groups of a launch: pipe 73 % busy (round 2).
// Here a workgroup is 4 waves x 512 registers.  Wave v keeps, for the whole work item,
//   - its 128-row slice of the super row (R = 2 block rows x 2 halves) x 16 NB block columns of direct partials,
//   - the X_I operand of the transposed product for those rows (it never changes within an item: no LDS, no re-reads),
// so one tile load and one LDS transposition feed 16 NB MFMAs per 32 x 16 sub-block instead of 16, and addresses are a scalar
// descriptor plus constant per-lane offsets.
// Where the other instructions stand matters as much as how many there are (profiles/ubench/loadcost.hip): a memory
// instruction that follows an MFMA issues in the shadow of that MFMA's 16 passes (~1-4 cycles), a burst of them queues in
// front of the next MFMA (8 cycles per buffer load, 29 per DS operation).  So a half-step is 16 NB asm statements of two
// MFMAs with ONE memory operation behind each: the transposition of the NEXT half-step (its LDS reads are 16 NB MFMAs old when
// their registers are used), the tile loads three half-steps ahead, the next unit's X_J, and - as second operations - the
// pieces of the exchange of the transposed partials.  Matrix pipe 93-94 % busy (PMC), 0.78 / 0.87 of the fp64 peak at 32 / 64
// columns at the clock the chip holds under this load.
// Same work items, slab layout, masking rules and fixed-order sums as matvec_sym9_kernel<2> (k_matvec_sym9.hip), so the
// same reduction kernel follows; results are bitwise reproducible run to run.
//
// Registers.  A wave's 512 registers are 256 VGPRs + 256 accumulation registers.  Measured (profiles/ubench/mfma_regfile.hip,
// r03_mfma_regfile.log): v_mfma_f64_16x16x4_f64 issues every 64.0 cycles with its accumulator (C / D) in VGPRs - for any
// distance between two MFMAs of one chain, back to back included - but only every 83.1 cycles with C / D in
// accumulation registers; A and B operands cost nothing in either half.  So: accumulators (direct partials, transposed
// partial) in VGPRs; what an MFMA only READS in accumulation registers - X_I (128), the load ring of tile entries (64),
// the X_J operand (16 NB), the Gram-layout operands (32).  The compiler's allocator does not keep such values there reliably (it prefers VGPRs for a
// value that may live in either half, runs out, and copies - a VALU write right in front of an MFMA it does not know to
// be one), so the ring, X_J and the Gram-layout operands live in FIXED accumulation registers a[128:255] that only inline
// assembly names: buffer loads and DS reads straight into them, DS writes and MFMAs straight out of them, vmcnt counted by hand (every vector-memory load of
// the loop is one of these; the compiler's own operations - prologue, strip stores - only make the counts conservative).
// What the compiler does not know about an asm MFMA is handled structurally: every MFMA operand is produced by a
// load, never by a VALU instruction (the B operand of a tile that is not there is loaded from a page of zeros instead
// of multiplied by zero; a chain starts with the literal-zero form; no branches around MFMAs), and MFMA results are read
// by other instructions only behind the 18 wait states a 16-pass MFMA needs (MFMA_DRAIN) or many MFMAs later.
// tests/test_isa_lint.py checks the emitted code: no compiler instruction touches a[128:255], no VALU write within two instructions of an MFMA that reads it.
#include "kernels.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

#define MFMA_DRAIN "s_nop 15\n\ts_nop 3"

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <class F, int... Is>
__device__ __forceinline__ void symw_static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void symw_static_for(F&& f) { symw_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// fixed accumulation registers: ring slot k (step mod NSLOT), load u: a[RING(k,u) .. +3] = rows 2c, 2c+1 of tile column 4u + g
// (x = +0..1, y = +2..3), the ring ending at a223; X_J operand set s (unit parity), column quad u, group bcb: a[XJ(s,u,bcb) .. +1]
// Gram-layout (transposed) operands of half-step parity par, 16-row block ib, row pair j: a[P(par,ib,j) .. +3] (x = +0..1, y = +2..3)
#define SYMW_RING(k, u) (224 - 16 * NSLOT + 16 * (k) + 4 * (u))
#define SYMW_XJ(s, u, bcb) (224 + 16 * (s) + 4 * (u) + 2 * (bcb))
#define SYMW_P(par, ib, j) (SYMW_PBASE + 16 * (par) + 8 * (ib) + 4 * (j))       // SYMW_PBASE: constexpr of the kernel (128; the harness variants: 168)
// fp32 tiles: raw ring slot k, load u in FIXED VGPRs v[224 + 8 k + 2 u .. + 1] (tests/test_isa_lint.py: the compiler stays below v224)
#define SYMW_RAW(k, u) (224 + 8 * (k) + 2 * (u))
// harness operator (GEN >= 2; the ring registers are free there): 2 log e of rows 2c, 2c + 1 of half-step hs: a[LROW(hs) .. + 3]; of tile
// column 4u + g of the unit: a[LCOL(u) .. + 1]
#ifndef SYMW_HARN_GROUP
#define SYMW_HARN_GROUP 1          // tile columns (= 2 Horner chains each) evaluated together in the burst
#endif
#define SYMW_LROW(hs) (200 + 4 * (hs))
#define SYMW_LCOL(u) (216 + 2 * (u))
constexpr int symw_fixed_lo(bool harness) { return harness ? 168 : 128; }   // first fixed register (tests/test_isa_lint.py)

__device__ double symw_zero_page[16 * 16];     // B operand of a tile that is not there

#ifdef DAV_SYMW_STAMPS
// diagnostic build only (scratch/): shader cycles per wave summed over units - [0] unit start to the barrier (half-step 0),
// [1] barrier wait, [2] LDS reads of the sum + half-step 1 + sum, [3] units, [4] / [5] shader clock / 100 MHz clock
__device__ unsigned long long symw_stamp[12];
#define STAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int dav_symw_stamps(unsigned long long* out) {
  unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(symw_stamp), sizeof(z)) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(symw_stamp), z, sizeof(z)) != hipSuccess;
}
#else
#define STAMP(t) do { } while (0)
#endif

// a wave-uniform value as the compiler must see it to keep it in scalar registers (asm "s" operands): through v_readfirstlane
__device__ __forceinline__ unsigned symw_uniform(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ int64_t symw_uniform(int64_t x) {
  const uint64_t u = (uint64_t)x;
  return (int64_t)(((uint64_t)symw_uniform((unsigned)(u >> 32)) << 32) | symw_uniform((unsigned)u));
}
__device__ __forceinline__ const char* symw_uniform(const char* p) { return reinterpret_cast<const char*>(symw_uniform((int64_t)reinterpret_cast<uintptr_t>(p))); }

// descriptor of a raw buffer (stride 0) at p: the hardware adds a per-lane and a scalar 32-bit offset
__device__ __forceinline__ i32x4 symw_desc(const void* p, int bytes) {
  // the two address words through v_readfirstlane: the compiler otherwise builds some descriptors in VGPRs (uniform values it
  // chose to compute on the VALU), which an asm "s" operand cannot take.  A buffer instruction that reads a scalar register
  // written by a VALU instruction needs 5 wait states in between: the first load of every descriptor opens with s_nop 4.
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  return i32x4{(int)symw_uniform((unsigned)a), (int)symw_uniform((unsigned)(a >> 32) & 0xffffu), bytes, 0x00020000};
}

// TALL (NB = 1 only; what 9-16 columns run): FOUR block rows per workgroup - every wave works through two 128-row slices of a unit
// one after the other (slice s: block row I0 + 2 s + wave / 2), so that the cross-wave sum of a unit's transposed partials covers
// four block rows and half as many of them are written (at 16 columns the sweep is HBM-bound and those writes are what
// separates it from the 8-column path).  The sequence the loop walks is (unit 0, slice 0), (unit 0, slice 1), (unit 1, slice 0)
// ...: a "pair" is then the two slices of ONE unit, everything else - ring, Gram operands, X_J sets (one per slice: a slice
// without a tile reads zeros), slots, exchange - is the two-block-row kernel's.
// F32 (NB = 1, two block rows; the opt-in mixed-precision inner sweeps of the GJD correction): the tiles are an fp32 copy of the
// stored operator - half the bytes of a sweep that is byte-bound at 16 columns.  The loads land in FIXED VGPRs v[224:255] (a
// 4-slot ring of raw fp32 pairs), a burst of 8 v_cvt_f64_f32 in front of a half-step's MFMAs widens the NEXT half-step's
// sub-block into ordinary registers, and from there on everything is fp64: the direct product's A operand and the LDS
// transposition take those registers instead of the ring, products and sums are the fp64 kernel's.
// GEN (NB = 2, two block rows; the hashed matrix-free operator at more than 16 columns): no tiles at all - every half-step the wave
// GENERATES the next half-step's 32 x 16 sub-block (8 entries per lane: one splitmix64 each, the layout of the tile loads) into
// ordinary registers, as one burst of VALU work in front of the half-step's MFMAs, and from there on it is the fp32-tile
// variant's data path: the direct product's A operand and the LDS transposition take those registers.  What it buys: a generated
// entry feeds 2 NB = 4 MFMAs instead of the 2 of the 16-column kernels (matvec_sym9_kernel<2, GEN>), whose 32- and 64-column
// blocks regenerate the whole operator per 16 columns - the generator (26 VALU instructions per 64 entries, on the issue port the
// MFMAs use) is paid once per 32 columns.
// GEN = 2 / 3 (round 6): the reference's matrix-free test operator (src/tests/test_utils.f90:72-116; cos / sin) in its one-variable form
// (common.h: dav_harness_poly - x = 1 - |l_i - l_j| from a table of 2 log e, a degree-17 / 19 polynomial: 19 / 21 fp64 VALU instructions
// per entry) on the same data path.  The table values a wave needs - its 128 rows (8 per lane, fixed for the work item) and a unit's 16
// tile columns (4 per lane) - wait in accumulation registers of the range the stored variants use for their load ring (a200.. : SYMW_LROW /
// SYMW_LCOL): the rows are put there once, the columns of unit q + 1 are loaded straight into them in half-step 2 of unit q (their
// last reader was that half-step's burst), and each burst fetches its 12 words with v_accvgpr_read - 12 instructions next to 160.
template <int NB, bool TALL, bool F32, int GEN>
__global__ __launch_bounds__(256, 1) void matvec_symw_kernel(const void* __restrict__ tiles_any, const int64_t* __restrict__ row_off,
                                                             const int* __restrict__ items, const int* __restrict__ zslot_begin,
                                                             const double* __restrict__ xt, double* __restrict__ slabD,
                                                             double* __restrict__ slabT, int kcols, int nwg, int64_t xt_gstride,
                                                             int64_t slabD_gstride, int64_t slabT_gstride, int nb, OpParams op, int64_t n) {
  static_assert(!TALL || NB == 1, "two slices per wave: 16 columns per workgroup");
  static_assert(GEN == 0 || (NB == 2 && !TALL && !F32), "generated entries: 32 columns, two block rows per workgroup");
  constexpr bool HARN = GEN >= 2;       // the reference's matrix-free test operator (GEN = 2: cos, 3: sin), polynomial form
  // The harness variants run at the limit of the 256 ordinary registers (128 of direct partials, 64 of transposed ones, 32 of generated
  // sub-blocks, and the burst's Horner chains): the allocator parks a few dozen lane constants in accumulation registers - the lowest
  // free ones, a128 up.  So these variants leave a128-a167 to it and keep their own fixed registers above (the ring's range is free
  // where nothing is loaded): Gram-layout operands a168-a199, table values a200-a223, X_J a224-a255.  tests/test_isa_lint.py checks that
  // compiler-generated code stays below symw_fixed_lo(harness).
  constexpr int SYMW_PBASE = symw_fixed_lo(HARN);
  constexpr bool REGT = F32 || GEN != 0;     // the sub-blocks reach the MFMAs and the transposition through ordinary registers
  (void)op; (void)n;
  static_assert(!F32 || (NB == 1 && !TALL), "fp32 tiles: 16 columns, two block rows per workgroup");
  using TileT = std::conditional_t<F32, float, double>;
  const TileT* __restrict__ tiles = static_cast<const TileT*>(tiles_any);
  constexpr int R = TALL ? 4 : 2;       // block rows per workgroup
  constexpr int NSL = TALL ? 2 : 1;     // 128-row slices per wave
  constexpr int NRS = 4;                // 128-row slices = waves
  constexpr unsigned UPJ = SYM_TB / 16; // units (16 tile columns) per tile column
  constexpr unsigned UPS = 4;           // units per 64-column strip of the stage
  // ring of tile loads: 4 slots, 3 half-steps of lookahead (an 8-slot ring at 16 columns, where the sweep is HBM-bound and
  // the registers are there, measured no gain: the sweep runs at what the partial-sum writes leave of the HBM rate)
  constexpr int NSLOT = 4;
  constexpr int DEPTH = NSLOT - 1;
  constexpr int TRS = 34, TRW = 16 * TRS, RS = 33;
  constexpr int ZW = 64, ZS = ZW + 2;   // stage strip: 64 tile columns, padded
  constexpr int NBL = 4 * NB;           // X_J loads per unit
  __shared__ __attribute__((aligned(16))) double tr[NRS * TRW];
  __shared__ __attribute__((aligned(16))) double zred[2][NRS][2][NB * 256];   // [pair parity][wave][unit of the pair][group][f64x4 per lane]
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
  const int rhalf = wave & 1;
  int Is[NSL];                                                    // this wave's block rows (may lie past the end)
  bool have_row[NSL];
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) { Is[sl] = I0 + 2 * sl + (wave >> 1); have_row[sl] = Is[sl] <= Imax; }

  // X_I of this wave's 128 rows, in the lane layout of the transposed product's B operand: for half-step hs, 16-row block
  // ib, row pair j and parity xy the lane (k = g, column c) holds X[row 32 hs + 16 ib + 4 g + 2 j + xy, block column c].
  // A wave without a block row reads the rows of block row Imax; its partial is multiplied by zero.
  double xI[NSL][4][2][2][2][NB];
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) {
    const double* xr = xt + ((int64_t)(have_row[sl] ? Is[sl] : Imax) * SYM_TB + 128 * rhalf) * 16;
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
              xI[sl][hs][ib][j][xy][bcb] = xr[bcb * xt_gstride + xo + (32 * hs + 16 * ib + 2 * j + xy) * 16];
  }

  f64x4 acc[NSL][4][2][NB];         // direct partials: [slice][half-step][row parity][group] x (4 row groups in the f64x4)
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
    for (int hs = 0; hs < 4; ++hs)
#pragma unroll
      for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int bcb = 0; bcb < NB; ++bcb) acc[sl][hs][par][bcb] = f64x4{0.0, 0.0, 0.0, 0.0};

  const unsigned nunits = symw_uniform((unsigned)(J1 - J0) * UPJ);
  const unsigned nseq = NSL * nunits;                             // length of the sequence the loop walks: (unit, slice) pairs
  double* tw = tr + wave * TRW;
  // LDS byte addresses (the low 32 bits of a generic pointer into LDS are the LDS offset)
  const unsigned tw_wr = (unsigned)reinterpret_cast<uintptr_t>(tw + g * TRS + 2 * c);    // + 4 u rows of TRS: the direct layout

  // tile loads: a scalar descriptor on the unit's 128 x 16 sub-block, constant per-lane offsets (rows 2c, 2c+1 of column
  // 4u + g), the half-step as a scalar offset: no vector address arithmetic in the loop
  unsigned voff[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) voff[u] = ((4 * u + g) * SYM_TB + 2 * c) * (unsigned)sizeof(TileT);
  // Tile this wave works on in tile column J: its own if it is stored, else the (stored) tile of block row Imax (masked).
  // The tiles of a block row are contiguous in J and a unit is 16 tile columns = 32 KiB of a tile: the sub-block of unit q
  // starts (J0 16 + q) 32 KiB behind the block row's first tile - linear in q, no division, no per-unit table look-up
  // (a scalar instruction costs ~5 cycles of matrix-pipe time here, profiles/ubench/r03_fatwave_vgpr_acc.log).
  constexpr int64_t UNIT_BYTES = 16 * SYM_TB * (int64_t)sizeof(TileT);
  const int64_t unit0 = (int64_t)J0 * UPJ * UNIT_BYTES;
  const char* const trow_max = GEN != 0 ? nullptr : symw_uniform(reinterpret_cast<const char*>(tiles + row_off[Imax] * (int64_t)(SYM_TB * SYM_TB) + 128 * rhalf) + unit0);
  const char* trow_own[NSL];
  unsigned qlim_d[NSL];              // units whose tile is stored for this wave's slice (J <= I): q < qlim_d
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) {
    trow_own[sl] = GEN != 0 ? nullptr : symw_uniform(reinterpret_cast<const char*>(tiles + row_off[have_row[sl] ? Is[sl] : Imax] * (int64_t)(SYM_TB * SYM_TB) + 128 * rhalf) + unit0);
    qlim_d[sl] = symw_uniform(have_row[sl] ? (Is[sl] >= J0 ? (unsigned)(Is[sl] - J0 + 1) * UPJ : 0u) : 0u);
  }
  // position i of the sequence: unit i / NSL, slice i % NSL
  auto unit_desc = [&](unsigned i) {
    i = i < nseq ? i : nseq - 1;
    const unsigned q = i / NSL, sl = i % NSL;
    return symw_desc((q < qlim_d[sl] ? trow_own[sl] : trow_max) + (int64_t)q * UNIT_BYTES, (int)UNIT_BYTES);
  };
  // One vector-memory / LDS operation per call, each its own asm statement.  In the loop they are placed BETWEEN the MFMAs, one
  // behind every second MFMA, never in bursts: next to 64-cycle MFMAs a lone wave pays 1-2 cycles for a memory instruction
  // that follows an MFMA, but 8 (buffer load) to 29 (DS) cycles for each instruction of a burst - 263 against 12 cycles per
  // half-step for its 8 loads + 8 DS operations (profiles/ubench/loadcost.hip, r03_loadcost.log).
  // tile load U of half-step hs of the unit behind `d` into ring slot SLOT
  auto t_load = [&](auto slot, auto uc, const i32x4& d, int hs, auto fresh) {
    constexpr int SLOT = decltype(slot)::value, U = decltype(uc)::value;
    const unsigned vo = voff[U];           // (operands of an asm statement inside a generic lambda are not captured by themselves)
    const i32x4 dd = d;
    const int soo = hs * 32 * (int)sizeof(TileT);
    if constexpr (F32) {
      // raw fp32 pair (rows 2c, 2c + 1 of tile column 4u + g) -> v[SYMW_RAW(SLOT, U) .. + 1]
      if constexpr (U == 0 && decltype(fresh)::value)
        asm volatile("s_nop 4\n\tbuffer_load_dwordx2 v[%c0:%c1], %2, %3, %4 offen"
                     :: "i"(SYMW_RAW(SLOT, U)), "i"(SYMW_RAW(SLOT, U) + 1), "v"(vo), "s"(dd), "s"(soo) : "memory");
      else
        asm volatile("buffer_load_dwordx2 v[%c0:%c1], %2, %3, %4 offen"
                     :: "i"(SYMW_RAW(SLOT, U)), "i"(SYMW_RAW(SLOT, U) + 1), "v"(vo), "s"(dd), "s"(soo) : "memory");
    } else if constexpr (U == 0 && decltype(fresh)::value)
      asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[%c0:%c1], %2, %3, %4 offen"
                   :: "i"(SYMW_RING(SLOT, U)), "i"(SYMW_RING(SLOT, U) + 3), "v"(vo), "s"(dd), "s"(soo) : "memory");
    else
      asm volatile("buffer_load_dwordx4 a[%c0:%c1], %2, %3, %4 offen"
                   :: "i"(SYMW_RING(SLOT, U)), "i"(SYMW_RING(SLOT, U) + 3), "v"(vo), "s"(dd), "s"(soo) : "memory");
  };
  // B operand of the direct product for unit q: X_J rows of the unit's 16 tile columns (16 rows of Xt = 2 KiB per unit,
  // contiguous over the whole run) - or, where this wave has no stored tile (above the diagonal inside the diagonal super
  // block, block row past the end), a page of zeros.  One descriptor per 16-column group, the column quad as one of four
  // constant scalar offsets.
  const unsigned boff = (g * 16 + c) * (unsigned)sizeof(double);
  const char* const xj0 = symw_uniform(reinterpret_cast<const char*>(xt + (int64_t)J0 * SYM_TB * 16));
  const char* xjg[NB];               // first X_J row of the run, per 16-column group
#pragma unroll
  for (int bcb = 0; bcb < NB; ++bcb) xjg[bcb] = symw_uniform(xj0 + bcb * xt_gstride * (int64_t)sizeof(double));
  const char* const zpage = symw_uniform(reinterpret_cast<const char*>(symw_zero_page));
  int xso[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) xso[u] = (4 * u) * 16 * (int)sizeof(double);
  auto x_desc = [&](unsigned i, int bcb) {
    i = i < nseq ? i : nseq - 1;
    const unsigned q = i / NSL, sl = i % NSL;
    return symw_desc(q < qlim_d[sl] ? xjg[bcb] + (int64_t)q * (16 * 16 * (int64_t)sizeof(double)) : zpage, 16 * 16 * (int)sizeof(double));
  };
  // column quad U of group B of the unit behind `d` -> X_J operand set SET
  auto x_load = [&](auto set, auto uc, auto bc, const i32x4& d, auto fresh) {
    constexpr int SET = decltype(set)::value, U = decltype(uc)::value, B = decltype(bc)::value;
    const unsigned bo = boff;
    const i32x4 dd = d;
    const int so = xso[U];
    if constexpr (U == 0 && decltype(fresh)::value)
      asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[%c0:%c1], %2, %3, %4 offen"
                   :: "i"(SYMW_XJ(SET, U, B)), "i"(SYMW_XJ(SET, U, B) + 1), "v"(bo), "s"(dd), "s"(so) : "memory");
    else
      asm volatile("buffer_load_dwordx2 a[%c0:%c1], %2, %3, %4 offen"
                   :: "i"(SYMW_XJ(SET, U, B)), "i"(SYMW_XJ(SET, U, B) + 1), "v"(bo), "s"(dd), "s"(so) : "memory");
  };
  // LDS transposition of the 32 x 16 sub-block in a ring slot: direct layout (rows 2c, 2c+1 of column 4u + g) -> Gram layout
  // (P(par, ib, j) = rows 16 ib + 4 g + 2 j, + 1 of column c); wave-private, DS operations of a wave execute in order: no
  // barrier, and the four reads may follow the four writes without a wait.
  const unsigned tw_rd = (unsigned)reinterpret_cast<uintptr_t>(tw + c * TRS + 4 * g);
  auto ds_w = [&](auto slot, auto uc) {
    constexpr int SLOT = decltype(slot)::value, U = decltype(uc)::value;
    const unsigned wa = tw_wr;
    asm volatile("ds_write_b128 %0, a[%c1:%c2] offset:%c3"
                 :: "v"(wa), "i"(SYMW_RING(SLOT, U)), "i"(SYMW_RING(SLOT, U) + 3), "i"(4 * U * TRS * (int)sizeof(double)) : "memory");
  };
  // F32: the widened sub-blocks, [half-step parity][column quad] = rows 2c, 2c + 1 of tile column 4u + g
  f64x2 wide[REGT ? 2 : 1][4];
  // GEN: step st = 4 q + hs of the run -> rows 2c, 2c + 1 of tile columns 4u + g of its sub-block, generated (the entries
  // matvec_sym9_kernel<2, GEN> generates: strictly below the diagonal and inside the matrix one splitmix64 per entry on a
  // key that is linear in row and column; the diagonal tile, the last ragged block row and tiles that are "not there" for this
  // wave - masked like stored ones - through dav_hashed_entry with its bounds).  All conditions are wave-uniform.
  const uint64_t seedmix = op.seed * 0x9E3779B97F4A7C15ull;
  const double gscale = op.sparsity * (1.0 / 9007199254740992.0);
  // harness operator: the table values of this wave's rows into their accumulation registers (once per work item); a wave without a
  // block row generates - and masks - the tile of block row Imax, as the stored kernels read it
  const int Irow = have_row[0] ? Is[0] : Imax;
  auto acc_read = [&](auto regc) {
    constexpr int RG = decltype(regc)::value;
    unsigned lo, hi;
    asm volatile("v_accvgpr_read_b32 %0, a%c2\n\tv_accvgpr_read_b32 %1, a%c3" : "=v"(lo), "=v"(hi) : "i"(RG), "i"(RG + 1));
    return __hiloint2double((int)hi, (int)lo);
  };
  // the 4 table values of unit q's tile columns 4u + g, straight into a[SYMW_LCOL]: one 16-value window per unit
  const unsigned lcoff = g * (unsigned)sizeof(double);
  auto lcol_desc = [&](unsigned q) {
    q = q < nunits ? q : nunits - 1;
    return symw_desc(reinterpret_cast<const char*>(op.l2_table + (int64_t)J0 * SYM_TB) + (int64_t)q * (16 * (int64_t)sizeof(double)), 16 * (int)sizeof(double));
  };
  auto lcol_load = [&](auto uc, const i32x4& d, auto fresh) {
    constexpr int U = decltype(uc)::value;
    const unsigned vo = lcoff;
    const i32x4 dd = d;
    if constexpr (U == 0 && decltype(fresh)::value)
      asm volatile("s_nop 4\n\tbuffer_load_dwordx2 a[%c0:%c1], %2, %3, 0 offen offset:%c4"
                   :: "i"(SYMW_LCOL(U)), "i"(SYMW_LCOL(U) + 1), "v"(vo), "s"(dd), "i"(4 * U * (int)sizeof(double)) : "memory");
    else
      asm volatile("buffer_load_dwordx2 a[%c0:%c1], %2, %3, 0 offen offset:%c4"
                   :: "i"(SYMW_LCOL(U)), "i"(SYMW_LCOL(U) + 1), "v"(vo), "s"(dd), "i"(4 * U * (int)sizeof(double)) : "memory");
  };
  if constexpr (HARN) {
    symw_static_for<4>([&](auto hsc) {
      constexpr int hs = decltype(hsc)::value;
      const f64x2 lr = *reinterpret_cast<const f64x2*>(op.l2_table + (int64_t)Irow * SYM_TB + 128 * rhalf + 32 * hs + 2 * c);
      const unsigned x0 = (unsigned)__double2loint(lr.x), x1 = (unsigned)__double2hiint(lr.x), y0 = (unsigned)__double2loint(lr.y), y1 = (unsigned)__double2hiint(lr.y);
      asm volatile("v_accvgpr_write_b32 a%c0, %4\n\tv_accvgpr_write_b32 a%c1, %5\n\tv_accvgpr_write_b32 a%c2, %6\n\tv_accvgpr_write_b32 a%c3, %7"
                   :: "i"(SYMW_LROW(hs)), "i"(SYMW_LROW(hs) + 1), "i"(SYMW_LROW(hs) + 2), "i"(SYMW_LROW(hs) + 3), "v"(x0), "v"(x1), "v"(y0), "v"(y1));
    });
  }
  auto generate = [&](unsigned st, f64x2 (&wv)[4], auto hsc) {
    constexpr int HSG = decltype(hsc)::value;           // half-step of step st (compile-time: it selects registers)
    (void)HSG;
    if constexpr (HARN) {
      st = st < 4 * nunits ? st : 4 * nunits - 1;
      const unsigned q = st >> 2;
      const int J = J0 + (int)(q / UPJ);
      if (J < Irow && ((int64_t)Irow + 1) * SYM_TB <= n) {
        // strictly below the diagonal, inside the matrix: x = 1 - |l_row - l_col|, one polynomial per entry
        // (2 SYMW_HARN_GROUP Horner chains at a time; all eight at once would need 32 more live registers than the wave has, and the
        // allocator would take them from the accumulation registers this kernel names by hand - tests/test_isa_lint.py checks that it does not)
        const double r0 = acc_read(std::integral_constant<int, SYMW_LROW(HSG)>{}), r1 = acc_read(std::integral_constant<int, SYMW_LROW(HSG) + 2>{});
        symw_static_for<4 / SYMW_HARN_GROUP>([&](auto hc) {
          constexpr int H = decltype(hc)::value;
          symw_static_for<SYMW_HARN_GROUP>([&](auto jc) {
            constexpr int U = SYMW_HARN_GROUP * H + decltype(jc)::value;
            const double lc = acc_read(std::integral_constant<int, SYMW_LCOL(U)>{});
            wv[U].x = dav_harness_poly<GEN == 3>(r0, lc);
            wv[U].y = dav_harness_poly<GEN == 3>(r1, lc);
          });
          __builtin_amdgcn_sched_barrier(0);
        });
      } else {
        // the diagonal tile, the ragged last block row, tiles that are "not there" for this wave (masked like stored ones)
        // (a few tiles per run, but the code sits in the loop sixteen times: 32-bit indices - the table is indexed by row / column,
        // n < 2^31 - and one entry at a time, so that this path asks for no more registers than the fast one)
        const int Ie = (have_row[0] && q < qlim_d[0]) ? Is[0] : Imax;
        int gi = Ie * SYM_TB + 128 * rhalf + 32 * HSG + 2 * (int)c;
        int gj = J * SYM_TB + 16 * (int)(q % UPJ) + (int)g;
        asm volatile("" : "+v"(gi), "+v"(gj));          // opaque: nothing of this path is computed ahead of the loop and kept in registers
        const int nn = (int)n;
        const double* __restrict__ l2 = op.l2_table;       // padded with zeros behind n: every index below is inside it
        // the diagonal entries of this lane's two rows, from a table (operator A: poly(1) + real(i), one rounding each as the other
        // kernels compute it; operator B: 1): at most one entry per row is diagonal, and the conversion in this place would cost the
        // wave a dozen registers it does not have
        const f64x2 dgv = GEN == 3 ? f64x2{1.0, 1.0} : *reinterpret_cast<const f64x2*>(op.dadd_table + gi);
        auto entry = [&](int i, int j, double dg) {
          const double v = dav_harness_poly<GEN == 3>(l2[i], l2[j]);
          const double e = i == j ? dg : v;
          return (i < nn && j < nn) ? e : 0.0;
        };
        symw_static_for<4>([&](auto uc) {
          constexpr int U = decltype(uc)::value;
          wv[U].x = entry(gi, gj + 4 * U, dgv.x);
          __builtin_amdgcn_sched_barrier(0);
          wv[U].y = entry(gi + 1, gj + 4 * U, dgv.y);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
    } else if constexpr (GEN != 0) {
      st = st < 4 * nunits ? st : 4 * nunits - 1;
      const unsigned q = st >> 2, hs = st & 3;
      const int J = J0 + (int)(q / UPJ);
      const int Ie = (have_row[0] && q < qlim_d[0]) ? Is[0] : Imax;
      const int64_t gi = (int64_t)Ie * SYM_TB + 128 * rhalf + 32 * hs + 2 * c;
      const int64_t gj = (int64_t)J * SYM_TB + 16 * (q % UPJ) + g;
      if (J < Ie && ((int64_t)Ie + 1) * SYM_TB <= n) {
        const uint64_t k0 = (uint64_t)gi + seedmix;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint64_t kc = ((uint64_t)(gj + 4 * u) << 32) + k0;
          const uint64_t m0 = dav_splitmix64(kc) >> 11, m1 = dav_splitmix64(kc + 1) >> 11;
          wv[u].x = __builtin_fma((double)(uint32_t)(m0 >> 32), 4294967296.0, (double)(uint32_t)m0) * gscale;
          wv[u].y = __builtin_fma((double)(uint32_t)(m1 >> 32), 4294967296.0, (double)(uint32_t)m1) * gscale;
        }
      } else {
        // branch-free (the code of this path is executed for a few tiles per run but sits in the loop sixteen times)
        auto entry = [&](int64_t i, int64_t j) {
          const uint64_t lo = (uint64_t)(i < j ? i : j), hi = (uint64_t)(i < j ? j : i);
          const uint64_t m = dav_splitmix64((lo << 32) + hi + seedmix) >> 11;
          const double off = __builtin_fma((double)(uint32_t)(m >> 32), 4294967296.0, (double)(uint32_t)m) * gscale;
          const double dg = op.use_diag ? op.diag_val : (double)(i + 1);
          const double v = i == j ? dg : off;
          return (i < n && j < n) ? v : 0.0;
        };
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          wv[u].x = entry(gi, gj + 4 * u);
          wv[u].y = entry(gi + 1, gj + 4 * u);
        }
      }
    }
  };
  auto ds_w_reg = [&](auto uc, const f64x2& v) {
    constexpr int U = decltype(uc)::value;
    const unsigned wa = tw_wr;
    const f64x2 vv = v;
    asm volatile("ds_write_b128 %0, %1 offset:%c2" :: "v"(wa), "v"(vv), "i"(4 * U * TRS * (int)sizeof(double)) : "memory");
  };
  auto widen = [&](auto slot, f64x2 (&w)[4]) {      // raw ring slot -> fp64
    constexpr int SLOT = decltype(slot)::value;
    symw_static_for<4>([&](auto uc) {
      constexpr int U = decltype(uc)::value;
      double x, y;
      asm volatile("v_cvt_f64_f32 %0, v%c2\n\tv_cvt_f64_f32 %1, v%c3" : "=&v"(x), "=&v"(y) : "i"(SYMW_RAW(SLOT, U)), "i"(SYMW_RAW(SLOT, U) + 1));
      w[U] = f64x2{x, y};
    });
  };
  auto ds_r = [&](auto parc, auto tc) {
    constexpr int PAR = decltype(parc)::value, IB = decltype(tc)::value >> 1, JJ = decltype(tc)::value & 1;
    const unsigned ra = tw_rd;
    asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3"
                 :: "v"(ra), "i"(SYMW_P(PAR, IB, JJ)), "i"(SYMW_P(PAR, IB, JJ) + 3), "i"((16 * IB + 2 * JJ) * (int)sizeof(double)) : "memory");
  };

  const int64_t zbase = zslot_begin[S];
  // The exchange of the transposed partials happens once per PAIR of units and is spread over the NEXT pair, every piece of it
  // in a slot between MFMAs: half-step 0 of the next pair's first unit: both partials of the pair -> LDS (they are thousands of
  // cycles old: no drain, no wait; the accumulators alternate with the pair parity); ONE workgroup barrier before its half-step
  // 1 (what it waits for is the skew between the four waves); then per unit of that pair: the LDS reads of one cross-wave sum
  // (half-step 1), the sum in fixed slice order, masked, -> stage (half-step 2).  zred alternates with the pair parity: a wave
  // that runs ahead cannot overwrite what a slower one still sums, because to get there it has to pass the next barrier.
  // Mask: the slices of block row r feed the transposed product of unit q only where its tile lies below the diagonal.
  unsigned qz[R];
#pragma unroll
  for (int r = 0; r < R; ++r) qz[r] = symw_uniform(I0 + r <= Imax && I0 + r > J0 ? (unsigned)(I0 + r - J0) * UPJ : 0u);
  unsigned zoff[NB];                 // stage offsets of the entries this lane sums (constant; + 16 (q & 3) + strip parity)
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    const unsigned e = wave * (NB * 64) + 64 * t + lane;
    const unsigned bcb = e >> 8, half = (e >> 7) & 1, ln = (e & 127) >> 1, jj = e & 1;
    zoff[t] = (16 * bcb + (ln & 15)) * ZS + (ln >> 4) + 4 * (2 * half + jj);
  }
  unsigned long long st_hs = 0, st_bar = 0, st_sum = 0, st_unit = 0, st_vm = 0, st_mf = 0;   // (diagnostic build; st_vm: cycles in the waits that open the half-steps, st_mf: in their MFMAs + slots)
  (void)st_hs; (void)st_bar; (void)st_sum; (void)st_unit; (void)st_vm; (void)st_mf;
#ifdef DAV_SYMW_STAMPS
  unsigned long long tk0, tr0;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk0), "=s"(tr0) :: "memory");
#endif
  double sreg[NRS][NB];              // the slices of the entries this lane sums, on their way from LDS
  auto sum_issue = [&](unsigned qp, int sl) {  // cross-wave sum of unit qp: LDS reads of slice sl
#pragma unroll
    for (int t = 0; t < NB; ++t) sreg[sl][t] = zred[(qp >> 1) & 1][sl][qp & 1][wave * (NB * 64) + 64 * t + lane];
  };
  double zkeep[NB];                  // (TALL) the first slice's sum, until the second slice's joins it
  auto sum_finish = [&](unsigned ip, int t) {  // position ip of the sequence: wave slices in fixed order, times 1 or 0 -> stage
    const unsigned qp = ip / NSL;               // the unit
    const int sl = (int)(ip % NSL);
    double* zs = zst[(qp / UPS) & 1] + (qp & 3) * 16;
    const double m0 = qp < qz[2 * sl] ? 1.0 : 0.0, m1 = qp < qz[2 * sl + 1] ? 1.0 : 0.0;
    const double v = __builtin_fma(m1, sreg[3][t], __builtin_fma(m1, sreg[2][t], __builtin_fma(m0, sreg[1][t], m0 * sreg[0][t])));
    if constexpr (!TALL) zs[zoff[t]] = v;
    else if (sl == 0) zkeep[t] = v;
    else zs[zoff[t]] = zkeep[t] + v;
  };
  // both partials of a pair (un = unit of the pair) -> zred[parity of the pair]: z[reg] = tile column col + g + 4 reg, block
  // column c of group bcb, summed over this wave's 128 rows
  auto z_write = [&](const f64x4 (&z)[2][NB], unsigned pair, int w) {
    const int un = w / (2 * NB), bcb = (w / 2) % NB, half = w & 1;
    double* zr = &zred[pair & 1][wave][un][0];
    *reinterpret_cast<f64x2*>(zr + 256 * bcb + 128 * half + 2 * lane) = f64x2{z[un][bcb][2 * half], z[un][bcb][2 * half + 1]};
  };
  // strip st of the run (64 tile columns x 16 NB block columns = 512 NB f64x2 = 2 NB per thread): stage -> slabT
  auto flush_strip = [&](unsigned st) {
    const int J = J0 + (int)(st >> 2);
    if (J >= Imax) return;                           // no block row of the super row lies below tile column J
    const double* zs = zst[st & 1];
#pragma unroll
    for (int bcb = 0; bcb < NB; ++bcb) {
      const int kc = kcols - 16 * bcb < 16 ? kcols - 16 * bcb : 16;
      double* outT = slabT + bcb * slabT_gstride + (zbase + J) * 16 * SYM_TB + (st & 3) * ZW;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = threadIdx.x + 256 * h, bc = e >> 5, pr = e & 31;
        if (bc < kc) *reinterpret_cast<f64x2*>(outT + bc * SYM_TB + 2 * pr) = *reinterpret_cast<const f64x2*>(zs + (16 * bcb + bc) * ZS + 2 * pr);
      }
    }
  };

  // Vector-memory operations in program order from here on: X_J(0), L(0) .. L(DEPTH - 1) (the prologue, in bursts); then per
  // half-step hs of unit q (step s = 4 q + hs), spread over its MFMAs: the 4 tile loads of step s + DEPTH and, in half-steps 0
  // and 1, one half (NBL / 2) of the X_J loads of unit q + 1.  The transposition in half-step hs reads step s + 1, whose
  // loads went out in step s - 2: issued after them are the X_J halves of step s - 2 (they follow the tile loads), the
  // tile loads of step s - 1 and its X_J half.  X_J(q) is older than the 8 tile loads of steps 4 q - 2, 4 q - 1.
  static_assert(DEPTH == 3 && NSLOT == 4, "the counts below hold for a 4-slot ring with 3 half-steps of lookahead");
  constexpr auto younger = [](int hs) { return 4 + (NBL / 2) * (hs == 0 ? 0 : hs == 2 ? 2 : 1); };
  i32x4 ud[3];                                                  // descriptors of units q, q + 1, q + 2
  if constexpr (GEN == 0) { ud[0] = unit_desc(0); ud[1] = unit_desc(1); ud[2] = unit_desc(2); }
  symw_static_for<NB>([&](auto bc) {
    const i32x4 d = x_desc(0, decltype(bc)::value);
    symw_static_for<4>([&](auto uc) { x_load(std::integral_constant<int, 0>{}, uc, bc, d, std::true_type{}); });
  });
  if constexpr (GEN == 0) {
    symw_static_for<DEPTH>([&](auto sc) {
      constexpr int st = decltype(sc)::value;
      symw_static_for<4>([&](auto uc) { t_load(std::integral_constant<int, st % NSLOT>{}, uc, ud[st / 4], st % 4, std::true_type{}); });
    });
    asm volatile("s_waitcnt vmcnt(%c0)" :: "i"(4 * (DEPTH - 1)) : "memory");
  }
  if constexpr (GEN != 0) {
    if constexpr (HARN) {
      const i32x4 ld0 = lcol_desc(0);
      symw_static_for<4>([&](auto uc) { lcol_load(uc, ld0, std::true_type{}); });
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    generate(0, wide[0], std::integral_constant<int, 0>{});
    symw_static_for<4>([&](auto uc) { ds_w_reg(uc, wide[0][decltype(uc)::value]); });
  } else if constexpr (F32) {
    widen(std::integral_constant<int, 0>{}, wide[0]);
    symw_static_for<4>([&](auto uc) { ds_w_reg(uc, wide[0][decltype(uc)::value]); });
  } else {
    symw_static_for<4>([&](auto uc) { ds_w(std::integral_constant<int, 0>{}, uc); });
  }
  symw_static_for<4>([&](auto tc) { ds_r(std::integral_constant<int, 0>{}, tc); });

  f64x4 zcs[2][2][NB];               // transposed partials: [pair parity][unit of the pair][group]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int un = 0; un < 2; ++un)
#pragma unroll
      for (int bcb = 0; bcb < NB; ++bcb) zcs[i][un][bcb] = f64x4{0.0, 0.0, 0.0, 0.0};
  auto unit = [&](unsigned q, auto set, auto ppc) {
    constexpr int SET = decltype(set)::value;          // unit of the pair = X_J operand set; the next unit's goes to the other
    constexpr int PP = decltype(ppc)::value;           // parity of the pair
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    (void)t0; (void)t1; (void)t2; (void)t3;
    STAMP(t0);
    i32x4 xd[NB];                                      // X_J of unit q + 1
#pragma unroll
    for (int bcb = 0; bcb < NB; ++bcb) xd[bcb] = x_desc(q + 1, bcb);
    i32x4 udn;
    if constexpr (GEN == 0) udn = unit_desc(q + 3);
    i32x4 lcd;                                         // harness variant: the table window of unit q + 1's tile columns
    if constexpr (HARN) lcd = lcol_desc(q + 1);
    (void)lcd;
    // the sums of strip st (units 4 st .. 4 st + 3) were staged by position NSL (4 st + 3) + NSL + 1 of the sequence; the barrier
    // of the next even position publishes them: 4 st + 6 (TALL: 8 st + 10)
    constexpr unsigned FP = 4 * NSL, F0 = TALL ? 10 : 6;
    const bool flush_due = SET == 0 && (q % FP) == 2 && q >= F0;
    f64x4(&zc)[NB] = zcs[PP][SET];             // transposed partials of the unit
    symw_static_for<4>([&](auto hsc) {
      constexpr int hs = decltype(hsc)::value;
      // (the first pair goes through the exchange of units "-2" and "-1" too: zeros into stage entries that units 6 and 7
      // overwrite before their strip leaves - cheaper than a branch per unit)
      if (hs == 1) {
        STAMP(t1);
        if constexpr (SET == 0) __syncthreads();
        STAMP(t2);
        if (flush_due) flush_strip((q - F0) / FP);     // once per four units: not worth registers across MFMAs
      }
      if (hs == 2) STAMP(t3);
      // this half-step's Gram operands (read from LDS during the previous half-step) and the tile entries its transposition
      // moves (step s + 1) have landed
#if DAV_SYMW_STAMPS > 1
      unsigned long long w0, w1;
      STAMP(w0);
#endif
      // GEN: the only vector-memory loads of the loop are X_J of unit q + 1, issued in half-steps 0 and 1 of unit q: all landed
      // long before the next unit's half-step 0 asks for them
      // (harness variant: the table values of unit q + 1's tile columns were loaded in half-step 2 and are first read by the burst of
      // half-step 3, which generates step 4 (q + 1))
      if constexpr (GEN == 0) asm volatile("s_waitcnt vmcnt(%c0) lgkmcnt(0)" :: "i"(younger(hs)) : "memory");
      else if constexpr (hs == 0 || (HARN && hs == 3)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if DAV_SYMW_STAMPS > 1
      STAMP(w1);
      st_vm += w1 - w0;
#endif
      if constexpr (F32) widen(std::integral_constant<int, (hs + 1) % NSLOT>{}, wide[(hs + 1) & 1]);   // ONE VALU burst per half-step
      if constexpr (GEN != 0) {
        // the generator of step s + 1, pinned as ONE burst in front of this half-step's MFMAs (VALU instructions between MFMAs
        // cost 12.5 + 4 n cycles per gap, profiles/ubench/r03_valucost.log; the scheduler would spread them)
        __builtin_amdgcn_sched_barrier(0);
        generate(4 * q + hs + 1, wide[(hs + 1) & 1], std::integral_constant<int, (hs + 1) & 3>{});
        __builtin_amdgcn_sched_barrier(0);
      }
      // The memory operations of the half-step, one slot behind every second MFMA: the transposition of step s + 1 (ring slot
      // -> LDS -> Gram operands of the other parity; its reads are 16 NB MFMAs old when the next half-step starts), the tile
      // loads of step s + DEPTH (their ring slot held step s - 1, whose MFMAs and DS reads have been issued), X_J of unit q + 1
      // (its set was last read by unit q - 1).
      auto mem_slot = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        using WS = std::integral_constant<int, (hs + 1) % NSLOT>;
        using LS = std::integral_constant<int, (hs + DEPTH) % NSLOT>;
        using NP = std::integral_constant<int, (hs + 1) & 1>;
        using XS = std::integral_constant<int, 1 - SET>;
        constexpr std::false_type old{};                // descriptors of the loop are scalar-ALU results: no wait states needed
        if constexpr (NB == 2) {
          if constexpr (k < 4) {
            if constexpr (REGT) ds_w_reg(std::integral_constant<int, k>{}, wide[(hs + 1) & 1][k]);
            else ds_w(WS{}, std::integral_constant<int, k>{});
          } else if constexpr (k < 8) ds_r(NP{}, std::integral_constant<int, k - 4>{});
          else if constexpr (k < 12) {
            if constexpr (GEN == 0) t_load(LS{}, std::integral_constant<int, k - 8>{}, ud[(hs + DEPTH) / 4], (hs + DEPTH) % 4, old);
            else if constexpr (HARN && hs == 2) lcol_load(std::integral_constant<int, k - 8>{}, lcd, old);
          } else if constexpr (hs < 2) x_load(XS{}, std::integral_constant<int, k - 12>{}, std::integral_constant<int, hs>{}, xd[hs < 2 ? hs : 0], old);
          // the exchange (compiler-visible LDS operations: the asm statements around them keep them in their slots)
          if constexpr (SET == 0 && hs == 0 && k < 8) z_write(zcs[1 - PP], (q >> 1) - 1, k);
          if constexpr (hs == 1 && k >= 8 && k < 12) sum_issue(q - 2, k - 8);
          if constexpr (hs == 2 && (k == 12 || k == 14)) sum_finish(q - 2, (k - 12) / 2);
        } else {
          if constexpr (k < 2) {
            if constexpr (F32) {
              ds_w_reg(std::integral_constant<int, 2 * k>{}, wide[(hs + 1) & 1][2 * k]);
              ds_w_reg(std::integral_constant<int, 2 * k + 1>{}, wide[(hs + 1) & 1][2 * k + 1]);
            } else {
              ds_w(WS{}, std::integral_constant<int, 2 * k>{});
              ds_w(WS{}, std::integral_constant<int, 2 * k + 1>{});
            }
          } else if constexpr (k < 4) {
            ds_r(NP{}, std::integral_constant<int, 2 * (k - 2)>{});
            ds_r(NP{}, std::integral_constant<int, 2 * (k - 2) + 1>{});
          } else if constexpr (k < 6) {
            t_load(LS{}, std::integral_constant<int, 2 * (k - 4)>{}, ud[(hs + DEPTH) / 4], (hs + DEPTH) % 4, old);
            t_load(LS{}, std::integral_constant<int, 2 * (k - 4) + 1>{}, ud[(hs + DEPTH) / 4], (hs + DEPTH) % 4, old);
          } else if constexpr (hs < 2) {
            x_load(XS{}, std::integral_constant<int, 2 * hs + k - 6>{}, std::integral_constant<int, 0>{}, xd[0], old);
          }
          if constexpr (SET == 0 && hs == 0 && k >= 4) z_write(zcs[1 - PP], (q >> 1) - 1, k - 4);
          if constexpr (hs == 1 && k >= 4) sum_issue(q - 2, k - 4);
          if constexpr (hs == 2 && k == 6) sum_finish(q - 2, 0);
        }
      };
#if DAV_SYMW_STAMPS > 1
      unsigned long long fa, fb;
      STAMP(fa);
#endif
      // direct: D[row 2 (g + 4 reg) + par, block column c] += sum_k A[row, tile column 4 u + k] X_J[tile column, c]
      // transposed: Z[tile column g + 4 reg, block column c] += sum_k P[row 16 ib + 4 k + 2 j + xy, tile column] X_I[row, c]
      symw_static_for<4>([&](auto uc) {
        constexpr int u = decltype(uc)::value, ib = u >> 1, j = u & 1;
        symw_static_for<NB>([&](auto bc) {
          constexpr int bcb = decltype(bc)::value, gi = u * NB + bcb;
          // (named here: operands of an asm statement inside a generic lambda are not captured by themselves)
          constexpr int SL = TALL ? SET : 0;     // the slice this position of the sequence works on
          f64x4 &d0 = acc[SL][hs][0][bcb], &d1 = acc[SL][hs][1][bcb], &zz = zc[bcb];
          const double x0 = xI[SL][hs][ib][j][0][bcb], x1 = xI[SL][hs][ib][j][1][bcb];
          // two MFMAs per statement: direct (rows of parity 0 / 1), transposed (the row pair's first / second row)
          if constexpr (REGT) {
            // the direct product's A operand is the widened / generated sub-block (ordinary registers) instead of the ring
            const double a0 = wide[hs & 1][u].x, a1 = wide[hs & 1][u].y;
            if constexpr (hs == 0 && u == 0)
              asm volatile("v_mfma_f64_16x16x4_f64 %0, %3, a[%c4:%c5], %0\n\t"
                           "v_mfma_f64_16x16x4_f64 %1, a[%c6:%c7], %2, 0"
                           : "+v"(d0), "=&v"(zz)
                           : "a"(x0), "v"(a0), "i"(SYMW_XJ(SET, u, bcb)), "i"(SYMW_XJ(SET, u, bcb) + 1), "i"(SYMW_P(hs & 1, ib, j)), "i"(SYMW_P(hs & 1, ib, j) + 1));
            else
              asm volatile("v_mfma_f64_16x16x4_f64 %0, %3, a[%c4:%c5], %0\n\t"
                           "v_mfma_f64_16x16x4_f64 %1, a[%c6:%c7], %2, %1"
                           : "+v"(d0), "+v"(zz)
                           : "a"(x0), "v"(a0), "i"(SYMW_XJ(SET, u, bcb)), "i"(SYMW_XJ(SET, u, bcb) + 1), "i"(SYMW_P(hs & 1, ib, j)), "i"(SYMW_P(hs & 1, ib, j) + 1));
            mem_slot(std::integral_constant<int, 2 * gi>{});
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %3, a[%c4:%c5], %0\n\t"
                         "v_mfma_f64_16x16x4_f64 %1, a[%c6:%c7], %2, %1"
                         : "+v"(d1), "+v"(zz)
                         : "a"(x1), "v"(a1), "i"(SYMW_XJ(SET, u, bcb)), "i"(SYMW_XJ(SET, u, bcb) + 1), "i"(SYMW_P(hs & 1, ib, j) + 2), "i"(SYMW_P(hs & 1, ib, j) + 3));
          } else {
          if constexpr (hs == 0 && u == 0)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, a[%c3:%c4], a[%c5:%c6], %0\n\t"
                         "v_mfma_f64_16x16x4_f64 %1, a[%c7:%c8], %2, 0"
                         : "+v"(d0), "=&v"(zz)
                         : "a"(x0), "i"(SYMW_RING(hs, u)), "i"(SYMW_RING(hs, u) + 1), "i"(SYMW_XJ(SET, u, bcb)), "i"(SYMW_XJ(SET, u, bcb) + 1),
                           "i"(SYMW_P(hs & 1, ib, j)), "i"(SYMW_P(hs & 1, ib, j) + 1));
          else
            asm volatile("v_mfma_f64_16x16x4_f64 %0, a[%c3:%c4], a[%c5:%c6], %0\n\t"
                         "v_mfma_f64_16x16x4_f64 %1, a[%c7:%c8], %2, %1"
                         : "+v"(d0), "+v"(zz)
                         : "a"(x0), "i"(SYMW_RING(hs, u)), "i"(SYMW_RING(hs, u) + 1), "i"(SYMW_XJ(SET, u, bcb)), "i"(SYMW_XJ(SET, u, bcb) + 1),
                           "i"(SYMW_P(hs & 1, ib, j)), "i"(SYMW_P(hs & 1, ib, j) + 1));
          mem_slot(std::integral_constant<int, 2 * gi>{});
          asm volatile("v_mfma_f64_16x16x4_f64 %0, a[%c3:%c4], a[%c5:%c6], %0\n\t"
                       "v_mfma_f64_16x16x4_f64 %1, a[%c7:%c8], %2, %1"
                       : "+v"(d1), "+v"(zz)
                       : "a"(x1), "i"(SYMW_RING(hs, u) + 2), "i"(SYMW_RING(hs, u) + 3), "i"(SYMW_XJ(SET, u, bcb)), "i"(SYMW_XJ(SET, u, bcb) + 1),
                         "i"(SYMW_P(hs & 1, ib, j) + 2), "i"(SYMW_P(hs & 1, ib, j) + 3));
          }
          mem_slot(std::integral_constant<int, 2 * gi + 1>{});
        });
      });
#if DAV_SYMW_STAMPS > 1
      STAMP(fb);
      st_mf += fb - fa;
#endif
    });
    if constexpr (GEN == 0) { ud[0] = ud[1]; ud[1] = ud[2]; ud[2] = udn; }
#ifdef DAV_SYMW_STAMPS
    unsigned long long t4;
    STAMP(t4);
    if (q > 0) { st_hs += t1 - t0; st_bar += t2 - t1; st_sum += t3 - t2; }
    st_unit += t4 - t0;
#endif
  };
  // nunits is a multiple of 16: two pairs per trip (the transposed partials alternate between two register sets)
  for (unsigned q = 0; q < nseq; q += 4) {
    unit(q, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    unit(q + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    unit(q + 2, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    unit(q + 3, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
  }
  // every load and LDS read of the loop has landed before anything else (the compiler does not know about them)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  // the last pair's partials, its sums, then the last strip
  if constexpr (NB == 2) asm volatile(MFMA_DRAIN : "+v"(zcs[1][0][0]), "+v"(zcs[1][0][1]), "+v"(zcs[1][1][0]), "+v"(zcs[1][1][1]));
  else asm volatile(MFMA_DRAIN : "+v"(zcs[1][0][0]), "+v"(zcs[1][1][0]));
#pragma unroll
  for (int w = 0; w < 4 * NB; ++w) z_write(zcs[1], (nseq >> 1) - 1, w);
  __syncthreads();
#pragma unroll
  for (int un = 2; un >= 1; --un) {
#pragma unroll
    for (int sl = 0; sl < NRS; ++sl) sum_issue(nseq - un, sl);
#pragma unroll
    for (int t = 0; t < NB; ++t) sum_finish(nseq - un, t);
  }
  __syncthreads();
  flush_strip(nunits / UPS - 1);
#ifdef DAV_SYMW_STAMPS
  if (lane == 0) {
    unsigned long long tk1, tr1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk1), "=s"(tr1) :: "memory");
    atomicAdd(&symw_stamp[0], st_hs); atomicAdd(&symw_stamp[1], st_bar); atomicAdd(&symw_stamp[2], st_sum);
    atomicAdd(&symw_stamp[3], (unsigned long long)nunits);
    atomicAdd(&symw_stamp[4], tk1 - tk0); atomicAdd(&symw_stamp[5], tr1 - tr0); atomicAdd(&symw_stamp[6], st_vm); atomicAdd(&symw_stamp[9], st_mf);
    atomicAdd(&symw_stamp[10], st_unit); atomicAdd(&symw_stamp[11], 1ull);
  }
#endif

  // end of the run: the direct partials of every row slice (complete: one column group per workgroup), one 32-row
  // half-step and one group at a time through the transposition scratch so that they leave as 256-byte rows
  double* outD = slabD + (int64_t)items[4 * item + 3] * R * 16 * SYM_TB;
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) {
#pragma unroll
    for (int bcb = 0; bcb < NB; ++bcb) {
      const int kc = kcols - 16 * bcb < 16 ? kcols - 16 * bcb : 16;
#pragma unroll
      for (int hs = 0; hs < 4; ++hs) {
        asm volatile(MFMA_DRAIN : "+v"(acc[sl][hs][0][bcb]), "+v"(acc[sl][hs][1][bcb]));
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) tw[c * RS + 2 * (g + 4 * reg) + par] = acc[sl][hs][par][bcb][reg];
        // wave-private scratch, in-order DS: the reads below see the writes above
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int e = lane + 64 * t;                      // (block column, row of the half-step)
          const int bc = e >> 5, r = e & 31;
          const double v = tw[bc * RS + r];
          if (bc < kc && have_row[sl])
            outD[bcb * slabD_gstride + ((int64_t)(2 * sl + (wave >> 1)) * 16 + bc) * SYM_TB + 128 * rhalf + 32 * hs + r] = v;
        }
      }
    }
  }
  // block rows of the super row past the end of the matrix: their slab rows are read by nobody
  // the fixed registers belong to this kernel: the descriptor must allocate all 256 accumulation registers
  asm volatile("" ::: "a128", "a255");
  if constexpr (F32) asm volatile("" ::: "v224", "v255");
}

void launch_matvec_symw(hipStream_t st, int nbw, bool tall, bool tiles_f32, const void* tiles, const int64_t* row_off, int nb, const int* items_dev,
                        int nitems, const int* zslot_begin_dev, const double* xt, int kcols, double* slabD, double* slabT, int nwg,
                        int64_t xt_gstride, int64_t slabD_gstride, int64_t slabT_gstride) {
  dim3 grid(nitems * nwg), block(256);
  const OpParams op{};
  const int64_t n = 0;
#define SYMW_LAUNCH(NBW, T, F)                                                                                                          \
  hipLaunchKernelGGL((matvec_symw_kernel<NBW, T, F, 0>), grid, block, 0, st, tiles, row_off, items_dev, zslot_begin_dev, xt, slabD, slabT, kcols, \
                     nwg, xt_gstride, slabD_gstride, slabT_gstride, nb, op, n)
  if (tiles_f32) SYMW_LAUNCH(1, false, true);
  else if (nbw == 2) SYMW_LAUNCH(2, false, false);
  else if (tall) SYMW_LAUNCH(1, true, false);
  else SYMW_LAUNCH(1, false, false);
#undef SYMW_LAUNCH
}

// the hashed operator at 17-32 columns per launch: no tiles, entries generated in the sweep (matvec_symw_kernel<2, false, false, GEN>)
void launch_matvec_symw_generated(hipStream_t st, OpParams op, int64_t n, int nb, const int* items_dev, int nitems, const int* zslot_begin_dev,
                                  const double* xt, int kcols, double* slabD, double* slabT, int nwg, int64_t xt_gstride, int64_t slabD_gstride,
                                  int64_t slabT_gstride) {
#define SYMW_GEN_LAUNCH(G)                                                                                                                       \
  hipLaunchKernelGGL((matvec_symw_kernel<2, false, false, G>), dim3(nitems * nwg), dim3(256), 0, st, (const void*)nullptr, (const int64_t*)nullptr, \
                     items_dev, zslot_begin_dev, xt, slabD, slabT, kcols, nwg, xt_gstride, slabD_gstride, slabT_gstride, nb, op, n)
  // the reference's test operator in its polynomial form (GEN = 2: cos, 3: sin); the hashed operator (1)
  if (op.kind == DAV_KIND_HARNESS) { if (op.trig != 0) SYMW_GEN_LAUNCH(3); else SYMW_GEN_LAUNCH(2); }
  else SYMW_GEN_LAUNCH(1);
#undef SYMW_GEN_LAUNCH
}
