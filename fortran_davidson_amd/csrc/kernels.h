// Host-side launchers of the gfx950 kernels (definitions in k_*.hip).  All pointers are device
// pointers; all launches go to the given stream; nothing here synchronises.
#pragma once
#include "common.h"

// ---- K1: block matvec ---------------------------------------------------------------------------
// Row tile of one workgroup (4 waves x 64 rows).  Panels and A are padded to a multiple of it.
constexpr int MV_ROWS = 256;
// Pack k columns of a column-major panel into the transposed MFMA-B layout Xt[group][row][16]
// (zero padded) for rows [0, nloc_pad) of this rank, written at row offset `row_off`.
void launch_pack_xt(hipStream_t st, const double* src, int64_t ld, int64_t nloc, int64_t nslab, int k,
                    double* xt, int64_t xt_group_stride, int64_t row_off);
// slab[s][col][row] = A[rows, chunk s] * X[chunk s, col]; ngroups = ceil(k/16) in {1,2,4}.
void launch_matvec_dense(hipStream_t st, const double* A, int64_t lda, int64_t nrows_pad, int64_t ncols_pad,
                         const double* xt, int64_t xt_group_stride, int ngroups, double* slab,
                         int nsplit, int jc);
void launch_matvec_free(hipStream_t st, OpParams op, int64_t row0, int64_t nloc, int64_t n,
                        int64_t nrows_pad, int64_t ncols_pad, const double* xt, int64_t xt_group_stride,
                        int ngroups, double* slab, int nsplit, int jc);
// dst[i, c] = sum_s slab[s][c][i] for i < nloc_pad (rows >= nloc are written as 0), c < k.
void launch_slab_reduce(hipStream_t st, const double* slab, int nsplit, int64_t nrows_pad, int ngroups,
                        int64_t nloc, int k, double* dst, int64_t ldd);
// scratch (in doubles) one matvec launch needs
size_t matvec_slab_doubles(int64_t nrows_pad, int ngroups, int nsplit);
void matvec_plan(int64_t nrows_pad, int64_t ncols_pad, int ngroups, int* nsplit, int* jc, int64_t target = 0, int64_t forced_nsplit = 0);

// ---- K2: tall-skinny Gram ------------------------------------------------------------------------
constexpr int GRAM_MIN_ROWS = 256; // ... and at least (small problems: more, shorter workgroups)
// out (p x q, column-major ld = p) = P^T Q over nrows_pad rows (multiple of 16; pad rows are zero).
// scratch must hold gram_scratch_doubles(...) doubles.  Deterministic two-stage reduction.
// counters: GRAM_MAX_COUNTERS zeroed device words - with at most GRAM_FUSE_CHUNKS row chunks the workgroup that finishes an output
// tile last sums its partial tiles itself (no second launch); nullptr: always the two-kernel route.
constexpr int GRAM_FUSE_CHUNKS = 24;
constexpr int GRAM_MAX_COUNTERS = 256;
void launch_gram(hipStream_t st, const double* P, int64_t ldp, int p, const double* Q, int64_t ldq, int q,
                 int64_t nrows_pad, double* scratch, double* out_dev, unsigned* counters = nullptr, int wg_target = 0);
size_t gram_scratch_doubles(int p, int q, int64_t nrows_pad);
void gram_set_fuse_chunks(int n);
// C = P^T [Q0 | Q1 | Q2] (p x nq*qeach): the columns of the right-hand side come from up to three blocks of qeach columns each
// (same leading dimension) - the projection and the Gram blocks of one iteration in ONE launch and one reduction (round 5)
void launch_gram_multi(hipStream_t st, const double* P, int64_t ldp, int p, const double* const* Qs, int nq, int qeach, int64_t ldq,
                       int64_t nrows_pad, double* scratch, double* out_dev, unsigned* counters = nullptr, int wg_target = 0);

// ---- K3/K4/K5: panel x small matrix ---------------------------------------------------------------
// The small matrices are passed as MFMA-B OPERAND IMAGES (pg_image_index): for step s (panel columns 4 s .. 4 s + 3) and column
// tile t (output columns 16 t .. 16 t + 15) the 64 values the lanes of a wave need - lane c + 16 g holds M[4 s + g][16 t + c] -
// are contiguous, so a wave fetches them with ONE coalesced 512-byte load.  (Round 3 read M column-major: 16 lines of the cache
// touched per 8-byte load, QT of them per step - what bound the kernel: 0.63 of 8 TB/s at QT = 1, 0.25 at QT = 4.)  tp = tiles
// per step of the image (a multiple of 4, >= ceil(q / 16)); rows past p and columns past q are zero.
struct PanelGemmArgs {
  const double* P1; int64_t ld1; int p1; const double* M1; int64_t tp1;   // term 1 (required)
  const double* P2; int64_t ld2; int p2; const double* M2; int64_t tp2;   // term 2 (p2 = 0: absent)
  double* out; int64_t ldo; int q;          // out[:, 0:q]
  int64_t nloc, nrows_pad;                  // valid rows / padded rows (multiple of 128)
  // epilogue: 0 = store; 1 = DPR: out = acc / (theta[col] * dB[row] - dA[row]) plus column norms;
  //           2 = store plus column norms
  int epilogue;
  const double* theta; const double* dA; const double* dB;   // dB == nullptr: B diagonal = 1
  int nnorm; double* norm_partial;          // [gridDim.x][nnorm] partial sums of acc^2 (cols < nnorm)
  // norm_out != nullptr: the LAST workgroup of the launch sums the partials into norm_out[0:nnorm] (fixed order; replaces the
  // launch of norm_finish_kernel); counter = a zeroed device word (dav_last_workgroup).  For small grids only (PG_FUSE_BLOCKS): the
  // device-scope fence every workgroup pays for the pattern writes its XCD's L2 back - measured at +0.15 ms on the 1563
  // workgroups of N=200000 (and +0.28 ms per sweep when the row-slab block matvec summed its column chunks this way: removed)
  double* norm_out; unsigned* counter;
  int pin;                                  // 1: the pinned software pipeline of the k loop (k_panel.hip), 0: the compiler's order
  // batch > 1 (round 5): the same product on `batch` panels in ONE launch (blockIdx.z): P1, P2 and out of panel z lie batch_stride
  // doubles behind those of panel z - 1 (the basis panel, its image under A and - generalized - under B are neighbours in the
  // engine's arena); same small matrices.  Epilogue 0 only.
  int batch; int64_t batch_stride;
};
// doubles of the operand image of a p x q matrix, its tiles per step, and the index of M[i][j] in it
static inline int64_t pg_image_tiles(int q) { return ((int64_t)(q > 0 ? q : 1) + 63) / 64 * 4; }
static inline int64_t pg_image_doubles(int p, int q) { return ((int64_t)(p > 0 ? p : 1) + 3) / 4 * pg_image_tiles(q) * 64; }
static inline int64_t pg_image_index(int i, int j, int64_t tp) { return (((int64_t)(i >> 2) * tp + (j >> 4)) << 6) + (j & 15) + 16 * (i & 3); }
constexpr int PG_ROWS = 128;
constexpr int PG_FUSE_BLOCKS = 48;         // row blocks up to which the Ritz kernel finishes its own norms (N <= 6144)
constexpr int PG_INPLACE_COLS = 64;       // q <= this: one workgroup column (grid.y == 1), so OUT may alias P1 (see k_panel.hip)
void launch_panel_gemm(hipStream_t st, const PanelGemmArgs& a);
// out[j] = sqrt(sum_b partial[b][j])
void launch_norm_finish(hipStream_t st, const double* partial, int nblocks, int nnorm, double* out);

// ---- setup / utilities ------------------------------------------------------------------------
void launch_generate_dense(hipStream_t st, double* A, int64_t lda, int64_t nrows_pad, int64_t ncols_pad,
                           int64_t row0, int64_t nloc, int64_t n, uint64_t seed, double sparsity,
                           int use_diag, double diag_val);
void launch_diag_dense(hipStream_t st, const double* A, int64_t lda, int64_t row0, int64_t nloc, double* diag);
void launch_diag_free(hipStream_t st, OpParams op, int64_t row0, int64_t nloc, double* diag);
// dst[i, c] = A[i, idx[c]] (column gather: A * unit vectors)
void launch_gather_columns(hipStream_t st, const double* A, int64_t lda, int64_t nrows_pad,
                           const int64_t* idx_dev, int k, double* dst, int64_t ldd);
// dst[:, c] = e_{idx[c]} restricted to local rows
void launch_unit_columns(hipStream_t st, const int64_t* idx_dev, int k, int64_t row0, int64_t nloc,
                         int64_t nrows_pad, double* dst, int64_t ldd);
void launch_copy_columns(hipStream_t st, const double* src, int64_t lds, double* dst, int64_t ldd,
                         int64_t nrows_pad, int k);
// stream microbenchmark over n doubles (n even): mode 0: a = b, mode 1: a = b + s c
void launch_stream(hipStream_t st, int mode, double* a, const double* b, const double* c, double s, int64_t n);
void launch_harness_rate(hipStream_t st, double* out, int wgs, int iters);
// out[i] = sum over p = 0 .. nparts-1 of part_p[i] in that order; part_p = own for p == self, stage + slot(p) * count otherwise (slot = p, minus one behind self)
void launch_compare_blocks(hipStream_t st, const double* a, const double* b, int64_t ld, int64_t rows, int k, double* out);
void launch_poke(hipStream_t st, double* p, double delta);
void launch_sum_parts(hipStream_t st, const double* own, const double* stage, int nparts, int self, size_t count, double* out);



// ---- ingest (k_ingest.hip): row-major staged rows -> resident slab / tiles ----------------------------
// stage: nrows complete rows, row-major (ld = ldr), global rows grow0...; dst = full slab (lda, rows
// [slab_row0, slab_row0 + slab_rows) kept) or, sym != 0, the lower block triangle of SYM_TB tiles.
void launch_rows_scatter(hipStream_t st, const double* stage, int64_t ldr, int64_t grow0, int64_t nrows, int64_t n,
                         double* dst, int64_t lda, int64_t slab_row0, int64_t slab_rows, int sym, const int64_t* row_off);

// ---- K7 helpers (k_gjd.hip) ------------------------------------------------------------------------
struct LincombArgs {      // out[:, j] = sum_t coef[t*ldc + j] * in[t][:, j]
  const double* in[4]; const double* coef; int ldc; int nterms;
  double* out; int64_t ld; int64_t nrows_pad; int m;
};
void launch_lincomb(hipStream_t st, const LincombArgs& a);
void launch_precond(hipStream_t st, const double* in, double* out, int64_t ld, int64_t nloc, int64_t nrows_pad, int m,
                    const double* theta, const double* dA, const double* dB, const double* active);
struct DotsArgs {         // partial[block][s*m + j] = <a[s][:, j], b[s][:, j]> over the block's rows
  const double* a[4]; const double* b[4]; int npairs; int64_t ld; int64_t nrows_pad; int m; double* partial;
};
int coldots_blocks(int64_t nrows_pad);
void launch_coldots(hipStream_t st, const DotsArgs& a);

// ---- K1s: symmetric-tiled storage (k_matvec_sym.hip, k_matvec_sym9.hip) ------------------------------------
constexpr int SYM_TB = 256;     // tile edge; tiles (I, J<=I) contiguous, column-major, ld = SYM_TB
// row_off (device, one entry per block row): first tile of block row I in this rank's storage, -1 = stored by another rank
void launch_matvec_sym(hipStream_t st, const double* tiles, const int64_t* row_off, const int* items_dev, int nitems, const double* xt, int kcols,
                       double* slabD, double* slabT, int npair, int64_t xt_gstride, int64_t slabD_gstride, int64_t slabT_gstride);
// the same sweep with the entries of the hashed operator generated in registers (no stored matrix)
void launch_matvec_sym_generated(hipStream_t st, OpParams op, int64_t n, const int* items_dev, int nitems, const double* xt, int kcols,
                                 double* slabD, double* slabT, int npair, int64_t xt_gstride, int64_t slabD_gstride,
                                 int64_t slabT_gstride);
// accumulate: the sums are ADDED to dst (panel layout only: the second part of an operator that is swept in two parts).
// chunk_rows = 0: dst = panel columns (ldd), rows >= nloc zeroed.  chunk_rows = nslab (several ranks): dst = this rank's
// partial product in reduce-scatter layout [rank][column][row of the rank's slab], rows < total_rows
void launch_sym_reduce(hipStream_t st, const double* slabD, const double* slabT, const int* row_item_begin_dev, const int64_t* owned,
                       int nb, int64_t nloc, int k, double* dst, int64_t ldd, int64_t chunk_rows, int64_t total_rows, bool accumulate = false);
// super-row schedules (k_matvec_sym9.hip): R = 2 or 4 block rows per workgroup, transposed partials summed on chip
void launch_matvec_sym9(hipStream_t st, int R, bool gen, const void* tiles, bool tiles_f32, const int64_t* row_off, OpParams op, int64_t n,
                        int nb, const int* items_dev, int nitems, const int* zslot_begin_dev, const double* xt, int kcols, double* slabD,
                        double* slabT, int npair, int64_t xt_gstride, int64_t slabD_gstride, int64_t slabT_gstride, bool mfma4 = true);
// the same sweep (stored fp64 tiles) with one wave per SIMD and 16 nbw block columns per workgroup (k_matvec_symw.hip): R = 2 block
// rows per workgroup, or (tall, nbw = 1: the work items of the R = 4 schedule) four; nwg workgroups per work item cover the
// 16-column groups [0, nbw nwg) of the block
// tiles_f32 (nbw = 1, R = 2): `tiles` is the fp32 copy of the stored tiles (mixed-precision inner sweeps)
void launch_matvec_symw(hipStream_t st, int nbw, bool tall, bool tiles_f32, const void* tiles, const int64_t* row_off, int nb, const int* items_dev,
                        int nitems, const int* zslot_begin_dev, const double* xt, int kcols, double* slabD, double* slabT, int nwg,
                        int64_t xt_gstride, int64_t slabD_gstride, int64_t slabT_gstride);
void launch_matvec_symw_generated(hipStream_t st, OpParams op, int64_t n, int nb, const int* items_dev, int nitems, const int* zslot_begin_dev,
                                  const double* xt, int kcols, double* slabD, double* slabT, int nwg, int64_t xt_gstride, int64_t slabD_gstride,
                                  int64_t slabT_gstride);
// fp32 copy of `count` stored tile entries (count a multiple of 4)
void launch_tiles_to_f32(hipStream_t st, const double* src, float* dst, int64_t count);
void launch_sym9_reduce(hipStream_t st, const double* slabD, const double* slabT, const int* row_item_begin_dev,
                        const int* zslot_begin_dev, const int64_t* owned, const int* next_owned, int R, int nb, int64_t nloc, int k, double* dst,
                        int64_t ldd, int64_t chunk_rows, int64_t total_rows, bool accumulate = false);
void launch_generate_sym_tiles(hipStream_t st, double* tiles, const int64_t* row_off_host, int nb, int64_t n, uint64_t seed,
                               double sparsity, int use_diag, double diag_val);
void launch_retile_panel(hipStream_t st, const double* panel, int64_t ldp, int64_t nrows, int ncols, int J, int nb,
                         const int64_t* row_off, double* tiles);
void launch_diag_sym(hipStream_t st, const double* tiles, const int64_t* row_off, int64_t n, int64_t nrows, double* diag);
void launch_gather_columns_sym_rs(hipStream_t st, const double* tiles, const int64_t* row_off, int64_t n, int64_t chunk_rows,
                                  int64_t total_rows, const int64_t* idx_dev, int k, double* dst);
// h0[i + j * k] = what this rank's tiles hold of the operator's entry (idx[i], idx[j]) (0 where another rank stores it)
void launch_entries_sym(hipStream_t st, const double* tiles, const int64_t* row_off, const int64_t* idx_dev, int k, double* h0);
void launch_zero_pad_rows(hipStream_t st, double* dst, int64_t ldd, int64_t nloc, int64_t nrows_pad, int k);
void launch_entries_free(hipStream_t st, OpParams op, const int64_t* idx_dev, int k, double* h0);
void launch_entries_dense(hipStream_t st, const double* A, int64_t lda, const int64_t* idx_dev, int k, double* h0);
void launch_gather_columns_free(hipStream_t st, OpParams op, int64_t row0, int64_t nloc, int64_t nrows_pad, const int64_t* idx_dev, int k,
                                double* dst, int64_t ldd);
void launch_gather_columns_sym(hipStream_t st, const double* tiles, const int64_t* row_off, int64_t n, int64_t nrows_pad,
                               const int64_t* idx_dev, int k, double* dst, int64_t ldd);
// dst[i, c] = i < nloc ? src[c * lds + i] : 0 for i < nrows_pad (the received chunk of a reduce-scatter -> panel columns)
void launch_chunk_to_panel(hipStream_t st, const double* src, int64_t lds, int64_t nloc, int64_t nrows_pad, int k, double* dst, int64_t ldd,
                           bool accumulate = false);

// ---- device-side Rayleigh-Ritz (k_smalleig.hip): all eigenpairs of H y = theta y / H y = theta S y, order m <= 128 ------
size_t small_eig_work_doubles(int m);
bool launch_small_eig(hipStream_t st, const double* H, int64_t ldh, const double* S, int64_t lds, int m, bool gev, double* theta,
                      double* Y, int64_t ldy, double* work, double* info);
void launch_rr_scatter(hipStream_t st, const double* blk, int mt, int k, int c0, double* Hd, int64_t ld);
void launch_rr_pack(hipStream_t st, const double* Y, int64_t ld, const double* theta, int m, int q, int ldm, int qpad, double* Ypk,
                    double* Y2pk, double* theta_pk, const double* info, double* result_tail);
