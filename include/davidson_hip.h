/* davidson_hip.h - C ABI of the MI355X-native Davidson engine (libdavidson_hip.so).
 *
 * This is the drop-in boundary for the hot path of NLESC-JCER/Fortran_Davidson: the body of the
 * `outer_loop` of generalized_eigensolver_dense (src/davidson.f90:138-229) and
 * generalized_eigensolver_free (src/davidson.f90:375-441).  The reference is pure Fortran and has
 * no FFI of its own; these entry points are what its Fortran modules bind through ISO_C_BINDING
 * (fortran_davidson_amd/fortran/davidson_hip_c.f90) in place of the
 * BLAS/LAPACK wrapper calls cited on each function.
 *
 * Conventions: every function returns 0 on success, non-zero on failure (dav_last_error() gives
 * the text; the Fortran shim prints it and `error stop`s like check_lapack_call,
 * src/lapack_wrapper.f90:395-408).  All matrices are IEEE binary64, column-major, with explicit
 * leading dimensions; pointers are HOST pointers unless the name says `_dev`; sizes that can exceed
 * 2^31 are int64_t.  The engine owns all device memory behind the opaque handle; nothing survives
 * dav_destroy.  Only the m x m Rayleigh-Ritz problem (DSYEV/DSYGV, src/lapack_wrapper.f90:14-91)
 * and the k x k basis transforms stay on the host.
 *
 * Row-slab sharding (one process per GPU): rank r of P owns global rows [r*nloc, r*nloc+nloc) of A,
 * B and of every N-long panel; the new basis block is exchanged with one RCCL all-gather, small
 * Gram blocks with one RCCL all-reduce.  P = 1 uses the same code without communication.
 */
#ifndef DAVIDSON_HIP_H
#define DAVIDSON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dav_engine* dav_handle_t;

/* which operator */
enum { DAV_OP_A = 0, DAV_OP_B = 1 };
/* device panels (N-long, column-major, resident in HBM) */
enum { DAV_PANEL_V = 0,   /* search-space basis V (and, in its tail columns, the correction block T) */
       DAV_PANEL_W = 1,   /* A*V            (src/davidson.f90:131,223 inner DGEMM)                 */
       DAV_PANEL_BV = 2,  /* B*V            (src/davidson.f90:134,226)                             */
       DAV_PANEL_X = 3,   /* Ritz vectors   (src/davidson.f90:159)                                 */
       DAV_PANEL_R = 4,   /* residues / scratch (src/davidson.f90:163-170)                         */
       DAV_PANEL_S = 5 }; /* scratch for out-of-place block transforms                             */
/* correction method (src/davidson.f90:656-669) */
enum { DAV_METHOD_DPR = 0, DAV_METHOD_GJD = 1,
       DAV_METHOD_NONE = 2 /* residues and norms only: the iteration will restart, no correction */ };

typedef struct dav_stats {
  int64_t n;               /* global order                                                          */
  int64_t nloc;            /* rows owned by this rank                                               */
  int32_t nranks, rank;
  int32_t m;               /* current basis width                                                   */
  int32_t applies;         /* block applies of A so far                                             */
  int64_t apply_cols;      /* total columns A was applied to                                        */
  double apply_ms;         /* device time of the A*X block applies END TO END (HIP events): operand   */
                           /* packing + (all-gather) + block-matvec kernel + partial-sum reduction   */
  double apply_bytes;      /* algorithmic bytes of those applies: 8*S + 16*N*k each, S = nloc*N      */
                           /* (full row slab) or N(N+1)/2 (symmetric-tiled)                          */
  double last_apply_ms;    /* duration of the most recent A apply (end to end)                      */
  double last_apply_bytes;
  double gram_ms, panel_ms, comm_ms;
  double apply_kernel_ms;  /* the block-matvec kernel alone inside apply_ms (same launches)          */
  double apply_flops;      /* 2*nloc*N*k per apply and rank (2*N*N*k / nranks with symmetric tiles:  */
                           /* every stored entry is used twice)                                      */
  int64_t apply_launches;  /* launches of the block-matvec kernel (an apply of > 32 / 64 columns is  */
                           /* several launches)                                                      */
  int64_t restarts;        /* collapse restarts (dav_restart / dav_rr_restart) so far                */
  /* collectives (dav_set_timing level 2; several ranks): HIP-event time on the stream they run on, and */
  /* the payload per rank - all-gather: bytes received (nranks * count * 8), reduce-scatter: bytes      */
  /* contributed (nranks * count * 8), all-reduce: count * 8                                            */
  double allgather_ms, reduce_scatter_ms, allreduce_ms;
  double allgather_bytes, reduce_scatter_bytes, allreduce_bytes;
  int64_t collectives;     /* collectives (or groups of them) issued so far, at any timing level      */
  int32_t comm_ranks;      /* ranks the RCCL communicator reports (ncclCommCount); 0 = no communicator */
  int32_t comm_overlap;    /* 1 = wide blocks run their collectives on a second stream under the sweeps */
  double apply_comm_ms;    /* (ABI 104) the part of the collectives' time that lies INSIDE apply_ms (all-gather / reduce-scatter of the */
                           /* applies, timing level 2): apply_ms - apply_comm_ms = packing + kernel + reduction of this rank            */
  /* (ABI 104) sweep kernels of the SECOND operator by what they read (timing level 2): stored tiles (a dense B, or the resident  */
  /* block rows of a generated one) - bytes = 8 * stored entries + 16 N k per launch - and generated block rows - entries =       */
  /* operator entries evaluated per launch                                                                                        */
  double b_stored_kernel_ms, b_stored_bytes, b_stored_flops;
  double b_generated_kernel_ms, b_generated_entries, b_generated_flops;
  int64_t b_stored_launches, b_generated_launches;
} dav_stats;

/* ABI version of this header.  dav_version() of the loaded library must return the same number: a     */
/* caller built against another layout of the statistics structure (it grew in 101 and 102; 103 added dav_device_memory, 104 dav_agree_next, 105 dav_free_buffers, 106 dav_set_operator_device, 107 DAV_NO_SUCH_ENTRY, 108 dav_comm_path) must not    */
/* use the unsized call - the sized one, which copies at most `bytes` bytes, is safe across versions.   */
#define DAV_HIP_ABI_VERSION 108
const char* dav_last_error(void);
int dav_version(void);

/* ---- lifetime ------------------------------------------------------------------------------- */
/* Create an engine for a problem of order n on `device`.  max_cols bounds the basis width
 * (2*max_dim as the reference can reach, src/davidson.f90:195-213).  gev != 0 reserves the B*V
 * panel.  rank/nranks describe the row-slab partition (0/1 for a single GPU). */
int dav_create(dav_handle_t* h, int device, int64_t n, int max_cols, int gev, int rank, int nranks);
int dav_destroy(dav_handle_t h);
/* RCCL bootstrap: rank 0 fills a 128-byte id, the launcher (torch.distributed / MPI / a file)
 * distributes it, every rank calls dav_comm_init.  Not needed when nranks == 1. */
int dav_comm_unique_id(void* id128);
int dav_comm_init(dav_handle_t h, const void* id128);
/* Several ranks (ABI 108): which way the collectives of a wide block (more than 32 columns) of the symmetric sweep go - RCCL's all-gather /
 * reduce-scatter in program order (0), direct exchanges with every peer over the point-to-point links (1), or 32-column chunks whose
 * collectives run on a second stream under the sweeps (2) - is decided by the engine at the first such block over a real communicator:
 * the block goes through all three, the results are compared with the program-order one (bitwise; the direct exchange's rank-order sums
 * to 1e-12 beyond two ranks), the times (HIP events, maximum over the ranks) are made common with one small all-reduce and the fastest
 * validated way is kept; a way that fails is left out with a message on stderr.  DAV_COLL_SELECT=0, DAV_SYM_OVERLAP or DAV_COLL_DIRECT in
 * the environment at dav_create switch the trial off.  This call reports the outcome: *selected = -1 before the trial, ms3 / valid3 per
 * way (0 / 1 / 2 as above). */
int dav_comm_path(dav_handle_t h, int* selected, int* trial_ran, int* columns, double* ms3, int* valid3);
int dav_synchronize(dav_handle_t h);
int dav_get_stats(dav_handle_t h, dav_stats* out);
int dav_get_stats_n(dav_handle_t h, void* out, size_t bytes);
int dav_reset_stats(dav_handle_t h);
/* What dav_get_stats measures with HIP events on the engine's stream: 0 = nothing, 1 (default) = the block
 * applies (apply_ms: pack + kernel + reduction, apply_kernel_ms: the roofline kernel alone), 2 = also the
 * Gram, panel and collective phases (gram_ms, panel_ms, comm_ms; each event pair costs ~5 us of host time
 * per launch group). */
int dav_set_timing(dav_handle_t h, int level);
/* rows of this rank: [row0, row0+nloc) */
int dav_local_rows(dav_handle_t h, int64_t* row0, int64_t* nloc);

/* ---- operators (replace `matrix` / `second_matrix` / fun_matrix_gemv) ------------------------- */
/* Storage of dense operators set AFTER this call: 0 = full row slabs (default), 1 = symmetric-tiled: only
 * the lower block triangle (256 x 256 tiles) is kept in HBM - N(N+1)/2 entries, e.g. 160 GB instead of
 * 320 GB at N = 200000 - and the block matvec uses every off-diagonal tile twice (A is assumed symmetric,
 * as the reference assumes).  Both work with any number of ranks: full = rank r keeps its rows of
 * A (all-gather of the new block per sweep); symmetric = the block rows of the triangle are dealt out over
 * the ranks, N*N/2P entries each (all-gather + reduce-scatter per sweep). */
int dav_set_storage(dav_handle_t h, int mode);
/* Free and total memory of the engine's device in bytes (hipMemGetInfo) - what a front end needs to decide, before it uploads, whether
 * a dense operator fits as full rows (8 N^2 / nranks bytes) or only as symmetric tiles (half of that). */
int dav_device_memory(dav_handle_t h, int64_t* free_bytes, int64_t* total_bytes);
/* Buffer cache.  dav_destroy keeps the device and pinned blocks of the engine it destroys (>= 64 KiB) for the next dav_create of the
 * same sizes - the reference's interface (src/davidson.f90:51-52) is one call per eigenproblem, so a caller in a loop creates and
 * destroys an engine per call; blocks idle through a whole create-destroy cycle are freed, a failing allocation frees them all and
 * tries again, dav_device_memory counts them as free.  dav_free_buffers() returns every idle block to the device now
 * (what mkl_free_buffers is to MKL); DAVIDSON_BUFFER_CACHE=0 in the environment turns the cache off; it never holds more than
 * DAVIDSON_BUFFER_CACHE_MB megabytes of device memory (default 4096: the call it exists for is the small and frequent one) and
 * DAVIDSON_BUFFER_CACHE_PINNED_MB of page-locked host memory (default 256); a block above 512 MiB that an engine frees in the middle of
 * its life goes straight back.  What the cache holds is invisible to every other allocator of the device (the caller's hipMalloc,
 * PyTorch, RCCL, other processes): a co-tenant that needs the memory calls dav_free_buffers() after the solve, or runs with the cache
 * off.  Nothing is released at process exit - the driver reclaims a dead process's memory, and HIP calls from exit handlers are
 * not safe against the runtime's own teardown. */
int dav_free_buffers(void);
/* Dense matrix from host memory, full storage a(lda, n), the caller's array as passed to
 * generalized_eigensolver_dense (src/davidson.f90:75-76).  Copies this rank's row slab to HBM and
 * extracts the diagonal (replaces array_utils.f90:115-134). */
int dav_set_dense_host(dav_handle_t h, int which, const double* a, int64_t lda);
/* Same with a(lda, n) already in device memory (e.g. produced by an upstream GPU stage): copied once
 * into the engine's padded slab / tile layout, the caller's buffer is not referenced afterwards. */
int dav_set_dense_dev(dav_handle_t h, int which, const double* a_dev, int64_t lda);
/* Streaming upload by blocks of COMPLETE ROWS, row-major - the order of the reference's on-disk format
 * (read_matrix / write_matrix, src/tests/test_utils.f90:118-135,150-166: list-directed text, one value
 * per line, row i outer, column j inner) - so that a matrix larger than host memory, or one produced row
 * by row, reaches HBM without a host N x N copy.  begin allocates and zeroes the resident storage (full
 * row slab or symmetric tiles, as dav_set_storage says); put_rows takes rows [row0, row0 + nrows) with
 * element (row0 + r, j) at rows[r * ldr + j], any order, global row numbers (rows of other ranks are
 * ignored; with symmetric-tiled storage the upper block triangle is dropped); the caller's buffer may be
 * reused as soon as the call returns; end extracts the diagonal (array_utils.f90:115-134) and makes the
 * operator usable. */
int dav_dense_begin(dav_handle_t h, int which);
int dav_dense_put_rows(dav_handle_t h, int which, int64_t row0, int64_t nrows, const double* rows, int64_t ldr);
int dav_dense_end(dav_handle_t h, int which);
/* Whole matrix from a file through the streaming path above.  DAV_FILE_TEXT: the reference's text format
 * (whitespace/comma separated reals, E/D exponents, n*n values, row-major; what test_utils.f90 read_matrix
 * reads and write_matrix writes).  DAV_FILE_F64: raw little-endian float64, row-major, exactly 8 n^2
 * bytes (each rank preads only its rows). */
enum { DAV_FILE_TEXT = 0, DAV_FILE_F64 = 1 };
int dav_set_dense_file(dav_handle_t h, int which, const char* path, int format);
/* The text parser of DAV_FILE_TEXT on a memory buffer (host only, no GPU needed): *nvals = numbers found,
 * the first min(*nvals, max_vals) are stored to out (out may be NULL to count). */
int dav_parse_text_f64(const char* text, size_t len, double* out, size_t max_vals, size_t* nvals);
/* Dense matrix generated directly in HBM with the semantics of generate_diagonal_dominant
 * (src/array_utils.f90:86-113) and the counter-based stream of oracle/davidson_oracle.py. */
int dav_set_dense_generated(dav_handle_t h, int which, uint64_t seed, double sparsity,
                            int use_diag_val, double diag_val);
/* Matrix-free operators evaluated on the fly inside the block matvec (no N x N storage):
 * the hashed diagonal-dominant operator (same entries as dav_set_dense_generated) ...          */
int dav_set_operator_hashed(dav_handle_t h, int which, uint64_t seed, double sparsity,
                            int use_diag_val, double diag_val);
/* ... the reference test harness operators (src/tests/test_utils.f90:38-116): which==A gives the
 * cos generator + i on the diagonal, which==B the sin generator with unit diagonal; e_table[n]
 * holds exp(real(i)/real(n)) as the host evaluates it in single precision ...                   */
int dav_set_operator_harness(dav_handle_t h, int which, const double* e_table);
/* ... and B = I (src/benchmark_free.f90:65-76). */
int dav_set_operator_identity(dav_handle_t h, int which);
/* The operator is applied by the host (API-faithful callback path of generalized_eigensolver_free,
 * src/davidson.f90:378-379): the driver moves blocks with dav_panel_get/put.  diag[n] is the
 * operator's diagonal (what extract_diagonal_free, src/davidson.f90:490-523, computes). */
int dav_set_operator_host(dav_handle_t h, int which, const double* diag);
/* The caller's OWN operator as a block apply on device memory - the device counterpart of the reference's matrix-free interface
 * (src/davidson.f90:277-337: a procedure X(N,k) -> (N,k) on host arrays): for a caller who has the operator as a HIP kernel (a stencil,
 * a sparse or structured product, ...) and wants the blocks to stay in HBM.  `fn` must ENQUEUE on `hip_stream` (a hipStream_t) the work
 * that leaves  Y[0:nloc, 0:k] = Op[row0 : row0 + nloc, 0:n] * X[0:n, 0:k]  in y_dev (column-major, leading dimension ldy) - x_dev
 * holds ALL n rows of the block (column-major, ldx; several ranks: the engine has all-gathered it), rows [row0, row0 + nloc) are this
 * rank's slab (dav_local_rows) - and return 0; it must not synchronise the device.  diag: the operator's diagonal, n doubles on the
 * host (the DPR preconditioner and the start vectors need it - src/davidson.f90:490-523 probes it with N unit vectors).  Since ABI 106. */
typedef int (*dav_device_apply_fn)(void* ctx, void* hip_stream, int64_t n, int64_t row0, int64_t nloc, int k, const double* x_dev, int64_t ldx,
                                   double* y_dev, int64_t ldy);
int dav_set_operator_device(dav_handle_t h, int which, dav_device_apply_fn fn, void* ctx, const double* diag);
int dav_get_diagonal(dav_handle_t h, int which, double* diag_out /* n, global */);

/* ---- the per-iteration hot path ---------------------------------------------------------------- */
/* K6 - replaces diagonal + generate_preconditioner + lapack_sort (src/davidson.f90:127-128,
 * array_utils.f90:136-160): V0 = unit vectors at the ncols smallest diagonal entries (stable order),
 * W0 = A*V0 (and B*V0).  idx_out[ncols] receives the 1-based positions.  Sets m = ncols. */
int dav_init_basis(dav_handle_t h, int ncols, int64_t* idx_out);
/* K1 - replaces lapack_matmul('N','N', matrix, V) (src/davidson.f90:131,223 inner;
 * lapack_wrapper.f90:279-328): dst[:, d0:d0+k] = Op(which) * src[:, c0:c0+k]. */
int dav_apply(dav_handle_t h, int which, int src_panel, int c0, int k, int dst_panel, int d0);
/* K2 - replaces lapack_matmul('T','N', V, .) (src/davidson.f90:131,223 outer):
 * out(p x q) = P[:, p0:p0+p]^T * Q[:, q0:q0+q], summed over all ranks, returned on the host. */
int dav_gram(dav_handle_t h, int panel_p, int p0, int p, int panel_q, int q0, int q,
             double* out, int64_t ldo);
/* New columns of the projected matrices after the basis grew from c0 to c0+k columns:
 * H[0:c0+k, c0:c0+k] = V^T W[:, c0:c0+k]; S likewise with B*V when gev (S may be NULL). */
int dav_project(dav_handle_t h, int c0, int k, double* H, int64_t ldh, double* S, int64_t lds);
/* K3 - replaces the Ritz-vector DGEMM, the m residual DGEMVs, norm() and compute_DPR_*
 * (src/davidson.f90:159-178, 673-698, 463-488): with Y (m x m) and theta (m) from the host
 * eigensolver computes X = V*Y(:, 1:nx), R = W*Y - Z*Y*diag(theta) (Z = B*V when gev else V),
 * resnorm[j] = ||R(:, j)||_2 for j < lowest and, for DPR, the correction block
 * T = R ./ (theta_j * diagB_i - diagA_i) written into V[:, m:2m].  For GJD R and X (nx = m) are
 * left in their panels for dav_gjd_correction. */
int dav_ritz_residual_correction(dav_handle_t h, int m, int lowest, const double* Y, int64_t ldy,
                                 const double* theta, int method, double* resnorm);
/* Opt-in variant (not in the reference): the same phase for the first ncorr Ritz pairs only
 * (Y is m x ncorr, theta has ncorr entries, lowest <= ncorr <= m): X = V*Y, R, norms of the first `lowest`
 * columns, DPR block T (ncorr columns) into V[:, m:m+ncorr].  dav_ritz_residual_correction is the case
 * ncorr = m.  dav_panel_select then keeps the chosen columns of a block (e.g. the corrections of the pairs
 * that have not converged): panel[:, c0+i] = panel[:, c0+sel[i]], sel ascending. */
int dav_ritz_residual_correction_n(dav_handle_t h, int m, int ncorr, int lowest, const double* Y, int64_t ldy,
                                   const double* theta, int method, double* resnorm);
int dav_panel_select(dav_handle_t h, int panel, int c0, int nsel, const int* sel);
/* DPR variant of dav_ritz_residual_correction_n that also returns the Gram blocks of the first
 * orthonormalisation pass, C = V[:, 0:m]^T T (m x ncorr) and G = T^T T (ncorr x ncorr) with T the correction
 * block just written - what dav_ortho_gram(h, m, ncorr, ...) would return - in the same reduction and the
 * same fetch as the residual norms: one host-device round trip less per iteration. */
int dav_ritz_residual_correction_g(dav_handle_t h, int m, int ncorr, int lowest, const double* Y, int64_t ldy,
                                   const double* theta, double* resnorm, double* C, int64_t ldc, double* G, int64_t ldg);
/* The Ritz vectors X = V*Y(:, 0:lowest) are what the caller gets back at the END (src/davidson.f90:186-187); the DPR correction
 * never reads them.  dav_set_lazy_ritz_vectors(h, 1): the Ritz phases above leave X alone unless the method is GJD (whose
 * correction solve needs x_k), and the driver asks for them once, when it stops: dav_ritz_vectors(h, m, nx, Y, ldy) makes
 * X[:, 0:nx] = V[:, 0:m] * Y(:, 0:nx).  One panel product (m + nx columns of N rows) and one launch less per outer iteration.
 * Default 0: every Ritz phase computes X, as documented above. */
int dav_set_lazy_ritz_vectors(dav_handle_t h, int on);
int dav_ritz_vectors(dav_handle_t h, int m, int nx, const double* Y, int64_t ldy);
/* K7 - replaces compute_GJD_generalized_dense (src/davidson.f90:700-734): solves
 * (I - x x^T)(A - theta_k B)(I - x x^T) t_k = -r_k for all m Ritz pairs at once with a block
 * preconditioned MINRES whose operator is the K1 block matvec; T goes to V[:, m:2m]. */
int dav_gjd_correction(dav_handle_t h, int m, const double* theta, int max_inner, double inner_tol,
                       int* inner_iters_out);
/* Same for ncols <= m pairs whose X and R sit in the first ncols columns of their panels; T goes to
 * V[:, m:m+ncols] (opt-in correction policy, see dav_ritz_residual_correction_n).  tol_per_col (ncols
 * entries, or NULL = inner_tol everywhere) sets the relative residual at which each pair's inner solve stops; a NEGATIVE entry
 * marks a follower: it stops at |tol| or as soon as every pair with a positive entry has stopped, whichever comes first (the
 * corrections of the Ritz pairs beyond `lowest` only enrich the basis). */
int dav_gjd_correction_n(dav_handle_t h, int m, int ncols, const double* theta, int max_inner, double inner_tol,
                         const double* tol_per_col, int* inner_iters_out);
/* K4 - replaces concatenate + lapack_qr (src/davidson.f90:210-213): block Gram-Schmidt of the
 * k = kt correction columns T = V[:, m:m+kt] against V[:, 0:m] and among themselves.
 * dav_ortho_gram returns C = V^T T (m x kt) and G = T^T T (kt x kt); the host factors
 * G - C^T C and calls dav_ortho_apply with M (kt x kt): T <- (T - V*C) * M. */
int dav_ortho_gram(dav_handle_t h, int m, int kt, double* C, int64_t ldc, double* G, int64_t ldg);
int dav_ortho_apply(dav_handle_t h, int m, int kt, const double* C, int64_t ldc,
                    const double* M, int64_t ldm);
/* (ABI 104) The LAST Gram-Schmidt pass fused with the projection: the driver sweeps the block as the first pass left it (T',
 * orthonormal to ~1e-8: dav_expand), then ONE reduction / fetch returns the raw projected blocks [V T']^T (A T') (and, S_raw
 * != NULL on a generalized engine, [V T']^T (B T')), each (m + k) x k, together with C = V^T T' and G = T'^T T' of that pass.
 * The pass is linear, T'' = (T' - V C) M, so dav_ortho_apply_all applies it to T' AND to its images A T', B T' in the W / BV
 * panels, and the projected blocks of T'' follow on the host: V^T A T'' = (H_raw_V - H C) M, T''^T A T'' = M^T (H_raw_T - C^T
 * H_raw_V - H_raw_V^T C + C^T H C) M.  Replaces dav_ortho_gram + dav_project of the same iteration: one host round trip and,
 * with several ranks, one all-reduce fewer per outer iteration.  Not with dav_rr_enable. */
int dav_project_ortho(dav_handle_t h, int m, int k, double* H_raw, int64_t ldh, double* S_raw, int64_t lds,
                      double* C, int64_t ldc, double* G, int64_t ldg);
int dav_ortho_apply_all(dav_handle_t h, int m, int kt, const double* C, int64_t ldc, const double* M, int64_t ldm);
/* Commit T as basis columns m..m+kt-1 and apply the operators to them (device operators only):
 * W[:, m:m+kt] = A*T (one block sweep of A - the only one per iteration), BV likewise. */
int dav_expand(dav_handle_t h, int m, int kt);
/* K5 - replaces V = V * Y(:, 1:keep) (src/davidson.f90:218) AND the re-application of the operators to the whole
 * basis that follows it in the reference (:223-226): V, W = A*V and (generalized problems) B*V are all contracted
 * with the same keep columns of Yk (m x keep), W*Yk = A*(V*Yk) to rounding, so no sweep of A or B follows a
 * restart; sets m = keep.  Generalized problems: pass Yk already multiplied by the k x k transform that makes V*Yk
 * Euclidean-orthonormal (Yk^T Yk = I), as the Fortran driver does, so that all three panels carry it. */
int dav_restart(dav_handle_t h, int m, int keep, const double* Yk, int64_t ldy);
/* Several ranks: returns an error if the ranks do not all pass the same `words` (the driver's control decisions of this
 * iteration) - one small all-reduce; a no-op on a single rank.  Lets a diverged rank end with a message instead of
 * leaving its peers in a collective. */
int dav_ranks_agree(dav_handle_t h, const double* words, int nwords);
/* The same check without a collective of its own (ABI 104): the words (at most 16) wait in the engine and ride on the NEXT
 * all-reduced small result - the residual norms and Gram blocks of dav_ritz_residual_correction_* - behind the payload; the call
 * that fetches that result fails on every rank when the words differ.  The Fortran driver calls it at the top of every outer
 * iteration with (iteration, basis width, grow-or-restart, corrections, tolerance, converged flags): together with the
 * all-reduced (hence bitwise identical) residual norms they determine every decision of the iteration.  One rank: a no-op. */
int dav_agree_next(dav_handle_t h, const double* words, int nwords);
/* The INPUTS of a solve (ABI 108; at most 16 words: order, wanted pairs, restart width, iteration limit, tolerance, policy, method ...),
 * verified across the ranks with dav_ranks_agree - a collective whose size does not depend on them, which the all-reduces the words of
 * dav_agree_next ride on are not: their element counts follow from the basis width - whenever they differ from what this engine verified
 * last; the first solve on an engine always verifies, repeated solves with unchanged inputs add no collective.  One rank: a no-op. */
int dav_agree_inputs(dav_handle_t h, const double* words, int nwords);
/* Mixed-precision correction path (opt-in).  bits = 32: the block sweeps inside the GJD correction solve
 * (replacing the dense projected solves of src/davidson.f90:700-734 + src/lapack_wrapper.f90:238-277) read an fp32 copy of
 * the stored symmetric tiles (made on first use; half the bytes per inner sweep), widen to fp64 in registers and
 * accumulate in fp64.  The expansion sweep, projections, residuals and the convergence test keep reading the fp64
 * matrix, so eigenvalues and residual norms are unaffected; only the correction vectors carry fp32-rounded operator
 * entries.  bits = 64 (default) is the reference's precision throughout. */
int dav_set_inner_precision(dav_handle_t h, int bits);
/* ---- device-resident Rayleigh-Ritz (opt-in) -------------------------------------------------------
 * Replaces lapack_generalized_eigensolver (src/lapack_wrapper.f90:14-91: DSYEV / DSYGV itype=1,'V','U', called at
 * src/davidson.f90:152-156 and :394) together with the transfers around it: after dav_rr_enable(h, 1) dav_project also
 * keeps the projected matrices H (and S) in HBM, dav_rr_ritz solves H y = theta y / H y = theta S y for ALL m pairs
 * on the device (one-workgroup cyclic Jacobi, generalized case through a Cholesky factor of S; m <= 128) and runs the
 * Ritz phase of dav_ritz_residual_correction_n / _g from the eigenpairs where they lie.  One synchronisation returns
 * theta_out[0..m) (ascending), resnorm[0..lowest) and, if C != NULL, the Gram blocks C (m x ncorr) and G (ncorr x
 * ncorr) of the first orthonormalisation pass.  Eigenvectors are normalised as DSYEV / DSYGV normalise them (Y^T Y = I,
 * Y^T S Y = I); their signs are the eigensolver's own.  sweeps_out: Jacobi sweeps used (may be NULL). */
int dav_rr_enable(dav_handle_t h, int on);
/* dav_project without host copies and without a synchronisation (the device keeps H and S) */
int dav_project_dev(dav_handle_t h, int c0, int k);
int dav_rr_ritz(dav_handle_t h, int m, int ncorr, int lowest, int method, double* theta_out, double* resnorm, double* C,
                int64_t ldc, double* G, int64_t ldg, int* sweeps_out);
/* collapse restart with the device-resident eigenvectors (standard problems): V, W <- V, W * Y(:, 1:keep)
 * (src/davidson.f90:218, :223) */
int dav_rr_restart(dav_handle_t h, int m, int keep);
/* the device-resident Ritz values / eigenvectors (m x ncols) on the host (either pointer may be NULL) */
int dav_rr_get(dav_handle_t h, int m, int ncols, double* theta, double* Y, int64_t ldy);
/* Generic block transform dst[:, d0:d0+q] = src[:, s0:s0+p] * M (p x q); used for the basis
 * re-orthonormalisation after a generalized restart. */
int dav_panel_transform(dav_handle_t h, int src_panel, int s0, int p, const double* M, int64_t ldm,
                        int q, int dst_panel, int d0);

/* ---- block movement (results, host-operator path, tests) --------------------------------------- */
/* host(ld, k) <- panel[:, c0:c0+k]: all N rows (gathered over ranks) */
int dav_panel_get(dav_handle_t h, int panel, int c0, int k, double* out, int64_t ld);
/* panel[:, c0:c0+k] <- host(ld, k) holding all N rows (each rank keeps its slab) */
int dav_panel_put(dav_handle_t h, int panel, int c0, int k, const double* in, int64_t ld);
/* panel[:, col] <- the unit vector at the (k+1)-th smallest diagonal entry of operator A (the order dav_init_basis takes its start
 * vectors from; k counts from 0).  What the driver completes a rank-deficient correction block with (the reference's Householder QR
 * leaves unit vectors in such columns, src/davidson.f90:197-215).  Returns DAV_NO_SUCH_ENTRY (2; until ABI 106: 1, the code of every
 * failure) - and leaves the column alone - when the engine keeps no (k+1)-th entry of that order: not an error, the caller takes
 * another direction; 1 is a failure like everywhere else (dav_last_error).  Since ABI 106. */
#define DAV_NO_SUCH_ENTRY 2
int dav_panel_unit_column(dav_handle_t h, int panel, int col, int k);
int dav_set_width(dav_handle_t h, int m);

#ifdef __cplusplus
}
#endif
#endif /* DAVIDSON_HIP_H */
