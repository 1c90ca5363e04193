// Engine, operators: storage modes (full row slabs, symmetric tiles dealt out over the ranks), work lists of the symmetric
// sweep, dense matrices from host / device memory / row streams / files (ingest glue), generated and matrix-free operators,
// diagonals.
#include "engine_internal.h"

int refresh_diag_host(E* e, int which) {
  // global diagonal on the host (stable top-k selection, dav_get_diagonal)
  if (which == DAV_OP_A) e->basis_order.clear();
  std::vector<double>& d = e->diag_host[which];
  d.assign((size_t)e->n, 0.0);
  if (!has_comm(e)) {
    CHK(need_comm(e));
    HIPCHK(hipMemcpyAsync(d.data(), e->op[which].diag, sizeof(double) * e->n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  } else {
    CHK(coll_allgather(e, e->op[which].diag, e->gather_dev, (size_t)e->nslab));
    HIPCHK(hipMemcpyAsync(d.data(), e->gather_dev, sizeof(double) * e->n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  return 0;
}

// Block rows per workgroup of the symmetric sweep for a launch of kk columns: 4 (k <= 8 - and 9-16 columns of stored fp64 tiles -
// from 200 block rows on), 2 (more than 8 columns, from 64 block rows on: the one-wave-per-SIMD kernel of k_matvec_symw.hip), or 1 (the one-block-row kernel of
// k_matvec_sym.hip: small matrices, where super rows leave too few work items and too much of the matrix in the masked
// diagonal super blocks).  Tune::sym_r (DAV_SYM_R at dav_create) = 1 | 2 | 4 forces a schedule (4 only if k <= 16).
int sym_schedule(const E* e, int kk, bool stored_fp64) {
  const int forced = e->tune.sym_r;                     // DAV_SYM_R at dav_create
  const int nb = (int)(e->ncols_pad / SYM_TB);          // block rows of the whole matrix
  // k <= 8, crossover measured end to end on one box (ms for R = 1 | 2 | 4): N=40000 (157 block rows) 1.33 | 1.38 | 1.39; N=60000
  // (235) 2.78 | 2.78 | 2.67; N=100000 7.28 | 7.39 | 6.84; N=140000 14.98 | 14.85 | 13.99.
  // More than 8 columns (R = 1 | 2, k = 16 / 32 / 64, profiles/experiments/r03_small_n_schedule.log): N=20000 (79 block rows)
  // 0.455 | 0.427, 0.665 | 0.647, 1.26 | 1.12 ms; N=40000 1.51 | 1.50, 2.31 | 2.05, 4.58 | 3.67; N=60000 3.04 | 2.79, 5.09 | 4.18,
  // 10.1 | 7.54.
  // (k <= 8 between 64 and 200 block rows, stored fp64 tiles: the wide kernel is 2-4 % ahead of the one-block-row kernel end to end:
  // N=20000 0.431 | 0.415, N=30000 0.863 | 0.847, N=50000 2.06 | 2.03 ms, profiles/experiments/r03_small_n_schedule.log)
  int R = nb >= 200 ? (kk <= 8 ? 4 : 2) : (nb >= 64 && stored_fp64 ? 2 : 1);   // generated / fp32 tiles: the two-wave kernels from 200 on, as before
  // 9-16 columns of stored fp64 tiles: the wide kernel on FOUR block rows per workgroup (half as many transposed partials written
  // by a sweep that is HBM-bound there); Tune::sym_tall = 0: two (A/B runs)
  const bool tall = stored_fp64 && kk > 8 && kk <= 16 && nb >= 200 && sym_wide_enabled(e) && e->tune.sym_tall != 0;
  if (R == 2 && tall) R = 4;
  if (forced == 1 || forced == 2 || forced == 4) R = forced;
  if (R == 4 && kk > 8 && !(stored_fp64 && kk <= 16 && sym_wide_enabled(e))) R = 2;
  return R;
}

// Owners of the groups of 4 block rows (what every schedule's super rows nest in): longest group first, each to the rank
// that holds the fewest tiles so far (ties: lowest rank) - every rank computes the same table.  Cyclic or boustrophedon
// dealing leaves the ranks 4-8 % apart at N=200000 on 8 ranks (the last, incomplete round hands out the longest block
// rows); this stays within 0.5 %, and the sweep time of the slowest rank is what every rank waits for.
std::vector<int> sym_group_owners(int nb, int nranks) {
  const int ng = (nb + 3) / 4;
  std::vector<int> owner(ng, 0);
  std::vector<int64_t> load(nranks, 0);
  for (int q = ng - 1; q >= 0; --q) {
    int64_t tiles = 0;
    for (int I = 4 * q; I < std::min(nb, 4 * q + 4); ++I) tiles += I + 1;
    int best = 0;
    for (int r = 1; r < nranks; ++r)
      if (load[r] < load[best]) best = r;
    owner[q] = best;
    load[best] += tiles;
  }
  return owner;
}

void sym_set_release(SymSet& s) {
  pool_free(s.row_off); pool_free(s.items); pool_free(s.row_begin);
  for (SymPlan& pl : s.plan) { pool_free(pl.items); pool_free(pl.row_begin); pool_free(pl.zslot_begin); pool_free(pl.next_owned); }
  s = SymSet();
}

int sym_setup(E* e) {
  // work lists of the symmetric sweep over the block rows THIS rank stores
  if (e->sym.built) return 0;
  const int nb = (int)(e->ncols_pad / SYM_TB);
  e->sym_nb = nb;
  return sym_build_set(e, 0, nb, e->sym);
}

// Work lists over the block rows [first, end) that this rank owns (first and end multiples of 4, or end = nb): storage offsets
// of the set, the runs of the one-block-row kernel, the items of the two super-row schedules.
int sym_build_set(E* e, int first, int end, SymSet& out) {
  const int nb = (int)(e->ncols_pad / SYM_TB);
  sym_set_release(out);
  out.row_off_h.assign(nb, -1);
  int64_t ntiles = 0;
  const std::vector<int> gowner = sym_group_owners(nb, e->nranks);
  for (int I = first; I < std::min(end, nb); ++I)
    if (gowner[I / 4] == e->rank) { out.row_off_h[I] = ntiles; ntiles += I + 1; }
  out.ntiles = ntiles;
  HIPCHK(pool_malloc(&out.row_off, sizeof(int64_t) * nb));
  HIPCHK(hipMemcpy(out.row_off, out.row_off_h.data(), sizeof(int64_t) * nb, hipMemcpyHostToDevice));
  auto owned = [&](int I) { return out.row_off_h[I] >= 0; };
  // One-block-row kernel: runs of <= C consecutive tiles of one block row.
  // Run length: ~12 rounds of the 256 resident workgroups, between 4 tiles (a workgroup costs ~7 us to start
  // and drain) and 32 (the tail of the sweep is at most one run long).  Slab slots stay in block-row order
  // (the reduction kernel walks them per block row); the dispatch order is longest run first, so the
  // short remainder runs of every block row fill the tail (same box, N=60000: 2.95-3.04 ms against 3.16-3.37 ms
  // in block-row order for run lengths 6..24; N=200000: flat within 1 % for 16..64).
  int64_t C = std::min<int64_t>(32, std::max<int64_t>(4, (ntiles + 3071) / 3072));
  if (e->tune.sym_run > 0) C = e->tune.sym_run;
  struct Item { int I, J0, J1, slot; };
  std::vector<Item> list;
  std::vector<int> row_begin(nb + 1, 0);
  for (int I = 0; I < nb; ++I) {
    row_begin[I] = (int)list.size();
    if (!owned(I)) continue;
    for (int J0 = 0; J0 <= I; J0 += (int)C)
      list.push_back({I, J0, (int)std::min<int64_t>(I + 1, J0 + C), (int)list.size()});
  }
  row_begin[nb] = (int)list.size();
  std::stable_sort(list.begin(), list.end(), [](const Item& a, const Item& b) { return a.J1 - a.J0 > b.J1 - b.J0; });
  std::vector<int> items;
  items.reserve(list.size() * 4 + 4);
  for (const Item& it : list) { items.push_back(it.I); items.push_back(it.J0); items.push_back(it.J1); items.push_back(it.slot); }
  items.resize(std::max<size_t>(items.size(), 4), 0);
  out.nitems = row_begin[nb];
  HIPCHK(pool_malloc(&out.items, sizeof(int) * items.size()));
  HIPCHK(pool_malloc(&out.row_begin, sizeof(int) * row_begin.size()));
  HIPCHK(hipMemcpy(out.items, items.data(), sizeof(int) * items.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(out.row_begin, row_begin.data(), sizeof(int) * row_begin.size(), hipMemcpyHostToDevice));
  // Super-row schedules: items = (super row of R block rows) x (run of C tile columns).  Per tile the schedule
  // writes 1/R of a transposed and 1/C of a direct partial; C is bounded by the tail of the sweep (an item is
  // R*C tiles long) and below by the number of items that keeps 256 workgroups busy.
  for (int p = 0; p < 2; ++p) {
    SymPlan& pl = out.plan[p];
    pl.R = p == 0 ? 2 : 4;
    pl.nsuper = (nb + pl.R - 1) / pl.R;
    int64_t Cp = std::min<int64_t>(64 / pl.R, std::max<int64_t>(1, (ntiles + 3071) / (3072 * pl.R)));
    if (e->tune.sym_run9 > 0) Cp = e->tune.sym_run9;
    std::vector<Item> plist;
    std::vector<int> prow(pl.nsuper + 1, 0), zbeg(pl.nsuper + 1, 0);
    for (int S = 0; S < pl.nsuper; ++S) {
      prow[S] = (int)plist.size();
      zbeg[S + 1] = zbeg[S];
      if (!owned(S * pl.R)) continue;                // a super row nests in a group of 4 block rows: one owner
      const int Imax = std::min(S * pl.R + pl.R - 1, nb - 1);
      for (int J0 = 0; J0 <= Imax; J0 += (int)Cp)
        plist.push_back({S, J0, (int)std::min<int64_t>(Imax + 1, J0 + Cp), (int)plist.size()});
      zbeg[S + 1] = zbeg[S] + Imax;                // tile columns J < Imax receive a transposed partial
    }
    prow[pl.nsuper] = (int)plist.size();
    std::stable_sort(plist.begin(), plist.end(), [](const Item& a, const Item& b) { return a.J1 - a.J0 > b.J1 - b.J0; });
    std::vector<int> pitems;
    pitems.reserve(plist.size() * 4 + 4);
    for (const Item& it : plist) { pitems.push_back(it.I); pitems.push_back(it.J0); pitems.push_back(it.J1); pitems.push_back(it.slot); }
    pitems.resize(std::max<size_t>(pitems.size(), 4), 0);
    pl.nitems = prow[pl.nsuper];
    pl.zslots = zbeg[pl.nsuper];
    HIPCHK(pool_malloc(&pl.items, sizeof(int) * pitems.size()));
    HIPCHK(pool_malloc(&pl.row_begin, sizeof(int) * prow.size()));
    HIPCHK(pool_malloc(&pl.zslot_begin, sizeof(int) * zbeg.size()));
    HIPCHK(hipMemcpy(pl.items, pitems.data(), sizeof(int) * pitems.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(pl.row_begin, prow.data(), sizeof(int) * prow.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(pl.zslot_begin, zbeg.data(), sizeof(int) * zbeg.size(), hipMemcpyHostToDevice));
    // the super rows of this set, as a skip list for the fixed-order reduction (several ranks: 7 of 8 super rows are another rank's)
    std::vector<int> nxt(pl.nsuper + 1, pl.nsuper);
    for (int S = pl.nsuper - 1; S >= 0; --S) nxt[S] = owned(S * pl.R) ? S : nxt[S + 1];
    HIPCHK(pool_malloc(&pl.next_owned, sizeof(int) * nxt.size()));
    HIPCHK(hipMemcpy(pl.next_owned, nxt.data(), sizeof(int) * nxt.size(), hipMemcpyHostToDevice));
  }
  out.built = true;
  return 0;
}

// diagonal of a stored symmetric-tiled operator -> o.diag (this rank's rows); with several ranks the diagonal tiles
// live where their block rows do: every rank contributes its pieces, one all-reduce of n doubles at set-up
int sym_diag(E* e, OpDesc& o) {
  if (e->nranks == 1) {
    launch_diag_sym(e->stream, o.a, e->sym.row_off, e->n, e->nloc_pad, o.diag);
    return 0;
  }
  if (!(e->comm || e->lg || e->shm)) return fail("multi-rank engine used before dav_comm_init");
  launch_diag_sym(e->stream, o.a, e->sym.row_off, e->n, e->ncols_pad, e->gather_dev);
  CHK(coll_allreduce(e, e->gather_dev, (size_t)e->ncols_pad));
  HIPCHK(hipMemcpyAsync(o.diag, e->gather_dev + e->row0, sizeof(double) * (size_t)e->nslab, hipMemcpyDeviceToDevice, e->stream));
  return 0;
}

// Slabs of the symmetric sweep - per launch [column groups x direct partials][column groups x transposed partials] -
// grown on demand to what the schedule and the number of column groups of a launch need: one transposed partial per
// TILE for the one-block-row kernel (N=200000, 32 columns: 20 GB), per (super row, tile column) for the super-row
// schedules (5-10 GB); N=10^6 matrix-free, 16 columns, R=2: 125 GB.
int sym_ensure_slabs(E* e, size_t doubles) {
  if (doubles <= e->sym_slab_doubles) return 0;
  HIPCHK(hipStreamSynchronize(e->stream));
  if (e->sym_slab) HIPCHK(pool_free(e->sym_slab));
  e->sym_slab = nullptr;
  e->sym_slab_doubles = 0;
  hipError_t r = pool_malloc(&e->sym_slab, sizeof(double) * doubles);
  if (r != hipSuccess) {
    (void)hipGetLastError();
    e->sym_slab = nullptr;
    return fail("hipMalloc of the symmetric sweep slabs failed: " + std::string(hipGetErrorString(r)));
  }
  e->sym_slab_doubles = doubles;
  return 0;
}

int alloc_dense(E* e, int which) {
  OpDesc& o = e->op[which];
  sym_resident_release(o);
  o.res_decided = false;
  o.a32_valid = false;       // new contents: the fp32 copy is rebuilt when the next inner sweep asks for it
  o.a32_refused = false;
  if (o.a && o.storage != e->storage) { pool_free(o.a); o.a = nullptr; }
  o.storage = e->storage;
  if (!o.a) {
    size_t bytes;
    if (o.storage == 1) {
      CHK(sym_setup(e));
      bytes = sizeof(double) * (size_t)std::max<int64_t>(e->sym.ntiles, 1) * SYM_TB * SYM_TB;
    } else {
      bytes = sizeof(double) * (size_t)e->nloc_pad * (size_t)e->ncols_pad;
    }
    hipError_t r = pool_malloc(&o.a, bytes);
    if (r != hipSuccess) {
      (void)hipGetLastError();
      return fail("hipMalloc of the dense matrix (" + std::to_string(bytes >> 20) + " MiB) failed: " + hipGetErrorString(r));
    }
  }
  if (o.storage == 1) CHK(sym_setup(e));
  return 0;
}

// Mixed-precision correction path.  bits = 32: the block sweeps INSIDE the GJD correction solve
// (src/davidson.f90:700-734: the solve only has to produce a good correction vector) read an fp32 copy of the stored
// symmetric tiles - half the bytes per inner sweep; entries are widened to fp64 in registers, every product and sum
// stays fp64.  Everything the answer is made of - the A*V sweep of the expansion, projections, residuals, the
// convergence test - keeps reading the fp64 matrix.  bits = 64 (default): the reference's precision throughout.
// Operators that are not stored symmetric tiles (row slabs, generated operators) are not affected.
extern "C" int dav_set_inner_precision(dav_handle_t e, int bits) {
  if (bits != 32 && bits != 64) return fail("dav_set_inner_precision: 32 or 64");
  if (bits == 32 && e->inner_bits != 32) {
    // the fp32 copy of A (made at the first inner sweep) needs room: a generated operator that is kept partly resident is split
    // again at its next sweep, with that copy in the reserve
    CHK(bind(e));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (OpDesc& o : e->op)
      if (o.res_decided && !e->op[DAV_OP_A].a32) { sym_resident_release(o); o.res_decided = false; }
  }
  e->inner_bits = bits;
  return 0;
}

extern "C" int dav_set_storage(dav_handle_t e, int mode) {
  if (mode != 0 && mode != 1) return fail("dav_set_storage: mode must be 0 (full) or 1 (symmetric-tiled)");
  e->storage = mode;
  return 0;
}

int set_dense_from(E* e, int which, const double* a, int64_t lda, hipMemcpyKind kind) {
  if (which < 0 || which > 1 || !a || lda < e->n) return fail("dav_set_dense: bad arguments");
  CHK(bind(e));
  CHK(alloc_dense(e, which));
  OpDesc& o = e->op[which];
  o.kind = DAV_KIND_DENSE;
  if (o.storage == 1) {
    // lower block triangle, tile by tile (edge tiles zero padded)
    // block column by block column: one long-row 2-D copy of the rows from the diagonal block down into a staging panel
    // (two panels alternate, so the copy of column J + 1 is queued behind the cut of column J), then cut into tiles
    const int nb = e->sym_nb;
    const int64_t ldp_stage = (int64_t)nb * SYM_TB;
    // G block columns per copy (round 5): a 2-D copy from pageable host memory carries a fixed cost (the runtime pins the pages it
    // touches), 79 copies of one block column each ran at 48 GB/s where one copy of the whole matrix reaches 53; four block columns
    // per copy (the part of the group above its diagonal blocks - 6 tiles in 4 block columns - crosses the link for nothing)
    const int G = (size_t)ldp_stage * SYM_TB * 4 * sizeof(double) <= ((size_t)512 << 20) ? 4 : 1;
    double* stage[2] = {nullptr, nullptr};
    for (int b = 0; b < 2; ++b) {
      hipError_t r = pool_malloc(&stage[b], sizeof(double) * (size_t)ldp_stage * SYM_TB * G);
      if (r != hipSuccess) {
        (void)hipGetLastError();
        if (stage[0]) pool_free(stage[0]);
        return fail("hipMalloc of the upload staging panel failed: " + std::string(hipGetErrorString(r)));
      }
    }
    int rc = 0;
    for (int J0 = 0, grp = 0; J0 < nb && rc == 0; J0 += G, ++grp) {
      double* st = stage[grp & 1];
      const int64_t r0 = (int64_t)J0 * SYM_TB, nr = e->n - r0;
      const int64_t ncg = std::min<int64_t>((int64_t)G * SYM_TB, e->n - r0);       // columns of the group inside the matrix
      if (nr > 0 && hipMemcpy2DAsync(st, sizeof(double) * ldp_stage, a + r0 + r0 * lda, sizeof(double) * lda, sizeof(double) * nr,
                                     (size_t)ncg, kind, e->stream) != hipSuccess) {
        (void)hipGetLastError();
        rc = fail("dav_set_dense: copy of a group of block columns failed");
        break;
      }
      for (int J = J0; J < std::min(nb, J0 + G); ++J) {
        const int64_t d = (int64_t)(J - J0) * SYM_TB;                            // the block column starts d rows down and d columns in
        const int64_t nrj = e->n - (int64_t)J * SYM_TB;
        const int ncj = (int)std::max<int64_t>(0, std::min<int64_t>(SYM_TB, nrj));
        // (block rows / columns wholly in the padding: zero tiles)
        launch_retile_panel(e->stream, st + d + d * ldp_stage, ldp_stage, std::max<int64_t>(nrj, 0), ncj, J, nb, e->sym.row_off, o.a);
      }
    }
    hipStreamSynchronize(e->stream);
    pool_free(stage[0]);
    pool_free(stage[1]);
    if (rc != 0) return rc;
    HIPCHK(hipGetLastError());
    CHK(sym_diag(e, o));
    CHK(refresh_diag_host(e, which));
    return 0;
  }
  HIPCHK(hipMemsetAsync(o.a, 0, sizeof(double) * (size_t)e->nloc_pad * (size_t)e->ncols_pad, e->stream));
  if (e->nloc > 0)
    HIPCHK(hipMemcpy2DAsync(o.a, sizeof(double) * e->nloc_pad, a + e->row0, sizeof(double) * lda,
                            sizeof(double) * e->nloc, (size_t)e->n, kind, e->stream));
  launch_diag_dense(e->stream, o.a, e->nloc_pad, e->row0, e->nloc, o.diag);
  CHK(refresh_diag_host(e, which));
  return 0;
}

extern "C" int dav_set_dense_host(dav_handle_t e, int which, const double* a, int64_t lda) {
  return set_dense_from(e, which, a, lda, hipMemcpyHostToDevice);
}

extern "C" int dav_set_dense_dev(dav_handle_t e, int which, const double* a_dev, int64_t lda) {
  return set_dense_from(e, which, a_dev, lda, hipMemcpyDeviceToDevice);
}

// ---- streaming ingest: rows arrive in the reference's on-disk order (row-major) -----------------------------
void ingest_release(E* e) {
  for (int b = 0; b < 2; ++b) {
    if (e->ing_done[b]) { hipEventSynchronize(e->ing_done[b]); hipEventDestroy(e->ing_done[b]); e->ing_done[b] = nullptr; }
    if (e->ing_host[b]) { pool_host_free(e->ing_host[b]); e->ing_host[b] = nullptr; }
    if (e->ing_dev[b]) { pool_free(e->ing_dev[b]); e->ing_dev[b] = nullptr; }
    e->ing_pending[b] = false;
  }
  e->ing_which = -1;
}

extern "C" int dav_dense_begin(dav_handle_t e, int which) {
  if (which < 0 || which > 1) return fail("dav_dense_begin: bad operator id");
  if (e->ing_which >= 0) return fail("dav_dense_begin: another streaming upload is open (call dav_dense_end)");
  CHK(bind(e));
  CHK(alloc_dense(e, which));
  OpDesc& o = e->op[which];
  o.kind = DAV_KIND_DENSE;
  size_t bytes = o.storage == 1 ? sizeof(double) * (size_t)e->sym.ntiles * SYM_TB * SYM_TB
                                : sizeof(double) * (size_t)e->nloc_pad * (size_t)e->ncols_pad;
  HIPCHK(hipMemsetAsync(o.a, 0, bytes, e->stream));
  // ~128 MiB per staging buffer, whole rows, at least 32 of them
  int64_t cap = std::max<int64_t>(32, ((int64_t)128 << 20) / (8 * e->n) / 32 * 32);
  cap = std::min<int64_t>(cap, roundup(e->n, 32));
  e->ing_cap_rows = cap;
  for (int b = 0; b < 2; ++b) {
    HIPCHK(pool_host_malloc(&e->ing_host[b], sizeof(double) * (size_t)(cap * e->n), hipHostMallocDefault));
    HIPCHK(pool_malloc(&e->ing_dev[b], sizeof(double) * (size_t)(cap * e->n)));
    HIPCHK(hipEventCreateWithFlags(&e->ing_done[b], hipEventDisableTiming));
  }
  e->ing_flip = 0;
  e->ing_which = which;
  return 0;
}

int ingest_acquire(E* e, double** buf, int64_t* cap_rows) {
  if (e->ing_which < 0) return fail("streaming upload is not open (call dav_dense_begin)");
  int b = e->ing_flip;
  if (e->ing_pending[b]) { HIPCHK(hipEventSynchronize(e->ing_done[b])); e->ing_pending[b] = false; }
  *buf = e->ing_host[b];
  *cap_rows = e->ing_cap_rows;
  return 0;
}

int ingest_commit(E* e, int64_t row0, int64_t nrows) {
  if (e->ing_which < 0) return fail("streaming upload is not open (call dav_dense_begin)");
  if (row0 < 0 || nrows < 0 || row0 + nrows > e->n || nrows > e->ing_cap_rows) return fail("dav_dense_put_rows: rows out of range");
  if (nrows == 0) return 0;
  CHK(bind(e));
  OpDesc& o = e->op[e->ing_which];
  int b = e->ing_flip;
  HIPCHK(hipMemcpyAsync(e->ing_dev[b], e->ing_host[b], sizeof(double) * (size_t)(nrows * e->n), hipMemcpyHostToDevice, e->stream));
  launch_rows_scatter(e->stream, e->ing_dev[b], e->n, row0, nrows, e->n, o.a, e->nloc_pad, e->row0, e->nloc, o.storage == 1, e->sym.row_off);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(e->ing_done[b], e->stream));
  e->ing_pending[b] = true;
  e->ing_flip ^= 1;
  return 0;
}

void ingest_wanted(E* e, int64_t* first, int64_t* count) {
  if (e->op[e->ing_which].storage == 1) { *first = 0; *count = e->n; }
  else { *first = e->row0; *count = e->nloc; }
}

extern "C" int dav_dense_put_rows(dav_handle_t e, int which, int64_t row0, int64_t nrows, const double* rows, int64_t ldr) {
  if (e->ing_which != which) return fail("dav_dense_put_rows: no streaming upload open for this operator");
  if (!rows || ldr < e->n || row0 < 0 || nrows < 0 || row0 + nrows > e->n) return fail("dav_dense_put_rows: bad arguments");
  int64_t w0, wn;
  ingest_wanted(e, &w0, &wn);
  int64_t lo = std::max(row0, w0), hi = std::min(row0 + nrows, w0 + wn);     // rows of other ranks are ignored
  for (int64_t r = lo; r < hi;) {
    double* buf; int64_t cap;
    CHK(ingest_acquire(e, &buf, &cap));
    int64_t take = std::min(cap, hi - r);
    // staging copy, by several threads when the block is large (one memcpy stream into pinned memory runs at
    // ~4 GB/s, far below the host-to-device copy that follows)
    const size_t blk_bytes = sizeof(double) * (size_t)take * (size_t)e->n;
    const int T = (int)std::min<size_t>(8, blk_bytes / ((size_t)8 << 20) + 1);
    auto copy_rows = [&](int64_t i0, int64_t i1) {
      for (int64_t i = i0; i < i1; ++i) memcpy(buf + i * e->n, rows + (r - row0 + i) * ldr, sizeof(double) * (size_t)e->n);
    };
    if (T <= 1) {
      copy_rows(0, take);
    } else {
      std::vector<std::thread> pool;
      for (int t = 0; t < T; ++t) pool.emplace_back(copy_rows, take * t / T, take * (t + 1) / T);
      for (auto& th : pool) th.join();
    }
    CHK(ingest_commit(e, r, take));
    r += take;
  }
  return 0;
}

extern "C" int dav_dense_end(dav_handle_t e, int which) {
  if (e->ing_which != which) return fail("dav_dense_end: no streaming upload open for this operator");
  CHK(bind(e));
  OpDesc& o = e->op[which];
  int rc = 0;
  if (o.storage == 1) rc = sym_diag(e, o);
  else launch_diag_dense(e->stream, o.a, e->nloc_pad, e->row0, e->nloc, o.diag);
  if (rc == 0) rc = refresh_diag_host(e, which);
  ingest_release(e);
  return rc;
}

namespace {
struct EngineSink : IngestSink {
  E* e;
  explicit EngineSink(E* e_) : e(e_) {}
  int acquire(double** buf, int64_t* cap_rows) override { return ingest_acquire(e, buf, cap_rows); }
  int commit(int64_t row0, int64_t nrows) override { return ingest_commit(e, row0, nrows); }
  void wanted(int64_t* first, int64_t* count) override { ingest_wanted(e, first, count); }
};
}  // namespace

extern "C" int dav_set_dense_file(dav_handle_t e, int which, const char* path, int format) {
  if (!path) return fail("dav_set_dense_file: null path");
  if (format != DAV_FILE_TEXT && format != DAV_FILE_F64) return fail("dav_set_dense_file: unknown format");
  CHK(dav_dense_begin(e, which));
  EngineSink sink(e);
  std::string err;
  int rc = format == DAV_FILE_TEXT ? ingest_text_file(path, e->n, sink, &err) : ingest_f64_file(path, e->n, sink, &err);
  if (rc != 0) {
    hipStreamSynchronize(e->stream);
    ingest_release(e);
    e->op[which].kind = DAV_KIND_NONE;
    return err.empty() ? rc : fail("dav_set_dense_file: " + err);
  }
  return dav_dense_end(e, which);
}

extern "C" int dav_parse_text_f64(const char* text, size_t len, double* out, size_t max_vals, size_t* nvals) {
  std::vector<double> v;
  std::string err;
  size_t used = ingest_parse_text_parallel(text, len, true, &v, 4, &err);
  if (used == (size_t)-1) return fail("dav_parse_text_f64: " + err);
  if (nvals) *nvals = v.size();
  if (out) memcpy(out, v.data(), sizeof(double) * std::min(v.size(), max_vals));
  return 0;
}

extern "C" int dav_set_dense_generated(dav_handle_t e, int which, uint64_t seed, double sparsity, int use_diag_val,
                                       double diag_val) {
  if (which < 0 || which > 1) return fail("dav_set_dense_generated: bad operator id");
  CHK(bind(e));
  CHK(alloc_dense(e, which));
  OpDesc& o = e->op[which];
  o.kind = DAV_KIND_DENSE;
  if (o.storage == 1) {
    launch_generate_sym_tiles(e->stream, o.a, e->sym.row_off_h.data(), e->sym_nb, e->n, seed, sparsity, use_diag_val, diag_val);
    CHK(sym_diag(e, o));
  } else {
    launch_generate_dense(e->stream, o.a, e->nloc_pad, e->nloc_pad, e->ncols_pad, e->row0, e->nloc, e->n, seed, sparsity,
                          use_diag_val, diag_val);
    launch_diag_dense(e->stream, o.a, e->nloc_pad, e->row0, e->nloc, o.diag);
  }
  CHK(refresh_diag_host(e, which));
  return 0;
}

OpParams op_params(const OpDesc& o) {
  OpParams p{};
  p.kind = o.kind; p.seed = o.seed; p.sparsity = o.sparsity; p.use_diag = o.use_diag; p.diag_val = o.diag_val;
  p.trig = o.trig; p.e_table = o.e_table; p.l2_table = o.l2_table; p.dadd_table = o.dadd_table; p.libm = o.harness_libm;
  return p;
}

extern "C" int dav_set_operator_hashed(dav_handle_t e, int which, uint64_t seed, double sparsity, int use_diag_val,
                                       double diag_val) {
  if (which < 0 || which > 1) return fail("dav_set_operator_hashed: bad operator id");
  CHK(bind(e));
  OpDesc& o = e->op[which];
  HIPCHK(hipStreamSynchronize(e->stream));      // sweeps in flight may still read the resident tiles of the previous definition
  sym_resident_release(o);
  o.res_decided = false;
  o.kind = DAV_KIND_HASHED; o.seed = seed; o.sparsity = sparsity; o.use_diag = use_diag_val; o.diag_val = diag_val;
  // storage mode "symmetric" (single rank) also applies to the generated operator: every entry of the lower
  // block triangle is produced once and used for both products
  o.storage = e->storage == 1 ? 1 : 0;
  if (o.storage == 1) CHK(sym_setup(e));
  launch_diag_free(e->stream, op_params(o), e->row0, e->nloc, o.diag);
  CHK(refresh_diag_host(e, which));
  return 0;
}

extern "C" int dav_set_operator_harness(dav_handle_t e, int which, const double* e_table) {
  if (which < 0 || which > 1 || !e_table) return fail("dav_set_operator_harness: bad arguments");
  CHK(bind(e));
  OpDesc& o = e->op[which];
  HIPCHK(hipStreamSynchronize(e->stream));
  sym_resident_release(o);
  o.res_decided = false;
  o.kind = DAV_KIND_HARNESS; o.trig = which == DAV_OP_A ? 0 : 1;
  o.storage = e->storage == 1 ? 1 : 0;      // symmetric mode: each entry generated once
  if (o.storage == 1) CHK(sym_setup(e));
  if (!o.e_table) HIPCHK(pool_malloc(&o.e_table, sizeof(double) * e->n));
  HIPCHK(hipMemcpyAsync(o.e_table, e_table, sizeof(double) * e->n, hipMemcpyHostToDevice, e->stream));
  // the one-variable form of the entries (common.h: dav_harness_poly) reads 2 log e_i; made here, once, from the table the caller
  // evaluated (host log: correctly rounded to within an ulp); padded with zeros so that the sweeps read whole tiles' worth of it
  o.harness_libm = e->tune.harness_libm != 0;
  const size_t l2n = (size_t)roundup(e->n, SYM_TB) + SYM_TB;
  std::vector<double> l2(l2n, 0.0);
  for (int64_t i = 0; i < e->n; ++i) {
    if (!(e_table[i] > 0.0)) return fail("dav_set_operator_harness: the table must hold positive numbers (exp(real(i) / real(n)))");
    l2[(size_t)i] = 2.0 * std::log(e_table[i]);
    // the one-variable form needs what exp(real(i) / real(n)) guarantees: a table that does not descend (then e_min / e_max are
    // the entries at the lower / higher index, atan2's quotient is at most 1) and spans at most a factor e; any other table is
    // served by the formula as written
    if (i > 0 && l2[(size_t)i] < l2[(size_t)i - 1]) o.harness_libm = true;
  }
  if (e->n > 0 && l2[(size_t)e->n - 1] - l2[0] > 2.0 * (1.0 + 1e-6)) o.harness_libm = true;
  if (!o.l2_table) HIPCHK(pool_malloc(&o.l2_table, sizeof(double) * l2n));
  HIPCHK(hipMemcpyAsync(o.l2_table, l2.data(), sizeof(double) * l2n, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  // the diagonal ENTRIES of operator A - poly(x = 1) * 1 + real(i) (src/tests/test_utils.f90:49), rounded as the kernels round it: the
  // Horner form at x = 1 is the sum of the coefficients, highest first, one rounding per step - as a table for the wide generating
  // kernel's diagonal tiles
  double c0 = 0.0;
  {
    static const double cf[] = {DAV_HARNESS_COS_COEFFS};
    c0 = cf[DAV_HARNESS_COS_DEGREE];
    for (int k = DAV_HARNESS_COS_DEGREE - 1; k >= 0; --k) c0 = std::fma(c0, 1.0, cf[k]);
  }
  for (size_t i = 0; i < l2n; ++i) l2[i] = (int64_t)i < e->n ? c0 + (double)(float)(i + 1) : 0.0;
  if (!o.dadd_table) HIPCHK(pool_malloc(&o.dadd_table, sizeof(double) * l2n));
  HIPCHK(hipMemcpyAsync(o.dadd_table, l2.data(), sizeof(double) * l2n, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  launch_diag_free(e->stream, op_params(o), e->row0, e->nloc, o.diag);
  CHK(refresh_diag_host(e, which));
  return 0;
}

extern "C" int dav_set_operator_identity(dav_handle_t e, int which) {
  if (which < 0 || which > 1) return fail("dav_set_operator_identity: bad operator id");
  CHK(bind(e));
  OpDesc& o = e->op[which];
  HIPCHK(hipStreamSynchronize(e->stream));
  sym_resident_release(o);
  o.res_decided = false;
  o.kind = DAV_KIND_IDENTITY;
  launch_diag_free(e->stream, op_params(o), e->row0, e->nloc, o.diag);
  CHK(refresh_diag_host(e, which));
  return 0;
}

extern "C" int dav_set_operator_host(dav_handle_t e, int which, const double* diag) {
  if (which < 0 || which > 1 || !diag) return fail("dav_set_operator_host: bad arguments");
  CHK(bind(e));
  OpDesc& o = e->op[which];
  HIPCHK(hipStreamSynchronize(e->stream));
  sym_resident_release(o);
  o.res_decided = false;
  o.kind = DAV_KIND_HOST;
  if (e->nloc > 0)
    HIPCHK(hipMemcpyAsync(o.diag, diag + e->row0, sizeof(double) * e->nloc, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->diag_host[which].assign(diag, diag + e->n);
  if (which == DAV_OP_A) e->basis_order.clear();
  return 0;
}

extern "C" int dav_set_operator_device(dav_handle_t e, int which, dav_device_apply_fn fn, void* ctx, const double* diag) {
  if (which < 0 || which > 1 || !diag || !fn) return fail("dav_set_operator_device: bad arguments");
  CHK(dav_set_operator_host(e, which, diag));          // the diagonal: device slab, host copy, start-vector order
  OpDesc& o = e->op[which];
  o.kind = DAV_KIND_DEVICE;
  o.dev_fn = fn;
  o.dev_ctx = ctx;
  return 0;
}

extern "C" int dav_get_diagonal(dav_handle_t e, int which, double* out) {
  if (which < 0 || which > 1 || e->diag_host[which].empty()) return fail("dav_get_diagonal: operator not set");
  std::memcpy(out, e->diag_host[which].data(), sizeof(double) * e->n);
  return 0;
}

// ---- a generated symmetric operator kept (partly) resident ------------------------------------------------------------------
// configs[3] (N=200000 generalized): A's lower block triangle takes 160.5 GB, B = the same generator with unit diagonal is never
// stored in full - but ~100 GB of HBM stand empty next to A, and a stored 16-column sweep costs half of a generated one (a
// quarter in the 32- / 64-column launches of the one-wave-per-SIMD kernel).  At the first sweep of such an operator the engine
// stores the tiles of as many of its LONGEST block rows (whole groups of four, from the bottom of the triangle up) as fit next to
// what the sweeps still have to allocate - the partial-sum slabs of the widest launch, the fp32 copy of A when the mixed-precision
// inner sweeps are on - and keeps generating the others.  Tune::b_resident (DAV_B_RESIDENT at dav_create): 0 = never, 1 = by the free
// memory (default; nothing below a fifth of the tiles), 2..100 = at most that percentage of the tiles (tests of the mixed path).
void sym_resident_release(OpDesc& o) {
  if (o.res) { sym_set_release(*o.res); delete o.res; o.res = nullptr; }
  if (o.gen) { sym_set_release(*o.gen); delete o.gen; o.gen = nullptr; }
  pool_free(o.res_a);
  o.res_a = nullptr;
  o.res_tiles = 0;
  o.res_first = 0;
  o.pass_res = o.pass_gen = false;
}

int sym_resident_split(E* e, int which) {
  OpDesc& o = e->op[which];
  sym_resident_release(o);
  o.res_decided = true;
  if (e->tune.b_resident == 0 || o.kind != DAV_KIND_HASHED || o.storage != 1) return 0;
  CHK(sym_setup(e));
  const int nb = e->sym_nb;
  const double tile_bytes = 8.0 * SYM_TB * SYM_TB;
  size_t free_b = 0, total_b = 0;
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  free_b += pool_idle_device_bytes(e->device);            // idle blocks of the buffer cache are given back when an allocation needs them
  // what the sweeps may still allocate: the slabs of a paired 32-column launch of the two-block-row schedule (a 64-column launch
  // that then finds no room for its four column groups runs as two paired launches - 1 % slower on that sweep, which is far less
  // than what the extra resident block rows save on every sweep of this operator), the fp32 copy
  const SymPlan& p2 = e->sym.plan[0];
  const double slabs = 2.0 * 8.0 * 16.0 * SYM_TB * ((double)p2.nitems * 2 + (double)p2.zslots);
  double reserve = std::max(0.0, slabs - 8.0 * (double)e->sym_slab_doubles) + 0.01 * (double)total_b + 2.0e9;
  const OpDesc& a = e->op[DAV_OP_A];
  if (e->inner_bits == 32 && a.kind == DAV_KIND_DENSE && a.storage == 1 && !a.a32) reserve += 4.0 * SYM_TB * SYM_TB * (double)e->sym.ntiles;
  int64_t fit = (int64_t)std::max(0.0, ((double)free_b - reserve) / tile_bytes);
  if (e->tune.b_resident > 1) fit = std::min<int64_t>(fit, e->sym.ntiles * std::min(e->tune.b_resident, 100) / 100);
  // whole groups of four block rows, from the bottom of the triangle up, while the rank's tiles of them fit
  int first = nb;
  int64_t tiles = 0;
  for (int q = (nb + 3) / 4 - 1; q >= 0; --q) {
    int64_t t = 0;
    for (int I = 4 * q; I < std::min(nb, 4 * q + 4); ++I)
      if (e->sym.row_off_h[I] >= 0) t += I + 1;
    if (tiles + t > fit) break;
    tiles += t;
    first = 4 * q;
  }
  if (tiles == 0 || (e->tune.b_resident == 1 && tiles * 5 < e->sym.ntiles)) { tiles = 0; first = nb; }      // too little to be worth a second pass
  // Several ranks: the sweeps of the parts carry collectives (all-gather of the block, reduce-scatter of the partial products), so
  // every rank must make the SAME passes whatever its own memory allowed: a pass over the resident block rows if any rank has
  // some, a pass over the generated ones if any rank has some (a rank without rows in a part sweeps an empty set)
  double flags[2] = {tiles > 0 ? 1.0 : 0.0, e->sym.ntiles - tiles > 0 ? 1.0 : 0.0};
  if (has_comm(e)) {
    HIPCHK(hipMemcpyAsync(e->gram_dev, flags, sizeof(flags), hipMemcpyHostToDevice, e->stream));
    CHK(coll_allreduce(e, e->gram_dev, 2));
    HIPCHK(hipMemcpyAsync(flags, e->gram_dev, sizeof(flags), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  if (flags[0] == 0.0) return 0;       // nothing resident anywhere: the operator stays generated (one pass over e->sym)
  o.pass_res = true;
  o.pass_gen = flags[1] > 0.0;
  o.res = new SymSet();
  o.gen = new SymSet();
  int rc = sym_build_set(e, first, nb, *o.res);
  if (rc == 0) rc = sym_build_set(e, 0, first, *o.gen);
  if (rc == 0 && pool_malloc(&o.res_a, (size_t)(tile_bytes * (double)std::max<int64_t>(o.res->ntiles, 1))) != hipSuccess) {
    (void)hipGetLastError();
    o.res_a = nullptr;
    rc = fail("hipMalloc of the resident tiles of a generated operator failed (" + std::to_string((size_t)(tile_bytes * (double)o.res->ntiles) >> 20) +
              " MiB): set DAV_B_RESIDENT=0");
  }
  if (rc != 0) {
    // the passes were agreed on across the ranks: no way back to the one-pass route from here.  The operator becomes unusable ON
    // THIS RANK - every later apply fails again, instead of sweeping one pass where the peers sweep two (a caller that catches the
    // error and goes on would otherwise walk into mismatched collectives; round-4 advisor)
    const std::string msg = g_err;
    sym_resident_release(o);
    o.kind = DAV_KIND_NONE;
    return fail(msg + " - the operator is now unset on this rank: set it again");
  }
  launch_generate_sym_tiles(e->stream, o.res_a, o.res->row_off_h.data(), nb, e->n, o.seed, o.sparsity, o.use_diag, o.diag_val);
  o.res_tiles = o.res->ntiles;
  o.res_first = first;
  // the split follows the free memory of the moment (a co-tenant or another box moves it, and with it the order of the sums of the
  // two parts): said once, so that a run can be repeated with the same split (DAV_B_RESIDENT = that percentage)
  if (getenv("DAVIDSON_VERBOSE"))
    std::fprintf(stderr, "davidson engine: rank %d keeps block rows %d.. of the generated operator %c resident (%lld of %lld tiles = %.1f %%; "
                         "DAV_B_RESIDENT=%d fixes an upper bound)\n", e->rank, first, which == DAV_OP_A ? 'A' : 'B', (long long)o.res_tiles,
                 (long long)e->sym.ntiles, 100.0 * (double)o.res_tiles / (double)std::max<int64_t>(e->sym.ntiles, 1),
                 (int)std::ceil(100.0 * (double)o.res_tiles / (double)std::max<int64_t>(e->sym.ntiles, 1)));
  HIPCHK(hipGetLastError());
  return 0;
}

// fraction of the block rows (by tiles) of operator `which` that a generated symmetric operator keeps resident as stored tiles
extern "C" int dav_device_memory(dav_handle_t e, int64_t* free_bytes, int64_t* total_bytes) {
  CHK(bind(e));
  size_t f = 0, t = 0;
  HIPCHK(hipMemGetInfo(&f, &t));
  f += pool_idle_device_bytes(e->device);                 // (the buffer cache's idle blocks are free memory to this library)
  if (free_bytes) *free_bytes = (int64_t)f;
  if (total_bytes) *total_bytes = (int64_t)t;
  return 0;
}

extern "C" int dav_resident_fraction(dav_handle_t e, int which, double* fraction) {
  if (which < 0 || which > 1 || !fraction) return fail("dav_resident_fraction: bad arguments");
  const OpDesc& o = e->op[which];
  *fraction = (o.res_tiles > 0 && e->sym.ntiles > 0) ? (double)o.res_tiles / (double)e->sym.ntiles : 0.0;
  return 0;
}
