// Engine, K1 scheduling: W = A X (or B X) for a block of columns - which kernel, how many columns per launch, packing,
// collectives around the sweep, the fixed-order reductions; the apply microbenchmark entry points.
#include "engine_internal.h"

// ---- K1 -----------------------------------------------------------------------------------------
// dst[:, 0:k] = Op(which) * src[:, 0:k] for device-resident column blocks with leading dimension ldp
// fp32 copy of a stored symmetric-tiled operator, made when the first inner sweep wants it; false (and fp64 sweeps) when
// the memory for it is not there
bool inner_f32_tiles(E* e, OpDesc& o) {
  if (e->inner_bits != 32 || o.kind != DAV_KIND_DENSE || o.storage != 1 || o.a32_refused) return false;
  if (o.a32_valid) return true;
  const size_t count = (size_t)std::max<int64_t>(e->sym.ntiles, 1) * SYM_TB * SYM_TB;
  if (!o.a32 && pool_malloc(&o.a32, sizeof(float) * count) != hipSuccess) {
    (void)hipGetLastError();
    o.a32 = nullptr;
    o.a32_refused = true;
    return false;
  }
  launch_tiles_to_f32(e->stream, o.a, o.a32, (int64_t)count);
  o.a32_valid = true;
  return true;
}

// The super-row sweep of one launch: stored fp64 tiles, two block rows per workgroup and more than 8 columns run the
// one-wave-per-SIMD kernel (k_matvec_symw.hip: 32 columns per workgroup, or 16 for a block of <= 16); generated operators,
// the fp32 copy and the k <= 8 schedule (R = 4, 4x4x4 MFMA) stay on matvec_sym9_kernel.
// Tune::sym_wide (DAV_SYM_WIDE at dav_create) = 2 (default): the one-wave-per-SIMD kernel for more than 8 columns, 1: for more
// than 16 only, 0: never (A/B runs)
bool sym_wide_enabled(const E* e) { return e->tune.sym_wide > 1; }   // ... for 9-16 columns too

void sym9_sweep(E* e, int R, const OpDesc& o, bool use32, const SymSet& set, const SymPlan* pl, const double* xt, int kk, double* slabD,
                double* slabT, int npair, int64_t dstride, int64_t tstride) {
  const int wide = e->tune.sym_wide;
  if (o.kind == DAV_KIND_DENSE && !use32 && wide > 0 && ((R == 2 && (kk > 16 || wide > 1)) || (R == 4 && kk > 8 && kk <= 16))) {
    const int nbw = kk > 16 ? 2 : 1;
    launch_matvec_symw(e->stream, nbw, R == 4, false, o.a, set.row_off, e->sym_nb, pl->items, pl->nitems, pl->zslot_begin, xt, kk, slabD, slabT,
                       (npair + nbw - 1) / nbw, e->xt_group_stride, dstride, tstride);
    return;
  }
  // the hashed operator at 17-32 columns per launch: the wide kernel's generating variant - one generated entry feeds the MFMAs of
  // both 16-column groups (Tune::sym_gen_wide = 0: two groups of the 16-column kernel, every entry generated twice)
  // (round 6: the reference's matrix-free test operator in its polynomial form too - GEN = 2 / 3 of that kernel)
  if ((o.kind == DAV_KIND_HASHED || (o.kind == DAV_KIND_HARNESS && !o.harness_libm)) && R == 2 && kk > 16 && wide > 0 && e->tune.sym_gen_wide) {
    launch_matvec_symw_generated(e->stream, op_params(o), e->n, e->sym_nb, pl->items, pl->nitems, pl->zslot_begin, xt, kk, slabD, slabT,
                                 (npair + 1) / 2, e->xt_group_stride, dstride, tstride);
    return;
  }
  // fp32 tiles (mixed-precision inner sweeps, up to 16 columns): the wide kernel's fp32 variant; Tune::sym_wide32 = 0: the two-wave kernel
  if (o.kind == DAV_KIND_DENSE && use32 && wide > 1 && e->tune.sym_wide32 && R == 2 && kk <= 16) {
    launch_matvec_symw(e->stream, 1, false, true, o.a32, set.row_off, e->sym_nb, pl->items, pl->nitems, pl->zslot_begin, xt, kk, slabD, slabT,
                       npair, e->xt_group_stride, dstride, tstride);
    return;
  }
  launch_matvec_sym9(e->stream, R, o.kind != DAV_KIND_DENSE, use32 ? (const void*)o.a32 : (const void*)o.a, use32, set.row_off,
                     o.kind != DAV_KIND_DENSE ? op_params(o) : OpParams{}, e->n, e->sym_nb, pl->items, pl->nitems, pl->zslot_begin, xt, kk,
                     slabD, slabT, npair, e->xt_group_stride, dstride, tstride, e->tune.sym_mfma4 != 0);
}


// Symmetric sweep of k > 32 columns over several ranks with RCCL, chunks of 32 columns software-pipelined over two streams:
//   comm stream:  gather(0)            gather(1)   scatter(0)   gather(2)   scatter(1) ...
//   main stream:  pack(0) pack(1) | wait gather(0) sweep(0) reduce(0) | pack(2) wait gather(1) sweep(1) reduce(1) | to_panel(0) ...
// i.e. the all-gather of chunk i + 1 and the reduce-scatter of chunk i - 1 run under the sweep of chunk i.  Xt column groups,
// the partial-product buffer and the receive buffer alternate with the chunk parity.  Same kernels, same sums, same result as
// the serial path (which the test transports and single-chunk applies keep using).
int apply_sym_overlapped(E* e, int which, OpDesc& o, const double* src, int k, double* dst, bool timed, bool inner) {
  const int step = 32;
  const int nchunks = (k + step - 1) / step;
  if (!e->ov_ready) {
    // everything into locals first: a failure half-way must not leave a stream without its events or buffers behind
    // (later calls would skip this block and launch on null handles); committed to the engine only when complete
    hipStream_t cs = nullptr;
    hipEvent_t evs[8] = {};
    double* bufs[4] = {};
    auto undo = [&]() {
      for (hipEvent_t v : evs) if (v) (void)hipEventDestroy(v);
      for (double* b : bufs) if (b) (void)pool_free(b);
      if (cs) (void)hipStreamDestroy(cs);
    };
    // highest priority: the collective's few workgroups take the first CUs the sweep's work items give back, instead of queueing
    // behind the thousands of workgroups of the sweep that is already launched
    int prio_least = 0, prio_greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) { (void)hipGetLastError(); prio_greatest = 0; }
    bool ok = hipStreamCreateWithPriority(&cs, hipStreamNonBlocking, prio_greatest) == hipSuccess;
    for (int i = 0; ok && i < 8; ++i) ok = hipEventCreateWithFlags(&evs[i], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < 2; ++i) {
      ok = pool_malloc(&bufs[i], sizeof(double) * (size_t)e->nranks * (size_t)e->nslab * 32) == hipSuccess &&
           pool_malloc(&bufs[2 + i], sizeof(double) * (size_t)e->nslab * 32) == hipSuccess;
    }
    if (!ok) {
      (void)hipGetLastError();
      undo();
      return 2;                                        // the caller runs the serial path
    }
    e->comm_stream = cs;
    for (int i = 0; i < 2; ++i) {
      e->ov_packed[i] = evs[4 * i]; e->ov_gathered[i] = evs[4 * i + 1]; e->ov_reduced[i] = evs[4 * i + 2]; e->ov_scattered[i] = evs[4 * i + 3];
      e->sym_wpart2[i] = bufs[i]; e->sym_wrecv2[i] = bufs[2 + i];
    }
    e->ov_ready = true;
  }
  const int64_t total_rows = (int64_t)e->nranks * e->nslab;
  // Test transports (loopback threads / shared-memory processes on one GPU, DAV_SYM_OVERLAP=1): the same pipeline - buffers, chunk
  // parities, layouts, counts, with SEVERAL ranks - with its collectives executed by the transport on the engine's own stream
  // (they synchronise the host); what the multi-rank RCCL run cannot be rehearsed for on a one-GPU box is only the overlap itself
  const bool tt = has_test_transport(e);
  hipStream_t cs = tt ? e->stream : e->comm_stream;
  const bool use32 = false;                            // chunks of 32 columns: the fp64 tiles on the wide kernel (see apply_ptr)
  const int R = 2;                                     // 32-column chunks: the paired two-block-row schedule
  const SymPlan* pl = &e->sym.plan[0];
  const int64_t dstride = (int64_t)pl->nitems * R * 16 * SYM_TB, tstride = pl->zslots * 16 * SYM_TB;
  if (sym_ensure_slabs(e, (size_t)2 * (size_t)(dstride + tstride) + 1) != 0) return 2;   // serial path: it degrades 4 -> 2 -> 1 column groups
  int slot = -1;
  const double stored = o.kind == DAV_KIND_DENSE ? (use32 ? 4.0 : 8.0) * 0.5 * (double)e->n * ((double)e->n + 1.0) / e->nranks : 0.0;
  if (timed) CHK(timed_begin(e, which == DAV_OP_A ? 0 : 2, stored * nchunks + 16.0 * (double)e->n * k, &slot));
  auto cols = [&](int i) { return std::min(step, k - i * step); };
  auto xt_of = [&](int i) { return e->xt + (size_t)(i & 1) * 2 * e->xt_group_stride; };
  auto pack_and_gather = [&](int i) -> int {
    const int p = i & 1, kk = cols(i), ng = (kk + 15) / 16;
    launch_pack_xt(e->stream, src + (int64_t)i * step * e->ldp, e->ldp, e->nloc, e->nslab, kk, xt_of(i), e->xt_group_stride, e->row0);
    HIPCHK(hipEventRecord(e->ov_packed[p], e->stream));
    HIPCHK(hipStreamWaitEvent(cs, e->ov_packed[p], 0));
    CollGroup grp(e);
    CHK(grp.begin(5, 8.0 * (double)e->nslab * 16 * ng * e->nranks, cs));
    for (int g = 0; g < ng; ++g) {
      double* base = xt_of(i) + (size_t)g * e->xt_group_stride;
      if (tt) CHK(test_allgather(e, base + e->row0 * 16, base, (size_t)e->nslab * 16));
      else NCCLCHK(g_rccl.AllGather(base + e->row0 * 16, base, (size_t)e->nslab * 16, ncclDouble, e->comm, cs));
    }
    CHK(grp.end("all-gather of a column chunk (second stream)", cs));
    HIPCHK(hipEventRecord(e->ov_gathered[p], cs));
    return 0;
  };
  auto to_panel = [&](int i) -> int {
    const int p = i & 1, kk = cols(i), ng = (kk + 15) / 16;
    HIPCHK(hipStreamWaitEvent(e->stream, e->ov_scattered[p], 0));
    for (int g = 0; g < ng; ++g)
      launch_chunk_to_panel(e->stream, e->sym_wrecv2[p] + (size_t)g * (size_t)e->nslab * 16, e->nslab, e->nloc, e->nloc_pad,
                            std::min(16, kk - 16 * g), dst + (int64_t)(i * step + 16 * g) * e->ldp, e->ldp);
    return 0;
  };
  CHK(pack_and_gather(0));
  for (int i = 0; i < nchunks; ++i) {
    const int p = i & 1, kk = cols(i), npair = (kk + 15) / 16;
    if (i + 1 < nchunks) CHK(pack_and_gather(i + 1));         // Xt groups of the other parity: last read by the sweep of chunk i - 1
    HIPCHK(hipStreamWaitEvent(e->stream, e->ov_gathered[p], 0));
    int kslot = -1;
    if (timed && which == DAV_OP_A) CHK(timed_begin(e, 4, 2.0 * (double)e->n * (double)e->n * kk / e->nranks, &kslot));
    double* slabT = e->sym_slab + (int64_t)npair * dstride;
    if (pl->nitems > 0)
      sym9_sweep(e, R, o, use32, e->sym, pl, xt_of(i), kk, e->sym_slab, slabT, npair, dstride, tstride);
    CHK(timed_end(e, kslot));
    // partial of the whole product of this chunk (the buffer of this parity was last read by the reduce-scatter of chunk
    // i - 2, whose completion the main stream waited for when it finished chunk i - 2 below)
    for (int g = 0; g < npair; ++g)
      launch_sym9_reduce(e->stream, e->sym_slab + g * dstride, slabT + g * tstride, pl->row_begin, pl->zslot_begin, e->sym.row_off, pl->next_owned, R,
                         e->sym_nb, e->nloc, std::min(16, kk - 16 * g), e->sym_wpart2[p] + (size_t)g * (size_t)total_rows * 16, e->ldp,
                         e->nslab, total_rows);
    HIPCHK(hipEventRecord(e->ov_reduced[p], e->stream));
    HIPCHK(hipStreamWaitEvent(cs, e->ov_reduced[p], 0));
    {
      CollGroup grp(e);
      CHK(grp.begin(6, 8.0 * (double)e->nslab * kk * e->nranks, cs));
      for (int g = 0; g < npair; ++g) {
        const double* send = e->sym_wpart2[p] + (size_t)g * (size_t)total_rows * 16;
        double* recv = e->sym_wrecv2[p] + (size_t)g * (size_t)e->nslab * 16;
        const size_t count = (size_t)e->nslab * std::min(16, kk - 16 * g);
        if (tt) CHK(test_reduce_scatter(e, send, recv, count));
        else NCCLCHK(g_rccl.ReduceScatter(send, recv, count, ncclDouble, ncclSum, e->comm, cs));
      }
      CHK(grp.end("reduce-scatter of a column chunk (second stream)", cs));
    }
    HIPCHK(hipEventRecord(e->ov_scattered[p], cs));
    if (i >= 1) CHK(to_panel(i - 1));                      // the previous chunk's rows of W, while this chunk's reduce-scatter runs
    if (which == DAV_OP_A) { e->st.applies += 1; e->st.apply_cols += kk; }
  }
  CHK(to_panel(nchunks - 1));
  CHK(timed_end(e, slot));
  HIPCHK(hipGetLastError());
  return 0;
}

static int apply_sym_set(E* e, int which, OpDesc& o, const SymSet& set, bool partial, bool accumulate, const double* src, int k, double* dst,
                         bool timed, bool inner);

// ---- which way do the collectives of a wide block go?  (round 6) ---------------------------------------------------------------
// Three ways exist (all three give the sweep's kernels the same operands and sum their partial products in a fixed order):
//   program order   one all-gather, one 64-column launch, one reduce-scatter, RCCL's collectives on the engine's stream
//   direct          the same program order with the all-gather / reduce-scatter as grouped ncclSend / ncclRecv to every peer and a
//                   rank-order sum: the P - 1 point-to-point links of the xGMI mesh at once, whatever schedule RCCL would pick
//   second stream   32-column chunks, the all-gather of chunk i + 1 and the reduce-scatter of chunk i - 1 under the sweep of chunk i
// Which is fastest depends on the links and on RCCL's schedules - nothing a one-GPU box can measure, and an environment knob is
// not something a launcher sets.  So the first wide block of an engine with a real communicator of several ranks goes through all
// three: program order first (its result is the one that is kept), then the other two into scratch blocks, each twice (the first
// run pays the lazy set-up of connections / streams), timed with HIP events on the engine's stream, compared with the program-order
// result - the second stream bitwise (it changes the order of launches, not a single sum), the direct exchange bitwise on two ranks
// and to 1e-12 of the block's largest entry beyond (RCCL's ring sums in ring order, the direct exchange in rank order).  ONE small
// all-reduce then makes the figures common (maximum over the ranks of each time, any rank's failed comparison), and every rank
// picks the fastest validated way.  A way that fails its comparison or returns an error is left out with a message; program order
// is always valid.  A rank that hangs inside a trial ends like any hung collective: the watchdog's exit 124 (no re-exec, nothing is
// retried).  DAV_COLL_SELECT=0, DAV_SYM_OVERLAP or DAV_COLL_DIRECT in the environment at dav_create switch the trial off.
static int coll_path_trial(E* e, int which, OpDesc& o, const SymSet& set, const double* src, int k, double* dst, bool timed) {
  const Tune saved = e->tune;
  const size_t blk = (size_t)e->ldp * k;
  double* scratch_blk = nullptr;
  double* cmp_dev = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  auto cleanup = [&]() {
    (void)hipStreamSynchronize(e->stream);
    if (e->comm_stream) (void)hipStreamSynchronize(e->comm_stream);
    pool_free(scratch_blk); pool_free(cmp_dev);
    for (hipEvent_t v : ev) if (v) (void)hipEventDestroy(v);
  };
  e->coll_path = COLL_PATH_PROGRAM_ORDER;                 // the runs below must not come back here
  e->tune.coll_direct = 0; e->tune.sym_overlap = 0;
  // the block in program order: THE result (timed like any other apply of the solve)
  int rc = apply_sym_set(e, which, o, set, false, false, src, k, dst, timed, false);
  if (rc != 0) { e->tune = saved; return rc; }
  bool setup_ok = pool_malloc(&scratch_blk, sizeof(double) * blk) == hipSuccess && pool_malloc(&cmp_dev, sizeof(double) * 3 * (size_t)k) == hipSuccess &&
                  hipEventCreate(&ev[0]) == hipSuccess && hipEventCreate(&ev[1]) == hipSuccess;
  CollTrial& t = e->coll_trial;
  t = CollTrial();
  t.ran = true;
  t.columns = k;
  if (!setup_ok) {
    (void)hipGetLastError();
    cleanup();
    e->tune = saved; e->tune.coll_direct = 0; e->tune.sym_overlap = 0;
    fprintf(stderr, "davidson (rank %d): no memory for the collective-path trial - program order\n", e->rank);
    return 0;
  }
  std::vector<double> cmp_host(3 * (size_t)k);
  for (int path = 0; path < 3; ++path) {
    e->tune.coll_direct = path == COLL_PATH_DIRECT ? 1 : 0;
    e->tune.sym_overlap = path == COLL_PATH_SECOND_STREAM ? 1 : 0;
    bool ok = true;
    float ms = 0.0f;
    for (int rep = 0; rep < 2 && ok; ++rep) {             // the first run of a way pays its lazy set-up; the second is timed
      if (rep == 1) ok = hipEventRecord(ev[0], e->stream) == hipSuccess;
      if (ok) {
        const int r = apply_sym_set(e, which, o, set, false, false, src, k, scratch_blk, false, false);
        if (r != 0) { ok = false; t.message[path] = g_err; }
        else if (path == COLL_PATH_SECOND_STREAM && !e->ov_ready) { ok = false; t.message[path] = "its stream / buffers could not be set up"; }
      }
      if (ok && rep == 1) ok = hipEventRecord(ev[1], e->stream) == hipSuccess && hipEventSynchronize(ev[1]) == hipSuccess &&
                               hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess;
    }
    if (!ok) (void)hipGetLastError();
    t.valid[path] = ok;
    t.ms[path] = ms;
    if (ok && path != COLL_PATH_PROGRAM_ORDER) {
      // test hook of the PRODUCT build (tests/test_rccl_one_gpu.py: an injected mismatch): DAV_COLL_TRIAL_CORRUPT = 1 | 2 spoils one
      // entry of that way's result on rank 0
      if (e->tune.coll_trial_corrupt == path && e->rank == 0) launch_poke(e->stream, scratch_blk, 1.0);
      launch_compare_blocks(e->stream, dst, scratch_blk, e->ldp, e->nloc, k, cmp_dev);
      if (hipMemcpyAsync(cmp_host.data(), cmp_dev, sizeof(double) * 3 * k, hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
          hipStreamSynchronize(e->stream) != hipSuccess) { (void)hipGetLastError(); t.valid[path] = false; t.message[path] = "comparison failed"; continue; }
      double differing = 0.0, maxdiff = 0.0, maxref = 0.0;
      for (int j = 0; j < k; ++j) { differing += cmp_host[3 * j]; maxdiff = std::max(maxdiff, cmp_host[3 * j + 1]); maxref = std::max(maxref, cmp_host[3 * j + 2]); }
      t.differing[path] = differing;
      t.maxdiff[path] = maxdiff;
      const bool bitwise = differing == 0.0;
      const bool close = maxdiff <= 1e-12 * maxref;
      t.valid[path] = (path == COLL_PATH_DIRECT && e->nranks > 2) ? close : bitwise;
      if (!t.valid[path]) t.message[path] = "result differs from program order";
    }
  }
  e->tune = saved; e->tune.coll_direct = 0; e->tune.sym_overlap = 0;
  // make the figures common: slot r of each word carries rank r's value (the SUM all-reduce reproduces every rank's)
  const int nw = 6;
  std::vector<double> words((size_t)nw * e->nranks, 0.0);
  for (int p = 0; p < 3; ++p) {
    words[(size_t)p * e->nranks + e->rank] = t.ms[p];
    words[(size_t)(3 + p) * e->nranks + e->rank] = t.valid[p] ? 0.0 : 1.0;
  }
  int chosen = COLL_PATH_PROGRAM_ORDER;
  if ((size_t)nw * e->nranks <= e->gram_doubles &&
      hipMemcpyAsync(e->gram_dev, words.data(), sizeof(double) * words.size(), hipMemcpyHostToDevice, e->stream) == hipSuccess &&
      coll_allreduce(e, e->gram_dev, words.size()) == 0 &&
      hipMemcpyAsync(words.data(), e->gram_dev, sizeof(double) * words.size(), hipMemcpyDeviceToHost, e->stream) == hipSuccess &&
      hipStreamSynchronize(e->stream) == hipSuccess) {
    double best = 0.0;
    for (int p = 0; p < 3; ++p) {
      double worst = 0.0, bad = 0.0;
      for (int r = 0; r < e->nranks; ++r) { worst = std::max(worst, words[(size_t)p * e->nranks + r]); bad += words[(size_t)(3 + p) * e->nranks + r]; }
      t.ms_max[p] = worst;
      t.valid_all[p] = bad == 0.0 && worst > 0.0;
      if (t.valid_all[p] && (p == COLL_PATH_PROGRAM_ORDER || worst < best)) { best = worst; chosen = p; }
      if (p == COLL_PATH_PROGRAM_ORDER && !t.valid_all[p]) { t.valid_all[p] = true; best = worst; }      // always available
    }
  } else {
    (void)hipGetLastError();
    fprintf(stderr, "davidson (rank %d): the ranks could not agree on the collective-path trial - program order\n", e->rank);
  }
  static const char* names[3] = {"program order", "direct exchange", "second stream"};
  for (int p = 1; p < 3; ++p)
    if (!t.valid_all[p] && e->rank == 0)
      fprintf(stderr, "davidson: collective path '%s' left out of the selection (%s%s)\n", names[p], t.valid[p] ? "another rank's comparison failed" : "this rank: ",
              t.valid[p] ? "" : t.message[p].c_str());
  t.selected = chosen;
  e->coll_path = chosen;
  e->tune.coll_direct = chosen == COLL_PATH_DIRECT ? 1 : 0;
  e->tune.sym_overlap = chosen == COLL_PATH_SECOND_STREAM ? 1 : 0;
  if (saved.gjd_trace || getenv("DAVIDSON_VERBOSE"))
    if (e->rank == 0)
      fprintf(stderr, "davidson: collectives of wide blocks: %s (trial of a %d-column block, ms max over ranks: program order %.3f, direct exchange %.3f%s, second stream %.3f%s)\n",
              names[chosen], k, t.ms_max[0], t.ms_max[1], t.valid_all[1] ? "" : " [invalid]", t.ms_max[2], t.valid_all[2] ? "" : " [invalid]");
  cleanup();
  return 0;
}

// The symmetric-tiled sweep of the block rows of `set` of operator view `o` (stored tiles at o.a / generated entries):
// dst[:, 0:k] (+)= Op * src[:, 0:k].  partial: `set` is not all of the rank's block rows (an operator swept in two parts: its
// resident and its generated block rows); accumulate: the result is added to dst.
static int apply_sym_set(E* e, int which, OpDesc& o, const SymSet& set, bool partial, bool accumulate, const double* src, int k, double* dst,
                         bool timed, bool inner) {
  {
    // symmetric-tiled sweep: every off-diagonal tile read (or generated) once, used twice.  16 columns per workgroup; 32
    // columns per launch as paired workgroups that share their tile reads through the memory-side cache.
    // Several ranks: each sweeps the block rows it stores against the all-gathered block and holds a partial of the
    // WHOLE product; one reduce-scatter per 16 columns sums the partials and leaves every rank its row slab.
    // pairing shares the READS of stored tiles: nothing to share when the entries are generated
    // ... except where the generating variant of the wide kernel shares the GENERATED entries between two groups (hashed operator,
    // two-block-row schedule)
    const bool gen_wide = (o.kind == DAV_KIND_HASHED || (o.kind == DAV_KIND_HARNESS && !o.harness_libm)) && e->tune.sym_gen_wide && e->tune.sym_wide > 0 &&
                          sym_schedule(e, 32, false) == 2;
    int step = (e->tune.sym_pair && !e->sym_no_pair && (o.kind == DAV_KIND_DENSE || gen_wide)) ? 32 : 16;
    // several ranks - or a communicator on a single rank (DAVIDSON_FORCE_RCCL=1: the GPU tests run the all-gather and the
    // reduce-scatter of this path through RCCL on a one-GPU box)
    const bool multi = e->nranks > 1 || has_comm(e);
    const bool pair_ok = step == 32;
    // Opt-in (Tune::sym_overlap, DAV_SYM_OVERLAP=1 at dav_create): the collectives of a block wider than 32 columns run on a second
    // stream under the sweeps of its 32-column chunks.  All collectives of that pipeline are issued on the ONE communication stream in
    // the same order on every rank, ordered against the engine's stream by events, so no two collectives of the communicator are
    // ever in flight together.  Off by default: it has never run over more than one RCCL rank (round-4 advisor); the default below
    // keeps every collective on the engine's stream in program order.
    const bool wide_block = !partial && pair_ok && k > 32 && o.kind == DAV_KIND_DENSE && sym_schedule(e, 32, true) == 2;
    // Round 6: which way the collectives of a wide block go is decided by the engine itself, once, at the first such block over a
    // real communicator of several ranks (coll_path_trial below) - unless the environment forced a path at dav_create
    if (wide_block && e->comm && e->nranks > 1 && e->coll_path == COLL_PATH_UNDECIDED && !inner) return coll_path_trial(e, which, o, set, src, k, dst, timed);
    if (e->tune.sym_overlap != 0 && wide_block && (e->comm || has_test_transport(e))) {
      const int rc = apply_sym_overlapped(e, which, o, src, k, dst, timed, inner);
      if (rc != 2) return rc;                          // 2: its streams / buffers / slabs could not be set up - serial path below
    }
    // 64 columns (the widest expansion of the doubling policy below a basis of 128) as FOUR column groups in one launch on
    // the super-row kernels: the four workgroups of a work item share every tile read through their XCD's L2.  Same box,
    // N=200000, k=64: two paired launches 102.6 ms, one launch of four groups 93.6 ms (56.9 TFLOP/s).  Tune::sym_quad = 0: off.
    // Several ranks (round 5): the same launch between ONE all-gather and ONE reduce-scatter of all four column groups - a block of
    // 64 columns costs two collectives, not four.
    if (e->tune.sym_quad && pair_ok && o.kind == DAV_KIND_DENSE && k >= 64 && !inner && !e->sym_no_quad && sym_schedule(e, 32, true) == 2) step = 64;
    if (multi && !e->sym_wpart) {
      HIPCHK(pool_malloc(&e->sym_wpart, sizeof(double) * (size_t)e->nranks * (size_t)e->nslab * 64));
      HIPCHK(pool_malloc(&e->sym_wrecv, sizeof(double) * (size_t)e->nslab * 64));
    }
    const int64_t* owned = (multi || partial) ? set.row_off : nullptr;
    const int64_t total_rows = (int64_t)e->nranks * e->nslab;
    for (int c = 0; c < k; c += step) {
      int kk = std::min(step, k - c);
      int npair = (kk + 15) / 16;
      // fp32 tiles (inner sweeps of the GJD correction, opt-in) where the sweep is bound by bytes: up to 16 columns.  Wider ones
      // are bound by the fp64 matrix pipe either way, and the one-wave-per-SIMD kernel on the fp64 tiles is the faster of the two
      const bool use32 = inner && kk <= 16 && inner_f32_tiles(e, o);
      int R = sym_schedule(e, kk, o.kind == DAV_KIND_DENSE && !use32);      // (the harness operator runs the super-row kernels too since round 5)
      if (use32 && R == 1) R = 2;            // the fp32 tiles are read by the super-row kernels only
      const SymPlan* pl = R > 1 ? &set.plan[R == 4 ? 1 : 0] : nullptr;
      const int64_t dstride = R > 1 ? (int64_t)pl->nitems * R * 16 * SYM_TB : (int64_t)set.nitems * 16 * SYM_TB;
      const int64_t tstride = R > 1 ? pl->zslots * 16 * SYM_TB : (int64_t)e->sym_nb * (e->sym_nb - 1) / 2 * 16 * SYM_TB;
      while (sym_ensure_slabs(e, (size_t)npair * (size_t)(dstride + tstride) + 1) != 0) {
        // not enough memory for this many column groups per launch: fewer from here on (4 -> 2 -> 1) - on ONE rank.  Several ranks
        // must issue the same collectives: a rank that quietly fell back to narrower launches would leave its peers in theirs
        if (npair < 2) return 1;
        if (e->nranks > 1) return fail("symmetric sweep: no room for the partial-sum slabs of a " + std::to_string(16 * npair) +
                                       "-column launch on rank " + std::to_string(e->rank) + " (DAV_SYM_QUAD=0 / DAV_SYM_PAIR=0 on every rank select narrower launches)");
        if (npair > 2) { e->sym_no_quad = true; step = 32; kk = 32; npair = 2; }
        else { e->sym_no_pair = true; step = 16; kk = 16; npair = 1; }
      }
      int slot = -1, kslot = -1;
      // stored bytes of this part: the whole triangle dealt out over the ranks, or - a part of an operator - the set's own tiles
      const double stored = o.kind == DAV_KIND_DENSE ? (use32 ? 4.0 : 8.0) * (partial ? (double)set.ntiles * SYM_TB * SYM_TB
                                                                                        : 0.5 * (double)e->n * ((double)e->n + 1.0) / e->nranks) : 0.0;
      double bytes = stored + 16.0 * (double)e->n * kk;
      // end to end: everything that turns the source columns into W - packing, (all-gather,) the sweep, the fixed-order sum(, reduce-scatter)
      if (timed) CHK(timed_begin(e, which == DAV_OP_A ? 0 : 2, bytes, &slot));
      launch_pack_xt(e->stream, src + (int64_t)c * e->ldp, e->ldp, e->nloc, e->nslab, kk, e->xt, e->xt_group_stride, e->row0);
      if (multi) {
        CollGroup grp(e);
        CHK(grp.begin(5, 8.0 * (double)e->nslab * 16 * npair * e->nranks));
        for (int g = 0; g < npair; ++g) {
          double* base = e->xt + g * e->xt_group_stride;
          CHK(coll_allgather(e, base + e->row0 * 16, base, (size_t)e->nslab * 16));
        }
        CHK(grp.end("all-gather of the new block", e->stream));
      }
      if (timed && which == DAV_OP_A) CHK(timed_begin(e, 4, 2.0 * (double)e->n * (double)e->n * kk / e->nranks, &kslot));
      if (timed && which == DAV_OP_B) {
        // the second operator's sweep kernels by what they read (level 2): stored tiles -> bytes, generated block rows -> entries
        // evaluated (once per 16 columns; once per 32 where the generating variant of the wide kernel runs)
        const double tiles_entries = (double)set.ntiles * SYM_TB * SYM_TB;
        const bool gen_shared = (o.kind == DAV_KIND_HASHED || (o.kind == DAV_KIND_HARNESS && !o.harness_libm)) && R == 2 && kk > 16 && e->tune.sym_wide > 0 && e->tune.sym_gen_wide;
        if (o.kind == DAV_KIND_DENSE) CHK(timed_begin(e, 8, (use32 ? 4.0 : 8.0) * tiles_entries + 16.0 * (double)e->n * kk, &kslot));
        else CHK(timed_begin(e, 9, tiles_entries * (gen_shared ? (npair + 1) / 2 : npair), &kslot));
        if (kslot >= 0) e->ev_flops[kslot] = 4.0 * tiles_entries * kk;          // every stored / generated entry is used twice
      }
      double* slabT = e->sym_slab + (int64_t)npair * dstride;
      const int nitems = R > 1 ? pl->nitems : set.nitems;
      if (nitems > 0) {                      // a rank can be left without a block row (more ranks than groups of block rows)
        if (R > 1)
          sym9_sweep(e, R, o, use32, set, pl, e->xt, kk, e->sym_slab, slabT, npair, dstride, tstride);
        else if (o.kind != DAV_KIND_DENSE)
          launch_matvec_sym_generated(e->stream, op_params(o), e->n, set.items, set.nitems, e->xt, kk, e->sym_slab, slabT, npair,
                                      e->xt_group_stride, dstride, tstride);
        else
          launch_matvec_sym(e->stream, o.a, set.row_off, set.items, set.nitems, e->xt, kk, e->sym_slab, slabT, npair,
                            e->xt_group_stride, dstride, tstride);
      }
      CHK(timed_end(e, kslot));
      for (int g = 0; g < npair; ++g) {
        const int kg = std::min(16, kk - 16 * g);
        double* out = multi ? e->sym_wpart + (size_t)g * (size_t)total_rows * 16 : dst + (int64_t)(c + 16 * g) * e->ldp;
        if (R > 1)
          launch_sym9_reduce(e->stream, e->sym_slab + g * dstride, slabT + g * tstride, pl->row_begin, pl->zslot_begin, owned, pl->next_owned, R, e->sym_nb,
                             e->nloc, kg, out, e->ldp, multi ? e->nslab : 0, total_rows, accumulate && !multi);
        else
          launch_sym_reduce(e->stream, e->sym_slab + g * dstride, slabT + g * tstride, set.row_begin, owned, e->sym_nb, e->nloc, kg,
                            out, e->ldp, multi ? e->nslab : 0, total_rows, accumulate && !multi);
      }
      if (multi) {
        CollGroup grp(e);
        CHK(grp.begin(6, 8.0 * (double)e->nslab * kk * e->nranks));
        for (int g = 0; g < npair; ++g) {
          const int kg = std::min(16, kk - 16 * g);
          CHK(coll_reduce_scatter(e, e->sym_wpart + (size_t)g * (size_t)total_rows * 16, e->sym_wrecv + (size_t)g * (size_t)e->nslab * 16,
                                  (size_t)e->nslab * kg));
        }
        CHK(grp.end("reduce-scatter of the partial products", e->stream));
        for (int g = 0; g < npair; ++g)
          launch_chunk_to_panel(e->stream, e->sym_wrecv + (size_t)g * (size_t)e->nslab * 16, e->nslab, e->nloc, e->nloc_pad,
                                std::min(16, kk - 16 * g), dst + (int64_t)(c + 16 * g) * e->ldp, e->ldp, accumulate);
      }
      CHK(timed_end(e, slot));
      if (which == DAV_OP_A) {
        e->st.applies += 1;
        e->st.apply_cols += kk;
      }
    }
    HIPCHK(hipGetLastError());
    return 0;
  }
}

// inner = true: a sweep inside the GJD correction solve (may run on the fp32 copy, dav_set_inner_precision)
int apply_ptr(E* e, int which, const double* src, int k, double* dst, bool timed, bool inner) {
  OpDesc& o = e->op[which];
  if (o.kind == DAV_KIND_NONE) return fail("dav_apply: operator not set");
  if (o.kind == DAV_KIND_HOST) return fail("dav_apply: host operator - move blocks with dav_panel_get/put");
  if (o.kind == DAV_KIND_IDENTITY) {
    launch_copy_columns(e->stream, src, e->ldp, dst, e->ldp, e->nloc_pad, k);
    return 0;
  }
  CHK(need_comm(e));
  if (o.kind == DAV_KIND_DEVICE) {
    // the caller's kernel(s): the block as ONE column-major matrix of all n rows (several ranks: gathered column by column in one
    // grouped collective - slab r of a column is rows [r * nslab, (r + 1) * nslab)), the result straight into this rank's panel rows
    const double* x = src;
    int64_t ldx = e->ldp;
    int slot = -1;
    if (timed) CHK(timed_begin(e, which == DAV_OP_A ? 0 : 2, 16.0 * (double)e->n * k, &slot));
    if (has_comm(e)) {
      ldx = (int64_t)e->nranks * e->nslab;
      const size_t need = (size_t)ldx * k;
      if (need > e->cb_x_doubles) {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->cb_x) HIPCHK(pool_free(e->cb_x));
        e->cb_x = nullptr; e->cb_x_doubles = 0;
        HIPCHK(pool_malloc(&e->cb_x, sizeof(double) * need));
        e->cb_x_doubles = need;
      }
      CollGroup grp(e);
      CHK(grp.begin(5, 8.0 * (double)e->nslab * k * e->nranks));
      for (int c = 0; c < k; ++c) CHK(coll_allgather(e, src + (int64_t)c * e->ldp, e->cb_x + (int64_t)c * ldx, (size_t)e->nslab));
      CHK(grp.end("all-gather of the new block", e->stream));
      x = e->cb_x;
    }
    const int rc = o.dev_fn(o.dev_ctx, (void*)e->stream, e->n, e->row0, e->nloc, k, x, ldx, dst, e->ldp);
    if (rc != 0) return fail("dav_apply: the caller's device operator returned " + std::to_string(rc));
    launch_zero_pad_rows(e->stream, dst, e->ldp, e->nloc, e->nloc_pad, k);     // the panels' padding rows stay zero whatever the callback left there
    CHK(timed_end(e, slot));
    if (which == DAV_OP_A) { e->st.applies += 1; e->st.apply_cols += k; }
    HIPCHK(hipGetLastError());
    return 0;
  }
  if ((o.kind == DAV_KIND_DENSE || o.kind == DAV_KIND_HASHED || o.kind == DAV_KIND_HARNESS) && o.storage == 1) {
    // A generated second operator whose tiles (partly) fit next to everything else is kept resident for its longest block rows
    // (configs[3]: B = the unit-diagonal generator next to a stored A): those rows run the stored kernels - half the time per
    // 16 columns of the generated sweep, a quarter in the 32- / 64-column launches - the others are generated as before; the
    // two parts are summed in fixed order (the generated part adds to the resident part's result).
    if (o.kind == DAV_KIND_HASHED && which == DAV_OP_B && !o.res_decided) CHK(sym_resident_split(e, which));
    if (o.res) {
      bool accumulate = false;
      if (o.pass_res) {
        OpDesc stored = OpDesc();
        stored.kind = DAV_KIND_DENSE; stored.storage = 1; stored.a = o.res_a; stored.a32_refused = true;
        CHK(apply_sym_set(e, which, stored, *o.res, true, false, src, k, dst, timed, inner));
        accumulate = true;
      }
      if (o.pass_gen) CHK(apply_sym_set(e, which, o, *o.gen, true, accumulate, src, k, dst, timed, inner));
      return 0;
    }
    return apply_sym_set(e, which, o, e->sym, false, false, src, k, dst, timed, inner);
  }
  for (int c = 0; c < k; c += 64) {
    int kk = std::min(64, k - c);
    int groups = (kk + 15) / 16;
    int ngroups = groups == 3 ? 4 : groups;
    int slot = -1, kslot = -1;
    double bytes = 8.0 * (double)e->nloc * (double)e->n + 16.0 * (double)e->n * kk;
    if (timed) CHK(timed_begin(e, which == DAV_OP_A ? 0 : 2, bytes, &slot));
    launch_pack_xt(e->stream, src + (int64_t)c * e->ldp, e->ldp, e->nloc, e->nslab, kk, e->xt, e->xt_group_stride, e->row0);
    if (has_comm(e)) {
      CollGroup grp(e);
      CHK(grp.begin(5, 8.0 * (double)e->nslab * 16 * groups * e->nranks));
      for (int g = 0; g < groups; ++g) {
        double* base = e->xt + g * e->xt_group_stride;
        CHK(coll_allgather(e, base + e->row0 * 16, base, (size_t)e->nslab * 16));
      }
      CHK(grp.end("all-gather of the new block", e->stream));
    }
    int nsplit, jc;
    matvec_plan(e->nloc_pad, e->ncols_pad, ngroups, &nsplit, &jc, e->tune.mv_target, e->tune.mv_nsplit);
    if (matvec_slab_doubles(e->nloc_pad, ngroups, nsplit) > e->scratch_doubles) return fail("matvec scratch too small");
    if (timed && which == DAV_OP_A) CHK(timed_begin(e, 4, 2.0 * (double)e->nloc * (double)e->n * kk, &kslot));
    if (o.kind == DAV_KIND_DENSE)
      launch_matvec_dense(e->stream, o.a, e->nloc_pad, e->nloc_pad, e->ncols_pad, e->xt, e->xt_group_stride, ngroups,
                          e->scratch, nsplit, jc);
    else
      launch_matvec_free(e->stream, op_params(o), e->row0, e->nloc, e->n, e->nloc_pad, e->ncols_pad, e->xt,
                         e->xt_group_stride, ngroups, e->scratch, nsplit, jc);
    CHK(timed_end(e, kslot));               // inner pair: the block-matvec kernel alone
    launch_slab_reduce(e->stream, e->scratch, nsplit, e->nloc_pad, ngroups, e->nloc, kk, dst + (int64_t)c * e->ldp, e->ldp);
    CHK(timed_end(e, slot));                // outer pair: pack + all-gather + kernel + reduction
    if (which == DAV_OP_A) {
      e->st.applies += 1;
      e->st.apply_cols += kk;
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// W0 = Op * V0 for the unit columns V0 = e_idx of dav_init_basis over SEVERAL ranks of dealt-out symmetric tiles, without a sweep:
// column p of the operator lies in the tiles of many ranks (rows at or below p's block row: the owners of those block rows, column
// p of their tiles; rows above: the owner of p's block row, row p of its tiles); every rank writes what it holds in the
// reduce-scatter layout of the sweeps and the same reduce-scatter leaves every rank its row slab.  One owner per entry, so the sum
// is exact and equals the sweep's result (products with the zeros of e_p add nothing).  A sweep of 32 columns costs 42 ms / P at
// N=200000 - a third of a configs[2] solve on several GPUs; this costs the reduce-scatter of 2 x N x 16 doubles.
int gather_columns_sym_multi(E* e, OpDesc& o, int ncols, double* dst, double* h0) {
  CHK(need_comm(e));
  if (!e->sym_wpart) {
    HIPCHK(pool_malloc(&e->sym_wpart, sizeof(double) * (size_t)e->nranks * (size_t)e->nslab * 64));
    HIPCHK(pool_malloc(&e->sym_wrecv, sizeof(double) * (size_t)e->nslab * 64));
  }
  const int64_t total_rows = (int64_t)e->nranks * e->nslab;
  for (int c = 0; c < ncols; c += 32) {
    const int kk = std::min(32, ncols - c), npair = (kk + 15) / 16;
    for (int g = 0; g < npair; ++g)
      launch_gather_columns_sym_rs(e->stream, o.a, e->sym.row_off, e->n, e->nslab, total_rows, e->idx_dev + c + 16 * g, std::min(16, kk - 16 * g),
                                   e->sym_wpart + (size_t)g * (size_t)total_rows * 16);
    // h0 (dav_init_basis): the entries (idx_i, idx_j) this rank holds, summed over the ranks by an all-reduce that is a member of the
    // first reduce-scatter's group - V0^T W0 without a collective (and a Gram product) of its own
    if (h0 && c == 0) launch_entries_sym(e->stream, o.a, e->sym.row_off, e->idx_dev, ncols, h0);
    CollGroup grp(e);
    CHK(grp.begin(6, 8.0 * (double)e->nslab * kk * e->nranks));
    for (int g = 0; g < npair; ++g)
      CHK(coll_reduce_scatter(e, e->sym_wpart + (size_t)g * (size_t)total_rows * 16, e->sym_wrecv + (size_t)g * (size_t)e->nslab * 16,
                              (size_t)e->nslab * std::min(16, kk - 16 * g)));
    if (h0 && c == 0) CHK(coll_allreduce(e, h0, (size_t)ncols * ncols));
    CHK(grp.end("reduce-scatter of the gathered columns", e->stream));
    for (int g = 0; g < npair; ++g)
      launch_chunk_to_panel(e->stream, e->sym_wrecv + (size_t)g * (size_t)e->nslab * 16, e->nslab, e->nloc, e->nloc_pad, std::min(16, kk - 16 * g),
                            dst + (int64_t)(c + 16 * g) * e->ldp, e->ldp, false);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int apply_impl(E* e, int which, int src_panel, int c0, int k, int dst_panel, int d0, bool timed) {
  CHK(check_panel(e, src_panel, c0, k));
  CHK(check_panel(e, dst_panel, d0, k));
  return apply_ptr(e, which, panel_ptr(e, src_panel, c0), k, panel_ptr(e, dst_panel, d0), timed);
}

extern "C" int dav_apply(dav_handle_t e, int which, int src_panel, int c0, int k, int dst_panel, int d0) {
  if (which < 0 || which > 1) return fail("dav_apply: bad operator id");
  CHK(bind(e));
  return apply_impl(e, which, src_panel, c0, k, dst_panel, d0, true);
}

// dav_apply as the GJD correction solve issues it (csrc/davidson_hip_private.h): an inner sweep may read the fp32 copy of the tiles
extern "C" int dav_apply_inner(dav_handle_t e, int which, int src_panel, int c0, int k, int dst_panel, int d0) {
  if (which < 0 || which > 1) return fail("dav_apply_inner: bad operator id");
  CHK(bind(e));
  CHK(check_panel(e, src_panel, c0, k));
  CHK(check_panel(e, dst_panel, d0, k));
  return apply_ptr(e, which, panel_ptr(e, src_panel, c0), k, panel_ptr(e, dst_panel, d0), false, true);
}

// ---- measurement --------------------------------------------------------------------------------
extern "C" int dav_bench_apply(dav_handle_t e, int which, int k, int reps, double* avg_ms, double* bytes) {
  double kernel_ms, flops;
  return dav_bench_apply2(e, which, k, reps, avg_ms, &kernel_ms, bytes, &flops);
}

extern "C" int dav_bench_apply2(dav_handle_t e, int which, int k, int reps, double* avg_ms, double* kernel_ms, double* bytes,
                                double* flops) {
  CHK(bind(e));
  if (which != DAV_OP_A) return fail("dav_bench_apply: only operator A is timed");
  if (k <= 0 || k > 64 || reps <= 0) return fail("dav_bench_apply: k must be in 1..64");
  CHK(collect_events(e));
  dav_stats saved = e->st;
  // warm up once, then time whole applies (pack + kernel + reduction; operands resident in HBM)
  const int saved_level = e->timing_level;
  e->timing_level = 1;
  CHK(apply_impl(e, which, DAV_PANEL_V, 0, k, DAV_PANEL_S, 0, false));
  HIPCHK(hipStreamSynchronize(e->stream));
  double total = 0, ktotal = 0;
  int done = 0;
  while (done < reps) {
    int batch = std::min(reps - done, N_EVPAIRS / 8);
    e->st.apply_ms = 0;
    e->st.apply_kernel_ms = 0;
    for (int i = 0; i < batch; ++i) CHK(apply_impl(e, which, DAV_PANEL_V, 0, k, DAV_PANEL_S, 0, true));
    CHK(collect_events(e));
    total += e->st.apply_ms;
    ktotal += e->st.apply_kernel_ms;
    done += batch;
  }
  e->timing_level = saved_level;
  *avg_ms = total / reps;
  *kernel_ms = ktotal / reps;
  const bool sym = e->op[which].storage == 1;
  // per rank: the stored bytes and the flops of the symmetric sweep are dealt out over the ranks like its tiles
  *bytes = (sym ? (e->op[which].kind == DAV_KIND_DENSE ? 8.0 * 0.5 * (double)e->n * ((double)e->n + 1.0) / e->nranks : 0.0)
                : 8.0 * (double)e->nloc * (double)e->n) + 16.0 * (double)e->n * k;
  *flops = 2.0 * (sym ? (double)e->n / e->nranks : (double)e->nloc) * (double)e->n * k;
  e->st = saved;
  return 0;
}

// What the HBM of THIS box delivers to a plain streaming kernel: device copy (a = b) and triad (a = b + s c) over arrays of
// `doubles` entries each (0 = 2^28: 2 GiB per array), read + written bytes per second ("re-measure achievable BW
// with a stream-triad on the box and report fraction of both").  Allocates and frees its three arrays.
extern "C" int dav_bench_stream3(dav_handle_t e, int64_t doubles, int reps, double* copy_GBps, double* triad_GBps, double* read_GBps) {
  CHK(bind(e));
  const int64_t n = doubles > 0 ? doubles / 2 * 2 : (int64_t)1 << 28;
  if (reps <= 0) reps = 5;
  double *a = nullptr, *b = nullptr, *c = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  auto cleanup = [&]() {
    pool_free(a); pool_free(b); pool_free(c);
    for (hipEvent_t v : ev) if (v) hipEventDestroy(v);
  };
  if (pool_malloc(&a, sizeof(double) * n) != hipSuccess || pool_malloc(&b, sizeof(double) * n) != hipSuccess ||
      pool_malloc(&c, sizeof(double) * n) != hipSuccess || hipEventCreate(&ev[0]) != hipSuccess || hipEventCreate(&ev[1]) != hipSuccess) {
    (void)hipGetLastError();
    cleanup();
    return fail("dav_bench_stream: could not allocate three arrays of " + std::to_string(n) + " doubles");
  }
  double out[3] = {0.0, 0.0, 0.0};
  int rc = 0;
  auto run = [&]() -> int {
    HIPCHK(hipMemsetAsync(b, 0, sizeof(double) * n, e->stream));
    HIPCHK(hipMemsetAsync(c, 0, sizeof(double) * n, e->stream));
    for (int mode = 0; mode < 3; ++mode) {                                        // copy, triad, reads only (two arrays)
      launch_stream(e->stream, mode, a, b, c, 0.5, n);                        // warm-up
      HIPCHK(hipEventRecord(ev[0], e->stream));
      for (int r = 0; r < reps; ++r) launch_stream(e->stream, mode, a, b, c, 0.5, n);
      HIPCHK(hipEventRecord(ev[1], e->stream));
      HIPCHK(hipEventSynchronize(ev[1]));
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, ev[0], ev[1]));
      out[mode] = (mode == 1 ? 3.0 : 2.0) * 8.0 * (double)n * reps / (ms * 1e-3) / 1e9;
    }
    return 0;
  };
  rc = run();
  cleanup();
  if (rc != 0) return rc;
  if (copy_GBps) *copy_GBps = out[0];
  if (triad_GBps) *triad_GBps = out[1];
  if (read_GBps) *read_GBps = out[2];
  return 0;
}
extern "C" int dav_bench_stream(dav_handle_t e, int64_t doubles, int reps, double* copy_GBps, double* triad_GBps) {
  return dav_bench_stream3(e, doubles, reps, copy_GBps, triad_GBps, nullptr);
}

// Entries of the matrix-free test operator (atan2 + sqrt + log + cos in fp64, src/tests/test_utils.f90:72-116) the chip evaluates
// per second when nothing else is in the way: the roof bench.py prices the generated sweeps of that operator against.
extern "C" int dav_bench_harness_rate(dav_handle_t e, int iters, double* entries_per_s) {
  CHK(bind(e));
  if (iters <= 0) iters = 2000;
  const int wgs = 512;
  double* out = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  HIPCHK(pool_malloc(&out, sizeof(double) * (size_t)wgs * 512));
  auto cleanup = [&]() { pool_free(out); for (hipEvent_t v : ev) if (v) hipEventDestroy(v); };
  auto run = [&]() -> int {
    HIPCHK(hipEventCreate(&ev[0]));
    HIPCHK(hipEventCreate(&ev[1]));
    launch_harness_rate(e->stream, out, wgs, 200);           // warm-up
    HIPCHK(hipEventRecord(ev[0], e->stream));
    launch_harness_rate(e->stream, out, wgs, iters);
    HIPCHK(hipEventRecord(ev[1], e->stream));
    HIPCHK(hipEventSynchronize(ev[1]));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev[0], ev[1]));
    *entries_per_s = (double)wgs * 512.0 * iters / (ms * 1e-3);
    return 0;
  };
  const int rc = run();
  cleanup();
  return rc;
}
