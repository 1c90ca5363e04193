// Engine, the phases of an outer iteration besides the sweep: K2 Gram / projection, K6 initial basis, K3 fused Ritz /
// residual / correction phase, K4 block orthonormalisation, expansion, K5 restart, and the opt-in device-side Rayleigh-Ritz.
#include "engine_internal.h"

// ---- K2 -----------------------------------------------------------------------------------------
// Small results (Gram blocks, norms, dots) reach the host without a copy command: a single rank lets
// the final reduction kernel write straight into device-visible pinned memory and only synchronises
// the stream; with a communicator the partial result is all-reduced in HBM first and then copied.
double* result_target(E* e) { return has_comm(e) ? e->gram_dev : e->gram_host_dev; }

// Several ranks: the all-reduce of `count` doubles at gram_dev - with the control words of dav_agree_next behind them when the
// driver left any (word i of rank r in slot i * nranks + r, zeros from the other ranks: the sum reproduces every rank's value).
// ONE collective carries the result and the agreement check (until round 4 the check was an all-reduce of its own per outer
// iteration).  total_out: doubles to bring to the host (result + words).
int allreduce_with_agreement(E* e, size_t count, size_t* total_out) {
  size_t total = count;
  const size_t nw = e->agree_words.size();
  if (nw > 0 && e->nranks > 1) {
    const size_t off = (count + 7) / 8 * 8, span = nw * (size_t)e->nranks;
    if (off + span <= e->gram_doubles) {
      // pinned staging that lives as long as the engine; every fetch that carries words ends in a stream synchronisation, so the
      // previous copy has left it
      std::memset(e->agree_pin, 0, sizeof(double) * span);
      for (size_t i = 0; i < nw; ++i) e->agree_pin[i * e->nranks + e->rank] = e->agree_words[i];
      if (off > count) HIPCHK(hipMemsetAsync(e->gram_dev + count, 0, sizeof(double) * (off - count), e->stream));
      HIPCHK(hipMemcpyAsync(e->gram_dev + off, e->agree_pin, sizeof(double) * span, hipMemcpyHostToDevice, e->stream));
      total = off + span;
    }
  }
  CHK(coll_allreduce(e, e->gram_dev, total));
  *total_out = total;
  return 0;
}
// after the copy to gram_host and the synchronisation: every rank must have contributed the same words
int agreement_verify(E* e, size_t count) {
  const size_t nw = e->agree_words.size();
  if (nw == 0 || e->nranks <= 1) { e->agree_words.clear(); return 0; }
  const size_t off = (count + 7) / 8 * 8, span = nw * (size_t)e->nranks;
  std::vector<double> words;
  words.swap(e->agree_words);                            // consumed: one check per dav_agree_next
  if (off + span > e->gram_doubles) return dav_ranks_agree(e, words.data(), (int)nw);     // no room behind this result: its own collective
  const double* got = e->gram_host + off;
  for (size_t i = 0; i < nw; ++i)
    for (int r = 0; r < e->nranks; ++r)
      if (got[i * e->nranks + r] != words[i])
        return fail("ranks disagree on a control decision of the driver loop (word " + std::to_string(i) + ": rank " + std::to_string(r) +
                    " has " + std::to_string(got[i * e->nranks + r]) + ", rank " + std::to_string(e->rank) + " has " +
                    std::to_string(words[i]) + "): inputs or environment differ between the ranks");
  return 0;
}

int result_fetch(E* e, size_t count) {
  if (has_comm(e)) {
    size_t total = count;
    CHK(allreduce_with_agreement(e, count, &total));
    HIPCHK(hipMemcpyAsync(e->gram_host, e->gram_dev, sizeof(double) * total, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return agreement_verify(e, count);
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  e->agree_words.clear();
  return 0;
}

// result left in e->gram_host (p x q, ld = p) after the call
int gram_impl(E* e, const double* P, int p, const double* Q, int q) {
  if ((size_t)p * q > e->gram_doubles) return fail("gram result exceeds engine capacity");
  if (gram_scratch_doubles(p, q, e->nloc_pad) > e->scratch_doubles) return fail("gram scratch too small");
  int slot;
  CHK(timed_begin(e, 1, 0, &slot));
  launch_gram(e->stream, P, e->ldp, p, Q, e->ldp, q, e->nloc_pad, e->scratch, result_target(e), e->counters, e->tune.gram_wgs);
  CHK(timed_end(e, slot));
  if (e->nranks > 1) CHK(need_comm(e));
  CHK(result_fetch(e, (size_t)p * q));
  return 0;
}

extern "C" int dav_gram(dav_handle_t e, int panel_p, int p0, int p, int panel_q, int q0, int q, double* out, int64_t ldo) {
  CHK(bind(e));
  CHK(check_panel(e, panel_p, p0, p));
  CHK(check_panel(e, panel_q, q0, q));
  if (p <= 0 || q <= 0 || ldo < p) return fail("dav_gram: bad shape");
  CHK(gram_impl(e, panel_ptr(e, panel_p, p0), p, panel_ptr(e, panel_q, q0), q));
  for (int j = 0; j < q; ++j) std::memcpy(out + j * ldo, e->gram_host + (size_t)j * p, sizeof(double) * p);
  return 0;
}

extern "C" int dav_project(dav_handle_t e, int c0, int k, double* H, int64_t ldh, double* S, int64_t lds) {
  CHK(bind(e));
  const int h0_take = e->h0_take;
  int mt = c0 + k;
  CHK(check_panel(e, DAV_PANEL_V, 0, mt));
  if (k <= 0 || (H && ldh < mt)) return fail("dav_project: bad shape");
  const bool both = e->gev && (S != nullptr || (!H && e->rr_on));
  if (both && S && lds < mt) return fail("dav_project: bad shape");
  if (h0_take > 0 && c0 == 0 && k == h0_take && H && !e->rr_on) {
    // the projection of the unit columns dav_init_basis has just set up: it summed the operators' entries (idx_i, idx_j) over the
    // ranks inside the reduce-scatter of W0 - no Gram product, no collective here
    HIPCHK(hipStreamSynchronize(e->stream));
    const size_t h0_blk = (size_t)e->h0_cap * e->h0_cap;
    for (int pass = 0; pass < (both ? 2 : 1); ++pass) {
      double* out = pass == 0 ? H : S;
      const int64_t ld = pass == 0 ? ldh : lds;
      for (int j = 0; j < k; ++j)
        for (int i = 0; i < k; ++i)
          out[j * ld + i] = e->h0_kind[pass] == 2 ? (i == j ? 1.0 : 0.0) : e->h0_host[pass * h0_blk + (size_t)j * k + i];
    }
    return 0;
  }
  const size_t blk = (size_t)mt * k;
  if ((both ? 2 : 1) * blk > e->gram_doubles) return fail("gram result exceeds engine capacity");
  if (gram_scratch_doubles(mt, k, e->nloc_pad) > e->scratch_doubles) return fail("gram scratch too small");
  // V^T W_new and (generalized) V^T (B V)_new: two Gram launches, ONE reduction/fetch of both blocks
  int slot;
  CHK(timed_begin(e, 1, 0, &slot));
  launch_gram(e->stream, panel_ptr(e, DAV_PANEL_V, 0), e->ldp, mt, panel_ptr(e, DAV_PANEL_W, c0), e->ldp, k, e->nloc_pad, e->scratch,
              result_target(e), e->counters, e->tune.gram_wgs);
  if (both)
    launch_gram(e->stream, panel_ptr(e, DAV_PANEL_V, 0), e->ldp, mt, panel_ptr(e, DAV_PANEL_BV, c0), e->ldp, k, e->nloc_pad,
                e->scratch, result_target(e) + blk, e->counters, e->tune.gram_wgs);
  CHK(timed_end(e, slot));
  if (e->nranks > 1) CHK(need_comm(e));
  if (e->rr_on) {
    // device-resident Rayleigh-Ritz: the new columns also go into the projected matrices kept in HBM; a caller that
    // passes H = NULL (the device-RR driver) gets no host copy and no synchronisation at all
    if (mt > e->rr_ld) return fail("dav_project: basis wider than the device-resident projected matrices");
    if (has_comm(e)) CHK(coll_allreduce(e, e->gram_dev, (both ? 2 : 1) * blk));
    launch_rr_scatter(e->stream, result_target(e), mt, k, c0, e->rr_H, e->rr_ld);
    if (both) launch_rr_scatter(e->stream, result_target(e) + blk, mt, k, c0, e->rr_S, e->rr_ld);
    if (!H) return 0;
    if (has_comm(e)) HIPCHK(hipMemcpyAsync(e->gram_host, e->gram_dev, sizeof(double) * (both ? 2 : 1) * blk, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  } else {
    if (!H) return fail("dav_project: H is NULL (only with dav_rr_enable)");
    CHK(result_fetch(e, (both ? 2 : 1) * blk));
  }
  for (int pass = 0; pass < (both ? 2 : 1); ++pass) {
    double* out = pass == 0 ? H : S;
    int64_t ld = pass == 0 ? ldh : lds;
    const double* res = e->gram_host + pass * blk;
    for (int j = 0; j < k; ++j)
      for (int i = 0; i < mt; ++i) {
        double v = res[(size_t)j * mt + i];
        out[(c0 + j) * ld + i] = v;
        if (i < c0) out[i * ld + (c0 + j)] = v;       // mirror: the projected matrices are symmetric
      }
  }
  return 0;
}

// Projection of the new block AND the Gram blocks of its last orthonormalisation pass in ONE reduction / fetch (round 5).
// The driver sweeps the block as it stands after the FIRST Gram-Schmidt pass (T', orthonormal to ~1e-8), W' = A T' (B T'), and
// asks here for  [V T']^T W'  ((m + k) x k; generalized: [V T']^T (B T') too)  together with  C2 = V^T T',  G2 = T'^T T'.  The
// second pass is linear - T'' = (T' - V C2) M2 carries W'' = (W' - W C2) M2 along (dav_ortho_apply_all) - so the projected blocks
// of T'' follow on the host from these numbers and the H it already has.  One collective and one round trip per outer iteration
// fewer than dav_ortho_gram + dav_project.  H_raw / S_raw: (m + k) x k, column-major.
extern "C" int dav_project_ortho(dav_handle_t e, int m, int k, double* H_raw, int64_t ldh, double* S_raw, int64_t lds, double* C, int64_t ldc,
                                 double* G, int64_t ldg) {
  CHK(bind(e));
  const int p = m + k;
  CHK(check_panel(e, DAV_PANEL_V, 0, p));
  if (m < 0 || k <= 0 || !H_raw || ldh < p || !C || !G || ldg < k || (m > 0 && ldc < m)) return fail("dav_project_ortho: bad shape");
  const bool both = e->gev && S_raw != nullptr;
  if (both && lds < p) return fail("dav_project_ortho: bad shape");
  if (e->rr_on) return fail("dav_project_ortho: not with the device-resident Rayleigh-Ritz (dav_project_dev)");
  const size_t blk = (size_t)p * k;
  const int nblk = both ? 3 : 2;
  if (nblk * blk > e->gram_doubles) return fail("gram result exceeds engine capacity");
  if (gram_scratch_doubles(p, k, e->nloc_pad) > e->scratch_doubles) return fail("gram scratch too small");
  int slot;
  CHK(timed_begin(e, 1, 0, &slot));
  const double* Vp = panel_ptr(e, DAV_PANEL_V, 0);
  const double* Qs[3] = {panel_ptr(e, DAV_PANEL_W, m), panel_ptr(e, DAV_PANEL_V, m), both ? panel_ptr(e, DAV_PANEL_BV, m) : nullptr};
  if (gram_scratch_doubles(p, nblk * k, e->nloc_pad) <= e->scratch_doubles) {
    // ONE launch and one reduction for all blocks: the right-hand side's columns come from the three panels
    launch_gram_multi(e->stream, Vp, e->ldp, p, Qs, nblk, k, e->ldp, e->nloc_pad, e->scratch, result_target(e), e->counters, e->tune.gram_wgs);
  } else {
    for (int b = 0; b < nblk; ++b)
      launch_gram(e->stream, Vp, e->ldp, p, Qs[b], e->ldp, k, e->nloc_pad, e->scratch, result_target(e) + b * blk, e->counters, e->tune.gram_wgs);
  }
  CHK(timed_end(e, slot));
  if (e->nranks > 1) CHK(need_comm(e));
  CHK(result_fetch(e, nblk * blk));
  const double* gh = e->gram_host;
  for (int j = 0; j < k; ++j) {
    std::memcpy(H_raw + (size_t)j * ldh, gh + (size_t)j * p, sizeof(double) * p);
    for (int i = 0; i < m; ++i) C[(size_t)j * ldc + i] = gh[blk + (size_t)j * p + i];
    for (int i = 0; i < k; ++i) G[(size_t)j * ldg + i] = gh[blk + (size_t)j * p + m + i];
    if (both) std::memcpy(S_raw + (size_t)j * lds, gh + 2 * blk + (size_t)j * p, sizeof(double) * p);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- K6 -----------------------------------------------------------------------------------------
extern "C" int dav_init_basis(dav_handle_t e, int ncols, int64_t* idx_out) {
  CHK(bind(e));
  if (ncols <= 0 || ncols > e->max_cols || ncols > e->n) return fail("dav_init_basis: bad column count");
  const std::vector<double>& d = e->diag_host[DAV_OP_A];
  if (d.empty()) return fail("dav_init_basis: operator A not set");
  // stable selection of the ncols smallest diagonal entries (ties -> lower index first); a property of the
  // resident operator, so it is kept until the diagonal changes (repeated solves on one engine)
  if ((int)e->basis_order.size() < ncols) {
    std::vector<int64_t> all((size_t)e->n);
    std::iota(all.begin(), all.end(), 0);
    int keep = (int)std::min<int64_t>(e->n, std::max(ncols, e->max_cols));
    std::partial_sort(all.begin(), all.begin() + keep, all.end(),
                      [&](int64_t a, int64_t b) { return d[a] < d[b] || (d[a] == d[b] && a < b); });
    all.resize(keep);
    e->basis_order.swap(all);
  }
  // the tickets of the last-workgroup finishes start every solve from zero (a launch that died half-way must not leave a count behind)
  HIPCHK(hipMemsetAsync(e->counters, 0, sizeof(unsigned) * (GRAM_MAX_COUNTERS + 8), e->stream));
  std::vector<int64_t> order(e->basis_order.begin(), e->basis_order.begin() + ncols);
  HIPCHK(hipMemcpyAsync(e->idx_dev, order.data(), sizeof(int64_t) * ncols, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  launch_unit_columns(e->stream, e->idx_dev, ncols, e->row0, e->nloc, e->nloc_pad, panel_ptr(e, DAV_PANEL_V, 0), e->ldp);
  // H0 = V0^T (Op V0) of unit columns is the operator's entries (idx_i, idx_j): read / generated here instead of a Gram product over
  // N rows in dav_project; several ranks of dealt-out tiles sum what each holds of them inside the reduce-scatter group of W0
  // (h0_dev in engine_internal.h), a generated operator's entries every rank generates for itself - no collective for H0 either way
  const bool h0_try = !e->rr_on && ncols <= e->h0_cap && e->tune.no_h0 == 0;
  const size_t h0_blk = (size_t)e->h0_cap * e->h0_cap;
  e->h0_kind[0] = e->h0_kind[1] = 0;
  for (int w = 0; w < (e->gev ? 2 : 1); ++w) {
    OpDesc& o = e->op[w];
    int dst = w == 0 ? DAV_PANEL_W : DAV_PANEL_BV;
    double* h0 = h0_try ? e->h0_dev + w * h0_blk : nullptr;
    bool stashed = false;
    if (o.kind == DAV_KIND_DENSE && o.storage == 1 && e->nranks == 1) {
      launch_gather_columns_sym(e->stream, o.a, e->sym.row_off, e->n, e->nloc_pad, e->idx_dev, ncols, panel_ptr(e, dst, 0), e->ldp);
      if (h0) { launch_entries_sym(e->stream, o.a, e->sym.row_off, e->idx_dev, ncols, h0); stashed = true; }
    } else if (o.kind == DAV_KIND_DENSE && o.storage == 1) {
      CHK(gather_columns_sym_multi(e, o, ncols, panel_ptr(e, dst, 0), h0));       // several ranks: one reduce-scatter instead of a sweep
      stashed = h0 != nullptr;
    } else if (o.kind == DAV_KIND_DENSE && o.storage == 0) {
      launch_gather_columns(e->stream, o.a, e->nloc_pad, e->nloc_pad, e->idx_dev, ncols, panel_ptr(e, dst, 0), e->ldp);
      if (h0 && e->nranks == 1) { launch_entries_dense(e->stream, o.a, e->nloc_pad, e->idx_dev, ncols, h0); stashed = true; }
    } else if (o.kind == DAV_KIND_HASHED || o.kind == DAV_KIND_HARNESS || o.kind == DAV_KIND_IDENTITY) {
      // a generated operator's columns are generated: N x ncols entries instead of a sweep of N^2 / 2 (configs[4]: one sweep in five)
      launch_gather_columns_free(e->stream, op_params(o), e->row0, e->nloc, e->nloc_pad, e->idx_dev, ncols, panel_ptr(e, dst, 0), e->ldp);
      if (o.kind == DAV_KIND_IDENTITY) e->h0_kind[w] = 2;                          // V0^T I V0 = I (distinct unit columns)
      else if (h0) { launch_entries_free(e->stream, op_params(o), e->idx_dev, ncols, h0); stashed = true; }
    }
    else if (o.kind == DAV_KIND_HOST) {
      /* the driver fills W / BV through dav_panel_put */
    } else
      CHK(apply_impl(e, w, DAV_PANEL_V, 0, ncols, dst, 0, false));
    if (stashed) {
      HIPCHK(hipMemcpyAsync(e->h0_host + w * h0_blk, h0, sizeof(double) * (size_t)ncols * ncols, hipMemcpyDeviceToHost, e->stream));
      e->h0_kind[w] = 1;
    }
  }
  e->m = ncols;
  if (idx_out)
    for (int i = 0; i < ncols; ++i) idx_out[i] = order[i] + 1;
  HIPCHK(hipGetLastError());
  // valid for the call that follows: H0 from the stash, and for a generalized problem S0 from the stash or the identity
  if (h0_try && e->h0_kind[0] == 1 && (!e->gev || e->h0_kind[1] != 0)) e->h0_cols = ncols;
  return 0;
}

// ---- K3 -----------------------------------------------------------------------------------------

extern "C" int dav_ritz_residual_correction_n(dav_handle_t e, int m, int ncorr, int lowest, const double* Y, int64_t ldy,
                                              const double* theta, int method, double* resnorm) {
  return ritz_impl(e, m, ncorr, lowest, Y, ldy, theta, method, resnorm, nullptr, 0, nullptr, 0, nullptr, nullptr);
}

extern "C" int dav_ritz_residual_correction_g(dav_handle_t e, int m, int ncorr, int lowest, const double* Y, int64_t ldy,
                                              const double* theta, double* resnorm, double* C, int64_t ldc, double* G,
                                              int64_t ldg) {
  if (!C || !G || ldc < m || ldg < ncorr) return fail("dav_ritz_residual_correction_g: bad shape");
  return ritz_impl(e, m, ncorr, lowest, Y, ldy, theta, DAV_METHOD_DPR, resnorm, C, ldc, G, ldg, nullptr, nullptr);
}

// Y == nullptr: the eigenpairs are the device-resident ones of dav_rr_ritz (theta_out receives all m Ritz values)
int ritz_impl(E* e, int m, int ncorr, int lowest, const double* Y, int64_t ldy, const double* theta, int method,
                     double* resnorm, double* C, int64_t ldc, double* G, int64_t ldg, double* theta_out,
                     double* info_out) {
  CHK(bind(e));
  const bool dev = Y == nullptr;
  if (m <= 0 || lowest <= 0 || lowest > ncorr || ncorr > m || (!dev && ldy < m)) return fail("dav_ritz_residual_correction: bad shape");
  if (method == DAV_METHOD_DPR && m + ncorr > e->cols_alloc) return fail("basis panel too narrow for the correction block");
  CHK(check_panel(e, DAV_PANEL_V, 0, m));
  const double *dY, *dY2, *dTheta;
  int64_t ldm_y, ldm_y2;
  std::vector<double> y2;
  if (dev) {
    dY = e->rr_Ypk; dY2 = e->rr_Y2pk; dTheta = e->rr_thpk;
    ldm_y = ldm_y2 = e->rr_tp;                 // tiles per step of the images launch_rr_pack made
  } else {
    y2.resize((size_t)m * ncorr);
    for (int j = 0; j < ncorr; ++j)
      for (int i = 0; i < m; ++i) y2[(size_t)j * m + i] = -Y[j * ldy + i] * theta[j];
    SmallMat sm3[3] = {{Y, ldy, m, ncorr, nullptr, 0, true}, {y2.data(), m, m, ncorr, nullptr, 0, true}, {theta, ncorr, ncorr, 1, nullptr, 0, false}};
    CHK(small_upload_multi(e, 0, sm3, 3));
    dY = sm3[0].dev; dY2 = sm3[1].dev; dTheta = sm3[2].dev;
    ldm_y = sm3[0].ldm; ldm_y2 = sm3[1].ldm;
  }

  int slot;
  CHK(timed_begin(e, 2, 0, &slot));
  // X = V * Y(:, 1:nx) - unless the caller asked for the Ritz vectors only at the end (dav_set_lazy_ritz_vectors) and the
  // correction does not read them (DPR, or no correction at all)
  int nx = method == DAV_METHOD_GJD ? ncorr : lowest;
  if (!(e->lazy_x && method != DAV_METHOD_GJD && !dev)) {
    PanelGemmArgs a{};
    a.P1 = panel_ptr(e, DAV_PANEL_V, 0); a.ld1 = e->ldp; a.p1 = m; a.M1 = dY; a.tp1 = ldm_y;
    a.p2 = 0;
    a.out = panel_ptr(e, DAV_PANEL_X, 0); a.ldo = e->ldp; a.q = nx;
    a.nloc = e->nloc; a.nrows_pad = e->nloc_pad; a.epilogue = 0;
    a.pin = e->tune.pg_pin;
    launch_panel_gemm(e->stream, a);
  }
  // R = W*Y + Z*(-Y*diag(theta)), norms, (DPR) T
  PanelGemmArgs r{};
  r.P1 = panel_ptr(e, DAV_PANEL_W, 0); r.ld1 = e->ldp; r.p1 = m; r.M1 = dY; r.tp1 = ldm_y;
  r.P2 = panel_ptr(e, e->gev ? DAV_PANEL_BV : DAV_PANEL_V, 0); r.ld2 = e->ldp; r.p2 = m; r.M2 = dY2; r.tp2 = ldm_y2;
  r.q = ncorr; r.nloc = e->nloc; r.nrows_pad = e->nloc_pad;
  r.theta = dTheta; r.dA = e->op[DAV_OP_A].diag; r.dB = e->gev ? e->op[DAV_OP_B].diag : nullptr;
  r.nnorm = lowest; r.norm_partial = e->norm_partial;
  const bool fuse_norms = e->nloc_pad / PG_ROWS <= PG_FUSE_BLOCKS;     // small grids: the last workgroup sums the partial norms itself
  if (fuse_norms) { r.norm_out = result_target(e); r.counter = e->counters + GRAM_MAX_COUNTERS; }
  if (method == DAV_METHOD_DPR) {
    r.out = panel_ptr(e, DAV_PANEL_V, m); r.ldo = e->ldp; r.epilogue = 1;
  } else {
    r.out = panel_ptr(e, DAV_PANEL_R, 0); r.ldo = e->ldp; r.epilogue = 2;
  }
  r.pin = e->tune.pg_pin;
  launch_panel_gemm(e->stream, r);
  if (!fuse_norms) launch_norm_finish(e->stream, e->norm_partial, (int)(e->nloc_pad / PG_ROWS), lowest, result_target(e));
  // optionally the Gram block the first orthonormalisation pass needs, [V T]^T T with T = V[:, m:m+ncorr] just
  // written: it rides on the same reduction and the same fetch as the norms (one synchronisation less)
  size_t count = (size_t)lowest;
  const size_t goff = ((size_t)lowest + 7) / 8 * 8;
  const int p = m + ncorr;
  if (C) {
    if (goff + (size_t)p * ncorr > e->gram_doubles) return fail("gram result exceeds engine capacity");
    if (gram_scratch_doubles(p, ncorr, e->nloc_pad) > e->scratch_doubles) return fail("gram scratch too small");
    launch_gram(e->stream, panel_ptr(e, DAV_PANEL_V, 0), e->ldp, p, panel_ptr(e, DAV_PANEL_V, m), e->ldp, ncorr, e->nloc_pad,
                e->scratch, result_target(e) + goff, e->counters, e->tune.gram_wgs);
    count = goff + (size_t)p * ncorr;
  }
  CHK(timed_end(e, slot));
  if (e->nranks > 1) CHK(need_comm(e));
  if (dev) {
    // the Ritz values (and the eigensolver's status word) ride on the same fetch, behind the all-reduced part
    const size_t toff = (count + 7) / 8 * 8;
    if (toff + (size_t)roundup(m + 1, 2) > e->gram_doubles) return fail("gram result exceeds engine capacity");   // the tail copy moves whole pairs
    if (has_comm(e)) CHK(coll_allreduce(e, e->gram_dev, count));
    launch_copy_columns(e->stream, e->rr_thpk + roundup(ncorr, 64), 2 * (int64_t)roundup(m + 1, 2), result_target(e) + toff,
                        2 * (int64_t)roundup(m + 1, 2), roundup(m + 1, 2), 1);
    if (has_comm(e)) HIPCHK(hipMemcpyAsync(e->gram_host, e->gram_dev, sizeof(double) * (toff + m + 1), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int j = 0; j < m; ++j) theta_out[j] = e->gram_host[toff + j];
    if (info_out) *info_out = e->gram_host[toff + m];
  } else {
    CHK(result_fetch(e, count));
  }
  for (int j = 0; j < lowest; ++j) resnorm[j] = std::sqrt(e->gram_host[j]);
  if (C) {
    const double* gh = e->gram_host + goff;
    for (int j = 0; j < ncorr; ++j) {
      for (int i = 0; i < m; ++i) C[j * ldc + i] = gh[(size_t)j * p + i];
      for (int i = 0; i < ncorr; ++i) G[j * ldg + i] = gh[(size_t)j * p + m + i];
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int dav_set_lazy_ritz_vectors(dav_handle_t e, int on) {
  e->lazy_x = on != 0;
  return 0;
}

extern "C" int dav_ritz_vectors(dav_handle_t e, int m, int nx, const double* Y, int64_t ldy) {
  if (m <= 0 || nx <= 0 || nx > m || ldy < m) return fail("dav_ritz_vectors: bad shape");
  return dav_panel_transform(e, DAV_PANEL_V, 0, m, Y, ldy, nx, DAV_PANEL_X, 0);
}

extern "C" int dav_ritz_residual_correction(dav_handle_t e, int m, int lowest, const double* Y, int64_t ldy,
                                            const double* theta, int method, double* resnorm) {
  return dav_ritz_residual_correction_n(e, m, m, lowest, Y, ldy, theta, method, resnorm);
}

extern "C" int dav_panel_select(dav_handle_t e, int panel, int c0, int nsel, const int* sel) {
  CHK(bind(e));
  if (nsel < 0 || (nsel > 0 && !sel)) return fail("dav_panel_select: bad arguments");
  for (int i = 0; i < nsel; ++i) {
    if (sel[i] < i || (i > 0 && sel[i] <= sel[i - 1])) return fail("dav_panel_select: indices must be ascending");
    CHK(check_panel(e, panel, c0 + sel[i], 1));
    if (sel[i] != i)      // columns only move to the left, in ascending order: no overlap
      launch_copy_columns(e->stream, panel_ptr(e, panel, c0 + sel[i]), e->ldp, panel_ptr(e, panel, c0 + i), e->ldp, e->nloc_pad, 1);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- K4 -----------------------------------------------------------------------------------------
extern "C" int dav_ortho_gram(dav_handle_t e, int m, int kt, double* C, int64_t ldc, double* G, int64_t ldg) {
  CHK(bind(e));
  if (m < 0 || kt <= 0 || ldg < kt || (m > 0 && ldc < m)) return fail("dav_ortho_gram: bad shape");
  CHK(check_panel(e, DAV_PANEL_V, 0, m + kt));
  int p = m + kt;
  CHK(gram_impl(e, panel_ptr(e, DAV_PANEL_V, 0), p, panel_ptr(e, DAV_PANEL_V, m), kt));
  for (int j = 0; j < kt; ++j) {
    for (int i = 0; i < m; ++i) C[j * ldc + i] = e->gram_host[(size_t)j * p + i];
    for (int i = 0; i < kt; ++i) G[j * ldg + i] = e->gram_host[(size_t)j * p + m + i];
  }
  return 0;
}

// T <- (T - V C) M on the basis panel; with_images: the same transform on W = A V (and B V), whose columns m.. hold the images of T
static int ortho_apply_impl(E* e, int m, int kt, const double* C, int64_t ldc, const double* M, int64_t ldm, bool with_images) {
  CHK(bind(e));
  if (m < 0 || kt <= 0 || ldm < kt) return fail("dav_ortho_apply: bad shape");
  CHK(check_panel(e, DAV_PANEL_V, 0, m + kt));
  std::vector<double> cm((size_t)std::max(m, 1) * kt, 0.0);       // -(C*M)
  for (int j = 0; j < kt && m > 0; ++j)
    for (int l = 0; l < kt; ++l) {
      double mlj = M[j * ldm + l];
      if (mlj == 0.0) continue;
      for (int i = 0; i < m; ++i) cm[(size_t)j * m + i] -= C[l * ldc + i] * mlj;
    }
  SmallMat sm2[2] = {{M, ldm, kt, kt, nullptr, 0, true}, {cm.data(), std::max(m, 1), m, kt, nullptr, 0, true}};
  CHK(small_upload_multi(e, 1, sm2, m > 0 ? 2 : 1));
  const int64_t ld_m = sm2[0].ldm, ld_cm = m > 0 ? sm2[1].ldm : 4;
  int slot;
  CHK(timed_begin(e, 2, 0, &slot));
  const int panels[3] = {DAV_PANEL_V, DAV_PANEL_W, DAV_PANEL_BV};
  const int npanels = with_images ? (e->gev ? 3 : 2) : 1;
  // the basis panel and its images are neighbours in the engine's arena: one batched launch (blockIdx.z = panel) where the
  // product runs in place
  const int64_t pstride = e->panel[DAV_PANEL_W] - e->panel[DAV_PANEL_V];
  const bool batched = npanels > 1 && kt <= PG_INPLACE_COLS && (!e->gev || e->panel[DAV_PANEL_BV] - e->panel[DAV_PANEL_W] == pstride);
  for (int i = 0; i < (batched ? 1 : npanels); ++i) {
    PanelGemmArgs a{};
    if (batched) { a.batch = npanels; a.batch_stride = pstride; }
    a.P1 = panel_ptr(e, panels[i], m); a.ld1 = e->ldp; a.p1 = kt; a.M1 = sm2[0].dev; a.tp1 = ld_m;
    a.P2 = panel_ptr(e, panels[i], 0); a.ld2 = e->ldp; a.p2 = m; a.M2 = sm2[1].dev; a.tp2 = ld_cm;
    // in place where one workgroup covers all kt output columns (k_panel.hip: a wave has read its rows of every input column before
    // it stores the first output); wider blocks go through the scratch panel
    const bool in_place = kt <= PG_INPLACE_COLS;
    a.out = in_place ? panel_ptr(e, panels[i], m) : panel_ptr(e, DAV_PANEL_S, 0); a.ldo = e->ldp; a.q = kt;
    a.nloc = e->nloc; a.nrows_pad = e->nloc_pad; a.epilogue = 0;
    a.pin = e->tune.pg_pin;
    launch_panel_gemm(e->stream, a);
    if (!in_place) launch_copy_columns(e->stream, panel_ptr(e, DAV_PANEL_S, 0), e->ldp, panel_ptr(e, panels[i], m), e->ldp, e->nloc_pad, kt);
  }
  CHK(timed_end(e, slot));
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int dav_ortho_apply(dav_handle_t e, int m, int kt, const double* C, int64_t ldc, const double* M, int64_t ldm) {
  return ortho_apply_impl(e, m, kt, C, ldc, M, ldm, false);
}
// ... for a block that has been swept already (dav_project_ortho): W[:, m:m+kt] = A T and (generalized) BV[:, m:m+kt] = B T follow T
extern "C" int dav_ortho_apply_all(dav_handle_t e, int m, int kt, const double* C, int64_t ldc, const double* M, int64_t ldm) {
  return ortho_apply_impl(e, m, kt, C, ldc, M, ldm, true);
}

extern "C" int dav_expand(dav_handle_t e, int m, int kt) {
  CHK(bind(e));
  if (m < 0 || kt <= 0 || m + kt > e->cols_alloc) return fail("dav_expand: bad shape");
  for (int w = 0; w < (e->gev ? 2 : 1); ++w) {
    if (e->op[w].kind == DAV_KIND_HOST) continue;     // driver moves the block through the host callback
    CHK(apply_impl(e, w, DAV_PANEL_V, m, kt, w == 0 ? DAV_PANEL_W : DAV_PANEL_BV, m, true));
  }
  e->m = m + kt;
  return 0;
}

// ---- K5 -----------------------------------------------------------------------------------------
extern "C" int dav_panel_transform(dav_handle_t e, int src_panel, int s0, int p, const double* M, int64_t ldm, int q,
                                   int dst_panel, int d0) {
  CHK(bind(e));
  CHK(check_panel(e, src_panel, s0, p));
  CHK(check_panel(e, dst_panel, d0, q));
  if (p <= 0 || q <= 0 || ldm < p) return fail("dav_panel_transform: bad shape");
  if (q > e->cols_alloc) return fail("dav_panel_transform: too many output columns");
  int64_t ld_m;
  CHK(small_upload_image(e, 3, M, ldm, p, q, &ld_m));
  int slot;
  CHK(timed_begin(e, 2, 0, &slot));
  PanelGemmArgs a{};
  a.P1 = panel_ptr(e, src_panel, s0); a.ld1 = e->ldp; a.p1 = p; a.M1 = e->sm[3].dev; a.tp1 = ld_m;
  a.p2 = 0;
  a.nloc = e->nloc; a.nrows_pad = e->nloc_pad; a.epilogue = 0; a.q = q; a.ldo = e->ldp;
  // source and destination in one panel: in place when the column ranges coincide and one workgroup covers all q output columns
  // (a wave reads its rows of all p input columns before it stores), through the scratch panel otherwise
  const bool same = (src_panel == dst_panel);
  const bool in_place = same && s0 == d0 && q <= p && q <= PG_INPLACE_COLS;
  bool overlap = same && !in_place;
  a.out = overlap ? panel_ptr(e, DAV_PANEL_S, 0) : panel_ptr(e, dst_panel, d0);
  if (overlap && src_panel == DAV_PANEL_S) return fail("dav_panel_transform: scratch panel cannot be transformed in place");
  a.pin = e->tune.pg_pin;
  launch_panel_gemm(e->stream, a);
  if (overlap) launch_copy_columns(e->stream, panel_ptr(e, DAV_PANEL_S, 0), e->ldp, panel_ptr(e, dst_panel, d0), e->ldp, e->nloc_pad, q);
  CHK(timed_end(e, slot));
  HIPCHK(hipGetLastError());
  return 0;
}

// V, W = A V and B V are contracted with the same keep columns (src/davidson.f90:218 contracts V and then re-applies the
// operators to the whole basis, :223-226; W Y = A (V Y) holds to rounding, so no sweep of A or B follows a restart)
// Mdev: operand image of the m x keep transform, ldm = its tiles per step
int restart_contract(E* e, int m, int keep, const double* Mdev, int64_t ldm) {
  int slot;
  CHK(timed_begin(e, 2, 0, &slot));
  const int panels[3] = {DAV_PANEL_V, DAV_PANEL_W, DAV_PANEL_BV};
  for (int i = 0; i < (e->gev ? 3 : 2); ++i) {
    PanelGemmArgs a{};
    a.P1 = panel_ptr(e, panels[i], 0); a.ld1 = e->ldp; a.p1 = m; a.M1 = Mdev; a.tp1 = ldm;
    a.p2 = 0;
    a.nloc = e->nloc; a.nrows_pad = e->nloc_pad; a.epilogue = 0; a.q = keep; a.ldo = e->ldp;
    const bool in_place = keep <= PG_INPLACE_COLS;      // the kept columns overwrite the leading columns of the panel they are made from
    a.out = in_place ? panel_ptr(e, panels[i], 0) : panel_ptr(e, DAV_PANEL_S, 0);
    a.pin = e->tune.pg_pin;
  launch_panel_gemm(e->stream, a);
    if (!in_place) launch_copy_columns(e->stream, panel_ptr(e, DAV_PANEL_S, 0), e->ldp, panel_ptr(e, panels[i], 0), e->ldp, e->nloc_pad, keep);
  }
  CHK(timed_end(e, slot));
  e->m = keep;
  e->st.restarts += 1;
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int dav_restart(dav_handle_t e, int m, int keep, const double* Yk, int64_t ldy) {
  CHK(bind(e));
  if (keep <= 0 || keep > m || m > e->cols_alloc || ldy < m) return fail("dav_restart: bad shape");
  int64_t ld_m;
  CHK(small_upload_image(e, 3, Yk, ldy, m, keep, &ld_m));
  return restart_contract(e, m, keep, e->sm[3].dev, ld_m);
}

// Several ranks: every rank takes the driver's control decisions (converged? grow or restart? how many columns?) from
// all-reduced small results, so they are identical by construction.  This makes that an enforced invariant instead of an
// assumption: the words (iteration number, basis width, decisions) are all-reduced as max and as -min in one collective;
// a rank that sees them differ returns an error - its process ends with a message, and the launcher tears the group down -
// instead of walking into the next collective alone and hanging everybody.  One tiny all-reduce per outer iteration.
// ---- device-resident Rayleigh-Ritz -----------------------------------------------------------------
extern "C" int dav_rr_enable(dav_handle_t e, int on) {
  CHK(bind(e));
  if (on && !e->rr_H) {
    // the device eigensolver handles projected problems of order <= 128 (+ one expansion block on top); an engine created
    // for a wider basis can still run narrower solves with it: the device-resident matrices are sized to what it can use
    e->rr_ld = std::min<int64_t>(e->cols_alloc, 160);
    const size_t sq = (size_t)e->rr_ld * e->rr_ld, pk = (size_t)roundup(e->rr_ld, 4) * roundup(e->rr_ld, 64);
    HIPCHK(pool_malloc(&e->rr_H, sizeof(double) * sq));
    HIPCHK(pool_malloc(&e->rr_S, sizeof(double) * sq));
    HIPCHK(pool_malloc(&e->rr_Y, sizeof(double) * sq));
    HIPCHK(pool_malloc(&e->rr_theta, sizeof(double) * e->rr_ld));
    HIPCHK(pool_malloc(&e->rr_work, sizeof(double) * small_eig_work_doubles((int)e->rr_ld)));
    HIPCHK(pool_malloc(&e->rr_info, sizeof(double) * 8));
    HIPCHK(pool_malloc(&e->rr_Ypk, sizeof(double) * pk));
    HIPCHK(pool_malloc(&e->rr_Y2pk, sizeof(double) * pk));
    HIPCHK(pool_malloc(&e->rr_thpk, sizeof(double) * (roundup(e->rr_ld, 64) + e->rr_ld + 8)));
    HIPCHK(hipMemsetAsync(e->rr_H, 0, sizeof(double) * sq, e->stream));
    HIPCHK(hipMemsetAsync(e->rr_S, 0, sizeof(double) * sq, e->stream));
  }
  e->rr_on = on != 0;
  return 0;
}

// dav_project without a host copy of the new block and without a synchronisation (device-resident Rayleigh-Ritz only)
extern "C" int dav_project_dev(dav_handle_t e, int c0, int k) {
  if (!e->rr_on) return fail("dav_project_dev: call dav_rr_enable first");
  return dav_project(e, c0, k, nullptr, 0, nullptr, 0);
}

// Rayleigh-Ritz on the device-resident projected matrices (filled by dav_project) followed by the Ritz phase of
// dav_ritz_residual_correction_n / _g from the eigenpairs where they lie: ONE host synchronisation returns all m Ritz
// values, the residual norms of the first `lowest` pairs and (C != NULL) the Gram blocks of the first
// orthonormalisation pass.  Replaces lapack_generalized_eigensolver (src/lapack_wrapper.f90:14-91) + the H-down /
// Y-up transfers.  sweeps_out: Jacobi sweeps used.
extern "C" int dav_rr_ritz(dav_handle_t e, int m, int ncorr, int lowest, int method, double* theta_out, double* resnorm, double* C,
                           int64_t ldc, double* G, int64_t ldg, int* sweeps_out) {
  CHK(bind(e));
  if (!e->rr_on) return fail("dav_rr_ritz: call dav_rr_enable first");
  if (m <= 0 || m > 128 || m > e->rr_ld || !theta_out || !resnorm) return fail("dav_rr_ritz: bad arguments (order <= 128)");
  // checked BEFORE the eigensolver and the operand packing are launched: they index the device-resident arrays with these
  if (lowest <= 0 || lowest > m || ncorr < 0 || ncorr > m) return fail("dav_rr_ritz: bad shape (0 < lowest <= m, 0 <= ncorr <= m)");
  if (C && (!G || ldc < m || ldg < ncorr)) return fail("dav_rr_ritz: bad shape");
  if (!e->agree_words.empty()) {
    // the device-resident Ritz phase keeps its own layout behind the all-reduced part of its result: the driver's control words get
    // a collective of their own here, BEFORE the kernels below write their partial results (dav_ranks_agree stages in the same buffer)
    std::vector<double> w;
    w.swap(e->agree_words);
    CHK(dav_ranks_agree(e, w.data(), (int)w.size()));
  }
  int slot;
  CHK(timed_begin(e, 1, 0, &slot));
  if (!launch_small_eig(e->stream, e->rr_H, e->rr_ld, e->rr_S, e->rr_ld, m, e->gev != 0, e->rr_theta, e->rr_Y, e->rr_ld, e->rr_work, e->rr_info))
    return fail("dav_rr_ritz: order out of range");
  const int nq = method == DAV_METHOD_GJD ? ncorr : std::max(ncorr, lowest);
  launch_rr_pack(e->stream, e->rr_Y, e->rr_ld, e->rr_theta, m, nq, (int)roundup(m, 4), (int)roundup(nq, 64), e->rr_Ypk, e->rr_Y2pk, e->rr_thpk,
                 e->rr_info, e->rr_thpk + roundup(ncorr, 64));
  e->rr_tp = roundup(nq, 64) / 16;
  CHK(timed_end(e, slot));
  double info = 0.0;
  CHK(ritz_impl(e, m, ncorr, lowest, nullptr, 0, nullptr, method, resnorm, C, ldc, G, ldg, theta_out, &info));
  if (info < 0.0) return fail("dav_rr_ritz: the projected overlap matrix is not positive definite (pivot " + std::to_string((int)-info) + ")");
  if (sweeps_out) *sweeps_out = (int)info;
  return 0;
}

// collapse restart with the device-resident eigenvectors: V <- V * Y(:, 1:keep)   (src/davidson.f90:218)
extern "C" int dav_rr_restart(dav_handle_t e, int m, int keep) {
  CHK(bind(e));
  if (!e->rr_on || keep <= 0 || keep > m || m > e->rr_ld) return fail("dav_rr_restart: bad shape");
  launch_rr_pack(e->stream, e->rr_Y, e->rr_ld, e->rr_theta, m, keep, (int)roundup(m, 4), (int)roundup(keep, 64), e->rr_Ypk, e->rr_Y2pk,
                 e->rr_thpk, e->rr_info, nullptr);
  e->rr_tp = roundup(keep, 64) / 16;
  return restart_contract(e, m, keep, e->rr_Ypk, e->rr_tp);
}

// the device-resident eigenvectors (m x ncols) and Ritz values, for tests and for callers that want them on the host
extern "C" int dav_rr_get(dav_handle_t e, int m, int ncols, double* theta, double* Y, int64_t ldy) {
  CHK(bind(e));
  if (!e->rr_on || m <= 0 || m > e->rr_ld || ncols > m || ldy < m) return fail("dav_rr_get: bad shape");
  if (theta) HIPCHK(hipMemcpyAsync(theta, e->rr_theta, sizeof(double) * m, hipMemcpyDeviceToHost, e->stream));
  if (Y) HIPCHK(hipMemcpy2DAsync(Y, sizeof(double) * ldy, e->rr_Y, sizeof(double) * e->rr_ld, sizeof(double) * m, (size_t)ncols,
                                 hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}
