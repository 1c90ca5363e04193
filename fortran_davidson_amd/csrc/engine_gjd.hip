// Engine, K7: the GJD correction - block MINRES on the projected systems with the K1 sweep as the operator.
#include "engine_internal.h"

// ---- K7: GJD correction ------------------------------------------------------------------------------
// Solves, for all m Ritz pairs at once,  (I - x x^T)(A - theta_k B)(I - x x^T) t_k = -r_k
// (the systems compute_GJD_generalized_dense builds densely and hands to DSYSV,
// src/davidson.f90:719-732; x is used exactly as the reference uses it - not re-normalised, so in
// the generalized case I - x x^T is not a projector, :721) with a Jacobi-preconditioned MINRES whose
// operator is the K1 block matvec: every inner step costs one sweep of A (and one of B) shared by all
// m right-hand sides.  Inputs: X (Ritz vectors, m columns) and R (residues) as left by
// dav_ritz_residual_correction(..., DAV_METHOD_GJD).  Output: T in V[:, m:2m].
namespace {
struct Gjd {
  E* e;
  int m;
  int64_t ldc;
  std::vector<double> coef;      // 4 x ldc staging
  int coef_slot = 0;
};

static int gjd_coef(Gjd& g, const std::vector<double>* c0, const std::vector<double>* c1, const std::vector<double>* c2,
                    const std::vector<double>* c3, double** dev) {
  // round-robin over two small buffers so that an upload never waits for the kernel that reads the other
  E* e = g.e;
  const std::vector<double>* cs[4] = {c0, c1, c2, c3};
  std::vector<double> flat((size_t)g.m * 4, 0.0);
  for (int t = 0; t < 4; ++t)
    if (cs[t]) std::copy(cs[t]->begin(), cs[t]->end(), flat.begin() + (size_t)t * g.m);
  int slot = 2 + (g.coef_slot++ & 1);
  int64_t ldm;
  CHK(small_upload(e, slot, flat.data(), g.m, g.m, 4, &ldm));
  g.ldc = ldm;
  *dev = e->sm[slot].dev;
  return 0;
}

static int gjd_lincomb(Gjd& g, double* out, const double* a0, const std::vector<double>* c0, const double* a1 = nullptr,
                       const std::vector<double>* c1 = nullptr, const double* a2 = nullptr,
                       const std::vector<double>* c2 = nullptr, const double* a3 = nullptr,
                       const std::vector<double>* c3 = nullptr) {
  E* e = g.e;
  double* dev;
  CHK(gjd_coef(g, c0, c1, c2, c3, &dev));
  LincombArgs a{};
  a.in[0] = a0; a.in[1] = a1 ? a1 : a0; a.in[2] = a2 ? a2 : a0; a.in[3] = a3 ? a3 : a0;
  a.nterms = a3 ? 4 : (a2 ? 3 : (a1 ? 2 : 1));
  a.coef = dev; a.ldc = (int)g.ldc; a.out = out; a.ld = e->ldp; a.nrows_pad = e->nloc_pad; a.m = g.m;
  launch_lincomb(e->stream, a);
  return 0;
}

// up to 4 column-wise dot products, all-reduced, returned as res[s][j]
static int gjd_dots(Gjd& g, int npairs, const double* const* a, const double* const* b, std::vector<double>* res) {
  E* e = g.e;
  DotsArgs d{};
  for (int s = 0; s < npairs; ++s) { d.a[s] = a[s]; d.b[s] = b[s]; }
  d.npairs = npairs; d.ld = e->ldp; d.nrows_pad = e->nloc_pad; d.m = g.m; d.partial = e->norm_partial;
  int nb = coldots_blocks(e->nloc_pad);
  int total = npairs * g.m;
  launch_coldots(e->stream, d);
  launch_norm_finish(e->stream, e->norm_partial, nb, total, result_target(e));
  CHK(result_fetch(e, (size_t)total));
  for (int s = 0; s < npairs; ++s) res[s].assign(e->gram_host + (size_t)s * g.m, e->gram_host + (size_t)(s + 1) * g.m);
  return 0;
}
}  // namespace

extern "C" int dav_gjd_correction(dav_handle_t e, int m, const double* theta, int max_inner, double inner_tol,
                                  int* inner_iters_out) {
  return dav_gjd_correction_n(e, m, m, theta, max_inner, inner_tol, nullptr, inner_iters_out);
}

extern "C" int dav_gjd_correction_n(dav_handle_t e, int mbasis, int m, const double* theta, int max_inner, double inner_tol,
                                    const double* tol_per_col, int* inner_iters_out) {
  CHK(bind(e));
  if (m <= 0 || mbasis < m || mbasis + m > e->cols_alloc || m > e->cols_alloc / 2) return fail("dav_gjd_correction: bad block width");
  if (e->op[DAV_OP_A].kind == DAV_KIND_HOST || e->op[DAV_OP_A].kind == DAV_KIND_NONE)
    return fail("dav_gjd_correction: needs a device operator A");
  const bool gev = e->gev != 0;
  // workspace: 9 column blocks of width cols_alloc/2, allocated on first use
  const int wcols = e->cols_alloc / 2 + 8;
  if (!e->gjd_ws) {
    size_t bytes = sizeof(double) * (size_t)e->ldp * wcols * 9;
    HIPCHK(pool_malloc(&e->gjd_ws, bytes));
    HIPCHK(hipMemsetAsync(e->gjd_ws, 0, bytes, e->stream));
  }
  auto ws = [&](int i) { return e->gjd_ws + (size_t)i * e->ldp * wcols; };
  double* X = panel_ptr(e, DAV_PANEL_X, 0);
  double* T = panel_ptr(e, DAV_PANEL_V, mbasis);
  double* r1 = panel_ptr(e, DAV_PANEL_R, 0);          // becomes b = -r in place
  double *r2 = ws(0), *y = ws(1), *v = ws(2), *w = ws(3), *w1 = ws(4), *w2 = ws(5), *ua = ws(6), *ub = ws(7), *mx = ws(8);

  Gjd g{e, m, 0, {}, 0};
  const std::vector<double> one(m, 1.0), minus_one(m, -1.0), zero(m, 0.0);
  std::vector<double> th(theta, theta + m), active(m, 1.0), res[4];
  int64_t ld_th, ld_act;
  CHK(small_upload(e, 0, th.data(), m, m, 1, &ld_th));
  const double* dA = e->op[DAV_OP_A].diag;
  const double* dB = gev ? e->op[DAV_OP_B].diag : nullptr;

  CHK(gjd_lincomb(g, r1, r1, &minus_one));                                     // b = -r
  CHK(gjd_lincomb(g, T, r1, &zero));                                           // t = 0
  CHK(gjd_lincomb(g, w, r1, &zero));
  CHK(gjd_lincomb(g, w2, r1, &zero));
  launch_copy_columns(e->stream, r1, e->ldp, r2, e->ldp, e->nloc_pad, m);      // r2 = r1
  CHK(small_upload(e, 1, active.data(), m, m, 1, &ld_act));
  // Jacobi preconditioner K = |diag(A) - theta_k diag(B)|, restricted to the complement of x_k:
  //   y = K^-1 r - (x^T K^-1 r / x^T K^-1 x) K^-1 x    (keeps every iterate orthogonal to x_k, so the
  //   null direction of the projected operator can never be amplified)
  launch_precond(e->stream, X, mx, e->ldp, e->nloc, e->nloc_pad, m, e->sm[0].dev, dA, dB, e->sm[1].dev);
  launch_precond(e->stream, r1, y, e->ldp, e->nloc, e->nloc_pad, m, e->sm[0].dev, dA, dB, e->sm[1].dev);
  {
    const double* a[4] = {X, X, r1, r1}; const double* b[4] = {mx, y, y, mx};
    CHK(gjd_dots(g, 4, a, b, res));
  }
  std::vector<double> xmx = res[0], cy(m, 0.0);
  std::vector<double> beta1(m), beta(m), oldb(m, 0.0), dbar(m, 0.0), epsln(m, 0.0), phibar(m), cs(m, -1.0), sn(m, 0.0);
  for (int j = 0; j < m; ++j) {
    cy[j] = xmx[j] > 0 ? -res[1][j] / xmx[j] : 0.0;
    double b2 = res[2][j] + cy[j] * res[3][j];
    beta1[j] = b2 > 0 ? std::sqrt(b2) : 0.0;
    beta[j] = phibar[j] = beta1[j];
    if (!(beta1[j] > 0.0)) active[j] = 0.0;
  }
  // tol_per_col[j] < 0 marks a FOLLOWER: its tolerance is |tol_per_col[j]|, and it also stops as soon as every column that is not a
  // follower has stopped (the corrections of the unwanted Ritz pairs only enrich the basis: they get the steps the wanted pairs
  // need, not a solve of their own)
  std::vector<char> follower(m, 0);
  bool any_leader = false;
  for (int j = 0; j < m; ++j) {
    follower[j] = tol_per_col && tol_per_col[j] < 0.0;
    any_leader = any_leader || !follower[j];
  }
  int itn = 0;
  std::vector<double> c0(m), c1(m), c2(m), c3(m);
  std::vector<int> stall(m, 0);
  while (itn < max_inner) {
    bool any = false;
    for (int j = 0; j < m; ++j) any = any || active[j] != 0.0;
    if (!any) break;
    ++itn;
    // v = (K^-1 r2 projected) / beta  - orthogonal to x by construction, so (I - x x^T) v = v
    for (int j = 0; j < m; ++j) {
      c0[j] = active[j] != 0.0 ? 1.0 / beta[j] : 0.0;
      c1[j] = c0[j] * cy[j];
    }
    CHK(gjd_lincomb(g, v, y, &c0, mx, &c1));
    // U = A v, UB = B v - only over the 16-column groups that still hold an active pair (a sweep costs one
    // pass per group in symmetric storage; columns outside the range are multiplied by zero below)
    int c_lo = m, c_hi = 0;
    for (int j = 0; j < m; ++j)
      if (active[j] != 0.0) { c_lo = std::min(c_lo, j); c_hi = std::max(c_hi, j + 1); }
    // (... or, when the active pairs fit one group of 8 - the wanted pairs outlive the others, which stop at a looser tolerance -
    // over those 8 columns: the 8-column sweep moves the same tiles but writes half the partial sums: 27.6 against 29.4 ms at N=200000)
    if (c_hi - c_lo / 8 * 8 <= 8) {
      c_lo = c_lo / 8 * 8;
      c_hi = std::min(m, c_lo + 8);
    } else {
      c_lo = c_lo / 16 * 16;
      c_hi = std::min(m, (c_hi + 15) / 16 * 16);
    }
    CHK(apply_ptr(e, DAV_OP_A, v + (size_t)c_lo * e->ldp, c_hi - c_lo, ua + (size_t)c_lo * e->ldp, true, true));
    const double* ubp = v;
    if (gev) {
      CHK(apply_ptr(e, DAV_OP_B, v + (size_t)c_lo * e->ldp, c_hi - c_lo, ub + (size_t)c_lo * e->ldp, true, true));
      ubp = ub;
    }
    if (e->tune.gjd_trace) {
      int na = 0;
      for (int j = 0; j < m; ++j) na += active[j] != 0.0;
      fprintf(stderr, "gjd inner %d: active %d of %d, columns [%d, %d)\n", itn, na, m, c_lo, c_hi);
    }
    // y = (U - theta UB) - (x^T(U - theta UB)) x - (beta/oldb) r1
    {
      const double* a[2] = {X, X}; const double* b[2] = {ua, ubp};
      CHK(gjd_dots(g, 2, a, b, res));
    }
    for (int j = 0; j < m; ++j) {
      double act = active[j];
      c0[j] = act;
      c1[j] = -th[j] * act;
      c2[j] = -(res[0][j] - th[j] * res[1][j]) * act;
      c3[j] = (itn >= 2 && act != 0.0) ? -beta[j] / oldb[j] : 0.0;
    }
    CHK(gjd_lincomb(g, y, ua, &c0, ubp, &c1, X, &c2, r1, &c3));
    // alfa = <v, y>;  y -= (alfa/beta) r2
    {
      const double* a[1] = {v}; const double* b[1] = {y};
      CHK(gjd_dots(g, 1, a, b, res));
    }
    std::vector<double> alfa = res[0];
    for (int j = 0; j < m; ++j) { c0[j] = active[j]; c1[j] = active[j] != 0.0 ? -alfa[j] / beta[j] : 0.0; }
    CHK(gjd_lincomb(g, y, y, &c0, r2, &c1));
    // rotate: r1 <- r2, r2 <- y, y <- (old r1 storage)
    { double* t = r1; r1 = r2; r2 = y; y = t; }
    CHK(small_upload(e, 1, active.data(), m, m, 1, &ld_act));
    launch_precond(e->stream, r2, y, e->ldp, e->nloc, e->nloc_pad, m, e->sm[0].dev, dA, dB, e->sm[1].dev);
    {
      const double* a[3] = {X, r2, r2}; const double* b[3] = {y, y, mx};
      CHK(gjd_dots(g, 3, a, b, res));
    }
    // scalar recurrences (Paige & Saunders), per column
    std::vector<double> oldeps(m), delta(m), gamma(m), phi(m);
    for (int j = 0; j < m; ++j) {
      if (active[j] == 0.0) { oldeps[j] = delta[j] = phi[j] = 0.0; gamma[j] = 1.0; continue; }
      cy[j] = -res[0][j] / xmx[j];
      double b2 = res[1][j] + cy[j] * res[2][j];
      oldb[j] = beta[j];
      beta[j] = b2 > 0 ? std::sqrt(b2) : 0.0;
      oldeps[j] = epsln[j];
      delta[j] = cs[j] * dbar[j] + sn[j] * alfa[j];
      double gbar = sn[j] * dbar[j] - cs[j] * alfa[j];
      epsln[j] = sn[j] * beta[j];
      dbar[j] = -cs[j] * beta[j];
      gamma[j] = std::max(std::sqrt(gbar * gbar + beta[j] * beta[j]), 1e-300);
      cs[j] = gbar / gamma[j];
      sn[j] = beta[j] / gamma[j];
      phi[j] = cs[j] * phibar[j];
      stall[j] = (sn[j] > 0.95) ? stall[j] + 1 : 0;       // |phibar| shrinks by sn each step
      phibar[j] = sn[j] * phibar[j];
    }
    // w_new = (v - oldeps w1 - delta w2) / gamma ;  t += phi w_new
    { double* t = w1; w1 = w2; w2 = w; w = t; }
    for (int j = 0; j < m; ++j) {
      double act = active[j];
      c0[j] = act / gamma[j];
      c1[j] = -oldeps[j] * act / gamma[j];
      c2[j] = -delta[j] * act / gamma[j];
    }
    CHK(gjd_lincomb(g, w, v, &c0, w1, &c1, w2, &c2));
    for (int j = 0; j < m; ++j) c1[j] = phi[j] * active[j];
    CHK(gjd_lincomb(g, T, T, &one, w, &c1));
    for (int j = 0; j < m; ++j)
      if (active[j] != 0.0 && (!(phibar[j] > (tol_per_col ? std::fabs(tol_per_col[j]) : inner_tol) * beta1[j]) || !(beta[j] > 0.0) ||
                               (stall[j] >= 8 && phibar[j] < 1e-6 * beta1[j])))
        active[j] = 0.0;      // converged, broke down, or stagnated at the attainable accuracy
    if (any_leader) {
      bool leader_active = false;
      for (int j = 0; j < m; ++j) leader_active = leader_active || (!follower[j] && active[j] != 0.0);
      if (!leader_active)
        for (int j = 0; j < m; ++j) active[j] = 0.0;
    }
  }
  if (inner_iters_out) *inner_iters_out = itn;
  HIPCHK(hipGetLastError());
  return 0;
}
