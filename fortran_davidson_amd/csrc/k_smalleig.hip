// Device-side Rayleigh-Ritz: all eigenpairs of the projected problem H y = theta y or H y = theta S y,
// order m <= 128, by ONE workgroup - what lapack_generalized_eigensolver (src/lapack_wrapper.f90:14-91: DSYEV / DSYGV
// itype = 1, 'V', 'U') computes on the host.  With it theta and Y never leave HBM and the H-down / Y-up round trip
// of an iteration disappears.
//
//   generalized:  S = L L^T (Cholesky, in place);  C = L^-1 H L^-T;  C = Z Theta Z^T;  Y = L^-T Z   (Y^T S Y = I)
//   standard:     H = Y Theta Y^T                                                                    (Y^T Y = I)
//
// Eigen-decomposition: cyclic two-sided Jacobi with the round-robin ("tournament") ordering - m/2 disjoint rotations
// per step (two barrier-separated phases: rotation angles, then every 2 x 2 block of A updated from both sides at once),
// m - 1 steps per sweep, until a whole sweep finds no |a_pq| > 1e-16 sqrt(|a_pp a_qq|); the projected matrices
// of a Davidson basis are close to diagonal (after a restart exactly diagonal), so 3-6 sweeps suffice.  The symmetric matrix lives in LDS (column-major, odd
// stride); the accumulated rotations too when both fit (m <= 96), else in global memory (L2).  Eigenvalues leave
// ascending (ties by index), eigenvectors in the matching order.  One workgroup: the order is at most 128, the sweeps
// are latency (barrier) bound, not throughput bound - measured against the host in docs/history/DESIGN_rounds_1_to_5.md section 10.
#include "kernels.h"
#include <algorithm>

namespace {
constexpr int EIG_THREADS = 1024;

__device__ __forceinline__ void tournament_pair(int M, int s, int k, int* p, int* q) {
  // round-robin over M (even) players, step s in [0, M-1), table k in [0, M/2): player M-1 stays, the others rotate
  int a, b;
  if (k == 0) { a = M - 1; b = s; }
  else { a = (s + k) % (M - 1); b = (s - k + (M - 1)) % (M - 1); }
  *p = a < b ? a : b;
  *q = a < b ? b : a;
}
}  // namespace

// H, S: device, column-major (ld), full symmetric m x m (S ignored unless gev).  On exit theta[0..m), Y (ldy) as above.
// work: >= 2 * m * m doubles of global scratch.  info[0] = sweeps used, or -j if the Cholesky met a non-positive pivot j
// (then nothing else is written).
template <bool VLDS>
__global__ __launch_bounds__(EIG_THREADS) void small_eig_kernel(const double* __restrict__ H, int64_t ldh, const double* __restrict__ S,
                                                               int64_t lds_, int m, int gev, double* __restrict__ theta,
                                                               double* __restrict__ Y, int64_t ldy, double* __restrict__ work,
                                                               double* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, NT = blockDim.x;   // as many waves as a step has work for: a barrier costs per wave
  const int la = m | 1;                        // odd leading dimension: column and row walks are both conflict free
  double* As = smem;                           // m x m
  double* Vs = VLDS ? smem + (size_t)la * m : work;   // accumulated rotations (ld = lv)
  const int lv = VLDS ? la : m;
  double* Lg = work + (size_t)m * m;           // generalized: the Cholesky factor (global, ld = m)
  __shared__ double rot_c[64], rot_s[64];
  __shared__ int flag;

  // ---- load; generalized: reduce to the standard problem ------------------------------------------------------
  for (int e = tid; e < m * m; e += NT) {
    const int i = e % m, j = e / m;
    As[i + j * la] = H[i + (int64_t)j * ldh];
  }
  if (tid == 0) flag = 0;
  __syncthreads();
  if (gev) {
    // Cholesky S = L L^T, right-looking, in global scratch (m steps, the whole workgroup updates the trailing block)
    for (int e = tid; e < m * m; e += NT) Lg[e] = S[(e % m) + (int64_t)(e / m) * lds_];
    __syncthreads();
    for (int j = 0; j < m; ++j) {
      const double d = Lg[j + j * m];
      if (!(d > 0.0)) { if (tid == 0) flag = -(j + 1); }
      __syncthreads();
      if (flag != 0) break;
      const double sd = sqrt(d);
      for (int i = j + tid; i < m; i += NT) Lg[i + j * m] = i == j ? sd : Lg[i + j * m] / sd;
      __syncthreads();
      // trailing update: S[i, k] -= L[i, j] L[k, j] for j < k <= i
      const int nt = m - j - 1;
      for (int e = tid; e < nt * nt; e += NT) {
        const int i = j + 1 + e % nt, k = j + 1 + e / nt;
        if (k <= i) Lg[i + k * m] -= Lg[i + j * m] * Lg[k + j * m];
      }
      __syncthreads();
    }
    if (flag != 0) { if (tid == 0) info[0] = (double)flag; return; }
    // C = L^-1 H L^-T.  X = L^-1 H: forward substitution down the rows, all columns at once (thread = column)
    for (int i = 0; i < m; ++i) {
      for (int j = tid; j < m; j += NT) {
        double acc = As[i + j * la];
        for (int k = 0; k < i; ++k) acc -= Lg[i + k * m] * As[k + j * la];
        As[i + j * la] = acc / Lg[i + i * m];
      }
      __syncthreads();
    }
    // C = X L^-T: the same substitution on the rows of X (thread = row), i.e. C^T = L^-1 X^T
    for (int jc = 0; jc < m; ++jc) {
      for (int i = tid; i < m; i += NT) {
        double acc = As[i + jc * la];
        for (int k = 0; k < jc; ++k) acc -= Lg[jc + k * m] * As[i + k * la];
        As[i + jc * la] = acc / Lg[jc + jc * m];
      }
      __syncthreads();
    }
    // symmetrise (rounding): the Jacobi sweeps read both triangles
    for (int e = tid; e < m * m; e += NT) {
      const int i = e % m, j = e / m;
      if (i > j) { const double v = 0.5 * (As[i + j * la] + As[j + i * la]); As[i + j * la] = v; As[j + i * la] = v; }
    }
    __syncthreads();
  }
  for (int e = tid; e < m * m; e += NT) Vs[(e % m) + (size_t)(e / m) * lv] = (e % m) == (e / m) ? 1.0 : 0.0;
  __syncthreads();

  // ---- cyclic Jacobi ------------------------------------------------------------------------------------------------
  const int M = (m + 1) & ~1, half = M / 2;
  // thread -> (pair slot, row or column) with power-of-two arithmetic: integer division by the run-time order and the
  // modulo of the round-robin table cost more than the rotations themselves (a configs[1] solve with the device eigensolver: 3.54 -> 2.80 ms)
  int lg = 0;
  while ((1 << lg) < m) ++lg;
  const int mp = 1 << lg;                      // rows / columns per pass, padded to a power of two
  const int kstep = NT >> lg;                  // pairs handled per pass (NT >= mp: NT = 1024 or >= m*m/2 rounded to 64)
  const int ti = tid & (mp - 1), tk = tid >> lg;
  __shared__ int pair_p[64], pair_q[64];
  int sweeps = 0;
  for (; sweeps < 30; ++sweeps) {
    if (tid == 0) flag = 0;
    __syncthreads();
    for (int s = 0; s < M - 1; ++s) {
      // rotations of this step.  A pair is rotated while |a_pq| > 1e-16 sqrt(|a_pp a_qq|): the criterion that gives the
      // small eigenvalues of a graded matrix their relative accuracy (a norm-wise test would stop at 1e-16 |H|, and the
      // projected matrices carry diagonal entries five orders of magnitude above the wanted Ritz values)
      if (tid < half) {
        int p, q;
        tournament_pair(M, s, tid, &p, &q);
        double c = 1.0, sn = 0.0;
        if (q < m) {
          const double apq = As[p + q * la], app = As[p + p * la], aqq = As[q + q * la];
          if (fabs(apq) > 1e-16 * sqrt(fabs(app * aqq)) + 1e-300) {
            const double tau = (aqq - app) / (2.0 * apq);
            const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
            c = 1.0 / sqrt(1.0 + t * t);
            sn = t * c;
            flag = 1;
          }
        }
        rot_c[tid] = c; rot_s[tid] = sn;
        pair_p[tid] = p; pair_q[tid] = q < m ? q : -1;
      }
      __syncthreads();
      // A <- J^T A J in ONE phase: the 2 x 2 block of row pair ki and column pair kj only depends on itself, so a thread
      // that owns a block applies the column rotation of pair kj and the row rotation of pair ki to its four entries
      // with no barrier in between (two phases per step instead of three); V <- V J alongside (columns only)
      {
        const int hp = mp >> 1 ? mp >> 1 : 1;                   // pair slots padded to a power of two
        const int lgh = lg > 0 ? lg - 1 : 0;
        for (int e = tid; e < hp * hp; e += NT) {
          const int ki = e >> lgh, kj = e & (hp - 1);
          if (ki >= half || kj >= half) continue;
          const int pi = pair_p[ki], qi = pair_q[ki], pj = pair_p[kj], qj = pair_q[kj];
          const double ci = rot_c[ki], si = rot_s[ki], cj = rot_c[kj], sj = rot_s[kj];
          if (si == 0.0 && sj == 0.0) continue;
          // entries (pi|qi, pj|qj); a missing partner (odd order: q = -1) has the identity rotation and no entries
          double app = As[pi + pj * la], apq = qj >= 0 ? As[pi + qj * la] : 0.0;
          double aqp = qi >= 0 ? As[qi + pj * la] : 0.0, aqq = (qi >= 0 && qj >= 0) ? As[qi + qj * la] : 0.0;
          const double tpp = cj * app - sj * apq, tpq = sj * app + cj * apq;      // columns
          const double tqp = cj * aqp - sj * aqq, tqq = sj * aqp + cj * aqq;
          As[pi + pj * la] = ci * tpp - si * tqp;                                 // rows
          if (qj >= 0) As[pi + qj * la] = ci * tpq - si * tqq;
          if (qi >= 0) As[qi + pj * la] = si * tpp + ci * tqp;
          if (qi >= 0 && qj >= 0) As[qi + qj * la] = si * tpq + ci * tqq;
        }
        if (ti < m)
          for (int k = tk; k < half; k += kstep) {
            const int p = pair_p[k], q = pair_q[k];
            const double c = rot_c[k], sn = rot_s[k];
            if (q < 0 || sn == 0.0) continue;
            const double vp = Vs[ti + (size_t)p * lv], vq = Vs[ti + (size_t)q * lv];
            Vs[ti + (size_t)p * lv] = c * vp - sn * vq;
            Vs[ti + (size_t)q * lv] = sn * vp + c * vq;
          }
      }
      __syncthreads();
    }
    // a whole sweep without a rotation ends the iteration.  The decision is latched in a register BEFORE the barrier
    // that lets thread 0 reset the flag for the next sweep: read after it, a slower wave could see the reset value,
    // leave the loop alone and strand the workgroup at different barriers.
    const int rotated = flag;
    __syncthreads();
    if (rotated == 0) break;
  }

  // ---- ascending order (ties by index), Y ---------------------------------------------------------------------------
  // rank of eigenvalue j = number of eigenvalues that sort before it
  __shared__ int ranks[128];
  for (int j = tid; j < m; j += NT) {
    const double dj = As[j + j * la];
    int r = 0;
    for (int i = 0; i < m; ++i) {
      const double di = As[i + i * la];
      r += (di < dj) || (di == dj && i < j);
    }
    ranks[j] = r;
    theta[r] = dj;
  }
  __syncthreads();
  if (!gev) {
    for (int e = tid; e < m * m; e += NT) {
      const int i = e % m, j = e / m;
      Y[i + (int64_t)ranks[j] * ldy] = Vs[i + (size_t)j * lv];
    }
  } else {
    // Y = L^-T Z: back substitution up the rows, thread = column
    for (int j = tid; j < m; j += NT) {
      double* y = Y + (int64_t)ranks[j] * ldy;
      for (int i = m - 1; i >= 0; --i) {
        double acc = Vs[i + (size_t)j * lv];
        for (int k = i + 1; k < m; ++k) acc -= Lg[k + i * m] * y[k];
        y[i] = acc / Lg[i + i * m];
      }
    }
  }
  if (tid == 0) info[0] = (double)sweeps;
}

size_t small_eig_work_doubles(int m) { return (size_t)2 * m * m + 64; }

// all on `st`; returns false if m is out of range
bool launch_small_eig(hipStream_t st, const double* H, int64_t ldh, const double* S, int64_t lds, int m, bool gev, double* theta,
                      double* Y, int64_t ldy, double* work, double* info) {
  if (m < 1 || m > 128) return false;
  const int la = m | 1;
  const bool vlds = (size_t)2 * la * m * sizeof(double) <= (size_t)150 * 1024;
  const size_t shmem = sizeof(double) * (size_t)la * m * (vlds ? 2 : 1);
  // A Jacobi step has (m/2) * m element pairs per phase: one per thread up to the 1024 of a workgroup (with four pairs per
  // thread - 512 threads at m = 64 - a configs[1] solve took 4.25 instead of 3.54 ms: the phases are LDS-latency bound)
  int mp = 1;
  while (mp < m) mp <<= 1;
  const int threads = std::max(64, std::min(EIG_THREADS, mp * std::max(1, mp / 2)));   // one (pair, row) per thread up to 1024; a multiple of mp
  if (vlds) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&small_eig_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024); attr = true; }
    hipLaunchKernelGGL(small_eig_kernel<true>, dim3(1), dim3(threads), shmem, st, H, ldh, S, lds, m, gev ? 1 : 0, theta, Y, ldy, work, info);
  } else {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&small_eig_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024); attr = true; }
    hipLaunchKernelGGL(small_eig_kernel<false>, dim3(1), dim3(threads), shmem, st, H, ldh, S, lds, m, gev ? 1 : 0, theta, Y, ldy, work, info);
  }
  return true;
}

// ---- glue kernels of the device-resident Rayleigh-Ritz step --------------------------------------------------------
// new columns c0..c0+k of the projected matrix: block (mt x k, ld = mt) -> Hd[:, c0:c0+k] and its mirror image
__global__ __launch_bounds__(256) void rr_scatter_kernel(const double* __restrict__ blk, int mt, int k, int c0, double* __restrict__ Hd,
                                                         int64_t ld) {
  for (int e = blockIdx.x * 256 + threadIdx.x; e < mt * k; e += gridDim.x * 256) {
    const int i = e % mt, j = e / mt;
    const double v = blk[e];
    Hd[i + (int64_t)(c0 + j) * ld] = v;
    if (i < c0) Hd[(c0 + j) + (int64_t)i * ld] = v;
  }
}
void launch_rr_scatter(hipStream_t st, const double* blk, int mt, int k, int c0, double* Hd, int64_t ld) {
  hipLaunchKernelGGL(rr_scatter_kernel, dim3((mt * k + 255) / 256), dim3(256), 0, st, blk, mt, k, c0, Hd, ld);
}

// Y (m x m, ld) and theta -> the operand images the panel kernels read: Ypk = Y[:, 0:q] and Y2pk = -Y diag(theta)
// with leading dimension ldm (rows m..ldm zero) and qpad columns (columns q.. zero), theta_pk[0:q], and theta[0:m] +
// info appended to the result buffer the host fetches
__global__ __launch_bounds__(256) void rr_pack_kernel(const double* __restrict__ Y, int64_t ld, const double* __restrict__ theta, int m, int q,
                                                      int ldm, int qpad, double* __restrict__ Ypk, double* __restrict__ Y2pk,
                                                      double* __restrict__ theta_pk, const double* __restrict__ info,
                                                      double* __restrict__ result_tail) {
  // MFMA-B operand images for panel_gemm_kernel (kernels.h: pg_image_index): entry e of the image = lane (c, g) of tile t of step s
  const int tp = qpad / 16;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < ldm * qpad; e += gridDim.x * 256) {
    const int lane = e & 63, t = (e >> 6) % tp, s = (e >> 6) / tp;
    const int i = 4 * s + (lane >> 4), j = 16 * t + (lane & 15);
    const double v = (i < m && j < q) ? Y[i + (int64_t)j * ld] : 0.0;
    Ypk[e] = v;
    Y2pk[e] = (i < m && j < q) ? -v * theta[j] : 0.0;
  }
  if (blockIdx.x == 0) {
    for (int j = threadIdx.x; j < qpad; j += 256) theta_pk[j] = j < q ? theta[j] : 0.0;
    if (result_tail) {
      for (int j = threadIdx.x; j < m; j += 256) result_tail[j] = theta[j];
      if (threadIdx.x == 0) result_tail[m] = info[0];
    }
  }
}
void launch_rr_pack(hipStream_t st, const double* Y, int64_t ld, const double* theta, int m, int q, int ldm, int qpad, double* Ypk,
                    double* Y2pk, double* theta_pk, const double* info, double* result_tail) {
  hipLaunchKernelGGL(rr_pack_kernel, dim3((ldm * qpad + 255) / 256), dim3(256), 0, st, Y, ld, theta, m, q, ldm, qpad, Ypk, Y2pk,
                     theta_pk, info, result_tail);
}
