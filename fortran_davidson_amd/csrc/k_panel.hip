// K3 / K4 / K5 - "panel x small matrix" block transforms
//     OUT[:, 0:q] = P1[:, 0:p1] * M1 (+ P2[:, 0:p2] * M2)
// with optional fused epilogues.  One kernel family replaces
//   * the Ritz-vector DGEMM X = V*Y                       (src/davidson.f90:159, :397)
//   * the m residual DGEMV sweeps / the three residual DGEMMs: R = W*Y + Z*(-Y*diag(theta))
//                                                          (src/davidson.f90:163-170, :401-410)
//   * norm() of the first `lowest` residual columns        (src/davidson.f90:173-178, :412-414)
//   * the DPR correction T = R ./ (theta_j*dB_i - dA_i)    (src/davidson.f90:673-698, :463-488)
//   * the block Gram-Schmidt update T <- T*M + V*(-C*M)    (instead of lapack_qr, :213)
//   * the collapse restart V <- V*Y(:, 1:2L)               (src/davidson.f90:218, :438)
//
// The panels are column-major with the long dimension contiguous, so the MFMA A operand (16 rows x 4 columns, here 32 rows via
// one 16-byte load per lane) is read straight from HBM in 256-byte row runs; the small matrix is the B operand, served by L2
// from its operand image (kernels.h: one coalesced 512-byte load per step and column tile).
// Accumulators: 32 rows x 16*QT columns per wave.  HBM-bound on the panel reads (m/8..m/4 flop/B), executed on
// v_mfma_f64_16x16x4_f64.
//
// Round 4: the k loop is software-pipelined through a static register ring.  A "step" is four panel columns = ONE 16-byte
// load per lane (1 KB per wave) + QT 8-byte loads of the small matrix + 2 QT MFMAs; the first version issued the loads of a step
// and waited for them before its MFMAs - a chain of p/4 memory latencies per wave (N=200000, p = 128 -> 64 columns: 154 us for
// 313 MB = 0.25 of 8 TB/s, neither bound: PMC traffic = algorithmic bytes).  Now the operands of step s + U are requested before
// the MFMAs of step s (U = 8: 8 KB of panel data in flight per wave, 64-128 KB per CU); the ring is indexed by compile-time
// constants only (fully unrolled rounds of U steps), so there are no register moves and the compiler counts vmcnt itself.
// Steps past the end of a term are made of a repeated valid load and a zeroed A operand: straight-line code, no branches.
// The two rows (2j, 2j + 1) a lane holds of an output column leave as ONE 16-byte store.  In-place use (OUT aliases P1, e.g.
// the Gram-Schmidt update of the new columns or the restart) is safe whenever one workgroup covers all q columns (q <= 64):
// a wave has read its 32 rows of every input column before it stores the first output, and no other wave touches those rows.
#include "kernels.h"

namespace {
struct PgStep {      // operands of one step, as loaded
  f64x2 a;
  double b[4];
};
}  // namespace

// acc += P[rows of the wave, 0:p] * M[0:p, q0 : q0 + 16 QT].  A "step" is four panel columns: one 16-byte load per lane of the
// panel (lane (c, g): rows 2c, 2c + 1 of column 4 s + g) and QT 8-byte loads of the small matrix (row 4 s + g of column
// q0 + 16 t + c), then 2 QT MFMAs.  Steps whose four columns all exist run through a ring of U slots (operands of step s + U
// requested behind the MFMAs of step s; past the last such step the last one is requested again: no branch, nothing masked);
// a trailing step with fewer than four columns - and what does not fill a round of U - runs unpipelined behind them: a missing
// column is replaced by the last valid one, which meets rows of the small matrix that are zero (the operand images are zero
// outside p x q: small_upload_image / rr_pack), so its product vanishes whatever the panel holds.
template <int QT, int U, bool PIN>
__device__ __forceinline__ void pg_term(const double* P, int64_t ld, int p, const double* __restrict__ M, int64_t tp, int64_t i0, int q0,
                                        int c, int g, f64x4 (&acc)[2][QT]) {
  const double* ap = P + i0 + 2 * c + (int64_t)g * ld;            // this lane's column of step 0
  const double* bp = M + (int64_t)(q0 >> 4) * 64 + c + 16 * g;    // this lane's entry of tile q0 / 16 of step 0 in the operand image
  const int64_t bstep = tp * 64;
  auto fetch = [&](const double* a, const double* b, PgStep& st) {
    st.a = *reinterpret_cast<const f64x2*>(a);
#pragma unroll
    for (int t = 0; t < QT; ++t) st.b[t] = b[64 * t];             // 64 lanes x 8 B contiguous: one coalesced 512-byte load per tile
  };
  auto mfmas = [&](const PgStep& st) {
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      acc[0][t] = mfma_f64(st.a.x, st.b[t], acc[0][t]);
      acc[1][t] = mfma_f64(st.a.y, st.b[t], acc[1][t]);
    }
  };
  const int64_t astep = 4 * ld;
  const int nfull = (p >> 2) / U * U;                             // steps of the pipelined rounds
  if (nfull > 0) {
    PgStep ring[U];
    const double* af = ap;                                        // where the next request goes
    const double* bf = bp;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      fetch(af, bf, ring[u]);
      af += astep; bf += bstep;
    }
    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
    for (int s0 = 0; s0 < nfull; s0 += U) {
      const bool more = s0 + U < nfull;                           // another round behind this one (uniform)
      const int64_t da = more ? astep : 0;
      const int64_t db = more ? bstep : 0;
      if (!more) { af -= astep; bf -= bstep; }                    // last round: the last pipelined step again and again
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // the MFMAs of step s first, then the requests for step s + U: the slot is dead by then (no second register set), and
        // the loads are issued in the shadow of the last MFMA
        mfmas(ring[u]);
        fetch(af, bf, ring[u]);
        af += da; bf += db;
        // the order above IS the schedule: without the barrier the machine scheduler hoists a round's requests to its top and the
        // round's first MFMAs wait for loads issued just in front of them (seen in the ISA: vmcnt(28) behind 30 loads, vmcnt(0)
        // at the end of every round) - a latency per round instead of a pipeline
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  for (int s = nfull; 4 * s < p; ++s) {
    const int col = min(4 * s + g, p - 1);
    PgStep st;
    fetch(P + i0 + 2 * c + (int64_t)col * ld, bp + (int64_t)s * bstep, st);
    mfmas(st);
  }
}

template <int QT, int U, bool PIN>
__global__ __launch_bounds__(256) void panel_gemm_kernel(PanelGemmArgs A) {
  __shared__ double nrm[4][16 * QT];
  if (A.batch > 1) {                                  // panel blockIdx.z of a batched launch (uniform)
    const int64_t shift = (int64_t)blockIdx.z * A.batch_stride;
    A.P1 += shift;
    if (A.p2 > 0) A.P2 += shift;
    A.out += shift;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int64_t i0 = (int64_t)blockIdx.x * PG_ROWS + wave * 32;
  const int q0 = blockIdx.y * 16 * QT;

  f64x4 acc[2][QT];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int t = 0; t < QT; ++t) acc[h][t] = f64x4{0.0, 0.0, 0.0, 0.0};

  pg_term<QT, U, PIN>(A.P1, A.ld1, A.p1, A.M1, A.tp1, i0, q0, c, g, acc);
  if (A.p2 > 0) pg_term<QT, U, PIN>(A.P2, A.ld2, A.p2, A.M2, A.tp2, i0, q0, c, g, acc);

  // acc[h][t][reg] = OUT[i0 + 2*(g + 4*reg) + h][q0 + 16 t + c]
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int col = q0 + 16 * t + c;
    double ssq = 0.0;
    const double th = (A.epilogue == 1 && col < A.q) ? A.theta[col] : 0.0;   // epilogue 2: store + norms
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int64_t row = i0 + 2 * (g + 4 * reg);
      f64x2 v = f64x2{acc[0][t][reg], acc[1][t][reg]};
      if (row >= A.nloc) v.x = 0.0;
      if (row + 1 >= A.nloc) v.y = 0.0;
      if (A.epilogue >= 1) ssq += v.x * v.x + v.y * v.y;
      if (A.epilogue == 1) {
        if (row < A.nloc) {
          const double den = th * (A.dB ? A.dB[row] : 1.0) - A.dA[row];
          v.x = (den != 0.0) ? v.x / den : 0.0;
        }
        if (row + 1 < A.nloc) {
          const double den = th * (A.dB ? A.dB[row + 1] : 1.0) - A.dA[row + 1];
          v.y = (den != 0.0) ? v.y / den : 0.0;
        }
      }
      if (col < A.q) *reinterpret_cast<f64x2*>(A.out + (int64_t)col * A.ldo + row) = v;
    }
    if (A.epilogue >= 1 && A.nnorm > 0) {
      ssq += __shfl_xor(ssq, 16);
      ssq += __shfl_xor(ssq, 32);
      if (g == 0) nrm[wave][16 * t + c] = ssq;
    }
  }
  if (A.epilogue >= 1 && A.nnorm > 0) {
    __syncthreads();
    if (threadIdx.x < 16 * QT) {
      int col = q0 + threadIdx.x;
      if (col < A.nnorm)
        A.norm_partial[(int64_t)blockIdx.x * A.nnorm + col] =
            nrm[0][threadIdx.x] + nrm[1][threadIdx.x] + nrm[2][threadIdx.x] + nrm[3][threadIdx.x];
    }
    if (A.norm_out) {
      // Last-workgroup finish: the workgroup that arrives last adds the per-workgroup partial sums in a fixed order (one wave per
      // column, lanes stride over the row blocks, fixed shuffle tree) - the same sum norm_finish_kernel makes, without its launch
      if (dav_last_workgroup(A.counter, gridDim.x * gridDim.y)) {
        const int nblocks = gridDim.x;
        for (int j = wave; j < A.nnorm; j += 4) {
          double s = 0.0;
          for (int b = lane; b < nblocks; b += 64) s += A.norm_partial[(int64_t)b * A.nnorm + j];
          for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
          if (lane == 0) A.norm_out[j] = s;     // squared norm; all-reduce (several ranks) and sqrt happen on the host side
        }
      }
    }
  }
}

// a.pin: the pinned schedule (MFMAs of step s, then the requests of step s + U) or the compiler's own order of a round
void launch_panel_gemm(hipStream_t st, const PanelGemmArgs& a) {
  unsigned gx = (unsigned)(a.nrows_pad / PG_ROWS);
  const unsigned gz = a.batch > 1 ? (unsigned)a.batch : 1u;
#define DAV_PG_LAUNCH(QT, GY)                                                                                              \
  do {                                                                                                                      \
    if (a.pin) hipLaunchKernelGGL((panel_gemm_kernel<QT, 8, true>), dim3(gx, GY, gz), dim3(256), 0, st, a);                \
    else hipLaunchKernelGGL((panel_gemm_kernel<QT, 8, false>), dim3(gx, GY, gz), dim3(256), 0, st, a);                     \
  } while (0)
  if (a.q <= 16) DAV_PG_LAUNCH(1, (a.q + 15) / 16);
  else if (a.q <= 32) DAV_PG_LAUNCH(2, (a.q + 31) / 32);
  else DAV_PG_LAUNCH(4, (a.q + 63) / 64);
#undef DAV_PG_LAUNCH
}

// out[j] = sum_b partial[b][j]: one wave per output column, lanes stride over the blocks, fixed
// shuffle tree -> reproducible.  (A single-thread serial sum cost 20 us at 157 blocks.)  Used when the panel kernel
// does not finish the norms itself (PanelGemmArgs::norm_out == nullptr).
__global__ __launch_bounds__(256) void norm_finish_kernel(const double* __restrict__ partial, int nblocks, int nnorm,
                                                          double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= nnorm) return;
  double s = 0.0;
  for (int b = lane; b < nblocks; b += 64) s += partial[(int64_t)b * nnorm + j];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) out[j] = s;     // squared norm / dot; all-reduce (multi-GPU) and sqrt happen on the host side
}

void launch_norm_finish(hipStream_t st, const double* partial, int nblocks, int nnorm, double* out) {
  hipLaunchKernelGGL(norm_finish_kernel, dim3((nnorm + 3) / 4), dim3(256), 0, st, partial, nblocks, nnorm, out);
}
