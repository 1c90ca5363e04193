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
// The panels are column-major with the long dimension contiguous, so the MFMA A operand
// (16 rows x 4 columns, here 32 rows via one 16-byte load per lane) is read straight from HBM/L2 in
// 256-byte row runs; the small matrix is the B operand, served by L1/L2.  Accumulators: 32 rows x
// 16*QT columns per wave.  HBM-bound on the panel reads (m/8..m/4 flop/B), executed on
// v_mfma_f64_16x16x4_f64.
#include "kernels.h"

template <int QT>
__device__ __forceinline__ void pg_term(const double* __restrict__ P, int64_t ld, int p, const double* __restrict__ M,
                                        int64_t ldm, int64_t i0, int q0, int c, int g, f64x4 (&acc)[2][QT]) {
  const double* ap = P + i0 + 2 * c + (int64_t)g * ld;
  const double* bp = M + (int64_t)(q0 + c) * ldm + g;
  for (int kk = 0; kk < p; kk += 4) {
    f64x2 a = *reinterpret_cast<const f64x2*>(ap + (int64_t)kk * ld);
    if (kk + g >= p) a = f64x2{0.0, 0.0};     // columns past the panel width may hold anything
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      double b = bp[(int64_t)(16 * t) * ldm + kk];
      acc[0][t] = mfma_f64(a.x, b, acc[0][t]);
      acc[1][t] = mfma_f64(a.y, b, acc[1][t]);
    }
  }
}

template <int QT>
__global__ __launch_bounds__(256) void panel_gemm_kernel(PanelGemmArgs A) {
  __shared__ double nrm[4][16 * QT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int64_t i0 = (int64_t)blockIdx.x * PG_ROWS + wave * 32;
  const int q0 = blockIdx.y * 16 * QT;

  f64x4 acc[2][QT];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int t = 0; t < QT; ++t) acc[h][t] = f64x4{0.0, 0.0, 0.0, 0.0};

  pg_term<QT>(A.P1, A.ld1, A.p1, A.M1, A.ldm1, i0, q0, c, g, acc);
  if (A.p2 > 0) pg_term<QT>(A.P2, A.ld2, A.p2, A.M2, A.ldm2, i0, q0, c, g, acc);

  // acc[h][t][reg] = OUT[i0 + 2*(g + 4*reg) + h][q0 + 16 t + c]
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int col = q0 + 16 * t + c;
    double ssq = 0.0;
    const double th = (A.epilogue == 1 && col < A.q) ? A.theta[col] : 0.0;   // epilogue 2: store + norms
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int64_t row = i0 + 2 * (g + 4 * reg) + h;
        double v = acc[h][t][reg];
        if (row >= A.nloc) v = 0.0;
        if (A.epilogue >= 1) ssq += v * v;
        if (A.epilogue == 1) {
          if (row < A.nloc) {
            double db = A.dB ? A.dB[row] : 1.0;
            double den = th * db - A.dA[row];
            v = (den != 0.0) ? v / den : 0.0;
          }
        }
        if (col < A.q) A.out[(int64_t)col * A.ldo + row] = v;
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
  }
}

void launch_panel_gemm(hipStream_t st, const PanelGemmArgs& a) {
  unsigned gx = (unsigned)(a.nrows_pad / PG_ROWS);
  if (a.q <= 16) {
    hipLaunchKernelGGL(panel_gemm_kernel<1>, dim3(gx, (a.q + 15) / 16), dim3(256), 0, st, a);
  } else if (a.q <= 32) {
    hipLaunchKernelGGL(panel_gemm_kernel<2>, dim3(gx, (a.q + 31) / 32), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(panel_gemm_kernel<4>, dim3(gx, (a.q + 63) / 64), dim3(256), 0, st, a);
  }
}

// out[j] = sum_b partial[b][j]: one wave per output column, lanes stride over the blocks, fixed
// shuffle tree -> reproducible.  (A single-thread serial sum cost 20 us at 157 blocks.)
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
