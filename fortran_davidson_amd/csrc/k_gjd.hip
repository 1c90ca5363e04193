// K7 helpers - column-wise vector kernels of the block MINRES that replaces the dense projected
// solves of compute_GJD_generalized_dense (src/davidson.f90:700-734).  All are single-pass,
// 16-byte-per-lane, HBM-bound elementwise or reduction kernels over N x m column blocks; per-column
// scalars come from small device arrays so that all m systems advance in lock step.
#include "kernels.h"

__global__ __launch_bounds__(256) void lincomb_kernel(LincombArgs a) {
  int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  int j = blockIdx.y;
  if (i >= a.nrows_pad) return;
  f64x2 acc = {0.0, 0.0};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t < a.nterms) {
      double c = a.coef[(int64_t)t * a.ldc + j];
      if (c != 0.0) {
        f64x2 v = *reinterpret_cast<const f64x2*>(a.in[t] + (int64_t)j * a.ld + i);
        acc.x += c * v.x;
        acc.y += c * v.y;
      }
    }
  }
  *reinterpret_cast<f64x2*>(a.out + (int64_t)j * a.ld + i) = acc;
}
void launch_lincomb(hipStream_t st, const LincombArgs& a) {
  hipLaunchKernelGGL(lincomb_kernel, dim3((unsigned)((a.nrows_pad / 2 + 255) / 256), a.m), dim3(256), 0, st, a);
}

// out[i,j] = in[i,j] / max(|dA_i - theta_j dB_i|, floor)   (Jacobi preconditioner of A - theta_j B, made SPD)
__global__ __launch_bounds__(256) void precond_kernel(const double* __restrict__ in, double* __restrict__ out, int64_t ld,
                                                      int64_t nloc, int64_t nrows_pad, const double* __restrict__ theta,
                                                      const double* __restrict__ dA, const double* __restrict__ dB,
                                                      const double* __restrict__ active) {
  int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  int j = blockIdx.y;
  if (i >= nrows_pad) return;
  f64x2 v = {0.0, 0.0};
  if (active[j] != 0.0) {
    const double th = theta[j];
    f64x2 x = *reinterpret_cast<const f64x2*>(in + (int64_t)j * ld + i);
    if (i < nloc) {
      double d = fabs(dA[i] - th * (dB ? dB[i] : 1.0));
      v.x = x.x / fmax(d, 1e-10);
    }
    if (i + 1 < nloc) {
      double d = fabs(dA[i + 1] - th * (dB ? dB[i + 1] : 1.0));
      v.y = x.y / fmax(d, 1e-10);
    }
  }
  *reinterpret_cast<f64x2*>(out + (int64_t)j * ld + i) = v;
}
void launch_precond(hipStream_t st, const double* in, double* out, int64_t ld, int64_t nloc, int64_t nrows_pad, int m,
                    const double* theta, const double* dA, const double* dB, const double* active) {
  hipLaunchKernelGGL(precond_kernel, dim3((unsigned)((nrows_pad / 2 + 255) / 256), m), dim3(256), 0, st, in, out, ld, nloc,
                     nrows_pad, theta, dA, dB, active);
}

// partial[blockIdx.x][s*m + j] = sum over this block's rows of a_s[i,j] * b_s[i,j]
constexpr int DOT_ROWS = 4096;
__global__ __launch_bounds__(256) void coldots_kernel(DotsArgs a) {
  __shared__ double red[4][4];
  const int j = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t i0 = (int64_t)blockIdx.x * DOT_ROWS;
  double sum[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t i = i0 + 2 * threadIdx.x; i < i0 + DOT_ROWS && i < a.nrows_pad; i += 512) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (s < a.npairs) {
        f64x2 x = *reinterpret_cast<const f64x2*>(a.a[s] + (int64_t)j * a.ld + i);
        f64x2 y = *reinterpret_cast<const f64x2*>(a.b[s] + (int64_t)j * a.ld + i);
        sum[s] += x.x * y.x + x.y * y.y;
      }
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    double v = sum[s];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) red[wave][s] = v;
  }
  __syncthreads();
  if (threadIdx.x < a.npairs)
    a.partial[(int64_t)blockIdx.x * (a.npairs * a.m) + threadIdx.x * a.m + j] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
int coldots_blocks(int64_t nrows_pad) { return (int)((nrows_pad + DOT_ROWS - 1) / DOT_ROWS); }
void launch_coldots(hipStream_t st, const DotsArgs& a) {
  hipLaunchKernelGGL(coldots_kernel, dim3(coldots_blocks(a.nrows_pad), a.m), dim3(256), 0, st, a);
}
