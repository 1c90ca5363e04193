// K2 - tall-skinny Gram / projection  C(p x q) = P^T Q  (replaces the DGEMM 'T','N' of
// src/davidson.f90:131,223 outer; also the inner products of the block Gram-Schmidt that replaces
// lapack_qr, src/lapack_wrapper.f90:176-236).
//
// The long dimension N is the MFMA contraction index.  Lane (c = lane & 15, g = lane >> 4) loads four
// consecutive rows n0 + 4g .. 4g+3 of panel column c (two 16-byte loads: the 16 lanes of a column
// group read one full 128-byte line per column), and MFMA step s uses element s of both fragments,
// so A and B operands see the same permutation of the contraction index.  Each wave owns a
// (16 PT) x (16 QT) tile of C over a 256-row chunk, the four waves of a workgroup are summed through
// LDS, workgroup partials go to a slab, and the slabs are added in a fixed order - by a second small kernel, or
// (few row chunks: the launch-bound sizes) by the workgroup that finishes last (dav_last_workgroup).
//
// Round 4: a "step" (16 rows of the wave's PT + QT panel columns = 2 (PT + QT) 16-byte loads per lane, 4 PT QT MFMAs) used to
// wait for its own loads - a chain of memory latencies, with 2 x 16 PT QT accumulator copies between the register halves around
// every step (N=200000, 64 x 32: 113 us for 296 MB = 0.33 of 8 TB/s).  The loop now runs through a static ring of U steps -
// the loads of step s + U are requested behind the MFMAs of step s (2 (PT + QT) U KB in flight per wave) - and the file is
// compiled with the accumulators in VGPRs (-amdgpu-mfma-vgpr-form, csrc/Makefile): no copies.
#include "kernels.h"

namespace {
template <int PT, int QT>
struct GramStep {
  f64x2 pf[PT][2], qf[QT][2];
};
}  // namespace

template <int PT, int QT, int U>
__global__ __launch_bounds__(256) void gram_kernel(const double* __restrict__ P, int64_t ldp, int p,
                                                   const double* __restrict__ Q, int64_t ldq, int q,
                                                   int64_t nrows_pad, int ptiles, int qtiles, int nchunks,
                                                   double* __restrict__ slab, int ppad, int qpad, int rows_per_wg,
                                                   double* __restrict__ out, unsigned* __restrict__ counters) {
  __shared__ double red[4][PT * QT * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  // 1-D grid of ntiles x nchunks workgroups.  Consecutive workgroups go to consecutive XCDs (8, each with its
  // own L2), so inside a block of 8 row chunks the workgroup index runs over the chunks first: all output
  // tiles of one row chunk - which re-read the same panel rows - land on ONE XCD and share its L2.
  const int ntiles = ptiles * qtiles;
  const int nfull = nchunks / 8 * 8;
  int tile, chunk;
  if ((int)blockIdx.x < nfull * ntiles) {
    chunk = (blockIdx.x / (8 * ntiles)) * 8 + (blockIdx.x % 8);
    tile = (blockIdx.x / 8) % ntiles;
  } else {
    const int r = blockIdx.x - nfull * ntiles;
    chunk = nfull + r / ntiles;
    tile = r % ntiles;
  }
  const int tp = tile / qtiles, tq = tile % qtiles;
  const int pc0 = tp * 16 * PT, qc0 = tq * 16 * QT;
  int64_t n0 = (int64_t)chunk * rows_per_wg + wave * (rows_per_wg / 4);
  int64_t n1 = n0 + rows_per_wg / 4;
  if (n1 > nrows_pad) n1 = nrows_pad;

  // column pointers; columns past the panel width are clamped (their results are discarded)
  const double* pp[PT];
  const double* qp[QT];
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    int col = pc0 + 16 * t + c;
    if (col >= p) col = p - 1;
    pp[t] = P + (int64_t)col * ldp + 4 * g;
  }
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    int col = qc0 + 16 * t + c;
    if (col >= q) col = q - 1;
    qp[t] = Q + (int64_t)col * ldq + 4 * g;
  }

  // NC accumulator chains per output tile so that consecutive MFMAs never depend on each other: a dependent
  // f64 MFMA issued fewer than ~4 slots behind its producer stalls the matrix pipe
  constexpr int NC = PT * QT >= 4 ? 1 : (PT * QT == 2 ? 2 : 4);
  f64x4 acc[PT][QT][NC];
#pragma unroll
  for (int a = 0; a < PT; ++a)
#pragma unroll
    for (int b = 0; b < QT; ++b)
#pragma unroll
      for (int ch = 0; ch < NC; ++ch) acc[a][b][ch] = f64x4{0.0, 0.0, 0.0, 0.0};

  auto fetch = [&](int64_t n, GramStep<PT, QT>& st) {
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      st.pf[t][0] = *reinterpret_cast<const f64x2*>(pp[t] + n);
      st.pf[t][1] = *reinterpret_cast<const f64x2*>(pp[t] + n + 2);
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      st.qf[t][0] = *reinterpret_cast<const f64x2*>(qp[t] + n);
      st.qf[t][1] = *reinterpret_cast<const f64x2*>(qp[t] + n + 2);
    }
  };
  auto mfmas = [&](const GramStep<PT, QT>& st) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int a = 0; a < PT; ++a)
#pragma unroll
        for (int b = 0; b < QT; ++b) {
          const double pa = (s & 1) ? st.pf[a][s >> 1].y : st.pf[a][s >> 1].x;
          const double qb = (s & 1) ? st.qf[b][s >> 1].y : st.qf[b][s >> 1].x;
          acc[a][b][s % NC] = mfma_f64(pa, qb, acc[a][b][s % NC]);
        }
  };

  // steps of 16 rows: full rounds of U through the ring (the loads of step s + U behind the MFMAs of step s; in the last round
  // the last step is requested again - no branch), the rest unpipelined
  const int64_t nsteps = n1 > n0 ? (n1 - n0) / 16 : 0;
  const int64_t nring = nsteps / U * U;
  if (nring > 0) {
    GramStep<PT, QT> ring[U];
    int64_t nf = n0;
#pragma unroll
    for (int u = 0; u < U; ++u) { fetch(nf, ring[u]); nf += 16; }
    __builtin_amdgcn_sched_barrier(0);
    for (int64_t s0 = 0; s0 < nring; s0 += U) {
      const bool more = s0 + U < nring;
      const int64_t dn = more ? 16 : 0;
      if (!more) nf -= 16;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        mfmas(ring[u]);
        fetch(nf, ring[u]);
        nf += dn;
        __builtin_amdgcn_sched_barrier(0);      // this order is the schedule (see k_panel.hip): MFMAs of step s, then the requests for step s + U
      }
    }
  }
  for (int64_t n = n0 + 16 * nring; n < n1; n += 16) {
    GramStep<PT, QT> st;
    fetch(n, st);
    mfmas(st);
  }

  // cross-wave sum through LDS, then one partial tile per workgroup
#pragma unroll
  for (int a = 0; a < PT; ++a)
#pragma unroll
    for (int b = 0; b < QT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double v = acc[a][b][0][r];
#pragma unroll
        for (int ch = 1; ch < NC; ++ch) v += acc[a][b][ch][r];
        red[wave][((a * QT + b) * 4 + r) * 64 + lane] = v;
      }
  __syncthreads();
  double* part = slab + (int64_t)chunk * ppad * qpad;
  for (int e = threadIdx.x; e < PT * QT * 256; e += 256) {
    double v = red[0][e] + red[1][e] + red[2][e] + red[3][e];
    int l = e & 63, r = (e >> 6) & 3, ab = e >> 8;
    int a = ab / QT, b = ab % QT;
    int prow = pc0 + 16 * a + (l >> 4) + 4 * r;   // index into P columns  (row of C)
    int qcol = qc0 + 16 * b + (l & 15);           // index into Q columns  (column of C)
    if (prow < p && qcol < q) part[(int64_t)qcol * ppad + prow] = v;
  }
  if (counters) {
    // few row chunks: the workgroup that finishes this output tile last adds the partial tiles of all chunks, chunk by chunk
    // with eight interleaved partial sums (a fixed order: reproducible run to run; not the order of gram_reduce_kernel)
    if (dav_last_workgroup(counters + tile, (unsigned)nchunks)) {
      const int64_t stride = (int64_t)ppad * qpad;
      for (int e = threadIdx.x; e < PT * QT * 256; e += 256) {
        const int prow = pc0 + (e & (16 * PT - 1)), qcol = qc0 + e / (16 * PT);
        if (prow >= p || qcol >= q) continue;
        const double* src = slab + (int64_t)qcol * ppad + prow;
        double s8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int ch = 0;
        for (; ch + 8 <= nchunks; ch += 8) {
#pragma unroll
          for (int u = 0; u < 8; ++u) s8[u] += src[(ch + u) * stride];
        }
        for (int u = 0; ch < nchunks; ++ch, ++u) s8[u] += src[ch * stride];
        out[(int64_t)qcol * p + prow] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
      }
    }
  }
}

// out[e] = sum over the row chunks of the partial tiles, fixed order: a workgroup of 1024 threads owns 64 output entries; wave w
// (16 of them) adds the chunks w, w + 16, ... of its entries with four interleaved partial sums, the 16 stripe sums are added in
// order through LDS.  (One thread per entry over all chunks - the first version - is a chain of nchunks / 8 load latencies:
// 19 us at 391 chunks, 39 us at 782, more than the Gram kernel itself for narrow blocks.)
__global__ __launch_bounds__(1024) void gram_reduce_kernel(const double* __restrict__ slab, int nchunks, int p, int q,
                                                           int ppad, int qpad, double* __restrict__ out) {
  __shared__ double stripe[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  double s4[4] = {0.0, 0.0, 0.0, 0.0};
  if (e < p * q) {
    const int row = e % p, col = e / p;
    const double* src = slab + (int64_t)col * ppad + row;
    const int64_t stride = (int64_t)ppad * qpad;
    int ch = w;
    for (; ch + 48 < nchunks; ch += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) s4[u] += src[(int64_t)(ch + 16 * u) * stride];
    }
    for (int u = 0; ch < nchunks; ch += 16, ++u) s4[u] += src[(int64_t)ch * stride];
  }
  stripe[w][lane] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  __syncthreads();
  if (w == 0 && e < p * q) {
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += stripe[u][lane];
    out[(int64_t)(e / p) * p + e % p] = s;
  }
}

static inline int pad16(int x) { return (x + 15) / 16 * 16; }

size_t gram_scratch_doubles(int p, int q, int64_t nrows_pad) {
  int64_t nchunks = (nrows_pad + GRAM_MIN_ROWS - 1) / GRAM_MIN_ROWS;      // worst case of gram_rows_per_wg
  return (size_t)nchunks * pad16(p) * pad16(q);
}

template <int PT, int QT, int U>
static void launch_gram_tiles(hipStream_t st, const double* P, int64_t ldp, int p, const double* Q, int64_t ldq, int q,
                              int64_t nrows_pad, double* scratch, int ppad, int qpad, double* out_dev, unsigned* counters) {
  const int ptiles = (p + 16 * PT - 1) / (16 * PT), qtiles = (q + 16 * QT - 1) / (16 * QT);
  // rows per workgroup: as tall as possible (fewer partial tiles) while the grid still fills the chip
  int rows_per_wg = GRAM_ROWS;
  while (rows_per_wg > GRAM_MIN_ROWS && (int64_t)ptiles * qtiles * ((nrows_pad + rows_per_wg - 1) / rows_per_wg) < 512) rows_per_wg /= 2;
  const int nchunks = (int)((nrows_pad + rows_per_wg - 1) / rows_per_wg);
  // last-workgroup finish where the sum over the chunks is short and a second launch is what costs (<= GRAM_FUSE_CHUNKS row
  // chunks, one counter per output tile); the two-kernel route where hundreds of chunks want more than one workgroup per tile
  const bool fuse = counters && nchunks <= GRAM_FUSE_CHUNKS && ptiles * qtiles <= GRAM_MAX_COUNTERS;
  hipLaunchKernelGGL((gram_kernel<PT, QT, U>), dim3(ptiles * qtiles * nchunks), dim3(256), 0, st, P, ldp, p, Q, ldq, q, nrows_pad, ptiles,
                     qtiles, nchunks, scratch, ppad, qpad, rows_per_wg, out_dev, fuse ? counters : (unsigned*)nullptr);
  if (!fuse) {
    const int total = p * q;
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((total + 63) / 64), dim3(1024), 0, st, scratch, nchunks, p, q, ppad, qpad, out_dev);
  }
}

// counters: GRAM_MAX_COUNTERS zeroed device words (nullptr: always the two-kernel route); tile_mode: 0 = by the shape,
// 1 = never the 64 x 32 register tile (A/B runs)
void launch_gram(hipStream_t st, const double* P, int64_t ldp, int p, const double* Q, int64_t ldq, int q,
                 int64_t nrows_pad, double* scratch, double* out_dev, unsigned* counters, int tile_mode) {
  int ppad = pad16(p), qpad = pad16(q);
  // register tile of a wave: 64 x 32 for wide blocks (every output tile re-reads its panel columns through L2: 128 x 64 as eight
  // 32 x 32 tiles moves 819 MB through L2 for 307 MB of panels), 32 x 32 down to 16 x 16 below
  if (p > 32 && q > 16 && tile_mode != 1)
    launch_gram_tiles<4, 2, 3>(st, P, ldp, p, Q, ldq, q, nrows_pad, scratch, ppad, qpad, out_dev, counters);
  else if (p > 16 && q > 16)
    launch_gram_tiles<2, 2, 3>(st, P, ldp, p, Q, ldq, q, nrows_pad, scratch, ppad, qpad, out_dev, counters);
  else if (p > 16)
    launch_gram_tiles<2, 1, 4>(st, P, ldp, p, Q, ldq, q, nrows_pad, scratch, ppad, qpad, out_dev, counters);
  else
    launch_gram_tiles<1, 1, 4>(st, P, ldp, p, Q, ldq, q, nrows_pad, scratch, ppad, qpad, out_dev, counters);
}
