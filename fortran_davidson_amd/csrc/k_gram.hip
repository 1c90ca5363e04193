// K2 - tall-skinny Gram / projection  C(p x q) = P^T Q  (replaces the DGEMM 'T','N' of
// src/davidson.f90:131,223 outer; also the inner products of the block Gram-Schmidt that replaces
// lapack_qr, src/lapack_wrapper.f90:176-236).
//
// The long dimension N is the MFMA contraction index, and the panels are column-major: an MFMA operand wants "lane <-> panel
// column" while a coalesced load wants "adjacent lanes <-> adjacent rows of one column".  Rounds 1-3 used v_mfma_f64_16x16x4_f64
// with lane (c, g) loading rows 4 g .. 4 g + 3 of column c: SIXTEEN columns - sixteen cache lines, 64 bytes of each - per load
// instruction, which is what bound the kernel (N=200000, 64 x 32: 0.33 of 8 TB/s; pipelining the loop changed nothing).  Round 4
// runs the products on v_mfma_f64_4x4x4_4b_f64 - four independent 4 x 4 x 4 blocks per instruction, A lane = i + 4 blk + 16 k,
// B lane = j + 4 blk + 16 k, D lane = j + 4 blk + 16 i (profiles/ubench/r02_mfma4x4.log), the same pipe time per flop - with the
// blocks as four groups of ROWS: lane (i, blk, k) loads 16 bytes = rows 2 (blk + 4 k), + 1 of column 4 f + i of a step of 32
// rows, so one load instruction reads FOUR columns x 256 contiguous bytes (the pattern of the block matvec's tile loads), P and
// Q alike, and feeds two MFMAs per fragment pair.  A wave owns a (4 PF) x (4 QF) tile of C over its rows, as PF x QF accumulators
// of one double per lane (the four blocks = four partial sums over disjoint rows); they are added across the blocks by two
// row-rotate DPP steps, across the four waves through LDS, workgroup partials go to a slab, and the slabs are added in a fixed
// order - by a second small kernel, or (few row chunks) by the workgroup that finishes last (dav_last_workgroup).
// The loop runs through a static ring of U steps: the loads of step s + U are requested behind the MFMAs of step s.
#include "kernels.h"
#include <algorithm>

namespace {
template <int PF, int QF>
struct GramStep {
  f64x2 pf[PF], qf[QF];
};
__device__ __forceinline__ double mfma4_f64(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
// value of the lane CTRL positions away inside its row of 16 lanes (DPP row rotate: 0x120 + n)
template <int CTRL>
__device__ __forceinline__ double dpp_row(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
}  // namespace

// columns [qeach * s, qeach * (s + 1)) of the right-hand side come from base[s] (launch_gram: one block, qeach = q)
struct GramQ { const double* base[3]; int qeach; };

template <int PF, int QF, int U>
__global__ __launch_bounds__(256) void gram_kernel(const double* __restrict__ P, int64_t ldp, int p,
                                                   GramQ Qs, int64_t ldq, int q,
                                                   int64_t nrows_pad, int ptiles, int qtiles, int nchunks,
                                                   double* __restrict__ slab, int ppad, int qpad, int rows_per_wg,
                                                   double* __restrict__ out, unsigned* __restrict__ counters) {
  __shared__ double red[4][PF * QF * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i4 = lane & 3, blk = (lane >> 2) & 3, kq = lane >> 4;
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
  const int pc0 = tp * 4 * PF, qc0 = tq * 4 * QF;
  int64_t n0 = (int64_t)chunk * rows_per_wg + wave * (rows_per_wg / 4);
  int64_t n1 = n0 + rows_per_wg / 4;
  if (n1 > nrows_pad) n1 = nrows_pad;

  // column pointers of this lane (rows 2 (blk + 4 k), + 1 of a step); columns past the panel width are clamped (their results
  // are discarded)
  const int roff = 2 * (blk + 4 * kq);
  const double* pp[PF];
  const double* qp[QF];
#pragma unroll
  for (int f = 0; f < PF; ++f) {
    int col = pc0 + 4 * f + i4;
    if (col >= p) col = p - 1;
    pp[f] = P + (int64_t)col * ldp + roff;
  }
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    int col = qc0 + 4 * f + i4;
    if (col >= q) col = q - 1;
    const int src = col / Qs.qeach;
    qp[f] = Qs.base[src] + (int64_t)(col - src * Qs.qeach) * ldq + roff;
  }

  double acc[PF][QF];
#pragma unroll
  for (int a = 0; a < PF; ++a)
#pragma unroll
    for (int b = 0; b < QF; ++b) acc[a][b] = 0.0;

  auto fetch = [&](int64_t n, GramStep<PF, QF>& st) {
#pragma unroll
    for (int f = 0; f < PF; ++f) st.pf[f] = *reinterpret_cast<const f64x2*>(pp[f] + n);
#pragma unroll
    for (int f = 0; f < QF; ++f) st.qf[f] = *reinterpret_cast<const f64x2*>(qp[f] + n);
  };
  // the two rows a lane holds = two MFMAs per fragment pair; an accumulator is touched once per PF QF MFMAs
  auto mfmas = [&](const GramStep<PF, QF>& st) {
#pragma unroll
    for (int a = 0; a < PF; ++a)
#pragma unroll
      for (int b = 0; b < QF; ++b) acc[a][b] = mfma4_f64(st.pf[a].x, st.qf[b].x, acc[a][b]);
#pragma unroll
    for (int a = 0; a < PF; ++a)
#pragma unroll
      for (int b = 0; b < QF; ++b) acc[a][b] = mfma4_f64(st.pf[a].y, st.qf[b].y, acc[a][b]);
  };

  // steps of 32 rows: full rounds of U through the ring (the loads of step s + U behind the MFMAs of step s; in the last round
  // the last step is requested again - no branch), the rest unpipelined
  const int64_t nsteps = n1 > n0 ? (n1 - n0) / 32 : 0;
  const int64_t nring = nsteps / U * U;
  if (nring > 0) {
    GramStep<PF, QF> ring[U];
    int64_t nf = n0;
#pragma unroll
    for (int u = 0; u < U; ++u) { fetch(nf, ring[u]); nf += 32; }
    __builtin_amdgcn_sched_barrier(0);
    for (int64_t s0 = 0; s0 < nring; s0 += U) {
      const bool more = s0 + U < nring;
      const int64_t dn = more ? 32 : 0;
      if (!more) nf -= 32;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        mfmas(ring[u]);
        fetch(nf, ring[u]);
        nf += dn;
        __builtin_amdgcn_sched_barrier(0);      // this order is the schedule (see k_panel.hip)
      }
    }
  }
  for (int64_t n = n0 + 32 * nring; n < n1; n += 32) {
    GramStep<PF, QF> st;
    fetch(n, st);
    mfmas(st);
  }

  // sum over the four blocks (lanes i + 4 blk' + 16 i' of a row of 16: two rotate-and-add steps leave the sum in every lane),
  // then over the four waves through LDS: one partial tile per workgroup
#pragma unroll
  for (int a = 0; a < PF; ++a)
#pragma unroll
    for (int b = 0; b < QF; ++b) {
      double v = acc[a][b];
      v += dpp_row<0x128>(v);                   // row_ror:8
      v += dpp_row<0x124>(v);                   // row_ror:4
      if (blk == 0) red[wave][(a * QF + b) * 16 + kq * 4 + i4] = v;     // D lane = j + 4 blk + 16 i: kq is the row i of C, i4 its column j
    }
  __syncthreads();
  double* part = slab + (int64_t)chunk * ppad * qpad;
  for (int e = threadIdx.x; e < PF * QF * 16; e += 256) {
    const double v = red[0][e] + red[1][e] + red[2][e] + red[3][e];
    const int ab = e >> 4, a = ab / QF, b = ab % QF;
    const int prow = pc0 + 4 * a + ((e >> 2) & 3);   // index into P columns  (row of C)
    const int qcol = qc0 + 4 * b + (e & 3);          // index into Q columns  (column of C)
    if (prow < p && qcol < q) part[(int64_t)qcol * ppad + prow] = v;
  }
  if (counters) {
    // few row chunks: the workgroup that finishes this output tile last adds the partial tiles of all chunks, chunk by chunk
    // with eight interleaved partial sums (a fixed order: reproducible run to run; not the order of gram_reduce_kernel)
    if (dav_last_workgroup(counters + tile, (unsigned)nchunks)) {
      const int64_t stride = (int64_t)ppad * qpad;
      for (int e = threadIdx.x; e < PF * QF * 16; e += 256) {
        const int prow = pc0 + e % (4 * PF), qcol = qc0 + e / (4 * PF);
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
static int g_fuse_chunks = GRAM_FUSE_CHUNKS;        // A/B knob (DAV_GRAM_FUSE at dav_create): row chunks up to which the last workgroup sums the partial tiles
void gram_set_fuse_chunks(int n) { g_fuse_chunks = n > 0 ? n : GRAM_FUSE_CHUNKS; }

size_t gram_scratch_doubles(int p, int q, int64_t nrows_pad) {
  int64_t nchunks = (nrows_pad + GRAM_MIN_ROWS - 1) / GRAM_MIN_ROWS;      // worst case of gram_rows_per_wg
  return (size_t)nchunks * pad16(p) * pad16(q);
}

template <int PF, int QF, int U>
static void launch_gram_tiles(hipStream_t st, const double* P, int64_t ldp, int p, const GramQ& Q, int64_t ldq, int q,
                              int64_t nrows_pad, double* scratch, int ppad, int qpad, double* out_dev, unsigned* counters, int wg_target) {
  const int ptiles = (p + 4 * PF - 1) / (4 * PF), qtiles = (q + 4 * QF - 1) / (4 * QF);
  // rows per workgroup: ONE workgroup per CU where the panels are long enough (wg_target = 256 workgroups over all output tiles:
  // a workgroup costs several microseconds of prologue, cross-wave sum and partial-tile write whatever its length - measured at
  // N=200000, 64 x 32: 782 workgroups of 512 rows 80 us, 392 of 1024 rows 59 us, 196 of 2048 rows 48 us), never below
  // GRAM_MIN_ROWS; a multiple of 128 (four waves x steps of 32 rows)
  const int64_t want_chunks = std::max<int64_t>(1, wg_target / ((int64_t)ptiles * qtiles));
  int64_t rows = (nrows_pad + want_chunks - 1) / want_chunks;
  rows = std::max<int64_t>(GRAM_MIN_ROWS, (rows + 127) / 128 * 128);
  const int rows_per_wg = (int)rows;
  const int nchunks = (int)((nrows_pad + rows_per_wg - 1) / rows_per_wg);
  // last-workgroup finish where the sum over the chunks is short and a second launch is what costs (<= GRAM_FUSE_CHUNKS row
  // chunks, one counter per output tile); the two-kernel route where hundreds of chunks want more than one workgroup per tile
  const bool fuse = counters && nchunks <= g_fuse_chunks && ptiles * qtiles <= GRAM_MAX_COUNTERS;
  hipLaunchKernelGGL((gram_kernel<PF, QF, U>), dim3(ptiles * qtiles * nchunks), dim3(256), 0, st, P, ldp, p, Q, ldq, q, nrows_pad, ptiles,
                     qtiles, nchunks, scratch, ppad, qpad, rows_per_wg, out_dev, fuse ? counters : (unsigned*)nullptr);
  if (!fuse) {
    const int total = p * q;
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((total + 63) / 64), dim3(1024), 0, st, scratch, nchunks, p, q, ppad, qpad, out_dev);
  }
}

// counters: GRAM_MAX_COUNTERS zeroed device words (nullptr: always the two-kernel route)
static void launch_gram_q(hipStream_t st, const double* P, int64_t ldp, int p, const GramQ& Q, int64_t ldq, int q,
                          int64_t nrows_pad, double* scratch, double* out_dev, unsigned* counters, int wg_target) {
  int ppad = pad16(p), qpad = pad16(q);
  if (wg_target <= 0) wg_target = 256;
  // register tile of a wave: 32 x 32 (64 accumulators), 32 x 16, 16 x 16 columns
  if (p > 16 && q > 16)
    launch_gram_tiles<8, 8, 3>(st, P, ldp, p, Q, ldq, q, nrows_pad, scratch, ppad, qpad, out_dev, counters, wg_target);
  else if (p > 16)
    launch_gram_tiles<8, 4, 4>(st, P, ldp, p, Q, ldq, q, nrows_pad, scratch, ppad, qpad, out_dev, counters, wg_target);
  else
    launch_gram_tiles<4, 4, 4>(st, P, ldp, p, Q, ldq, q, nrows_pad, scratch, ppad, qpad, out_dev, counters, wg_target);
}

void launch_gram(hipStream_t st, const double* P, int64_t ldp, int p, const double* Q, int64_t ldq, int q,
                 int64_t nrows_pad, double* scratch, double* out_dev, unsigned* counters, int wg_target) {
  GramQ src{{Q, Q, Q}, q > 0 ? q : 1};
  launch_gram_q(st, P, ldp, p, src, ldq, q, nrows_pad, scratch, out_dev, counters, wg_target);
}

void launch_gram_multi(hipStream_t st, const double* P, int64_t ldp, int p, const double* const* Qs, int nq, int qeach, int64_t ldq,
                       int64_t nrows_pad, double* scratch, double* out_dev, unsigned* counters, int wg_target) {
  GramQ src{{Qs[0], Qs[nq > 1 ? 1 : 0], Qs[nq > 2 ? 2 : 0]}, qeach > 0 ? qeach : 1};
  launch_gram_q(st, P, ldp, p, src, ldq, nq * qeach, nrows_pad, scratch, out_dev, counters, wg_target);
}
