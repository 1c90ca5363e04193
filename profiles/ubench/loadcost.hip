// What ONE tile load costs a wave that is alone on its SIMD and otherwise issues fp64 MFMAs back to back (matvec_symw_kernel's
// regime), by the form of the load: buffer_load with a VGPR offset (what the kernel uses), buffer_load with the lane term
// supplied by the descriptor (ADD_TID_ENABLE, no VGPR operand), global_load with an SGPR base, buffer_load straight into LDS,
// and the same loads placed between the MFMAs instead of in front of them.
//   hipcc --offload-arch=gfx950 -O3 loadcost.hip -o loadcost && ./loadcost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MF8(A0, A1, B0, B1)                                                                                              \
  "v_mfma_f64_16x16x4_f64 %0, " A0 ", " B0 ", %0\n\tv_mfma_f64_16x16x4_f64 %1, " A0 ", " B1 ", %1\n\t"                   \
  "v_mfma_f64_16x16x4_f64 %2, " A1 ", " B0 ", %2\n\tv_mfma_f64_16x16x4_f64 %3, " A1 ", " B1 ", %3\n\t"                   \
  "v_mfma_f64_16x16x4_f64 %4, " A0 ", " B0 ", %4\n\tv_mfma_f64_16x16x4_f64 %5, " A0 ", " B1 ", %5\n\t"                   \
  "v_mfma_f64_16x16x4_f64 %6, " A1 ", " B0 ", %6\n\tv_mfma_f64_16x16x4_f64 %7, " A1 ", " B1 ", %7\n\t"

enum { NONE = 0, BUF_OFFEN = 1, BUF_TID = 2, GLOBAL_SADDR = 3, BUF_LDS = 4, BUF_OFFEN_SPREAD = 5, BUF_TID_SPREAD = 6, BUF_OFFEN_X2 = 7, BUF_TID_X2 = 8, DS_BURST = 9, DS_SPREAD = 10, MIX_BURST = 11, MIX_SPREAD = 12, MIX_SPREAD2 = 13, DEP_NONE = 14, DEP_SPREAD = 15, DEP_SPREAD_SALU = 16, DEP_SPREAD_NOP = 17 };

// the destination registers of the loads are fixed (a[0:15] / a[16:31]) so that no variant pays for a copy
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const double* __restrict__ src, double* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
  __shared__ __attribute__((aligned(16))) double lds[4 * 544 + 1024];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  f64x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f64x4{0, 0, 0, 0};
  double a0 = src[lane], a1 = src[64 + lane], b0 = src[128 + lane], b1 = src[192 + lane];
  const double* base = src + blockIdx.x % 8 * 8192;
  u32x4 d_raw, d_tid;
  d_raw[0] = (unsigned)(uintptr_t)base; d_raw[1] = (unsigned)((uintptr_t)base >> 32) & 0xffff; d_raw[2] = 65536; d_raw[3] = 0x00020000;
  d_tid = d_raw; d_tid[1] |= 16u << 16; d_tid[2] = 0xffffffffu; d_tid[3] = 0x00800000;      // stride 16 B, ADD_TID_ENABLE
  for (int i = 0; i < 4; ++i) { d_raw[i] = __builtin_amdgcn_readfirstlane(d_raw[i]); d_tid[i] = __builtin_amdgcn_readfirstlane(d_tid[i]); }
  unsigned voff = (g * 256 + 2 * c) * 8;
  unsigned m0v = __builtin_amdgcn_readfirstlane(wave * 64 * 16 * 4);
  unsigned ldsa = (unsigned)(uintptr_t)(lds) + wave * 4352 + lane * 16;
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_mov_b32 m0, %0" ::"s"(m0v));
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int it = 0; it < iters; ++it) {
    unsigned so = __builtin_amdgcn_readfirstlane((it & 3) * 1024);
#define OPS : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]) \
            : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(voff), "s"(d_raw), "s"(d_tid), "s"(so), "s"(base), "v"(ldsa)                   \
            : "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "s90", "scc"
#define M32 MF8("%8", "%9", "%10", "%11") MF8("%8", "%9", "%10", "%11") MF8("%8", "%9", "%10", "%11") MF8("%8", "%9", "%10", "%11")
    if (MODE == NONE) {
      asm volatile(M32 OPS);
    } else if (MODE == BUF_OFFEN) {
      asm volatile("s_waitcnt vmcnt(0)\n\t"
                   "buffer_load_dwordx4 a[0:3], %12, %13, %15 offen\n\tbuffer_load_dwordx4 a[4:7], %12, %13, %15 offen offset:64\n\t"
                   "buffer_load_dwordx4 a[8:11], %12, %13, %15 offen offset:128\n\tbuffer_load_dwordx4 a[12:15], %12, %13, %15 offen offset:192\n\t" M32 OPS);
    } else if (MODE == BUF_TID) {
      asm volatile("s_waitcnt vmcnt(0)\n\t"
                   "buffer_load_dwordx4 a[0:3], off, %14, %15\n\tbuffer_load_dwordx4 a[4:7], off, %14, %15 offset:1024\n\t"
                   "buffer_load_dwordx4 a[8:11], off, %14, %15 offset:2048\n\tbuffer_load_dwordx4 a[12:15], off, %14, %15 offset:3072\n\t" M32 OPS);
    } else if (MODE == GLOBAL_SADDR) {
      asm volatile("s_waitcnt vmcnt(0)\n\t"
                   "global_load_dwordx4 a[0:3], %12, %16\n\tglobal_load_dwordx4 a[4:7], %12, %16 offset:64\n\t"
                   "global_load_dwordx4 a[8:11], %12, %16 offset:128\n\tglobal_load_dwordx4 a[12:15], %12, %16 offset:192\n\t" M32 OPS);
    } else if (MODE == BUF_LDS) {
      asm volatile("s_waitcnt vmcnt(0)\n\t"
                   "buffer_load_dwordx4 %12, %13, %15 offen lds\n\tbuffer_load_dwordx4 %12, %13, %15 offen offset:64 lds\n\t"
                   "buffer_load_dwordx4 %12, %13, %15 offen offset:128 lds\n\tbuffer_load_dwordx4 %12, %13, %15 offen offset:192 lds\n\t" M32 OPS);
    } else if (MODE == BUF_OFFEN_SPREAD) {
      asm volatile("s_waitcnt vmcnt(0)\n\t"
                   "buffer_load_dwordx4 a[0:3], %12, %13, %15 offen\n\t" MF8("%8", "%9", "%10", "%11")
                   "buffer_load_dwordx4 a[4:7], %12, %13, %15 offen offset:64\n\t" MF8("%8", "%9", "%10", "%11")
                   "buffer_load_dwordx4 a[8:11], %12, %13, %15 offen offset:128\n\t" MF8("%8", "%9", "%10", "%11")
                   "buffer_load_dwordx4 a[12:15], %12, %13, %15 offen offset:192\n\t" MF8("%8", "%9", "%10", "%11") OPS);
    } else if (MODE == BUF_TID_SPREAD) {
      asm volatile("s_waitcnt vmcnt(0)\n\t"
                   "buffer_load_dwordx4 a[0:3], off, %14, %15\n\t" MF8("%8", "%9", "%10", "%11")
                   "buffer_load_dwordx4 a[4:7], off, %14, %15 offset:1024\n\t" MF8("%8", "%9", "%10", "%11")
                   "buffer_load_dwordx4 a[8:11], off, %14, %15 offset:2048\n\t" MF8("%8", "%9", "%10", "%11")
                   "buffer_load_dwordx4 a[12:15], off, %14, %15 offset:3072\n\t" MF8("%8", "%9", "%10", "%11") OPS);
    } else if (MODE == BUF_OFFEN_X2) {
      asm volatile("s_waitcnt vmcnt(0)\n\t"
                   "buffer_load_dwordx4 a[0:3], %12, %13, %15 offen\n\tbuffer_load_dwordx4 a[4:7], %12, %13, %15 offen offset:64\n\t"
                   "buffer_load_dwordx4 a[8:11], %12, %13, %15 offen offset:128\n\tbuffer_load_dwordx4 a[12:15], %12, %13, %15 offen offset:192\n\t"
                   "buffer_load_dwordx4 a[16:19], %12, %13, %15 offen offset:256\n\tbuffer_load_dwordx4 a[20:23], %12, %13, %15 offen offset:320\n\t"
                   "buffer_load_dwordx4 a[24:27], %12, %13, %15 offen offset:384\n\tbuffer_load_dwordx4 a[28:31], %12, %13, %15 offen offset:448\n\t" M32 OPS);
    } else if (MODE == BUF_TID_X2) {
      asm volatile("s_waitcnt vmcnt(0)\n\t"
                   "buffer_load_dwordx4 a[0:3], off, %14, %15\n\tbuffer_load_dwordx4 a[4:7], off, %14, %15 offset:1024\n\t"
                   "buffer_load_dwordx4 a[8:11], off, %14, %15 offset:2048\n\tbuffer_load_dwordx4 a[12:15], off, %14, %15 offset:3072\n\t"
                   "buffer_load_dwordx4 a[16:19], off, %14, %15\n\tbuffer_load_dwordx4 a[20:23], off, %14, %15 offset:1024\n\t"
                   "buffer_load_dwordx4 a[24:27], off, %14, %15 offset:2048\n\tbuffer_load_dwordx4 a[28:31], off, %14, %15 offset:3072\n\t" M32 OPS);
    } else if (MODE == DS_BURST) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                   "ds_write_b128 %17, a[0:3]\n\tds_write_b128 %17, a[4:7] offset:1088\n\tds_write_b128 %17, a[8:11] offset:2176\n\tds_write_b128 %17, a[12:15] offset:3264\n\t"
                   "ds_read_b128 v[200:203], %17\n\tds_read_b128 v[204:207], %17 offset:16\n\tds_read_b128 v[208:211], %17 offset:128\n\tds_read_b128 v[212:215], %17 offset:144\n\t" M32 OPS);
    } else if (MODE == DS_SPREAD) {
#define M4 "v_mfma_f64_16x16x4_f64 %0, %8, %10, %0\n\tv_mfma_f64_16x16x4_f64 %1, %8, %11, %1\n\tv_mfma_f64_16x16x4_f64 %2, %9, %10, %2\n\tv_mfma_f64_16x16x4_f64 %3, %9, %11, %3\n\t"
#define M2 "v_mfma_f64_16x16x4_f64 %4, %8, %10, %4\n\tv_mfma_f64_16x16x4_f64 %5, %8, %11, %5\n\t"
      asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                   "ds_write_b128 %17, a[0:3]\n\t" M4 "ds_write_b128 %17, a[4:7] offset:1088\n\t" M4 "ds_write_b128 %17, a[8:11] offset:2176\n\t" M4
                   "ds_write_b128 %17, a[12:15] offset:3264\n\t" M4
                   "ds_read_b128 v[200:203], %17\n\t" M4 "ds_read_b128 v[204:207], %17 offset:16\n\t" M4 "ds_read_b128 v[208:211], %17 offset:128\n\t" M4
                   "ds_read_b128 v[212:215], %17 offset:144\n\t" M4 OPS);
    } else if (MODE == MIX_BURST) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
                   "buffer_load_dwordx4 a[0:3], %12, %13, %15 offen\n\tbuffer_load_dwordx4 a[4:7], %12, %13, %15 offen offset:64\n\t"
                   "buffer_load_dwordx4 a[8:11], %12, %13, %15 offen offset:128\n\tbuffer_load_dwordx4 a[12:15], %12, %13, %15 offen offset:192\n\t"
                   "buffer_load_dwordx4 a[16:19], %12, %13, %15 offen offset:256\n\tbuffer_load_dwordx4 a[20:23], %12, %13, %15 offen offset:320\n\t"
                   "buffer_load_dwordx4 a[24:27], %12, %13, %15 offen offset:384\n\tbuffer_load_dwordx4 a[28:31], %12, %13, %15 offen offset:448\n\t"
                   "ds_write_b128 %17, a[0:3]\n\tds_write_b128 %17, a[4:7] offset:1088\n\tds_write_b128 %17, a[8:11] offset:2176\n\tds_write_b128 %17, a[12:15] offset:3264\n\t"
                   "ds_read_b128 v[200:203], %17\n\tds_read_b128 v[204:207], %17 offset:16\n\tds_read_b128 v[208:211], %17 offset:128\n\tds_read_b128 v[212:215], %17 offset:144\n\t" M32 OPS);
    } else if (MODE == MIX_SPREAD) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
                   "ds_write_b128 %17, a[0:3]\n\t" M2 "ds_write_b128 %17, a[4:7] offset:1088\n\t" M2 "ds_write_b128 %17, a[8:11] offset:2176\n\t" M2
                   "ds_write_b128 %17, a[12:15] offset:3264\n\t" M2
                   "buffer_load_dwordx4 a[0:3], %12, %13, %15 offen\n\t" M2 "buffer_load_dwordx4 a[4:7], %12, %13, %15 offen offset:64\n\t" M2
                   "buffer_load_dwordx4 a[8:11], %12, %13, %15 offen offset:128\n\t" M2 "buffer_load_dwordx4 a[12:15], %12, %13, %15 offen offset:192\n\t" M2
                   "buffer_load_dwordx4 a[16:19], %12, %13, %15 offen offset:256\n\t" M2 "buffer_load_dwordx4 a[20:23], %12, %13, %15 offen offset:320\n\t" M2
                   "buffer_load_dwordx4 a[24:27], %12, %13, %15 offen offset:384\n\t" M2 "buffer_load_dwordx4 a[28:31], %12, %13, %15 offen offset:448\n\t" M2
                   "ds_read_b128 v[200:203], %17\n\t" M2 "ds_read_b128 v[204:207], %17 offset:16\n\t" M2 "ds_read_b128 v[208:211], %17 offset:128\n\t" M2
                   "ds_read_b128 v[212:215], %17 offset:144\n\t" M2 OPS);
    } else if (MODE == MIX_SPREAD2) {   // the same 16 instructions, two per 4 MFMAs
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
                   "ds_write_b128 %17, a[0:3]\n\tds_write_b128 %17, a[4:7] offset:1088\n\t" M4 "ds_write_b128 %17, a[8:11] offset:2176\n\t"
                   "ds_write_b128 %17, a[12:15] offset:3264\n\t" M4
                   "buffer_load_dwordx4 a[0:3], %12, %13, %15 offen\n\tbuffer_load_dwordx4 a[4:7], %12, %13, %15 offen offset:64\n\t" M4
                   "buffer_load_dwordx4 a[8:11], %12, %13, %15 offen offset:128\n\tbuffer_load_dwordx4 a[12:15], %12, %13, %15 offen offset:192\n\t" M4
                   "buffer_load_dwordx4 a[16:19], %12, %13, %15 offen offset:256\n\tbuffer_load_dwordx4 a[20:23], %12, %13, %15 offen offset:320\n\t" M4
                   "buffer_load_dwordx4 a[24:27], %12, %13, %15 offen offset:384\n\tbuffer_load_dwordx4 a[28:31], %12, %13, %15 offen offset:448\n\t" M4
                   "ds_read_b128 v[200:203], %17\n\tds_read_b128 v[204:207], %17 offset:16\n\t" M4 "ds_read_b128 v[208:211], %17 offset:128\n\t"
                   "ds_read_b128 v[212:215], %17 offset:144\n\t" M4 OPS);
    } else if (MODE >= DEP_NONE) {
      // the kernel's dependency pattern: per group two direct accumulators and ONE transposed accumulator used by every second MFMA
#define D2(d, z) "v_mfma_f64_16x16x4_f64 " d ", %8, %10, " d "\n\tv_mfma_f64_16x16x4_f64 " z ", %9, %11, " z "\n\t"
#define SL(op) op
#define GRP(o1, o2, o3, o4) D2("%0", "%4") o1 D2("%1", "%4") o2 D2("%2", "%5") o3 D2("%3", "%5") o4
      if (MODE == DEP_NONE)
        asm volatile(GRP("", "", "", "") GRP("", "", "", "") GRP("", "", "", "") GRP("", "", "", "") OPS);
      else if (MODE == DEP_SPREAD)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
                     GRP("ds_write_b128 %17, a[0:3]\n\t", "ds_write_b128 %17, a[4:7] offset:1088\n\t", "ds_write_b128 %17, a[8:11] offset:2176\n\t", "ds_write_b128 %17, a[12:15] offset:3264\n\t")
                     GRP("ds_read_b128 v[200:203], %17\n\t", "ds_read_b128 v[204:207], %17 offset:16\n\t", "ds_read_b128 v[208:211], %17 offset:128\n\t", "ds_read_b128 v[212:215], %17 offset:144\n\t")
                     GRP("buffer_load_dwordx4 a[0:3], %12, %13, %15 offen\n\t", "buffer_load_dwordx4 a[4:7], %12, %13, %15 offen offset:64\n\t", "buffer_load_dwordx4 a[8:11], %12, %13, %15 offen offset:128\n\t", "buffer_load_dwordx4 a[12:15], %12, %13, %15 offen offset:192\n\t")
                     GRP("buffer_load_dwordx2 a[16:17], %12, %13, %15 offen offset:256\n\t", "buffer_load_dwordx2 a[20:21], %12, %13, %15 offen offset:320\n\t", "buffer_load_dwordx2 a[24:25], %12, %13, %15 offen offset:384\n\t", "buffer_load_dwordx2 a[28:29], %12, %13, %15 offen offset:448\n\t") OPS);
      else if (MODE == DEP_SPREAD_SALU)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
                     GRP("ds_write_b128 %17, a[0:3]\n\ts_add_u32 s90, s90, 1\n\t", "ds_write_b128 %17, a[4:7] offset:1088\n\t", "ds_write_b128 %17, a[8:11] offset:2176\n\ts_add_u32 s90, s90, 1\n\t", "ds_write_b128 %17, a[12:15] offset:3264\n\t")
                     GRP("ds_read_b128 v[200:203], %17\n\ts_add_u32 s90, s90, 1\n\t", "ds_read_b128 v[204:207], %17 offset:16\n\t", "ds_read_b128 v[208:211], %17 offset:128\n\ts_add_u32 s90, s90, 1\n\t", "ds_read_b128 v[212:215], %17 offset:144\n\t")
                     GRP("buffer_load_dwordx4 a[0:3], %12, %13, %15 offen\n\ts_add_u32 s90, s90, 1\n\t", "buffer_load_dwordx4 a[4:7], %12, %13, %15 offen offset:64\n\t", "buffer_load_dwordx4 a[8:11], %12, %13, %15 offen offset:128\n\ts_add_u32 s90, s90, 1\n\t", "buffer_load_dwordx4 a[12:15], %12, %13, %15 offen offset:192\n\t")
                     GRP("buffer_load_dwordx2 a[16:17], %12, %13, %15 offen offset:256\n\ts_add_u32 s90, s90, 1\n\t", "buffer_load_dwordx2 a[20:21], %12, %13, %15 offen offset:320\n\t", "buffer_load_dwordx2 a[24:25], %12, %13, %15 offen offset:384\n\ts_add_u32 s90, s90, 1\n\t", "buffer_load_dwordx2 a[28:29], %12, %13, %15 offen offset:448\n\t") OPS);
      else
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
                     GRP("s_nop 0\n\t", "s_nop 0\n\t", "s_nop 0\n\t", "s_nop 0\n\t") GRP("s_nop 0\n\t", "s_nop 0\n\t", "s_nop 0\n\t", "s_nop 0\n\t")
                     GRP("s_nop 0\n\t", "s_nop 0\n\t", "s_nop 0\n\t", "s_nop 0\n\t") GRP("s_nop 0\n\t", "s_nop 0\n\t", "s_nop 0\n\t", "s_nop 0\n\t") OPS);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 3\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  double s = lds[threadIdx.x];
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  double chk;
  asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(*reinterpret_cast<unsigned*>(&chk)));
  out[blockIdx.x * 256 + threadIdx.x] = s + chk;
  if (lane == 0) { cyc[2 * (blockIdx.x * 4 + wave)] = t1 - t0; cyc[2 * (blockIdx.x * 4 + wave) + 1] = r1 - r0; }
}

// what the ADD_TID descriptor returns: lane l must see bytes [16 l, 16 l + 16) of the buffer (+ soffset + offset)
__global__ void check_tid(const double* __restrict__ src, double* __restrict__ out) {
  u32x4 d;
  d[0] = (unsigned)(uintptr_t)src; d[1] = ((unsigned)((uintptr_t)src >> 32) & 0xffff) | (16u << 16); d[2] = 0xffffffffu; d[3] = 0x00800000;
  for (int i = 0; i < 4; ++i) d[i] = __builtin_amdgcn_readfirstlane(d[i]);
  f64x2 v;
  unsigned so = __builtin_amdgcn_readfirstlane(2048u);
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, off, %1, %2 offset:1024\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "s"(d), "s"(so) : "memory");
  out[2 * threadIdx.x] = v.x; out[2 * threadIdx.x + 1] = v.y;
}

template <int MODE>
void run(const char* name, const double* src, double* out, unsigned long long* cyc, double base_cycles, int nloads) {
  const int iters = 4000, nwg = 256;
  std::vector<unsigned long long> h(2 * nwg * 4);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<MODE>), dim3(nwg), dim3(256), 0, 0, src, out, cyc, iters);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double c = 0, r = 0;
  for (int i = 0; i < nwg * 4; ++i) { c += h[2 * i]; r += h[2 * i + 1]; }
  const double per = c / (nwg * 4) / iters;
  printf("%-52s %8.1f cycles per 32 MFMAs", name, per);
  if (nloads) printf("  = %5.1f per load", (per - base_cycles) / nloads);
  printf("   clock %.3f GHz\n", c / r * 0.1);
}

int main() {
  double *src, *out; unsigned long long* cyc;
  hipMalloc(&src, 1 << 20); hipMalloc(&out, 256 * 256 * 8); hipMalloc(&cyc, 256 * 4 * 2 * 8);
  std::vector<double> h((1 << 20) / 8);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)i;
  hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check_tid, dim3(1), dim3(64), 0, 0, src, out);
  std::vector<double> o(128);
  hipMemcpy(o.data(), out, 128 * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) bad += o[2 * l] != (double)((2048 + 1024) / 8 + 2 * l) || o[2 * l + 1] != (double)((2048 + 1024) / 8 + 2 * l + 1);
  printf("ADD_TID_ENABLE descriptor: lane l reads bytes 16 l .. 16 l + 15 past soffset + offset: %s (lane 1 got %.0f, %.0f)\n", bad ? "NO" : "yes", o[2], o[3]);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice);
  const double base = 2048.0;
  run<NONE>("32 MFMAs", src, out, cyc, base, 0);
  run<BUF_OFFEN>("+ 4 buffer_load_dwordx4 offen (VGPR offset)", src, out, cyc, base, 4);
  run<BUF_TID>("+ 4 buffer_load_dwordx4 off (ADD_TID descriptor)", src, out, cyc, base, 4);
  run<GLOBAL_SADDR>("+ 4 global_load_dwordx4 v, s[base]", src, out, cyc, base, 4);
  run<BUF_LDS>("+ 4 buffer_load_dwordx4 offen lds", src, out, cyc, base, 4);
  run<BUF_OFFEN_SPREAD>("+ 4 offen loads, one per 8 MFMAs", src, out, cyc, base, 4);
  run<BUF_TID_SPREAD>("+ 4 ADD_TID loads, one per 8 MFMAs", src, out, cyc, base, 4);
  run<BUF_OFFEN_X2>("+ 8 offen loads", src, out, cyc, base, 8);
  run<BUF_TID_X2>("+ 8 ADD_TID loads", src, out, cyc, base, 8);
  run<DS_BURST>("+ 4 ds_write_b128 + 4 ds_read_b128 in a burst", src, out, cyc, base, 8);
  run<DS_SPREAD>("+ the same, one per 4 MFMAs", src, out, cyc, base, 8);
  run<MIX_BURST>("+ 8 loads + 4 ds_write + 4 ds_read in a burst", src, out, cyc, base, 16);
  run<MIX_SPREAD>("+ the same 16, one per 2 MFMAs", src, out, cyc, base, 16);
  run<MIX_SPREAD2>("+ the same 16, two per 4 MFMAs", src, out, cyc, base, 16);
  run<DEP_NONE>("32 MFMAs, the kernel's accumulator pattern", src, out, cyc, base, 0);
  run<DEP_SPREAD>("+ 16 memory operations, one per 2 MFMAs", src, out, cyc, base, 16);
  run<DEP_SPREAD_SALU>("+ 16 memory + 8 scalar operations", src, out, cyc, base, 24);
  run<DEP_SPREAD_NOP>("+ 16 s_nop 0, one per 2 MFMAs", src, out, cyc, base, 16);
  return 0;
}
