// How many integer VALU instructions fit in the shadow of an fp64 MFMA of the SAME wave (one wave per SIMD)?  NV instructions behind
// every MFMA: v_add_u32 (full rate), v_mul_lo_u32 (quarter rate), v_mad_u64_u32, v_xor + v_lshrrev mixes, v_cvt / fp64 multiply.
//   hipcc --offload-arch=gfx950 -O3 valucost.hip -o valucost && ./valucost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));

#define REP1(x) x
#define REP2(x) x x
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP12(x) REP8(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)
#define MF(d) "v_mfma_f64_16x16x4_f64 " d ", %8, %9, " d "\n\t"
#define BODY(V) MF("%0") V MF("%1") V MF("%2") V MF("%3") V MF("%4") V MF("%5") V MF("%6") V MF("%7") V

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const double* __restrict__ src, double* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f64x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f64x4{0, 0, 0, 0};
  double a0 = src[lane], b0 = src[128 + lane];
  unsigned x = lane * 2654435761u + 1, y = lane + 7;
  unsigned long long w = lane;
  double f = src[lane + 256];
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int it = 0; it < iters; ++it) {
#define OPS : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]) \
            : "v"(a0), "v"(b0), "v"(x), "v"(y), "v"(w), "v"(f) : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "vcc"
#define ADD "v_add_u32 v200, %10, v200\n\t"
#define XOR "v_xor_b32 v201, %11, v201\n\t"
#define MUL "v_mul_lo_u32 v202, %10, v202\n\t"
#define MAD "v_mad_u64_u32 v[204:205], vcc, %10, %11, v[204:205]\n\t"
#define FMU "v_mul_f64 v[206:207], %13, v[206:207]\n\t"
    if (MODE == 0) asm volatile(BODY("") OPS);
    else if (MODE == 1) asm volatile(BODY(REP1(ADD)) OPS);
    else if (MODE == 2) asm volatile(BODY(REP2(ADD)) OPS);
    else if (MODE == 4) asm volatile(BODY(REP4(ADD)) OPS);
    else if (MODE == 8) asm volatile(BODY(REP8(ADD)) OPS);
    else if (MODE == 12) asm volatile(BODY(REP12(ADD)) OPS);
    else if (MODE == 16) asm volatile(BODY(REP16(ADD)) OPS);
    else if (MODE == 108) asm volatile(BODY(REP4(ADD XOR)) OPS);
    else if (MODE == 116) asm volatile(BODY(REP8(ADD XOR)) OPS);
    else if (MODE == 201) asm volatile(BODY(REP1(MUL)) OPS);
    else if (MODE == 202) asm volatile(BODY(REP2(MUL)) OPS);
    else if (MODE == 204) asm volatile(BODY(REP4(MUL)) OPS);
    else if (MODE == 301) asm volatile(BODY(REP1(MAD)) OPS);
    else if (MODE == 302) asm volatile(BODY(REP2(MAD)) OPS);
    else if (MODE == 401) asm volatile(BODY(REP1(FMU)) OPS);
    else if (MODE == 404) asm volatile(BODY(REP4(FMU)) OPS);
    else if (MODE == 1016) asm volatile(REP8(REP16(ADD)) OPS);          // no MFMA: 128 adds
    else if (MODE == 1204) asm volatile(REP8(REP4(MUL)) OPS);           // no MFMA: 32 quarter-rate multiplies
    else if (MODE == 1302) asm volatile(REP8(REP2(MAD)) OPS);           // no MFMA: 16 v_mad_u64_u32
  }
  asm volatile("s_nop 15\n\ts_nop 3\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) { cyc[2 * (blockIdx.x * 4 + wave)] = t1 - t0; cyc[2 * (blockIdx.x * 4 + wave) + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, const double* src, double* out, unsigned long long* cyc, int nv) {
  const int iters = 4000, nwg = 256;
  std::vector<unsigned long long> h(2 * nwg * 4);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<MODE>), dim3(nwg), dim3(256), 0, 0, src, out, cyc, iters);
    (void)hipDeviceSynchronize();
  }
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double c = 0, r = 0;
  for (int i = 0; i < nwg * 4; ++i) { c += h[2 * i]; r += h[2 * i + 1]; }
  const double per = c / (nwg * 4) / iters;
  printf("%-46s %8.1f cycles per 8 MFMAs", name, per);
  if (nv) printf("  = %5.2f per VALU instruction beyond 512", (per - 512.0) / nv);
  printf("   clock %.3f GHz\n", c / r * 0.1);
}

int main() {
  double *src, *out; unsigned long long* cyc;
  (void)hipMalloc(&src, 1 << 20); (void)hipMalloc(&out, 256 * 256 * 8); (void)hipMalloc(&cyc, 256 * 4 * 2 * 8);
  std::vector<double> h((1 << 20) / 8);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  (void)hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice);
  run<0>("8 MFMAs", src, out, cyc, 0);
  run<1>("+ 1 v_add_u32 behind each", src, out, cyc, 8);
  run<2>("+ 2 v_add_u32 behind each", src, out, cyc, 16);
  run<4>("+ 4 v_add_u32 behind each", src, out, cyc, 32);
  run<8>("+ 8 v_add_u32 behind each", src, out, cyc, 64);
  run<12>("+ 12 v_add_u32 behind each", src, out, cyc, 96);
  run<16>("+ 16 v_add_u32 behind each", src, out, cyc, 128);
  run<108>("+ 4 (v_add_u32, v_xor_b32) behind each", src, out, cyc, 64);
  run<116>("+ 8 (v_add_u32, v_xor_b32) behind each", src, out, cyc, 128);
  run<201>("+ 1 v_mul_lo_u32 behind each", src, out, cyc, 8);
  run<202>("+ 2 v_mul_lo_u32 behind each", src, out, cyc, 16);
  run<204>("+ 4 v_mul_lo_u32 behind each", src, out, cyc, 32);
  run<301>("+ 1 v_mad_u64_u32 behind each", src, out, cyc, 8);
  run<302>("+ 2 v_mad_u64_u32 behind each", src, out, cyc, 16);
  run<401>("+ 1 v_mul_f64 behind each", src, out, cyc, 8);
  run<404>("+ 4 v_mul_f64 behind each", src, out, cyc, 32);
  run<1016>("128 v_add_u32, no MFMA", src, out, cyc, 0);
  run<1204>("32 v_mul_lo_u32, no MFMA", src, out, cyc, 0);
  run<1302>("16 v_mad_u64_u32, no MFMA", src, out, cyc, 0);
  return 0;
}
