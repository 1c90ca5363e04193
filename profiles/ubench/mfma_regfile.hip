// Issue rate of v_mfma_f64_16x16x4_f64, one wave per SIMD, by register file of its operands (VGPR or accumulation
// registers) and by the distance between two MFMAs of one accumulator chain.  Cycles by s_memtime.
//   hipcc --offload-arch=gfx950 -O3 mfma_regfile.hip -o mfma_regfile && ./mfma_regfile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));

// CD: 0 = accumulators in VGPRs, 1 = in accumulation registers, 2 = alternating.  BA: B operand in accumulation registers.
// AA: A operand in accumulation registers (C/D in VGPRs).
// DIST: an accumulator is used again DIST MFMAs later (1, 2, 4, 8).
template <int CD, bool BA, int DIST, bool AA = false>
__global__ __launch_bounds__(256, 1) void k(const double* __restrict__ src, double* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f64x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f64x4{0, 0, 0, 0};
  double a = src[lane], b = src[64 + lane];
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      const int i = m % DIST;
      const bool ca = CD == 1 || (CD == 2 && (i & 1));
      if (AA) {
        if (BA) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "a"(a), "a"(b));
        else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "a"(a), "v"(b));
      } else if (ca) {
        if (BA) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "a"(b));
        else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      } else {
        if (BA) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "a"(b));
        else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 3\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

// A and B operands named literally: register numbers (and with them the register banks, number mod 4) chosen by hand
template <int AF, int A0, int BF, int B0>   // AF / BF: 0 = VGPR, 1 = accumulation register; A0 / B0: first register of the pair
__global__ __launch_bounds__(256, 1) void kb(const double* __restrict__ src, double* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f64x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f64x4{0, 0, 0, 0};
  double a = src[lane], b = src[64 + lane];
  // operands into the named registers (v[200:207] / a[200:207] are declared clobbered below, so the compiler keeps out)
  asm volatile("v_mov_b32 v200, %0\n\tv_mov_b32 v201, %1\n\tv_mov_b32 v202, %0\n\tv_mov_b32 v203, %1\n\tv_mov_b32 v204, %2\n\tv_mov_b32 v205, %3\n\tv_mov_b32 v206, %2\n\tv_mov_b32 v207, %3\n\t"
               "v_accvgpr_write_b32 a200, %0\n\tv_accvgpr_write_b32 a201, %1\n\tv_accvgpr_write_b32 a202, %0\n\tv_accvgpr_write_b32 a203, %1\n\t"
               "v_accvgpr_write_b32 a204, %2\n\tv_accvgpr_write_b32 a205, %3\n\tv_accvgpr_write_b32 a206, %2\n\tv_accvgpr_write_b32 a207, %3\n\ts_nop 4"
               :: "v"((unsigned)__double2loint(a)), "v"((unsigned)__double2hiint(a)), "v"((unsigned)__double2loint(b)), "v"((unsigned)__double2hiint(b))
               : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207");
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      if (AF == 0 && BF == 0) asm volatile("v_mfma_f64_16x16x4_f64 %0, v[%c1:%c2], v[%c3:%c4], %0" : "+v"(acc[m % 8]) : "i"(A0), "i"(A0 + 1), "i"(B0), "i"(B0 + 1));
      if (AF == 1 && BF == 1) asm volatile("v_mfma_f64_16x16x4_f64 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(acc[m % 8]) : "i"(A0), "i"(A0 + 1), "i"(B0), "i"(B0 + 1));
      if (AF == 0 && BF == 1) asm volatile("v_mfma_f64_16x16x4_f64 %0, v[%c1:%c2], a[%c3:%c4], %0" : "+v"(acc[m % 8]) : "i"(A0), "i"(A0 + 1), "i"(B0), "i"(B0 + 1));
    }
  }
  asm volatile("s_nop 15\n\ts_nop 3\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}
template <int AF, int A0, int BF, int B0>
void runb(const char* name, const double* src, double* out, unsigned long long* cyc) {
  const int iters = 2000, nwg = 256;
  std::vector<unsigned long long> h(nwg * 4);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((kb<AF, A0, BF, B0>), dim3(nwg), dim3(256), 0, 0, src, out, cyc, iters);
    (void)hipDeviceSynchronize();
  }
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double c = 0;
  for (int i = 0; i < nwg * 4; ++i) c += h[i];
  printf("%-64s %6.2f cycles per MFMA\n", name, c / (nwg * 4) / iters / 32);
}

template <int CD, bool BA, int DIST, bool AA = false>
void run(const char* name, const double* src, double* out, unsigned long long* cyc) {
  const int iters = 2000, nwg = 256;
  std::vector<unsigned long long> h(nwg * 4);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<CD, BA, DIST, AA>), dim3(nwg), dim3(256), 0, 0, src, out, cyc, iters);
    (void)hipDeviceSynchronize();
  }
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double c = 0;
  for (int i = 0; i < nwg * 4; ++i) c += h[i];
  printf("%-64s %6.2f cycles per MFMA\n", name, c / (nwg * 4) / iters / 32);
}

int main() {
  double *src, *out; unsigned long long* cyc;
  (void)hipMalloc(&src, 4096); (void)hipMalloc(&out, 256 * 256 * 8); (void)hipMalloc(&cyc, 256 * 4 * 8);
  std::vector<double> h(512);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  (void)hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
  run<0, false, 8>("C/D VGPR, B VGPR, chain distance 8", src, out, cyc);
  run<1, false, 8>("C/D acc regs, B VGPR, chain distance 8", src, out, cyc);
  run<0, true, 8>("C/D VGPR, B acc reg, chain distance 8", src, out, cyc);
  run<1, true, 8>("C/D acc regs, B acc reg, chain distance 8", src, out, cyc);
  run<2, false, 8>("C/D alternating VGPR / acc regs, B VGPR, chain distance 8", src, out, cyc);
  run<0, false, 4>("C/D VGPR, B VGPR, chain distance 4", src, out, cyc);
  run<0, false, 2>("C/D VGPR, B VGPR, chain distance 2", src, out, cyc);
  run<0, false, 1>("C/D VGPR, B VGPR, chain distance 1 (dependent)", src, out, cyc);
  run<1, false, 4>("C/D acc regs, B VGPR, chain distance 4", src, out, cyc);
  run<1, false, 2>("C/D acc regs, B VGPR, chain distance 2", src, out, cyc);
  run<1, false, 1>("C/D acc regs, B VGPR, chain distance 1 (dependent)", src, out, cyc);
  run<0, false, 8, true>("C/D VGPR, A acc reg, B VGPR, chain distance 8", src, out, cyc);
  run<0, true, 8, true>("C/D VGPR, A and B acc regs, chain distance 8", src, out, cyc);
  run<0, true, 1, true>("C/D VGPR, A and B acc regs, chain distance 1", src, out, cyc);
  runb<1, 200, 1, 204>("A a[200:201], B a[204:205] (same banks)", src, out, cyc);
  runb<1, 200, 1, 206>("A a[200:201], B a[206:207] (different banks)", src, out, cyc);
  runb<0, 200, 0, 204>("A v[200:201], B v[204:205] (same banks)", src, out, cyc);
  runb<0, 200, 0, 206>("A v[200:201], B v[206:207] (different banks)", src, out, cyc);
  runb<0, 200, 1, 204>("A v[200:201], B a[204:205] (same banks, different halves)", src, out, cyc);
  return 0;
}
