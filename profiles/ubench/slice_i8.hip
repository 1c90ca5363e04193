// Exploratory (round 5, verdict item 8): is there anything above the fp64 matrix pipe for the MFMA-bound 32- / 64-column sweeps?
// Error-free slicing: an fp64 tile entry a = 2^E * sum_s a_s 2^(-7 s) with 8 slices a_s of 7 bits (E = the tile's exponent), the block's
// entries x likewise per column; every slice product a_s * x_t is exact in the int8 matrix core's int32 accumulator (256 * 127^2 < 2^22),
// the pairs with s + t <= 9 (36 of 64) carry everything above 2^-56 of the tile's scale, and the 8 partial sums of equal s + t are
// folded in fp64.  Per 32 x 16 sub-block of a tile (the unit of the sweep kernels: 8 entries per lane) and 32 columns, both products:
//   fp64 pipe : 32 v_mfma_f64_16x16x4_f64            = 2048 cycles (measured: 64.0 each)
//   int8 pipe : 36 pairs x 2 products x (512 x 32 / 16384) = 72 v_mfma_i32_16x16x64_i8
// What this measures, one wave per SIMD and two, every CU busy:
//   mode 0  32 fp64 MFMAs per sub-block                                   (the baseline)
//   mode 1  72 int8 MFMAs per sub-block                                   (the matrix-core time of the sliced form)
//   mode 2  slicing 8 fp64 entries per lane into 8 int8 slices (integer route: exponent, 64-bit shift, bit fields, packing)
//   mode 3  mode 1 + mode 2 in one loop body                              (do they overlap?  fp64 MFMAs and VALU work do not)
//   hipcc --offload-arch=gfx950 -O3 slice_i8.hip -o slice_i8 && ./slice_i8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// 8 slices of 7 bits of |a| * 2^(-E) (a < 2^E), most significant first, as two dwords of four int8 each; the sign is applied to every
// slice (two's complement negate of the aligned mantissa would do the same in 2 more operations: the generator's entries are >= 0)
__device__ __forceinline__ void slice8(double a, int E, uint32_t& hi4, uint32_t& lo4) {
  const uint64_t bits = (uint64_t)__double_as_longlong(a);
  const int e = (int)((bits >> 52) & 0x7FF);
  uint64_t mant = (bits & 0x000FFFFFFFFFFFFFull) | (e ? 0x0010000000000000ull : 0ull);     // 53 bits, value = mant * 2^(e - 1075)
  // aligned to 56 fraction bits below 2^E: frac = |a| / 2^E * 2^56 = mant * 2^(e - 1075 - E + 56)
  const int sh = E + 1075 - 56 - e;                    // right shift (>= -3 for |a| < 2^E)
  uint64_t frac = sh >= 64 ? 0ull : (sh >= 0 ? (mant >> sh) : (mant << (-sh)));
  const uint32_t f_hi = (uint32_t)(frac >> 28), f_lo = (uint32_t)(frac & 0x0FFFFFFFu);     // 2 x 28 bits = 2 x four 7-bit fields
  hi4 = ((f_hi >> 21) & 0x7F) | (((f_hi >> 14) & 0x7F) << 8) | (((f_hi >> 7) & 0x7F) << 16) | ((f_hi & 0x7F) << 24);
  lo4 = ((f_lo >> 21) & 0x7F) | (((f_lo >> 14) & 0x7F) << 8) | (((f_lo >> 7) & 0x7F) << 16) | ((f_lo & 0x7F) << 24);
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const double* __restrict__ src, double* __restrict__ out, unsigned long long* __restrict__ cyc, int iters, int E) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f64x4 facc[8];
  i32x4 iacc[8];
  for (int i = 0; i < 8; ++i) { facc[i] = f64x4{0, 0, 0, 0}; iacc[i] = i32x4{0, 0, 0, 0}; }
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = src[lane + 64 * i];
  double b0 = src[512 + lane];
  i32x4 xa = {(int)lane, (int)lane * 3, 7, 11}, xb = {5, (int)lane, 13, 17};
  uint32_t packed[16];
  for (int i = 0; i < 16; ++i) packed[i] = lane + i;
  unsigned long long t0 = 0, t1 = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 32; ++j) facc[j & 7] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j & 7], b0, facc[j & 7], 0, 0, 0);
    }
    if (MODE == 2 || MODE == 3) {
      // the NEXT sub-block's 8 entries per lane -> 16 dwords of slices (kept live through the accumulator below)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        uint32_t h, l;
        slice8(a[i], E, h, l);
        packed[2 * i] ^= h; packed[2 * i + 1] += l;
        a[i] += 1e-9;                      // new data every iteration (keeps the slicing in the loop)
      }
    }
    if (MODE == 1 || MODE == 3) {
      // 72 int8 MFMAs on operands built from the packed slices (any four dwords: the issue rate does not depend on the values)
#pragma unroll
      for (int j = 0; j < 72; ++j) {
        i32x4 av = {(int)packed[(j) & 15], (int)packed[(j + 1) & 15], (int)packed[(j + 2) & 15], (int)packed[(j + 3) & 15]};
        iacc[j & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, (j & 1) ? xa : xb, iacc[j & 7], 0, 0, 0);
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 3\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  double s = 0;
  for (int i = 0; i < 8; ++i) s += facc[i][0] + facc[i][3] + (double)iacc[i][0] + (double)iacc[i][2] + a[i];
  for (int i = 0; i < 16; ++i) s += (double)packed[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
static void run(const char* what, int waves_per_simd, const double* src, double* out, unsigned long long* cyc, int E) {
  // one workgroup of 4 waves per CU and wave slot: 256 (one wave per SIMD) or 512 (two) workgroups, all resident at once
  const int iters = 2000, grid = 256 * waves_per_simd;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, src, out, cyc, 10, E);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, src, out, cyc, iters, E);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / iters;                // one sub-block per wave
  printf("%-58s %d wave(s)/SIMD: %7.3f us per sub-block and wave = %6.0f cycles at 2.4 GHz; per SIMD %6.0f cycles per sub-block\n", what,
         waves_per_simd, us, us * 2400.0, us * 2400.0 / waves_per_simd);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
  double* src; double* out; unsigned long long* cyc;
  std::vector<double> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 1e-3 * ((i * 2654435761u) % 1000003u) / 1000003.0;       // generator-like entries in [0, 1e-3)
  hipMalloc(&src, sizeof(double) * 1024); hipMalloc(&out, sizeof(double) * 256 * 4096); hipMalloc(&cyc, sizeof(unsigned long long) * 8 * 4096);
  hipMemcpy(src, h.data(), sizeof(double) * 1024, hipMemcpyHostToDevice);
  const int E = -9;                                 // 2^-9 = 1.95e-3 > every entry
  for (int w = 1; w <= 2; ++w) {
    run<0>("mode 0: 32 x v_mfma_f64_16x16x4_f64", w, src, out, cyc, E);
    run<1>("mode 1: 72 x v_mfma_i32_16x16x64_i8", w, src, out, cyc, E);
    run<2>("mode 2: slicing of 8 entries per lane into 8 x 7 bits", w, src, out, cyc, E);
    run<3>("mode 3: slicing + 72 int8 MFMAs in one loop body", w, src, out, cyc, E);
  }
  return 0;
}
