// (fatwave.hip with the accumulators in VGPRs: the MFMAs issue every 64 cycles and nothing hides behind them)
// What the instructions of a half-step cost a wave that is ALONE on its SIMD (matvec_symw_kernel's regime): 32 fp64 MFMAs
// (64 cycles each if nothing else is issued) plus, per variant, the 4 tile loads, the LDS transposition (4 ds_write_b128 +
// 4 ds_read_b128), the compiler's s_nop after asm statements.  Cycles per iteration by s_memtime, clock by s_memrealtime.
//   hipcc --offload-arch=gfx950 -O3 fatwave.hip -o fatwave && ./fatwave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA_A(d, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b))
#define MFMA_V(d, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b))

// LOADS: 4 buffer_load_dwordx4 per iteration (L2-resident source); LDS: transposition writes + reads; ONEASM: the 32 MFMAs in ONE
// asm statement (no compiler padding between them)
template <bool LOADS, bool LDS, bool ONEASM, int NSALU = 0, int NVALU = 0>
__global__ __launch_bounds__(256, 1) void k(const double* __restrict__ src, double* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
  __shared__ __attribute__((aligned(16))) double tr[4 * 16 * 34];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  double* tw = tr + wave * 16 * 34;
  f64x4 acc[8], z[2];
  for (int i = 0; i < 8; ++i) acc[i] = f64x4{0, 0, 0, 0};
  z[0] = z[1] = f64x4{0, 0, 0, 0};
  f64x2 a[4], p[4];
  double b[8], x[8];
  for (int i = 0; i < 8; ++i) { b[i] = src[lane + 64 * i]; x[i] = src[512 + lane + 64 * i]; }
  for (int u = 0; u < 4; ++u) { a[u] = *reinterpret_cast<const f64x2*>(src + 1024 + 2 * lane + 128 * u); p[u] = a[u]; }
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(src) + blockIdx.x % 8 * 8192, 0, 65536, 0x00020000);
  unsigned voff[4];
  for (int u = 0; u < 4; ++u) voff[u] = ((4 * u + g) * 256 + 2 * c) * 8;
  unsigned sacc = 0, vacc = lane;
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int it = 0; it < iters; ++it) {
    f64x2 an[4];
    if (LOADS) {
#pragma unroll
      for (int u = 0; u < 4; ++u) an[u] = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, voff[u] + (it & 3) * 256, 0, 0));
    }
    if (LDS) {
#pragma unroll
      for (int u = 0; u < 4; ++u) *reinterpret_cast<f64x2*>(tw + (4 * u + g) * 34 + 2 * c) = a[u];
#pragma unroll
      for (int u = 0; u < 4; ++u) p[u] = *reinterpret_cast<const f64x2*>(tw + c * 34 + 16 * (u >> 1) + 4 * g + 2 * (u & 1));
    }
    if (NSALU > 0) {
#pragma unroll
      for (int i = 0; i < NSALU; ++i) asm volatile("s_add_u32 s90, s90, 1\n\ts_cselect_b32 s91, s90, s91" ::: "s90", "s91", "scc");   // 2 scalar ALU instructions
    }
    if (NVALU > 0) {
#pragma unroll
      for (int i = 0; i < NVALU; ++i) asm volatile("v_add_u32 %0, %0, 1" : "+v"(vacc));
    }
    if (ONEASM) {
      asm volatile(
          "v_mfma_f64_16x16x4_f64 %0, %10, %18, %0\n\tv_mfma_f64_16x16x4_f64 %8, %14, %22, %8\n\t"
          "v_mfma_f64_16x16x4_f64 %1, %10, %19, %1\n\tv_mfma_f64_16x16x4_f64 %9, %14, %23, %9\n\t"
          "v_mfma_f64_16x16x4_f64 %2, %11, %18, %2\n\tv_mfma_f64_16x16x4_f64 %8, %15, %24, %8\n\t"
          "v_mfma_f64_16x16x4_f64 %3, %11, %19, %3\n\tv_mfma_f64_16x16x4_f64 %9, %15, %25, %9\n\t"
          "v_mfma_f64_16x16x4_f64 %4, %12, %20, %4\n\tv_mfma_f64_16x16x4_f64 %8, %16, %22, %8\n\t"
          "v_mfma_f64_16x16x4_f64 %5, %12, %21, %5\n\tv_mfma_f64_16x16x4_f64 %9, %16, %23, %9\n\t"
          "v_mfma_f64_16x16x4_f64 %6, %13, %20, %6\n\tv_mfma_f64_16x16x4_f64 %8, %17, %24, %8\n\t"
          "v_mfma_f64_16x16x4_f64 %7, %13, %21, %7\n\tv_mfma_f64_16x16x4_f64 %9, %17, %25, %9\n\t"
          "v_mfma_f64_16x16x4_f64 %0, %10, %18, %0\n\tv_mfma_f64_16x16x4_f64 %8, %14, %22, %8\n\t"
          "v_mfma_f64_16x16x4_f64 %1, %10, %19, %1\n\tv_mfma_f64_16x16x4_f64 %9, %14, %23, %9\n\t"
          "v_mfma_f64_16x16x4_f64 %2, %11, %18, %2\n\tv_mfma_f64_16x16x4_f64 %8, %15, %24, %8\n\t"
          "v_mfma_f64_16x16x4_f64 %3, %11, %19, %3\n\tv_mfma_f64_16x16x4_f64 %9, %15, %25, %9\n\t"
          "v_mfma_f64_16x16x4_f64 %4, %12, %20, %4\n\tv_mfma_f64_16x16x4_f64 %8, %16, %22, %8\n\t"
          "v_mfma_f64_16x16x4_f64 %5, %12, %21, %5\n\tv_mfma_f64_16x16x4_f64 %9, %16, %23, %9\n\t"
          "v_mfma_f64_16x16x4_f64 %6, %13, %20, %6\n\tv_mfma_f64_16x16x4_f64 %8, %17, %24, %8\n\t"
          "v_mfma_f64_16x16x4_f64 %7, %13, %21, %7\n\tv_mfma_f64_16x16x4_f64 %9, %17, %25, %9"
          : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(z[0]), "+v"(z[1])
          : "v"(a[0].x), "v"(a[0].y), "v"(a[1].x), "v"(a[1].y), "v"(p[0].x), "v"(p[0].y), "v"(p[1].x), "v"(p[1].y), "v"(b[0]), "v"(b[1]), "v"(b[2]),
            "v"(b[3]), "a"(x[0]), "a"(x[1]), "a"(x[2]), "a"(x[3]));
    } else {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          MFMA_A(acc[2 * u], u < 2 ? a[u].x : a[u - 2].y, b[2 * (u >> 1)]);
          MFMA_V(z[0], u < 2 ? p[u].x : p[u - 2].y, x[u & 1]);
          MFMA_A(acc[2 * u + 1], u < 2 ? a[u].x : a[u - 2].y, b[2 * (u >> 1) + 1]);
          MFMA_V(z[1], u < 2 ? p[u].x : p[u - 2].y, x[2 + (u & 1)]);
        }
    }
    if (LOADS) {
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = an[u];
    }
  }
  asm volatile("s_nop 15\n\ts_nop 3\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  s += z[0][1] + z[1][2] + p[0].x + (double)sacc + (double)vacc;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) { cyc[2 * (blockIdx.x * 4 + wave)] = t1 - t0; cyc[2 * (blockIdx.x * 4 + wave) + 1] = r1 - r0; }
}

template <bool LOADS, bool LDS, bool ONEASM, int NSALU = 0, int NVALU = 0>
void run(const char* name, const double* src, double* out, unsigned long long* cyc, int nmfma) {
  const int iters = 4000, nwg = 256;
  std::vector<unsigned long long> h(2 * nwg * 4);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<LOADS, LDS, ONEASM, NSALU, NVALU>), dim3(nwg), dim3(256), 0, 0, src, out, cyc, iters);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double c = 0, r = 0;
  for (int i = 0; i < nwg * 4; ++i) { c += h[2 * i]; r += h[2 * i + 1]; }
  printf("%-34s %8.1f cycles per iteration = %6.2f per MFMA   clock %.3f GHz\n", name, c / (nwg * 4) / iters, c / (nwg * 4) / iters / nmfma,
         c / r * 0.1);
}

int main() {
  double *src, *out; unsigned long long* cyc;
  hipMalloc(&src, 1 << 20); hipMalloc(&out, 256 * 256 * 8); hipMalloc(&cyc, 256 * 4 * 2 * 8);
  std::vector<double> h((1 << 20) / 8);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice);
  run<false, false, true>("32 MFMAs, one asm statement", src, out, cyc, 32);
  run<false, false, false>("32 MFMAs, one statement each", src, out, cyc, 32);
  run<true, false, true>("+ 4 buffer_load_dwordx4", src, out, cyc, 32);
  run<false, true, true>("+ 4 ds_write_b128 + 4 ds_read_b128", src, out, cyc, 32);
  run<true, true, true>("+ loads + LDS transposition", src, out, cyc, 32);
  run<true, true, false>("the same, one statement per MFMA", src, out, cyc, 32);
  run<false, false, true, 16, 0>("+ 32 scalar ALU instructions", src, out, cyc, 32);
  run<false, false, true, 0, 16>("+ 16 v_add_u32", src, out, cyc, 32);
  return 0;
}
