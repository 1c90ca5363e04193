// What overlaps with v_mfma_f64_16x16x4_f64 on gfx950?  One wave per SIMD (256 threads / WG, 1 WG / CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

template <int MODE, int NX>
__global__ __launch_bounds__(256, 1) void k(double* out, const double* in, long long* cyc, int iters) {
  __shared__ double lds[4096];
  f64x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f64x4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
  __syncthreads();
  const double* gp = in + threadIdx.x * 2;
  double g[8];
  for (int i = 0; i < 8; ++i) g[i] = 0;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE != 9) acc[u & 3] = MFMA(a, b, acc[u & 3]);
      if (MODE == 1 || MODE == 9) {          // NX independent f64 adds per MFMA
#pragma unroll
        for (int x = 0; x < NX; ++x) v[(u * NX + x) & 7] += 1.0;
      }
      if (MODE == 2) {                        // NX int VALU ops per MFMA
#pragma unroll
        for (int x = 0; x < NX; ++x) { int t = __builtin_bit_cast(long long, v[x & 7]); t = t * 3 + u; v[x & 7] = __builtin_bit_cast(double, (long long)t | (__builtin_bit_cast(long long, v[x & 7]) & ~0xffffffffll)); }
      }
      if (MODE == 3) {                        // NX ds_read_b128 per MFMA
#pragma unroll
        for (int x = 0; x < NX; ++x) {
          double2 r = *reinterpret_cast<double2*>(&lds[((threadIdx.x & 63) * 2 + ((u * NX + x) & 7) * 128) & 4095]);
          g[(u * NX + x) & 7] += r.x;
        }
      }
      if (MODE == 4) {                        // NX global_load_dwordx4 per MFMA (L2 hits)
#pragma unroll
        for (int x = 0; x < NX; ++x) {
          double2 r = *reinterpret_cast<const double2*>(gp + (((it * 16 + u) * NX + x) & 1023) * 512);
          g[(u * NX + x) & 7] += r.x;
        }
      }
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i] + g[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int NX>
void run(const char* name, double* out, double* in, long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<MODE, NX>), dim3(256), dim3(256), 0, 0, out, in, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NX>), dim3(256), dim3(256), 0, 0, out, in, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-28s NX=%d  %8.1f ticks per 16-MFMA group (%.1f per MFMA), %.3f ms\n", name, NX, (double)c / iters, (double)c / iters / 16, ms);
}

int main() {
  double *out, *in; long long* cyc;
  hipMalloc(&out, 256 * 256 * 8); hipMalloc(&in, 8 << 20); hipMemset(in, 0, 8 << 20); hipMalloc(&cyc, 8);
  run<0, 0>("mfma only", out, in, cyc);
  run<9, 4>("f64 add only (no mfma)", out, in, cyc);
  run<1, 1>("mfma + f64 add", out, in, cyc); run<1, 2>("mfma + f64 add", out, in, cyc); run<1, 4>("mfma + f64 add", out, in, cyc); run<1, 8>("mfma + f64 add", out, in, cyc);
  run<2, 2>("mfma + int valu", out, in, cyc); run<2, 4>("mfma + int valu", out, in, cyc);
  run<3, 1>("mfma + ds_read_b128", out, in, cyc); run<3, 2>("mfma + ds_read_b128", out, in, cyc); run<3, 4>("mfma + ds_read_b128", out, in, cyc);
  run<4, 1>("mfma + global_load_x4", out, in, cyc); run<4, 2>("mfma + global_load_x4", out, in, cyc);
  return 0;
}
