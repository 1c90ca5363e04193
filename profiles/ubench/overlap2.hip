// Two waves per SIMD: waves 0-3 issue f64 MFMAs only; waves 4-7 issue another instruction type only.
// Does the second wave's work proceed in the shadow of the first wave's MFMAs?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(double* out, const double* in, long long* cyc, int iters, int mfma_on) {
  __shared__ double lds[8192];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i;
  __syncthreads();
  double s = 0;
  long long t0 = clock64();
  if (wave < 4) {
    if (mfma_on) {
      f64x4 acc[4];
      for (int i = 0; i < 4; ++i) acc[i] = f64x4{0, 0, 0, 0};
      double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 3] = MFMA(a, b, acc[u & 3]);
      }
      for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
  } else {
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane + i;
    const double* gp = in + (size_t)(blockIdx.x & 7) * 65536 + lane * 2;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 64; ++u) {
        if (MODE == 1) v[u & 7] += 1.0;
        if (MODE == 2) { double2 r = *reinterpret_cast<double2*>(&lds[(lane * 2 + (u & 7) * 128 + (wave - 4) * 1024) & 8191]); v[u & 7] += r.x; }
        if (MODE == 3) { double2 r = *reinterpret_cast<const double2*>(gp + ((it * 64 + u) & 127) * 128); v[u & 7] += r.x; }
        if (MODE == 4) { int t = (int)v[u & 7]; asm volatile("v_add_u32 %0, %0, 1" : "+v"(t)); v[u & 7] = t; }
      }
    }
    for (int i = 0; i < 8; ++i) s += v[i];
  }
  long long t1 = clock64();
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (lane == 0 && blockIdx.x == 0 && (wave == 0 || wave == 4)) cyc[wave >> 2] = t1 - t0;
}

template <int MODE>
void run(const char* name, double* out, double* in, long long* cyc) {
  const int iters = 1000;
  for (int mf = 0; mf < 2; ++mf) {
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, in, cyc, iters, mf);
    hipDeviceSynchronize();
    long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-22s mfma wave %s: mfma wave %7.1f ticks/MFMA | other wave %7.2f ticks per op\n", name, mf ? "ON " : "off",
           mf ? (double)c[0] / iters / 16 : 0.0, (double)c[1] / iters / 64);
  }
}

int main() {
  double *out, *in; long long* cyc;
  hipMalloc(&out, 256 * 512 * 8); hipMalloc(&in, 8 << 20); hipMemset(in, 0, 8 << 20); hipMalloc(&cyc, 16);
  run<1>("f64 add", out, in, cyc);
  run<4>("int add (asm)", out, in, cyc);
  run<2>("ds_read_b128 + add", out, in, cyc);
  run<3>("global_load_x4 + add", out, in, cyc);
  return 0;
}
