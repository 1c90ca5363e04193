// v_mfma_f64_4x4x4_4b_f64 on gfx950: issue rate against v_mfma_f64_16x16x4_f64, and the lane layout of its operands.
// hipcc --offload-arch=gfx950 -O3 mfma4x4.hip -o mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0)

// MODE 0: 16x16x4, NACC independent accumulators; MODE 1: 4x4x4
template <int MODE, int NACC, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void rate(double* out, long long* cyc, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  f64x4 acc16[NACC];
  double acc4[NACC];
  for (int i = 0; i < NACC; ++i) { acc16[i] = f64x4{0, 0, 0, 0}; acc4[i] = 0; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      if (MODE == 0) acc16[u % NACC] = MFMA16(a, b, acc16[u % NACC]);
      else acc4[u % NACC] = MFMA4(a, b, acc4[u % NACC]);
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3] + acc4[i];
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int NACC, int WAVES>
void run(const char* name, double* out, long long* cyc) {
  const int iters = 1000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((rate<MODE, NACC, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
  }
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-12s accumulators=%2d waves/SIMD=%d : %7.2f ticks per MFMA (one wave's stream)\n", name, NACC, WAVES / 4,
         (double)c / iters / 32);
}

__global__ void layout(const double* a, const double* b, double* d) {
  d[threadIdx.x] = MFMA4(a[threadIdx.x], b[threadIdx.x], 0.0);
}

int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 8);
  run<0, 1, 4>("16x16x4", out, cyc); run<0, 4, 4>("16x16x4", out, cyc); run<0, 8, 4>("16x16x4", out, cyc); run<0, 4, 8>("16x16x4", out, cyc);
  run<1, 1, 4>("4x4x4", out, cyc); run<1, 2, 4>("4x4x4", out, cyc); run<1, 4, 4>("4x4x4", out, cyc); run<1, 8, 4>("4x4x4", out, cyc);
  run<1, 16, 4>("4x4x4", out, cyc); run<1, 4, 8>("4x4x4", out, cyc); run<1, 8, 8>("4x4x4", out, cyc);

  // layout: A = delta at lane la, B[lane] = 1 + lane  ->  D[l] = B[lane that pairs with la for output l]
  double *da, *db, *dd;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
  std::vector<double> ha(64), hb(64), hd(64);
  for (int l = 0; l < 64; ++l) hb[l] = 1 + l;
  hipMemcpy(db, hb.data(), 512, hipMemcpyHostToDevice);
  printf("A lane -> (D lane : B lane) pairs\n");
  for (int la = 0; la < 64; ++la) {
    for (int l = 0; l < 64; ++l) ha[l] = l == la ? 1.0 : 0.0;
    hipMemcpy(da, ha.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd.data(), dd, 512, hipMemcpyDeviceToHost);
    printf("A[%2d]:", la);
    for (int l = 0; l < 64; ++l) if (hd[l] != 0.0) printf(" (%d:%d)", l, (int)hd[l] - 1);
    printf("\n");
  }
  return 0;
}
