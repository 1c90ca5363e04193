// A caller's own matrix-free operator, written the way a user of the device interface writes it (include/davidson_hip.h:
// dav_set_operator_device / dav_device_apply_fn): its own HIP kernel on the stream the engine hands over, no library of ours.
// Test infrastructure (tests/test_device_operator_gpu.py) and the example INTEGRATION.md points to.
//   Op = diag(d0 + dstep * i) + eps * (first neighbours) + eps / 2 * (second neighbours)     - a banded, symmetric stencil
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC user_operator.hip -o libuser_operator.so
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>

struct UserOp { double d0, dstep, eps; };

__global__ __launch_bounds__(256) void stencil_kernel(UserOp op, int64_t n, int64_t row0, int64_t nloc, const double* __restrict__ x, int64_t ldx,
                                                      double* __restrict__ y, int64_t ldy) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;        // row of this rank's slab
  if (i >= nloc) return;
  const int64_t g = row0 + i;                                        // its global index: x holds all n rows
  const double* xc = x + (int64_t)blockIdx.y * ldx;
  double v = (op.d0 + op.dstep * (double)g) * xc[g];
  if (g >= 1) v += op.eps * xc[g - 1];
  if (g + 1 < n) v += op.eps * xc[g + 1];
  if (g >= 2) v += 0.5 * op.eps * xc[g - 2];
  if (g + 2 < n) v += 0.5 * op.eps * xc[g + 2];
  y[(int64_t)blockIdx.y * ldy + i] = v;
}

extern "C" void* user_op_create(double d0, double dstep, double eps) {
  UserOp* op = (UserOp*)malloc(sizeof(UserOp));
  op->d0 = d0; op->dstep = dstep; op->eps = eps;
  return op;
}
extern "C" void user_op_destroy(void* ctx) { free(ctx); }

// dav_device_apply_fn
extern "C" int user_op_apply(void* ctx, void* hip_stream, int64_t n, int64_t row0, int64_t nloc, int k, const double* x_dev, int64_t ldx,
                             double* y_dev, int64_t ldy) {
  if (nloc <= 0 || k <= 0) return 0;
  hipLaunchKernelGGL(stencil_kernel, dim3((unsigned)((nloc + 255) / 256), k), dim3(256), 0, (hipStream_t)hip_stream, *(const UserOp*)ctx, n, row0,
                     nloc, x_dev, ldx, y_dev, ldy);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
