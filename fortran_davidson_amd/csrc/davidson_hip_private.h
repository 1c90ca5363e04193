/* davidson_hip_private.h - entry points of libdavidson_hip.so that are NOT part of the drop-in ABI (include/davidson_hip.h):
 * measurement doors of bench.py / the profiling scripts, and the doors of the TEST build (lib/test/libdavidson_hip.so,
 * -DDAV_TEST_TRANSPORTS=1) that let several ranks of one problem share the one GPU of the test box.  Nothing here has a
 * counterpart in the reference; a host program that replaces the reference's BLAS/LAPACK calls never needs them. */
#ifndef DAVIDSON_HIP_PRIVATE_H
#define DAVIDSON_HIP_PRIVATE_H
#include "../../include/davidson_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- measurement (both builds) ------------------------------------------------------------------------------------ */
/* Time `reps` block applies of A on k columns with HIP events on the engine's stream (inputs resident).
 * Returns the average milliseconds per apply END TO END (operand packing + block-matvec kernel + reduction
 * of the partial sums - everything that produces W from V) and the algorithmic bytes per apply
 * (8*S + 16*N*k).  dav_bench_apply2 also returns the average of the block-matvec kernel
 * alone (kernel_ms) and the flops per apply. */
int dav_bench_apply(dav_handle_t h, int which, int k, int reps, double* avg_ms, double* bytes);
int dav_bench_apply2(dav_handle_t h, int which, int k, int reps, double* avg_ms, double* kernel_ms, double* bytes,
                     double* flops);
/* What the HBM of this box delivers to a plain streaming kernel (16 B per lane): device copy a = b and triad a = b + s c over
 * three arrays of `doubles` entries (0 = 2^28, i.e. 2 GiB each), read + written GB/s - the measured counterpart of the data
 * sheet's 8 TB/s that every HBM fraction of bench.py is also quoted against. */
int dav_bench_stream(dav_handle_t h, int64_t doubles, int reps, double* copy_GBps, double* triad_GBps);
/* The same, plus the rate of a kernel that only READS (two of the arrays, same access pattern, one partial sum per workgroup
 * written): the practical roof of the operator sweeps, which read 8*S bytes and write 8*N*k. */
int dav_bench_stream3(dav_handle_t h, int64_t doubles, int reps, double* copy_GBps, double* triad_GBps, double* read_GBps);
/* Entries per second of the arithmetic of the reference's matrix-free test operator (one fp64 atan2, sqrt, log and cos per matrix
 * entry, src/tests/test_utils.f90:72-116) on register operands, two waves per SIMD, no memory traffic: the measured roof of the
 * generated sweeps of that operator (bench.py: benchmark_free leg). */
int dav_bench_harness_rate(dav_handle_t h, int iters, double* entries_per_s);
/* dav_apply as the GJD correction solve calls it (an "inner" sweep: may read the fp32 copy of the stored tiles,
 * dav_set_inner_precision) - so that the parity tests can compare that sweep with the oracle directly. */
int dav_apply_inner(dav_handle_t h, int which, int src_panel, int c0, int k, int dst_panel, int d0);
/* Fraction of the block rows of a generated second operator that is kept resident as stored tiles (configs[3]); 0 when nothing is. */
int dav_resident_fraction(dav_handle_t h, int which, double* fraction);
/* what the buffer cache (include/davidson_hip.h: dav_free_buffers) holds right now: idle device blocks, idle page-locked host blocks */
int dav_buffer_cache_held(int64_t* device_bytes, int64_t* pinned_bytes);

/* The host-side packing of a small matrix into the MFMA-B operand image the panel kernel reads (csrc/kernels.h: pg_image_index:
 * entry (i, j) at ((i / 4) * tiles_per_step + j / 16) * 64 + (j % 16) + 16 * (i % 4)); no GPU needed.  out == NULL: sizes only. */
int dav_pack_operand_image(const double* src, int64_t ld, int p, int q, double* out, int64_t* doubles_out, int64_t* tiles_per_step_out);

/* ---- TEST build only ---------------------------------------------------------------------------------------------- */
/* Test transport: the n engines (created with rank r of n, same process, same GPU) exchange through
 * device copies and thread barriers instead of RCCL; each rank must then be driven by its own thread. */
int dav_local_group_join(dav_handle_t* handles, int n);
/* DAV_TEST_SERIALIZE=1 at dav_local_group_join: between two collectives the ranks' threads run ONE AT A TIME, in rank order, so
 * that a rank's HIP-event times are those of a rank that owns a GPU.  A thread that has finished its work calls this to pass the
 * turn on (the ranks behind it would wait for ever otherwise). */
int dav_local_group_yield(dav_handle_t h);
/* Second test transport: ranks are PROCESSES sharing one GPU; collectives go through the POSIX shared-memory
 * segment `name` ("/something", created by rank 0).  Exercises the complete multi-process launch flow
 * (one engine per process, as under torch.distributed.run) on a single-GPU box. */
int dav_comm_init_shm(dav_handle_t h, const char* name);

#ifdef __cplusplus
}
#endif
#endif
