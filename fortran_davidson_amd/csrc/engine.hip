// Engine + C ABI (include/davidson_hip.h).  Owns every N-long object in HBM and sequences the kernels
// of k_*.hip on one HIP stream; the host (Fortran driver) keeps only m x m matrices.
#include "../../include/davidson_hip.h"
#include "kernels.h"
#include "ingest.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <ctime>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(const std::string& msg) {
  g_err = msg;
  return 1;
}
#define HIPCHK(call)                                                                                  \
  do {                                                                                                \
    hipError_t e_ = (call);                                                                           \
    if (e_ != hipSuccess) {                                                                           \
      (void)hipGetLastError(); /* reported here: do not leave it for a later hipGetLastError() */     \
      return fail(std::string(#call) + " failed: " + hipGetErrorString(e_) + " (" __FILE__ ":" +      \
                  std::to_string(__LINE__) + ")");                                                    \
    }                                                                                                 \
  } while (0)
#define CHK(call)            \
  do {                       \
    int r_ = (call);         \
    if (r_ != 0) return r_;  \
  } while (0)

// ---- RCCL, loaded lazily so that single-GPU use never touches it ------------------------------------
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_rccl;
static int rccl_load() {
  if (g_rccl.lib) return 0;
  void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!lib) return fail(std::string("cannot load librccl.so: ") + dlerror());
#define SYM(field, name)                                              \
  *(void**)(&g_rccl.field) = dlsym(lib, name);                        \
  if (!g_rccl.field) return fail(std::string("librccl.so lacks ") + name);
  SYM(GetUniqueId, "ncclGetUniqueId")
  SYM(CommInitRank, "ncclCommInitRank")
  SYM(CommDestroy, "ncclCommDestroy")
  SYM(AllGather, "ncclAllGather")
  SYM(AllReduce, "ncclAllReduce")
  SYM(Broadcast, "ncclBroadcast")
  SYM(ReduceScatter, "ncclReduceScatter")
  SYM(GroupStart, "ncclGroupStart")
  SYM(GroupEnd, "ncclGroupEnd")
  SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  g_rccl.lib = lib;
  return 0;
}
#define NCCLCHK(call)                                                                              \
  do {                                                                                             \
    ncclResult_t r_ = (call);                                                                      \
    if (r_ != ncclSuccess) return fail(std::string(#call) + " failed: " + g_rccl.GetErrorString(r_)); \
  } while (0)

struct LocalGroup;
struct ShmGroup;

// ------------------------------------------------------------------------------------------------
struct OpDesc {
  int kind = DAV_KIND_NONE;
  double* a = nullptr;       // dense: nloc_pad x ncols_pad, column-major, lda = nloc_pad
  uint64_t seed = 0;
  double sparsity = 0;
  int use_diag = 0;
  double diag_val = 0;
  int trig = 0;
  double* e_table = nullptr; // device
  double* diag = nullptr;    // device, nloc_pad (local rows)
  int storage = 0;           // dense: 0 = full, 1 = symmetric-tiled (lower block triangle)
  float* a32 = nullptr;      // fp32 copy of the symmetric tiles: operand of the mixed-precision inner sweeps (lazy)
  bool a32_valid = false, a32_refused = false;
};

struct SmallBuf {            // device small matrix + pinned staging
  double* dev = nullptr;
  double* host = nullptr;
  hipEvent_t done = nullptr;
  bool pending = false;
};

constexpr int N_SMALL = 4;
constexpr int N_EVPAIRS = 64;

struct Watchdog;
struct dav_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  int64_t n = 0, nslab = 0, nloc = 0, row0 = 0, nloc_pad = 0, ncols_pad = 0;
  int rank = 0, nranks = 1, gev = 0;
  int max_cols = 0, cols_alloc = 0;
  int m = 0;
  double* panel[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int64_t ldp = 0;
  double* xt = nullptr;
  int64_t xt_group_stride = 0;
  double* scratch = nullptr;
  size_t scratch_doubles = 0;
  double* gram_dev = nullptr;     // result of gram / norms on device
  double* gram_host = nullptr;    // pinned, device-visible (zero-copy target of the reduction kernels)
  double* gram_host_dev = nullptr;  // device address of gram_host
  size_t gram_doubles = 0;
  double* gather_dev = nullptr;   // nranks*nslab staging for panel_get / diagonal gather
  int64_t* idx_dev = nullptr;
  double* norm_partial = nullptr;
  double* gjd_ws = nullptr;       // GJD inner-solver workspace (lazy)
  int storage = 0;                // storage mode for dense operators set after dav_set_storage
  int sym_nb = 0, sym_nitems = 0; // symmetric-tiled sweep: block rows, work items (runs of tiles)
  int* sym_items = nullptr;       // device: (I, J0, J1) per item
  int* sym_row_begin = nullptr;   // device: first item of each block row (nb + 1)
  double* sym_slab = nullptr;     // device: direct slabs (per item) followed by transposed slabs (per tile)
  size_t sym_slab_doubles = 0;    // grown on demand: what the largest launch so far needed (schedule x column groups)
  bool sym_no_pair = false;       // paired 32-column launches did not fit the memory: 16 columns per launch
  bool sym_no_quad = false;       // ... four column groups (64 columns) per launch did not
  int inner_bits = 64;            // 32: the sweeps INSIDE the GJD correction read an fp32 copy of the stored tiles (dav_set_inner_precision)
  // Several ranks: the lower block triangle is dealt out by groups of 4 block rows (what every schedule's super rows
  // nest in), longest group first to the least loaded rank (sym_group_owners).  row_off[I] = first tile of block row I
  // in this rank's storage, -1 = another rank's.
  std::vector<int64_t> sym_row_off_h;
  int64_t* sym_row_off = nullptr; // device copy
  int64_t sym_ntiles_local = 0;
  double* sym_wpart = nullptr;    // several ranks: this rank's partial of the whole product, [rank][column][row of its slab]
  double* sym_wrecv = nullptr;    // ... and the summed chunk the reduce-scatter hands back (nslab x 32)
  // RCCL only: a second stream for the collectives of the symmetric sweep, so that the all-gather of the NEXT 32 columns
  // and the reduce-scatter of the PREVIOUS ones run under the sweep of the current ones; buffers alternate by chunk parity
  Watchdog* wd = nullptr;         // watches the RCCL collectives of this engine (dav_comm_init)
  int group_depth = 0;            // inside ncclGroupStart / ncclGroupEnd: the group is marked once, at its end
  long iter_hint = -1;            // outer iteration the driver is in (dav_ranks_agree), for the watchdog's message
  hipStream_t comm_stream = nullptr;
  bool ov_ready = false;          // stream, events and buffers of apply_sym_overlapped all exist
  hipEvent_t ov_packed[2] = {nullptr, nullptr}, ov_gathered[2] = {nullptr, nullptr}, ov_reduced[2] = {nullptr, nullptr},
             ov_scattered[2] = {nullptr, nullptr};
  double* sym_wpart2[2] = {nullptr, nullptr};
  double* sym_wrecv2[2] = {nullptr, nullptr};
  // super-row schedules (k_matvec_sym9.hip): plan p = 0 / 1 for R = 2 / 4 block rows per workgroup
  struct SymPlan {
    int R = 0, nitems = 0, nsuper = 0;
    int64_t zslots = 0;               // transposed-partial slots: one per (super row, tile column below its last block row)
    int* items = nullptr;             // device: (super row, J0, J1, slab slot) per item, longest first
    int* row_begin = nullptr;         // device: first item of each super row (nsuper + 1)
    int* zslot_begin = nullptr;       // device: first slot of each super row (nsuper + 1)
  } sym_plan[2];
  // device-resident Rayleigh-Ritz (dav_rr_enable): projected matrices, eigenpairs and their operand images stay in HBM
  bool rr_on = false;
  int64_t rr_ld = 0;
  double *rr_H = nullptr, *rr_S = nullptr, *rr_Y = nullptr, *rr_theta = nullptr, *rr_work = nullptr, *rr_info = nullptr;
  double *rr_Ypk = nullptr, *rr_Y2pk = nullptr, *rr_thpk = nullptr;
  SmallBuf sm[N_SMALL];
  size_t small_doubles = 0;
  ncclComm_t comm = nullptr;
  LocalGroup* lg = nullptr;       // loopback transport (tests); owned by rank 0
  ShmGroup* shm = nullptr;        // shared-memory transport (tests of the multi-process launch flow)
  OpDesc op[2];
  std::vector<double> diag_host[2];
  std::vector<int64_t> basis_order;   // indices of the smallest diagonal entries of A (cache of dav_init_basis)
  // streaming ingest (dav_dense_begin .. dav_dense_end): two pinned row-major staging buffers + device twins
  double* ing_host[2] = {nullptr, nullptr};
  double* ing_dev[2] = {nullptr, nullptr};
  hipEvent_t ing_done[2] = {nullptr, nullptr};
  bool ing_pending[2] = {false, false};
  int64_t ing_cap_rows = 0;
  int ing_flip = 0, ing_which = -1;
  // statistics
  dav_stats st{};
  hipEvent_t ev[N_EVPAIRS][2];
  double ev_bytes[N_EVPAIRS];
  int ev_kind[N_EVPAIRS];
  bool ev_done[N_EVPAIRS];        // end event recorded (a call that fails between begin and end leaves a pair without one)
  int ev_used = 0, ev_open = 0;
  int timing_level = 1;           // 0 = nothing, 1 = block matvec only, 2 = every phase
};
typedef dav_engine E;

// ---- collective watchdog (failure detection: the reference's convention is print + stop,
// src/lapack_wrapper.f90:395-408) ------------------------------------------------------------------------------------
// A rank whose peer died inside RCCL would wait for ever: the collectives are asynchronous stream operations, the host
// only notices at its next synchronisation, which never returns.  Every RCCL collective (or group of them) is therefore
// followed by an event, and one thread per engine checks that events complete: one that has not after
// DAVIDSON_COLLECTIVE_TIMEOUT seconds (default 600; 0 = no watchdog) prints rank / collective / outer iteration and ends
// the process with exit code 124 - the launcher then tears the group down.  No re-exec, nothing is retried.
struct Watchdog {
  static constexpr int NW = 32;
  struct Item { hipEvent_t ev = nullptr; const char* what = ""; uint64_t seq = 0; double t0 = 0.0; long iter = -1; bool active = false; };
  Item it[NW];
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  bool stop = false;
  uint64_t seq = 0;
  double timeout_s = 0.0;
  int device = 0, rank = 0, nranks = 1;
};
static double wall_seconds() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void watchdog_loop(Watchdog* w) {
  (void)hipSetDevice(w->device);
  std::unique_lock<std::mutex> lk(w->mu);
  while (!w->stop) {
    w->cv.wait_for(lk, std::chrono::milliseconds(200));
    const double now = wall_seconds();
    for (Watchdog::Item& x : w->it) {
      if (!x.active) continue;
      const hipError_t q = hipEventQuery(x.ev);
      if (q == hipSuccess) { x.active = false; continue; }
      (void)hipGetLastError();
      if (q == hipErrorNotReady && now - x.t0 > w->timeout_s) {
        std::fprintf(stderr, "davidson engine: rank %d of %d: collective \"%s\" (number %llu, outer iteration %ld) has not completed after %.0f s "
                             "- a peer is gone or stuck; ending this process (DAVIDSON_COLLECTIVE_TIMEOUT sets the bound)\n",
                     w->rank, w->nranks, x.what, (unsigned long long)x.seq, x.iter, now - x.t0);
        std::fflush(stderr);
        _exit(124);
      }
    }
  }
}
static void ingest_release(E* e);
static void shm_release(E* e);

static inline int64_t roundup(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

static int bind(E* e) { HIPCHK(hipSetDevice(e->device)); return 0; }

// Called only where no pair can legitimately be open (API entry points, or timed_begin with ev_open == 0): a pair left
// without its end event by a failed call is dropped here, and the open count starts from zero again - a failure does
// not switch the timing off for the rest of the engine's life.
static int collect_events(E* e) {
  e->ev_open = 0;
  for (int i = 0; i < e->ev_used; ++i) {
    if (!e->ev_done[i]) continue;
    HIPCHK(hipEventSynchronize(e->ev[i][1]));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e->ev[i][0], e->ev[i][1]));
    if (e->ev_kind[i] == 0) {
      e->st.apply_ms += ms;
      e->st.apply_bytes += e->ev_bytes[i];
      e->st.last_apply_ms = ms;
      e->st.last_apply_bytes = e->ev_bytes[i];
    } else if (e->ev_kind[i] == 4) {           // the block-matvec kernel alone; ev_bytes carries its flops
      e->st.apply_kernel_ms += ms;
      e->st.apply_flops += e->ev_bytes[i];
      e->st.apply_launches += 1;
    } else if (e->ev_kind[i] == 1) {
      e->st.gram_ms += ms;
    } else if (e->ev_kind[i] == 2) {
      e->st.panel_ms += ms;
    } else {
      e->st.comm_ms += ms;
    }
  }
  e->ev_used = 0;
  return 0;
}
// begin/end record an event pair on the stream; collect_events() turns pairs into milliseconds
static int timed_begin(E* e, int kind, double bytes, int* slot) {
  // an event pair costs ~5 us of host time: by default only the block apply is timed - kind 0 = end to end
  // (pack + all-gather + kernel + reduction), kind 4 = the block-matvec kernel alone (the roofline kernel);
  // dav_set_timing(h, 2) adds the Gram / panel / collective phases
  if (e->timing_level < 1 || (kind != 0 && kind != 4 && e->timing_level < 2)) { *slot = -1; return 0; }
  // pairs nest (kernel inside apply): collect only while no pair is open, and leave room for the inner ones
  if (e->ev_open == 0 && e->ev_used > N_EVPAIRS - 12) CHK(collect_events(e));   // room for the inner pairs of one apply (<= 8 chunks + collectives)
  if (e->ev_used == N_EVPAIRS) { *slot = -1; return 0; }
  ++e->ev_open;
  *slot = e->ev_used++;
  e->ev_done[*slot] = false;
  e->ev_kind[*slot] = kind;
  e->ev_bytes[*slot] = bytes;
  HIPCHK(hipEventRecord(e->ev[*slot][0], e->stream));
  return 0;
}
static int timed_end(E* e, int slot) {
  if (slot < 0) return 0;
  if (e->ev_open > 0) --e->ev_open;
  HIPCHK(hipEventRecord(e->ev[slot][1], e->stream));
  e->ev_done[slot] = true;
  return 0;
}

// upload a p x q host matrix (ld) into small buffer i, zero padded to (pad4(p)) x (pad64(q)); returns ldm
static int small_upload(E* e, int i, const double* src, int64_t ld, int p, int q, int64_t* ldm_out) {
  SmallBuf& b = e->sm[i];
  int64_t ldm = roundup(std::max(p, 1), 4), qp = roundup(std::max(q, 1), 64);
  if ((size_t)(ldm * qp) > e->small_doubles) return fail("small matrix exceeds engine capacity");
  if (b.pending) {
    HIPCHK(hipEventSynchronize(b.done));
    b.pending = false;
  }
  std::memset(b.host, 0, sizeof(double) * ldm * qp);
  for (int j = 0; j < q; ++j) std::memcpy(b.host + j * ldm, src + j * ld, sizeof(double) * p);
  HIPCHK(hipMemcpyAsync(b.dev, b.host, sizeof(double) * ldm * qp, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipEventRecord(b.done, e->stream));
  b.pending = true;
  *ldm_out = ldm;
  return 0;
}

// several small matrices in ONE staging buffer and ONE host-to-device copy (each H2D command costs
// ~10 us of launch latency, which is what the small phases are made of)
struct SmallMat {
  const double* src; int64_t ld; int p, q;   // in
  double* dev; int64_t ldm;                  // out
};
static int small_upload_multi(E* e, int i, SmallMat* mats, int n) {
  SmallBuf& b = e->sm[i];
  size_t total = 0;
  for (int k = 0; k < n; ++k) {
    mats[k].ldm = roundup(std::max(mats[k].p, 1), 4);
    total += (size_t)mats[k].ldm * roundup(std::max(mats[k].q, 1), 64);
  }
  if (total > e->small_doubles) return fail("small matrices exceed engine capacity");
  if (b.pending) {
    HIPCHK(hipEventSynchronize(b.done));
    b.pending = false;
  }
  std::memset(b.host, 0, sizeof(double) * total);
  size_t off = 0;
  for (int k = 0; k < n; ++k) {
    SmallMat& mt = mats[k];
    for (int j = 0; j < mt.q; ++j) std::memcpy(b.host + off + j * mt.ldm, mt.src + j * mt.ld, sizeof(double) * mt.p);
    mt.dev = b.dev + off;
    off += (size_t)mt.ldm * roundup(std::max(mt.q, 1), 64);
  }
  HIPCHK(hipMemcpyAsync(b.dev, b.host, sizeof(double) * total, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipEventRecord(b.done, e->stream));
  b.pending = true;
  return 0;
}

static double* panel_ptr(E* e, int panel, int col) {
  return e->panel[panel] + (int64_t)col * e->ldp;
}
static int check_panel(E* e, int panel, int c0, int k) {
  if (panel < 0 || panel > 5 || !e->panel[panel]) return fail("invalid or unallocated panel id");
  if (c0 < 0 || k < 0 || c0 + k > e->cols_alloc) return fail("panel column range out of bounds");
  return 0;
}

// ------------------------------------------------------------------------------------------------
extern "C" const char* dav_last_error(void) { return g_err.c_str(); }
extern "C" int dav_version(void) { return 100; }

static int create_impl(E* e, int device, int64_t n, int max_cols, int gev, int rank, int nranks);

extern "C" int dav_create(dav_handle_t* h, int device, int64_t n, int max_cols, int gev, int rank, int nranks) {
  if (!h || n <= 0 || max_cols <= 0 || nranks <= 0 || rank < 0 || rank >= nranks) return fail("dav_create: bad arguments");
  *h = nullptr;
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) return fail("dav_create: no HIP device visible - the HIP path has no CPU fallback");
  if (device < 0 || device >= ndev) return fail("dav_create: device index out of range");
  E* e = new E();
  int rc = create_impl(e, device, n, max_cols, gev, rank, nranks);
  if (rc != 0) {                     // e.g. out of device memory: release what was allocated, keep the message
    std::string msg = g_err;
    dav_destroy(e);
    g_err = msg;
    return rc;
  }
  *h = e;
  return 0;
}

static int create_impl(E* e, int device, int64_t n, int max_cols, int gev, int rank, int nranks) {
  e->device = device;
  e->n = n;
  e->rank = rank;
  e->nranks = nranks;
  e->gev = gev ? 1 : 0;
  e->max_cols = max_cols;
  e->cols_alloc = (int)roundup(max_cols, 16) + 16;
  e->nslab = roundup((n + nranks - 1) / nranks, 16);
  e->row0 = (int64_t)rank * e->nslab;
  e->nloc = std::max<int64_t>(0, std::min<int64_t>(e->nslab, n - e->row0));
  e->nloc_pad = roundup(e->nslab, MV_ROWS);
  e->ncols_pad = roundup((int64_t)nranks * e->nslab, SYM_TB);   // whole 256-row blocks: the symmetric sweeps index Xt by tile
  e->ldp = e->nloc_pad;
  e->st.n = n;
  e->st.nloc = e->nloc;
  e->st.rank = rank;
  e->st.nranks = nranks;
  CHK(bind(e));
  HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  size_t pbytes = sizeof(double) * (size_t)e->ldp * e->cols_alloc;
  for (int p = 0; p < 6; ++p) {
    if (p == DAV_PANEL_BV && !e->gev) continue;
    HIPCHK(hipMalloc(&e->panel[p], pbytes));
    HIPCHK(hipMemsetAsync(e->panel[p], 0, pbytes, e->stream));
  }
  e->xt_group_stride = std::max(e->ncols_pad, e->nloc_pad) * 16;   // sym-tiled sweeps index whole 256-row blocks
  HIPCHK(hipMalloc(&e->xt, sizeof(double) * e->xt_group_stride * 4));
  HIPCHK(hipMemsetAsync(e->xt, 0, sizeof(double) * e->xt_group_stride * 4, e->stream));
  int nsplit, jc;
  matvec_plan(e->nloc_pad, e->ncols_pad, 4, &nsplit, &jc);
  size_t s1 = matvec_slab_doubles(e->nloc_pad, 4, nsplit);
  size_t s2 = gram_scratch_doubles(e->cols_alloc, e->cols_alloc, e->nloc_pad);
  e->scratch_doubles = std::max(s1, s2);
  HIPCHK(hipMalloc(&e->scratch, sizeof(double) * e->scratch_doubles));
  e->gram_doubles = 2 * (size_t)e->cols_alloc * e->cols_alloc;      // H and S blocks of one projection side by side
  HIPCHK(hipMalloc(&e->gram_dev, sizeof(double) * e->gram_doubles));
  HIPCHK(hipHostMalloc(&e->gram_host, sizeof(double) * e->gram_doubles, hipHostMallocMapped));
  HIPCHK(hipHostGetDevicePointer((void**)&e->gram_host_dev, e->gram_host, 0));
  HIPCHK(hipMalloc(&e->gather_dev, sizeof(double) * (size_t)e->ncols_pad));
  HIPCHK(hipMalloc(&e->idx_dev, sizeof(int64_t) * e->cols_alloc));
  HIPCHK(hipMalloc(&e->norm_partial, sizeof(double) * (size_t)(e->nloc_pad / PG_ROWS) * e->cols_alloc));
  e->small_doubles = 3 * (size_t)roundup(e->cols_alloc, 4) * roundup(e->cols_alloc, 64);
  for (int i = 0; i < N_SMALL; ++i) {
    HIPCHK(hipMalloc(&e->sm[i].dev, sizeof(double) * e->small_doubles));
    HIPCHK(hipHostMalloc(&e->sm[i].host, sizeof(double) * e->small_doubles, hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&e->sm[i].done, hipEventDisableTiming));
  }
  for (int i = 0; i < N_EVPAIRS; ++i) {
    HIPCHK(hipEventCreate(&e->ev[i][0]));
    HIPCHK(hipEventCreate(&e->ev[i][1]));
  }
  for (int w = 0; w < 2; ++w) {
    HIPCHK(hipMalloc(&e->op[w].diag, sizeof(double) * e->nloc_pad));
    HIPCHK(hipMemsetAsync(e->op[w].diag, 0, sizeof(double) * e->nloc_pad, e->stream));
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

extern "C" int dav_destroy(dav_handle_t e) {
  if (!e) return 0;
  hipSetDevice(e->device);
  if (e->stream) hipStreamSynchronize(e->stream);
  if (e->comm && g_rccl.lib) g_rccl.CommDestroy(e->comm);
  for (int p = 0; p < 6; ++p)
    if (e->panel[p]) hipFree(e->panel[p]);
  hipFree(e->xt);
  hipFree(e->scratch);
  hipFree(e->gram_dev);
  if (e->gram_host) hipHostFree(e->gram_host);
  hipFree(e->gather_dev);
  hipFree(e->idx_dev);
  hipFree(e->norm_partial);
  hipFree(e->gjd_ws);
  ingest_release(e);
  shm_release(e);
  hipFree(e->sym_items);
  hipFree(e->sym_row_begin);
  hipFree(e->sym_slab);
  hipFree(e->rr_H); hipFree(e->rr_S); hipFree(e->rr_Y); hipFree(e->rr_theta); hipFree(e->rr_work); hipFree(e->rr_info);
  hipFree(e->rr_Ypk); hipFree(e->rr_Y2pk); hipFree(e->rr_thpk);
  for (int i = 0; i < 2; ++i) {
    if (e->ov_packed[i]) hipEventDestroy(e->ov_packed[i]);
    if (e->ov_gathered[i]) hipEventDestroy(e->ov_gathered[i]);
    if (e->ov_reduced[i]) hipEventDestroy(e->ov_reduced[i]);
    if (e->ov_scattered[i]) hipEventDestroy(e->ov_scattered[i]);
    hipFree(e->sym_wpart2[i]);
    hipFree(e->sym_wrecv2[i]);
  }
  if (e->comm_stream) hipStreamDestroy(e->comm_stream);
  if (e->wd) {
    { std::lock_guard<std::mutex> lk(e->wd->mu); e->wd->stop = true; }
    e->wd->cv.notify_all();
    e->wd->th.join();
    for (Watchdog::Item& x : e->wd->it) if (x.ev) hipEventDestroy(x.ev);
    delete e->wd;
    e->wd = nullptr;
  }
  hipFree(e->sym_row_off);
  hipFree(e->sym_wpart);
  hipFree(e->sym_wrecv);
  for (auto& pl : e->sym_plan) { hipFree(pl.items); hipFree(pl.row_begin); hipFree(pl.zslot_begin); }
  for (int i = 0; i < N_SMALL; ++i) {
    hipFree(e->sm[i].dev);
    if (e->sm[i].host) hipHostFree(e->sm[i].host);
    if (e->sm[i].done) hipEventDestroy(e->sm[i].done);
  }
  for (int i = 0; i < N_EVPAIRS; ++i) {
    if (e->ev[i][0]) hipEventDestroy(e->ev[i][0]);
    if (e->ev[i][1]) hipEventDestroy(e->ev[i][1]);
  }
  for (int w = 0; w < 2; ++w) {
    hipFree(e->op[w].a);
    hipFree(e->op[w].a32);
    hipFree(e->op[w].e_table);
    hipFree(e->op[w].diag);
  }
  if (e->stream) hipStreamDestroy(e->stream);
  delete e;
  return 0;
}

extern "C" int dav_comm_unique_id(void* id128) {
  CHK(rccl_load());
  ncclUniqueId id;
  NCCLCHK(g_rccl.GetUniqueId(&id));
  std::memcpy(id128, &id, sizeof(id));
  return 0;
}

extern "C" int dav_comm_init(dav_handle_t e, const void* id128) {
  // A single rank needs no communicator.  DAVIDSON_FORCE_RCCL=1 builds a 1-rank communicator anyway so
  // that every collective of the sharded path (all-gather of the packed block, all-reduce of the Gram
  // blocks and norms) runs through RCCL on a single-GPU box - used by the GPU tests.
  if (e->nranks == 1) {
    const char* force = getenv("DAVIDSON_FORCE_RCCL");
    if (!force || force[0] != '1') return 0;
  }
  CHK(rccl_load());
  CHK(bind(e));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  NCCLCHK(g_rccl.CommInitRank(&e->comm, e->nranks, id, e->rank));
  // the watchdog of this communicator's collectives (DAVIDSON_COLLECTIVE_TIMEOUT seconds; default 600, 0 = none)
  double timeout = 600.0;
  if (const char* ev = getenv("DAVIDSON_COLLECTIVE_TIMEOUT")) timeout = atof(ev);
  if (timeout > 0.0 && !e->wd) {
    Watchdog* w = new Watchdog;
    w->timeout_s = timeout; w->device = e->device; w->rank = e->rank; w->nranks = e->nranks;
    for (Watchdog::Item& x : w->it)
      if (hipEventCreateWithFlags(&x.ev, hipEventDisableTiming) != hipSuccess) {
        for (Watchdog::Item& y : w->it) if (y.ev) (void)hipEventDestroy(y.ev);
        delete w;
        return fail("dav_comm_init: could not create the watchdog's events");
      }
    w->th = std::thread(watchdog_loop, w);
    e->wd = w;
  }
  return 0;
}

extern "C" int dav_synchronize(dav_handle_t e) {
  CHK(bind(e));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

extern "C" int dav_get_stats(dav_handle_t e, dav_stats* out) {
  CHK(bind(e));
  CHK(collect_events(e));
  e->st.m = e->m;
  *out = e->st;
  return 0;
}

extern "C" int dav_set_timing(dav_handle_t e, int level) {
  if (level < 0 || level > 2) return fail("dav_set_timing: level must be 0, 1 or 2");
  CHK(bind(e));
  CHK(collect_events(e));
  e->timing_level = level;
  return 0;
}

extern "C" int dav_reset_stats(dav_handle_t e) {
  CHK(bind(e));
  CHK(collect_events(e));
  dav_stats z{};
  z.n = e->n; z.nloc = e->nloc; z.rank = e->rank; z.nranks = e->nranks;
  e->st = z;
  return 0;
}

extern "C" int dav_local_rows(dav_handle_t e, int64_t* row0, int64_t* nloc) {
  *row0 = e->row0;
  *nloc = e->nloc;
  return 0;
}

// ---- operators ---------------------------------------------------------------------------------
static bool has_comm(E* e) { return e->comm != nullptr || e->lg != nullptr || e->shm != nullptr; }
static int need_comm(E* e) {
  if (e->nranks > 1 && !has_comm(e)) return fail("multi-rank engine used before dav_comm_init");
  return 0;
}

// ---- test transports (build flag DAV_TEST_TRANSPORTS: off in the product lib/libdavidson_hip.so, on in lib/test/libdavidson_hip.so,
// the build pytest loads because the GPU tests run on a one-GPU box - csrc/Makefile) -------------------------------------
#ifndef DAV_TEST_TRANSPORTS
#define DAV_TEST_TRANSPORTS 0
#endif
#if DAV_TEST_TRANSPORTS
// ---- loopback transport: several ranks of one problem as threads of ONE process on ONE GPU ----------
// Same collective semantics as the RCCL path (in-place all-gather of equal slabs, sum all-reduce with
// a rank-ordered, hence identical, result on every rank).  It exists so that the row-slab logic of a
// multi-rank engine (offsets, padding, gathered indices) can be verified on a single-GPU box; the
// multi-GPU data path is RCCL.
struct LocalGroup {
  int n = 0;
  pthread_barrier_t bar;
  const double* send[16] = {nullptr};
};

// ---- shared-memory transport: several ranks of one problem as PROCESSES that share one GPU ---------------
// Same collective semantics again, through a POSIX shared-memory segment (staging via the host).  It lets
// the complete multi-process launch flow (torch.distributed.run, id broadcast, one engine per process,
// barriers) run on a single-GPU box; the multi-GPU data path is RCCL.
struct ShmHeader {
  pthread_barrier_t bar;
  int nranks;
  size_t slot_doubles;
};
struct ShmGroup {
  ShmHeader* hdr = nullptr;
  double* slots = nullptr;      // nranks x slot_doubles
  size_t bytes = 0;
  std::string name;
  bool owner = false;
};

static bool has_test_transport(const E* e) { return e->lg != nullptr || e->shm != nullptr; }
static size_t test_transport_max_message(const E* e) { return e->shm ? e->shm->hdr->slot_doubles : (size_t)-1; }

static int test_allgather(E* e, const double* send, double* recv, size_t count) {
  if (e->lg) {
    LocalGroup* g = e->lg;
    HIPCHK(hipStreamSynchronize(e->stream));
    g->send[e->rank] = send;
    pthread_barrier_wait(&g->bar);
    for (int p = 0; p < g->n; ++p)
      if (recv + (size_t)p * count != g->send[p])
        HIPCHK(hipMemcpyAsync(recv + (size_t)p * count, g->send[p], sizeof(double) * count, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    pthread_barrier_wait(&g->bar);
    return 0;
  }
  if (e->shm) {
    ShmGroup* g = e->shm;
    if (count > g->hdr->slot_doubles) return fail("shared-memory transport: message larger than a slot");
    HIPCHK(hipMemcpyAsync(g->slots + (size_t)e->rank * g->hdr->slot_doubles, send, sizeof(double) * count, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    pthread_barrier_wait(&g->hdr->bar);
    for (int p = 0; p < e->nranks; ++p)
      if (p != e->rank || recv + (size_t)p * count != send)
        HIPCHK(hipMemcpyAsync(recv + (size_t)p * count, g->slots + (size_t)p * g->hdr->slot_doubles, sizeof(double) * count,
                              hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    pthread_barrier_wait(&g->hdr->bar);
    return 0;
  }
  return fail("no test transport");
}

static int test_allreduce(E* e, double* buf, size_t count) {
  if (e->lg) {
    LocalGroup* g = e->lg;
    HIPCHK(hipStreamSynchronize(e->stream));
    g->send[e->rank] = buf;
    pthread_barrier_wait(&g->bar);
    std::vector<double> sum(count, 0.0), tmp(count);
    for (int p = 0; p < g->n; ++p) {
      HIPCHK(hipMemcpy(tmp.data(), g->send[p], sizeof(double) * count, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < count; ++i) sum[i] += tmp[i];
    }
    pthread_barrier_wait(&g->bar);            // everyone has read every buffer
    HIPCHK(hipMemcpy(buf, sum.data(), sizeof(double) * count, hipMemcpyHostToDevice));
    return 0;
  }
  if (e->shm) {
    ShmGroup* g = e->shm;
    if (count > g->hdr->slot_doubles) return fail("shared-memory transport: message larger than a slot");
    HIPCHK(hipMemcpyAsync(g->slots + (size_t)e->rank * g->hdr->slot_doubles, buf, sizeof(double) * count, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    pthread_barrier_wait(&g->hdr->bar);
    std::vector<double> sum(count, 0.0);
    for (int p = 0; p < e->nranks; ++p) {          // rank order: the same bits on every rank
      const double* src = g->slots + (size_t)p * g->hdr->slot_doubles;
      for (size_t i = 0; i < count; ++i) sum[i] += src[i];
    }
    pthread_barrier_wait(&g->hdr->bar);
    HIPCHK(hipMemcpy(buf, sum.data(), sizeof(double) * count, hipMemcpyHostToDevice));
    return 0;
  }
  return fail("no test transport");
}

static int test_reduce_scatter(E* e, const double* send, double* recv, size_t count) {
  if (e->lg) {
    LocalGroup* g = e->lg;
    HIPCHK(hipStreamSynchronize(e->stream));
    g->send[e->rank] = send;
    pthread_barrier_wait(&g->bar);
    std::vector<double> sum(count, 0.0), tmp(count);
    for (int p = 0; p < g->n; ++p) {                  // rank order: reproducible
      HIPCHK(hipMemcpy(tmp.data(), g->send[p] + (size_t)e->rank * count, sizeof(double) * count, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < count; ++i) sum[i] += tmp[i];
    }
    pthread_barrier_wait(&g->bar);
    HIPCHK(hipMemcpy(recv, sum.data(), sizeof(double) * count, hipMemcpyHostToDevice));
    return 0;
  }
  if (e->shm) {
    ShmGroup* g = e->shm;
    if (count * (size_t)e->nranks > g->hdr->slot_doubles)
      return fail("shared-memory transport: reduce-scatter message larger than a slot (test transport: small orders only)");
    HIPCHK(hipMemcpyAsync(g->slots + (size_t)e->rank * g->hdr->slot_doubles, send, sizeof(double) * count * e->nranks,
                          hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    pthread_barrier_wait(&g->hdr->bar);
    std::vector<double> sum(count, 0.0);
    for (int p = 0; p < e->nranks; ++p) {
      const double* src = g->slots + (size_t)p * g->hdr->slot_doubles + (size_t)e->rank * count;
      for (size_t i = 0; i < count; ++i) sum[i] += src[i];
    }
    pthread_barrier_wait(&g->hdr->bar);
    HIPCHK(hipMemcpy(recv, sum.data(), sizeof(double) * count, hipMemcpyHostToDevice));
    return 0;
  }
  return fail("no test transport");
}

static void shm_release(E* e) {
  ShmGroup* g = e->shm;
  if (!g) return;
  if (g->hdr) munmap(g->hdr, g->bytes);
  if (g->owner) shm_unlink(g->name.c_str());
  delete g;
  e->shm = nullptr;
}

extern "C" int dav_comm_init_shm(dav_handle_t e, const char* name) {
  if (!name || name[0] != '/') return fail("dav_comm_init_shm: name must start with '/'");
  if (has_comm(e)) return fail("dav_comm_init_shm: the engine already has a transport");
  if (e->nranks == 1) return 0;
  // one slot holds the largest message: an all-gathered slab block (nslab x 16) or a small result matrix
  size_t slot = std::max<size_t>((size_t)e->nslab * 16, std::max(e->gram_doubles, (size_t)e->ncols_pad));
  if ((size_t)e->ncols_pad * 32 * sizeof(double) * e->nranks <= ((size_t)1 << 30))     // symmetric storage: the partial products
    slot = std::max(slot, (size_t)e->ncols_pad * 32);
  size_t bytes = sizeof(ShmHeader) + 64 + sizeof(double) * slot * (size_t)e->nranks;
  ShmGroup* g = new ShmGroup();
  g->name = name;
  g->bytes = bytes;
  int fd = -1;
  if (e->rank == 0) {
    shm_unlink(name);
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { delete g; return fail(std::string("dav_comm_init_shm: cannot create ") + name); }
    g->owner = true;
  } else {
    for (int tries = 0; tries < 3000 && fd < 0; ++tries) {       // rank 0 creates it: wait up to 30 s
      fd = shm_open(name, O_RDWR, 0600);
      if (fd < 0) usleep(10000);
    }
    if (fd < 0) { delete g; return fail(std::string("dav_comm_init_shm: cannot open ") + name); }
    struct stat sb;
    for (int tries = 0; tries < 3000; ++tries) {                   // ... and sizes it
      if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= bytes) break;
      usleep(10000);
    }
  }
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { delete g; return fail("dav_comm_init_shm: mmap failed"); }
  g->hdr = (ShmHeader*)p;
  g->slots = (double*)((char*)p + ((sizeof(ShmHeader) + 63) / 64) * 64);
  if (e->rank == 0) {
    pthread_barrierattr_t attr;
    pthread_barrierattr_init(&attr);
    pthread_barrierattr_setpshared(&attr, PTHREAD_PROCESS_SHARED);
    pthread_barrier_init(&g->hdr->bar, &attr, (unsigned)e->nranks);
    pthread_barrierattr_destroy(&attr);
    g->hdr->slot_doubles = slot;
    __atomic_store_n(&g->hdr->nranks, e->nranks, __ATOMIC_RELEASE);   // published last
  } else {
    for (int tries = 0; tries < 3000 && __atomic_load_n(&g->hdr->nranks, __ATOMIC_ACQUIRE) != e->nranks; ++tries) usleep(10000);
    if (__atomic_load_n(&g->hdr->nranks, __ATOMIC_ACQUIRE) != e->nranks) { munmap(p, bytes); delete g; return fail("dav_comm_init_shm: rank 0 did not initialise the segment"); }
  }
  e->shm = g;
  pthread_barrier_wait(&g->hdr->bar);
  return 0;
}

extern "C" int dav_local_group_join(dav_handle_t* handles, int n) {
  if (!handles || n < 1 || n > 16) return fail("dav_local_group_join: 1..16 engines");
  for (int r = 0; r < n; ++r)
    if (!handles[r] || handles[r]->nranks != n || handles[r]->rank != r || handles[r]->lg || handles[r]->comm)
      return fail("dav_local_group_join: engine r must be created with rank r of n and have no transport yet");
  LocalGroup* g = new LocalGroup();
  g->n = n;
  pthread_barrier_init(&g->bar, nullptr, (unsigned)n);
  for (int r = 0; r < n; ++r) handles[r]->lg = g;
  return 0;
}

#else
static bool has_test_transport(const E*) { return false; }
static size_t test_transport_max_message(const E*) { return (size_t)-1; }
static int test_allgather(E*, const double*, double*, size_t) { return fail("built without test transports"); }
static int test_allreduce(E*, double*, size_t) { return fail("built without test transports"); }
static int test_reduce_scatter(E*, const double*, double*, size_t) { return fail("built without test transports"); }
static void shm_release(E*) {}
extern "C" int dav_comm_init_shm(dav_handle_t, const char*) { return fail("dav_comm_init_shm: built without DAV_TEST_TRANSPORTS"); }
extern "C" int dav_local_group_join(dav_handle_t*, int) { return fail("dav_local_group_join: built without DAV_TEST_TRANSPORTS"); }
#endif

// recv[p*count .. (p+1)*count) = send of rank p, for every rank (send may alias recv + rank*count)
#if DAV_TEST_TRANSPORTS
// test hook (DAV_TEST_STALL_MS): a finite single-thread kernel that holds the stream for that long in front of a
// collective's event, so that the watchdog can be seen to fire on a one-GPU box
__global__ void watchdog_stall_kernel(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
#endif
// an event behind the collective(s) just enqueued on `stream`, handed to the watchdog
static int watch_mark(E* e, const char* what, hipStream_t stream) {
  Watchdog* w = e->wd;
  if (!w || e->group_depth > 0) return 0;
#if DAV_TEST_TRANSPORTS
  if (const char* ev = getenv("DAV_TEST_STALL_MS"))
    hipLaunchKernelGGL(watchdog_stall_kernel, dim3(1), dim3(1), 0, stream, (unsigned long long)atoll(ev) * 100000ull);   // 100 MHz counter
#endif
  std::lock_guard<std::mutex> lk(w->mu);
  Watchdog::Item& x = w->it[w->seq % Watchdog::NW];
  if (x.active) return 0;                       // the ring is full of unfinished collectives: the oldest of them is being watched
  HIPCHK(hipEventRecord(x.ev, stream));
  x.what = what; x.seq = w->seq++; x.t0 = wall_seconds(); x.iter = e->iter_hint; x.active = true;
  return 0;
}
static int coll_group_begin(E* e) {
  if (e->comm) { NCCLCHK(g_rccl.GroupStart()); ++e->group_depth; }
  return 0;
}
static int coll_group_end(E* e, const char* what, hipStream_t stream) {
  if (e->comm) { NCCLCHK(g_rccl.GroupEnd()); --e->group_depth; CHK(watch_mark(e, what, stream)); }
  return 0;
}

static int coll_allgather(E* e, const double* send, double* recv, size_t count) {
  if (has_test_transport(e)) return test_allgather(e, send, recv, count);
  NCCLCHK(g_rccl.AllGather(send, recv, count, ncclDouble, e->comm, e->stream));
  return watch_mark(e, "all-gather", e->stream);
}

// buf <- sum over ranks of buf (same bits on every rank)
static int coll_allreduce(E* e, double* buf, size_t count) {
  if (has_test_transport(e)) return test_allreduce(e, buf, count);
  NCCLCHK(g_rccl.AllReduce(buf, buf, count, ncclDouble, ncclSum, e->comm, e->stream));
  return watch_mark(e, "all-reduce", e->stream);
}

// recv[0 .. count) = sum over ranks p of send_p[rank*count .. (rank+1)*count)  (send holds nranks chunks)
static int coll_reduce_scatter(E* e, const double* send, double* recv, size_t count) {
  if (has_test_transport(e)) return test_reduce_scatter(e, send, recv, count);
  NCCLCHK(g_rccl.ReduceScatter(send, recv, count, ncclDouble, ncclSum, e->comm, e->stream));
  return watch_mark(e, "reduce-scatter", e->stream);
}

static int refresh_diag_host(E* e, int which) {
  // global diagonal on the host (stable top-k selection, dav_get_diagonal)
  if (which == DAV_OP_A) e->basis_order.clear();
  std::vector<double>& d = e->diag_host[which];
  d.assign((size_t)e->n, 0.0);
  if (!has_comm(e)) {
    CHK(need_comm(e));
    HIPCHK(hipMemcpyAsync(d.data(), e->op[which].diag, sizeof(double) * e->n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  } else {
    CHK(coll_allgather(e, e->op[which].diag, e->gather_dev, (size_t)e->nslab));
    HIPCHK(hipMemcpyAsync(d.data(), e->gather_dev, sizeof(double) * e->n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  return 0;
}

// Block rows per workgroup of the symmetric sweep for a block of kk <= 16 columns: 4 (k <= 8), 2, or 1 (the
// one-block-row kernel of k_matvec_sym.hip: below 200 block rows, where super rows leave too few work items and too much
// of the matrix in the masked diagonal super blocks).  DAV_SYM_R = 1 | 2 | 4 forces a schedule (4 only if k <= 8).
static int sym_schedule(const E* e, int kk) {
  const char* ev = getenv("DAV_SYM_R");                 // read per call: A/B runs flip it inside one process
  const int forced = ev ? atoi(ev) : 0;
  const int nb = (int)(e->ncols_pad / SYM_TB);          // block rows of the whole matrix
  // crossover measured end to end on one box (k = 8 / 32; ms for R = 1 | 2 | 4): N=40000 (157 block rows) 1.33 | 1.38 | 1.39
  // and 2.30 | 2.45 | 2.42; N=60000 (235) 2.78 | 2.78 | 2.67 and 5.04 | 5.00 | 5.01; N=100000 7.28 | 7.39 | 6.84 and
  // 13.78 | 13.61 | 13.57; N=140000 14.98 | 14.85 | 13.99 and 26.81 | 26.31 | 26.30
  int R = nb >= 200 ? (kk <= 8 ? 4 : 2) : 1;
  if (forced == 1 || forced == 2 || forced == 4) R = forced;
  if (R == 4 && kk > 8) R = 2;
  if (R > 1 && !matvec_sym_can_pair()) R = 1;          // DAV_SYM_V8=0: the one-wave-per-SIMD kernel, A/B runs only
  return R;
}

// Owners of the groups of 4 block rows (what every schedule's super rows nest in): longest group first, each to the rank
// that holds the fewest tiles so far (ties: lowest rank) - every rank computes the same table.  Cyclic or boustrophedon
// dealing leaves the ranks 4-8 % apart at N=200000 on 8 ranks (the last, incomplete round hands out the longest block
// rows); this stays within 0.5 %, and the sweep time of the slowest rank is what every rank waits for.
static std::vector<int> sym_group_owners(int nb, int nranks) {
  const int ng = (nb + 3) / 4;
  std::vector<int> owner(ng, 0);
  std::vector<int64_t> load(nranks, 0);
  for (int q = ng - 1; q >= 0; --q) {
    int64_t tiles = 0;
    for (int I = 4 * q; I < std::min(nb, 4 * q + 4); ++I) tiles += I + 1;
    int best = 0;
    for (int r = 1; r < nranks; ++r)
      if (load[r] < load[best]) best = r;
    owner[q] = best;
    load[best] += tiles;
  }
  return owner;
}

static int sym_setup(E* e) {
  // work lists of the symmetric sweep over the block rows THIS rank stores
  if (e->sym_items) return 0;
  const int nb = (int)(e->ncols_pad / SYM_TB);
  e->sym_row_off_h.assign(nb, -1);
  int64_t ntiles = 0;
  const std::vector<int> gowner = sym_group_owners(nb, e->nranks);
  for (int I = 0; I < nb; ++I)
    if (gowner[I / 4] == e->rank) { e->sym_row_off_h[I] = ntiles; ntiles += I + 1; }
  e->sym_ntiles_local = ntiles;
  HIPCHK(hipMalloc(&e->sym_row_off, sizeof(int64_t) * nb));
  HIPCHK(hipMemcpy(e->sym_row_off, e->sym_row_off_h.data(), sizeof(int64_t) * nb, hipMemcpyHostToDevice));
  auto owned = [&](int I) { return e->sym_row_off_h[I] >= 0; };
  // One-block-row kernel: runs of <= C consecutive tiles of one block row.
  // Run length: ~12 rounds of the 256 resident workgroups, between 4 tiles (a workgroup costs ~7 us to start
  // and drain) and 32 (the tail of the sweep is at most one run long).  Slab slots stay in block-row order
  // (the reduction kernel walks them per block row); the dispatch order is longest run first, so the
  // short remainder runs of every block row fill the tail (same box, N=60000: 2.95-3.04 ms against 3.16-3.37 ms
  // in block-row order for run lengths 6..24; N=200000: flat within 1 % for 16..64).
  int64_t C = std::min<int64_t>(32, std::max<int64_t>(4, (ntiles + 3071) / 3072));
  if (const char* ev = getenv("DAV_SYM_RUN")) C = std::max(1, atoi(ev));
  struct Item { int I, J0, J1, slot; };
  std::vector<Item> list;
  std::vector<int> row_begin(nb + 1, 0);
  for (int I = 0; I < nb; ++I) {
    row_begin[I] = (int)list.size();
    if (!owned(I)) continue;
    for (int J0 = 0; J0 <= I; J0 += (int)C)
      list.push_back({I, J0, (int)std::min<int64_t>(I + 1, J0 + C), (int)list.size()});
  }
  row_begin[nb] = (int)list.size();
  std::stable_sort(list.begin(), list.end(), [](const Item& a, const Item& b) { return a.J1 - a.J0 > b.J1 - b.J0; });
  std::vector<int> items;
  items.reserve(list.size() * 4 + 4);
  for (const Item& it : list) { items.push_back(it.I); items.push_back(it.J0); items.push_back(it.J1); items.push_back(it.slot); }
  items.resize(std::max<size_t>(items.size(), 4), 0);
  e->sym_nb = nb;
  e->sym_nitems = row_begin[nb];
  HIPCHK(hipMalloc(&e->sym_items, sizeof(int) * items.size()));
  HIPCHK(hipMalloc(&e->sym_row_begin, sizeof(int) * row_begin.size()));
  HIPCHK(hipMemcpy(e->sym_items, items.data(), sizeof(int) * items.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->sym_row_begin, row_begin.data(), sizeof(int) * row_begin.size(), hipMemcpyHostToDevice));
  // Super-row schedules: items = (super row of R block rows) x (run of C tile columns).  Per tile the schedule
  // writes 1/R of a transposed and 1/C of a direct partial; C is bounded by the tail of the sweep (an item is
  // R*C tiles long) and below by the number of items that keeps 256 workgroups busy.
  for (int p = 0; p < 2; ++p) {
    E::SymPlan& pl = e->sym_plan[p];
    pl.R = p == 0 ? 2 : 4;
    pl.nsuper = (nb + pl.R - 1) / pl.R;
    int64_t Cp = std::min<int64_t>(64 / pl.R, std::max<int64_t>(1, (ntiles + 3071) / (3072 * pl.R)));
    if (const char* ev = getenv("DAV_SYM_RUN9")) Cp = std::max(1, atoi(ev));
    std::vector<Item> plist;
    std::vector<int> prow(pl.nsuper + 1, 0), zbeg(pl.nsuper + 1, 0);
    for (int S = 0; S < pl.nsuper; ++S) {
      prow[S] = (int)plist.size();
      zbeg[S + 1] = zbeg[S];
      if (!owned(S * pl.R)) continue;                // a super row nests in a group of 4 block rows: one owner
      const int Imax = std::min(S * pl.R + pl.R - 1, nb - 1);
      for (int J0 = 0; J0 <= Imax; J0 += (int)Cp)
        plist.push_back({S, J0, (int)std::min<int64_t>(Imax + 1, J0 + Cp), (int)plist.size()});
      zbeg[S + 1] = zbeg[S] + Imax;                // tile columns J < Imax receive a transposed partial
    }
    prow[pl.nsuper] = (int)plist.size();
    std::stable_sort(plist.begin(), plist.end(), [](const Item& a, const Item& b) { return a.J1 - a.J0 > b.J1 - b.J0; });
    std::vector<int> pitems;
    pitems.reserve(plist.size() * 4 + 4);
    for (const Item& it : plist) { pitems.push_back(it.I); pitems.push_back(it.J0); pitems.push_back(it.J1); pitems.push_back(it.slot); }
    pitems.resize(std::max<size_t>(pitems.size(), 4), 0);
    pl.nitems = prow[pl.nsuper];
    pl.zslots = zbeg[pl.nsuper];
    HIPCHK(hipMalloc(&pl.items, sizeof(int) * pitems.size()));
    HIPCHK(hipMalloc(&pl.row_begin, sizeof(int) * prow.size()));
    HIPCHK(hipMalloc(&pl.zslot_begin, sizeof(int) * zbeg.size()));
    HIPCHK(hipMemcpy(pl.items, pitems.data(), sizeof(int) * pitems.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(pl.row_begin, prow.data(), sizeof(int) * prow.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(pl.zslot_begin, zbeg.data(), sizeof(int) * zbeg.size(), hipMemcpyHostToDevice));
  }
  return 0;
}

// diagonal of a stored symmetric-tiled operator -> o.diag (this rank's rows); with several ranks the diagonal tiles
// live where their block rows do: every rank contributes its pieces, one all-reduce of n doubles at set-up
static int coll_allreduce(E* e, double* buf, size_t count);
static int sym_diag(E* e, OpDesc& o) {
  if (e->nranks == 1) {
    launch_diag_sym(e->stream, o.a, e->sym_row_off, e->n, e->nloc_pad, o.diag);
    return 0;
  }
  if (!(e->comm || e->lg || e->shm)) return fail("multi-rank engine used before dav_comm_init");
  launch_diag_sym(e->stream, o.a, e->sym_row_off, e->n, e->ncols_pad, e->gather_dev);
  CHK(coll_allreduce(e, e->gather_dev, (size_t)e->ncols_pad));
  HIPCHK(hipMemcpyAsync(o.diag, e->gather_dev + e->row0, sizeof(double) * (size_t)e->nslab, hipMemcpyDeviceToDevice, e->stream));
  return 0;
}

// Slabs of the symmetric sweep - per launch [column groups x direct partials][column groups x transposed partials] -
// grown on demand to what the schedule and the number of column groups of a launch need: one transposed partial per
// TILE for the one-block-row kernel (N=200000, 32 columns: 20 GB), per (super row, tile column) for the super-row
// schedules (5-10 GB); N=10^6 matrix-free, 16 columns, R=2: 125 GB.
static int sym_ensure_slabs(E* e, size_t doubles) {
  if (doubles <= e->sym_slab_doubles) return 0;
  HIPCHK(hipStreamSynchronize(e->stream));
  if (e->sym_slab) HIPCHK(hipFree(e->sym_slab));
  e->sym_slab = nullptr;
  e->sym_slab_doubles = 0;
  hipError_t r = hipMalloc(&e->sym_slab, sizeof(double) * doubles);
  if (r != hipSuccess) {
    (void)hipGetLastError();
    e->sym_slab = nullptr;
    return fail("hipMalloc of the symmetric sweep slabs failed: " + std::string(hipGetErrorString(r)));
  }
  e->sym_slab_doubles = doubles;
  return 0;
}

static int alloc_dense(E* e, int which) {
  OpDesc& o = e->op[which];
  o.a32_valid = false;       // new contents: the fp32 copy is rebuilt when the next inner sweep asks for it
  o.a32_refused = false;
  if (o.a && o.storage != e->storage) { hipFree(o.a); o.a = nullptr; }
  o.storage = e->storage;
  if (!o.a) {
    size_t bytes;
    if (o.storage == 1) {
      CHK(sym_setup(e));
      bytes = sizeof(double) * (size_t)std::max<int64_t>(e->sym_ntiles_local, 1) * SYM_TB * SYM_TB;
    } else {
      bytes = sizeof(double) * (size_t)e->nloc_pad * (size_t)e->ncols_pad;
    }
    hipError_t r = hipMalloc(&o.a, bytes);
    if (r != hipSuccess) {
      (void)hipGetLastError();
      return fail("hipMalloc of the dense matrix (" + std::to_string(bytes >> 20) + " MiB) failed: " + hipGetErrorString(r));
    }
  }
  if (o.storage == 1) CHK(sym_setup(e));
  return 0;
}

// Mixed-precision correction path.  bits = 32: the block sweeps INSIDE the GJD correction solve
// (src/davidson.f90:700-734: the solve only has to produce a good correction vector) read an fp32 copy of the stored
// symmetric tiles - half the bytes per inner sweep; entries are widened to fp64 in registers, every product and sum
// stays fp64.  Everything the answer is made of - the A*V sweep of the expansion, projections, residuals, the
// convergence test - keeps reading the fp64 matrix.  bits = 64 (default): the reference's precision throughout.
// Operators that are not stored symmetric tiles (row slabs, generated operators) are not affected.
extern "C" int dav_set_inner_precision(dav_handle_t e, int bits) {
  if (bits != 32 && bits != 64) return fail("dav_set_inner_precision: 32 or 64");
  e->inner_bits = bits;
  return 0;
}

extern "C" int dav_set_storage(dav_handle_t e, int mode) {
  if (mode != 0 && mode != 1) return fail("dav_set_storage: mode must be 0 (full) or 1 (symmetric-tiled)");
  e->storage = mode;
  return 0;
}

static int set_dense_from(E* e, int which, const double* a, int64_t lda, hipMemcpyKind kind) {
  if (which < 0 || which > 1 || !a || lda < e->n) return fail("dav_set_dense: bad arguments");
  CHK(bind(e));
  CHK(alloc_dense(e, which));
  OpDesc& o = e->op[which];
  o.kind = DAV_KIND_DENSE;
  if (o.storage == 1) {
    // lower block triangle, tile by tile (edge tiles zero padded)
    // block column by block column: one long-row 2-D copy of the rows from the diagonal block down into a staging panel
    // (two panels alternate, so the copy of column J + 1 is queued behind the cut of column J), then cut into tiles
    const int nb = e->sym_nb;
    const int64_t ldp_stage = (int64_t)nb * SYM_TB;
    double* stage[2] = {nullptr, nullptr};
    for (int b = 0; b < 2; ++b) {
      hipError_t r = hipMalloc(&stage[b], sizeof(double) * (size_t)ldp_stage * SYM_TB);
      if (r != hipSuccess) {
        (void)hipGetLastError();
        if (stage[0]) hipFree(stage[0]);
        return fail("hipMalloc of the upload staging panel failed: " + std::string(hipGetErrorString(r)));
      }
    }
    int rc = 0;
    for (int J = 0; J < nb && rc == 0; ++J) {
      const int64_t r0 = (int64_t)J * SYM_TB, nr = e->n - r0;
      const int nc = (int)std::min<int64_t>(SYM_TB, e->n - r0);
      if (nr <= 0) {                                    // block rows / columns wholly in the padding: zero tiles
        launch_retile_panel(e->stream, stage[J & 1], ldp_stage, 0, 0, J, nb, e->sym_row_off, o.a);
        continue;
      }
      if (hipMemcpy2DAsync(stage[J & 1], sizeof(double) * ldp_stage, a + r0 + r0 * lda, sizeof(double) * lda, sizeof(double) * nr,
                           (size_t)nc, kind, e->stream) != hipSuccess) {
        (void)hipGetLastError();
        rc = fail("dav_set_dense: copy of a block column failed");
        break;
      }
      launch_retile_panel(e->stream, stage[J & 1], ldp_stage, nr, nc, J, nb, e->sym_row_off, o.a);
    }
    hipStreamSynchronize(e->stream);
    hipFree(stage[0]);
    hipFree(stage[1]);
    if (rc != 0) return rc;
    HIPCHK(hipGetLastError());
    CHK(sym_diag(e, o));
    CHK(refresh_diag_host(e, which));
    return 0;
  }
  HIPCHK(hipMemsetAsync(o.a, 0, sizeof(double) * (size_t)e->nloc_pad * (size_t)e->ncols_pad, e->stream));
  if (e->nloc > 0)
    HIPCHK(hipMemcpy2DAsync(o.a, sizeof(double) * e->nloc_pad, a + e->row0, sizeof(double) * lda,
                            sizeof(double) * e->nloc, (size_t)e->n, kind, e->stream));
  launch_diag_dense(e->stream, o.a, e->nloc_pad, e->row0, e->nloc, o.diag);
  CHK(refresh_diag_host(e, which));
  return 0;
}

extern "C" int dav_set_dense_host(dav_handle_t e, int which, const double* a, int64_t lda) {
  return set_dense_from(e, which, a, lda, hipMemcpyHostToDevice);
}

extern "C" int dav_set_dense_dev(dav_handle_t e, int which, const double* a_dev, int64_t lda) {
  return set_dense_from(e, which, a_dev, lda, hipMemcpyDeviceToDevice);
}

// ---- streaming ingest: rows arrive in the reference's on-disk order (row-major) -----------------------------
static void ingest_release(E* e) {
  for (int b = 0; b < 2; ++b) {
    if (e->ing_done[b]) { hipEventSynchronize(e->ing_done[b]); hipEventDestroy(e->ing_done[b]); e->ing_done[b] = nullptr; }
    if (e->ing_host[b]) { hipHostFree(e->ing_host[b]); e->ing_host[b] = nullptr; }
    if (e->ing_dev[b]) { hipFree(e->ing_dev[b]); e->ing_dev[b] = nullptr; }
    e->ing_pending[b] = false;
  }
  e->ing_which = -1;
}

extern "C" int dav_dense_begin(dav_handle_t e, int which) {
  if (which < 0 || which > 1) return fail("dav_dense_begin: bad operator id");
  if (e->ing_which >= 0) return fail("dav_dense_begin: another streaming upload is open (call dav_dense_end)");
  CHK(bind(e));
  CHK(alloc_dense(e, which));
  OpDesc& o = e->op[which];
  o.kind = DAV_KIND_DENSE;
  size_t bytes = o.storage == 1 ? sizeof(double) * (size_t)e->sym_ntiles_local * SYM_TB * SYM_TB
                                : sizeof(double) * (size_t)e->nloc_pad * (size_t)e->ncols_pad;
  HIPCHK(hipMemsetAsync(o.a, 0, bytes, e->stream));
  // ~128 MiB per staging buffer, whole rows, at least 32 of them
  int64_t cap = std::max<int64_t>(32, ((int64_t)128 << 20) / (8 * e->n) / 32 * 32);
  cap = std::min<int64_t>(cap, roundup(e->n, 32));
  e->ing_cap_rows = cap;
  for (int b = 0; b < 2; ++b) {
    HIPCHK(hipHostMalloc(&e->ing_host[b], sizeof(double) * (size_t)(cap * e->n), hipHostMallocDefault));
    HIPCHK(hipMalloc(&e->ing_dev[b], sizeof(double) * (size_t)(cap * e->n)));
    HIPCHK(hipEventCreateWithFlags(&e->ing_done[b], hipEventDisableTiming));
  }
  e->ing_flip = 0;
  e->ing_which = which;
  return 0;
}

static int ingest_acquire(E* e, double** buf, int64_t* cap_rows) {
  if (e->ing_which < 0) return fail("streaming upload is not open (call dav_dense_begin)");
  int b = e->ing_flip;
  if (e->ing_pending[b]) { HIPCHK(hipEventSynchronize(e->ing_done[b])); e->ing_pending[b] = false; }
  *buf = e->ing_host[b];
  *cap_rows = e->ing_cap_rows;
  return 0;
}

static int ingest_commit(E* e, int64_t row0, int64_t nrows) {
  if (e->ing_which < 0) return fail("streaming upload is not open (call dav_dense_begin)");
  if (row0 < 0 || nrows < 0 || row0 + nrows > e->n || nrows > e->ing_cap_rows) return fail("dav_dense_put_rows: rows out of range");
  if (nrows == 0) return 0;
  CHK(bind(e));
  OpDesc& o = e->op[e->ing_which];
  int b = e->ing_flip;
  HIPCHK(hipMemcpyAsync(e->ing_dev[b], e->ing_host[b], sizeof(double) * (size_t)(nrows * e->n), hipMemcpyHostToDevice, e->stream));
  launch_rows_scatter(e->stream, e->ing_dev[b], e->n, row0, nrows, e->n, o.a, e->nloc_pad, e->row0, e->nloc, o.storage == 1, e->sym_row_off);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(e->ing_done[b], e->stream));
  e->ing_pending[b] = true;
  e->ing_flip ^= 1;
  return 0;
}

static void ingest_wanted(E* e, int64_t* first, int64_t* count) {
  if (e->op[e->ing_which].storage == 1) { *first = 0; *count = e->n; }
  else { *first = e->row0; *count = e->nloc; }
}

extern "C" int dav_dense_put_rows(dav_handle_t e, int which, int64_t row0, int64_t nrows, const double* rows, int64_t ldr) {
  if (e->ing_which != which) return fail("dav_dense_put_rows: no streaming upload open for this operator");
  if (!rows || ldr < e->n || row0 < 0 || nrows < 0 || row0 + nrows > e->n) return fail("dav_dense_put_rows: bad arguments");
  int64_t w0, wn;
  ingest_wanted(e, &w0, &wn);
  int64_t lo = std::max(row0, w0), hi = std::min(row0 + nrows, w0 + wn);     // rows of other ranks are ignored
  for (int64_t r = lo; r < hi;) {
    double* buf; int64_t cap;
    CHK(ingest_acquire(e, &buf, &cap));
    int64_t take = std::min(cap, hi - r);
    // staging copy, by several threads when the block is large (one memcpy stream into pinned memory runs at
    // ~4 GB/s, far below the host-to-device copy that follows)
    const size_t blk_bytes = sizeof(double) * (size_t)take * (size_t)e->n;
    const int T = (int)std::min<size_t>(8, blk_bytes / ((size_t)8 << 20) + 1);
    auto copy_rows = [&](int64_t i0, int64_t i1) {
      for (int64_t i = i0; i < i1; ++i) memcpy(buf + i * e->n, rows + (r - row0 + i) * ldr, sizeof(double) * (size_t)e->n);
    };
    if (T <= 1) {
      copy_rows(0, take);
    } else {
      std::vector<std::thread> pool;
      for (int t = 0; t < T; ++t) pool.emplace_back(copy_rows, take * t / T, take * (t + 1) / T);
      for (auto& th : pool) th.join();
    }
    CHK(ingest_commit(e, r, take));
    r += take;
  }
  return 0;
}

extern "C" int dav_dense_end(dav_handle_t e, int which) {
  if (e->ing_which != which) return fail("dav_dense_end: no streaming upload open for this operator");
  CHK(bind(e));
  OpDesc& o = e->op[which];
  int rc = 0;
  if (o.storage == 1) rc = sym_diag(e, o);
  else launch_diag_dense(e->stream, o.a, e->nloc_pad, e->row0, e->nloc, o.diag);
  if (rc == 0) rc = refresh_diag_host(e, which);
  ingest_release(e);
  return rc;
}

namespace {
struct EngineSink : IngestSink {
  E* e;
  explicit EngineSink(E* e_) : e(e_) {}
  int acquire(double** buf, int64_t* cap_rows) override { return ingest_acquire(e, buf, cap_rows); }
  int commit(int64_t row0, int64_t nrows) override { return ingest_commit(e, row0, nrows); }
  void wanted(int64_t* first, int64_t* count) override { ingest_wanted(e, first, count); }
};
}  // namespace

extern "C" int dav_set_dense_file(dav_handle_t e, int which, const char* path, int format) {
  if (!path) return fail("dav_set_dense_file: null path");
  if (format != DAV_FILE_TEXT && format != DAV_FILE_F64) return fail("dav_set_dense_file: unknown format");
  CHK(dav_dense_begin(e, which));
  EngineSink sink(e);
  std::string err;
  int rc = format == DAV_FILE_TEXT ? ingest_text_file(path, e->n, sink, &err) : ingest_f64_file(path, e->n, sink, &err);
  if (rc != 0) {
    hipStreamSynchronize(e->stream);
    ingest_release(e);
    e->op[which].kind = DAV_KIND_NONE;
    return err.empty() ? rc : fail("dav_set_dense_file: " + err);
  }
  return dav_dense_end(e, which);
}

extern "C" int dav_parse_text_f64(const char* text, size_t len, double* out, size_t max_vals, size_t* nvals) {
  std::vector<double> v;
  std::string err;
  size_t used = ingest_parse_text_parallel(text, len, true, &v, 4, &err);
  if (used == (size_t)-1) return fail("dav_parse_text_f64: " + err);
  if (nvals) *nvals = v.size();
  if (out) memcpy(out, v.data(), sizeof(double) * std::min(v.size(), max_vals));
  return 0;
}

extern "C" int dav_set_dense_generated(dav_handle_t e, int which, uint64_t seed, double sparsity, int use_diag_val,
                                       double diag_val) {
  if (which < 0 || which > 1) return fail("dav_set_dense_generated: bad operator id");
  CHK(bind(e));
  CHK(alloc_dense(e, which));
  OpDesc& o = e->op[which];
  o.kind = DAV_KIND_DENSE;
  if (o.storage == 1) {
    launch_generate_sym_tiles(e->stream, o.a, e->sym_row_off_h.data(), e->sym_nb, e->n, seed, sparsity, use_diag_val, diag_val);
    CHK(sym_diag(e, o));
  } else {
    launch_generate_dense(e->stream, o.a, e->nloc_pad, e->nloc_pad, e->ncols_pad, e->row0, e->nloc, e->n, seed, sparsity,
                          use_diag_val, diag_val);
    launch_diag_dense(e->stream, o.a, e->nloc_pad, e->row0, e->nloc, o.diag);
  }
  CHK(refresh_diag_host(e, which));
  return 0;
}

static OpParams op_params(const OpDesc& o) {
  OpParams p;
  p.kind = o.kind; p.seed = o.seed; p.sparsity = o.sparsity; p.use_diag = o.use_diag; p.diag_val = o.diag_val;
  p.trig = o.trig; p.e_table = o.e_table;
  return p;
}

extern "C" int dav_set_operator_hashed(dav_handle_t e, int which, uint64_t seed, double sparsity, int use_diag_val,
                                       double diag_val) {
  if (which < 0 || which > 1) return fail("dav_set_operator_hashed: bad operator id");
  CHK(bind(e));
  OpDesc& o = e->op[which];
  o.kind = DAV_KIND_HASHED; o.seed = seed; o.sparsity = sparsity; o.use_diag = use_diag_val; o.diag_val = diag_val;
  // storage mode "symmetric" (single rank) also applies to the generated operator: every entry of the lower
  // block triangle is produced once and used for both products
  o.storage = e->storage == 1 ? 1 : 0;
  if (o.storage == 1) CHK(sym_setup(e));
  launch_diag_free(e->stream, op_params(o), e->row0, e->nloc, o.diag);
  CHK(refresh_diag_host(e, which));
  return 0;
}

extern "C" int dav_set_operator_harness(dav_handle_t e, int which, const double* e_table) {
  if (which < 0 || which > 1 || !e_table) return fail("dav_set_operator_harness: bad arguments");
  CHK(bind(e));
  OpDesc& o = e->op[which];
  o.kind = DAV_KIND_HARNESS; o.trig = which == DAV_OP_A ? 0 : 1;
  o.storage = e->storage == 1 ? 1 : 0;      // symmetric mode: each entry generated once
  if (o.storage == 1) CHK(sym_setup(e));
  if (!o.e_table) HIPCHK(hipMalloc(&o.e_table, sizeof(double) * e->n));
  HIPCHK(hipMemcpyAsync(o.e_table, e_table, sizeof(double) * e->n, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  launch_diag_free(e->stream, op_params(o), e->row0, e->nloc, o.diag);
  CHK(refresh_diag_host(e, which));
  return 0;
}

extern "C" int dav_set_operator_identity(dav_handle_t e, int which) {
  if (which < 0 || which > 1) return fail("dav_set_operator_identity: bad operator id");
  CHK(bind(e));
  OpDesc& o = e->op[which];
  o.kind = DAV_KIND_IDENTITY;
  launch_diag_free(e->stream, op_params(o), e->row0, e->nloc, o.diag);
  CHK(refresh_diag_host(e, which));
  return 0;
}

extern "C" int dav_set_operator_host(dav_handle_t e, int which, const double* diag) {
  if (which < 0 || which > 1 || !diag) return fail("dav_set_operator_host: bad arguments");
  CHK(bind(e));
  OpDesc& o = e->op[which];
  o.kind = DAV_KIND_HOST;
  if (e->nloc > 0)
    HIPCHK(hipMemcpyAsync(o.diag, diag + e->row0, sizeof(double) * e->nloc, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->diag_host[which].assign(diag, diag + e->n);
  if (which == DAV_OP_A) e->basis_order.clear();
  return 0;
}

extern "C" int dav_get_diagonal(dav_handle_t e, int which, double* out) {
  if (which < 0 || which > 1 || e->diag_host[which].empty()) return fail("dav_get_diagonal: operator not set");
  std::memcpy(out, e->diag_host[which].data(), sizeof(double) * e->n);
  return 0;
}

// ---- K1 -----------------------------------------------------------------------------------------
// dst[:, 0:k] = Op(which) * src[:, 0:k] for device-resident column blocks with leading dimension ldp
// fp32 copy of a stored symmetric-tiled operator, made when the first inner sweep wants it; false (and fp64 sweeps) when
// the memory for it is not there
static bool inner_f32_tiles(E* e, OpDesc& o) {
  if (e->inner_bits != 32 || o.kind != DAV_KIND_DENSE || o.storage != 1 || o.a32_refused) return false;
  if (o.a32_valid) return true;
  const size_t count = (size_t)std::max<int64_t>(e->sym_ntiles_local, 1) * SYM_TB * SYM_TB;
  if (!o.a32 && hipMalloc(&o.a32, sizeof(float) * count) != hipSuccess) {
    (void)hipGetLastError();
    o.a32 = nullptr;
    o.a32_refused = true;
    return false;
  }
  launch_tiles_to_f32(e->stream, o.a, o.a32, (int64_t)count);
  o.a32_valid = true;
  return true;
}

// The super-row sweep of one launch: stored fp64 tiles, two block rows per workgroup and more than 8 columns run the
// one-wave-per-SIMD kernel (k_matvec_symw.hip: 32 columns per workgroup, or 16 for a block of <= 16); generated operators,
// the fp32 copy and the k <= 8 schedule (R = 4, 4x4x4 MFMA) stay on matvec_sym9_kernel.
// DAV_SYM_WIDE = 0: never (A/B runs), 1: blocks wider than 16 columns only, 2 (default): from 9 columns on.  Read per call.
static void sym9_sweep(E* e, int R, const OpDesc& o, bool use32, const E::SymPlan* pl, const double* xt, int kk, double* slabD, double* slabT,
                       int npair, int64_t dstride, int64_t tstride) {
  const char* ev = getenv("DAV_SYM_WIDE");
  const int wide = ev ? atoi(ev) : 2;
  if (R == 2 && o.kind == DAV_KIND_DENSE && !use32 && wide > 0 && (kk > 16 || wide > 1)) {
    const int nbw = kk > 16 ? 2 : 1;
    launch_matvec_symw(e->stream, nbw, o.a, e->sym_row_off, e->sym_nb, pl->items, pl->nitems, pl->zslot_begin, xt, kk, slabD, slabT,
                       (npair + nbw - 1) / nbw, e->xt_group_stride, dstride, tstride);
    return;
  }
  launch_matvec_sym9(e->stream, R, o.kind != DAV_KIND_DENSE, use32 ? (const void*)o.a32 : (const void*)o.a, use32, e->sym_row_off,
                     o.kind != DAV_KIND_DENSE ? op_params(o) : OpParams{}, e->n, e->sym_nb, pl->items, pl->nitems, pl->zslot_begin, xt, kk,
                     slabD, slabT, npair, e->xt_group_stride, dstride, tstride);
}


// Symmetric sweep of k > 32 columns over several ranks with RCCL, chunks of 32 columns software-pipelined over two streams:
//   comm stream:  gather(0)            gather(1)   scatter(0)   gather(2)   scatter(1) ...
//   main stream:  pack(0) pack(1) | wait gather(0) sweep(0) reduce(0) | pack(2) wait gather(1) sweep(1) reduce(1) | to_panel(0) ...
// i.e. the all-gather of chunk i + 1 and the reduce-scatter of chunk i - 1 run under the sweep of chunk i.  Xt column groups,
// the partial-product buffer and the receive buffer alternate with the chunk parity.  Same kernels, same sums, same result as
// the serial path (which the test transports and single-chunk applies keep using).
static int apply_sym_overlapped(E* e, int which, OpDesc& o, const double* src, int k, double* dst, bool timed, bool inner) {
  const int step = 32;
  const int nchunks = (k + step - 1) / step;
  if (!e->ov_ready) {
    // everything into locals first: a failure half-way must not leave a stream without its events or buffers behind
    // (later calls would skip this block and launch on null handles); committed to the engine only when complete
    hipStream_t cs = nullptr;
    hipEvent_t evs[8] = {};
    double* bufs[4] = {};
    auto undo = [&]() {
      for (hipEvent_t v : evs) if (v) (void)hipEventDestroy(v);
      for (double* b : bufs) if (b) (void)hipFree(b);
      if (cs) (void)hipStreamDestroy(cs);
    };
    bool ok = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < 8; ++i) ok = hipEventCreateWithFlags(&evs[i], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < 2; ++i) {
      ok = hipMalloc(&bufs[i], sizeof(double) * (size_t)e->nranks * (size_t)e->nslab * 32) == hipSuccess &&
           hipMalloc(&bufs[2 + i], sizeof(double) * (size_t)e->nslab * 32) == hipSuccess;
    }
    if (!ok) {
      (void)hipGetLastError();
      undo();
      return 2;                                        // the caller runs the serial path
    }
    e->comm_stream = cs;
    for (int i = 0; i < 2; ++i) {
      e->ov_packed[i] = evs[4 * i]; e->ov_gathered[i] = evs[4 * i + 1]; e->ov_reduced[i] = evs[4 * i + 2]; e->ov_scattered[i] = evs[4 * i + 3];
      e->sym_wpart2[i] = bufs[i]; e->sym_wrecv2[i] = bufs[2 + i];
    }
    e->ov_ready = true;
  }
  const int64_t total_rows = (int64_t)e->nranks * e->nslab;
  const bool use32 = inner && inner_f32_tiles(e, o);
  const int R = 2;                                     // 32-column chunks: the paired two-block-row schedule
  const E::SymPlan* pl = &e->sym_plan[0];
  const int64_t dstride = (int64_t)pl->nitems * R * 16 * SYM_TB, tstride = pl->zslots * 16 * SYM_TB;
  if (sym_ensure_slabs(e, (size_t)2 * (size_t)(dstride + tstride) + 1) != 0) return 2;   // serial path: it degrades 4 -> 2 -> 1 column groups
  int slot = -1;
  const double stored = o.kind == DAV_KIND_DENSE ? (use32 ? 4.0 : 8.0) * 0.5 * (double)e->n * ((double)e->n + 1.0) / e->nranks : 0.0;
  if (timed) CHK(timed_begin(e, which == DAV_OP_A ? 0 : 2, stored * nchunks + 16.0 * (double)e->n * k, &slot));
  auto cols = [&](int i) { return std::min(step, k - i * step); };
  auto xt_of = [&](int i) { return e->xt + (size_t)(i & 1) * 2 * e->xt_group_stride; };
  auto pack_and_gather = [&](int i) -> int {
    const int p = i & 1, kk = cols(i), ng = (kk + 15) / 16;
    launch_pack_xt(e->stream, src + (int64_t)i * step * e->ldp, e->ldp, e->nloc, e->nslab, kk, xt_of(i), e->xt_group_stride, e->row0);
    HIPCHK(hipEventRecord(e->ov_packed[p], e->stream));
    HIPCHK(hipStreamWaitEvent(e->comm_stream, e->ov_packed[p], 0));
    CHK(coll_group_begin(e));
    for (int g = 0; g < ng; ++g) {
      double* base = xt_of(i) + (size_t)g * e->xt_group_stride;
      NCCLCHK(g_rccl.AllGather(base + e->row0 * 16, base, (size_t)e->nslab * 16, ncclDouble, e->comm, e->comm_stream));
    }
    CHK(coll_group_end(e, "all-gather of a column chunk (second stream)", e->comm_stream));
    HIPCHK(hipEventRecord(e->ov_gathered[p], e->comm_stream));
    return 0;
  };
  auto to_panel = [&](int i) -> int {
    const int p = i & 1, kk = cols(i), ng = (kk + 15) / 16;
    HIPCHK(hipStreamWaitEvent(e->stream, e->ov_scattered[p], 0));
    for (int g = 0; g < ng; ++g)
      launch_chunk_to_panel(e->stream, e->sym_wrecv2[p] + (size_t)g * (size_t)e->nslab * 16, e->nslab, e->nloc, e->nloc_pad,
                            std::min(16, kk - 16 * g), dst + (int64_t)(i * step + 16 * g) * e->ldp, e->ldp);
    return 0;
  };
  CHK(pack_and_gather(0));
  for (int i = 0; i < nchunks; ++i) {
    const int p = i & 1, kk = cols(i), npair = (kk + 15) / 16;
    if (i + 1 < nchunks) CHK(pack_and_gather(i + 1));         // Xt groups of the other parity: last read by the sweep of chunk i - 1
    HIPCHK(hipStreamWaitEvent(e->stream, e->ov_gathered[p], 0));
    int kslot = -1;
    if (timed && which == DAV_OP_A) CHK(timed_begin(e, 4, 2.0 * (double)e->n * (double)e->n * kk / e->nranks, &kslot));
    double* slabT = e->sym_slab + (int64_t)npair * dstride;
    if (pl->nitems > 0)
      sym9_sweep(e, R, o, use32, pl, xt_of(i), kk, e->sym_slab, slabT, npair, dstride, tstride);
    CHK(timed_end(e, kslot));
    // partial of the whole product of this chunk (the buffer of this parity was last read by the reduce-scatter of chunk
    // i - 2, whose completion the main stream waited for when it finished chunk i - 2 below)
    for (int g = 0; g < npair; ++g)
      launch_sym9_reduce(e->stream, e->sym_slab + g * dstride, slabT + g * tstride, pl->row_begin, pl->zslot_begin, e->sym_row_off, R,
                         e->sym_nb, e->nloc, std::min(16, kk - 16 * g), e->sym_wpart2[p] + (size_t)g * (size_t)total_rows * 16, e->ldp,
                         e->nslab, total_rows);
    HIPCHK(hipEventRecord(e->ov_reduced[p], e->stream));
    HIPCHK(hipStreamWaitEvent(e->comm_stream, e->ov_reduced[p], 0));
    CHK(coll_group_begin(e));
    for (int g = 0; g < npair; ++g)
      NCCLCHK(g_rccl.ReduceScatter(e->sym_wpart2[p] + (size_t)g * (size_t)total_rows * 16, e->sym_wrecv2[p] + (size_t)g * (size_t)e->nslab * 16,
                                   (size_t)e->nslab * std::min(16, kk - 16 * g), ncclDouble, ncclSum, e->comm, e->comm_stream));
    CHK(coll_group_end(e, "reduce-scatter of a column chunk (second stream)", e->comm_stream));
    HIPCHK(hipEventRecord(e->ov_scattered[p], e->comm_stream));
    if (i >= 1) CHK(to_panel(i - 1));                      // the previous chunk's rows of W, while this chunk's reduce-scatter runs
    if (which == DAV_OP_A) { e->st.applies += 1; e->st.apply_cols += kk; }
  }
  CHK(to_panel(nchunks - 1));
  CHK(timed_end(e, slot));
  HIPCHK(hipGetLastError());
  return 0;
}

// inner = true: a sweep inside the GJD correction solve (may run on the fp32 copy, dav_set_inner_precision)
static int apply_ptr(E* e, int which, const double* src, int k, double* dst, bool timed, bool inner = false) {
  OpDesc& o = e->op[which];
  if (o.kind == DAV_KIND_NONE) return fail("dav_apply: operator not set");
  if (o.kind == DAV_KIND_HOST) return fail("dav_apply: host operator - move blocks with dav_panel_get/put");
  if (o.kind == DAV_KIND_IDENTITY) {
    launch_copy_columns(e->stream, src, e->ldp, dst, e->ldp, e->nloc_pad, k);
    return 0;
  }
  CHK(need_comm(e));
  if ((o.kind == DAV_KIND_DENSE || o.kind == DAV_KIND_HASHED || o.kind == DAV_KIND_HARNESS) && o.storage == 1) {
    // symmetric-tiled sweep: every off-diagonal tile read (or generated) once, used twice.  16 columns per workgroup; 32
    // columns per launch as paired workgroups that share their tile reads through the memory-side cache.
    // Several ranks: each sweeps the block rows it stores against the all-gathered block and holds a partial of the
    // WHOLE product; one reduce-scatter per 16 columns sums the partials and leaves every rank its row slab.
    static const int pair_env = [] { const char* ev = getenv("DAV_SYM_PAIR"); return ev ? atoi(ev) : 1; }();
    // pairing shares the READS of stored tiles: nothing to share when the entries are generated
    int step = (pair_env && matvec_sym_can_pair() && !e->sym_no_pair && o.kind == DAV_KIND_DENSE) ? 32 : 16;
    // 64 columns (the widest expansion of the doubling policy below a basis of 128) as FOUR column groups in one launch on
    // the super-row kernels: the four workgroups of a work item share every tile read through their XCD's L2.  Same box,
    // N=200000, k=64: two paired launches 102.6 ms, one launch of four groups 93.6 ms (56.9 TFLOP/s).  DAV_SYM_QUAD=0: off.
    static const int quad_env = [] { const char* ev = getenv("DAV_SYM_QUAD"); return ev ? atoi(ev) : 1; }();
    if (quad_env && step == 32 && k >= 64 && !inner && !e->sym_no_quad && sym_schedule(e, 16) == 2 && !has_comm(e)) step = 64;
    // several ranks - or a communicator on a single rank (DAVIDSON_FORCE_RCCL=1: the GPU tests run the all-gather and the
    // reduce-scatter of this path through RCCL on a one-GPU box)
    const bool multi = e->nranks > 1 || has_comm(e);
    if (multi && !e->sym_wpart) {
      HIPCHK(hipMalloc(&e->sym_wpart, sizeof(double) * (size_t)e->nranks * (size_t)e->nslab * 32));
      HIPCHK(hipMalloc(&e->sym_wrecv, sizeof(double) * (size_t)e->nslab * 32));
    }
    const int64_t* owned = multi ? e->sym_row_off : nullptr;
    const int64_t total_rows = (int64_t)e->nranks * e->nslab;
    {
      // Opt-in (DAV_SYM_OVERLAP=1) until it has run on a multi-GPU node: the pipeline is exercised through a 1-rank RCCL
      // communicator only, and a second stream on one communicator is exactly the kind of thing that must be seen on real
      // links before it becomes the default of a run nobody can watch
      static const int overlap_env = [] { const char* ev = getenv("DAV_SYM_OVERLAP"); return ev ? atoi(ev) : 0; }();
      if (overlap_env && e->comm && step == 32 && k > 32 && o.kind == DAV_KIND_DENSE && sym_schedule(e, 16) == 2) {
        const int rc = apply_sym_overlapped(e, which, o, src, k, dst, timed, inner);
        if (rc != 2) return rc;                        // 2: its streams / buffers / slabs could not be set up - serial path below
      }
    }
    for (int c = 0; c < k; c += step) {
      int kk = std::min(step, k - c);
      int npair = (kk + 15) / 16;
      const bool use32 = inner && inner_f32_tiles(e, o);
      int R = o.kind == DAV_KIND_HARNESS ? 1 : sym_schedule(e, std::min(kk, 16));
      if (use32 && R == 1) R = 2;            // the fp32 tiles are read by the super-row kernels only
      const E::SymPlan* pl = R > 1 ? &e->sym_plan[R == 4 ? 1 : 0] : nullptr;
      const int64_t dstride = R > 1 ? (int64_t)pl->nitems * R * 16 * SYM_TB : (int64_t)e->sym_nitems * 16 * SYM_TB;
      const int64_t tstride = R > 1 ? pl->zslots * 16 * SYM_TB : (int64_t)e->sym_nb * (e->sym_nb - 1) / 2 * 16 * SYM_TB;
      while (sym_ensure_slabs(e, (size_t)npair * (size_t)(dstride + tstride) + 1) != 0) {
        // not enough memory for this many column groups per launch: fewer from here on (4 -> 2 -> 1)
        if (npair < 2) return 1;
        if (npair > 2) { e->sym_no_quad = true; step = 32; kk = 32; npair = 2; }
        else { e->sym_no_pair = true; step = 16; kk = 16; npair = 1; }
      }
      int slot = -1, kslot = -1, cslot = -1;
      const double stored = o.kind == DAV_KIND_DENSE ? (use32 ? 4.0 : 8.0) * 0.5 * (double)e->n * ((double)e->n + 1.0) / e->nranks : 0.0;
      double bytes = stored + 16.0 * (double)e->n * kk;
      // end to end: everything that turns the source columns into W - packing, (all-gather,) the sweep, the fixed-order sum(, reduce-scatter)
      if (timed) CHK(timed_begin(e, which == DAV_OP_A ? 0 : 2, bytes, &slot));
      launch_pack_xt(e->stream, src + (int64_t)c * e->ldp, e->ldp, e->nloc, e->nslab, kk, e->xt, e->xt_group_stride, e->row0);
      if (multi) {
        CHK(timed_begin(e, 3, 0, &cslot));
        CHK(coll_group_begin(e));
        for (int g = 0; g < npair; ++g) {
          double* base = e->xt + g * e->xt_group_stride;
          CHK(coll_allgather(e, base + e->row0 * 16, base, (size_t)e->nslab * 16));
        }
        CHK(coll_group_end(e, "all-gather of the new block", e->stream));
        CHK(timed_end(e, cslot));
      }
      if (timed && which == DAV_OP_A) CHK(timed_begin(e, 4, 2.0 * (double)e->n * (double)e->n * kk / e->nranks, &kslot));
      double* slabT = e->sym_slab + (int64_t)npair * dstride;
      const int nitems = R > 1 ? pl->nitems : e->sym_nitems;
      if (nitems > 0) {                      // a rank can be left without a block row (more ranks than groups of block rows)
        if (R > 1)
          sym9_sweep(e, R, o, use32, pl, e->xt, kk, e->sym_slab, slabT, npair, dstride, tstride);
        else if (o.kind != DAV_KIND_DENSE)
          launch_matvec_sym_generated(e->stream, op_params(o), e->n, e->sym_items, e->sym_nitems, e->xt, kk, e->sym_slab, slabT, npair,
                                      e->xt_group_stride, dstride, tstride);
        else
          launch_matvec_sym(e->stream, o.a, e->sym_row_off, e->sym_items, e->sym_nitems, e->xt, kk, e->sym_slab, slabT, npair,
                            e->xt_group_stride, dstride, tstride);
      }
      CHK(timed_end(e, kslot));
      for (int g = 0; g < npair; ++g) {
        const int kg = std::min(16, kk - 16 * g);
        double* out = multi ? e->sym_wpart + (size_t)g * (size_t)total_rows * 16 : dst + (int64_t)(c + 16 * g) * e->ldp;
        if (R > 1)
          launch_sym9_reduce(e->stream, e->sym_slab + g * dstride, slabT + g * tstride, pl->row_begin, pl->zslot_begin, owned, R, e->sym_nb,
                             e->nloc, kg, out, e->ldp, multi ? e->nslab : 0, total_rows);
        else
          launch_sym_reduce(e->stream, e->sym_slab + g * dstride, slabT + g * tstride, e->sym_row_begin, owned, e->sym_nb, e->nloc, kg,
                            out, e->ldp, multi ? e->nslab : 0, total_rows);
      }
      if (multi) {
        CHK(timed_begin(e, 3, 0, &cslot));
        CHK(coll_group_begin(e));
        for (int g = 0; g < npair; ++g) {
          const int kg = std::min(16, kk - 16 * g);
          CHK(coll_reduce_scatter(e, e->sym_wpart + (size_t)g * (size_t)total_rows * 16, e->sym_wrecv + (size_t)g * (size_t)e->nslab * 16,
                                  (size_t)e->nslab * kg));
        }
        CHK(coll_group_end(e, "reduce-scatter of the partial products", e->stream));
        CHK(timed_end(e, cslot));
        for (int g = 0; g < npair; ++g)
          launch_chunk_to_panel(e->stream, e->sym_wrecv + (size_t)g * (size_t)e->nslab * 16, e->nslab, e->nloc, e->nloc_pad,
                                std::min(16, kk - 16 * g), dst + (int64_t)(c + 16 * g) * e->ldp, e->ldp);
      }
      CHK(timed_end(e, slot));
      if (which == DAV_OP_A) {
        e->st.applies += 1;
        e->st.apply_cols += kk;
      }
    }
    HIPCHK(hipGetLastError());
    return 0;
  }
  for (int c = 0; c < k; c += 64) {
    int kk = std::min(64, k - c);
    int groups = (kk + 15) / 16;
    int ngroups = groups == 3 ? 4 : groups;
    int slot = -1, kslot = -1;
    double bytes = 8.0 * (double)e->nloc * (double)e->n + 16.0 * (double)e->n * kk;
    if (timed) CHK(timed_begin(e, which == DAV_OP_A ? 0 : 2, bytes, &slot));
    launch_pack_xt(e->stream, src + (int64_t)c * e->ldp, e->ldp, e->nloc, e->nslab, kk, e->xt, e->xt_group_stride, e->row0);
    if (has_comm(e)) {
      int cslot;
      CHK(timed_begin(e, 3, 0, &cslot));
      CHK(coll_group_begin(e));
      for (int g = 0; g < groups; ++g) {
        double* base = e->xt + g * e->xt_group_stride;
        CHK(coll_allgather(e, base + e->row0 * 16, base, (size_t)e->nslab * 16));
      }
      CHK(coll_group_end(e, "all-gather of the new block", e->stream));
      CHK(timed_end(e, cslot));
    }
    int nsplit, jc;
    matvec_plan(e->nloc_pad, e->ncols_pad, ngroups, &nsplit, &jc);
    if (matvec_slab_doubles(e->nloc_pad, ngroups, nsplit) > e->scratch_doubles) return fail("matvec scratch too small");
    if (timed && which == DAV_OP_A) CHK(timed_begin(e, 4, 2.0 * (double)e->nloc * (double)e->n * kk, &kslot));
    if (o.kind == DAV_KIND_DENSE)
      launch_matvec_dense(e->stream, o.a, e->nloc_pad, e->nloc_pad, e->ncols_pad, e->xt, e->xt_group_stride, ngroups,
                          e->scratch, nsplit, jc);
    else
      launch_matvec_free(e->stream, op_params(o), e->row0, e->nloc, e->n, e->nloc_pad, e->ncols_pad, e->xt,
                         e->xt_group_stride, ngroups, e->scratch, nsplit, jc);
    CHK(timed_end(e, kslot));               // inner pair: the block-matvec kernel alone
    launch_slab_reduce(e->stream, e->scratch, nsplit, e->nloc_pad, ngroups, e->nloc, kk, dst + (int64_t)c * e->ldp, e->ldp);
    CHK(timed_end(e, slot));                // outer pair: pack + all-gather + kernel + reduction
    if (which == DAV_OP_A) {
      e->st.applies += 1;
      e->st.apply_cols += kk;
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

static int apply_impl(E* e, int which, int src_panel, int c0, int k, int dst_panel, int d0, bool timed) {
  CHK(check_panel(e, src_panel, c0, k));
  CHK(check_panel(e, dst_panel, d0, k));
  return apply_ptr(e, which, panel_ptr(e, src_panel, c0), k, panel_ptr(e, dst_panel, d0), timed);
}

extern "C" int dav_apply(dav_handle_t e, int which, int src_panel, int c0, int k, int dst_panel, int d0) {
  if (which < 0 || which > 1) return fail("dav_apply: bad operator id");
  CHK(bind(e));
  return apply_impl(e, which, src_panel, c0, k, dst_panel, d0, true);
}

// ---- K2 -----------------------------------------------------------------------------------------
// Small results (Gram blocks, norms, dots) reach the host without a copy command: a single rank lets
// the final reduction kernel write straight into device-visible pinned memory and only synchronises
// the stream; with a communicator the partial result is all-reduced in HBM first and then copied.
static double* result_target(E* e) { return has_comm(e) ? e->gram_dev : e->gram_host_dev; }
static int result_fetch(E* e, size_t count) {
  if (has_comm(e)) {
    CHK(coll_allreduce(e, e->gram_dev, count));
    HIPCHK(hipMemcpyAsync(e->gram_host, e->gram_dev, sizeof(double) * count, hipMemcpyDeviceToHost, e->stream));
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

// result left in e->gram_host (p x q, ld = p) after the call
static int gram_impl(E* e, const double* P, int p, const double* Q, int q) {
  if ((size_t)p * q > e->gram_doubles) return fail("gram result exceeds engine capacity");
  if (gram_scratch_doubles(p, q, e->nloc_pad) > e->scratch_doubles) return fail("gram scratch too small");
  int slot;
  CHK(timed_begin(e, 1, 0, &slot));
  launch_gram(e->stream, P, e->ldp, p, Q, e->ldp, q, e->nloc_pad, e->scratch, result_target(e));
  CHK(timed_end(e, slot));
  if (e->nranks > 1) CHK(need_comm(e));
  CHK(result_fetch(e, (size_t)p * q));
  return 0;
}

extern "C" int dav_gram(dav_handle_t e, int panel_p, int p0, int p, int panel_q, int q0, int q, double* out, int64_t ldo) {
  CHK(bind(e));
  CHK(check_panel(e, panel_p, p0, p));
  CHK(check_panel(e, panel_q, q0, q));
  if (p <= 0 || q <= 0 || ldo < p) return fail("dav_gram: bad shape");
  CHK(gram_impl(e, panel_ptr(e, panel_p, p0), p, panel_ptr(e, panel_q, q0), q));
  for (int j = 0; j < q; ++j) std::memcpy(out + j * ldo, e->gram_host + (size_t)j * p, sizeof(double) * p);
  return 0;
}

extern "C" int dav_project(dav_handle_t e, int c0, int k, double* H, int64_t ldh, double* S, int64_t lds) {
  CHK(bind(e));
  int mt = c0 + k;
  CHK(check_panel(e, DAV_PANEL_V, 0, mt));
  if (k <= 0 || (H && ldh < mt)) return fail("dav_project: bad shape");
  const bool both = e->gev && (S != nullptr || (!H && e->rr_on));
  if (both && S && lds < mt) return fail("dav_project: bad shape");
  const size_t blk = (size_t)mt * k;
  if ((both ? 2 : 1) * blk > e->gram_doubles) return fail("gram result exceeds engine capacity");
  if (gram_scratch_doubles(mt, k, e->nloc_pad) > e->scratch_doubles) return fail("gram scratch too small");
  // V^T W_new and (generalized) V^T (B V)_new: two Gram launches, ONE reduction/fetch of both blocks
  int slot;
  CHK(timed_begin(e, 1, 0, &slot));
  launch_gram(e->stream, panel_ptr(e, DAV_PANEL_V, 0), e->ldp, mt, panel_ptr(e, DAV_PANEL_W, c0), e->ldp, k, e->nloc_pad, e->scratch,
              result_target(e));
  if (both)
    launch_gram(e->stream, panel_ptr(e, DAV_PANEL_V, 0), e->ldp, mt, panel_ptr(e, DAV_PANEL_BV, c0), e->ldp, k, e->nloc_pad,
                e->scratch, result_target(e) + blk);
  CHK(timed_end(e, slot));
  if (e->nranks > 1) CHK(need_comm(e));
  if (e->rr_on) {
    // device-resident Rayleigh-Ritz: the new columns also go into the projected matrices kept in HBM; a caller that
    // passes H = NULL (the device-RR driver) gets no host copy and no synchronisation at all
    if (mt > e->rr_ld) return fail("dav_project: basis wider than the device-resident projected matrices");
    if (has_comm(e)) CHK(coll_allreduce(e, e->gram_dev, (both ? 2 : 1) * blk));
    launch_rr_scatter(e->stream, result_target(e), mt, k, c0, e->rr_H, e->rr_ld);
    if (both) launch_rr_scatter(e->stream, result_target(e) + blk, mt, k, c0, e->rr_S, e->rr_ld);
    if (!H) return 0;
    if (has_comm(e)) HIPCHK(hipMemcpyAsync(e->gram_host, e->gram_dev, sizeof(double) * (both ? 2 : 1) * blk, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  } else {
    if (!H) return fail("dav_project: H is NULL (only with dav_rr_enable)");
    CHK(result_fetch(e, (both ? 2 : 1) * blk));
  }
  for (int pass = 0; pass < (both ? 2 : 1); ++pass) {
    double* out = pass == 0 ? H : S;
    int64_t ld = pass == 0 ? ldh : lds;
    const double* res = e->gram_host + pass * blk;
    for (int j = 0; j < k; ++j)
      for (int i = 0; i < mt; ++i) {
        double v = res[(size_t)j * mt + i];
        out[(c0 + j) * ld + i] = v;
        if (i < c0) out[i * ld + (c0 + j)] = v;       // mirror: the projected matrices are symmetric
      }
  }
  return 0;
}

// ---- K6 -----------------------------------------------------------------------------------------
extern "C" int dav_init_basis(dav_handle_t e, int ncols, int64_t* idx_out) {
  CHK(bind(e));
  if (ncols <= 0 || ncols > e->max_cols || ncols > e->n) return fail("dav_init_basis: bad column count");
  const std::vector<double>& d = e->diag_host[DAV_OP_A];
  if (d.empty()) return fail("dav_init_basis: operator A not set");
  // stable selection of the ncols smallest diagonal entries (ties -> lower index first); a property of the
  // resident operator, so it is kept until the diagonal changes (repeated solves on one engine)
  if ((int)e->basis_order.size() < ncols) {
    std::vector<int64_t> all((size_t)e->n);
    std::iota(all.begin(), all.end(), 0);
    int keep = (int)std::min<int64_t>(e->n, std::max(ncols, e->max_cols));
    std::partial_sort(all.begin(), all.begin() + keep, all.end(),
                      [&](int64_t a, int64_t b) { return d[a] < d[b] || (d[a] == d[b] && a < b); });
    all.resize(keep);
    e->basis_order.swap(all);
  }
  std::vector<int64_t> order(e->basis_order.begin(), e->basis_order.begin() + ncols);
  HIPCHK(hipMemcpyAsync(e->idx_dev, order.data(), sizeof(int64_t) * ncols, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  launch_unit_columns(e->stream, e->idx_dev, ncols, e->row0, e->nloc, e->nloc_pad, panel_ptr(e, DAV_PANEL_V, 0), e->ldp);
  for (int w = 0; w < (e->gev ? 2 : 1); ++w) {
    OpDesc& o = e->op[w];
    int dst = w == 0 ? DAV_PANEL_W : DAV_PANEL_BV;
    if (o.kind == DAV_KIND_DENSE && o.storage == 1 && e->nranks == 1)
      launch_gather_columns_sym(e->stream, o.a, e->sym_row_off, e->n, e->nloc_pad, e->idx_dev, ncols, panel_ptr(e, dst, 0), e->ldp);
    else if (o.kind == DAV_KIND_DENSE && o.storage == 0)
      launch_gather_columns(e->stream, o.a, e->nloc_pad, e->nloc_pad, e->idx_dev, ncols, panel_ptr(e, dst, 0), e->ldp);
    else if (o.kind == DAV_KIND_HOST) {
      /* the driver fills W / BV through dav_panel_put */
    } else
      CHK(apply_impl(e, w, DAV_PANEL_V, 0, ncols, dst, 0, false));
  }
  e->m = ncols;
  if (idx_out)
    for (int i = 0; i < ncols; ++i) idx_out[i] = order[i] + 1;
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- K3 -----------------------------------------------------------------------------------------
static int ritz_impl(E* e, int m, int ncorr, int lowest, const double* Y, int64_t ldy, const double* theta, int method,
                     double* resnorm, double* C, int64_t ldc, double* G, int64_t ldg, double* theta_out, double* info_out);

extern "C" int dav_ritz_residual_correction_n(dav_handle_t e, int m, int ncorr, int lowest, const double* Y, int64_t ldy,
                                              const double* theta, int method, double* resnorm) {
  return ritz_impl(e, m, ncorr, lowest, Y, ldy, theta, method, resnorm, nullptr, 0, nullptr, 0, nullptr, nullptr);
}

extern "C" int dav_ritz_residual_correction_g(dav_handle_t e, int m, int ncorr, int lowest, const double* Y, int64_t ldy,
                                              const double* theta, double* resnorm, double* C, int64_t ldc, double* G,
                                              int64_t ldg) {
  if (!C || !G || ldc < m || ldg < ncorr) return fail("dav_ritz_residual_correction_g: bad shape");
  return ritz_impl(e, m, ncorr, lowest, Y, ldy, theta, DAV_METHOD_DPR, resnorm, C, ldc, G, ldg, nullptr, nullptr);
}

// Y == nullptr: the eigenpairs are the device-resident ones of dav_rr_ritz (theta_out receives all m Ritz values)
static int ritz_impl(E* e, int m, int ncorr, int lowest, const double* Y, int64_t ldy, const double* theta, int method,
                     double* resnorm, double* C, int64_t ldc, double* G, int64_t ldg, double* theta_out,
                     double* info_out) {
  CHK(bind(e));
  const bool dev = Y == nullptr;
  if (m <= 0 || lowest <= 0 || lowest > ncorr || ncorr > m || (!dev && ldy < m)) return fail("dav_ritz_residual_correction: bad shape");
  if (method == DAV_METHOD_DPR && m + ncorr > e->cols_alloc) return fail("basis panel too narrow for the correction block");
  CHK(check_panel(e, DAV_PANEL_V, 0, m));
  const double *dY, *dY2, *dTheta;
  int64_t ldm_y, ldm_y2;
  std::vector<double> y2;
  if (dev) {
    dY = e->rr_Ypk; dY2 = e->rr_Y2pk; dTheta = e->rr_thpk;
    ldm_y = ldm_y2 = roundup(m, 4);
  } else {
    y2.resize((size_t)m * ncorr);
    for (int j = 0; j < ncorr; ++j)
      for (int i = 0; i < m; ++i) y2[(size_t)j * m + i] = -Y[j * ldy + i] * theta[j];
    SmallMat sm3[3] = {{Y, ldy, m, ncorr, nullptr, 0}, {y2.data(), m, m, ncorr, nullptr, 0}, {theta, ncorr, ncorr, 1, nullptr, 0}};
    CHK(small_upload_multi(e, 0, sm3, 3));
    dY = sm3[0].dev; dY2 = sm3[1].dev; dTheta = sm3[2].dev;
    ldm_y = sm3[0].ldm; ldm_y2 = sm3[1].ldm;
  }

  int slot;
  CHK(timed_begin(e, 2, 0, &slot));
  // X = V * Y(:, 1:nx)
  int nx = method == DAV_METHOD_GJD ? ncorr : lowest;
  PanelGemmArgs a{};
  a.P1 = panel_ptr(e, DAV_PANEL_V, 0); a.ld1 = e->ldp; a.p1 = m; a.M1 = dY; a.ldm1 = ldm_y;
  a.p2 = 0;
  a.out = panel_ptr(e, DAV_PANEL_X, 0); a.ldo = e->ldp; a.q = nx;
  a.nloc = e->nloc; a.nrows_pad = e->nloc_pad; a.epilogue = 0;
  launch_panel_gemm(e->stream, a);
  // R = W*Y + Z*(-Y*diag(theta)), norms, (DPR) T
  PanelGemmArgs r{};
  r.P1 = panel_ptr(e, DAV_PANEL_W, 0); r.ld1 = e->ldp; r.p1 = m; r.M1 = dY; r.ldm1 = ldm_y;
  r.P2 = panel_ptr(e, e->gev ? DAV_PANEL_BV : DAV_PANEL_V, 0); r.ld2 = e->ldp; r.p2 = m; r.M2 = dY2; r.ldm2 = ldm_y2;
  r.q = ncorr; r.nloc = e->nloc; r.nrows_pad = e->nloc_pad;
  r.theta = dTheta; r.dA = e->op[DAV_OP_A].diag; r.dB = e->gev ? e->op[DAV_OP_B].diag : nullptr;
  r.nnorm = lowest; r.norm_partial = e->norm_partial;
  if (method == DAV_METHOD_DPR) {
    r.out = panel_ptr(e, DAV_PANEL_V, m); r.ldo = e->ldp; r.epilogue = 1;
  } else {
    r.out = panel_ptr(e, DAV_PANEL_R, 0); r.ldo = e->ldp; r.epilogue = 2;
  }
  launch_panel_gemm(e->stream, r);
  launch_norm_finish(e->stream, e->norm_partial, (int)(e->nloc_pad / PG_ROWS), lowest, result_target(e));
  // optionally the Gram block the first orthonormalisation pass needs, [V T]^T T with T = V[:, m:m+ncorr] just
  // written: it rides on the same reduction and the same fetch as the norms (one synchronisation less)
  size_t count = (size_t)lowest;
  const size_t goff = ((size_t)lowest + 7) / 8 * 8;
  const int p = m + ncorr;
  if (C) {
    if (goff + (size_t)p * ncorr > e->gram_doubles) return fail("gram result exceeds engine capacity");
    if (gram_scratch_doubles(p, ncorr, e->nloc_pad) > e->scratch_doubles) return fail("gram scratch too small");
    launch_gram(e->stream, panel_ptr(e, DAV_PANEL_V, 0), e->ldp, p, panel_ptr(e, DAV_PANEL_V, m), e->ldp, ncorr, e->nloc_pad,
                e->scratch, result_target(e) + goff);
    count = goff + (size_t)p * ncorr;
  }
  CHK(timed_end(e, slot));
  if (e->nranks > 1) CHK(need_comm(e));
  if (dev) {
    // the Ritz values (and the eigensolver's status word) ride on the same fetch, behind the all-reduced part
    const size_t toff = (count + 7) / 8 * 8;
    if (toff + (size_t)roundup(m + 1, 2) > e->gram_doubles) return fail("gram result exceeds engine capacity");   // the tail copy moves whole pairs
    if (has_comm(e)) CHK(coll_allreduce(e, e->gram_dev, count));
    launch_copy_columns(e->stream, e->rr_thpk + roundup(ncorr, 64), 2 * (int64_t)roundup(m + 1, 2), result_target(e) + toff,
                        2 * (int64_t)roundup(m + 1, 2), roundup(m + 1, 2), 1);
    if (has_comm(e)) HIPCHK(hipMemcpyAsync(e->gram_host, e->gram_dev, sizeof(double) * (toff + m + 1), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int j = 0; j < m; ++j) theta_out[j] = e->gram_host[toff + j];
    if (info_out) *info_out = e->gram_host[toff + m];
  } else {
    CHK(result_fetch(e, count));
  }
  for (int j = 0; j < lowest; ++j) resnorm[j] = std::sqrt(e->gram_host[j]);
  if (C) {
    const double* gh = e->gram_host + goff;
    for (int j = 0; j < ncorr; ++j) {
      for (int i = 0; i < m; ++i) C[j * ldc + i] = gh[(size_t)j * p + i];
      for (int i = 0; i < ncorr; ++i) G[j * ldg + i] = gh[(size_t)j * p + m + i];
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int dav_ritz_residual_correction(dav_handle_t e, int m, int lowest, const double* Y, int64_t ldy,
                                            const double* theta, int method, double* resnorm) {
  return dav_ritz_residual_correction_n(e, m, m, lowest, Y, ldy, theta, method, resnorm);
}

extern "C" int dav_panel_select(dav_handle_t e, int panel, int c0, int nsel, const int* sel) {
  CHK(bind(e));
  if (nsel < 0 || (nsel > 0 && !sel)) return fail("dav_panel_select: bad arguments");
  for (int i = 0; i < nsel; ++i) {
    if (sel[i] < i || (i > 0 && sel[i] <= sel[i - 1])) return fail("dav_panel_select: indices must be ascending");
    CHK(check_panel(e, panel, c0 + sel[i], 1));
    if (sel[i] != i)      // columns only move to the left, in ascending order: no overlap
      launch_copy_columns(e->stream, panel_ptr(e, panel, c0 + sel[i]), e->ldp, panel_ptr(e, panel, c0 + i), e->ldp, e->nloc_pad, 1);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- K4 -----------------------------------------------------------------------------------------
extern "C" int dav_ortho_gram(dav_handle_t e, int m, int kt, double* C, int64_t ldc, double* G, int64_t ldg) {
  CHK(bind(e));
  if (m < 0 || kt <= 0 || ldg < kt || (m > 0 && ldc < m)) return fail("dav_ortho_gram: bad shape");
  CHK(check_panel(e, DAV_PANEL_V, 0, m + kt));
  int p = m + kt;
  CHK(gram_impl(e, panel_ptr(e, DAV_PANEL_V, 0), p, panel_ptr(e, DAV_PANEL_V, m), kt));
  for (int j = 0; j < kt; ++j) {
    for (int i = 0; i < m; ++i) C[j * ldc + i] = e->gram_host[(size_t)j * p + i];
    for (int i = 0; i < kt; ++i) G[j * ldg + i] = e->gram_host[(size_t)j * p + m + i];
  }
  return 0;
}

extern "C" int dav_ortho_apply(dav_handle_t e, int m, int kt, const double* C, int64_t ldc, const double* M, int64_t ldm) {
  CHK(bind(e));
  if (m < 0 || kt <= 0 || ldm < kt) return fail("dav_ortho_apply: bad shape");
  CHK(check_panel(e, DAV_PANEL_V, 0, m + kt));
  std::vector<double> cm((size_t)std::max(m, 1) * kt, 0.0);       // -(C*M)
  for (int j = 0; j < kt && m > 0; ++j)
    for (int l = 0; l < kt; ++l) {
      double mlj = M[j * ldm + l];
      if (mlj == 0.0) continue;
      for (int i = 0; i < m; ++i) cm[(size_t)j * m + i] -= C[l * ldc + i] * mlj;
    }
  SmallMat sm2[2] = {{M, ldm, kt, kt, nullptr, 0}, {cm.data(), std::max(m, 1), m, kt, nullptr, 0}};
  CHK(small_upload_multi(e, 1, sm2, m > 0 ? 2 : 1));
  const int64_t ld_m = sm2[0].ldm, ld_cm = m > 0 ? sm2[1].ldm : 4;
  int slot;
  CHK(timed_begin(e, 2, 0, &slot));
  PanelGemmArgs a{};
  a.P1 = panel_ptr(e, DAV_PANEL_V, m); a.ld1 = e->ldp; a.p1 = kt; a.M1 = sm2[0].dev; a.ldm1 = ld_m;
  a.P2 = panel_ptr(e, DAV_PANEL_V, 0); a.ld2 = e->ldp; a.p2 = m; a.M2 = sm2[1].dev; a.ldm2 = ld_cm;
  a.out = panel_ptr(e, DAV_PANEL_S, 0); a.ldo = e->ldp; a.q = kt;
  a.nloc = e->nloc; a.nrows_pad = e->nloc_pad; a.epilogue = 0;
  launch_panel_gemm(e->stream, a);
  launch_copy_columns(e->stream, panel_ptr(e, DAV_PANEL_S, 0), e->ldp, panel_ptr(e, DAV_PANEL_V, m), e->ldp, e->nloc_pad, kt);
  CHK(timed_end(e, slot));
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int dav_expand(dav_handle_t e, int m, int kt) {
  CHK(bind(e));
  if (m < 0 || kt <= 0 || m + kt > e->cols_alloc) return fail("dav_expand: bad shape");
  for (int w = 0; w < (e->gev ? 2 : 1); ++w) {
    if (e->op[w].kind == DAV_KIND_HOST) continue;     // driver moves the block through the host callback
    CHK(apply_impl(e, w, DAV_PANEL_V, m, kt, w == 0 ? DAV_PANEL_W : DAV_PANEL_BV, m, true));
  }
  e->m = m + kt;
  return 0;
}

// ---- K5 -----------------------------------------------------------------------------------------
extern "C" int dav_panel_transform(dav_handle_t e, int src_panel, int s0, int p, const double* M, int64_t ldm, int q,
                                   int dst_panel, int d0) {
  CHK(bind(e));
  CHK(check_panel(e, src_panel, s0, p));
  CHK(check_panel(e, dst_panel, d0, q));
  if (p <= 0 || q <= 0 || ldm < p) return fail("dav_panel_transform: bad shape");
  if (q > e->cols_alloc) return fail("dav_panel_transform: too many output columns");
  int64_t ld_m;
  CHK(small_upload(e, 3, M, ldm, p, q, &ld_m));
  int slot;
  CHK(timed_begin(e, 2, 0, &slot));
  PanelGemmArgs a{};
  a.P1 = panel_ptr(e, src_panel, s0); a.ld1 = e->ldp; a.p1 = p; a.M1 = e->sm[3].dev; a.ldm1 = ld_m;
  a.p2 = 0;
  a.nloc = e->nloc; a.nrows_pad = e->nloc_pad; a.epilogue = 0; a.q = q; a.ldo = e->ldp;
  bool overlap = (src_panel == dst_panel);
  a.out = overlap ? panel_ptr(e, DAV_PANEL_S, 0) : panel_ptr(e, dst_panel, d0);
  if (overlap && src_panel == DAV_PANEL_S) return fail("dav_panel_transform: scratch panel cannot be transformed in place");
  launch_panel_gemm(e->stream, a);
  if (overlap) launch_copy_columns(e->stream, panel_ptr(e, DAV_PANEL_S, 0), e->ldp, panel_ptr(e, dst_panel, d0), e->ldp, e->nloc_pad, q);
  CHK(timed_end(e, slot));
  HIPCHK(hipGetLastError());
  return 0;
}

// V, W = A V and B V are contracted with the same keep columns (src/davidson.f90:218 contracts V and then re-applies the
// operators to the whole basis, :223-226; W Y = A (V Y) holds to rounding, so no sweep of A or B follows a restart)
static int restart_contract(E* e, int m, int keep, const double* Mdev, int64_t ldm) {
  int slot;
  CHK(timed_begin(e, 2, 0, &slot));
  const int panels[3] = {DAV_PANEL_V, DAV_PANEL_W, DAV_PANEL_BV};
  for (int i = 0; i < (e->gev ? 3 : 2); ++i) {
    PanelGemmArgs a{};
    a.P1 = panel_ptr(e, panels[i], 0); a.ld1 = e->ldp; a.p1 = m; a.M1 = Mdev; a.ldm1 = ldm;
    a.p2 = 0;
    a.nloc = e->nloc; a.nrows_pad = e->nloc_pad; a.epilogue = 0; a.q = keep; a.ldo = e->ldp;
    a.out = panel_ptr(e, DAV_PANEL_S, 0);
    launch_panel_gemm(e->stream, a);
    launch_copy_columns(e->stream, panel_ptr(e, DAV_PANEL_S, 0), e->ldp, panel_ptr(e, panels[i], 0), e->ldp, e->nloc_pad, keep);
  }
  CHK(timed_end(e, slot));
  e->m = keep;
  e->st.restarts += 1;
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int dav_restart(dav_handle_t e, int m, int keep, const double* Yk, int64_t ldy) {
  CHK(bind(e));
  if (keep <= 0 || keep > m || m > e->cols_alloc || ldy < m) return fail("dav_restart: bad shape");
  int64_t ld_m;
  CHK(small_upload(e, 3, Yk, ldy, m, keep, &ld_m));
  return restart_contract(e, m, keep, e->sm[3].dev, ld_m);
}

// Several ranks: every rank takes the driver's control decisions (converged? grow or restart? how many columns?) from
// all-reduced small results, so they are identical by construction.  This makes that an enforced invariant instead of an
// assumption: the words (iteration number, basis width, decisions) are all-reduced as max and as -min in one collective;
// a rank that sees them differ returns an error - its process ends with a message, and the launcher tears the group down -
// instead of walking into the next collective alone and hanging everybody.  One tiny all-reduce per outer iteration.
extern "C" int dav_ranks_agree(dav_handle_t e, const double* words, int nwords) {
  if (nwords > 0) e->iter_hint = (long)words[0];       // the driver's first word is its iteration number (the watchdog's message)
  if (e->nranks <= 1 || !has_comm(e)) return 0;
  CHK(bind(e));
  if (nwords <= 0 || (size_t)(2 * nwords) > e->gram_doubles) return fail("dav_ranks_agree: bad word count");
  // max(x) and max(-x) through the SUM all-reduce of the transports: encode every word of rank r in slot r of a
  // nranks-wide row, so that the sum reproduces each rank's value
  const size_t total = (size_t)nwords * e->nranks;
  if (total > e->gram_doubles) return fail("dav_ranks_agree: too many words");
  std::vector<double> buf(total, 0.0);
  for (int i = 0; i < nwords; ++i) buf[(size_t)i * e->nranks + e->rank] = words[i];
  HIPCHK(hipMemcpyAsync(e->gram_dev, buf.data(), sizeof(double) * total, hipMemcpyHostToDevice, e->stream));
  CHK(coll_allreduce(e, e->gram_dev, total));
  HIPCHK(hipMemcpyAsync(buf.data(), e->gram_dev, sizeof(double) * total, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  for (int i = 0; i < nwords; ++i)
    for (int r = 0; r < e->nranks; ++r)
      if (buf[(size_t)i * e->nranks + r] != words[i])
        return fail("ranks disagree on a control decision of the driver loop (word " + std::to_string(i) + ": rank " + std::to_string(r) +
                    " has " + std::to_string(buf[(size_t)i * e->nranks + r]) + ", rank " + std::to_string(e->rank) + " has " +
                    std::to_string(words[i]) + "): inputs or environment differ between the ranks");
  return 0;
}

// ---- device-resident Rayleigh-Ritz -----------------------------------------------------------------
extern "C" int dav_rr_enable(dav_handle_t e, int on) {
  CHK(bind(e));
  if (on && !e->rr_H) {
    // the device eigensolver handles projected problems of order <= 128 (+ one expansion block on top); an engine created
    // for a wider basis can still run narrower solves with it: the device-resident matrices are sized to what it can use
    e->rr_ld = std::min<int64_t>(e->cols_alloc, 160);
    const size_t sq = (size_t)e->rr_ld * e->rr_ld, pk = (size_t)roundup(e->rr_ld, 4) * roundup(e->rr_ld, 64);
    HIPCHK(hipMalloc(&e->rr_H, sizeof(double) * sq));
    HIPCHK(hipMalloc(&e->rr_S, sizeof(double) * sq));
    HIPCHK(hipMalloc(&e->rr_Y, sizeof(double) * sq));
    HIPCHK(hipMalloc(&e->rr_theta, sizeof(double) * e->rr_ld));
    HIPCHK(hipMalloc(&e->rr_work, sizeof(double) * small_eig_work_doubles((int)e->rr_ld)));
    HIPCHK(hipMalloc(&e->rr_info, sizeof(double) * 8));
    HIPCHK(hipMalloc(&e->rr_Ypk, sizeof(double) * pk));
    HIPCHK(hipMalloc(&e->rr_Y2pk, sizeof(double) * pk));
    HIPCHK(hipMalloc(&e->rr_thpk, sizeof(double) * (roundup(e->rr_ld, 64) + e->rr_ld + 8)));
    HIPCHK(hipMemsetAsync(e->rr_H, 0, sizeof(double) * sq, e->stream));
    HIPCHK(hipMemsetAsync(e->rr_S, 0, sizeof(double) * sq, e->stream));
  }
  e->rr_on = on != 0;
  return 0;
}

// dav_project without a host copy of the new block and without a synchronisation (device-resident Rayleigh-Ritz only)
extern "C" int dav_project_dev(dav_handle_t e, int c0, int k) {
  if (!e->rr_on) return fail("dav_project_dev: call dav_rr_enable first");
  return dav_project(e, c0, k, nullptr, 0, nullptr, 0);
}

// Rayleigh-Ritz on the device-resident projected matrices (filled by dav_project) followed by the Ritz phase of
// dav_ritz_residual_correction_n / _g from the eigenpairs where they lie: ONE host synchronisation returns all m Ritz
// values, the residual norms of the first `lowest` pairs and (C != NULL) the Gram blocks of the first
// orthonormalisation pass.  Replaces lapack_generalized_eigensolver (src/lapack_wrapper.f90:14-91) + the H-down /
// Y-up transfers.  sweeps_out: Jacobi sweeps used.
extern "C" int dav_rr_ritz(dav_handle_t e, int m, int ncorr, int lowest, int method, double* theta_out, double* resnorm, double* C,
                           int64_t ldc, double* G, int64_t ldg, int* sweeps_out) {
  CHK(bind(e));
  if (!e->rr_on) return fail("dav_rr_ritz: call dav_rr_enable first");
  if (m <= 0 || m > 128 || m > e->rr_ld || !theta_out || !resnorm) return fail("dav_rr_ritz: bad arguments (order <= 128)");
  // checked BEFORE the eigensolver and the operand packing are launched: they index the device-resident arrays with these
  if (lowest <= 0 || lowest > m || ncorr < 0 || ncorr > m) return fail("dav_rr_ritz: bad shape (0 < lowest <= m, 0 <= ncorr <= m)");
  if (C && (!G || ldc < m || ldg < ncorr)) return fail("dav_rr_ritz: bad shape");
  int slot;
  CHK(timed_begin(e, 1, 0, &slot));
  if (!launch_small_eig(e->stream, e->rr_H, e->rr_ld, e->rr_S, e->rr_ld, m, e->gev != 0, e->rr_theta, e->rr_Y, e->rr_ld, e->rr_work, e->rr_info))
    return fail("dav_rr_ritz: order out of range");
  const int nq = method == DAV_METHOD_GJD ? ncorr : std::max(ncorr, lowest);
  launch_rr_pack(e->stream, e->rr_Y, e->rr_ld, e->rr_theta, m, nq, (int)roundup(m, 4), (int)roundup(nq, 64), e->rr_Ypk, e->rr_Y2pk, e->rr_thpk,
                 e->rr_info, e->rr_thpk + roundup(ncorr, 64));
  CHK(timed_end(e, slot));
  double info = 0.0;
  CHK(ritz_impl(e, m, ncorr, lowest, nullptr, 0, nullptr, method, resnorm, C, ldc, G, ldg, theta_out, &info));
  if (info < 0.0) return fail("dav_rr_ritz: the projected overlap matrix is not positive definite (pivot " + std::to_string((int)-info) + ")");
  if (sweeps_out) *sweeps_out = (int)info;
  return 0;
}

// collapse restart with the device-resident eigenvectors: V <- V * Y(:, 1:keep)   (src/davidson.f90:218)
extern "C" int dav_rr_restart(dav_handle_t e, int m, int keep) {
  CHK(bind(e));
  if (!e->rr_on || keep <= 0 || keep > m || m > e->rr_ld) return fail("dav_rr_restart: bad shape");
  launch_rr_pack(e->stream, e->rr_Y, e->rr_ld, e->rr_theta, m, keep, (int)roundup(m, 4), (int)roundup(keep, 64), e->rr_Ypk, e->rr_Y2pk,
                 e->rr_thpk, e->rr_info, nullptr);
  return restart_contract(e, m, keep, e->rr_Ypk, roundup(m, 4));
}

// the device-resident eigenvectors (m x ncols) and Ritz values, for tests and for callers that want them on the host
extern "C" int dav_rr_get(dav_handle_t e, int m, int ncols, double* theta, double* Y, int64_t ldy) {
  CHK(bind(e));
  if (!e->rr_on || m <= 0 || m > e->rr_ld || ncols > m || ldy < m) return fail("dav_rr_get: bad shape");
  if (theta) HIPCHK(hipMemcpyAsync(theta, e->rr_theta, sizeof(double) * m, hipMemcpyDeviceToHost, e->stream));
  if (Y) HIPCHK(hipMemcpy2DAsync(Y, sizeof(double) * ldy, e->rr_Y, sizeof(double) * e->rr_ld, sizeof(double) * m, (size_t)ncols,
                                 hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

// ---- block movement --------------------------------------------------------------------------------
extern "C" int dav_panel_get(dav_handle_t e, int panel, int c0, int k, double* out, int64_t ld) {
  CHK(bind(e));
  CHK(check_panel(e, panel, c0, k));
  if (ld < e->n) return fail("dav_panel_get: leading dimension too small");
  if (!has_comm(e)) {
    CHK(need_comm(e));
    HIPCHK(hipMemcpy2DAsync(out, sizeof(double) * ld, panel_ptr(e, panel, c0), sizeof(double) * e->ldp,
                            sizeof(double) * e->n, (size_t)k, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
  }
  CHK(need_comm(e));
  // ONE all-gather per batch of columns: the slabs go out as a contiguous nslab x kb block ([rank][column][row] on
  // arrival), in batches the staging buffer holds
  int kmax = (int)std::max<size_t>(1, e->scratch_doubles / ((size_t)e->nranks * (size_t)e->nslab));
  kmax = (int)std::max<size_t>(1, std::min<size_t>((size_t)kmax, test_transport_max_message(e) / (size_t)e->nslab));
  for (int j0 = 0; j0 < k; j0 += kmax) {
    const int kb = std::min(kmax, k - j0);
    const size_t chunk = (size_t)e->nslab * kb;
    double* mine = e->scratch + (size_t)e->rank * chunk;
    HIPCHK(hipMemcpy2DAsync(mine, sizeof(double) * e->nslab, panel_ptr(e, panel, c0 + j0), sizeof(double) * e->ldp,
                            sizeof(double) * e->nslab, (size_t)kb, hipMemcpyDeviceToDevice, e->stream));
    CHK(coll_allgather(e, mine, e->scratch, chunk));
    for (int p = 0; p < e->nranks; ++p) {
      const int64_t r0 = (int64_t)p * e->nslab, nr = std::min<int64_t>(e->nslab, e->n - r0);
      if (nr <= 0) break;
      HIPCHK(hipMemcpy2DAsync(out + (int64_t)j0 * ld + r0, sizeof(double) * ld, e->scratch + (size_t)p * chunk, sizeof(double) * e->nslab,
                              sizeof(double) * nr, (size_t)kb, hipMemcpyDeviceToHost, e->stream));
    }
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  return 0;
}

extern "C" int dav_panel_put(dav_handle_t e, int panel, int c0, int k, const double* in, int64_t ld) {
  CHK(bind(e));
  CHK(check_panel(e, panel, c0, k));
  if (ld < e->n) return fail("dav_panel_put: leading dimension too small");
  if (e->nloc > 0) {
    HIPCHK(hipMemcpy2DAsync(panel_ptr(e, panel, c0), sizeof(double) * e->ldp, in + e->row0, sizeof(double) * ld,
                            sizeof(double) * e->nloc, (size_t)k, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
  }
  return 0;
}

extern "C" int dav_set_width(dav_handle_t e, int m) {
  if (m < 0 || m > e->cols_alloc) return fail("dav_set_width: out of range");
  e->m = m;
  return 0;
}

// ---- measurement --------------------------------------------------------------------------------
extern "C" int dav_bench_apply(dav_handle_t e, int which, int k, int reps, double* avg_ms, double* bytes) {
  double kernel_ms, flops;
  return dav_bench_apply2(e, which, k, reps, avg_ms, &kernel_ms, bytes, &flops);
}

extern "C" int dav_bench_apply2(dav_handle_t e, int which, int k, int reps, double* avg_ms, double* kernel_ms, double* bytes,
                                double* flops) {
  CHK(bind(e));
  if (which != DAV_OP_A) return fail("dav_bench_apply: only operator A is timed");
  if (k <= 0 || k > 64 || reps <= 0) return fail("dav_bench_apply: k must be in 1..64");
  CHK(collect_events(e));
  dav_stats saved = e->st;
  // warm up once, then time whole applies (pack + kernel + reduction; operands resident in HBM)
  const int saved_level = e->timing_level;
  e->timing_level = 1;
  CHK(apply_impl(e, which, DAV_PANEL_V, 0, k, DAV_PANEL_S, 0, false));
  HIPCHK(hipStreamSynchronize(e->stream));
  double total = 0, ktotal = 0;
  int done = 0;
  while (done < reps) {
    int batch = std::min(reps - done, N_EVPAIRS / 8);
    e->st.apply_ms = 0;
    e->st.apply_kernel_ms = 0;
    for (int i = 0; i < batch; ++i) CHK(apply_impl(e, which, DAV_PANEL_V, 0, k, DAV_PANEL_S, 0, true));
    CHK(collect_events(e));
    total += e->st.apply_ms;
    ktotal += e->st.apply_kernel_ms;
    done += batch;
  }
  e->timing_level = saved_level;
  *avg_ms = total / reps;
  *kernel_ms = ktotal / reps;
  const bool sym = e->op[which].storage == 1;
  // per rank: the stored bytes and the flops of the symmetric sweep are dealt out over the ranks like its tiles
  *bytes = (sym ? (e->op[which].kind == DAV_KIND_DENSE ? 8.0 * 0.5 * (double)e->n * ((double)e->n + 1.0) / e->nranks : 0.0)
                : 8.0 * (double)e->nloc * (double)e->n) + 16.0 * (double)e->n * k;
  *flops = 2.0 * (sym ? (double)e->n / e->nranks : (double)e->nloc) * (double)e->n * k;
  e->st = saved;
  return 0;
}

// ---- K7: GJD correction ------------------------------------------------------------------------------
// Solves, for all m Ritz pairs at once,  (I - x x^T)(A - theta_k B)(I - x x^T) t_k = -r_k
// (the systems compute_GJD_generalized_dense builds densely and hands to DSYSV,
// src/davidson.f90:719-732; x is used exactly as the reference uses it - not re-normalised, so in
// the generalized case I - x x^T is not a projector, :721) with a Jacobi-preconditioned MINRES whose
// operator is the K1 block matvec: every inner step costs one sweep of A (and one of B) shared by all
// m right-hand sides.  Inputs: X (Ritz vectors, m columns) and R (residues) as left by
// dav_ritz_residual_correction(..., DAV_METHOD_GJD).  Output: T in V[:, m:2m].
namespace {
struct Gjd {
  E* e;
  int m;
  int64_t ldc;
  std::vector<double> coef;      // 4 x ldc staging
  int coef_slot = 0;
};

static int gjd_coef(Gjd& g, const std::vector<double>* c0, const std::vector<double>* c1, const std::vector<double>* c2,
                    const std::vector<double>* c3, double** dev) {
  // round-robin over two small buffers so that an upload never waits for the kernel that reads the other
  E* e = g.e;
  const std::vector<double>* cs[4] = {c0, c1, c2, c3};
  std::vector<double> flat((size_t)g.m * 4, 0.0);
  for (int t = 0; t < 4; ++t)
    if (cs[t]) std::copy(cs[t]->begin(), cs[t]->end(), flat.begin() + (size_t)t * g.m);
  int slot = 2 + (g.coef_slot++ & 1);
  int64_t ldm;
  CHK(small_upload(e, slot, flat.data(), g.m, g.m, 4, &ldm));
  g.ldc = ldm;
  *dev = e->sm[slot].dev;
  return 0;
}

static int gjd_lincomb(Gjd& g, double* out, const double* a0, const std::vector<double>* c0, const double* a1 = nullptr,
                       const std::vector<double>* c1 = nullptr, const double* a2 = nullptr,
                       const std::vector<double>* c2 = nullptr, const double* a3 = nullptr,
                       const std::vector<double>* c3 = nullptr) {
  E* e = g.e;
  double* dev;
  CHK(gjd_coef(g, c0, c1, c2, c3, &dev));
  LincombArgs a{};
  a.in[0] = a0; a.in[1] = a1 ? a1 : a0; a.in[2] = a2 ? a2 : a0; a.in[3] = a3 ? a3 : a0;
  a.nterms = a3 ? 4 : (a2 ? 3 : (a1 ? 2 : 1));
  a.coef = dev; a.ldc = (int)g.ldc; a.out = out; a.ld = e->ldp; a.nrows_pad = e->nloc_pad; a.m = g.m;
  launch_lincomb(e->stream, a);
  return 0;
}

// up to 4 column-wise dot products, all-reduced, returned as res[s][j]
static int gjd_dots(Gjd& g, int npairs, const double* const* a, const double* const* b, std::vector<double>* res) {
  E* e = g.e;
  DotsArgs d{};
  for (int s = 0; s < npairs; ++s) { d.a[s] = a[s]; d.b[s] = b[s]; }
  d.npairs = npairs; d.ld = e->ldp; d.nrows_pad = e->nloc_pad; d.m = g.m; d.partial = e->norm_partial;
  int nb = coldots_blocks(e->nloc_pad);
  int total = npairs * g.m;
  launch_coldots(e->stream, d);
  launch_norm_finish(e->stream, e->norm_partial, nb, total, result_target(e));
  CHK(result_fetch(e, (size_t)total));
  for (int s = 0; s < npairs; ++s) res[s].assign(e->gram_host + (size_t)s * g.m, e->gram_host + (size_t)(s + 1) * g.m);
  return 0;
}
}  // namespace

extern "C" int dav_gjd_correction(dav_handle_t e, int m, const double* theta, int max_inner, double inner_tol,
                                  int* inner_iters_out) {
  return dav_gjd_correction_n(e, m, m, theta, max_inner, inner_tol, nullptr, inner_iters_out);
}

extern "C" int dav_gjd_correction_n(dav_handle_t e, int mbasis, int m, const double* theta, int max_inner, double inner_tol,
                                    const double* tol_per_col, int* inner_iters_out) {
  CHK(bind(e));
  if (m <= 0 || mbasis < m || mbasis + m > e->cols_alloc || m > e->cols_alloc / 2) return fail("dav_gjd_correction: bad block width");
  if (e->op[DAV_OP_A].kind == DAV_KIND_HOST || e->op[DAV_OP_A].kind == DAV_KIND_NONE)
    return fail("dav_gjd_correction: needs a device operator A");
  const bool gev = e->gev != 0;
  // workspace: 9 column blocks of width cols_alloc/2, allocated on first use
  const int wcols = e->cols_alloc / 2 + 8;
  if (!e->gjd_ws) {
    size_t bytes = sizeof(double) * (size_t)e->ldp * wcols * 9;
    HIPCHK(hipMalloc(&e->gjd_ws, bytes));
    HIPCHK(hipMemsetAsync(e->gjd_ws, 0, bytes, e->stream));
  }
  auto ws = [&](int i) { return e->gjd_ws + (size_t)i * e->ldp * wcols; };
  double* X = panel_ptr(e, DAV_PANEL_X, 0);
  double* T = panel_ptr(e, DAV_PANEL_V, mbasis);
  double* r1 = panel_ptr(e, DAV_PANEL_R, 0);          // becomes b = -r in place
  double *r2 = ws(0), *y = ws(1), *v = ws(2), *w = ws(3), *w1 = ws(4), *w2 = ws(5), *ua = ws(6), *ub = ws(7), *mx = ws(8);

  Gjd g{e, m, 0, {}, 0};
  const std::vector<double> one(m, 1.0), minus_one(m, -1.0), zero(m, 0.0);
  std::vector<double> th(theta, theta + m), active(m, 1.0), res[4];
  int64_t ld_th, ld_act;
  CHK(small_upload(e, 0, th.data(), m, m, 1, &ld_th));
  const double* dA = e->op[DAV_OP_A].diag;
  const double* dB = gev ? e->op[DAV_OP_B].diag : nullptr;

  CHK(gjd_lincomb(g, r1, r1, &minus_one));                                     // b = -r
  CHK(gjd_lincomb(g, T, r1, &zero));                                           // t = 0
  CHK(gjd_lincomb(g, w, r1, &zero));
  CHK(gjd_lincomb(g, w2, r1, &zero));
  launch_copy_columns(e->stream, r1, e->ldp, r2, e->ldp, e->nloc_pad, m);      // r2 = r1
  CHK(small_upload(e, 1, active.data(), m, m, 1, &ld_act));
  // Jacobi preconditioner K = |diag(A) - theta_k diag(B)|, restricted to the complement of x_k:
  //   y = K^-1 r - (x^T K^-1 r / x^T K^-1 x) K^-1 x    (keeps every iterate orthogonal to x_k, so the
  //   null direction of the projected operator can never be amplified)
  launch_precond(e->stream, X, mx, e->ldp, e->nloc, e->nloc_pad, m, e->sm[0].dev, dA, dB, e->sm[1].dev);
  launch_precond(e->stream, r1, y, e->ldp, e->nloc, e->nloc_pad, m, e->sm[0].dev, dA, dB, e->sm[1].dev);
  {
    const double* a[4] = {X, X, r1, r1}; const double* b[4] = {mx, y, y, mx};
    CHK(gjd_dots(g, 4, a, b, res));
  }
  std::vector<double> xmx = res[0], cy(m, 0.0);
  std::vector<double> beta1(m), beta(m), oldb(m, 0.0), dbar(m, 0.0), epsln(m, 0.0), phibar(m), cs(m, -1.0), sn(m, 0.0);
  for (int j = 0; j < m; ++j) {
    cy[j] = xmx[j] > 0 ? -res[1][j] / xmx[j] : 0.0;
    double b2 = res[2][j] + cy[j] * res[3][j];
    beta1[j] = b2 > 0 ? std::sqrt(b2) : 0.0;
    beta[j] = phibar[j] = beta1[j];
    if (!(beta1[j] > 0.0)) active[j] = 0.0;
  }
  int itn = 0;
  std::vector<double> c0(m), c1(m), c2(m), c3(m);
  std::vector<int> stall(m, 0);
  while (itn < max_inner) {
    bool any = false;
    for (int j = 0; j < m; ++j) any = any || active[j] != 0.0;
    if (!any) break;
    ++itn;
    // v = (K^-1 r2 projected) / beta  - orthogonal to x by construction, so (I - x x^T) v = v
    for (int j = 0; j < m; ++j) {
      c0[j] = active[j] != 0.0 ? 1.0 / beta[j] : 0.0;
      c1[j] = c0[j] * cy[j];
    }
    CHK(gjd_lincomb(g, v, y, &c0, mx, &c1));
    // U = A v, UB = B v - only over the 16-column groups that still hold an active pair (a sweep costs one
    // pass per group in symmetric storage; columns outside the range are multiplied by zero below)
    int c_lo = m, c_hi = 0;
    for (int j = 0; j < m; ++j)
      if (active[j] != 0.0) { c_lo = std::min(c_lo, j); c_hi = std::max(c_hi, j + 1); }
    c_lo = c_lo / 16 * 16;
    c_hi = std::min(m, (c_hi + 15) / 16 * 16);
    CHK(apply_ptr(e, DAV_OP_A, v + (size_t)c_lo * e->ldp, c_hi - c_lo, ua + (size_t)c_lo * e->ldp, true, true));
    const double* ubp = v;
    if (gev) {
      CHK(apply_ptr(e, DAV_OP_B, v + (size_t)c_lo * e->ldp, c_hi - c_lo, ub + (size_t)c_lo * e->ldp, true, true));
      ubp = ub;
    }
    if (getenv("DAV_GJD_TRACE")) {
      int na = 0;
      for (int j = 0; j < m; ++j) na += active[j] != 0.0;
      fprintf(stderr, "gjd inner %d: active %d of %d, columns [%d, %d)\n", itn, na, m, c_lo, c_hi);
    }
    // y = (U - theta UB) - (x^T(U - theta UB)) x - (beta/oldb) r1
    {
      const double* a[2] = {X, X}; const double* b[2] = {ua, ubp};
      CHK(gjd_dots(g, 2, a, b, res));
    }
    for (int j = 0; j < m; ++j) {
      double act = active[j];
      c0[j] = act;
      c1[j] = -th[j] * act;
      c2[j] = -(res[0][j] - th[j] * res[1][j]) * act;
      c3[j] = (itn >= 2 && act != 0.0) ? -beta[j] / oldb[j] : 0.0;
    }
    CHK(gjd_lincomb(g, y, ua, &c0, ubp, &c1, X, &c2, r1, &c3));
    // alfa = <v, y>;  y -= (alfa/beta) r2
    {
      const double* a[1] = {v}; const double* b[1] = {y};
      CHK(gjd_dots(g, 1, a, b, res));
    }
    std::vector<double> alfa = res[0];
    for (int j = 0; j < m; ++j) { c0[j] = active[j]; c1[j] = active[j] != 0.0 ? -alfa[j] / beta[j] : 0.0; }
    CHK(gjd_lincomb(g, y, y, &c0, r2, &c1));
    // rotate: r1 <- r2, r2 <- y, y <- (old r1 storage)
    { double* t = r1; r1 = r2; r2 = y; y = t; }
    CHK(small_upload(e, 1, active.data(), m, m, 1, &ld_act));
    launch_precond(e->stream, r2, y, e->ldp, e->nloc, e->nloc_pad, m, e->sm[0].dev, dA, dB, e->sm[1].dev);
    {
      const double* a[3] = {X, r2, r2}; const double* b[3] = {y, y, mx};
      CHK(gjd_dots(g, 3, a, b, res));
    }
    // scalar recurrences (Paige & Saunders), per column
    std::vector<double> oldeps(m), delta(m), gamma(m), phi(m);
    for (int j = 0; j < m; ++j) {
      if (active[j] == 0.0) { oldeps[j] = delta[j] = phi[j] = 0.0; gamma[j] = 1.0; continue; }
      cy[j] = -res[0][j] / xmx[j];
      double b2 = res[1][j] + cy[j] * res[2][j];
      oldb[j] = beta[j];
      beta[j] = b2 > 0 ? std::sqrt(b2) : 0.0;
      oldeps[j] = epsln[j];
      delta[j] = cs[j] * dbar[j] + sn[j] * alfa[j];
      double gbar = sn[j] * dbar[j] - cs[j] * alfa[j];
      epsln[j] = sn[j] * beta[j];
      dbar[j] = -cs[j] * beta[j];
      gamma[j] = std::max(std::sqrt(gbar * gbar + beta[j] * beta[j]), 1e-300);
      cs[j] = gbar / gamma[j];
      sn[j] = beta[j] / gamma[j];
      phi[j] = cs[j] * phibar[j];
      stall[j] = (sn[j] > 0.95) ? stall[j] + 1 : 0;       // |phibar| shrinks by sn each step
      phibar[j] = sn[j] * phibar[j];
    }
    // w_new = (v - oldeps w1 - delta w2) / gamma ;  t += phi w_new
    { double* t = w1; w1 = w2; w2 = w; w = t; }
    for (int j = 0; j < m; ++j) {
      double act = active[j];
      c0[j] = act / gamma[j];
      c1[j] = -oldeps[j] * act / gamma[j];
      c2[j] = -delta[j] * act / gamma[j];
    }
    CHK(gjd_lincomb(g, w, v, &c0, w1, &c1, w2, &c2));
    for (int j = 0; j < m; ++j) c1[j] = phi[j] * active[j];
    CHK(gjd_lincomb(g, T, T, &one, w, &c1));
    for (int j = 0; j < m; ++j)
      if (active[j] != 0.0 && (!(phibar[j] > (tol_per_col ? tol_per_col[j] : inner_tol) * beta1[j]) || !(beta[j] > 0.0) ||
                               (stall[j] >= 8 && phibar[j] < 1e-6 * beta1[j])))
        active[j] = 0.0;      // converged, broke down, or stagnated at the attainable accuracy
  }
  if (inner_iters_out) *inner_iters_out = itn;
  HIPCHK(hipGetLastError());
  return 0;
}
