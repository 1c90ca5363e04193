// Internal header of the engine translation units (engine*.hip): the engine object, error / check macros, the RCCL entry
// points, and the functions the parts share.  Nothing here is part of the C ABI (include/davidson_hip.h); internal functions
// have hidden visibility.
#pragma once
#include "davidson_hip_private.h"
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

#pragma GCC visibility push(hidden)
extern thread_local std::string g_err;
int fail(const std::string& msg);
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

// Buffer cache (engine.hip).  The blocks of the engine destroyed last (>= 64 KiB: tiles, panels, slabs, pinned staging) stay allocated and
// are handed to the next engine that asks for exactly those sizes: the drop-in call creates and destroys an engine per call, and
// hipMalloc / hipFree of gigabytes cost it 5 ms of 40 at N=20000.  Blocks that sat idle through a whole create-destroy cycle are freed;
// an allocation that fails frees the idle blocks and tries again, so the cache never turns into an out-of-memory of this process.
// DAVIDSON_BUFFER_CACHE=0 turns it off; dav_free_buffers() gives the idle blocks back (mkl_free_buffers' role).
hipError_t pool_malloc_raw(void** p, size_t bytes);
hipError_t pool_host_malloc_raw(void** p, size_t bytes, unsigned flags);
hipError_t pool_free(void* p);
hipError_t pool_host_free(void* p);
size_t pool_idle_device_bytes(int device);
void pool_begin_of_destroy();
void pool_end_of_destroy();
hipError_t pool_stream_get(hipStream_t* st);          // the engine's stream (non-blocking): created, or the one of the engine destroyed last
void pool_stream_put(hipStream_t st, int device);
template <class T> inline hipError_t pool_malloc(T** p, size_t bytes) { return pool_malloc_raw((void**)p, bytes); }
template <class T> inline hipError_t pool_host_malloc(T** p, size_t bytes, unsigned flags) { return pool_host_malloc_raw((void**)p, bytes, flags); }


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
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
};
extern Rccl g_rccl;
int rccl_load();
#define NCCLCHK(call)                                                                              \
  do {                                                                                             \
    ncclResult_t r_ = (call);                                                                      \
    if (r_ != ncclSuccess) return fail(std::string(#call) + " failed: " + g_rccl.GetErrorString(r_)); \
  } while (0)

struct LocalGroup;
struct ShmGroup;

// ------------------------------------------------------------------------------------------------
// super-row schedules (k_matvec_sym9.hip / k_matvec_symw.hip): plan p = 0 / 1 for R = 2 / 4 block rows per workgroup
struct SymPlan {
  int R = 0, nitems = 0, nsuper = 0;
  int64_t zslots = 0;               // transposed-partial slots: one per (super row, tile column below its last block row)
  int* items = nullptr;             // device: (super row, J0, J1, slab slot) per item, longest first
  int* row_begin = nullptr;         // device: first item of each super row (nsuper + 1)
  int* zslot_begin = nullptr;       // device: first slot of each super row (nsuper + 1)
  int* next_owned = nullptr;        // device: smallest super row >= S of this set (nsuper + 1; nsuper = none): the reduction walks only those
};
// Work lists of the symmetric sweep over a SET of block rows: all block rows of this rank (dav_engine::sym), or - for a
// generated operator that is kept partly resident (OpDesc::res / gen) - the stored and the generated ones.  Several ranks: the
// lower block triangle is dealt out by groups of 4 block rows (what every schedule's super rows nest in), longest group first to
// the least loaded rank (sym_group_owners); a set always consists of whole groups.
struct SymSet {
  std::vector<int64_t> row_off_h;   // per block row: first tile of the row inside the set's storage, -1 = not in the set
  int64_t* row_off = nullptr;       // device copy
  int64_t ntiles = 0;
  int nitems = 0;                   // one-block-row kernel: runs of tiles
  int* items = nullptr;             // device: (I, J0, J1, slot) per item
  int* row_begin = nullptr;         // device: first item of each block row (nb + 1)
  SymPlan plan[2];
  bool built = false;
};
void sym_set_release(SymSet& s);

struct OpDesc {
  int kind = DAV_KIND_NONE;
  double* a = nullptr;       // dense: nloc_pad x ncols_pad, column-major, lda = nloc_pad
  uint64_t seed = 0;
  double sparsity = 0;
  int use_diag = 0;
  double diag_val = 0;
  int trig = 0;
  bool harness_libm = false; // harness operator: entries by the four library calls (Tune::harness_libm when the operator was set)
  double* e_table = nullptr; // device: the caller's exp(real(i) / real(n)) table (harness operator)
  double* dadd_table = nullptr; // device, same length: the diagonal entries of operator A, poly(1) + (double)(float)(i + 1)
  double* l2_table = nullptr; // device: 2 log e_i, roundup(n, 256) + 256 entries, zero behind n (polynomial form of the harness operator)
  double* diag = nullptr;    // device, nloc_pad (local rows)
  int storage = 0;           // dense: 0 = full, 1 = symmetric-tiled (lower block triangle)
  float* a32 = nullptr;      // fp32 copy of the symmetric tiles: operand of the mixed-precision inner sweeps (lazy)
  bool a32_valid = false, a32_refused = false;
  // generated symmetric operator (hashed, storage 1) kept PARTLY resident: the tiles of its longest block rows (I >= res_first)
  // are stored like a dense operator's (res_tiles of them at res_a, addressed through res_row_off), the others are generated in the sweep
  double* res_a = nullptr;
  int64_t res_tiles = 0;
  int res_first = 0;                // first resident block row (resident rows form the tail: the longest ones)
  bool res_decided = false;         // the split was made (at the first sweep of the operator); cleared when the operator is set again
  bool pass_res = false, pass_gen = false;   // the passes every rank makes (agreed on across the ranks when the split was made)
  SymSet* res = nullptr;            // work lists over the resident block rows (tiles at res_a) ...
  SymSet* gen = nullptr;            // ... and over the generated ones
  // DAV_KIND_DEVICE: the caller's own block apply on device memory (dav_set_operator_device)
  dav_device_apply_fn dev_fn = nullptr;
  void* dev_ctx = nullptr;
};

struct SmallBuf {            // device small matrix + pinned staging
  double* dev = nullptr;
  double* host = nullptr;
  hipEvent_t done = nullptr;
  bool pending = false;
};

constexpr int N_SMALL = 4;
constexpr int N_EVPAIRS = 64;

// Tuning / A-B knobs of the sweeps.  Read from the environment ONCE, when the engine is created (tune_from_env in
// engine.hip, called by dav_create): nothing on the apply path calls getenv.  A/B runs set the variables and create a new engine.
struct Tune {
  int sym_wide = 2;       // DAV_SYM_WIDE: 2 = the one-wave-per-SIMD kernel (k_matvec_symw.hip) from 9 columns on, 1 = for more than 16 only, 0 = never
  int sym_wide32 = 1;     // DAV_SYM_WIDE32: its fp32-tile variant for the mixed-precision inner sweeps of up to 16 columns
  int sym_pair = 1;       // DAV_SYM_PAIR: 32 columns per launch (two 16-column groups share their tile reads)
  int sym_quad = 1;       // DAV_SYM_QUAD: 64 columns per launch
  int sym_overlap = 0;    // DAV_SYM_OVERLAP=1: collectives of wide blocks on a second stream under the sweeps (opt-in until it has run over real links)
  int sym_r = 0;          // DAV_SYM_R: 1 | 2 | 4 forces the block rows per workgroup
  int sym_tall = 1;       // DAV_SYM_TALL: four block rows per workgroup at 9-16 columns
  int sym_gen_wide = 1;   // DAV_SYM_GEN_WIDE: the hashed operator at more than 16 columns generates its entries once per 32 columns (0: once per 16)
  int sym_run = 0;        // DAV_SYM_RUN / DAV_SYM_RUN9: run length of the work items (0 = by the tile count)
  int sym_run9 = 0;
  int sym_mfma4 = 1;      // DAV_SYM_MFMA4: the 4x4x4 MFMA at k <= 8
  int64_t mv_target = 0;  // DAV_MV_TARGET / DAV_MV_NSPLIT: grid of the row-slab kernel (0 = by the shape)
  int64_t mv_nsplit = 0;
  int gram_wgs = 0;       // DAV_GRAM_WGS: workgroups the Gram kernel's grid aims at (0 = default)
  int pg_pin = 1;         // DAV_PG_PIN: the panel kernel's k loop in the pinned order (1) or the compiler's (0) (A/B runs)
  int coll_direct = 0;    // DAV_COLL_DIRECT=1: all-gather / reduce-scatter as direct exchanges (grouped send / receive to every peer) instead of RCCL's collectives (opt-in)
  int b_resident = 1;     // DAV_B_RESIDENT: keep what fits of a generated second operator resident as stored tiles (dav_set_operator_hashed, storage 1)
  bool gjd_trace = false; // DAV_GJD_TRACE
  int coll_select = 1;    // DAV_COLL_SELECT=0: no trial of the collective paths at the first wide block (program order, or what DAV_SYM_OVERLAP / DAV_COLL_DIRECT force)
  int coll_trial_corrupt = 0;   // DAV_COLL_TRIAL_CORRUPT=1|2: test hook - spoil that way's trial result on rank 0
  bool coll_forced = false;     // DAV_SYM_OVERLAP or DAV_COLL_DIRECT present in the environment
  int no_h0 = 0;          // DAV_NO_H0=1: the first projection by a Gram product instead of the operator's entries at the start indices (A/B)
  int harness_libm = 0;   // DAV_HARNESS_LIBM=1: the harness operator's entries by four library calls each (A/B; default: the one-variable polynomial)
};
Tune tune_from_env();

// which way the collectives of a wide block of the symmetric sweep go (engine_apply.hip: coll_path_trial)
enum { COLL_PATH_UNDECIDED = -1, COLL_PATH_PROGRAM_ORDER = 0, COLL_PATH_DIRECT = 1, COLL_PATH_SECOND_STREAM = 2 };
struct CollTrial {
  bool ran = false;
  int selected = COLL_PATH_PROGRAM_ORDER, columns = 0;
  double ms[3] = {0, 0, 0}, ms_max[3] = {0, 0, 0};       // this rank's time of the trial block per way; maximum over the ranks
  bool valid[3] = {true, false, false}, valid_all[3] = {true, false, false};
  double differing[3] = {0, 0, 0}, maxdiff[3] = {0, 0, 0};
  std::string message[3];
};

struct Watchdog;
struct dav_engine {
  Tune tune;
  int device = 0;
  hipStream_t stream = nullptr;
  int64_t n = 0, nslab = 0, nloc = 0, row0 = 0, nloc_pad = 0, ncols_pad = 0;
  int rank = 0, nranks = 1, gev = 0;
  int max_cols = 0, cols_alloc = 0;
  int m = 0;
  // Everything dav_create allocates lives in TWO allocations (round 5): `arena` (device: panels, Xt, scratch, small results,
  // staging twins, diagonals, counters) and `arena_host` (pinned, device-visible: result target, small-matrix staging).  A
  // hipFree / hipHostFree costs ~0.2 ms whatever its size (it synchronises the device): 25 of them were most of what
  // dav_destroy took - and with it a sixth of a drop-in call at N=20000.  The members below point into the arenas.
  char* arena = nullptr;
  char* arena_host = nullptr;
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
  unsigned* counters = nullptr;   // zeroed words of the last-workgroup finishes: [0, GRAM_MAX_COUNTERS) Gram tiles, [GRAM_MAX_COUNTERS] the panel norms
  double* gjd_ws = nullptr;       // GJD inner-solver workspace (lazy)
  int storage = 0;                // storage mode for dense operators set after dav_set_storage
  int sym_nb = 0;                 // symmetric-tiled sweep: block rows of the whole matrix
  SymSet sym;                     // work lists over the block rows this rank stores (or generates)
  double* sym_slab = nullptr;     // device: direct slabs (per item) followed by transposed slabs (per tile)
  size_t sym_slab_doubles = 0;    // grown on demand: what the largest launch so far needed (schedule x column groups)
  bool sym_no_pair = false;       // paired 32-column launches did not fit the memory: 16 columns per launch
  bool sym_no_quad = false;       // ... four column groups (64 columns) per launch did not
  int inner_bits = 64;            // 32: the sweeps INSIDE the GJD correction read an fp32 copy of the stored tiles (dav_set_inner_precision)
  // Several ranks: the lower block triangle is dealt out by groups of 4 block rows (what every schedule's super rows
  // nest in), longest group first to the least loaded rank (sym_group_owners).  row_off[I] = first tile of block row I
  // in this rank's storage, -1 = another rank's.
  double* sym_wpart = nullptr;    // several ranks: this rank's partial of the whole product, [rank][column][row of its slab]
  double* sym_wrecv = nullptr;    // ... and the summed chunk the reduce-scatter hands back (nslab x 64)
  double* coll_stage = nullptr;   // DAV_COLL_DIRECT: the peers' chunks of a direct reduce-scatter before the fixed-order sum (grown on demand)
  size_t coll_stage_doubles = 0;
  // RCCL only: a second stream for the collectives of the symmetric sweep, so that the all-gather of the NEXT 32 columns
  // and the reduce-scatter of the PREVIOUS ones run under the sweep of the current ones; buffers alternate by chunk parity
  int coll_path = COLL_PATH_PROGRAM_ORDER;   // UNDECIDED once a communicator of several ranks exists and the environment forced nothing
  CollTrial coll_trial;
  Watchdog* wd = nullptr;         // watches the RCCL collectives of this engine (dav_comm_init)
  int group_depth = 0;            // inside ncclGroupStart / ncclGroupEnd: the group is marked once, at its end
  bool group_timed = false;       // a CollGroup times its members as a whole
  long iter_hint = -1;            // outer iteration the driver is in (dav_ranks_agree), for the watchdog's message
  std::vector<double> agree_words;   // dav_agree_next: the driver's control words, riding on the next all-reduced small result
  std::vector<double> agreed_inputs; // dav_agree_inputs: the solve inputs the ranks verified last
  double* agree_pin = nullptr;       // pinned staging of those words (16 x nranks doubles)
  // H0 = V0^T (Op V0) of the unit columns of dav_init_basis is the operator's entries at (idx_i, idx_j): several ranks of dealt-out
  // tiles sum what each holds of them in the SAME grouped collective that reduce-scatters W0 (one collective per solve fewer);
  // h0_cols is what dav_init_basis left for the dav_project(0, ncols) that follows it - every other API call drops it (bind)
  double* h0_dev = nullptr;          // 2 x h0_cap x h0_cap (operator A, operator B)
  double* cb_x = nullptr;            // several ranks, DAV_KIND_DEVICE: the gathered block, column-major (nranks * nslab) x k
  size_t cb_x_doubles = 0;
  double* h0_host = nullptr;
  int h0_cap = 0, h0_cols = 0, h0_take = 0;
  int h0_kind[2] = {0, 0};           // per operator: 0 none, 1 summed entries in the stash, 2 identity
  hipStream_t comm_stream = nullptr;
  bool ov_ready = false;          // stream, events and buffers of apply_sym_overlapped all exist
  hipEvent_t ov_packed[2] = {nullptr, nullptr}, ov_gathered[2] = {nullptr, nullptr}, ov_reduced[2] = {nullptr, nullptr},
             ov_scattered[2] = {nullptr, nullptr};
  double* sym_wpart2[2] = {nullptr, nullptr};
  double* sym_wrecv2[2] = {nullptr, nullptr};
  // device-resident Rayleigh-Ritz (dav_rr_enable): projected matrices, eigenpairs and their operand images stay in HBM
  bool rr_on = false;
  int64_t rr_ld = 0;
  double *rr_H = nullptr, *rr_S = nullptr, *rr_Y = nullptr, *rr_theta = nullptr, *rr_work = nullptr, *rr_info = nullptr;
  double *rr_Ypk = nullptr, *rr_Y2pk = nullptr, *rr_thpk = nullptr;   // operand images of Y and -Y Theta (kernels.h: pg_image_index), packed theta
  int64_t rr_tp = 4;              // tiles per step of those images as last packed
  SmallBuf sm[N_SMALL];
  size_t small_doubles = 0;
  ncclComm_t comm = nullptr;
  int comm_ranks = 0;             // ncclCommCount of `comm`
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
  double ev_flops[N_EVPAIRS];     // kinds 8 / 9 (sweep kernels of the second operator): flops next to bytes / entries
  int ev_kind[N_EVPAIRS];
  bool ev_done[N_EVPAIRS];        // end event recorded (a call that fails between begin and end leaves a pair without one)
  bool ev_inside[N_EVPAIRS];      // a collective's pair opened inside another pair (the end-to-end pair of an apply)
  int ev_used = 0, ev_open = 0;
  int timing_level = 1;           // 0 = nothing, 1 = block matvec only, 2 = every phase
  bool lazy_x = false;            // dav_set_lazy_ritz_vectors: the Ritz phases compute X only for GJD
};
typedef dav_engine E;

// ---- collective watchdog (failure detection: the reference's convention is print + stop,
// src/lapack_wrapper.f90:395-408) ------------------------------------------------------------------------------------
// A rank whose peer died inside RCCL would wait for ever: the collectives are asynchronous stream operations, the host
// only notices at its next synchronisation, which never returns.  Every RCCL collective (or group of them) is therefore
// followed by an event, and one thread per engine checks that events complete: one that has not after
// DAVIDSON_COLLECTIVE_TIMEOUT seconds (default 600; 0 = no watchdog) prints rank / collective / outer iteration and ends
// the process with exit code 124 - the launcher then tears the group down.  No re-exec, nothing is retried.
// Per stream ("lane": the engine's stream, the communication stream) two events cover the whole tail of the stream however many
// collectives are enqueued between two polls: `check` is the one being watched (with the time it was recorded), `latest` is
// re-recorded behind every further collective.  When `check` completes the watchdog moves on to `latest` (its record time becomes
// the new reference), so the newest collective is always behind a watched event, and a stream that stays busy for longer than
// the timeout with collectives that DO complete is never mistaken for a hang.
struct Watchdog {
  static constexpr int NL = 2;
  struct Mark { hipEvent_t ev = nullptr; const char* what = ""; uint64_t seq = 0; double t0 = 0.0; long iter = -1; };
  struct Lane { hipStream_t stream = nullptr; bool used = false, check_active = false, latest_valid = false; Mark check, latest; };
  Lane lane[NL];
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  bool stop = false;
  uint64_t seq = 0;
  double timeout_s = 0.0;
  int device = 0, rank = 0, nranks = 1;
};

static inline int64_t roundup(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// several small matrices in ONE staging buffer and ONE host-to-device copy (each H2D command costs
// ~10 us of launch latency, which is what the small phases are made of)
struct SmallMat {
  const double* src; int64_t ld; int p, q;   // in
  double* dev; int64_t ldm;                  // out: device address; leading dimension of the padded copy, or - image - tiles per step
  bool image = false;                        // in: upload as MFMA-B operand image (kernels.h: pg_image_index) for panel_gemm_kernel
};

// ---- engine.hip ------------------------------------------------------------------------------------------
int fail(const std::string& msg);
int bind(E* e);
int collect_events(E* e);
int timed_begin(E* e, int kind, double bytes, int* slot);
int timed_end(E* e, int slot);
int timed_begin_on(E* e, int kind, double bytes, int* slot, hipStream_t stream);   // the pair is recorded on `stream`
int timed_end_on(E* e, int slot, hipStream_t stream);
int small_upload(E* e, int i, const double* src, int64_t ld, int p, int q, int64_t* ldm_out);
int small_upload_multi(E* e, int i, SmallMat* mats, int n);
int small_upload_image(E* e, int i, const double* src, int64_t ld, int p, int q, int64_t* tp_out);
double* panel_ptr(E* e, int panel, int col);
int check_panel(E* e, int panel, int c0, int k);
int create_impl(E* e, int device, int64_t n, int max_cols, int gev, int rank, int nranks);
// ---- engine_comm.hip -------------------------------------------------------------------------------------
int rccl_load();
double wall_seconds();
void watchdog_loop(Watchdog* w);
bool has_comm(E* e);
int need_comm(E* e);
bool has_test_transport(const E* e);
size_t test_transport_max_message(const E* e);
int test_allgather(E* e, const double* send, double* recv, size_t count);
int test_allreduce(E* e, double* buf, size_t count);
int test_reduce_scatter(E* e, const double* send, double* recv, size_t count);
void shm_release(E* e);
bool has_test_transport(const E*);
size_t test_transport_max_message(const E*);
int test_allgather(E*, const double*, double*, size_t);
int test_allreduce(E*, double*, size_t);
int test_reduce_scatter(E*, const double*, double*, size_t);
void shm_release(E*);
int watch_mark(E* e, const char* what, hipStream_t stream);
int coll_group_begin(E* e);
int coll_group_end(E* e, const char* what, hipStream_t stream);
void coll_group_abort(E* e);
// RAII around a group of collectives: begin() ... end(); leaving the scope without end() (an early return on an error) closes
// the NCCL group and restores the depth counter
struct CollGroup {
  E* e; bool open = false; int slot = -1;
  explicit CollGroup(E* e_) : e(e_) {}
  // kind 5 / 6: the group is an all-gather / a reduce-scatter of `bytes` payload per rank, timed as a whole on `stream`
  int begin(int kind = 0, double bytes = 0.0, hipStream_t stream = nullptr) {
    if (kind != 0) {         // any transport: with a test transport the members run one by one inside this pair
      int rc = timed_begin_on(e, kind, bytes, &slot, stream ? stream : e->stream);
      if (rc != 0) return rc;
      e->st.collectives += 1;
      e->group_timed = true;
    }
    int rc = coll_group_begin(e); open = rc == 0; return rc;
  }
  int end(const char* what, hipStream_t stream) {
    open = false;
    e->group_timed = false;
    int rc = coll_group_end(e, what, stream);
    if (rc != 0) return rc;
    return timed_end_on(e, slot, stream);
  }
  ~CollGroup() { e->group_timed = false; if (open) coll_group_abort(e); }
};
int coll_allgather(E* e, const double* send, double* recv, size_t count);
int coll_allreduce(E* e, double* buf, size_t count);
int coll_reduce_scatter(E* e, const double* send, double* recv, size_t count);
// ---- engine_operators.hip --------------------------------------------------------------------------------
int refresh_diag_host(E* e, int which);
int sym_schedule(const E* e, int kk, bool stored_fp64);
std::vector<int> sym_group_owners(int nb, int nranks);
int sym_setup(E* e);
int sym_build_set(E* e, int first_block_row, int end_block_row, SymSet& out);
int sym_resident_split(E* e, int which);
void sym_resident_release(OpDesc& o);
int sym_diag(E* e, OpDesc& o);
int sym_ensure_slabs(E* e, size_t doubles);
int alloc_dense(E* e, int which);
int set_dense_from(E* e, int which, const double* a, int64_t lda, hipMemcpyKind kind);
void ingest_release(E* e);
int ingest_acquire(E* e, double** buf, int64_t* cap_rows);
int ingest_commit(E* e, int64_t row0, int64_t nrows);
void ingest_wanted(E* e, int64_t* first, int64_t* count);
OpParams op_params(const OpDesc& o);
// ---- engine_apply.hip ------------------------------------------------------------------------------------
bool inner_f32_tiles(E* e, OpDesc& o);
bool sym_wide_enabled(const E* e);
void sym9_sweep(E* e, int R, const OpDesc& o, bool use32, const SymPlan* pl, const double* xt, int kk, double* slabD, double* slabT,
                       int npair, int64_t dstride, int64_t tstride);
int apply_sym_overlapped(E* e, int which, OpDesc& o, const double* src, int k, double* dst, bool timed, bool inner);
int apply_ptr(E* e, int which, const double* src, int k, double* dst, bool timed, bool inner = false);
int apply_impl(E* e, int which, int src_panel, int c0, int k, int dst_panel, int d0, bool timed);
int gather_columns_sym_multi(E* e, OpDesc& o, int ncols, double* dst, double* h0 = nullptr);
// ---- engine_solver.hip -----------------------------------------------------------------------------------
double* result_target(E* e);
int result_fetch(E* e, size_t count);
int allreduce_with_agreement(E* e, size_t count, size_t* total_out);
int agreement_verify(E* e, size_t count);
int gram_impl(E* e, const double* P, int p, const double* Q, int q);
int ritz_impl(E* e, int m, int ncorr, int lowest, const double* Y, int64_t ldy, const double* theta, int method,
                     double* resnorm, double* C, int64_t ldc, double* G, int64_t ldg, double* theta_out,
                     double* info_out);
int restart_contract(E* e, int m, int keep, const double* Mdev, int64_t ldm);
#pragma GCC visibility pop
