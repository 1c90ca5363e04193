// Engine core + C ABI (include/davidson_hip.h): lifetime, statistics and HIP-event timing, small-matrix staging, panels and
// block movement.  The engine owns every N-long object in HBM and sequences the kernels of k_*.hip on one HIP stream; the host
// (Fortran driver) keeps only m x m matrices.  Other parts: engine_comm.hip (RCCL, watchdog, collectives, test transports),
// engine_operators.hip (operators and their storage, ingest glue), engine_apply.hip (K1 scheduling), engine_solver.hip
// (projection, Ritz phase, orthonormalisation, restart, device-side Rayleigh-Ritz), engine_gjd.hip (K7).
#include "engine_internal.h"

thread_local std::string g_err;
int fail(const std::string& msg) {
  g_err = msg;
  return 1;
}

int bind(E* e) {
  e->h0_take = e->h0_cols;        // only the call that directly follows dav_init_basis may use its H0 (engine_internal.h)
  e->h0_cols = 0;
  HIPCHK(hipSetDevice(e->device));
  return 0;
}

// Called only where no pair can legitimately be open (API entry points, or timed_begin with ev_open == 0): a pair left
// without its end event by a failed call is dropped here, and the open count starts from zero again - a failure does
// not switch the timing off for the rest of the engine's life.
int collect_events(E* e) {
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
    } else if (e->ev_kind[i] == 8) {           // sweep kernel of the second operator over stored tiles; ev_bytes = its bytes
      e->st.b_stored_kernel_ms += ms; e->st.b_stored_bytes += e->ev_bytes[i]; e->st.b_stored_flops += e->ev_flops[i]; e->st.b_stored_launches += 1;
    } else if (e->ev_kind[i] == 9) {           // ... over generated block rows; ev_bytes = entries evaluated
      e->st.b_generated_kernel_ms += ms; e->st.b_generated_entries += e->ev_bytes[i]; e->st.b_generated_flops += e->ev_flops[i]; e->st.b_generated_launches += 1;
    } else {
      e->st.comm_ms += ms;
      if (e->ev_inside[i]) e->st.apply_comm_ms += ms;
      if (e->ev_kind[i] == 5) { e->st.allgather_ms += ms; e->st.allgather_bytes += e->ev_bytes[i]; }
      else if (e->ev_kind[i] == 6) { e->st.reduce_scatter_ms += ms; e->st.reduce_scatter_bytes += e->ev_bytes[i]; }
      else if (e->ev_kind[i] == 7) { e->st.allreduce_ms += ms; e->st.allreduce_bytes += e->ev_bytes[i]; }
    }
  }
  e->ev_used = 0;
  return 0;
}
// begin/end record an event pair on the stream; collect_events() turns pairs into milliseconds
int timed_begin(E* e, int kind, double bytes, int* slot) { return timed_begin_on(e, kind, bytes, slot, e->stream); }
int timed_end(E* e, int slot) { return timed_end_on(e, slot, e->stream); }
int timed_begin_on(E* e, int kind, double bytes, int* slot, hipStream_t stream) {
  // an event pair costs ~5 us of host time: by default only the block apply is timed - kind 0 = end to end
  // (pack + all-gather + kernel + reduction), kind 4 = the block-matvec kernel alone (the roofline kernel);
  // dav_set_timing(h, 2) adds the Gram / panel / collective phases
  if (e->timing_level < 1 || (kind != 0 && kind != 4 && e->timing_level < 2)) { *slot = -1; return 0; }
  // pairs nest (kernel inside apply): collect only while no pair is open, and leave room for the inner ones
  if (e->ev_open == 0 && e->ev_used > N_EVPAIRS - 12) CHK(collect_events(e));   // room for the inner pairs of one apply (<= 8 chunks + collectives)
  if (e->ev_used == N_EVPAIRS) { *slot = -1; return 0; }
  *slot = e->ev_used++;
  e->ev_inside[*slot] = e->ev_open > 0 && kind >= 5 && kind <= 7;
  e->ev_flops[*slot] = 0.0;
  ++e->ev_open;
  e->ev_done[*slot] = false;
  e->ev_kind[*slot] = kind;
  e->ev_bytes[*slot] = bytes;
  HIPCHK(hipEventRecord(e->ev[*slot][0], stream));
  return 0;
}
int timed_end_on(E* e, int slot, hipStream_t stream) {
  if (slot < 0) return 0;
  if (e->ev_open > 0) --e->ev_open;
  HIPCHK(hipEventRecord(e->ev[slot][1], stream));
  e->ev_done[slot] = true;
  return 0;
}

// MFMA-B operand image of a p x q column-major matrix into dst (pg_image_doubles(p, q) doubles, zeroed by the caller), written in
// the order of the image: per step (4 rows) and tile (16 columns) the 64 entries a wave loads are contiguous
static void pack_operand_image(const double* src, int64_t ld, int p, int q, double* dst) {
  const int64_t tp = pg_image_tiles(q);
  const int nstep = (p + 3) / 4, ntile = (q + 15) / 16;
  for (int s = 0; s < nstep; ++s)
    for (int t = 0; t < ntile; ++t) {
      double* blk = dst + ((int64_t)s * tp + t) * 64;
      const int gmax = std::min(4, p - 4 * s), cmax = std::min(16, q - 16 * t);
      for (int g = 0; g < gmax; ++g) {
        const double* col = src + (int64_t)(16 * t) * ld + 4 * s + g;
        for (int c = 0; c < cmax; ++c) blk[16 * g + c] = col[(int64_t)c * ld];
      }
    }
}
// the same for tests on a host without a GPU (csrc/davidson_hip_private.h): out must hold *doubles_out doubles
extern "C" int dav_pack_operand_image(const double* src, int64_t ld, int p, int q, double* out, int64_t* doubles_out, int64_t* tiles_per_step_out) {
  if (p <= 0 || q <= 0 || ld < p) return fail("dav_pack_operand_image: bad shape");
  if (doubles_out) *doubles_out = pg_image_doubles(p, q);
  if (tiles_per_step_out) *tiles_per_step_out = pg_image_tiles(q);
  if (out) {
    std::memset(out, 0, sizeof(double) * (size_t)pg_image_doubles(p, q));
    pack_operand_image(src, ld, p, q, out);
  }
  return 0;
}

// several small matrices in ONE staging buffer and ONE host-to-device copy (each H2D command costs ~10 us of launch latency,
// which is what the small phases are made of).  Plain: zero padded to pad4(p) x pad64(q), column-major (ldm out); image: the
// MFMA-B operand image panel_gemm_kernel reads (tiles per step out) - the same number of doubles.
int small_upload_multi(E* e, int i, SmallMat* mats, int n) {
  SmallBuf& b = e->sm[i];
  size_t total = 0;
  for (int k = 0; k < n; ++k) total += (size_t)pg_image_doubles(mats[k].p, mats[k].q);
  if (total > e->small_doubles) return fail("small matrices exceed engine capacity");
  if (b.pending) {
    HIPCHK(hipEventSynchronize(b.done));
    b.pending = false;
  }
  std::memset(b.host, 0, sizeof(double) * total);
  size_t off = 0;
  for (int k = 0; k < n; ++k) {
    SmallMat& mt = mats[k];
    double* dst = b.host + off;
    if (mt.image) {
      mt.ldm = pg_image_tiles(mt.q);
      pack_operand_image(mt.src, mt.ld, mt.p, mt.q, dst);
    } else {
      mt.ldm = roundup(std::max(mt.p, 1), 4);
      for (int j = 0; j < mt.q; ++j) std::memcpy(dst + j * mt.ldm, mt.src + j * mt.ld, sizeof(double) * mt.p);
    }
    mt.dev = b.dev + off;
    off += (size_t)pg_image_doubles(mt.p, mt.q);
  }
  HIPCHK(hipMemcpyAsync(b.dev, b.host, sizeof(double) * total, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipEventRecord(b.done, e->stream));
  b.pending = true;
  return 0;
}

// one matrix into small buffer i, plain (see above); returns ldm
int small_upload(E* e, int i, const double* src, int64_t ld, int p, int q, int64_t* ldm_out) {
  SmallMat m{src, ld, p, q, nullptr, 0, false};
  CHK(small_upload_multi(e, i, &m, 1));
  *ldm_out = m.ldm;
  return 0;
}
// ... as operand image; returns its tiles per step
int small_upload_image(E* e, int i, const double* src, int64_t ld, int p, int q, int64_t* tp_out) {
  SmallMat m{src, ld, p, q, nullptr, 0, true};
  CHK(small_upload_multi(e, i, &m, 1));
  *tp_out = m.ldm;
  return 0;
}

double* panel_ptr(E* e, int panel, int col) {
  return e->panel[panel] + (int64_t)col * e->ldp;
}
int check_panel(E* e, int panel, int c0, int k) {
  if (panel < 0 || panel > 5 || !e->panel[panel]) return fail("invalid or unallocated panel id");
  if (c0 < 0 || k < 0 || c0 + k > e->cols_alloc) return fail("panel column range out of bounds");
  return 0;
}

// ------------------------------------------------------------------------------------------------
extern "C" const char* dav_last_error(void) { return g_err.c_str(); }
extern "C" int dav_version(void) { return DAV_HIP_ABI_VERSION; }


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

// every tuning / A-B knob of the engine, read here and nowhere else
Tune tune_from_env() {
  Tune t;
  auto geti = [](const char* name, int dflt) { const char* ev = getenv(name); return ev && *ev ? atoi(ev) : dflt; };
  t.sym_wide = geti("DAV_SYM_WIDE", t.sym_wide);
  t.sym_wide32 = geti("DAV_SYM_WIDE32", t.sym_wide32);
  t.sym_pair = geti("DAV_SYM_PAIR", t.sym_pair);
  t.sym_quad = geti("DAV_SYM_QUAD", t.sym_quad);
  t.sym_overlap = geti("DAV_SYM_OVERLAP", t.sym_overlap);
  t.sym_r = geti("DAV_SYM_R", t.sym_r);
  t.sym_tall = geti("DAV_SYM_TALL", t.sym_tall);
  t.sym_gen_wide = geti("DAV_SYM_GEN_WIDE", 1);
  t.sym_run = std::max(0, geti("DAV_SYM_RUN", 0));
  t.sym_run9 = std::max(0, geti("DAV_SYM_RUN9", 0));
  t.sym_mfma4 = geti("DAV_SYM_MFMA4", t.sym_mfma4);
  t.mv_target = std::max(0, geti("DAV_MV_TARGET", 0));
  t.mv_nsplit = std::max(0, geti("DAV_MV_NSPLIT", 0));
  t.b_resident = geti("DAV_B_RESIDENT", t.b_resident);
  t.coll_direct = geti("DAV_COLL_DIRECT", t.coll_direct);
  t.pg_pin = geti("DAV_PG_PIN", t.pg_pin);
  t.gram_wgs = geti("DAV_GRAM_WGS", t.gram_wgs);
  gram_set_fuse_chunks(geti("DAV_GRAM_FUSE", 0));
  t.gjd_trace = getenv("DAV_GJD_TRACE") != nullptr;
  t.harness_libm = geti("DAV_HARNESS_LIBM", 0);
  t.no_h0 = geti("DAV_NO_H0", 0);
  t.coll_select = geti("DAV_COLL_SELECT", 1);
  t.coll_trial_corrupt = geti("DAV_COLL_TRIAL_CORRUPT", 0);
  t.coll_forced = getenv("DAV_SYM_OVERLAP") != nullptr || getenv("DAV_COLL_DIRECT") != nullptr;
  return t;
}

// ---- buffer cache (engine_internal.h) -------------------------------------------------------------------------------------------
namespace {
struct PoolBlock { void* p; size_t bytes; int device; int kind; uint64_t gen; };      // kind 0: device, 1 + flags: pinned host
struct Pool {
  std::mutex mu;
  std::vector<PoolBlock> live, idle;
  std::vector<std::pair<int, hipStream_t>> streams;      // idle engine streams (device, stream), synchronised before they came here
  uint64_t gen = 0;
  int enabled = -1;
  size_t cap_bytes = 0;          // device blocks
  size_t cap_pinned_bytes = 0;   // pinned host blocks: their own, much smaller cap (page-locked memory is the scarcer resource)
  int destroying = 0;            // inside dav_destroy: the blocks that come back are an engine's whole set, kept for the next dav_create
};
Pool& pool() { static Pool* p = new Pool; return *p; }                               // never destroyed: outlives every engine
constexpr size_t POOL_MIN_BYTES = 64 << 10;

bool pool_enabled(Pool& P) {
  if (P.enabled < 0) {
    const char* v = getenv("DAVIDSON_BUFFER_CACHE");
    P.enabled = (v && atoi(v) == 0) ? 0 : 1;
    // what the cache may hold: 4 GiB by default (DAVIDSON_BUFFER_CACHE_MB) - the call it exists for is the small and frequent one
    // (N=20000: 1.6 GB of tiles, 5 ms of 39 saved); an engine of 160 GB costs its solve, not its allocation, and must not leave the
    // device full behind a single drop-in call
    const char* mb = getenv("DAVIDSON_BUFFER_CACHE_MB");
    P.cap_bytes = (size_t)(mb && atol(mb) >= 0 ? atol(mb) : 4096) << 20;
    const char* pmb = getenv("DAVIDSON_BUFFER_CACHE_PINNED_MB");
    P.cap_pinned_bytes = (size_t)(pmb && atol(pmb) >= 0 ? atol(pmb) : 256) << 20;
  }
  return P.enabled == 1;
}
void pool_release_block(const PoolBlock& b) { if (b.kind == 0) (void)hipFree(b.p); else (void)hipHostFree(b.p); }
// idle blocks older than `before` (all of them: UINT64_MAX) leave the cache; called with the lock held, frees outside of it
std::vector<PoolBlock> pool_take_idle(Pool& P, uint64_t before) {
  std::vector<PoolBlock> out;
  for (size_t i = 0; i < P.idle.size();)
    if (P.idle[i].gen < before) { out.push_back(P.idle[i]); P.idle[i] = P.idle.back(); P.idle.pop_back(); } else ++i;
  return out;
}
hipError_t pool_get(void** p, size_t bytes, int kind, unsigned flags) {
  Pool& P = pool();
  int device = 0;
  (void)hipGetDevice(&device);
  const bool cached = bytes >= POOL_MIN_BYTES;
  if (cached) {
    std::lock_guard<std::mutex> lk(P.mu);
    if (pool_enabled(P))
      for (size_t i = 0; i < P.idle.size(); ++i)
        if (P.idle[i].bytes == bytes && P.idle[i].kind == kind && (kind != 0 || P.idle[i].device == device)) {
          *p = P.idle[i].p;
          P.live.push_back(P.idle[i]);
          P.idle[i] = P.idle.back();
          P.idle.pop_back();
          return hipSuccess;
        }
  }
  hipError_t r = kind == 0 ? hipMalloc(p, bytes) : hipHostMalloc(p, bytes, flags);
  if (r != hipSuccess) {                                   // give the idle blocks back and try once more
    (void)hipGetLastError();
    std::vector<PoolBlock> drop;
    { std::lock_guard<std::mutex> lk(P.mu); drop = pool_take_idle(P, UINT64_MAX); }
    if (!drop.empty()) {
      for (const PoolBlock& b : drop) pool_release_block(b);
      r = kind == 0 ? hipMalloc(p, bytes) : hipHostMalloc(p, bytes, flags);
    }
  }
  if (r == hipSuccess && cached) {
    std::lock_guard<std::mutex> lk(P.mu);
    P.live.push_back(PoolBlock{*p, bytes, device, kind, 0});
  }
  return r;
}
hipError_t pool_put(void* p, int kind) {
  if (!p) return hipSuccess;
  Pool& P = pool();
  {
    std::lock_guard<std::mutex> lk(P.mu);
    for (size_t i = 0; i < P.live.size(); ++i)
      if (P.live[i].p == p) {
        PoolBlock b = P.live[i];
        P.live[i] = P.live.back();
        P.live.pop_back();
        if (!pool_enabled(P)) break;
        // a block freed in the MIDDLE of an engine's life (a regrown slab, the staging panels of an upload) is not what the next
        // dav_create will ask for: above 512 MiB it goes straight back instead of sitting idle through the solve (round-5 advisor; the
        // staging panels of a small drop-in upload - 2 x 165 MB at N=20000 - stay: the next call's upload takes them)
        if (P.destroying == 0 && b.bytes > ((size_t)512 << 20)) break;
        size_t held = 0;
        for (const PoolBlock& q : P.idle) if ((q.kind == 0) == (b.kind == 0)) held += q.bytes;
        if (held + b.bytes > (b.kind == 0 ? P.cap_bytes : P.cap_pinned_bytes)) break;   // over its cap (device and pinned blocks have their own): back it goes
        b.gen = P.gen;
        P.idle.push_back(b);
        return hipSuccess;
      }
  }
  return kind == 0 ? hipFree(p) : hipHostFree(p);
}
}  // namespace

hipError_t pool_malloc_raw(void** p, size_t bytes) { return pool_get(p, bytes, 0, 0); }
hipError_t pool_host_malloc_raw(void** p, size_t bytes, unsigned flags) { return pool_get(p, bytes, 1 + (int)flags, flags); }
hipError_t pool_free(void* p) { return pool_put(p, 0); }
hipError_t pool_host_free(void* p) { return pool_put(p, 1); }
// hipStreamCreate / hipStreamDestroy cost 1.5-2.7 ms each on this runtime (measured, DAV_TIME_LIFECYCLE): the engine's stream is kept too
hipError_t pool_stream_get(hipStream_t* st) {
  Pool& P = pool();
  int device = 0;
  (void)hipGetDevice(&device);
  {
    std::lock_guard<std::mutex> lk(P.mu);
    if (pool_enabled(P))
      for (size_t i = 0; i < P.streams.size(); ++i)
        if (P.streams[i].first == device) {
          *st = P.streams[i].second;
          P.streams[i] = P.streams.back();
          P.streams.pop_back();
          return hipSuccess;
        }
  }
  return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}
void pool_stream_put(hipStream_t st, int device) {
  if (!st) return;
  Pool& P = pool();
  {
    std::lock_guard<std::mutex> lk(P.mu);
    if (pool_enabled(P) && P.streams.size() < 2) { P.streams.emplace_back(device, st); return; }
  }
  (void)hipStreamDestroy(st);
}
size_t pool_idle_device_bytes(int device) {
  Pool& P = pool();
  std::lock_guard<std::mutex> lk(P.mu);
  size_t sum = 0;
  for (const PoolBlock& b : P.idle) if (b.kind == 0 && b.device == device) sum += b.bytes;
  return sum;
}
// measurement door (davidson_hip_private.h): what the cache holds right now - idle device blocks (all devices), idle pinned host blocks
extern "C" int dav_buffer_cache_held(int64_t* device_bytes, int64_t* pinned_bytes) {
  Pool& P = pool();
  std::lock_guard<std::mutex> lk(P.mu);
  int64_t dev = 0, pin = 0;
  for (const PoolBlock& b : P.idle) (b.kind == 0 ? dev : pin) += (int64_t)b.bytes;
  if (device_bytes) *device_bytes = dev;
  if (pinned_bytes) *pinned_bytes = pin;
  return 0;
}
// end of a dav_destroy: what this engine returned stays; what was idle before the engine before it was destroyed goes
void pool_begin_of_destroy() {
  Pool& P = pool();
  std::lock_guard<std::mutex> lk(P.mu);
  ++P.destroying;
}
void pool_end_of_destroy() {
  Pool& P = pool();
  std::vector<PoolBlock> drop;
  { std::lock_guard<std::mutex> lk(P.mu); if (P.destroying > 0) --P.destroying; ++P.gen; if (P.gen >= 2) drop = pool_take_idle(P, P.gen - 1); }
  for (const PoolBlock& b : drop) pool_release_block(b);
}
extern "C" int dav_free_buffers(void) {
  Pool& P = pool();
  std::vector<PoolBlock> drop;
  std::vector<std::pair<int, hipStream_t>> streams;
  { std::lock_guard<std::mutex> lk(P.mu); drop = pool_take_idle(P, UINT64_MAX); streams.swap(P.streams); }
  for (const PoolBlock& b : drop) pool_release_block(b);
  for (auto& s : streams) (void)hipStreamDestroy(s.second);
  return 0;
}

// DAV_TIME_LIFECYCLE=1: where dav_create / dav_destroy spend their time (stderr; the drop-in call pays both per eigenproblem)
struct LapTimer {
  bool on = getenv("DAV_TIME_LIFECYCLE") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  std::string line;
  void lap(const char* what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    char buf[96];
    snprintf(buf, sizeof buf, " %s=%.3f", what, std::chrono::duration<double, std::milli>(now - t).count());
    line += buf;
    t = now;
  }
  void print(const char* head) { if (on) fprintf(stderr, "%s ms:%s\n", head, line.c_str()); }
};

int create_impl(E* e, int device, int64_t n, int max_cols, int gev, int rank, int nranks) {
  LapTimer lt;
  e->tune = tune_from_env();
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
  lt.lap("bind");
  HIPCHK(pool_stream_get(&e->stream));
  lt.lap("stream");
  // sizes first, then ONE device allocation and ONE pinned allocation carved up (256-byte aligned pieces)
  const size_t pbytes = sizeof(double) * (size_t)e->ldp * e->cols_alloc;
  e->xt_group_stride = std::max(e->ncols_pad, e->nloc_pad) * 16;   // sym-tiled sweeps index whole 256-row blocks
  int nsplit, jc;
  matvec_plan(e->nloc_pad, e->ncols_pad, 4, &nsplit, &jc, e->tune.mv_target, e->tune.mv_nsplit);
  const size_t s1 = matvec_slab_doubles(e->nloc_pad, 4, nsplit);
  const size_t s2 = gram_scratch_doubles(e->cols_alloc, e->cols_alloc, e->nloc_pad);
  e->scratch_doubles = std::max(s1, s2);
  e->gram_doubles = 2 * (size_t)e->cols_alloc * e->cols_alloc;      // H and S blocks of one projection side by side
  e->small_doubles = 3 * (size_t)roundup(e->cols_alloc, 4) * roundup(e->cols_alloc, 64);
  size_t dev_total = 0, host_total = 0;
  auto carve = [](size_t& total, size_t bytes) { const size_t off = total; total += (bytes + 255) / 256 * 256; return off; };
  size_t off_panel[6] = {0, 0, 0, 0, 0, 0};
  for (int p = 0; p < 6; ++p)
    if (p != DAV_PANEL_BV || e->gev) off_panel[p] = carve(dev_total, pbytes);
  const size_t off_xt = carve(dev_total, sizeof(double) * e->xt_group_stride * 4);
  const size_t off_diag0 = carve(dev_total, sizeof(double) * e->nloc_pad), off_diag1 = carve(dev_total, sizeof(double) * e->nloc_pad);
  const size_t off_counters = carve(dev_total, sizeof(unsigned) * (GRAM_MAX_COUNTERS + 8));
  const size_t zeroed = dev_total;                                  // everything up to here starts as zeros
  const size_t off_scratch = carve(dev_total, sizeof(double) * e->scratch_doubles);
  const size_t off_gram = carve(dev_total, sizeof(double) * e->gram_doubles);
  const size_t off_gather = carve(dev_total, sizeof(double) * (size_t)e->ncols_pad);
  const size_t off_idx = carve(dev_total, sizeof(int64_t) * e->cols_alloc);
  const size_t off_norm = carve(dev_total, sizeof(double) * (size_t)(e->nloc_pad / PG_ROWS) * e->cols_alloc);
  e->h0_cap = std::min(e->cols_alloc, 128);
  const size_t off_h0 = carve(dev_total, sizeof(double) * 2 * (size_t)e->h0_cap * e->h0_cap);
  size_t off_sm[N_SMALL], off_smh[N_SMALL];
  for (int i = 0; i < N_SMALL; ++i) off_sm[i] = carve(dev_total, sizeof(double) * e->small_doubles);
  const size_t off_gramh = carve(host_total, sizeof(double) * e->gram_doubles);
  const size_t off_agree = carve(host_total, sizeof(double) * 16 * (size_t)e->nranks);
  const size_t off_h0h = carve(host_total, sizeof(double) * 2 * (size_t)e->h0_cap * e->h0_cap);
  for (int i = 0; i < N_SMALL; ++i) off_smh[i] = carve(host_total, sizeof(double) * e->small_doubles);
  {
    hipError_t r = pool_malloc(&e->arena, dev_total);
    if (r != hipSuccess) {
      (void)hipGetLastError();
      e->arena = nullptr;
      return fail("dav_create: hipMalloc of the engine's panels and work space (" + std::to_string(dev_total >> 20) + " MiB) failed: " + hipGetErrorString(r));
    }
  }
  lt.lap("arena");
  HIPCHK(pool_host_malloc(&e->arena_host, host_total, hipHostMallocMapped));
  lt.lap("arena_host");
  char* host_dev = nullptr;
  HIPCHK(hipHostGetDevicePointer((void**)&host_dev, e->arena_host, 0));
  HIPCHK(hipMemsetAsync(e->arena, 0, zeroed, e->stream));
  for (int p = 0; p < 6; ++p)
    if (p != DAV_PANEL_BV || e->gev) e->panel[p] = (double*)(e->arena + off_panel[p]);
  e->xt = (double*)(e->arena + off_xt);
  e->op[0].diag = (double*)(e->arena + off_diag0);
  e->op[1].diag = (double*)(e->arena + off_diag1);
  e->counters = (unsigned*)(e->arena + off_counters);
  e->scratch = (double*)(e->arena + off_scratch);
  e->gram_dev = (double*)(e->arena + off_gram);
  e->gather_dev = (double*)(e->arena + off_gather);
  e->idx_dev = (int64_t*)(e->arena + off_idx);
  e->norm_partial = (double*)(e->arena + off_norm);
  e->gram_host = (double*)(e->arena_host + off_gramh);
  e->gram_host_dev = (double*)(host_dev + off_gramh);
  e->agree_pin = (double*)(e->arena_host + off_agree);
  e->h0_dev = (double*)(e->arena + off_h0);
  e->h0_host = (double*)(e->arena_host + off_h0h);
  for (int i = 0; i < N_SMALL; ++i) {
    e->sm[i].dev = (double*)(e->arena + off_sm[i]);
    e->sm[i].host = (double*)(e->arena_host + off_smh[i]);
    HIPCHK(hipEventCreateWithFlags(&e->sm[i].done, hipEventDisableTiming));
  }
  lt.lap("carve");
  for (int i = 0; i < N_EVPAIRS; ++i) {
    HIPCHK(hipEventCreate(&e->ev[i][0]));
    HIPCHK(hipEventCreate(&e->ev[i][1]));
  }
  lt.lap("events");
  HIPCHK(hipStreamSynchronize(e->stream));
  lt.lap("memset_sync");
  lt.print("dav_create");
  return 0;
}

extern "C" int dav_destroy(dav_handle_t e) {
  if (!e) return 0;
  LapTimer lt;
  hipSetDevice(e->device);
  if (e->stream) hipStreamSynchronize(e->stream);
  if (e->comm_stream) hipStreamSynchronize(e->comm_stream);   // (hipFree used to wait for the device; a cached block is handed on without)
  lt.lap("sync");
  if (e->comm && g_rccl.lib) g_rccl.CommDestroy(e->comm);
  lt.lap("comm");
  pool_begin_of_destroy();
  if (e->arena) pool_free(e->arena);
  if (e->arena_host) pool_host_free(e->arena_host);
  lt.lap("arenas");
  pool_free(e->gjd_ws);
  ingest_release(e);
  shm_release(e);
  pool_free(e->sym_slab);
  pool_free(e->rr_H); pool_free(e->rr_S); pool_free(e->rr_Y); pool_free(e->rr_theta); pool_free(e->rr_work); pool_free(e->rr_info);
  pool_free(e->rr_Ypk); pool_free(e->rr_Y2pk); pool_free(e->rr_thpk);
  lt.lap("ingest_shm_slab_rr");
  for (int i = 0; i < 2; ++i) {
    if (e->ov_packed[i]) hipEventDestroy(e->ov_packed[i]);
    if (e->ov_gathered[i]) hipEventDestroy(e->ov_gathered[i]);
    if (e->ov_reduced[i]) hipEventDestroy(e->ov_reduced[i]);
    if (e->ov_scattered[i]) hipEventDestroy(e->ov_scattered[i]);
    pool_free(e->sym_wpart2[i]);
    pool_free(e->sym_wrecv2[i]);
  }
  if (e->comm_stream) hipStreamDestroy(e->comm_stream);
  if (e->wd) {
    { std::lock_guard<std::mutex> lk(e->wd->mu); e->wd->stop = true; }
    e->wd->cv.notify_all();
    e->wd->th.join();
    for (Watchdog::Lane& l : e->wd->lane) { if (l.check.ev) hipEventDestroy(l.check.ev); if (l.latest.ev) hipEventDestroy(l.latest.ev); }
    delete e->wd;
    e->wd = nullptr;
  }
  lt.lap("overlap_watchdog");
  sym_set_release(e->sym);
  lt.lap("sym_set");
  pool_free(e->sym_wpart);
  pool_free(e->coll_stage);
  pool_free(e->cb_x);
  pool_free(e->sym_wrecv);
  for (int i = 0; i < N_SMALL; ++i)
    if (e->sm[i].done) hipEventDestroy(e->sm[i].done);
  for (int i = 0; i < N_EVPAIRS; ++i) {
    if (e->ev[i][0]) hipEventDestroy(e->ev[i][0]);
    if (e->ev[i][1]) hipEventDestroy(e->ev[i][1]);
  }
  lt.lap("wpart_events");
  for (int w = 0; w < 2; ++w) {
    sym_resident_release(e->op[w]);
    pool_free(e->op[w].a);
    pool_free(e->op[w].a32);
    pool_free(e->op[w].e_table);
    pool_free(e->op[w].l2_table);
    pool_free(e->op[w].dadd_table);
  }
  lt.lap("operators");
  if (e->stream) { (void)hipStreamSynchronize(e->stream); pool_stream_put(e->stream, e->device); }
  lt.lap("stream");
  delete e;
  pool_end_of_destroy();
  lt.lap("pool");
  lt.print("dav_destroy");
  return 0;
}

extern "C" int dav_synchronize(dav_handle_t e) {
  CHK(bind(e));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

extern "C" int dav_get_stats(dav_handle_t e, dav_stats* out) { return dav_get_stats_n(e, out, sizeof(dav_stats)); }

// at most `bytes` bytes of the structure: a caller built against an older, shorter layout passes its own sizeof
extern "C" int dav_get_stats_n(dav_handle_t e, void* out, size_t bytes) {
  if (!out) return fail("dav_get_stats: null output");
  CHK(bind(e));
  CHK(collect_events(e));
  e->st.m = e->m;
  e->st.comm_ranks = e->comm_ranks;
  e->st.comm_overlap = (e->comm && e->tune.sym_overlap != 0) ? 1 : 0;
  std::memcpy(out, &e->st, std::min(bytes, sizeof(dav_stats)));
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

extern "C" int dav_panel_unit_column(dav_handle_t e, int panel, int col, int k) {
  CHK(bind(e));
  CHK(check_panel(e, panel, col, 1));
  if (k < 0) return fail("dav_panel_unit_column: bad index");
  if ((size_t)k >= e->basis_order.size()) return DAV_NO_SUCH_ENTRY;       // not an error: the caller falls back to another direction
  const int64_t idx = e->basis_order[(size_t)k];
  int64_t* slot = e->idx_dev + (e->cols_alloc - 1);                        // (dav_init_basis uses the front of idx_dev; stream order keeps them apart)
  HIPCHK(hipMemcpyAsync(slot, &idx, sizeof(int64_t), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));                                 // idx is a local
  launch_unit_columns(e->stream, slot, 1, e->row0, e->nloc, e->nloc_pad, panel_ptr(e, panel, col), e->ldp);
  HIPCHK(hipGetLastError());
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
