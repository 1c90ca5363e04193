// Engine, communication: RCCL (loaded lazily), communicator set-up, the collective watchdog, the collectives the other parts
// call (all-gather / all-reduce / reduce-scatter, grouped or not), the agreement check of the driver's control decisions - and,
// in the TEST build only (-DDAV_TEST_TRANSPORTS=1, csrc/Makefile), the loopback / shared-memory transports.
#include "engine_internal.h"

Rccl g_rccl;
int rccl_load() {
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
  SYM(Send, "ncclSend")
  SYM(Recv, "ncclRecv")
  SYM(GroupStart, "ncclGroupStart")
  SYM(GroupEnd, "ncclGroupEnd")
  SYM(GetErrorString, "ncclGetErrorString")
  SYM(CommCount, "ncclCommCount")
#undef SYM
  g_rccl.lib = lib;
  return 0;
}

double wall_seconds() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
void watchdog_loop(Watchdog* w) {
  (void)hipSetDevice(w->device);
  std::unique_lock<std::mutex> lk(w->mu);
  while (!w->stop) {
    w->cv.wait_for(lk, std::chrono::milliseconds(200));
    const double now = wall_seconds();
    for (Watchdog::Lane& l : w->lane) {
      if (!l.used || !l.check_active) continue;
      const hipError_t q = hipEventQuery(l.check.ev);
      if (q == hipSuccess) {
        // everything up to `check` is done; what was enqueued behind it is covered by `latest`: watch that one next
        // (its clock starts NOW, when its predecessor has completed - not when it was enqueued: a long backlog of queued collectives
        // that do complete must not eat the newest one's allowance; round-4 advisor)
        if (l.latest_valid) { std::swap(l.check, l.latest); l.latest_valid = false; l.check.t0 = now; }
        else l.check_active = false;
        continue;
      }
      (void)hipGetLastError();
      if (q == hipErrorNotReady && now - l.check.t0 > w->timeout_s) {
        std::fprintf(stderr, "davidson engine: rank %d of %d: collective \"%s\" (number %llu, outer iteration %ld) has not completed after %.0f s "
                             "- a peer is gone or stuck; ending this process (DAVIDSON_COLLECTIVE_TIMEOUT sets the bound)\n",
                     w->rank, w->nranks, l.check.what, (unsigned long long)l.check.seq, l.check.iter, now - l.check.t0);
        std::fflush(stderr);
        _exit(124);
      }
    }
  }
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
  int count = 0;
  NCCLCHK(g_rccl.CommCount(e->comm, &count));          // what RCCL itself reports: goes into dav_stats.comm_ranks (bench.py prints it)
  if (count != e->nranks) return fail("dav_comm_init: the communicator has " + std::to_string(count) + " ranks, the engine was created for " + std::to_string(e->nranks));
  e->comm_ranks = count;
  // the first wide block of the symmetric sweep decides which way its collectives go (engine_apply.hip: coll_path_trial) - with a
  // communicator of several ranks, and unless the environment chose at dav_create
  e->coll_path = (e->nranks > 1 && e->tune.coll_select != 0 && !e->tune.coll_forced) ? COLL_PATH_UNDECIDED : COLL_PATH_PROGRAM_ORDER;
  // the watchdog of this communicator's collectives (DAVIDSON_COLLECTIVE_TIMEOUT seconds; default 600, 0 = none)
  double timeout = 600.0;
  if (const char* ev = getenv("DAVIDSON_COLLECTIVE_TIMEOUT")) timeout = atof(ev);
  if (timeout > 0.0 && !e->wd) {
    Watchdog* w = new Watchdog;
    w->timeout_s = timeout; w->device = e->device; w->rank = e->rank; w->nranks = e->nranks;
    bool ok = true;
    for (Watchdog::Lane& l : w->lane)
      ok = ok && hipEventCreateWithFlags(&l.check.ev, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&l.latest.ev, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      for (Watchdog::Lane& l : w->lane) { if (l.check.ev) (void)hipEventDestroy(l.check.ev); if (l.latest.ev) (void)hipEventDestroy(l.latest.ev); }
      delete w;
      return fail("dav_comm_init: could not create the watchdog's events");
    }
    w->th = std::thread(watchdog_loop, w);
    e->wd = w;
  }
  return 0;
}

// ---- operators ---------------------------------------------------------------------------------
bool has_comm(E* e) { return e->comm != nullptr || e->lg != nullptr || e->shm != nullptr; }
int need_comm(E* e) {
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
  // DAV_TEST_SERIALIZE=1 (read at dav_local_group_join): between two collectives only ONE rank's thread runs at a time, in rank
  // order - each rank has the GPU to itself for its segment, so the HIP-event times of its sweeps are those of a rank that owns
  // a GPU (what a P-rank run on P GPUs would see per rank), not of P kernels time-slicing one device.  A rehearsal tool: the
  // wall time of such a run is the SUM over the ranks.
  bool serialize = false;
  int first = 0;                     // DAV_TEST_SERIALIZE_FIRST: the rank that goes first in every segment (the order is first, first + 1, ... mod n)
  std::mutex mu;
  std::condition_variable cv;
  uint64_t turn = 0;                 // ticket being served: segment * n + rank
  uint64_t segment[16] = {0};        // per rank: collectives passed so far
};
// entry of a collective (the rank's stream is idle): hand the GPU to the next rank
static void lg_release_turn(LocalGroup* g, int rank) {
  if (!g->serialize || g->segment[rank] == 0) return;            // segment 0 (set-up, before the first collective) runs concurrently
  const uint64_t pos = (uint64_t)((rank - g->first + g->n) % g->n);
  { std::lock_guard<std::mutex> lk(g->mu); g->turn = std::max(g->turn, g->segment[rank] * (uint64_t)g->n + pos + 1); }
  g->cv.notify_all();
}
// exit of a collective: wait until the ranks before this one have finished their segment
static void lg_acquire_turn(LocalGroup* g, int rank) {
  if (!g->serialize) return;
  g->segment[rank] += 1;
  const uint64_t pos = (uint64_t)((rank - g->first + g->n) % g->n);
  const uint64_t ticket = g->segment[rank] * (uint64_t)g->n + pos;
  std::unique_lock<std::mutex> lk(g->mu);
  if (pos == 0 && g->turn < ticket) g->turn = ticket;            // the first rank opens the segment (everyone has passed the collective's last barrier)
  g->cv.notify_all();
  g->cv.wait(lk, [&] { return g->turn >= ticket; });
}

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

bool has_test_transport(const E* e) { return e->lg != nullptr || e->shm != nullptr; }
size_t test_transport_max_message(const E* e) { return e->shm ? e->shm->hdr->slot_doubles : (size_t)-1; }

int test_allgather(E* e, const double* send, double* recv, size_t count) {
  if (e->lg) {
    LocalGroup* g = e->lg;
    HIPCHK(hipStreamSynchronize(e->stream));
    lg_release_turn(g, e->rank);
    g->send[e->rank] = send;
    pthread_barrier_wait(&g->bar);
    for (int p = 0; p < g->n; ++p)
      if (recv + (size_t)p * count != g->send[p])
        HIPCHK(hipMemcpyAsync(recv + (size_t)p * count, g->send[p], sizeof(double) * count, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    pthread_barrier_wait(&g->bar);
    lg_acquire_turn(g, e->rank);
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

int test_allreduce(E* e, double* buf, size_t count) {
  if (e->lg) {
    LocalGroup* g = e->lg;
    HIPCHK(hipStreamSynchronize(e->stream));
    lg_release_turn(g, e->rank);
    g->send[e->rank] = buf;
    pthread_barrier_wait(&g->bar);
    std::vector<double> sum(count, 0.0), tmp(count);
    for (int p = 0; p < g->n; ++p) {
      HIPCHK(hipMemcpy(tmp.data(), g->send[p], sizeof(double) * count, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < count; ++i) sum[i] += tmp[i];
    }
    pthread_barrier_wait(&g->bar);            // everyone has read every buffer
    HIPCHK(hipMemcpy(buf, sum.data(), sizeof(double) * count, hipMemcpyHostToDevice));
    lg_acquire_turn(g, e->rank);
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

int test_reduce_scatter(E* e, const double* send, double* recv, size_t count) {
  if (e->lg) {
    LocalGroup* g = e->lg;
    HIPCHK(hipStreamSynchronize(e->stream));
    lg_release_turn(g, e->rank);
    g->send[e->rank] = send;
    pthread_barrier_wait(&g->bar);
    std::vector<double> sum(count, 0.0), tmp(count);
    for (int p = 0; p < g->n; ++p) {                  // rank order: reproducible
      HIPCHK(hipMemcpy(tmp.data(), g->send[p] + (size_t)e->rank * count, sizeof(double) * count, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < count; ++i) sum[i] += tmp[i];
    }
    pthread_barrier_wait(&g->bar);
    HIPCHK(hipMemcpy(recv, sum.data(), sizeof(double) * count, hipMemcpyHostToDevice));
    lg_acquire_turn(g, e->rank);
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

void shm_release(E* e) {
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

// A rank whose thread has no further collective to enter (the end of its work) must pass the turn on, or the ranks behind it
// would wait for ever (DAV_TEST_SERIALIZE=1); harmless otherwise.
extern "C" int dav_local_group_yield(dav_handle_t e) {
  if (!e || !e->lg) return 0;
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  lg_release_turn(e->lg, e->rank);
  return 0;
}

extern "C" int dav_local_group_join(dav_handle_t* handles, int n) {
  if (!handles || n < 1 || n > 16) return fail("dav_local_group_join: 1..16 engines");
  for (int r = 0; r < n; ++r)
    if (!handles[r] || handles[r]->nranks != n || handles[r]->rank != r || handles[r]->lg || handles[r]->comm)
      return fail("dav_local_group_join: engine r must be created with rank r of n and have no transport yet");
  LocalGroup* g = new LocalGroup();
  g->n = n;
  if (const char* ev = getenv("DAV_TEST_SERIALIZE")) g->serialize = ev[0] == '1';
  if (const char* ev = getenv("DAV_TEST_SERIALIZE_FIRST")) g->first = std::max(0, std::min(n - 1, atoi(ev)));
  pthread_barrier_init(&g->bar, nullptr, (unsigned)n);
  for (int r = 0; r < n; ++r) handles[r]->lg = g;
  return 0;
}

#else
bool has_test_transport(const E*) { return false; }
size_t test_transport_max_message(const E*) { return (size_t)-1; }
int test_allgather(E*, const double*, double*, size_t) { return fail("built without test transports"); }
int test_allreduce(E*, double*, size_t) { return fail("built without test transports"); }
int test_reduce_scatter(E*, const double*, double*, size_t) { return fail("built without test transports"); }
void shm_release(E*) {}
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
int watch_mark(E* e, const char* what, hipStream_t stream) {
  Watchdog* w = e->wd;
  if (!w || e->group_depth > 0) return 0;
#if DAV_TEST_TRANSPORTS
  // DAV_TEST_STALL_MS (with DAV_TEST_STALL_FROM = n: only from the n-th collective of this engine on)
  static const long stall_ms = [] { const char* ev = getenv("DAV_TEST_STALL_MS"); return ev ? atol(ev) : 0L; }();
  static const long stall_from = [] { const char* ev = getenv("DAV_TEST_STALL_FROM"); return ev ? atol(ev) : 0L; }();
  if (stall_ms > 0 && (long)w->seq >= stall_from)
    hipLaunchKernelGGL(watchdog_stall_kernel, dim3(1), dim3(1), 0, stream, (unsigned long long)stall_ms * 100000ull);   // 100 MHz counter
#endif
  std::lock_guard<std::mutex> lk(w->mu);
  Watchdog::Lane* l = nullptr;
  for (Watchdog::Lane& c : w->lane)
    if (c.used && c.stream == stream) { l = &c; break; }
  if (!l)
    for (Watchdog::Lane& c : w->lane)
      if (!c.used) { c.used = true; c.stream = stream; l = &c; break; }
  if (!l) return fail("collective watchdog: more streams than lanes");
  Watchdog::Mark& m = l->check_active ? l->latest : l->check;
  HIPCHK(hipEventRecord(m.ev, stream));          // `latest` is re-recorded: it then stands behind the newest collective
  m.what = what; m.seq = w->seq++; m.t0 = wall_seconds(); m.iter = e->iter_hint;
  if (l->check_active) l->latest_valid = true;
  else l->check_active = true;
  return 0;
}
// ncclGroupStart ... ncclGroupEnd around several collectives; the group is marked for the watchdog once, at its end.  A failure
// between begin and end must go through coll_group_abort (CollGroup below does), or the open group would silently switch the
// watchdog off for the rest of the engine's life.
int coll_group_begin(E* e) {
  if (e->comm) { NCCLCHK(g_rccl.GroupStart()); ++e->group_depth; }
  return 0;
}
int coll_group_end(E* e, const char* what, hipStream_t stream) {
  if (e->comm) {
    if (e->group_depth > 0) --e->group_depth;
    NCCLCHK(g_rccl.GroupEnd());
    CHK(watch_mark(e, what, stream));
  }
  return 0;
}
void coll_group_abort(E* e) {
  if (e->comm && e->group_depth > 0) {
    --e->group_depth;
    (void)g_rccl.GroupEnd();
  }
}

// Timing (dav_set_timing level 2): an event pair around the collective on the stream it runs on - kinds 5 / 6 / 7 =
// all-gather / reduce-scatter / all-reduce, with the payload per rank; inside a group the group is timed as a whole.
int coll_allgather(E* e, const double* send, double* recv, size_t count) {
  int slot = -1;
  if (e->group_depth == 0 && !e->group_timed) { CHK(timed_begin(e, 5, 8.0 * (double)count * e->nranks, &slot)); e->st.collectives += 1; }
  if (has_test_transport(e)) {
    CHK(test_allgather(e, send, recv, count));
    return timed_end(e, slot);
  }
  if (e->tune.coll_direct) {
    // opt-in (DAV_COLL_DIRECT=1): a direct exchange - this rank's slab to each of its P - 1 peers, theirs into place - so that the
    // P - 1 point-to-point links of the xGMI mesh carry the all-gather at once, whatever schedule RCCL would pick for ncclAllGather
    NCCLCHK(g_rccl.GroupStart());
    for (int p = 0; p < e->nranks; ++p) {
      if (p == e->rank) continue;
      NCCLCHK(g_rccl.Send(send, count, ncclDouble, p, e->comm, e->stream));
      NCCLCHK(g_rccl.Recv(recv + (size_t)p * count, count, ncclDouble, p, e->comm, e->stream));
    }
    NCCLCHK(g_rccl.GroupEnd());
    if (recv + (size_t)e->rank * count != send)
      HIPCHK(hipMemcpyAsync(recv + (size_t)e->rank * count, send, sizeof(double) * count, hipMemcpyDeviceToDevice, e->stream));
  } else {
    NCCLCHK(g_rccl.AllGather(send, recv, count, ncclDouble, e->comm, e->stream));
  }
  CHK(timed_end(e, slot));
  return watch_mark(e, "all-gather", e->stream);
}

// buf <- sum over ranks of buf (same bits on every rank)
int coll_allreduce(E* e, double* buf, size_t count) {
  int slot = -1;
  if (e->group_depth == 0 && !e->group_timed) { CHK(timed_begin(e, 7, 8.0 * (double)count, &slot)); e->st.collectives += 1; }
  if (has_test_transport(e)) {
    CHK(test_allreduce(e, buf, count));
    return timed_end(e, slot);
  }
  NCCLCHK(g_rccl.AllReduce(buf, buf, count, ncclDouble, ncclSum, e->comm, e->stream));
  CHK(timed_end(e, slot));
  return watch_mark(e, "all-reduce", e->stream);
}

// recv[0 .. count) = sum over ranks p of send_p[rank*count .. (rank+1)*count)  (send holds nranks chunks)
int coll_reduce_scatter(E* e, const double* send, double* recv, size_t count) {
  int slot = -1;
  if (e->group_depth == 0 && !e->group_timed) { CHK(timed_begin(e, 6, 8.0 * (double)count * e->nranks, &slot)); e->st.collectives += 1; }
  if (has_test_transport(e)) {
    CHK(test_reduce_scatter(e, send, recv, count));
    return timed_end(e, slot);
  }
  if (e->tune.coll_direct) {
    // opt-in (DAV_COLL_DIRECT=1): chunk p of this rank's contribution goes straight to rank p, the peers' chunks for this rank arrive
    // in a staging buffer, and a fixed-order sum (rank order) leaves the result - the same volume as the ring, over all links at once.
    // The sum needs the data its exchange moves, and an NCCL group only moves data when it ends: a direct reduce-scatter that is a
    // member of an open CollGroup therefore closes that group in front of itself, runs its exchange as a group of its own, enqueues
    // its sum, and reopens the group for the members behind it (the staging buffer is reused member by member, in stream order).
    const size_t need = (size_t)std::max(e->nranks - 1, 1) * count;
    if (need > e->coll_stage_doubles) {
      HIPCHK(hipStreamSynchronize(e->stream));
      if (e->coll_stage) HIPCHK(pool_free(e->coll_stage));
      e->coll_stage = nullptr; e->coll_stage_doubles = 0;
      HIPCHK(pool_malloc(&e->coll_stage, sizeof(double) * need));
      e->coll_stage_doubles = need;
    }
    const bool reopen = e->group_depth > 0;
    if (reopen) { NCCLCHK(g_rccl.GroupEnd()); }                 // what the open group holds so far goes out now
    NCCLCHK(g_rccl.GroupStart());
    for (int p = 0; p < e->nranks; ++p) {
      if (p == e->rank) continue;
      NCCLCHK(g_rccl.Send(send + (size_t)p * count, count, ncclDouble, p, e->comm, e->stream));
      NCCLCHK(g_rccl.Recv(e->coll_stage + (size_t)(p < e->rank ? p : p - 1) * count, count, ncclDouble, p, e->comm, e->stream));
    }
    NCCLCHK(g_rccl.GroupEnd());
    launch_sum_parts(e->stream, send + (size_t)e->rank * count, e->coll_stage, e->nranks, e->rank, count, recv);
    if (reopen) { NCCLCHK(g_rccl.GroupStart()); }
  } else {
    NCCLCHK(g_rccl.ReduceScatter(send, recv, count, ncclDouble, ncclSum, e->comm, e->stream));
  }
  CHK(timed_end(e, slot));
  return watch_mark(e, "reduce-scatter", e->stream);
}

// The inputs of a solve (order, wanted pairs, restart width, tolerance, method, policy ...) must be the same on every rank: everything the
// driver decides follows from them and from all-reduced numbers.  Verified with dav_ranks_agree - a collective whose size does not depend
// on the words - when they differ from what THIS engine verified last (the first solve on an engine always verifies): repeated solves
// with unchanged inputs add no collective.  Ranks of an SPMD caller change their inputs together, so all of them come here together; a
// rank that changes them alone is the error this check exists for - its peers are then in another collective, and the watchdog ends the run.
extern "C" int dav_agree_inputs(dav_handle_t e, const double* words, int nwords) {
  if (nwords <= 0 || nwords > 16 || !words) return fail("dav_agree_inputs: 1..16 words");
  if (e->nranks <= 1 || !has_comm(e)) return 0;
  if ((int)e->agreed_inputs.size() == nwords && std::equal(words, words + nwords, e->agreed_inputs.begin())) return 0;
  CHK(dav_ranks_agree(e, words, nwords));
  e->agreed_inputs.assign(words, words + nwords);
  return 0;
}

// What coll_path_trial decided and measured (engine_apply.hip).  selected: 0 program order, 1 direct exchange, 2 second stream, -1 = not
// decided yet (no wide block has been swept over a communicator of several ranks); ms[3] / valid[3]: maximum over the ranks of the trial
// block's time per way and whether every rank validated it (ms = 0: not tried - the environment forced a way, or one rank).
extern "C" int dav_comm_path(dav_handle_t e, int* selected, int* trial_ran, int* columns, double* ms3, int* valid3) {
  if (!e) return fail("dav_comm_path: null handle");
  const CollTrial& t = e->coll_trial;
  if (selected) *selected = e->coll_path == COLL_PATH_UNDECIDED ? -1 : (e->tune.sym_overlap ? COLL_PATH_SECOND_STREAM : e->tune.coll_direct ? COLL_PATH_DIRECT : COLL_PATH_PROGRAM_ORDER);
  if (trial_ran) *trial_ran = t.ran ? 1 : 0;
  if (columns) *columns = t.columns;
  for (int p = 0; p < 3; ++p) {
    if (ms3) ms3[p] = t.ms_max[p];
    if (valid3) valid3[p] = t.valid_all[p] ? 1 : 0;
  }
  return 0;
}

// The same check without a collective of its own: the words wait in the engine and ride on the NEXT all-reduced small result
// (the residual norms + Gram blocks of the Ritz phase: allreduce_with_agreement in engine_solver.hip).  One rank: only the
// iteration hint of the watchdog's message is kept.
extern "C" int dav_agree_next(dav_handle_t e, const double* words, int nwords) {
  if (nwords < 0 || nwords > 16 || (nwords > 0 && !words)) return fail("dav_agree_next: 0..16 words");
  if (nwords > 0) e->iter_hint = (long)words[0];
  e->agree_words.clear();
  if (e->nranks > 1 && has_comm(e)) e->agree_words.assign(words, words + nwords);
  return 0;
}

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
