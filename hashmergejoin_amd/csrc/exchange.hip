// Multi-GPU partition exchange behind the C ABI (include/hmj.h, "multi-GPU" section).
// One process per GPU.  The radix fan-out shards the join: every row has an OWNER rank, the ranks exchange
// their rows (all-to-all-v) and each joins what it owns -- partition p of the probe side only ever meets
// table p (hashjoin_bench.cc:92-96), so no further communication is needed.  The reference reaches all of its
// parallelism from the ctor (hashjoin.h:56-68 -> radix_hash.h:375-405, threads of one address space); this is
// the same fork-join across address spaces.
//
//   first radix pass on every rank (digit-major, stable)  ->  counts all-gather  ->  rounds of grouped send/recv, a round =
//   a range of digits  ->  every arrived round joined at once (remaining radix passes, build, probe)
// Rank g owns a contiguous range of the first pass's digits (hmj_exchange_digit_plan): the radix fan-out is the owner,
// nothing is partitioned twice.  Fallbacks: hash owner / key-range owner with a separate owner split (owner_digit).
//
// Transport: RCCL (ncclSend / ncclRecv per peer inside ncclGroupStart / ncclGroupEnd, on the communicator's own
// HIP stream; librccl is loaded with dlopen, so the library has no link-time dependency on it) or callbacks
// the host supplies (an existing communicator; the tests drive several ranks on one GPU through gloo that way).
// Everything else -- owner function, split, round plan, receive layout, overlap with the local join -- is the
// same code for both.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "hmj_ctx.h"

using hmj::u32;
typedef uint64_t u64;  // host-side arrays cross the C ABI as uint64_t
using namespace hmj_host;

namespace {

// ---- librccl, loaded on first use --------------------------------------------------------------------
struct RcclApi {
  void* dl = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                        // optional: absent -> a timed-out communicator is leaked
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;  // optional
  std::string error;
};

RcclApi g_rccl;
bool rccl_load(RcclApi& api) {
  // a process that already holds RCCL (PyTorch bundles one under the same SONAME) gets that copy back
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    api.dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (api.dl) break;
  }
  if (!api.dl) {
    const char* e = dlerror();
    api.error = e ? e : "librccl not found";
    return false;
  }
#define HMJ_SYM(field, name)                                          \
  api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.dl, name)); \
  if (!api.field) {                                                   \
    api.error = std::string("librccl lacks ") + name;                \
    api.dl = nullptr;                                                 \
    return false;                                                     \
  }
  HMJ_SYM(GetUniqueId, "ncclGetUniqueId")
  HMJ_SYM(CommInitRank, "ncclCommInitRank")
  HMJ_SYM(CommDestroy, "ncclCommDestroy")
  HMJ_SYM(GroupStart, "ncclGroupStart")
  HMJ_SYM(GroupEnd, "ncclGroupEnd")
  HMJ_SYM(Send, "ncclSend")
  HMJ_SYM(Recv, "ncclRecv")
  HMJ_SYM(AllGather, "ncclAllGather")
  HMJ_SYM(GetErrorString, "ncclGetErrorString")
#undef HMJ_SYM
  api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(api.dl, "ncclCommAbort"));
  api.CommGetAsyncError = reinterpret_cast<decltype(api.CommGetAsyncError)>(dlsym(api.dl, "ncclCommGetAsyncError"));
  return true;
}
// loaded once per process, whichever thread / context asks first (contexts on several threads may initialise
// their communicators at the same time)
RcclApi* rccl_api() {
  static std::once_flag once;
  std::call_once(once, [] { (void)rccl_load(g_rccl); });
  return g_rccl.dl ? &g_rccl : nullptr;
}

constexpr int kSampleKeys = 2048;  // per relation and rank, for the ordered mode's splitters

}  // namespace

struct hmj_comm {
  int n_ranks = 1, rank = 0;
  ncclComm_t nccl = nullptr;  // RCCL transport
  hmj_transport cb;           // callback transport
  bool has_cb = false;
  hipStream_t stream = nullptr;  // communication stream
  hipEvent_t ev_split = nullptr, ev_split_s = nullptr, ev_build = nullptr, ev_t0 = nullptr, ev_t1 = nullptr, ev_t2 = nullptr, ev_t3 = nullptr;
  std::vector<hipEvent_t> round_ev;
  DevBuf parted_r, parted_s, recv_r, recv_s, offs, gather_dev, sample_dev;
  HostBuf gather_host, offs_host;
  bool self_exchange = false;  // one rank: run the whole exchange path (tests) instead of the plain local join
  // Sizes every rank has already allocated for in an earlier successful step (per rank: shard rows and owned rows of
  // both relations).  All ranks hold the same values, so "does any rank have to grow a buffer in this step" is
  // decided identically everywhere -- only then is the extra status all-gather needed that keeps an out-of-memory
  // rank from leaving its peers blocked.
  std::vector<u64> peak_rows;  // [n_ranks][4]
  u64 max_msg_bytes = 1ull << 30;       // RCCL 2.26 truncates a single message of 2 GiB or more
  u64 target_round_bytes = 128ull << 20;  // probe side: several rounds, so the local pass A starts on arrived rows
  bool round_bytes_set = false;           // the host chose the round size itself (hmj_comm_set_message_bytes)
  int owner_path = HMJ_OWNER_DIGIT;       // non-ordered joins: digit ranges (default) or round 2's hash owner split (hmj_comm_set_owner_path, HMJ_EXCHANGE_OWNER=split)
  hmj_exchange_info info;
  // ---- the deadline of a step (hmj_comm_set_timeout_ms).  Every host wait of a step that depends on another rank is a
  // poll under this deadline; a watchdog thread aborts the RCCL communicator when the host itself is blocked inside an
  // RCCL call (ncclGroupEnd setting up a connection to a peer that never arrives).  `mu` orders the owner thread's RCCL
  // calls against the watchdog's ncclCommAbort: enqueue calls are made under it, calls that may block are made without
  // it (in_blocking_call), which is the one use ncclCommAbort from another thread is meant for.
  u64 timeout_ms = 120000;
  std::chrono::steady_clock::time_point deadline;
  bool step_open = false;
  std::mutex mu;
  std::condition_variable cv;
  std::thread watchdog;
  bool wd_stop = false, wd_armed = false, in_blocking_call = false;
  std::atomic<bool> broken{false};  // the communicator was aborted (timeout / asynchronous RCCL error): every later step fails
  int broken_code = HMJ_OK;
  // test hook HMJ_FAULT_STALL=<round>:<ms>: before that round's messages are posted the communication stream is held by a
  // kernel that spins for <ms> milliseconds and then exits -- what a peer that stops sending looks like from this rank
  // (RCCL's receive kernel spinning on data that does not come), with an end every wave reaches
  int fault_stall_round = -1;
  unsigned fault_stall_ms = 0;
};

namespace {

#define HIP_TRY(expr)                                           \
  do {                                                          \
    hipError_t _e = (expr);                                     \
    if (_e != hipSuccess) return fail(c, HMJ_E_HIP, #expr, _e); \
  } while (0)

int rccl_fail(hmj_ctx* c, const char* what, ncclResult_t r) {
  RcclApi* a = rccl_api();
  std::string m = std::string(what) + ": " + (a && a->GetErrorString ? a->GetErrorString(r) : "RCCL error");
  return fail(c, HMJ_E_RCCL, m.c_str());
}
#define RCCL_TRY(expr)                                  \
  do {                                                  \
    ncclResult_t _r = (expr);                           \
    if (_r != ncclSuccess) return rccl_fail(c, #expr, _r); \
  } while (0)

// ---- the deadline of a step -----------------------------------------------------------------------------
using Clock = std::chrono::steady_clock;

// Mark the communicator broken and, with abort_now, abort it: RCCL's kernels leave their wait loops, the communication
// stream drains, a host thread blocked inside an RCCL call returns.  The owner thread, which notices an expired deadline in
// one of its own polls, only marks (it is not blocked, and ncclCommAbort itself waits for the communicator's kernels: the
// step must return at its deadline, the abort happens at teardown); the watchdog aborts at once -- that is what unblocks
// an owner stuck inside RCCL.  m->mu held.
void abort_locked(hmj_comm* m, int code, bool abort_now) {
  if (!m->broken.load()) {
    m->broken_code = code;
    m->broken.store(true);
  }
  if (abort_now && m->nccl) {
    RcclApi* a = rccl_api();
    if (a && a->CommAbort) (void)a->CommAbort(m->nccl);
    m->nccl = nullptr;  // (a librccl without ncclCommAbort: the handle is leaked, never used again)
  }
}

// The watchdog: sleeps until a step is open, then until one second past its deadline (the owner thread's own polls
// notice an expired deadline first; the watchdog is for an owner that is blocked INSIDE an RCCL call).
void watchdog_main(hmj_comm* m) {
  std::unique_lock<std::mutex> lk(m->mu);
  while (!m->wd_stop) {
    if (!m->wd_armed) {
      m->cv.wait(lk);
      continue;
    }
    const auto fire = m->deadline + std::chrono::milliseconds(1000);
    if (m->cv.wait_until(lk, fire) == std::cv_status::timeout && m->wd_armed && !m->wd_stop && Clock::now() >= fire) {
      abort_locked(m, HMJ_E_TIMEOUT, true);
      m->wd_armed = false;
    }
  }
}

bool expired(const hmj_comm* m) { return m->timeout_ms && Clock::now() >= m->deadline; }

// The step cannot complete: abort the communicator and report why.
int give_up(hmj_ctx* c, const char* waiting_for) {
  hmj_comm* m = c->comm;
  int code;
  {
    std::lock_guard<std::mutex> lk(m->mu);
    abort_locked(m, HMJ_E_TIMEOUT, false);
    code = m->broken_code;
  }
  char msg[256];
  if (code == HMJ_E_TIMEOUT)
    std::snprintf(msg, sizeof(msg),
                  "rank %d of %d: no progress within %llu ms while waiting for %s -- a peer did not take part in the step; "
                  "the communicator is unusable", m->rank, m->n_ranks, (unsigned long long)m->timeout_ms, waiting_for);
  else
    std::snprintf(msg, sizeof(msg), "rank %d of %d: the communicator failed while waiting for %s and is unusable", m->rank,
                  m->n_ranks, waiting_for);
  return fail(c, code == HMJ_E_TIMEOUT ? HMJ_E_TIMEOUT : HMJ_E_RCCL, msg);
}

// A host wait that may depend on another rank: poll `query` (hipStreamQuery / hipEventQuery) under the step's deadline,
// looking at the communicator's asynchronous error state once a millisecond.  timeout 0: the plain blocking call.
template <class Query, class Block>
int wait_bounded(hmj_ctx* c, Query query, Block block, const char* what) {
  hmj_comm* m = c->comm;
  if (!m->timeout_ms) {
    const hipError_t e = block();
    return e == hipSuccess ? HMJ_OK : fail(c, HMJ_E_HIP, what, e);
  }
  auto next_look = Clock::now() + std::chrono::milliseconds(1);
  for (int spin = 0;; spin++) {
    const hipError_t e = query();
    if (e == hipSuccess) return HMJ_OK;
    if (e != hipErrorNotReady) return fail(c, HMJ_E_HIP, what, e);
    if (spin < 256) continue;  // the usual case: done within microseconds
    std::this_thread::sleep_for(std::chrono::microseconds(spin < 2048 ? 20 : 200));
    const auto now = Clock::now();
    if (now < next_look) continue;
    next_look = now + std::chrono::milliseconds(1);
    if (m->broken.load()) return give_up(c, what);  // (the watchdog was faster)
    if (m->nccl && !m->has_cb) {
      RcclApi* a = rccl_api();
      std::lock_guard<std::mutex> lk(m->mu);
      ncclResult_t st = ncclSuccess;
      if (m->nccl && a && a->CommGetAsyncError && a->CommGetAsyncError(m->nccl, &st) == ncclSuccess && st != ncclSuccess &&
          st != ncclInProgress) {
        abort_locked(m, HMJ_E_RCCL, false);
        return rccl_fail(c, "asynchronous RCCL error (ncclCommGetAsyncError)", st);
      }
    }
    if (expired(m)) return give_up(c, what);
  }
}
int wait_stream(hmj_ctx* c, hipStream_t s, const char* what) {
  return wait_bounded(c, [s] { return hipStreamQuery(s); }, [s] { return hipStreamSynchronize(s); }, what);
}
int wait_event(hmj_ctx* c, hipEvent_t ev, const char* what) {
  return wait_bounded(c, [ev] { return hipEventQuery(ev); }, [ev] { return hipEventSynchronize(ev); }, what);
}
// hmj_ctx::arrive_wait during a step: probe rows still on the links are waited for on the HOST, under the deadline
// (api.hip would queue a device-side wait, and the join's next synchronisation would block for ever on a lost peer)
int arrive_wait_hook(hmj_ctx* c, hipEvent_t ev) { return wait_event(c, ev, "an exchange round (probe rows)"); }

// Is the stream idle within `ms`?  (teardown: never block for ever on a stream a lost peer left busy)
bool drained_within(hipStream_t s, int ms) {
  const auto until = Clock::now() + std::chrono::milliseconds(ms);
  for (;;) {
    const hipError_t e = hipStreamQuery(s);
    if (e != hipErrorNotReady) return true;  // idle, or an error that no wait will cure
    if (Clock::now() >= until) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
}

struct StepScope {  // opens the step's deadline, arms the watchdog (RCCL transport); closes both on every exit
  hmj_ctx* c;
  explicit StepScope(hmj_ctx* ctx) : c(ctx) {
    hmj_comm* m = c->comm;
    std::lock_guard<std::mutex> lk(m->mu);
    m->deadline = Clock::now() + std::chrono::milliseconds(m->timeout_ms ? m->timeout_ms : 1000ull * 3600 * 24 * 365);
    m->step_open = true;
    if (m->timeout_ms && !m->has_cb && m->nccl) {
      if (!m->watchdog.joinable()) m->watchdog = std::thread(watchdog_main, m);
      m->wd_armed = true;
      m->cv.notify_all();
    }
    if (m->timeout_ms) c->arrive_wait = arrive_wait_hook;
  }
  ~StepScope() {
    hmj_comm* m = c->comm;
    c->arrive_wait = nullptr;
    if (!m) return;
    std::lock_guard<std::mutex> lk(m->mu);
    m->step_open = false;
    m->wd_armed = false;
    m->cv.notify_all();
  }
};

void comm_free(hmj_comm* m) {  // everything a (possibly half-built) communicator holds, except the RCCL handle
  if (m->watchdog.joinable()) {
    {
      std::lock_guard<std::mutex> lk(m->mu);
      m->wd_stop = true;
      m->cv.notify_all();
    }
    m->watchdog.join();
  }
  if (m->stream && !drained_within(m->stream, 5000)) {
    // a transfer that can never complete is still on the stream (no ncclCommAbort in this librccl, or the abort did not
    // take): freeing the buffers it uses would block or fault -- leak them with the communicator
    return;
  }
  DevBuf* devs[] = {&m->parted_r, &m->parted_s, &m->recv_r, &m->recv_s, &m->offs, &m->gather_dev, &m->sample_dev};
  for (DevBuf* b : devs) free_dev(*b);
  free_host(m->gather_host);
  free_host(m->offs_host);
  hipEvent_t evs[] = {m->ev_split, m->ev_split_s, m->ev_build, m->ev_t0, m->ev_t1, m->ev_t2, m->ev_t3};
  for (hipEvent_t e : evs)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : m->round_ev) (void)hipEventDestroy(e);
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m;
}

int comm_ensure(hmj_ctx* c) {
  if (c->comm) return HMJ_OK;
  hmj_comm* m = new hmj_comm();
  std::memset(&m->cb, 0, sizeof(m->cb));
  std::memset(&m->info, 0, sizeof(m->info));
  bool ok = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) == hipSuccess;
  hipEvent_t* evs[] = {&m->ev_split, &m->ev_split_s, &m->ev_build};
  for (hipEvent_t* e : evs) ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
  hipEvent_t* tevs[] = {&m->ev_t0, &m->ev_t1, &m->ev_t2, &m->ev_t3};
  for (hipEvent_t* e : tevs) ok = ok && hipEventCreate(e) == hipSuccess;
  if (!ok) {
    comm_free(m);
    return fail(c, HMJ_E_HIP, "communication stream / events");
  }
  if (const char* e = getenv("HMJ_EXCHANGE_OWNER")) {  // "split": round 2's owner-split path for all non-ordered joins
    if (std::strcmp(e, "split") == 0) m->owner_path = HMJ_OWNER_SPLIT;
  }
  if (const char* e = getenv("HMJ_COMM_TIMEOUT_MS")) m->timeout_ms = std::strtoull(e, nullptr, 10);  // (0: no deadline)
  if (const char* e = getenv("HMJ_FAULT_STALL")) {  // test hook, see hmj_comm
    int r = -1;
    unsigned ms = 0;
    if (std::sscanf(e, "%d:%u", &r, &ms) == 2 && ms <= 60000) {
      m->fault_stall_round = r;
      m->fault_stall_ms = ms;
    }
  }
  c->comm = m;
  return HMJ_OK;
}

// what a host-supplied collective returned: 0 ok, HMJ_E_TIMEOUT = it gave up waiting for a peer, anything else = failed
int transport_cb_status(hmj_ctx* c, int rc, const char* what) {
  if (rc == 0) return HMJ_OK;
  if (rc == HMJ_E_TIMEOUT) return give_up(c, what);
  return fail(c, HMJ_E_RCCL, what);
}

// (test hook) one wave that holds its stream for `ticks` of the constant-rate wall clock, then exits
__global__ void fault_stall_kernel(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
}

// ---- transport ----------------------------------------------------------------------------------------
// all ranks learn every rank's `count` values: recv[r * count + i] = rank r's send[i].  Host memory, blocking.
int transport_allgather(hmj_ctx* c, const u64* send, u64* recv, int count) {
  hmj_comm* m = c->comm;
  // (one rank: nothing to gather -- except in the forced self-exchange of the tests, which then also drives
  //  ncclAllGather, its device staging and its stream order on the one GPU a test box has)
  if (m->n_ranks == 1 && !(m->self_exchange && !m->has_cb && m->nccl)) {
    std::memcpy(recv, send, (size_t)count * 8);
    return HMJ_OK;
  }
  if (m->has_cb) return transport_cb_status(c, m->cb.allgather_u64(m->cb.user, send, recv, count), "transport: allgather_u64");
  RcclApi* a = rccl_api();
  int rc;
  const size_t bytes = (size_t)count * 8;
  if ((rc = ensure_dev(c, m->gather_dev, bytes * (m->n_ranks + 1))) != HMJ_OK) return rc;
  char* d = static_cast<char*>(m->gather_dev.p);
  HIP_TRY(hipMemcpyAsync(d, send, bytes, hipMemcpyHostToDevice, m->stream));
  {
    // (the call may block while RCCL connects to its peers: made without the lock, so that the watchdog can abort it)
    std::unique_lock<std::mutex> lk(m->mu);
    if (m->broken.load() || !m->nccl) {
      lk.unlock();
      return give_up(c, "the all-gather");
    }
    ncclComm_t h = m->nccl;
    lk.unlock();
    const ncclResult_t r = a->AllGather(d, d + bytes, (size_t)count, ncclUint64, h, m->stream);
    if (m->broken.load()) return give_up(c, "the all-gather");
    if (r != ncclSuccess) return rccl_fail(c, "ncclAllGather", r);
  }
  HIP_TRY(hipMemcpyAsync(recv, d + bytes, bytes * m->n_ranks, hipMemcpyDeviceToHost, m->stream));
  return wait_stream(c, m->stream, "the all-gather");
}

// one round of the all-to-all-v on device memory, queued on the communication stream
int transport_round(hmj_ctx* c, int round, const void* const* sp, const u64* sb, void* const* rp, const u64* rb) {
  hmj_comm* m = c->comm;
  const int G = m->n_ranks, me = m->rank;
  if (m->has_cb) return transport_cb_status(c, m->cb.alltoallv(m->cb.user, round, sp, sb, rp, rb, (void*)m->stream), "transport: alltoallv");
  RcclApi* a = rccl_api();
  // this rank's own bucket: a device copy when there are peers (it overlaps the link traffic); with a single
  // rank the copy goes through RCCL's send/recv pair as well, which keeps that path exercised on one GPU
  const bool self_rccl = G == 1;
  if (!self_rccl && sb[me]) HIP_TRY(hipMemcpyAsync(rp[me], sp[me], sb[me], hipMemcpyDeviceToDevice, m->stream));
  bool any = false;
  for (int g = 0; g < G; g++)
    if ((g != me || self_rccl) && (sb[g] || rb[g])) any = true;
  if (!any) return HMJ_OK;
  // No single ncclSend / ncclRecv above max_msg_bytes (RCCL 2.26 truncates a message of 2 GiB or more): a larger
  // message goes as consecutive pieces; both sides cut the same byte count the same way, and pieces between one
  // pair of ranks match in the order they are posted.
  const u64 lim = m->max_msg_bytes;
  if (m->fault_stall_round == round && m->fault_stall_ms) {  // (test hook)
    int khz = 100000;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device);
    hipLaunchKernelGGL(fault_stall_kernel, dim3(1), dim3(64), 0, m->stream, (unsigned long long)khz * m->fault_stall_ms);
  }
  // Sends and receives are only recorded inside the group (under the lock: the watchdog cannot abort the communicator
  // between two of them); ncclGroupEnd launches -- and may block while RCCL connects to a peer -- so it runs without.
  std::unique_lock<std::mutex> lk(m->mu);
  if (m->broken.load() || !m->nccl) {
    lk.unlock();
    return give_up(c, "an exchange round");
  }
  ncclResult_t r = a->GroupStart();
  for (int g = 0; g < G && r == ncclSuccess; g++) {
    if (g == me && !self_rccl) continue;
    for (u64 off = 0; off < sb[g] && r == ncclSuccess; off += lim)
      r = a->Send(static_cast<const char*>(sp[g]) + off, (size_t)std::min<u64>(lim, sb[g] - off), ncclUint8, g, m->nccl, m->stream);
    for (u64 off = 0; off < rb[g] && r == ncclSuccess; off += lim)
      r = a->Recv(static_cast<char*>(rp[g]) + off, (size_t)std::min<u64>(lim, rb[g] - off), ncclUint8, g, m->nccl, m->stream);
  }
  lk.unlock();
  const ncclResult_t re = a->GroupEnd();  // (always: a group that was started must be ended)
  if (m->broken.load()) return give_up(c, "an exchange round");
  if (r != ncclSuccess) return rccl_fail(c, "ncclSend / ncclRecv", r);
  if (re != ncclSuccess) return rccl_fail(c, "ncclGroupEnd", re);
  return HMJ_OK;
}

// ---- kernels ------------------------------------------------------------------------------------------

// K keys of a relation: one from each of K equal strata, at a pseudo-random position inside the stratum (a regular
// stride aliases with periodic key patterns: every 128th row of "every fourth key is in another cluster" never
// sees that cluster)
__global__ void sample_keys_kernel(const hmj::Tup* __restrict__ a, u64 n, u32 K, u64* __restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < K) {
    const u64 lo = (u64)i * n / K, hi = (u64)(i + 1) * n / K;
    const u64 span = hi > lo ? hi - lo : 1;
    u64 at = lo + hmj::mix64(0x9E3779B97F4A7C15ull * (i + 1)) % span;
    out[i] = a[at < n ? at : n - 1].key;
  }
}

// owner split of one relation: stable histogram / scan / scatter on owner_digit (radix.hip)
int owner_split(hmj_ctx* c, const void* in, u32 n, const hmj::OwnerFn& own, void* out, u64* offsets_dev) {
  const int G = (int)own.G;
  int bits = 0;
  while ((1 << bits) < G) bits++;
  if (bits == 0) bits = 1;
  if (n == 0) {
    HIP_TRY(hipMemsetAsync(offsets_dev, 0, ((size_t)(1u << bits) + 1) * 8, c->stream));
    return HMJ_OK;
  }
  u32 nblk, rpb;
  hmj::radix_pass_geometry(n, hmj::RP_TILE, &nblk, &rpb);
  int rc;
  if ((rc = ensure_dev(c, c->hist, (size_t)(1u << bits) * nblk * sizeof(u32))) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->totals, (size_t)hmj::RP_MAXD * sizeof(u32))) != HMJ_OK) return rc;
  HIP_TRY(hmj::launch_radix_hist(in, n, hmj::RP_TILE, 0, bits, (u32*)c->hist.p, nblk, rpb, c->stream, &own));
  HIP_TRY(hmj::launch_radix_rowscan((u32*)c->hist.p, nblk, bits, (u32*)c->totals.p, c->stream));
  HIP_TRY(hmj::launch_owner_scatter(in, out, n, bits, own, (const u32*)c->hist.p, (const u32*)c->totals.p, nblk, rpb,
                                    reinterpret_cast<hmj::u64*>(offsets_dev), c->stream));
  return HMJ_OK;
}

}  // namespace

// ---- host-only planning (exported: usable without a GPU, covered by the CPU tests) ----------------------
extern "C" uint32_t hmj_exchange_rounds(int n_ranks, const uint64_t* counts, uint64_t max_msg_rows) {
  if (n_ranks < 1 || !counts || max_msg_rows == 0) return 0;
  uint64_t biggest = 0;
  for (int i = 0; i < n_ranks * n_ranks; i++) biggest = std::max<uint64_t>(biggest, counts[i]);
  const uint64_t r = (biggest + max_msg_rows - 1) / max_msg_rows;
  return (uint32_t)(r < 1 ? 1 : r);
}

extern "C" int hmj_exchange_layout(int n_ranks, int rank, const uint64_t* counts, uint32_t n_rounds, int layout,
                                   uint64_t* send_off, uint64_t* send_rows, uint64_t* recv_off, uint64_t* recv_rows,
                                   uint64_t* round_end) {
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks || !counts || n_rounds == 0 || !send_off || !send_rows || !recv_off ||
      !recv_rows)
    return HMJ_E_ARG;
  const int G = n_ranks;
  // round r carries rows [r * c / R, (r + 1) * c / R) of every bucket of c rows
  auto lo = [&](uint64_t cnt, uint32_t r) { return (uint64_t)((unsigned __int128)cnt * r / n_rounds); };
  uint64_t s0 = 0;
  for (int g = 0; g < G; g++) {  // the split buffer is owner-major: bucket g starts at s0
    const uint64_t cnt = counts[(size_t)rank * G + g];
    for (uint32_t r = 0; r < n_rounds; r++) {
      send_off[(size_t)r * G + g] = s0 + lo(cnt, r);
      send_rows[(size_t)r * G + g] = lo(cnt, r + 1) - lo(cnt, r);
    }
    s0 += cnt;
  }
  if (layout == 0) {  // source-major: the rows of one source are contiguous, sources in rank order
    uint64_t r0 = 0;
    for (int g = 0; g < G; g++) {
      const uint64_t cnt = counts[(size_t)g * G + rank];
      for (uint32_t r = 0; r < n_rounds; r++) {
        recv_off[(size_t)r * G + g] = r0 + lo(cnt, r);
        recv_rows[(size_t)r * G + g] = lo(cnt, r + 1) - lo(cnt, r);
      }
      r0 += cnt;
    }
    if (round_end)
      for (uint32_t r = 0; r < n_rounds; r++) round_end[r] = (r + 1 == n_rounds) ? r0 : 0;  // complete only at the end
  } else {  // round-major: everything a round delivers is contiguous
    uint64_t r0 = 0;
    for (uint32_t r = 0; r < n_rounds; r++) {
      for (int g = 0; g < G; g++) {
        const uint64_t cnt = counts[(size_t)g * G + rank];
        recv_off[(size_t)r * G + g] = r0;
        recv_rows[(size_t)r * G + g] = lo(cnt, r + 1) - lo(cnt, r);
        r0 += recv_rows[(size_t)r * G + g];
      }
      if (round_end) round_end[r] = r0;
    }
  }
  return HMJ_OK;
}


// ---- digit-range owners (the radix fan-out as the owner): host arithmetic, exported ---------------------------
extern "C" int hmj_exchange_digit_plan(int n_ranks, const uint64_t* keys, uint64_t n, uint32_t n_rounds, hmj_digit_plan* out) {
  if (n_ranks < 1 || n_ranks > HMJ_MAX_RANKS || !out || (n && !keys) || n_rounds < 1 || n_rounds > HMJ_MAX_ROUNDS) return HMJ_E_ARG;
  std::memset(out, 0, sizeof(*out));
  const int G = n_ranks;
  out->n_rounds = n_rounds;
  // the bits all sampled keys share are no use as a digit (dense integer keys: SURVEY.md D5); the digit is the
  // top <= 8 bits right under them -- the reference's top-bits routing (radix_hash.h:369) under a prefix
  uint64_t diff = 0;
  for (uint64_t i = 1; i < n; i++) diff |= keys[i] ^ keys[0];
  const int prefix = diff ? __builtin_clzll(diff) : 64;
  int ba = 64 - prefix < 8 ? 64 - prefix : 8;
  if (n == 0) ba = 8;  // nothing sampled (empty relations): any window does
  const int low = n == 0 ? 56 : 64 - prefix - ba;
  out->digit_bits = ba;
  out->digit_low = low;
  const uint32_t D = 1u << (ba > 0 ? ba : 0);
  if (ba < 1 || D < 2u * (uint32_t)G) {  // fewer than two digits per rank: no balance to be had
    out->digit_bits = ba < 0 ? 0 : ba;
    for (int g = 0; g <= G; g++) out->owner_first[g] = (uint32_t)((uint64_t)D * g / G);
    for (int g = 0; g < G; g++)
      for (uint32_t r = 0; r <= n_rounds; r++) out->round_first[g][r] = r == 0 ? out->owner_first[g] : out->owner_first[g + 1];
    out->usable = 0;
    return HMJ_OK;
  }
  std::vector<uint64_t> cum(D + 1, 0);
  for (uint64_t i = 0; i < n; i++) cum[((keys[i] >> low) & (D - 1)) + 1]++;
  for (uint32_t d = 0; d < D; d++) cum[d + 1] += cum[d];
  const bool tiny = n < 64ull * G;  // too few samples to say anything: equal digit ranges
  // boundary j of `parts` over digits [a, b): where the cumulative sample count is closest to its share
  auto cut = [&](uint32_t a, uint32_t b, uint32_t parts, uint32_t j, uint32_t not_before) -> uint32_t {
    if (j == 0) return a;
    if (j >= parts) return b;
    if (tiny || cum[b] == cum[a]) {
      uint32_t d = a + (uint32_t)((uint64_t)(b - a) * j / parts);
      return d < not_before ? not_before : d;
    }
    const unsigned __int128 target = (unsigned __int128)(cum[b] - cum[a]) * j;  // compare (cum[d] - cum[a]) * parts with it
    uint32_t best = not_before;
    unsigned __int128 best_err = ~(unsigned __int128)0;
    for (uint32_t d = not_before; d <= b; d++) {
      const unsigned __int128 v = (unsigned __int128)(cum[d] - cum[a]) * parts;
      const unsigned __int128 err = v > target ? v - target : target - v;
      if (err < best_err) {
        best_err = err;
        best = d;
      }
    }
    return best;
  };
  out->owner_first[0] = 0;
  for (int g = 1; g <= G; g++) out->owner_first[g] = cut(0, D, (uint32_t)G, (uint32_t)g, out->owner_first[g - 1]);
  uint64_t biggest = 0;
  for (int g = 0; g < G; g++) biggest = std::max<uint64_t>(biggest, cum[out->owner_first[g + 1]] - cum[out->owner_first[g]]);
  out->max_share = n ? (float)((double)biggest * G / (double)n) : 1.0f;
  out->usable = (tiny || out->max_share <= 1.3f) ? 1 : 0;
  for (int g = 0; g < G; g++) {
    const uint32_t a = out->owner_first[g], b = out->owner_first[g + 1];
    out->round_first[g][0] = a;
    for (uint32_t r = 1; r <= n_rounds; r++) out->round_first[g][r] = cut(a, b, n_rounds, r, out->round_first[g][r - 1]);
  }
  return HMJ_OK;
}

extern "C" int hmj_exchange_digit_layout(int n_ranks, int rank, const hmj_digit_plan* plan, const uint64_t* counts,
                                         uint64_t* send_off, uint64_t* send_rows, uint64_t* recv_off, uint64_t* recv_rows,
                                         uint64_t* round_off) {
  if (n_ranks < 1 || n_ranks > HMJ_MAX_RANKS || rank < 0 || rank >= n_ranks || !plan || !counts || !send_off || !send_rows ||
      !recv_off || !recv_rows || !round_off || plan->n_rounds < 1 || plan->n_rounds > HMJ_MAX_ROUNDS || plan->digit_bits < 0 ||
      plan->digit_bits > 8)
    return HMJ_E_ARG;
  const int G = n_ranks;
  const uint32_t D = 1u << plan->digit_bits, R = plan->n_rounds;
  std::vector<uint64_t> offs(D + 1, 0);  // this rank's digit-major buffer
  for (uint32_t d = 0; d < D; d++) offs[d + 1] = offs[d] + counts[(size_t)rank * D + d];
  for (uint32_t r = 0; r < R; r++)
    for (int g = 0; g < G; g++) {
      const uint32_t d0 = plan->round_first[g][r], d1 = plan->round_first[g][r + 1];
      if (d0 > d1 || d1 > D) return HMJ_E_ARG;
      send_off[(size_t)r * G + g] = offs[d0];
      send_rows[(size_t)r * G + g] = offs[d1] - offs[d0];
    }
  uint64_t pos = 0;
  for (uint32_t r = 0; r < R; r++) {
    round_off[r] = pos;
    const uint32_t d0 = plan->round_first[rank][r], d1 = plan->round_first[rank][r + 1];
    for (int s = 0; s < G; s++) {
      uint64_t rows = 0;
      for (uint32_t d = d0; d < d1; d++) rows += counts[(size_t)s * D + d];
      recv_off[(size_t)r * G + s] = pos;
      recv_rows[(size_t)r * G + s] = rows;
      pos += rows;
    }
  }
  round_off[R] = pos;
  return HMJ_OK;
}

namespace {

// Exchange one relation: rounds of alltoallv queued on the communication stream.  ev_after_round (optional):
// events recorded after each round (n_rounds of them).  Returns rows received.
// soft_err: a failed event record does not stop the queuing (the peers' receives are matched by this rank's rounds);
// it becomes the rank's pending error, which the caller carries to the final reduction
int exchange_relation(hmj_ctx* c, const void* parted, const u64* counts_matrix, u32 n_rounds, int layout, void* recv,
                      hipEvent_t* ev_after_round, u64* round_end, int* soft_err) {
  hmj_comm* m = c->comm;
  const int G = m->n_ranks;
  std::vector<u64> so((size_t)n_rounds * G), sr(so.size()), ro(so.size()), rr(so.size());
  int rc = hmj_exchange_layout(G, m->rank, counts_matrix, n_rounds, layout, so.data(), sr.data(), ro.data(), rr.data(), round_end);
  if (rc != HMJ_OK) return fail(c, rc, "exchange layout");
  std::vector<const void*> sp(G);
  std::vector<void*> rp(G);
  std::vector<u64> sb(G), rb(G);
  for (u32 r = 0; r < n_rounds; r++) {
    for (int g = 0; g < G; g++) {
      sp[g] = static_cast<const char*>(parted) + so[(size_t)r * G + g] * 16;
      rp[g] = static_cast<char*>(recv) + ro[(size_t)r * G + g] * 16;
      sb[g] = sr[(size_t)r * G + g] * 16;
      rb[g] = rr[(size_t)r * G + g] * 16;
    }
    if ((rc = transport_round(c, (int)r, sp.data(), sb.data(), rp.data(), rb.data())) != HMJ_OK) return rc;
    if (ev_after_round) {
      const hipError_t e = hipEventRecord(ev_after_round[r], m->stream);
      if (e != hipSuccess && *soft_err == HMJ_OK) *soft_err = fail(c, HMJ_E_HIP, "hipEventRecord", e);
    }
  }
  return HMJ_OK;
}

}  // namespace

namespace hmj_host {
void comm_destroy(hmj_ctx* c) {
  hmj_comm* m = c->comm;
  if (!m) return;
  c->arrive_wait = nullptr;
  {
    // ncclCommDestroy waits for everything queued on the communicator: only when the communication stream drains;
    // otherwise (a peer was lost and nobody noticed yet) the communicator is aborted
    const bool idle = !m->stream || drained_within(m->stream, 5000);
    std::lock_guard<std::mutex> lk(m->mu);
    if (m->nccl) {
      RcclApi* a = rccl_api();
      if (idle && a && !m->broken.load())
        (void)a->CommDestroy(m->nccl);
      else if (a && a->CommAbort)
        (void)a->CommAbort(m->nccl);
      m->nccl = nullptr;
    }
  }
  comm_free(m);
  c->comm = nullptr;
}
}  // namespace hmj_host

extern "C" {

int hmj_comm_unique_id(void* id128) {
  if (!id128) return HMJ_E_ARG;
  RcclApi* a = rccl_api();
  if (!a) return HMJ_E_RCCL;
  ncclUniqueId id;
  if (a->GetUniqueId(&id) != ncclSuccess) return HMJ_E_RCCL;
  static_assert(sizeof(id) == HMJ_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
  std::memcpy(id128, &id, sizeof(id));
  return HMJ_OK;
}

int hmj_comm_init_rank(hmj_ctx* c, int n_ranks, int rank, const void* id128) {
  if (!c || n_ranks < 1 || rank < 0 || rank >= n_ranks || !id128) return c ? fail(c, HMJ_E_ARG, "hmj_comm_init_rank") : HMJ_E_ARG;
  if (n_ranks > hmj::kMaxRanks) return fail(c, HMJ_E_UNSUPPORTED, "more than 16 ranks in one exchange");
  HIP_TRY(hipSetDevice(c->device));
  RcclApi* a = rccl_api();
  if (!a) return fail(c, HMJ_E_RCCL, "librccl could not be loaded");
  int rc = comm_ensure(c);
  if (rc != HMJ_OK) return rc;
  hmj_comm* m = c->comm;
  if (m->nccl) {
    (void)a->CommDestroy(m->nccl);
    m->nccl = nullptr;
  }
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  RCCL_TRY(a->CommInitRank(&m->nccl, n_ranks, id, rank));
  m->n_ranks = n_ranks;
  m->rank = rank;
  m->has_cb = false;
  return HMJ_OK;
}

int hmj_comm_set_transport(hmj_ctx* c, const hmj_transport* t) {
  if (!c || !t || t->n_ranks < 1 || t->rank < 0 || t->rank >= t->n_ranks || !t->allgather_u64 || !t->alltoallv)
    return c ? fail(c, HMJ_E_ARG, "hmj_comm_set_transport") : HMJ_E_ARG;
  if (t->n_ranks > hmj::kMaxRanks) return fail(c, HMJ_E_UNSUPPORTED, "more than 16 ranks in one exchange");
  HIP_TRY(hipSetDevice(c->device));
  int rc = comm_ensure(c);
  if (rc != HMJ_OK) return rc;
  hmj_comm* m = c->comm;
  m->cb = *t;
  m->has_cb = true;
  m->n_ranks = t->n_ranks;
  m->rank = t->rank;
  return HMJ_OK;
}

int hmj_comm_destroy(hmj_ctx* c) {
  if (!c) return HMJ_E_ARG;
  (void)hipSetDevice(c->device);
  comm_destroy(c);
  return HMJ_OK;
}

int hmj_comm_set_message_bytes(hmj_ctx* c, uint64_t max_message_bytes, uint64_t probe_round_bytes) {
  if (!c || !c->comm) return c ? fail(c, HMJ_E_ARG, "no communicator") : HMJ_E_ARG;
  if (max_message_bytes) {
    if (max_message_bytes < 16 || max_message_bytes > (1ull << 30)) return fail(c, HMJ_E_ARG, "max_message_bytes must be in [16, 2^30]");
    c->comm->max_msg_bytes = max_message_bytes;
  }
  if (probe_round_bytes) {
    c->comm->target_round_bytes = probe_round_bytes < 16 ? 16 : probe_round_bytes;
    c->comm->round_bytes_set = true;
  }
  return HMJ_OK;
}

int hmj_comm_set_timeout_ms(hmj_ctx* c, uint64_t timeout_ms) {
  if (!c || !c->comm) return c ? fail(c, HMJ_E_ARG, "no communicator") : HMJ_E_ARG;
  std::lock_guard<std::mutex> lk(c->comm->mu);
  if (c->comm->step_open) return fail(c, HMJ_E_ARG, "hmj_comm_set_timeout_ms during a step");
  c->comm->timeout_ms = timeout_ms;
  return HMJ_OK;
}

int hmj_comm_get_timeout_ms(hmj_ctx* c, uint64_t* timeout_ms) {
  if (!c || !c->comm || !timeout_ms) return HMJ_E_ARG;
  *timeout_ms = c->comm->timeout_ms;
  return HMJ_OK;
}

int hmj_comm_set_owner_path(hmj_ctx* c, int owner_path) {
  if (!c || !c->comm) return c ? fail(c, HMJ_E_ARG, "no communicator") : HMJ_E_ARG;
  if (owner_path != HMJ_OWNER_DIGIT && owner_path != HMJ_OWNER_SPLIT) return fail(c, HMJ_E_ARG, "owner_path");
  c->comm->owner_path = owner_path;
  return HMJ_OK;
}

int hmj_comm_set_self_exchange(hmj_ctx* c, int on) {
  if (!c || !c->comm) return c ? fail(c, HMJ_E_ARG, "no communicator") : HMJ_E_ARG;
  c->comm->self_exchange = on != 0;
  return HMJ_OK;
}

int hmj_owner_split_u64_device(hmj_ctx* c, const void* in_aos_dev, uint64_t n, int n_ranks, const uint64_t* splitters,
                               void* out_aos_dev, uint64_t* offsets_dev) {
  if (!c) return HMJ_E_ARG;
  if (n_ranks < 2 || n_ranks > hmj::kMaxRanks) return fail(c, HMJ_E_ARG, "n_ranks must be in 2..16");
  if (n > 0xFFFFFFFFull || (n && (!in_aos_dev || !out_aos_dev)) || !offsets_dev) return fail(c, HMJ_E_ARG, "hmj_owner_split_u64_device");
  HIP_TRY(hipSetDevice(c->device));
  hmj::OwnerFn own;
  std::memset(&own, 0, sizeof(own));
  own.G = (u32)n_ranks;
  own.mode = splitters ? 2u : 1u;
  if (splitters)
    for (int i = 0; i + 1 < n_ranks; i++) own.spl[i] = splitters[i];
  int rc = owner_split(c, in_aos_dev, (u32)n, own, out_aos_dev, offsets_dev);
  if (rc != HMJ_OK) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return HMJ_OK;
}

int hmj_last_exchange_info(hmj_ctx* c, hmj_exchange_info* out) {
  if (!c || !out || !c->comm) return HMJ_E_ARG;
  *out = c->comm->info;
  return HMJ_OK;
}


}  // extern "C"

// ---- the distributed join ----------------------------------------------------------------------------------
namespace {

// Message of the first all-gather of a step: this rank's status, sample sizes, shard sizes and key sample.
constexpr int kSampleWords = 2 * kSampleKeys + 5;  // [0] status [1] kr [2] ks [3] n_build [4] n_probe [5..] keys

struct StepSample {
  std::vector<u64> keys;    // pooled over all ranks and both relations
  std::vector<u64> nb, np;  // shard rows of every rank
};

// A collective must be left by all ranks together.  `mine`: this rank's pending error (HMJ_OK if none); `flags[g]`:
// what rank g reported.  Returns HMJ_OK when nobody failed, this rank's own code when it failed, HMJ_E_PEER otherwise.
int settle(hmj_ctx* c, int mine, const std::vector<u64>& flags) {
  int bad = -1;
  for (size_t g = 0; g < flags.size(); g++)
    if (flags[g] && bad < 0) bad = (int)g;
  if (bad < 0 && mine == HMJ_OK) return HMJ_OK;
  if (mine != HMJ_OK) return mine;  // (its message was set where it happened)
  char msg[96];
  std::snprintf(msg, sizeof(msg), "rank %d failed in this collective call (status %lld)", bad, (long long)(int64_t)flags[bad]);
  return fail(c, HMJ_E_PEER, msg);
}

// One word per rank: did anybody fail since the last collective?  (Only called when every rank calls it.)
int status_round(hmj_ctx* c, int mine) {
  hmj_comm* m = c->comm;
  const u64 w = mine == HMJ_OK ? 0ull : (u64)(int64_t)mine;
  std::vector<u64> all((size_t)m->n_ranks, 0);
  int rc = transport_allgather(c, &w, all.data(), 1);
  if (rc != HMJ_OK) return rc;
  return settle(c, mine, all);
}

// every rank's evenly spaced key sample of both shards + its shard sizes: one all-gather
int gather_samples(hmj_ctx* c, const void* R, u64 nb, const void* S, u64 np, int* err, StepSample* out) {
  hmj_comm* m = c->comm;
  const int G = m->n_ranks, K = kSampleKeys, W = kSampleWords;
  std::vector<u64> mine((size_t)W, 0), all((size_t)W * G, 0);
  const u32 kr = (u32)std::min<u64>(K, nb), ks = (u32)std::min<u64>(K, np);
  if (*err == HMJ_OK) {
    int rc = ensure_dev(c, m->sample_dev, (size_t)2 * K * 8);
    if (rc == HMJ_OK) {
      u64* sd = static_cast<u64*>(m->sample_dev.p);
      if (kr) hipLaunchKernelGGL(sample_keys_kernel, dim3((kr + 255) / 256), dim3(256), 0, c->stream, static_cast<const hmj::Tup*>(R), nb, kr, sd);
      if (ks) hipLaunchKernelGGL(sample_keys_kernel, dim3((ks + 255) / 256), dim3(256), 0, c->stream, static_cast<const hmj::Tup*>(S), np, ks, sd + K);
      hipError_t e = hipGetLastError();
      if (e == hipSuccess) e = hipMemcpyAsync(mine.data() + 5, sd, (size_t)2 * K * 8, hipMemcpyDeviceToHost, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      if (e != hipSuccess) rc = fail(c, HMJ_E_HIP, "key sample", e);
    }
    if (rc != HMJ_OK) *err = rc;
  }
  mine[0] = *err == HMJ_OK ? 0ull : (u64)(int64_t)*err;
  mine[1] = kr;
  mine[2] = ks;
  mine[3] = nb;
  mine[4] = np;
  int rc = transport_allgather(c, mine.data(), all.data(), W);
  if (rc != HMJ_OK) return rc;
  std::vector<u64> st((size_t)G);
  out->keys.clear();
  out->nb.assign((size_t)G, 0);
  out->np.assign((size_t)G, 0);
  for (int g = 0; g < G; g++) {
    const u64* p = &all[(size_t)g * W];
    st[g] = p[0];
    out->nb[g] = p[3];
    out->np[g] = p[4];
    if (!p[0]) {
      out->keys.insert(out->keys.end(), p + 5, p + 5 + std::min<u64>(p[1], K));
      out->keys.insert(out->keys.end(), p + 5 + K, p + 5 + K + std::min<u64>(p[2], K));
    }
  }
  return settle(c, *err, st);
}

void add_result(hmj_result* acc, const hmj_result& r) {
  acc->n_matches += r.n_matches;
  acc->sum_r += r.sum_r;
  acc->sum_s += r.sum_s;
  acc->xor_fold ^= r.xor_fold;
  acc->mix_sum += r.mix_sum;
  acc->sum_probe_all += r.sum_probe_all;
}

float kernel_ms(const hmj_timing& t) {
  return t.ms_hist + t.ms_scan + t.ms_scatter + t.ms_offsets + t.ms_probe_count + t.ms_out_scan + t.ms_probe_write + t.ms_order;
}

// Does any rank have to grow one of its receive buffers in this step?  Decided from numbers all ranks hold.
bool any_rank_grows(hmj_comm* m, int which, const std::vector<u64>& need) {
  const int G = m->n_ranks;
  if (m->peak_rows.size() != (size_t)G * 4) m->peak_rows.assign((size_t)G * 4, 0);
  bool grows = false;
  for (int g = 0; g < G; g++)
    if (need[g] > m->peak_rows[(size_t)g * 4 + which]) grows = true;
  return grows;
}
void note_peaks(hmj_comm* m, int which, const std::vector<u64>& need) {
  for (int g = 0; g < m->n_ranks; g++)
    m->peak_rows[(size_t)g * 4 + which] = std::max(m->peak_rows[(size_t)g * 4 + which], need[g]);
}

int ensure_round_events(hmj_ctx* c, size_t n) {
  hmj_comm* m = c->comm;
  while (m->round_ev.size() < n) {
    hipEvent_t e;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    m->round_ev.push_back(e);
  }
  return HMJ_OK;
}

// The reduction over all ranks that ends every step; `mine`: this rank's pending error.  All ranks return together.
int final_reduction(hmj_ctx* c, int mine, const hmj_result* local_out, hmj_result* global_out) {
  hmj_comm* m = c->comm;
  const int G = m->n_ranks;
  u64 w[7] = {mine == HMJ_OK ? 0ull : (u64)(int64_t)mine, local_out->n_matches, local_out->sum_r, local_out->sum_s,
              local_out->xor_fold, local_out->mix_sum, local_out->sum_probe_all};
  if (mine != HMJ_OK) std::memset(w + 1, 0, 6 * sizeof(u64));
  std::vector<u64> all(7 * (size_t)G), st((size_t)G);
  int rc = transport_allgather(c, w, all.data(), 7);
  if (rc != HMJ_OK) return mine != HMJ_OK ? mine : rc;
  for (int g = 0; g < G; g++) st[g] = all[7 * (size_t)g];
  if ((rc = settle(c, mine, st)) != HMJ_OK) return rc;
  if (global_out) {
    std::memset(global_out, 0, sizeof(*global_out));
    for (int g = 0; g < G; g++) {
      const u64* p = &all[7 * (size_t)g + 1];
      global_out->n_matches += p[0];
      global_out->sum_r += p[1];
      global_out->sum_s += p[2];
      global_out->xor_fold ^= p[3];
      global_out->mix_sum += p[4];
      global_out->sum_probe_all += p[5];
    }
  }
  return HMJ_OK;
}

float ms_since(std::chrono::steady_clock::time_point t) {
  return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t).count();
}

// ---- digit path: the first radix pass is the owner --------------------------------------------------------
// compute stream: pre-pass R | pre-pass S | join of round 0 | join of round 1 | ...
// comm stream   :            | R rounds ..........| S round 0 | S round 1 | ...      (round = a range of digits)
int exchange_digit_path(hmj_ctx* c, const void* R, u64 nb, const void* S, u64 np, u32 flags, const hmj_digit_plan& plan,
                        hmj_result* local_out, int* err) {
  hmj_comm* m = c->comm;
  const int G = m->n_ranks, me = m->rank;
  const int ba = plan.digit_bits, low = plan.digit_low;
  const u32 D = 1u << ba, NR = plan.n_rounds;
  const bool materialize = (flags & HMJ_MATERIALIZE) != 0;
  m->info.owner_mode = 3;
  m->info.digit_bits = ba;
  m->info.digit_low = low;
  m->info.sample_max_share = plan.max_share;
  m->info.rounds_build = m->info.rounds_probe = NR;
  hmj_timing acc;
  std::memset(&acc, 0, sizeof(acc));
  const auto t_split = std::chrono::steady_clock::now();

  // ---- 1. the first radix pass of both shards (digit-major, dense, stable) + bucket starts to the host
  const size_t noff = (size_t)D + 1;
  u64* oh = nullptr;
  if (*err == HMJ_OK) {
    int rc = ensure_dev(c, m->parted_r, (size_t)nb * 16 + 16);
    if (rc == HMJ_OK) rc = ensure_dev(c, m->parted_s, (size_t)np * 16 + 16);
    if (rc == HMJ_OK) rc = ensure_dev(c, m->offs, 2 * noff * 8);
    if (rc == HMJ_OK) rc = ensure_host(c, m->offs_host, 2 * noff * 8);
    if (rc == HMJ_OK) rc = ensure_round_events(c, NR);
    if (rc == HMJ_OK) {
      oh = static_cast<u64*>(m->offs_host.p);
      u64* od = static_cast<u64*>(m->offs.p);
      spans_reset(c);
      struct { const void* in; void* out; u64 n; u64* o_dev; u64* o_host; hipEvent_t ev; int rel; } side[2] = {
          {R, m->parted_r.p, nb, od, oh, m->ev_split, 0}, {S, m->parted_s.p, np, od + noff, oh + noff, m->ev_split_s, 1}};
      for (auto& sd : side) {
        if (rc != HMJ_OK) break;
        hipError_t e = hipSuccess;
        if (sd.n == 0)
          e = hipMemsetAsync(sd.o_dev, 0, noff * 8, c->stream);
        else
          rc = radix_pass(c, sd.in, sd.out, (u32)sd.n, low, ba, sd.rel, reinterpret_cast<hmj::u64*>(sd.o_dev), 0);
        if (rc == HMJ_OK && e == hipSuccess) e = hipMemcpyAsync(sd.o_host, sd.o_dev, noff * 8, hipMemcpyDeviceToHost, c->stream);
        if (rc == HMJ_OK && e == hipSuccess) e = hipEventRecord(sd.ev, c->stream);
        if (rc == HMJ_OK && e != hipSuccess) rc = fail(c, HMJ_E_HIP, "digit pre-pass", e);
      }
    }
    if (rc != HMJ_OK) *err = rc;
  }

  // ---- 2. per relation: counts all-gather -> layout -> receive buffer -> rounds queued on the communication stream.
  // The build side's rounds are on the links while the probe side is still being partitioned.
  std::vector<u64> so((size_t)NR * G), sr(so.size()), ro(so.size()), rr(so.size());
  std::vector<u64> round_off_r(NR + 1, 0), round_off_s(NR + 1, 0);
  std::vector<const void*> sp(G);
  std::vector<void*> rp(G);
  std::vector<u64> sb(G), rb(G);
  for (int rel = 0; rel < 2; rel++) {
    std::vector<u64> msg((size_t)D + 1, 0), all(((size_t)D + 1) * G, 0), st((size_t)G), need((size_t)G, 0);
    if (*err == HMJ_OK) {
      hipError_t e = hipEventSynchronize(rel == 0 ? m->ev_split : m->ev_split_s);
      if (e != hipSuccess) *err = fail(c, HMJ_E_HIP, "digit pre-pass", e);
    }
    if (*err == HMJ_OK) {
      const u64* o = oh + (size_t)rel * noff;
      for (u32 d = 0; d < D; d++) msg[1 + d] = o[d + 1] - o[d];
    }
    // the communication stream waits for this relation's pre-pass -- queued BEFORE the all-gather, so that a failure
    // here travels in its status word and every rank leaves together (local failures after the first
    // all-gather must not return on this rank alone).  Order on the stream: [wait R] R rounds [wait S] S rounds.
    if (*err == HMJ_OK) {
      hipError_t e = hipStreamWaitEvent(m->stream, rel == 0 ? m->ev_split : m->ev_split_s, 0);
      if (e != hipSuccess) *err = fail(c, HMJ_E_HIP, "hipStreamWaitEvent", e);
    }
    if (rel == 1 && c->profiling && *err == HMJ_OK) {  // both pre-passes are complete: their spans
      spans_collect(c);
      add_timing(&acc, c->timing);
      m->info.ms_split = ms_since(t_split);
    }
    msg[0] = *err == HMJ_OK ? 0ull : (u64)(int64_t)*err;
    int rc = transport_allgather(c, msg.data(), all.data(), (int)D + 1);
    if (rc != HMJ_OK) return rc;
    std::vector<u64> counts((size_t)G * D);
    for (int g = 0; g < G; g++) {
      st[g] = all[(size_t)g * (D + 1)];
      std::memcpy(&counts[(size_t)g * D], &all[(size_t)g * (D + 1) + 1], (size_t)D * 8);
    }
    if ((rc = settle(c, *err, st)) != HMJ_OK) return rc;
    // rows every rank will own: the 2^32-1 limit of one local join is checked for ALL ranks by ALL ranks
    for (int g = 0; g < G; g++)
      for (int s = 0; s < G; s++)
        for (u32 d = plan.owner_first[g]; d < plan.owner_first[g + 1]; d++) need[g] += counts[(size_t)s * D + d];
    for (int g = 0; g < G; g++)
      if (need[g] > 0xFFFFFFFFull) {
        char msgb[96];
        std::snprintf(msgb, sizeof(msgb), "rank %d would own more than 2^32-1 %s rows", g, rel == 0 ? "build" : "probe");
        return fail(c, HMJ_E_UNSUPPORTED, msgb);
      }
    std::vector<u64>& round_off = rel == 0 ? round_off_r : round_off_s;
    rc = hmj_exchange_digit_layout(G, me, &plan, counts.data(), so.data(), sr.data(), ro.data(), rr.data(), round_off.data());
    if (rc != HMJ_OK) return fail(c, rc, "digit layout");  // (same inputs on every rank: all fail alike)
    (rel == 0 ? m->info.recv_build : m->info.recv_probe) = need[me];
    DevBuf& recv = rel == 0 ? m->recv_r : m->recv_s;
    const bool grows = any_rank_grows(m, 2 + rel, need);
    int mine = ensure_dev(c, recv, (size_t)need[me] * 16 + 16);
    if (grows) {
      if ((rc = status_round(c, mine)) != HMJ_OK) return rc;
      note_peaks(m, 2 + rel, need);
    } else if (mine != HMJ_OK) {
      return mine;  // cannot happen: the buffer is at least as large as in an earlier step
    }
    // queue the rounds.  From here on every rank queues ALL its rounds whatever happens to it locally (its peers'
    // receives are matched by them); only a failure of the transport itself returns at once.  A failed event record
    // becomes this rank's pending error: its local joins are skipped and the final reduction tells everybody.
    auto note = [&](hipError_t e, const char* what) {
      if (e != hipSuccess && *err == HMJ_OK) *err = fail(c, HMJ_E_HIP, what, e);
    };
    note(hipEventRecord(rel == 0 ? m->ev_t0 : m->ev_t1, m->stream), "hipEventRecord");
    const char* parted = static_cast<const char*>(rel == 0 ? m->parted_r.p : m->parted_s.p);
    char* rbase = static_cast<char*>(recv.p);
    const bool in_place = G == 1 && !m->self_exchange;  // (not reached today: one rank takes the plain join)
    for (u32 r = 0; r < NR && !in_place; r++) {
      for (int g = 0; g < G; g++) {
        sp[g] = parted + so[(size_t)r * G + g] * 16;
        rp[g] = rbase + ro[(size_t)r * G + g] * 16;
        sb[g] = sr[(size_t)r * G + g] * 16;
        rb[g] = rr[(size_t)r * G + g] * 16;
      }
      if ((rc = transport_round(c, (int)(rel * NR + r), sp.data(), sb.data(), rp.data(), rb.data())) != HMJ_OK) return rc;
      if (rel == 1) note(hipEventRecord(m->round_ev[r], m->stream), "hipEventRecord");
    }
    if (rel == 0) {
      note(hipEventRecord(m->ev_t3, m->stream), "hipEventRecord");
      note(hipEventRecord(m->ev_build, m->stream), "hipEventRecord");
    } else {
      note(hipEventRecord(m->ev_t2, m->stream), "hipEventRecord");
    }
  }
  if (!c->profiling) m->info.ms_split = ms_since(t_split);

  // ---- 3. a round that has arrived is a complete key range of both relations: join it (remaining radix passes,
  // build, probe) while later rounds are on the links.  The digit bits are consumed: the local partitions start
  // right under them (the reference's recursion masks off consumed bits the same way, radix_hash.h:219-220).
  const auto t_local = std::chrono::steady_clock::now();
  std::memset(local_out, 0, sizeof(*local_out));
  int lerr = HMJ_OK;
  // The waits for rows from other ranks are HOST waits under the step's deadline (a join cannot start before its rows
  // are there anyway): the compute stream never depends on a peer, so nothing this rank queues on it can block for ever.
  if (*err == HMJ_OK) lerr = wait_event(c, m->ev_build, "the build side's exchange rounds");
  if (m->broken.load()) return lerr != HMJ_OK ? lerr : give_up(c, "the build side's exchange rounds");
  if (*err != HMJ_OK) lerr = *err;  // (a local failure while the rounds were queued: no joins, straight to the reduction)
  for (u32 r = 0; r < NR && lerr == HMJ_OK; r++) {
    const u64 nr_ = round_off_r[r + 1] - round_off_r[r], ns_ = round_off_s[r + 1] - round_off_s[r];
    lerr = wait_event(c, m->round_ev[r], "an exchange round (probe rows)");
    if (m->broken.load()) return lerr != HMJ_OK ? lerr : give_up(c, "an exchange round (probe rows)");
    if (lerr != HMJ_OK) break;
    if (ns_ == 0) continue;                              // no probe rows: no result rows, no probe payloads
    if (nr_ == 0 && !(flags & HMJ_SUM_PROBE)) continue;  // nothing can match (and no probe payload sum is asked for)
    hmj_result res;
    spans_reset(c);
    // count modes: skip the bits the digit consumed (a round of several digits differs in them, which any
    // partition function may ignore); materialising joins (one round) let the planner sample the keys itself
    // (as a MINIMUM: the key sample of the round's join still runs, so sorted shards, hot keys and dense build keys
    //  keep their plans -- forcing prefix_bits switched the sample off)
    if (!materialize) c->min_prefix_bits = 64 - low;
    const int rc = join_device(c, static_cast<const char*>(m->recv_r.p) + round_off_r[r] * 16, nr_,
                               static_cast<const char*>(m->recv_s.p) + round_off_s[r] * 16, ns_, flags, &res, false);
    c->min_prefix_bits = 0;
    if (rc != HMJ_OK) {
      lerr = rc;
      break;
    }
    m->info.n_subjoins++;
    if (materialize)
      *local_out = res;  // (one round: the columns of this join)
    else
      add_result(local_out, res);
    if (c->profiling) {
      (void)hipStreamSynchronize(c->stream);
      spans_collect(c);
      add_timing(&acc, c->timing);
    }
  }
  if (lerr != HMJ_OK) *err = lerr;
  // the sends of this rank must have left its buffers before the next step may overwrite them
  {
    const int w = wait_stream(c, m->stream, "this rank's sends (communication stream)");
    if (m->broken.load()) return w != HMJ_OK ? w : give_up(c, "this rank's sends");
    if (w != HMJ_OK && *err == HMJ_OK) *err = w;
  }
  m->info.ms_local = ms_since(t_local);
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, m->ev_t0, m->ev_t3) == hipSuccess) m->info.ms_exchange_build = ms;
  if (hipEventElapsedTime(&ms, m->ev_t1, m->ev_t2) == hipSuccess) m->info.ms_exchange_probe = ms;
  if (c->profiling) {
    const u32 path = acc.path;
    c->timing = acc;
    c->timing.path = path;
    m->info.ms_kernels = kernel_ms(acc);
  }
  return HMJ_OK;
}

// ---- owner-split path: hash owner (fallback for keys the digit ranges cannot balance) or key-range owner
// (HMJ_ORDERED): a separate stable split by owner, then ONE local join of everything this rank received.
int exchange_owner_path(hmj_ctx* c, const void* build_shard_dev, u64 n_build, const void* probe_shard_dev, u64 n_probe,
                        u32 flags, const hmj::OwnerFn& own, hmj_result* local_out, int* err) {
  hmj_comm* m = c->comm;
  const int G = m->n_ranks, me = m->rank;
  int rc;
  m->info.owner_mode = (int)own.mode;
  // ---- split both shards by owner, counts
  const void *parted_r = build_shard_dev, *parted_s = probe_shard_dev;
  std::vector<u64> cnt(2 * (size_t)G + 1, 0), allcnt((2 * (size_t)G + 1) * G, 0);
  const auto t_split = std::chrono::steady_clock::now();
  if (G > 1) {
    int bits = 0;
    while ((1 << bits) < G) bits++;
    const size_t noff = ((size_t)1 << bits) + 1;
    if (*err == HMJ_OK) {
      rc = ensure_dev(c, m->parted_r, (size_t)n_build * 16);
      if (rc == HMJ_OK) rc = ensure_dev(c, m->parted_s, (size_t)n_probe * 16);
      if (rc == HMJ_OK) rc = ensure_dev(c, m->offs, 2 * noff * 8);
      if (rc == HMJ_OK) rc = ensure_host(c, m->offs_host, 2 * noff * 8);
      if (rc == HMJ_OK) {
        u64* od = static_cast<u64*>(m->offs.p);
        u64* oh = static_cast<u64*>(m->offs_host.p);
        rc = owner_split(c, build_shard_dev, (u32)n_build, own, m->parted_r.p, od);
        if (rc == HMJ_OK) rc = owner_split(c, probe_shard_dev, (u32)n_probe, own, m->parted_s.p, od + noff);
        hipError_t e = hipSuccess;
        if (rc == HMJ_OK) e = hipMemcpyAsync(oh, od, 2 * noff * 8, hipMemcpyDeviceToHost, c->stream);
        if (rc == HMJ_OK && e == hipSuccess) e = hipEventRecord(m->ev_split, c->stream);
        if (rc == HMJ_OK && e == hipSuccess) e = hipStreamWaitEvent(m->stream, m->ev_split, 0);  // (before the all-gather: a failure travels in its status word)
        if (rc == HMJ_OK && e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (rc == HMJ_OK && e != hipSuccess) rc = fail(c, HMJ_E_HIP, "owner split", e);
        if (rc == HMJ_OK)
          for (int g = 0; g < G; g++) {
            cnt[1 + g] = oh[g + 1] - oh[g];
            cnt[1 + G + g] = oh[noff + g + 1] - oh[noff + g];
          }
      }
      if (rc != HMJ_OK) *err = rc;
    }
    parted_r = m->parted_r.p;
    parted_s = m->parted_s.p;
  } else {
    cnt[1] = n_build;
    cnt[2] = n_probe;
    if (*err == HMJ_OK) {
      hipError_t e = hipEventRecord(m->ev_split, c->stream);
      if (e == hipSuccess) e = hipStreamWaitEvent(m->stream, m->ev_split, 0);
      if (e != hipSuccess) *err = fail(c, HMJ_E_HIP, "owner split", e);
    }
  }
  m->info.ms_split = ms_since(t_split);
  cnt[0] = *err == HMJ_OK ? 0ull : (u64)(int64_t)*err;
  if ((rc = transport_allgather(c, cnt.data(), allcnt.data(), 2 * G + 1)) != HMJ_OK) return rc;
  std::vector<u64> st((size_t)G);
  for (int g = 0; g < G; g++) st[g] = allcnt[(size_t)g * (2 * G + 1)];
  if ((rc = settle(c, *err, st)) != HMJ_OK) return rc;
  std::vector<u64> MR((size_t)G * G), MS((size_t)G * G);  // [src][dst]
  for (int s = 0; s < G; s++)
    for (int d = 0; d < G; d++) {
      MR[(size_t)s * G + d] = allcnt[(size_t)s * (2 * G + 1) + 1 + d];
      MS[(size_t)s * G + d] = allcnt[(size_t)s * (2 * G + 1) + 1 + G + d];
    }
  // every rank holds the whole matrix: the row limit of a local join is checked for ALL destination ranks, so
  // all ranks fail together (a skewed key range in ordered mode is the usual way to reach it on ONE rank)
  std::vector<u64> need_r((size_t)G, 0), need_s((size_t)G, 0);
  for (int d = 0; d < G; d++)
    for (int s = 0; s < G; s++) {
      need_r[d] += MR[(size_t)s * G + d];
      need_s[d] += MS[(size_t)s * G + d];
    }
  for (int d = 0; d < G; d++)
    if (need_r[d] > 0xFFFFFFFFull || need_s[d] > 0xFFFFFFFFull) {
      char msg[96];
      std::snprintf(msg, sizeof(msg), "rank %d would own more than 2^32-1 rows", d);
      return fail(c, HMJ_E_UNSUPPORTED, msg);
    }
  const u64 nr = need_r[me], ns = need_s[me];
  m->info.recv_build = nr;
  m->info.recv_probe = ns;

  // ---- rounds and receive layout
  // Build side: source-major (sources in rank order = global input order, which HMJ_FIRST_WINS relies on).
  // Probe side: round-major and in several rounds, so that the local pass A can start on the rows that have
  // arrived while later rounds are still on the links.
  const u64 max_rows = std::max<u64>(1, m->max_msg_bytes / 16);
  const u32 rounds_r = hmj_exchange_rounds(G, MR.data(), max_rows);
  u64 biggest_s = 0;
  for (u64 v : MS) biggest_s = std::max(biggest_s, v);
  u64 per_round = std::max<u64>(1, std::min<u64>(max_rows, m->target_round_bytes / 16));
  if (biggest_s / per_round > 16) per_round = (biggest_s + 15) / 16;  // at most 16 rounds
  if (per_round > max_rows) per_round = max_rows;
  const u32 rounds_s = hmj_exchange_rounds(G, MS.data(), per_round);
  m->info.rounds_build = rounds_r;
  m->info.rounds_probe = rounds_s;
  {
    const bool grows = any_rank_grows(m, 2, need_r) || any_rank_grows(m, 3, need_s);
    int mine = ensure_dev(c, m->recv_r, (size_t)nr * 16 + 16);
    if (mine == HMJ_OK) mine = ensure_dev(c, m->recv_s, (size_t)ns * 16 + 16);
    if (mine == HMJ_OK) mine = ensure_round_events(c, rounds_s);
    if (grows) {
      if ((rc = status_round(c, mine)) != HMJ_OK) return rc;
      note_peaks(m, 2, need_r);
      note_peaks(m, 3, need_s);
    } else if (mine != HMJ_OK) {
      return mine;
    }
  }

  // ---- exchange: everything is queued on the communication stream behind the splits (the stream was told to wait
  // for them before the counts were gathered).  Every rank queues all its rounds; a failed event record becomes
  // this rank's pending error (no local join, reported by the final reduction); only the transport returns at once.
  auto note = [&](hipError_t e2, const char* what) {
    if (e2 != hipSuccess && *err == HMJ_OK) *err = fail(c, HMJ_E_HIP, what, e2);
  };
  note(hipEventRecord(m->ev_t0, m->stream), "hipEventRecord");
  if ((rc = exchange_relation(c, parted_r, MR.data(), rounds_r, 0, m->recv_r.p, nullptr, nullptr, err)) != HMJ_OK) return rc;
  note(hipEventRecord(m->ev_build, m->stream), "hipEventRecord");
  note(hipEventRecord(m->ev_t1, m->stream), "hipEventRecord");
  std::vector<u64> round_end(rounds_s);
  if ((rc = exchange_relation(c, parted_s, MS.data(), rounds_s, 1, m->recv_s.p, m->round_ev.data(), round_end.data(), err)) != HMJ_OK)
    return rc;
  note(hipEventRecord(m->ev_t2, m->stream), "hipEventRecord");

  // ---- local join: build side as soon as it is complete, probe side as its rounds arrive
  const auto t_local = std::chrono::steady_clock::now();
  int lerr = *err;
  // (host waits under the step's deadline, as in the digit path; the probe side's rounds are waited for inside the
  //  join through hmj_ctx::arrive_wait)
  if (lerr == HMJ_OK) lerr = wait_event(c, m->ev_build, "the build side's exchange rounds");
  if (m->broken.load()) return lerr != HMJ_OK ? lerr : give_up(c, "the build side's exchange rounds");
  spans_reset(c);
  c->sample_build_only = true;
  // count modes (checksums and first-wins included): the build side is partitioned while the probe rows are on
  // the links -- the plan of a join does not depend on those flags, only materialising joins may plan differently
  if (lerr == HMJ_OK && !(flags & HMJ_MATERIALIZE)) lerr = prepare_build(c, m->recv_r.p, nr, ns);
  if (lerr == HMJ_OK) {
    c->arrive_rows.assign(round_end.begin(), round_end.end());
    c->arrive_ev.assign(m->round_ev.begin(), m->round_ev.begin() + rounds_s);
    lerr = join_device(c, m->recv_r.p, nr, m->recv_s.p, ns, flags, local_out, false);
    if (lerr == HMJ_OK) m->info.n_subjoins = 1;
  }
  c->sample_build_only = false;
  c->arrive_rows.clear();
  c->arrive_ev.clear();
  if (c->profiling) {
    (void)hipStreamSynchronize(c->stream);
    spans_collect(c);
    m->info.ms_kernels = kernel_ms(c->timing);
  }
  if (lerr != HMJ_OK) *err = lerr;
  {
    const int w = m->broken.load() ? HMJ_E_TIMEOUT : wait_stream(c, m->stream, "this rank's sends (communication stream)");
    if (m->broken.load()) return lerr == HMJ_E_TIMEOUT ? lerr : give_up(c, "the exchange rounds");
    if (w != HMJ_OK && *err == HMJ_OK) *err = w;
  }
  m->info.ms_local = ms_since(t_local);
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, m->ev_t0, m->ev_t1) == hipSuccess) m->info.ms_exchange_build = ms;
  if (hipEventElapsedTime(&ms, m->ev_t1, m->ev_t2) == hipSuccess) m->info.ms_exchange_probe = ms;
  return HMJ_OK;
}

}  // namespace

extern "C" int hmj_exchange_join_u64_device(hmj_ctx* c, const void* build_shard_dev, uint64_t n_build, const void* probe_shard_dev,
                                            uint64_t n_probe, uint32_t flags, hmj_result* local_out, hmj_result* global_out) {
  if (!c) return HMJ_E_ARG;
  // (argument errors every rank makes alike -- the same program runs everywhere -- return at once)
  if (!local_out) return fail(c, HMJ_E_ARG, "local_out is NULL");
  if (!c->comm) return fail(c, HMJ_E_ARG, "no communicator: call hmj_comm_init_rank or hmj_comm_set_transport first");
  HIP_TRY(hipSetDevice(c->device));
  hmj_comm* m = c->comm;
  const int G = m->n_ranks;
  if (m->broken.load())
    return fail(c, m->broken_code == HMJ_E_TIMEOUT ? HMJ_E_TIMEOUT : HMJ_E_RCCL,
                "the communicator was aborted by an earlier step (timeout or RCCL error): destroy it and create a new one");
  if (!m->has_cb && !m->nccl) return fail(c, HMJ_E_ARG, "communicator has no transport");
  StepScope step_scope(c);  // the deadline of this step; closed on every return below
  if (flags & HMJ_ORDERED) flags |= HMJ_MATERIALIZE;
  std::memset(&m->info, 0, sizeof(m->info));
  std::memset(local_out, 0, sizeof(*local_out));
  m->info.n_ranks = G;
  const auto t_begin = std::chrono::steady_clock::now();
  // errors that depend on THIS rank's shard are carried to the next collective, where all ranks leave together
  int err = HMJ_OK;
  if (n_build > 0xFFFFFFFFull || n_probe > 0xFFFFFFFFull) err = fail(c, HMJ_E_ARG, "more than 2^32-1 rows in one shard");
  else if ((n_build && !build_shard_dev) || (n_probe && !probe_shard_dev)) err = fail(c, HMJ_E_ARG, "shard pointer is NULL");

  if (G == 1 && !m->self_exchange) {
    // one rank owns every key: nothing to exchange, the step is the plain local join
    if (err != HMJ_OK) return err;
    spans_reset(c);
    int rc = join_device(c, build_shard_dev, n_build, probe_shard_dev, n_probe, flags, local_out, false);
    if (c->profiling) {
      (void)hipStreamSynchronize(c->stream);
      spans_collect(c);
      m->info.ms_kernels = kernel_ms(c->timing);
    }
    if (rc != HMJ_OK) return rc;
    m->info.recv_build = n_build;
    m->info.recv_probe = n_probe;
    m->info.n_subjoins = 1;
    if (global_out) {
      *global_out = *local_out;
      global_out->key = global_out->rval = global_out->sval = nullptr;
    }
    m->info.ms_total = m->info.ms_local = ms_since(t_begin);
    m->info.ms_exposed = c->profiling ? std::max(0.f, m->info.ms_total - m->info.ms_kernels) : 0.f;
    return HMJ_OK;
  }

  // ---- 1. every rank's key sample and shard sizes (one all-gather; also the first point all ranks can fail at)
  StepSample smp;
  int rc = gather_samples(c, build_shard_dev, n_build, probe_shard_dev, n_probe, &err, &smp);
  if (rc != HMJ_OK) return rc;

  // ---- 2. the owner function
  hmj::OwnerFn own;
  std::memset(&own, 0, sizeof(own));
  own.G = (u32)G;
  if (flags & HMJ_ORDERED) {
    // Ordered results: rank g owns the g-th key range between splitters all ranks agree on (their pooled key
    // samples' quantiles), so per-rank ordered results concatenate in key order.
    own.mode = 2;
    std::vector<u64> keys = smp.keys;
    std::sort(keys.begin(), keys.end());
    for (int i = 0; i + 1 < G; i++)
      own.spl[i] = keys.empty() ? ~0ull : keys[std::min(keys.size() - 1, keys.size() * (size_t)(i + 1) / (size_t)G)];
    rc = exchange_owner_path(c, build_shard_dev, n_build, probe_shard_dev, n_probe, flags, own, local_out, &err);
  } else {
    // rounds: about target_round_bytes per message of the probe side (materialising joins: one round -- their
    // result columns are those of ONE local join)
    u64 pair_rows = 0;
    for (int g = 0; g < G; g++) pair_rows = std::max<u64>(pair_rows, std::max(smp.np[g], smp.nb[g]) / (u64)G);
    const u64 per_round = std::max<u64>(1, m->target_round_bytes / 16);
    u64 nr64 = (flags & HMJ_MATERIALIZE) ? 1 : (pair_rows + per_round - 1) / per_round;
    // a round's join should keep the efficient kernels busy: with the default round size, no more rounds than give
    // every round about 2^26 probe rows per rank (smaller joins cost more per row: 2^24 rows 0.039 ns, 2^25 0.037,
    // 2^28 0.029 -- and every round costs a host round trip for its result)
    if (!m->round_bytes_set) {
      u64 total_np = 0;
      for (int g = 0; g < G; g++) total_np += smp.np[g];
      nr64 = std::min<u64>(nr64, std::max<u64>(1, (total_np / (u64)G) >> 26));
    }
    nr64 = std::min<u64>(std::max<u64>(nr64, 1), HMJ_MAX_ROUNDS);
    hmj_digit_plan plan;
    if (hmj_exchange_digit_plan(G, smp.keys.data(), smp.keys.size(), (u32)nr64, &plan) != HMJ_OK) return fail(c, HMJ_E_ARG, "digit plan");
    if (plan.usable && m->owner_path == HMJ_OWNER_DIGIT) {
      rc = exchange_digit_path(c, build_shard_dev, n_build, probe_shard_dev, n_probe, flags, plan, local_out, &err);
    } else {
      // a few clusters of keys: no contiguous digit ranges balance the ranks -- the hash owner spreads ANY key set
      own.mode = 1;
      m->info.fallback = plan.usable ? 0 : 1;  // (0: the caller asked for this path, hmj_comm_set_owner_path)
      m->info.sample_max_share = plan.max_share;
      rc = exchange_owner_path(c, build_shard_dev, n_build, probe_shard_dev, n_probe, flags, own, local_out, &err);
    }
  }
  if (rc != HMJ_OK) return rc;

  // ---- 3. the reduction over all ranks (and the last point a failed local join is reported at)
  rc = final_reduction(c, err, local_out, global_out);
  m->info.ms_total = ms_since(t_begin);
  m->info.ms_exposed = c->profiling ? std::max(0.f, m->info.ms_total - m->info.ms_kernels) : 0.f;
  return rc;
}
