// Multi-GPU partition exchange behind the C ABI (include/hmj.h, "multi-GPU" section).
// One process per GPU.  The radix fan-out shards the join: every row has an OWNER rank, the ranks exchange
// their rows (all-to-all-v) and each joins what it owns -- partition p of the probe side only ever meets
// table p (hashjoin_bench.cc:92-96), so no further communication is needed.  The reference reaches all of its
// parallelism from the ctor (hashjoin.h:56-68 -> radix_hash.h:375-405, threads of one address space); this is
// the same fork-join across address spaces.
//
//   owner split (radix.hip, owner_digit)  ->  counts all-gather  ->  rounds of grouped send/recv  ->  local join
//
// Transport: RCCL (ncclSend / ncclRecv per peer inside ncclGroupStart / ncclGroupEnd, on the communicator's own
// HIP stream; librccl is loaded with dlopen, so the library has no link-time dependency on it) or callbacks
// the host supplies (an existing communicator; the tests drive several ranks on one GPU through gloo that way).
// Everything else -- owner function, split, round plan, receive layout, overlap with the local join -- is the
// same code for both.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstring>

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
  std::string error;
};

RcclApi* rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api.dl ? &api : nullptr;
  tried = true;
  // a process that already holds RCCL (PyTorch bundles one under the same SONAME) gets that copy back
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    api.dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (api.dl) break;
  }
  if (!api.dl) {
    api.error = dlerror() ? dlerror() : "librccl not found";
    return nullptr;
  }
#define HMJ_SYM(field, name)                                          \
  api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.dl, name)); \
  if (!api.field) {                                                   \
    api.error = std::string("librccl lacks ") + name;                \
    api.dl = nullptr;                                                 \
    return nullptr;                                                   \
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
  return &api;
}

constexpr int kSampleKeys = 2048;  // per relation and rank, for the ordered mode's splitters

}  // namespace

struct hmj_comm {
  int n_ranks = 1, rank = 0;
  ncclComm_t nccl = nullptr;  // RCCL transport
  hmj_transport cb;           // callback transport
  bool has_cb = false;
  hipStream_t stream = nullptr;  // communication stream
  hipEvent_t ev_split = nullptr, ev_build = nullptr, ev_t0 = nullptr, ev_t1 = nullptr, ev_t2 = nullptr;
  std::vector<hipEvent_t> round_ev;
  DevBuf parted_r, parted_s, recv_r, recv_s, offs, gather_dev, sample_dev;
  HostBuf gather_host;
  u64 max_msg_bytes = 1ull << 30;       // RCCL 2.26 truncates a single message of 2 GiB or more
  u64 target_round_bytes = 128ull << 20;  // probe side: several rounds, so the local pass A starts on arrived rows
  hmj_exchange_info info;
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

void comm_free(hmj_comm* m) {  // everything a (possibly half-built) communicator holds, except the RCCL handle
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  DevBuf* devs[] = {&m->parted_r, &m->parted_s, &m->recv_r, &m->recv_s, &m->offs, &m->gather_dev, &m->sample_dev};
  for (DevBuf* b : devs) free_dev(*b);
  free_host(m->gather_host);
  hipEvent_t evs[] = {m->ev_split, m->ev_build, m->ev_t0, m->ev_t1, m->ev_t2};
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
  hipEvent_t* evs[] = {&m->ev_split, &m->ev_build};
  for (hipEvent_t* e : evs) ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
  hipEvent_t* tevs[] = {&m->ev_t0, &m->ev_t1, &m->ev_t2};
  for (hipEvent_t* e : tevs) ok = ok && hipEventCreate(e) == hipSuccess;
  if (!ok) {
    comm_free(m);
    return fail(c, HMJ_E_HIP, "communication stream / events");
  }
  c->comm = m;
  return HMJ_OK;
}

// ---- transport ----------------------------------------------------------------------------------------
// all ranks learn every rank's `count` values: recv[r * count + i] = rank r's send[i].  Host memory, blocking.
int transport_allgather(hmj_ctx* c, const u64* send, u64* recv, int count) {
  hmj_comm* m = c->comm;
  if (m->n_ranks == 1) {
    std::memcpy(recv, send, (size_t)count * 8);
    return HMJ_OK;
  }
  if (m->has_cb) {
    if (m->cb.allgather_u64(m->cb.user, send, recv, count) != 0) return fail(c, HMJ_E_RCCL, "transport: allgather_u64 failed");
    return HMJ_OK;
  }
  RcclApi* a = rccl_api();
  int rc;
  const size_t bytes = (size_t)count * 8;
  if ((rc = ensure_dev(c, m->gather_dev, bytes * (m->n_ranks + 1))) != HMJ_OK) return rc;
  char* d = static_cast<char*>(m->gather_dev.p);
  HIP_TRY(hipMemcpyAsync(d, send, bytes, hipMemcpyHostToDevice, m->stream));
  RCCL_TRY(a->AllGather(d, d + bytes, (size_t)count, ncclUint64, m->nccl, m->stream));
  HIP_TRY(hipMemcpyAsync(recv, d + bytes, bytes * m->n_ranks, hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  return HMJ_OK;
}

// one round of the all-to-all-v on device memory, queued on the communication stream
int transport_round(hmj_ctx* c, int round, const void* const* sp, const u64* sb, void* const* rp, const u64* rb) {
  hmj_comm* m = c->comm;
  const int G = m->n_ranks, me = m->rank;
  if (m->has_cb) {
    if (m->cb.alltoallv(m->cb.user, round, sp, sb, rp, rb, (void*)m->stream) != 0)
      return fail(c, HMJ_E_RCCL, "transport: alltoallv failed");
    return HMJ_OK;
  }
  RcclApi* a = rccl_api();
  // this rank's own bucket: a device copy when there are peers (it overlaps the link traffic); with a single
  // rank the copy goes through RCCL's send/recv pair as well, which keeps that path exercised on one GPU
  const bool self_rccl = G == 1;
  if (!self_rccl && sb[me]) HIP_TRY(hipMemcpyAsync(rp[me], sp[me], sb[me], hipMemcpyDeviceToDevice, m->stream));
  bool any = false;
  for (int g = 0; g < G; g++)
    if ((g != me || self_rccl) && (sb[g] || rb[g])) any = true;
  if (!any) return HMJ_OK;
  RCCL_TRY(a->GroupStart());
  for (int g = 0; g < G; g++) {
    if (g == me && !self_rccl) continue;
    if (sb[g]) RCCL_TRY(a->Send(sp[g], (size_t)sb[g], ncclUint8, g, m->nccl, m->stream));
    if (rb[g]) RCCL_TRY(a->Recv(rp[g], (size_t)rb[g], ncclUint8, g, m->nccl, m->stream));
  }
  RCCL_TRY(a->GroupEnd());
  return HMJ_OK;
}

// ---- kernels ------------------------------------------------------------------------------------------
__global__ void sample_keys_kernel(const hmj::Tup* __restrict__ a, u64 n, u32 K, u64* __restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < K) out[i] = a[(u64)i * n / K].key;
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

namespace {

// Exchange one relation: rounds of alltoallv queued on the communication stream.  ev_after_round (optional):
// events recorded after each round (n_rounds of them).  Returns rows received.
int exchange_relation(hmj_ctx* c, const void* parted, const u64* counts_matrix, u32 n_rounds, int layout, void* recv,
                      hipEvent_t* ev_after_round, u64* round_end) {
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
    if (ev_after_round) HIP_TRY(hipEventRecord(ev_after_round[r], m->stream));
  }
  return HMJ_OK;
}

}  // namespace

namespace hmj_host {
void comm_destroy(hmj_ctx* c) {
  hmj_comm* m = c->comm;
  if (!m) return;
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  if (m->nccl) {
    RcclApi* a = rccl_api();
    if (a) (void)a->CommDestroy(m->nccl);
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
  if (probe_round_bytes) c->comm->target_round_bytes = probe_round_bytes < 16 ? 16 : probe_round_bytes;
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

int hmj_exchange_join_u64_device(hmj_ctx* c, const void* build_shard_dev, uint64_t n_build, const void* probe_shard_dev,
                                 uint64_t n_probe, uint32_t flags, hmj_result* local_out, hmj_result* global_out) {
  if (!c) return HMJ_E_ARG;
  if (!local_out) return fail(c, HMJ_E_ARG, "local_out is NULL");
  if (!c->comm) return fail(c, HMJ_E_ARG, "no communicator: call hmj_comm_init_rank or hmj_comm_set_transport first");
  if (n_build > 0xFFFFFFFFull || n_probe > 0xFFFFFFFFull) return fail(c, HMJ_E_ARG, "more than 2^32-1 rows in one shard");
  if ((n_build && !build_shard_dev) || (n_probe && !probe_shard_dev)) return fail(c, HMJ_E_ARG, "shard pointer is NULL");
  HIP_TRY(hipSetDevice(c->device));
  hmj_comm* m = c->comm;
  const int G = m->n_ranks, me = m->rank;
  if (!m->has_cb && !m->nccl) return fail(c, HMJ_E_ARG, "communicator has no transport");
  if (flags & HMJ_ORDERED) flags |= HMJ_MATERIALIZE;
  int rc;
  std::memset(&m->info, 0, sizeof(m->info));
  m->info.n_ranks = G;
  const auto t_begin = std::chrono::steady_clock::now();
  auto ms_since = [](std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t).count();
  };

  // ---- 1. owner function ---------------------------------------------------------------------------
  // Ordered results: rank g owns the g-th key range between splitters all ranks agree on (their pooled key
  // samples' quantiles), so per-rank ordered results concatenate in key order.  Every other mode: the owner
  // is a mixing hash of the key, which spreads any key set -- dense integer keys included -- evenly.
  hmj::OwnerFn own;
  std::memset(&own, 0, sizeof(own));
  own.G = (u32)G;
  own.mode = (flags & HMJ_ORDERED) ? 2u : 1u;
  m->info.owner_mode = (int)own.mode;
  if (own.mode == 2 && G > 1) {
    const int K = kSampleKeys, W = 2 * K + 2;
    if ((rc = ensure_dev(c, m->sample_dev, (size_t)2 * K * 8)) != HMJ_OK) return rc;
    if ((rc = ensure_host(c, m->gather_host, (size_t)W * 8 * (G + 1))) != HMJ_OK) return rc;
    u64* mine = static_cast<u64*>(m->gather_host.p);
    u64* all = mine + W;
    u64* sd = static_cast<u64*>(m->sample_dev.p);
    const u32 kr = (u32)std::min<u64>(K, n_build), ks = (u32)std::min<u64>(K, n_probe);
    if (kr) hipLaunchKernelGGL(sample_keys_kernel, dim3((kr + 255) / 256), dim3(256), 0, c->stream,
                               static_cast<const hmj::Tup*>(build_shard_dev), (u64)n_build, kr, sd);
    if (ks) hipLaunchKernelGGL(sample_keys_kernel, dim3((ks + 255) / 256), dim3(256), 0, c->stream,
                               static_cast<const hmj::Tup*>(probe_shard_dev), (u64)n_probe, ks, sd + K);
    HIP_TRY(hipGetLastError());
    mine[0] = kr;
    mine[1] = ks;
    HIP_TRY(hipMemcpyAsync(mine + 2, sd, (size_t)2 * K * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if ((rc = transport_allgather(c, mine, all, W)) != HMJ_OK) return rc;
    std::vector<u64> keys;
    for (int g = 0; g < G; g++) {
      const u64* p = all + (size_t)g * W;
      keys.insert(keys.end(), p + 2, p + 2 + p[0]);
      keys.insert(keys.end(), p + 2 + K, p + 2 + K + p[1]);
    }
    std::sort(keys.begin(), keys.end());
    for (int i = 0; i + 1 < G; i++)
      own.spl[i] = keys.empty() ? ~0ull : keys[std::min(keys.size() - 1, keys.size() * (size_t)(i + 1) / (size_t)G)];
  }

  // ---- 2. split both shards by owner, 3. counts -------------------------------------------------------
  const void *parted_r = build_shard_dev, *parted_s = probe_shard_dev;
  std::vector<u64> cnt(2 * (size_t)G, 0), allcnt(2 * (size_t)G * G, 0);
  const auto t_split = std::chrono::steady_clock::now();
  if (G > 1) {
    int bits = 0;
    while ((1 << bits) < G) bits++;
    const size_t noff = ((size_t)1 << bits) + 1;
    if ((rc = ensure_dev(c, m->parted_r, (size_t)n_build * 16)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, m->parted_s, (size_t)n_probe * 16)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, m->offs, 2 * noff * 8)) != HMJ_OK) return rc;
    if ((rc = ensure_host(c, m->gather_host, std::max<size_t>(2 * noff * 8, (size_t)(2 * kSampleKeys + 2) * 8 * (G + 1)))) != HMJ_OK)
      return rc;
    u64* od = static_cast<u64*>(m->offs.p);
    if ((rc = owner_split(c, build_shard_dev, (u32)n_build, own, m->parted_r.p, od)) != HMJ_OK) return rc;
    if ((rc = owner_split(c, probe_shard_dev, (u32)n_probe, own, m->parted_s.p, od + noff)) != HMJ_OK) return rc;
    u64* oh = static_cast<u64*>(m->gather_host.p);
    HIP_TRY(hipMemcpyAsync(oh, od, 2 * noff * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(m->ev_split, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int g = 0; g < G; g++) {
      cnt[g] = oh[g + 1] - oh[g];
      cnt[G + g] = oh[noff + g + 1] - oh[noff + g];
    }
    parted_r = m->parted_r.p;
    parted_s = m->parted_s.p;
  } else {
    cnt[0] = n_build;
    cnt[1] = n_probe;
    HIP_TRY(hipEventRecord(m->ev_split, c->stream));
  }
  m->info.ms_split = ms_since(t_split);
  if ((rc = transport_allgather(c, cnt.data(), allcnt.data(), 2 * G)) != HMJ_OK) return rc;
  std::vector<u64> MR((size_t)G * G), MS((size_t)G * G);  // [src][dst]
  u64 nr = 0, ns = 0;
  for (int s = 0; s < G; s++)
    for (int d = 0; d < G; d++) {
      MR[(size_t)s * G + d] = allcnt[(size_t)s * 2 * G + d];
      MS[(size_t)s * G + d] = allcnt[(size_t)s * 2 * G + G + d];
    }
  for (int s = 0; s < G; s++) {
    nr += MR[(size_t)s * G + me];
    ns += MS[(size_t)s * G + me];
  }
  if (nr > 0xFFFFFFFFull || ns > 0xFFFFFFFFull) return fail(c, HMJ_E_UNSUPPORTED, "this rank would own more than 2^32-1 rows");
  m->info.recv_build = nr;
  m->info.recv_probe = ns;

  // ---- 4. rounds and receive layout -------------------------------------------------------------------
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
  if ((rc = ensure_dev(c, m->recv_r, (size_t)nr * 16 + 16)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, m->recv_s, (size_t)ns * 16 + 16)) != HMJ_OK) return rc;
  while (m->round_ev.size() < rounds_s) {
    hipEvent_t e;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    m->round_ev.push_back(e);
  }

  // ---- 5. exchange: everything is queued on the communication stream behind the splits -----------------
  HIP_TRY(hipStreamWaitEvent(m->stream, m->ev_split, 0));
  HIP_TRY(hipEventRecord(m->ev_t0, m->stream));
  if ((rc = exchange_relation(c, parted_r, MR.data(), rounds_r, 0, m->recv_r.p, nullptr, nullptr)) != HMJ_OK) return rc;
  HIP_TRY(hipEventRecord(m->ev_build, m->stream));
  HIP_TRY(hipEventRecord(m->ev_t1, m->stream));
  std::vector<u64> round_end(rounds_s);
  if ((rc = exchange_relation(c, parted_s, MS.data(), rounds_s, 1, m->recv_s.p, m->round_ev.data(), round_end.data())) != HMJ_OK)
    return rc;
  HIP_TRY(hipEventRecord(m->ev_t2, m->stream));

  // ---- 6. local join: build side as soon as it is complete, probe side as its rounds arrive ------------
  const auto t_local = std::chrono::steady_clock::now();
  HIP_TRY(hipStreamWaitEvent(c->stream, m->ev_build, 0));
  spans_reset(c);
  c->sample_build_only = true;
  if (flags == 0) {  // plain count join: the build side is partitioned while the probe rows are on the links
    rc = prepare_build(c, m->recv_r.p, nr, ns);
    if (rc != HMJ_OK) {
      c->sample_build_only = false;
      return rc;
    }
  }
  c->arrive_rows.assign(round_end.begin(), round_end.end());
  c->arrive_ev.assign(m->round_ev.begin(), m->round_ev.begin() + rounds_s);
  rc = join_device(c, m->recv_r.p, nr, m->recv_s.p, ns, flags, local_out, false);
  c->sample_build_only = false;
  c->arrive_rows.clear();
  c->arrive_ev.clear();
  if (c->profiling) {
    (void)hipStreamSynchronize(c->stream);
    spans_collect(c);
  }
  if (rc != HMJ_OK) return rc;
  HIP_TRY(hipStreamSynchronize(m->stream));
  m->info.ms_local = ms_since(t_local);
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, m->ev_t0, m->ev_t1) == hipSuccess) m->info.ms_exchange_build = ms;
  if (hipEventElapsedTime(&ms, m->ev_t1, m->ev_t2) == hipSuccess) m->info.ms_exchange_probe = ms;

  // ---- 7. the reduction over all ranks ------------------------------------------------------------------
  if (global_out) {
    std::memset(global_out, 0, sizeof(*global_out));
    u64 mine[6] = {local_out->n_matches, local_out->sum_r, local_out->sum_s, local_out->xor_fold, local_out->mix_sum,
                   local_out->sum_probe_all};
    std::vector<u64> all(6 * (size_t)G);
    if ((rc = transport_allgather(c, mine, all.data(), 6)) != HMJ_OK) return rc;
    for (int g = 0; g < G; g++) {
      const u64* p = &all[6 * (size_t)g];
      global_out->n_matches += p[0];
      global_out->sum_r += p[1];
      global_out->sum_s += p[2];
      global_out->xor_fold ^= p[3];
      global_out->mix_sum += p[4];
      global_out->sum_probe_all += p[5];
    }
  }
  m->info.ms_total = ms_since(t_begin);
  return HMJ_OK;
}

}  // extern "C"
