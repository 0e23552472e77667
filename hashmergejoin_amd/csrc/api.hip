// extern "C" ABI of libhmj_hip.so (include/hmj.h): context, workspace, planner and the join
// driver that sequences the gfx950 kernels on one HIP stream.  No CPU compute path exists here:
// every entry point either runs the HIP kernels or returns an error code.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "hmj_ctx.h"

using hmj::u32;
using hmj::u64;
using namespace hmj_host;

namespace hmj_host {

constexpr int kMaxEvents = 1024;  // two per span; a multi-round exchange step keeps the spans of all its sub-joins

int fail(hmj_ctx* c, int code, const char* what, hipError_t e) {
  if (c) {
    c->last_error = what;
    if (e != hipSuccess) {
      c->last_error += ": ";
      c->last_error += hipGetErrorString(e);
    }
  }
  return code;
}

// the per-phase times and byte counts of one (sub-)join, added to the step's totals
void add_timing(hmj_timing* acc, const hmj_timing& t) {
  acc->ms_total += t.ms_total;
  acc->ms_partition_build += t.ms_partition_build;
  acc->ms_partition_probe += t.ms_partition_probe;
  acc->ms_hist += t.ms_hist;
  acc->ms_scan += t.ms_scan;
  acc->ms_scatter += t.ms_scatter;
  acc->ms_offsets += t.ms_offsets;
  acc->ms_probe_count += t.ms_probe_count;
  acc->ms_out_scan += t.ms_out_scan;
  acc->ms_probe_write += t.ms_probe_write;
  acc->ms_order += t.ms_order;
  acc->n_scatter_launches += t.n_scatter_launches;
  acc->n_split_retries += t.n_split_retries;
  acc->bytes_scatter += t.bytes_scatter;
  acc->bytes_hist += t.bytes_hist;
  acc->bytes_probe_count += t.bytes_probe_count;
  acc->bytes_probe_write += t.bytes_probe_write;
  acc->path |= t.path;
  acc->ms_scatter_pass[0] += t.ms_scatter_pass[0];
  acc->ms_scatter_pass[1] += t.ms_scatter_pass[1];
  if (t.radix_passes) {  // the plan of the latest sub-join that partitioned anything
    acc->radix_bits = t.radix_bits;
    acc->radix_passes = t.radix_passes;
    acc->key_prefix_bits = t.key_prefix_bits;
    acc->key_window_low = t.key_window_low;
  }
  acc->n_probe_items += t.n_probe_items;
}

#define HIP_TRY(expr)                                                   \
  do {                                                                  \
    hipError_t _e = (expr);                                             \
    if (_e != hipSuccess) return fail(c, HMJ_E_HIP, #expr, _e);         \
  } while (0)

// ---- what the context has learnt, per workload ------------------------------------------------------------------
static int ilog2_u64(uint64_t v) { return v ? 63 - __builtin_clzll(v) : 0; }
uint64_t workload_signature(uint64_t n_build, uint64_t n_probe, uint32_t flags, int kind) {
  const uint32_t mode = ((flags & HMJ_ORDERED) ? 4u : 0u) | ((flags & (HMJ_MATERIALIZE | HMJ_ORDERED)) ? 2u : 0u) |
                        ((flags & HMJ_FIRST_WINS) ? 1u : 0u);
  return ((uint64_t)(kind & 15) << 20) | ((uint64_t)mode << 16) | ((uint64_t)ilog2_u64(n_build) << 8) | (uint64_t)ilog2_u64(n_probe);
}
WorkloadMemo* memo_for(hmj_ctx* c, uint64_t sig) {
  auto it = c->memos.find(sig);
  if (it == c->memos.end()) {
    if (c->memos.size() >= 1024) c->memos.clear();  // (a context that has seen a thousand shapes starts over)
    it = c->memos.emplace(sig, c->memo_init).first;
  }
  c->wm = &it->second;
  c->wm_sig = sig;
  return c->wm;
}
uint32_t WorkloadMemo::cooling() const {
  return (uniq_cooldown ? HMJ_COOL_UNIQ_WRITE : 0u) | (sorted_cooldown ? HMJ_COOL_SORTED_WRITE : 0u) | (gtable_cooldown ? HMJ_COOL_GTABLE : 0u) |
         (gtable_write_cooldown ? HMJ_COOL_GTABLE_WRITE : 0u) | (gtable_sort_cooldown ? HMJ_COOL_RANK_SORT : 0u) |
         (gtable_sort_slab_cooldown ? HMJ_COOL_RANK_SORT_SLAB : 0u) | (expand_cooldown ? HMJ_COOL_EXPANSION : 0u) |
         (sort_slab_cooldown ? HMJ_COOL_SORT_SLAB : 0u) | (slab_cooldown ? HMJ_COOL_SLAB : 0u) | (slab_probe_cooldown ? HMJ_COOL_SLAB_PROBE : 0u) |
         (one_pass_write_cooldown ? HMJ_COOL_ONE_PASS_WRITE : 0u) | (exact_prefix_joins ? HMJ_COOL_EXACT_PREFIX : 0u) |
         (rank_runs_cooldown ? HMJ_COOL_RANK_RUNS : 0u) | (sort_msd_cooldown ? HMJ_COOL_SORT_MSD : 0u);
}

int wait_arrival(hmj_ctx* c, hipEvent_t ev) {
  if (c->arrive_wait) return c->arrive_wait(c, ev);
  HIP_TRY(hipStreamWaitEvent(c->stream, ev, 0));
  return HMJ_OK;
}

int ensure_dev(hmj_ctx* c, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return HMJ_OK;
  // a prepared build side (hmj_prepare_build_u64_device) lives in these buffers: once one of them is
  // reallocated the prepared state points at freed memory, so it is dropped here, whoever the caller is
  if (c && (&b == &c->rbuf[0] || &b == &c->rbuf[1] || &b == &c->r_off || &b == &c->slab_br || &b == &c->cnt_br))
    c->prep.valid = false;
  if (b.p) {
    hipError_t e = hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    if (e != hipSuccess) return fail(c, HMJ_E_HIP, "hipFree", e);
  }
  size_t want = bytes + (bytes >> 4) + 256;  // a little headroom against regrowth
#ifdef HMJ_DEV
  if (const char* e = getenv("HMJ_ALLOC_PAD_KB")) want += (size_t)atoll(e) << 10;       // placement experiments
  if (const char* e = getenv("HMJ_ALLOC_ROUND_MB")) {
    const size_t g = (size_t)atoll(e) << 20;
    if (g) want = (want + g - 1) / g * g;
  }
#endif
  const auto t_alloc0 = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    b.p = nullptr;
    (void)hipGetLastError();
    return fail(c, HMJ_E_OOM, "hipMalloc", e);
  }
  const float ms_alloc0 = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_alloc0).count();
  b.cap = want;
  // ---- placement.  How fast a buffer can be WRITTEN depends on the physical memory the driver backs it with
  // (tools/micro/place_bw.hip, place_vmm.hip: allocations of one process fill at 4.4-4.6 TB/s or
  // at 5.3-5.7 TB/s, the same buffer always the same, whichever API created it; reads do not care).  Only the buffers
  // the partition passes write -- where the effect was measured -- of 2 GiB and more (HMJ_PLACE_MIN_MB) are looked at:
  //   * every such allocation is PROBED: filled twice, the second fill timed (2.4 ms for 6 GB), the rate logged
  //     (hmj_placement_info) -- that is all a join does on its own;
  //   * a SEARCH for better memory runs only when the caller asked for workspace ahead of time (hmj_reserve) or set
  //     HMJ_PLACE=n in the environment: while the buffer fills slower than a good one does, another candidate is
  //     allocated and the faster of the two kept -- never more than two alive (2 x want), at most c->place_tries
  //     candidates, and under a wall-clock budget (c->place_budget_ms, HMJ_PLACE_BUDGET_MS; 50 ms): round 3's driver
  //     run paid 1.3 s in the first join for one buffer whose three extra candidates took 435 ms each to create
  //     (BENCH_r03: the other two buffers' searches cost 13 ms), for a gain of at most 0.3 ms per join.  The search
  //     stops as soon as what it has spent plus what the last candidate cost would exceed the budget; every
  //     candidate's hipMalloc and fill times are recorded.
  // Exchange buffers, result columns and upload targets are never probed (their (re)growth must not stall
  // in-flight rounds with extra device-synchronising hipMalloc / hipFree calls).
  const char* pname = !c ? nullptr
                      : &b == &c->slab_a ? "slab_a" : &b == &c->slab_br ? "slab_b_build" : &b == &c->slab_bs ? "slab_b_probe"
                      : &b == &c->rbuf[0] ? "rbuf0" : &b == &c->rbuf[1] ? "rbuf1" : &b == &c->sbuf[0] ? "sbuf0"
                      : &b == &c->sbuf[1] ? "sbuf1" : nullptr;
  if (c && c->place_tune && pname && want >= c->place_min_bytes) {
    if (!c->place_ev[0]) {
      if (hipEventCreate(&c->place_ev[0]) != hipSuccess || hipEventCreate(&c->place_ev[1]) != hipSuccess) {
        (void)hipGetLastError();
        c->place_tune = false;
      }
    }
    typedef std::chrono::steady_clock clk;
    const auto t_search = clk::now();
    auto ms_since = [](clk::time_point t) { return std::chrono::duration<float, std::milli>(clk::now() - t).count(); };
    // "good": 5.2 TB/s (bytes per ms) -- or, once this context has seen what the box gives, 92 % of the best fill so
    // far if that is less (a box where every allocation writes slowly should not pay for more candidates each time)
    const double good = c->place_best > 0.0 && 0.92 * c->place_best < 5.2e9 ? 0.92 * c->place_best : 5.2e9;
    auto probe = [&](void* p) -> double {
      float ms = 0.f;
      if (hmj::launch_fill_probe(p, want, c->stream) != hipSuccess) return 0.0;  // (first touch)
      if (hipEventRecord(c->place_ev[0], c->stream) != hipSuccess) return 0.0;
      if (hmj::launch_fill_probe(p, want, c->stream) != hipSuccess) return 0.0;
      if (hipEventRecord(c->place_ev[1], c->stream) != hipSuccess) return 0.0;
      if (hipEventSynchronize(c->place_ev[1]) != hipSuccess) return 0.0;
      if (hipEventElapsedTime(&ms, c->place_ev[0], c->place_ev[1]) != hipSuccess || ms <= 0.f) return 0.0;
      return (double)want / (double)ms;
    };
    hmj_place_info pi;
    std::memset(&pi, 0, sizeof(pi));
    std::snprintf(pi.name, sizeof(pi.name), "%s", pname);
    pi.bytes = want;
    pi.budget_ms = c->place_budget_ms;
    pi.cand_ms_alloc[0] = ms_alloc0;
    auto t_c = clk::now();
    double best_rate = c->place_tune ? probe(b.p) : 0.0;
    pi.cand_ms_fill[0] = ms_since(t_c);
    pi.cand_TBps[0] = (float)(best_rate * 1e-9);
    int tried = 1;
    const bool search = c->place_search_always || c->in_reserve;
    for (; search && c->place_tune && tried < c->place_tries && tried < HMJ_PLACE_MAX_CAND && best_rate > 0.0 &&
           best_rate < good; tried++) {
      // budget: what the search has spent so far plus what the previous extra candidate cost (the first extra one is
      // assumed to cost what the first allocation did) must stay inside it
      const float spent = ms_since(t_search);
      const float last = tried == 1 ? ms_alloc0 + pi.cand_ms_fill[0] : pi.cand_ms_alloc[tried - 1] + pi.cand_ms_fill[tried - 1];
      if (spent + last > c->place_budget_ms) {
        pi.aborted = 1;
        break;
      }
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) break;
      if (free_b < want + (8ull << 30)) break;
      t_c = clk::now();
      void* cand = nullptr;
      if (hipMalloc(&cand, want) != hipSuccess) {
        (void)hipGetLastError();
        break;
      }
      pi.cand_ms_alloc[tried] = ms_since(t_c);
      t_c = clk::now();
      const double r = probe(cand);
      pi.cand_TBps[tried] = (float)(r * 1e-9);
      void* loser = cand;
      if (r > best_rate) {
        loser = b.p;
        b.p = cand;
        best_rate = r;
      }
      (void)hipFree(loser);  // (never more than two candidates alive: the kept one and the one being probed)
      pi.cand_ms_fill[tried] = ms_since(t_c);  // fill + the loser's hipFree (device-synchronising)
    }
    if (best_rate > c->place_best) c->place_best = best_rate;
    pi.fill_TBps = (float)(best_rate * 1e-9);
    pi.candidates = tried;
    pi.searched = search ? 1 : 0;
    pi.ms_search = ms_since(t_search);
    bool replaced = false;
    for (hmj_place_info& e : c->place_log)
      if (std::strcmp(e.name, pi.name) == 0) {
        e = pi;
        replaced = true;
      }
    if (!replaced) c->place_log.push_back(pi);
    if (c->trace)
      std::fprintf(stderr, "[hmj]   placement: %s %.1f MiB fill at %.2f TB/s after %d candidate%s (%.1f ms%s%s)\n", pname,
                   (double)want / 1048576.0, best_rate * 1e-9, tried, tried == 1 ? "" : "s", pi.ms_search,
                   search ? "" : ", probe only", pi.aborted ? ", budget reached" : "");
  }
  if (c && c->trace && want >= (64u << 20))
    std::fprintf(stderr, "[hmj] hipMalloc %.1f MiB -> %p (low 30 bits %#llx)\n", (double)want / 1048576.0, b.p,
                 (unsigned long long)((uintptr_t)b.p & ((1ull << 30) - 1)));
  return HMJ_OK;
}

// Host memory is expensive to create, so released buffers go to a process-wide pool and are handed out
// again (smallest fit within 2x).  Two kinds (measured on the MI355X box, tools/micro/alloc_cost*.hip):
//   pinned   (hipHostMalloc): 0.15-0.2 ms per MiB to create -- only the small staging slots use it;
//   pageable on transparent huge pages: the result columns.  A device-to-host copy into FRESH such
//   memory runs at 24 GB/s, into memory used before at 54 GB/s -- the same as into pinned memory --
//   so a one-shot join of 2^26 rows gets its 1.5 GiB of result columns in 64 ms instead of 258 ms.
std::mutex g_pool_mu;
std::vector<HostBuf> g_pool;  // oldest first
size_t g_pool_bytes = 0;
// The pool keeps at most this many bytes; what exceeds it is returned to the OS (oldest buffers first).
// HMJ_HOST_POOL_MAX_MB (read once per process) overrides the default of 8 GiB; hmj_host_pool_trim() empties it.
size_t pool_limit() {
  static const size_t lim = [] {
    const char* e = getenv("HMJ_HOST_POOL_MAX_MB");
    const long long mb = e ? atoll(e) : 8192;
    return (size_t)(mb < 0 ? 0 : mb) << 20;
  }();
  return lim;
}

void host_release(HostBuf& b) {
  if (!b.p) return;
  if (b.pinned)
    (void)hipHostFree(b.p);
  else
    free(b.p);
  b.p = nullptr;
  b.cap = 0;
}

// caller holds g_pool_mu
size_t pool_trim_locked(size_t keep_bytes) {
  size_t released = 0;
  while (g_pool_bytes > keep_bytes && !g_pool.empty()) {
    HostBuf b = g_pool.front();
    g_pool.erase(g_pool.begin());
    g_pool_bytes -= b.cap;
    released += b.cap;
    host_release(b);
  }
  return released;
}

void pool_give(HostBuf& b) {
  if (!b.p) return;
  std::lock_guard<std::mutex> g(g_pool_mu);
  g_pool.push_back(b);
  g_pool_bytes += b.cap;
  b.p = nullptr;
  b.cap = 0;
  pool_trim_locked(pool_limit());
}

bool pool_take(size_t bytes, bool pinned, HostBuf* out) {
  std::lock_guard<std::mutex> g(g_pool_mu);
  int best = -1;
  for (int i = 0; i < (int)g_pool.size(); i++)
    if (g_pool[i].pinned == pinned && g_pool[i].cap >= bytes && g_pool[i].cap <= 2 * bytes + (1u << 20) &&
        (best < 0 || g_pool[i].cap < g_pool[best].cap))
      best = i;
  if (best < 0) return false;
  *out = g_pool[best];
  g_pool.erase(g_pool.begin() + best);
  g_pool_bytes -= out->cap;
  return true;
}

int ensure_host(hmj_ctx* c, HostBuf& b, size_t bytes, bool pinned) {
  if (bytes <= b.cap) return HMJ_OK;
  pool_give(b);
  if (pool_take(bytes, pinned, &b)) return HMJ_OK;
  size_t want = bytes + (bytes >> 4) + 256;
  b.pinned = pinned;
  if (pinned) {
    hipError_t e = hipHostMalloc(&b.p, want, hipHostMallocDefault);
    if (e != hipSuccess) {
      b.p = nullptr;
      (void)hipGetLastError();
      return fail(c, HMJ_E_OOM, "hipHostMalloc", e);
    }
  } else {
    const size_t huge = 2u << 20;
    want = (want + huge - 1) & ~(huge - 1);
    b.p = aligned_alloc(huge, want);
    if (!b.p) return fail(c, HMJ_E_OOM, "host memory for the result columns");
    (void)madvise(b.p, want, MADV_HUGEPAGE);  // advisory: plain pages work too, only slower to fault in
  }
  b.cap = want;
  return HMJ_OK;
}

void free_dev(DevBuf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}
void free_host(HostBuf& b) { pool_give(b); }

// ---- profiling spans ---------------------------------------------------------------------------
int span_begin(hmj_ctx* c, int kind, int rel, int pass) {
  if (!c->profiling || c->ev_used + 2 > (int)c->events.size()) return -1;
  Span s{kind, rel, c->ev_used, c->ev_used + 1, pass};
  c->ev_used += 2;
  (void)hipEventRecord(c->events[s.e0], c->stream);
  c->spans.push_back(s);
  return (int)c->spans.size() - 1;
}
void span_end(hmj_ctx* c, int id) {
  if (id < 0) return;
  (void)hipEventRecord(c->events[c->spans[id].e1], c->stream);
}
void spans_reset(hmj_ctx* c) {
  c->spans.clear();
  c->ev_used = 0;
  std::memset(&c->timing, 0, sizeof(c->timing));
}
void spans_collect(hmj_ctx* c) {  // call after the stream is synchronized
  hmj_timing& t = c->timing;
  for (const Span& s : c->spans) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->events[s.e0], c->events[s.e1]) != hipSuccess) continue;
    switch (s.kind) {
      case K_TOTAL: t.ms_total += ms; break;
      case K_H2D: t.ms_h2d += ms; break;
      case K_D2H: t.ms_d2h += ms; break;
      case K_HIST: t.ms_hist += ms; break;
      case K_SCAN: t.ms_scan += ms; break;
      case K_SCATTER:
        t.ms_scatter += ms;
        t.n_scatter_launches++;
        t.ms_scatter_pass[s.pass ? 1 : 0] += ms;
        break;
      case K_OFFSETS: t.ms_offsets += ms; break;
      case K_PROBE_COUNT: t.ms_probe_count += ms; break;
      case K_OUT_SCAN: t.ms_out_scan += ms; break;
      case K_PROBE_WRITE: t.ms_probe_write += ms; break;
      case K_ORDER: t.ms_order += ms; break;
    }
    if (s.kind == K_HIST || s.kind == K_SCAN || s.kind == K_SCATTER) {
      if (s.rel == 0) t.ms_partition_build += ms;
      if (s.rel == 1) t.ms_partition_probe += ms;
    }
  }
}

// ---- planner -------------------------------------------------------------------------------------
// Partition count: enough most-significant key bits that an average build partition holds
// <= PB_TARGET_AVG rows and so fits the LDS table of probe.hip (the reference's optimal_partition,
// radix_hash.h:38-57, sizes for CPU caches instead).  Passes: LSD order, <= RP_MAX_BITS each.
// The most probe rows ANY of P partitions is expected to hold in a foreign-key join: a partition gets N ~ Poisson(lam)
// build keys (lam = avg_np / f) and about f probe rows for each.  N's upper quantile at 0.05 / P is found by summing
// the Poisson tail (the normal approximation "mean + 5 sigma" is 15 % short at lam = 8: fan-out 256 was planned into
// partitions that did not fit, and the whole join fell back to the two-step ordered form); the rows-per-key noise is
// added in quadrature.
double fk_probe_rows_hi(double avg_np, double f, double P) {
  if (avg_np <= 0.0 || f <= 0.0) return 0.0;
  const double lam = avg_np / f, p_tail = 0.05 / (P > 1.0 ? P : 1.0);
  const double step = lam > 256.0 ? std::floor(std::sqrt(lam) / 8.0) : 1.0;
  double k = std::floor(lam);
  for (const double k_end = lam + 40.0 * std::sqrt(lam) + 100.0; k < k_end; k += step) {
    const double lp = -lam + (k + 1.0) * std::log(lam) - std::lgamma(k + 2.0);  // log P(N = k + 1)
    if (std::exp(lp) / (1.0 - lam / (k + 2.0)) <= p_tail) break;                // (geometric bound on P(N > k))
  }
  const double z = (k - lam) / std::sqrt(lam), from_keys = (k - lam) * f, from_rows = z * std::sqrt(avg_np);
  return avg_np + std::sqrt(from_keys * from_keys + from_rows * from_rows);
}

void plan_bits(u64 n_build, int force, int* total, int* passes, int pass_bits[4]) {
  int B = 0;
  if (force >= 0) {
    B = force;
  } else {
    const u64 avg_max = hmj::PB_TARGET_AVG + hmj::PB_PLAN_SLACK;
    u64 parts = (n_build + avg_max - 1) / avg_max;
    while ((1ull << B) < parts) B++;
  }
  if (B > 27) B = 27;
  int np = (B + hmj::RP_MAX_BITS - 1) / hmj::RP_MAX_BITS;
  for (int i = 0; i < 4; i++) pass_bits[i] = 0;
  // pass 0 handles the LEAST significant of the B bits; earlier passes get the extra bit
  for (int i = 0; i < np; i++) pass_bits[i] = B / np + (i < B % np ? 1 : 0);
  *total = B;
  *passes = np;
}

// Sizes for which the histogram-free slab partitioning pays: both relations have at least slab_min_rows rows
// (2^21: every size that plans two passes; hmj_ctx.h has the measurements).
// small_side_ok (materialising joins): the smaller relation may be as small as 2^18 rows -- a dimension table beside a
// fact table; the plan then follows the probe rows either way, and 32 instead of 48 bytes per row and pass on the big
// side pays (2^20 x 2^26: materialise 2.36 -> 2.04 ms, ordered 3.39 -> 3.15).  Count joins keep the stricter bound: for
// them a small build side means ONE exact pass with big probe partitions, which beats two slab passes (2^20 x 2^26: 1.07
// against 1.47 ms); a small PROBE side beside a big build side is fine for them too (the plan follows the build side).
static inline bool slab_sizes_ok(const hmj_ctx* c, u64 nb, u64 np, bool small_side_ok = false) {
  const u64 big = nb > np ? nb : np, small = nb > np ? np : nb;
  const u64 floor_rows = small_side_ok ? (1u << 18) : (1u << 22);
  const u64 small_min = c->slab_min_rows < floor_rows ? c->slab_min_rows : floor_rows;
  return big >= c->slab_min_rows && small >= small_min;
}

// one stable LSD pass src -> dst
int radix_pass(hmj_ctx* c, const void* src, void* dst, u32 n, int shift, int bits, int rel,
               u64* offsets_out, int pass_index) {
  u32 nblk, rpb;
  const int variant = c->scatter_variant;
  const int tile = hmj::radix_tile_rows(bits, variant);
  hmj::radix_pass_geometry(n, tile, &nblk, &rpb);
  int rc;
  if ((rc = ensure_dev(c, c->hist, (size_t)(1u << bits) * nblk * sizeof(u32))) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->totals, (size_t)hmj::RP_MAXD * sizeof(u32))) != HMJ_OK) return rc;
  int s = span_begin(c, K_HIST, rel);
  HIP_TRY(hmj::launch_radix_hist(src, n, tile, shift, bits, (u32*)c->hist.p, nblk, rpb, c->stream));
  span_end(c, s);
  s = span_begin(c, K_SCAN, rel);
  HIP_TRY(hmj::launch_radix_rowscan((u32*)c->hist.p, nblk, bits, (u32*)c->totals.p, c->stream));
  span_end(c, s);
  s = span_begin(c, K_SCATTER, rel, pass_index);
  HIP_TRY(hmj::launch_radix_scatter(src, dst, n, variant, shift, bits, (const u32*)c->hist.p,
                                    (const u32*)c->totals.p, nblk, rpb, offsets_out, c->stream));
  span_end(c, s);
  c->timing.bytes_hist += 16ull * n;
  c->timing.bytes_scatter += 32ull * n;
  return HMJ_OK;
}

// all passes of one relation; *result = where the partitioned rows ended up
// try_in_place: the key sample found the relation in ascending key order.  If its rows' partition numbers never
// decrease (one pass over the keys: a sixth of what two radix passes move) the relation IS partitioned where it lies
// -- the output of an ordered join fed into the next one, two tables exported in key order -- and no pass runs.
int partition_relation(hmj_ctx* c, const void* in, u32 n, DevBuf buf[2], int top, int passes,
                       const int pass_bits[4], int rel, const void** result, bool try_in_place = false) {
  *result = in;
  if (passes == 0 || n == 0) return HMJ_OK;
  int rc;
  if (try_in_place) {
    int bits = 0;
    for (int i = 0; i < passes; i++) bits += pass_bits[i];
    if ((rc = ensure_dev(c, c->offs64, 8 * sizeof(u64))) != HMJ_OK) return rc;
    if ((rc = ensure_host(c, c->h_accum, 8 * sizeof(u64))) != HMJ_OK) return rc;
    u32* flag = (u32*)((u64*)c->offs64.p + 7);
    u32* hflag = (u32*)((u64*)c->h_accum.p + 7);
    HIP_TRY(hipMemsetAsync(flag, 0, 4, c->stream));
    const int sp = span_begin(c, K_HIST, rel);
    HIP_TRY(hmj::launch_check_partitioned(in, n, top, bits, flag, c->num_cus, c->stream));
    span_end(c, sp);
    HIP_TRY(hipMemcpyAsync(hflag, flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (*hflag == 0) {
      c->timing.path |= HMJ_PATH_PRESORTED;
      return HMJ_OK;  // *result == in
    }
  }
  if ((rc = ensure_dev(c, buf[0], (size_t)n * 16)) != HMJ_OK) return rc;
  if (passes > 1 && (rc = ensure_dev(c, buf[1], (size_t)n * 16)) != HMJ_OK) return rc;
  const void* src = in;
  int shift = top;  // lowest of the partition bits
  for (int i = 0; i < passes; i++) {
    void* dst = buf[i & 1].p;
    if ((rc = radix_pass(c, src, dst, n, shift, pass_bits[i], rel, nullptr, i)) != HMJ_OK) return rc;
    shift += pass_bits[i];
    src = dst;
  }
  *result = src;
  return HMJ_OK;
}

// Pageable host memory -> device.  hipMemcpy from pageable memory is staged by the runtime through
// one thread; here T host threads each copy their chunks into their own pinned double buffer and
// push them over PCIe on their own stream, so the copy is PCIe-bound instead of memcpy-bound.
constexpr size_t kUpChunk = 8u << 20;

int upload_host(hmj_ctx* c, void* dst_dev, const void* src_host, size_t bytes) {
  if (bytes == 0) return HMJ_OK;
  int T = c->host_threads > 0 ? c->host_threads : (int)std::thread::hardware_concurrency();
  if (c->host_threads <= 0 && T > 8) T = 8;
  const size_t nchunks = (bytes + kUpChunk - 1) / kUpChunk;
  if (T > (int)nchunks) T = (int)nchunks;
  // One hipMemcpy from the caller's pageable memory moves 54 GB/s on the MI355X box (2 GiB in 38 ms), as fast
  // as from pinned memory, so that is the default; HMJ_UPLOAD=staged selects the multi-threaded pinned
  // staging below for hosts whose runtime stages pageable copies slowly through one thread.
  if (!c->staged_upload || T <= 1 || bytes < 4 * kUpChunk) {
    HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream));
    return HMJ_OK;
  }
  while ((int)c->up_streams.size() < T) {
    hipStream_t st;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    c->up_streams.push_back(st);
    for (int k = 0; k < 2; k++) {
      hipEvent_t ev;
      HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      c->up_events.push_back(ev);
      HostBuf hb;
      int rc = ensure_host(c, hb, kUpChunk);
      if (rc != HMJ_OK) return rc;
      c->up_slots.push_back(hb);
    }
  }
  HIP_TRY(hipStreamSynchronize(c->stream));  // dst may still be read by earlier work on the ctx stream
  std::vector<int> status(T, 0);
  std::vector<std::thread> th;
  const int dev = c->device;
  for (int t = 0; t < T; t++) {
    th.emplace_back([=, &status]() {
      if (hipSetDevice(dev) != hipSuccess) { status[t] = 1; return; }
      hipStream_t st = c->up_streams[t];
      size_t it = 0;
      for (size_t ci = t; ci < nchunks; ci += T, it++) {
        const int slot = (int)(it & 1);
        hipEvent_t ev = c->up_events[2 * t + slot];
        char* pin = static_cast<char*>(c->up_slots[2 * t + slot].p);
        if (it >= 2 && hipEventSynchronize(ev) != hipSuccess) { status[t] = 1; return; }
        const size_t off = ci * kUpChunk, len = (bytes - off < kUpChunk) ? bytes - off : kUpChunk;
        std::memcpy(pin, static_cast<const char*>(src_host) + off, len);
        if (hipMemcpyAsync(static_cast<char*>(dst_dev) + off, pin, len, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipEventRecord(ev, st) != hipSuccess) { status[t] = 1; return; }
      }
      if (hipStreamSynchronize(st) != hipSuccess) status[t] = 1;
    });
  }
  for (auto& x : th) x.join();
  for (int t = 0; t < T; t++)
    if (status[t]) return fail(c, HMJ_E_HIP, "staged host-to-device copy failed");
  return HMJ_OK;
}

int check_rel(hmj_ctx* c, const void* p, uint64_t n, const char* name) {
  if (n > 0xFFFFFFFFull) return fail(c, HMJ_E_ARG, "more than 2^32-1 rows in one call");
  if (n && !p) return fail(c, HMJ_E_ARG, name);
  if (((uintptr_t)p & 15) != 0) return fail(c, HMJ_E_ARG, "relation pointer must be 16-byte aligned");
  return HMJ_OK;
}

constexpr int kRetryNoWinOrdered = 1003;  // internal: window + sort ordered path cannot index the result -> prefix rule
constexpr int kRetryNoFastWrite = 1002;  // internal: unique-key write mode gave up -> general materialise
constexpr int kRetryNoSlab = 1001;    // internal: the slab path gave up (skew) -> exact path
constexpr int kRetryNoSlabProbe = 1004;  // internal: the probe-side slab partitioning overflowed -> exact path
constexpr int kRetryMoreBits = 1005;  // internal: a partition did not fit the ordered expansion but would with one more radix bit -> re-plan
constexpr int kRetryNoPrefix = 1000;  // internal: ordered join must be re-planned without the sampled prefix

// The ordered epilogue.  order_kernel sorts every partition's rows from the unsorted columns (c->out_*) into
// c->ord_*; a segment too large for its in-LDS paths (hundreds of thousands of rows of a few keys) is only
// copied and flagged, and the whole result is then sorted by (key, rval, sval) with three stable LSD sorts
// over {column value, row index} handles (sval, then rval, then key).  `by_key_only`: the partitions are not
// key ranges (window partitioning of keys with structure) but internally sorted -- one stable sort by key
// finishes the order.  On return *rk/*rr/*rs are the ordered columns.
int order_rows(hmj_ctx* c, const u32* vstart, const u32* in_base32, const u64* in_base64, u32 P, u32 Q, int low,
               u64 n, bool by_key_only, bool many_per_key, const u64** rk, const u64** rr, const u64** rs, int* retry_code) {
  int rc;
  const size_t bytes = (size_t)n * 8;
  if ((rc = ensure_dev(c, c->ord_key, bytes)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->ord_rval, bytes)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->ord_sval, bytes)) != HMJ_OK) return rc;
  const bool can_sort = n <= 0xFFFFFFFFull;  // row indices of the global sorts are 32-bit
  if (by_key_only && !can_sort) {
    *retry_code = kRetryNoWinOrdered;
    return HMJ_OK;
  }
  int sp = span_begin(c, K_ORDER, -1);
  HIP_TRY(hmj::launch_order((const u64*)c->part_out_off.p, vstart, in_base32, in_base64, P, Q, low,
                            (const u64*)c->out_key.p, (const u64*)c->out_rval.p, (const u64*)c->out_sval.p,
                            (u64*)c->ord_key.p, (u64*)c->ord_rval.p, (u64*)c->ord_sval.p, (u64*)c->accum.p,
                            can_sort ? 65536u : 0u, many_per_key, c->num_cus * 4, c->stream));
  span_end(c, sp);
  u64* h = (u64*)c->h_accum.p;
  HIP_TRY(hipMemcpyAsync(&h[hmj::ACC_ERR], (u64*)c->accum.p + hmj::ACC_ERR, 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (h[hmj::ACC_ERR] & hmj::ERR_ORDER_FAIL)
    return fail(c, HMJ_E_UNSUPPORTED, "HMJ_ORDERED: one partition's result exceeds 2^31-1 rows and the whole result 2^32-1 rows");
  const bool deferred = (h[hmj::ACC_ERR] & hmj::ERR_ORDER_DEFER) != 0;
  *rk = (const u64*)c->ord_key.p;
  *rr = (const u64*)c->ord_rval.p;
  *rs = (const u64*)c->ord_sval.p;
  if (!deferred && !by_key_only) return HMJ_OK;
  c->timing.path |= (deferred ? HMJ_PATH_ORDER_DEFERRED : 0u) | (by_key_only ? HMJ_PATH_ORDER_BY_KEY : 0u);
  c->prep.valid = false;  // the partition buffers become the sort's ping-pong pair
  if ((rc = ensure_dev(c, c->rbuf[0], (size_t)n * 16)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->rbuf[1], (size_t)n * 16)) != HMJ_OK) return rc;
  sp = span_begin(c, K_ORDER, -1);
  const u64* cols[3] = {*rs, *rr, *rk};  // least significant sort key first
  int cur = 0;  // rbuf[cur] holds the {column value, row} pairs
  for (int kcol = deferred ? 0 : 2; kcol < 3; kcol++) {
    if (kcol == (deferred ? 0 : 2))
      HIP_TRY(hmj::launch_key_idx(cols[kcol], n, c->rbuf[cur].p, c->stream));
    else
      HIP_TRY(hmj::launch_rekey(c->rbuf[cur].p, n, cols[kcol], c->stream));
    // a pass only for the digits in which some value of the column differs (payloads that are row numbers, build
    // payloads of a few thousand keys: half of the 24 passes and more are copies; as in hmj_sort_u64_device)
    u32 digits = 0xFFu;
    if (n >= (1u << 22)) {
      if ((rc = ensure_dev(c, c->offs64, 8 * sizeof(u64))) != HMJ_OK) return rc;
      const u64 init[3] = {0, ~0ull, 0};
      HIP_TRY(hipMemcpyAsync(c->offs64.p, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
      HIP_TRY(hmj::launch_key_exact(c->rbuf[cur].p, (u32)n, nullptr, 0u, 0, (u64*)c->offs64.p, c->num_cus, c->stream, true));
      HIP_TRY(hipMemcpyAsync(&h[7], c->offs64.p, sizeof(u64), hipMemcpyDeviceToHost, c->stream));  // (slot 7: no accumulator lives there)
      HIP_TRY(hipStreamSynchronize(c->stream));
      digits = 0;
      for (int d = 0; d < 8; d++) digits |= ((h[7] >> (8 * d)) & 0xFFu) ? (1u << d) : 0u;
    }
    for (int d = 0; d < 8; d++) {
      if (!(digits & (1u << d))) continue;
      if ((rc = radix_pass(c, c->rbuf[cur].p, c->rbuf[cur ^ 1].p, (u32)n, 8 * d, 8, -1, nullptr)) != HMJ_OK) return rc;
      cur ^= 1;
    }
  }
  HIP_TRY(hmj::launch_gather3(c->rbuf[cur].p, n, *rr, *rs, (u64*)c->out_key.p, (u64*)c->out_rval.p,
                              (u64*)c->out_sval.p, c->stream));
  span_end(c, sp);
  *rk = (const u64*)c->out_key.p;
  *rr = (const u64*)c->out_rval.p;
  *rs = (const u64*)c->out_sval.p;
  return HMJ_OK;
}

// Ordered join, unique-build-key fast path: ONE probe pass writes each partition's rows into the slots
// of its own probe rows (no count pass), a scan of the per-partition row counts gives the final offsets,
// and the ordered epilogue moves every partition to its place while sorting it.  Returns
// kRetryNoFastWrite when the kernel met duplicate build keys or an oversized partition.
int unique_key_write(hmj_ctx* c, hmj::ProbeArgs& wa, bool slab, u32 nb, u32 np, int low, bool extra,
                         int verify_prefix, u64 pfx_ref, bool ordered, bool to_host, bool fk_wide_plan, double density,
                         hmj_result* out) {
  int rc;
  const u32 P = wa.P;
  const size_t cap_bytes = ((size_t)np + 8) * 8;  // at most one row per probe row
  if ((rc = ensure_dev(c, c->part_count, (size_t)P * 8)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->part_out_off, ((size_t)P + 1) * 8)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->out_key, cap_bytes)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->out_rval, cap_bytes)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->out_sval, cap_bytes)) != HMJ_OK) return rc;
  wa.part_count = (u64*)c->part_count.p;
  wa.out_key = (u64*)c->out_key.p;
  wa.out_rval = (u64*)c->out_rval.p;
  wa.out_sval = (u64*)c->out_sval.p;
  wa.extra = extra ? 1u : 0u;
  if (verify_prefix > 0) {
    wa.pfx_shift = (u32)(64 - verify_prefix);
    wa.pfx_val = pfx_ref >> (64 - verify_prefix);
  }
  const u64* in_base64 = nullptr;
  const u32* in_base32 = nullptr;
  if (slab) {  // first output slot of partition p = number of probe rows in partitions < p
    if ((rc = ensure_dev(c, c->offs64, ((size_t)P * 2 + 2) * 8)) != HMJ_OK) return rc;
    u64* npv = (u64*)c->offs64.p;
    u64* base = npv + P;
    HIP_TRY(hmj::launch_slab_np(wa.s_cnt, P, npv, c->stream));
    HIP_TRY(hmj::launch_scan_u64(npv, base, P, c->stream));
    wa.item_base = base;
    in_base64 = base;
  } else {
    in_base32 = wa.s_off;
  }
  u64* h = (u64*)c->h_accum.p;
  // the result delivered from three dense device columns
  auto deliver = [&](const u64* rk, const u64* rr, const u64* rs) -> int {
    const size_t bytes = (size_t)out->n_matches * 8;
    if (to_host) {
      if ((rc = ensure_host(c, c->h_key, bytes, false)) != HMJ_OK) return rc;
      if ((rc = ensure_host(c, c->h_rval, bytes, false)) != HMJ_OK) return rc;
      if ((rc = ensure_host(c, c->h_sval, bytes, false)) != HMJ_OK) return rc;
      const int s2 = span_begin(c, K_D2H, -1);
      HIP_TRY(hipMemcpyAsync(c->h_key.p, rk, bytes, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipMemcpyAsync(c->h_rval.p, rr, bytes, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipMemcpyAsync(c->h_sval.p, rs, bytes, hipMemcpyDeviceToHost, c->stream));
      span_end(c, s2);
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    out->key = to_host ? (const uint64_t*)c->h_key.p : (const uint64_t*)rk;
    out->rval = to_host ? (const uint64_t*)c->h_rval.p : (const uint64_t*)rr;
    out->sval = to_host ? (const uint64_t*)c->h_sval.p : (const uint64_t*)rs;
    return HMJ_OK;
  };
  // ---- ordered joins: probe, sort and write in one pass (probe_write_sorted_kernel) when the keys allow it
  if (ordered && c->wm->sorted_cooldown > 0) c->wm->sorted_cooldown--;
  // two forms of the kernel: unique probe keys (a match bitmap), and repeating probe keys (a foreign-key join:
  // match counts and a rank by payload inside every key's run).  The context remembers which one the last join
  // needed, and asks the cheaper one again every 64 ordered joins.
  if (ordered && c->wm->sorted_fk && ++c->wm->sorted_fk_age >= 64) c->wm->sorted_fk = false;
  // The foreign-key form has a small shape (512 threads, 3072 probe / 2048 build rows per partition, two workgroups
  // per CU) for the partitions the planner makes for fan-out >= 6: about 2048 probe rows + 5 sigma, sigma =
  // sqrt(fan-out x mean).  A partition beyond it makes the kernel give up ("does not fit"): the big shape then runs.
  bool no_half = !c->sorted_half;
  // a probe side of more than twice the build side cannot have unique keys if its rows match: start with the
  // foreign-key form (the bitmap form would only find out: 1.9 ms at 2^24 x 2^28)
  const bool must_repeat = nb > 0 && (double)np >= 2.0 * (double)nb;
  for (int form = (c->wm->sorted_fk || must_repeat || fk_wide_plan) ? 1 : 0;
       form < 2 && ordered && c->sorted_mode && c->wm->sorted_cooldown == 0; form++) {
    const bool fk = form == 1;
    bool half = false;
    int shape = 0;
    if (fk && nb > 0 && P > 0) {
      // (density: the populated partitions hold that many times the mean -- the dense-build plan)
      const double avg_np = density * (double)np / (double)P, avg_nb = density * (double)nb / (double)P, f = (double)np / (double)nb;
      const double hi_np = fk_probe_rows_hi(avg_np, f, (double)P), hi_nb = avg_nb + 6.0 * __builtin_sqrt(avg_nb) + 8.0;
      half = !no_half && f >= 2.0 && hi_np <= 3072.0 && hi_nb <= 2048.0;
      shape = half ? 1 : (fk_wide_plan && hi_nb <= 2560.0) ? 2 : 0;
    }
    if ((rc = ensure_dev(c, c->lookback, ((size_t)P + 1) * 8)) != HMJ_OK) return rc;
    HIP_TRY(hipMemsetAsync(c->lookback.p, 0, ((size_t)P + 1) * 8, c->stream));
    // long runs (from 24 probe rows per build row on): the kernel's instantiation that buckets the output slots by (build
    // rank, payload position) and ranks inside those buckets instead of inside whole runs (HMJ_FK_PAYLOAD_BUCKETS=0: never)
    wa.extra = (wa.extra & 3u) | ((fk && c->fk_payload_buckets && nb > 0 && (double)np >= (double)c->fk_payload_buckets * (double)nb) ? 4u : 0u);
    int sp = span_begin(c, K_PROBE_WRITE, -1);
    HIP_TRY(hmj::launch_probe_write_sorted(wa, slab, fk, shape, (u64*)c->lookback.p, c->wm->sorted_chained, low, c->num_cus, c->stream));
    span_end(c, sp);
    HIP_TRY(hipMemcpyAsync(h, c->accum.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (h[hmj::ACC_ERR] & hmj::ERR_SLAB) {
      c->wm->slab_cooldown = 8;
      return kRetryNoSlab;
    }
    if (h[hmj::ACC_ERR] & hmj::ERR_PREFIX) return kRetryNoPrefix;
    if (!(h[hmj::ACC_ERR] & hmj::ERR_SORTED)) {
      out->n_matches = h[hmj::ACC_N];
      out->sum_r = h[hmj::ACC_SUM_R];
      out->sum_s = h[hmj::ACC_SUM_S];
      out->xor_fold = h[hmj::ACC_XOR];
      out->mix_sum = h[hmj::ACC_MIX];
      out->sum_probe_all = h[hmj::ACC_SUM_P];
      c->timing.bytes_probe_write = 16ull * ((u64)nb + np) + 24ull * out->n_matches;
      c->timing.path |= HMJ_PATH_UNIQ_WRITE | HMJ_PATH_SORTED_WRITE | (fk ? HMJ_PATH_SORTED_FK : 0u) | (half ? HMJ_PATH_SORTED_FK_HALF : 0u) |
                        (shape == 2 ? HMJ_PATH_SORTED_FK_WIDE : 0u);
      if (out->n_matches == 0) return HMJ_OK;
      const u64 *rk = wa.out_key, *rr = wa.out_rval, *rs = wa.out_sval;
      // the next ordered join: chained output offsets if this one had unmatched probe rows (dense without an
      // epilogue, 5.6 instead of 4.0 + 3.1 ms at 2^28 rows), the probe rows' own slots if it had none
      const bool was_chained = c->wm->sorted_chained;
      if (!c->sorted_chained_forced) c->wm->sorted_chained = out->n_matches != (u64)np;
      if (!was_chained && out->n_matches != (u64)np) {
        // unmatched probe rows left gaps at the end of every partition's slots: the ordered epilogue closes them
        HIP_TRY(hmj::launch_scan_u64((const u64*)c->part_count.p, (u64*)c->part_out_off.p, P, c->stream));
        int retry = 0;
        if ((rc = order_rows(c, nullptr, in_base32, in_base64, P, 1, low, out->n_matches, false, out->n_matches > 2ull * nb,
                             &rk, &rr, &rs, &retry)) != HMJ_OK)
          return rc;
      }
      return deliver(rk, rr, rs);
    }
    // the partitions stay valid: repeating probe keys -> the foreign-key form; anything else (duplicate build keys,
    // clustered keys, a hot key, an oversized partition) -> the two-step form, and do not ask again for a while
    const u64 why = h[hmj::ACC_ERR];
    const bool try_fk = !fk && (why & 256) && !(why & (128 | 512 | 1024 | 2048));
    const bool try_big = fk && half && (why & 512) && !(why & (128 | 1024 | 2048));
    if (try_fk) {
      c->wm->sorted_fk = true;
      c->wm->sorted_fk_age = 0;
    } else if (try_big) {
      no_half = true;  // same form again, one workgroup per CU with the full capacities
      form--;
    } else {
      c->wm->sorted_cooldown = 64;
      form = 2;
    }
    if (fk_wide_plan && !try_big) c->wm->sorted_wide = false;  // the 16-bit plan did not pay off here: plan for 5120 rows from now on
    if (why & 1024) {  // never expected: say so whether or not tracing is on, and leave a mark the tests can see
      c->timing.path |= HMJ_PATH_LOOKBACK_TIMEOUT;
      std::fprintf(stderr, "[hmj] one-pass ordered write: a chained partition waited beyond the spin limit for its predecessor; "
                           "the join falls back to the write + order epilogue (result unaffected)\n");
    }
    if (c->trace)
      std::fprintf(stderr, "[hmj]   one-pass ordered write%s gave up (%s%s%s%s%s) -> %s\n", fk ? " (foreign-key form)" : "",
                   (why & 128) ? "a bucket too long " : "", (why & 256) ? "repeating probe keys " : "",
                   (why & 512) ? "a partition does not fit " : "", (why & 1024) ? "look-back timeout " : "",
                   (why & 2048) ? "duplicate build keys or a hot key" : "", try_fk ? "foreign-key form" : try_big ? "foreign-key form, big shape" : "write + order epilogue");
    std::vector<Span> keep;
    for (const Span& s2 : c->spans)
      if (s2.kind != K_PROBE_WRITE) keep.push_back(s2);
    c->spans.swap(keep);
    HIP_TRY(hipMemsetAsync(c->accum.p, 0, 8 * sizeof(u64), c->stream));
  }
  int sp = span_begin(c, K_PROBE_WRITE, -1);
  HIP_TRY(hmj::launch_probe_write_uniq(wa, slab, c->num_cus, c->stream));
  span_end(c, sp);
  sp = span_begin(c, K_OUT_SCAN, -1);
  HIP_TRY(hmj::launch_scan_u64((const u64*)c->part_count.p, (u64*)c->part_out_off.p, P, c->stream));
  span_end(c, sp);
  HIP_TRY(hipMemcpyAsync(h, c->accum.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (h[hmj::ACC_ERR] & hmj::ERR_SLAB) {
    c->wm->slab_cooldown = 8;
    return kRetryNoSlab;
  }
  if (h[hmj::ACC_ERR] & hmj::ERR_PREFIX) return kRetryNoPrefix;
  if (h[hmj::ACC_ERR] & hmj::ERR_FASTPATH) {
    c->wm->uniq_cooldown = 8;
    return kRetryNoFastWrite;
  }
  out->n_matches = h[hmj::ACC_N];
  out->sum_r = h[hmj::ACC_SUM_R];
  out->sum_s = h[hmj::ACC_SUM_S];
  out->xor_fold = h[hmj::ACC_XOR];
  out->mix_sum = h[hmj::ACC_MIX];
  out->sum_probe_all = h[hmj::ACC_SUM_P];
  c->timing.bytes_probe_write = 16ull * ((u64)nb + np) + 24ull * out->n_matches;
  c->timing.path |= HMJ_PATH_UNIQ_WRITE;
  if (out->n_matches == 0) return HMJ_OK;
  // every probe row matched: the columns have no gaps, an unordered result is complete as it stands.
  // Otherwise the ordered epilogue closes the gaps (and sorts, which an unordered caller may ignore).
  const bool dense_out = !ordered && out->n_matches == (u64)np;
  const u64 *rk = wa.out_key, *rr = wa.out_rval, *rs = wa.out_sval;
  if (!dense_out && ordered) {
    int retry = 0;
    if ((rc = order_rows(c, nullptr, in_base32, in_base64, P, 1, low, out->n_matches, false, out->n_matches > 2ull * nb,
                         &rk, &rr, &rs, &retry)) !=
        HMJ_OK)
      return rc;
  } else if (!dense_out) {
    // an unordered result only has to lose the gaps unmatched probe rows left: every partition's rows move to the
    // partition's dense offset as they are.  (Until round 3 this went through the ordered epilogue, which also sorts:
    // a foreign-key join with fan-out 16 and a fifth of its probe rows unmatched took 16.4 ms instead of 2.)
    const size_t bytes = (size_t)out->n_matches * 8;
    if ((rc = ensure_dev(c, c->ord_key, bytes)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->ord_rval, bytes)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->ord_sval, bytes)) != HMJ_OK) return rc;
    const int sp2 = span_begin(c, K_ORDER, -1);
    HIP_TRY(hmj::launch_compact((const u64*)c->part_out_off.p, in_base32, in_base64, P, rk, rr, rs, (u64*)c->ord_key.p,
                                (u64*)c->ord_rval.p, (u64*)c->ord_sval.p, c->num_cus * 8, c->stream));
    span_end(c, sp2);
    rk = (const u64*)c->ord_key.p;
    rr = (const u64*)c->ord_rval.p;
    rs = (const u64*)c->ord_sval.p;
  }
  return deliver(rk, rr, rs);
}

// ---- one attempt of the partitioned join = plan_join (key sample, radix bits, partition window, which kernels are on
// offer: everything decided before a row moves) + execute_join (partition, build + probe, retries reported as kRetry*).
// What the attempt may try is the caller's (join_device's retry loop); what it decided is a JoinPlan.
struct JoinPlan {
  bool allow_slab, allow_slab_probe;   // still allowed after the sample (a relation in key order takes no slab pass)
  int B, passes, pass_bits[4];         // total radix bits, LSD passes, bits per pass
  bool fk_wide_plan;                   // the bits rely on the wide shape of the one-pass ordered write (foreign-key form)
  double dense_scale;                  // the populated partitions hold this many times the mean (dense-build plan)
  bool asc_r, asc_s;                   // the sample found the build / probe rows in ascending key order
  u32 P, Q;                            // partitions, probe slices per partition
  u64 items;
  int prefix, low;                     // partition id = (key >> low) & (P - 1)
  bool win_ordered, hot_hint, verify_pfx;
  u64 pfx_ref;
  bool probe_fits, fast_write;
  u32 np_plan;                         // probe size the plan was made for
};

static int plan_join(hmj_ctx* c, const void* R, uint64_t n_build, const void* S, uint64_t n_probe, uint32_t flags,
                     bool allow_auto_prefix, bool allow_slab, bool allow_fast_write, bool allow_win_ordered, bool prefix_unsafe,
                     bool allow_slab_probe, JoinPlan* plan_out) {
  int rc;
  const bool materialize = flags & HMJ_MATERIALIZE;
  const u32 nb = (u32)n_build, np = (u32)n_probe;
  const u32 np_plan = c->prepare_only ? (u32)c->probe_hint : np;  // probe size the plan is made for


  int B, passes, pass_bits[4];
  bool fk_wide_plan = false;  // the plan relies on the wide shape of the one-pass ordered write (foreign-key form)
  plan_bits(n_build, c->force_bits, &B, &passes, pass_bits);
  // ---- key sample (one small kernel): the top bits all keys share, a hot-key hint, and the range of the build keys
  u64 smp[8] = {0, 0, 0, 0, ~0ull, 0, 0, 0};
  double dense_scale = 1.0;   // dense-build plan: the populated partitions hold this many times the mean
  bool asc_r = false, asc_s = false;  // the sample found the build / probe rows in ascending key order
  bool prefix_exact = false;  // smp[0] / [4] / [5] come from a pass over ALL keys (the retry after a prefix violation)
  const bool have_sample = c->prefix_bits < 0 && allow_auto_prefix && B > 0 && (u64)nb + np > 0;
  if (have_sample) {
    if ((rc = ensure_dev(c, c->offs64, 8 * sizeof(u64))) != HMJ_OK) return rc;
    if ((rc = ensure_host(c, c->h_accum, 8 * sizeof(u64))) != HMJ_OK) return rc;
    HIP_TRY(hmj::launch_key_sample(R, nb, c->sample_build_only ? nullptr : S, c->sample_build_only ? 0u : np,
                                   (u64*)c->offs64.p, c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_accum.p, c->offs64.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::memcpy(smp, c->h_accum.p, sizeof(smp));
    if (prefix_unsafe && (flags & HMJ_ORDERED)) {
      // The first attempt found rows outside the sampled prefix (keys 0 .. n - 1 with n a little above a power of
      // two: the handful of keys with the top bit set escaped the sample).  One pass over all keys gives the exact
      // set of varying bits and the build keys' exact range; the plan below then sees a key range that is filled
      // only in part (the dense-build plan: more bits) and the partitions stay key ranges.  (Round 2 kept the
      // sampled prefix and finished such a join with a sort of all result rows by key: 2^25 + 5 dense keys, ordered,
      // 7.2 ms against 1.4 ms for uniform keys, tools/exp_cliffs2.py.)
      // (A probe side that was still arriving when the sample ran -- host pipeline, exchange rounds -- has arrived by
      //  now, or will have once its events fire: this IS the retry.  Its keys must be in the exact prefix, or an
      //  ordered join whose probe keys lie outside the build keys' prefix would fail a second time)
      for (hipEvent_t ev : c->arrive_ev)
        if ((rc = wait_arrival(c, ev)) != HMJ_OK) return rc;
      const bool probe_readable = !c->sample_build_only || !c->arrive_ev.empty();
      u64* ex3 = (u64*)c->offs64.p;
      const u64 init[3] = {0, ~0ull, 0};
      HIP_TRY(hipMemcpyAsync(ex3, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
      HIP_TRY(hmj::launch_key_exact(R, nb, probe_readable ? S : nullptr, probe_readable ? np : 0u, smp[1], ex3,
                                    c->num_cus, c->stream));
      HIP_TRY(hipMemcpyAsync(c->h_accum.p, ex3, 3 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      const u64* e3 = (const u64*)c->h_accum.p;
      smp[0] = e3[0];
      if (nb) {
        smp[4] = e3[1];
        smp[5] = e3[2];
      }
      prefix_exact = true;
    }
    {
      // A relation that arrives (nearly) sorted by key gives every worker of the slab pass rows of one digit: its
      // slab for that digit overflows, the join starts over on the exact path and the context avoids slabs for eight
      // joins, then tries again.  Nine tenths of the neighbouring sample pairs in one order: skip the attempt.
      auto sorted_like = [](u64 w) {
        const u64 up = (w >> 16) & 0xFFFF, dn = (w >> 32) & 0xFFFF;
        return up + dn >= 64 && (up * 10 >= (up + dn) * 9 || dn * 10 >= (up + dn) * 9);
      };
      if (sorted_like(smp[2]) || sorted_like(smp[3])) {
        if (allow_slab || allow_slab_probe) c->plan.refused |= HMJ_REFUSED_SLAB_SORTED_INPUT;
        allow_slab = allow_slab_probe = false;
      }
      auto ascending = [](u64 w) {
        const u64 up = (w >> 16) & 0xFFFF, dn = (w >> 32) & 0xFFFF;
        return up + dn >= 64 && up * 10 >= (up + dn) * 9;
      };
      asc_r = ascending(smp[2]);
      asc_s = ascending(smp[3]);
    }
    // Build keys that cover only part of the key range the partition bits span (a dimension table's ids under a
    // fact table with a wider key domain; keys below 2^63 against full 64-bit keys) crowd into a fraction of
    // the partitions: each then holds 1 / fraction times the planned rows, overflows the LDS table and takes the
    // chunked generic kernels (2^26 sorted keys below 2^63 against uniform 64-bit probe keys: count 7.6 ms,
    // ordered 36 ms).  Plan for the density the build side really has instead -- if the sampled build keys fill
    // their range evenly (at least 48 of its 64ths hold a sample, none more than four times its share); keys in a
    // few clusters (a tag, a gap, an id) are the key window's business below.
    const int pfx = smp[0] ? __builtin_clzll(smp[0]) : 64;
    const u64 n_smp = nb < 2048 ? nb : 2048;
    const bool even = smp[6] >= 48 && smp[7] * 64 <= 4 * n_smp + 64;
    // (a caller's minimum prefix hides top bits the keys DO differ in: the build keys' range says nothing about the window below it)
    if (c->dense_plan && c->force_bits < 0 && nb > 0 && smp[5] > smp[4] && pfx < 64 && even && c->min_prefix_bits <= pfx) {
      const double span = (double)(smp[5] - smp[4]) + 1.0, full = __builtin_ldexp(1.0, 64 - pfx);
      const double fraction = span / full;
      // (between 0.7 and 1 the bits stay -- dense ids fill whole partitions of a power-of-two size either way -- but the
      //  populated partitions still hold 1 / fraction times the mean, which the slab capacities, the foreign-key plan and
      //  the kernel shapes must know: 1.5 x 2^24 dense ids overflowed a slab, count 1.20 ms against 0.80 / 1.07 ms for
      //  1.25 x and 1.75 x 2^24)
      if (fraction < 0.97) dense_scale = fraction > 1.0 / 64.0 ? 1.0 / fraction : 64.0;
      if (fraction < 0.7) {
        const double scale = fraction > 1.0 / 64.0 ? 1.0 / fraction : 64.0;
        const double eff = (double)n_build * scale;
        plan_bits(eff > 4.0e9 ? 4000000000ull : (u64)eff, -1, &B, &passes, pass_bits);
        c->timing.path |= HMJ_PATH_DENSE_BUILD;
        dense_scale = scale;
      }
    }
  }
  if (c->force_bits < 0 && np_plan > nb && nb > 0) {
    // A probe side larger than the build side (a foreign-key join with fan-out f): size the partitions by the
    // PROBE rows where that pays.  A partition's probe rows vary with the number of build keys it happens to
    // hold, times f: sigma ~ sqrt(f * avg); spend another bit while avg + 5 sigma would not fit the 5120-row
    // pipelines.
    //  * materialising joins: result rows are written and (ordered) sorted partition by partition, so the
    //    probe rows of a partition must fit the single-pass write mode and the LDS sort (up to 18 bits: still
    //    two passes, the same number as before for these sizes);
    //  * count joins: only when that makes the histogram-free slab partitioning applicable (both relations
    //    large, two passes of at most 8 bits): 2^26 x 2^28 rows 8.3 -> 6.0 ms.  Otherwise the build-side plan
    //    with its big probe partitions is the faster one (one table serves tens of thousands of probe rows).
    int Bp, passes_p, pb_p[4];
    // (keys that fill 1 / dense_scale of the key range the partition bits span -- dimension ids 0 .. n - 1 with n a little
    //  above a power of two -- crowd into that share of the partitions: plan for dense_scale times the rows.  Without
    //  it an ordered join of 2^22 + 9 dense ids with 2^26 foreign keys made 8192-row partitions, had them cut into
    //  virtual partitions and took 40.7 ms instead of 2.3.)
    const u64 np_dense = (u64)((double)np_plan * dense_scale);
    plan_bits(np_dense > 4000000000ull ? 4000000000ull : np_dense, -1, &Bp, &passes_p, pb_p);
    const double f = (double)np_plan / (double)nb;
    // ordered joins with fan-out >= 2.5 (their probe keys MUST repeat): the foreign-key form of the one-pass ordered
    // write has a 6144-row shape, so 16 bits still hold mean + 5 sigma of a 2^28-row probe side at fan-out 16 -- and
    // 16 bits partition on the slab path (32 instead of 48 B per row and pass)
    const bool wide_ok = materialize && (flags & HMJ_ORDERED) && allow_fast_write && !c->prepare_only && c->sorted_mode &&
                         c->wm->sorted_wide && c->wm->sorted_cooldown <= 1 && f >= 2.5;
    bool wide_ok_override = true;
    const int B_narrow = [&] {
      int b = Bp;
      while (b < 18 && fk_probe_rows_hi((double)np_dense / (double)(1ull << b), f, (double)(1ull << b)) > 5120.0) b++;
      return b;
    }();
    // ... and its small shape (512 threads, 3072 probe / 2048 build rows, TWO workgroups per CU, which overlap each
    // other's barrier-separated LDS phases with memory) wants partitions of about half that size.  Since round 4 the
    // slab path partitions 17 and 18 bits too (9-bit passes), so the extra bit costs 0.1-0.2 ms of partitioning instead
    // of a fall to the exact path: 2^24 x 2^28 ordered, wide 16-bit plan 3.79 + 4.27 ms, 17-bit plan + small shape
    // 3.93 + 3.78 ms (profiles/r04c_*).  HMJ_FK_PLAN=wide|half|narrow pins the choice (measurements).
    const bool half_ok = materialize && (flags & HMJ_ORDERED) && allow_fast_write && !c->prepare_only && c->sorted_mode &&
                         c->sorted_half && c->wm->sorted_cooldown <= 1 && f >= 2.0 && c->fk_plan != 1 && c->fk_plan != 3;
    int B_half = Bp;
    while (B_half < 18 && fk_probe_rows_hi((double)np_dense / (double)(1ull << B_half), f, (double)(1ull << B_half)) > 3072.0) B_half++;
    const bool half_fits = fk_probe_rows_hi((double)np_dense / (double)(1ull << B_half), f, (double)(1ull << B_half)) <= 3072.0 &&
                           (double)nb * dense_scale / (double)(1ull << B_half) <= 1536.0;
    // Measured (tools/exp_fk_plans.py, profiles/r04c_side_fk_plans.txt, 2^b x 2^28 ordered, ms: wide or narrow plan | half
    // plan): b = 26 8.00 | 8.17, 25 7.65 | 7.80, 24 8.16 | 8.00, 23 8.78 | 10.76, 22 12.02 | 12.40 -- the second
    // workgroup per CU does not pay for the extra partition bit: the write kernel is bound by its own instructions
    // (run ranking, LDS), not by exposed memory latency.  The half plan therefore stays an option (HMJ_FK_PLAN=half).
    const bool take_half = half_ok && half_fits && c->fk_plan == 2;
    if (c->fk_plan == 3) wide_ok_override = false;
    if (take_half) {
      Bp = B_half;
    } else {
      while (Bp < 18 && fk_probe_rows_hi((double)np_dense / (double)(1ull << Bp), f, (double)(1ull << Bp)) > ((wide_ok && wide_ok_override) ? 6144.0 : 5120.0)) Bp++;
    }
    // count joins on the slab path: the pipelined count kernel takes a partition's rows piece by piece, a quarter of
    // its threads per piece (1280 rows); a key's f probe rows spread over the four pieces (f = 8 at 4096-row partitions
    // overflowed a piece, the join started over with probe-side slabs only: 3.9 ms against 2.7 ms, tools/exp_cliffs_fk.py)
    if (!materialize)
      while (Bp < 16 && fk_probe_rows_hi((double)np_dense / (double)(1ull << Bp) / 4.0, f / 4.0, 4.0 * (double)(1ull << Bp)) > 1280.0) Bp++;
    // (the wide shape where it saves a bit -- or where 18 bits, the most two slab passes make, still leave the fullest
    //  partition beyond the narrow shape's 5120 rows: 2^25 x 2^30 rows were planned narrow at 18 bits, overflowed, and took
    //  the exact path, 107 ms)
    const bool narrow_holds = fk_probe_rows_hi((double)np_dense / (double)(1ull << B_narrow), f, (double)(1ull << B_narrow)) <= 5120.0;
    fk_wide_plan = wide_ok && wide_ok_override && !take_half && (Bp < B_narrow || !narrow_holds);
    const bool slab_ok = allow_slab && c->slab_mode && c->wm->slab_cooldown == 0 && slab_sizes_ok(c, nb, np_plan);
    // (count joins whose build-side plan is ONE pass keep it when the probe side can stay in that pass's slabs -- the
    //  one-pass slab path below: 2^21 x 2^28 rows, 18-bit probe-side plan 7.0 ms, 9-bit build-side plan 2.9 ms)
    const bool one_pass_count = !(flags & HMJ_ORDERED) && (!materialize || c->wm->one_pass_write_cooldown == 0) && passes == 1 && B >= 5 && c->one_pass_slab && allow_slab_probe && c->slab_mode &&
                                c->wm->slab_probe_cooldown == 0 && !c->prepare_only && np_plan >= (1u << 22) && (u64)np_plan >= 8ull * nb;
    if (Bp > B && !one_pass_count &&
        // (count joins: up to 16 bits only.  The 9-bit slab passes would allow 17 and 18, but a probe side that needs them
        //  is >= 3 * 10^8 rows over a much smaller build side -- BASELINE configs[4] -- where partitioning BOTH sides that
        //  finely costs more than the build-side plan with probe-side slabs (20 against 17.4 ms), skewed build keys overflow
        //  the build slabs, and the cool-down of that failed attempt took the next eight joins of the context off the slab
        //  path: a 5 * 10^8-row join after it ran 21.9 instead of 16.3 ms, tools/exp_after_configs4.py)
        ((materialize && !c->prepare_only && Bp <= 18) || (!materialize && Bp <= 16 && slab_ok)))
      plan_bits(0, Bp, &B, &passes, pass_bits);  // the pass split of Bp bits
  }
  const u32 P = 1u << B;
  // probe slices: when there are few partitions, several workgroups share one partition's table
  u32 Q = 1;
  const u32 target_items = (u32)hmj::probe_default_grid(c->num_cus);
  if (P < target_items) {
    u64 by_grid = (target_items + P - 1) / P;
    u64 by_rows = ((u64)np_plan / P + hmj::PB_TARGET_AVG - 1) / hmj::PB_TARGET_AVG;
    Q = (u32)(by_grid < by_rows ? by_grid : by_rows);
    if (Q < 1) Q = 1;
  }
  u64 items = (u64)P * Q;
  c->timing.radix_bits = B;
  c->timing.radix_passes = passes;
  int prefix = c->prefix_bits < 0 ? 0 : c->prefix_bits;
  bool sampled = false, win_ordered = false, hot_hint = false;
  u64 pfx_ref = 0;
  if (have_sample) {
    // dense / small-integer keys: skip the top bits every (sampled) key shares
    const u64* hs = smp;
    prefix = hs[0] ? __builtin_clzll(hs[0]) : 64;
    // (a caller that consumed top key bits itself -- the exchange's digit owner -- says so with a MINIMUM: the sample
    //  still runs for its hot-key / sorted-input / density hints, the partition window starts at or below the hint)
    const bool hinted = c->min_prefix_bits > prefix;
    if (hinted) prefix = c->min_prefix_bits > 64 ? 64 : c->min_prefix_bits;
    pfx_ref = hs[1];
    hot_hint = (hs[2] & 0xFFFF) >= 2 || (hs[3] & 0xFFFF) >= 2;  // neighbouring sample positions with equal keys: a hot key
    if (hot_hint && P >= 16 && Q > 1 && c->split_mode && !c->prepare_only) {
      // uniform probe slices would leave the hot partition to a few workgroups: take one item per partition
      // and let the split step below cut the oversized ones into as many virtual partitions as they need
      Q = 1;
      items = P;
    }
    sampled = prefix > 0 && !hinted;  // (a hinted prefix is not shared by the keys: nothing may verify or rely on it)
    if (prefix + B > 64) prefix = 64 - B;
    if (!c->prepare_only && c->window_mode && (!(flags & HMJ_ORDERED) || allow_win_ordered)) {
      // hs[0] has a 1 wherever two sampled keys differ.  Keys with structure (a tag in the top bits,
      // zeros below it, an id in the low bits) vary in few of the B bits right under the shared prefix;
      // any B-bit window is a valid partition function, so take the highest one that covers the most
      // varying bits.  (Partitions are then no key ranges: an ordered result gets a final sort by key.)
      const u64 M = hs[0], wmask = (B >= 64) ? ~0ull : ((1ull << B) - 1);
      int best_low = 64 - prefix - B, best_pop = __builtin_popcountll((M >> best_low) & wmask);
      for (int l = best_low - 1; l >= 0 && best_pop < B; l--) {
        const int pop = __builtin_popcountll((M >> l) & wmask);
        if (pop > best_pop) {
          best_pop = pop;
          best_low = l;
        }
      }
      if (best_low != 64 - prefix - B) {
        prefix = 64 - B - best_low;  // "prefix" now only positions the window; nothing relies on it
        c->timing.path |= HMJ_PATH_WINDOW;
        sampled = false;
        // an ordered result is then finished by a stable sort of the rows on the whole key (below)
        win_ordered = (flags & HMJ_ORDERED) != 0;
      }
    }
    if (prefix_unsafe && !prefix_exact && (flags & HMJ_ORDERED) && allow_win_ordered && sampled) {
      // a first attempt found rows outside the sampled prefix (a few outliers above an otherwise dense key
      // range): keep the plan -- dropping the prefix would put every row into one partition -- but stop
      // relying on partitions being key ranges: the result gets the final stable sort by key
      sampled = false;
      win_ordered = true;
    }
  }
  if (prefix + B > 64) prefix = 64 - B;
  const int low = 64 - prefix - B;  // partition id = (key >> low) & (P - 1)
  const bool verify_pfx = sampled && prefix > 0 && (flags & HMJ_ORDERED);  // ordered output relies on it
  c->timing.key_prefix_bits = prefix;
  c->timing.key_window_low = low;
  if (hot_hint) c->timing.path |= HMJ_PATH_HOT_KEY_HINT;
  if (((u64)nb >> B) > hmj::PB_CAP) c->timing.path |= HMJ_PATH_CHUNKED_BUILD;
  // materialising joins whose build keys are unique take the unique-key write mode (one probe pass,
  // no count pass); it works on either partition layout
  if (c->wm->uniq_cooldown > 0 && allow_fast_write && materialize) c->wm->uniq_cooldown--;
  // the pipelined probe kernels hold one partition's probe rows in registers (5120 at most)
  const bool probe_fits = ((u64)((double)np_plan * dense_scale) >> B) <= 4608;  // (the wide foreign-key plan keeps the mean at <= 4608 too)
  // A small materialising join (fewer partitions than the probe grid has workgroups) would share every table among
  // several probe slices, which the unique-key write mode cannot take (a partition's rows go out as one piece): such
  // a join keeps Q = 1 while that mode is on offer -- one pass on half the chip beats count + write + order on all of
  // it (ordered, 2^20 + 3000 rows: 0.34 -> 0.22 ms, 2^21 + 3000: 0.45 -> 0.27 ms).  After a failed attempt (duplicate
  // build keys: uniq_cooldown) the slices are back.
  if (Q > 1 && allow_fast_write && c->wm->uniq_cooldown == 0 && materialize && !win_ordered && P >= 2 && probe_fits &&
      !c->prepare_only) {
    Q = 1;
    items = P;
  }
  // (HMJ_FIRST_WINS included: with unique build keys "the first row of a key" is the only one, and the unique-key
  //  kernels give up the moment a probe row meets a key twice -- the general first-wins passes then run as before.
  //  Until round 3 first-wins joins never tried: first + ordered 3.2 ms against 1.4 ms at 2^25 unique keys.)
  // (ordered foreign-key joins with LONG runs -- expand_fk_fanout probe rows per build row and more -- skip the one-pass
  //  ordered write, whose in-run ranking is linear in the run, for count + scan + the ordered expansion, whose sort buckets
  //  cut the runs by payload: probe_expand_ordered_kernel)
  const bool long_runs = (flags & HMJ_ORDERED) && c->expand_mode && c->wm->expand_cooldown == 0 && c->expand_fk_fanout > 0 && nb > 0 &&
                         (double)np >= (double)c->expand_fk_fanout * (double)nb && !(flags & HMJ_FIRST_WINS);
  const bool fast_write = allow_fast_write && c->wm->uniq_cooldown == 0 && materialize && !win_ordered && !long_runs &&
                          Q == 1 && P >= 2 && probe_fits && !c->prepare_only;
  // what this attempt will not even try because an earlier join of the workload gave up on it
  if (!c->prepare_only) {
    if (allow_fast_write && materialize && c->wm->uniq_cooldown > 0) c->plan.refused |= HMJ_REFUSED_FAST_WRITE_COOLING;
    if (allow_slab && c->slab_mode && c->wm->slab_cooldown > 0) c->plan.refused |= HMJ_REFUSED_SLAB_COOLING;
    if (allow_slab_probe && c->slab_mode && c->wm->slab_probe_cooldown > 0) c->plan.refused |= HMJ_REFUSED_SLAB_PROBE_COOLING;
  }
  JoinPlan& p = *plan_out;
  p.allow_slab = allow_slab;
  p.allow_slab_probe = allow_slab_probe;
  p.B = B;
  p.passes = passes;
  for (int i = 0; i < 4; i++) p.pass_bits[i] = i < passes ? pass_bits[i] : 0;
  p.fk_wide_plan = fk_wide_plan;
  p.dense_scale = dense_scale;
  p.asc_r = asc_r;
  p.asc_s = asc_s;
  p.P = P;
  p.Q = Q;
  p.items = items;
  p.prefix = prefix;
  p.low = low;
  p.win_ordered = win_ordered;
  p.hot_hint = hot_hint;
  p.verify_pfx = verify_pfx;
  p.pfx_ref = pfx_ref;
  p.probe_fits = probe_fits;
  p.fast_write = fast_write;
  p.np_plan = np_plan;
  for (int i = 0; i < 4; i++) c->plan.pass_bits[i] = p.pass_bits[i];
  (void)rc;
  return HMJ_OK;
}

static int execute_join(hmj_ctx* c, const void* R, uint64_t n_build, const void* S, uint64_t n_probe, uint32_t flags,
                        hmj_result* out, bool to_host, const JoinPlan& p) {
  int rc;
  const bool materialize = flags & HMJ_MATERIALIZE, first = flags & HMJ_FIRST_WINS;
  const bool extra = flags & (HMJ_CHECKSUM | HMJ_SUM_PROBE);
  const u32 nb = (u32)n_build, np = (u32)n_probe, np_plan = p.np_plan;
  const bool allow_slab = p.allow_slab, allow_slab_probe = p.allow_slab_probe;
  const int B = p.B, passes = p.passes, prefix = p.prefix, low = p.low;
  const int* pass_bits = p.pass_bits;
  const bool fk_wide_plan = p.fk_wide_plan, asc_r = p.asc_r, asc_s = p.asc_s, win_ordered = p.win_ordered, hot_hint = p.hot_hint;
  const bool verify_pfx = p.verify_pfx, probe_fits = p.probe_fits, fast_write = p.fast_write;
  const double dense_scale = p.dense_scale;
  const u32 P = p.P, Q = p.Q;
  const u64 items = p.items, pfx_ref = p.pfx_ref;
  if ((rc = ensure_dev(c, c->r_off, ((size_t)P + 1) * 4)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->s_off, ((size_t)P + 1) * 4)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->accum, 8 * sizeof(u64))) != HMJ_OK) return rc;
  if ((rc = ensure_host(c, c->h_accum, 8 * sizeof(u64))) != HMJ_OK) return rc;
  if (materialize) {
    if ((rc = ensure_dev(c, c->part_count, (size_t)items * 8)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->part_out_off, ((size_t)items + 1) * 8)) != HMJ_OK) return rc;
  }
  HIP_TRY(hipMemsetAsync(c->accum.p, 0, 8 * sizeof(u64), c->stream));
  const void *Rp, *Sp;
  // ---- histogram-free slab path (plain count joins of large, evenly distributed relations)
  hmj::SlabGeom gr, gs;
  if (allow_slab && c->slab_mode && c->wm->slab_cooldown == 0 && !c->prepare_only && !(c->timing.path & HMJ_PATH_SLAB) &&
      !((!materialize || fast_write) && Q == 1 && probe_fits && passes == 2 && pass_bits[0] <= hmj::SLAB_MAX_BITS &&
        pass_bits[1] <= hmj::SLAB_MAX_BITS && dense_scale <= 2.5))
    c->plan.refused |= HMJ_REFUSED_SLAB_SHAPE;
  if (allow_slab && c->slab_mode && c->wm->slab_cooldown == 0 && (!materialize || fast_write) && Q == 1 &&
      probe_fits &&
      passes == 2 && pass_bits[0] <= hmj::SLAB_MAX_BITS && pass_bits[1] <= hmj::SLAB_MAX_BITS &&
      slab_sizes_ok(c, nb, np_plan, materialize || np_plan < nb) &&
      dense_scale <= 2.5 &&  // (beyond: slabs of several times the relation's size; the exact path needs none)
      hmj::slab_geometry(nb, pass_bits[0], pass_bits[1], &gr, 0, 1.0, dense_scale) &&
      hmj::slab_geometry(np_plan, pass_bits[0], pass_bits[1], &gs, 0,
                         (nb && np_plan > nb) ? (double)np_plan / (double)nb : 1.0, dense_scale)) {
    // (a foreign-key probe side repeats every key np / nb times: the rows of a slab piece then vary sqrt(np / nb) times
    //  more, and the slabs are sized for that.  Until round 3 only the wide ordered plan passed the fan-out; count joins
    //  such as 2^22 x 2^26 overflowed a slab at their first attempt, started over with probe-side slabs only (2.2 ms
    //  against 1.4 ms) and tried again every ninth join.)
    const bool reuse = c->prep.valid && c->prep.slab && c->prep.ptr == R && c->prep.n == nb &&
                       c->prep.low == low && c->prep.B == B;
    c->prep.valid = false;  // one-shot; slab_br is about to be (re)written unless reused
    c->timing.path |= HMJ_PATH_SLAB | (reuse ? HMJ_PATH_PREPARED : 0u);
    c->timing.n_probe_items = P;
    const int ba = pass_bits[0], bb = pass_bits[1];
    const u64 rows_a = gr.rows_a > gs.rows_a ? gr.rows_a : gs.rows_a;
    const u32 wa = gr.WA > gs.WA ? gr.WA : gs.WA;
    if ((rc = ensure_dev(c, c->slab_a, rows_a * 16)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->cnt_a, ((size_t)wa << ba) * 4)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->slab_br, gr.rows_b * 16)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->slab_bs, gs.rows_b * 16)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->cnt_br, (size_t)P * gr.KB * 4)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->cnt_bs, (size_t)P * gs.KB * 4)) != HMJ_OK) return rc;
    u64* acc = (u64*)c->accum.p;
    struct { const void* in; u32 n; const hmj::SlabGeom* g; DevBuf* sb; DevBuf* cb; int rel; } side[2] = {
        {R, nb, &gr, &c->slab_br, &c->cnt_br, 0}, {S, np, &gs, &c->slab_bs, &c->cnt_bs, 1}};
    const u64 a_rows = c->slab_a.cap / 16, a_cnt = c->cnt_a.cap / 4;  // what is allocated: the launchers check it
    for (auto& sd : side) {
      if (sd.rel == 0 && reuse) continue;            // build side already in slab_br / cnt_br
      if (sd.rel == 1 && c->prepare_only) continue;  // hmj_prepare_build: build side only
      int sp = span_begin(c, K_SCATTER, sd.rel);
      if (sd.rel == 1 && !c->arrive_ev.empty()) {
        // the probe rows are still arriving over the links (exchange.hip): pass-A workers own contiguous input
        // ranges, so the workers whose rows are complete start after each round's event
        u32 w_done = 0;
        for (size_t i = 0; i < c->arrive_ev.size(); i++) {
          if ((rc = wait_arrival(c, c->arrive_ev[i])) != HMJ_OK) return rc;
          const bool last = i + 1 == c->arrive_ev.size();
          const u32 w_end = last ? sd.g->WA : (u32)std::min<u64>(sd.g->WA, c->arrive_rows[i] / sd.g->rpw);
          if (w_end > w_done) {
            HIP_TRY(hmj::launch_slab_a(sd.in, sd.n, low, ba, *sd.g, c->slab_a.p, a_rows, (u32*)c->cnt_a.p, a_cnt, acc, c->stream, w_done, w_end));
            w_done = w_end;
          }
        }
      } else {
        HIP_TRY(hmj::launch_slab_a(sd.in, sd.n, low, ba, *sd.g, c->slab_a.p, a_rows, (u32*)c->cnt_a.p, a_cnt, acc, c->stream));
      }
      span_end(c, sp);
      sp = span_begin(c, K_SCATTER, sd.rel, 1);
      HIP_TRY(hmj::launch_slab_b(c->slab_a.p, (const u32*)c->cnt_a.p, ba, low + ba, bb, *sd.g, sd.sb->p, sd.sb->cap / 16,
                                 (u32*)sd.cb->p, sd.cb->cap / 4, acc, c->stream));
      span_end(c, sp);
      c->timing.bytes_scatter += 2 * 32ull * sd.n;
    }
    u64* hh = (u64*)c->h_accum.p;
    if (c->prepare_only) {
      HIP_TRY(hipMemcpyAsync(hh, c->accum.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (hh[hmj::ACC_ERR] & hmj::ERR_SLAB) {
        c->wm->slab_cooldown = 8;
        return kRetryNoSlab;
      }
      c->prep.valid = true;
      c->prep.slab = true;
      c->prep.ptr = R;
      c->prep.n = nb;
      c->prep.low = low;
      c->prep.B = B;
      return HMJ_OK;
    }
    if (fast_write) {
      hmj::ProbeArgs wa;
      std::memset(&wa, 0, sizeof(wa));
      wa.R = c->slab_br.p;
      wa.S = c->slab_bs.p;
      wa.r_cnt = (const u32*)c->cnt_br.p;
      wa.s_cnt = (const u32*)c->cnt_bs.p;
      wa.r_cap = gr.CB;
      wa.s_cap = gs.CB;
      wa.r_cnt_n = c->cnt_br.cap / 4;
      wa.s_cnt_n = c->cnt_bs.cap / 4;
      wa.r_rows = c->slab_br.cap / 16;
      wa.s_rows = c->slab_bs.cap / 16;
      wa.P = P;
      wa.Q = 1;
      wa.accum = acc;
      return unique_key_write(c, wa, true, nb, np, low, extra, verify_pfx ? prefix : 0, pfx_ref,
                                  (flags & HMJ_ORDERED) != 0, to_host, fk_wide_plan, dense_scale, out);
    }
    hmj::ProbeArgs sa;
    std::memset(&sa, 0, sizeof(sa));
    sa.R = c->slab_br.p;
    sa.S = c->slab_bs.p;
    sa.r_cnt = (const u32*)c->cnt_br.p;
    sa.s_cnt = (const u32*)c->cnt_bs.p;
    sa.r_cap = gr.CB;
    sa.s_cap = gs.CB;
    sa.r_cnt_n = c->cnt_br.cap / 4;
    sa.s_cnt_n = c->cnt_bs.cap / 4;
    sa.r_rows = c->slab_br.cap / 16;
    sa.s_rows = c->slab_bs.cap / 16;
    sa.P = P;
    sa.Q = 1;
    sa.accum = acc;
#ifdef HMJ_DEV
    sa.debug = c->dev_ablate;
#endif
    int sp = span_begin(c, K_PROBE_COUNT, -1);
    if (first || extra) {
      sa.extra = (extra ? 1u : 0u) | (first ? 2u : 0u);
      HIP_TRY(hmj::launch_probe_count_ext(sa, nullptr, nullptr, true, c->num_cus, c->stream));
    } else {
      HIP_TRY(hmj::launch_probe_count_slab(sa, c->num_cus, c->stream));
    }
    span_end(c, sp);
    c->timing.bytes_probe_count = 16ull * ((u64)nb + np);
    HIP_TRY(hipMemcpyAsync(hh, c->accum.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (hh[hmj::ACC_ERR] & hmj::ERR_SLAB) {  // skewed digits: remember, and take the exact path
      c->wm->slab_cooldown = 8;
      return kRetryNoSlab;
    }
    out->n_matches = hh[hmj::ACC_N];
    out->sum_r = hh[hmj::ACC_SUM_R];
    out->sum_s = hh[hmj::ACC_SUM_S];
    out->xor_fold = hh[hmj::ACC_XOR];
    out->mix_sum = hh[hmj::ACC_MIX];
    out->sum_probe_all = hh[hmj::ACC_SUM_P];
    return HMJ_OK;
  }

  const bool reuse_exact = c->prep.valid && !c->prep.slab && c->prep.ptr == R && c->prep.n == nb &&
                           c->prep.low == low && c->prep.B == B;
  c->prep.valid = false;  // one-shot; rbuf / r_off are about to be (re)written unless reused
  c->timing.path |= HMJ_PATH_EXACT | (reuse_exact ? HMJ_PATH_PREPARED : 0u);
  int s;
  if (reuse_exact) {
    Rp = c->prep.Rp;
  } else {
    if ((rc = partition_relation(c, R, nb, c->rbuf, low, passes, pass_bits, 0, &Rp, asc_r)) != HMJ_OK) return rc;
    s = span_begin(c, K_OFFSETS, -1);
    HIP_TRY(hmj::launch_part_offsets(Rp, nb, low, B, (u32*)c->r_off.p, c->stream));
    span_end(c, s);
  }
  if (c->prepare_only) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->prep.valid = true;
    c->prep.slab = false;
    c->prep.ptr = R;
    c->prep.Rp = Rp;
    c->prep.n = nb;
    c->prep.low = low;
    c->prep.B = B;
    return HMJ_OK;
  }
  for (hipEvent_t ev : c->arrive_ev)
    if ((rc = wait_arrival(c, ev)) != HMJ_OK) return rc;  // probe rows still on the links
  // ---- probe-heavy count joins (BASELINE configs[4]: 2^24 build rows, 2^30 probe rows): the build side is
  // small and was partitioned exactly above; the probe side -- almost all of the bytes -- takes the
  // histogram-free slab partitioning (32 instead of 48 B per row and pass) and the generic kernel reads a
  // partition's KB slab pieces as its KB probe slices.  Skewed probe keys overflow a slab: exact path.
  {
    hmj::SlabGeom gp;
    // 2^bits_a * KB >= 512 pass-B workers (HMJ_SLAB_PROBE_KB: any piece count, for the bounds regression tests)
    const u32 kb = c->slab_probe_kb ? c->slab_probe_kb : (pass_bits[0] < 7 ? (512u >> pass_bits[0]) : 4u);
    if (allow_slab_probe && c->slab_mode && c->wm->slab_probe_cooldown == 0 && !materialize && !probe_fits && Q == 1 && passes == 2 &&
        pass_bits[0] <= hmj::SLAB_MAX_BITS && pass_bits[1] <= hmj::SLAB_MAX_BITS && np >= c->slab_min_rows && (u64)np >= 4ull * nb &&
        hmj::slab_geometry(np, pass_bits[0], pass_bits[1], &gp, kb, (double)np / (double)(nb ? nb : 1))) {  // (a foreign-key
      // probe side repeats every key np / nb times: the slabs are sized for that spread)
      const int ba = pass_bits[0], bb = pass_bits[1];
      if ((rc = ensure_dev(c, c->slab_a, gp.rows_a * 16)) != HMJ_OK) return rc;
      if ((rc = ensure_dev(c, c->cnt_a, ((size_t)gp.WA << ba) * 4)) != HMJ_OK) return rc;
      if ((rc = ensure_dev(c, c->slab_bs, gp.rows_b * 16)) != HMJ_OK) return rc;
      if ((rc = ensure_dev(c, c->cnt_bs, (size_t)P * kb * 4)) != HMJ_OK) return rc;
      u64* acc = (u64*)c->accum.p;
      int sp = span_begin(c, K_SCATTER, 1, 0);
      HIP_TRY(hmj::launch_slab_a(S, np, low, ba, gp, c->slab_a.p, c->slab_a.cap / 16, (u32*)c->cnt_a.p, c->cnt_a.cap / 4, acc, c->stream));
      span_end(c, sp);
      sp = span_begin(c, K_SCATTER, 1, 1);
      HIP_TRY(hmj::launch_slab_b(c->slab_a.p, (const u32*)c->cnt_a.p, ba, low + ba, bb, gp, c->slab_bs.p, c->slab_bs.cap / 16,
                                 (u32*)c->cnt_bs.p, c->cnt_bs.cap / 4, acc, c->stream));
      span_end(c, sp);
      c->timing.bytes_scatter += 2 * 32ull * np;
      c->timing.path |= HMJ_PATH_SLAB_PROBE;
      hmj::ProbeArgs a;
      std::memset(&a, 0, sizeof(a));
      a.R = Rp;
      a.r_off = (const u32*)c->r_off.p;
      a.S = c->slab_bs.p;
      a.s_cnt = (const u32*)c->cnt_bs.p;
      a.s_cap = gp.CB;
      a.s_cnt_n = c->cnt_bs.cap / 4;
      a.s_rows = c->slab_bs.cap / 16;
      a.P = P;
      a.Q = kb;
      a.accum = acc;
      c->timing.n_probe_items = P * kb;
      if (first) {  // one bit per probe row SLOT of the slab layout
        const size_t mb = ((size_t)gp.rows_b / 32 + 1) * 4;
        if ((rc = ensure_dev(c, c->matched, mb)) != HMJ_OK) return rc;
        a.matched = (u32*)c->matched.p;
        HIP_TRY(hipMemsetAsync(c->matched.p, 0, mb, c->stream));
      }
      sp = span_begin(c, K_PROBE_COUNT, -1);
      HIP_TRY(hmj::launch_probe(a, 0, first, extra, hmj::probe_default_grid(c->num_cus), c->stream));
      span_end(c, sp);
      c->timing.bytes_probe_count = 16ull * ((u64)nb + np);
      u64* hh = (u64*)c->h_accum.p;
      HIP_TRY(hipMemcpyAsync(hh, c->accum.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (hh[hmj::ACC_ERR] & hmj::ERR_SLAB) {  // a probe slab overflowed (skewed probe keys)
        c->wm->slab_probe_cooldown = 8;
        return kRetryNoSlabProbe;
      }
      out->n_matches = hh[hmj::ACC_N];
      out->sum_r = hh[hmj::ACC_SUM_R];
      out->sum_s = hh[hmj::ACC_SUM_S];
      out->xor_fold = hh[hmj::ACC_XOR];
      out->mix_sum = hh[hmj::ACC_MIX];
      out->sum_probe_all = hh[hmj::ACC_SUM_P];
      return HMJ_OK;
    }
  }
  // ---- count joins of a mid-size build side (2^18 ... 2^20 rows: ONE radix pass, too big for the global table of
  // gtable.hip) with a probe side several times larger -- a dimension table under a fact table.  The exact path costs the
  // probe side 16 (histogram) + 32 (scatter) + 16 (probe) bytes per row.  Here its one pass is the histogram-free slab
  // pass A (32 B), and the generic kernel reads a partition straight out of the pass's worker-private slabs: partition p =
  // the WA pieces [p][0 .. WA), an item = (partition, a run of pieces), a wave per piece.  48 instead of 64 B per row.
  if (allow_slab_probe && c->slab_mode && c->one_pass_slab && c->wm->slab_probe_cooldown == 0 && !(flags & HMJ_ORDERED) &&
      (!materialize || c->wm->one_pass_write_cooldown == 0) && passes == 1 &&
      B >= 5 && B <= hmj::SLAB_MAX_BITS && np >= (1u << 22) && (u64)np >= 8ull * nb && nb > 0) {
    hmj::SlabGeom g1;
    if (hmj::slab_geometry_one_pass(np, B, dense_scale * (double)nb / (double)P, 512, &g1)) {
      if ((rc = ensure_dev(c, c->slab_a, g1.rows_a * 16)) != HMJ_OK) return rc;
      if ((rc = ensure_dev(c, c->cnt_a, ((size_t)g1.WA << B) * 4)) != HMJ_OK) return rc;
      u64* acc = (u64*)c->accum.p;
      int sp = span_begin(c, K_SCATTER, 1, 0);
      HIP_TRY(hmj::launch_slab_a(S, np, low, B, g1, c->slab_a.p, c->slab_a.cap / 16, (u32*)c->cnt_a.p, c->cnt_a.cap / 4, acc, c->stream));
      span_end(c, sp);
      c->timing.bytes_scatter += 32ull * np;
      c->timing.path |= HMJ_PATH_SLAB_PROBE | HMJ_PATH_SLAB_ONE_PASS;
      // items: about eight per CU; an item walks ppi of its partition's WA pieces
      u32 qi = 2048u / P;
      if (qi < 1) qi = 1;
      if (qi > g1.WA / 16) qi = g1.WA / 16;  // (a wave per piece, 16 waves per workgroup: an item of fewer pieces idles some)
      if (qi < 1) qi = 1;
      const u32 ppi = (g1.WA + qi - 1) / qi;
      qi = (g1.WA + ppi - 1) / ppi;
      hmj::ProbeArgs a;
      std::memset(&a, 0, sizeof(a));
      a.R = Rp;
      a.r_off = (const u32*)c->r_off.p;
      a.S = c->slab_a.p;
      a.s_cnt = (const u32*)c->cnt_a.p;
      a.s_cap = g1.CA;
      a.s_wa = g1.WA;
      a.s_ppi = ppi;
      a.s_cnt_n = c->cnt_a.cap / 4;
      a.s_rows = c->slab_a.cap / 16;
      a.P = P;
      a.Q = qi;
      a.accum = acc;
      c->timing.n_probe_items = P * qi;
      if (first) {  // one bit per probe row SLOT of the slab buffer (used where a build partition needs several tables)
        const size_t mb = ((size_t)g1.rows_a / 32 + 1) * 4;
        if ((rc = ensure_dev(c, c->matched, mb)) != HMJ_OK) return rc;
        a.matched = (u32*)c->matched.p;
        HIP_TRY(hipMemsetAsync(c->matched.p, 0, mb, c->stream));
      }
      if (materialize) {
        // unordered rows: counted AND written in the same walk, behind one output cursor (probe_kernel<3>); the columns hold
        // one row per probe row -- more (duplicate build keys: a probe row expands) raises ERR_FASTPATH, and the
        // count / scan / write passes run instead
        const size_t bytes = (size_t)np * 8;
        if ((rc = ensure_dev(c, c->out_key, bytes)) != HMJ_OK) return rc;
        if ((rc = ensure_dev(c, c->out_rval, bytes)) != HMJ_OK) return rc;
        if ((rc = ensure_dev(c, c->out_sval, bytes)) != HMJ_OK) return rc;
        a.out_key = (u64*)c->out_key.p;
        a.out_rval = (u64*)c->out_rval.p;
        a.out_sval = (u64*)c->out_sval.p;
        a.out_cap = np;
      }
      sp = span_begin(c, materialize ? K_PROBE_WRITE : K_PROBE_COUNT, -1);
      HIP_TRY(hmj::launch_probe(a, materialize ? 3 : 0, first, extra, hmj::probe_default_grid(c->num_cus) * 2, c->stream));
      span_end(c, sp);
      if (materialize)
        c->timing.bytes_probe_write = 16ull * ((u64)nb + np);
      else
        c->timing.bytes_probe_count = 16ull * ((u64)nb + np);
      u64* hh = (u64*)c->h_accum.p;
      HIP_TRY(hipMemcpyAsync(hh, c->accum.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (hh[hmj::ACC_ERR] & hmj::ERR_SLAB) {  // a slab overflowed (skewed probe keys): the exact path, and not again for a while
        c->wm->slab_probe_cooldown = 8;
        return kRetryNoSlabProbe;
      }
      if (hh[hmj::ACC_ERR] & hmj::ERR_FASTPATH) {  // more result rows than probe rows: the general materialising passes
        c->wm->one_pass_write_cooldown = 8;
        return kRetryNoSlabProbe;
      }
      out->n_matches = hh[hmj::ACC_N];
      out->sum_r = hh[hmj::ACC_SUM_R];
      out->sum_s = hh[hmj::ACC_SUM_S];
      out->xor_fold = hh[hmj::ACC_XOR];
      out->mix_sum = hh[hmj::ACC_MIX];
      out->sum_probe_all = hh[hmj::ACC_SUM_P];
      if (materialize && out->n_matches) {
        c->timing.bytes_probe_write += 24ull * out->n_matches;
        const size_t bytes = (size_t)out->n_matches * 8;
        if (to_host) {
          if ((rc = ensure_host(c, c->h_key, bytes, false)) != HMJ_OK) return rc;
          if ((rc = ensure_host(c, c->h_rval, bytes, false)) != HMJ_OK) return rc;
          if ((rc = ensure_host(c, c->h_sval, bytes, false)) != HMJ_OK) return rc;
          const int s2 = span_begin(c, K_D2H, -1);
          HIP_TRY(hipMemcpyAsync(c->h_key.p, c->out_key.p, bytes, hipMemcpyDeviceToHost, c->stream));
          HIP_TRY(hipMemcpyAsync(c->h_rval.p, c->out_rval.p, bytes, hipMemcpyDeviceToHost, c->stream));
          HIP_TRY(hipMemcpyAsync(c->h_sval.p, c->out_sval.p, bytes, hipMemcpyDeviceToHost, c->stream));
          span_end(c, s2);
          HIP_TRY(hipStreamSynchronize(c->stream));
        }
        out->key = (const uint64_t*)(to_host ? c->h_key.p : c->out_key.p);
        out->rval = (const uint64_t*)(to_host ? c->h_rval.p : c->out_rval.p);
        out->sval = (const uint64_t*)(to_host ? c->h_sval.p : c->out_sval.p);
      }
      return HMJ_OK;
    }
  }
  if ((rc = partition_relation(c, S, np, c->sbuf, low, passes, pass_bits, 1, &Sp, asc_s)) != HMJ_OK) return rc;
  s = span_begin(c, K_OFFSETS, -1);
  HIP_TRY(hmj::launch_part_offsets(Sp, np, low, B, (u32*)c->s_off.p, c->stream));
  span_end(c, s);

  // Skewed probe side (a hot foreign key): partitions with far more probe rows than the rest are cut into
  // virtual partitions so that no single workgroup is left with millions of rows (probe.hip,
  // split_*_kernel).  Pi = number of (virtual) partitions the probe kernels iterate over.
  u32 Pi = P;
  const u32 *v_start = nullptr, *vr_beg = nullptr, *vr_end = nullptr, *vs_beg = nullptr, *vs_end = nullptr;
  // (not with HMJ_SUM_PROBE: a probe row would be added once per build slice)
  const bool enumerating = (materialize || extra) && !first && !(flags & HMJ_SUM_PROBE);
  // the step costs one 4-byte read-back: taken from 1024 partitions on (joins of a few ms), and from 16 on when
  // the key sample saw a repeated key (a hot key can mean one partition with most of the rows)
  if (Q == 1 && (P >= 1024 || (hot_hint && P >= 16)) && np > 0 && c->split_mode) {
    // average partition fits the pipelines: cut everything above them into 4096-row slices.  Otherwise (big
    // probe partitions by plan) a slice is 1/2048 of the probe side -- enough slices to fill the chip even when
    // one partition holds most of the rows -- and only partitions of more than four slices are cut.
    u32 slice = probe_fits ? 4096u : np / 2048u;
    if (!probe_fits && slice < 16384u) slice = 16384u;
    const u32 thr = probe_fits ? (fk_wide_plan ? 6144u : 5120u) : 4u * slice;  // (the one-pass ordered write's partition capacity)
    // modes that enumerate every pair (materialise, checksums) and are not first-wins also cut partitions with
    // thousands of copies of a key on the build side: the cross product of a hot key is then written by many
    // workgroups (each build slice meets every probe slice of the partition)
    const u32 build_slice = 4096u;
    // virtual partitions: at most P + np / slice from probe slices; build slices multiply a partition's count,
    // bounded by giving the table room for 8x that (a larger total makes the split be ignored)
    u32 build_thr = enumerating ? 6144u : 0u;
    u32 cap_v = (P + np / slice + 1) * (enumerating ? 8u : 1u);
    if (enumerating && cap_v < (1u << 18)) cap_v = 1u << 18;  // room for the 256-row slices of build-heavy partitions
    u32* hnv = (u32*)c->h_accum.p;
    u32 *d_vstart = nullptr, *d_rb = nullptr, *d_re = nullptr, *d_sb = nullptr, *d_se = nullptr;
    // The count step reports how many virtual partitions the split needs.  If the table is too small it is
    // grown (up to 2^24 entries) and the split repeated; a split that would still not fit falls back to
    // probe-only slices (bounded by P + np / slice) instead of being dropped -- one workgroup enumerating a
    // whole hot cross product is a cliff of minutes.
    for (int attempt = 0; attempt < 4; attempt++) {
      if ((rc = ensure_dev(c, c->vparts, ((size_t)cap_v * 4 + P + 3) * 4)) != HMJ_OK) return rc;
      u32* vp = (u32*)c->vparts.p;
      d_vstart = vp;
      u32* d_nv = vp + P + 1;
      d_rb = vp + P + 2;
      d_re = d_rb + cap_v;
      d_sb = d_re + cap_v;
      d_se = d_sb + cap_v;
      HIP_TRY(hmj::launch_split_parts((const u32*)c->r_off.p, (const u32*)c->s_off.p, P, thr, slice, build_thr, build_slice,
                                      cap_v, d_vstart, d_rb, d_re, d_sb, d_se, d_nv, c->stream));
      HIP_TRY(hipMemcpyAsync(hnv, d_nv, 4, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (*hnv <= cap_v) break;
      c->timing.n_split_retries++;
      if (attempt < 2 && cap_v < (1u << 24)) {
        u64 want = (u64)cap_v * 8;  // per-partition counts are clamped at cap_v + 1: the total is a lower bound
        cap_v = want > (1ull << 24) ? (1u << 24) : (u32)want;
      } else {
        build_thr = 0;  // probe slices only
        cap_v = P + np / slice + 1;
      }
    }
    if (*hnv > P && *hnv <= cap_v) {
      Pi = *hnv;
      v_start = d_vstart;
      vr_beg = d_rb;
      vr_end = d_re;
      vs_beg = d_sb;
      vs_end = d_se;
      if (materialize) {
        if ((rc = ensure_dev(c, c->part_count, (size_t)Pi * 8)) != HMJ_OK) return rc;
        if ((rc = ensure_dev(c, c->part_out_off, ((size_t)Pi + 1) * 8)) != HMJ_OK) return rc;
      }
    }
  }
  const bool split = Pi != P;
  if (split) c->timing.path |= HMJ_PATH_SPLIT;
  c->timing.n_probe_items = split ? Pi : (u32)items;

  if (fast_write && !split) {  // (with split partitions a partition's rows are not in one piece: general passes)
    hmj::ProbeArgs wa;
    std::memset(&wa, 0, sizeof(wa));
    wa.R = Rp;
    wa.r_off = (const u32*)c->r_off.p;
    wa.S = Sp;
    wa.s_off = (const u32*)c->s_off.p;
    wa.P = P;
    wa.Q = 1;
    wa.accum = (u64*)c->accum.p;
    rc = unique_key_write(c, wa, false, nb, np, low, extra, verify_pfx ? prefix : 0, pfx_ref,
                                  (flags & HMJ_ORDERED) != 0, to_host, fk_wide_plan, dense_scale, out);
    if (rc != kRetryNoFastWrite) return rc;
    // duplicate build keys: the partitions stay valid, carry on with the count / scan / write passes
    HIP_TRY(hipMemsetAsync(c->accum.p, 0, 8 * sizeof(u64), c->stream));
  }
  hmj::ProbeArgs a;
  std::memset(&a, 0, sizeof(a));
  a.R = Rp;
  a.r_off = split ? vr_beg : (const u32*)c->r_off.p;
  a.r_end = vr_end;
  a.S = Sp;
  a.s_off = split ? vs_beg : (const u32*)c->s_off.p;
  a.s_end = vs_end;
  a.P = Pi;
  a.Q = Q;
  a.accum = (u64*)c->accum.p;
  if (verify_pfx) {
    a.pfx_shift = (u32)(64 - prefix);
    a.pfx_val = pfx_ref >> (64 - prefix);
  }
#ifdef HMJ_DEV
  a.debug = c->dev_ablate;
#endif
  a.part_count = (u64*)c->part_count.p;
  a.part_out_off = (const u64*)c->part_out_off.p;
  const int grid = hmj::probe_default_grid(c->num_cus);
  const size_t matched_bytes = ((size_t)np / 32 + 1) * 4;
  if (first) {  // bitmap of probe rows already paired (used when a build partition needs chunks)
    if ((rc = ensure_dev(c, c->matched, matched_bytes)) != HMJ_OK) return rc;
    a.matched = (u32*)c->matched.p;
    HIP_TRY(hipMemsetAsync(c->matched.p, 0, matched_bytes, c->stream));
  }

  s = span_begin(c, K_PROBE_COUNT, -1);
  if (!first && !extra && Q == 1 && Pi >= 2) {
    // pipelined count kernel (with per-partition counts when materialising), then the generic
    // kernel over the partitions it set aside
    if ((rc = ensure_dev(c, c->irregular, ((size_t)Pi + 1) * 4)) != HMJ_OK) return rc;
    u32* n_irr = (u32*)c->irregular.p;
    u32* irr = n_irr + 1;
    HIP_TRY(hipMemsetAsync(n_irr, 0, 4, c->stream));
    const bool big = ((u64)nb >> B) > 2300;  // average build partition beyond the 2560-row pipeline
    HIP_TRY(hmj::launch_probe_count_fast(a, irr, n_irr, big, materialize, c->num_cus, c->stream));
    hmj::ProbeArgs a2 = a;
    a2.item_list = irr;
    a2.n_item_list = n_irr;
    HIP_TRY(hmj::launch_probe(a2, materialize ? 1 : 0, false, false, c->num_cus, c->stream));
  } else if (!materialize && Q == 1 && Pi >= 2) {
    // count mode with checksums / first-wins: the pipelined kernel's extended variant, then the generic
    // kernel (same flags) over the partitions it set aside
    if ((rc = ensure_dev(c, c->irregular, ((size_t)Pi + 1) * 4)) != HMJ_OK) return rc;
    u32* n_irr = (u32*)c->irregular.p;
    u32* irr = n_irr + 1;
    HIP_TRY(hipMemsetAsync(n_irr, 0, 4, c->stream));
    hmj::ProbeArgs ax = a;
    ax.extra = (extra ? 1u : 0u) | (first ? 2u : 0u);
    HIP_TRY(hmj::launch_probe_count_ext(ax, irr, n_irr, false, c->num_cus, c->stream));
    hmj::ProbeArgs a2 = a;
    a2.item_list = irr;
    a2.n_item_list = n_irr;
    HIP_TRY(hmj::launch_probe(a2, 0, first, extra, c->num_cus, c->stream));
  } else {
    HIP_TRY(hmj::launch_probe(a, materialize ? 1 : 0, first, extra, grid, c->stream));
  }
  span_end(c, s);
  c->timing.bytes_probe_count = 16ull * ((u64)nb + np);
  if (materialize) {
    s = span_begin(c, K_OUT_SCAN, -1);
    HIP_TRY(hmj::launch_scan_u64((const u64*)c->part_count.p, (u64*)c->part_out_off.p,
                                 split ? Pi : (u32)items, c->stream));
    span_end(c, s);
  }
  u64* h = (u64*)c->h_accum.p;
  HIP_TRY(hipMemcpyAsync(h, c->accum.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (h[hmj::ACC_ERR] & hmj::ERR_PREFIX) return kRetryNoPrefix;
  out->n_matches = h[hmj::ACC_N];
  out->sum_r = h[hmj::ACC_SUM_R];
  out->sum_s = h[hmj::ACC_SUM_S];
  out->xor_fold = h[hmj::ACC_XOR];
  out->mix_sum = h[hmj::ACC_MIX];
  out->sum_probe_all = h[hmj::ACC_SUM_P];

  if (materialize && out->n_matches) {
    const size_t bytes = (size_t)out->n_matches * 8;
    if ((rc = ensure_dev(c, c->out_key, bytes)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->out_rval, bytes)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->out_sval, bytes)) != HMJ_OK) return rc;
    a.out_key = (u64*)c->out_key.p;
    a.out_rval = (u64*)c->out_rval.p;
    a.out_sval = (u64*)c->out_sval.p;
    // Ordered, and the unique-key forms were not taken (duplicate build keys): while the partitions are key ranges and fit
    // the kernel, the result is written IN ORDER partition by partition (probe_expand_ordered_kernel: both sides sorted in
    // LDS, every build row against its key's run of probe rows) instead of written in probe order and sorted afterwards --
    // 8 x 8 rows per key, 1.3 * 10^8 result rows: write 1.9 + order 16.1 ms -> one kernel.
    bool expanded = false;
    // (An unordered result takes it too -- rows in order are rows: 1.15 against 1.9 ms for the general write pass, whose
    //  workgroups re-probe their partition; there the partitions need not be key ranges.)
    if (c->wm->expand_cooldown > 0) c->wm->expand_cooldown--;
    //  Only where the result is several times the input: the kernel runs one workgroup per CU through a dozen barriers per
    //  partition -- at about one result row per input row the general write pass is as fast: 2^28 x 2^28, every 4th build key
    //  doubled: 6.4 against 6.9 ms.)
    else if (c->expand_mode && ((flags & HMJ_ORDERED) ? !win_ordered : out->n_matches >= 2 * ((u64)nb + np)) && !split && Q == 1 &&
             !first && P >= 2) {
      s = span_begin(c, K_PROBE_WRITE, -1);
      HIP_TRY(hmj::launch_probe_expand_ordered(a, low, c->num_cus, c->stream));
      span_end(c, s);
      HIP_TRY(hipMemcpyAsync(&h[hmj::ACC_ERR], (u64*)c->accum.p + hmj::ACC_ERR, 16, hipMemcpyDeviceToHost, c->stream));  // (+ ACC_PAD)
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (h[hmj::ACC_ERR] & hmj::ERR_FASTPATH) {  // a partition beyond the kernel's capacity: write + sort, as before ...
        // ... unless ONE more radix bit would make every partition fit (the biggest is under twice the capacity: keys with
        // many copies make partition sizes vary like copies x keys, not a hot key) and sorting the result rows would cost more
        // than partitioning again (0.12 ns per result row against ~0.06 per input row): the join starts over with B + 1
        if (c->expand_allow_rebits && (flags & HMJ_ORDERED) && c->force_bits < 0 && h[hmj::ACC_PAD] <= 7800 && B + 1 <= 18 &&
            2 * out->n_matches >= (u64)nb + np) {
          c->expand_rebits = B + 1;
          return kRetryMoreBits;
        }
        c->wm->expand_cooldown = 8;
        HIP_TRY(hipMemsetAsync((u64*)c->accum.p + hmj::ACC_ERR, 0, sizeof(u64), c->stream));
        std::vector<Span> keep;
        for (const Span& s2 : c->spans)
          if (s2.kind != K_PROBE_WRITE) keep.push_back(s2);
        c->spans.swap(keep);
      } else {
        expanded = true;
        c->timing.path |= HMJ_PATH_ORDERED_EXPANSION;
        c->timing.bytes_probe_write = 16ull * ((u64)nb + np) + 24ull * out->n_matches;
      }
    }
    if (!expanded) {
    if (first) HIP_TRY(hipMemsetAsync(c->matched.p, 0, matched_bytes, c->stream));
    s = span_begin(c, K_PROBE_WRITE, -1);
    HIP_TRY(hmj::launch_probe(a, 2, first, false, grid, c->stream));
    span_end(c, s);
    c->timing.bytes_probe_write = 16ull * ((u64)nb + np) + 24ull * out->n_matches;
    }
    if ((flags & HMJ_ORDERED) && !expanded) {  // out of place: unsorted columns -> sorted columns (order_rows)
      const u64 *rk = nullptr, *rr = nullptr, *rs = nullptr;
      int retry = 0;
      if ((rc = order_rows(c, v_start, nullptr, nullptr, P, Q, low, out->n_matches, win_ordered,
                           out->n_matches > 2ull * nb, &rk, &rr, &rs, &retry)) != HMJ_OK)
        return rc;
      if (retry) return retry;
      a.out_key = const_cast<u64*>(rk);
      a.out_rval = const_cast<u64*>(rr);
      a.out_sval = const_cast<u64*>(rs);
    }
    if (to_host) {
      if ((rc = ensure_host(c, c->h_key, bytes, false)) != HMJ_OK) return rc;
      if ((rc = ensure_host(c, c->h_rval, bytes, false)) != HMJ_OK) return rc;
      if ((rc = ensure_host(c, c->h_sval, bytes, false)) != HMJ_OK) return rc;
      s = span_begin(c, K_D2H, -1);
      HIP_TRY(hipMemcpyAsync(c->h_key.p, a.out_key, bytes, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipMemcpyAsync(c->h_rval.p, a.out_rval, bytes, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipMemcpyAsync(c->h_sval.p, a.out_sval, bytes, hipMemcpyDeviceToHost, c->stream));
      span_end(c, s);
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    out->key = to_host ? (const uint64_t*)c->h_key.p : (const uint64_t*)a.out_key;
    out->rval = to_host ? (const uint64_t*)c->h_rval.p : (const uint64_t*)a.out_rval;
    out->sval = to_host ? (const uint64_t*)c->h_sval.p : (const uint64_t*)a.out_sval;
  }
  return HMJ_OK;
}

int join_device_impl(hmj_ctx* c, const void* R, uint64_t n_build, const void* S, uint64_t n_probe,
                     uint32_t flags, hmj_result* out, bool to_host, bool allow_auto_prefix,
                     bool allow_slab, bool allow_fast_write, bool allow_win_ordered, bool prefix_unsafe,
                     bool allow_slab_probe) {
  int rc;
  if (!out) return fail(c, HMJ_E_ARG, "out is NULL");
  std::memset(out, 0, sizeof(*out));
  if ((rc = check_rel(c, R, n_build, "build_aos is NULL")) != HMJ_OK) return rc;
  if ((rc = check_rel(c, S, n_probe, "probe_aos is NULL")) != HMJ_OK) return rc;
  if (flags & HMJ_ORDERED) flags |= HMJ_MATERIALIZE;
  // a build side partitioned ahead by hmj_prepare_build pins the partitioning path of the join that uses it (a
  // slab cool-down that ran out in between must not make the join partition the build side a second time)
  if (!c->prepare_only && c->prep.valid && !c->prep.slab && c->prep.ptr == R && c->prep.n == (u32)n_build) allow_slab = false;
  // one decision per join about every cool-down the plan asks about
  const bool materialize = flags & HMJ_MATERIALIZE;
  if (c->wm->slab_cooldown > 0 && allow_slab) c->wm->slab_cooldown--;
  if (c->wm->slab_probe_cooldown > 0 && allow_slab_probe) c->wm->slab_probe_cooldown--;
  if (c->wm->one_pass_write_cooldown > 0 && materialize && allow_slab_probe) c->wm->one_pass_write_cooldown--;
  JoinPlan p;
  if ((rc = plan_join(c, R, n_build, S, n_probe, flags, allow_auto_prefix, allow_slab, allow_fast_write, allow_win_ordered,
                      prefix_unsafe, allow_slab_probe, &p)) != HMJ_OK)
    return rc;
  return execute_join(c, R, n_build, S, n_probe, flags, out, to_host, p);
}

// Small build side, count modes: one global hash table, the probe side streamed once (gtable.hip).  *done = the join
// was answered here.  Not taken (or given up: duplicates / a clustering key set raise ERR_GTABLE, then the context
// skips it for the next 8 joins) -> the partitioned paths run as before.
// When: the table (16 B x 2^k slots, load factor <= 0.5) must stay cache-resident while the probe side streams by, and
// the probe side must be big enough for the saved partitioning pass to matter -- build rows <= c->gtable_max_rows,
// probe rows >= c->gtable_min_fanout x build rows.
int try_global_table(hmj_ctx* c, const void* R, uint64_t n_build, const void* S, uint64_t n_probe, uint32_t flags,
                     hmj_result* out, bool to_host, bool* done) {
  *done = false;
  // (materialising joins: the unordered form only -- an ordered result needs the probe rows in key order, which IS the
  //  partitioning -- and only while no earlier attempt met duplicate build keys: gtable_write_cooldown)
  const bool materialize = (flags & HMJ_MATERIALIZE) != 0;
  c->plan.refused |= HMJ_REFUSED_GTABLE_SHAPE;  // (cleared below once the join is inside the table's window)
  if (!c->gtable_mode || (flags & HMJ_ORDERED) || c->prepare_only || c->force_bits >= 0 ||
      !c->arrive_ev.empty() || n_build == 0 || n_probe > 0xFFFFFFFFull || n_probe < c->gtable_min_probe ||
      // big probe sides: the table must stay in an XCD's L2.  Small joins (a few hundred microseconds of dependent launches on
      // the partitioned paths: sample + read-back, histogram, scan, scatter, offsets, probe) take it up to 2^20 build rows:
      // 2^18 x 2^18 rows 0.124 -> 0.078 ms, tools/exp_gtable.py / profiles/r04h_sweep_build_x_probe_sizes.txt
      // (materialising joins that the one-pass slab walk does not take: up to twice the rows -- what they replace costs more)
      (n_build > (materialize && !(c->one_pass_slab && c->slab_mode && n_probe >= (1ull << 22) && n_probe >= 8 * n_build) ? 2 : 1) * c->gtable_max_rows && !(n_build <= 8 * c->gtable_max_rows && n_build + n_probe <= 16 * c->gtable_max_rows)) ||
      n_probe < (uint64_t)c->gtable_min_fanout * n_build ||
      (c->prep.valid && c->prep.ptr == R && c->prep.n == (u32)n_build))
    return HMJ_OK;
  c->plan.refused &= ~HMJ_REFUSED_GTABLE_SHAPE;
  // (the cool-downs count joins that COULD have taken the table: checked after the eligibility tests)
  if (materialize && c->wm->gtable_write_cooldown > 0) {
    c->wm->gtable_write_cooldown--;
    c->plan.refused |= HMJ_REFUSED_GTABLE_COOLING;
    return HMJ_OK;
  }
  if (c->wm->gtable_cooldown > 0) {
    c->wm->gtable_cooldown--;
    c->plan.refused |= HMJ_REFUSED_GTABLE_COOLING;
    return HMJ_OK;
  }
  int rc;
  if (!out) return fail(c, HMJ_E_ARG, "out is NULL");
  if ((rc = check_rel(c, R, n_build, "build_aos is NULL")) != HMJ_OK) return rc;
  if ((rc = check_rel(c, S, n_probe, "probe_aos is NULL")) != HMJ_OK) return rc;
  int log_cap = 10;
  // slots: 16 per build row while the table stays within 2^18 slots (4 MiB, an XCD's L2), never fewer than 4
  while (((u64)1 << log_cap) < (u64)c->gtable_slots_per_row * n_build) log_cap++;
  while (log_cap > c->gtable_max_log_cap && ((u64)1 << (log_cap - 1)) >= (n_build > c->gtable_max_rows ? 2 : 4) * n_build) log_cap--;  // (small joins of a bigger build side: 2 slots per row)
  const size_t tab_bytes = (size_t)16 << log_cap;
  // tiny build sides, count modes: the table in LDS, one copy per workgroup (gtable.hip, ltable_probe_kernel) -- a lookup in
  // the L2-resident table moves a 128-byte line into L1 per probe row, which bounds that kernel at 0.42 ms per 2^26 probe rows
  // (while its load factor stays low -- a wave walks until its longest walk ends: up to 2048 build rows, 1024 with the checksum
  //  accumulators; 2^10 x 2^26 rows 0.36 -> 0.24 ms, 2^11: 0.41 -> 0.30, 2^12: 0.42 -> 0.57 and not taken)
  const bool lds_table = !materialize && c->ltable_mode && n_probe >= (1u << 16) &&
                         n_build <= (uint64_t)((flags & (HMJ_CHECKSUM | HMJ_SUM_PROBE)) ? hmj::ltable_max_rows() / 4 : hmj::ltable_max_rows() / 2);
  if (!lds_table && (rc = ensure_dev(c, c->gtab, tab_bytes)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->accum, 8 * sizeof(u64))) != HMJ_OK) return rc;
  if ((rc = ensure_host(c, c->h_accum, 8 * sizeof(u64))) != HMJ_OK) return rc;
  const bool first = flags & HMJ_FIRST_WINS, extra = flags & (HMJ_CHECKSUM | HMJ_SUM_PROBE);
  HIP_TRY(hipMemsetAsync(c->accum.p, 0, 8 * sizeof(u64), c->stream));
  int sp = span_begin(c, K_PROBE_COUNT, -1);
  // (a launcher that refuses its operands -- a device without 128 KiB of LDS per workgroup, a table beyond 2^30 slots after
  //  HMJ_GTABLE_MAX_LOG2 was raised -- means "not taken", not a failed join: the partitioned paths run)
  auto not_taken = [&]() {
    (void)hipGetLastError();
    c->wm->gtable_cooldown = 64;
    c->plan.refused |= HMJ_REFUSED_GTABLE_SHAPE;
    std::vector<Span> keep;
    for (const Span& s2 : c->spans)
      if (s2.kind == K_TOTAL || s2.kind == K_H2D) keep.push_back(s2);
    c->spans.swap(keep);
    return HMJ_OK;
  };
  if (lds_table) {
    const hipError_t e = hmj::launch_ltable_probe(R, (u32)n_build, S, (u32)n_probe, (u64*)c->accum.p, first, extra, c->num_cus, c->stream);
    if (e == hipErrorInvalidValue) return not_taken();
    HIP_TRY(e);
  } else {
    HIP_TRY(hipMemsetAsync(c->gtab.p, 0xFF, tab_bytes, c->stream));  // every slot empty (key of all ones)
    const hipError_t e = hmj::launch_gtable_build(R, (u32)n_build, c->gtab.p, log_cap, (u64*)c->accum.p, first, c->num_cus, c->stream);
    if (e == hipErrorInvalidValue) return not_taken();
    HIP_TRY(e);
  }
  if (lds_table) {
  } else if (materialize) {
    const size_t bytes = (size_t)(n_probe ? n_probe : 1) * 8;  // (at most one result row per probe row)
    if ((rc = ensure_dev(c, c->out_key, bytes)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->out_rval, bytes)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->out_sval, bytes)) != HMJ_OK) return rc;
    span_end(c, sp);
    sp = span_begin(c, K_PROBE_WRITE, -1);
    HIP_TRY(hmj::launch_gtable_write(S, (u32)n_probe, c->gtab.p, log_cap, R, (u64*)c->accum.p, (u64*)c->out_key.p,
                                     (u64*)c->out_rval.p, (u64*)c->out_sval.p, first, extra, c->num_cus, c->gtable_wg_per_cu,
                                     c->stream));
  } else {
    HIP_TRY(hmj::launch_gtable_probe(S, (u32)n_probe, c->gtab.p, log_cap, R, (u64*)c->accum.p, first, extra, c->num_cus,
                                     c->gtable_wg_per_cu, c->stream));
  }
  span_end(c, sp);
  u64* hh = (u64*)c->h_accum.p;
  HIP_TRY(hipMemcpyAsync(hh, c->accum.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const bool dup_write = materialize && !first && hh[hmj::ACC_PAD] != 0;  // duplicate build keys: a probe row may expand to several rows
  if (dup_write && !(hh[hmj::ACC_ERR] & hmj::ERR_GTABLE)) c->wm->gtable_write_cooldown = 8;
  if ((hh[hmj::ACC_ERR] & hmj::ERR_GTABLE) || dup_write) {
    if (hh[hmj::ACC_ERR] & hmj::ERR_GTABLE) c->wm->gtable_cooldown = 8;
    c->plan.refused |= HMJ_REFUSED_GTABLE_GAVE_UP;
    std::vector<Span> keep;  // forget the abandoned attempt's span
    for (const Span& s2 : c->spans)
      if (s2.kind == K_TOTAL || s2.kind == K_H2D) keep.push_back(s2);
    c->spans.swap(keep);
    if (c->trace) std::fprintf(stderr, "[hmj] join nb=%llu np=%llu: global table gave up (long walk, reserved key, or duplicate build keys under a materialising join) -> partitioned path\n",
                               (unsigned long long)n_build, (unsigned long long)n_probe);
    return HMJ_OK;
  }
  std::memset(out, 0, sizeof(*out));
  out->n_matches = hh[hmj::ACC_N];
  out->sum_r = hh[hmj::ACC_SUM_R];
  out->sum_s = hh[hmj::ACC_SUM_S];
  out->xor_fold = hh[hmj::ACC_XOR];
  out->mix_sum = hh[hmj::ACC_MIX];
  out->sum_probe_all = hh[hmj::ACC_SUM_P];
  c->timing.path |= HMJ_PATH_GLOBAL_TABLE | (lds_table ? HMJ_PATH_LDS_TABLE : 0u);
  c->timing.radix_bits = 0;
  c->timing.radix_passes = 0;
  c->timing.n_probe_items = 1;
  if (materialize)
    c->timing.bytes_probe_write = 16ull * (n_build + n_probe) + 24ull * out->n_matches;
  else
    c->timing.bytes_probe_count = 16ull * (n_build + n_probe);
  c->prep.valid = false;
  if (materialize && out->n_matches) {
    const size_t bytes = (size_t)out->n_matches * 8;
    if (to_host) {
      if ((rc = ensure_host(c, c->h_key, bytes, false)) != HMJ_OK) return rc;
      if ((rc = ensure_host(c, c->h_rval, bytes, false)) != HMJ_OK) return rc;
      if ((rc = ensure_host(c, c->h_sval, bytes, false)) != HMJ_OK) return rc;
      const int s2 = span_begin(c, K_D2H, -1);
      HIP_TRY(hipMemcpyAsync(c->h_key.p, c->out_key.p, bytes, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipMemcpyAsync(c->h_rval.p, c->out_rval.p, bytes, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipMemcpyAsync(c->h_sval.p, c->out_sval.p, bytes, hipMemcpyDeviceToHost, c->stream));
      span_end(c, s2);
      HIP_TRY(hipStreamSynchronize(c->stream));
    }
    out->key = (const uint64_t*)(to_host ? c->h_key.p : c->out_key.p);
    out->rval = (const uint64_t*)(to_host ? c->h_rval.p : c->out_rval.p);
    out->sval = (const uint64_t*)(to_host ? c->h_sval.p : c->out_sval.p);
  }
  if (c->trace) std::fprintf(stderr, "[hmj] join nb=%llu np=%llu flags=%#x: global table of 2^%d slots\n",
                             (unsigned long long)n_build, (unsigned long long)n_probe, flags, log_cap);
  *done = true;
  return HMJ_OK;
}

// Stable LSD passes over n dense 16-byte rows as a CHAIN of histogram-free slab passes (32 instead of 48 B per row and pass).
// Pass 1 is slab pass A over the dense rows (<= 512 workers); every later pass is slab pass B launched with the geometry of the
// pass before: B reads "the worker-private slabs [digit][worker] of the previous pass" -- worker (dA, k) gathers a run of the
// pieces of previous digit dA -- and writes [digit'][worker'] with worker' = dA * KB + k: the same shape, so the kernel chains
// into itself unchanged, and a pass's stable output order IS the order of its pieces in memory.  Result (*ok): *pc_n pieces of
// *pc_cap rows each in c->slab_a / c->cnt_a (*pc_which == 0) or c->slab_bs / c->cnt_bs (1), in sorted order; the caller walks
// or compacts them.  Slab capacities assume evenly filled digits; dg[i].dens >= 1 says how much fuller than the mean the
// values in use of digit i are (a digit that holds the top bits of a range filled in part) -- it widens pass i's slabs AND pass
// i + 1's, because a pass-B worker belongs to one value of the previous digit and the values in use carry all its rows.  Skew
// the sizes do not cover overflows a slab: *ok = false, the input is untouched, the error word is cleared, the chain's spans
// are dropped.  One read-back.
struct ChainDigit {
  int shift, bits;
  double dens;
};
static int slab_chain(hmj_ctx* c, const void* dense_in, u32 n, const ChainDigit* dg, int nd, int rel, bool* ok_out, u32* pc_n,
                      u32* pc_cap, int* pc_which) {
  *ok_out = false;
  *pc_n = 0;
  int rc;
  if ((rc = ensure_dev(c, c->accum, 8 * sizeof(u64))) != HMJ_OK) return rc;
  if ((rc = ensure_host(c, c->h_accum, 8 * sizeof(u64))) != HMJ_OK) return rc;
  DevBuf* sl[2] = {&c->slab_a, &c->slab_bs};
  DevBuf* cn[2] = {&c->cnt_a, &c->cnt_bs};
  u64* acc = (u64*)c->accum.p;
  u64* hh = (u64*)c->h_accum.p;
  // the chain's verdict is the ERR_SLAB bit of this word: it starts from zero whatever an earlier call (or a fresh
  // allocation: hmj_sort_u64_device on a new context) left there
  HIP_TRY(hipMemsetAsync(acc + hmj::ACC_ERR, 0, sizeof(u64), c->stream));
  u32 W = 0, C = 0;
  int bprev = 0;
  double dens_prev = 1.0;
  bool ok = nd >= 1 && n > 0;
  const u64 bytes_before = c->timing.bytes_scatter;
  for (int i = 0; i < nd && ok; i++) {
    const int bits = dg[i].bits, shift = dg[i].shift;
    const double dens = dg[i].dens < 1.0 ? 1.0 : dg[i].dens;
    const int sp2 = span_begin(c, K_SCATTER, rel, i);
    if (i == 0) {
      hmj::SlabGeom g;
      if (!hmj::slab_geometry_one_pass(n, bits, 1e18, 512, &g, dens)) {
        ok = false;
      } else {
        if ((rc = ensure_dev(c, *sl[0], g.rows_a * 16)) != HMJ_OK) return rc;
        if ((rc = ensure_dev(c, *cn[0], ((size_t)g.WA << bits) * 4)) != HMJ_OK) return rc;
        HIP_TRY(hmj::launch_slab_a(dense_in, n, shift, bits, g, sl[0]->p, sl[0]->cap / 16, (u32*)cn[0]->p, cn[0]->cap / 4, acc, c->stream));
        W = g.WA;
        C = g.CA;
      }
    } else {
      hmj::SlabGeom g;
      std::memset(&g, 0, sizeof(g));
      g.WA = W;
      g.CA = C;
      g.KB = 512u >> bprev;  // 512 workers again, whatever the last digit's width
      if (g.KB < 1) g.KB = 1;
      if (g.KB > W) g.KB = W;
      const u32 Wn = g.KB << bprev;
      g.CB = hmj::slab_capacity(dens * dens_prev * (double)n / ((double)Wn * (double)(1u << bits)), 1.0);
      const u64 pieces = (u64)Wn << bits;
      if (pieces * g.CB >= 0xFFFFFFF0ull) {
        ok = false;
      } else {
        DevBuf& so = *sl[i & 1];
        DevBuf& co = *cn[i & 1];
        if ((rc = ensure_dev(c, so, pieces * g.CB * 16)) != HMJ_OK) return rc;
        if ((rc = ensure_dev(c, co, pieces * 4)) != HMJ_OK) return rc;
        HIP_TRY(hmj::launch_slab_b(sl[(i - 1) & 1]->p, (const u32*)cn[(i - 1) & 1]->p, bprev, shift, bits, g, so.p, so.cap / 16,
                                   (u32*)co.p, co.cap / 4, acc, c->stream));
        W = Wn;
        C = g.CB;
      }
    }
    span_end(c, sp2);
    if (ok) {
      c->timing.bytes_scatter += 32ull * n;
      bprev = bits;
      dens_prev = dens;
      *pc_which = i & 1;
    }
    // keys with many duplicates (a digit's row count then varies like rows-per-key x its key count) or digits that follow
    // from one another show in the first two passes: look at the error word there instead of running a long chain to its end
    if (ok && i == 1 && nd > 3) {
      HIP_TRY(hipMemcpyAsync(hh, (u64*)c->accum.p + hmj::ACC_ERR, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (hh[0] & hmj::ERR_SLAB) ok = false;
    }
  }
  if (ok) {
    HIP_TRY(hipMemcpyAsync(hh, c->accum.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (hh[hmj::ACC_ERR] & hmj::ERR_SLAB) ok = false;
  }
  if (ok) {
    *pc_n = W << bprev;
    *pc_cap = C;
  } else {
    HIP_TRY(hipMemsetAsync((u64*)c->accum.p + hmj::ACC_ERR, 0, sizeof(u64), c->stream));
    std::vector<Span> keep;
    for (const Span& s2 : c->spans)
      if (!(s2.kind == K_SCATTER && s2.rel == rel)) keep.push_back(s2);
    c->spans.swap(keep);
    c->timing.bytes_scatter = bytes_before;
  }
  *ok_out = ok;
  return HMJ_OK;
}

// Ordered result of a SMALL build side under a LONG probe side (fan-outs in the hundreds and thousands: a dimension table of a
// few thousand rows under a fact table), unique build keys.  The operator's order (key, rval, sval) is then the probe rows
// sorted by (rank of their key among the sorted build keys, sval), and both fit one 64-bit sort key when the probe payloads
// span few enough bits (row ids, timestamps: rank_bits + bits(max - min) <= 64): sort the build side, global table key ->
// rank, one composite per matching probe row, LSD radix passes over the composite's low bits only, expand (gtable.hip).
// The partitioned paths rank every probe row inside its key's run, linear in the run length, and plan 18 bits for these
// shapes: 2^16 x 2^26 rows 14-22 ms, 2^10 x 2^22 6 ms.  Anything else -- duplicate build keys, payloads too wide, a table
// that gives up -- leaves *done false and the partitioned paths run (8 joins of cool-down).
// ns per probe row of the partitioned one-pass ordered foreign-key write at fan-out f (profiles/r04u_side_fk_payload_buckets.txt)
static double ordered_part_ns(const OrderedCostModel& m, double f, double n_probe) {
  if (f > 700.0) return m.part_epilogue_ns;  // (runs beyond the kernel's partitions: write + order epilogue)
  // ... and so are a few keys too many in one partition of the finest plan there is (18 bits): 2^19 x 2^28 rows -- two keys
  // of 512 probe rows per partition on average, twelve in the fullest of 2^18 -- overflowed the write's 6144-row shape,
  // the exact path took over with the order epilogue: 137 ms, against 19 ms on the composites and 15 ms for 2^20 x 2^28
  // (profiles/r05s_*).  The plan's own estimate of the fullest partition (fk_probe_rows_hi) says when.
  const double P18 = (double)(1u << (2 * hmj::SLAB_MAX_BITS));
  if (fk_probe_rows_hi(n_probe / P18, f, P18) > 6144.0) return m.part_epilogue_ns;
  const double lin = m.part_ns + m.part_ns_per_f * f;
  if (f < 24.0) return lin;
  const double bucketed = m.part_bucket_ns + (f > 128.0 ? m.part_bucket_ns_per_f * (f - 128.0) : 0.0);
  return lin < bucketed ? lin : bucketed;
}
// The rank-run form (gtable.hip): every partition of probe rows must fit one workgroup's LDS sort with room for its
// spread (mean + 8 sigma of a run of a uniform foreign key), both digits of the partition number must be slab passes, and
// the partitions must be long enough for a workgroup each to pay.  A partition is a rank's whole run up to fan-outs of
// ~1700 (*tb = 0); beyond, a run is cut into 2^tb partitions by the position of the payload in the payloads' range.
static bool rank_runs_fit(const hmj_ctx* c, uint64_t n_build, uint64_t n_probe, int* tb, int* level) {
  *tb = 0;
  *level = 0;
  if (!c->rank_runs_mode || !c->slab_mode || n_build < 4 || n_probe > 0xFFFFFFFFull ||
      n_probe < (1u << 16))  // (a slab pass wants a few dozen tiles of rows)
    return false;
  const int rank_bits = 64 - __builtin_clzll(n_build - 1), max_bits = 2 * hmj::SLAB_MAX_BITS;
  if (rank_bits < 2) return false;
  const double f0 = (double)n_probe / (double)n_build;
  if (f0 < 16.0) return false;
  auto fits = [](double m, int lv) { return m + 8.0 * std::sqrt(m) + 24.0 <= (double)hmj::rank_sort_max_run(lv); };
  if (rank_bits > max_bits) {
    // more ranks than two slab passes tell apart: 2^gb consecutive ranks to a partition (*tb = -gb), sorted there by (rank's
    // low bits, payload) as one word -- where 2^gb runs still fit one workgroup's sort
    const int gb = rank_bits - max_bits;
    if (gb > c->rank_runs_max_group) return false;
    const double m = f0 * (double)(1u << gb);
    int lv = c->rank_runs_wave ? -1 : 0;
    while (lv <= c->rank_runs_max_level && !fits(m, lv)) lv++;
    if (lv > c->rank_runs_max_level) return false;
    *tb = -gb;
    *level = lv;
    return true;
  }
  // the smallest workgroup shape (the fastest: most sorters per CU) whose cut stays within two slab passes; the wave
  // shape (level -1: 512 rows) only for whole runs -- cutting a run further to reach it would only add partitions
  if (c->rank_runs_wave && fits(f0, -1)) {
    *level = -1;
    return true;
  }
  for (int lv = 0; lv <= c->rank_runs_max_level; lv++) {
    double f = f0;
    int t = 0;
    while (!fits(f, lv)) {
      f *= 0.5;
      t++;
    }
    if (rank_bits + t > max_bits || t > c->rank_runs_max_cut) continue;
    if (t > 0) {
      // Payloads that grow with the row's position (row ids, timestamps): a worker of pass A reads `places` chunks from all
      // over the probe side (radix.hip, slab_a_body's strided form), each inside ONE piece; the fullest of its slabs must
      // still be within a slab's capacity (slab_geometry: mean + 8 sqrt(mean) + 24), or the attempt is known to overflow.
      const int TB = rank_bits + t, ba = TB - TB / 2;
      const double tile = ba > 8 ? 4096.0 : 2048.0, tiles = std::ceil((double)n_probe / tile), tpw = std::ceil(tiles / 2048.0);
      const double rpw = tpw * tile, places = rpw / (double)hmj::RANK_PASS_CHUNK_ROWS, pieces = (double)(1u << t);
      const double skew = places >= pieces ? std::ceil(places / pieces) / (places / pieces) : pieces / places;
      const double mean = rpw / (double)(1u << ba);
      if (skew * mean + 4.0 * std::sqrt(skew * mean) > mean + 8.0 * std::sqrt(mean) + 24.0) return false;
    }
    *tb = t;
    *level = lv;
    return true;
  }
  return false;
}
// (what a workload's memo says on top: the form rests after it gave up; cut runs exist only with the lookup inside pass A)
static bool rank_runs_rested(const hmj_ctx* c, int tb) {
  return c->wm->rank_runs_cooldown == 0 && (tb <= 0 || c->wm->rank_lookup_cooldown == 0);
}

int try_small_build_ordered(hmj_ctx* c, const void* R, uint64_t n_build, const void* S, uint64_t n_probe, uint32_t flags,
                            hmj_result* out, bool to_host, bool* done) {
  *done = false;
  if (!c->gtable_mode || !c->gtable_sort_mode || !(flags & HMJ_ORDERED) || c->prepare_only || c->force_bits >= 0 ||
      !c->arrive_ev.empty() || n_build == 0 || n_build > 16 * c->gtable_max_rows || n_probe > 0xFFFFFFFFull ||
      n_probe < (uint64_t)c->gtable_sort_fanout * n_build ||
      (c->prep.valid && c->prep.ptr == R && c->prep.n == (u32)n_build))
    return HMJ_OK;
  {
    // Which is faster is a matter of fan-out AND size (profiles/r04i_*, r04l_*: both paths over a grid of sizes).  This path:
    // ~0.65 ms of dependent launches and read-backs + 0.06-0.09 ns per probe row (below).  The
    // partitioned one-pass ordered write: ~0.15 ms + (0.030 + 0.00023 f) ns per row up to fan-out ~200, 0.00041 f ns beyond
    // (the plan changes), and from ~700 rows per key on it declines (runs beyond its capacity) and the write + order
    // epilogue takes over at 0.25 ns per row and more.  HMJ_GTABLE_SORT_FANOUT=1 (the experiments) skips the model.
    const double f = (double)n_probe / (double)n_build, rows = (double)n_probe * 1e-6;
    // (later in round 4 the one-pass write learned to rank inside (build rank, payload position) buckets from fan-out 24 on:
    //  0.043 ns per row up to fan-out ~130, + 0.0001 per further probe row per key -- 2^18 x 2^26: 7.5 -> 3.9 ms, 2^20 x 2^28:
    //  29.3 -> 15.8 -- so this path is now for fan-outs beyond ~500 and for runs the kernel's partitions cannot hold)
    const OrderedCostModel& m = c->ordered_model;
    const double part_ns = ordered_part_ns(m, f, (double)n_probe);
    // (round 4, later: 0.060 where the composites' passes are the chain of slab passes below, 0.072 on exact passes, + 0.02
    //  where the table leaves the L2 -- profiles/r04p_side_rank_sort_slab_chain.txt)
    const bool chain = c->gtable_sort_slab && c->slab_mode && c->wm->gtable_sort_slab_cooldown == 0 && n_probe >= c->gtable_sort_slab_min;
    // (round 5: runs that fit one workgroup's LDS sort -- fan-out up to ~1700 -- are partitioned by rank with two slab passes
    //  and sorted run by run, rank_runs_fit, whatever the payloads' width;
    //  measured over 2^12 ... 2^18 build x 2^22 ... 2^28 probe rows, profiles/r05k_sweep_ordered_small_build.txt: 0.25 ms of
    //  launches and read-backs + 0.0215 ns per probe row + 7.5 ns per RUN -- a workgroup's load -> sort -> store chain per key.
    //  Longer runs are cut into 2^tb pieces by the position of the payload in the payloads' range.  Row ids and timestamps
    //  are monotone in the row's POSITION, so a worker of a slab pass that reads one contiguous range of the probe side sees
    //  one piece only and its slab for that digit overflows, whichever digit the piece bits go into: the pass with the
    //  lookup reads 4 KiB chunks from all over the relation instead (radix.hip, RankXform::kStrided); + one pass over the
    //  payloads for their range)
    int run_tb = 0, run_level = 0;
    const bool runs = rank_runs_fit(c, n_build, n_probe, &run_tb, &run_level) && rank_runs_rested(c, run_tb);
    // (2^20 ... 2^21 build rows: the rank-run form only -- there for the joins whose fullest 18-bit partition outgrows the
    //  one-pass ordered write: 2^21 x 2^29 rows took 166 ms on the exact path with the order epilogue, profiles/r05y_*)
    const double P18 = (double)(1u << (2 * hmj::SLAB_MAX_BITS));
    if (n_build > 8 * c->gtable_max_rows && (!runs || fk_probe_rows_hi((double)n_probe / P18, f, P18) <= 6144.0)) return HMJ_OK;
    const double rank_ns = (chain ? m.comp_ns_chain : m.comp_ns_exact) + (n_build > c->gtable_max_rows ? m.comp_ns_beyond_l2 : 0.0);
    const double n_parts = run_tb >= 0 ? (double)(n_build << run_tb) : (double)(n_build >> -run_tb);
    const double lb = std::log2((double)n_build) - 14.0;
    const bool wave = run_level < 0;
    const double rank_ms = runs ? m.runs_fixed_ms + (m.runs_ns + (wave ? m.runs_wave_ns : 0.0) + (run_tb > 0 ? m.runs_range_ns : 0.0) +
                                                    (lb > 0.0 ? lb * m.runs_ns_per_log2_build : 0.0)) * rows +
                                      (wave ? m.runs_wave_ns_per_run : m.runs_ns_per_run) * n_parts * 1e-6
                                : m.comp_fixed_ms + rank_ns * rows;
    if (c->gtable_sort_fanout > 1 && rank_ms >= m.part_fixed_ms + part_ns * rows) {
      c->plan.refused |= HMJ_REFUSED_RANK_SORT_MODEL;
      return HMJ_OK;
    }
  }
  if (c->wm->gtable_sort_cooldown > 0) {
    c->wm->gtable_sort_cooldown--;
    c->plan.refused |= HMJ_REFUSED_RANK_SORT_COOLING;
    return HMJ_OK;
  }
  int rc;
  if (!out) return fail(c, HMJ_E_ARG, "out is NULL");
  if ((rc = check_rel(c, R, n_build, "build_aos is NULL")) != HMJ_OK) return rc;
  if ((rc = check_rel(c, S, n_probe, "probe_aos is NULL")) != HMJ_OK) return rc;
  const u32 nb = (u32)n_build, np = (u32)n_probe;
  const bool extra = flags & (HMJ_CHECKSUM | HMJ_SUM_PROBE);
  c->prep.valid = false;  // rbuf is the build side's sort buffer
  if ((rc = ensure_dev(c, c->rbuf[0], (size_t)nb * 16)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->rbuf[1], (size_t)nb * 16)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->sbuf[0], (size_t)np * 16)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->sbuf[1], (size_t)np * 16)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->accum, 8 * sizeof(u64))) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->offs64, 8 * sizeof(u64))) != HMJ_OK) return rc;
  if ((rc = ensure_host(c, c->h_accum, 8 * sizeof(u64))) != HMJ_OK) return rc;
  u64* hh = (u64*)c->h_accum.p;
  auto give_up = [&](const char* why) {
    c->wm->gtable_sort_cooldown = 8;
    c->plan.refused |= HMJ_REFUSED_RANK_SORT_GAVE_UP;
    std::vector<Span> keep;  // forget the abandoned attempt's spans
    for (const Span& s2 : c->spans)
      if (s2.kind == K_TOTAL || s2.kind == K_H2D) keep.push_back(s2);
    c->spans.swap(keep);
    c->timing.path = 0;  // (only what this attempt set: its path bits and pass bytes)
    c->timing.bytes_scatter = c->timing.bytes_hist = 0;
    c->timing.n_scatter_launches = 0;
    if (c->trace) std::fprintf(stderr, "[hmj] join nb=%u np=%u ordered: sort-by-rank path gave up (%s) -> partitioned path\n", nb, np, why);
    return HMJ_OK;
  };
  // ---- 1. which 8-bit digits the build keys differ in, and -- for the composite form -- the range of the probe payloads
  // (one read-back)
  int run_tb = 0, run_level = 0;
  bool use_runs = rank_runs_fit(c, n_build, n_probe, &run_tb, &run_level);
  if (use_runs && !rank_runs_rested(c, run_tb)) {
    if (c->wm->rank_runs_cooldown > 0)
      c->wm->rank_runs_cooldown--;
    else
      c->wm->rank_lookup_cooldown--;
    use_runs = false;
  }
  bool have_range = !use_runs;  // (the exact range; cut runs take a sample's: rows outside it go to the end pieces)
  const u32 range_every = (use_runs && run_tb > 0 && np >= (1u << 18)) ? ((np >> 17) | 1u) : 1u;
  {
    const u64 init[5] = {0, ~0ull, 0, ~0ull, 0};
    HIP_TRY(hipMemcpyAsync(c->offs64.p, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hmj::launch_key_exact(R, nb, nullptr, 0u, 0, (u64*)c->offs64.p, c->num_cus, c->stream, true));
    if (have_range || (use_runs && run_tb > 0)) HIP_TRY(hmj::launch_sval_range(S, np, (u64*)c->offs64.p + 3, c->num_cus, c->stream, range_every));
    HIP_TRY(hipMemcpyAsync(hh, c->offs64.p, 5 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  const u64 key_diff = hh[0];
  const bool any_range = have_range || (use_runs && run_tb > 0);
  u64 svmin = (np && any_range) ? hh[3] : 0, svmax = (np && any_range) ? hh[4] : 0;
  have_range = have_range || (any_range && range_every == 1);
  int range_bits = svmax > svmin ? 64 - __builtin_clzll(svmax - svmin) : 0;
  const int rank_bits = nb > 1 ? 32 - __builtin_clz(nb - 1) : 0;
  // rank and payload in ONE word where they fit; else as two (payloads that are hashes, doubles, pointers): sorted by the
  // payload's varying digits first, then stably by the rank -- up to 8 + 3 passes instead of 4-5, still well under the
  // partitioned paths' run ranking at these fan-outs (a wide-payload join needs twice the fan-out to take this path)
  bool wide = rank_bits + range_bits > 64;
  if (wide && !use_runs && c->gtable_sort_fanout > 1) {  // (ten passes instead of four or five: about 1.6 x the time per row; the rank-run form does not care)
    const double f = (double)n_probe / (double)n_build, rows = (double)n_probe * 1e-6;
    const OrderedCostModel& m = c->ordered_model;
    const double part_ns = ordered_part_ns(m, f, (double)n_probe);
    if (m.comp_fixed_ms + m.comp_ns_wide * rows >= m.part_fixed_ms + part_ns * rows) {
      c->wm->gtable_sort_cooldown = 8;
      return HMJ_OK;
    }
  }
  if (nb > 1 && key_diff == 0) return give_up("duplicate build keys");
  // ---- 2. the build side in key order.  MSD: ONE exact pass on the top varying key bits -- partitions of ~512 rows -- and a
  // stable LDS sort of every partition on the rest (gtable.hip, sort_runs_write_kernel on dense partitions): 4 launches
  // instead of the 24 of eight LSD passes, which were 0.3 of the 2.4 ms of a 2^16 x 2^26-row join (profiles/r05u_small16_ord_*).
  // Keys crowded into few partitions or clustered inside one raise ERR_FASTPATH (one 8-byte read-back): the LSD passes run.
  const void* sortedR = R;
  bool build_sorted = false;
  if (c->build_sort_msd && key_diff != 0 && nb >= 2) {
    const int hb = 63 - __builtin_clzll(key_diff);
    int h = 0;
    while (h < hmj::RP_MAX_BITS && ((u64)nb >> h) > 512) h++;
    if (h > hb + 1) h = hb + 1;
    const double mean = (double)nb / (double)(1u << h);
    int level = 0;
    while (level < 2 && mean + 8.0 * std::sqrt(mean) + 24.0 > (double)hmj::rank_sort_max_run(level)) level++;
    if (mean + 8.0 * std::sqrt(mean) + 24.0 <= (double)hmj::rank_sort_max_run(level)) {
      const u32 P = 1u << h;
      if ((rc = ensure_dev(c, c->msd_off, ((size_t)P + 1) * 8)) != HMJ_OK) return rc;
      HIP_TRY(hipMemsetAsync(c->accum.p, 0, 8 * sizeof(u64), c->stream));
      const void* parted = R;
      if (h > 0) {
        if ((rc = radix_pass(c, R, c->rbuf[0].p, nb, hb + 1 - h, h, 0, (u64*)c->msd_off.p, 0)) != HMJ_OK) return rc;
        parted = c->rbuf[0].p;
      } else {
        const u64 off01[2] = {0, nb};
        HIP_TRY(hipMemcpyAsync(c->msd_off.p, off01, sizeof(off01), hipMemcpyHostToDevice, c->stream));
      }
      const int so = span_begin(c, K_ORDER, -1);
      HIP_TRY(hmj::launch_sort_runs_write(parted, nullptr, 0, P, (const u64*)c->msd_off.p, c->rbuf[1].p, (u64*)c->accum.p, level, c->num_cus,
                                          c->stream));
      span_end(c, so);
      HIP_TRY(hipMemcpyAsync(hh, (u64*)c->accum.p + hmj::ACC_ERR, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (!(hh[0] & hmj::ERR_FASTPATH)) {
        sortedR = c->rbuf[1].p;
        build_sorted = true;
      }
    }
  }
  if (!build_sorted) {
    u32 digits = 0;
    for (int i = 0; i < 8; i++) digits |= ((key_diff >> (8 * i)) & 0xFFu) ? (1u << i) : 0u;
    int k = 0;
    for (int d = 0; d < 8; d++) {
      if (!(digits & (1u << d))) continue;
      void* dst = c->rbuf[k & 1].p;
      if ((rc = radix_pass(c, sortedR, dst, nb, 8 * d, 8, 0, nullptr, k ? 1 : 0)) != HMJ_OK) return rc;
      sortedR = dst;
      k++;
    }
  }
  // ---- 3. key -> rank
  int log_cap = 10;
  while (((u64)1 << log_cap) < (u64)c->gtable_slots_per_row * nb) log_cap++;
  while (log_cap > c->gtable_max_log_cap && ((u64)1 << (log_cap - 1)) >= (nb > c->gtable_max_rows ? 2 : 4) * (u64)nb) log_cap--;
  const size_t tab_bytes = (size_t)16 << log_cap;
  if ((rc = ensure_dev(c, c->gtab, tab_bytes)) != HMJ_OK) return rc;
  HIP_TRY(hipMemsetAsync(c->accum.p, 0, 8 * sizeof(u64), c->stream));
  HIP_TRY(hipMemsetAsync(c->gtab.p, 0xFF, tab_bytes, c->stream));
  int sp = span_begin(c, K_PROBE_COUNT, -1);
  HIP_TRY(hmj::launch_gtable_build(sortedR, nb, c->gtab.p, log_cap, (u64*)c->accum.p, true, c->num_cus, c->stream));
  // ---- 4a. the rank-run form: {rank, sval} rows, two slab passes on the rank's digits, every rank's run sorted in LDS and
  // written at its offset (gtable.hip).  First with the rank lookup INSIDE pass A (radix.hip, radix_slab_a_rank_kernel: the
  // probe rows are read once, nothing is emitted in between); a probe row without its build row cannot be dropped from the
  // middle of a tile there, so that attempt gives way to emit + pass A.  A slab that overflows or a run beyond the kernel
  // (a hot foreign key) leaves the table as it is; the composite form below starts over from the emit, and the workload
  // skips what gave up for its next 8 joins.
  bool runs_done = false;
  u64 n = 0;
  if (use_runs) {
    // the partition number: rank << tb | piece of the run (hmj_dev.h, rank_run_bucket); tb < 0: rank >> -tb
    const int TB = rank_bits + run_tb, gb = run_tb < 0 ? -run_tb : 0, cut = run_tb > 0 ? run_tb : 0;
    const int bb = TB / 2, ba = TB - bb;  // LSD: pass A on the low digit, pass B on the high one
    const u32 P = 1u << TB;
    const int pre = range_bits > 32 ? range_bits - 32 : 0;
    const u64 mult = cut ? ((u64)1 << (cut + 32)) / (((svmax - svmin) >> pre) + 1) : 0;
    u64* acc = (u64*)c->accum.p;
    const char* why = "";
    // rows -> partitions (fused: straight from the probe rows) -> offsets -> sorted runs; 0 ok, 1 gave up, < 0 error
    auto passes = [&](u64 rows, bool fused) -> int {
      hmj::SlabGeom g;
      if (rows == 0 || !hmj::slab_geometry((u32)rows, ba, bb, &g, 0, 1.0, (double)(1u << rank_bits) / (double)nb)) {
        why = "no slab geometry";
        return 1;
      }
      int r2;
      if ((r2 = ensure_dev(c, c->slab_a, g.rows_a * 16)) != HMJ_OK) return r2;
      if ((r2 = ensure_dev(c, c->cnt_a, ((size_t)g.WA << ba) * 4)) != HMJ_OK) return r2;
      if ((r2 = ensure_dev(c, c->slab_bs, g.rows_b * 16)) != HMJ_OK) return r2;
      if ((r2 = ensure_dev(c, c->cnt_bs, (size_t)P * g.KB * 4)) != HMJ_OK) return r2;
      if ((r2 = ensure_dev(c, c->part_out_off, ((size_t)P + 1) * 8)) != HMJ_OK) return r2;
      if ((r2 = ensure_dev(c, c->part_count, ((size_t)P / 1024 + 1) * 8)) != HMJ_OK) return r2;  // (chunk totals of the offsets scan)
      const size_t bytes = (size_t)rows * 8;
      if ((r2 = ensure_dev(c, c->out_key, bytes)) != HMJ_OK) return r2;
      if ((r2 = ensure_dev(c, c->out_rval, bytes)) != HMJ_OK) return r2;
      if ((r2 = ensure_dev(c, c->out_sval, bytes)) != HMJ_OK) return r2;
      auto launch = [&](hipError_t e, const char* what) { return e == hipSuccess ? HMJ_OK : fail(c, HMJ_E_HIP, what, e); };
      int s2 = span_begin(c, K_SCATTER, 1, 0);
      if (fused)
        r2 = launch(hmj::launch_slab_a_ranks(S, np, gb, ba, g, c->slab_a.p, c->slab_a.cap / 16, (u32*)c->cnt_a.p, c->cnt_a.cap / 4, acc, c->gtab.p,
                                             log_cap, extra, cut, svmin, svmax - svmin, pre, mult, c->stream), "launch_slab_a_ranks");
      else
        r2 = launch(hmj::launch_slab_a(c->sbuf[0].p, (u32)rows, gb, ba, g, c->slab_a.p, c->slab_a.cap / 16, (u32*)c->cnt_a.p, c->cnt_a.cap / 4,
                                       acc, c->stream), "launch_slab_a");
      span_end(c, s2);
      if (r2 != HMJ_OK) return r2;
      s2 = span_begin(c, K_SCATTER, 1, 1);
      r2 = launch(hmj::launch_slab_b(c->slab_a.p, (const u32*)c->cnt_a.p, ba, gb + ba, bb, g, c->slab_bs.p, c->slab_bs.cap / 16, (u32*)c->cnt_bs.p,
                                     c->cnt_bs.cap / 4, acc, c->stream), "launch_slab_b");
      span_end(c, s2);
      if (r2 != HMJ_OK) return r2;
      s2 = span_begin(c, K_OUT_SCAN, -1);
      r2 = launch(hmj::launch_slab_offsets((const u32*)c->cnt_bs.p, P, (u64*)c->part_out_off.p, (u64*)c->part_count.p, c->stream), "launch_slab_offsets");
      span_end(c, s2);
      if (r2 != HMJ_OK) return r2;
      s2 = span_begin(c, K_PROBE_WRITE, -1);
      r2 = launch(hmj::launch_rank_sort_write(c->slab_bs.p, (const u32*)c->cnt_bs.p, g.CB, P, (const u64*)c->part_out_off.p, sortedR, nb,
                                              run_tb, (u64*)c->out_key.p, (u64*)c->out_rval.p, (u64*)c->out_sval.p, acc, extra, run_level, c->num_cus, c->stream),
                  "launch_rank_sort_write");
      span_end(c, s2);
      if (r2 != HMJ_OK) return r2;
      HIP_TRY(hipMemcpyAsync(hh, c->accum.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (hh[hmj::ACC_ERR] & hmj::ERR_GTABLE) return 2;
      if (hh[hmj::ACC_PAD] != 0) return 3;
      if (hh[hmj::ACC_ERR] & (hmj::ERR_SLAB | hmj::ERR_FASTPATH)) {
        why = (hh[hmj::ACC_ERR] & hmj::ERR_SLAB) ? "a slab overflowed" : fused ? "a probe row without its build row, or a run beyond the kernel" : "a run beyond the kernel";
        return 1;
      }
      c->timing.bytes_scatter += 2 * 32ull * rows - (fused ? 16ull * rows : 0ull);
      c->timing.path |= HMJ_PATH_SLAB | HMJ_PATH_RANK_RUNS | (fused ? HMJ_PATH_RANK_LOOKUP_IN_PASS : 0u);
      c->timing.radix_bits = TB;
      c->timing.radix_passes = 2;
      return 0;
    };
    auto drop_attempt = [&]() {
      std::vector<Span> keep;
      for (const Span& s3 : c->spans)
        if (s3.kind == K_TOTAL || s3.kind == K_H2D) keep.push_back(s3);
      c->spans.swap(keep);
      c->timing.bytes_scatter = 0;
    };
    int st = 1;
    if (c->wm->rank_lookup_cooldown > 0) {
      c->wm->rank_lookup_cooldown--;
    } else {
      span_end(c, sp);  // (the table build)
      st = passes(np, true);
      if (st < 0) return st;
      if (st == 2) return give_up("the table gave up");
      if (st == 3) return give_up("duplicate build keys");
      if (st == 1) {
        c->wm->rank_lookup_cooldown = 8;
        if (c->trace) std::fprintf(stderr, "[hmj] join nb=%u np=%u ordered: rank lookup inside pass A gave up (%s) -> emit + pass A\n", nb, np, why);
        drop_attempt();
        // (the table stays; the accumulators start over, with the build kernel's verdict -- no duplicates, no give-up -- known)
        HIP_TRY(hipMemsetAsync(c->accum.p, 0, 8 * sizeof(u64), c->stream));
        sp = span_begin(c, K_PROBE_COUNT, -1);
      } else {
        n = hh[hmj::ACC_N];
        runs_done = true;
      }
    }
    if (!runs_done && run_tb <= 0) {  // (cut runs: the emit below writes ranks, not partition numbers -- on to the composites)
      HIP_TRY(hmj::launch_gtable_emit_ranks(S, np, c->gtab.p, log_cap, (u64*)c->accum.p, c->sbuf[0].p, extra, c->num_cus, c->gtable_wg_per_cu, c->stream));
      span_end(c, sp);
      HIP_TRY(hipMemcpyAsync(hh, c->accum.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (hh[hmj::ACC_ERR] & hmj::ERR_GTABLE) return give_up("the table gave up");
      if (hh[hmj::ACC_PAD] != 0) return give_up("duplicate build keys");
      n = hh[hmj::ACC_N];
      if (n == 0) {
        runs_done = true;  // no probe row matched: an empty result
      } else {
        st = passes(n, false);
        if (st < 0) return st;
        runs_done = st == 0;
      }
    }
    if (!runs_done) {
      // start over on the composite form: the table stays, the accumulators and the emit are redone with the payload range
      c->wm->rank_runs_cooldown = 8;
      if (c->trace) std::fprintf(stderr, "[hmj] join nb=%u np=%u ordered: rank-run form gave up (%s) -> composite sort\n", nb, np, why);
      drop_attempt();
      use_runs = false;
      if (!have_range) {
        const u64 init2[2] = {~0ull, 0};
        HIP_TRY(hipMemcpyAsync((u64*)c->offs64.p + 3, init2, sizeof(init2), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hmj::launch_sval_range(S, np, (u64*)c->offs64.p + 3, c->num_cus, c->stream));
        HIP_TRY(hipMemcpyAsync(hh, (u64*)c->offs64.p + 3, 2 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        svmin = np ? hh[0] : 0;
        svmax = np ? hh[1] : 0;
        range_bits = svmax > svmin ? 64 - __builtin_clzll(svmax - svmin) : 0;
        wide = rank_bits + range_bits > 64;
        have_range = true;
      }
      HIP_TRY(hipMemsetAsync(c->accum.p, 0, 8 * sizeof(u64), c->stream));
      sp = span_begin(c, K_PROBE_COUNT, -1);
    }
  }
  int which = 0, n_passes = 0;  // `sorted` lies in sbuf[which]
  u32 pc_n = 0, pc_cap = 0;  // the sorted composites as pieces (pc_n != 0): c->slab_* / c->cnt_* [pc_which]
  int pc_which = 0;
  const void* sorted = c->sbuf[0].p;
  if (!runs_done) {
  // ---- 4. one composite per matching probe row
  HIP_TRY(hmj::launch_gtable_emit(S, np, c->gtab.p, log_cap, svmin, range_bits, (u64*)c->accum.p, c->sbuf[0].p, extra, wide,
                                  c->num_cus, c->gtable_wg_per_cu, c->stream));
  span_end(c, sp);
  HIP_TRY(hipMemcpyAsync(hh, c->accum.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (hh[hmj::ACC_ERR] & hmj::ERR_GTABLE) return give_up("the table gave up");
  if (hh[hmj::ACC_PAD] != 0) return give_up("duplicate build keys");
  n = hh[hmj::ACC_N];
  // ---- 5. sort the composites: only their rank_bits + range_bits low bits differ
  auto lsd = [&](int total) -> int {  // stable LSD passes over the key word's bits [0, total), 9-bit digits at most
    if (total <= 0 || n <= 1) return HMJ_OK;
    const int passes = (total + hmj::RP_MAX_BITS - 1) / hmj::RP_MAX_BITS;
    int shift = 0;
    for (int i = 0; i < passes; i++) {
      const int bits = total / passes + (i < total % passes ? 1 : 0);
      void* dst = c->sbuf[which ^ 1].p;
      const int r2 = radix_pass(c, sorted, dst, (u32)n, shift, bits, 1, nullptr, n_passes ? 1 : 0);
      if (r2 != HMJ_OK) return r2;
      sorted = dst;
      which ^= 1;
      shift += bits;
      n_passes++;
    }
    return HMJ_OK;
  };
  // The narrow form's passes as a chain of histogram-free slab passes (slab_chain above): 32 instead of 48 B per row and pass,
  // the last pass's pieces expanded in place (gtable_expand_pieces_kernel).  The digits that hold the payload range's and the
  // rank's top bits are sized for how much of their range is in use.  An overflow (skewed payload bits, a hot key): the exact
  // passes run from the dense composites, which the chain has not touched, and the chain is left alone for the next 8 such joins.
  if (c->wm->gtable_sort_slab_cooldown > 0 && !wide) c->wm->gtable_sort_slab_cooldown--;
  else if (!wide && c->slab_mode && c->gtable_sort_slab && n >= c->gtable_sort_slab_min && rank_bits + range_bits > 0) {
    const int total = rank_bits + range_bits;
    const int passes = (total + hmj::RP_MAX_BITS - 1) / hmj::RP_MAX_BITS;
    const double dens_range = range_bits ? std::ldexp(1.0, range_bits) / ((double)(svmax - svmin) + 1.0) : 1.0;
    const double dens_rank = rank_bits ? std::ldexp(1.0, rank_bits) / (double)nb : 1.0;
    ChainDigit dg[8];
    int shift = 0;
    for (int i = 0; i < passes; i++) {
      dg[i].shift = shift;
      dg[i].bits = total / passes + (i < total % passes ? 1 : 0);
      dg[i].dens = 1.0;
      if (range_bits && shift <= range_bits - 1 && range_bits - 1 < shift + dg[i].bits) dg[i].dens *= dens_range;
      if (rank_bits && i == passes - 1) dg[i].dens *= dens_rank;
      shift += dg[i].bits;
    }
    bool ok = false;
    if ((rc = slab_chain(c, c->sbuf[0].p, (u32)n, dg, passes, 1, &ok, &pc_n, &pc_cap, &pc_which)) != HMJ_OK) return rc;
    if (ok) {
      n_passes = passes;
      c->timing.path |= HMJ_PATH_SLAB;
    } else {
      c->wm->gtable_sort_slab_cooldown = 8;
      if (c->trace) std::fprintf(stderr, "[hmj] join nb=%u np=%u ordered: a slab of the composite sort's chain overflowed -> exact passes\n", nb, np);
    }
  }
  if (pc_n) {
    // (sorted: the pieces of the chain's last pass)
  } else if (!wide) {
    if ((rc = lsd(rank_bits + range_bits)) != HMJ_OK) return rc;
  } else {
    if ((rc = lsd(range_bits)) != HMJ_OK) return rc;  // {payload, rank} by payload
    if (n) {
      HIP_TRY(hmj::launch_gtable_swap(sorted, c->sbuf[which ^ 1].p, n, c->num_cus, c->stream));  // -> {rank, payload}
      sorted = c->sbuf[which ^ 1].p;
      which ^= 1;
    }
    if ((rc = lsd(rank_bits)) != HMJ_OK) return rc;  // stably by rank
  }
  // ---- 6. composites -> result rows
  if (n) {
    const size_t bytes = (size_t)n * 8;
    if ((rc = ensure_dev(c, c->out_key, bytes)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->out_rval, bytes)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->out_sval, bytes)) != HMJ_OK) return rc;
    sp = span_begin(c, K_PROBE_WRITE, -1);
    if (pc_n) {
      DevBuf& so = pc_which ? c->slab_bs : c->slab_a;
      DevBuf& co = pc_which ? c->cnt_bs : c->cnt_a;
      if ((rc = ensure_dev(c, c->piece_off, (size_t)pc_n * 8)) != HMJ_OK) return rc;
      HIP_TRY(hmj::launch_piece_offsets((const u32*)co.p, pc_n, (u64*)c->piece_off.p, c->stream));
      HIP_TRY(hmj::launch_gtable_expand_pieces(so.p, (const u32*)co.p, (const u64*)c->piece_off.p, pc_n, pc_cap, sortedR, svmin, range_bits,
                                               (u64*)c->out_key.p, (u64*)c->out_rval.p, (u64*)c->out_sval.p, (u64*)c->accum.p, extra,
                                               c->num_cus, c->stream));
    } else {
      HIP_TRY(hmj::launch_gtable_expand(sorted, n, sortedR, svmin, range_bits, (u64*)c->out_key.p, (u64*)c->out_rval.p,
                                        (u64*)c->out_sval.p, (u64*)c->accum.p, extra, wide, c->num_cus, c->stream));
    }
    span_end(c, sp);
    HIP_TRY(hipMemcpyAsync(hh, c->accum.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  }  // (!runs_done)
  std::memset(out, 0, sizeof(*out));
  out->n_matches = n;
  out->sum_r = hh[hmj::ACC_SUM_R];
  out->sum_s = hh[hmj::ACC_SUM_S];
  out->xor_fold = hh[hmj::ACC_XOR];
  out->mix_sum = hh[hmj::ACC_MIX];
  out->sum_probe_all = hh[hmj::ACC_SUM_P];
  c->timing.path |= HMJ_PATH_GLOBAL_TABLE | HMJ_PATH_ORDER_BY_RANK_SORT;
  if (!runs_done) {
    c->timing.radix_bits = rank_bits + range_bits;
    c->timing.radix_passes = n_passes;
  }
  c->timing.n_probe_items = 1;
  c->timing.bytes_probe_write = 16ull * (n_build + n_probe) + 24ull * n;
  if (n) {
    const size_t bytes = (size_t)n * 8;
    if (to_host) {
      if ((rc = ensure_host(c, c->h_key, bytes, false)) != HMJ_OK) return rc;
      if ((rc = ensure_host(c, c->h_rval, bytes, false)) != HMJ_OK) return rc;
      if ((rc = ensure_host(c, c->h_sval, bytes, false)) != HMJ_OK) return rc;
      const int s2 = span_begin(c, K_D2H, -1);
      HIP_TRY(hipMemcpyAsync(c->h_key.p, c->out_key.p, bytes, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipMemcpyAsync(c->h_rval.p, c->out_rval.p, bytes, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipMemcpyAsync(c->h_sval.p, c->out_sval.p, bytes, hipMemcpyDeviceToHost, c->stream));
      span_end(c, s2);
      HIP_TRY(hipStreamSynchronize(c->stream));
    }
    out->key = (const uint64_t*)(to_host ? c->h_key.p : c->out_key.p);
    out->rval = (const uint64_t*)(to_host ? c->h_rval.p : c->out_rval.p);
    out->sval = (const uint64_t*)(to_host ? c->h_sval.p : c->out_sval.p);
  }
  if (c->trace) std::fprintf(stderr, "[hmj] join nb=%u np=%u flags=%#x: ordered by a sort on (rank, payload): %d + %d bits\n", nb, np, flags,
                             rank_bits, range_bits);
  *done = true;
  return HMJ_OK;
}

static int join_device_planned(hmj_ctx* c, const void* R, uint64_t n_build, const void* S, uint64_t n_probe, uint32_t flags,
                               hmj_result* out, bool to_host);
// ---- ordered foreign-key joins beyond what ONE 18-bit plan holds (round 5) ----------------------------------------------
// Two 9-bit slab passes make 2^18 partitions; under 2^30 probe rows those average 4096 rows, the fullest outgrow the
// one-pass ordered write's 6144-row shape, and with more than 2^21 build rows the rank-run form is out of reach too: the
// exact path with split partitions and the order epilogue ran, 134-432 ms (profiles/r05y_grid_fk_2p29_2p30.txt).  Ordered
// output is a concatenation over KEY RANGES: both relations are cut on their top h varying key bits (one exact radix pass:
// dense ranges + offsets), range after range is joined by the planner as a join of its own -- half or a quarter of the
// rows, so its plan fits again -- and its rows are appended to the result columns.  One more pass over both relations
// (48 B per row) and one copy of the result (48 B per row) buy a plan that works.
static bool key_ranges_wanted(const hmj_ctx* c, uint64_t nb, uint64_t np, uint32_t flags, bool to_host, int* h) {
  if (!(flags & HMJ_ORDERED) || to_host || c->prepare_only || !c->arrive_ev.empty() || c->force_bits >= 0 || nb == 0 || np == 0 ||
      np > 0xFFFFFFFFull || nb > 0xFFFFFFFFull)
    return false;
  if (c->key_ranges_force > 0) {  // (tests: any ordered device-resident join, 2^force ranges)
    *h = c->key_ranges_force;
    return true;
  }
  // (the reasoning below is about 18-bit plans: what the planner takes from ~2^29 probe rows on)
  if (!c->key_ranges || np < c->big_join_rows || np < 8 * nb || (c->prep.valid && c->prep.n == (u32)nb)) return false;
  const double P18 = (double)(1u << (2 * hmj::SLAB_MAX_BITS));
  auto held = [&](double b, double p) {  // does one plan hold a join of b x p rows: the one-pass ordered write, or the rank-run form
    if (fk_probe_rows_hi(p / P18, p / b, P18) <= 6144.0) return true;
    int tb = 0, lv = 0;
    return b <= 16.0 * (double)c->gtable_max_rows && rank_runs_fit(c, (uint64_t)b, (uint64_t)p, &tb, &lv);
  };
  if (held((double)nb, (double)np)) return false;
  // (a range's share of the rows is not exactly 1 / 2^h: a twentieth of room)
  for (int hh = 1; hh <= 3; hh++)
    if (held(1.05 * (double)nb / (double)(1 << hh), 1.05 * (double)np / (double)(1 << hh))) {
      *h = hh;
      return true;
    }
  return false;
}

static int join_by_key_ranges(hmj_ctx* c, const void* R, uint64_t n_build, const void* S, uint64_t n_probe, uint32_t flags, hmj_result* out,
                              int h, bool* done) {
  *done = false;
  int rc;
  if (!out) return fail(c, HMJ_E_ARG, "out is NULL");
  if ((rc = check_rel(c, R, n_build, "build_aos is NULL")) != HMJ_OK) return rc;
  if ((rc = check_rel(c, S, n_probe, "probe_aos is NULL")) != HMJ_OK) return rc;
  const u32 nb = (u32)n_build, np = (u32)n_probe;
  if ((rc = ensure_dev(c, c->offs64, 8 * sizeof(u64))) != HMJ_OK) return rc;
  if ((rc = ensure_host(c, c->h_accum, 8 * sizeof(u64))) != HMJ_OK) return rc;
  u64* hh = (u64*)c->h_accum.p;
  // ---- 1. the bits in which keys differ, over both relations
  {
    const u64 init[3] = {0, ~0ull, 0};
    HIP_TRY(hipMemcpyAsync(c->offs64.p, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hmj::launch_key_exact(R, nb, S, np, 0ull, (u64*)c->offs64.p, c->num_cus, c->stream, true));
    HIP_TRY(hipMemcpyAsync(hh, c->offs64.p, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  const u64 diff = hh[0];
  if (diff == 0) return HMJ_OK;  // one key value in all: nothing to cut on, the planner's own paths answer
  const int hb = 63 - __builtin_clzll(diff);
  if (h > hb + 1) h = hb + 1;
  const int shift = hb + 1 - h;
  const u32 D = 1u << h;
  // ---- 2. both relations cut into their 2^h key ranges (stable, dense, + range starts)
  if ((rc = ensure_dev(c, c->split_r, (size_t)nb * 16 + 16)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->split_s, (size_t)np * 16 + 16)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->split_off, 2 * ((size_t)D + 1) * 8)) != HMJ_OK) return rc;
  std::vector<u64> off(2 * ((size_t)D + 1));
  u64* od = (u64*)c->split_off.p;
  if ((rc = radix_pass(c, R, c->split_r.p, nb, shift, h, 0, od, 0)) != HMJ_OK) return rc;
  if ((rc = radix_pass(c, S, c->split_s.p, np, shift, h, 1, od + D + 1, 0)) != HMJ_OK) return rc;
  HIP_TRY(hipMemcpyAsync(off.data(), od, off.size() * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  // ---- 3. range after range: a join of its own, its rows appended
  hmj_result tot;
  std::memset(&tot, 0, sizeof(tot));
  hmj_timing tsum;  // (the call's phases: the sum over its ranges)
  std::memset(&tsum, 0, sizeof(tsum));
  u64 rows_done = 0;
  auto grow = [&](u64 need_rows) -> int {  // the three result columns hold need_rows (what is there stays)
    const size_t bytes = (size_t)need_rows * 8;
    DevBuf* cols[3] = {&c->cat_key, &c->cat_rval, &c->cat_sval};
    for (DevBuf* b : cols) {
      if (bytes <= b->cap) continue;
      DevBuf nw;
      int r2 = ensure_dev(c, nw, bytes + (bytes >> 2));
      if (r2 != HMJ_OK) return r2;
      hipError_t e = rows_done ? hipMemcpyAsync(nw.p, b->p, (size_t)rows_done * 8, hipMemcpyDeviceToDevice, c->stream) : hipSuccess;
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      if (e != hipSuccess) {
        free_dev(nw);  // (the column keeps what it held; the join fails)
        return fail(c, HMJ_E_HIP, "growing a result column of a join by key ranges", e);
      }
      free_dev(*b);
      *b = nw;
    }
    return HMJ_OK;
  };
  for (u32 d = 0; d < D; d++) {
    const u64 r0 = off[d], nr = off[d + 1] - off[d], s0 = off[D + 1 + d], ns = off[D + 1 + d + 1] - off[D + 1 + d];
    if (ns == 0) continue;                              // no probe rows: no result rows, no probe payloads
    if (nr == 0 && !(flags & HMJ_SUM_PROBE)) continue;  // nothing can match
    hmj_result sub;
    std::memset(&sub, 0, sizeof(sub));
    rc = join_device_planned(c, (const char*)c->split_r.p + r0 * 16, nr, (const char*)c->split_s.p + s0 * 16, ns, flags, &sub, false);
    if (rc != HMJ_OK) return rc;
    add_timing(&tsum, c->timing);
    if (sub.n_matches) {
      if (rows_done == 0 && d + 1 < D) {  // (first range: room for the whole result at this range's rate, so that growing is rare)
        const double est = (double)sub.n_matches * (double)np / (double)ns * 1.05 + 4096.0;
        if ((rc = grow(est > (double)sub.n_matches ? (u64)est : sub.n_matches)) != HMJ_OK) return rc;
      }
      if ((rc = grow(rows_done + sub.n_matches)) != HMJ_OK) return rc;
      const size_t bytes = (size_t)sub.n_matches * 8;
      HIP_TRY(hipMemcpyAsync((u64*)c->cat_key.p + rows_done, sub.key, bytes, hipMemcpyDeviceToDevice, c->stream));
      HIP_TRY(hipMemcpyAsync((u64*)c->cat_rval.p + rows_done, sub.rval, bytes, hipMemcpyDeviceToDevice, c->stream));
      HIP_TRY(hipMemcpyAsync((u64*)c->cat_sval.p + rows_done, sub.sval, bytes, hipMemcpyDeviceToDevice, c->stream));
      rows_done += sub.n_matches;
    }
    tot.n_matches += sub.n_matches;
    tot.sum_r += sub.sum_r;
    tot.sum_s += sub.sum_s;
    tot.xor_fold ^= sub.xor_fold;
    tot.mix_sum += sub.mix_sum;
    tot.sum_probe_all += sub.sum_probe_all;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  *out = tot;
  out->key = rows_done ? (const uint64_t*)c->cat_key.p : nullptr;
  out->rval = rows_done ? (const uint64_t*)c->cat_rval.p : nullptr;
  out->sval = rows_done ? (const uint64_t*)c->cat_sval.p : nullptr;
  c->timing = tsum;
  c->timing.path |= HMJ_PATH_KEY_RANGES;
  *done = true;
  return HMJ_OK;
}

int join_device(hmj_ctx* c, const void* R, uint64_t n_build, const void* S, uint64_t n_probe,
                uint32_t flags, hmj_result* out, bool to_host) {
  // An UNORDERED materialising foreign-key join whose fullest 18-bit partition outgrows the unique-key write's 5120 rows but
  // not the ordered write's wide shape (6144) is run as an ordered one -- ordered rows are a valid answer, and the exact
  // path with split partitions that would run otherwise is slower than the ordered write (2^25 ... 2^26.6 x 2^30 rows:
  // 46-56 ms against 34; profiles/r05af_*).
  if ((flags & HMJ_MATERIALIZE) && !(flags & HMJ_ORDERED) && !c->prepare_only && c->promote_to_ordered && n_build > 0 &&
      n_probe <= 0xFFFFFFFFull && n_probe >= c->big_join_rows && (double)n_probe >= 2.5 * (double)n_build && c->force_bits < 0) {
    const double P18 = (double)(1u << (2 * hmj::SLAB_MAX_BITS));
    const double hi = fk_probe_rows_hi((double)n_probe / P18, (double)n_probe / (double)n_build, P18);
    if (hi > 5120.0 && hi <= 6144.0) flags |= HMJ_ORDERED;
  }
  // the workload this call belongs to (a prepared build side: the join it was announced for) and what is known about it
  memo_for(c, workload_signature(n_build, c->prepare_only ? c->probe_hint : n_probe, flags, 0));
  std::memset(&c->plan, 0, sizeof(c->plan));
  int rc = HMJ_OK, kr_h = 0;
  bool kr_done = false;
  if (key_ranges_wanted(c, n_build, n_probe, flags, to_host, &kr_h)) {
    rc = join_by_key_ranges(c, R, n_build, S, n_probe, flags, out, kr_h, &kr_done);
  } else if (c->cat_key.p) {  // (the appended columns of an earlier join by key ranges: results live until the next join)
    free_dev(c->cat_key);
    free_dev(c->cat_rval);
    free_dev(c->cat_sval);
  }
  if (rc == HMJ_OK && !kr_done) rc = join_device_planned(c, R, n_build, S, n_probe, flags, out, to_host);
  hmj_plan_desc& p = c->plan;
  p.struct_size = sizeof(p);
  p.path = c->timing.path;
  p.radix_bits = c->timing.radix_bits;
  p.radix_passes = c->timing.radix_passes;
  p.key_prefix_bits = c->timing.key_prefix_bits;
  p.key_window_low = c->timing.key_window_low;
  p.n_partitions = (c->timing.path & (HMJ_PATH_GLOBAL_TABLE | HMJ_PATH_LDS_TABLE)) ? 0u : (p.radix_bits < 32 ? 1u << p.radix_bits : 0u);
  p.probe_items = c->timing.n_probe_items;
  p.cooling = c->wm->cooling();
  p.workload = c->wm_sig;
  return rc;
}
static int join_device_planned(hmj_ctx* c, const void* R, uint64_t n_build, const void* S, uint64_t n_probe, uint32_t flags,
                               hmj_result* out, bool to_host) {
  {
    bool done = false;
    int rc = try_global_table(c, R, n_build, S, n_probe, flags, out, to_host, &done);
    if (rc != HMJ_OK || done) return rc;
    rc = try_small_build_ordered(c, R, n_build, S, n_probe, flags, out, to_host, &done);
    if (rc != HMJ_OK || done) return rc;
  }
  bool auto_prefix = true, slab = true, fast_write = true, win_ordered = true, prefix_unsafe = false, slab_probe = true;
  if ((flags & HMJ_ORDERED) && c->wm->exact_prefix_joins > 0) {  // (a failed attempt costs more than the pass over the keys)
    c->wm->exact_prefix_joins--;
    prefix_unsafe = true;
  }
  bool forced_here = false;  // (kRetryMoreBits pins the plan's bits for the rest of THIS join)
  struct Unforce {
    hmj_ctx* c;
    bool* on;
    ~Unforce() {
      if (*on) c->force_bits = -1;
      c->expand_allow_rebits = true;
    }
  } unforce{c, &forced_here};
  for (int attempt = 0; attempt < 8; attempt++) {
    c->plan.attempts = (uint32_t)attempt + 1;
    int rc = join_device_impl(c, R, n_build, S, n_probe, flags, out, to_host, auto_prefix, slab, fast_write,
                              win_ordered, prefix_unsafe, slab_probe);
    c->plan.refused |= rc == kRetryNoSlab ? HMJ_REFUSED_SLAB_OVERFLOW : rc == kRetryNoPrefix ? HMJ_REFUSED_PREFIX_VIOLATED
                       : rc == kRetryNoFastWrite ? HMJ_REFUSED_FAST_WRITE_GAVE_UP : rc == kRetryNoSlabProbe ? HMJ_REFUSED_SLAB_PROBE_OVERFLOW
                       : rc == kRetryMoreBits ? HMJ_REFUSED_EXPANSION_GAVE_UP : 0u;
    if (c->trace)
      std::fprintf(stderr, "[hmj] join nb=%llu np=%llu flags=%#x attempt %d: rc=%d bits=%d passes=%d path=%#x prefix=%d low=%d items=%u%s\n",
                   (unsigned long long)n_build, (unsigned long long)n_probe, flags, attempt, rc, c->timing.radix_bits,
                   c->timing.radix_passes, c->timing.path, c->timing.key_prefix_bits, c->timing.key_window_low,
                   c->timing.n_probe_items,
                   rc == kRetryNoSlab ? " -> retry without the slab path (a slab overflowed)"
                   : rc == kRetryNoPrefix ? " -> retry: a row outside the sampled key prefix"
                   : rc == kRetryNoFastWrite ? " -> retry without the unique-key write mode"
                   : rc == kRetryNoWinOrdered ? " -> retry without the key window"
                   : rc == kRetryNoSlabProbe ? " -> retry without probe-side slabs (one overflowed)"
                   : rc == kRetryMoreBits ? " -> retry with one more radix bit (a partition did not fit the ordered expansion)" : "");
    if (rc == kRetryNoSlab || rc == kRetryNoPrefix || rc == kRetryNoFastWrite || rc == kRetryNoWinOrdered || rc == kRetryNoSlabProbe ||
        rc == kRetryMoreBits) {
      // forget the abandoned attempt's phase spans (the enclosing total / h2d spans stay)
      std::vector<Span> keep;
      for (const Span& sp : c->spans)
        if (sp.kind == K_TOTAL || sp.kind == K_H2D) keep.push_back(sp);
      c->spans.swap(keep);
      std::memset(&c->timing, 0, sizeof(c->timing));
      if (rc == kRetryNoSlab) slab = false;
      else if (rc == kRetryMoreBits) {  // once per join: the next attempt plans the bits asked for
        c->force_bits = c->expand_rebits;
        c->expand_allow_rebits = false;
        forced_here = true;
      }
      else if (rc == kRetryNoSlabProbe) slab_probe = false;
      else if (rc == kRetryNoFastWrite) fast_write = false;
      else if (rc == kRetryNoWinOrdered) win_ordered = false;
      else if (!prefix_unsafe && win_ordered) {  // outliers: the exact prefix (join_device_impl), also for the next joins
        prefix_unsafe = true;
        c->wm->exact_prefix_joins = 32;
      }
      else auto_prefix = false;
      continue;
    }
    return rc;
  }
  return fail(c, HMJ_E_HIP, "join could not be planned");
}

int prepare_build(hmj_ctx* c, const void* R, uint64_t n_build, uint64_t n_probe_hint) {
  hmj_result dummy;
  c->prepare_only = true;
  c->probe_hint = n_probe_hint;
  c->prep.valid = false;
  const int rc = join_device(c, R, n_build, nullptr, 0, 0, &dummy, false);
  c->prepare_only = false;
  return rc;
}

}  // namespace hmj_host

extern "C" {

int hmj_create(hmj_ctx** out, int device_id) {
  if (!out) return HMJ_E_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return HMJ_E_NODEV;
  }
  if (device_id < 0 && hipGetDevice(&device_id) != hipSuccess) return HMJ_E_NODEV;
  if (device_id >= ndev) return HMJ_E_ARG;
  if (hipSetDevice(device_id) != hipSuccess) return HMJ_E_NODEV;
  hmj_ctx* c = new hmj_ctx();
  c->device = device_id;
  if (const char* e = getenv("HMJ_SLAB")) c->slab_mode = atoi(e);
  if (const char* e = getenv("HMJ_ONE_PASS_SLAB")) c->one_pass_slab = atoi(e) != 0;  // 0: mid-size build sides keep the exact one-pass plan
  if (const char* e = getenv("HMJ_GTABLE")) c->gtable_mode = atoi(e) != 0;  // 0: small build sides are partitioned too
  if (const char* e = getenv("HMJ_RANK_RUNS")) c->rank_runs_mode = atoi(e) != 0;  // 0: ordered small-build joins always sort composites
  if (const char* e = getenv("HMJ_RANK_RUNS_MAX_CUT")) c->rank_runs_max_cut = atoi(e) < 0 ? 0 : atoi(e) > 16 ? 16 : atoi(e);
  if (const char* e = getenv("HMJ_RANK_RUNS_MAX_GROUP")) c->rank_runs_max_group = atoi(e) < 0 ? 0 : atoi(e) > 4 ? 4 : atoi(e);
  if (const char* e = getenv("HMJ_PROMOTE_TO_ORDERED")) c->promote_to_ordered = atoi(e) != 0;  // 0: an unordered materialising join is never run as an ordered one
  if (const char* e = getenv("HMJ_KEY_RANGES")) c->key_ranges = atoi(e) != 0;  // 0: ordered joins are never cut into key ranges joined one after the other
  if (const char* e = getenv("HMJ_KEY_RANGES_FORCE")) c->key_ranges_force = atoi(e) < 0 ? 0 : atoi(e) > 4 ? 4 : atoi(e);  // tests: every ordered device-resident join in 2^n key ranges
  if (const char* e = getenv("HMJ_RANK_RUNS_WAVE")) c->rank_runs_wave = atoi(e) != 0;  // 0: partitions of <= 512 rows are sorted by 256-thread workgroups too
  if (const char* e = getenv("HMJ_RANK_RUNS_MAX_LEVEL")) c->rank_runs_max_level = atoi(e) < 0 ? 0 : atoi(e) > 2 ? 2 : atoi(e);  // 0: the LDS sorts' 256-thread shape only (2048 rows per partition)
  if (const char* e = getenv("HMJ_GTABLE_SORT")) c->gtable_sort_mode = atoi(e) != 0;  // 0: ordered joins never sort composites of (rank, payload)
  if (const char* e = getenv("HMJ_EXPAND_FK_FANOUT")) c->expand_fk_fanout = atoi(e) > 0 ? (u32)atoi(e) : 0u;  // 0: never for unique build keys
  if (const char* e = getenv("HMJ_SORT_MSD")) c->sort_msd = atoi(e) != 0;  // 0: hmj_sort_u64_device never takes its MSD form (two slab passes + an LDS sort per partition)
  if (const char* e = getenv("HMJ_SORT_MSD_MEAN")) c->sort_msd_mean = atof(e) >= 64.0 ? atof(e) : c->sort_msd_mean;  // the window narrows until partitions average at most this many rows
  if (const char* e = getenv("HMJ_SORT_MSD_MAX_BITS")) c->sort_msd_max_bits = atoi(e) < 2 ? 2 : atoi(e) > 18 ? 18 : atoi(e);  // fewer, larger partitions (up to 8192 rows each)
  if (const char* e = getenv("HMJ_SORT_MSD_MIN_LOG2")) {
    const int l = atoi(e);
    if (l >= 16 && l <= 32) c->sort_msd_min = 1ull << l;
  }
  if (const char* e = getenv("HMJ_SORT_SLAB_MIN_LOG2")) {
    const int l = atoi(e);
    if (l >= 12 && l <= 32) c->sort_slab_min = 1ull << l;
  }
  if (const char* e = getenv("HMJ_GTABLE_SORT_SLAB_MIN_LOG2")) {
    const int l = atoi(e);
    if (l >= 12 && l <= 32) c->gtable_sort_slab_min = 1ull << l;
  }
  if (const char* e = getenv("HMJ_GTABLE_SORT_FANOUT")) c->gtable_sort_fanout = atoi(e) > 0 ? (u32)atoi(e) : 1u;
  if (const char* e = getenv("HMJ_PLACE")) {  // 0: nothing is probed; n: joins search too (at most n candidates per buffer)
    c->place_tune = atoi(e) != 0;
    if (atoi(e) > 0) {
      c->place_tries = atoi(e);
      c->place_search_always = true;
    }
  }
  if (const char* e = getenv("HMJ_PLACE_BUDGET_MS")) {
    if (atof(e) >= 0.0) c->place_budget_ms = (float)atof(e);
  }
  if (const char* e = getenv("HMJ_PLACE_MIN_MB")) {
    const long long mb = atoll(e);
    if (mb >= 1) c->place_min_bytes = (size_t)mb << 20;
  }
  if (const char* e = getenv("HMJ_UPLOAD")) c->staged_upload = std::strcmp(e, "staged") == 0;
  if (const char* e = getenv("HMJ_HOST_PIPELINE")) c->host_pipeline = atoi(e) != 0;
  if (const char* e = getenv("HMJ_SLAB_MIN_LOG2")) {
    const int l = atoi(e);
    if (l >= 16 && l <= 31) c->slab_min_rows = 1u << l;
  }
  if (const char* e = getenv("HMJ_SLAB_PROBE_KB")) {
    const int k = atoi(e);
    if (k >= 1 && k <= 512) c->slab_probe_kb = (u32)k;
  }
  if (const char* e = getenv("HMJ_TRACE")) c->trace = atoi(e) != 0;
  // Switches that only serve ablations and the experiments under tools/ (developer builds, tools/build_variant.sh): the
  // release library plans from the data, hmj_last_plan says how, and tests assert on that instead of pinning paths.
#ifdef HMJ_DEV
  if (const char* e = getenv("HMJ_LTABLE")) c->ltable_mode = atoi(e) != 0;  // 0: tiny build sides take the L2-resident table too
  if (const char* e = getenv("HMJ_GTABLE_MAX_LOG2")) {
    const int l = atoi(e);
    if (l >= 0 && l <= 28) c->gtable_max_rows = 1ull << l;
  }
  if (const char* e = getenv("HMJ_GTABLE_FANOUT")) c->gtable_min_fanout = atoi(e) > 0 ? (u32)atoi(e) : 0u;
  if (const char* e = getenv("HMJ_GTABLE_WG")) c->gtable_wg_per_cu = atoi(e) > 0 ? atoi(e) : 8;
  if (const char* e = getenv("HMJ_BUILD_SORT_MSD")) c->build_sort_msd = atoi(e) != 0;  // 0: the rank forms sort their build side by LSD passes
  if (const char* e = getenv("HMJ_GTABLE_SORT_SLAB")) c->gtable_sort_slab = atoi(e) != 0;  // 0: the composites' LSD passes are exact passes (hist + scan + scatter)
  if (const char* e = getenv("HMJ_FK_PAYLOAD_BUCKETS")) c->fk_payload_buckets = atoi(e) > 0 ? (u32)atoi(e) : 0u;  // fan-out from which the ordered foreign-key write buckets by payload (0: never)
  if (const char* e = getenv("HMJ_ORDERED_EXPANSION")) c->expand_mode = atoi(e) != 0;  // 0: ordered joins with duplicate build keys write in probe order and sort the rows
  if (const char* e = getenv("HMJ_SORT_SLAB")) c->sort_slab = atoi(e) != 0;  // 0: hmj_sort_u64_device runs exact passes (hist + scan + scatter)
  if (const char* e = getenv("HMJ_GTABLE_SLOTS")) c->gtable_slots_per_row = atoi(e) >= 2 ? (u32)atoi(e) : 2u;
  if (const char* e = getenv("HMJ_GTABLE_MAX_LOG_CAP")) {
    const int l = atoi(e);
    if (l >= 10 && l <= 30) c->gtable_max_log_cap = l;  // (the build launcher takes tables of up to 2^30 slots)
  }
  if (const char* e = getenv("HMJ_DENSE_PLAN")) c->dense_plan = atoi(e) != 0;
  if (const char* e = getenv("HMJ_SORTED_WRITE")) {
    c->sorted_mode = atoi(e) != 0;
    c->memo_init.sorted_chained = atoi(e) == 2;
    c->sorted_chained_forced = atoi(e) == 2 || atoi(e) == 3;  // 3: never chained
  }
  if (const char* e = getenv("HMJ_SORTED_HALF")) c->sorted_half = atoi(e) != 0;
  if (const char* e = getenv("HMJ_FK_PLAN"))  // ordered foreign-key joins: pin the plan (1 wide, 2 half, 3 narrow; else automatic)
    c->fk_plan = std::strcmp(e, "wide") == 0 ? 1 : std::strcmp(e, "half") == 0 ? 2 : std::strcmp(e, "narrow") == 0 ? 3 : 0;
  if (const char* e = getenv("HMJ_SORTED_WIDE")) c->memo_init.sorted_wide = atoi(e) != 0;
  if (const char* e = getenv("HMJ_WINDOW")) c->window_mode = atoi(e) != 0;
  if (const char* e = getenv("HMJ_SPLIT")) c->split_mode = atoi(e) != 0;
  if (const char* e = getenv("HMJ_SCATTER")) c->scatter_variant = (std::strcmp(e, "plain") == 0) ? 0 : 1;
  if (const char* e = getenv("HMJ_DEBUG_ABLATE")) c->dev_ablate = (u32)atoi(e);
  if (const char* e = getenv("HMJ_ORDERED_MODEL")) {  // "runs_ns=0.03,comp_fixed_ms=0.5": single constants of the ordered cost model
    struct { const char* name; double* v; } f[] = {
        {"part_fixed_ms", &c->ordered_model.part_fixed_ms}, {"part_ns", &c->ordered_model.part_ns},
        {"part_ns_per_f", &c->ordered_model.part_ns_per_f}, {"part_bucket_ns", &c->ordered_model.part_bucket_ns},
        {"part_bucket_ns_per_f", &c->ordered_model.part_bucket_ns_per_f}, {"part_epilogue_ns", &c->ordered_model.part_epilogue_ns},
        {"comp_fixed_ms", &c->ordered_model.comp_fixed_ms}, {"comp_ns_chain", &c->ordered_model.comp_ns_chain},
        {"comp_ns_exact", &c->ordered_model.comp_ns_exact}, {"comp_ns_wide", &c->ordered_model.comp_ns_wide},
        {"comp_ns_beyond_l2", &c->ordered_model.comp_ns_beyond_l2}, {"runs_fixed_ms", &c->ordered_model.runs_fixed_ms},
        {"runs_ns", &c->ordered_model.runs_ns}, {"runs_ns_per_run", &c->ordered_model.runs_ns_per_run},
        {"runs_range_ns", &c->ordered_model.runs_range_ns}, {"runs_ns_per_log2_build", &c->ordered_model.runs_ns_per_log2_build},
        {"runs_wave_ns_per_run", &c->ordered_model.runs_wave_ns_per_run}, {"runs_wave_ns", &c->ordered_model.runs_wave_ns}};
    std::string spec(e);
    size_t pos = 0;
    while (pos < spec.size()) {
      size_t end = spec.find(',', pos);
      if (end == std::string::npos) end = spec.size();
      const std::string item = spec.substr(pos, end - pos);
      const size_t eq = item.find('=');
      if (eq != std::string::npos)
        for (auto& x : f)
          if (item.compare(0, eq, x.name) == 0 && std::strlen(x.name) == eq) *x.v = atof(item.c_str() + eq + 1);
      pos = end + 1;
    }
  }
#endif
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->num_cus = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return HMJ_E_HIP;
  }
  c->stream = c->own_stream;
  c->events.resize(kMaxEvents);
  for (auto& e : c->events)
    if (hipEventCreate(&e) != hipSuccess) {
      delete c;
      return HMJ_E_HIP;
    }
  std::memset(&c->timing, 0, sizeof(c->timing));
  *out = c;
  return HMJ_OK;
}

void hmj_destroy(hmj_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  comm_destroy(c);
  DevBuf* devs[] = {&c->rbuf[0], &c->rbuf[1], &c->sbuf[0], &c->sbuf[1], &c->in_r, &c->in_s,
                    &c->hist, &c->totals, &c->r_off, &c->s_off, &c->part_count,
                    &c->part_out_off, &c->accum, &c->out_key, &c->out_rval, &c->out_sval,
                    &c->offs64, &c->irregular, &c->ord_key, &c->ord_rval, &c->ord_sval, &c->matched, &c->vparts,
                    &c->slab_a, &c->slab_br, &c->slab_bs, &c->cnt_a, &c->cnt_br, &c->cnt_bs, &c->lookback, &c->gtab, &c->piece_off,
                    &c->split_r, &c->split_s, &c->split_off, &c->cat_key, &c->cat_rval, &c->cat_sval, &c->msd_off};
  for (DevBuf* b : devs) free_dev(*b);
  HostBuf* hosts[] = {&c->h_accum, &c->h_key, &c->h_rval, &c->h_sval};
  for (HostBuf* b : hosts) free_host(*b);
  for (auto& e : c->events)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : c->up_events) (void)hipEventDestroy(e);
  for (auto& e : c->place_ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& st : c->up_streams) (void)hipStreamDestroy(st);
  for (auto& e : c->copy_ev) (void)hipEventDestroy(e);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  for (auto& b : c->up_slots) free_host(b);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int hmj_set_stream(hmj_ctx* c, void* hip_stream) {
  if (!c) return HMJ_E_ARG;
  c->stream = (hip_stream == HMJ_STREAM_OWN) ? c->own_stream : (hipStream_t)hip_stream;
  return HMJ_OK;
}

int hmj_set_radix_bits(hmj_ctx* c, int total_bits) {
  if (!c || total_bits > 27) return HMJ_E_ARG;
  c->force_bits = total_bits < 0 ? -1 : total_bits;
  return HMJ_OK;
}

int hmj_autotune_radix_bits(hmj_ctx* c, uint64_t n_build, uint64_t n_probe, int apply, int* best_bits,
                            double ms[3]) {
  if (!c || !best_bits) return HMJ_E_ARG;
  if (n_build == 0 || n_probe == 0 || n_build >= (1ull << 32) || n_probe >= (1ull << 32))
    return fail(c, HMJ_E_ARG, "autotune needs 0 < rows < 2^32 per relation");
  HIP_TRY(hipSetDevice(c->device));
  int B0, passes, pb[4];
  plan_bits(n_build, -1, &B0, &passes, pb);
  void *R = nullptr, *S = nullptr;
  if (hipMalloc(&R, n_build * 16) != hipSuccess || hipMalloc(&S, n_probe * 16) != hipSuccess) {
    if (R) (void)hipFree(R);
    return fail(c, HMJ_E_OOM, "autotune: device memory for the synthetic relations");
  }
  const u64 seed = 0x243F6A8885A308D3ull;
  int rc = HMJ_OK;
  const int saved = c->force_bits;
  double best = -1.0;
  int best_b = B0;
  if (hmj::launch_gen_build(R, n_build, 0, seed, c->stream) != hipSuccess ||
      hmj::launch_gen_probe(S, n_probe, 0, n_build, seed, 0, c->stream) != hipSuccess ||
      hipStreamSynchronize(c->stream) != hipSuccess)
    rc = fail(c, HMJ_E_HIP, "autotune: generating the synthetic relations failed");
  for (int k = 0; k < 3 && rc == HMJ_OK; k++) {
    const int b = B0 - 1 + k;
    if (ms) ms[k] = -1.0;
    if (b < 0 || b > 27) continue;
    c->force_bits = b;
    double t_best = -1.0;
    for (int rep = 0; rep < 3 && rc == HMJ_OK; rep++) {  // first run also pays for the workspace
      hmj_result res;
      const auto t0 = std::chrono::steady_clock::now();
      rc = join_device(c, R, n_build, S, n_probe, 0, &res, false);
      const double t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (rc == HMJ_OK && res.n_matches != n_probe) rc = fail(c, HMJ_E_HIP, "autotune: wrong match count");
      if (rep > 0 && (t_best < 0 || t < t_best)) t_best = t;
    }
    if (ms) ms[k] = t_best;
    if (rc == HMJ_OK && (best < 0 || t_best < best)) {
      best = t_best;
      best_b = b;
    }
  }
  (void)hipFree(R);
  (void)hipFree(S);
  c->force_bits = (rc == HMJ_OK && apply) ? best_b : saved;
  c->prep.valid = false;
  if (rc == HMJ_OK) *best_bits = best_b;
  return rc;
}

int hmj_set_key_prefix_bits(hmj_ctx* c, int bits) {
  if (!c || bits < -1 || bits > 48) return HMJ_E_ARG;
  c->prefix_bits = bits;
  return HMJ_OK;
}

int hmj_plan(uint64_t n_build, int* total_bits, int* n_passes, int pass_bits[4]) {
  if (!total_bits || !n_passes || !pass_bits) return HMJ_E_ARG;
  plan_bits(n_build, -1, total_bits, n_passes, pass_bits);
  return HMJ_OK;
}

int hmj_set_profiling(hmj_ctx* c, int enabled) {
  if (!c) return HMJ_E_ARG;
  c->profiling = enabled != 0;
  return HMJ_OK;
}

int hmj_placement_info(hmj_ctx* c, hmj_place_info* out, int max_entries) {
  if (!c || !out || max_entries <= 0) return 0;
  int n = 0;
  for (const hmj_place_info& e : c->place_log)
    if (n < max_entries) out[n++] = e;
  return n;
}

int hmj_last_plan(hmj_ctx* c, hmj_plan_desc* out) {
  if (!c || !out || out->struct_size < 8) return HMJ_E_ARG;
  const uint32_t room = out->struct_size < sizeof(hmj_plan_desc) ? out->struct_size : (uint32_t)sizeof(hmj_plan_desc);
  hmj_plan_desc p = c->plan;
  p.struct_size = room;
  std::memcpy(out, &p, room);
  return HMJ_OK;
}

int hmj_forget_workloads(hmj_ctx* c) {
  if (!c) return HMJ_E_ARG;
  c->memos.clear();
  c->wm = &c->memo_init;
  c->wm_sig = 0;
  return HMJ_OK;
}

int hmj_last_timing(hmj_ctx* c, hmj_timing* out) {
  if (!c || !out) return HMJ_E_ARG;
  *out = c->timing;
  return HMJ_OK;
}

const char* hmj_strerror(int code) {
  switch (code) {
    case HMJ_OK: return "ok";
    case HMJ_E_ARG: return "invalid argument";
    case HMJ_E_NODEV: return "no usable HIP device";
    case HMJ_E_OOM: return "out of device or pinned host memory";
    case HMJ_E_HIP: return "HIP runtime error";
    case HMJ_E_UNSUPPORTED: return "unsupported flag combination for this input";
    case HMJ_E_RCCL: return "RCCL / transport error";
    case HMJ_E_PEER: return "another rank of the collective failed";
    case HMJ_E_TIMEOUT: return "the exchange step timed out (a peer never took part); the communicator was aborted";
    default: return "unknown error";
  }
}

const char* hmj_last_error(hmj_ctx* c) { return c ? c->last_error.c_str() : ""; }
const char* hmj_version(void) { return "hashmergejoin_amd 0.5 (gfx950)"; }
int hmj_abi_version(void) { return HMJ_ABI_VERSION; }

int hmj_reserve(hmj_ctx* c, uint64_t n_build, uint64_t n_probe, uint64_t max_matches,
                uint32_t flags) {
  if (!c) return HMJ_E_ARG;
  if (n_build > 0xFFFFFFFFull || n_probe > 0xFFFFFFFFull) return fail(c, HMJ_E_ARG, "too many rows");
  HIP_TRY(hipSetDevice(c->device));
  c->prep.valid = false;  // "any other call discards the prepared state" (hmj.h)
  struct InReserve {  // the caller asked for the workspace ahead of time: big partition buffers may search for fast memory
    hmj_ctx* c;
    explicit InReserve(hmj_ctx* c_) : c(c_) { c->in_reserve = true; }
    ~InReserve() { c->in_reserve = false; }
  } in_reserve(c);
  int B, passes, pass_bits[4], rc;
  plan_bits(n_build, c->force_bits, &B, &passes, pass_bits);
  const size_t P = (size_t)1 << B;
  const size_t items = P < 4096 ? 4096 : P;
  // which partitioning path joins of these sizes take: the histogram-free slab path needs its slabs and nothing of the
  // exact path's ping-pong buffers (4 x 4.5 GB at 2^28 rows, which round 3 reserved -- and probed -- for nothing; a join
  // that falls back to the exact path on skewed keys creates them then)
  hmj::SlabGeom gr, gs;
  const bool count_mode = !(flags & (HMJ_MATERIALIZE | HMJ_ORDERED));
  const bool slab_plan = c->slab_mode && passes == 2 && pass_bits[0] <= hmj::SLAB_MAX_BITS && pass_bits[1] <= hmj::SLAB_MAX_BITS &&
                         slab_sizes_ok(c, n_build, n_probe, !count_mode) &&
                         hmj::slab_geometry((u32)n_build, pass_bits[0], pass_bits[1], &gr) &&
                         hmj::slab_geometry((u32)n_probe, pass_bits[0], pass_bits[1], &gs);
  if (passes >= 1 && !slab_plan) {
    if ((rc = ensure_dev(c, c->rbuf[0], n_build * 16)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->sbuf[0], n_probe * 16)) != HMJ_OK) return rc;
  }
  if (passes >= 2 && !slab_plan) {
    if ((rc = ensure_dev(c, c->rbuf[1], n_build * 16)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->sbuf[1], n_probe * 16)) != HMJ_OK) return rc;
  }
  if (slab_plan) {
    const u64 rows_a = gr.rows_a > gs.rows_a ? gr.rows_a : gs.rows_a;
    const u32 wa = gr.WA > gs.WA ? gr.WA : gs.WA;
    if ((rc = ensure_dev(c, c->slab_a, rows_a * 16)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->cnt_a, ((size_t)wa << pass_bits[0]) * 4)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->slab_br, gr.rows_b * 16)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->slab_bs, gs.rows_b * 16)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->cnt_br, P * gr.KB * 4)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->cnt_bs, P * gs.KB * 4)) != HMJ_OK) return rc;
  }
  if ((rc = ensure_dev(c, c->hist, (size_t)hmj::RP_MAXD * hmj::RP_MAX_BLOCKS * 4)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->totals, (size_t)hmj::RP_MAXD * 4)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->r_off, (P + 1) * 4)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->s_off, (P + 1) * 4)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->accum, 64)) != HMJ_OK) return rc;
  if ((rc = ensure_host(c, c->h_accum, 64)) != HMJ_OK) return rc;
  if (flags & (HMJ_MATERIALIZE | HMJ_ORDERED)) {
    if ((rc = ensure_dev(c, c->part_count, items * 8)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->part_out_off, (items + 1) * 8)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->out_key, max_matches * 8)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->out_rval, max_matches * 8)) != HMJ_OK) return rc;
    if ((rc = ensure_dev(c, c->out_sval, max_matches * 8)) != HMJ_OK) return rc;
    if (flags & HMJ_ORDERED) {
      if ((rc = ensure_dev(c, c->ord_key, max_matches * 8)) != HMJ_OK) return rc;
      if ((rc = ensure_dev(c, c->ord_rval, max_matches * 8)) != HMJ_OK) return rc;
      if ((rc = ensure_dev(c, c->ord_sval, max_matches * 8)) != HMJ_OK) return rc;
    }
  }
  return HMJ_OK;
}

int hmj_prepare_build_u64_device(hmj_ctx* c, const void* build_aos_dev, uint64_t n_build,
                                 uint64_t n_probe_hint) {
  if (!c) return HMJ_E_ARG;
  if (n_probe_hint > 0xFFFFFFFFull) return fail(c, HMJ_E_ARG, "too many rows");
  HIP_TRY(hipSetDevice(c->device));
  spans_reset(c);
  int rc = prepare_build(c, build_aos_dev, n_build, n_probe_hint);
  if (c->profiling) {
    (void)hipStreamSynchronize(c->stream);
    spans_collect(c);
  }
  return rc;
}

int hmj_join_u64_device(hmj_ctx* c, const void* build_aos_dev, uint64_t n_build,
                        const void* probe_aos_dev, uint64_t n_probe, uint32_t flags,
                        hmj_result* out) {
  if (!c) return HMJ_E_ARG;
  HIP_TRY(hipSetDevice(c->device));
  spans_reset(c);
  int st = span_begin(c, K_TOTAL, -1);
  int rc = join_device(c, build_aos_dev, n_build, probe_aos_dev, n_probe, flags, out, false);
  span_end(c, st);
  if (c->profiling) {
    (void)hipStreamSynchronize(c->stream);
    spans_collect(c);
  }
  return rc;
}

// The host-resident entry point (what a caller of the reference ctor experiences, hashjoin_bench.cc:126-133) as a
// pipeline:
//   copy stream   :  H2D R | H2D S chunk 0 | chunk 1 | ...
//   compute stream:        | partition R   | pass A of S per arrived chunk (slab path) ... | pass B, build + probe | D2H
// The build side is partitioned while the probe side is still on the PCIe link (hmj_prepare_build; dropped if the
// join plans differently), and the probe side's first pass follows its chunks -- the machinery the multi-GPU
// exchange uses for rows arriving over xGMI.  The result columns go back in one copy per column: they exist only
// once the last partition is probed, and the 2^26-row join's whole GPU part is 4 ms beside 38 ms up and 29 ms down,
// so the PCIe link bounds the call.
static int join_host_impl(hmj_ctx* c, const void* build_aos_host, uint64_t n_build,
                          const void* probe_aos_host, uint64_t n_probe, uint32_t flags,
                          hmj_result* out) {
  if (!c) return HMJ_E_ARG;
  if (n_build > 0xFFFFFFFFull || n_probe > 0xFFFFFFFFull) return fail(c, HMJ_E_ARG, "too many rows");
  if ((n_build && !build_aos_host) || (n_probe && !probe_aos_host))
    return fail(c, HMJ_E_ARG, "relation pointer is NULL");
  HIP_TRY(hipSetDevice(c->device));
  spans_reset(c);
  int rc;
  if ((rc = ensure_dev(c, c->in_r, n_build * 16 + 16)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->in_s, n_probe * 16 + 16)) != HMJ_OK) return rc;
  const auto t0 = std::chrono::steady_clock::now();
  // small inputs, the staged multi-thread upload, or no pipeline wanted (HMJ_HOST_PIPELINE=0): the serial form
  constexpr uint64_t kPipeMinRows = 1ull << 22;
  const int kChunks = 8;
  const bool pipeline = c->host_pipeline && !c->staged_upload && n_build >= kPipeMinRows && n_probe >= kPipeMinRows;
  if (!pipeline) {
    if ((rc = upload_host(c, c->in_r.p, build_aos_host, n_build * 16)) != HMJ_OK) return rc;
    if ((rc = upload_host(c, c->in_s.p, probe_aos_host, n_probe * 16)) != HMJ_OK) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    const float ms_h2d = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    int st = span_begin(c, K_TOTAL, -1);
    rc = join_device(c, c->in_r.p, n_build, c->in_s.p, n_probe, flags, out, true);
    span_end(c, st);
    if (c->profiling) {
      (void)hipStreamSynchronize(c->stream);
      spans_collect(c);
      c->timing.ms_h2d = ms_h2d;  // wall clock: staging threads + PCIe
      c->timing.ms_total += ms_h2d;
    }
    return rc;
  }
  if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  while ((int)c->copy_ev.size() < kChunks + 2) {
    hipEvent_t e;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->copy_ev.push_back(e);
  }
  // the staging buffers may still be read by earlier work on the compute stream
  HIP_TRY(hipEventRecord(c->copy_ev[kChunks + 1], c->stream));
  HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->copy_ev[kChunks + 1], 0));
  // From here on copies that read the CALLER'S buffers are queued: whichever way this function is left, they have
  // completed first (free on success -- every join path has waited for all arrive events by then)
  struct CopySync {
    hipStream_t s;
    ~CopySync() { (void)hipStreamSynchronize(s); }
  } copy_sync{c->copy_stream};
  HIP_TRY(hipMemcpyAsync(c->in_r.p, build_aos_host, n_build * 16, hipMemcpyHostToDevice, c->copy_stream));
  HIP_TRY(hipEventRecord(c->copy_ev[kChunks], c->copy_stream));
  std::vector<u64> ends;
  for (int i = 0; i < kChunks; i++) {
    const u64 lo = n_probe * (u64)i / kChunks, hi = n_probe * (u64)(i + 1) / kChunks;
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(c->in_s.p) + lo * 16, static_cast<const char*>(probe_aos_host) + lo * 16,
                           (hi - lo) * 16, hipMemcpyHostToDevice, c->copy_stream));
    HIP_TRY(hipEventRecord(c->copy_ev[i], c->copy_stream));
    ends.push_back(hi);
  }
  int st = span_begin(c, K_TOTAL, -1);
  HIP_TRY(hipStreamWaitEvent(c->stream, c->copy_ev[kChunks], 0));  // R is on the device
  c->sample_build_only = true;  // (the key sample must not read probe rows that are still on the link)
  // (a materialising join with the larger probe side plans by the probe rows: a build side prepared for the count
  //  plan would only be partitioned a second time)
  const bool prep_pays = !((flags & (HMJ_MATERIALIZE | HMJ_ORDERED)) && n_probe > n_build);
  rc = prep_pays ? prepare_build(c, c->in_r.p, n_build, n_probe) : HMJ_OK;
  if (rc == HMJ_OK) {
    c->arrive_rows.assign(ends.begin(), ends.end());
    c->arrive_ev.assign(c->copy_ev.begin(), c->copy_ev.begin() + kChunks);
    rc = join_device(c, c->in_r.p, n_build, c->in_s.p, n_probe, flags, out, true);
  }
  c->sample_build_only = false;
  c->arrive_rows.clear();
  c->arrive_ev.clear();
  span_end(c, st);
  if (c->profiling) {
    (void)hipStreamSynchronize(c->stream);
    spans_collect(c);
    c->timing.path |= HMJ_PATH_HOST_PIPELINE;
  }
  return rc;
}

int hmj_join_u64(hmj_ctx* c, const void* build_aos_host, uint64_t n_build,
                 const void* probe_aos_host, uint64_t n_probe, uint32_t flags, hmj_result* out) {
  return join_host_impl(c, build_aos_host, n_build, probe_aos_host, n_probe, flags, out);
}

struct hmj_rows {
  HostBuf key, rval, sval;
};

int hmj_join_u64_rows(hmj_ctx* c, const void* build_aos_host, uint64_t n_build,
                      const void* probe_aos_host, uint64_t n_probe, uint32_t flags, hmj_result* out,
                      hmj_rows** rows) {
  if (!rows) return c ? fail(c, HMJ_E_ARG, "rows is NULL") : HMJ_E_ARG;
  *rows = nullptr;
  int rc = join_host_impl(c, build_aos_host, n_build, probe_aos_host, n_probe, flags | HMJ_MATERIALIZE, out);
  if (rc != HMJ_OK) return rc;
  hmj_rows* r = new hmj_rows();
  if (out->n_matches) {  // hand the pinned columns over; the ctx will take fresh ones from the pool
    r->key = c->h_key;
    r->rval = c->h_rval;
    r->sval = c->h_sval;
    c->h_key = HostBuf();
    c->h_rval = HostBuf();
    c->h_sval = HostBuf();
  }
  *rows = r;
  return HMJ_OK;
}

void hmj_rows_free(hmj_rows* r) {
  if (!r) return;
  pool_give(r->key);
  pool_give(r->rval);
  pool_give(r->sval);
  delete r;
}

uint64_t hmj_host_pool_trim(uint64_t keep_bytes) {
  std::lock_guard<std::mutex> g(g_pool_mu);
  return (uint64_t)pool_trim_locked((size_t)keep_bytes);
}

uint64_t hmj_host_pool_bytes(void) {
  std::lock_guard<std::mutex> g(g_pool_mu);
  return (uint64_t)g_pool_bytes;
}

int hmj_set_host_threads(hmj_ctx* c, int n) {
  if (!c || n < 0 || n > 64) return HMJ_E_ARG;
  c->host_threads = n;
  return HMJ_OK;
}

void hmj_release_result(hmj_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  free_dev(c->out_key);
  free_dev(c->out_rval);
  free_dev(c->out_sval);
  free_dev(c->cat_key);
  free_dev(c->cat_rval);
  free_dev(c->cat_sval);
  free_dev(c->ord_key);
  free_dev(c->ord_rval);
  free_dev(c->ord_sval);
  free_host(c->h_key);
  free_host(c->h_rval);
  free_host(c->h_sval);
}

int hmj_partition_u64_device(hmj_ctx* c, const void* in_aos_dev, uint64_t n, int shift, int bits,
                             void* out_aos_dev, uint64_t* offsets_dev) {
  if (!c) return HMJ_E_ARG;
  if (bits < 1 || bits > hmj::RP_MAX_BITS || shift < 0 || shift + bits > 64)
    return fail(c, HMJ_E_ARG, "bits must be in 1..9 and shift+bits <= 64");
  int rc;
  if ((rc = check_rel(c, in_aos_dev, n, "in_aos is NULL")) != HMJ_OK) return rc;
  if ((rc = check_rel(c, out_aos_dev, n, "out_aos is NULL")) != HMJ_OK) return rc;
  if (!offsets_dev) return fail(c, HMJ_E_ARG, "offsets_dev is NULL");
  HIP_TRY(hipSetDevice(c->device));
  spans_reset(c);
  if (n == 0) {
    HIP_TRY(hipMemsetAsync(offsets_dev, 0, ((size_t)(1u << bits) + 1) * 8, c->stream));
  } else if ((rc = radix_pass(c, in_aos_dev, out_aos_dev, (u32)n, shift, bits, -1,
                              (u64*)offsets_dev)) != HMJ_OK) {
    return rc;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->profiling) spans_collect(c);
  return HMJ_OK;
}

int hmj_sort_u64_device(hmj_ctx* c, const void* in_aos_dev, uint64_t n, void* out_aos_dev) {
  if (!c) return HMJ_E_ARG;
  int rc;
  if ((rc = check_rel(c, in_aos_dev, n, "in_aos is NULL")) != HMJ_OK) return rc;
  if ((rc = check_rel(c, out_aos_dev, n, "out_aos is NULL")) != HMJ_OK) return rc;
  HIP_TRY(hipSetDevice(c->device));
  spans_reset(c);
  if (n == 0) return HMJ_OK;
  memo_for(c, workload_signature(n, 0, 0, 1));  // (what earlier sorts of this size learnt: the chain's cool-down)
  c->prep.valid = false;  // rbuf[0] is the sort's ping-pong buffer
  if ((rc = ensure_dev(c, c->rbuf[0], (size_t)n * 16)) != HMJ_OK) return rc;
  // Digits in which no key differs need no pass (a stable pass on a constant digit is a copy): integer ids below 2^32
  // sort in four passes instead of eight (dense keys, 2^26 rows: 5.58 -> 2.93 ms, 2^28: 21.4 -> 10.8 ms = 24.8 G keys/s,
  // tools/exp_cliffs_sort.py; from 2^22 rows on -- below, the two host round trips cost more than the passes saved).  A sample of the keys
  // says whether that can be the case at all -- uniform 64-bit keys then pay nothing --, one pass over all keys (16 of a
  // radix pass's 48 bytes per row) makes it exact.
  u32 digits = 0xFFu;  // bit d: the 8-bit digit d varies
  u64 diff = ~0ull, kmin = 0, kmax = ~0ull;  // the bits in which keys differ; exact (with the extremes) when != ~0
  u64 sdiff = ~0ull;                          // the bits in which the SAMPLED keys differ
  if (n >= (1u << 22)) {
    if ((rc = ensure_dev(c, c->offs64, 8 * sizeof(u64))) != HMJ_OK) return rc;
    if ((rc = ensure_host(c, c->h_accum, 8 * sizeof(u64))) != HMJ_OK) return rc;
    u64* h = (u64*)c->h_accum.p;
    HIP_TRY(hmj::launch_key_sample(in_aos_dev, (u32)n, nullptr, 0u, (u64*)c->offs64.p, c->stream));
    HIP_TRY(hipMemcpyAsync(h, c->offs64.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    auto varying = [](u64 mask) {
      u32 d = 0;
      for (int i = 0; i < 8; i++) d |= ((mask >> (8 * i)) & 0xFFu) ? (1u << i) : 0u;
      return d;
    };
    sdiff = h[0];
    if (varying(h[0]) != 0xFFu) {
      const u64 init[3] = {0, ~0ull, 0};
      HIP_TRY(hipMemcpyAsync(c->offs64.p, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
      HIP_TRY(hmj::launch_key_exact(in_aos_dev, (u32)n, nullptr, 0u, 0, (u64*)c->offs64.p, c->num_cus, c->stream, true));
      HIP_TRY(hipMemcpyAsync(h, c->offs64.p, 3 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      digits = varying(h[0]);
      diff = h[0];
      kmin = h[1];
      kmax = h[2];
    }
  }
  const int k = __builtin_popcount(digits);
  void* const tmp = c->rbuf[0].p;
  // From 2^25 rows on the passes are a CHAIN of histogram-free slab passes (slab_chain: 32 instead of 48 B per row and pass)
  // and one compaction of the last pass's pieces into the output (32 B per row): uniform 64-bit keys 9 x 32 instead of 8 x 48
  // B per row.  A digit is trimmed to the bits that vary inside it (ids below 2^28: 8 + 8 + 8 + 4 bits), the top digit's slabs
  // are sized for the share of its values in use; digits that are not evenly filled overflow a slab -> the exact passes below
  // (the input is untouched: the chain writes only its own buffers), and the chain is left alone for the next 8 sorts.
  bool chained = false;
  // Round 5, out-of-place sorts from 2^22 rows on: MSD.  The join's two slab passes partition the rows on the TOP 12 ... 18
  // varying key bits, one workgroup per partition sorts its run on the remaining bits in LDS and writes it at the partition's
  // offset (gtable.hip, sort_runs_write_kernel): 3 x 32 B per row whatever the key width, against one 32-byte pass per varying
  // digit + a compaction in the chain below (uniform 64-bit keys: 9 x 32 B).  The window's top is the highest bit in which
  // keys differ (exact), or bit 63 when only the sample is known; partitions hold ~1000 rows and at most 2048.  Keys that
  // crowd into few partitions, or runs with many equal keys, raise an error word: the input is untouched, the chain runs,
  // and this size is left alone for 8 sorts.
  if (c->wm->sort_msd_cooldown > 0) c->wm->sort_msd_cooldown--;
  else if (c->sort_msd && c->slab_mode && out_aos_dev != in_aos_dev && n >= c->sort_msd_min && n < 0xFFFFFFF0ull) {
    const bool exact = diff != ~0ull;
    const int hi = exact ? (diff ? 64 - __builtin_clzll(diff) : 0) : 64;  // keys agree in every bit from `hi` up
    // how much of the window's top digit is in use: exact -> from the extremes; sampled -> a top bit the sample never saw
    // varying halves it (63-bit keys)
    // the narrowest window whose partitions average <= 1200 rows, counting that the keys may fill only part of it
    int TB = 1;
    double dens = 1.0, mean = (double)n;
    do {
      TB++;
      if (TB > hi) break;
      if (exact) {
        const u64 in_use = (kmax >> (hi - TB)) - (kmin >> (hi - TB)) + 1;
        dens = std::ldexp(1.0, TB) / (double)in_use;
      } else if (sdiff != ~0ull) {
        dens = std::ldexp(1.0, __builtin_clzll(sdiff));  // (the sample's top varying bit is below bit 63)
        if (dens > 64.0) dens = 64.0;
      }
      mean = dens * (double)n / std::ldexp(1.0, TB);
      // (a 17th / 18th bit makes a pass a 9-bit pass -- 1.95 / 2.17 ms per 2^28 rows against 1.5 / 1.6 -- and is only worth
      //  partitions beyond the 512-thread sort's comfort: 2^27 / 2^28 rows 3.43 / 7.06 -> 3.32 / 6.84 ms, profiles/r05ad_*)
    } while (TB < c->sort_msd_max_bits && mean > (TB >= 16 ? 2.0 : 1.0) * c->sort_msd_mean);
    if (TB > hi) TB = 0;
    // (beyond 4.5 * 10^8 rows the 2^18 partitions outgrow the 256-thread sort: 512 / 1024 threads hold 4096 / 8192 rows)
    int level = 0;
    while (level < c->rank_runs_max_level && mean + 8.0 * std::sqrt(mean) + 24.0 > (double)hmj::rank_sort_max_run(level)) level++;
    const int bb = TB / 2, ba = TB - bb, shift_a = hi - TB, shift_b = shift_a + ba;
    hmj::SlabGeom g;
    if (TB >= 2 && ba <= hmj::SLAB_MAX_BITS && mean + 8.0 * std::sqrt(mean) + 24.0 <= (double)hmj::rank_sort_max_run(level) &&
        hmj::slab_geometry((u32)n, ba, bb, &g, 0, 1.0, dens)) {
      const u32 P = 1u << TB;
      if ((rc = ensure_dev(c, c->accum, 8 * sizeof(u64))) != HMJ_OK) return rc;
      if ((rc = ensure_host(c, c->h_accum, 8 * sizeof(u64))) != HMJ_OK) return rc;
      if ((rc = ensure_dev(c, c->slab_a, g.rows_a * 16)) != HMJ_OK) return rc;
      if ((rc = ensure_dev(c, c->cnt_a, ((size_t)g.WA << ba) * 4)) != HMJ_OK) return rc;
      if ((rc = ensure_dev(c, c->slab_bs, g.rows_b * 16)) != HMJ_OK) return rc;
      if ((rc = ensure_dev(c, c->cnt_bs, (size_t)P * g.KB * 4)) != HMJ_OK) return rc;
      if ((rc = ensure_dev(c, c->part_out_off, ((size_t)P + 1) * 8)) != HMJ_OK) return rc;
      if ((rc = ensure_dev(c, c->part_count, ((size_t)P / 1024 + 1) * 8)) != HMJ_OK) return rc;
      u64* acc = (u64*)c->accum.p;
      u64* hh = (u64*)c->h_accum.p;
      HIP_TRY(hipMemsetAsync(c->accum.p, 0, 8 * sizeof(u64), c->stream));
      int sp = span_begin(c, K_SCATTER, -1, 0);
      HIP_TRY(hmj::launch_slab_a(in_aos_dev, (u32)n, shift_a, ba, g, c->slab_a.p, c->slab_a.cap / 16, (u32*)c->cnt_a.p, c->cnt_a.cap / 4, acc, c->stream));
      span_end(c, sp);
      sp = span_begin(c, K_SCATTER, -1, 1);
      HIP_TRY(hmj::launch_slab_b(c->slab_a.p, (const u32*)c->cnt_a.p, ba, shift_b, bb, g, c->slab_bs.p, c->slab_bs.cap / 16, (u32*)c->cnt_bs.p,
                                 c->cnt_bs.cap / 4, acc, c->stream));
      span_end(c, sp);
      HIP_TRY(hmj::launch_slab_offsets((const u32*)c->cnt_bs.p, P, (u64*)c->part_out_off.p, (u64*)c->part_count.p, c->stream));
      sp = span_begin(c, K_ORDER, -1);
      HIP_TRY(hmj::launch_sort_runs_write(c->slab_bs.p, (const u32*)c->cnt_bs.p, g.CB, P, (const u64*)c->part_out_off.p, out_aos_dev, acc, level,
                                          c->num_cus, c->stream));
      span_end(c, sp);
      HIP_TRY(hipMemcpyAsync(hh, c->accum.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (hh[hmj::ACC_ERR] & (hmj::ERR_SLAB | hmj::ERR_FASTPATH)) {
        c->wm->sort_msd_cooldown = 8;
        HIP_TRY(hipMemsetAsync((u64*)c->accum.p + hmj::ACC_ERR, 0, sizeof(u64), c->stream));
        std::vector<Span> keep;
        c->spans.swap(keep);
        if (c->trace) std::fprintf(stderr, "[hmj] sort n=%llu: MSD form gave up (%s) -> chain of LSD passes\n", (unsigned long long)n,
                                   (hh[hmj::ACC_ERR] & hmj::ERR_SLAB) ? "a slab overflowed" : "a run beyond the kernel, or many equal keys");
      } else {
        c->timing.bytes_scatter += 2 * 32ull * n;
        c->timing.path |= HMJ_PATH_SLAB | HMJ_PATH_SORT_MSD;
        c->timing.radix_bits = TB;
        c->timing.radix_passes = 2;
        if (c->profiling) spans_collect(c);
        return HMJ_OK;
      }
    }
  }
  if (c->wm->sort_slab_cooldown > 0) c->wm->sort_slab_cooldown--;
  else if (k >= 2 && c->slab_mode && c->sort_slab && n >= c->sort_slab_min && n < 0xFFFFFFF0ull) {
    // exact: `diff` holds every bit in which keys differ (digits are trimmed to them).  Otherwise the sample saw all eight
    // digits vary and `sdiff` holds the bits IT saw varying: every digit keeps its eight bits (a bit the sample missed must
    // still be sorted on), but a bit of a digit that (as far as the sample tells) never varies halves the values in use, so
    // the rows crowd into the others -- 63-bit keys fill half of the top digit -- and the slabs are sized for that.
    const bool exact = diff != ~0ull || sdiff == ~0ull;
    hmj_host::ChainDigit dg[8];
    int nd = 0;
    for (int d = 0; d < 8; d++) {
      const u32 sub = (u32)((exact ? diff : sdiff) >> (8 * d)) & 0xFFu;
      if (exact && !sub) continue;
      const int lo = exact ? __builtin_ctz(sub) : 0, hi = exact ? 31 - __builtin_clz(sub) : 7;
      dg[nd].shift = 8 * d + lo;
      dg[nd].bits = hi - lo + 1;
      dg[nd].dens = std::ldexp(1.0, dg[nd].bits - __builtin_popcount(sub));
      nd++;
    }
    if (exact && diff != ~0ull && nd) {  // (kmin / kmax are the keys' extremes, all bits above the top digit agree)
      const hmj_host::ChainDigit& t = dg[nd - 1];
      const u64 in_use = (kmax >> t.shift) - (kmin >> t.shift) + 1;
      const double dt = std::ldexp(1.0, t.bits) / (double)in_use;
      if (dt > dg[nd - 1].dens) dg[nd - 1].dens = dt;
    }
    bool ok = false;
    u32 pc_n = 0, pc_cap = 0;
    int pc_which = 0;
    if ((rc = hmj_host::slab_chain(c, in_aos_dev, (u32)n, dg, nd, -1, &ok, &pc_n, &pc_cap, &pc_which)) != HMJ_OK) return rc;
    if (ok) {
      DevBuf& so = pc_which ? c->slab_bs : c->slab_a;
      DevBuf& co = pc_which ? c->cnt_bs : c->cnt_a;
      if ((rc = ensure_dev(c, c->piece_off, (size_t)pc_n * 8)) != HMJ_OK) return rc;
      HIP_TRY(hmj::launch_piece_offsets((const u32*)co.p, pc_n, (u64*)c->piece_off.p, c->stream));
      const int sp = span_begin(c, K_SCATTER, -1, nd);
      HIP_TRY(hmj::launch_pieces_compact(so.p, (const u32*)co.p, (const u64*)c->piece_off.p, pc_n, pc_cap, out_aos_dev, c->num_cus, c->stream));
      span_end(c, sp);
      c->timing.bytes_scatter += 32ull * n;
      c->timing.path |= HMJ_PATH_SLAB;
      chained = true;
    } else {
      c->wm->sort_slab_cooldown = 8;
    }
  }
  if (chained) {
  } else if (k == 0) {  // one distinct key: the input order is the sorted order
    if (out_aos_dev != in_aos_dev) HIP_TRY(hipMemcpyAsync(out_aos_dev, in_aos_dev, (size_t)n * 16, hipMemcpyDeviceToDevice, c->stream));
  } else {
    // the passes alternate between the caller's output and the scratch buffer and end in the output -- except in place
    // with an odd number of passes (the first pass must not write what it reads): those end in the scratch buffer
    const bool end_in_out = !(out_aos_dev == in_aos_dev && (k & 1));
    const void* src = in_aos_dev;
    int left = k;
    for (int d = 0; d < 8; d++) {  // LSD: bits [8 d, 8 d + 8)
      if (!(digits & (1u << d))) continue;
      void* dst = ((left & 1) == (end_in_out ? 1 : 0)) ? out_aos_dev : tmp;
      if ((rc = radix_pass(c, src, dst, (u32)n, 8 * d, 8, -1, nullptr)) != HMJ_OK) return rc;
      src = dst;
      left--;
    }
    if (!end_in_out) HIP_TRY(hipMemcpyAsync(out_aos_dev, tmp, (size_t)n * 16, hipMemcpyDeviceToDevice, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->profiling) spans_collect(c);
  return HMJ_OK;
}

// {key, index} pairs in rbuf[0] -> sorted by key (stable), back in rbuf[0]: eight 8-bit LSD passes
static int sort_pairs_in_rbuf(hmj_ctx* c, u64 n) {
  int rc;
  c->prep.valid = false;
  if ((rc = ensure_dev(c, c->rbuf[0], (size_t)n * 16)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->rbuf[1], (size_t)n * 16)) != HMJ_OK) return rc;
  for (int pass = 0; pass < 8; pass++)
    if ((rc = radix_pass(c, c->rbuf[pass & 1].p, c->rbuf[(pass & 1) ^ 1].p, (u32)n, 8 * pass, 8, -1, nullptr)) != HMJ_OK) return rc;
  return HMJ_OK;
}

int hmj_sort_rows_by_u64_host(hmj_ctx* c, void* rows_host, uint64_t n, uint32_t row_bytes, uint32_t key_offset) {
  if (!c) return HMJ_E_ARG;
  if (row_bytes < 16 || row_bytes > 64 || (row_bytes & 7) || (key_offset & 7) || key_offset + 8 > row_bytes)
    return fail(c, HMJ_E_ARG, "row_bytes must be a multiple of 8 in 16..64 and hold the 8-byte key at an 8-byte offset");
  if (n > 0xFFFFFFFFull || (n && !rows_host)) return fail(c, HMJ_E_ARG, "hmj_sort_rows_by_u64_host");
  HIP_TRY(hipSetDevice(c->device));
  spans_reset(c);
  if (n < 2) return HMJ_OK;
  int rc;
  const size_t bytes = (size_t)n * row_bytes;
  const u32 words = row_bytes / 8;
  if ((rc = ensure_dev(c, c->in_r, bytes)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->in_s, bytes)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->rbuf[0], (size_t)n * 16)) != HMJ_OK) return rc;
  HIP_TRY(hipMemcpyAsync(c->in_r.p, rows_host, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hmj::launch_rows_key_idx(c->in_r.p, n, words, key_offset / 8, c->rbuf[0].p, c->stream));
  if ((rc = sort_pairs_in_rbuf(c, n)) != HMJ_OK) return rc;
  HIP_TRY(hmj::launch_rows_gather(c->in_r.p, c->rbuf[0].p, n, words, c->in_s.p, c->stream));
  HIP_TRY(hipMemcpyAsync(rows_host, c->in_s.p, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return HMJ_OK;
}

int hmj_argsort_u64_host(hmj_ctx* c, const void* keys_host, uint64_t n, uint32_t stride_bytes, uint32_t* perm_out) {
  if (!c) return HMJ_E_ARG;
  if (stride_bytes < 8 || n > 0xFFFFFFFFull || (n && (!keys_host || !perm_out))) return fail(c, HMJ_E_ARG, "hmj_argsort_u64_host");
  HIP_TRY(hipSetDevice(c->device));
  spans_reset(c);
  if (n == 0) return HMJ_OK;
  int rc;
  if ((rc = ensure_dev(c, c->in_r, (size_t)n * 8)) != HMJ_OK) return rc;
  if ((rc = ensure_dev(c, c->rbuf[0], (size_t)n * 16)) != HMJ_OK) return rc;
  HIP_TRY(hipMemcpy2DAsync(c->in_r.p, 8, keys_host, stride_bytes, 8, n, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hmj::launch_key_idx((const u64*)c->in_r.p, n, c->rbuf[0].p, c->stream));
  if ((rc = sort_pairs_in_rbuf(c, n)) != HMJ_OK) return rc;
  HIP_TRY(hmj::launch_pairs_val_u32(c->rbuf[0].p, n, (u32*)c->in_r.p, c->stream));
  HIP_TRY(hipMemcpyAsync(perm_out, c->in_r.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return HMJ_OK;
}

#define GEN_PROLOGUE                                                        \
  if (!c) return HMJ_E_ARG;                                                 \
  if (n && !out_aos_dev) return fail(c, HMJ_E_ARG, "out_aos is NULL");      \
  HIP_TRY(hipSetDevice(c->device));                                         \
  if (n == 0) return HMJ_OK;

int hmj_gen_build_u64_device(hmj_ctx* c, void* out_aos_dev, uint64_t n, uint64_t start,
                             uint64_t seed) {
  GEN_PROLOGUE
  HIP_TRY(hmj::launch_gen_build(out_aos_dev, n, start, seed, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return HMJ_OK;
}

int hmj_gen_probe_u64_device(hmj_ctx* c, void* out_aos_dev, uint64_t n, uint64_t start,
                             uint64_t n_build, uint64_t seed, uint64_t miss_mod) {
  GEN_PROLOGUE
  HIP_TRY(hmj::launch_gen_probe(out_aos_dev, n, start, n_build, seed, miss_mod, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return HMJ_OK;
}

int hmj_gen_from_cdf_u64_device(hmj_ctx* c, void* out_aos_dev, uint64_t n, uint64_t start,
                                const uint64_t* thr_dev, uint64_t domain, uint64_t seed,
                                uint64_t zseed) {
  GEN_PROLOGUE
  if (!thr_dev || domain == 0) return fail(c, HMJ_E_ARG, "thr_dev/domain");
  HIP_TRY(hmj::launch_gen_from_cdf(out_aos_dev, n, start, (const u64*)thr_dev, domain, seed, zseed,
                                   c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return HMJ_OK;
}

int hmj_gen_uniform_domain_u64_device(hmj_ctx* c, void* out_aos_dev, uint64_t n, uint64_t start,
                                      uint64_t domain, uint64_t seed, uint64_t zseed) {
  GEN_PROLOGUE
  if (domain == 0) return fail(c, HMJ_E_ARG, "domain");
  HIP_TRY(hmj::launch_gen_uniform_domain(out_aos_dev, n, start, domain, seed, zseed, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return HMJ_OK;
}

}  // extern "C"
