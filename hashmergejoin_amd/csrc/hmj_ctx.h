// Host-side state of one hmj_ctx and the internal entry points shared by api.hip (the join driver) and
// exchange.hip (the multi-GPU partition exchange).  Internal header.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/hmj.h"
#include "hmj_dev.h"
#include "hmj_launch.h"

struct hmj_comm;  // exchange.hip

namespace hmj_host {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};
struct HostBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool pinned = true;  // false: pageable memory on transparent huge pages (the result columns)
};

enum Kind {
  K_TOTAL = 0, K_H2D, K_D2H, K_HIST, K_SCAN, K_SCATTER, K_OFFSETS, K_PROBE_COUNT, K_OUT_SCAN,
  K_PROBE_WRITE, K_ORDER, K_NKINDS
};
struct Span {
  int kind, rel, e0, e1;  // rel: 0 = build side, 1 = probe side, -1 = n/a
  int pass;               // K_SCATTER: 0 = a relation's first radix pass, 1 = a later one
};

// What a context has LEARNT about a workload -- which fast paths gave up on it and are skipped for its next joins, which
// form of the ordered kernels its keys need -- keyed by the workload's signature (log2 of both sizes, the mode flags):
// a join that overflows a slab or meets duplicate build keys changes the plans of later joins OF THE SAME SHAPE only.
// (Until round 4 these were eleven counters of the context: BASELINE configs[4]'s failed slab attempt put the next eight
//  joins of ANY shape on the exact path, VERDICT r4 #4.)  A cool-down of n: the next n joins of this workload skip the path.
struct WorkloadMemo {
  int uniq_cooldown = 0;            // unique-key write mode (after duplicate build keys)
  int sorted_cooldown = 0;          // one-pass ordered write
  bool sorted_chained = false;      // dense output offsets by a chained scan over the partitions (on after an ordered join
                                    // with unmatched probe rows; HMJ_SORTED_WRITE=2 always, 3 never)
  bool sorted_wide = true;          // ordered foreign-key joins may be planned for the 6144-row shape (HMJ_SORTED_WIDE=0: never)
  bool sorted_fk = false;           // the last ordered join's probe keys repeated: start with the foreign-key form
  int sorted_fk_age = 0;
  int gtable_cooldown = 0;          // global table, count modes (after it gave up: duplicates, a clustering key set)
  int gtable_write_cooldown = 0;    // ... materialising (after duplicate build keys)
  int gtable_sort_cooldown = 0;     // ordered by a sort on (key rank, payload) composites
  int gtable_sort_slab_cooldown = 0;  // ... its passes as a chain of slab passes (after a slab overflowed)
  int rank_runs_cooldown = 0;       // ... the rank-run form (after a slab overflowed or a run did not fit one workgroup's sort)
  int rank_lookup_cooldown = 0;     // ... its rank lookup inside pass A (after a probe row without a build row)
  int expand_cooldown = 0;          // ordered expansion (after a partition did not fit the kernel)
  int sort_slab_cooldown = 0;       // hmj_sort_u64_device: chain of slab passes (after a slab overflowed: skewed digits)
  int sort_msd_cooldown = 0;        // ... its MSD form (after keys crowded into few partitions, or runs of equal keys)
  int slab_cooldown = 0;            // histogram-free slab partitioning of both sides (after a slab overflowed: skewed keys)
  int slab_probe_cooldown = 0;      // ... of the probe side only (probe-heavy count joins, one-pass slab walk)
  int one_pass_write_cooldown = 0;  // materialising on the one-pass slab walk (after duplicate build keys)
  int exact_prefix_joins = 0;       // ordered joins that still take the shared key prefix from a pass over ALL keys: the
                                    // sample of an earlier join missed a few keys above an otherwise dense range
  uint32_t cooling() const;         // HMJ_COOL_* bits of the non-zero counters (hmj_plan_desc.cooling)
};

}  // namespace hmj_host

namespace hmj_host {
// The constants of the cost model that chooses, for an ordered join of a small build side under a long probe side, between
// the partitioned one-pass ordered write and the forms that order through the key's rank (api.hip, try_small_build_ordered).
// ALL of them are fits to sweeps on MI355X boxes (profiles/r04i_*, r04l_*, r04p_*, r04u_*, r05k_*): ms per join =
// fixed + ns_per_row * probe rows / 10^6 (+ per-run term).  One struct, so that a port to another part re-measures one place
// (they were literals in three functions); HMJ_ORDERED_MODEL="name=value,..." overrides single fields at
// hmj_create (developer builds).
struct OrderedCostModel {
  // partitioned one-pass ordered write at fan-out f (probe rows per build row)
  double part_fixed_ms = 0.15;
  double part_ns = 0.030, part_ns_per_f = 0.00023;                // whole-run ranking: linear in the run (f < 24, or cheaper than the buckets)
  double part_bucket_ns = 0.043, part_bucket_ns_per_f = 0.0001;  // (build rank, payload position) buckets from f = 24; + per f beyond 128
  double part_epilogue_ns = 0.25;                                 // f > 700: runs beyond the kernel's partitions, write + order epilogue
  // (rank, payload) composites sorted by global LSD passes
  double comp_fixed_ms = 0.65;
  double comp_ns_chain = 0.060, comp_ns_exact = 0.072;  // passes as a chain of slab passes / exact passes
  double comp_ns_wide = 0.145;                          // rank and payload as two words (payloads spanning > 64 - rank bits)
  double comp_ns_beyond_l2 = 0.02;                      // + where the key -> rank table leaves an XCD's L2
  // rank runs: two slab passes on the rank + an LDS sort per run
  double runs_fixed_ms = 0.25, runs_ns = 0.0215, runs_ns_per_run = 7.5;
  double runs_ns_per_log2_build = 0.0023;  // ... + this per probe row and doubling of the build side beyond 2^14 rows (the rank table and the build rows leave the L2: 2^14 / 2^18 / 2^20 x 2^28 rows 8.5 / 11.0 / 12.6 ms)
  double runs_wave_ns_per_run = 1.2, runs_wave_ns = 0.007;  // partitions of <= 512 rows, one wave each: per partition, and on top of runs_ns per row (profiles/r05aa_*)
  double runs_range_ns = 0.0;    // cut runs: the range of the payloads comes from a sample of ~2^17 rows (a pass over all of them cost 0.003)
};
}  // namespace hmj_host

struct hmj_ctx {
  using DevBuf = hmj_host::DevBuf;
  using HostBuf = hmj_host::HostBuf;
  using Span = hmj_host::Span;
  using WorkloadMemo = hmj_host::WorkloadMemo;
  using u32 = hmj::u32;
  using u64 = hmj::u64;
  int device = 0, num_cus = 256;
  hipStream_t own_stream = nullptr, stream = nullptr;
  DevBuf rbuf[2], sbuf[2], in_r, in_s, hist, totals, r_off, s_off, part_count, part_out_off, accum,
      out_key, out_rval, out_sval, ord_key, ord_rval, ord_sval, offs64, irregular, matched, vparts,
      slab_a, slab_br, slab_bs, cnt_a, cnt_br, cnt_bs, lookback, gtab, piece_off,
      split_r, split_s, split_off, cat_key, cat_rval, cat_sval,  // joins by key ranges: both relations cut, the appended result columns
      msd_off;  // the rank forms' build-side sort: partition offsets of its one MSD pass
  HostBuf h_accum, h_key, h_rval, h_sval;
  int host_threads = 0;  // staging threads for pageable input (0 = default)
  std::vector<hipStream_t> up_streams;
  std::vector<hipEvent_t> up_events;  // 2 per staging thread
  std::vector<HostBuf> up_slots;      // 2 per staging thread
  // adaptive state, per workload signature (hmj_host::memo_for); `wm` is the memo of the call in progress
  std::unordered_map<uint64_t, WorkloadMemo> memos;
  WorkloadMemo memo_init;   // what a new workload starts from (HMJ_SORTED_WRITE / HMJ_SORTED_WIDE set its flags)
  WorkloadMemo* wm = &memo_init;
  uint64_t wm_sig = 0;
  hmj_plan_desc plan;       // hmj_last_plan: how the last join was planned and why
  hmj_host::OrderedCostModel ordered_model;
  int force_bits = -1;
  int prefix_bits = -1;  // top key bits known to be constant; -1 = sample the relations (default)
  int min_prefix_bits = 0;  // with sampling: the partition window starts at or below this many top bits (internal: exchange rounds)
  // build side partitioned ahead of the join by hmj_prepare_build_u64_device (one-shot)
  struct Prep {
    bool valid = false, slab = false;
    const void* ptr = nullptr;
    const void* Rp = nullptr;
    u32 n = 0;
    int low = 0, B = 0;
  } prep;
  bool prepare_only = false;
  // placement of big allocations (ensure_dev, api.hip): candidates are probed with a fill and the fastest kept
  int place_tries = 4;     // candidates per allocation when a search runs (HMJ_PLACE=n); a fresh candidate of 6 GB costs 3-450 ms
  float place_budget_ms = 50.f;  // wall-clock budget of one buffer's search (HMJ_PLACE_BUDGET_MS)
  bool place_search_always = false;  // HMJ_PLACE=n set in the environment: joins search too, not only hmj_reserve
  bool in_reserve = false;           // inside hmj_reserve: the caller asked for workspace ahead of time -> search
  hipEvent_t place_ev[2] = {nullptr, nullptr};
  double place_best = 0.0;  // best fill rate (bytes per ms) any probed allocation of this context reached
  bool place_tune = true;  // HMJ_PLACE=0: take the buffers as the driver hands them out
  size_t place_min_bytes = 2048ull << 20;  // only allocations of this size and more are probed (HMJ_PLACE_MIN_MB)
  std::vector<hmj_place_info> place_log;  // one entry per probed buffer (hmj_placement_info)
  bool dense_plan = true;    // HMJ_DENSE_PLAN=0: never size the plan by the build keys' share of the key range
  bool sorted_mode = true;   // HMJ_SORTED_WRITE=0: ordered joins always take write + order epilogue
  bool sorted_chained_forced = false;
  bool sorted_half = true;   // HMJ_SORTED_HALF=0: the foreign-key form never takes its two-workgroups-per-CU shape
  int fk_plan = 0;           // HMJ_FK_PLAN: 0 automatic, 1 wide (6144-row shape), 2 half (3072-row shape, two workgroups per CU), 3 narrow (5120 rows)
  u64 probe_hint = 0;
  // small build sides: one global hash table, probe side streamed unpartitioned (gtable.hip, count modes).  Measured
  // (tools/exp_gtable.py, profiles/r04b_*; 2^26 probe rows, ms per join, partitioned path -> global table): build rows
  // 2^13 1.01 -> 0.42, 2^14 0.98 -> 0.42, 2^15 0.99 -> 0.46, 2^16 1.00 -> 0.58, 2^17 0.99 -> 0.72, 2^18 1.01 -> 1.05 (a
  // 16 MiB table: past an XCD's 4 MiB of L2 every lookup is a 128-byte line from the Infinity Cache), 2^20 1.05 -> 1.5.
  // The load factor matters as much as the size (a wave walks until its longest walk ends): 16 slots per build row
  // while the table stays within 2^18 slots = 4 MiB, never fewer than 4.
  bool gtable_mode = true;         // HMJ_GTABLE=0 disables
  uint64_t gtable_max_rows = 1ull << 17;  // build rows (HMJ_GTABLE_MAX_LOG2); 8 x that for joins of <= 16 x that many rows in all
  u32 gtable_min_fanout = 0;       // probe rows >= this x build rows (HMJ_GTABLE_FANOUT); it wins at every fan-out (2^17 x 2^18: 0.125 -> 0.06 ms)
  uint64_t gtable_min_probe = 1;
  int gtable_wg_per_cu = 8;        // probe grid (HMJ_GTABLE_WG)
  u32 gtable_slots_per_row = 16;   // table slots per build row (HMJ_GTABLE_SLOTS) ...
  int gtable_max_log_cap = 18;     // ... while the table has at most 2^this slots; beyond, down to 4 per row (HMJ_GTABLE_MAX_LOG_CAP)
  bool ltable_mode = true;         // build sides <= 2048 rows (1024 with checksums) under >= 2^16 probe rows, count modes: the table in LDS, one copy per workgroup (HMJ_LTABLE=0, developer builds: the L2-resident table)
  bool gtable_sort_mode = true;    // HMJ_GTABLE_SORT=0: ordered joins of a small build side under a long probe side stay partitioned
  u32 gtable_sort_fanout = 128;    // ... from this many probe rows per build row on (HMJ_GTABLE_SORT_FANOUT)
  bool rank_runs_mode = true;      // ordered, small build side, fan-out from 16: partition by rank, sort every rank's run in LDS (HMJ_RANK_RUNS=0: composites)
  int rank_runs_max_cut = 10;       // ... runs beyond ~1700 rows cut into up to 2^this pieces by payload position (HMJ_RANK_RUNS_MAX_CUT; 0: such joins sort composites)
  int rank_runs_max_group = 3;     // ... more than 2^18 build rows: up to 2^this consecutive ranks share a partition (HMJ_RANK_RUNS_MAX_GROUP; 0: such joins take other paths)
  int rank_runs_max_level = 2;     // the LDS sorts (rank runs, hmj_sort_u64_device's MSD form): workgroups of up to 256 << this threads, 2048 << this rows per partition (HMJ_RANK_RUNS_MAX_LEVEL)
  uint64_t big_join_rows = 1ull << 29;  // probe rows from which the planner is at its 18-bit limit: what the two rules below reason about
  bool promote_to_ordered = true;  // unordered materialising foreign-key joins between the narrow and the wide write's capacity at 18 bits run ordered (HMJ_PROMOTE_TO_ORDERED=0: off)
  bool key_ranges = true;          // ordered foreign-key joins no single plan holds are cut into key ranges joined one after the other (HMJ_KEY_RANGES=0: off)
  int key_ranges_force = 0;        // HMJ_KEY_RANGES_FORCE=n (tests): every ordered device-resident join in 2^n key ranges
  bool rank_runs_wave = true;      // ... partitions of <= 512 rows: one WAVE sorts a partition, four partitions per workgroup (HMJ_RANK_RUNS_WAVE=0: off)
  bool build_sort_msd = true;      // the rank forms sort their build side by ONE radix pass + an LDS sort per partition (HMJ_BUILD_SORT_MSD=0: LSD passes)
  bool gtable_sort_slab = true;    // the composites' LSD passes are histogram-free slab passes chained one into the next (HMJ_GTABLE_SORT_SLAB=0: exact passes)
  u64 gtable_sort_slab_min = 1ull << 25;  // ... from this many composites on (HMJ_GTABLE_SORT_SLAB_MIN_LOG2; below: no gain, 2^24 rows 1.7 ms either way)
  bool expand_mode = true;         // ordered joins with duplicate build keys write their rows in order, partition by partition (HMJ_ORDERED_EXPANSION=0: write + sort)
  u32 fk_payload_buckets = 24;     // the one-pass ordered foreign-key write ranks inside (build rank, payload position) buckets from this fan-out on (HMJ_FK_PAYLOAD_BUCKETS; 0: never)
  u32 expand_fk_fanout = 0;        // ordered foreign-key joins (unique build keys) take the expansion from this fan-out on (HMJ_EXPAND_FK_FANOUT; 0: never)
  bool expand_allow_rebits = true;  // (false during the retry that already took one more bit)
  int expand_rebits = 0;            // the bits that retry plans         // ordered joins to keep on write + sort after a partition did not fit the expansion kernel
  bool sort_msd = true;            // hmj_sort_u64_device, out of place: two MSD slab passes + an LDS sort per partition (HMJ_SORT_MSD=0: the LSD chain)
  double sort_msd_mean = 1200.0;   // ... narrowed until its partitions average at most this many rows (HMJ_SORT_MSD_MEAN)
  int sort_msd_max_bits = 18;      // ... its window: at most this many key bits make the partitions (HMJ_SORT_MSD_MAX_BITS)
  u64 sort_msd_min = 1ull << 22;   // ... from this many rows on (HMJ_SORT_MSD_MIN_LOG2)
  bool sort_slab = true;           // hmj_sort_u64_device: LSD passes as a chain of slab passes + one compaction (HMJ_SORT_SLAB=0: exact passes)
  u64 sort_slab_min = 1ull << 25;  // ... from this many rows on (HMJ_SORT_SLAB_MIN_LOG2)
  int slab_mode = 1;      // 1 = try the histogram-free slab path for plain count joins (HMJ_SLAB=0 disables)
  bool staged_upload = false;  // HMJ_UPLOAD=staged
  bool host_pipeline = true;   // HMJ_HOST_PIPELINE=0: host entry points upload, join and download one after the other
  hipStream_t copy_stream = nullptr;  // host entry points: uploads run here, beside the partitioning on `stream`
  std::vector<hipEvent_t> copy_ev;
  bool split_mode = true;      // HMJ_SPLIT=0: never split oversized probe partitions
  bool window_mode = true;     // HMJ_WINDOW=0: always partition right below the shared key prefix
  u32 slab_min_rows = 1u << 21;  // per relation: every size that plans two passes.  (Round 1 measured the exact path
                                 // faster below 2^25 -- 2^22 0.33 vs 0.45 ms -- and the threshold stayed there while the slab
                                 // kernels got faster; round 3's size sweep, tools/exp_cliffs.py, count mode, exact vs slab:
                                 // 2.7 M rows 0.295 vs 0.251 ms, 2^22 0.359 vs 0.304, 2^24 0.827 vs 0.651, 28.5 M 1.345 vs
                                 // 0.970; ordered and materialising joins alike.)  HMJ_SLAB_MIN_LOG2 overrides (tests).
  u32 slab_probe_kb = 0;  // HMJ_SLAB_PROBE_KB: pieces per partition of the probe-side slabs (0 = 512 >> bits of pass A)
  bool one_pass_slab = true;  // HMJ_ONE_PASS_SLAB=0: count joins of a one-pass plan never leave the probe side in pass-A slabs
  int scatter_variant = 1;  // 1 = write-combining scatter (default), 0 = plain (HMJ_SCATTER=plain)
  bool profiling = false;
  bool trace = false;  // HMJ_TRACE=1: one stderr line per join attempt (plan, paths taken, why an attempt was retried)
#ifdef HMJ_DEV
  u32 dev_ablate = 0;  // developer builds: HMJ_DEBUG_ABLATE, read once at hmj_create
#endif
  std::vector<hipEvent_t> events;
  std::vector<Span> spans;
  int ev_used = 0;
  hmj_timing timing;
  std::string last_error;
  // ---- multi-GPU exchange (exchange.hip) ----
  hmj_comm* comm = nullptr;
  // Probe rows that are still arriving over the links: rows [0, arrive_rows[i]) of the probe relation are
  // complete once arrive_ev[i] has fired (ascending).  The slab path starts pass A on the arrived part; every
  // other path waits for the last event before it touches the probe side.
  std::vector<u64> arrive_rows;
  std::vector<hipEvent_t> arrive_ev;
  bool sample_build_only = false;  // the key sample must not read the probe side (it is still arriving)
  // How the join waits for arrive_ev[i].  nullptr (host pipeline): a device-side wait queued on the ctx stream.  Set by
  // exchange.hip during a step with a deadline: a host wait that gives up (HMJ_E_TIMEOUT) when a peer never sends.
  int (*arrive_wait)(hmj_ctx*, hipEvent_t) = nullptr;
};


namespace hmj_host {
int fail(hmj_ctx* c, int code, const char* what, hipError_t e = hipSuccess);
// the per-phase times and byte counts of one (sub-)join, added to a call's totals (exchange rounds, key ranges); api.hip
void add_timing(hmj_timing* acc, const hmj_timing& t);
int ensure_dev(hmj_ctx* c, DevBuf& b, size_t bytes);
int ensure_host(hmj_ctx* c, HostBuf& b, size_t bytes, bool pinned = true);
void free_dev(DevBuf& b);
void free_host(HostBuf& b);
void spans_reset(hmj_ctx* c);
void spans_collect(hmj_ctx* c);
int span_begin(hmj_ctx* c, int kind, int rel, int pass = 0);
void span_end(hmj_ctx* c, int id);
// the whole local join (planning, retries) on device-resident relations
int join_device(hmj_ctx* c, const void* R, uint64_t n_build, const void* S, uint64_t n_probe, uint32_t flags,
                hmj_result* out, bool to_host);
// one stable radix pass src -> dst (histogram, scan, write-combining scatter); offsets_out: 2^bits + 1 bucket starts
int radix_pass(hmj_ctx* c, const void* src, void* dst, hmj::u32 n, int shift, int bits, int rel, hmj::u64* offsets_out,
               int pass_index = 0);
// partition the build side only (hmj_prepare_build_u64_device without the API prologue)
int prepare_build(hmj_ctx* c, const void* R, uint64_t n_build, uint64_t n_probe_hint);
void comm_destroy(hmj_ctx* c);  // exchange.hip: called by hmj_destroy
// probe rows [.., arrive_rows[i]) are complete after this (device-side wait, or hmj_ctx::arrive_wait on the host)
int wait_arrival(hmj_ctx* c, hipEvent_t ev);
// the signature adaptive state is keyed by, and the memo of that workload (created on first use)
uint64_t workload_signature(uint64_t n_build, uint64_t n_probe, uint32_t flags, int kind);
WorkloadMemo* memo_for(hmj_ctx* c, uint64_t sig);
}  // namespace hmj_host
