/*
 * hmj.h -- C ABI of libhmj_hip.so: the MI355X (gfx950) radix-partitioned hash-join executor.
 *
 * This is the drop-in boundary for ONE path of dryman/HashMergeJoin: radix partition ->
 * bucket-local build -> probe, behind the reference's join operator.  Every entry point cites
 * the reference interface it replaces (file:line in the reference tree).  Plain pointers and
 * sizes only; no C++/STL/torch types cross this line; nothing throws.
 *
 * Relation layout (reference: hashjoin.h:29-31 KeyValVec, strgen.h:24-25):
 *   n x { uint64_t key; uint64_t val; }  ==  std::pair<uint64_t,uint64_t>[n], contiguous AoS.
 * Key hash: std::hash<uint64_t>, the identity (radix_hash.h:353) -- partitions are
 * taken from the most significant bits of the key, as the reference does (radix_hash.h:369).
 * R is the build side, S the probe side; a result row is (key, rval, sval) exactly as
 * HashMergeJoin::iterator::operator* yields it (hashjoin.h:168-173).
 *
 * Semantics: relational equi-join.  For relations whose keys are unique within each relation
 * (the reference generator's invariant, strgen_test.cc:24-33) the result equals what iterating
 * the reference's HashMergeJoin yields, and with HMJ_ORDERED the order equals its iteration order
 * (ascending key).  With duplicate keys: cross product per key (or HMJ_FIRST_WINS); the reference
 * iterator's "staircase"/tail-cut artefact is not reproduced.
 *
 * Threading: an hmj_ctx is not thread-safe; use one per calling thread / per GPU.
 */
#ifndef HMJ_H
#define HMJ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hmj_ctx hmj_ctx;

/* status codes: 0 ok, negative = error (the reference has none: assert/UB, strgen.cc:60) */
#define HMJ_OK 0
#define HMJ_E_ARG (-1)         /* bad argument / size beyond 2^32-1 tuples per call */
#define HMJ_E_NODEV (-2)       /* no usable HIP device */
#define HMJ_E_OOM (-3)         /* device or host allocation failed */
#define HMJ_E_HIP (-4)         /* HIP runtime error, see hmj_last_error() */
#define HMJ_E_UNSUPPORTED (-5) /* the flags cannot be honoured for this input (e.g. HMJ_ORDERED with a */
                               /* single partition's result beyond 2^31-1 rows)                      */

#define HMJ_E_RCCL (-6)        /* RCCL (or the host's transport callbacks) failed, or librccl is absent  */
#define HMJ_E_PEER (-7)        /* a collective call failed on ANOTHER rank; every rank returns together   */
#define HMJ_E_TIMEOUT (-8)     /* a step of the exchange made no progress within hmj_comm_set_timeout_ms: a peer */
                               /* never took part (or died).  The communicator is unusable: hmj_comm_destroy it   */

/* flags for hmj_join_* */
#define HMJ_MATERIALIZE 0x01u /* produce the (key, rval, sval) columns; else count/sums only --   */
                              /* the reduction hashjoin_bench.cc:131-133 performs.  Row order is  */
                              /* unspecified without HMJ_ORDERED (a join may come back ordered)   */
#define HMJ_ORDERED 0x02u     /* rows sorted by (key, rval, sval): HashMergeJoin iteration order  */
                              /* (hashjoin.h:104-154); implies HMJ_MATERIALIZE                    */
#define HMJ_FIRST_WINS 0x04u  /* each probe row pairs with the FIRST build row of its key in      */
                              /* input order: unordered_map::insert, partitioned_hash.h:166-170   */
#define HMJ_CHECKSUM 0x08u    /* also fill xor_fold / mix_sum (parity checks)                     */
#define HMJ_SUM_PROBE 0x10u   /* also fill sum_probe_all (the 'miss inserts 0' sum of             */
                              /* hashjoin_bench.cc:92-96 is sum_probe_all + sum_r w/ FIRST_WINS)  */

typedef struct {
  uint64_t n_matches;
  uint64_t sum_r;         /* sum of rval over result rows, mod 2^64                               */
  uint64_t sum_s;         /* sum of sval over result rows; sum_r + sum_s == hashjoin_bench.cc:132 */
  uint64_t xor_fold;      /* XOR over rows of tmix(key,rval,sval)        (HMJ_CHECKSUM)           */
  uint64_t mix_sum;       /* sum over rows of tmix(key,rval,sval)        (HMJ_CHECKSUM)           */
  uint64_t sum_probe_all; /* sum of val over ALL probe rows              (HMJ_SUM_PROBE)          */
  /* Result columns, n_matches entries each, NULL unless HMJ_MATERIALIZE.  Owned by the ctx,
   * valid until the next hmj_join_* / hmj_release_result / hmj_destroy on it.
   * hmj_join_u64_device: device pointers.  hmj_join_u64: host pointers.                       */
  const uint64_t* key;
  const uint64_t* rval;
  const uint64_t* sval;
} hmj_result;

/* Per-phase device time of the last join (HIP events on the ctx stream) and the algorithmic
 * bytes each phase moves (DESIGN.md "algorithmic bytes").  Valid after hmj_set_profiling(ctx,1). */
typedef struct {
  float ms_total;
  float ms_h2d, ms_d2h;           /* hmj_join_u64 only                                          */
  float ms_partition_build;       /* all radix passes over R                                    */
  float ms_partition_probe;       /* all radix passes over S                                    */
  float ms_hist, ms_scan, ms_scatter; /* the same time split by kernel kind (both relations)    */
  float ms_offsets;               /* partition boundary search                                  */
  float ms_probe_count;           /* build+probe kernel, count/sum mode                         */
  float ms_out_scan;              /* exclusive scan of per-partition match counts               */
  float ms_probe_write;           /* build+probe kernel, materialising mode                     */
  float ms_order;                 /* in-partition sort for HMJ_ORDERED                          */
  int radix_bits;                 /* total partition bits B (2^B partitions)                    */
  int radix_passes;               /* LSD passes per relation                                    */
  int n_scatter_launches;         /* scatter kernel launches in this join                       */
  int n_split_retries;            /* skewed joins: times the virtual-partition table was regrown        */
  uint64_t bytes_scatter;         /* algorithmic bytes of all scatter launches (32 B/tuple)     */
  uint64_t bytes_hist;            /* 16 B/tuple per pass                                        */
  uint64_t bytes_probe_count;     /* 16*(n_build + n_probe)                                     */
  uint64_t bytes_probe_write;     /* 16*(n_build + n_probe) + 24*n_matches                      */
  /* which code paths the last join took -- filled with or without profiling; the tests assert on these
   * instead of on wall-clock budgets                                                                */
  uint32_t path;                  /* HMJ_PATH_* bits                                            */
  int32_t key_prefix_bits;        /* top key bits skipped as shared by all rows (sampled or set)*/
  int32_t key_window_low;         /* lowest key bit of the B-bit partition window               */
  uint32_t n_probe_items;         /* probe work items: (virtual) partitions x probe slices      */
  float ms_scatter_pass[2];       /* ms_scatter split: [0] first radix pass of a relation (slab A),
                                     [1] later passes (slab B); both relations                  */
} hmj_timing;
#define HMJ_PATH_SLAB 0x001u           /* histogram-free slab partitioning                                */
#define HMJ_PATH_EXACT 0x002u          /* histogram + scan + scatter passes                               */
#define HMJ_PATH_UNIQ_WRITE 0x004u     /* materialised in one probe pass (unique build keys)              */
#define HMJ_PATH_SPLIT 0x008u          /* oversized partitions cut into virtual partitions                */
#define HMJ_PATH_WINDOW 0x010u         /* partition window moved below the shared prefix (structured keys)*/
#define HMJ_PATH_ORDER_DEFERRED 0x020u /* ordered epilogue finished by global LSD sorts of the result     */
#define HMJ_PATH_ORDER_BY_KEY 0x040u   /* partitions are not key ranges: final stable sort by key         */
#define HMJ_PATH_PREPARED 0x080u       /* build side taken from hmj_prepare_build_u64_device              */
#define HMJ_PATH_CHUNKED_BUILD 0x100u  /* plan with build partitions beyond the LDS table (forced bits)   */
#define HMJ_PATH_HOT_KEY_HINT 0x200u   /* the key sample saw a repeated key                               */
#define HMJ_PATH_SLAB_PROBE 0x400u     /* probe-heavy count join: build side exact, probe side in slabs   */
#define HMJ_PATH_SORTED_WRITE 0x800u   /* ordered join probed, sorted and written in one pass             */
#define HMJ_PATH_SORTED_FK 0x1000u     /* ... in its foreign-key form (probe keys repeat)                 */
#define HMJ_PATH_DENSE_BUILD 0x2000u   /* build keys cover part of the key range: plan sized by their density */
#define HMJ_PATH_SORTED_FK_HALF 0x8000u /* ... in its small shape: 512-thread workgroups, two per CU                */
#define HMJ_PATH_SORTED_FK_WIDE 0x20000u /* ... in its wide shape: 6144 probe rows per partition (16-bit plan, slab path)    */
#define HMJ_PATH_LOOKBACK_TIMEOUT 0x10000u /* a chained partition gave up waiting for its predecessor (a bug if seen) */
#define HMJ_PATH_PRESORTED 0x40000u /* a relation arrived already partitioned (sorted by key): its radix passes were skipped */
#define HMJ_PATH_SLAB_ONE_PASS 0x80000u /* ... of a ONE-pass plan: the probe kernel reads the pass's worker-private slabs directly */
#define HMJ_PATH_ORDER_BY_RANK_SORT 0x200000u /* ordered, small build side under a long probe side: rows sorted as (key rank, payload) composites */
#define HMJ_PATH_RANK_RUNS 0x1000000u /* ... rows partitioned by key rank -- longer runs: by (rank, piece of the payloads' range) -- with two slab passes, each partition sorted in LDS */
#define HMJ_PATH_RANK_LOOKUP_IN_PASS 0x2000000u /* ... with the key -> rank lookup inside the first slab pass (every probe row had its build row) */
#define HMJ_PATH_SORT_MSD 0x4000000u /* hmj_sort_u64_device: two slab passes on the top varying key bits + an LDS sort of every partition */
#define HMJ_PATH_KEY_RANGES 0x8000000u /* ordered join no single plan holds: both relations cut into key ranges, joined one after the other, rows appended */
#define HMJ_PATH_ORDERED_EXPANSION 0x400000u /* ordered, duplicate build keys: rows written in order partition by partition (no sort of result rows) */
#define HMJ_PATH_LDS_TABLE 0x800000u /* ... of <= 2048 build rows (1024 with HMJ_CHECKSUM / HMJ_SUM_PROBE) under >= 2^16 probe rows, count modes: that table in LDS, one copy per workgroup */
#define HMJ_PATH_GLOBAL_TABLE 0x100000u /* small build side: one global hash table, the probe side streamed unpartitioned */
#define HMJ_PATH_HOST_PIPELINE 0x4000u /* host entry: build side partitioned while the probe side was uploading   */

/* How the last join on this ctx was planned, and why (hmj_last_plan).  The planner chooses among the formulations listed
 * at hmj_join_u64_device from the sizes, the flags, a key sample and what EARLIER JOINS OF THE SAME WORKLOAD taught it:
 * a fast path that gives up (a slab overflowed on skewed keys, the unique-key write met duplicate build keys, the global
 * table's walk got too long) is retried on the general path within the same join and then skipped for the next 8 (64)
 * joins of that workload.  A workload = floor(log2) of both sizes + the mode flags (materialise / ordered / first-wins);
 * joins of other shapes on the same ctx are not affected (round 5; until then the cool-downs belonged to the ctx).
 * Tests assert on these fields instead of pinning paths through the environment.  Nothing in the reference corresponds
 * (its one plan is optimal_partition, radix_hash.h:38-57).                                                         */
typedef struct {
  uint32_t struct_size;   /* bytes the library filled in (callers built against an older header get a prefix)      */
  uint32_t path;          /* HMJ_PATH_* of the attempt that produced the result (= hmj_timing.path)                */
  int32_t radix_bits, radix_passes, pass_bits[4];
  int32_t key_prefix_bits, key_window_low;
  uint32_t n_partitions;  /* 2^radix_bits, or 0 where nothing was partitioned (global / LDS table)                 */
  uint32_t probe_items;   /* work items of the probe phase ((virtual) partitions x probe slices)                   */
  uint32_t attempts;      /* plans executed for this join (1 = the first plan held)                                */
  uint32_t refused;       /* HMJ_REFUSED_*: faster formulations not taken by this join, and why                    */
  uint32_t cooling;       /* HMJ_COOL_*: fast paths this WORKLOAD is skipping after this join (0 = nothing learnt) */
  uint64_t workload;      /* the signature the adaptive state is keyed by                                          */
} hmj_plan_desc;
#define HMJ_REFUSED_GTABLE_SHAPE 0x0001u        /* global table: build side / flags / sizes outside its window          */
#define HMJ_REFUSED_GTABLE_COOLING 0x0002u      /* ... skipped: an earlier join of this workload gave up on it          */
#define HMJ_REFUSED_GTABLE_GAVE_UP 0x0004u      /* ... tried by THIS join and abandoned (duplicate keys, long walks)     */
#define HMJ_REFUSED_RANK_SORT_MODEL 0x0008u     /* ordered small build side: the cost model preferred partitioning       */
#define HMJ_REFUSED_RANK_SORT_COOLING 0x0010u
#define HMJ_REFUSED_RANK_SORT_GAVE_UP 0x0020u
#define HMJ_REFUSED_SLAB_SHAPE 0x0040u          /* slab partitioning: not a two-pass plan of <= 9-bit passes, sizes below */
                                                /* the threshold, or probe partitions beyond the pipelined kernels        */
#define HMJ_REFUSED_SLAB_COOLING 0x0080u
#define HMJ_REFUSED_SLAB_SORTED_INPUT 0x0100u   /* the sample found a relation in key order (its digits are not mixed)   */
#define HMJ_REFUSED_SLAB_OVERFLOW 0x0200u       /* tried by this join: a slab overflowed (skewed digits), exact passes    */
#define HMJ_REFUSED_FAST_WRITE_COOLING 0x0400u  /* unique-key write mode skipped: duplicate build keys seen earlier       */
#define HMJ_REFUSED_FAST_WRITE_GAVE_UP 0x0800u  /* ... tried by this join: duplicate build keys / an oversized partition  */
#define HMJ_REFUSED_SLAB_PROBE_COOLING 0x1000u  /* probe-side-only slabs / one-pass slab walk skipped                     */
#define HMJ_REFUSED_SLAB_PROBE_OVERFLOW 0x2000u
#define HMJ_REFUSED_PREFIX_VIOLATED 0x4000u     /* a row outside the sampled key prefix: the join re-planned              */
#define HMJ_REFUSED_EXPANSION_GAVE_UP 0x8000u   /* ordered expansion: a partition beyond the kernel's capacity            */
#define HMJ_COOL_UNIQ_WRITE 0x001u
#define HMJ_COOL_SORTED_WRITE 0x002u
#define HMJ_COOL_GTABLE 0x004u
#define HMJ_COOL_GTABLE_WRITE 0x008u
#define HMJ_COOL_RANK_SORT 0x010u
#define HMJ_COOL_RANK_SORT_SLAB 0x020u
#define HMJ_COOL_EXPANSION 0x040u
#define HMJ_COOL_SORT_SLAB 0x080u
#define HMJ_COOL_SLAB 0x100u
#define HMJ_COOL_SLAB_PROBE 0x200u
#define HMJ_COOL_ONE_PASS_WRITE 0x400u
#define HMJ_COOL_EXACT_PREFIX 0x800u
#define HMJ_COOL_RANK_RUNS 0x1000u
#define HMJ_COOL_SORT_MSD 0x2000u

/* ---- lifecycle ------------------------------------------------------------------------------- */
/* Replaces: nothing in the reference (no device); one ctx per GPU. device_id < 0 = current.     */
int hmj_create(hmj_ctx** out, int device_id);
void hmj_destroy(hmj_ctx* ctx);
/* Launch on the caller's HIP stream (hipStream_t passed as void*).  NULL is the HIP default (null)
 * stream, exactly as in the HIP API; HMJ_STREAM_OWN selects the ctx's private non-blocking stream
 * (the initial setting).  Device inputs must be complete on, or ordered before, the chosen stream:
 * the private stream does NOT synchronise with work queued on other streams.                      */
#define HMJ_STREAM_OWN ((void*)(intptr_t)-1)
int hmj_set_stream(hmj_ctx* ctx, void* hip_stream);
/* Pre-allocate workspace for joins up to these sizes (so the timed call allocates nothing).     */
int hmj_reserve(hmj_ctx* ctx, uint64_t n_build, uint64_t n_probe, uint64_t max_matches,
                uint32_t flags);
/* Override the planner (the reference's optimal_partition heuristic, radix_hash.h:38-57, is tuned
 * for CPU caches; ours targets LDS capacity).  total_bits < 0 restores automatic planning.       */
int hmj_set_radix_bits(hmj_ctx* ctx, int total_bits);
/* Measure the plan instead of trusting the heuristic (the reference tunes its k with
 * find_k_bench.cc:116-129): joins synthetic relations of the given sizes (unique uniform keys, every
 * probe row matching once -- the hmj_gen_* generators) with B-1, B and B+1 total radix bits, B being
 * hmj_plan's choice, and reports the time of each (ms[0..2]; a candidate that does not exist is < 0).
 * apply != 0 keeps the fastest as the forced plan of this ctx (as hmj_set_radix_bits would).  Needs
 * 16*(n_build+n_probe) bytes of device memory beside the join's workspace.                         */
int hmj_autotune_radix_bits(hmj_ctx* ctx, uint64_t n_build, uint64_t n_probe, int apply, int* best_bits,
                            double ms[3]);
/* Tell the executor that the top `bits` key bits are equal in all rows of both relations (an outer
 * radix split of the caller's already consumed them; hmj_exchange_join_u64_device needs none): partitioning then starts
 * below them, as the reference's recursion masks off consumed bits (radix_hash.h:219-220).
 * bits = -1 (the default): the executor samples both relations and skips the top bits all sampled
 * keys share (dense / small-integer keys would otherwise all fall into partition 0).
 * Any value is safe for the result; ordered output additionally verifies a sampled prefix on every
 * row and re-plans without it if a row disagrees.                                                */
int hmj_set_key_prefix_bits(hmj_ctx* ctx, int bits);
/* The automatic plan for a build side of n_build rows: total bits and per-pass bits (LSD order).*/
int hmj_plan(uint64_t n_build, int* total_bits, int* n_passes, int pass_bits[4]);
/* Placement of the big partition buffers (how fast a buffer can be written is a property of the
 * physical memory behind it).  Every partition buffer of 2 GiB and more is probed when it is created (two fills, the
 * second timed: ~2.4 ms for 6 GB).  A SEARCH for faster memory -- further candidate allocations, the fastest kept --
 * runs only inside hmj_reserve (the caller asked for the workspace ahead of time) or when HMJ_PLACE=n is set in the
 * environment; a join that allocates on its own never searches.  A search keeps at most two candidates alive, tries
 * at most n (default 4) and stops when its wall-clock budget (HMJ_PLACE_BUDGET_MS, default 50 ms per buffer) would be
 * exceeded.  HMJ_PLACE=0: nothing is probed.  One entry per probed buffer of this ctx; returns the number of entries
 * (<= max_entries).  Diagnostic only; nothing in the reference corresponds to it (its buffers are std::vector
 * storage, hashjoin.h:62-63).                                                                                    */
/* hmj_place_info, hmj_timing and hmj_exchange_info are DIAGNOSTIC structs: they grow between releases of this library and
 * carry no size field -- build callers against the header of the library they load (hmj_abi_version() == HMJ_ABI_VERSION).
 * hmj_plan_desc, added later, is size-versioned instead.                                                              */
#define HMJ_ABI_VERSION 5
int hmj_abi_version(void);
#define HMJ_PLACE_MAX_CAND 4
typedef struct {
  char name[16];      /* slab_a, slab_b_build, slab_b_probe, rbuf0/1, sbuf0/1                                   */
  uint64_t bytes;
  float fill_TBps;    /* of the allocation that was kept: second of two sequential fills                         */
  int candidates;     /* allocations tried (1 = the first one was kept without a search)                         */
  float ms_search;    /* host time of probe(s) and search, part of the call that allocated the buffer            */
  int searched;       /* 1: a search was allowed (hmj_reserve / HMJ_PLACE=n); 0: probe only                      */
  int aborted;        /* 1: the search stopped because its budget was reached                                    */
  float budget_ms;
  float cand_ms_alloc[HMJ_PLACE_MAX_CAND]; /* per candidate: hipMalloc                                           */
  float cand_ms_fill[HMJ_PLACE_MAX_CAND];  /* per candidate: the two fills (+ the loser's hipFree)               */
  float cand_TBps[HMJ_PLACE_MAX_CAND];     /* per candidate: fill rate                                           */
} hmj_place_info;
int hmj_placement_info(hmj_ctx* ctx, hmj_place_info* out, int max_entries);
/* out->struct_size must hold sizeof(hmj_plan_desc) of the CALLER's header on entry; at most that many bytes are written. */
int hmj_last_plan(hmj_ctx* ctx, hmj_plan_desc* out);
/* Forget what the ctx has learnt about every workload (all cool-downs, the ordered kernels' adaptive forms).          */
int hmj_forget_workloads(hmj_ctx* ctx);
int hmj_set_profiling(hmj_ctx* ctx, int enabled);
int hmj_last_timing(hmj_ctx* ctx, hmj_timing* out);
const char* hmj_strerror(int code);
const char* hmj_last_error(hmj_ctx* ctx);
const char* hmj_version(void);

/* ---- the join ---------------------------------------------------------------------------------- */
/* Replaces HashMergeJoin<RIter,SIter>::HashMergeJoin(r_begin,r_end,s_begin,s_end,num_threads)
 * (hashjoin.h:56-68) plus the iteration that consumes it (hashjoin.h:183-191, driven as in
 * hashjoin_bench.cc:126-133), and equally the partition_only + partitioned_hash_table + probe loop
 * of hashjoin_bench.cc:88-96 (partitioned_hash.h:82-124, :173-215).
 * Inputs are device-resident, borrowed, read-only, not retained after return.  Blocking: on
 * return `out` is filled and the result columns are complete on the stream.
 * Which formulation runs is the executor's choice (hmj_timing.path says which; results are the same):
 *   - two radix passes + LDS build/probe per partition (the default from ~2^21 rows per side on; histogram-free slab
 *     passes of up to 9 bits, so up to 2^30 rows per side stay at 32 B per row and pass);
 *   - count modes and unordered HMJ_MATERIALIZE, build side of <= 2^17 rows (or a join of <= 2^21 rows in all): ONE global
 *     hash table, the probe side streamed unpartitioned -- the reference's BM_hash_join_raw formulation
 *     (hashjoin_bench.cc:29-63), HMJ_PATH_GLOBAL_TABLE (count modes, <= 2048 build rows: the table in LDS, HMJ_PATH_LDS_TABLE);
 *   - the same modes, build side of 2^17 ... 2^21 rows under a probe side >= 8 x larger: one radix pass, the probe side left
 *     in the worker-private slabs of its slab pass and probed there, HMJ_PATH_SLAB_ONE_PASS;
 *   - HMJ_ORDERED, small build side under a probe side >= 128 x larger (a cost model over fan-out and size decides): the
 *     rows ordered through the RANK of their key among the sorted build keys, HMJ_PATH_ORDER_BY_RANK_SORT -- partitioned
 *     by rank (runs beyond ~1700 rows: by rank and the position of the payload in the payloads' range), every partition
 *     sorted in LDS (HMJ_PATH_RANK_RUNS); where that does not apply (more than 2^18 partitions, probe rows without a
 *     build row in a cut run, a hot key): (rank, payload) composites sorted by global LSD passes.
 * hmj_last_plan says which was taken and why the faster ones were not.                                               */
int hmj_join_u64_device(hmj_ctx* ctx, const void* build_aos_dev, uint64_t n_build,
                        const void* probe_aos_dev, uint64_t n_probe, uint32_t flags,
                        hmj_result* out);
/* Partition the build side ahead of the join (e.g. while the probe side is still arriving over
 * xGMI).  One-shot: the NEXT hmj_join_u64_device on this ctx whose build pointer, row count and plan
 * match (count modes and materialising joins of relations of similar size plan alike; a join that plans
 * differently simply partitions R again) skips re-partitioning R; any other call discards the prepared state.  The caller promises the build rows do not change in between.
 * n_probe_hint: the probe size the join will have (it selects the partitioning path).
 * Corresponds to the first radix_non_inplace_par call of the reference ctor (hashjoin.h:65).       */
int hmj_prepare_build_u64_device(hmj_ctx* ctx, const void* build_aos_dev, uint64_t n_build,
                                 uint64_t n_probe_hint);
/* Same with host-resident relations (what a caller of the reference's ctor holds: pointers into
 * std::vector<std::pair<uint64_t,uint64_t>>).  Copies in over PCIe, joins, copies results out.   */
int hmj_join_u64(hmj_ctx* ctx, const void* build_aos_host, uint64_t n_build,
                 const void* probe_aos_host, uint64_t n_probe, uint32_t flags, hmj_result* out);
void hmj_release_result(hmj_ctx* ctx);

/* Host-resident join whose result columns are DETACHED from the ctx: *rows owns the three
 * host columns (out->key/rval/sval point into them) until hmj_rows_free, independent of later joins
 * on the ctx or of the ctx's lifetime -- the ownership HashMergeJoin's _r_sorted/_s_sorted vectors
 * have in the reference (hashjoin.h:197-198).  Freed buffers return to a process-wide pool of
 * huge-page host memory, so a construct/clear loop like hashjoin_bench.cc:120-134 does not fault its
 * result memory in again every iteration.
 * Implies HMJ_MATERIALIZE.  hmj_rows_free(NULL) is a no-op; it may be called from any thread.     */
typedef struct hmj_rows hmj_rows;
int hmj_join_u64_rows(hmj_ctx* ctx, const void* build_aos_host, uint64_t n_build,
                      const void* probe_aos_host, uint64_t n_probe, uint32_t flags, hmj_result* out,
                      hmj_rows** rows);
void hmj_rows_free(hmj_rows* rows);
/* Released result columns and staging slots return to a process-wide pool of host memory that keeps at
 * most HMJ_HOST_POOL_MAX_MB (environment, default 8192) MiB; beyond that the oldest buffers go back to the
 * OS.  hmj_host_pool_trim(keep) shrinks the pool to at most `keep` bytes now (0 empties it) and returns the
 * bytes released; hmj_host_pool_bytes() reports what the pool holds.  Both are thread-safe.  (The
 * reference frees its _r_sorted/_s_sorted vectors in clear(), hashjoin.h:192-195.)                     */
uint64_t hmj_host_pool_trim(uint64_t keep_bytes);
uint64_t hmj_host_pool_bytes(void);
/* Host threads of the optional staged upload (pageable input -> pinned chunks -> PCIe), used only
 * with HMJ_UPLOAD=staged in the environment; by default each relation goes up in one copy straight
 * from the caller's memory (54 GB/s on the MI355X box).  The reference ctor's num_threads argument,
 * hashjoin.h:58, maps to this.  Default min(8, cores).                                            */
int hmj_set_host_threads(hmj_ctx* ctx, int n);

/* ---- multi-GPU: the radix partition exchange --------------------------------------- */
/* One process per GPU, one ctx per process.  The reference reaches all of its parallelism from the ctor
 * (hashjoin.h:56-68 -> radix_hash.h:375-405: fork-join over threads of one address space); this is the same
 * fork-join over GPUs: every rank passes its row shard of R and S, the ranks exchange rows so that each owns
 * a disjoint set of keys (radix fan-out over ranks; partition p of the probe side only ever meets table p,
 * hashjoin_bench.cc:92-96), and each joins what it owns.
 *
 * Communicator.  RCCL over xGMI: rank 0 makes an id (hmj_comm_unique_id), the host program hands the 128 bytes
 * to every rank by whatever it has (torch.distributed, MPI, a file), every rank calls hmj_comm_init_rank.
 * librccl is loaded at the first of these calls (dlopen: the library itself does not link it; a process that
 * already holds RCCL, e.g. through PyTorch, shares that copy).  Alternatively the host supplies the two
 * collectives itself (hmj_comm_set_transport): an existing communicator, or -- as the tests do -- several
 * ranks on one GPU over gloo.  Both are collective calls: all ranks, same order.                           */
#define HMJ_UNIQUE_ID_BYTES 128
int hmj_comm_unique_id(void* id128);
int hmj_comm_init_rank(hmj_ctx* ctx, int n_ranks, int rank, const void* id128);
typedef struct hmj_transport {
  void* user;
  int n_ranks, rank;
  /* both callbacks return 0 = ok, HMJ_E_TIMEOUT = gave up waiting for a peer (the step then returns HMJ_E_TIMEOUT and
   * the communicator is marked unusable), anything else = failed (HMJ_E_RCCL).
   * every rank contributes `count` values; recv[r * count + i] = rank r's send[i].  Host memory, blocking.  */
  int (*allgather_u64)(void* user, const uint64_t* send, uint64_t* recv, int count);
  /* one round of the all-to-all-v on DEVICE memory: send send_bytes[g] bytes at send_ptrs[g] to rank g and
   * receive recv_bytes[g] bytes from it into recv_ptrs[g] (g == own rank: a local copy).  hip_stream
   * (hipStream_t): the inputs are complete on it and the received bytes must be usable by later work on it
   * -- queue the transfers there, or synchronise it, move the data and return.  0 = ok.                      */
  int (*alltoallv)(void* user, int round, const void* const* send_ptrs, const uint64_t* send_bytes,
                   void* const* recv_ptrs, const uint64_t* recv_bytes, void* hip_stream);
} hmj_transport;
int hmj_comm_set_transport(hmj_ctx* ctx, const hmj_transport* t);
int hmj_comm_destroy(hmj_ctx* ctx); /* also done by hmj_destroy */
/* Message sizes (0 = leave unchanged).  max_message_bytes: no single send exceeds it (default and maximum
 * 2^30: RCCL 2.26 truncates a message of 2 GiB or more); a bucket larger than that goes in rounds.
 * probe_round_bytes: target size of a probe-side message (default 128 MiB): the probe side travels in several
 * rounds so that the local partitioning starts on the rows that have arrived.                              */
int hmj_comm_set_message_bytes(hmj_ctx* ctx, uint64_t max_message_bytes, uint64_t probe_round_bytes);
/* The deadline of ONE hmj_exchange_join_u64_device call, in milliseconds from its start (default 120 000, or
 * HMJ_COMM_TIMEOUT_MS in the environment when the communicator is created; 0 = wait for ever).  The reference's
 * workers all return from the one call that started them (hashjoin.h:56-68 -> radix_hash.h:375-405: pthread_join);
 * ranks in different processes can lose a peer, so every wait of a step that depends on another rank -- the
 * all-gathers, the build side's rounds, each probe round -- is a host-side poll (hipEventQuery / hipStreamQuery +
 * ncclCommGetAsyncError) under this deadline, and a watchdog thread covers a host blocked inside RCCL itself (it calls
 * ncclCommAbort, which is what makes such a call return).  When the deadline passes the call returns HMJ_E_TIMEOUT, and
 * so does every later step on this communicator: hmj_comm_destroy it (that is where a communicator whose stream is still
 * busy is aborted -- RCCL's kernels leave their wait loops, the stream drains -- instead of destroyed) and make a new one
 * (all ranks).  Each rank notices on its own -- there is nobody left to tell
 * it -- so a step ends everywhere within the deadline plus about a second.  Callback transports must bound their own
 * waits by the same number (hmj_comm_get_timeout_ms) and return HMJ_E_TIMEOUT from the callback.  Not during a step. */
int hmj_comm_set_timeout_ms(hmj_ctx* ctx, uint64_t timeout_ms);
int hmj_comm_get_timeout_ms(hmj_ctx* ctx, uint64_t* timeout_ms);

/* The distributed join.  Replaces the HashMergeJoin ctor + iteration (hashjoin.h:56-68, :183-191) for
 * relations sharded by rows over the ranks.  The radix fan-out itself is the owner (round 3): every rank runs the
 * FIRST radix pass of the join on its own shards -- stable histogram / scan / scatter on the top `digit_bits` key
 * bits under the prefix all ranks' keys share (the window is agreed through one all-gather of key samples), exactly
 * pass 1 of the reference's sort (radix_hash.h:313-345, top-bits routing radix_hash.h:369) -- and rank g owns a
 * contiguous RANGE of those digits, chosen from the pooled sample so that the ranks balance.  The digit-major rows
 * then travel in rounds of digit sub-ranges (grouped send/recv on the communicator's own stream; build side first);
 * a round that has arrived is a complete key range of both relations and is joined at once by the remaining
 * radix passes + build + probe, while later rounds are still on the links.  Nothing is partitioned twice.
 * Fallbacks: a key sample whose digits cannot be balanced over the ranks (a few clusters of keys) selects the
 * hash owner floor(mix64(key) * n_ranks / 2^64) with a separate owner split (round 2's path, even for any key
 * set); HMJ_ORDERED uses key-range owners between splitters = quantiles of the pooled sample, so rank g's ordered
 * rows precede rank g+1's and the concatenation in rank order is the reference's iteration order.
 * HMJ_FIRST_WINS is global when rank r's build shard precedes rank r+1's in input order (rows are received
 * source-major inside a round and every pass is stable).
 * local_out: this rank's result (materialising flags: columns are device pointers owned by the ctx; count modes:
 * the sums over the key ranges this rank owns).  global_out (may be NULL): counts and checksums over all ranks,
 * columns NULL.  Collective: all ranks, same flags; an error on one rank (bad sizes, out of memory, a failed
 * local join) makes EVERY rank return -- the failing one its own code, the others HMJ_E_PEER -- instead of
 * leaving its peers blocked in the next collective.  One rank: the plain local join (nothing to exchange),
 * unless hmj_comm_set_self_exchange asked for the whole path.
 * STATUS: the digit-owner path (round 3, the default for non-ordered joins) has been verified with several ranks
 * sharing one GPU over the callback transport and with one rank over RCCL; it has NOT yet run over RCCL between
 * two real GPUs (no such box was available).  hmj_comm_set_owner_path(ctx, HMJ_OWNER_SPLIT) -- or
 * HMJ_EXCHANGE_OWNER=split in the environment when the communicator is created -- selects round 2's owner-split
 * path (hash owner, separate split, ONE local join) for all non-ordered joins instead.                         */
int hmj_exchange_join_u64_device(hmj_ctx* ctx, const void* build_shard_dev, uint64_t n_build_shard,
                                 const void* probe_shard_dev, uint64_t n_probe_shard, uint32_t flags,
                                 hmj_result* local_out, hmj_result* global_out);
/* One-rank communicators only (tests, rehearsals on a one-GPU box): on != 0 runs the whole exchange -- digit
 * pre-pass, rounds through the transport (RCCL self send/recv), per-round joins -- where the default is the plain
 * local join.                                                                                                */
int hmj_comm_set_self_exchange(hmj_ctx* ctx, int on);
/* Which owner function non-ordered distributed joins use (collective setting: the same on every rank).
 * HMJ_OWNER_DIGIT (default): ranges of the first radix pass's digit, per-round joins.  HMJ_OWNER_SPLIT: the hash
 * owner with its separate owner split and one local join -- the path every clustered key set falls back to anyway.
 * Nothing in the reference corresponds (one address space, radix_hash.h:375-405).                              */
#define HMJ_OWNER_DIGIT 0
#define HMJ_OWNER_SPLIT 1
int hmj_comm_set_owner_path(hmj_ctx* ctx, int owner_path);
/* The owner split on its own: rows grouped by owner rank, stably (owner-major; within an owner in input
 * order), offsets_dev[g] = first row of owner g (2^ceil(log2 n_ranks) + 1 uint64, device).  splitters: NULL =
 * hash owner; else n_ranks - 1 ascending keys (host memory) = key-range owner.  in/out must not overlap.   */
int hmj_owner_split_u64_device(hmj_ctx* ctx, const void* in_aos_dev, uint64_t n, int n_ranks,
                               const uint64_t* splitters, void* out_aos_dev, uint64_t* offsets_dev);
typedef struct {
  int n_ranks, owner_mode;            /* owner_mode: 1 = hash of the key, 2 = key ranges (splitters),          */
                                      /* 3 = ranges of the first radix pass's digit, 0 = one rank, no exchange */
  uint32_t rounds_build, rounds_probe;
  uint64_t recv_build, recv_probe;    /* rows this rank owns                                                */
  float ms_split;                     /* digit pre-pass (or owner split) of both shards + the read-back of the */
                                      /* counts (host clock)                                                 */
  float ms_exchange_build, ms_exchange_probe; /* on the communication stream                                */
  float ms_local;                     /* from "build side complete" to the end of the local join(s) (host clock) */
  float ms_total;
  int32_t digit_bits, digit_low;      /* owner_mode 3: digit = (key >> digit_low) & (2^digit_bits - 1)         */
  uint32_t n_subjoins;                /* local joins run (one per round of the digit path)                     */
  uint32_t fallback;                  /* 1: the digit owner was rejected (sample not balanceable): hash owner  */
  float sample_max_share;             /* owner_mode 3: largest sampled share of a rank x n_ranks (1 = even)    */
  float ms_kernels;                   /* device time of this rank's own kernels in the step (pre-pass + local  */
                                      /* joins; HIP events, needs hmj_set_profiling)                           */
  float ms_exposed;                   /* ms_total - ms_kernels: what the exchange and its synchronisation added */
} hmj_exchange_info;
int hmj_last_exchange_info(hmj_ctx* ctx, hmj_exchange_info* out);
/* Digit-range owners and rounds, exported because they are pure host arithmetic (no GPU needed; every rank
 * computes the same plan from the same pooled key sample).  sample_keys: the pooled sample (any order).
 * digit = (key >> digit_low) & (2^digit_bits - 1): the top <= 8 bits under the prefix all sampled keys share.
 * owner_first[g] .. owner_first[g+1]: the digits rank g owns (boundaries where the sample's cumulative count is
 * closest to g / n_ranks); round_first[g][r] .. [r+1]: the digits of rank g that travel in round r.
 * usable = 0: fewer than 2 digits per rank, or the fullest rank would get more than 1.3 x its share.          */
#define HMJ_MAX_RANKS 16
#define HMJ_MAX_ROUNDS 16
typedef struct {
  int32_t usable;
  int32_t digit_bits, digit_low;
  uint32_t n_rounds;
  uint32_t owner_first[HMJ_MAX_RANKS + 1];
  uint32_t round_first[HMJ_MAX_RANKS][HMJ_MAX_ROUNDS + 1];
  float max_share;
} hmj_digit_plan;
int hmj_exchange_digit_plan(int n_ranks, const uint64_t* sample_keys, uint64_t n_sample, uint32_t n_rounds,
                            hmj_digit_plan* out);
/* One relation's messages under a digit plan.  counts[src * 2^digit_bits + d] = rows of digit d in rank src's
 * shard.  For round r and peer g (index r * n_ranks + g): rows [send_off, +send_rows) of this rank's digit-major
 * buffer go to g; rows [recv_off, +recv_rows) of its receive buffer are filled by g.  The receive buffer is
 * round-major, inside a round source-major (sources in rank order = global input order): round r occupies rows
 * [round_off[r], round_off[r+1]) and is a complete key range once every source's message has arrived.        */
int hmj_exchange_digit_layout(int n_ranks, int rank, const hmj_digit_plan* plan, const uint64_t* counts,
                              uint64_t* send_off, uint64_t* send_rows, uint64_t* recv_off, uint64_t* recv_rows,
                              uint64_t* round_off);
/* The round plan of the owner-split path (hash / key-range owners), exported because it is pure host arithmetic
 * (no GPU needed; every rank computes the same from the same count matrix).  counts[src * n_ranks + dst] = rows rank src sends to rank dst.
 * hmj_exchange_rounds: rounds so that no message exceeds max_msg_rows.  hmj_exchange_layout: for round r and
 * peer g (index r * n_ranks + g) the rows [send_off, +send_rows) of this rank's owner-major split buffer that
 * go to g, and the rows [recv_off, +recv_rows) of its receive buffer that g's message fills.  layout 0 =
 * source-major (a source's rows contiguous, sources in rank order = global input order), 1 = round-major
 * (a round's rows contiguous; round_end[r] = rows complete after round r).                                  */
uint32_t hmj_exchange_rounds(int n_ranks, const uint64_t* counts, uint64_t max_msg_rows);
int hmj_exchange_layout(int n_ranks, int rank, const uint64_t* counts, uint32_t n_rounds, int layout,
                        uint64_t* send_off, uint64_t* send_rows, uint64_t* recv_off, uint64_t* recv_rows,
                        uint64_t* round_end);

/* ---- one radix pass ---------------------------------------------------------------------------- */
/* Replaces pass 1 of radix_int_non_inplace / radix_non_inplace_par: per-worker histogram, exclusive
 * scan partition-major/worker-minor, STABLE scatter (radix_sort.h:418-449, radix_hash.h:313-345) on
 * digit = (key >> shift) & (2^bits - 1), 1 <= bits <= 9.  offsets_dev: 2^bits + 1 uint64 bucket
 * starts (device).  in/out: n x {key,val}, device, must not overlap.  Also the multi-GPU owner
 * split.                                                                          */
int hmj_partition_u64_device(hmj_ctx* ctx, const void* in_aos_dev, uint64_t n, int shift, int bits,
                             void* out_aos_dev, uint64_t* offsets_dev);

/* ---- full radix sort ------------------------------------------------------------ */
/* Replaces radix_int_non_inplace<uint64_t,uint64_t>(begin, end, dst, num_threads)
 * (radix_sort.h:452-522) -- the call radix_bench_par.cc:126-127 times: rows sorted by key, ascending,
 * out of place.  From 2^22 rows on (out != in): MSD, as the reference's own sort is (radix_hash.h:202-292) -- two
 * histogram-free slab passes on the top 12 ... 18 varying key bits, every partition sorted on the remaining bits in LDS
 * (HMJ_PATH_SORT_MSD: 3 x 32 B per row whatever the key width); otherwise, or where the keys crowd into few partitions:
 * stable LSD passes over the 8-bit digits in which keys differ (write-combining scatter; from 2^25 rows on
 * histogram-free slab passes chained one into the next + one compaction).  Equal keys keep their
 * input order in every form (the reference is stable in pass 1 only, so on duplicate keys its payload order may
 * differ; the key column and the multiset of rows are identical).  in/out: n x {key,val}, device.
 * out == in sorts in place -- the replacement of radix_int_inplace<uint64_t,uint64_t>(begin, n,
 * num_threads) (radix_sort.h:333-398; radix_bench_par.cc:96), which is unstable: same key column, same
 * multiset of rows.  A partial overlap of in and out is not allowed.                               */
int hmj_sort_u64_device(hmj_ctx* ctx, const void* in_aos_dev, uint64_t n, void* out_aos_dev);

/* Replaces radix_hash::radix_inplace_par on a caller's pre-hashed tuple buffer (radix_hash.h:589-654) -- what the
 * HashMergeJoin2 ctor does to BOTH relations as a side effect callers may rely on (hashjoin.h:234-235): n rows
 * of row_bytes bytes in HOST memory are sorted IN PLACE, ascending on the uint64 at byte offset key_offset.
 * Stable (equal keys keep their input order; the reference's swap chains are not).  The rows go to the GPU,
 * {key, index} pairs are sorted there (eight 8-bit LSD passes), the rows are gathered and copied back.
 * row_bytes: a multiple of 8 in 16..64 (std::tuple<size_t, uint64_t, uint64_t> is 24; libstdc++ stores the
 * elements in reverse order, so its hash sits at offset 16).                                               */
int hmj_sort_rows_by_u64_host(hmj_ctx* ctx, void* rows_host, uint64_t n, uint32_t row_bytes, uint32_t key_offset);
/* The sorted order only, for rows the GPU cannot move (non-trivial types such as std::string keys): key i is the
 * uint64 at keys_host + i * stride_bytes; perm_out[j] = input index of the row that belongs at position j.   */
int hmj_argsort_u64_host(hmj_ctx* ctx, const void* keys_host, uint64_t n, uint32_t stride_bytes, uint32_t* perm_out);

/* ---- synthetic relations on device (same integer arithmetic as the oracle) ------- */
/* key = mix64(i + seed), val = i, i in [start, start+n)                                          */
int hmj_gen_build_u64_device(hmj_ctx* ctx, void* out_aos_dev, uint64_t n, uint64_t start,
                             uint64_t seed);
/* j in [start,start+n): idx = (0x9E3779B1*j + 12345) mod n_build (+ n_build when miss_mod > 0 and
 * j % miss_mod == 0); key = mix64(idx + seed); val = j ^ 0x9E3779B97F4A7C15                      */
int hmj_gen_probe_u64_device(hmj_ctx* ctx, void* out_aos_dev, uint64_t n, uint64_t start,
                             uint64_t n_build, uint64_t seed, uint64_t miss_mod);
/* rank = lower_bound(thr_dev[0..domain), mix64(i ^ zseed)); key = mix64(rank + seed); val = i     */
int hmj_gen_from_cdf_u64_device(hmj_ctx* ctx, void* out_aos_dev, uint64_t n, uint64_t start,
                                const uint64_t* thr_dev, uint64_t domain, uint64_t seed,
                                uint64_t zseed);
/* key = mix64((mix64(j ^ zseed) % domain) + seed); val = j ^ 0x9E3779B97F4A7C15                   */
int hmj_gen_uniform_domain_u64_device(hmj_ctx* ctx, void* out_aos_dev, uint64_t n, uint64_t start,
                                      uint64_t domain, uint64_t seed, uint64_t zseed);

#ifdef __cplusplus
}
#endif
#endif /* HMJ_H */
