// Host-callable launchers of the gfx950 kernels.  Internal header (api.hip <-> *.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

namespace hmj {
typedef unsigned long long u64;
typedef unsigned int u32;

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per-DEVICE function state, and one process may hold a
// ctx per GPU (hmj.h): remember per device (bit = device id) which kernels already carry the attribute.
struct SmemAttrOnce {
  std::atomic<unsigned long long> devices{0};
};
inline hipError_t ensure_max_smem(SmemAttrOnce& once, const void* kernel, size_t bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (dev < 64 && (once.devices.load(std::memory_order_acquire) & bit)) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return e;
  if (dev < 64) once.devices.fetch_or(bit, std::memory_order_release);
  return hipSuccess;
}

// radix.hip
constexpr int kMaxRanks = 16;  // ranks of one exchange (owner split fan-out; one node has 8 GPUs)
// what the owner split of the multi-GPU exchange partitions on (radix.hip owner_digit)
struct OwnerFn {
  u32 mode;  // 0 = radix digit of the key, 1 = floor(mix64(key) * G / 2^64), 2 = splitters
  u32 G;
  u64 spl[kMaxRanks - 1];
};
// variant: 0 = plain staged scatter (4096-row tile), 1 = write-combining scatter (WC_X / WC_Y)
int radix_tile_rows(int bits, int variant);
void radix_pass_geometry(u32 n, int tile, u32* nblk, u32* rows_per_block);
size_t radix_scatter_smem_bytes();
hipError_t launch_radix_hist(const void* in, u32 n, int tile, int shift, int bits, u32* hist,
                             u32 nblk, u32 rows_per_block, hipStream_t st, const OwnerFn* own = nullptr);
hipError_t launch_owner_scatter(const void* in, void* out, u32 n, int bits, const OwnerFn& own,
                                const u32* hist_scanned, const u32* totals, u32 nblk, u32 rows_per_block,
                                u64* offsets_out, hipStream_t st);
hipError_t launch_radix_rowscan(u32* hist, u32 nblk, int bits, u32* totals, hipStream_t st);
hipError_t launch_radix_scatter(const void* in, void* out, u32 n, int variant, int shift, int bits,
                                const u32* hist_scanned, const u32* totals, u32 nblk,
                                u32 rows_per_block, u64* offsets_out, hipStream_t st);
// slab (histogram-free) partitioning, radix.hip
struct SlabGeom {
  u32 WA, rpw, CA, CB, KB;
  u64 rows_a, rows_b;
};
// kb: pieces per final partition (0 = 4, what the pipelined probe kernel reads)
// fan: expected rows per distinct key (>= 1): widens the slabs by sqrt(fan) standard deviations
// density: the populated partitions hold this many times the mean (keys that fill only part of the key range)
bool slab_geometry(u32 n, int bits_a, int bits_b, SlabGeom* g, u32 kb = 0, double fan = 1.0, double density = 1.0);
// one slab pass only (its worker-private slabs are what the probe kernel reads): workers, rows per worker, slab capacity.
// keys_per_digit: distinct keys a digit holds on average (the digit's share of a worker's rows varies with it)
bool slab_geometry_one_pass(u32 n, int bits, double keys_per_digit, u32 max_workers, SlabGeom* g, double density = 1.0);
// rows a slab holds for `mean` expected rows whose count varies like fan * (a Poisson count): mean + 8 sigma + slack
u32 slab_capacity(double mean, double fan);
// slab_*_rows / cnt_*_n: what the caller ALLOCATED (rows of 16 bytes, u32 entries).  The launchers compare them with
// what the kernel and its grid will touch for this geometry and refuse (hipErrorInvalidValue) instead of launching
// a kernel that would write past a buffer.
hipError_t launch_slab_a(const void* in, u32 n, int shift, int bits, const SlabGeom& g, void* slab_a, u64 slab_a_rows,
                         u32* cnt_a, u64 cnt_a_n, u64* accum, hipStream_t st, u32 w_begin = 0, u32 w_end = 0xFFFFFFFFu);
hipError_t launch_slab_b(const void* slab_a, const u32* cnt_a, int bits_a, int shift, int bits,
                         const SlabGeom& g, void* slab_b, u64 slab_b_rows, u32* cnt_b, u64 cnt_b_n, u64* accum, hipStream_t st);
hipError_t launch_key_sample(const void* R, u32 nb, const void* S, u32 np, u64* out, hipStream_t st);
hipError_t launch_key_exact(const void* R, u32 nb, const void* S, u32 np, u64 ref, u64* out, int num_cus, hipStream_t st,
                            bool ref_is_first_key = false);
// *flag (zeroed by the caller) becomes 1 unless the rows' partition numbers (key >> low, `bits` bits) never decrease
hipError_t launch_check_partitioned(const void* a, u32 n, int low, int bits, u32* flag, int num_cus, hipStream_t st);
hipError_t launch_part_offsets(const void* a, u32 n, int low, int bits, u32* off, hipStream_t st);

// probe.hip
struct ProbeArgs {
  const void* R;        // partitioned build rows
  const u32* r_off;     // P+1
  const void* S;        // partitioned probe rows
  const u32* s_off;     // P+1
  u32 P;                   // partitions
  u32 Q;                   // probe slices per partition; work item w = p*Q + q
  u64* part_count;         // P*Q    (count mode with per-item counts)
  const u64* part_out_off; // P*Q+1  (write mode)
  u64* out_key;
  u64* out_rval;
  u64* out_sval;
  u64 out_cap;             // mode 3 (count + write behind a cursor): rows the three columns hold
  u64* accum;              // 8 x u64, see hmj_dev.h ACC_*
  const u32* item_list;    // optional: process only these items (set aside by the fast kernel)
  const u32* n_item_list;  //           their count (device)
  // slab layout (probe_count_slab_kernel): partition p = 4 pieces, piece j = rows
  // [(p*4 + j) * cap, + cnt[p*4 + j]) of R / S
  const u32* r_cnt;
  const u32* s_cnt;
  u32 r_cap, s_cap;
  u32 s_wa, s_ppi;         // generic kernel, probe side after ONE slab pass: partition p = pieces [p * s_wa, (p + 1) * s_wa), item
                           // (p, q) walks s_ppi of them (s_ppi == 0: off); count modes without first-wins
  u64 r_cnt_n, s_cnt_n;    // slab layout: u32 entries allocated behind r_cnt / s_cnt, rows allocated behind R / S --
  u64 r_rows, s_rows;      // checked by the launchers against P * pieces (* cap) before a kernel runs
  const u64* item_base;    // unique-key write mode, slab layout: first output slot of partition p
  const u32* r_end;        // optional: end of partition p's build rows (NULL: r_off[p + 1]); with s_end this lets
  const u32* s_end;        // several "virtual" partitions share one build range (oversized probe partitions are split)
  u32 extra;               // unique-key write mode: bit 0 also accumulate checksums / sum_probe_all, bit 1 first-wins (count ext), bit 2 the ordered foreign-key write buckets its output slots by payload position
  u32 pfx_shift;           // ordered mode: verify (key >> pfx_shift) == pfx_val for every row (0 = off)
  u64 pfx_val;
  u32* matched;            // HMJ_FIRST_WINS + chunked build: one bit per probe row already paired
  u32 debug;               // developer builds (-DHMJ_DEV) only: ablation bits, 1 = loads only, 2 = no probe walk
};
// mode: 0 = count/sums only, 1 = count + per-partition counts, 2 = write, 3 = count + write behind the cursor accum[ACC_N] (piece walk only)
hipError_t launch_probe(const ProbeArgs& a, int mode, bool first_wins, bool extra, int grid,
                        hipStream_t st);
int probe_default_grid(int num_cus);
hipError_t launch_probe_count_ext(const ProbeArgs& a, u32* irregular, u32* n_irregular, bool slab,
                                  int num_cus, hipStream_t st);
hipError_t launch_probe_count_slab(const ProbeArgs& a, int num_cus, hipStream_t st);
hipError_t launch_probe_write_sorted(const ProbeArgs& a, bool slab, bool fk, int shape, u64* lookback, bool chained, int key_low,
                                     int num_cus, hipStream_t st);
hipError_t launch_probe_write_uniq(const ProbeArgs& a, bool slab, int num_cus, hipStream_t st);
hipError_t launch_slab_np(const u32* cnt, u32 P, u64* out, hipStream_t st);
// count-mode fast path (Q == 1, no flags); partitions it cannot take go to irregular[]
hipError_t launch_probe_count_fast(const ProbeArgs& a, u32* irregular, u32* n_irregular, bool big,
                                   bool per_partition_counts, int num_cus, hipStream_t st);
hipError_t launch_scan_u64(const u64* in, u64* out_excl, u32 n, hipStream_t st);  // out: n+1
// ordered result of a join with duplicate build keys, written in order partition by partition (probe_expand_ordered_kernel)
hipError_t launch_probe_expand_ordered(const ProbeArgs& a, int key_low, int num_cus, hipStream_t st);
hipError_t launch_order(const u64* part_out_off, const u32* vstart, const u32* in_base32, const u64* in_base64, u32 P, u32 Q,
                        int low, const u64* akey, const u64* arval, const u64* asval, u64* bkey, u64* brval,
                        u64* bsval, u64* accum, u32 defer_rows, bool many_per_key, int grid, hipStream_t st);
// partition p's rows [in_base[p], + out_off[p + 1] - out_off[p]) of the three columns a -> b at out_off[p] (closes the gaps
// unmatched probe rows leave in the unique-key write mode's slot layout; no sorting)
hipError_t launch_compact(const u64* part_out_off, const u32* in_base32, const u64* in_base64, u32 P, const u64* akey,
                          const u64* arval, const u64* asval, u64* bkey, u64* brval, u64* bsval, int grid, hipStream_t st);
hipError_t launch_rekey(void* pairs, u64 n, const u64* col, hipStream_t st);  // pairs[j].key = col[pairs[j].val]

// gtable.hip: small build sides -- one global open-addressing table (2^log_cap slots of 16 bytes, pre-filled with 0xFF),
// the probe side streamed once.  Count modes only.  R: the build relation (first-wins fetches payloads from it).
hipError_t launch_gtable_build(const void* R, u32 nb, void* tab, int log_cap, u64* accum, bool first, int num_cus,
                               hipStream_t st);
hipError_t launch_gtable_probe(const void* S, u32 np, const void* tab, int log_cap, const void* R, u64* accum, bool first,
                               bool extra, int num_cus, int wg_per_cu, hipStream_t st);
// materialising form: at most one result row per probe row (unique build keys or first-wins); out_*: room for np rows;
// accum[ACC_N] is the output cursor (zeroed by the caller, n_matches afterwards)
hipError_t launch_gtable_write(const void* S, u32 np, const void* tab, int log_cap, const void* R, u64* accum, u64* out_key,
                               u64* out_rval, u64* out_sval, bool first, bool extra, int num_cus, int wg_per_cu, hipStream_t st);

// ordered results of a small build side under a long probe side: composites rank << range_bits | (sval - svmin), sorted, expanded
// tiny build sides (<= ltable_max_rows()): the table in LDS, one copy per workgroup; count modes (FIRST: R's payload of the first row)
hipError_t launch_ltable_probe(const void* R, u32 nb, const void* S, u32 np, u64* accum, bool first, bool extra, int num_cus, hipStream_t st);
int ltable_max_rows();
hipError_t launch_sval_range(const void* S, u32 np, u64* out2 /* {min, max}; caller: {~0, 0} */, int num_cus, hipStream_t st,
                             u32 every = 1 /* > 1: of the rows 0, every, 2 every, ... only */);
// (wide: rank and payload as two words -- {payload, rank} emitted, sorted by payload, swapped, sorted by rank, expanded)
hipError_t launch_gtable_emit(const void* S, u32 np, const void* tab, int log_cap, u64 svmin, int range_bits, u64* accum,
                              void* pairs, bool extra, bool wide, int num_cus, int wg_per_cu, hipStream_t st);
hipError_t launch_gtable_swap(const void* in, void* out, u64 n, int num_cus, hipStream_t st);
// the rank-run form: {rank, sval} rows, partitioned by rank with two slab passes, every rank's run sorted in LDS and written
hipError_t launch_gtable_emit_ranks(const void* S, u32 np, const void* tab, int log_cap, u64* accum, void* pairs, bool extra, int num_cus,
                                    int wg_per_cu, hipStream_t st);
// rows one sorter of the LDS sorts holds: level 0 / 1 / 2 = workgroups of 256 / 512 / 1024 threads = 2048 / 4096 / 8192 rows;
// level -1 (rank_sort_write only) = one wave = 512 rows, four partitions in flight per 256-thread workgroup
int rank_sort_max_run(int level = 0);
// hmj_sort_u64_device's MSD form: the rows partitioned on their top varying key bits by two slab passes (partition p = four
// pieces of `cap` rows, counts cnt[p * 4 ..]), every partition sorted on the remaining bits in LDS (stable) and written at out_off[p]
hipError_t launch_sort_runs_write(const void* slabs, const u32* cnt, u32 cap, u32 P, const u64* out_off, void* out, u64* accum, int level,
                                  int num_cus, hipStream_t st);
hipError_t launch_rank_sort_write(const void* slabs, const u32* cnt, u32 cap, u32 P, const u64* out_off, const void* sortedR, u32 nb, int tb,
                                  u64* out_key, u64* out_rval, u64* out_sval, u64* accum, bool extra, int level, int num_cus, hipStream_t st);
hipError_t launch_slab_a_ranks(const void* in, u32 n, int shift, int bits, const SlabGeom& g, void* slab_a, u64 slab_a_rows, u32* cnt_a,
                               u64 cnt_a_n, u64* accum, const void* tab, int log_cap, bool extra, int tb, u64 svmin, u64 svrange, int pre,
                               u64 mult, hipStream_t st);
hipError_t launch_slab_offsets(const u32* cnt /* P x SLAB_KB piece counts */, u32 P, u64* off /* P + 1 */, u64* scratch /* ceil(P / 1024) */,
                               hipStream_t st);
hipError_t launch_gtable_expand(const void* pairs, u64 n, const void* sortedR, u64 svmin, int range_bits, u64* out_key,
                                u64* out_rval, u64* out_sval, u64* accum, bool extra, bool wide, int num_cus, hipStream_t st);
// Sorted composites that lie in the worker-private slabs of a chain of slab passes (piece i = rows [i * cap, + cnt[i]); the
// sorted order is the order of the pieces): launch_piece_offsets gives every piece its first result row (exclusive scan of
// the counts), launch_gtable_expand_pieces turns the pieces into result rows.
hipError_t launch_piece_offsets(const u32* cnt, u32 n_pieces, u64* off, hipStream_t st);
hipError_t launch_pieces_compact(const void* slabs, const u32* cnt, const u64* off, u32 n_pieces, u32 cap, void* out, int num_cus,
                                 hipStream_t st);  // pieces -> dense 16-byte rows, in piece order
hipError_t launch_gtable_expand_pieces(const void* slabs, const u32* cnt, const u64* off, u32 n_pieces, u32 cap, const void* sortedR,
                                       u64 svmin, int range_bits, u64* out_key, u64* out_rval, u64* out_sval, u64* accum, bool extra,
                                       int num_cus, hipStream_t st);

// gen.hip
hipError_t launch_gen_build(void* out, u64 n, u64 start, u64 seed, hipStream_t st);
hipError_t launch_gen_probe(void* out, u64 n, u64 start, u64 n_build, u64 seed, u64 miss_mod,
                            hipStream_t st);
hipError_t launch_gen_from_cdf(void* out, u64 n, u64 start, const u64* thr, u64 domain, u64 seed,
                               u64 zseed, hipStream_t st);
// split oversized probe partitions into virtual partitions (api.hip, skewed probe sides)
hipError_t launch_split_parts(const u32* r_off, const u32* s_off, u32 P, u32 thr_rows, u32 slice_rows, u32 build_thr,
                              u32 build_slice, u32 cap_v, u32* vstart, u32* vr_beg, u32* vr_end, u32* vs_beg,
                              u32* vs_end, u32* nv_out, hipStream_t st);
hipError_t launch_key_idx(const u64* key, u64 n, void* out, hipStream_t st);
hipError_t launch_fill_probe(void* p, size_t bytes, hipStream_t st);
hipError_t launch_rows_key_idx(const void* rows, u64 n, u32 words, u32 key_word, void* out, hipStream_t st);
hipError_t launch_rows_gather(const void* in, const void* sorted, u64 n, u32 words, void* out, hipStream_t st);
hipError_t launch_pairs_val_u32(const void* sorted, u64 n, u32* out, hipStream_t st);
hipError_t launch_gather3(const void* sorted, u64 n, const u64* rval, const u64* sval, u64* okey, u64* orval,
                          u64* osval, hipStream_t st);
hipError_t launch_gen_uniform_domain(void* out, u64 n, u64 start, u64 domain, u64 seed, u64 zseed,
                                     hipStream_t st);
}  // namespace hmj
