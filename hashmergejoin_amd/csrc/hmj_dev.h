// Shared device-side definitions for the gfx950 kernels (wave64, CDNA4).  Internal header.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hmj {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned short u16;

// One relation row: std::pair<uint64_t,uint64_t>, moved as one dwordx4.
struct __attribute__((aligned(16))) Tup {
  u64 key;
  u64 val;
};

constexpr int kWave = 64;
#ifndef HMJ_SCAN_DPP
#define HMJ_SCAN_DPP 1
#endif

// Streaming access to rows that are read or written exactly once per kernel: nontemporal policy, so
// the line does not linger in L2 / Infinity Cache (a plain copy of 4 GiB runs 5.4 -> 6.0 TB/s with it).
// HMJ_STREAM_POLICY: 0 = default policy, 1 = nontemporal stores, 2 = nontemporal loads and stores.
#ifndef HMJ_STREAM_POLICY
#define HMJ_STREAM_POLICY 2
#endif
typedef u64 u64x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_stream(Tup* p, const Tup& v) {
#if HMJ_STREAM_POLICY >= 1
  u64x2_t x = {v.key, v.val};
  __builtin_nontemporal_store(x, reinterpret_cast<u64x2_t*>(p));
#else
  *p = v;
#endif
}
__device__ __forceinline__ Tup load_stream(const Tup* p) {
#if HMJ_STREAM_POLICY >= 2
  const u64x2_t x = __builtin_nontemporal_load(reinterpret_cast<const u64x2_t*>(p));
  Tup t;
  t.key = x.x;
  t.val = x.y;
  return t;
#else
  return *p;
#endif
}

// ---- the global open-addressing table of gtable.hip (also read by radix.hip's pass A with a rank lookup in front) ----------
constexpr u64 GT_EMPTY = ~0ull;   // key of an empty slot
constexpr int GT_MAXWALK = 64;    // a build row that would walk further gives the table up
__device__ __forceinline__ u32 gt_hash(u64 key, int shift) {
  u64 h = key * 0x9E3779B97F4A7C15ull;
  h ^= h >> 32;
  h *= 0xD6E8FEB86659FD93ull;
  return (u32)(h >> shift);
}

// The rank-run form cuts a key's run by the position of the payload in the payloads' range: bucket = floor(off * 2^tb /
// (range + 1)) for off = sval - svmin, computed as ((off >> pre) * mult) >> 32 with pre = max(0, bits(range) - 32) and
// mult = floor(2^(tb + 32) / ((range >> pre) + 1)) (host: api.hip).  Monotone in sval, < 2^tb, and even over ANY range -- the
// top tb bits of the range would leave up to half of the buckets empty when the range is not a power of two.
#ifndef HMJ_RANK_PASS_CHUNK_ROWS
#define HMJ_RANK_PASS_CHUNK_ROWS 64  // (64 / 128 / 256 measure the same, profiles/r05s_*; 64 reaches the longest runs)
#endif
constexpr int RANK_PASS_CHUNK_ROWS = HMJ_RANK_PASS_CHUNK_ROWS;  // the pass with the rank lookup reads chunks of this many rows from all over the probe side (radix.hip)
__device__ __forceinline__ u64 rank_run_bucket(u64 off, int pre, u64 mult) { return ((off >> pre) * mult) >> 32; }

// ONE walker for every kernel that looks keys up in that table (the lockstep walk was written out four times,
// and the copies had begun to differ).  Each lane walks ROWS keys at once -- ROWS independent loads in flight -- from
// slot[r] on, while live[r]; an empty slot ends a row's walk; on_slot(r, entry) sees every OCCUPIED slot a live row
// visits and returns true when that row is done (a hit in a table of unique keys; a multi-map walks on to the empty slot).
// `tab` is anything indexable by a slot number that yields a Tup: the global table, or its copy in LDS.
template <int ROWS, typename Table, typename OnSlot>
__device__ __forceinline__ void gt_walk(Table tab, u32 mask, u32 (&slot)[ROWS], bool (&live)[ROWS], OnSlot on_slot) {
  bool any_live = false;
#pragma unroll
  for (int r = 0; r < ROWS; r++) any_live |= live[r];
  while (any_live) {
    Tup e[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; r++)
      if (live[r]) e[r] = tab[slot[r]];
    any_live = false;
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
      if (live[r]) {
        if (e[r].key == GT_EMPTY || on_slot(r, e[r])) {
          live[r] = false;
        } else {
          slot[r] = (slot[r] + 1) & mask;
          any_live = true;
        }
      }
    }
  }
}

// ---- radix pass geometry (radix.hip) -------------------------------------------------------
constexpr int RP_THREADS = 512;                  // 8 waves
constexpr int RP_WAVES = RP_THREADS / kWave;
constexpr int RP_ITEMS = 8;                      // rows per thread per tile
constexpr int RP_TILE = RP_THREADS * RP_ITEMS;   // 4096 rows = 64 KiB staged in LDS
constexpr int RP_MAX_BITS = 9;                   // fan-out <= 512 per pass
constexpr int RP_MAXD = 1 << RP_MAX_BITS;
constexpr int RP_MAX_BLOCKS = 2048;              // "workers" (reference: threads) per pass

// ---- build+probe geometry (probe.hip) ------------------------------------------------------
constexpr int PB_THREADS = 1024;       // generic kernel: 106 KiB LDS -> one 16-wave workgroup per CU
constexpr int PB_CAP = 5120;           // build rows resident in LDS per chunk
constexpr int PB_LOG_NB = 12;          // 4096 chain heads
constexpr int PB_TARGET_AVG = 4096;    // planner: average build rows per partition
constexpr int PB_PLAN_SLACK = 256;     // planner: tolerated excess of that average before another bit is spent
                                       // (a shard of 2^28 + a few rows after an exchange must not fall to 17 bits)
// Measured at |R|=|S|=2^28 (tools/exp_bits.py): 16 bits (avg 4096 rows/partition, two 8-bit
// write-combining passes) 12.45 ms per join vs 17 bits (avg 2048, 9+8) 13.02 ms, 18 bits 13.9 ms,
// 15 bits 14.6 ms: fewer, larger partitions win as long as one still fits the LDS table.

// ---- slab layout (radix.hip slab kernels -> probe.hip pipelined kernels) -------------------------------------
// A final partition of the histogram-free path is SLAB_KB pieces (one per pass-B worker of its bucket).  The pipelined
// probe kernels unroll exactly this many pieces; every count array is P * SLAB_KB entries, every slab buffer
// P * SLAB_KB * cap rows.  ONE definition: the host sizes its buffers from it (api.hip, SlabGeom::KB) and the
// launchers check the operands against it before a kernel runs.  (The generic probe kernel takes any KB as its
// number of probe slices: probe-side slabs of probe-heavy joins.)
constexpr int SLAB_KB = 4;
constexpr int SLAB_MAX_BITS = 9;  // digit width of one slab pass (9: the 1024-thread, one-workgroup-per-CU shape)

// accumulator slots (global u64[8])
enum { ACC_N = 0, ACC_SUM_R, ACC_SUM_S, ACC_XOR, ACC_MIX, ACC_SUM_P, ACC_ERR, ACC_PAD };
constexpr u64 ERR_SLAB = 2;    // slab path: a slab overflowed or a partition does not fit -> exact path
constexpr u64 ERR_ORDER_DEFER = 16;  // ordered epilogue left a very large segment unsorted: the host sorts the result
constexpr u64 ERR_ORDER_FAIL = 32;   // ordered epilogue: a segment too large to sort in place AND to defer (result > 2^32-1 rows)
constexpr u64 ERR_SORTED = 64;   // one-pass ordered write met duplicate keys / clustered keys / an oversized partition
constexpr u64 ERR_FASTPATH = 8;  // unique-key write mode met duplicate build keys / an oversized partition
constexpr u64 ERR_GTABLE = 128;  // global-table path: a build row walked too far (duplicates / clustering hash) or has the empty marker as its key
constexpr u64 ERR_PREFIX = 4;  // a key does not carry the sampled common prefix (ordered mode re-plans)

__device__ __forceinline__ u64 mix64(u64 x) {  // same constants as oracle/hmj_oracle.c
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

__device__ __forceinline__ u64 tmix(u64 key, u64 rval, u64 sval) {
  u64 t = mix64(key);
  t = mix64(t ^ rval);
  t = mix64(t + sval);
  return t;
}

__device__ __forceinline__ int lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// number of set bits of `m` strictly below this lane
__device__ __forceinline__ u32 popc_below(u64 m) {
  return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}

// Inclusive wave64 scan on the DPP path (VALU speed; __shfl_up goes through ds_bpermute: six LDS round trips).
// row_shr:1,2,4,8 scan the four 16-lane rows, row_bcast:15 / :31 carry the row totals on (gfx9 family).
__device__ __forceinline__ u32 wave_incl_scan_u32(u32 v, int /*lane*/) {
#if HMJ_SCAN_DPP
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);  // row_shr:1
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);  // row_shr:2
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);  // row_shr:4
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);  // row_shr:8
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
  return v;
#else
  const int lane = lane_id();
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    u32 t = __shfl_up(v, o, kWave);
    if (lane >= o) v += t;
  }
  return v;
#endif
}

__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}
__device__ __forceinline__ u64 wave_xor_u64(u64 v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v ^= __shfl_xor(v, o, kWave);
  return v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter (s_waitcnt vmcnt(0)), which would stall every barrier on the global loads a kernel
// keeps in flight as prefetch and on its fire-and-forget stores.  Use where all cross-thread
// communication goes through LDS.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Fold per-thread 64-bit accumulators into the global accumulator block with ONE global atomic
// per value per WORKGROUP.  (One per wave costs ~12 ns each when tens of thousands hit the same
// few addresses at kernel end: measured as a fixed 0.6 ms tail on a 1024-workgroup launch.)
// red: >= 8 u64 of LDS, zeroed by the caller before a barrier.  Contains one barrier.
__device__ __forceinline__ void block_accumulate(u64* red, u64* accum, const u64 (&v)[6],
                                                 u32 xor_mask) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const bool is_xor = (xor_mask >> k) & 1u;
    u64 w = is_xor ? wave_xor_u64(v[k]) : wave_sum_u64(v[k]);
    if (lane == 0 && w) {
      if (is_xor)
        atomicXor(&red[k], w);
      else
        atomicAdd(&red[k], w);
    }
  }
  lds_barrier();
  if (threadIdx.x < 6 && red[threadIdx.x]) {
    if ((xor_mask >> threadIdx.x) & 1u)
      atomicXor(&accum[threadIdx.x], red[threadIdx.x]);
    else
      atomicAdd(&accum[threadIdx.x], red[threadIdx.x]);
  }
}

// Block-wide exclusive scan of one u32 per thread.  scratch: >= THREADS/64 + 1 words of LDS.
// Returns the exclusive prefix; *total receives the block sum.  Contains two barriers.
// TRAILING_BARRIER = false: the caller guarantees a workgroup barrier before `scratch` is written again.
// tid_: the caller's thread index (default: threadIdx.x) -- a kernel that keeps its thread index opaque inside a loop
// passes it, so that &scratch[wave] is not hoisted out of that loop as one more live register.
template <int THREADS, bool TRAILING_BARRIER = true>
__device__ __forceinline__ u32 block_excl_scan_u32(u32 v, u32* scratch, u32* total, int tid_ = -1) {
  const int t_ = tid_ < 0 ? (int)threadIdx.x : tid_;
  const int lane = t_ & 63, w = t_ >> 6;
  constexpr int NW = THREADS / kWave;
  u32 incl = wave_incl_scan_u32(v, lane);
  if (lane == 63) scratch[w] = incl;
  lds_barrier();
  u32 pre = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < NW; k++) {
    u32 s = scratch[k];
    if (k < w) pre += s;
    tot += s;
  }
  if (TRAILING_BARRIER) lds_barrier();
  *total = tot;
  return pre + incl - v;
}

}  // namespace hmj
