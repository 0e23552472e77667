// Small build sides: ONE global open-addressing table, the probe side streamed once and never partitioned.
//
// The reference benchmarks this formulation next to the partitioned one (BM_hash_join_raw, hashjoin_bench.cc:29-63:
// one std::unordered_map over the whole build side, one loop over the probe side; semantics of the lookups as in
// partitioned_hash.h:166-170 / hashjoin_bench.cc:92-96).  On the GPU it pays when the build side is small enough for
// its table to stay in the L2 / Infinity Cache while the probe side streams by: radix-partitioning a 2^26-row probe
// side costs 32-48 B per row before the first lookup, reading it once costs 16.
//
//   table   cap = 2^k slots of 16 bytes {key, val}, k chosen by the host for a load factor <= 0.5; an empty slot has
//           key == GT_EMPTY (all ones).  Linear probing from hash(key); a MULTI-map: every build row takes a slot
//           of its own, so duplicate build keys need no per-key aggregate and a probe simply walks on to the first
//           empty slot -- 2.5 slots on average at load factor 0.5, nearly always inside one 128-byte line.
//   build   one atomicCAS per visited slot on the key word, the payload stored by the winner.  A row that has to
//           walk further than GT_MAXWALK slots (thousands of copies of one key; a hash that clusters on this key set)
//           raises ERR_GTABLE: the host discards the attempt and takes the partitioned path.
//   probe   grid-stride over tiles of the probe side, four rows per thread walking in lockstep (four independent
//           loads in flight per lane); probe rows are loaded nontemporally so they do not push the table out of L2.
//   key == GT_EMPTY cannot live in the table: a build row with that key raises ERR_GTABLE as well (partitioned path).
//   HMJ_FIRST_WINS: the slots carry the build row's INDEX instead of its payload; a probe row keeps the smallest
//           index among its hits (= first in input order, unordered_map::insert semantics) and fetches that row's
//           payload from the build relation.
// HBM-bound integer work (the probe stream) + random 16-byte reads served by L2 / MALL; no LDS staging needed.
#include "hmj_dev.h"
#include "hmj_launch.h"

namespace hmj {

constexpr int GT_THREADS = 256;
#ifndef HMJ_GT_ROWS
#define HMJ_GT_ROWS 8
#endif
constexpr int GT_ROWS = HMJ_GT_ROWS;  // probe rows per thread and tile
#ifndef HMJ_GTW_THREADS
#define HMJ_GTW_THREADS 256
#endif
constexpr int GTW_THREADS = HMJ_GTW_THREADS;  // the kernels that place rows behind the result cursor: threads per workgroup

// A workgroup's rows of one tile behind the result cursor: ONE add per workgroup.  Adds on one address cost ~11 ns each
// wherever they come from -- one per wave tile (512 rows) was 1.4 of gtable_write_kernel's 1.75 ms at 2^26 rows; one per
// workgroup tile (2048 rows) leaves 0.63-0.80 ms for the whole join (512- and 1024-thread workgroups: no better).
// Every thread of the workgroup calls this the same number of times (two barriers inside).
template <int THREADS>
__device__ __forceinline__ u64 wg_reserve(u32 wave_rows, u64* cursor, u64* s_base, u32* s_rows, int tid, int lane) {
  constexpr int NW = THREADS / kWave;
  const int wv = tid >> 6;
  if (lane == 0) s_rows[wv] = wave_rows;
  __syncthreads();
  const u32 wr = lane < NW ? s_rows[lane] : 0u;
  const u32 wincl = wave_incl_scan_u32(wr, lane);
  const u32 all = (u32)__builtin_amdgcn_readlane((int)wincl, NW - 1);
  if (tid == 0 && all) *s_base = atomicAdd(reinterpret_cast<unsigned long long*>(cursor), (unsigned long long)all);
  __syncthreads();
  return *s_base + (u32)__shfl((int)(wincl - wr), wv, kWave);
}


template <bool FIRST>
__global__ __launch_bounds__(GT_THREADS) void gtable_build_kernel(const Tup* __restrict__ R, u32 nb, Tup* __restrict__ tab,
                                                                   int log_cap, u64* __restrict__ accum) {
  const u32 mask = (1u << log_cap) - 1;
  const int shift = 64 - log_cap;
  bool bad = false, dup = false;
  for (u64 i = (u64)blockIdx.x * GT_THREADS + threadIdx.x; i < nb; i += (u64)gridDim.x * GT_THREADS) {
    const Tup t = R[i];
    if (t.key == GT_EMPTY) {
      bad = true;
      continue;
    }
    u32 s = gt_hash(t.key, shift);
    int walk = 0;
    for (; walk < GT_MAXWALK; walk++) {
      const u64 old = atomicCAS(reinterpret_cast<unsigned long long*>(&tab[s].key), (unsigned long long)GT_EMPTY,
                                (unsigned long long)t.key);
      if (old == GT_EMPTY) {
        tab[s].val = FIRST ? i : t.val;
        break;
      }
      dup |= old == t.key;
      s = (s + 1) & mask;
    }
    bad |= walk == GT_MAXWALK;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(reinterpret_cast<unsigned long long*>(&accum[ACC_ERR]), (unsigned long long)ERR_GTABLE);
  // a key met twice: probe walks must go on past a hit (multi-map).  Otherwise -- unique build keys, the usual case --
  // the probe kernel stops a walk at its hit (accum[ACC_PAD] is read there as a wave-uniform flag)
  if (__any(dup) && (threadIdx.x & 63) == 0) atomicOr(reinterpret_cast<unsigned long long*>(&accum[ACC_PAD]), 1ull);
}

template <bool FIRST, bool EXTRA>
__global__ __launch_bounds__(GT_THREADS) void gtable_probe_kernel(const Tup* __restrict__ S, u32 np, const Tup* __restrict__ tab,
                                                                   int log_cap, const Tup* __restrict__ R,
                                                                   u64* __restrict__ accum) {
  __shared__ u64 red[8];
  const int tid = threadIdx.x;
  if (tid < 8) red[tid] = 0;
  const u32 mask = (1u << log_cap) - 1;
  const int shift = 64 - log_cap;
  constexpr u32 TILE = GT_THREADS * GT_ROWS;
  u64 acc_n = 0, acc_r = 0, acc_s = 0, acc_x = 0, acc_m = 0, acc_p = 0;
  const bool multi = accum[ACC_PAD] != 0;  // duplicate build keys: a walk ends at an empty slot only
  for (u64 base = (u64)blockIdx.x * TILE; base < np; base += (u64)gridDim.x * TILE) {
    Tup t[GT_ROWS];
    u32 slot[GT_ROWS];
    bool live[GT_ROWS];
    u64 first_idx[GT_ROWS];
#pragma unroll
    for (int r = 0; r < GT_ROWS; r++) {
      const u64 i = base + (u64)r * GT_THREADS + tid;
      const bool valid = i < np;
      t[r] = load_stream(&S[valid ? i : (u64)np - 1]);
      live[r] = valid;
    }
#pragma unroll
    for (int r = 0; r < GT_ROWS; r++) {
      slot[r] = gt_hash(t[r].key, shift);
      first_idx[r] = ~0ull;
      if (EXTRA && live[r]) acc_p += t[r].val;
      if (t[r].key == GT_EMPTY) live[r] = false;  // never a table key (the build kernel refuses such rows)
    }
    gt_walk<GT_ROWS>(tab, mask, slot, live, [&](int r, const Tup& e) {
      const bool eq = e.key == t[r].key;
      if (eq) {
        if (FIRST) {
          first_idx[r] = e.val < first_idx[r] ? e.val : first_idx[r];
        } else {
          acc_n++;
          acc_r += e.val;
          acc_s += t[r].val;
          if (EXTRA) {
            const u64 m = tmix(t[r].key, e.val, t[r].val);
            acc_x ^= m;
            acc_m += m;
          }
        }
      }
      return eq && !multi;  // (duplicate build keys: a walk ends at an empty slot only)
    });
    if (FIRST) {
#pragma unroll
      for (int r = 0; r < GT_ROWS; r++) {
        if (first_idx[r] != ~0ull) {
          const u64 rv = R[first_idx[r]].val;
          acc_n++;
          acc_r += rv;
          acc_s += t[r].val;
          if (EXTRA) {
            const u64 m = tmix(t[r].key, rv, t[r].val);
            acc_x ^= m;
            acc_m += m;
          }
        }
      }
    }
  }
  __syncthreads();
  const u64 v[6] = {acc_n, acc_r, acc_s, acc_x, acc_m, acc_p};  // ACC_N .. ACC_SUM_P order
  block_accumulate(red, accum, v, 1u << ACC_XOR);
}

// ---- tiny build sides (<= LT_MAX_ROWS rows: a dimension table of a few thousand rows): the SAME table in LDS ---------------
// A lookup in the L2-resident table costs a 128-byte line from L2 to L1 per probe row whatever the slot's 16 bytes -- 2^26
// lookups take 0.25 ms of the CUs' L1 fill bandwidth alone (0.42 ms measured).  A table of 8192 slots is 128 KiB: every
// workgroup builds its own copy in LDS (the build side is read from L2: 64 KiB per workgroup) and the probe side streams
// against it at HBM speed.  One workgroup of 1024 threads per CU; count modes.
constexpr int LT_THREADS = 1024, LT_LOG_SLOTS = 13, LT_MAX_ROWS = 4096;  // load factor <= 0.5
template <bool FIRST, bool EXTRA>
__global__ __launch_bounds__(LT_THREADS) void ltable_probe_kernel(const Tup* __restrict__ R, u32 nb, const Tup* __restrict__ S, u32 np,
                                                                  u64* __restrict__ accum) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lt_smem[];
  Tup* tab = reinterpret_cast<Tup*>(lt_smem);                                   // [1 << LT_LOG_SLOTS]
  u64* red = reinterpret_cast<u64*>(lt_smem + (sizeof(Tup) << LT_LOG_SLOTS));   // [8]
  u32* flags = reinterpret_cast<u32*>(red + 8);                                 // [0]: duplicate build keys, [1]: give up
  const int tid = threadIdx.x;
  constexpr u32 NS = 1u << LT_LOG_SLOTS, mask = NS - 1;
  constexpr int shift = 64 - LT_LOG_SLOTS;
  for (u32 i = tid; i < NS; i += LT_THREADS) {
    tab[i].key = GT_EMPTY;
    tab[i].val = 0;
  }
  if (tid < 8) red[tid] = 0;
  if (tid < 2) flags[tid] = 0;
  __syncthreads();
  for (u32 i = tid; i < nb; i += LT_THREADS) {
    const Tup t = R[i];
    if (t.key == GT_EMPTY) {
      flags[1] = 1;
      continue;
    }
    u32 s = gt_hash(t.key, shift);
    int walk = 0;
    for (; walk < GT_MAXWALK; walk++) {
      const u64 old = atomicCAS(reinterpret_cast<unsigned long long*>(&tab[s].key), (unsigned long long)GT_EMPTY, (unsigned long long)t.key);
      if (old == GT_EMPTY) {
        tab[s].val = FIRST ? (u64)i : t.val;
        break;
      }
      if (old == t.key) flags[0] = 1;
      s = (s + 1) & mask;
    }
    if (walk == GT_MAXWALK) flags[1] = 1;
  }
  __syncthreads();
  if (flags[1]) {  // (uniform; every workgroup sees the same build side)
    if (tid == 0 && blockIdx.x == 0) atomicOr(reinterpret_cast<unsigned long long*>(&accum[ACC_ERR]), (unsigned long long)ERR_GTABLE);
    return;
  }
  const bool multi = flags[0] != 0;
  constexpr u32 TILE = LT_THREADS * GT_ROWS;
  u64 acc_n = 0, acc_r = 0, acc_s = 0, acc_x = 0, acc_m = 0, acc_p = 0;
  for (u64 base = (u64)blockIdx.x * TILE; base < np; base += (u64)gridDim.x * TILE) {
    Tup t[GT_ROWS];
    u32 slot[GT_ROWS];
    bool live[GT_ROWS];
    u64 first_idx[GT_ROWS];
#pragma unroll
    for (int r = 0; r < GT_ROWS; r++) {
      const u64 i = base + (u64)r * LT_THREADS + tid;
      const bool valid = i < np;
      t[r] = load_stream(&S[valid ? i : (u64)np - 1]);
      live[r] = valid;
    }
#pragma unroll
    for (int r = 0; r < GT_ROWS; r++) {
      slot[r] = gt_hash(t[r].key, shift);
      first_idx[r] = ~0ull;
      if (EXTRA && live[r]) acc_p += t[r].val;
      if (t[r].key == GT_EMPTY) live[r] = false;
    }
    gt_walk<GT_ROWS>(tab, mask, slot, live, [&](int r, const Tup& e) {
      const bool eq = e.key == t[r].key;
      if (eq) {
        if (FIRST) {
          first_idx[r] = e.val < first_idx[r] ? e.val : first_idx[r];
        } else {
          acc_n++;
          acc_r += e.val;
          acc_s += t[r].val;
          if (EXTRA) {
            const u64 m = tmix(t[r].key, e.val, t[r].val);
            acc_x ^= m;
            acc_m += m;
          }
        }
      }
      return eq && !multi;  // (duplicate build keys: a walk ends at an empty slot only)
    });
    if (FIRST) {
#pragma unroll
      for (int r = 0; r < GT_ROWS; r++) {
        if (first_idx[r] != ~0ull) {
          const u64 rv = R[first_idx[r]].val;
          acc_n++;
          acc_r += rv;
          acc_s += t[r].val;
          if (EXTRA) {
            const u64 m = tmix(t[r].key, rv, t[r].val);
            acc_x ^= m;
            acc_m += m;
          }
        }
      }
    }
  }
  __syncthreads();
  const u64 v[6] = {acc_n, acc_r, acc_s, acc_x, acc_m, acc_p};
  block_accumulate(red, accum, v, 1u << ACC_XOR);
}

hipError_t launch_ltable_probe(const void* R, u32 nb, const void* S, u32 np, u64* accum, bool first, bool extra, int num_cus,
                               hipStream_t st) {
  if (nb == 0 || nb > (u32)LT_MAX_ROWS || !R || !accum) return hipErrorInvalidValue;
  const size_t smem = (sizeof(Tup) << LT_LOG_SLOTS) + 8 * sizeof(u64) + 2 * sizeof(u32);
  const u64 tiles = ((u64)np + LT_THREADS * GT_ROWS - 1) / (LT_THREADS * GT_ROWS);
  u64 grid = (u64)num_cus;
  if (grid > tiles) grid = tiles;
  if (grid < 1) grid = 1;
#define HMJ_LT(F, E)                                                                                                      \
  {                                                                                                                         \
    static SmemAttrOnce once;                                                                                               \
    if (hipError_t e = ensure_max_smem(once, reinterpret_cast<const void*>(ltable_probe_kernel<F, E>), smem); e != hipSuccess) return e; \
    hipLaunchKernelGGL((ltable_probe_kernel<F, E>), dim3((u32)grid), dim3(LT_THREADS), smem, st, static_cast<const Tup*>(R), nb,       \
                       static_cast<const Tup*>(S), np, accum);                                                              \
  }
  if (first) {
    if (extra) HMJ_LT(true, true) else HMJ_LT(true, false)
  } else {
    if (extra) HMJ_LT(false, true) else HMJ_LT(false, false)
  }
#undef HMJ_LT
  return hipGetLastError();
}
int ltable_max_rows() { return LT_MAX_ROWS; }

// Materialising form (HMJ_MATERIALIZE without HMJ_ORDERED; unique build keys, or HMJ_FIRST_WINS): every probe row has at most
// one result row.  No count pass: a wave compacts the hits of each of its row slots with a ballot, reserves its output
// rows with ONE atomic add per tile on the result cursor (accum[ACC_N], which ends as n_matches) and writes them as runs
// of consecutive rows -- an unordered result is a multiset, so any placement is a valid one.  Duplicate build keys
// without first-wins (a probe row would expand to several rows) make the kernel return at once; the host sees the build
// kernel's flag and takes the partitioned path.
template <bool FIRST, bool EXTRA>
__global__ __launch_bounds__(GTW_THREADS) void gtable_write_kernel(const Tup* __restrict__ S, u32 np, const Tup* __restrict__ tab,
                                                                   int log_cap, const Tup* __restrict__ R, u64* __restrict__ accum,
                                                                   u64* __restrict__ out_key, u64* __restrict__ out_rval,
                                                                   u64* __restrict__ out_sval) {
  __shared__ u64 red[8];
  __shared__ u64 obase;
  __shared__ u32 wrows[GTW_THREADS / kWave];
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid < 8) red[tid] = 0;
  const bool dups = accum[ACC_PAD] != 0;
  if (dups && !FIRST) return;  // (uniform for the whole grid)
  const u32 mask = (1u << log_cap) - 1;
  const int shift = 64 - log_cap;
  constexpr u32 TILE = GTW_THREADS * GT_ROWS;
  u64 acc_r = 0, acc_s = 0, acc_x = 0, acc_m = 0, acc_p = 0;
  for (u64 base = (u64)blockIdx.x * TILE; base < np; base += (u64)gridDim.x * TILE) {
    Tup t[GT_ROWS];
    u32 slot[GT_ROWS];
    bool live[GT_ROWS];
    u64 hitv[GT_ROWS];  // the matching build row's payload (FIRST: its index, smallest so far)
    u32 hit = 0;        // bit r: row slot r has a match
#pragma unroll
    for (int r = 0; r < GT_ROWS; r++) {
      const u64 i = base + (u64)r * GTW_THREADS + tid;
      const bool valid = i < np;
      t[r] = load_stream(&S[valid ? i : (u64)np - 1]);
      live[r] = valid;
    }
#pragma unroll
    for (int r = 0; r < GT_ROWS; r++) {
      slot[r] = gt_hash(t[r].key, shift);
      hitv[r] = ~0ull;
      if (EXTRA && live[r]) acc_p += t[r].val;
      if (t[r].key == GT_EMPTY) live[r] = false;
    }
    gt_walk<GT_ROWS>(tab, mask, slot, live, [&](int r, const Tup& e) {
      const bool eq = e.key == t[r].key;
      if (eq) {
        hit |= 1u << r;
        hitv[r] = (FIRST && e.val > hitv[r]) ? hitv[r] : e.val;
      }
      return eq && !dups;
    });
    // this wave's output rows: slot r's hits form a run, the runs follow each other
    u32 pre[GT_ROWS], run = 0;
    u64 m[GT_ROWS];
#pragma unroll
    for (int r = 0; r < GT_ROWS; r++) {
      m[r] = __ballot((hit >> r) & 1u);
      pre[r] = run;
      run += (u32)__popcll(m[r]);
    }
    const u64 ob = wg_reserve<GTW_THREADS>(run, &accum[ACC_N], &obase, wrows, tid, lane);
#pragma unroll
    for (int r = 0; r < GT_ROWS; r++) {
      if ((hit >> r) & 1u) {
        const u64 rv = FIRST ? R[hitv[r]].val : hitv[r];
        const u64 o = ob + pre[r] + popc_below(m[r]);
        out_key[o] = t[r].key;
        out_rval[o] = rv;
        out_sval[o] = t[r].val;
        acc_r += rv;
        acc_s += t[r].val;
        if (EXTRA) {
          const u64 mx = tmix(t[r].key, rv, t[r].val);
          acc_x ^= mx;
          acc_m += mx;
        }
      }
    }
  }
  __syncthreads();
  const u64 v[6] = {0, acc_r, acc_s, acc_x, acc_m, acc_p};  // (ACC_N is the output cursor: already complete)
  block_accumulate(red, accum, v, 1u << ACC_XOR);
}

// ---- ordered results for a small build side under a long probe side (fan-outs in the hundreds and thousands) ----------
// The operator's order is (key, rval, sval).  With unique build keys that is: the probe rows sorted by (rank of their key
// among the sorted build keys, sval).  rank < 2^17 and the payloads of one relation usually span far less than 64 bits
// (row ids, timestamps of one epoch), so the pair fits ONE 64-bit sort key: composite = rank << range_bits | (sval - svmin).
// Pipeline (api.hip, try_small_build_ordered): sort the build side (tiny) -> global table key -> rank -> this kernel emits
// one composite per matching probe row -> LSD radix passes over the composite's rank_bits + range_bits low bits ->
// gtable_expand_kernel turns every composite back into its result row.  The partitioned paths rank every probe row inside
// its key's run, linear in the run: 2^16 x 2^26 rows ordered 14-22 ms, 2^10 x 2^22 6 ms (an 18-bit plan of 2^18 partitions).
// (one atomic pair per WORKGROUP and four loads in flight per thread: with one pair per wave of a 2048-workgroup grid, the
//  16 384 same-address atomics were half of the kernel's 0.38 ms at 2^26 rows -- profiles/r05a_small16_ord_summary.txt)
// every > 1: a SAMPLE, the rows 0, every, 2 every, ... (np counts the sampled rows) -- the rank-run form's pieces only need a
// range that nearly all payloads lie in (rows outside it go to the first / last piece, api.hip), not a pass over the relation.
__global__ __launch_bounds__(256) void sval_range_kernel(const Tup* __restrict__ S0, u32 np, u32 every, u64* __restrict__ out) {
  __shared__ u64 wmn[4], wmx[4];
  u64 mn = ~0ull, mx = 0;
  const u64 stride = (u64)gridDim.x * 256;
  u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (every > 1) {  // (uniform)
    for (; i < np; i += stride) {
      const u64 v = S0[i * every].val;
      mn = v < mn ? v : mn;
      mx = v > mx ? v : mx;
    }
  }
  const Tup* __restrict__ S = S0;
  for (; i + 3 * stride < np; i += 4 * stride) {
    const u64 v0 = load_stream(&S[i]).val, v1 = load_stream(&S[i + stride]).val, v2 = load_stream(&S[i + 2 * stride]).val,
              v3 = load_stream(&S[i + 3 * stride]).val;
    const u64 lo01 = v0 < v1 ? v0 : v1, lo23 = v2 < v3 ? v2 : v3, hi01 = v0 > v1 ? v0 : v1, hi23 = v2 > v3 ? v2 : v3;
    const u64 lo = lo01 < lo23 ? lo01 : lo23, hi = hi01 > hi23 ? hi01 : hi23;
    mn = lo < mn ? lo : mn;
    mx = hi > mx ? hi : mx;
  }
  for (; i < np; i += stride) {
    const u64 v = S[i].val;
    mn = v < mn ? v : mn;
    mx = v > mx ? v : mx;
  }
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) {
    const u64 a2 = __shfl_xor(mn, o, kWave), b2 = __shfl_xor(mx, o, kWave);
    mn = a2 < mn ? a2 : mn;
    mx = b2 > mx ? b2 : mx;
  }
  if ((threadIdx.x & 63) == 0) {
    wmn[threadIdx.x >> 6] = mn;
    wmx[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0 && np) {
    for (int w = 1; w < 4; w++) {
      mn = wmn[w] < mn ? wmn[w] : mn;
      mx = wmx[w] > mx ? wmx[w] : mx;
    }
    atomicMin(reinterpret_cast<unsigned long long*>(out), (unsigned long long)mn);
    atomicMax(reinterpret_cast<unsigned long long*>(out + 1), (unsigned long long)mx);
  }
}

// table: key -> rank (built by gtable_build_kernel<true> over the SORTED build rows: the row index is the rank).
// pairs[cursor++] = {rank << range_bits | (sval - svmin), 0}; accum[ACC_N] is the cursor.
// WIDE (rank and payload do not fit one word together): pairs[..] = {sval - svmin, rank} -- sorted by the payload first, then
// (after gtable_swap_kernel) stably by the rank.
// RANKKEY (round 5, the rank-run form below): pairs[..] = {rank, sval} -- no payload range needed, any payload width.
template <bool EXTRA, bool WIDE, bool RANKKEY = false>
__global__ __launch_bounds__(GTW_THREADS) void gtable_emit_kernel(const Tup* __restrict__ S, u32 np, const Tup* __restrict__ tab,
                                                                  int log_cap, u64 svmin, int range_bits, u64* __restrict__ accum,
                                                                  Tup* __restrict__ pairs) {
  __shared__ u64 red[8];
  __shared__ u64 obase;
  __shared__ u32 wrows[GTW_THREADS / kWave];
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid < 8) red[tid] = 0;
  if (accum[ACC_PAD] != 0) return;  // duplicate build keys: not this path's case (uniform for the grid)
  const u32 mask = (1u << log_cap) - 1;
  const int shift = 64 - log_cap;
  constexpr u32 TILE = GTW_THREADS * GT_ROWS;
  u64 acc_s = 0, acc_p = 0;
  for (u64 base = (u64)blockIdx.x * TILE; base < np; base += (u64)gridDim.x * TILE) {
    Tup t[GT_ROWS];
    u32 slot[GT_ROWS];
    bool live[GT_ROWS];
    u64 rank[GT_ROWS];
    u32 hit = 0;
#pragma unroll
    for (int r = 0; r < GT_ROWS; r++) {
      const u64 i = base + (u64)r * GTW_THREADS + tid;
      const bool valid = i < np;
      t[r] = load_stream(&S[valid ? i : (u64)np - 1]);
      live[r] = valid;
    }
#pragma unroll
    for (int r = 0; r < GT_ROWS; r++) {
      slot[r] = gt_hash(t[r].key, shift);
      rank[r] = 0;
      if (EXTRA && live[r]) acc_p += t[r].val;
      if (t[r].key == GT_EMPTY) live[r] = false;
    }
    gt_walk<GT_ROWS>(tab, mask, slot, live, [&](int r, const Tup& e) {
      const bool eq = e.key == t[r].key;  // (unique build keys: the only hit)
      if (eq) {
        hit |= 1u << r;
        rank[r] = e.val;
      }
      return eq;
    });
    u32 pre[GT_ROWS], run = 0;
    u64 m[GT_ROWS];
#pragma unroll
    for (int r = 0; r < GT_ROWS; r++) {
      m[r] = __ballot((hit >> r) & 1u);
      pre[r] = run;
      run += (u32)__popcll(m[r]);
    }
    const u64 ob = wg_reserve<GTW_THREADS>(run, &accum[ACC_N], &obase, wrows, tid, lane);
#pragma unroll
    for (int r = 0; r < GT_ROWS; r++) {
      if ((hit >> r) & 1u) {
        Tup o;
        if (RANKKEY) {
          o.key = rank[r];
          o.val = t[r].val;
        } else if (WIDE) {
          o.key = t[r].val - svmin;
          o.val = rank[r];
        } else {
          o.key = (range_bits >= 64 ? 0ull : rank[r] << range_bits) | (t[r].val - svmin);
          o.val = 0;
        }
        store_stream(&pairs[ob + pre[r] + popc_below(m[r])], o);
        acc_s += t[r].val;
      }
    }
  }
  __syncthreads();
  const u64 v[6] = {0, 0, acc_s, 0, 0, acc_p};
  block_accumulate(red, accum, v, 1u << ACC_XOR);
}

// {a, b} -> {b, a}: the wide form's second sort is on the rank
__global__ __launch_bounds__(256) void gtable_swap_kernel(const Tup* __restrict__ in, Tup* __restrict__ out, u64 n) {
  for (u64 j = (u64)blockIdx.x * 256 + threadIdx.x; j < n; j += (u64)gridDim.x * 256) {
    const Tup t = load_stream(&in[j]);
    Tup o;
    o.key = t.val;
    o.val = t.key;
    store_stream(&out[j], o);
  }
}

// sorted composites -> result rows.  sortedR: the build rows in key order (rank = index).
// WIDE: pairs[j] = {rank, sval - svmin}.
template <bool EXTRA, bool WIDE>
__global__ __launch_bounds__(256) void gtable_expand_kernel(const Tup* __restrict__ pairs, u64 n, const Tup* __restrict__ sortedR,
                                                            u64 svmin, int range_bits, u64* __restrict__ out_key,
                                                            u64* __restrict__ out_rval, u64* __restrict__ out_sval,
                                                            u64* __restrict__ accum) {
  __shared__ u64 red[8];
  if (threadIdx.x < 8) red[threadIdx.x] = 0;
  const u64 lowmask = range_bits >= 64 ? ~0ull : ((1ull << range_bits) - 1);
  u64 acc_r = 0, acc_x = 0, acc_m = 0;
  for (u64 j = (u64)blockIdx.x * 256 + threadIdx.x; j < n; j += (u64)gridDim.x * 256) {
    const Tup pr = load_stream(&pairs[j]);
    const u64 c = pr.key;
    const Tup b = sortedR[WIDE ? c : (range_bits >= 64 ? 0ull : c >> range_bits)];
    const u64 sv = (WIDE ? pr.val : (c & lowmask)) + svmin;
    out_key[j] = b.key;
    out_rval[j] = b.val;
    out_sval[j] = sv;
    acc_r += b.val;
    if (EXTRA) {
      const u64 mx = tmix(b.key, b.val, sv);
      acc_x ^= mx;
      acc_m += mx;
    }
  }
  __syncthreads();
  const u64 v[6] = {0, acc_r, 0, acc_x, acc_m, 0};
  block_accumulate(red, accum, v, 1u << ACC_XOR);
}

// ---- the same, from the worker-private slabs of a chain of histogram-free slab passes (api.hip, try_small_build_ordered) ------
// off[i] = cnt[0] + ... + cnt[i - 1]: one workgroup, a contiguous share of the counts per thread (a few hundred thousand counts).
__global__ __launch_bounds__(1024) void piece_offsets_kernel(const u32* __restrict__ cnt, u32 n, u64* __restrict__ off) {
  __shared__ u64 wtot[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const u32 per = (n + 1023u) / 1024u, b = (u32)tid * per, e = b + per < n ? b + per : n;
  u64 s = 0;
  for (u32 i = b; i < e; i++) s += cnt[i];
  u64 incl = s;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const u64 up = __shfl_up(incl, o, kWave);
    if (lane >= o) incl += up;
  }
  if (lane == kWave - 1) wtot[wv] = incl;
  __syncthreads();
  u64 before = 0;
  for (int w2 = 0; w2 < wv; w2++) before += wtot[w2];
  u64 run = before + incl - s;
  for (u32 i = b; i < e; i++) {
    off[i] = run;
    run += cnt[i];
  }
}

// A wave walks a contiguous range of pieces -- counts and offsets fetched 64 at a time, one per lane, handed out by readlane --
// and calls row(piece, r, first result row of the piece) for every row r of every piece, lanes over the rows.  ONE walker
// for the kernels that consume a chain's last pass (the walk was written out twice).
template <typename OnRow>
__device__ __forceinline__ void walk_pieces(const u32* __restrict__ cnt, const u64* __restrict__ off, u32 n_pieces, u32 ppw, OnRow on_row) {
  const int lane = threadIdx.x & 63;
  const u32 wave = blockIdx.x * 4u + (threadIdx.x >> 6);
  const u64 p0 = (u64)wave * ppw;
  const u32 p1 = p0 + ppw < n_pieces ? (u32)(p0 + ppw) : n_pieces;
  for (u64 pb = p0; pb < p1; pb += kWave) {
    const u32 c_l = pb + lane < p1 ? cnt[pb + lane] : 0u;
    const u64 o_l = pb + lane < p1 ? off[pb + lane] : 0ull;
    const u32 pe = p1 - pb < (u64)kWave ? (u32)(p1 - pb) : (u32)kWave;
    for (u32 pi = 0; pi < pe; pi++) {
      const u32 c = (u32)__builtin_amdgcn_readlane((int)c_l, (int)pi);
      const u64 o = ((u64)(u32)__builtin_amdgcn_readlane((int)(o_l >> 32), (int)pi) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)o_l, (int)pi);
      for (u32 r = (u32)lane; r < c; r += kWave) on_row(pb + pi, r, o);
    }
  }
}

template <bool EXTRA>
__global__ __launch_bounds__(256) void gtable_expand_pieces_kernel(const Tup* __restrict__ slabs, const u32* __restrict__ cnt,
                                                                   const u64* __restrict__ off, u32 n_pieces, u32 cap, u32 ppw,
                                                                   const Tup* __restrict__ sortedR, u64 svmin, int range_bits,
                                                                   u64* __restrict__ out_key, u64* __restrict__ out_rval,
                                                                   u64* __restrict__ out_sval, u64* __restrict__ accum) {
  __shared__ u64 red[8];
  if (threadIdx.x < 8) red[threadIdx.x] = 0;
  const u64 lowmask = range_bits >= 64 ? ~0ull : ((1ull << range_bits) - 1);
  u64 acc_r = 0, acc_x = 0, acc_m = 0;
  walk_pieces(cnt, off, n_pieces, ppw, [&](u64 piece, u32 r, u64 o) {
    const u64 comp = load_stream(&slabs[piece * cap + r]).key;
    const Tup b = sortedR[range_bits >= 64 ? 0ull : comp >> range_bits];
    const u64 sv = (comp & lowmask) + svmin;
    out_key[o + r] = b.key;
    out_rval[o + r] = b.val;
    out_sval[o + r] = sv;
    acc_r += b.val;
    if (EXTRA) {
      const u64 mx = tmix(b.key, b.val, sv);
      acc_x ^= mx;
      acc_m += mx;
    }
  });
  __syncthreads();
  const u64 v[6] = {0, acc_r, 0, acc_x, acc_m, 0};
  block_accumulate(red, accum, v, 1u << ACC_XOR);
}

// pieces -> dense rows (the sort API's last step: hmj_sort_u64_device over a chain of slab passes)
__global__ __launch_bounds__(256) void pieces_compact_kernel(const Tup* __restrict__ slabs, const u32* __restrict__ cnt,
                                                             const u64* __restrict__ off, u32 n_pieces, u32 cap, u32 ppw,
                                                             Tup* __restrict__ out) {
  walk_pieces(cnt, off, n_pieces, ppw, [&](u64 piece, u32 r, u64 o) { store_stream(&out[o + r], load_stream(&slabs[piece * cap + r])); });
}

// ---- the rank-run form (round 5): ordered result of a small build side under a probe side of fan-out 16 and more -------
// With unique build keys the operator's order (key, rval, sval) is: the probe rows grouped by the RANK of their key, every
// group sorted by sval.  The composite form above sorts (rank, sval - svmin) with 4-10 global LSD passes of 32 B per row
// each (2^16 x 2^26 rows: 4.7 ms = 0.07 of the HBM peak, 8.8 ms when the payloads span 64 bits: profiles/r05a_small16_ord*).
// Here the rows {rank, sval} are PARTITIONED by rank with the join's own two histogram-free slab passes (rank bits = the
// two digits, so partition p = the rows of rank p, in four slab pieces), and one workgroup per rank sorts its run of svals
// in LDS -- a register-tiled bitonic network on 32-bit keys sval - min (or the 64-bit svals where a run spans more) -- and
// writes (key, rval, sval) at the rank's offset.  Two global passes instead of five, no payload range needed, payload
// width irrelevant.  Fan-outs beyond ~1700: the host cuts every run into 2^tb pieces by payload position (partition =
// rank << tb | piece, api.hip / radix.hip RankXform<true>).  Partitions beyond the workgroup's capacity (a hot key) raise ERR_FASTPATH: the
// composite form runs.
constexpr int RS_EPT = 8;  // rows per thread: a workgroup of T = 256 / 512 / 1024 threads sorts up to 2048 / 4096 / 8192 rows
// (the 256-thread shape is the fast one -- five workgroups per CU; the larger ones exist for partitions that two 9-bit slab
//  passes cannot make smaller: sorts beyond 4.5 * 10^8 rows, ordered joins under > 2^29 probe rows)
// LDS of one workgroup (dynamic: 129 KiB at T = 1024).  OIDX: the stable sort's original positions (hmj_sort_u64_device only)
template <int T, bool OIDX>
struct RsSmem {
  u64 keys[T * RS_EPT];                       // the run's keys (64-bit, or 32-bit in its first half); later payloads by the same slots
  u32 bcnt[T * RS_EPT + 2];                   // bucket counts / starts (+ pad: what follows stays 8-byte aligned)
  unsigned short sidx[T * RS_EPT];            // bucket sort: slot of the i-th smallest key
  unsigned short oidx[OIDX ? T * RS_EPT : 4];
  u32 wsc[2 * (T / kWave)];
  u64 wmn[T / kWave], wmx[T / kWave];
  u64 red[8];
};
// T == 64 is the WAVE shape: a 256-thread workgroup whose four waves each sort their own partitions of up to 512 rows, with
// their own RsSmem<64> and no workgroup barrier -- a wave's LDS operations complete in order, so waiting for them is all the
// synchronisation its lanes need.  For partitions of a few hundred rows (fan-outs 16 ... ~330) the 256-thread shape left
// three quarters of its threads idle through a dozen barriers per partition (7.5 ns per partition; profiles/r05k_*).
template <int T>
__device__ __forceinline__ void rs_sync() {
  if constexpr (T == 64) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  } else {
    lds_barrier();
  }
}
// LDS position of element i: the low three index bits XORed with bits 5..7, so that the eight consecutive elements a
// thread takes in the stride-1 steps fall into different banks than its neighbours' (16-way conflicts otherwise)
__device__ __forceinline__ u32 rs_sw(u32 i) { return i ^ ((i >> 5) & 7u); }

template <typename K>
__device__ __forceinline__ void rs_ce(K& a, K& b, bool asc) {
  const K lo = a < b ? a : b, hi = a < b ? b : a;
  a = asc ? lo : hi;
  b = asc ? hi : lo;
}
// three levels of a bitonic merge on the eight elements v[i] = element base + i * q: strides 4q, 2q, q
template <typename K>
__device__ __forceinline__ void rs_three_levels(K (&v)[RS_EPT], bool asc) {
#pragma unroll
  for (int i = 0; i < 4; i++) rs_ce(v[i], v[i + 4], asc);
#pragma unroll
  for (int i = 0; i < 8; i++)
    if (!(i & 2)) rs_ce(v[i], v[i | 2], asc);
#pragma unroll
  for (int i = 0; i < 8; i += 2) rs_ce(v[i], v[i + 1], asc);
}
// Sort s[rs_sw(0 .. N)) ascending; N a power of two in [8, 8 T]; threads tid < N / 8 work, all threads pass the barriers.
// (not inlined: the network is the fallback of the bucket sort below, and its 16 + 16 key registers would otherwise count
//  against the occupancy of the kernel's common path)
template <typename K, int T>
__device__ __attribute__((noinline)) void rs_bitonic(K* __restrict__ s, u32 N, int tid) {
  const bool active = (u32)tid < (N >> 3);
  K v[RS_EPT];
  // phases k = 2, 4, 8: inside a thread's eight consecutive elements
  if (active) {
    const u32 base = (u32)tid << 3;
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = s[rs_sw(base + i)];
#pragma unroll
    for (int i = 0; i < 8; i += 2) rs_ce(v[i], v[i + 1], (i & 2) == 0);  // k = 2
#pragma unroll
    for (int i = 0; i < 8; i++)
      if (!(i & 2)) rs_ce(v[i], v[i | 2], (i & 4) == 0);  // k = 4, j = 2
#pragma unroll
    for (int i = 0; i < 8; i += 2) rs_ce(v[i], v[i + 1], (i & 4) == 0);  // k = 4, j = 1
    rs_three_levels(v, (base & 8u) == 0);  // k = 8
#pragma unroll
    for (int i = 0; i < 8; i++) s[rs_sw(base + i)] = v[i];
  }
  rs_sync<T>();
  for (u32 k = 16; k <= N; k <<= 1) {
    // Levels k/2 ... 1 of this merge, three per round trip through LDS: a group works on the strides (4q, 2q, q).  q starts
    // at k/8 and drops by 8 while a full group is left; the last group is always (4, 2, 1).  Where that repeats a level
    // already done (k = 16: (8, 4, 2) then (4, 2, 1)) the repeat is a no-op: after level j every element of a 2j-block's
    // lower half is <= every element of its upper half, which the lower levels -- permutations inside the halves -- keep.
    for (u32 q = k >> 3;;) {
      if (active) {
        const u32 lq = 31u - (u32)__builtin_clz(q);
        const u32 base = (((u32)tid >> lq) << (lq + 3)) | ((u32)tid & (q - 1));
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = s[rs_sw(base + (u32)i * q)];
        rs_three_levels(v, (base & k) == 0);
#pragma unroll
        for (int i = 0; i < 8; i++) s[rs_sw(base + (u32)i * q)] = v[i];
      }
      rs_sync<T>();
      if (q == 1) break;
      q = q >= 8 ? (q >> 3) : 1u;
    }
  }
}

// The cheap sort of a run: payloads of one key are usually spread evenly over their range (row ids, timestamps), so a run of
// n keys falls into NB >= n value-range buckets of about one key each.  Count the buckets (one LDS atomic per key), scan,
// move every key to its bucket, let it rank itself among the one to three keys that share the bucket, write it to its
// final slot: O(n) with ~60 instructions per key where the bitonic network spends ~420 (profiles/r05b_small16_ord_*: the
// network alone was 0.7 ms of VALU time at 2^26 rows, and a run of 1025 keys paid for 2048).  Payloads that tie or cluster
// make long buckets: beyond RS_MAXBUCKET keys in one the function returns false and the network sorts the run.
// tmp: n keys; idx: n 16-bit slots; cnt: NB + 1 words.  On success the i-th smallest key is tmp[idx[i]].  All threads call it.
// STABLE: equal keys keep the order of their positions i in the run (oidx: n 16-bit slots; without it ties are broken by
// the order the atomics happened to number them -- fine where only the keys are of interest).  slot_out[r]: where the
// thread's r-th key was put in tmp (the caller moves the keys' payloads through the same slots).
constexpr u32 RS_MAXBUCKET = 24;
template <typename K, int T, bool STABLE = false>
__device__ __forceinline__ bool rs_bucket_sort(const u64 (&sv)[RS_EPT], u64 mn, u64 range, u32 n, K* __restrict__ tmp, unsigned short* __restrict__ idx,
                                               u32* __restrict__ cnt, u32* __restrict__ wsc, int tid, unsigned short* __restrict__ oidx = nullptr,
                                               u32* slot_out = nullptr) {
  const int lane = tid & 63, wv = tid >> 6;
  u32 NB = T;
  while (NB < n) NB <<= 1;
  const int lg = 31 - __builtin_clz(NB);
  const int bits = 64 - __builtin_clzll(range);  // (range > 0)
  const int sh = bits > lg ? bits - lg : 0;
  for (u32 i = (u32)tid; i <= NB; i += T) cnt[i] = 0;
  rs_sync<T>();
  u32 bk[RS_EPT], ar[RS_EPT];
#pragma unroll
  for (int r = 0; r < RS_EPT; r++) {
    const u32 i = (u32)tid + (u32)r * T;
    bk[r] = ar[r] = 0;
    if (i < n) {
      bk[r] = (u32)((sv[r] - mn) >> sh);
      ar[r] = atomicAdd(&cnt[bk[r]], 1u);
    }
  }
  rs_sync<T>();
  // exclusive scan of the NB counts in place (thread t: entries [t * E, (t + 1) * E)), and the longest bucket
  const u32 E = NB / T;  // 1 ... 8
  u32 loc[RS_EPT], sum = 0, big = 0;
#pragma unroll
  for (int e = 0; e < RS_EPT; e++) {
    loc[e] = (u32)e < E ? cnt[(u32)tid * E + (u32)e] : 0u;
    sum += loc[e];
    big = loc[e] > big ? loc[e] : big;
  }
  u32 incl = sum;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const u32 up = __shfl_up(incl, o, kWave);
    if (lane >= o) incl += up;
  }
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) {
    const u32 b2 = __shfl_xor(big, o, kWave);
    big = b2 > big ? b2 : big;
  }
  if (lane == kWave - 1) wsc[wv] = incl;
  if (lane == 0) wsc[T / kWave + wv] = big;
  rs_sync<T>();
  u32 before = 0;
#pragma unroll
  for (int w = 0; w < T / kWave; w++) {
    if (w < wv) before += wsc[w];
    big = wsc[T / kWave + w] > big ? wsc[T / kWave + w] : big;
  }
  if (big > RS_MAXBUCKET) {  // (uniform) ties or clustered payloads
    rs_sync<T>();
    return false;
  }
  u32 run = before + incl - sum;
#pragma unroll
  for (int e = 0; e < RS_EPT; e++)
    if ((u32)e < E) {
      cnt[(u32)tid * E + (u32)e] = run;
      run += loc[e];
    }
  if (tid == T - 1) cnt[NB] = n;
  rs_sync<T>();
  // keys to their buckets (arrival order), then every key ranks itself inside its bucket
#pragma unroll
  for (int r = 0; r < RS_EPT; r++) {
    const u32 i = (u32)tid + (u32)r * T;
    if (i < n) {
      const u32 me = cnt[bk[r]] + ar[r];
      tmp[me] = (K)(sv[r] - mn);
      if (STABLE) oidx[me] = (unsigned short)i;
      if (slot_out) slot_out[r] = me;
    }
  }
  rs_sync<T>();
#pragma unroll
  for (int r = 0; r < RS_EPT; r++) {
    const u32 i = (u32)tid + (u32)r * T;
    if (i < n) {
      const u32 s0 = cnt[bk[r]], e0 = cnt[bk[r] + 1], me = s0 + ar[r];
      const K key = (K)(sv[r] - mn);
      u32 rank = 0;
      for (u32 t = s0; t < e0; t++) {
        const K kt = tmp[t];
        const bool before = STABLE ? (u32)oidx[t] < i : t < me;
        rank += (kt < key || (kt == key && before)) ? 1u : 0u;
      }
      idx[s0 + rank] = (unsigned short)me;
    }
  }
  rs_sync<T>();
  return true;
}

// GROUP: more than 2^18 ranks -- a partition holds the runs of 2^gb consecutive ranks (gb = -tb), its rows are sorted by
// (rank's low gb bits, sval) as ONE word, sub << bits(range) | (sval - min): the same sorts on a composite.  Payloads that
// leave no room for the gb bits (a range beyond 2^(64 - gb)) raise ERR_FASTPATH.
template <bool EXTRA, bool GROUP, int T>
__global__ __launch_bounds__(T == 64 ? 256 : T, T <= 256 ? 5 : 4) void rank_sort_write_kernel(const Tup* __restrict__ slabs, const u32* __restrict__ cnt, u32 cap,
                                                                     u32 P, const u64* __restrict__ out_off,
                                                                     const Tup* __restrict__ sortedR, u32 nb, int tb, u64* __restrict__ out_key,
                                                                     u64* __restrict__ out_rval, u64* __restrict__ out_sval,
                                                                     u64* __restrict__ accum) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem_raw[];
  constexpr bool WAVE = T == 64;  // (rs_sync above: every wave of the workgroup is a sorter of its own)
  RsSmem<T, false>* const sms = reinterpret_cast<RsSmem<T, false>*>(rs_smem_raw);
  RsSmem<T, false>& sm = sms[WAVE ? threadIdx.x >> 6 : 0];
  u64* const keys = sm.keys;
  unsigned short* const sidx = sm.sidx;
  u32* const bcnt = sm.bcnt;
  u32* const wsc = sm.wsc;
  u64* const red = sms[0].red;
  u64* const wmn = sm.wmn;
  u64* const wmx = sm.wmx;
  constexpr u32 RS_CAP = T * RS_EPT, RS_THREADS = T;
  const int tid = WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (threadIdx.x < 8) red[threadIdx.x] = 0;
  u64 acc_r = 0, acc_x = 0, acc_m = 0;
  bool bad = false;
  const u32 p_first = WAVE ? blockIdx.x * 4u + (threadIdx.x >> 6) : blockIdx.x, p_step = WAVE ? gridDim.x * 4u : gridDim.x;
  for (u32 p = p_first; p < P; p += p_step) {
    const u32* q = cnt + (u64)p * SLAB_KB;
    const u32 c0 = q[0], c1 = q[1], c2 = q[2], c3 = q[3];
    const u32 n = c0 + c1 + c2 + c3;
    if (n == 0) continue;
    const int gb = GROUP ? -tb : 0;
    if (n > (u32)RS_CAP || (GROUP ? ((u64)p << gb) : (u64)(p >> tb)) >= nb || c0 > cap || c1 > cap || c2 > cap || c3 > cap) {  // (uniform) a run beyond the kernel, or counts no slab pass wrote
      bad = true;
      continue;
    }
    const Tup* __restrict__ base = slabs + (u64)p * SLAB_KB * cap;
    // ---- the run's payloads, coalesced piece by piece; their range
    u64 sv[RS_EPT];
    u32 sub[GROUP ? RS_EPT : 1];
    u64 mn = ~0ull, mx = 0;
#pragma unroll
    for (int r = 0; r < RS_EPT; r++) {
      const u32 i = (u32)tid + (u32)r * RS_THREADS;
      sv[r] = 0;
      if (GROUP) sub[r] = 0;
      if (i < n) {
        const u32 k = i < c0 ? 0u : i < c0 + c1 ? 1u : i < c0 + c1 + c2 ? 2u : 3u;
        const u32 start = k == 0 ? 0u : k == 1 ? c0 : k == 2 ? c0 + c1 : c0 + c1 + c2;
        if (GROUP) {
          const Tup row = load_stream(&base[(u64)k * cap + (i - start)]);
          sv[r] = row.val;
          sub[r] = (u32)row.key & ((1u << gb) - 1u);
        } else {
          sv[r] = load_stream(&base[(u64)k * cap + (i - start)]).val;
        }
        mn = sv[r] < mn ? sv[r] : mn;
        mx = sv[r] > mx ? sv[r] : mx;
      }
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
      const u64 a2 = __shfl_xor(mn, o, kWave), b2 = __shfl_xor(mx, o, kWave);
      mn = a2 < mn ? a2 : mn;
      mx = b2 > mx ? b2 : mx;
    }
    if (lane == 0) {
      wmn[wv] = mn;
      wmx[wv] = mx;
    }
    rs_sync<T>();
#pragma unroll
    for (int w = 0; w < RS_THREADS / kWave; w++) {
      mn = wmn[w] < mn ? wmn[w] : mn;
      mx = wmx[w] > mx ? wmx[w] : mx;
    }
    u64 range = mx - mn;
    const u64 mn_s = mn;
    int rb = 0;
    if (GROUP) {  // the composite: (rank's low bits, sval - min) in one word
      rb = range ? 64 - __builtin_clzll(range) : 0;
      if (rb + gb > 64) {  // (uniform)
        bad = true;
        rs_sync<T>();
        continue;
      }
#pragma unroll
      for (int r = 0; r < RS_EPT; r++) sv[r] = (rb < 64 ? (u64)sub[r] << rb : 0ull) | (sv[r] - mn_s);
      range = (rb < 64 ? (u64)((1u << gb) - 1u) << rb : 0ull) | range;
      mn = 0;
    }
    u32 N = 8;
    while (N < n) N <<= 1;
    // (tb > 0: a key's run is cut into 2^tb partitions by the position of the payload in the payloads' range)
    const Tup b = GROUP ? Tup{0, 0} : sortedR[p >> tb];
    const u64 off = out_off[p];
    if (range == 0) {
      // one payload value in the whole run: nothing to sort
    } else if (range < 0xFFFFFFFFull) {  // 32-bit keys sval - min
      u32* k32 = reinterpret_cast<u32*>(keys);
      if (rs_bucket_sort<u32, T>(sv, mn, range, n, k32, sidx, bcnt, wsc, tid)) {
#pragma unroll
        for (int r = 0; r < RS_EPT; r++) {
          const u32 i = (u32)tid + (u32)r * RS_THREADS;
          if (i < n) sv[r] = (u64)k32[sidx[i]] + mn;
        }
      } else {  // the padding (all ones) is above every real key
#pragma unroll
        for (int r = 0; r < RS_EPT; r++) {
          const u32 i = (u32)tid + (u32)r * RS_THREADS;
          if (i < N) k32[rs_sw(i)] = i < n ? (u32)(sv[r] - mn) : 0xFFFFFFFFu;
        }
        rs_sync<T>();
        rs_bitonic<u32, T>(k32, N, tid);
#pragma unroll
        for (int r = 0; r < RS_EPT; r++) {
          const u32 i = (u32)tid + (u32)r * RS_THREADS;
          if (i < n) sv[r] = (u64)k32[rs_sw(i)] + mn;
        }
      }
    } else if (range != ~0ull && rs_bucket_sort<u64, T>(sv, mn, range, n, keys, sidx, bcnt, wsc, tid)) {
#pragma unroll
      for (int r = 0; r < RS_EPT; r++) {
        const u32 i = (u32)tid + (u32)r * RS_THREADS;
        if (i < n) sv[r] = keys[sidx[i]] + mn;
      }
    } else if (range != ~0ull) {
#pragma unroll
      for (int r = 0; r < RS_EPT; r++) {
        const u32 i = (u32)tid + (u32)r * RS_THREADS;
        if (i < N) keys[rs_sw(i)] = i < n ? sv[r] - mn : ~0ull;
      }
      rs_sync<T>();
      rs_bitonic<u64, T>(keys, N, tid);
#pragma unroll
      for (int r = 0; r < RS_EPT; r++) {
        const u32 i = (u32)tid + (u32)r * RS_THREADS;
        if (i < n) sv[r] = keys[rs_sw(i)] + mn;
      }
    } else {  // payloads 0 and 2^64 - 1 in one run: no value is free for the padding
      bad = true;
      rs_sync<T>();
      continue;
    }
    // ---- rows out: the rank's key and payload, the sorted probe payloads
#pragma unroll
    for (int r = 0; r < RS_EPT; r++) {
      const u32 i = (u32)tid + (u32)r * RS_THREADS;
      if (i < n) {
        Tup br = b;
        u64 s1 = sv[r];
        if (GROUP) {
          br = sortedR[((u64)p << gb) + (rb < 64 ? s1 >> rb : 0ull)];
          s1 = (rb < 64 ? s1 & (((u64)1 << rb) - 1) : s1) + mn_s;
          acc_r += br.val;
        }
        out_key[off + i] = br.key;
        out_rval[off + i] = br.val;
        out_sval[off + i] = s1;
        if (EXTRA) {
          const u64 m = tmix(br.key, br.val, s1);
          acc_x ^= m;
          acc_m += m;
        }
      }
    }
    if (!GROUP && tid == 0) acc_r += b.val * (u64)n;
    rs_sync<T>();  // (keys / wmn / wmx are rewritten by the next run)
  }
  if (bad && tid == 0) atomicOr(reinterpret_cast<unsigned long long*>(&accum[ACC_ERR]), (unsigned long long)ERR_FASTPATH);
  __syncthreads();
  const u64 v[6] = {0, acc_r, 0, acc_x, acc_m, 0};
  block_accumulate(red, accum, v, 1u << ACC_XOR);
}

// off[p] = rows of the slab partitions 0 .. p - 1 (each partition SLAB_KB piece counts, one 16-byte load), off[P] = all rows.
// Two small launches over ceil(P / 1024) workgroups: chunk totals, then every workgroup adds up the totals before its chunk
// (<= 256 values) and scans its 1024 partitions.  (ONE workgroup for the whole array pulled 2 MB through one CU: 0.07-0.11 ms
// at 2^16 partitions, profiles/r05c / r05d.)
constexpr u32 SO_CHUNK = 1024;
__device__ __forceinline__ u64 so_total(const uint4* __restrict__ c4, u32 i, u32 P) {
  if (i >= P) return 0ull;
  const uint4 q = c4[i];
  return (u64)q.x + q.y + q.z + q.w;
}
__global__ __launch_bounds__(256) void slab_totals_kernel(const u32* __restrict__ cnt, u32 P, u64* __restrict__ tot) {
  static_assert(SLAB_KB == 4, "one 16-byte load per partition");
  __shared__ u64 w4[4];
  const uint4* __restrict__ c4 = reinterpret_cast<const uint4*>(cnt);
  const u32 b = blockIdx.x * SO_CHUNK + threadIdx.x;
  u64 s = so_total(c4, b, P) + so_total(c4, b + 256, P) + so_total(c4, b + 512, P) + so_total(c4, b + 768, P);
  s = wave_sum_u64(s);
  if ((threadIdx.x & 63) == 0) w4[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) tot[blockIdx.x] = w4[0] + w4[1] + w4[2] + w4[3];
}
__global__ __launch_bounds__(256) void slab_offsets_kernel(const u32* __restrict__ cnt, u32 P, const u64* __restrict__ tot, u32 n_chunks,
                                                           u64* __restrict__ off) {
  __shared__ u64 w4[4];
  const uint4* __restrict__ c4 = reinterpret_cast<const uint4*>(cnt);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  u64 before = 0, all = 0;  // chunks before this one / all chunks (n_chunks <= 2^22 / 1024: a few values per thread)
  for (u32 i = (u32)tid; i < n_chunks; i += 256) {
    const u64 t = tot[i];
    all += t;
    if (i < blockIdx.x) before += t;
  }
  before = wave_sum_u64(before);
  all = wave_sum_u64(all);
  if (lane == 0) w4[wv] = before;
  __syncthreads();
  const u64 base = w4[0] + w4[1] + w4[2] + w4[3];
  __syncthreads();
  if (blockIdx.x == 0) {
    if (lane == 0) w4[wv] = all;
    __syncthreads();
    if (tid == 0) off[P] = w4[0] + w4[1] + w4[2] + w4[3];
    __syncthreads();
  }
  u64 carry = base;
#pragma unroll
  for (int k = 0; k < 4; k++) {  // 256 partitions per step, in order
    const u32 i = blockIdx.x * SO_CHUNK + (u32)k * 256 + (u32)tid;
    const u64 v = so_total(c4, i, P);
    u64 incl = v;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const u64 up = __shfl_up(incl, o, kWave);
      if (lane >= o) incl += up;
    }
    if (lane == kWave - 1) w4[wv] = incl;
    __syncthreads();
    u64 pre = 0, step = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      if (w < wv) pre += w4[w];
      step += w4[w];
    }
    if (i < P) off[i] = carry + pre + incl - v;
    carry += step;
    __syncthreads();
  }
}
hipError_t launch_slab_offsets(const u32* cnt, u32 P, u64* off, u64* scratch, hipStream_t st) {
  if (!cnt || !off || !scratch || P == 0) return hipErrorInvalidValue;
  const u32 chunks = (P + SO_CHUNK - 1) / SO_CHUNK;
  hipLaunchKernelGGL(slab_totals_kernel, dim3(chunks), dim3(256), 0, st, cnt, P, scratch);
  hipLaunchKernelGGL(slab_offsets_kernel, dim3(chunks), dim3(256), 0, st, cnt, P, scratch, chunks, off);
  return hipGetLastError();
}

// ---- hmj_sort_u64_device, round 5: two MSD slab passes + this kernel -------------------------------------------------------
// The LSD chain moves every row through one pass per varying digit (nine passes of 32 B for uniform 64-bit keys).  The
// reference's own sort is MSD (radix_hash.h:202-292: partition on the top bits, recurse into the buckets, insertion-sort the
// small ones); so is this: the join's two slab passes partition the rows on the top 12 ... 18 varying key bits (LSD inside
// that window, so partition p = the window's value, pieces in input order), and one workgroup per partition sorts its run
// of <= 2048 rows on the REMAINING low bits in LDS -- the bucket-rank sort above, stable (ties by position in the run = input
// order, as the passes are stable), 32-bit keys where the run spans < 2^32 -- and writes the rows at the partition's offset:
// 3 x 32 B per row whatever the key width.  A run beyond the kernel or a bucket of > 24 equal / clustered keys raises
// ERR_FASTPATH: the output is garbage, the host runs the chain from the untouched input (out-of-place calls only).
template <int T>
__global__ __launch_bounds__(T, 4) void sort_runs_write_kernel(const Tup* __restrict__ slabs, const u32* __restrict__ cnt, u32 cap, u32 P,
                                                                        const u64* __restrict__ out_off, Tup* __restrict__ out,
                                                                        u64* __restrict__ accum) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem_raw[];
  RsSmem<T, true>& sm = *reinterpret_cast<RsSmem<T, true>*>(rs_smem_raw);
  u64* const keys = sm.keys;  // the run's keys by bucket slot, later its payloads by the same slots
  unsigned short* const sidx = sm.sidx;
  unsigned short* const oidx = sm.oidx;
  u32* const bcnt = sm.bcnt;
  u32* const wsc = sm.wsc;
  u64* const wmn = sm.wmn;
  u64* const wmx = sm.wmx;
  constexpr u32 RS_CAP = T * RS_EPT, RS_THREADS = T;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  bool bad = false;
  for (u32 p = blockIdx.x; p < P; p += gridDim.x) {
    u32 c0, c1 = 0, c2 = 0, c3 = 0;
    const Tup* __restrict__ base;
    if (cnt) {
      const u32* q = cnt + (u64)p * SLAB_KB;
      c0 = q[0], c1 = q[1], c2 = q[2], c3 = q[3];
      base = slabs + (u64)p * SLAB_KB * cap;
    } else {  // dense partitions (one exact radix pass in front): partition p is rows [out_off[p], out_off[p + 1]) of `slabs` too
      const u64 o0 = out_off[p], o1 = out_off[p + 1];
      c0 = o1 - o0 > (u64)RS_CAP ? (u32)RS_CAP + 1u : (u32)(o1 - o0);
      base = slabs + o0;
    }
    const u32 n = c0 + c1 + c2 + c3;
    if (n == 0) continue;
    if (n > (u32)RS_CAP || c0 > cap || c1 > cap || c2 > cap || c3 > cap) {  // (uniform)
      bad = true;
      continue;
    }
    u64 kv[RS_EPT], pv[RS_EPT];
    u64 mn = ~0ull, mx = 0;
#pragma unroll
    for (int r = 0; r < RS_EPT; r++) {
      const u32 i = (u32)tid + (u32)r * RS_THREADS;
      kv[r] = pv[r] = 0;
      if (i < n) {
        const u32 k = i < c0 ? 0u : i < c0 + c1 ? 1u : i < c0 + c1 + c2 ? 2u : 3u;
        const u32 start = k == 0 ? 0u : k == 1 ? c0 : k == 2 ? c0 + c1 : c0 + c1 + c2;
        const Tup t = load_stream(&base[(u64)k * cap + (i - start)]);
        kv[r] = t.key;
        pv[r] = t.val;
        mn = kv[r] < mn ? kv[r] : mn;
        mx = kv[r] > mx ? kv[r] : mx;
      }
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
      const u64 a2 = __shfl_xor(mn, o, kWave), b2 = __shfl_xor(mx, o, kWave);
      mn = a2 < mn ? a2 : mn;
      mx = b2 > mx ? b2 : mx;
    }
    if (lane == 0) {
      wmn[wv] = mn;
      wmx[wv] = mx;
    }
    lds_barrier();
#pragma unroll
    for (int w = 0; w < RS_THREADS / kWave; w++) {
      mn = wmn[w] < mn ? wmn[w] : mn;
      mx = wmx[w] > mx ? wmx[w] : mx;
    }
    const u64 range = mx - mn;
    const u64 off = out_off[p];
    u32 slot[RS_EPT];
    bool sorted_ok = true;
    if (range == 0) {  // one key in the whole run: input order is the sorted order
#pragma unroll
      for (int r = 0; r < RS_EPT; r++) {
        const u32 i = (u32)tid + (u32)r * RS_THREADS;
        if (i < n) {
          Tup t;
          t.key = kv[r];
          t.val = pv[r];
          store_stream(&out[off + i], t);
        }
      }
      lds_barrier();
      continue;
    } else if (range < 0xFFFFFFFFull) {
      u32* k32 = reinterpret_cast<u32*>(keys);
      sorted_ok = rs_bucket_sort<u32, T, true>(kv, mn, range, n, k32, sidx, bcnt, wsc, tid, oidx, slot);
      if (sorted_ok) {
#pragma unroll
        for (int r = 0; r < RS_EPT; r++) {
          const u32 i = (u32)tid + (u32)r * RS_THREADS;
          if (i < n) kv[r] = (u64)k32[sidx[i]] + mn;
        }
      }
    } else {
      sorted_ok = rs_bucket_sort<u64, T, true>(kv, mn, range, n, keys, sidx, bcnt, wsc, tid, oidx, slot);
      if (sorted_ok) {
#pragma unroll
        for (int r = 0; r < RS_EPT; r++) {
          const u32 i = (u32)tid + (u32)r * RS_THREADS;
          if (i < n) kv[r] = keys[sidx[i]] + mn;
        }
      }
    }
    if (!sorted_ok) {  // (uniform) ties or clustered keys: the host takes the chain
      bad = true;
      continue;
    }
    lds_barrier();  // every thread has read its sorted keys: the array now carries the payloads, by the same slots
#pragma unroll
    for (int r = 0; r < RS_EPT; r++) {
      const u32 i = (u32)tid + (u32)r * RS_THREADS;
      if (i < n) keys[slot[r]] = pv[r];
    }
    lds_barrier();
#pragma unroll
    for (int r = 0; r < RS_EPT; r++) {
      const u32 i = (u32)tid + (u32)r * RS_THREADS;
      if (i < n) {
        Tup t;
        t.key = kv[r];
        t.val = keys[sidx[i]];
        store_stream(&out[off + i], t);
      }
    }
    lds_barrier();
  }
  if (bad && tid == 0) atomicOr(reinterpret_cast<unsigned long long*>(&accum[ACC_ERR]), (unsigned long long)ERR_FASTPATH);
}

// level: 0 / 1 / 2 = workgroups of 256 / 512 / 1024 threads, partitions of up to 2048 / 4096 / 8192 rows (rank_sort_max_run)
template <typename Kern>
static hipError_t rs_launch_prep(Kern kern, size_t smem, SmemAttrOnce& once) {
  return smem > 64 * 1024 ? ensure_max_smem(once, reinterpret_cast<const void*>(kern), smem) : hipSuccess;
}
hipError_t launch_sort_runs_write(const void* slabs, const u32* cnt, u32 cap, u32 P, const u64* out_off, void* out, u64* accum, int level,
                                  int num_cus, hipStream_t st) {
  // cnt == nullptr: dense partitions -- `slabs` holds the rows partitioned in place of the output, partition p = rows
  // [out_off[p], out_off[p + 1]) (P + 1 offsets), cap is not looked at
  if (!slabs || !out_off || !out || !accum || P == 0 || (cnt && cap == 0) || level < 0 || level > 2) return hipErrorInvalidValue;
  if (!cnt) cap = 0xFFFFFFFFu;
  u32 grid = (u32)num_cus * (level == 0 ? 4u : level == 1 ? 2u : 1u);  // what is resident at once (launch bounds; 32 / 64 / 129 KiB of LDS per workgroup)
  if (grid > P) grid = P;
#define HMJ_SRW(T)                                                                                                                    \
  {                                                                                                                                   \
    static SmemAttrOnce once;                                                                                                         \
    const size_t smem = sizeof(RsSmem<T, true>);                                                                                      \
    if (hipError_t e = rs_launch_prep(sort_runs_write_kernel<T>, smem, once); e != hipSuccess) return e;                              \
    hipLaunchKernelGGL((sort_runs_write_kernel<T>), dim3(grid), dim3(T), smem, st, static_cast<const Tup*>(slabs), cnt, cap, P, out_off, \
                       static_cast<Tup*>(out), accum);                                                                                \
  }
  if (level == 0) HMJ_SRW(256) else if (level == 1) HMJ_SRW(512) else HMJ_SRW(1024)
#undef HMJ_SRW
  return hipGetLastError();
}

int rank_sort_max_run(int level) { return level < 0 ? 64 * RS_EPT : (256 << (level > 2 ? 2 : level)) * RS_EPT; }
hipError_t launch_rank_sort_write(const void* slabs, const u32* cnt, u32 cap, u32 P, const u64* out_off, const void* sortedR, u32 nb, int tb,
                                  u64* out_key, u64* out_rval, u64* out_sval, u64* accum, bool extra, int level, int num_cus, hipStream_t st) {
  // tb > 0: a rank's run is 2^tb partitions; tb < 0: a partition is the runs of 2^-tb consecutive ranks
  if (!slabs || !cnt || !out_off || !sortedR || !out_key || !out_rval || !out_sval || !accum || P == 0 || cap == 0 || nb == 0 || tb < -4 || tb > 16 ||
      level < -1 || level > 2)
    return hipErrorInvalidValue;
  // what is resident at once: 96 registers (launch bounds) and 28 KiB of LDS per 256-thread workgroup -- five per CU; two of 512 threads; one of 1024
  u32 grid = (u32)num_cus * (level <= 0 ? 5u : level == 1 ? 2u : 1u);
  const u32 per_wg = level < 0 ? 4u : 1u;  // (the wave shape: four partitions in flight per workgroup)
  if (grid > (P + per_wg - 1) / per_wg) grid = (P + per_wg - 1) / per_wg;
#define HMJ_RSW(E, G, T)                                                                                                          \
  {                                                                                                                               \
    static SmemAttrOnce once;                                                                                                     \
    const size_t smem = sizeof(RsSmem<T, false>) * (T == 64 ? 4 : 1);                                                             \
    if (hipError_t e = rs_launch_prep(rank_sort_write_kernel<E, G, T>, smem, once); e != hipSuccess) return e;                    \
    hipLaunchKernelGGL((rank_sort_write_kernel<E, G, T>), dim3(grid), dim3(T == 64 ? 256 : T), smem, st, static_cast<const Tup*>(slabs), cnt, cap, \
                       P, out_off, static_cast<const Tup*>(sortedR), nb, tb, out_key, out_rval, out_sval, accum);                  \
  }
#define HMJ_RSW_T(E, G)                                                                                              \
  {                                                                                                                  \
    if (level < 0) HMJ_RSW(E, G, 64) else if (level == 0) HMJ_RSW(E, G, 256) else if (level == 1) HMJ_RSW(E, G, 512) \
    else HMJ_RSW(E, G, 1024)                                                                                         \
  }
  if (tb < 0) {
    if (extra) HMJ_RSW_T(true, true) else HMJ_RSW_T(false, true)
  } else {
    if (extra) HMJ_RSW_T(true, false) else HMJ_RSW_T(false, false)
  }
#undef HMJ_RSW_T
#undef HMJ_RSW
  return hipGetLastError();
}


hipError_t launch_pieces_compact(const void* slabs, const u32* cnt, const u64* off, u32 n_pieces, u32 cap, void* out, int num_cus,
                                 hipStream_t st) {
  if (!slabs || !cnt || !off || !out || n_pieces == 0 || cap == 0) return hipErrorInvalidValue;
  u32 waves = (u32)num_cus * 32u;
  if (waves > n_pieces) waves = n_pieces;
  const u32 ppw = (n_pieces + waves - 1) / waves;
  const u32 grid = ((n_pieces + ppw - 1) / ppw + 3) / 4;
  hipLaunchKernelGGL(pieces_compact_kernel, dim3(grid), dim3(256), 0, st, static_cast<const Tup*>(slabs), cnt, off, n_pieces, cap, ppw,
                     static_cast<Tup*>(out));
  return hipGetLastError();
}

hipError_t launch_piece_offsets(const u32* cnt, u32 n_pieces, u64* off, hipStream_t st) {
  if (!cnt || !off || n_pieces == 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(piece_offsets_kernel, dim3(1), dim3(1024), 0, st, cnt, n_pieces, off);
  return hipGetLastError();
}

hipError_t launch_gtable_expand_pieces(const void* slabs, const u32* cnt, const u64* off, u32 n_pieces, u32 cap, const void* sortedR,
                                       u64 svmin, int range_bits, u64* out_key, u64* out_rval, u64* out_sval, u64* accum, bool extra,
                                       int num_cus, hipStream_t st) {
  if (!slabs || !cnt || !off || !sortedR || !out_key || !out_rval || !out_sval || n_pieces == 0 || cap == 0 || range_bits < 0 ||
      range_bits > 64)
    return hipErrorInvalidValue;
  u32 waves = (u32)num_cus * 32u;  // eight workgroups of four waves per CU
  if (waves > n_pieces) waves = n_pieces;
  const u32 ppw = (n_pieces + waves - 1) / waves;
  const u32 grid = ((n_pieces + ppw - 1) / ppw + 3) / 4;
  if (extra)
    hipLaunchKernelGGL((gtable_expand_pieces_kernel<true>), dim3(grid), dim3(256), 0, st, static_cast<const Tup*>(slabs), cnt, off, n_pieces,
                       cap, ppw, static_cast<const Tup*>(sortedR), svmin, range_bits, out_key, out_rval, out_sval, accum);
  else
    hipLaunchKernelGGL((gtable_expand_pieces_kernel<false>), dim3(grid), dim3(256), 0, st, static_cast<const Tup*>(slabs), cnt, off, n_pieces,
                       cap, ppw, static_cast<const Tup*>(sortedR), svmin, range_bits, out_key, out_rval, out_sval, accum);
  return hipGetLastError();
}

hipError_t launch_gtable_build(const void* R, u32 nb, void* tab, int log_cap, u64* accum, bool first, int num_cus,
                               hipStream_t st) {
  if (log_cap < 4 || log_cap > 30 || ((u64)1 << log_cap) < (u64)nb + nb / 4 + 1) return hipErrorInvalidValue;  // (load factor <= 0.8: walks must end)
  u64 grid = ((u64)nb + GT_THREADS - 1) / GT_THREADS;
  if (grid > (u64)num_cus * 16) grid = (u64)num_cus * 16;
  if (grid < 1) grid = 1;
  if (first)
    hipLaunchKernelGGL((gtable_build_kernel<true>), dim3((u32)grid), dim3(GT_THREADS), 0, st, static_cast<const Tup*>(R), nb,
                       static_cast<Tup*>(tab), log_cap, accum);
  else
    hipLaunchKernelGGL((gtable_build_kernel<false>), dim3((u32)grid), dim3(GT_THREADS), 0, st, static_cast<const Tup*>(R), nb,
                       static_cast<Tup*>(tab), log_cap, accum);
  return hipGetLastError();
}

hipError_t launch_gtable_probe(const void* S, u32 np, const void* tab, int log_cap, const void* R, u64* accum, bool first,
                               bool extra, int num_cus, int wg_per_cu, hipStream_t st) {
  if (log_cap < 4 || log_cap > 30) return hipErrorInvalidValue;
  const u64 tiles = ((u64)np + GT_THREADS * GT_ROWS - 1) / (GT_THREADS * GT_ROWS);
  u64 grid = (u64)num_cus * (u64)(wg_per_cu > 0 ? wg_per_cu : 8);
  if (grid > tiles) grid = tiles;
  if (grid < 1) grid = 1;
#define HMJ_GT(F, E)                                                                                                   \
  hipLaunchKernelGGL((gtable_probe_kernel<F, E>), dim3((u32)grid), dim3(GT_THREADS), 0, st, static_cast<const Tup*>(S), np, \
                     static_cast<const Tup*>(tab), log_cap, static_cast<const Tup*>(R), accum)
  if (first) {
    if (extra) HMJ_GT(true, true); else HMJ_GT(true, false);
  } else {
    if (extra) HMJ_GT(false, true); else HMJ_GT(false, false);
  }
#undef HMJ_GT
  return hipGetLastError();
}

hipError_t launch_gtable_write(const void* S, u32 np, const void* tab, int log_cap, const void* R, u64* accum, u64* out_key,
                               u64* out_rval, u64* out_sval, bool first, bool extra, int num_cus, int wg_per_cu, hipStream_t st) {
  if (log_cap < 4 || log_cap > 30 || !out_key || !out_rval || !out_sval) return hipErrorInvalidValue;
  const u64 tiles = ((u64)np + GTW_THREADS * GT_ROWS - 1) / (GTW_THREADS * GT_ROWS);
  u64 grid = (u64)num_cus * (u64)(wg_per_cu > 0 ? wg_per_cu : 8) * GT_THREADS / GTW_THREADS;  // (wg_per_cu counts 256-thread workgroups)
  if (grid > tiles) grid = tiles;
  if (grid < 1) grid = 1;
#define HMJ_GTW(F, E)                                                                                                  \
  hipLaunchKernelGGL((gtable_write_kernel<F, E>), dim3((u32)grid), dim3(GTW_THREADS), 0, st, static_cast<const Tup*>(S), np, \
                     static_cast<const Tup*>(tab), log_cap, static_cast<const Tup*>(R), accum, out_key, out_rval, out_sval)
  if (first) {
    if (extra) HMJ_GTW(true, true); else HMJ_GTW(true, false);
  } else {
    if (extra) HMJ_GTW(false, true); else HMJ_GTW(false, false);
  }
#undef HMJ_GTW
  return hipGetLastError();
}

hipError_t launch_sval_range(const void* S, u32 np, u64* out2, int num_cus, hipStream_t st, u32 every) {
  if (every < 1) every = 1;
  const u32 rows = (u32)(((u64)np + every - 1) / every);
  u64 grid = ((u64)rows + 1023) / 1024;  // (a workgroup per 1024 rows at most: a tiny relation does not launch 2048 of them)
  if (grid > (u64)num_cus * 8) grid = (u64)num_cus * 8;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(sval_range_kernel, dim3((u32)grid), dim3(256), 0, st, static_cast<const Tup*>(S), rows, every, out2);
  return hipGetLastError();
}

hipError_t launch_gtable_emit_ranks(const void* S, u32 np, const void* tab, int log_cap, u64* accum, void* pairs, bool extra, int num_cus,
                                    int wg_per_cu, hipStream_t st) {
  if (log_cap < 4 || log_cap > 30 || !pairs) return hipErrorInvalidValue;
  const u64 tiles = ((u64)np + GTW_THREADS * GT_ROWS - 1) / (GTW_THREADS * GT_ROWS);
  u64 grid = (u64)num_cus * (u64)(wg_per_cu > 0 ? wg_per_cu : 8) * GT_THREADS / GTW_THREADS;
  if (grid > tiles) grid = tiles;
  if (grid < 1) grid = 1;
  if (extra)
    hipLaunchKernelGGL((gtable_emit_kernel<true, false, true>), dim3((u32)grid), dim3(GTW_THREADS), 0, st, static_cast<const Tup*>(S), np,
                       static_cast<const Tup*>(tab), log_cap, 0ull, 0, accum, static_cast<Tup*>(pairs));
  else
    hipLaunchKernelGGL((gtable_emit_kernel<false, false, true>), dim3((u32)grid), dim3(GTW_THREADS), 0, st, static_cast<const Tup*>(S), np,
                       static_cast<const Tup*>(tab), log_cap, 0ull, 0, accum, static_cast<Tup*>(pairs));
  return hipGetLastError();
}

hipError_t launch_gtable_emit(const void* S, u32 np, const void* tab, int log_cap, u64 svmin, int range_bits, u64* accum,
                              void* pairs, bool extra, bool wide, int num_cus, int wg_per_cu, hipStream_t st) {
  if (log_cap < 4 || log_cap > 30 || range_bits < 0 || range_bits > 64 || !pairs) return hipErrorInvalidValue;
  const u64 tiles = ((u64)np + GTW_THREADS * GT_ROWS - 1) / (GTW_THREADS * GT_ROWS);
  u64 grid = (u64)num_cus * (u64)(wg_per_cu > 0 ? wg_per_cu : 8) * GT_THREADS / GTW_THREADS;  // (wg_per_cu counts 256-thread workgroups)
  if (grid > tiles) grid = tiles;
  if (grid < 1) grid = 1;
#define HMJ_GTE(E, W)                                                                                                  \
  hipLaunchKernelGGL((gtable_emit_kernel<E, W>), dim3((u32)grid), dim3(GTW_THREADS), 0, st, static_cast<const Tup*>(S), np, \
                     static_cast<const Tup*>(tab), log_cap, svmin, range_bits, accum, static_cast<Tup*>(pairs))
  if (extra) {
    if (wide) HMJ_GTE(true, true); else HMJ_GTE(true, false);
  } else {
    if (wide) HMJ_GTE(false, true); else HMJ_GTE(false, false);
  }
#undef HMJ_GTE
  return hipGetLastError();
}

hipError_t launch_gtable_swap(const void* in, void* out, u64 n, int num_cus, hipStream_t st) {
  u64 grid = (n + 255) / 256;
  if (grid > (u64)num_cus * 16) grid = (u64)num_cus * 16;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(gtable_swap_kernel, dim3((u32)grid), dim3(256), 0, st, static_cast<const Tup*>(in), static_cast<Tup*>(out), n);
  return hipGetLastError();
}

hipError_t launch_gtable_expand(const void* pairs, u64 n, const void* sortedR, u64 svmin, int range_bits, u64* out_key,
                                u64* out_rval, u64* out_sval, u64* accum, bool extra, bool wide, int num_cus, hipStream_t st) {
  if (!pairs || !sortedR || !out_key || !out_rval || !out_sval || range_bits < 0 || range_bits > 64) return hipErrorInvalidValue;
  u64 grid = (n + 255) / 256;
  if (grid > (u64)num_cus * 16) grid = (u64)num_cus * 16;
  if (grid < 1) grid = 1;
#define HMJ_GTX(E, W)                                                                                            \
  hipLaunchKernelGGL((gtable_expand_kernel<E, W>), dim3((u32)grid), dim3(256), 0, st, static_cast<const Tup*>(pairs), n, \
                     static_cast<const Tup*>(sortedR), svmin, range_bits, out_key, out_rval, out_sval, accum)
  if (extra) {
    if (wide) HMJ_GTX(true, true); else HMJ_GTX(true, false);
  } else {
    if (wide) HMJ_GTX(false, true); else HMJ_GTX(false, false);
  }
#undef HMJ_GTX
  return hipGetLastError();
}

}  // namespace hmj
