// One LSD radix pass over 16-byte {key,val} rows for gfx950: histogram -> row scan -> stable
// scatter.  It restates pass 1 of the reference's parallel radix sort for the GPU:
//   * a "worker" is a workgroup owning a contiguous row range, as a reference thread owns
//     rows [t*N/T, (t+1)*N/T)                         (radix_sort.h:476-488, radix_hash.h:375-388);
//   * per-worker digit histogram                      (radix_sort.h:418-421, radix_hash.h:313-316);
//   * exclusive scan digit-major / worker-minor       (radix_sort.h:424-437, radix_hash.h:319-333);
//   * stable scatter through per-(worker,digit) cursors (radix_sort.h:444-449, radix_hash.h:339-345).
// The result of a pass is therefore bit-identical to oracle orc_stable_partition() for any
// worker count.  HBM-bound integer work: coalesced dwordx4 loads, per-wave ballot ranking,
// LDS-staged tiles so each digit's rows leave as one contiguous run (write-combined scatter).
#include <cstdlib>
#include <cstring>

#include <type_traits>

#include "hmj_dev.h"
#include "hmj_launch.h"

namespace hmj {

// What a pass splits on.  OWN == 0: a radix digit of the key, (key >> shift) & mask -- every pass of the
// local join.  The multi-GPU owner split (exchange.hip) uses the same stable histogram / scan / scatter with
//   OWN == 1: owner = floor(mix64(key) * G / 2^64): spreads ANY key set evenly over G ranks (dense integer
//             keys share their top bits and would all belong to rank 0);
//   OWN == 2: owner = number of splitters <= key (G - 1 ascending splitters agreed by all ranks): rank g
//             owns a key range, so per-rank ordered results concatenate in key order (HMJ_ORDERED).
template <int OWN>
__device__ __forceinline__ u32 owner_digit(u64 key, int shift, u32 mask, const OwnerFn& f) {
  if (OWN == 1) return (u32)__umul64hi(mix64(key), (u64)f.G);
  if (OWN == 2) {
    u32 d = 0;
#pragma unroll
    for (int i = 0; i < kMaxRanks - 1; i++) d += ((u32)i + 1 < f.G && key >= f.spl[i]) ? 1u : 0u;
    return d;
  }
  return (u32)(key >> shift) & mask;
}

// ---------------------------------------------------------------------------------------------
// K1: per-worker digit histogram.  hist[d * nblk + worker].  Algorithmic traffic: 16 B/row read
// (the key shares its 16-B row with the payload, so whole lines are fetched).
// ---------------------------------------------------------------------------------------------
template <int OWN>
__global__ __launch_bounds__(RP_THREADS) void radix_hist_kernel(const Tup* __restrict__ in, u32 n,
                                                                int shift, u32 mask,
                                                                u32 rows_per_block, u32 n_full,
                                                                u32* __restrict__ hist, u32 nblk, OwnerFn own) {
  __shared__ u32 h[RP_MAXD];
  const u32 D = mask + 1;
  for (u32 d = threadIdx.x; d < D; d += RP_THREADS) h[d] = 0;
  __syncthreads();
  // workers 0..k-1 own whole tiles of [0, n_full); one extra worker owns the ragged tail
  u64 begin = (u64)blockIdx.x * rows_per_block, end;
  if (begin < n_full) {
    end = begin + rows_per_block;
    if (end > n_full) end = n_full;
  } else {
    begin = n_full;
    end = n;
  }
  const u64* __restrict__ keys = reinterpret_cast<const u64*>(in);
  for (u64 base = begin; base < end; base += (u64)RP_THREADS * RP_ITEMS) {
    u64 k[RP_ITEMS];
#pragma unroll
    for (int r = 0; r < RP_ITEMS; r++) {
      u64 i = base + (u64)r * RP_THREADS + threadIdx.x;
      k[r] = (i < end) ? keys[2 * i] : 0;
    }
#pragma unroll
    for (int r = 0; r < RP_ITEMS; r++) {
      u64 i = base + (u64)r * RP_THREADS + threadIdx.x;
      if (i < end) atomicAdd(&h[owner_digit<OWN>(k[r], shift, mask, own)], 1u);
    }
  }
  __syncthreads();
  for (u32 d = threadIdx.x; d < D; d += RP_THREADS) hist[(u64)d * nblk + blockIdx.x] = h[d];
}

// ---------------------------------------------------------------------------------------------
// K2: one workgroup per digit: exclusive scan of that digit's worker counts (in place) + total.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void radix_rowscan_kernel(u32* __restrict__ hist, u32 nblk,
                                                            u32* __restrict__ totals) {
  __shared__ u32 scratch[8];
  u32* row = hist + (u64)blockIdx.x * nblk;
  u32 carry = 0;
  for (u32 base = 0; base < nblk; base += 256) {
    u32 i = base + threadIdx.x;
    u32 v = (i < nblk) ? row[i] : 0;
    u32 tot;
    u32 ex = block_excl_scan_u32<256>(v, scratch, &tot);
    if (i < nblk) row[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// ---------------------------------------------------------------------------------------------
// K3: stable scatter.  Algorithmic traffic: 16 B/row read + 16 B/row written.
// LDS: 64 KiB tile + 8 KiB per-wave digit counters + 6 KiB per-digit state  -> 2 workgroups/CU.
//
// Ranking (the hot part) is wave-synchronous and needs no ballots: every wave owns a table of
// one 64-bit LANE MASK per digit (aliased onto the tile buffer, which is idle while ranking).
// A lane ORs its lane bit into mask[digit] (ds_or_b64), reads the word back -- LDS executes one
// wave's instructions in order, so it now holds exactly the lanes of this round with the same
// digit -- and the lowest such lane adds the population to the per-wave digit counter and
// clears the mask.  rank = counter before + matching lanes below me: stable by construction.
// ---------------------------------------------------------------------------------------------
struct ScatterSmem {
  union {
    Tup stage[RP_TILE];             // rows of the tile, grouped by digit
    u64 lanemask[RP_WAVES][RP_MAXD];  // ranking scratch (first 32 KiB), all-zero between uses
  };
  u16 wcnt[RP_WAVES][RP_MAXD];  // per-wave digit counts, then exclusive prefix across waves
  u32 tile_off[RP_MAXD];        // start of digit d inside the staged tile
  u32 delta[RP_MAXD];           // global cursor of d minus tile_off[d]
  u32 cursor[RP_MAXD];          // next output row of digit d for this worker
  u32 scratch[RP_WAVES + 1];
};
static_assert(sizeof(u64) * RP_WAVES * RP_MAXD <= sizeof(Tup) * RP_TILE / 2, "mask table must fit half the tile");

template <bool HI>  // HI: the digit lies entirely in the key's high 32 bits (shift >= 32)
__device__ __forceinline__ u32 digit_of(u64 key, int shift, u32 mask) {
  if (HI) return ((u32)(key >> 32) >> (shift - 32)) & mask;
  return (u32)(key >> shift) & mask;
}

// Loads this lane's 8 rows of the tile at `src`.  FULL tiles use unpredicated loads so the eight
// dwordx4 requests issue back to back.
template <bool FULL>
__device__ __forceinline__ void load_tile(Tup (&t)[RP_ITEMS], const Tup* __restrict__ src,
                                          u32 wbase, u32 tile_n) {
#pragma unroll
  for (int r = 0; r < RP_ITEMS; r++)
    if (FULL || wbase + r * 64 < tile_n) t[r] = load_stream(&src[wbase + r * 64]);
}

// FULL = true : workers that own only whole tiles (the bulk; no predication anywhere).
// FULL = false: the single tail worker (rows [n_full, n), fewer than one tile), launched apart so
//               its predicated code does not cost the bulk kernel registers.
template <bool HI, bool FULL, int OWN = 0>
__global__ __launch_bounds__(RP_THREADS, 4) void radix_scatter_kernel(
    const Tup* __restrict__ in, Tup* __restrict__ out, u32 n, u32 n_full, int shift, int bits,
    u32 rows_per_block, u32 worker_base, const u32* __restrict__ hist_scanned,
    const u32* __restrict__ totals, u32 nblk, u64* __restrict__ offsets_out, OwnerFn own) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  ScatterSmem& sm = *reinterpret_cast<ScatterSmem*>(smem_raw);
  const u32 D = 1u << bits, mask = D - 1;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 worker = worker_base + blockIdx.x;

  // digit bases = exclusive scan of the digit totals (D <= 512 == RP_THREADS: one per thread)
  {
    u32 t = ((u32)tid < D) ? totals[tid] : 0, tot;
    u32 base = block_excl_scan_u32<RP_THREADS>(t, sm.scratch, &tot);
    if ((u32)tid < D) {
      sm.cursor[tid] = base + hist_scanned[(u64)tid * nblk + worker];
      if (offsets_out && worker == 0) {
        offsets_out[tid] = base;
        if ((u32)tid == D - 1) offsets_out[D] = n;
      }
    }
    for (int i = tid; i < RP_WAVES * RP_MAXD / 2; i += RP_THREADS)
      reinterpret_cast<u32*>(&sm.wcnt[0][0])[i] = 0;
    for (int i = tid; i < RP_WAVES * RP_MAXD; i += RP_THREADS) (&sm.lanemask[0][0])[i] = 0;
  }
  __syncthreads();

  u64 begin, end;
  if (FULL) {
    begin = (u64)blockIdx.x * rows_per_block;
    end = begin + rows_per_block;
    if (end > n_full) end = n_full;
  } else {
    begin = n_full;
    end = n;
  }
  if (begin >= end) return;

  // relaxed workgroup-scope atomics: plain LDS instructions, never cached in registers, and --
  // unlike volatile accesses -- no s_waitcnt vmcnt(0) around them (the prefetch stays in flight)
  u16* wc = &sm.wcnt[w][0];
  u64* lm = &sm.lanemask[w][0];
  const u64 lanebit = 1ull << lane;
  const u32 wbase = (u32)w * (RP_ITEMS * 64) + lane;  // this lane's row in round 0

  // wave w owns rows [w*512, w*512+512) of a tile, 8 rounds of 64 consecutive rows: lane order
  // inside a round + round order == row order, which is what makes the rank stable.
  Tup t[RP_ITEMS];
  load_tile<FULL>(t, in + begin, wbase, (u32)(end - begin));

  for (u64 tile = begin; tile < end; tile += RP_TILE) {
    const u32 tile_n = FULL ? (u32)RP_TILE : (u32)(end - tile);
    u32 dr[RP_ITEMS];  // digit << 16 | rank inside this wave's 512 rows
#pragma unroll
    for (int r = 0; r < RP_ITEMS; r++) {
      u32 d = 0, rank = 0;
      if (FULL || wbase + r * 64 < tile_n) {
        d = OWN ? owner_digit<OWN>(t[r].key, shift, mask, own) : digit_of<HI>(t[r].key, shift, mask);
        __hip_atomic_fetch_or(&lm[d], lanebit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const u64 m = __hip_atomic_load(&lm[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const u32 old = __hip_atomic_load(&wc[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const u32 below = popc_below(m);
        if (below == 0) {
          __hip_atomic_store(&lm[d], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_store(&wc[d], (u16)(old + (u32)__popcll(m)), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        rank = old + below;
      }
      dr[r] = (d << 16) | rank;
    }
    lds_barrier();

    // per digit: exclusive prefix over the 8 waves, tile totals, tile offsets, global delta
    {
      u32 cnt = 0;
      if ((u32)tid < D) {
#pragma unroll
        for (int k = 0; k < RP_WAVES; k++) {
          u32 c = sm.wcnt[k][tid];
          sm.wcnt[k][tid] = (u16)cnt;
          cnt += c;
        }
      }
      u32 tot;
      u32 off = block_excl_scan_u32<RP_THREADS>(cnt, sm.scratch, &tot);
      if ((u32)tid < D) {
        sm.tile_off[tid] = off;
        u32 cur = sm.cursor[tid];
        sm.delta[tid] = cur - off;
        sm.cursor[tid] = cur + cnt;
      }
    }
    lds_barrier();

#pragma unroll
    for (int r = 0; r < RP_ITEMS; r++) {
      if (FULL || wbase + r * 64 < tile_n) {
        const u32 d = dr[r] >> 16;
        const u32 pos = sm.tile_off[d] + sm.wcnt[w][d] + (dr[r] & 0xFFFFu);
        sm.stage[pos] = t[r];
      }
    }
    // the rows now live in LDS: start fetching the next tile while this one is written out
    if (FULL && tile + RP_TILE < end) load_tile<true>(t, in + tile + RP_TILE, wbase, RP_TILE);
    lds_barrier();

    // copy out: consecutive lanes -> consecutive rows of one digit run -> contiguous 16-B stores;
    // the first half of the tile buffer is handed back to the ranking masks as zeros
#pragma unroll
    for (int h = 0; h < 2; h++) {
      Tup v[RP_ITEMS / 2];
#pragma unroll
      for (int r = 0; r < RP_ITEMS / 2; r++) {
        const u32 i = (h * (RP_ITEMS / 2) + r) * RP_THREADS + tid;
        if (FULL || i < tile_n) v[r] = sm.stage[i];
      }
      if (h == 0) {
#pragma unroll
        for (int r = 0; r < RP_ITEMS / 2; r++) {
          Tup z;
          z.key = 0;
          z.val = 0;
          sm.stage[r * RP_THREADS + tid] = z;
        }
      }
#pragma unroll
      for (int r = 0; r < RP_ITEMS / 2; r++) {
        const u32 i = (h * (RP_ITEMS / 2) + r) * RP_THREADS + tid;
        if (FULL || i < tile_n) {
          const u32 d = OWN ? owner_digit<OWN>(v[r].key, shift, mask, own) : digit_of<HI>(v[r].key, shift, mask);
          store_stream(&out[sm.delta[d] + i], v[r]);
        }
      }
    }
    for (int i = tid; i < RP_WAVES * RP_MAXD / 2; i += RP_THREADS)
      reinterpret_cast<u32*>(&sm.wcnt[0][0])[i] = 0;
    lds_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// K3': stable scatter with software write-combining.  Same ranking and staging as K3, but the
// output of every digit is only ever stored in WHOLE 128-byte lines (8 rows): the rows of a
// digit's trailing partial line wait in an LDS carry buffer (carry[digit][8]) until a later tile
// of the same worker completes the line.  Measured motivation (tools/exp_align.py, 2^28 rows,
// 9-bit fan-out): line-aligned runs 1.70 ms (5.1 TB/s) vs unaligned runs 2.90 ms (2.9 TB/s) --
// partial lines, not run length, are what the plain scatter pays for.
//   WC_X: 512 threads, 2048-row tile, fan-out <= 256  -> 74 KiB LDS, 2 workgroups/CU
//   WC_Y: 1024 threads, 4096-row tile, fan-out <= 512 -> 156 KiB LDS, 1 workgroup/CU
// Worker-boundary lines (first/last line of a worker's digit segment) are still partial:
// 2 per (worker, digit).
// ---------------------------------------------------------------------------------------------
constexpr int WC_ITEMS = 4;
constexpr int WC_LINE = 8;  // rows per 128-byte line

// Developer builds (-DHMJ_STAMPS): shader-clock stamps around the phases of wc_tile, summed over all waves of
// a launch into g_wc_stamps[phase] (read and reset by hmj_dev_stamps, api.hip).  Phases: 0 wait for the tile's
// rows, 1 rank, 2 barrier, 3 plan, 4 barrier, 5 stage + prefetch issue, 6 barrier, 7 copy out + keep, 8 barrier,
// 9 carry + clear, 10 barrier, 11 = tiles.  The release library compiles none of this.
#ifndef HMJ_WC_PLAN1
#define HMJ_WC_PLAN1 0   // 1: one wave plans all digits with wave scans -- measured 40 % SLOWER (the other seven waves
                         // wait 6000 cycles at the next barrier); 0: one thread per digit + block scan
#endif
#ifndef HMJ_WC_FULLTILE
#define HMJ_WC_FULLTILE 1  // slab kernels: whole tiles run the predicate-free instantiation of wc_tile
#endif
#ifndef HMJ_WC_NOBAR5
#define HMJ_WC_NOBAR5 1  // 1: waves zero their own mask slab / counters, no barrier at the end of a tile
#endif
#ifdef HMJ_STAMPS
__device__ unsigned long long g_wc_stamps[32];  // [0..15] pass A / exact scatter, [16..31] slab pass B
struct WcStamps {
  u32 acc[14];
  u64 last;
  int base;
  __device__ __forceinline__ void begin() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = __builtin_amdgcn_s_memtime();
  }
  __device__ __forceinline__ void mark(int i) {
    const u64 now = __builtin_amdgcn_s_memtime();
    acc[i] += (u32)(now - last);
    last = now;
  }
  __device__ __forceinline__ void flush() {
    if ((threadIdx.x & 63) == 0)
      for (int i = 0; i < 14; i++) atomicAdd(&g_wc_stamps[base + i], (unsigned long long)acc[i]);
  }
};
#define WC_STAMP_ARG , WcStamps& stamps
#define WC_STAMP_PASS , stamps
#define WC_MARK(i) stamps.mark(i)
#define WC_VMWAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define WC_STAMP_ARG
#define WC_STAMP_PASS
#define WC_MARK(i)
#define WC_VMWAIT()
#endif

template <int THREADS, int MAXD>
struct WcSmem {
  static constexpr int TILE = THREADS * WC_ITEMS;
  static constexpr int WAVES = THREADS / kWave;
  static constexpr int ZROWS = WAVES * MAXD * 8 / 16;  // tile rows aliased by the lane masks
  static constexpr int MAXLINES = TILE / WC_LINE + 2 * MAXD;
  static_assert(ZROWS <= TILE && ZROWS % THREADS == 0, "lane masks must alias whole row slabs");
  static_assert(MAXD <= THREADS && MAXD * WC_LINE == TILE, "one digit per thread; carry == tile slots");
  union {
    Tup stage[TILE];
    u64 lanemask[WAVES][MAXD];
  };
  Tup carry[MAXD][WC_LINE];
  u16 wcnt[WAVES][MAXD];
  u32 tile_off[MAXD];  // start of digit d's new rows inside the staged tile
  u32 cflush[MAXD];    // global row where digit d's next unflushed row belongs
  // cflush at the start of this tile.  The 512-digit shape has no LDS to spare (stage 64 + carry 64 + counters 16 KiB
  // of the CU's 160): there the copy-out recomputes it as cflush[d] - (rows flushed this tile)
  static constexpr bool HAS_CF_TILE = MAXD <= 256;
  u32 cf_tile[HAS_CF_TILE ? MAXD : 1];
  u32 plan[MAXD];      // rows to flush this tile << 8 | rows pending from earlier tiles
  u32 pend[MAXD];      // rows waiting in carry[d]
  u32 line_off[MAXD];  // first entry of digit d in line_tab
  u16 line_tab[MAXLINES];  // digit of every 128-byte line this tile completes
  u32 total_lines;
  u32 scratch[WAVES + 1];
};

// One tile of the write-combining scatter: rank -> plan -> stage -> (prefetch) -> copy out whole
// lines -> keep the tails.  Shared by the exact kernel (global offsets from the histogram) and the
// slab kernels (private over-allocated output regions, no histogram).
//   ALLVALID : every row slot of the tile holds a row (no per-row predicate)
//   LIMIT    : digit d may only write below limit[d] (an LDS array); a digit that would cross it sets *ovf and
//              drops its rows (the caller discards the whole result and re-runs the exact path)
//              (`limit` is a callable d -> first row past digit d's slab: the slab kernels compute it from the slab
//              geometry in registers; the plan phase asks once per digit and tile)
struct NoLimit {
  __device__ __forceinline__ u32 operator()(u32) const { return 0xFFFFFFFFu; }
};
template <int THREADS, int MAXD, bool HI, bool ALLVALID, bool LIMIT, typename Prefetch, typename Limit>
__device__ __forceinline__ void wc_tile(WcSmem<THREADS, MAXD>& sm, Tup (&t)[WC_ITEMS], u32 tile_n,
                                        Tup* __restrict__ out, int shift, u32 mask, u32 D,
                                        Prefetch&& prefetch, const Limit& limit, bool* ovf WC_STAMP_ARG) {
  typedef WcSmem<THREADS, MAXD> Smem;
  constexpr int WAVES = Smem::WAVES;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  u16* wc = &sm.wcnt[w][0];
  u64* lm = &sm.lanemask[w][0];
  const u64 lanebit = 1ull << lane;
  const u32 wbase = (u32)w * (WC_ITEMS * 64) + lane;

  // relaxed workgroup-scope atomics: plain LDS instructions, never cached in registers, and --
  // unlike volatile accesses -- no s_waitcnt vmcnt(0) around them (the prefetch stays in flight)
  u32 dr[WC_ITEMS];
#ifdef HMJ_STAMPS
  stamps.last = __builtin_amdgcn_s_memtime();
  WC_VMWAIT();
  WC_MARK(0);
  stamps.acc[11]++;
#endif
#pragma unroll
  for (int r = 0; r < WC_ITEMS; r++) {
    u32 d = 0, rank = 0;
    if (ALLVALID || wbase + r * 64 < tile_n) {
      d = digit_of<HI>(t[r].key, shift, mask);
      __hip_atomic_fetch_or(&lm[d], lanebit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const u64 m = __hip_atomic_load(&lm[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const u32 old = __hip_atomic_load(&wc[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const u32 below = popc_below(m);
      if (below == 0) {
        __hip_atomic_store(&lm[d], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&wc[d], (u16)(old + (u32)__popcll(m)), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      rank = old + below;
    }
    dr[r] = (d << 16) | rank;
  }
  WC_MARK(1);
  lds_barrier();
  WC_MARK(2);

  // per digit: prefix over waves, tile offsets, this tile's flush plan and its 128-byte lines
#if HMJ_WC_PLAN1
  // ONE wave plans all digits (lane l: digits l, l + 64, ...): the scan over digits is then a wave scan per
  // 64 digits with a running carry -- no block scan, no barriers inside the phase (it took 12 % of a tile with
  // half the threads idle and two extra barriers; the other waves go straight to the next barrier).
  if (w == 0) {
    constexpr int DPL = MAXD / kWave;  // digits per lane
    u32 cnt[DPL], nl[DPL], pd[DPL], cf[DPL], fl[DPL];
#pragma unroll
    for (int j = 0; j < DPL; j++) {
      const u32 d = (u32)j * kWave + lane;
      cnt[j] = nl[j] = pd[j] = cf[j] = fl[j] = 0;
      if (d < D) {
        u32 c[WAVES];
#pragma unroll
        for (int k = 0; k < WAVES; k++) c[k] = sm.wcnt[k][d];
        u32 run = 0;
#pragma unroll
        for (int k = 0; k < WAVES; k++) {
          sm.wcnt[k][d] = (u16)run;
          run += c[k];
        }
        cnt[j] = run;
        pd[j] = sm.pend[d];
        cf[j] = sm.cflush[d];
        u32 avail = pd[j] + run;
        if (LIMIT && cf[j] + avail > limit(d)) {  // slab full: give up on this digit (and the run)
          *ovf = true;
          avail = 0;
          pd[j] = 0;
        }
        const u32 tail = (cf[j] + avail) & (WC_LINE - 1);  // rows past the last whole line
        fl[j] = (tail >= avail) ? 0u : avail - tail;
        nl[j] = fl[j] ? (((cf[j] & (WC_LINE - 1)) + fl[j] + WC_LINE - 1) >> 3) : 0u;
        sm.pend[d] = avail - fl[j];
        sm.cflush[d] = cf[j] + fl[j];
      }
    }
    u32 running = 0;  // lines << 16 | rows of the digits scanned so far
#pragma unroll
    for (int j = 0; j < DPL; j++) {
      const u32 d = (u32)j * kWave + lane;
      const u32 v = (nl[j] << 16) | cnt[j];
      const u32 incl = wave_incl_scan_u32(v, lane);
      const u32 sc = running + incl - v;
      running += (u32)__builtin_amdgcn_readlane((int)incl, 63);
      if (d < D) {
        const u32 loff = sc >> 16;
        sm.tile_off[d] = sc & 0xFFFFu;
        if constexpr (Smem::HAS_CF_TILE) sm.cf_tile[d] = cf[j];
        sm.plan[d] = (fl[j] << 8) | pd[j];
        sm.line_off[d] = loff;
        for (u32 l = 0; l < nl[j]; l++) sm.line_tab[loff + l] = (u16)d;
      }
    }
    if (lane == 0) sm.total_lines = running >> 16;
  }
#else
  {
    u32 cnt = 0, nl = 0, pd = 0, cf = 0, fl = 0;
    if ((u32)tid < D) {
#pragma unroll
      for (int k = 0; k < WAVES; k++) {
        u32 c = sm.wcnt[k][tid];
        sm.wcnt[k][tid] = (u16)cnt;
        cnt += c;
      }
      pd = sm.pend[tid];
      cf = sm.cflush[tid];
      u32 avail = pd + cnt;
      if (LIMIT && cf + avail > limit((u32)tid)) {  // slab full: give up on this digit (and the run)
        *ovf = true;
        avail = 0;
        pd = 0;
      }
      const u32 tail = (cf + avail) & (WC_LINE - 1);  // rows past the last whole line
      fl = (tail >= avail) ? 0u : avail - tail;
      nl = fl ? (((cf & (WC_LINE - 1)) + fl + WC_LINE - 1) >> 3) : 0u;
      sm.pend[tid] = avail - fl;
      sm.cflush[tid] = cf + fl;
    }
    u32 tot;
    // (no trailing barrier: scratch is next written a whole tile -- four barriers -- later)
    const u32 sc = block_excl_scan_u32<THREADS, false>((nl << 16) | cnt, sm.scratch, &tot);
    if ((u32)tid < D) {
      const u32 loff = sc >> 16;
      sm.tile_off[tid] = sc & 0xFFFFu;
      if constexpr (Smem::HAS_CF_TILE) sm.cf_tile[tid] = cf;
      sm.plan[tid] = (fl << 8) | pd;
      sm.line_off[tid] = loff;
      for (u32 l = 0; l < nl; l++) sm.line_tab[loff + l] = (u16)tid;
    }
    if (tid == 0) sm.total_lines = tot >> 16;
  }
#endif
  WC_MARK(3);
  lds_barrier();
  WC_MARK(4);

#pragma unroll
  for (int r = 0; r < WC_ITEMS; r++) {
    if (ALLVALID || wbase + r * 64 < tile_n) {
      const u32 d = dr[r] >> 16;
      sm.stage[sm.tile_off[d] + sm.wcnt[w][d] + (dr[r] & 0xFFFFu)] = t[r];
    }
  }
  WC_MARK(5);
  prefetch();  // the rows now live in LDS: start fetching the next tile while this one leaves
  WC_MARK(12);
  lds_barrier();
  WC_MARK(6);

  // copy out WHOLE LINES: 8 consecutive lanes own one 128-byte line (carried rows first, then
  // this tile's rows), so every line leaves in a single store instruction
  {
    const u32 TL = sm.total_lines, k = (u32)tid & (WC_LINE - 1);
    for (u32 L = (u32)tid >> 3; L < TL; L += THREADS / WC_LINE) {
      const u32 d = sm.line_tab[L];
      const u32 pl = sm.plan[d], fl = pl >> 8, pd = pl & 0xFFu;
      u32 cfd;
      if constexpr (Smem::HAS_CF_TILE) cfd = sm.cf_tile[d];
      else cfd = sm.cflush[d] - fl;  // (the plan phase advanced cflush by the rows this tile flushes)
      const u32 grow = (cfd & ~(u32)(WC_LINE - 1)) + (L - sm.line_off[d]) * WC_LINE + k;
      const u32 j = grow - cfd;  // wraps (huge) for the slots before an unaligned segment start
      if (j < fl) store_stream(&out[grow], (j < pd) ? sm.carry[d][j] : sm.stage[sm.tile_off[d] + j - pd]);
    }
  }
  // rows that stay behind (each digit's new partial line): stage -> registers
  u64 keep_k[WC_ITEMS], keep_v[WC_ITEMS];
  u32 has = 0;
#pragma unroll
  for (int q = 0; q < WC_ITEMS; q++) {
    const u32 s = q * THREADS + tid, d = s >> 3, k = s & (WC_LINE - 1);
    u32 idx = 0;
    if (d < D && k < sm.pend[d]) {
      const u32 pl = sm.plan[d], fl = pl >> 8, pd = pl & 0xFFu;
      if (fl != 0 || k >= pd) {
        idx = sm.tile_off[d] + (fl ? fl - pd + k : k - pd);
        has |= 1u << q;
      }
    }
    const Tup kv = sm.stage[idx];  // unconditional read (row 0 when nothing stays)
    keep_k[q] = kv.key;
    keep_v[q] = kv.val;
  }
  WC_MARK(7);
  lds_barrier();
  WC_MARK(8);
#pragma unroll
  for (int q = 0; q < WC_ITEMS; q++) {
    const u32 s = q * THREADS + tid;
    if (has & (1u << q)) {
      Tup kv;
      kv.key = keep_k[q];
      kv.val = keep_v[q];
      sm.carry[s >> 3][s & (WC_LINE - 1)] = kv;
    }
  }
#if HMJ_WC_NOBAR5
  // every wave hands its OWN lane-mask slab (aliased on the tile buffer) and its own digit counters back as
  // zeros: the next tile's ranking touches only those, so no barrier is needed between here and there.
  // (The carry rows written above are next read after three more barriers, or behind wc_flush_carry's.)
  {
    constexpr int ZR = MAXD * 8 / 16;  // tile rows aliased by one wave's lane masks
    Tup z;
    z.key = 0;
    z.val = 0;
#pragma unroll
    for (int r = 0; r < ZR / kWave; r++) sm.stage[w * ZR + r * kWave + lane] = z;
    u32* wz = reinterpret_cast<u32*>(&sm.wcnt[w][0]);
#pragma unroll
    for (int r = 0; r < MAXD / 2 / kWave; r++) wz[r * kWave + lane] = 0;
  }
  WC_MARK(9);
  WC_MARK(10);
#else
#pragma unroll
  for (int r = 0; r < WC_ITEMS; r++) {
    if (r * THREADS < Smem::ZROWS) {  // hand the slab back to the lane masks as zeros
      Tup z;
      z.key = 0;
      z.val = 0;
      sm.stage[r * THREADS + tid] = z;
    }
  }
  for (int i = tid; i < WAVES * MAXD / 2; i += THREADS) reinterpret_cast<u32*>(&sm.wcnt[0][0])[i] = 0;
  WC_MARK(9);
  lds_barrier();
  WC_MARK(10);
#endif
}

// worker done: whatever still waits in the carry buffers ends this worker's digit segments
template <int THREADS, int MAXD>
__device__ __forceinline__ void wc_flush_carry(WcSmem<THREADS, MAXD>& sm, Tup* __restrict__ out, u32 D) {
  const int tid = threadIdx.x;
  lds_barrier();  // the last tile's carry rows are complete
#pragma unroll
  for (int q = 0; q < WC_ITEMS; q++) {
    const u32 s = q * THREADS + tid, d = s >> 3, k = s & (WC_LINE - 1);
    if (d < D && k < sm.pend[d]) store_stream(&out[sm.cflush[d] + k], sm.carry[d][k]);
  }
}

template <int THREADS, int MAXD>
__device__ __forceinline__ void wc_clear(WcSmem<THREADS, MAXD>& sm) {
  constexpr int WAVES = WcSmem<THREADS, MAXD>::WAVES;
  const int tid = threadIdx.x;
  for (int i = tid; i < WAVES * MAXD / 2; i += THREADS) reinterpret_cast<u32*>(&sm.wcnt[0][0])[i] = 0;
  for (int i = tid; i < WAVES * MAXD; i += THREADS) (&sm.lanemask[0][0])[i] = 0;
}

template <int THREADS, int MAXD, bool HI, bool FULL>
__global__ __launch_bounds__(THREADS, (THREADS * (MAXD <= 256 ? 2 : 1)) / 256) void radix_scatter_wc_kernel(
    const Tup* __restrict__ in, Tup* __restrict__ out, u32 n, u32 n_full, int shift, int bits,
    u32 rows_per_block, u32 worker_base, const u32* __restrict__ hist_scanned,
    const u32* __restrict__ totals, u32 nblk, u64* __restrict__ offsets_out) {
  typedef WcSmem<THREADS, MAXD> Smem;
  constexpr int TILE = Smem::TILE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem& sm = *reinterpret_cast<Smem*>(smem_raw);
  const u32 D = 1u << bits, mask = D - 1;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 worker = worker_base + blockIdx.x;

  {
    u32 t = ((u32)tid < D) ? totals[tid] : 0, tot;
    u32 base = block_excl_scan_u32<THREADS>(t, sm.scratch, &tot);
    if ((u32)tid < D) {
      sm.cflush[tid] = base + hist_scanned[(u64)tid * nblk + worker];
      sm.pend[tid] = 0;
      if (offsets_out && worker == 0) {
        offsets_out[tid] = base;
        if ((u32)tid == D - 1) offsets_out[D] = n;
      }
    }
    wc_clear(sm);
  }
  __syncthreads();

  u64 begin, end;
  if (FULL) {
    begin = (u64)blockIdx.x * rows_per_block;
    end = begin + rows_per_block;
    if (end > n_full) end = n_full;
  } else {
    begin = n_full;
    end = n;
  }
  if (begin >= end) return;
  const u32 wbase = (u32)w * (WC_ITEMS * 64) + lane;

  Tup t[WC_ITEMS];
#pragma unroll
  for (int r = 0; r < WC_ITEMS; r++)
    if (FULL || begin + wbase + r * 64 < end) t[r] = load_stream(&in[begin + wbase + r * 64]);

  bool ovf = false;
#ifdef HMJ_STAMPS
  WcStamps stamps = {};
#endif
  for (u64 tile = begin; tile < end; tile += TILE) {
    const u32 tile_n = FULL ? (u32)TILE : (u32)(end - tile);
    wc_tile<THREADS, MAXD, HI, FULL, false>(
        sm, t, tile_n, out, shift, mask, D,
        [&]() {
          if (FULL && tile + TILE < end) {
            const Tup* src = in + tile + TILE;
#pragma unroll
            for (int r = 0; r < WC_ITEMS; r++) t[r] = load_stream(&src[wbase + r * 64]);
          }
        },
        NoLimit(), &ovf WC_STAMP_PASS);
  }
  wc_flush_carry(sm, out, D);
#ifdef HMJ_STAMPS
  stamps.flush();
#endif
}

// ---------------------------------------------------------------------------------------------
// Slab kernels: the same write-combining scatter WITHOUT a histogram.  Every worker owns a
// private, over-allocated output region ("slab") per digit, so output positions need no global
// prefix sums and each pass reads its input exactly once (16 B read + 16 B written per row instead
// of 16 + 16 + 16).  Slabs hold mean + 8 sigma rows for uniformly distributed digits; a worker whose
// digit outgrows its slab raises ERR_SLAB and the caller re-runs the exact (histogram) path.
//
//   pass A  worker wA = contiguous input rows; writes slabA[dA][wA][CA], cntA[dA][wA]
//   pass B  worker (dA, k) = the A-slabs [k*WA/KB, (k+1)*WA/KB) of bucket dA, gathered in order
//           (so the pass stays stable); writes slabB[p][k][CB], cntB[p][k] with p = dB << bitsA | dA
//   probe   partition p = its KB pieces in order (probe.hip, probe_count_slab_kernel)
// ---------------------------------------------------------------------------------------------
// (Xform: what happens to a tile's rows between the load and the scatter.  NoXform: nothing -- the join's pass A;
//  RankXform: every probe row becomes {rank of its key, payload} through the global table, radix_slab_a_rank_kernel below.)
struct NoXform {
  static constexpr bool kStrided = false;
};
template <int THREADS, int MAXD, bool HI, typename Xform>
__device__ __forceinline__ void slab_a_body(const Tup* __restrict__ in, u32 n, int shift, int bits, u32 rows_per_worker,
                                            Tup* __restrict__ slab, u32 CA, u32 WA, u32* __restrict__ cnt_out,
                                            u64* __restrict__ accum, u32 worker_base, Xform& xf) {
  typedef WcSmem<THREADS, MAXD> Smem;
  constexpr int TILE = Smem::TILE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem& sm = *reinterpret_cast<Smem*>(smem_raw);
  const u32 D = 1u << bits, mask = D - 1;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 worker = worker_base + blockIdx.x;  // (a launch may cover a range of workers: rows that have arrived)
  const auto limit = [=](u32 d) { return (d * WA + worker) * CA + CA; };  // first row past this worker's slab of digit d
  if ((u32)tid < D) {
    sm.cflush[tid] = ((u32)tid * WA + worker) * CA;
    sm.pend[tid] = 0;
  }
  wc_clear(sm);
  __syncthreads();
  const u64 begin = (u64)worker * rows_per_worker;
  u64 end = begin + rows_per_worker;
  if (end > n) end = n;
  const u32 wbase = (u32)w * (WC_ITEMS * 64) + lane;
  bool ovf = false;
#ifdef HMJ_STAMPS
  WcStamps stamps = {};
#endif
  if constexpr (Xform::kStrided) {
    // STRIDED chunks: a tile is made of chunks of RANK_PASS_CHUNK_ROWS rows, and a worker's chunks are c, c + WA, c + 2 WA, ... --
    // places all over the relation instead of one contiguous range.  Digits that are correlated with a row's POSITION (the
    // piece of a run a payload falls into, when payloads are row ids or timestamps) then spread evenly over a worker's
    // slabs instead of filling one of them.  The price: a (digit, worker) slab is no longer a contiguous range of the
    // input in order, so the pass is not stable -- used only where equal keys are equal rows (radix_slab_a_rank_kernel).
    // A tile's valid rows are still a prefix of it: chunk numbers grow with the wave, and only the last chunk is ragged.
    constexpr u32 CH = RANK_PASS_CHUNK_ROWS, IPC = CH / 64, CPT = TILE / CH;  // items of a thread per chunk; chunks per tile
    static_assert(WC_ITEMS % IPC == 0 && IPC >= 1, "a wave's rows of a tile are whole chunks");
    const u64 n_chunks = ((u64)n + CH - 1) / CH, whole_chunks = n / CH;
    const u32 tail = n % CH;
    // Which chunk of a stride a worker takes is scrambled (a multiplier coprime to WA, near WA / golden ratio): pass B's
    // worker k gathers the slabs of a RANGE of workers, and with chunk = worker a piece shorter than a stride would sit in
    // the slabs of one range only.
    const u32 slot = (u32)(((u64)worker * xf.slot_mult) % WA);
    const u64 my_whole = whole_chunks > slot ? (whole_chunks - slot + WA - 1) / WA : 0;  // this worker's whole chunks
    auto chunk_of = [&](u64 k, u32 c) { return (k * CPT + c) * WA + slot; };  // chunk c of the worker's tile k
    auto rows_of = [&](u64 k) {
      const u64 first = k * CPT;
      const u32 nv = my_whole > first ? (u32)((my_whole - first < CPT) ? my_whole - first : CPT) : 0u;
      return nv * CH + ((nv < CPT && tail && chunk_of(k, nv) == whole_chunks) ? tail : 0u);
    };
    if (chunk_of(0, 0) < n_chunks) {
      Tup t[WC_ITEMS];
      auto load_tile = [&](u64 k) {
#pragma unroll
        for (int r = 0; r < WC_ITEMS; r++) {
          const u64 q = chunk_of(k, (u32)w * (WC_ITEMS / IPC) + r / IPC) * CH + (r % IPC) * 64 + lane;
          t[r] = load_stream(&in[q < n ? q : n - 1]);  // clamped, unpredicated: loads issue back to back
        }
      };
      load_tile(0);
      for (u64 k = 0; chunk_of(k, 0) < n_chunks; k++) {
        const u32 tile_n = rows_of(k);
        auto prefetch = [&]() {
          if (chunk_of(k + 1, 0) < n_chunks) load_tile(k + 1);
        };
        xf(t, tile_n, wbase);
        if (HMJ_WC_FULLTILE && tile_n == (u32)TILE)
          wc_tile<THREADS, MAXD, HI, true, true>(sm, t, tile_n, slab, shift, mask, D, prefetch, limit, &ovf WC_STAMP_PASS);
        else
          wc_tile<THREADS, MAXD, HI, false, true>(sm, t, tile_n, slab, shift, mask, D, prefetch, limit, &ovf WC_STAMP_PASS);
      }
      wc_flush_carry(sm, slab, D);
    }
  } else if (begin < end) {
    Tup t[WC_ITEMS];
    {
      const u32 tn = (u32)((end - begin < TILE) ? end - begin : TILE);
#pragma unroll
      for (int r = 0; r < WC_ITEMS; r++) {
        const u32 q = wbase + r * 64;
        t[r] = load_stream(&in[begin + (q < tn ? q : tn - 1)]);  // clamped, unpredicated: loads issue back to back
      }
    }
    for (u64 tile = begin; tile < end; tile += TILE) {
      const u32 tile_n = (u32)((end - tile < TILE) ? end - tile : TILE);
      auto prefetch = [&]() {
        if (tile + TILE < end) {
          const u64 nt = tile + TILE;
          const u32 tn = (u32)((end - nt < TILE) ? end - nt : TILE);
          const Tup* src = in + nt;
#pragma unroll
          for (int r = 0; r < WC_ITEMS; r++) {
            const u32 q = wbase + r * 64;
            t[r] = load_stream(&src[q < tn ? q : tn - 1]);
          }
        }
      };
      if constexpr (!std::is_same<Xform, NoXform>::value) xf(t, tile_n, wbase);
      // whole tiles (all but a worker's last) run the predicate-free instantiation
      if (HMJ_WC_FULLTILE && tile_n == (u32)TILE)
        wc_tile<THREADS, MAXD, HI, true, true>(sm, t, tile_n, slab, shift, mask, D, prefetch, limit, &ovf WC_STAMP_PASS);
      else
        wc_tile<THREADS, MAXD, HI, false, true>(sm, t, tile_n, slab, shift, mask, D, prefetch, limit, &ovf WC_STAMP_PASS);
    }
    wc_flush_carry(sm, slab, D);
  }
  if ((u32)tid < D) cnt_out[(u32)tid * WA + worker] = sm.cflush[tid] + sm.pend[tid] - ((u32)tid * WA + worker) * CA;
  if (ovf) atomicOr(&accum[ACC_ERR], ERR_SLAB);
#ifdef HMJ_STAMPS
  stamps.flush();
#endif
}

#ifndef HMJ_B_ADDR
#define HMJ_B_ADDR 2  // 2: ONE wave-uniform lookup per wave and tile (64-entry window, ballot); 1: per-round slab
                      // lookup in lanes 0..3, planned one tile ahead; 0: per-row walk (round 1)
#endif
#ifndef HMJ_B_NT
#define HMJ_B_NT 0    // nontemporal loads of the A-slab rows
#endif
#ifdef HMJ_DEV
// developer builds: timing-only ablations of pass B's gather, switched between joins of one process (the same
// buffers, so the placement lottery cancels).  1 = read the bucket contiguously, 2 = 128-byte aligned gather.
// Results are wrong while it is set.
__device__ u32 g_b_ablate;
static bool g_b_addr_alt = false;  // launch the slab lookup that is NOT the default (HMJ_B_ADDR)
#endif
// (SLAB_KB, hmj_dev.h: pass-B workers per bucket == pieces per final partition of the full slab path)
constexpr int SLAB_MAXSEG = 512;  // A-slabs one pass-B worker gathers (WA / KB <= 512)

template <int THREADS, int MAXD, bool HI, int ADDR>
__global__ __launch_bounds__(THREADS, (THREADS * (MAXD <= 256 ? 2 : 1)) / 256) void radix_slab_b_kernel(
    const Tup* __restrict__ slab_a, const u32* __restrict__ cnt_a, u32 CA, u32 WA, int bits_a, int shift,
    int bits, Tup* __restrict__ slab_b, u32 CB, u32* __restrict__ cnt_b, u64* __restrict__ accum, u32 KB) {
  typedef WcSmem<THREADS, MAXD> Smem;
  constexpr int TILE = Smem::TILE;
  static_assert(SLAB_MAXSEG <= THREADS, "one A-slab count per thread in the prologue scan");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem& sm = *reinterpret_cast<Smem*>(smem_raw);
  u32* pre = reinterpret_cast<u32*>(smem_raw + sizeof(Smem));  // SLAB_MAXSEG + 1 prefix sums
  const u32 D = 1u << bits, mask = D - 1;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 dA = blockIdx.x / KB, k = blockIdx.x % KB;
  const auto limit = [=](u32 d) { return (((d << bits_a) | dA) * KB + k) * CB + CB; };  // first row past piece k of partition (d, dA)
  const u32 w0 = (u32)((u64)k * WA / KB), w1 = (u32)((u64)(k + 1) * WA / KB), ns = w1 - w0;
  {
    const u32 c = ((u32)tid < ns) ? cnt_a[dA * WA + w0 + tid] : 0;
    u32 tot;
    const u32 ex = block_excl_scan_u32<THREADS>(c, sm.scratch, &tot);
    if ((u32)tid < ns) pre[tid] = ex;
    if (tid == 0) pre[ns] = tot;
    if ((u32)tid < D) {
      sm.cflush[tid] = ((((u32)tid << bits_a) | dA) * KB + k) * CB;
      sm.pend[tid] = 0;
    }
    wc_clear(sm);
  }
  __syncthreads();
  const u32 total = pre[ns];
#ifdef HMJ_DEV
  const u32 abl = g_b_ablate;
#else
  constexpr u32 abl = 0;
#endif
  const u32 wbase = (u32)w * (WC_ITEMS * 64) + lane;
  (void)wbase;
  const Tup* __restrict__ bucket = slab_a + (u64)(dA * WA + w0) * CA;  // A-slab j of this worker: + j*CA
  bool ovf = false;
#ifdef HMJ_STAMPS
  WcStamps stamps = {};
  stamps.base = 16;
#endif
  if (total) {
    Tup t[WC_ITEMS];
    if constexpr (ADDR != 0) {
    // Row q of the worker's input (the concatenation of its A-slabs) lives in slab s = the last one with
    // pre[s] <= q, at bucket[s * CA + (q - pre[s])].  A wave reads 4 ROUNDS of 64 consecutive rows per tile;
    // lane r < 4 of every wave looks up round r's first slab by binary search ONE TILE AHEAD (nine dependent
    // LDS reads that overlap the copy-out of the current tile), together with the next three slab starts.
    // The 64 rows of a round then need no search: slabs average hundreds of rows, so a round crosses at most
    // two slab starts (else: the per-row walk below), and a lane finds its slab with two compares.
    u32 d_s0 = 0, d_b0 = 0, d_b1 = 0, d_b2 = 0, d_b3 = 0;  // descriptors of the tile to load next (lanes 0..3)
    const float slabs_per_row = (float)ns / (float)total;
    auto plan1 = [&](u32 tile_begin) {
      u32 q0 = tile_begin + (u32)w * (WC_ITEMS * 64) + (u32)(lane & 3) * 64;
      q0 = q0 < total ? q0 : total - 1;
      // evenly filled slabs: slab(q0) ~ q0 * ns / total, off by a slab or two (the prefix sums wander by
      // sqrt(#slabs) * sigma).  Look at the eight slab starts around the guess -- independent LDS reads, one
      // latency -- and fall back to the binary search (nine dependent reads) when they do not bracket q0.
      u32 g = (u32)((float)q0 * slabs_per_row);
      g = g < ns ? g : ns - 1;
      u32 lo = g >= 3 ? g - 3 : 0;
      u32 pw[8];
#pragma unroll
      for (int i = 0; i < 8; i++) pw[i] = pre[lo + i < ns ? lo + i : ns];
      u32 j = 0;
#pragma unroll
      for (int i = 1; i < 8; i++) j += pw[i] <= q0 ? 1u : 0u;
      const bool bracketed = pw[0] <= q0 && pw[7] > q0;  // (pre[ns] = total > q0)
      lo += j;
      if (__any(!bracketed)) {
        u32 hi = ns;  // pre[lo] <= q0, and hi == ns or pre[hi] > q0
        lo = 0;
#pragma unroll
        for (int it = 0; it < 9; it++) {  // ns <= SLAB_MAXSEG = 512
          const u32 mid = (lo + hi) >> 1;
          const bool le = pre[mid] <= q0, act = hi - lo > 1;
          lo = (act && le) ? mid : lo;
          hi = (act && !le) ? mid : hi;
        }
      }
      d_s0 = lo;
      d_b0 = pre[lo];
      d_b1 = pre[lo + 1 < ns ? lo + 1 : ns];
      d_b2 = pre[lo + 2 < ns ? lo + 2 : ns];
      d_b3 = pre[lo + 3 < ns ? lo + 3 : ns];  // (pre[ns] = total: never <= a row index)
    };
    auto issue1 = [&](u32 tile_begin) {
      u64 off[WC_ITEMS];
      const u32 wq0 = tile_begin + (u32)w * (WC_ITEMS * 64);  // the wave's first row of the tile
      u32 wlast = wq0 + WC_ITEMS * 64 - 1;
      wlast = wlast < total ? wlast : total - 1;
      const u32 s00 = (u32)__builtin_amdgcn_readlane((int)d_s0, 0), b00 = (u32)__builtin_amdgcn_readlane((int)d_b0, 0),
                b10 = (u32)__builtin_amdgcn_readlane((int)d_b1, 0);
      if (abl == 4) {  // (developer ablation: no gather, no plan)
#pragma unroll
        for (int r = 0; r < WC_ITEMS; r++) {
          u32 q = wq0 + lane + r * 64;
          off[r] = q < total ? q : total - 1;
        }
      } else if (b10 > wlast) {  // wave-uniform, the common case: all 256 rows of the wave lie in one slab
        const u64 o0 = (u64)s00 * CA;
#pragma unroll
        for (int r = 0; r < WC_ITEMS; r++) {
          u32 q = wq0 + lane + r * 64;
          q = q < total ? q : total - 1;
          off[r] = abl == 1 ? (u64)q : o0 + (q - (abl == 2 ? b00 & ~7u : b00));
        }
      } else {
#pragma unroll
        for (int r = 0; r < WC_ITEMS; r++) {
          const u32 s0 = (u32)__builtin_amdgcn_readlane((int)d_s0, r), b0 = (u32)__builtin_amdgcn_readlane((int)d_b0, r),
                    b1 = (u32)__builtin_amdgcn_readlane((int)d_b1, r), b2 = (u32)__builtin_amdgcn_readlane((int)d_b2, r),
                    b3 = (u32)__builtin_amdgcn_readlane((int)d_b3, r);
          u32 q = wq0 + lane + r * 64;
          q = q < total ? q : total - 1;
          u32 qlast = wq0 + r * 64 + 63;
          qlast = qlast < total ? qlast : total - 1;
          if (b3 > qlast) {  // wave-uniform: the round crosses at most the slab starts b1 and b2
            const u32 j = (q >= b1 ? 1u : 0u) + (q >= b2 ? 1u : 0u);
            const u32 base = j == 0 ? b0 : (j == 1 ? b1 : b2);
            off[r] = abl == 1 ? (u64)q : (u64)(s0 + j) * CA + (q - (abl == 2 ? base & ~7u : base));
          } else {  // tiny or empty slabs in a row: walk
            u32 sx = s0;
            while (sx + 1 < ns && pre[sx + 1] <= q) sx++;
            off[r] = (u64)sx * CA + (q - pre[sx]);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < WC_ITEMS; r++) t[r] = HMJ_B_NT ? load_stream(&bucket[off[r]]) : bucket[off[r]];
    };
    // ADDR == 2: the same lookup ONCE per wave and tile, in scalar registers, and mostly without touching memory.
    // Each wave keeps a WINDOW of 64 consecutive slab starts in one vector register (lane i: pre[win_lo + i]).  A
    // ballot of "start <= first row" counts the slabs before the wave's first row of the next tile, and five
    // readlanes fetch that slab's start and the next four.  The wave's first row advances by one tile (four
    // average slabs) at a time, so the window serves about a dozen tiles before it is read again (one LDS
    // instruction at the interpolated position; a binary search if that misses).  The wave's 256 rows cross at
    // most three slab starts (else: the per-row walk), so a row finds its slab with three compares against
    // scalars -- no per-round descriptors, no lane 0..3 detour.
    u32 u_s0 = 0, u_b0 = 0, u_b1 = 0, u_b2 = 0, u_b3 = 0, u_b4 = 0;  // the tile to load next (wave-uniform)
    const u32 wq = (u32)__builtin_amdgcn_readfirstlane(w) * (WC_ITEMS * 64);
    u32 win_lo = 0;
    u32 win_v = pre[(u32)lane < ns ? (u32)lane : ns];
    auto plan2 = [&](u32 tile_begin) {
      u32 q0 = tile_begin + wq;
      q0 = q0 < total ? q0 : total - 1;
      u64 m = __ballot(win_v <= q0);  // pre[] is nondecreasing: a prefix of ones (pre[ns] = total > q0)
      u32 j = (u32)__popcll(m) - 1;
      if (!(m & 1) || j > 59) {  // q0 has left the window (or is too near its end): move it
        u32 g = (u32)__builtin_amdgcn_readfirstlane((int)(u32)((float)q0 * slabs_per_row));
        g = g < ns ? g : ns - 1;
        win_lo = g >= 8 ? g - 8 : 0;
        win_v = pre[win_lo + (u32)lane < ns ? win_lo + (u32)lane : ns];
        m = __ballot(win_v <= q0);
        j = (u32)__popcll(m) - 1;
        if (!(m & 1) || j > 59) {  // the guess is off by more than the window: binary search
          u32 lo = 0, hi = ns;     // pre[lo] <= q0, and hi == ns or pre[hi] > q0
#pragma unroll
          for (int it = 0; it < 9; it++) {  // ns <= SLAB_MAXSEG = 512
            const u32 mid = (lo + hi) >> 1;
            const bool le = pre[mid] <= q0, act = hi - lo > 1;
            lo = (act && le) ? mid : lo;
            hi = (act && !le) ? mid : hi;
          }
          win_lo = (u32)__builtin_amdgcn_readfirstlane((int)lo);
          // slabs lo + 1 ... may be empty (same start): the last one with pre <= q0 is what the ballot counts
          win_v = pre[win_lo + (u32)lane < ns ? win_lo + (u32)lane : ns];
          m = __ballot(win_v <= q0);
          j = (u32)__popcll(m) - 1;
          j = j > 59 ? 59 : j;  // (60 empty slabs in a row: the per-row walk below finds the way)
        }
      }
      u_s0 = win_lo + j;
      u_b0 = (u32)__builtin_amdgcn_readlane((int)win_v, (int)j);
      u_b1 = (u32)__builtin_amdgcn_readlane((int)win_v, (int)j + 1);
      u_b2 = (u32)__builtin_amdgcn_readlane((int)win_v, (int)j + 2);
      u_b3 = (u32)__builtin_amdgcn_readlane((int)win_v, (int)j + 3);
      u_b4 = (u32)__builtin_amdgcn_readlane((int)win_v, (int)j + 4);
    };
    auto issue2 = [&](u32 tile_begin) {
      u64 off[WC_ITEMS];
      const u32 wq0 = tile_begin + wq;
      u32 wlast = wq0 + WC_ITEMS * 64 - 1;
      wlast = wlast < total ? wlast : total - 1;
      if (abl == 4) {  // (developer ablation: no gather, no plan)
#pragma unroll
        for (int r = 0; r < WC_ITEMS; r++) {
          u32 q = wq0 + lane + r * 64;
          off[r] = q < total ? q : total - 1;
        }
      } else if (u_b1 > wlast) {  // the common case: all 256 rows of the wave lie in one slab
        const u64 o0 = (u64)u_s0 * CA;
#pragma unroll
        for (int r = 0; r < WC_ITEMS; r++) {
          u32 q = wq0 + lane + r * 64;
          q = q < total ? q : total - 1;
          off[r] = o0 + (q - u_b0);
        }
      } else if (u_b4 > wlast) {  // at most the slab starts b1, b2, b3 inside the wave's rows
#pragma unroll
        for (int r = 0; r < WC_ITEMS; r++) {
          u32 q = wq0 + lane + r * 64;
          q = q < total ? q : total - 1;
          const u32 j = (q >= u_b1 ? 1u : 0u) + (q >= u_b2 ? 1u : 0u) + (q >= u_b3 ? 1u : 0u);
          const u32 base = j == 0 ? u_b0 : (j == 1 ? u_b1 : (j == 2 ? u_b2 : u_b3));
          off[r] = (u64)(u_s0 + j) * CA + (q - base);
        }
      } else {  // tiny or empty slabs in a row: walk
#pragma unroll
        for (int r = 0; r < WC_ITEMS; r++) {
          u32 q = wq0 + lane + r * 64;
          q = q < total ? q : total - 1;
          u32 sx = u_s0;
          while (sx + 1 < ns && pre[sx + 1] <= q) sx++;
          off[r] = (u64)sx * CA + (q - pre[sx]);
        }
      }
#pragma unroll
      for (int r = 0; r < WC_ITEMS; r++) t[r] = HMJ_B_NT ? load_stream(&bucket[off[r]]) : bucket[off[r]];
    };
    auto plan = [&](u32 tb) {
      if constexpr (ADDR == 2) plan2(tb); else plan1(tb);
    };
    auto issue = [&](u32 tb) {
      if constexpr (ADDR == 2) issue2(tb); else issue1(tb);
    };
    if (abl != 4) plan(0);
    issue(0);
    if ((u32)TILE < total && abl != 4) plan(TILE);
    for (u32 tile = 0; tile < total; tile += TILE) {
      const u32 tile_n = (total - tile < (u32)TILE) ? total - tile : (u32)TILE;
      auto prefetch = [&]() {
        if (tile + TILE < total) {
          issue(tile + TILE);
#ifdef HMJ_STAMPS
          stamps.mark(13);
#endif
          if (tile + 2 * TILE < total && abl != 4) plan(tile + 2 * TILE);
        }
      };
#ifdef HMJ_DEV
      if (abl == 1 || abl == 4) {  // never-written slab tails hold zero keys: give them spread digits (the rank phase waits for the rows anyway)
#pragma unroll
        for (int r = 0; r < WC_ITEMS; r++)
          if (t[r].key == 0) t[r].key = (u64)(tile + (u32)tid + r * THREADS + blockIdx.x * 7919u + 1) * 0x9E3779B97F4A7C15ull;
      }
#endif
      if (HMJ_WC_FULLTILE && tile_n == (u32)TILE)
        wc_tile<THREADS, MAXD, HI, true, true>(sm, t, tile_n, slab_b, shift, mask, D, prefetch, limit, &ovf WC_STAMP_PASS);
      else
        wc_tile<THREADS, MAXD, HI, false, true>(sm, t, tile_n, slab_b, shift, mask, D, prefetch, limit, &ovf WC_STAMP_PASS);
    }
    } else {
    u32 seg = 0;  // first A-slab overlapping the tile being loaded (same value in every thread)
    auto load = [&](u32 tile_begin) {
      const u32 tn = (total - tile_begin < (u32)TILE) ? total - tile_begin : (u32)TILE;
      while (seg + 1 < ns && pre[seg + 1] <= tile_begin) seg++;
      u32 s = seg;
      u64 off[WC_ITEMS];
#pragma unroll
      for (int r = 0; r < WC_ITEMS; r++) {
        u32 q = wbase + r * 64;
        q = tile_begin + (q < tn ? q : tn - 1);
        while (s + 1 < ns && pre[s + 1] <= q) s++;
        off[r] = (u64)s * CA + (q - pre[s]);
      }
#pragma unroll
      for (int r = 0; r < WC_ITEMS; r++) t[r] = HMJ_B_NT ? load_stream(&bucket[off[r]]) : bucket[off[r]];
    };
    load(0);
    for (u32 tile = 0; tile < total; tile += TILE) {
      const u32 tile_n = (total - tile < (u32)TILE) ? total - tile : (u32)TILE;
      wc_tile<THREADS, MAXD, HI, false, true>(
          sm, t, tile_n, slab_b, shift, mask, D,
          [&]() {
            if (tile + TILE < total) load(tile + TILE);
          },
          limit, &ovf WC_STAMP_PASS);
    }
    }
    wc_flush_carry(sm, slab_b, D);
  }
#ifdef HMJ_STAMPS
  stamps.flush();
#endif
  if ((u32)tid < D) {
    const u32 pid = (((u32)tid << bits_a) | dA) * KB + k;
    cnt_b[pid] = sm.cflush[tid] + sm.pend[tid] - pid * CB;
  }
  if (ovf && !abl) atomicOr(&accum[ACC_ERR], ERR_SLAB);
}

// ---------------------------------------------------------------------------------------------
// Partition boundaries of an array sorted by partition id = (key >> low) & (2^bits - 1):
// off[p] = first row whose id >= p (binary search; P+1 entries).  bits == 0 -> {0, n}.
// ---------------------------------------------------------------------------------------------
__global__ void part_offsets_kernel(const Tup* __restrict__ a, u32 n, int low, int bits,
                                    u32* __restrict__ off, u32 P) {
  u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > P) return;
  if (p == P || bits == 0) {
    off[p] = (p == P) ? n : 0;
    return;
  }
  const u64* keys = reinterpret_cast<const u64*>(a);
  u32 lo = 0, hi = n;
  while (lo < hi) {
    u32 mid = lo + ((hi - lo) >> 1);
    if (((u32)(keys[2 * (u64)mid] >> low) & (P - 1)) < p)
      lo = mid + 1;
    else
      hi = mid;
  }
  off[p] = lo;
}

// ---------------------------------------------------------------------------------------------
// Key-range probe for the planner: OR of (key ^ reference key) over a strided sample of both
// relations.  Its leading zeros are the top key bits all sampled keys share (dense integer keys:
// the zero high bits), which carry no information for partitioning; the reference's MSD recursion
// reaches the informative bits by recursing (radix_hash.h:277-289), the GPU plan skips to them.
// out[0] = OR, out[1] = reference key.  A sample can miss outliers: the join stays correct for any
// prefix (the partition id is a function of the key); ordered output additionally verifies it.
// ---------------------------------------------------------------------------------------------
// out[4] / out[5]: smallest / largest sampled build key (the planner's estimate of the build side's key range).
// out[2] / out[3]: number of neighbouring sample positions of the build / probe relation that carry the SAME
// key (a hint that costs two shuffles: unique keys give 0; a key that holds 5 % of a relation makes ~5 of the
// 2048 neighbouring pairs equal).
__global__ __launch_bounds__(1024) void key_sample_kernel(const Tup* __restrict__ R, u32 nb,
                                                          const Tup* __restrict__ S, u32 np,
                                                          u64* __restrict__ out) {
  __shared__ u64 acc;
  __shared__ unsigned long long rng[2];  // smallest / largest sampled build key
  __shared__ u32 bins[64];               // sampled build keys per 64th of that range
  __shared__ u32 same[2];
  __shared__ u32 order[4];  // ascending / descending neighbouring sample pairs of R, of S
  const int tid = threadIdx.x, lane = tid & 63;
  const u64* rk = reinterpret_cast<const u64*>(R);
  const u64* sk = reinterpret_cast<const u64*>(S);
  const u64 ref = nb ? rk[0] : (np ? sk[0] : 0);
  if (tid == 0) {
    acc = 0;
    rng[0] = ~0ull;
    rng[1] = 0;
  }
  if (tid < 2) same[tid] = 0;
  if (tid < 4) order[tid] = 0;
  if (tid < 64) bins[tid] = 0;
  __syncthreads();
  u64 x = 0, mn = ~0ull, mx = 0;
  u64 mykey[2] = {0, 0};
  bool myvalid[2] = {false, false};
  u32 eq_r = 0, eq_s = 0, up_r = 0, dn_r = 0, up_s = 0, dn_s = 0;
  const u32 sr = nb / 2048 + 1, ss = np / 2048 + 1;  // ~2048 samples per relation
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const u64 i = (u64)(tid + k * 1024) * sr, j = (u64)(tid + k * 1024) * ss;
    const bool vr = i < nb, vs = j < np;
    const u64 kr = vr ? rk[2 * i] : 0, ks = vs ? sk[2 * j] : 0;
    if (vr) {
      x |= kr ^ ref;
      mn = kr < mn ? kr : mn;
      mx = kr > mx ? kr : mx;
      mykey[k] = kr;
      myvalid[k] = true;
    }
    if (vs) x |= ks ^ ref;
    const u64 kr_next = __shfl_down(kr, 1, kWave), ks_next = __shfl_down(ks, 1, kWave);
    const bool vr_next = __shfl_down((int)vr, 1, kWave) != 0, vs_next = __shfl_down((int)vs, 1, kWave) != 0;
    if (lane < 63 && vr && vr_next && kr == kr_next) eq_r++;
    if (lane < 63 && vs && vs_next && ks == ks_next) eq_s++;
    // neighbouring samples in ascending / descending key order: a relation that arrives sorted hands every worker
    // of the histogram-free slab pass rows of ONE digit (its slab for that digit overflows)
    if (lane < 63 && vr && vr_next) (kr < kr_next ? up_r : dn_r)++;
    if (lane < 63 && vs && vs_next) (ks < ks_next ? up_s : dn_s)++;
  }
  if (nb) {
    const u64 kl = rk[2 * (u64)(nb - 1)];
    x |= kl ^ ref;
    mn = kl < mn ? kl : mn;
    mx = kl > mx ? kl : mx;
  }
  if (np) x |= sk[2 * (u64)(np - 1)] ^ ref;
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) x |= __shfl_xor(x, o, kWave);
  if ((tid & 63) == 0 && x) atomicOr(&acc, x);
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) {
    const u64 a2 = __shfl_xor(mn, o, kWave), b2 = __shfl_xor(mx, o, kWave);
    mn = a2 < mn ? a2 : mn;
    mx = b2 > mx ? b2 : mx;
  }
  if ((tid & 63) == 0 && nb) {
    atomicMin(&rng[0], (unsigned long long)mn);
    atomicMax(&rng[1], (unsigned long long)mx);
  }
  if (eq_r) atomicAdd(&same[0], eq_r);
  if (eq_s) atomicAdd(&same[1], eq_s);
  if (up_r) atomicAdd(&order[0], up_r);
  if (dn_r) atomicAdd(&order[1], dn_r);
  if (up_s) atomicAdd(&order[2], up_s);
  if (dn_s) atomicAdd(&order[3], dn_s);
  __syncthreads();
  {  // how evenly the sampled build keys fill their range: 64 equal bins
    const u64 lo = rng[0], span = rng[1] - rng[0];
#pragma unroll
    for (int k = 0; k < 2; k++)
      if (myvalid[k] && span) atomicAdd(&bins[(u32)((mykey[k] - lo) / ((span >> 6) + 1)) & 63u], 1u);
  }
  __syncthreads();
  if (tid == 0) {
    out[0] = acc;
    out[1] = ref;
    // (bits 0-15: neighbouring samples with equal keys; 16-31 / 32-47: pairs in ascending / descending order)
    out[2] = (u64)same[0] | ((u64)order[0] << 16) | ((u64)order[1] << 32);
    out[3] = (u64)same[1] | ((u64)order[2] << 16) | ((u64)order[3] << 32);
    out[4] = rng[0];  // (~0, 0 without build rows)
    out[5] = rng[1];
    u32 nonempty = 0, big = 0;
    for (int i = 0; i < 64; i++) {
      nonempty += bins[i] ? 1u : 0u;
      big = bins[i] > big ? bins[i] : big;
    }
    out[6] = nonempty;  // bins (64ths of the build key range) that hold a sampled build key
    out[7] = big;       // ... and the fullest bin
  }
}

// Two shapes per slab kernel: digits of up to 8 bits run 512 threads x 2048-row tiles, two workgroups per CU (74 KiB
// LDS); a 9-bit digit needs 512 carry lines (64 KiB) beside a 4096-row tile and runs 1024 threads, one workgroup per
// CU (157 + 2 KiB) -- the shape radix_scatter_wc_kernel<1024, 512> has on the exact path.
template <int THREADS, int MAXD, bool HI>
__global__ __launch_bounds__(THREADS, (THREADS * (MAXD <= 256 ? 2 : 1)) / 256) void radix_slab_a_kernel(
    const Tup* __restrict__ in, u32 n, int shift, int bits, u32 rows_per_worker, Tup* __restrict__ slab,
    u32 CA, u32 WA, u32* __restrict__ cnt_out, u64* __restrict__ accum, u32 worker_base) {
  NoXform xf;
  slab_a_body<THREADS, MAXD, HI, NoXform>(in, n, shift, bits, rows_per_worker, slab, CA, WA, cnt_out, accum, worker_base, xf);
}

// Pass A with the rank lookup in front (the rank-run form of ordered small-build joins, gtable.hip / api.hip): the rows
// read are PROBE rows, the rows scattered are {rank of the probe key among the sorted build keys, probe payload} -- what
// gtable_emit_kernel would have written to memory for this pass to read back (32 B per probe row saved).  The table maps
// key -> rank (gtable_build_kernel<true> over the sorted build rows).  A probe row whose key is not in the table cannot be
// dropped from the middle of a tile: the kernel raises ERR_FASTPATH and the host takes emit + pass A (foreign keys that
// all have their dimension row -- the usual case -- never get there).  Sums of the probe payloads as gtable_emit_kernel.
// CUT: runs beyond one workgroup's sort are cut into 2^tb pieces by the position of the payload in the payloads' range, and
// the pass reads chunks from all over the probe side (slab_a_body's strided form) so that payloads which grow with the
// row's position do not land a worker's rows in one piece.
template <bool CUT>
struct RankXform {
  static constexpr bool kStrided = CUT;
  u32 slot_mult;  // (strided form: worker w reads the chunks w * slot_mult mod WA of every stride)
  const Tup* __restrict__ tab;
  u32 mask;
  int hshift;
  int tb, pre;  // runs beyond one workgroup: the partition key is rank << tb | rank_run_bucket(sval - svmin, pre, mult) (hmj_dev.h)
  u64 svmin, svrange, mult;  // (the range is a SAMPLE's: payloads outside it go to the first / last piece)
  u64 acc_s;
  u64 n_rows;
  bool miss;
  __device__ __forceinline__ void operator()(Tup (&t)[WC_ITEMS], u32 tile_n, u32 wbase) {
    u32 slot[WC_ITEMS];
    bool live[WC_ITEMS], want[WC_ITEMS];
    u32 found = 0;
#pragma unroll
    for (int r = 0; r < WC_ITEMS; r++) {
      want[r] = wbase + (u32)r * 64 < tile_n;
      slot[r] = gt_hash(t[r].key, hshift);
      if (want[r]) acc_s += t[r].val;
      if (want[r]) n_rows++;
      live[r] = want[r] && t[r].key != GT_EMPTY;  // (the empty marker is never a table key)
    }
    u64 rank[WC_ITEMS] = {};
    gt_walk<WC_ITEMS>(tab, mask, slot, live, [&](int r, const Tup& e) {
      const bool eq = e.key == t[r].key;
      if (eq) {
        rank[r] = e.val;
        found |= 1u << r;
      }
      return eq;
    });
#pragma unroll
    for (int r = 0; r < WC_ITEMS; r++) {
      miss |= want[r] && !((found >> r) & 1u);
      if (CUT) {
        u64 off = t[r].val - svmin;
        off = t[r].val < svmin ? 0ull : off > svrange ? svrange : off;
        t[r].key = (rank[r] << tb) | rank_run_bucket(off, pre, mult);
      } else {
        t[r].key = rank[r];
      }  // (.val stays the probe payload)
    }
  }
};
template <int THREADS, int MAXD, bool CUT>
__global__ __launch_bounds__(THREADS, (THREADS * (MAXD <= 256 ? 2 : 1)) / 256) void radix_slab_a_rank_kernel(
    const Tup* __restrict__ in, u32 n, int shift, int bits, u32 rows_per_worker, Tup* __restrict__ slab, u32 CA, u32 WA,
    u32* __restrict__ cnt_out, u64* __restrict__ accum, const Tup* __restrict__ tab, int log_cap, bool extra, int tb, u64 svmin, u64 svrange,
    int pre, u64 mult, u32 slot_mult) {
  // (duplicate build keys or a table that gave up -- the host sees the same words and discards this pass -- need no early
  //  exit: the table is fully initialised either way, every walk ends at an empty slot, every rank is a build row's index;
  //  and pass B trusts the counts written below, so they must be written)
  RankXform<CUT> xf{slot_mult, tab, (1u << log_cap) - 1, 64 - log_cap, tb, pre, svmin, svrange, mult, 0ull, 0ull, false};
  slab_a_body<THREADS, MAXD, false, RankXform<CUT>>(in, n, shift, bits, rows_per_worker, slab, CA, WA, cnt_out, accum, 0u, xf);  // (shift: ranks grouped 2^shift to a partition)
  // sums: one set of atomics per workgroup
  __shared__ u64 wsum[THREADS / kWave], wrows[THREADS / kWave];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const u64 ws = wave_sum_u64(xf.acc_s), wr = wave_sum_u64(xf.n_rows);
  const bool any_miss = __any(xf.miss);
  if (lane == 0) {
    wsum[wv] = ws;
    wrows[wv] = wr;
  }
  if (any_miss && lane == 0) atomicOr(reinterpret_cast<unsigned long long*>(&accum[ACC_ERR]), (unsigned long long)ERR_FASTPATH);
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 tot = 0, rows = 0;
    for (int w = 0; w < THREADS / kWave; w++) {
      tot += wsum[w];
      rows += wrows[w];
    }
    if (tot) atomicAdd(reinterpret_cast<unsigned long long*>(&accum[ACC_SUM_S]), (unsigned long long)tot);
    if (extra && tot) atomicAdd(reinterpret_cast<unsigned long long*>(&accum[ACC_SUM_P]), (unsigned long long)tot);
    if (rows) atomicAdd(reinterpret_cast<unsigned long long*>(&accum[ACC_N]), (unsigned long long)rows);
  }
}
template <int THREADS, int MAXD, bool HI>
static hipError_t launch_slab_a_t(const void* in, u32 n, int shift, int bits, u32 rpw, void* slab, u32 CA,
                                  u32 WA, u32* cnt, u64* accum, hipStream_t st, u32 w_begin, u32 w_end) {
  typedef WcSmem<THREADS, MAXD> Smem;
  const size_t smem = sizeof(Smem);
  static_assert(sizeof(Smem) <= 160 * 1024, "slab pass A: LDS");
  static SmemAttrOnce attr_once;
  if (hipError_t e = ensure_max_smem(attr_once, reinterpret_cast<const void*>(radix_slab_a_kernel<THREADS, MAXD, HI>), (size_t)smem); e != hipSuccess) return e;
  if (w_end > WA) w_end = WA;
  if (w_begin >= w_end) return hipSuccess;
  hipLaunchKernelGGL((radix_slab_a_kernel<THREADS, MAXD, HI>), dim3(w_end - w_begin), dim3(THREADS), smem, st,
                     static_cast<const Tup*>(in), n, shift, bits, rpw, static_cast<Tup*>(slab), CA, WA, cnt,
                     accum, w_begin);
  return hipGetLastError();
}

template <int THREADS, int MAXD, bool HI>
static hipError_t launch_slab_b_t(const void* slab_a, const u32* cnt_a, u32 CA, u32 WA, int bits_a, int shift,
                                  int bits, void* slab_b, u32 CB, u32* cnt_b, u64* accum, hipStream_t st, u32 KB) {
  typedef WcSmem<THREADS, MAXD> Smem;
  const size_t smem = sizeof(Smem) + (SLAB_MAXSEG + 1) * sizeof(u32);
  static_assert(sizeof(Smem) + (SLAB_MAXSEG + 1) * sizeof(u32) <= 160 * 1024, "slab pass B: LDS");
#ifdef HMJ_DEV
  if (g_b_addr_alt) {  // the other lookup, for A/B timing inside one process (hmj_dev_set_b_addr_alt)
    constexpr int ALT = HMJ_B_ADDR == 2 ? 1 : 2;
    static SmemAttrOnce attr_alt;
    if (hipError_t e = ensure_max_smem(attr_alt, reinterpret_cast<const void*>(radix_slab_b_kernel<THREADS, MAXD, HI, ALT>), (size_t)smem); e != hipSuccess) return e;
    hipLaunchKernelGGL((radix_slab_b_kernel<THREADS, MAXD, HI, ALT>), dim3((1u << bits_a) * KB), dim3(THREADS), smem, st,
                       static_cast<const Tup*>(slab_a), cnt_a, CA, WA, bits_a, shift, bits,
                       static_cast<Tup*>(slab_b), CB, cnt_b, accum, KB);
    return hipGetLastError();
  }
#endif
  static SmemAttrOnce attr_once;
  if (hipError_t e = ensure_max_smem(attr_once, reinterpret_cast<const void*>(radix_slab_b_kernel<THREADS, MAXD, HI, HMJ_B_ADDR>), (size_t)smem); e != hipSuccess) return e;
  hipLaunchKernelGGL((radix_slab_b_kernel<THREADS, MAXD, HI, HMJ_B_ADDR>), dim3((1u << bits_a) * KB), dim3(THREADS), smem, st,
                     static_cast<const Tup*>(slab_a), cnt_a, CA, WA, bits_a, shift, bits,
                     static_cast<Tup*>(slab_b), CB, cnt_b, accum, KB);
  return hipGetLastError();
}

// Geometry of the slab path for a relation of n rows split by bits_a then bits_b (both <= 9).
// cap(m) = m + 8 sqrt(m) + 24 rounded up to whole 128-byte lines.
// fan: rows per distinct key (a probe side drawn from a smaller key domain): a digit's row count then varies
// like fan * (number of its keys), i.e. sigma = sqrt(fan * mean) instead of sqrt(mean)
static u32 slab_cap(double mean, double fan) {
  double c = mean + 8.0 * __builtin_sqrt(fan * mean) + 24.0;
  return ((u32)c + 8) & ~7u;
}
u32 slab_capacity(double mean, double fan) { return slab_cap(mean, fan < 1.0 ? 1.0 : fan); }
bool slab_geometry(u32 n, int bits_a, int bits_b, SlabGeom* g, u32 kb, double fan, double density) {
  if (bits_a < 1 || bits_a > SLAB_MAX_BITS || bits_b < 1 || bits_b > SLAB_MAX_BITS) return false;
  const u32 tile = bits_a > 8 ? 4096 : 2048;  // pass A's tile: a worker owns whole tiles (its last one may be ragged)
  u64 tiles = ((u64)n + tile - 1) / tile;
  u64 max_workers = 2048;
#ifdef HMJ_DEV
  if (const char* e = getenv("HMJ_SLAB_WORKERS")) max_workers = (u64)atoi(e) > 0 ? (u64)atoi(e) : max_workers;
#endif
  u64 tpw = (tiles + max_workers - 1) / max_workers;
  if (tpw == 0) tpw = 1;
  g->WA = (u32)((tiles + tpw - 1) / tpw);
  g->rpw = (u32)(tpw * tile);
  if (kb == 0) kb = SLAB_KB;
  g->KB = kb;
  if (g->WA < kb || (g->WA + kb - 1) / kb > (u32)SLAB_MAXSEG) return false;
  if (fan < 1.0) fan = 1.0;
  if (density < 1.0) density = 1.0;  // keys that fill 1 / density of the partitions: those hold density x the mean
  // (pass A: a worker's rows are a SAMPLE of the relation, rpw of n rows, so a digit's share of them varies with the
  //  sampling -- mean -- and with the digit's share of the KEYS, n / fan / 2^bits_a of them: variance mean * (1 + rpw * fan
  //  / n), four times that term for safety.  Until round 5 this was `fan` as for pass B: 2^19 x 2^28 rows, fan-out 512,
  //  asked for 54 GiB of pass-A slabs for 4 GiB of rows.)
  const double fan_a = 1.0 + 4.0 * (double)g->rpw * fan / (double)(n ? n : 1);
  g->CA = slab_cap(density * (double)g->rpw / (double)(1u << bits_a), fan_a < fan ? fan_a : fan);
  g->CB = slab_cap(density * (double)n / (double)(1u << bits_a) / (double)kb / (double)(1u << bits_b), fan);
  const u64 rows_a = (u64)(1u << bits_a) * g->WA * g->CA;
  const u64 rows_b = (u64)(1u << (bits_a + bits_b)) * kb * g->CB;
  g->rows_a = rows_a;
  g->rows_b = rows_b;
  return rows_a < 0xFFFFFFF0ull && rows_b < 0xFFFFFFF0ull;
}

// One slab pass whose worker-private slabs the probe kernel reads directly (probe_kernel's piece walk).  A worker's m rows
// fall into digit d with probability p_d = (keys of d) / (all keys); p_d varies over the digits like a Poisson count of
// keys_per_digit, so a slab's row count has variance mean + mean^2 / keys_per_digit (sampling noise + key-count noise) --
// NOT fan-out x mean, which is the spread of a digit's rows over the WHOLE relation.
bool slab_geometry_one_pass(u32 n, int bits, double keys_per_digit, u32 max_workers, SlabGeom* g, double density) {
  if (bits < 1 || bits > SLAB_MAX_BITS || max_workers == 0) return false;
  const u32 tile = bits > 8 ? 4096 : 2048;
  const u64 tiles = ((u64)n + tile - 1) / tile;
  u64 tpw = (tiles + max_workers - 1) / max_workers;
  if (tpw == 0) tpw = 1;
  g->WA = (u32)((tiles + tpw - 1) / tpw);
  g->rpw = (u32)(tpw * tile);
  g->KB = 0;
  g->CB = 0;
  g->rows_b = 0;
  if (density < 1.0) density = 1.0;  // digits that fill 1 / density of their range: the filled ones hold density x the mean
  const double mean = density * (double)g->rpw / (double)(1u << bits);
  if (keys_per_digit < 1.0) keys_per_digit = 1.0;
  g->CA = slab_cap(mean, 1.0 + mean / keys_per_digit);
  g->rows_a = (u64)(1u << bits) * g->WA * g->CA;
  return g->WA >= 1 && g->rows_a < 0xFFFFFFF0ull;
}

// workers [w_begin, w_end) of the pass (default: all): worker w reads input rows [w * g.rpw, (w + 1) * g.rpw)
hipError_t launch_slab_a(const void* in, u32 n, int shift, int bits, const SlabGeom& g, void* slab_a, u64 slab_a_rows,
                         u32* cnt_a, u64 cnt_a_n, u64* accum, hipStream_t st, u32 w_begin, u32 w_end) {
  // worker w writes slab [d][w][CA] and cnt[d * WA + w] for every digit d < 2^bits, and reads rows [w * rpw, ...)
  if (bits < 1 || bits > SLAB_MAX_BITS || g.WA == 0 || g.CA == 0 || (u64)g.WA * g.rpw < n) return hipErrorInvalidValue;
  if (slab_a_rows < ((u64)g.WA << bits) * g.CA || cnt_a_n < ((u64)g.WA << bits)) return hipErrorInvalidValue;
  if (((u64)g.WA << bits) * g.CA >= 0xFFFFFFF0ull) return hipErrorInvalidValue;  // (u32 row indices inside the kernel)
  const bool hi = shift >= 32;
#define HMJ_SLAB_A(T, M, H) launch_slab_a_t<T, M, H>(in, n, shift, bits, g.rpw, slab_a, g.CA, g.WA, cnt_a, accum, st, w_begin, w_end)
  if (bits > 8) return hi ? HMJ_SLAB_A(1024, 512, true) : HMJ_SLAB_A(1024, 512, false);
  return hi ? HMJ_SLAB_A(512, 256, true) : HMJ_SLAB_A(512, 256, false);
#undef HMJ_SLAB_A
}

// pass A of the rank-run form with the key -> rank lookup in front (radix_slab_a_rank_kernel): in = PROBE rows
hipError_t launch_slab_a_ranks(const void* in, u32 n, int shift, int bits, const SlabGeom& g, void* slab_a, u64 slab_a_rows, u32* cnt_a,
                               u64 cnt_a_n, u64* accum, const void* tab, int log_cap, bool extra, int tb, u64 svmin, u64 svrange, int pre,
                               u64 mult, hipStream_t st) {
  if (bits < 1 || bits > SLAB_MAX_BITS || g.WA == 0 || g.CA == 0 || (u64)g.WA * g.rpw < n || !tab || log_cap < 4 || log_cap > 30 || tb < 0 ||
      tb > 16 || pre < 0 || pre > 32 || shift < 0 || shift > 4 || (shift && tb))
    return hipErrorInvalidValue;
  if (slab_a_rows < ((u64)g.WA << bits) * g.CA || cnt_a_n < ((u64)g.WA << bits)) return hipErrorInvalidValue;
  if (((u64)g.WA << bits) * g.CA >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
  u32 slot_mult = 1;
  if (tb > 0 && g.WA > 2) {
    auto gcd = [](u32 a, u32 b) {
      while (b) {
        const u32 t = a % b;
        a = b;
        b = t;
      }
      return a;
    };
    slot_mult = (u32)((double)g.WA * 0.6180339887498949);
    if (slot_mult < 1) slot_mult = 1;
    while (gcd(slot_mult, g.WA) != 1) slot_mult++;
  }
#define HMJ_SLAB_AR(T, M, C)                                                                                                          \
  {                                                                                                                                   \
    typedef WcSmem<T, M> Smem;                                                                                                        \
    static SmemAttrOnce once;                                                                                                         \
    if (hipError_t e = ensure_max_smem(once, reinterpret_cast<const void*>(radix_slab_a_rank_kernel<T, M, C>), sizeof(Smem)); e != hipSuccess) \
      return e;                                                                                                                       \
    hipLaunchKernelGGL((radix_slab_a_rank_kernel<T, M, C>), dim3(g.WA), dim3(T), sizeof(Smem), st, static_cast<const Tup*>(in), n, shift, bits, g.rpw, \
                       static_cast<Tup*>(slab_a), g.CA, g.WA, cnt_a, accum, static_cast<const Tup*>(tab), log_cap, extra, tb, svmin, svrange, pre, mult, \
                       slot_mult);                                                                                                    \
  }
  if (tb > 0) {
    if (bits > 8) HMJ_SLAB_AR(1024, 512, true) else HMJ_SLAB_AR(512, 256, true)
  } else {
    if (bits > 8) HMJ_SLAB_AR(1024, 512, false) else HMJ_SLAB_AR(512, 256, false)
  }
#undef HMJ_SLAB_AR
  return hipGetLastError();
}

hipError_t launch_slab_b(const void* slab_a, const u32* cnt_a, int bits_a, int shift, int bits,
                         const SlabGeom& g, void* slab_b, u64 slab_b_rows, u32* cnt_b, u64 cnt_b_n, u64* accum, hipStream_t st) {
  // worker (dA, k) writes piece (d << bits_a | dA) * KB + k of every digit d < 2^bits: P * KB pieces of CB rows, one
  // count each; it gathers the A-slabs [k * WA / KB, (k + 1) * WA / KB) -- at most SLAB_MAXSEG of them
  if (bits < 1 || bits > SLAB_MAX_BITS || bits_a < 1 || bits_a > SLAB_MAX_BITS || g.KB == 0 || g.WA < g.KB ||
      (g.WA + g.KB - 1) / g.KB > (u32)SLAB_MAXSEG)
    return hipErrorInvalidValue;
  const u64 pieces = ((u64)g.KB << (bits_a + bits));
  if (slab_b_rows < pieces * g.CB || cnt_b_n < pieces || pieces * g.CB >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
  const bool hi = shift >= 32;
#define HMJ_SLAB_B(T, M, H) launch_slab_b_t<T, M, H>(slab_a, cnt_a, g.CA, g.WA, bits_a, shift, bits, slab_b, g.CB, cnt_b, accum, st, g.KB)
  if (bits > 8) return hi ? HMJ_SLAB_B(1024, 512, true) : HMJ_SLAB_B(1024, 512, false);
  return hi ? HMJ_SLAB_B(512, 256, true) : HMJ_SLAB_B(512, 256, false);
#undef HMJ_SLAB_B
}

// Every key of both relations against the sample's reference key: the bits in which ANY key differs (atomicOr into
// *out, which the caller zeroes), and the smallest / largest build key (out[1], out[2]; caller: ~0, 0).  The retry of an
// ordered join whose sampled prefix some row did not share (a handful of keys above an otherwise dense range) plans
// with these exact values instead.
__global__ __launch_bounds__(256) void key_exact_kernel(const Tup* __restrict__ R, u32 nb, const Tup* __restrict__ S, u32 np,
                                                        u64 ref, bool ref_is_first_key, u64* __restrict__ out) {
  if (ref_is_first_key) ref = nb ? R[0].key : (np ? S[0].key : 0);  // (callers that have not seen a key yet)
  u64 x = 0, mn = ~0ull, mx = 0;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += stride) {
    const u64 k = R[i].key;
    x |= k ^ ref;
    mn = k < mn ? k : mn;
    mx = k > mx ? k : mx;
  }
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += stride) x |= S[i].key ^ ref;
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) {
    x |= __shfl_xor(x, o, kWave);
    const u64 a2 = __shfl_xor(mn, o, kWave), b2 = __shfl_xor(mx, o, kWave);
    mn = a2 < mn ? a2 : mn;
    mx = b2 > mx ? b2 : mx;
  }
  // one set of atomics per WORKGROUP (same-address atomics cost ~11 ns each wherever they come from: a set per wave of a
  // 2048-workgroup grid was 0.2 ms for a 2^16-row build side, profiles/r05a_small16_ord_summary.txt)
  __shared__ u64 wx[4], wmn[4], wmx[4];
  if ((threadIdx.x & 63) == 0) {
    wx[threadIdx.x >> 6] = x;
    wmn[threadIdx.x >> 6] = mn;
    wmx[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) {
      x |= wx[w];
      mn = wmn[w] < mn ? wmn[w] : mn;
      mx = wmx[w] > mx ? wmx[w] : mx;
    }
    if (x) atomicOr(reinterpret_cast<unsigned long long*>(out), (unsigned long long)x);
    if (nb) {
      atomicMin(reinterpret_cast<unsigned long long*>(out + 1), (unsigned long long)mn);
      atomicMax(reinterpret_cast<unsigned long long*>(out + 2), (unsigned long long)mx);
    }
  }
}

hipError_t launch_key_exact(const void* R, u32 nb, const void* S, u32 np, u64 ref, u64* out, int num_cus, hipStream_t st,
                            bool ref_is_first_key) {
  u64 grid = ((u64)nb + np + 1023) / 1024;  // (a workgroup per 1024 rows at most)
  if (grid > (u64)num_cus * 8) grid = (u64)num_cus * 8;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(key_exact_kernel, dim3((u32)grid), dim3(256), 0, st, static_cast<const Tup*>(R), nb,
                     static_cast<const Tup*>(S), np, ref, ref_is_first_key, out);
  return hipGetLastError();
}

// *flag |= 1 if some row's partition number is below its predecessor's: the relation is NOT already partitioned.
__global__ __launch_bounds__(256) void check_partitioned_kernel(const Tup* __restrict__ a, u32 n, int low, u64 mask,
                                                                u32* __restrict__ flag) {
  bool bad = false;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x + 1; i < n; i += stride)
    bad |= ((a[i].key >> low) & mask) < ((a[i - 1].key >> low) & mask);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

hipError_t launch_check_partitioned(const void* a, u32 n, int low, int bits, u32* flag, int num_cus, hipStream_t st) {
  const u64 mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
  hipLaunchKernelGGL(check_partitioned_kernel, dim3(num_cus * 8), dim3(256), 0, st, static_cast<const Tup*>(a), n, low, mask, flag);
  return hipGetLastError();
}

hipError_t launch_key_sample(const void* R, u32 nb, const void* S, u32 np, u64* out, hipStream_t st) {
  hipLaunchKernelGGL(key_sample_kernel, dim3(1), dim3(1024), 0, st, static_cast<const Tup*>(R), nb,
                     static_cast<const Tup*>(S), np, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
// Workers: k = ceil(full_tiles / tiles_per_worker) workers own whole tiles; if n is not a multiple
// of the tile, one more worker owns the ragged tail.  *nblk counts all of them.
// Kernel choice per pass.  Fan-out <= 16: runs are >= 4 KiB anyway, the plain staged scatter is
// fastest.  Otherwise the write-combining kernels (WC_X up to 8 bits, WC_Y for 9).
static inline int scatter_kind(int bits, int variant) {
  if (variant != 1 || bits <= 4) return 0;
  return bits <= 8 ? 1 : 2;
}
int radix_tile_rows(int bits, int variant) {
  const int k = scatter_kind(bits, variant);
  return k == 1 ? 2048 : (k == 2 ? 4096 : RP_TILE);
}

void radix_pass_geometry(u32 n, int tile, u32* nblk, u32* rows_per_block) {
  const u64 RP_TILE = (u64)tile;
  u64 full_tiles = (u64)n / RP_TILE;
  u64 tpb = (full_tiles + RP_MAX_BLOCKS - 1) / RP_MAX_BLOCKS;
  if (tpb == 0) tpb = 1;
  u64 k = (full_tiles + tpb - 1) / tpb;
  *nblk = (u32)(k + ((u64)n % RP_TILE ? 1 : 0));
  if (*nblk == 0) *nblk = 1;
  *rows_per_block = (u32)(tpb * RP_TILE);
}

size_t radix_scatter_smem_bytes() { return sizeof(ScatterSmem); }

hipError_t launch_radix_hist(const void* in, u32 n, int tile, int shift, int bits, u32* hist,
                             u32 nblk, u32 rows_per_block, hipStream_t st, const OwnerFn* own) {
  const u32 n_full = (n / (u32)tile) * (u32)tile;
  OwnerFn f;
  std::memset(&f, 0, sizeof(f));
  if (own) f = *own;
#define HMJ_HIST(O)                                                                                      \
  hipLaunchKernelGGL((radix_hist_kernel<O>), dim3(nblk), dim3(RP_THREADS), 0, st, static_cast<const Tup*>(in), n, \
                     shift, (1u << bits) - 1, rows_per_block, n_full, hist, nblk, f)
  if (f.mode == 1)
    HMJ_HIST(1);
  else if (f.mode == 2)
    HMJ_HIST(2);
  else
    HMJ_HIST(0);
#undef HMJ_HIST
  return hipGetLastError();
}

hipError_t launch_radix_rowscan(u32* hist, u32 nblk, int bits, u32* totals, hipStream_t st) {
  hipLaunchKernelGGL(radix_rowscan_kernel, dim3(1u << bits), dim3(256), 0, st, hist, nblk, totals);
  return hipGetLastError();
}

template <bool HI, bool FULL, int OWN = 0>
static hipError_t launch_radix_scatter_t(const void* in, void* out, u32 n, u32 n_full, int shift,
                                         int bits, u32 rows_per_block, u32 worker_base, u32 grid,
                                         const u32* hist_scanned, const u32* totals, u32 nblk,
                                         u64* offsets_out, hipStream_t st, const OwnerFn* own = nullptr) {
  static SmemAttrOnce attr_once;
  if (hipError_t e = ensure_max_smem(attr_once, reinterpret_cast<const void*>(radix_scatter_kernel<HI, FULL, OWN>), (size_t)sizeof(ScatterSmem)); e != hipSuccess) return e;
  OwnerFn f;
  std::memset(&f, 0, sizeof(f));
  if (own) f = *own;
  hipLaunchKernelGGL((radix_scatter_kernel<HI, FULL, OWN>), dim3(grid), dim3(RP_THREADS),
                     sizeof(ScatterSmem), st, static_cast<const Tup*>(in), static_cast<Tup*>(out), n,
                     n_full, shift, bits, rows_per_block, worker_base, hist_scanned, totals, nblk,
                     offsets_out, f);
  return hipGetLastError();
}

template <int THREADS, int MAXD, bool HI, bool FULL>
static hipError_t launch_wc_t(const void* in, void* out, u32 n, u32 n_full, int shift, int bits,
                              u32 rows_per_block, u32 worker_base, u32 grid, const u32* hist_scanned,
                              const u32* totals, u32 nblk, u64* offsets_out, hipStream_t st) {
  typedef WcSmem<THREADS, MAXD> Smem;
  static SmemAttrOnce attr_once;
  if (hipError_t e = ensure_max_smem(attr_once, reinterpret_cast<const void*>(radix_scatter_wc_kernel<THREADS, MAXD, HI, FULL>), (size_t)sizeof(Smem)); e != hipSuccess) return e;
  hipLaunchKernelGGL((radix_scatter_wc_kernel<THREADS, MAXD, HI, FULL>), dim3(grid), dim3(THREADS),
                     sizeof(Smem), st, static_cast<const Tup*>(in), static_cast<Tup*>(out), n, n_full,
                     shift, bits, rows_per_block, worker_base, hist_scanned, totals, nblk, offsets_out);
  return hipGetLastError();
}

template <int THREADS, int MAXD>
static hipError_t launch_wc(const void* in, void* out, u32 n, int shift, int bits,
                            const u32* hist_scanned, const u32* totals, u32 nblk, u32 rows_per_block,
                            u64* offsets_out, hipStream_t st) {
  const u32 tile = THREADS * WC_ITEMS, n_full = (n / tile) * tile;
  const bool tail = n_full != n, hi = shift >= 32;
  const u32 k = nblk - (tail ? 1 : 0);
  hipError_t e = hipSuccess;
  if (n_full) {
    e = hi ? launch_wc_t<THREADS, MAXD, true, true>(in, out, n, n_full, shift, bits, rows_per_block, 0,
                                                    k, hist_scanned, totals, nblk, offsets_out, st)
           : launch_wc_t<THREADS, MAXD, false, true>(in, out, n, n_full, shift, bits, rows_per_block,
                                                     0, k, hist_scanned, totals, nblk, offsets_out, st);
    if (e != hipSuccess) return e;
  }
  if (tail) {
    const u32 wb = n_full ? k : 0;
    e = hi ? launch_wc_t<THREADS, MAXD, true, false>(in, out, n, n_full, shift, bits, rows_per_block,
                                                     wb, 1, hist_scanned, totals, nblk, offsets_out, st)
           : launch_wc_t<THREADS, MAXD, false, false>(in, out, n, n_full, shift, bits, rows_per_block,
                                                      wb, 1, hist_scanned, totals, nblk, offsets_out, st);
  }
  return e;
}

hipError_t launch_radix_scatter(const void* in, void* out, u32 n, int variant, int shift, int bits,
                                const u32* hist_scanned, const u32* totals, u32 nblk,
                                u32 rows_per_block, u64* offsets_out, hipStream_t st) {
  const int kind = scatter_kind(bits, variant);
  if (kind == 1)
    return launch_wc<512, 256>(in, out, n, shift, bits, hist_scanned, totals, nblk, rows_per_block,
                               offsets_out, st);
  if (kind == 2)
    return launch_wc<1024, 512>(in, out, n, shift, bits, hist_scanned, totals, nblk, rows_per_block,
                                offsets_out, st);
  const u32 n_full = (n / RP_TILE) * RP_TILE;
  const bool tail = n_full != n;
  const u32 k = nblk - (tail ? 1 : 0);  // workers owning whole tiles
  const bool hi = shift >= 32;
  hipError_t e = hipSuccess;
  if (n_full) {
    e = hi ? launch_radix_scatter_t<true, true>(in, out, n, n_full, shift, bits, rows_per_block, 0, k,
                                                hist_scanned, totals, nblk, offsets_out, st)
           : launch_radix_scatter_t<false, true>(in, out, n, n_full, shift, bits, rows_per_block, 0,
                                                 k, hist_scanned, totals, nblk, offsets_out, st);
    if (e != hipSuccess) return e;
  }
  if (tail) {
    const u32 wb = n_full ? k : 0;
    e = hi ? launch_radix_scatter_t<true, false>(in, out, n, n_full, shift, bits, rows_per_block, wb,
                                                 1, hist_scanned, totals, nblk, offsets_out, st)
           : launch_radix_scatter_t<false, false>(in, out, n, n_full, shift, bits, rows_per_block,
                                                  wb, 1, hist_scanned, totals, nblk, offsets_out, st);
  }
  return e;
}

// The owner split of the multi-GPU exchange: plain staged scatter (fan-out <= 16) on owner_digit<mode>.
hipError_t launch_owner_scatter(const void* in, void* out, u32 n, int bits, const OwnerFn& own,
                                const u32* hist_scanned, const u32* totals, u32 nblk, u32 rows_per_block,
                                u64* offsets_out, hipStream_t st) {
  const u32 n_full = (n / RP_TILE) * RP_TILE;
  const bool tail = n_full != n;
  const u32 k = nblk - (tail ? 1 : 0);
  hipError_t e = hipSuccess;
  if (n_full) {
    e = own.mode == 2 ? launch_radix_scatter_t<false, true, 2>(in, out, n, n_full, 0, bits, rows_per_block, 0, k,
                                                              hist_scanned, totals, nblk, offsets_out, st, &own)
                      : launch_radix_scatter_t<false, true, 1>(in, out, n, n_full, 0, bits, rows_per_block, 0, k,
                                                              hist_scanned, totals, nblk, offsets_out, st, &own);
    if (e != hipSuccess) return e;
  }
  if (tail) {
    const u32 wb = n_full ? k : 0;
    e = own.mode == 2 ? launch_radix_scatter_t<false, false, 2>(in, out, n, n_full, 0, bits, rows_per_block, wb, 1,
                                                               hist_scanned, totals, nblk, offsets_out, st, &own)
                      : launch_radix_scatter_t<false, false, 1>(in, out, n, n_full, 0, bits, rows_per_block, wb, 1,
                                                               hist_scanned, totals, nblk, offsets_out, st, &own);
  }
  return e;
}

hipError_t launch_part_offsets(const void* a, u32 n, int low, int bits, u32* off, hipStream_t st) {
  u32 P = 1u << bits;
  hipLaunchKernelGGL(part_offsets_kernel, dim3((P + 1 + 255) / 256), dim3(256), 0, st,
                     static_cast<const Tup*>(a), n, low, bits, off, P);
  return hipGetLastError();
}

}  // namespace hmj

#ifdef HMJ_DEV
extern "C" int hmj_dev_set_b_addr_alt(int on) {
  hmj::g_b_addr_alt = on != 0;
  return 0;
}
extern "C" int hmj_dev_set_b_ablate(unsigned v) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(hmj::g_b_ablate), &v, sizeof(v));
}
#endif
#ifdef HMJ_STAMPS
// developer builds only: read (and clear) the phase stamps summed by the write-combining scatter kernels
extern "C" int hmj_dev_stamps(unsigned long long out[32], int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return -4;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(hmj::g_wc_stamps), 32 * sizeof(unsigned long long)) != hipSuccess) return -4;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(hmj::g_wc_stamps), z, sizeof(z)) != hipSuccess) return -4;
  }
  return 0;
}
#endif
