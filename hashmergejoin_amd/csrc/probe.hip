// Bucket-local build + probe for gfx950.  One workgroup per radix partition (grid-stride):
//   build : the partition's R rows are copied into LDS and inserted into a chained hash table of
//           DISTINCT keys (find-or-insert with one ds compare-and-swap; a row whose key is already
//           there is folded into / listed under that key's first entry) -- the GPU form of
//           tables[p].insert(item), partitioned_hash.h:166-170 (partitioned_table_worker) / :173-215;
//   probe : the partition's S rows stream through in coalesced 16-byte loads and walk the chain
//           -- the loop of hashjoin_bench.cc:92-96, which only ever meets table p with bucket p.
// The rows it yields are the (key, rval, sval) of HashMergeJoin::iterator::operator*
// (hashjoin.h:168-173); count mode computes the reduction hashjoin_bench.cc:131-133 performs.
// Build partitions larger than PB_CAP rows (skew) are processed in LDS-sized chunks, re-streaming
// the probe partition per chunk.
#include "hmj_dev.h"
#include "hmj_launch.h"
#include <type_traits>

namespace hmj {

constexpr u32 PB_NB = 1u << PB_LOG_NB;
constexpr u32 NIL = 0xFFFFu;
static_assert(PB_CAP < 0xFFFF, "chain links are 16-bit");

struct ProbeSmem {
  u64 key[PB_CAP];
  u64 val[PB_CAP];
  u32 head[PB_NB];
  u32 aux[PB_CAP];  // per distinct key: row count | first row | list of its other rows (by kernel mode)
  u16 next[PB_CAP];
  u32 scratch[PB_THREADS / kWave + 1];
  u64 pcount;
  u64 red[8];
  u64 obase;                       // MODE 3: where the workgroup's rows of this round start in the result
  u32 wtot[PB_THREADS / kWave];    // MODE 3: rows each wave found in this round (bit 31: the wave had rows to look at)
  u32 nslot;
};
constexpr u32 PB_BATCH = 2 * PB_THREADS;  // build rows inserted per step
static_assert(PB_BATCH <= PB_CAP, "a batch must fit an empty table");

// The bucket of a key inside its partition: the key bits right under the partition bits, as many as the bucket count
// takes.  Keys with FEWER varying bits under the partition bits than that (dense integer ids: 2^22 ids in 2^14 partitions
// leave eight) have sh < 0: their low bits become the TOP bits of the bucket number, which keeps buckets in key order.
__device__ __forceinline__ u32 key_bucket(u64 key, int sh, u32 mask) {
  return (sh >= 0 ? (u32)(key >> sh) : (u32)(key << (-sh))) & mask;
}

__device__ __forceinline__ u32 tab_hash(u64 k) {
  u32 x = (u32)k ^ ((u32)(k >> 32) * 0x85EBCA6Bu);
  x *= 0x9E3779B1u;
  return x >> (32 - PB_LOG_NB);
}

// HMJ_FIRST_WINS when the build partition is processed in several chunks: chunks follow input order,
// so the first chunk that holds a probe row's key holds its first build row; the row is then marked
// in a bitmap (one bit per probe row) and skipped in later chunks.  Only the thread that owns the
// row ever tests or sets its bit; the load bypasses L1 because the earlier atomic went to L2.
__device__ __forceinline__ bool first_claim(u32* matched, bool multi, u32 row) {
  if (!multi) return true;
  u32* w = matched + (row >> 5);
  const u32 bit = 1u << (row & 31);
  if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) return false;
  atomicOr(w, bit);
  return true;
}

// MODE 0: count + sums.  MODE 1: also per-partition match counts.  MODE 2: write result columns.
// MODE 3 (round 4; probe side in the slabs of one pass, piece walk): count AND write in one pass -- an unordered result may
//         be placed anywhere, so a wave reserves its rows with one atomic add on the result cursor (accum[ACC_N]).
// FIRST: HMJ_FIRST_WINS.  EXTRA: HMJ_CHECKSUM / HMJ_SUM_PROBE accumulators.
// The table holds one chain entry per DISTINCT key (skewed build sides put hundreds of thousands of
// equal keys into one partition; chaining them all would make every probe of that bucket walk them).
// What an entry knows about the other rows with its key depends on what the mode needs:
//   AGG  (count modes without checksums): aux = number of rows, val = sum of their payloads;
//   FIRST: aux = smallest row position in the partition (first in input order; its payload is read
//          from the partition when a probe row pairs with it);
//   else : aux threads a list through the rows with that key (enumerated per match: output-sized work).
template <int MODE, bool FIRST, bool EXTRA>
__global__ __launch_bounds__(PB_THREADS, PB_THREADS / 256) void probe_kernel(ProbeArgs a) {
  constexpr bool AGG = !FIRST && !EXTRA && MODE != 2 && MODE != 3;
  constexpr bool PERSIST = AGG || FIRST;  // rows with a key already in the table take no slot
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  ProbeSmem& sm = *reinterpret_cast<ProbeSmem*>(smem_raw);
  const Tup* __restrict__ R = static_cast<const Tup*>(a.R);
  const Tup* __restrict__ S = static_cast<const Tup*>(a.S);
  const int tid = threadIdx.x, lane = tid & 63;

  u64 acc_n = 0, acc_r = 0, acc_s = 0, acc_x = 0, acc_m = 0, acc_p = 0;
  bool pfx_bad = false;
  if (MODE == 1 && tid == 0) sm.pcount = 0;

  // either all P*Q items, or (after the fast kernel) just the partitions it set aside
  const u64 items = a.item_list ? (u64)*a.n_item_list : (u64)a.P * a.Q;
  for (u64 wi = blockIdx.x; wi < items; wi += gridDim.x) {
    const u64 w = a.item_list ? (u64)a.item_list[wi] : wi;
    const u32 p = (u32)(w / a.Q), q = (u32)(w % a.Q);
    const u32 rb = a.r_off[p], nb = (a.r_end ? a.r_end[p] : a.r_off[p + 1]) - rb;
    u32 sb, np;  // this item's slice of partition p's probe rows
    // probe side left in the worker-private slabs of ONE slab pass (round 4, count modes): partition p is its s_wa pieces
    // (one per pass-A worker: count s_cnt[p * s_wa + j], rows [(p * s_wa + j) * s_cap, ...)), item (p, q) takes s_ppi of them
    u32 npieces = 0;
    u64 piece0 = 0;
    if (a.s_ppi) {
      piece0 = (u64)p * a.s_wa + (u64)q * a.s_ppi;
      npieces = a.s_wa - q * a.s_ppi < a.s_ppi ? a.s_wa - q * a.s_ppi : a.s_ppi;
      sb = 0;
      np = 1;  // (unknown without summing the piece counts; "some" is all the code below needs)
    } else if (a.s_cnt) {  // probe side in the slab layout (radix.hip): slice q is the partition's q-th piece
      sb = (u32)w * a.s_cap;
      np = a.s_cnt[w];
    } else {
      const u32 sb0 = a.s_off[p], np0 = (a.s_end ? a.s_end[p] : a.s_off[p + 1]) - sb0;
      const u32 lo = (u32)((u64)np0 * q / a.Q), hi = (u32)((u64)np0 * (q + 1) / a.Q);
      sb = sb0 + lo;
      np = hi - lo;
    }
    if (nb == 0 || np == 0) {
      if (EXTRA && MODE != 2 && npieces) {
        for (u32 pi = 0; pi < npieces; pi++) {
          const u32 cnt = a.s_cnt[piece0 + pi];
          const Tup* __restrict__ base = S + (piece0 + pi) * a.s_cap;
          for (u32 j = tid; j < cnt; j += PB_THREADS) acc_p += base[j].val;
        }
      } else if (EXTRA && MODE != 2)
        for (u32 j = tid; j < np; j += PB_THREADS) acc_p += S[sb + j].val;
      if (MODE == 1 && tid == 0) a.part_count[w] = 0;
      continue;
    }
    const bool multi = FIRST && nb > PB_CAP;  // first-wins across tables: remember paired probe rows
    u64 pc = 0;                                   // this thread's matches in partition p
    u64 run = (MODE == 2) ? a.part_out_off[w] : 0;  // next output row of this item

    // The build rows go into the table in batches; a table is probed (and cleared) when the next batch
    // might not fit.  PERSIST modes give a slot only to a key not yet in the table, so a partition
    // of few distinct keys -- the skewed case -- is one table however many rows it has.
    u32 c0 = 0;
    bool first_table = true;
    while (c0 < nb) {
      __syncthreads();
      for (u32 i = tid; i < PB_NB; i += PB_THREADS) sm.head[i] = NIL;
      if (tid == 0) sm.nslot = 0;
      __syncthreads();
      u32 used = 0;
      while (c0 < nb) {
        const u32 cn = (nb - c0 < PB_BATCH) ? nb - c0 : PB_BATCH;
        if (used + cn > PB_CAP) break;
        Tup t[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const u32 i = k * PB_THREADS + tid;
          if (i < cn) t[k] = R[(u64)rb + c0 + i];
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const u32 i = k * PB_THREADS + tid;
          bool valid = i < cn;
          const u64 key = t[k].key;
          u64 sumv = t[k].val;
          u32 cntv = 1;
          if (valid && a.pfx_shift && (key >> a.pfx_shift) != a.pfx_val) pfx_bad = true;
          if (PERSIST) {
            // a wave whose rows all carry one key (the hot key of a skewed partition) sends one lane
            const u64 vm = __ballot(valid);
            if (__popcll(vm) > 1) {
              const int leader = __ffsll((long long)vm) - 1;
              const u64 k0 = __shfl(key, leader, kWave);
              if (__all(!valid || key == k0)) {
                if (AGG) {
                  sumv = wave_sum_u64(valid ? sumv : 0);
                  cntv = (u32)__popcll(vm);
                }
                valid = lane == leader;  // FIRST: the leader holds the smallest position
              }
            }
          }
          if (valid) {
            const u32 pos = c0 + i;  // position in the partition = input order (stable partitioning)
            u32 slot = PERSIST ? NIL : used + i;
            if (!PERSIST) {
              sm.key[slot] = key;
              sm.val[slot] = sumv;
              sm.aux[slot] = NIL;
            }
            u32* hp = &sm.head[tab_hash(key)];
            u32 seen = NIL;  // entries from here on were already compared
            for (;;) {
              const u32 hd = __hip_atomic_load(hp, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
              u32 found = NIL;
              for (u32 n = hd; n != seen; n = sm.next[n])
                if (sm.key[n] == key) {
                  found = n;
                  break;
                }
              if (found != NIL) {
                if (AGG) {
                  atomicAdd(&sm.aux[found], cntv);
                  atomicAdd(reinterpret_cast<unsigned long long*>(&sm.val[found]), (unsigned long long)sumv);
                } else if (FIRST) {
                  if (pos < __hip_atomic_load(&sm.aux[found], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                    atomicMin(&sm.aux[found], pos);
                } else {
                  sm.aux[slot] = atomicExch(&sm.aux[found], slot);
                }
                break;
              }
              if (PERSIST && slot == NIL) {  // a key not seen so far: now it needs a slot
                slot = atomicAdd(&sm.nslot, 1u);
                sm.key[slot] = key;
                if (AGG) sm.val[slot] = sumv;
                sm.aux[slot] = AGG ? cntv : pos;
              }
              sm.next[slot] = (u16)hd;
              u32 expect = hd;
              if (__hip_atomic_compare_exchange_strong(hp, &expect, slot, __ATOMIC_RELEASE, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP))
                break;
              seen = hd;  // lost the race: only the entries pushed since then are new
            }
          }
        }
        c0 += cn;
        __syncthreads();
        used = PERSIST ? sm.nslot : used + cn;  // PERSIST: at most one slot per row, so <= used + cn
        if (PERSIST) __syncthreads();           // everyone has read nslot before the next batch bumps it
      }

      // ---- probe
      if (MODE != 2) {
        // one probe row against the table (row: its index in the probe relation -- first-wins' bitmap of paired rows)
        auto probe_row = [&](const Tup& tr, u32 row) __attribute__((always_inline)) {
              const u64 key = tr.key, sval = tr.val;
              if (a.pfx_shift && (key >> a.pfx_shift) != a.pfx_val) pfx_bad = true;
              if (EXTRA && first_table) acc_p += sval;
              u32 i = sm.head[tab_hash(key)];
              while (i != NIL && sm.key[i] != key) i = sm.next[i];
              u32 best = 0;
              bool hit = false;  // FIRST: `best` is a position in the partition, any 32-bit value is a valid one
              if (i != NIL) {
                hit = true;
                if (AGG) {
                  const u64 c = sm.aux[i];
                  pc += c;
                  acc_r += sm.val[i];
                  acc_s += sval * c;
                } else if (FIRST) {
                  best = sm.aux[i];
                } else {
                  do {  // every build row with this key
                    const u64 rval = sm.val[i];
                    pc++;
                    acc_r += rval;
                    acc_s += sval;
                    if (EXTRA) {
                      u64 m = tmix(key, rval, sval);
                      acc_x ^= m;
                      acc_m += m;
                    }
                    i = sm.aux[i];
                  } while (i != NIL);
                }
              }
              if (FIRST && hit && first_claim(a.matched, multi, row)) {
                const u64 rval = R[(u64)rb + best].val;  // the row itself is not kept in LDS
                pc++;
                acc_r += rval;
                acc_s += sval;
                if (EXTRA) {
                  u64 m = tmix(key, rval, sval);
                  acc_x ^= m;
                  acc_m += m;
                }
              }
        };
        if (MODE == 3) {
          // count + write: per group of 4 x 64 rows a wave finds every row's matches (m: how many, first: where their
          // list starts) and scans the counts slot by slot; the WORKGROUP then reserves the rows of its sixteen groups on the
          // cursor with ONE atomic add (same-address atomics cost 11 ns each: one per wave group was 3 of this kernel's
          // 3.6 ms at 2^26 rows) and every wave writes its share -- slot k's rows form one contiguous block, a lane's m
          // rows inside it are consecutive.  Waves walk their pieces as a state (pb, pi, j0) so that all sixteen reach the
          // round's two barriers the same number of times.
          constexpr u32 NW = PB_THREADS / kWave, PR = 4;  // (eight rows in flight spill here: 136-356 bytes of scratch per lane)
          const u32 wv = (u32)__builtin_amdgcn_readfirstlane(tid >> 6);
          u32 pb = 0, pi = wv, j0 = 0;
          u32 cnt_l = (u32)lane < npieces ? a.s_cnt[piece0 + lane] : 0u;
          for (;;) {
            u32 cnt = 0;
            bool have = false;
            while (pb < npieces) {  // (uniform per wave)
              const u32 pe = npieces - pb < (u32)kWave ? npieces - pb : (u32)kWave;
              if (pi < pe) {
                cnt = (u32)__builtin_amdgcn_readlane((int)cnt_l, (int)pi);
                if (j0 < cnt) { have = true; break; }
                pi += NW;
                j0 = 0;
              } else {
                pb += kWave;
                pi = wv;
                j0 = 0;
                if (pb < npieces) cnt_l = pb + (u32)lane < npieces ? a.s_cnt[piece0 + pb + lane] : 0u;
              }
            }
            Tup t[PR];
            u32 m[PR], first[PR], excl[PR], pre[PR];
            u32 total = 0;
            const u32 prow0 = (piece0 + pb + pi) * a.s_cap;  // (this piece's first slab slot; unused without a group)
            if (have) {
              const Tup* __restrict__ base = S + (u64)prow0;
#pragma unroll
              for (int k = 0; k < (int)PR; k++) {
                const u32 j = j0 + k * kWave + lane;
                t[k] = load_stream(&base[j < cnt ? j : cnt - 1]);
              }
#pragma unroll
              for (int k = 0; k < (int)PR; k++) {
                const u32 j = j0 + k * kWave + lane;
                m[k] = 0;
                first[k] = NIL;
                if (j < cnt) {
                  const u64 key = t[k].key;
                  if (a.pfx_shift && (key >> a.pfx_shift) != a.pfx_val) pfx_bad = true;
                  if (EXTRA && first_table) acc_p += t[k].val;
                  u32 i = sm.head[tab_hash(key)];
                  while (i != NIL && sm.key[i] != key) i = sm.next[i];
                  if (i != NIL) {
                    if (FIRST) {
                      first[k] = sm.aux[i];
                      m[k] = first_claim(a.matched, multi, prow0 + j) ? 1u : 0u;
                    } else {
                      first[k] = i;
                      for (u32 n = i; n != NIL; n = sm.aux[n]) m[k]++;
                    }
                  }
                }
                const u32 incl = wave_incl_scan_u32(m[k], lane);
                excl[k] = incl - m[k];
                pre[k] = total;
                total += (u32)__builtin_amdgcn_readlane((int)incl, 63);
              }
              j0 += kWave * PR;
            }
            if (lane == 0) sm.wtot[wv] = total | (have ? 0x80000000u : 0u);
            __syncthreads();
            const u32 wt = (u32)lane < NW ? sm.wtot[lane] : 0u;
            if (__ballot(wt & 0x80000000u) == 0) break;  // no wave had a group: the item is done (same answer in every wave)
            const u32 wn = wt & 0x7FFFFFFFu;
            const u32 wincl = wave_incl_scan_u32(wn, lane);
            const u32 all = (u32)__builtin_amdgcn_readlane((int)wincl, (int)NW - 1);
            const u32 mine = (u32)__shfl((int)(wincl - wn), (int)wv, kWave);
            if (tid == 0 && all) sm.obase = atomicAdd(reinterpret_cast<unsigned long long*>(&a.accum[ACC_N]), (unsigned long long)all);
            __syncthreads();
            if (total == 0) continue;  // (uniform per wave; every wave still takes both barriers of every round)
            const u64 ob = sm.obase + mine;
            if (ob + total > a.out_cap) {  // more result rows than the caller's columns hold (duplicate build keys): not written
              if (lane == 0) atomicOr(reinterpret_cast<unsigned long long*>(&a.accum[ACC_ERR]), (unsigned long long)ERR_FASTPATH);
              continue;
            }
#pragma unroll
            for (int k = 0; k < (int)PR; k++) {
              if (m[k]) {
                u64 o = ob + pre[k] + excl[k];
                const u64 key = t[k].key, sval = t[k].val;
                if (FIRST) {
                  const u64 rval = R[(u64)rb + first[k]].val;
                  a.out_key[o] = key;
                  a.out_rval[o] = rval;
                  a.out_sval[o] = sval;
                  acc_r += rval;
                  acc_s += sval;
                  if (EXTRA) {
                    const u64 mx = tmix(key, rval, sval);
                    acc_x ^= mx;
                    acc_m += mx;
                  }
                } else {
                  for (u32 n = first[k]; n != NIL; n = sm.aux[n]) {
                    const u64 rval = sm.val[n];
                    a.out_key[o] = key;
                    a.out_rval[o] = rval;
                    a.out_sval[o] = sval;
                    o++;
                    acc_r += rval;
                    acc_s += sval;
                    if (EXTRA) {
                      const u64 mx = tmix(key, rval, sval);
                      acc_x ^= mx;
                      acc_m += mx;
                    }
                  }
                }
              }
            }
          }
        } else if (npieces) {
          // a wave per piece (pieces are a few hundred to a few thousand contiguous rows), eight rows per lane in flight.
          // The piece counts come in 64 at a time, one per lane (a count read per piece put a dependent global load --
          // a microsecond -- in front of every few hundred rows), and are handed out by readlane.
          constexpr u32 NW = PB_THREADS / kWave, PR = EXTRA ? 4 : 8;  // (with the checksum accumulators eight rows in flight spill: 96 bytes of scratch per lane)
          const u32 wv = (u32)__builtin_amdgcn_readfirstlane(tid >> 6);
          for (u32 pb = 0; pb < npieces; pb += kWave) {
            const u32 cnt_l = pb + (u32)lane < npieces ? a.s_cnt[piece0 + pb + lane] : 0u;
            const u32 pe = npieces - pb < (u32)kWave ? npieces - pb : (u32)kWave;
            for (u32 pi = wv; pi < pe; pi += NW) {
              const u32 cnt = (u32)__builtin_amdgcn_readlane((int)cnt_l, (int)pi);
              const Tup* __restrict__ base = S + (piece0 + pb + pi) * a.s_cap;
              for (u32 j0 = 0; j0 < cnt; j0 += kWave * PR) {
                Tup t[PR];
#pragma unroll
                for (int k = 0; k < (int)PR; k++) {
                  const u32 j = j0 + k * kWave + lane;
                  t[k] = load_stream(&base[j < cnt ? j : cnt - 1]);  // (clamped, unpredicated: the loads issue back to back)
                }
#pragma unroll
                for (int k = 0; k < (int)PR; k++) {
                  const u32 j = j0 + k * kWave + lane;
                  if (j < cnt) probe_row(t[k], (u32)((piece0 + pb + pi) * a.s_cap) + j);  // (row = its slot in the slab buffer: first-wins' bitmap)
                }
              }
            }
          }
        } else {
        for (u32 j0 = 0; j0 < np; j0 += PB_THREADS * 4) {
          Tup t[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            u32 j = j0 + k * PB_THREADS + tid;
            if (j < np) t[k] = S[(u64)sb + j];
          }
#pragma unroll
          for (int k = 0; k < 4; k++) {
            u32 j = j0 + k * PB_THREADS + tid;
            if (j < np) probe_row(t[k], sb + j);
          }
        }
        }
      } else {
        for (u32 j0 = 0; j0 < np; j0 += PB_THREADS) {
          const u32 j = j0 + tid;
          u64 key = 0, sval = 0;
          u32 m = 0, first = NIL;
          if (j < np) {
            Tup t = S[(u64)sb + j];
            key = t.key;
            sval = t.val;
            u32 i = sm.head[tab_hash(key)];
            while (i != NIL && sm.key[i] != key) i = sm.next[i];
            if (i != NIL) {
              if (FIRST) {
                first = sm.aux[i];
                m = first_claim(a.matched, multi, sb + j) ? 1u : 0u;
              } else {
                first = i;
                for (u32 n = i; n != NIL; n = sm.aux[n]) m++;
              }
            }
          }
          u32 tot;
          const u32 off = block_excl_scan_u32<PB_THREADS>(m, sm.scratch, &tot);
          if (m) {
            u64 o = run + off;
            if (m == 1) {
              a.out_key[o] = key;
              a.out_rval[o] = FIRST ? R[(u64)rb + first].val : sm.val[first];
              a.out_sval[o] = sval;
            } else {
              for (u32 n = first; n != NIL; n = sm.aux[n]) {
                a.out_key[o] = key;
                a.out_rval[o] = sm.val[n];
                a.out_sval[o] = sval;
                o++;
              }
            }
          }
          run += tot;
        }
      }
      first_table = false;
    }

    if (MODE != 3) acc_n += pc;  // (MODE 3: accum[ACC_N] is the output cursor, complete as it is)
    if (MODE == 1) {
      u64 ws = wave_sum_u64(pc);
      if (lane == 0 && ws) atomicAdd(&sm.pcount, ws);
      __syncthreads();
      if (tid == 0) {
        a.part_count[w] = sm.pcount;
        sm.pcount = 0;
      }
    }
  }

  if (MODE != 2 && __any(pfx_bad) && (tid & 63) == 0) atomicOr(&a.accum[ACC_ERR], ERR_PREFIX);
  if (MODE != 2) {
    __syncthreads();
    if (tid < 8) sm.red[tid] = 0;
    __syncthreads();
    const u64 v[6] = {acc_n, acc_r, acc_s, acc_x, acc_m, acc_p};  // ACC_N .. ACC_SUM_P order
    block_accumulate(sm.red, a.accum, v, 1u << ACC_XOR);
  }
}

// ---------------------------------------------------------------------------------------------
// Fast path of count mode (the headline: hashjoin_bench.cc:131-133 reduction, no flags).
// Same build+probe, software-pipelined so HBM latency is never exposed: while partition p is
// probed, the build rows of the workgroup's NEXT partition are already in flight, and p's probe
// rows were requested before its table was built.  Chain heads carry a 16-bit epoch, so the table
// is never cleared between partitions.  3 workgroups/CU (53 KiB LDS each).
// Partitions that do not fit the register pipeline (more than FP_CAP build or probe rows: skew)
// are appended to `irregular` and joined afterwards by the generic kernel.
// ---------------------------------------------------------------------------------------------
constexpr int FP_ROWS = 5;  // rows per thread per side
// Ablation switches of the pipelined kernel (1 = loads only, 2 = no chain walk) exist in developer builds
// only (-DHMJ_DEV, read from HMJ_DEBUG_ABLATE); the release library compiles them out.
#ifdef HMJ_DEV
#define HMJ_ABLATE(bit) ((a.debug & (bit)) != 0u)
#else
#define HMJ_ABLATE(bit) false
#endif
#ifndef BIG_LOG_NB
#define BIG_LOG_NB 13
#endif
#ifndef HMJ_PROBE_PIECES
#define HMJ_PROBE_PIECES 1  // slab layout, count mode: rows are taken piece by piece (no flattening arithmetic)
#endif

template <int THREADS, int LOG_NB>
struct FastSmem {
  static constexpr int CAP = THREADS * FP_ROWS;
  u64 key[CAP];
  u64 val[CAP];
  u32 head[1u << LOG_NB];  // epoch << 16 | row
  u16 next[CAP];
  u64 red[8];
  u32 itemcnt[2];  // per-partition match count, double-buffered by partition parity
  u64 wscan[THREADS / kWave + 1];  // OUT == 1: packed 64-bit block scan
  u32 wdup[THREADS / kWave + 1];   // OUT == 1: wave saw a probe row with two matches
};

template <int LOG_NB>
__device__ __forceinline__ u32 fast_hash(u64 k) {
  u32 x = (u32)k ^ ((u32)(k >> 32) * 0x85EBCA6Bu);
  x *= 0x9E3779B1u;
  return x >> (32 - LOG_NB);
}

template <int THREADS, int ROWS = FP_ROWS>
__device__ __forceinline__ void fp_load(Tup (&t)[ROWS], const Tup* __restrict__ base, u32 n,
                                        int tid) {
  // unpredicated loads (index clamped into the partition) so all five issue back to back
#pragma unroll
  for (int k = 0; k < ROWS; k++) {
    u32 i = k * THREADS + tid;
    t[k] = base[i < n ? i : n - 1];
  }
}

// <512, 11>: partitions up to 2560 rows, 53 KiB LDS, 3 workgroups/CU (the planner's default size)
// <1024, 12>: partitions up to 5120 rows, 106 KiB LDS, 1 workgroup/CU
// Slab layout loader: the partition is 4 pieces of up to `cap` rows; p1..p3 = prefix sums of the
// first three piece counts (wave-uniform).
static_assert(SLAB_KB == 4, "fp_load_slab / fp_load_pieces / slab_np_kernel unroll exactly four pieces per partition");
template <int THREADS, int ROWS = FP_ROWS>
__device__ __forceinline__ void fp_load_slab(Tup (&t)[ROWS], const Tup* __restrict__ base, u32 cap,
                                             u32 p1, u32 p2, u32 p3, u32 n, int tid) {
#pragma unroll
  for (int k = 0; k < ROWS; k++) {
    u32 i = k * THREADS + tid;
    i = i < n ? i : n - 1;
    const u32 j = (i >= p1) + (i >= p2) + (i >= p3);
    const u32 pre = j == 0 ? 0u : (j == 1 ? p1 : (j == 2 ? p2 : p3));
    t[k] = base[(u64)j * cap + (i - pre)];
  }
}

// Slab layout, count mode: which (thread, slot) holds which row of a partition does not matter to a count join,
// so instead of flattening the 4 pieces (three compares and selects per row) every quarter of the workgroup
// takes one piece: thread tid reads rows (tid mod T/4) + k * T/4 of piece tid / (T/4).  A wave lies inside one
// quarter, so its piece -- base and row count -- is wave-uniform and the address is base + row.  A piece may
// hold up to FP_ROWS * T/4 rows (1280; the slabs' capacity is mean + 8 sigma = 1304 at the headline size:
// beyond 1280 the partition is reported like any other that does not fit).
template <int THREADS>
__device__ __forceinline__ void fp_load_pieces(Tup (&t)[FP_ROWS], const Tup* __restrict__ base, u32 cap, u32 cnt_mine, int tid) {
  constexpr int QT = THREADS / 4;
  const u32 g = (u32)tid / QT, l = (u32)tid % QT;
  const Tup* __restrict__ piece = base + (u64)g * cap;
  const u32 last = cnt_mine ? cnt_mine - 1 : 0;
#pragma unroll
  for (int k = 0; k < FP_ROWS; k++) {
    const u32 r = l + (u32)k * QT;
    t[k] = piece[r < last ? r : last];  // clamped, unpredicated: the five loads issue back to back
  }
}

// PCOUNT: also store each partition's match count in a.part_count[p] (first pass of materialising).
// SLAB: the partitioned relations are in the histogram-free slab layout (radix.hip, slab kernels);
//       a partition that does not fit the pipeline raises ERR_SLAB (the caller re-runs the exact path).
// OUT == 1: "unique build keys" write mode for ordered joins.  Every probe row yields at most one
//       result row, so partition p's rows go to the slots of its own probe rows (base = a.s_off[p] /
//       a.item_base[p]; no count pass, no output atomics): (key, rval, sval) columns are written in probe
//       order, the row count goes to a.part_count[p].  A probe row with two matches (duplicate build
//       keys) or a partition that does not fit raises ERR_FASTPATH: the caller re-runs the general path.
template <int THREADS, int LOG_NB, bool PCOUNT, bool SLAB, int OUT>
__global__ __launch_bounds__(THREADS, 4) void probe_count_fast_kernel(
    ProbeArgs a, u32* __restrict__ irregular, u32* __restrict__ n_irregular) {
  typedef FastSmem<THREADS, LOG_NB> Smem;
  constexpr u32 CAP = Smem::CAP, NB = 1u << LOG_NB;
  constexpr bool PIECES = SLAB && OUT == 0 && !PCOUNT && HMJ_PROBE_PIECES;  // rows are taken piece by piece (see fp_load_pieces)
  constexpr u32 QT = THREADS / 4, PIECE_CAP = FP_ROWS * QT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem& sm = *reinterpret_cast<Smem*>(smem_raw);
  const Tup* __restrict__ R = static_cast<const Tup*>(a.R);
  const Tup* __restrict__ S = static_cast<const Tup*>(a.S);
  const u32* __restrict__ r_off = a.r_off;
  const u32* __restrict__ s_off = a.s_off;
  // end of partition p: the next partition's start, unless oversized probe partitions were split into
  // virtual partitions that share a build range (then explicit end arrays are passed)
  const u32* __restrict__ r_end = (SLAB || a.r_end == nullptr) ? a.r_off + 1 : a.r_end;
  const u32* __restrict__ s_end = (SLAB || a.s_end == nullptr) ? a.s_off + 1 : a.s_end;
  const u32 P = a.P;
  const int tid = threadIdx.x;
  u64 acc_n = 0, acc_r = 0, acc_s = 0, acc_x = 0, acc_m = 0, acc_p = 0;
  bool pfx_bad = false, giveup = false;
  for (u32 i = tid; i < NB; i += THREADS) sm.head[i] = 0;
  if (tid < 2) sm.itemcnt[tid] = 0;
  u32 epoch = 0;
  u32 prev_p = 0xFFFFFFFFu, parity = 0;  // PCOUNT: partition whose count still sits in itemcnt[parity^1]

  // Pipeline (per workgroup, partitions p, p+G, p+2G, ...): when partition p is processed its build
  // rows (br) AND probe rows (pr) are already in registers; the probe rows of the next partition
  // (pq) are requested before p's table is built, its build rows right after p's rows went to LDS.
  u32 p = blockIdx.x;
  u32 rb = 0, nb = 0, sb = 0, np = 0;
  bool regular = false;
  Tup br[FP_ROWS], pr[FP_ROWS], pq[FP_ROWS];
  // slab layout: piece prefix sums of the current (r1..3, s1..3) and next (…n) partition
  u32 r1 = 0, r2 = 0, r3 = 0, s1 = 0, s2 = 0, s3 = 0, r1n = 0, r2n = 0, r3n = 0, s1n = 0, s2n = 0, s3n = 0;
  // PIECES: rows of this thread's quarter's piece, current (rm, sm_) and next (rmn, smn) partition
  u32 rm = 0, sm_ = 0, rmn = 0, smn = 0;
  const u32 my_piece = (u32)tid / QT;
  bool slab_bad = false;
  if (p < P) {
    if (SLAB) {
      const u32* rc = a.r_cnt + (u64)p * SLAB_KB;
      const u32* sc = a.s_cnt + (u64)p * SLAB_KB;
      r1 = rc[0]; r2 = r1 + rc[1]; r3 = r2 + rc[2]; nb = r3 + rc[3];
      s1 = sc[0]; s2 = s1 + sc[1]; s3 = s2 + sc[2]; np = s3 + sc[3];
      if (nb > CAP || np > CAP) slab_bad = true;
      if (PIECES) {  // (scalar loads + selects: indexing by the piece number would be a per-lane global load)
        rm = my_piece == 0 ? rc[0] : my_piece == 1 ? rc[1] : my_piece == 2 ? rc[2] : rc[3];
        sm_ = my_piece == 0 ? sc[0] : my_piece == 1 ? sc[1] : my_piece == 2 ? sc[2] : sc[3];
        if (rc[0] > PIECE_CAP || rc[1] > PIECE_CAP || rc[2] > PIECE_CAP || rc[3] > PIECE_CAP || sc[0] > PIECE_CAP ||
            sc[1] > PIECE_CAP || sc[2] > PIECE_CAP || sc[3] > PIECE_CAP) {
          slab_bad = true;
          nb = CAP + 1;  // not "regular"
        }
      }
    } else {
      rb = r_off[p]; nb = r_end[p] - rb;
      sb = s_off[p]; np = s_end[p] - sb;
    }
    regular = nb && np && nb <= CAP && np <= CAP;
    if (regular) {
      if (PIECES) {
        fp_load_pieces<THREADS>(br, R + (u64)p * SLAB_KB * a.r_cap, a.r_cap, rm, tid);
        fp_load_pieces<THREADS>(pr, S + (u64)p * SLAB_KB * a.s_cap, a.s_cap, sm_, tid);
      } else if (SLAB) {
        fp_load_slab<THREADS>(br, R + (u64)p * SLAB_KB * a.r_cap, a.r_cap, r1, r2, r3, nb, tid);
        if (OUT == 0) fp_load_slab<THREADS>(pr, S + (u64)p * SLAB_KB * a.s_cap, a.s_cap, s1, s2, s3, np, tid);
      } else {
        fp_load<THREADS>(br, R + rb, nb, tid);
        if (OUT == 0) fp_load<THREADS>(pr, S + sb, np, tid);
      }
    }
  }
  // The prefetches of the NEXT partition are issued unconditionally: when there is no next partition, or it does
  // not fit the pipeline, the same five loads read one dummy row (the accumulator block) instead.  Inside an
  // `if` the compiler cannot count them, and every wait for older rows becomes s_waitcnt vmcnt(0) -- i.e. the
  // probe walk of partition p waited for partition p + G's rows, which had only just been requested.
  const Tup* __restrict__ dummy = reinterpret_cast<const Tup*>(a.accum);
  auto load_build = [&](Tup (&t)[FP_ROWS], bool ok, u32 pp, u32 rb_, u32 nb_, u32 q1, u32 q2, u32 q3, u32 mine) {
    if (PIECES)
      fp_load_pieces<THREADS>(t, ok ? R + (u64)pp * SLAB_KB * a.r_cap : dummy, ok ? a.r_cap : 0u, ok ? mine : 1u, tid);
    else if (SLAB)
      fp_load_slab<THREADS>(t, ok ? R + (u64)pp * SLAB_KB * a.r_cap : dummy, ok ? a.r_cap : 0u, ok ? q1 : 1u, ok ? q2 : 1u,
                            ok ? q3 : 1u, ok ? nb_ : 1u, tid);
    else
      fp_load<THREADS>(t, ok ? R + rb_ : dummy, ok ? nb_ : 1u, tid);
  };
  auto load_probe = [&](Tup (&t)[FP_ROWS], bool ok, u32 pp, u32 sb_, u32 np_, u32 q1, u32 q2, u32 q3, u32 mine) {
    if (PIECES)
      fp_load_pieces<THREADS>(t, ok ? S + (u64)pp * SLAB_KB * a.s_cap : dummy, ok ? a.s_cap : 0u, ok ? mine : 1u, tid);
    else if (SLAB)
      fp_load_slab<THREADS>(t, ok ? S + (u64)pp * SLAB_KB * a.s_cap : dummy, ok ? a.s_cap : 0u, ok ? q1 : 1u, ok ? q2 : 1u,
                            ok ? q3 : 1u, ok ? np_ : 1u, tid);
    else
      fp_load<THREADS>(t, ok ? S + sb_ : dummy, ok ? np_ : 1u, tid);
  };
  const int tid_outer = tid;
  while (p < P) {
    // write modes: keep the compiler from hoisting the per-row LDS / output addresses out of the partition loop
    // (it then spills them and reloads each behind s_waitcnt vmcnt(0): see probe_write_sorted_kernel)
    int tid = tid_outer;
    if constexpr (OUT != 0) asm volatile("" : "+v"(tid));
    // write mode, slab layout: this partition's first output slot through the SCALAR cache (item_base was written by an
    // earlier kernel).  As a vector load at its point of use it sat behind the prefetch of the next partition's build
    // rows in the vmcnt queue and the copy-out waited for them (s_waitcnt vmcnt(0); see probe_write_sorted_kernel).
    u64 out_base = (u64)sb;
    if constexpr (OUT == 1 && SLAB) {
      const u64* ibp = a.item_base + p;
      asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(out_base) : "s"(ibp) : "memory");
    }
    const u32 pn = __builtin_amdgcn_readfirstlane(p + gridDim.x);  // keep the offsets on the scalar path
    u32 rb2 = 0, nb2 = 0, sb2 = 0, np2 = 0;
    bool regular2 = false;
    if (pn < P) {
      if (SLAB) {
        const u32* rc = a.r_cnt + (u64)pn * SLAB_KB;
        const u32* sc = a.s_cnt + (u64)pn * SLAB_KB;
        r1n = rc[0]; r2n = r1n + rc[1]; r3n = r2n + rc[2]; nb2 = r3n + rc[3];
        s1n = sc[0]; s2n = s1n + sc[1]; s3n = s2n + sc[2]; np2 = s3n + sc[3];
        if (nb2 > CAP || np2 > CAP) slab_bad = true;
        if (PIECES) {
          rmn = my_piece == 0 ? rc[0] : my_piece == 1 ? rc[1] : my_piece == 2 ? rc[2] : rc[3];
          smn = my_piece == 0 ? sc[0] : my_piece == 1 ? sc[1] : my_piece == 2 ? sc[2] : sc[3];
          if (rc[0] > PIECE_CAP || rc[1] > PIECE_CAP || rc[2] > PIECE_CAP || rc[3] > PIECE_CAP || sc[0] > PIECE_CAP ||
              sc[1] > PIECE_CAP || sc[2] > PIECE_CAP || sc[3] > PIECE_CAP) {
            slab_bad = true;
            nb2 = CAP + 1;
          }
        }
      } else {
        rb2 = r_off[pn]; nb2 = r_end[pn] - rb2;
        sb2 = s_off[pn]; np2 = s_end[pn] - sb2;
      }
      regular2 = nb2 && np2 && nb2 <= CAP && np2 <= CAP;
    }
    if (OUT == 0) {
      load_probe(pq, regular2, pn, sb2, np2, s1n, s2n, s3n, smn);  // next partition's probe rows, a whole partition ahead
    } else if (regular) {  // write mode keeps fewer rows in flight (registers): this partition's probe rows
      if (SLAB)
        fp_load_slab<THREADS>(pr, S + (u64)p * SLAB_KB * a.s_cap, a.s_cap, s1, s2, s3, np, tid);
      else
        fp_load<THREADS>(pr, S + sb, np, tid);
    }
    if (regular) {
      lds_barrier();                   // everyone is done probing the previous table
      if (PCOUNT && prev_p != 0xFFFFFFFFu && tid == 0) {  // its count is complete now
        a.part_count[prev_p] = sm.itemcnt[parity ^ 1];
        sm.itemcnt[parity ^ 1] = 0;
      }
      if (PCOUNT) prev_p = 0xFFFFFFFFu;
      epoch++;
      if (epoch == 0x10000u) {         // 16-bit epoch wrapped: clear once
        for (u32 i = tid; i < NB; i += THREADS) sm.head[i] = 0;
        epoch = 1;
        lds_barrier();
      }
      const u32 tag = epoch << 16;
      if (!HMJ_ABLATE(1u)) {
        // build: the five exchanges are independent -> issue them together, then link
        u32 old[FP_ROWS];
#pragma unroll
        for (int k = 0; k < FP_ROWS; k++) {
          const u32 i = k * THREADS + tid;
          old[k] = tag | NIL;
          if (PIECES ? ((u32)tid % QT + (u32)k * QT < rm) : (i < nb)) {
            if (a.pfx_shift && (br[k].key >> a.pfx_shift) != a.pfx_val) pfx_bad = true;
            sm.key[i] = br[k].key;
            sm.val[i] = br[k].val;
            old[k] = atomicExch(&sm.head[fast_hash<LOG_NB>(br[k].key)], tag | i);
          }
        }
#pragma unroll
        for (int k = 0; k < FP_ROWS; k++) {
          const u32 i = k * THREADS + tid;
          if (PIECES ? ((u32)tid % QT + (u32)k * QT < rm) : (i < nb)) sm.next[i] = ((old[k] >> 16) == epoch) ? (u16)old[k] : (u16)NIL;
        }
      } else {  // developer builds only (HMJ_ABLATE): stream the rows, no LDS work
#pragma unroll
        for (int k = 0; k < FP_ROWS; k++) acc_r += br[k].key;
      }
      load_build(br, regular2, pn, rb2, nb2, r1n, r2n, r3n, rmn);  // next partition's build rows
      lds_barrier();                                          // table complete
      const u64 n_before = acc_n;
      if (!HMJ_ABLATE(1u)) {
        // probe: walk the five chains in lockstep so their LDS latencies overlap
        u32 cur[FP_ROWS];
        u32 cnt[FP_ROWS], first[FP_ROWS];  // OUT == 1
#pragma unroll
        for (int k = 0; k < FP_ROWS; k++) {
          const u32 j = k * THREADS + tid;
          cur[k] = NIL;
          cnt[k] = 0;
          first[k] = OUT == 2 ? NIL : 0;
          if (PIECES ? ((u32)tid % QT + (u32)k * QT < sm_) : (j < np)) {
            if (a.pfx_shift && (pr[k].key >> a.pfx_shift) != a.pfx_val) pfx_bad = true;
            if (OUT >= 1 && (a.extra & 1u)) acc_p += pr[k].val;
            const u32 hv = sm.head[fast_hash<LOG_NB>(pr[k].key)];
            cur[k] = ((hv >> 16) == epoch) ? (hv & 0xFFFFu) : NIL;
          }
        }
        for (; !HMJ_ABLATE(2u);) {
          bool any = false;
#pragma unroll
          for (int k = 0; k < FP_ROWS; k++) any |= (cur[k] != NIL);
          if (!__any(any)) break;
#pragma unroll
          for (int k = 0; k < FP_ROWS; k++) {
            if (cur[k] != NIL) {
              const u32 i = cur[k];
              const u64 kk = sm.key[i], vv = sm.val[i];  // val read unconditionally: one LDS
              cur[k] = sm.next[i];                       // round trip per chain step, not two
              if (kk == pr[k].key) {
                if (OUT == 2 && (a.extra & 2u)) {  // first wins: smallest position = first in input order
                  first[k] = i < first[k] ? i : first[k];
                } else {
                  acc_n++;
                  acc_r += vv;
                  acc_s += pr[k].val;
                  if (OUT == 1) {
                    if (cnt[k] == 0) first[k] = i;
                    cnt[k]++;
                  }
                  if (OUT >= 1 && (a.extra & 1u)) {
                    const u64 m = tmix(kk, vv, pr[k].val);
                    acc_x ^= m;
                    acc_m += m;
                  }
                }
              }
            }
          }
        }
        if (OUT == 2 && (a.extra & 2u)) {
#pragma unroll
          for (int k = 0; k < FP_ROWS; k++) {
            if (first[k] != NIL) {
              const u64 vv = sm.val[first[k]];
              acc_n++;
              acc_r += vv;
              acc_s += pr[k].val;
              if (a.extra & 1u) {
                const u64 m = tmix(pr[k].key, vv, pr[k].val);
                acc_x ^= m;
                acc_m += m;
              }
            }
          }
        }
        if (OUT == 1) {
          // compact this partition's result rows in probe order: one packed block scan gives every
          // thread the offsets of its five rows (12 bits per row slot k).  "A row matched twice" is
          // OR-ed per wave beside it (summing such flags in the packed word would wrap: 16 of them
          // already overflow bit 60).
          u64 packed = 0;
          bool cx = false;
#pragma unroll
          for (int k = 0; k < FP_ROWS; k++) {
            cx |= cnt[k] > 1;
            packed |= (u64)(cnt[k] ? 1u : 0u) << (12 * k);
          }
          const int lane = tid & 63, wv = tid >> 6;
          const bool wave_dup = __any(cx);
          u64 incl = packed;
#pragma unroll
          for (int o = 1; o < kWave; o <<= 1) {
            u64 t = __shfl_up(incl, o, kWave);
            if (lane >= o) incl += t;
          }
          if (lane == 63) {
            sm.wscan[wv] = incl;
            sm.wdup[wv] = wave_dup ? 1u : 0u;
          }
          lds_barrier();
          u64 pre = 0, tot = 0;
          u32 dup = 0;
#pragma unroll
          for (int q = 0; q < THREADS / kWave; q++) {
            const u64 ws = sm.wscan[q];
            dup |= sm.wdup[q];
            if (q < wv) pre += ws;
            tot += ws;
          }
          const u64 ex = pre + incl - packed;
          if (dup) {
            giveup = true;  // duplicate build keys here: not the unique-key case
          } else {
            u64 o = out_base;
            u32 total = 0;
#pragma unroll
            for (int k = 0; k < FP_ROWS; k++) {
              if (cnt[k]) {
                const u64 d = o + ((ex >> (12 * k)) & 0xFFFu);
                a.out_key[d] = pr[k].key;
                a.out_rval[d] = sm.val[first[k]];
                a.out_sval[d] = pr[k].val;
              }
              const u32 tk = (u32)((tot >> (12 * k)) & 0xFFFu);
              o += tk;
              total += tk;
            }
            if (tid == 0) a.part_count[p] = total;
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < FP_ROWS; k++) acc_r += pr[k].key;
      }
      if (PCOUNT) {
        const u32 mine = (u32)(acc_n - n_before);
        const u32 ws = (u32)wave_sum_u64(mine);
        if ((tid & 63) == 0 && ws) atomicAdd(&sm.itemcnt[parity], ws);
        prev_p = p;
        parity ^= 1;
      }
    } else {
      if ((PCOUNT || OUT == 1) && tid == 0 && !(nb && np)) a.part_count[p] = 0;  // empty side: no rows
      if (OUT == 1 && nb && np) giveup = true;  // does not fit the pipeline: general path
      if (OUT >= 1 && (a.extra & 1u) && !nb) {  // probe rows without a build partition still count in sum_probe_all
        if (SLAB) {
          const Tup* base = S + (u64)p * SLAB_KB * a.s_cap;
          for (u32 j = tid; j < np; j += THREADS) {
            const u32 pc = (j >= s1) + (j >= s2) + (j >= s3);
            const u32 pre = pc == 0 ? 0 : pc == 1 ? s1 : pc == 2 ? s2 : s3;
            acc_p += base[(u64)pc * a.s_cap + (j - pre)].val;
          }
        } else {
          for (u32 j = tid; j < np; j += THREADS) acc_p += S[sb + j].val;
        }
      }
      if (OUT != 1 && !SLAB && tid == 0 && nb && np) irregular[atomicAdd(n_irregular, 1u)] = p;
      load_build(br, regular2, pn, rb2, nb2, r1n, r2n, r3n, rmn);
    }
    if (OUT == 0) {
#pragma unroll
      for (int k = 0; k < FP_ROWS; k++) pr[k] = pq[k];
    }
    p = pn; rb = rb2; nb = nb2; sb = sb2; np = np2; regular = regular2;
    r1 = r1n; r2 = r2n; r3 = r3n; s1 = s1n; s2 = s2n; s3 = s3n;
    rm = rmn; sm_ = smn;
  }
  if (SLAB && slab_bad && tid == 0) atomicOr(&a.accum[ACC_ERR], ERR_SLAB);
  if (OUT == 1 && __any(giveup) && (tid & 63) == 0) atomicOr(&a.accum[ACC_ERR], ERR_FASTPATH);
  if (__any(pfx_bad) && (tid & 63) == 0) atomicOr(&a.accum[ACC_ERR], ERR_PREFIX);
  lds_barrier();
  if (PCOUNT && prev_p != 0xFFFFFFFFu && tid == 0) a.part_count[prev_p] = sm.itemcnt[parity ^ 1];
  if (tid < 8) sm.red[tid] = 0;
  lds_barrier();
  const u64 v[6] = {acc_n, acc_r, acc_s, acc_x, acc_m, acc_p};
  block_accumulate(sm.red, a.accum, v, 1u << ACC_XOR);
}

// ---------------------------------------------------------------------------------------------
// Ordered unique-key write in ONE pass: probe, sort and write (HMJ_ORDERED; build keys unique, probe keys unique).
//
// The two-step form (probe_count_fast_kernel<OUT=1> writes rows in probe order at the slots of their probe rows,
// order_kernel sorts every partition and closes the gaps) moves every result row twice: 32 B read + 24 B written,
// then 24 B read + 24 B written.  Here the result rows leave once, already sorted and dense:
//   * the partition's BUILD rows are bucket-sorted by key into LDS (bucket = the 12 key bits below the partition
//     bits, a counting pass + a rank among the handful of rows of a bucket): the sorted array IS the table --
//     a probe row scans its key's bucket, a contiguous run of about one row;
//   * a matching probe row sets its build row's bit in a bitmap (a bit set twice = two probe rows with that key:
//     not this kernel's case) and leaves its payload beside it; a popcount prefix of the bitmap then gives every
//     matched build row its rank among the partition's result rows -- in key order;
//   * the partition's first output row is the number of result rows of all partitions before it: partitions are
//     handed out by a ticket counter and chained by a decoupled look-back over their published counts (a
//     partition waits only for partitions with smaller tickets, all of which are running or done);
//   * copy-out by sorted build index: consecutive lanes write consecutive output rows.
// Anything else -- duplicate build or probe keys, a bucket of more than SW_MAXBUCKET rows (keys that do not vary
// in those 12 bits), a partition that does not fit -- raises ERR_SORTED: the caller re-runs the two-step form.
// ---------------------------------------------------------------------------------------------
constexpr int SW_LOGB = 12, SW_NB = 1 << SW_LOGB, SW_MAXBUCKET = 32;
constexpr u64 LB_AGG = 1ull << 62, LB_PFX = 2ull << 62, LB_MASK = (1ull << 62) - 1;
constexpr u32 LB_SPIN_LIMIT = 1u << 22;  // ~ seconds of polling: a predecessor that never publishes is a bug

template <int THREADS>
struct SortedSmem {
  static constexpr int CAP = THREADS * FP_ROWS, CAPB = CAP, LOGB = SW_LOGB;
  u64 key[CAP];   // build keys, grouped by bucket (buckets ascend; inside a bucket: order of arrival)
  u64 val[CAP];   // build payloads, same order
  u64 sval[CAP];  // the payload of the probe row that matched the build row of sorted rank i
  u32 cnt[SW_NB];
  u16 bstart[SW_NB + 2];
  u16 perm[CAP];  // where the build row of sorted rank i lies in key[] / val[] (written by the probe row that matched it)
  u32 mbits[2][CAP / 32];   // the build row of sorted rank i was matched (double-buffered by partition parity)
  u32 scratch[THREADS / kWave + 1];
  u32 tick[2];
  u32 flag;
  u64 obase;
  u64 red[8];
};

// FK = true: the probe keys may repeat (a foreign-key join: every build key is hit by f probe rows).  The build rows
// are then really sorted in LDS (bucket, then a rank among the handful of keys of a bucket: position = sorted rank),
// a build row counts its matches (the old count is the probe row's arrival number among the rows of its key), an
// exclusive scan of the counts in sorted order gives every key its run of output slots, and the probe rows drop
// payload and sorted rank there.  The copy-out walks the OUTPUT slots -- consecutive lanes = consecutive slots, so
// the lanes of a run sit side by side -- and every slot ranks its payload inside its run by reading the run once
// (the result order is (key, rval, sval), and rval is the same for the whole run): lanes of one run read the SAME
// LDS word in every step (a broadcast, no bank conflict), where round 2's row-order ranking made 64 lanes read 64
// unrelated runs.  The row goes straight to its final place in global memory; a run's rows permute inside the
// run's own 128-byte lines, so the stores coalesce as before.  Build rows: at most 4608 per partition (a foreign-key
// join has far fewer), 2048 buckets -- that is what fits beside the per-slot arrays.
constexpr int SWF_LOGB = 11, SWF_CAPB = 4608, SWF_MAXDUP = 1024;
// Two shapes: <1024 threads, 5 rows> -- 5120 probe rows, 4608 build rows, 2048 buckets, 152 KB: one workgroup per CU;
// <512 threads, 6 rows> -- 3072 probe rows, 2048 build rows, 1024 buckets, 76 KB: TWO workgroups per CU, for the
// partitions of a join with fan-out >= 6 (the planner sizes those for about 2048 probe rows: half of the big
// shape's thread slots would idle, and one workgroup per CU leaves the memory pipeline empty during every one of
// its barrier-separated phases -- with two, one computes while the other loads and stores).
constexpr int SWF_HALF_THREADS = 512, SWF_HALF_ROWS = 6, SWF_HALF_CAPB = 2048, SWF_HALF_LOGB = 10;
// <1024 threads, 6 rows>, 2560 build rows -- 6144 probe rows, 125 KB: the partitions of a 16-bit plan for fan-out 2.5 ... 30
// (4096 probe rows + 5 sigma, sigma = sqrt(fan-out x 4096) <= 390), which lets those joins partition on the slab path
// (two 8-bit passes, 32 B per row and pass) instead of 17 bits on the exact path (48 B).
constexpr int SWF_WIDE_ROWS = 6, SWF_WIDE_CAPB = 2560;
template <int THREADS, int ROWS = FP_ROWS, int CAPB_ = SWF_CAPB, int LOGB_ = SWF_LOGB>
struct SortedFkSmem {
  static constexpr int CAP = THREADS * ROWS, CAPB = CAPB_, LOGB = LOGB_;
  u64 key[CAPB];   // build keys in sorted order
  u64 val[CAPB];   // build payloads, same order
  u64 sval[CAP];   // the probe payload of every OUTPUT slot (before that: the build keys in bucket / arrival order)
  // matches of the build row of sorted rank i, then the first output slot of its run; [CAPB] = "some key has more than
  // one probe row", [CAPB + 1] = "some key has more than SWF_MAXDUP probe rows" (two words of the same array, so that one
  // loop over thread-relative addresses zeroes them all: as words of their own their constant addresses each took a
  // register, got spilled, and came back behind an s_waitcnt vmcnt(0) in wave 0)
  u32 mcnt[CAPB + 2];
  u16 srank[CAP];  // sorted rank of every output slot's build row
  u32 cnt[1 << LOGB_];
  u16 bstart[(1 << LOGB_) + 2];
  u32 scratch[THREADS / kWave + 1];
  u32 tick[2];
  u32 flag;
  u64 obase;
  u64 red[8];
  unsigned long long vmin, vmax;  // the partition's matched probe payloads (round 4: buckets over (key, payload position))
};
static_assert(sizeof(SortedFkSmem<SWF_HALF_THREADS, SWF_HALF_ROWS, SWF_HALF_CAPB, SWF_HALF_LOGB>) <= 80 * 1024, "two workgroups per CU");
static_assert(SWF_HALF_CAPB <= SWF_HALF_THREADS * SWF_HALF_ROWS, "tmpkey aliases sval");
static_assert(sizeof(SortedFkSmem<1024, SWF_WIDE_ROWS, SWF_WIDE_CAPB, SWF_LOGB>) <= 160 * 1024, "one workgroup per CU");
static_assert((SWF_HALF_CAPB + 2) * 4 + SWF_HALF_THREADS * SWF_HALF_ROWS * 2 >= (SWF_MAXDUP + 8) * 8, "the run ranking reads up to a run past sval[]");

// Publish partition p's row count and return the number of result rows of partitions 0 .. p-1.  Called by one
// whole wave: lane l inspects partition p-1-l, 64 predecessors per round, back to the nearest one that has
// published its inclusive prefix (typically within the few hundred partitions in flight).
__device__ __forceinline__ u64 lookback_publish(u64* __restrict__ state, u32 p, u64 total, int lane, bool* timeout) {
  unsigned long long* st = reinterpret_cast<unsigned long long*>(state);
  if (lane == 0) atomicExch(&st[p], LB_AGG | total);
  u64 excl = 0;
  for (long long base = (long long)p - 1; base >= 0; base -= kWave) {
    const long long q = base - lane;
    u64 v = LB_PFX;  // lanes before partition 0: "prefix 0"
    if (q >= 0) v = atomicAdd(&st[q], 0ull);  // (an RMW: always served by the coherent level)
    u32 spins = 0;
    while (__any((v >> 62) == 0)) {
      if (++spins > LB_SPIN_LIMIT) {  // a predecessor that never publishes: give up loudly instead of hanging
        *timeout = true;
        if ((v >> 62) == 0) v = LB_PFX;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
      if ((v >> 62) == 0) v = atomicAdd(&st[q], 0ull);
    }
    const u64 pfx = __ballot((v >> 62) == 2);
    const int first = pfx ? __ffsll((unsigned long long)pfx) - 1 : kWave;  // nearest published prefix
    excl += wave_sum_u64(lane <= first ? (v & LB_MASK) : 0ull);
    if (pfx) break;
  }
  if (lane == 0) atomicExch(&st[p], LB_PFX | ((excl + total) & LB_MASK));
  return excl;
}

// EXTRA: also accumulate the checksums and the sum of all probe payloads (HMJ_CHECKSUM / HMJ_SUM_PROBE); the
// operator's own calls do not ask for them, and without them the kernel keeps six 64-bit accumulators and the
// mixing arithmetic out of its registers (it ran at the 128-register limit of a 1024-thread workgroup with up to
// 96 bytes of scratch per lane; the scratch reloads showed as 38 % more fetched bytes than the rows themselves,
// profiles/r03a_ordered_unique_before_summary.txt).
template <int THREADS, bool SLAB, bool FK, bool EXTRA, int ROWS = FP_ROWS, int CAPB_ = SWF_CAPB, int LOGB_ = SWF_LOGB, bool PB = false>
__global__ __launch_bounds__(THREADS, 4) void probe_write_sorted_kernel(ProbeArgs a, u64* __restrict__ lookback, int key_low, bool chained) {
  static_assert(FK || ROWS == ROWS, "the bitmap form has one shape");
  typedef typename std::conditional<FK, SortedFkSmem<THREADS, ROWS, CAPB_, LOGB_>, SortedSmem<THREADS>>::type Smem;
  constexpr u32 CAP = Smem::CAP, CAPB = Smem::CAPB, WORDS = CAP / 32, NB = 1u << Smem::LOGB, BPT = NB / THREADS;
  static_assert(BPT * THREADS == NB && BPT >= 1, "whole buckets per thread in the scan");
  // The big shapes (one workgroup per CU) request the next partition's probe rows as soon as this partition's are
  // consumed; the small shape (two workgroups per CU: its sibling covers the latency) asks for a partition's probe
  // rows at the top of its own iteration and keeps those 24 registers free during the copy-out.
  constexpr bool EARLY_PROBE = ROWS == FP_ROWS;
  const int bsh = key_low - Smem::LOGB;  // bucket = the key bits right under the partition bits
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Smem& sm = *reinterpret_cast<Smem*>(smem_raw);
  const Tup* __restrict__ R = static_cast<const Tup*>(a.R);
  const Tup* __restrict__ S = static_cast<const Tup*>(a.S);
  const u32* __restrict__ r_off = a.r_off;
  const u32* __restrict__ s_off = a.s_off;
  const u32* __restrict__ r_end = (SLAB || a.r_end == nullptr) ? a.r_off + 1 : a.r_end;
  const u32* __restrict__ s_end = (SLAB || a.s_end == nullptr) ? a.s_off + 1 : a.s_end;
  const u32 P = a.P;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  unsigned long long* ticket = reinterpret_cast<unsigned long long*>(lookback);
  u64* state = lookback + 1;
  u64 acc_n = 0, acc_r = 0, acc_s = 0, acc_x = 0, acc_m = 0, acc_p = 0;
  bool pfx_bad = false, giveup = false, slab_bad = false, lb_timeout = false;
  u32 why = 0;  // why the kernel gave up (bits 7..11 of the error word, for HMJ_TRACE and the choice of the next form)

  // Partitions are handed out by a ticket counter only when the output offsets are chained (a partition may wait
  // only for smaller tickets, all running or done); otherwise a static stride does, and wave 0 does not have to wait
  // at the top of every partition for a returning atomic that sits behind its own stores of the previous one.
  if (tid == 0) sm.tick[0] = chained ? (u32)atomicAdd(ticket, 1ull) : blockIdx.x;
  if (tid < 8) sm.red[tid] = 0;
#pragma unroll
  for (u32 q = 0; q < BPT; q++) sm.cnt[tid * BPT + q] = 0;
  if constexpr (!FK) {
    if ((u32)tid < WORDS) sm.mbits[0][tid] = 0;
  }
  if (tid == 0) sm.flag = 0;
  __syncthreads();
  // (wave-uniform, and known to be: everything indexed by p -- piece counts, item_base[p] -- then takes the scalar
  //  path.  As a vector load, item_base[p] sat behind the prefetches of the next partition's rows in the vmcnt queue
  //  and the copy-out waited for all of them.)
  u32 p = (u32)__builtin_amdgcn_readfirstlane((int)sm.tick[0]), par = 0;
  u32 rb = 0, nb = 0, sb = 0, np = 0;
  u32 r1 = 0, r2 = 0, r3 = 0, s1 = 0, s2 = 0, s3 = 0, r1n = 0, r2n = 0, r3n = 0, s1n = 0, s2n = 0, s3n = 0;
  bool regular = false;
  Tup br[ROWS], pr[ROWS];
  const Tup* __restrict__ dummy = reinterpret_cast<const Tup*>(a.accum);
  if (p < P) {
    if (SLAB) {
      const u32* rc = a.r_cnt + (u64)p * SLAB_KB;
      const u32* sc = a.s_cnt + (u64)p * SLAB_KB;
      r1 = rc[0]; r2 = r1 + rc[1]; r3 = r2 + rc[2]; nb = r3 + rc[3];
      s1 = sc[0]; s2 = s1 + sc[1]; s3 = s2 + sc[2]; np = s3 + sc[3];
      if (nb > CAPB || np > CAP) slab_bad = true;
    } else {
      rb = r_off[p]; nb = r_end[p] - rb;
      sb = s_off[p]; np = s_end[p] - sb;
    }
    regular = nb && np && nb <= CAPB && np <= CAP;
    // the first partition's rows, both sides (unconditional loads: see probe_count_fast_kernel)
    if (SLAB) {
      fp_load_slab<THREADS, ROWS>(br, regular ? R + (u64)p * SLAB_KB * a.r_cap : dummy, regular ? a.r_cap : 0u, regular ? r1 : 1u,
                            regular ? r2 : 1u, regular ? r3 : 1u, regular ? nb : 1u, tid);
      if (EARLY_PROBE)
        fp_load_slab<THREADS, ROWS>(pr, regular ? S + (u64)p * SLAB_KB * a.s_cap : dummy, regular ? a.s_cap : 0u, regular ? s1 : 1u,
                              regular ? s2 : 1u, regular ? s3 : 1u, regular ? np : 1u, tid);
    } else {
      fp_load<THREADS, ROWS>(br, regular ? R + rb : dummy, regular ? nb : 1u, tid);
      if (EARLY_PROBE) fp_load<THREADS, ROWS>(pr, regular ? S + sb : dummy, regular ? np : 1u, tid);
    }
  }
  const int tid_outer = tid;
  while (p < P) {
    // The thread index as the loop body sees it is opaque to the compiler: otherwise it hoists every per-row LDS
    // address (&sm.sval[k * THREADS + tid], &sm.perm[...], ...) out of the partition loop, runs out of registers,
    // spills them, and reloads each one in the copy-out behind an s_waitcnt vmcnt(0) -- i.e. every output row waited
    // for all stores and prefetches in flight (found in the ISA; profiles/r03a_ordered_unique_before_summary.txt:
    // 71 % of the wave cycles parked, 38 % more bytes fetched than the rows themselves).
    int tid = tid_outer;
    asm volatile("" : "+v"(tid));
    // (the same for the few words single lanes write at constant addresses -- the flag, this partition's row count:
    //  indexed by an opaque zero their addresses are computed where they are used, not kept, spilled and reloaded
    //  behind s_waitcnt vmcnt(0))
    u32 zero = 0;
    asm volatile("" : "+v"(zero));
    if (chained && tid == 0) sm.tick[par ^ 1] = (u32)atomicAdd(ticket, 1ull);  // the partition after this one
    // this partition's first output slot, through the SCALAR cache (item_base was written by an earlier kernel).
    // The compiler cannot prove it invariant and would use a vector load, and vector-memory operations complete in
    // order: the copy-out's first use of it then waits (s_waitcnt vmcnt(0)) for the prefetches of the next
    // partition's rows that were requested in between.
    u64 ob = (u64)sb;
    if (SLAB) {
      const u64* ibp = a.item_base + p;
      asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ob) : "s"(ibp) : "memory");
    }
    if (!EARLY_PROBE) {  // this partition's probe rows (unconditional loads: see probe_count_fast_kernel)
      if (SLAB)
        fp_load_slab<THREADS, ROWS>(pr, regular ? S + (u64)p * SLAB_KB * a.s_cap : dummy, regular ? a.s_cap : 0u, regular ? s1 : 1u,
                              regular ? s2 : 1u, regular ? s3 : 1u, regular ? np : 1u, tid);
      else
        fp_load<THREADS, ROWS>(pr, regular ? S + sb : dummy, regular ? np : 1u, tid);
    }
    // (no barrier here: the first LDS arrays this partition writes -- cnt, then bstart -- are not read by the
    //  previous partition's copy-out, and two barriers lie between here and the first write to sval / key / val)
    u32 ha[ROWS];  // build row k: bucket | arrival number in the bucket << 12
    if (regular) {
      // count the build rows per bucket; the old count is the row's arrival number inside its bucket
#pragma unroll
      for (int k = 0; k < ROWS; k++) {
        const u32 i = k * THREADS + tid;
        ha[k] = 0;
        if (i < nb) {
          if (a.pfx_shift && (br[k].key >> a.pfx_shift) != a.pfx_val) pfx_bad = true;
          const u32 h = key_bucket(br[k].key, bsh, NB - 1);
          ha[k] = h | (atomicAdd(&sm.cnt[h], 1u) << 12);
        }
      }
    }
    lds_barrier();  // bucket counts complete; the next ticket is visible
    if constexpr (FK) {  // the match counts (the previous partition's copy-out is over: one barrier since)
      for (u32 i = tid; i < CAPB + 2; i += THREADS) sm.mcnt[i] = 0;  // (+ anydup, hot: the two words behind the array)
      if (tid == 0) {
        sm.vmin = ~0ull;
        sm.vmax = 0;
      }
    } else {
      if ((u32)tid < WORDS) sm.mbits[par ^ 1][tid] = 0;  // the next partition's bitmap (the previous one's copy-out is over)
    }
    const u32 pn = chained ? (u32)__builtin_amdgcn_readfirstlane((int)sm.tick[par ^ 1]) : p + gridDim.x;
    u32 rb2 = 0, nb2 = 0, sb2 = 0, np2 = 0;
    bool regular2 = false;
    if (pn < P) {
      if (SLAB) {
        const u32* rc = a.r_cnt + (u64)pn * SLAB_KB;
        const u32* sc = a.s_cnt + (u64)pn * SLAB_KB;
        r1n = rc[0]; r2n = r1n + rc[1]; r3n = r2n + rc[2]; nb2 = r3n + rc[3];
        s1n = sc[0]; s2n = s1n + sc[1]; s3n = s2n + sc[2]; np2 = s3n + sc[3];
        if (nb2 > CAPB || np2 > CAP) slab_bad = true;
      } else {
        rb2 = r_off[pn]; nb2 = r_end[pn] - rb2;
        sb2 = s_off[pn]; np2 = s_end[pn] - sb2;
      }
      regular2 = nb2 && np2 && nb2 <= CAPB && np2 <= CAP;
    }
    auto load_next_build = [&]() {
      if (SLAB)
        fp_load_slab<THREADS, ROWS>(br, regular2 ? R + (u64)pn * SLAB_KB * a.r_cap : dummy, regular2 ? a.r_cap : 0u,
                              regular2 ? r1n : 1u, regular2 ? r2n : 1u, regular2 ? r3n : 1u, regular2 ? nb2 : 1u, tid);
      else
        fp_load<THREADS, ROWS>(br, regular2 ? R + rb2 : dummy, regular2 ? nb2 : 1u, tid);
    };
    // The NEXT partition's probe rows are requested as soon as this partition's are consumed -- before the bitmap
    // scan, the look-back and the copy-out, which do not touch them -- so they have the whole write phase and the
    // next partition's count / scan / place phases to arrive (round 2 asked for them at the top of the iteration
    // and waited for them three barriers later).
    auto load_next_probe = [&]() {
      if (!EARLY_PROBE) return;
      if (SLAB)
        fp_load_slab<THREADS, ROWS>(pr, regular2 ? S + (u64)pn * SLAB_KB * a.s_cap : dummy, regular2 ? a.s_cap : 0u,
                              regular2 ? s1n : 1u, regular2 ? s2n : 1u, regular2 ? s3n : 1u, regular2 ? np2 : 1u, tid);
      else
        fp_load<THREADS, ROWS>(pr, regular2 ? S + sb2 : dummy, regular2 ? np2 : 1u, tid);
    };
    u32 total = 0;       // result rows of this partition
    constexpr int WROUNDS = (int)((WORDS + kWave - 1) / kWave);
    u32 wpre[WROUNDS];   // bitmap form: matched rows before words lane, lane + 64, ...
#pragma unroll
    for (int r = 0; r < WROUNDS; r++) wpre[r] = 0;
    bool sorted_ok = false;
    // The phases below are guarded one by one instead of nested in one `if (regular) ... else ...`: the prefetches of
    // the next partition's rows then have ONE call site each.  With a call site per branch the loaded rows of the
    // branches met in a phi, the register allocator gave them registers other than the loop header's, and the copies
    // at the loop's back edge waited for every load and store in flight (s_waitcnt vmcnt(0)) once per partition.
    if (regular) {  // exclusive scan of the bucket counts, BPT per thread (and the counts go back to zero for the next partition)
      u32 c[BPT], sum = 0, mx = 0;
#pragma unroll
      for (u32 q = 0; q < BPT; q++) {
        c[q] = sm.cnt[tid * BPT + q];
        sm.cnt[tid * BPT + q] = 0;
        sum += c[q];
        mx = c[q] > mx ? c[q] : mx;
      }
      if (mx > (u32)SW_MAXBUCKET) (&sm.flag)[zero] = 1;
      u32 tot;
      u32 ex = block_excl_scan_u32<THREADS, false>(sum, sm.scratch, &tot, tid);  // (a barrier follows below)
#pragma unroll
      for (u32 q = 0; q < BPT; q++) {
        sm.bstart[tid * BPT + q] = (u16)ex;
        ex += c[q];
      }
      if (tid == THREADS - 1) sm.bstart[tid * BPT + BPT] = (u16)ex;  // = bstart[NB] = nb: the end of the last bucket
      lds_barrier();
    }
    if constexpr (!FK) {
      if (regular && sm.flag == 0) {  // uniform
        // key and payload go to their bucket, in order of arrival.  A row's sorted rank -- its bucket's start plus
        // the number of smaller keys in the bucket -- is what the output order needs; the probe row that matches it
        // scans the whole bucket anyway and counts them on the way, so the build rows are never moved again.
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
          const u32 i = k * THREADS + tid;
          if (i < nb) {
            const u32 pos = (u32)sm.bstart[ha[k] & (NB - 1)] + (ha[k] >> 12);
            sm.key[pos] = br[k].key;
            sm.val[pos] = br[k].val;
          }
        }
      }
    } else {
      // ---- foreign-key form: the build rows are sorted for real (position = sorted rank)
      u64* tmpkey = sm.sval;  // keys in bucket / arrival order (sval is free until the payloads are dropped)
      if (regular && sm.flag == 0) {  // uniform
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
          const u32 i = k * THREADS + tid;
          if (i < nb) tmpkey[(u32)sm.bstart[ha[k] & (NB - 1)] + (ha[k] >> 12)] = br[k].key;
        }
      }
      if (regular) lds_barrier();
      if (regular && sm.flag == 0) {
        bool dupb = false;
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
          const u32 i = k * THREADS + tid;
          if (i < nb) {
            const u32 h = ha[k] & (NB - 1), b0 = sm.bstart[h], b1 = sm.bstart[h + 1], mine = b0 + (ha[k] >> 12);
            u32 less = 0;
            for (u32 t = b0; t < b1; t++) {  // (a bucket holds a handful of keys: <= SW_MAXBUCKET, flagged above)
              const u64 kk = tmpkey[t];
              less += kk < br[k].key ? 1u : 0u;
              dupb |= kk == br[k].key && t != mine;
            }
            sm.key[b0 + less] = br[k].key;
            sm.val[b0 + less] = br[k].val;
          }
        }
        if (dupb) (&sm.flag)[zero] = 3;  // two build rows with one key: not this kernel's case
      }
    }
    load_next_build();              // (the only call site)
    if (regular) lds_barrier();     // table complete (foreign-key form: and tmpkey is dead)
    u32 found[FK ? ROWS : 1], slot[FK ? ROWS : 1];  // foreign-key form: a probe row's match, sorted rank + 1 (0 = none); its arrival number in the key's run
#pragma unroll
    for (int k = 0; k < (FK ? ROWS : 1); k++) found[k] = slot[k] = 0;
    // Foreign-key form, round 4: where the partition's build rows leave bucket bits over (2^b >= build rows, b < LOGB) the
    // output slots are bucketed by (sorted build rank, position of the payload in the partition's payload range) instead of by
    // build rank alone: a row then ranks itself among the few rows of its bucket, not among the f rows of its key's run --
    // the ranking was half of this kernel's instructions at fan-out 16 and grew linearly with the fan-out.
    int svb = 0, vsh = 0;
    u64 vmin = 0;
    // (The caller asks for it from ~24 probe rows per build row on: below, the extra barrier, the payload range and the scan
    //  over all buckets cost more than the shorter ranking saves -- write phase at fan-out 8 / 16 / 32 / 64 / 256, 2^28 probe
    //  rows: 4.3 / 5.4 / 6.0 / 8.8 / 24.2 ms without, 4.8 / 5.8 / 5.9 / 6.8 / 10.9 with.)
    // (PB: its own instantiation -- as a run-time branch the added code cost the short-run joins 10 % in registers and scratch)
    if constexpr (FK && PB) {
      if (regular) {
        const int nbits = nb > 1 ? 32 - __builtin_clz(nb - 1) : 0;
        svb = nbits < (int)Smem::LOGB ? (int)Smem::LOGB - nbits : 0;
      }
    }
    if constexpr (!FK) {
      if (regular && sm.flag == 0) {
        // probe: scan the key's bucket (five rows in lockstep so their LDS latencies overlap).  Per row one packed
        // word: smaller keys seen (bits 0-5), step of the match (6-11), matches (12-13)
        u32 cur[ROWS], len[ROWS], st[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
          const u32 j = k * THREADS + tid;
          cur[k] = len[k] = st[k] = 0;
          if (j < np) {
            if (a.pfx_shift && (pr[k].key >> a.pfx_shift) != a.pfx_val) pfx_bad = true;
            if (EXTRA) acc_p += pr[k].val;
            const u32 hh = key_bucket(pr[k].key, bsh, NB - 1);
            cur[k] = sm.bstart[hh];
            len[k] = (u32)sm.bstart[hh + 1] - cur[k];
          }
        }
        for (u32 step = 0; step < (u32)SW_MAXBUCKET; step++) {
          bool any = false;
#pragma unroll
          for (int k = 0; k < ROWS; k++) any |= step < len[k];
          if (!__any(any)) break;
#pragma unroll
          for (int k = 0; k < ROWS; k++) {
            if (step < len[k]) {
              const u64 kk = sm.key[cur[k] + step];
              st[k] += kk < pr[k].key ? 1u : 0u;
              if (kk == pr[k].key) st[k] = (st[k] & ~(63u << 6)) + (step << 6) + (1u << 12);
            }
          }
        }
        bool dup = false, dupb = false;
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
          const u32 hits = st[k] >> 12;
          if (hits) {
            const u32 si = cur[k] + (st[k] & 63u), bit = 1u << (si & 31);  // the matched build row's sorted rank
            const u32 mpos = cur[k] + ((st[k] >> 6) & 63u);
            dupb |= hits > 1;                                            // two build rows with this key
            dup |= (atomicOr(&sm.mbits[par][si >> 5], bit) & bit) != 0;  // two probe rows with this key
            sm.sval[si] = pr[k].val;
            sm.perm[si] = (u16)mpos;
            acc_n++;
            acc_s += pr[k].val;
            if (EXTRA) {
              const u64 m = tmix(pr[k].key, sm.val[mpos], pr[k].val);
              acc_x ^= m;
              acc_m += m;
            }
          }
        }
        if (dup) (&sm.flag)[zero] = 2;
        if (dupb) (&sm.flag)[zero] = 3;
      }
    } else {
      if (regular && sm.flag == 0) {
        u32 cur[ROWS], len[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
          const u32 j = k * THREADS + tid;
          cur[k] = len[k] = 0;
          if (j < np) {
            if (a.pfx_shift && (pr[k].key >> a.pfx_shift) != a.pfx_val) pfx_bad = true;
            if (EXTRA) acc_p += pr[k].val;
            const u32 hh = key_bucket(pr[k].key, bsh, NB - 1);
            cur[k] = sm.bstart[hh];
            len[k] = (u32)sm.bstart[hh + 1] - cur[k];
          }
        }
        for (u32 step = 0; step < (u32)SW_MAXBUCKET; step++) {
          bool any = false;
#pragma unroll
          for (int k = 0; k < ROWS; k++) any |= step < len[k];
          if (!__any(any)) break;
#pragma unroll
          for (int k = 0; k < ROWS; k++) {
            if (step < len[k] && sm.key[cur[k] + step] == pr[k].key) {
              found[k] = cur[k] + step + 1;
              len[k] = 0;  // (keys are unique in the table: done)
            }
          }
        }
        if (svb) {  // the partition's payload range, for the payload-position bits of the bucket number
          u64 mn = ~0ull, mx = 0;
#pragma unroll
          for (int k = 0; k < ROWS; k++) {
            if (found[k]) {
              mn = pr[k].val < mn ? pr[k].val : mn;
              mx = pr[k].val > mx ? pr[k].val : mx;
            }
          }
#pragma unroll
          for (int o = kWave / 2; o > 0; o >>= 1) {
            const u64 a2 = __shfl_xor(mn, o, kWave), b2 = __shfl_xor(mx, o, kWave);
            mn = a2 < mn ? a2 : mn;
            mx = b2 > mx ? b2 : mx;
          }
          if (lane == 0 && mn <= mx) {
            atomicMin(&sm.vmin, (unsigned long long)mn);
            atomicMax(&sm.vmax, (unsigned long long)mx);
          }
        }
      }
      if (svb && regular) lds_barrier();
      if (regular && sm.flag == 0) {
        if (svb) {
          vmin = sm.vmin;
          const u64 vmax = sm.vmax;
          const int rbits = vmax > vmin ? 64 - __builtin_clzll(vmax - vmin) : 0;
          vsh = rbits > svb ? rbits - svb : 0;
        }
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
          if (found[k]) {
            // arrival number among the probe rows of the key -- or, with payload buckets, of the key's bucket
            slot[k] = svb ? atomicAdd(&sm.cnt[((found[k] - 1) << svb) | (u32)((pr[k].val - vmin) >> vsh)], 1u)
                          : atomicAdd(&sm.mcnt[found[k] - 1], 1u);
            acc_n++;
            acc_s += pr[k].val;
            if (EXTRA) {
              const u64 m = tmix(pr[k].key, sm.val[found[k] - 1], pr[k].val);
              acc_x ^= m;
              acc_m += m;
            }
          }
        }
      }
      if (regular) lds_barrier();
      if (regular && sm.flag == 0 && svb) {  // uniform
        // every bucket's run of output slots: buckets are numbered (sorted build rank, payload position), i.e. in output order
        u32 c[BPT], sum = 0, mx = 0;
#pragma unroll
        for (u32 q = 0; q < BPT; q++) {
          c[q] = sm.cnt[tid * BPT + q];
          sm.cnt[tid * BPT + q] = 0;  // (zero again for the next partition's build buckets)
          sum += c[q];
          mx = c[q] > mx ? c[q] : mx;
        }
        if (mx > (u32)SWF_MAXDUP) sm.mcnt[CAPB + 1] = 1;
        if (mx > 1) sm.mcnt[CAPB] = 1;
        u32 ex = block_excl_scan_u32<THREADS, false>(sum, sm.scratch, &total, tid);  // (a barrier follows below)
#pragma unroll
        for (u32 q = 0; q < BPT; q++) {
          sm.bstart[tid * BPT + q] = (u16)ex;
          ex += c[q];
        }
        if (tid == THREADS - 1) sm.bstart[tid * BPT + BPT] = (u16)ex;
        lds_barrier();
      }
      if (regular && sm.flag == 0 && !svb) {  // uniform
        // every build row's run of output slots: exclusive scan of the match counts in sorted build order
        u32 c[ROWS], sum = 0, mx = 0;
#pragma unroll
        for (int q = 0; q < ROWS; q++) {
          const u32 i = (u32)tid * ROWS + q;
          c[q] = i < CAPB ? sm.mcnt[i] : 0u;
          sum += c[q];
          mx = c[q] > mx ? c[q] : mx;
        }
        if (mx > (u32)SWF_MAXDUP) sm.mcnt[CAPB + 1] = 1;  // (its own word: sm.flag is being read by slower threads right now)
        if (mx > 1) sm.mcnt[CAPB] = 1;
        u32 ex = block_excl_scan_u32<THREADS, false>(sum, sm.scratch, &total, tid);  // (a barrier follows below)
#pragma unroll
        for (int q = 0; q < ROWS; q++) {
          const u32 i = (u32)tid * ROWS + q;
          if (i < CAPB) sm.mcnt[i] = ex;
          ex += c[q];
        }
        lds_barrier();
      }
      if (regular) {
        sorted_ok = sm.flag == 0 && sm.mcnt[CAPB + 1] == 0;
        if (sorted_ok) {
          // payload and sorted rank to the key's run, in arrival order
#pragma unroll
          for (int k = 0; k < ROWS; k++) {
            if (found[k]) {
              const u32 o = (svb ? (u32)sm.bstart[((found[k] - 1) << svb) | (u32)((pr[k].val - vmin) >> vsh)] : sm.mcnt[found[k] - 1]) + slot[k];
              sm.sval[o] = pr[k].val;
              sm.srank[o] = (u16)(found[k] - 1);
            }
          }
        } else {
          giveup = true;
          if (sm.flag == 0) why |= 2048u;
          total = 0;
        }
      }
    }
    load_next_probe();              // (the only call site; a no-op in the shapes that ask at the top of the iteration)
    if constexpr (!FK) {
      if (regular) {
        lds_barrier();
        sorted_ok = sm.flag == 0;
        if (!sorted_ok) giveup = true;
        // matched rows before every bitmap word, and the partition's row count: every wave scans the 160 words for
        // itself (lane l keeps the prefixes of words l, l + 64, l + 128), so nobody waits for anybody
        u32 run = 0;
#pragma unroll
        for (int r = 0; r < WROUNDS; r++) {
          const u32 w = (u32)r * kWave + lane;
          const u32 c = (sorted_ok && w < WORDS) ? (u32)__popc(sm.mbits[par][w]) : 0u;
          const u32 incl = wave_incl_scan_u32(c, lane);
          wpre[r] = run + incl - c;
          run += (u32)__builtin_amdgcn_readlane((int)incl, 63);
        }
        total = run;
      }
    }
    if (!regular) {
      if (nb && np) {  // does not fit the pipeline
        giveup = true;
        why |= 512u;
      }
      if (EXTRA && !nb) {  // probe rows without a build partition still count in sum_probe_all
        if (SLAB) {
          const Tup* base = S + (u64)p * SLAB_KB * a.s_cap;
          for (u32 j = tid; j < np; j += THREADS) {
            const u32 pc = (j >= s1) + (j >= s2) + (j >= s3);
            const u32 pre = pc == 0 ? 0 : pc == 1 ? s1 : pc == 2 ? s2 : s3;
            acc_p += base[(u64)pc * a.s_cap + (j - pre)].val;
          }
        } else {
          for (u32 j = tid; j < np; j += THREADS) acc_p += S[sb + j].val;
        }
      }
    }
    // every partition publishes its count (zero if it has no rows or gave up): its successors wait for it
    // chained: the rows of all partitions before this one (wave 0 finds out, the others wait); slots: the
    // partition's own probe-row slots (the result is then dense only if every probe row matched -- the caller
    // checks and closes the gaps otherwise)
    if (tid == 0) a.part_count[p + zero] = total;
    if (chained) {
      if (wv == 0) {
        const u64 excl = lookback_publish(state, p, (u64)total, lane, &lb_timeout);
        if (lane == 0) sm.obase = excl;
      }
      lds_barrier();
      ob = sm.obase;
    } else if (FK) {
      lds_barrier();  // the payloads and ranks of all output slots are in place
    }
    if (FK && sorted_ok) {
      if constexpr (FK) {
        // copy-out in SLOT order: lane <-> output slot, so the lanes of a run are neighbours and read the run's
        // payloads as broadcasts.  rank = smaller payloads in the run + equal ones in earlier slots.
        const bool rank_runs = sm.mcnt[CAPB] != 0;  // uniform
        // (all LDS work of the five slots first, then the stores back to back from registers of their own: see the
        //  bitmap form below)
        u32 rel[ROWS];
        u64 ok[ROWS], orv[ROWS], osv[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
          const u32 j = k * THREADS + tid;
          const bool live = j < total;
          u32 si = 0, base = 0, c = 0;
          u64 v = 0;
          if (live) {
            si = sm.srank[j];
            v = sm.sval[j];
            if (svb) {  // rank inside the (key, payload position) bucket: a few rows however long the key's run is
              const u32 b = (si << svb) | (u32)((v - vmin) >> vsh);
              base = sm.bstart[b];
              c = (u32)sm.bstart[b + 1] - base;
            } else {
              base = sm.mcnt[si];
              c = (si + 1 < CAPB ? sm.mcnt[si + 1] : total) - base;
            }
          }
          u32 r = j - base;  // (runs of one row, or no repeating key in the partition: the slot is final)
          if (rank_runs) {
            // rank = payloads of the run below mine + equal ones in earlier slots.  Eight reads per step at constant
            // offsets from one address (unclamped: a read past the run's end is ignored, one past sval[] lands in the
            // arrays behind it -- a run is at most SWF_MAXDUP rows); round 2's form clamped every index and spent 18
            // instructions per pair, which made fan-outs from 64 on compute-bound (profiles/r03c_side_ordered_kernels.txt)
            const u32 rel = j - base;
            const u64* rp = &sm.sval[base];
            r = 0;
            for (u32 t = 0; __any(t < c); t += 8) {
              u64 o[8];
#pragma unroll
              for (int u = 0; u < 8; u++) o[u] = rp[t + u];
#pragma unroll
              for (int u = 0; u < 8; u++) {
                const u32 at = t + u;
                r += ((at < c) & ((o[u] < v) | ((o[u] == v) & (at < rel)))) ? 1u : 0u;
              }
            }
          }
          rel[k] = live ? base + r : 0xFFFFFFFFu;
          ok[k] = live ? sm.key[si] : 0;
          orv[k] = live ? sm.val[si] : 0;
          osv[k] = v;
        }
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
          if (rel[k] != 0xFFFFFFFFu) {
            const u64 d = ob + rel[k];
            a.out_key[d] = ok[k];
            a.out_rval[d] = orv[k];
            a.out_sval[d] = osv[k];
            acc_r += orv[k];
          }
        }
      }
    } else if (sorted_ok) {
      if constexpr (!FK) {
      static_assert(THREADS == 1024, "word of slot k: 32 k + tid / 32");
      // all LDS reads of the five rows first, then the fifteen stores back to back, every row in registers of its
      // own: with the rows interleaved the compiler reused one row's store-data registers for the next row's LDS
      // reads and put an s_waitcnt vmcnt(1) between them -- each row then waited for the previous row's stores
      // AND for the prefetches of the next partition that are in flight (vector-memory operations retire in order)
      u32 rel[ROWS];  // output row relative to ob, or ~0
      u64 ok[ROWS], orv[ROWS], osv[ROWS];
#pragma unroll
      for (int k = 0; k < ROWS; k++) {
        const u32 si = k * THREADS + tid;
        // word si / 32 = 32 k + tid / 32: scan round k / 2, lane 32 (k & 1) + tid / 32 of this wave's registers
        const u32 before = (u32)__shfl((int)wpre[k >> 1], (int)((k & 1) * 32 + ((u32)tid >> 5) % 32u), kWave);
        rel[k] = 0xFFFFFFFFu;
        ok[k] = orv[k] = osv[k] = 0;
        if (si < nb) {
          const u32 w = sm.mbits[par][si >> 5], b = si & 31;
          if ((w >> b) & 1u) {
            rel[k] = before + (u32)__popc(w & ((1u << b) - 1u));
            const u32 m = sm.perm[si];
            ok[k] = sm.key[m];
            orv[k] = sm.val[m];
            osv[k] = sm.sval[si];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < ROWS; k++) {
        if (rel[k] != 0xFFFFFFFFu) {
          const u64 d = ob + rel[k];
          a.out_key[d] = ok[k];
          a.out_rval[d] = orv[k];
          a.out_sval[d] = osv[k];
          acc_r += orv[k];  // (summed here, where the payload is read anyway: one random LDS read less in the probe walk)
        }
      }
      }
    }
    p = pn; par ^= 1;
    rb = rb2; nb = nb2; sb = sb2; np = np2; regular = regular2;
    r1 = r1n; r2 = r2n; r3 = r3n; s1 = s1n; s2 = s2n; s3 = s3n;
  }
  if (SLAB && slab_bad && tid == 0) atomicOr(&a.accum[ACC_ERR], ERR_SLAB);
  if (__any(giveup || lb_timeout) && lane == 0) {
    const u32 f = sm.flag;  // 1: a bucket too long, 2: repeating probe keys (bitmap form), 3: duplicate build keys
    atomicOr(&a.accum[ACC_ERR], ERR_SORTED | why | (f == 1 ? 128u : 0u) | (f == 2 ? 256u : 0u) | (f == 3 ? 2048u : 0u) |
                                    (lb_timeout ? 1024u : 0u));
  }
  if (__any(pfx_bad) && lane == 0) atomicOr(&a.accum[ACC_ERR], ERR_PREFIX);
  lds_barrier();
  const u64 v[6] = {acc_n, acc_r, acc_s, acc_x, acc_m, acc_p};
  block_accumulate(sm.red, a.accum, v, 1u << ACC_XOR);
}

// Unordered result of the unique-key write mode with unmatched probe rows: every partition's rows, written from the
// partition's first probe-row slot on, move to the partition's dense offset (the order inside a partition stays).
__global__ __launch_bounds__(256) void compact_rows_kernel(const u64* __restrict__ out_off, const u32* __restrict__ base32,
                                                           const u64* __restrict__ base64, u32 P,
                                                           const u64* __restrict__ akey, const u64* __restrict__ arval,
                                                           const u64* __restrict__ asval, u64* __restrict__ bkey,
                                                           u64* __restrict__ brval, u64* __restrict__ bsval) {
  for (u32 p = blockIdx.x; p < P; p += gridDim.x) {
    const u64 dst = out_off[p], n = out_off[p + 1] - dst;
    const u64 src = base64 ? base64[p] : (u64)base32[p];
    for (u64 i = threadIdx.x; i < n; i += blockDim.x) {
      bkey[dst + i] = akey[src + i];
      brval[dst + i] = arval[src + i];
      bsval[dst + i] = asval[src + i];
    }
  }
}

hipError_t launch_compact(const u64* part_out_off, const u32* in_base32, const u64* in_base64, u32 P, const u64* akey,
                          const u64* arval, const u64* asval, u64* bkey, u64* brval, u64* bsval, int grid, hipStream_t st) {
  if (P == 0) return hipSuccess;
  if ((u32)grid > P) grid = (int)P;
  hipLaunchKernelGGL(compact_rows_kernel, dim3(grid), dim3(256), 0, st, part_out_off, in_base32, in_base64, P, akey, arval,
                     asval, bkey, brval, bsval);
  return hipGetLastError();
}

// Exclusive scan of n u64 counts into n+1 offsets (single workgroup; n = partition count).  A wave owns a contiguous chunk
// and walks it 256 entries at a time -- one entry per lane, coalesced, four loads in flight: first the chunk totals, then the
// running offsets.  (Until round 5 the workgroup walked the array 1024 entries at a time behind three barriers each:
// 0.19 ms for 2^17 partitions, 2 % of an ordered foreign-key join -- profiles/r05a_fk22_ord_summary.txt.)
__global__ __launch_bounds__(1024) void scan_u64_kernel(const u64* __restrict__ in,
                                                        u64* __restrict__ out, u32 n) {
  __shared__ u64 wtot[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 chunk = ((n + 15u) / 16u + 63u) & ~63u;
  const u32 b = (u32)w * chunk < n ? (u32)w * chunk : n, e = b + chunk < n ? b + chunk : n;
  auto at = [&](u32 i) -> u64 { return i < e ? in[i] : 0ull; };
  u64 s = 0;
  for (u32 i = b + (u32)lane; i < e; i += 256) s += at(i) + at(i + 64) + at(i + 128) + at(i + 192);
  s = wave_sum_u64(s);
  if (lane == 0) wtot[w] = s;
  __syncthreads();
  u64 carry = 0, all = 0;
  for (int k = 0; k < 16; k++) {
    if (k < w) carry += wtot[k];
    all += wtot[k];
  }
  for (u32 i0 = b; i0 < e; i0 += 256) {
    u64 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = at(i0 + (u32)k * 64 + (u32)lane);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      u64 incl = v[k];
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) {
        const u64 t = __shfl_up(incl, o, kWave);
        if (lane >= o) incl += t;
      }
      const u32 i = i0 + (u32)k * 64 + (u32)lane;
      if (i < e) out[i] = carry + incl - v[k];
      carry += __shfl(incl, kWave - 1, kWave);
    }
  }
  if (tid == 0) out[n] = all;
}

// HMJ_ORDERED epilogue: sort each partition's result rows by (key, rval, sval).  Partitions are
// key ranges in ascending order (partitioning uses the most significant key bits, radix_hash.h:369),
// so sorted partitions concatenate to the ascending-key order HashMergeJoin iteration has
// (hashjoin.h:86-101).
//
// Fast path (segments of <= OS_CAP rows): one bucket pass on the 12 key bits just below the
// partition bits (4096 buckets for ~4096 rows: about one row per bucket for the uniform keys the
// planner assumes), then every row ranks itself inside its bucket by comparing with the few rows
// that share it; rows are staged through LDS column by column and leave as coalesced stores into
// the second set of result columns.  Out of place: A (unsorted) -> B (sorted).
// Slow path (longer segments, or a bucket with more than OS_MAXBUCKET rows: duplicate-heavy keys):
// copy A -> B, then a bitonic network on the global columns ("all ascending" form, no padding).
constexpr int OS_THREADS = 1024, OS_ROWS = 5, OS_CAP = OS_THREADS * OS_ROWS, OS_LOGB = 12;
constexpr int OS_NB = 1 << OS_LOGB, OS_MAXBUCKET = 48;
// MANY (more result rows than twice the build rows: keys repeat on the probe side, or on both): a key's rows share
// their bucket, so buckets of f_build x f_probe rows are the normal case, not a sign of trouble -- 8 copies of every
// key on both sides make 64-row buckets, which the 48-row limit sent to the one-workgroup bitonic network in global
// memory (2^24 x 2^24 rows of 2^21 keys, ordered: 116 ms for 1.3 * 10^8 rows against 3.1 ms unordered).  The
// in-bucket ranking is linear in the bucket, which stays cheap up to a few hundred rows.
constexpr int OS_MAXBUCKET_MANY = 640;
template <bool MANY>
struct OrderLimits {
  static constexpr u32 MAXBUCKET = MANY ? OS_MAXBUCKET_MANY : OS_MAXBUCKET;
  static constexpr u32 CHUNK_ROWS = OS_CAP - MAXBUCKET;  // a chunk = the buckets that START inside one such window
};

struct OrderSmem {
  u64 stage[OS_CAP];  // the segment's three columns: unsorted while ranking (rows of one key are ranked on
  u64 srv[OS_CAP];    // (rval, sval) -- a foreign-key join has many per key -- without going back to global
  u64 ssv[OS_CAP];    // memory), then sorted for the coalesced copy-out
  u16 sidx[OS_CAP];   // row indices grouped by bucket
  u16 bstart[OS_NB + 2];
  u32 bcur[OS_NB];    // bucket counts, then insertion cursors
  u32 scratch[OS_THREADS / kWave + 1];
  u32 fallback;
  u32 mixed;                      // the segment holds more than one (key, rval) pair
  unsigned long long svmin, svmax;  // smallest / largest sval of the segment
};

// How the rows of a segment are spread over the 4096 buckets of the in-LDS sort.  By key bits (what a join result
// normally varies in) -- or, for a segment that is ONE key's run (a hot foreign key: thousands of rows that agree in key
// and rval), by the position of the sval inside the segment's sval range: buckets then ascend in (key, rval, sval) order
// all the same, and hold a handful of rows each where the key bits would put the whole run into one.  (Without it such a
// segment went to the one-workgroup bitonic network in global memory: 2^12 build keys x 2^26 probe rows, ordered: 49 of
// 55 ms; tools/exp_cliffs2.py.)
struct OrderBuckets {
  int by_sval;
  int bsh;     // by key: bucket = key bits [bsh, bsh + 12) (key_bucket)
  int svsh;    // by sval: bucket = (sval - svbase) >> svsh
  u64 svbase;
};
__device__ __forceinline__ u32 order_bucket(const OrderBuckets& f, u64 key, u64 sval) {
  return f.by_sval ? (u32)((sval - f.svbase) >> f.svsh) : key_bucket(key, f.bsh, OS_NB - 1);
}
// by-sval buckets for a segment whose svals span [mn, mx] (mx > mn)
__device__ __forceinline__ OrderBuckets order_buckets_by_sval(u64 mn, u64 mx, int bsh) {
  OrderBuckets f;
  const int bits = 64 - __clzll((long long)(mx - mn));
  f.by_sval = 1;
  f.bsh = bsh;
  f.svsh = bits > OS_LOGB ? bits - OS_LOGB : 0;
  f.svbase = mn;
  return f;
}

__device__ __forceinline__ void order_network_global(u64* k, u64* r, u64* s, u32 L, int tid) {
  u32 n2 = 2;
  while (n2 < L) n2 <<= 1;
  for (u32 size = 2; size <= n2; size <<= 1) {
    for (u32 j = size >> 1; j > 0; j >>= 1) {
      const bool mirror = (j == (size >> 1));
      for (u32 t = tid; t < (n2 >> 1); t += OS_THREADS) {
        u32 i, l;
        if (mirror) {
          u32 blk = t / j, idx = t % j;
          i = blk * size + idx;
          l = blk * size + (size - 1 - idx);
        } else {
          i = 2 * t - (t & (j - 1));
          l = i + j;
        }
        if (l < L) {
          u64 ki = k[i], kl = k[l], ri = r[i], rl = r[l], si = s[i], sl = s[l];
          bool gt = (ki != kl) ? (ki > kl) : (ri != rl) ? (ri > rl) : (si > sl);
          if (gt) {
            k[i] = kl; k[l] = ki;
            r[i] = rl; r[l] = ri;
            s[i] = sl; s[l] = si;
          }
        }
      }
      __syncthreads();
    }
  }
}

// Sort L <= OS_CAP rows held in registers (row i = k * OS_THREADS + tid in key[k], rv[k], sv[k]) by
// (key, rval, sval) and store them at bkey/brval/bsval[ob ..).  One bucket pass on key bits [bsh, bsh+12)
// (all rows must agree in the bits above), then every row ranks itself among the rows of its bucket.
// Returns false -- nothing stored -- if a bucket holds more than OS_MAXBUCKET rows (duplicate-heavy keys).
// MANY: the caller expects several rows per key (fan-out): the ranking loop is then branch-free and unrolled, so
// that the loads of consecutive candidates overlap (-11 % on a 16-fold fan-out, +13 % on unique keys).
template <bool MANY>
__device__ __forceinline__ bool order_sort_registers(OrderSmem& sm, u32 L, const u64 (&key)[OS_ROWS],
                                                     const u64 (&rv)[OS_ROWS], const u64 (&sv)[OS_ROWS], const OrderBuckets& bf,
                                                     u64 ob, u64* __restrict__ bkey, u64* __restrict__ brval,
                                                     u64* __restrict__ bsval, int tid) {
  u32 bk[OS_ROWS], dest[OS_ROWS];
  __syncthreads();
  for (u32 i = tid; i < (u32)OS_NB; i += OS_THREADS) sm.bcur[i] = 0;
  if (tid == 0) sm.fallback = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < OS_ROWS; k++) {
    const u32 i = k * OS_THREADS + tid;
    bk[k] = 0;
    if (i < L) {
      bk[k] = order_bucket(bf, key[k], sv[k]);
      sm.stage[i] = key[k];
      sm.srv[i] = rv[k];
      sm.ssv[i] = sv[k];
      atomicAdd(&sm.bcur[bk[k]], 1u);
    }
  }
  __syncthreads();
  {  // exclusive scan of the 4096 bucket counts, 4 per thread
    u32 c[4], sum = 0, mx = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      c[q] = sm.bcur[tid * 4 + q];
      sum += c[q];
      mx = c[q] > mx ? c[q] : mx;
    }
    if (mx > OrderLimits<MANY>::MAXBUCKET) sm.fallback = 1;
    u32 tot;
    u32 ex = block_excl_scan_u32<OS_THREADS>(sum, sm.scratch, &tot);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      sm.bstart[tid * 4 + q] = (u16)ex;
      sm.bcur[tid * 4 + q] = ex;
      ex += c[q];
    }
    if (tid == 0) sm.bstart[OS_NB] = (u16)L;
  }
  __syncthreads();
  if (sm.fallback != 0) return false;  // uniform for the workgroup
#pragma unroll
  for (int k = 0; k < OS_ROWS; k++) {
    const u32 i = k * OS_THREADS + tid;
    if (i < L) sm.sidx[atomicAdd(&sm.bcur[bk[k]], 1u)] = (u16)i;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < OS_ROWS; k++) {
    const u32 i = k * OS_THREADS + tid;
    dest[k] = 0;
    if (i < L) {
      const u32 s0 = sm.bstart[bk[k]], e0 = sm.bstart[bk[k] + 1];
      u32 rank = 0;
      if (MANY) {  // order by key, then (rval, sval), then position; the row itself is "not less"
#pragma unroll 4
        for (u32 j = s0; j < e0; j++) {
          const u32 o = sm.sidx[j];
          const u64 ok = sm.stage[o], orv = sm.srv[o], osv = sm.ssv[o];
          const bool less = (ok < key[k]) | ((ok == key[k]) & ((orv < rv[k]) | ((orv == rv[k]) &
                            ((osv < sv[k]) | ((osv == sv[k]) & (o < i))))));
          rank += less ? 1u : 0u;
        }
      } else {
        for (u32 j = s0; j < e0; j++) {
          const u32 o = sm.sidx[j];
          if (o == i) continue;
          const u64 ok = sm.stage[o];
          bool less = ok < key[k];
          if (ok == key[k]) {  // duplicate key: order by (rval, sval), then by position
            const u64 orv = sm.srv[o], osv = sm.ssv[o];
            less = (orv != rv[k]) ? (orv < rv[k]) : (osv != sv[k]) ? (osv < sv[k]) : (o < i);
          }
          rank += less ? 1u : 0u;
        }
      }
      dest[k] = s0 + rank;
    }
  }
  __syncthreads();
  // the three columns in sorted order (the unsorted copies are no longer needed), copied out coalesced
#pragma unroll
  for (int k = 0; k < OS_ROWS; k++) {
    const u32 i = k * OS_THREADS + tid;
    if (i < L) {
      sm.stage[dest[k]] = key[k];
      sm.srv[dest[k]] = rv[k];
      sm.ssv[dest[k]] = sv[k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < OS_ROWS; k++) {
    const u32 i = k * OS_THREADS + tid;
    if (i < L) {
      bkey[ob + i] = sm.stage[i];
      brval[ob + i] = sm.srv[i];
      bsval[ob + i] = sm.ssv[i];
    }
  }
  __syncthreads();
  return true;
}

// Segment p: input rows A[in_base(p) .. + len(p)), output rows B[off[p*Q] ..).  By default the input is
// laid out like the output (in_base32 == in_base64 == NULL); the unique-key write mode passes where
// each partition's rows were written (slot of its first probe row) and their count.
//   L <= OS_CAP            : one in-LDS sort;
//   L <= OS_CHUNKS * chunk : (a many-to-many join: more result rows than probe rows) the 4096 buckets are
//                            counted over the whole segment, consecutive buckets are grouped into chunks of
//                            at most OS_CAP rows, and every chunk is gathered from the segment and sorted in
//                            LDS -- the segment's keys are re-read once per chunk;
//   otherwise, or when one bucket holds more than OS_MAXBUCKET rows: the global bitonic network.
constexpr u32 OS_CHUNKS = 16;

template <bool MANY>
__global__ __launch_bounds__(OS_THREADS, 4) void order_kernel(
    const u64* __restrict__ off, const u32* __restrict__ vstart, const u32* __restrict__ in_base32,
    const u64* __restrict__ in_base64,
    u32 P, u32 Q, int low, const u64* __restrict__ akey,
    const u64* __restrict__ arval, const u64* __restrict__ asval, u64* __restrict__ bkey,
    u64* __restrict__ brval, u64* __restrict__ bsval, u64* __restrict__ accum, u32 defer_rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  OrderSmem& sm = *reinterpret_cast<OrderSmem*>(smem_raw);
  const int tid = threadIdx.x;
  const int bsh = low - OS_LOGB;  // bucket = key bits [low-12, low)
  constexpr u32 OS_CHUNK_ROWS = OrderLimits<MANY>::CHUNK_ROWS;
  OrderBuckets bf_key;
  bf_key.by_sval = 0;
  bf_key.bsh = bsh;
  bf_key.svsh = 0;
  bf_key.svbase = 0;
  for (u32 p = blockIdx.x; p < P; p += gridDim.x) {
    // output segment: partition p's items are off[p*Q .. (p+1)*Q), or -- when oversized probe partitions
    // were split into virtual partitions -- the virtual partitions vstart[p] .. vstart[p+1]
    const u64 i0 = vstart ? (u64)vstart[p] : (u64)p * Q, i1 = vstart ? (u64)vstart[p + 1] : ((u64)p + 1) * Q;
    const u64 ob = off[i0], L64 = off[i1] - ob;
    const u64 b = in_base64 ? in_base64[p] : (in_base32 ? (u64)in_base32[p] : ob);  // input segment start
    if (L64 == 0) continue;
    bool done = false;
    u64 key[OS_ROWS], rv[OS_ROWS], sv[OS_ROWS];
    if (L64 <= OS_CAP) {
      const u32 L = (u32)L64;
#pragma unroll
      for (int k = 0; k < OS_ROWS; k++) {
        const u32 i = k * OS_THREADS + tid;
        key[k] = rv[k] = sv[k] = 0;
        if (i < L) {
          key[k] = akey[b + i];
          rv[k] = arval[b + i];
          sv[k] = asval[b + i];
        }
      }
      done = order_sort_registers<MANY>(sm, L, key, rv, sv, bf_key, ob, bkey, brval, bsval, tid);
      if (MANY && !done) {  // a bucket too long: one key's run?  then spread the rows by sval (uniform for the workgroup)
        __syncthreads();
        if (tid == 0) {
          sm.mixed = 0;
          sm.svmin = ~0ull;
          sm.svmax = 0;
        }
        __syncthreads();
        const u64 k0 = akey[b], r0 = arval[b];
        bool mixed = false;
        u64 mn = ~0ull, mx = 0;
#pragma unroll
        for (int k = 0; k < OS_ROWS; k++) {
          if ((u32)(k * OS_THREADS + tid) < L) {
            mixed |= key[k] != k0 || rv[k] != r0;
            mn = sv[k] < mn ? sv[k] : mn;
            mx = sv[k] > mx ? sv[k] : mx;
          }
        }
        if (mixed) sm.mixed = 1;
        atomicMin(&sm.svmin, (unsigned long long)mn);
        atomicMax(&sm.svmax, (unsigned long long)mx);
        __syncthreads();
        if (sm.mixed == 0 && sm.svmax > sm.svmin)
          done = order_sort_registers<MANY>(sm, L, key, rv, sv, order_buckets_by_sval(sm.svmin, sm.svmax, bsh), ob, bkey, brval, bsval, tid);
      }
    } else if (L64 <= (u64)OS_CHUNKS * OS_CHUNK_ROWS) {
      const u32 L = (u32)L64;
      // bucket counts of the whole segment -> bucket starts (kept in bstart32, aliased on sidx + bstart:
      // they are free until a chunk is sorted, so the starts are rebuilt per chunk from the counts instead)
      OrderBuckets bf = bf_key;
      u32 st[4];  // start of this thread's four buckets in the sorted segment
      for (int attempt = 0; attempt < (MANY ? 2 : 1); attempt++) {
        __syncthreads();
        for (u32 i = tid; i < (u32)OS_NB; i += OS_THREADS) sm.bcur[i] = 0;
        if (tid == 0) sm.fallback = 0;
        __syncthreads();
        if (bf.by_sval) {
          for (u32 i = tid; i < L; i += OS_THREADS) atomicAdd(&sm.bcur[order_bucket(bf, 0, asval[b + i])], 1u);
        } else {
          for (u32 i = tid; i < L; i += OS_THREADS) atomicAdd(&sm.bcur[order_bucket(bf, akey[b + i], 0)], 1u);
        }
        __syncthreads();
        {
          u32 c[4], sum = 0, mx = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            c[q] = sm.bcur[tid * 4 + q];
            sum += c[q];
            mx = c[q] > mx ? c[q] : mx;
          }
          if (mx > OrderLimits<MANY>::MAXBUCKET) sm.fallback = 1;
          u32 tot;
          u32 ex = block_excl_scan_u32<OS_THREADS>(sum, sm.scratch, &tot);
#pragma unroll
          for (int q = 0; q < 4; q++) {
            st[q] = ex;
            ex += c[q];
          }
        }
        __syncthreads();
        if (sm.fallback == 0 || attempt == 1 || !MANY) break;
        // a bucket too long: is the segment one key's run?  then spread its rows by sval and count again
        if (tid == 0) {
          sm.mixed = 0;
          sm.svmin = ~0ull;
          sm.svmax = 0;
        }
        __syncthreads();
        {
          const u64 k0 = akey[b], r0 = arval[b];
          bool mixed = false;
          u64 mn = ~0ull, mx = 0;
          for (u32 i = tid; i < L; i += OS_THREADS) {
            mixed |= akey[b + i] != k0 || arval[b + i] != r0;
            const u64 x = asval[b + i];
            mn = x < mn ? x : mn;
            mx = x > mx ? x : mx;
          }
          if (mixed) sm.mixed = 1;
          atomicMin(&sm.svmin, (unsigned long long)mn);
          atomicMax(&sm.svmax, (unsigned long long)mx);
        }
        __syncthreads();
        if (sm.mixed != 0 || sm.svmax <= sm.svmin) break;  // (fallback stays set)
        bf = order_buckets_by_sval(sm.svmin, sm.svmax, bsh);
      }
      if (sm.fallback == 0) {
        // chunk c = the buckets whose start lies in [c * OS_CHUNK_ROWS, (c+1) * OS_CHUNK_ROWS): at most
        // OS_CHUNK_ROWS + OS_MAXBUCKET = OS_CAP rows.  chunk_of[bucket] goes to sidx (u16, free here).
#pragma unroll
        for (int q = 0; q < 4; q++) sm.sidx[tid * 4 + q] = (u16)(st[q] / OS_CHUNK_ROWS);
        const u32 nchunks = (L + OS_CHUNK_ROWS - 1) / OS_CHUNK_ROWS;
        for (u32 cnk = 0; cnk < nchunks; cnk++) {
          __syncthreads();
          if (tid == 0) {
            sm.scratch[0] = 0;        // rows gathered so far
            sm.scratch[1] = 0xFFFFFFFFu;  // smallest bucket start of the chunk = where its rows begin
          }
          __syncthreads();
          // first row of the chunk in the sorted segment: the smallest start among its non-empty buckets
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (st[q] / OS_CHUNK_ROWS == cnk && sm.bcur[tid * 4 + q] != 0) atomicMin(&sm.scratch[1], st[q]);
          // gather the chunk's rows (any order) into LDS
          for (u32 i0r = 0; i0r < L; i0r += OS_THREADS) {
            const u32 i = i0r + tid;
            if (i < L) {
              const u64 kk = akey[b + i], ss = asval[b + i];
              if (sm.sidx[order_bucket(bf, kk, ss)] == (u16)cnk) {
                const u32 slot = atomicAdd(&sm.scratch[0], 1u);
                sm.stage[slot] = kk;
                sm.srv[slot] = arval[b + i];
                sm.ssv[slot] = ss;
              }
            }
          }
          __syncthreads();
          const u32 n_c = sm.scratch[0], base_c = sm.scratch[1];
          __syncthreads();
          if (n_c == 0) continue;
#pragma unroll
          for (int k = 0; k < OS_ROWS; k++) {
            const u32 i = k * OS_THREADS + tid;
            key[k] = rv[k] = sv[k] = 0;
            if (i < n_c) {
              key[k] = sm.stage[i];
              rv[k] = sm.srv[i];
              sv[k] = sm.ssv[i];
            }
          }
          // (the sort below overwrites sidx and bcur: chunk_of and the counts are rebuilt from st[] / c after it)
          u32 cnt_keep[4];
#pragma unroll
          for (int q = 0; q < 4; q++) cnt_keep[q] = sm.bcur[tid * 4 + q];
          order_sort_registers<MANY>(sm, n_c, key, rv, sv, bf, ob + base_c, bkey, brval, bsval, tid);
#pragma unroll
          for (int q = 0; q < 4; q++) {
            sm.sidx[tid * 4 + q] = (u16)(st[q] / OS_CHUNK_ROWS);
            sm.bcur[tid * 4 + q] = cnt_keep[q];
          }
        }
        __syncthreads();
        done = true;
      }
    }
    if (!done) {  // uniform for the workgroup
      const u64 n = L64;
      __syncthreads();
      for (u64 i = tid; i < n; i += OS_THREADS) {
        bkey[ob + i] = akey[b + i];
        brval[ob + i] = arval[b + i];
        bsval[ob + i] = asval[b + i];
      }
      __syncthreads();
      if (defer_rows && n > defer_rows) {
        // hundreds of thousands of rows of few keys: a bitonic network run by one workgroup would take
        // seconds.  Leave the segment as copied; the host sorts the whole result instead (api.hip).
        if (tid == 0) atomicOr(reinterpret_cast<unsigned long long*>(&accum[ACC_ERR]), (unsigned long long)ERR_ORDER_DEFER);
      } else if (n >= 2 && n <= 0x7FFFFFFFull) {
        order_network_global(bkey + ob, brval + ob, bsval + ob, (u32)n, tid);
      } else if (n >= 2) {  // cannot be sorted here nor deferred (row indices of the global sorts are 32-bit)
        if (tid == 0) atomicOr(reinterpret_cast<unsigned long long*>(&accum[ACC_ERR]), (unsigned long long)ERR_ORDER_FAIL);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Skewed probe sides: a partition with far more probe rows than the others would keep ONE workgroup
// busy long after the rest of the grid has drained.  Such partitions are cut into slices of
// slice_rows probe rows; every slice becomes a "virtual partition" with the same build range, so the
// kernels above treat it like any other partition (the table is rebuilt per slice: small next to
// the probe rows it serves).  vstart[p] = first virtual partition of partition p (P + 1 entries).
// ---------------------------------------------------------------------------------------------
// build_thr != 0 (modes that enumerate every matching pair, not first-wins): a partition with more than
// build_thr build rows -- thousands of copies of a key -- is cut on the BUILD side too (slices of build_slice
// rows, each joined with all of the partition's probe slices), so that one hot key's cross product is written
// by many workgroups.
__device__ __forceinline__ void split_shape(const u32* r_off, const u32* s_off, u32 p, u32 thr_rows, u32 slice_rows,
                                            u32 build_thr, u32 build_slice, u32* ns, u32* nbs) {
  const u32 np = s_off[p + 1] - s_off[p], nb = r_off[p + 1] - r_off[p];
  *ns = (np > thr_rows) ? (np + slice_rows - 1) / slice_rows : 1u;
  *nbs = (build_thr && nb > build_thr && np) ? (nb + build_slice - 1) / build_slice : 1u;
  // thousands of copies of a build key: every probe row that hits it yields thousands of result rows, written
  // by the one thread that owns the probe row -- so give such a partition many small probe slices too
  if (*nbs > 1 && np > 256u) *ns = (np + 255u) / 256u;
}
__global__ __launch_bounds__(1024) void split_count_kernel(const u32* __restrict__ r_off, const u32* __restrict__ s_off,
                                                           u32 P, u32 thr_rows, u32 slice_rows, u32 build_thr,
                                                           u32 build_slice, u32 cap_v, u32* __restrict__ vstart,
                                                           u32* __restrict__ nv_out) {
  __shared__ u32 scratch[17];
  __shared__ u32 carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (u32 base = 0; base < P; base += 1024) {
    const u32 p = base + threadIdx.x;
    u32 ns = 0;
    if (p < P) {
      u32 a, b;
      split_shape(r_off, s_off, p, thr_rows, slice_rows, build_thr, build_slice, &a, &b);
      const unsigned long long prod = (unsigned long long)a * b;
      ns = prod > cap_v ? cap_v + 1 : (u32)prod;  // (an overflowing total makes the host ignore the split)
    }
    u32 tot;
    const u32 ex = block_excl_scan_u32<1024>(ns, scratch, &tot);
    if (p < P) vstart[p] = carry_s + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    vstart[P] = carry_s;
    *nv_out = carry_s;
  }
}
__global__ void split_fill_kernel(const u32* __restrict__ r_off, const u32* __restrict__ s_off, u32 P, u32 thr_rows,
                                  u32 slice_rows, u32 build_thr, u32 build_slice, u32 cap_v,
                                  const u32* __restrict__ vstart,
                                  u32* __restrict__ vr_beg, u32* __restrict__ vr_end, u32* __restrict__ vs_beg,
                                  u32* __restrict__ vs_end) {
  const u32 nv = vstart[P];
  for (u32 v = blockIdx.x * blockDim.x + threadIdx.x; v < nv && v < cap_v; v += gridDim.x * blockDim.x) {
    u32 lo = 0, hi = P;  // last p with vstart[p] <= v
    while (hi - lo > 1) {
      const u32 mid = (lo + hi) >> 1;
      if (vstart[mid] <= v) lo = mid; else hi = mid;
    }
    const u32 p = lo, q = v - vstart[p];
    u32 ns, nbs;
    split_shape(r_off, s_off, p, thr_rows, slice_rows, build_thr, build_slice, &ns, &nbs);
    const u32 qb = q / ns, qs = q % ns;  // build slice major, probe slice minor
    const u32 sb = s_off[p], np = s_off[p + 1] - sb, rb = r_off[p], nb = r_off[p + 1] - rb;
    if (nbs == 1) {
      vr_beg[v] = rb;
      vr_end[v] = rb + nb;
    } else {
      const u32 b = qb * build_slice, e = (b + build_slice < nb) ? b + build_slice : nb;
      vr_beg[v] = rb + b;
      vr_end[v] = rb + e;
    }
    if (ns == 1) {
      vs_beg[v] = sb;
      vs_end[v] = sb + np;
    } else {
      const u32 step = (np + ns - 1) / ns;  // (a build-heavy partition uses 256-row probe slices)
      const u32 b = qs * step, e = (b + step < np) ? b + step : np;
      vs_beg[v] = sb + (b < np ? b : np);
      vs_end[v] = sb + e;
    }
  }
}
hipError_t launch_split_parts(const u32* r_off, const u32* s_off, u32 P, u32 thr_rows, u32 slice_rows, u32 build_thr,
                              u32 build_slice, u32 cap_v, u32* vstart, u32* vr_beg, u32* vr_end, u32* vs_beg,
                              u32* vs_end, u32* nv_out, hipStream_t st) {
  hipLaunchKernelGGL(split_count_kernel, dim3(1), dim3(1024), 0, st, r_off, s_off, P, thr_rows, slice_rows, build_thr,
                     build_slice, cap_v, vstart, nv_out);
  hipLaunchKernelGGL(split_fill_kernel, dim3((cap_v + 255) / 256), dim3(256), 0, st, r_off, s_off, P, thr_rows,
                     slice_rows, build_thr, build_slice, cap_v, vstart, vr_beg, vr_end, vs_beg, vs_end);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
template <int MODE, bool FIRST, bool EXTRA>
static hipError_t launch_probe_t(const ProbeArgs& a, int grid, hipStream_t st) {
  static SmemAttrOnce attr_once;
  if (hipError_t e = ensure_max_smem(attr_once, reinterpret_cast<const void*>(probe_kernel<MODE, FIRST, EXTRA>), (size_t)sizeof(ProbeSmem)); e != hipSuccess) return e;
  hipLaunchKernelGGL((probe_kernel<MODE, FIRST, EXTRA>), dim3(grid), dim3(PB_THREADS),
                     sizeof(ProbeSmem), st, a);
  return hipGetLastError();
}

template <int THREADS, int LOG_NB, bool PCOUNT, bool SLAB = false, int OUT = 0>
static hipError_t launch_fast_t(const ProbeArgs& a, u32* irregular, u32* n_irregular, int grid,
                                hipStream_t st) {
  typedef FastSmem<THREADS, LOG_NB> Smem;
  static SmemAttrOnce attr_once;
  if (hipError_t e = ensure_max_smem(attr_once, reinterpret_cast<const void*>(probe_count_fast_kernel<THREADS, LOG_NB, PCOUNT, SLAB, OUT>), (size_t)sizeof(Smem)); e != hipSuccess) return e;
  if ((u32)grid > a.P) grid = (int)a.P;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((probe_count_fast_kernel<THREADS, LOG_NB, PCOUNT, SLAB, OUT>), dim3(grid), dim3(THREADS),
                     sizeof(Smem), st, a, irregular, n_irregular);
  return hipGetLastError();
}

// big == false: partitions of <= 2560 rows (3 workgroups/CU); big == true: <= 5120 rows (1/CU)
hipError_t launch_probe_count_fast(const ProbeArgs& a, u32* irregular, u32* n_irregular, bool big,
                                   bool per_partition_counts, int num_cus, hipStream_t st) {
  if (per_partition_counts) {
    if (big) return launch_fast_t<1024, BIG_LOG_NB, true>(a, irregular, n_irregular, num_cus * 4, st);
    return launch_fast_t<512, 11, true>(a, irregular, n_irregular, num_cus * 3 * 4, st);
  }
  if (big) return launch_fast_t<1024, BIG_LOG_NB, false>(a, irregular, n_irregular, num_cus * 4, st);
  return launch_fast_t<512, 11, false>(a, irregular, n_irregular, num_cus * 3 * 4, st);
}

// Slab layout: the pipelined kernels read P * SLAB_KB counts and piece p * SLAB_KB + j at row (p * SLAB_KB + j) * cap
// of each side -- refuse operands that were allocated for anything less
static bool slab_operands_ok(const ProbeArgs& a) {
  const u64 pieces = (u64)a.P * SLAB_KB;
  return a.r_cnt && a.s_cnt && a.r_cnt_n >= pieces && a.s_cnt_n >= pieces && a.r_rows >= pieces * a.r_cap &&
         a.s_rows >= pieces * a.s_cap;
}

// unique-build-key write mode (ordered joins); slab or dense layout
hipError_t launch_probe_write_uniq(const ProbeArgs& a, bool slab, int num_cus, hipStream_t st) {
  if (slab && !slab_operands_ok(a)) return hipErrorInvalidValue;
  if (slab) return launch_fast_t<1024, BIG_LOG_NB, false, true, 1>(a, nullptr, nullptr, num_cus * 4, st);
  return launch_fast_t<1024, BIG_LOG_NB, false, false, 1>(a, nullptr, nullptr, num_cus * 4, st);
}

// ordered unique-key write in one pass (probe_write_sorted_kernel); lookback: P + 1 words, zeroed by the caller
template <bool SLAB, bool FK, bool EXTRA, bool PB = false>
static hipError_t launch_sorted_t(const ProbeArgs& a, u64* lookback, bool chained, int key_low, int grid, hipStream_t st) {
  typedef typename std::conditional<FK, SortedFkSmem<1024>, SortedSmem<1024>>::type Smem;
  static_assert(sizeof(Smem) <= 160 * 1024, "one workgroup's LDS");
  if constexpr (FK && !PB) {
    if (a.extra & 4u) return launch_sorted_t<SLAB, FK, EXTRA, true>(a, lookback, chained, key_low, grid, st);  // payload buckets
  }
  static SmemAttrOnce attr_once;
  const void* fn = reinterpret_cast<const void*>(probe_write_sorted_kernel<1024, SLAB, FK, EXTRA, FP_ROWS, SWF_CAPB, SWF_LOGB, PB>);
  if (hipError_t e = ensure_max_smem(attr_once, fn, sizeof(Smem)); e != hipSuccess) return e;
  hipLaunchKernelGGL((probe_write_sorted_kernel<1024, SLAB, FK, EXTRA, FP_ROWS, SWF_CAPB, SWF_LOGB, PB>), dim3(grid), dim3(1024), sizeof(Smem), st, a,
                     lookback, key_low, chained);
  return hipGetLastError();
}
// the foreign-key form's other shapes: 512 threads x 6 rows (two workgroups per CU), 1024 threads x 6 rows
template <int THREADS, int ROWS, int CAPB_, int LOGB_, bool SLAB, bool EXTRA, bool PB = false>
static hipError_t launch_sorted_shape_t(const ProbeArgs& a, u64* lookback, bool chained, int key_low, int grid, hipStream_t st) {
  typedef SortedFkSmem<THREADS, ROWS, CAPB_, LOGB_> Smem;
  if constexpr (!PB) {
    if (a.extra & 4u) return launch_sorted_shape_t<THREADS, ROWS, CAPB_, LOGB_, SLAB, EXTRA, true>(a, lookback, chained, key_low, grid, st);
  }
  static SmemAttrOnce attr_once;
  const void* fn = reinterpret_cast<const void*>(probe_write_sorted_kernel<THREADS, SLAB, true, EXTRA, ROWS, CAPB_, LOGB_, PB>);
  if (hipError_t e = ensure_max_smem(attr_once, fn, sizeof(Smem)); e != hipSuccess) return e;
  hipLaunchKernelGGL((probe_write_sorted_kernel<THREADS, SLAB, true, EXTRA, ROWS, CAPB_, LOGB_, PB>), dim3(grid), dim3(THREADS),
                     sizeof(Smem), st, a, lookback, key_low, chained);
  return hipGetLastError();
}
template <int THREADS, int ROWS, int CAPB_, int LOGB_>
static hipError_t launch_sorted_shape(const ProbeArgs& a, bool slab, u64* lookback, bool chained, int key_low, int grid, hipStream_t st) {
  if (slab) return (a.extra & 1u) ? launch_sorted_shape_t<THREADS, ROWS, CAPB_, LOGB_, true, true>(a, lookback, chained, key_low, grid, st)
                                  : launch_sorted_shape_t<THREADS, ROWS, CAPB_, LOGB_, true, false>(a, lookback, chained, key_low, grid, st);
  return (a.extra & 1u) ? launch_sorted_shape_t<THREADS, ROWS, CAPB_, LOGB_, false, true>(a, lookback, chained, key_low, grid, st)
                        : launch_sorted_shape_t<THREADS, ROWS, CAPB_, LOGB_, false, false>(a, lookback, chained, key_low, grid, st);
}
template <bool SLAB, bool FK>
static hipError_t launch_sorted_x(const ProbeArgs& a, u64* lookback, bool chained, int key_low, int grid, hipStream_t st) {
  return (a.extra & 1u) ? launch_sorted_t<SLAB, FK, true>(a, lookback, chained, key_low, grid, st)
                        : launch_sorted_t<SLAB, FK, false>(a, lookback, chained, key_low, grid, st);
}
// fk: the probe keys may repeat (SortedFkSmem); shape (fk only): 0 = 5120 probe / 4608 build rows per partition,
// 1 = the small shape (3072 / 2048, two workgroups per CU), 2 = the wide shape (6144 / 2560) -- a partition beyond the
// shape's capacities makes the kernel give up with "a partition does not fit"; key_low: the partition id's lowest key bit
hipError_t launch_probe_write_sorted(const ProbeArgs& a, bool slab, bool fk, int shape, u64* lookback, bool chained, int key_low,
                                     int num_cus, hipStream_t st) {
  if (key_low < 0 || key_low > 63) return hipErrorInvalidValue;  // (fewer key bits than buckets: key_bucket shifts left)
  if (slab && !slab_operands_ok(a)) return hipErrorInvalidValue;
  int grid = (fk && shape == 1) ? 2 * num_cus : num_cus;  // workgroups that fit a CU (LDS); partitions by stride or ticket
  if ((u32)grid > a.P) grid = (int)a.P;
  if (grid < 1) grid = 1;
  if (fk && shape == 1)
    return launch_sorted_shape<SWF_HALF_THREADS, SWF_HALF_ROWS, SWF_HALF_CAPB, SWF_HALF_LOGB>(a, slab, lookback, chained, key_low, grid, st);
  if (fk && shape == 2)
    return launch_sorted_shape<1024, SWF_WIDE_ROWS, SWF_WIDE_CAPB, SWF_LOGB>(a, slab, lookback, chained, key_low, grid, st);
  if (slab) return fk ? launch_sorted_x<true, true>(a, lookback, chained, key_low, grid, st)
                      : launch_sorted_x<true, false>(a, lookback, chained, key_low, grid, st);
  return fk ? launch_sorted_x<false, true>(a, lookback, chained, key_low, grid, st)
            : launch_sorted_x<false, false>(a, lookback, chained, key_low, grid, st);
}

// np of every slab partition (sum of its 4 piece counts) as u64, for the exclusive scan that gives
// each partition's first output slot in the unique-key write mode
__global__ void slab_np_kernel(const u32* __restrict__ cnt, u32 P, u64* __restrict__ out) {
  u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P) {
    const u32* q = cnt + (u64)p * SLAB_KB;
    out[p] = (u64)q[0] + q[1] + q[2] + q[3];
  }
}
hipError_t launch_slab_np(const u32* cnt, u32 P, u64* out, hipStream_t st) {
  hipLaunchKernelGGL(slab_np_kernel, dim3((P + 255) / 256), dim3(256), 0, st, cnt, P, out);
  return hipGetLastError();
}

// count mode with checksums / sum of all probe payloads (a.extra bit 0) and first-wins (bit 1)
hipError_t launch_probe_count_ext(const ProbeArgs& a, u32* irregular, u32* n_irregular, bool slab,
                                  int num_cus, hipStream_t st) {
  if (slab && !slab_operands_ok(a)) return hipErrorInvalidValue;
  if (slab) return launch_fast_t<1024, BIG_LOG_NB, false, true, 2>(a, nullptr, nullptr, num_cus * 4, st);
  return launch_fast_t<1024, BIG_LOG_NB, false, false, 2>(a, irregular, n_irregular, num_cus * 4, st);
}

hipError_t launch_probe_count_slab(const ProbeArgs& a, int num_cus, hipStream_t st) {
  if (!slab_operands_ok(a)) return hipErrorInvalidValue;
  return launch_fast_t<1024, BIG_LOG_NB, false, true>(a, nullptr, nullptr, num_cus * 4, st);
}

hipError_t launch_probe(const ProbeArgs& a, int mode, bool first_wins, bool extra, int grid,
                        hipStream_t st) {
  if (!a.item_list && (u64)grid > (u64)a.P * a.Q) grid = (int)((u64)a.P * a.Q);
  if (grid < 1) grid = 1;
  // probe side in slabs (probe-heavy count joins): item w = p * Q + q is piece w -- count w, rows [w * cap, + count)
  if (a.s_ppi) {  // pieces of one slab pass: P * s_wa pieces, Q items of s_ppi pieces cover a partition's s_wa
    if (!a.s_cnt || (mode != 0 && mode != 3) || (mode == 3 && (!a.out_key || !a.out_rval || !a.out_sval || a.out_cap == 0)) || (first_wins && !a.matched) || a.s_wa == 0 || (u64)a.P * a.s_wa * a.s_cap > 0xFFFFFFFFull || (u64)a.Q * a.s_ppi < a.s_wa || (u64)(a.Q - 1) * a.s_ppi >= a.s_wa ||
        a.s_cnt_n < (u64)a.P * a.s_wa || a.s_rows < (u64)a.P * a.s_wa * a.s_cap)
      return hipErrorInvalidValue;
  } else if (a.s_cnt && (a.s_cnt_n < (u64)a.P * a.Q || a.s_rows < (u64)a.P * a.Q * a.s_cap || (u64)a.P * a.Q * a.s_cap > 0xFFFFFFFFull))
    return hipErrorInvalidValue;
#define HMJ_DISPATCH(M)                                                          \
  if (first_wins)                                                                \
    return extra ? launch_probe_t<M, true, true>(a, grid, st)                    \
                 : launch_probe_t<M, true, false>(a, grid, st);                  \
  else                                                                           \
    return extra ? launch_probe_t<M, false, true>(a, grid, st)                   \
                 : launch_probe_t<M, false, false>(a, grid, st);
  if (mode == 3 && !a.s_ppi) return hipErrorInvalidValue;  // (count + write by cursor exists for the piece walk only)
  if (mode == 0) {
    HMJ_DISPATCH(0)
  } else if (mode == 3) {
    HMJ_DISPATCH(3)
  } else if (mode == 1) {
    HMJ_DISPATCH(1)
  } else {
    if (first_wins) return launch_probe_t<2, true, false>(a, grid, st);
    return launch_probe_t<2, false, false>(a, grid, st);
  }
#undef HMJ_DISPATCH
}

int probe_default_grid(int num_cus) { return num_cus * 4; }

hipError_t launch_scan_u64(const u64* in, u64* out_excl, u32 n, hipStream_t st) {
  hipLaunchKernelGGL(scan_u64_kernel, dim3(1), dim3(1024), 0, st, in, out_excl, n);
  return hipGetLastError();
}

template <bool MANY>
static hipError_t launch_order_t(const u64* part_out_off, const u32* vstart, const u32* in_base32, const u64* in_base64,
                                 u32 P, u32 Q, int low, const u64* akey, const u64* arval, const u64* asval, u64* bkey,
                                 u64* brval, u64* bsval, u64* accum, u32 defer_rows, int grid, hipStream_t st) {
  static SmemAttrOnce attr_once;
  if (hipError_t e = ensure_max_smem(attr_once, reinterpret_cast<const void*>(order_kernel<MANY>), (size_t)sizeof(OrderSmem)); e != hipSuccess) return e;
  if ((u32)grid > P) grid = (int)P;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((order_kernel<MANY>), dim3(grid), dim3(OS_THREADS), sizeof(OrderSmem), st, part_out_off, vstart,
                     in_base32, in_base64, P, Q, low, akey, arval, asval, bkey, brval, bsval, accum, defer_rows);
  return hipGetLastError();
}
// many_per_key: the result has clearly more rows than the build side has keys (fan-out)
// ---------------------------------------------------------------------------------------------
// Ordered result written IN ORDER, partition by partition, instead of written in probe order and sorted afterwards
// (order_kernel: 16 of the 19 ms of a join of 8 x 8 rows per key with 1.3 * 10^8 result rows), for what the one-pass unique-key
// forms do not take: duplicate build keys, and foreign-key joins whose runs are long (their in-run ranking is linear in the
// run).  The operator's order is (key, rval, sval).  Per partition (a key range; one workgroup):
//   1. the build rows are SORTED in LDS by (key, rval): a row goes to its bucket (the key bits under the partition bits, in
//      arrival order), then ranks itself among the handful of rows of its bucket and moves to its place once everybody has read;
//   2. every probe row looks its key up in the sorted build keys (binary search): its KEY ID is the index of the key's first
//      build row; rows without a match are dropped here;
//   3. the matched probe rows are sorted by (key id, sval) the same way, with buckets over BOTH: a partition with few keys
//      (2^b >= build rows) spends its 11 bucket bits as b bits of key id and 11 - b bits of the payload's position in the
//      partition's payload range -- a foreign-key partition of 16 keys x 256 rows ranks inside buckets of a few rows instead of
//      inside runs of 256; a partition with thousands of keys buckets by key id alone;
//   4. every build row takes its key's run of sorted probe rows (bucket boundaries where a key owns whole buckets): that many
//      result rows; an exclusive scan over the build rows gives each its first result row;
//   5. the result rows are written by OUTPUT index: lane <-> result row, so the stores coalesce; a row finds its build row by a
//      binary search over the scan (13 LDS reads) and its probe row by the offset inside the run.  Build rows that agree in key
//      AND payload share one block in which every probe row appears g times in a row (the order by sval runs across them).
// The partition's first result row comes from the count pass's scan (part_out_off), which also made the sums.  A partition
// that does not fit (XE_CAP rows a side) raises ERR_FASTPATH and reports its size: the caller plans one more bit or writes
// and sorts as before.
constexpr int XE_THREADS = 1024, XE_ROWS = 5, XE_CAP = 4608, XE_LOGB = 11;
struct ExpandSmem {
  u64 bkey[XE_CAP];  // build keys, sorted
  u64 bval[XE_CAP];  // build payloads, same order
  u64 sval[XE_CAP];  // payloads of the matched probe rows, sorted by (key id, payload)
  u32 off[XE_CAP];   // first result row of build row i
  u16 sid[XE_CAP];   // key id of every sorted probe row
  u16 s0[XE_CAP];    // where build row i's run of probe rows starts; bit 15: the row equals its predecessor (key and payload)
  u32 cnt[1 << XE_LOGB];
  u16 bstart[(1 << XE_LOGB) + 2];
  u32 scratch[XE_THREADS / kWave + 1];
  u32 flag;
  unsigned long long vmin, vmax;
};
static_assert(sizeof(ExpandSmem) <= 160 * 1024, "one workgroup per CU");
static_assert(XE_THREADS * XE_ROWS >= XE_CAP && XE_CAP < 0x8000, "every row has a thread slot; row indices fit 15 bits");

// exclusive scan of sm.cnt[] -> sm.bstart[] (+ the total behind it); barriers before and after are the caller's
__device__ __forceinline__ u32 xe_scan_buckets(ExpandSmem& sm, int tid) {
  constexpr u32 NB = 1u << XE_LOGB, BPT = NB / XE_THREADS;
  u32 c[BPT], sum = 0;
#pragma unroll
  for (u32 q = 0; q < BPT; q++) {
    c[q] = sm.cnt[tid * BPT + q];
    sum += c[q];
  }
  u32 tot;
  u32 ex = block_excl_scan_u32<XE_THREADS>(sum, sm.scratch, &tot, tid);
#pragma unroll
  for (u32 q = 0; q < BPT; q++) {
    sm.bstart[tid * BPT + q] = (u16)ex;
    ex += c[q];
  }
  if (tid == 0) sm.bstart[NB] = (u16)tot;
  return tot;
}

// build rows t[0 .. XE_ROWS) of this thread (row j = k * XE_THREADS + tid, valid while j < n) -> bkey[], bval[] sorted by (key, val)
__device__ __forceinline__ void xe_sort_build(ExpandSmem& sm, Tup (&t)[XE_ROWS], u32 n, int bsh, int tid) {
  constexpr u32 NB = 1u << XE_LOGB, BPT = NB / XE_THREADS;
  const u32 mask = NB - 1;
#pragma unroll
  for (u32 q = 0; q < BPT; q++) sm.cnt[tid * BPT + q] = 0;
  lds_barrier();
  u32 arr[XE_ROWS], bk[XE_ROWS];
#pragma unroll
  for (int k = 0; k < XE_ROWS; k++) {
    const u32 j = (u32)k * XE_THREADS + tid;
    bk[k] = key_bucket(t[k].key, bsh, mask);
    arr[k] = j < n ? atomicAdd(&sm.cnt[bk[k]], 1u) : 0u;
  }
  lds_barrier();
  xe_scan_buckets(sm, tid);
  lds_barrier();
#pragma unroll
  for (int k = 0; k < XE_ROWS; k++) {
    const u32 j = (u32)k * XE_THREADS + tid;
    if (j < n) {
      const u32 slot = sm.bstart[bk[k]] + arr[k];
      sm.bkey[slot] = t[k].key;
      sm.bval[slot] = t[k].val;
    }
  }
  lds_barrier();
  // slot j's row ranks itself inside its bucket: rows with a smaller (key, val), and equal ones in earlier slots
  u32 pos[XE_ROWS];
#pragma unroll
  for (int k = 0; k < XE_ROWS; k++) {
    const u32 j = (u32)k * XE_THREADS + tid;
    pos[k] = 0xFFFFFFFFu;
    if (j < n) {
      const u64 key = sm.bkey[j], val = sm.bval[j];
      t[k].key = key;
      t[k].val = val;
      const u32 b = key_bucket(key, bsh, mask), b0 = sm.bstart[b], b1 = sm.bstart[b + 1];
      u32 r = 0;
      for (u32 i = b0; i < b1; i++) {
        const u64 k2 = sm.bkey[i], v2 = sm.bval[i];
        r += (k2 < key || (k2 == key && (v2 < val || (v2 == val && i < j)))) ? 1u : 0u;
      }
      pos[k] = b0 + r;
    }
  }
  lds_barrier();
#pragma unroll
  for (int k = 0; k < XE_ROWS; k++) {
    if (pos[k] != 0xFFFFFFFFu) {
      sm.bkey[pos[k]] = t[k].key;
      sm.bval[pos[k]] = t[k].val;
    }
  }
  lds_barrier();
}

__global__ __launch_bounds__(XE_THREADS) void probe_expand_ordered_kernel(ProbeArgs a, int key_low) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  ExpandSmem& sm = *reinterpret_cast<ExpandSmem*>(smem_raw);
  const Tup* __restrict__ R = static_cast<const Tup*>(a.R);
  const Tup* __restrict__ S = static_cast<const Tup*>(a.S);
  const int tid = threadIdx.x, lane = tid & 63;
  const int bsh = key_low - XE_LOGB;
  constexpr u32 NB = 1u << XE_LOGB, BPT = NB / XE_THREADS, NONE = 0xFFFFu;
  bool toobig = false;
  if (tid == 0) sm.flag = 0;
  __syncthreads();
  for (u32 p = blockIdx.x; p < a.P; p += gridDim.x) {
    const u32 rb = a.r_off[p], nb = a.r_off[p + 1] - rb, sb = a.s_off[p], np = a.s_off[p + 1] - sb;
    const u64 ob = a.part_out_off[p], total64 = a.part_out_off[p + 1] - ob;
    if (total64 == 0) continue;  // (uniform)
    if (nb > (u32)XE_CAP || np > (u32)XE_CAP || total64 > 0xFFFFFFFFull) {
      toobig = true;
      // (the caller decides from the biggest partition whether one more radix bit would make them all fit)
      if (tid == 0) atomicMax(reinterpret_cast<unsigned long long*>(&a.accum[ACC_PAD]), (unsigned long long)(total64 > 0xFFFFFFFFull ? 0xFFFFFFFFull : (nb > np ? nb : np)));
      continue;
    }
    const u32 total = (u32)total64;
    Tup br[XE_ROWS], pr[XE_ROWS];
#pragma unroll
    for (int k = 0; k < XE_ROWS; k++) {
      const u32 j = (u32)k * XE_THREADS + tid;
      br[k] = load_stream(&R[(u64)rb + (j < nb ? j : nb - 1)]);
      pr[k] = load_stream(&S[(u64)sb + (j < np ? j : np - 1)]);
    }
    if (tid == 0) {
      sm.vmin = ~0ull;
      sm.vmax = 0;
    }
    xe_sort_build(sm, br, nb, bsh, tid);
    // ---- 2. key ids: lower bound of every probe key in the sorted build keys (the five searches of a thread in lockstep)
    u32 id[XE_ROWS];
    {
      u32 lo[XE_ROWS], hi[XE_ROWS];  // first index with bkey >= key lies in [lo, hi]
#pragma unroll
      for (int k = 0; k < XE_ROWS; k++) {
        lo[k] = 0;
        hi[k] = nb;
      }
#pragma unroll 1
      for (u32 span = nb; span > 0; span >>= 1) {  // (13 rounds at most: every round halves hi - lo)
#pragma unroll
        for (int k = 0; k < XE_ROWS; k++) {
          if (lo[k] < hi[k]) {
            const u32 mid = (lo[k] + hi[k]) >> 1;
            if (sm.bkey[mid] < pr[k].key) lo[k] = mid + 1; else hi[k] = mid;
          }
        }
      }
      u64 mn = ~0ull, mx = 0;
#pragma unroll
      for (int k = 0; k < XE_ROWS; k++) {
        const u32 j = (u32)k * XE_THREADS + tid;
        id[k] = NONE;
        if (j < np && lo[k] < nb && sm.bkey[lo[k]] == pr[k].key) {
          id[k] = lo[k];
          mn = pr[k].val < mn ? pr[k].val : mn;
          mx = pr[k].val > mx ? pr[k].val : mx;
        }
      }
#pragma unroll
      for (int o = kWave / 2; o > 0; o >>= 1) {
        const u64 a2 = __shfl_xor(mn, o, kWave), b2 = __shfl_xor(mx, o, kWave);
        mn = a2 < mn ? a2 : mn;
        mx = b2 > mx ? b2 : mx;
      }
      if (lane == 0 && mn <= mx) {
        atomicMin(&sm.vmin, (unsigned long long)mn);
        atomicMax(&sm.vmax, (unsigned long long)mx);
      }
    }
#pragma unroll
    for (u32 q = 0; q < BPT; q++) sm.cnt[tid * BPT + q] = 0;
    lds_barrier();
    // ---- 3. the matched probe rows sorted by (key id, payload): bucket = key id bits, then payload-position bits
    const u64 vmin = sm.vmin, vmax = sm.vmax;
    const int nbits = nb > 1 ? 32 - __builtin_clz(nb - 1) : 0;             // key ids < 2^nbits
    const int svb = nbits < XE_LOGB ? XE_LOGB - nbits : 0;                 // bucket bits left for the payload
    const int idsh = nbits > XE_LOGB ? nbits - XE_LOGB : 0;                // ... or several key ids per bucket
    const int rbits = vmax > vmin ? 64 - __builtin_clzll(vmax - vmin) : 0;
    const int vsh = rbits > svb ? rbits - svb : 0;
    auto bucket_of = [&](u32 kid, u64 v) -> u32 { return svb ? ((kid << svb) | (u32)((v - vmin) >> vsh)) : (kid >> idsh); };
    u32 arr[XE_ROWS], bk[XE_ROWS];
#pragma unroll
    for (int k = 0; k < XE_ROWS; k++) {
      bk[k] = 0;
      arr[k] = 0;
      if (id[k] != NONE) {
        bk[k] = bucket_of(id[k], pr[k].val);
        arr[k] = atomicAdd(&sm.cnt[bk[k]], 1u);
      }
    }
    lds_barrier();
    const u32 nm = xe_scan_buckets(sm, tid);  // matched probe rows
    lds_barrier();
#pragma unroll
    for (int k = 0; k < XE_ROWS; k++) {
      if (id[k] != NONE) {
        const u32 slot = sm.bstart[bk[k]] + arr[k];
        sm.sid[slot] = (u16)id[k];
        sm.sval[slot] = pr[k].val;
      }
    }
    lds_barrier();
    {
      u32 pos[XE_ROWS], kid[XE_ROWS];
      u64 val[XE_ROWS];
#pragma unroll
      for (int k = 0; k < XE_ROWS; k++) {
        const u32 j = (u32)k * XE_THREADS + tid;
        pos[k] = 0xFFFFFFFFu;
        kid[k] = 0;
        val[k] = 0;
        if (j < nm) {
          kid[k] = sm.sid[j];
          val[k] = sm.sval[j];
          const u32 b = bucket_of(kid[k], val[k]), b0 = sm.bstart[b], b1 = sm.bstart[b + 1];
          u32 r = 0;
          for (u32 i = b0; i < b1; i++) {
            const u32 k2 = sm.sid[i];
            const u64 v2 = sm.sval[i];
            r += (k2 < kid[k] || (k2 == kid[k] && (v2 < val[k] || (v2 == val[k] && i < j)))) ? 1u : 0u;
          }
          pos[k] = b0 + r;
        }
      }
      lds_barrier();
#pragma unroll
      for (int k = 0; k < XE_ROWS; k++) {
        if (pos[k] != 0xFFFFFFFFu) {
          sm.sid[pos[k]] = (u16)kid[k];
          sm.sval[pos[k]] = val[k];
        }
      }
      lds_barrier();
    }
    // ---- 4. every build row: its key's run of sorted probe rows
    u32 c[XE_ROWS], st[XE_ROWS], sum = 0;
#pragma unroll
    for (int k = 0; k < XE_ROWS; k++) {
      const u32 i = (u32)tid * XE_ROWS + k;  // (consecutive build rows per thread: the scan below is over them)
      c[k] = 0;
      st[k] = 0;
      if (i < nb) {
        const u64 key = sm.bkey[i];
        u32 kid = i;  // the key's first build row
        while (kid > 0 && sm.bkey[kid - 1] == key) kid--;
        // (a build row that equals its predecessor in key AND payload: the two rows' result rows are the same rows, and
        //  the order by sval runs across both -- see the copy-out)
        const u32 tie = (i > 0 && sm.bkey[i - 1] == key && sm.bval[i - 1] == sm.bval[i]) ? 0x8000u : 0u;
        u32 first, n;
        if (idsh == 0) {  // the key id owns whole buckets: its run is what lies between their boundaries
          first = sm.bstart[kid << svb];
          n = sm.bstart[(kid + 1) << svb] - first;
        } else {
          const u32 b = kid >> idsh, b0 = sm.bstart[b], b1 = sm.bstart[b + 1];
          first = b1;
          n = 0;
          for (u32 j = b0; j < b1; j++) {
            if (sm.sid[j] == kid) {
              first = j < first ? j : first;
              n++;
            }
          }
        }
        c[k] = n;
        st[k] = tie | first;
      }
      sum += c[k];
    }
    u32 tot;
    u32 ex = block_excl_scan_u32<XE_THREADS>(sum, sm.scratch, &tot, tid);
#pragma unroll
    for (int k = 0; k < XE_ROWS; k++) {
      const u32 i = (u32)tid * XE_ROWS + k;
      if (i < nb) {
        sm.off[i] = ex;
        sm.s0[i] = (u16)st[k];
      }
      ex += c[k];
    }
    if (tid == 0 && tot != total) sm.flag = 1;  // (the count pass and this kernel disagree: never expected; the caller falls back)
    lds_barrier();
    // ---- 5. result rows by output index.  Build row lo's rows are (its payload) x (its key's run of probe rows, by sval); g
    // build rows that agree in key and payload share ONE block of g * cs rows in which every probe row appears g times in a row.
    for (u32 o = (u32)tid; o < tot && tot == total; o += XE_THREADS) {
      u32 lo = 0, hi = nb;  // off[lo] <= o < off[hi] (off[nb] = tot, not stored)
#pragma unroll 1
      while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (sm.off[mid] <= o) lo = mid; else hi = mid;
      }
      const u32 cs = (lo + 1 < nb ? sm.off[lo + 1] : tot) - sm.off[lo];
      u32 q = 0, g = 1;  // lo is the q-th of g equal build rows
#pragma unroll 1
      while (sm.s0[lo - q] & 0x8000u) q++;
#pragma unroll 1
      while (lo + g - q < nb && (sm.s0[lo + g - q] & 0x8000u)) g++;
      const u32 x = (o - sm.off[lo]) + q * cs;
      const u32 s = ((u32)sm.s0[lo] & 0x7FFFu) + (g == 1 ? x : x / g);
      a.out_key[ob + o] = sm.bkey[lo];
      a.out_rval[ob + o] = sm.bval[lo];
      a.out_sval[ob + o] = sm.sval[s];
    }
    lds_barrier();  // (the next partition overwrites the arrays)
  }
  if (tid == 0 && sm.flag) toobig = true;
  if (__any(toobig) && (tid & 63) == 0) atomicOr(reinterpret_cast<unsigned long long*>(&a.accum[ACC_ERR]), (unsigned long long)ERR_FASTPATH);
}

// a: R / r_off, S / s_off (contiguous partitions, P of them), part_out_off (P + 1, from the count pass), out_*, accum
hipError_t launch_probe_expand_ordered(const ProbeArgs& a, int key_low, int num_cus, hipStream_t st) {
  if (!a.R || !a.S || !a.r_off || !a.s_off || !a.part_out_off || !a.out_key || !a.out_rval || !a.out_sval || !a.accum || a.P == 0)
    return hipErrorInvalidValue;
  static SmemAttrOnce attr_once;
  if (hipError_t e = ensure_max_smem(attr_once, reinterpret_cast<const void*>(probe_expand_ordered_kernel), sizeof(ExpandSmem)); e != hipSuccess) return e;
  u32 grid = (u32)num_cus;
  if (grid > a.P) grid = a.P;
  hipLaunchKernelGGL(probe_expand_ordered_kernel, dim3(grid), dim3(XE_THREADS), sizeof(ExpandSmem), st, a, key_low);
  return hipGetLastError();
}

hipError_t launch_order(const u64* part_out_off, const u32* vstart, const u32* in_base32, const u64* in_base64, u32 P, u32 Q,
                        int low, const u64* akey, const u64* arval, const u64* asval, u64* bkey, u64* brval,
                        u64* bsval, u64* accum, u32 defer_rows, bool many_per_key, int grid, hipStream_t st) {
  return many_per_key ? launch_order_t<true>(part_out_off, vstart, in_base32, in_base64, P, Q, low, akey, arval, asval,
                                             bkey, brval, bsval, accum, defer_rows, grid, st)
                      : launch_order_t<false>(part_out_off, vstart, in_base32, in_base64, P, Q, low, akey, arval, asval,
                                              bkey, brval, bsval, accum, defer_rows, grid, st);
}

}  // namespace hmj
