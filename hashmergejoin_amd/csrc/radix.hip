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
#include "hmj_dev.h"
#include "hmj_launch.h"

namespace hmj {

// ---------------------------------------------------------------------------------------------
// K1: per-worker digit histogram.  hist[d * nblk + worker].  Algorithmic traffic: 16 B/row read
// (the key shares its 16-B row with the payload, so whole lines are fetched).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RP_THREADS) void radix_hist_kernel(const Tup* __restrict__ in, u32 n,
                                                                int shift, u32 mask,
                                                                u32 rows_per_block,
                                                                u32* __restrict__ hist, u32 nblk) {
  __shared__ u32 h[RP_MAXD];
  const u32 D = mask + 1;
  for (u32 d = threadIdx.x; d < D; d += RP_THREADS) h[d] = 0;
  __syncthreads();
  const u64 begin = (u64)blockIdx.x * rows_per_block;
  u64 end = begin + rows_per_block;
  if (end > n) end = n;
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
      if (i < end) atomicAdd(&h[(u32)(k[r] >> shift) & mask], 1u);
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
// ---------------------------------------------------------------------------------------------
struct ScatterSmem {
  Tup stage[RP_TILE];
  u16 wcnt[RP_WAVES][RP_MAXD];  // per-wave digit counts, then exclusive prefix across waves
  u32 tile_off[RP_MAXD];        // start of digit d inside the staged tile
  u32 delta[RP_MAXD];           // global cursor of d minus tile_off[d]
  u32 cursor[RP_MAXD];          // next output row of digit d for this worker
  u32 scratch[RP_WAVES + 1];
};

__global__ __launch_bounds__(RP_THREADS, 4) void radix_scatter_kernel(
    const Tup* __restrict__ in, Tup* __restrict__ out, u32 n, int shift, int bits,
    u32 rows_per_block, const u32* __restrict__ hist_scanned, const u32* __restrict__ totals,
    u32 nblk, u64* __restrict__ offsets_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  ScatterSmem& sm = *reinterpret_cast<ScatterSmem*>(smem_raw);
  const u32 D = 1u << bits, mask = D - 1;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

  // digit bases = exclusive scan of the digit totals (D <= 512 == RP_THREADS: one per thread)
  {
    u32 t = ((u32)tid < D) ? totals[tid] : 0, tot;
    u32 base = block_excl_scan_u32<RP_THREADS>(t, sm.scratch, &tot);
    if ((u32)tid < D) {
      sm.cursor[tid] = base + hist_scanned[(u64)tid * nblk + blockIdx.x];
      if (offsets_out && blockIdx.x == 0) {
        offsets_out[tid] = base;
        if ((u32)tid == D - 1) offsets_out[D] = n;
      }
    }
    for (int i = tid; i < RP_WAVES * RP_MAXD; i += RP_THREADS) (&sm.wcnt[0][0])[i] = 0;
  }
  __syncthreads();

  const u64 begin = (u64)blockIdx.x * rows_per_block;
  u64 end = begin + rows_per_block;
  if (end > n) end = n;
  volatile u16* wc = &sm.wcnt[w][0];

  for (u64 tile = begin; tile < end; tile += RP_TILE) {
    const u32 tile_n = (u32)((end - tile < RP_TILE) ? end - tile : RP_TILE);
    // wave w owns rows [w*512, w*512+512) of the tile, 8 rounds of 64 consecutive rows: lane
    // order inside a round + round order == row order, which is what makes the rank stable.
    Tup t[RP_ITEMS];
    u32 dg[RP_ITEMS];
    u32 rk[RP_ITEMS];
#pragma unroll
    for (int r = 0; r < RP_ITEMS; r++) {
      u32 li = (u32)w * (RP_ITEMS * 64) + r * 64 + lane;
      if (li < tile_n) t[r] = in[tile + li];
    }
#pragma unroll
    for (int r = 0; r < RP_ITEMS; r++) {
      u32 li = (u32)w * (RP_ITEMS * 64) + r * 64 + lane;
      bool valid = li < tile_n;
      u32 d = valid ? ((u32)(t[r].key >> shift) & mask) : 0;
      // lanes holding the same digit (wave match-any by bit ballots)
      u64 m = __ballot(valid);
      for (int b = 0; b < bits; b++) {
        bool bit = (d >> b) & 1;
        u64 bal = __ballot(bit);
        m &= bit ? bal : ~bal;
      }
      u32 below = popc_below(m);
      u32 old = 0;
      if (valid) {
        old = wc[d];
        if (below == 0) wc[d] = (u16)(old + (u32)__popcll(m));
      }
      dg[r] = d;
      rk[r] = old + below;
    }
    __syncthreads();

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
    __syncthreads();

#pragma unroll
    for (int r = 0; r < RP_ITEMS; r++) {
      u32 li = (u32)w * (RP_ITEMS * 64) + r * 64 + lane;
      if (li < tile_n) {
        u32 pos = sm.tile_off[dg[r]] + sm.wcnt[w][dg[r]] + rk[r];
        sm.stage[pos] = t[r];
      }
    }
    __syncthreads();

    // copy out: consecutive lanes -> consecutive rows of one digit run -> contiguous 16-B stores
#pragma unroll
    for (int r = 0; r < RP_ITEMS; r++) {
      u32 i = r * RP_THREADS + tid;
      if (i < tile_n) {
        Tup v = sm.stage[i];
        u32 d = (u32)(v.key >> shift) & mask;
        out[sm.delta[d] + i] = v;
      }
    }
    for (int i = tid; i < RP_WAVES * RP_MAXD / 2; i += RP_THREADS)
      reinterpret_cast<u32*>(&sm.wcnt[0][0])[i] = 0;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Partition boundaries of an array sorted by its top `bits` key bits: off[p] = first row whose
// top bits >= p (binary search; P+1 entries).  bits == 0 -> {0, n}.
// ---------------------------------------------------------------------------------------------
__global__ void part_offsets_kernel(const Tup* __restrict__ a, u32 n, int bits,
                                    u32* __restrict__ off, u32 P) {
  u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > P) return;
  if (p == P || bits == 0) {
    off[p] = (p == P) ? n : 0;
    return;
  }
  const u64* keys = reinterpret_cast<const u64*>(a);
  const int sh = 64 - bits;
  u32 lo = 0, hi = n;
  while (lo < hi) {
    u32 mid = lo + ((hi - lo) >> 1);
    if ((u32)(keys[2 * (u64)mid] >> sh) < p)
      lo = mid + 1;
    else
      hi = mid;
  }
  off[p] = lo;
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
void radix_pass_geometry(u32 n, u32* nblk, u32* rows_per_block) {
  u64 tiles = ((u64)n + RP_TILE - 1) / RP_TILE;
  if (tiles == 0) tiles = 1;
  u64 tpb = (tiles + RP_MAX_BLOCKS - 1) / RP_MAX_BLOCKS;
  *nblk = (u32)((tiles + tpb - 1) / tpb);
  *rows_per_block = (u32)(tpb * RP_TILE);
}

size_t radix_scatter_smem_bytes() { return sizeof(ScatterSmem); }

hipError_t launch_radix_hist(const void* in, u32 n, int shift, int bits, u32* hist, u32 nblk,
                             u32 rows_per_block, hipStream_t st) {
  hipLaunchKernelGGL(radix_hist_kernel, dim3(nblk), dim3(RP_THREADS), 0, st,
                     static_cast<const Tup*>(in), n, shift, (1u << bits) - 1, rows_per_block, hist,
                     nblk);
  return hipGetLastError();
}

hipError_t launch_radix_rowscan(u32* hist, u32 nblk, int bits, u32* totals, hipStream_t st) {
  hipLaunchKernelGGL(radix_rowscan_kernel, dim3(1u << bits), dim3(256), 0, st, hist, nblk, totals);
  return hipGetLastError();
}

hipError_t launch_radix_scatter(const void* in, void* out, u32 n, int shift, int bits,
                                const u32* hist_scanned, const u32* totals, u32 nblk,
                                u32 rows_per_block, u64* offsets_out, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(radix_scatter_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sizeof(ScatterSmem));
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblk), dim3(RP_THREADS), sizeof(ScatterSmem), st,
                     static_cast<const Tup*>(in), static_cast<Tup*>(out), n, shift, bits,
                     rows_per_block, hist_scanned, totals, nblk, offsets_out);
  return hipGetLastError();
}

hipError_t launch_part_offsets(const void* a, u32 n, int bits, u32* off, hipStream_t st) {
  u32 P = 1u << bits;
  hipLaunchKernelGGL(part_offsets_kernel, dim3((P + 1 + 255) / 256), dim3(256), 0, st,
                     static_cast<const Tup*>(a), n, bits, off, P);
  return hipGetLastError();
}

}  // namespace hmj
