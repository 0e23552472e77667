// Synthetic relation generators on device.  Integer-only arithmetic identical to
// oracle/hmj_oracle.c orc_gen_*, so CPU oracle and GPU run on the same rows without a PCIe copy.
// They mirror what the reference's benches do before timing: two relations over the same key set
// in different orders, so every probe key matches once (hashjoin_bench.cc:112-113, strgen.cc:51).
#include "hmj_dev.h"
#include "hmj_launch.h"

namespace hmj {

constexpr u64 kPiA = 0x9E3779B1ull, kPiB = 12345ull, kValXor = 0x9E3779B97F4A7C15ull;

__global__ void gen_build_kernel(Tup* __restrict__ out, u64 n, u64 start, u64 seed) {
  for (u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (u64)gridDim.x * blockDim.x) {
    u64 i = start + k;
    Tup t;
    t.key = mix64(i + seed);
    t.val = i;
    out[k] = t;
  }
}

__global__ void gen_probe_kernel(Tup* __restrict__ out, u64 n, u64 start, u64 n_build, u64 seed,
                                 u64 miss_mod) {
  for (u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (u64)gridDim.x * blockDim.x) {
    u64 j = start + k;
    u64 idx = n_build ? (kPiA * j + kPiB) % n_build : j;
    if (miss_mod && (j % miss_mod) == 0) idx += n_build;
    Tup t;
    t.key = mix64(idx + seed);
    t.val = j ^ kValXor;
    out[k] = t;
  }
}

__global__ void gen_from_cdf_kernel(Tup* __restrict__ out, u64 n, u64 start,
                                    const u64* __restrict__ thr, u64 domain, u64 seed, u64 zseed) {
  for (u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (u64)gridDim.x * blockDim.x) {
    u64 i = start + k;
    u64 u = mix64(i ^ zseed);
    u64 lo = 0, hi = domain;
    while (lo < hi) {
      u64 mid = lo + (hi - lo) / 2;
      if (thr[mid] < u)
        lo = mid + 1;
      else
        hi = mid;
    }
    if (lo >= domain) lo = domain - 1;
    Tup t;
    t.key = mix64(lo + seed);
    t.val = i;
    out[k] = t;
  }
}

__global__ void gen_uniform_domain_kernel(Tup* __restrict__ out, u64 n, u64 start, u64 domain,
                                          u64 seed, u64 zseed) {
  for (u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (u64)gridDim.x * blockDim.x) {
    u64 j = start + k;
    Tup t;
    t.key = mix64((mix64(j ^ zseed) % domain) + seed);
    t.val = j ^ kValXor;
    out[k] = t;
  }
}

static inline int gen_grid(u64 n) {
  u64 b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

hipError_t launch_gen_build(void* out, u64 n, u64 start, u64 seed, hipStream_t st) {
  hipLaunchKernelGGL(gen_build_kernel, dim3(gen_grid(n)), dim3(256), 0, st, static_cast<Tup*>(out),
                     n, start, seed);
  return hipGetLastError();
}
hipError_t launch_gen_probe(void* out, u64 n, u64 start, u64 n_build, u64 seed, u64 miss_mod,
                            hipStream_t st) {
  hipLaunchKernelGGL(gen_probe_kernel, dim3(gen_grid(n)), dim3(256), 0, st, static_cast<Tup*>(out),
                     n, start, n_build, seed, miss_mod);
  return hipGetLastError();
}
hipError_t launch_gen_from_cdf(void* out, u64 n, u64 start, const u64* thr, u64 domain, u64 seed,
                               u64 zseed, hipStream_t st) {
  hipLaunchKernelGGL(gen_from_cdf_kernel, dim3(gen_grid(n)), dim3(256), 0, st,
                     static_cast<Tup*>(out), n, start, thr, domain, seed, zseed);
  return hipGetLastError();
}
hipError_t launch_gen_uniform_domain(void* out, u64 n, u64 start, u64 domain, u64 seed, u64 zseed,
                                     hipStream_t st) {
  hipLaunchKernelGGL(gen_uniform_domain_kernel, dim3(gen_grid(n)), dim3(256), 0, st,
                     static_cast<Tup*>(out), n, start, domain, seed, zseed);
  return hipGetLastError();
}

// ---- ordered results of keys with structure (api.hip, "window + sort") -----------------------------
// rows {key[i], i}: the sortable handle of result row i
__global__ void key_idx_kernel(const u64* __restrict__ key, u64 n, Tup* __restrict__ out) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    Tup t;
    t.key = key[i];
    t.val = i;
    out[i] = t;
  }
}
// result columns in the order of the sorted handles
__global__ void gather3_kernel(const Tup* __restrict__ sorted, u64 n, const u64* __restrict__ rval,
                               const u64* __restrict__ sval, u64* __restrict__ okey, u64* __restrict__ orval,
                               u64* __restrict__ osval) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    const Tup t = sorted[i];
    okey[i] = t.key;
    orval[i] = rval[t.val];
    osval[i] = sval[t.val];
  }
}
// rows of `words` u64 each: pairs[i] = {rows[i][key_word], i}
__global__ void rows_key_idx_kernel(const u64* __restrict__ rows, u64 n, u32 words, u32 key_word, Tup* __restrict__ out) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    Tup t;
    t.key = rows[i * words + key_word];
    t.val = i;
    out[i] = t;
  }
}
// out row i = in row sorted[i].val   (one thread per u64 word: consecutive threads write consecutive words)
__global__ void rows_gather_kernel(const u64* __restrict__ in, const Tup* __restrict__ sorted, u64 n, u32 words,
                                   u64* __restrict__ out) {
  const u64 total = n * words;
  for (u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (u64)gridDim.x * blockDim.x) {
    const u64 i = x / words;
    const u32 w = (u32)(x - i * words);
    out[x] = in[sorted[i].val * words + w];
  }
}
__global__ void pairs_val_u32_kernel(const Tup* __restrict__ sorted, u64 n, u32* __restrict__ out) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) out[i] = (u32)sorted[i].val;
}
// Fill a buffer with 16-byte nontemporal stores (the allocator's write-bandwidth probe, api.hip ensure_dev).
__global__ __launch_bounds__(512) void fill_probe_kernel(uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    __builtin_nontemporal_store(0u, &dst[i].x);
    __builtin_nontemporal_store(0u, &dst[i].y);
    __builtin_nontemporal_store(0u, &dst[i].z);
    __builtin_nontemporal_store(0u, &dst[i].w);
  }
}
hipError_t launch_fill_probe(void* p, size_t bytes, hipStream_t st) {
  hipLaunchKernelGGL(fill_probe_kernel, dim3(4096), dim3(512), 0, st, static_cast<uint4*>(p), bytes / 16);
  return hipGetLastError();
}

hipError_t launch_rows_key_idx(const void* rows, u64 n, u32 words, u32 key_word, void* out, hipStream_t st) {
  hipLaunchKernelGGL(rows_key_idx_kernel, dim3(2048), dim3(256), 0, st, static_cast<const u64*>(rows), n, words, key_word,
                     static_cast<Tup*>(out));
  return hipGetLastError();
}
hipError_t launch_rows_gather(const void* in, const void* sorted, u64 n, u32 words, void* out, hipStream_t st) {
  hipLaunchKernelGGL(rows_gather_kernel, dim3(4096), dim3(256), 0, st, static_cast<const u64*>(in),
                     static_cast<const Tup*>(sorted), n, words, static_cast<u64*>(out));
  return hipGetLastError();
}
hipError_t launch_pairs_val_u32(const void* sorted, u64 n, u32* out, hipStream_t st) {
  hipLaunchKernelGGL(pairs_val_u32_kernel, dim3(2048), dim3(256), 0, st, static_cast<const Tup*>(sorted), n, out);
  return hipGetLastError();
}

hipError_t launch_key_idx(const u64* key, u64 n, void* out, hipStream_t st) {
  hipLaunchKernelGGL(key_idx_kernel, dim3(2048), dim3(256), 0, st, key, n, static_cast<Tup*>(out));
  return hipGetLastError();
}
hipError_t launch_gather3(const void* sorted, u64 n, const u64* rval, const u64* sval, u64* okey, u64* orval,
                          u64* osval, hipStream_t st) {
  hipLaunchKernelGGL(gather3_kernel, dim3(2048), dim3(256), 0, st, static_cast<const Tup*>(sorted), n, rval, sval,
                     okey, orval, osval);
  return hipGetLastError();
}

// handles {., idx} -> {col[idx], idx}, in place: the next key of a multi-key LSD sort
__global__ void rekey_kernel(Tup* __restrict__ pairs, u64 n, const u64* __restrict__ col) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    Tup t = pairs[i];
    t.key = col[t.val];
    pairs[i] = t;
  }
}
hipError_t launch_rekey(void* pairs, u64 n, const u64* col, hipStream_t st) {
  hipLaunchKernelGGL(rekey_kernel, dim3(2048), dim3(256), 0, st, static_cast<Tup*>(pairs), n, col);
  return hipGetLastError();
}

}  // namespace hmj
