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

}  // namespace hmj
