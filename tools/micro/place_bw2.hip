// Follow-up to place_bw.hip: is a slowly written buffer slow from every XCD, for plain stores too, and for smaller grids?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <bool NT>
__global__ __launch_bounds__(512) void fill_kernel(uint4* __restrict__ dst, size_t n16, int xcd_mod, int xcd_sel) {
  if (xcd_mod && (int)(blockIdx.x % xcd_mod) != xcd_sel) return;
  const size_t active = xcd_mod ? gridDim.x / xcd_mod : gridDim.x, me = xcd_mod ? blockIdx.x / xcd_mod : blockIdx.x;
  const uint4 v = make_uint4(1, 2, 3, 4);
  for (size_t i = me * blockDim.x + threadIdx.x; i < n16; i += active * blockDim.x) {
    if (NT) { __builtin_nontemporal_store(v.x, &dst[i].x); __builtin_nontemporal_store(v.y, &dst[i].y); __builtin_nontemporal_store(v.z, &dst[i].z); __builtin_nontemporal_store(v.w, &dst[i].w); }
    else dst[i] = v;
  }
}
int main(int argc, char** argv) {
  const int CAND = argc > 1 ? atoi(argv[1]) : 8;
  const size_t bytes = (size_t)6 << 30;
  void* buf[32];
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < CAND; i++) CK(hipMalloc(&buf[i], bytes));
  auto timeit = [&](auto&& launch) { float ms; launch(); CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); return bytes / ms * 1e-9; };
  for (int i = 0; i < CAND; i++) {
    uint4* p = (uint4*)buf[i];
    printf("buf %d: nt %.2f plain %.2f | grid 1024: %.2f  grid 512: %.2f | memset %.2f | per XCD (1/8 of the grid):", i,
           timeit([&] { fill_kernel<true><<<4096, 512>>>(p, bytes / 16, 0, 0); }), timeit([&] { fill_kernel<false><<<4096, 512>>>(p, bytes / 16, 0, 0); }),
           timeit([&] { fill_kernel<true><<<1024, 512>>>(p, bytes / 16, 0, 0); }), timeit([&] { fill_kernel<true><<<512, 512>>>(p, bytes / 16, 0, 0); }),
           timeit([&] { CK(hipMemsetAsync(p, 0, bytes, 0)); }));
    for (int x = 0; x < 8; x++) printf(" %.2f", timeit([&] { fill_kernel<true><<<4096, 512>>>(p, bytes / 16, 8, x); }));
    printf("\n");
  }
  return 0;
}
