// Follow-up to place_bw.hip: does a slowly written buffer suffer less when every visit of a write stream leaves 256 or 512
// bytes instead of one 128-byte line (same total bytes, same 256 streams per workgroup)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ __launch_bounds__(512) void fill_kernel(uint4* __restrict__ dst, size_t n16) {
  const uint4 v = make_uint4(1, 2, 3, 4);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
// LPV lines of 128 bytes per visit of a stream; 512 threads write 64 lines per step = 64 / LPV streams per step
template <int LPV>
__global__ __launch_bounds__(512) void scatter_kernel(uint4* __restrict__ dst, unsigned WA, unsigned CA, unsigned lines_per_slab) {
  const unsigned w = blockIdx.x, t = threadIdx.x, sub = t & 7, ln = t >> 3;  // ln: 0..63
  const uint4 v = make_uint4(w, t, 3, 4);
  const unsigned steps = lines_per_slab * 256 / 64;
  const unsigned spv = 64 / LPV;  // streams per step
  for (unsigned s = 0; s < steps; s++) {
    const unsigned visit = s * spv + ln / LPV;            // global visit number
    const unsigned d = visit & 255, round = visit >> 8;   // stream, how often it was visited before
    const size_t row = ((size_t)d * WA + w) * CA + ((size_t)round * LPV + ln % LPV) * 8 + sub;
    __builtin_nontemporal_store(v.x, &dst[row].x), __builtin_nontemporal_store(v.y, &dst[row].y),
        __builtin_nontemporal_store(v.z, &dst[row].z), __builtin_nontemporal_store(v.w, &dst[row].w);
  }
}
int main(int argc, char** argv) {
  const int CAND = argc > 1 ? atoi(argv[1]) : 8;
  const unsigned WA = 2048, CA = 720, lines = 64;
  const size_t rows = (size_t)256 * WA * CA, bytes = rows * 16;
  void* buf[32];
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < CAND; i++) CK(hipMalloc(&buf[i], bytes));
  const double sb = 256.0 * WA * lines * 128;
  auto timeit = [&](auto&& launch) { float ms; launch(); CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); return (double)ms; };
  for (int i = 0; i < CAND; i++) {
    uint4* p = (uint4*)buf[i];
    printf("buf %d: fill %.2f TB/s | scatter, bytes per stream visit: 128: %.2f  256: %.2f  512: %.2f  1024: %.2f TB/s\n", i,
           bytes / timeit([&] { fill_kernel<<<4096, 512>>>(p, bytes / 16); }) * 1e-9,
           sb / timeit([&] { scatter_kernel<1><<<WA, 512>>>(p, WA, CA, lines); }) * 1e-9, sb / timeit([&] { scatter_kernel<2><<<WA, 512>>>(p, WA, CA, lines); }) * 1e-9,
           sb / timeit([&] { scatter_kernel<4><<<WA, 512>>>(p, WA, CA, lines); }) * 1e-9, sb / timeit([&] { scatter_kernel<8><<<WA, 512>>>(p, WA, CA, lines); }) * 1e-9);
  }
  return 0;
}
