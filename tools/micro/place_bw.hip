// Micro-benchmark for the placement lottery: does the PHYSICAL memory behind a buffer change
// how fast it is written -- by a plain sequential fill, and by a scatter shaped like slab pass A (every workgroup
// appends 128-byte lines to 256 streams that lie ~22 MiB apart)?  Allocates CAND buffers of the slab size (all stay
// allocated, so they are different memory) and times both patterns on each, three times round robin.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(512) void fill_kernel(uint4* __restrict__ dst, size_t n16) {
  const uint4 v = make_uint4(1, 2, 3, 4);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(v.x, &dst[i].x), __builtin_nontemporal_store(v.y, &dst[i].y),
        __builtin_nontemporal_store(v.z, &dst[i].z), __builtin_nontemporal_store(v.w, &dst[i].w);
}
// worker w (a workgroup) owns slab [d][w][CA rows] of every digit d < 256; per step the 512 threads write 64 lines of
// 128 bytes: thread t -> line t / 8 of the step, 16 bytes each; the step's lines go to digits (step * 64 + line) & 255,
// each digit's slab filling up line by line -- the addresses of pass A with perfectly even digits
__global__ __launch_bounds__(512) void scatter_kernel(uint4* __restrict__ dst, unsigned WA, unsigned CA, unsigned lines_per_slab,
                                                      unsigned stream_shuffle) {
  const unsigned w = blockIdx.x, t = threadIdx.x, sub = t & 7, ln = t >> 3;
  const uint4 v = make_uint4(w, t, 3, 4);
  const unsigned steps = lines_per_slab * 256 / 64;
  for (unsigned s = 0; s < steps; s++) {
    unsigned d = (s * 64 + ln) & 255;
    if (stream_shuffle) d = (d * 167u + 13u) & 255u;
    const unsigned line = (s * 64 + ln) >> 8;  // how far this digit's slab has filled
    const size_t row = ((size_t)d * WA + w) * CA + (size_t)line * 8 + sub;
    __builtin_nontemporal_store(v.x, &dst[row].x), __builtin_nontemporal_store(v.y, &dst[row].y),
        __builtin_nontemporal_store(v.z, &dst[row].z), __builtin_nontemporal_store(v.w, &dst[row].w);
  }
}
__global__ __launch_bounds__(512) void read_kernel(const uint4* __restrict__ src, size_t n16, u64* out) {
  u64 acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = src[i];
    acc += v.x + v.w;
  }
  if (acc == 0x123456789ull) *out = acc;
}
int main(int argc, char** argv) {
  const int CAND = argc > 1 ? atoi(argv[1]) : 10;
  const unsigned WA = 2048, CA = 720, lines = 64;  // 512 rows of 16 B written per slab (as at 2^28 rows), slabs of 720 rows
  const size_t rows = (size_t)256 * WA * CA, bytes = rows * 16;
  void* buf[32];
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < CAND; i++) CK(hipMalloc(&buf[i], bytes));
  printf("%d buffers of %.2f GiB; scatter writes %.2f GiB per launch\n", CAND, bytes / 1073741824.0, 256.0 * WA * lines * 128 / 1073741824.0);
  for (int rep = 0; rep < 1; rep++)
    for (int i = 0; i < CAND; i++) {
      float ms_f, ms_s, ms_x;
      fill_kernel<<<4096, 512>>>((uint4*)buf[i], bytes / 16);
      CK(hipEventRecord(e0)); fill_kernel<<<4096, 512>>>((uint4*)buf[i], bytes / 16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms_f, e0, e1));
      scatter_kernel<<<WA, 512>>>((uint4*)buf[i], WA, CA, lines, 0);
      CK(hipEventRecord(e0)); scatter_kernel<<<WA, 512>>>((uint4*)buf[i], WA, CA, lines, 0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms_s, e0, e1));
      CK(hipEventRecord(e0)); scatter_kernel<<<WA, 512>>>((uint4*)buf[i], WA, CA, lines, 1); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms_x, e0, e1));
      const double sb = 256.0 * WA * lines * 128;
      printf("rep %d buf %2d %p : fill %.3f ms %.2f TB/s | scatter %.3f ms %.2f TB/s | shuffled digits %.3f ms %.2f TB/s\n", rep, i, buf[i],
             ms_f, bytes / ms_f * 1e-9, ms_s, sb / ms_s * 1e-9, ms_x, sb / ms_x * 1e-9);
    }
  // reads, and the fill rate of every 256 MiB piece of the first four buffers (20 fills each)
  u64* sink; CK(hipMalloc(&sink, 8));
  for (int i = 0; i < CAND; i++) {
    float ms;
    read_kernel<<<4096, 512>>>((const uint4*)buf[i], bytes / 16, sink);
    CK(hipEventRecord(e0)); read_kernel<<<4096, 512>>>((const uint4*)buf[i], bytes / 16, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("buf %2d read %.3f ms %.2f TB/s\n", i, ms, bytes / ms * 1e-9);
  }
  const size_t piece = 256ull << 20;
  for (int i = 0; i < (CAND < 6 ? CAND : 6); i++) {
    printf("buf %2d fill TB/s per 256 MiB piece:", i);
    for (size_t off = 0; off + piece <= bytes; off += piece) {
      float ms;
      fill_kernel<<<4096, 512>>>((uint4*)((char*)buf[i] + off), piece / 16);
      CK(hipEventRecord(e0));
      for (int r = 0; r < 20; r++) fill_kernel<<<4096, 512>>>((uint4*)((char*)buf[i] + off), piece / 16);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf(" %.1f", 20.0 * piece / ms * 1e-9);
    }
    printf("\n");
  }
  return 0;
}
