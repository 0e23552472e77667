// What does device memory cost to CREATE on this box?  (the driver's box charged 166 / 159 / 165 ms for the three
// 6 GB partition buffers of the headline join -- 26 ms per GB -- where the builder's boxes charged 0.3 ms.)  Fresh process, no
// other GPU work: hipMalloc of 2 / 6 / 18 GB, three 6 GB buffers against one 18 GB arena, the first fill (first touch) and a
// second fill of each, hipFree.  Usage: malloc_cost [reps]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

__global__ void fill_kernel(uint4* p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, 4);
}
static double fill_ms(void* p, size_t bytes) {
  const double t0 = now_ms();
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (uint4*)p, bytes / 16);
  (void)hipDeviceSynchronize();
  return now_ms() - t0;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 2;
  double t0 = now_ms();
  CK(hipFree(0));
  std::printf("runtime init %.1f ms\n", now_ms() - t0);
  const size_t GB = 1000ull * 1000 * 1000;
  for (int rep = 0; rep < reps; rep++) {
    for (size_t gb : {2, 6, 18}) {
      void* p = nullptr;
      t0 = now_ms();
      CK(hipMalloc(&p, gb * GB));
      const double a = now_ms() - t0;
      const double f1 = fill_ms(p, gb * GB), f2 = fill_ms(p, gb * GB);
      t0 = now_ms();
      CK(hipFree(p));
      std::printf("rep %d: one buffer of %2zu GB: hipMalloc %8.2f ms (%6.2f ms/GB)  first fill %7.2f ms  second fill %6.2f ms  hipFree %7.2f ms\n", rep, gb, a,
                  a / gb, f1, f2, now_ms() - t0);
    }
    void* q[3] = {nullptr, nullptr, nullptr};
    double a3[3];
    for (int i = 0; i < 3; i++) {
      t0 = now_ms();
      CK(hipMalloc(&q[i], 6 * GB));
      a3[i] = now_ms() - t0;
    }
    std::printf("rep %d: three buffers of 6 GB: hipMalloc %.2f + %.2f + %.2f = %.2f ms\n", rep, a3[0], a3[1], a3[2], a3[0] + a3[1] + a3[2]);
    for (int i = 0; i < 3; i++) CK(hipFree(q[i]));
  }
  return 0;
}
