// Micro-benchmark: what copy / read / write rate does this MI355X reach, and with which launch shape and
// cache policy?  (Sets the practical ceiling for the radix scatter, which is a copy with a permutation.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
struct alignas(16) Row { u64 k, v; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int U, int NT>  // NT: 0 plain, 1 nontemporal stores, 2 nontemporal loads+stores
__global__ void copy_kernel(const Row* __restrict__ src, Row* __restrict__ dst, size_t n) {
  const size_t tile = (size_t)blockDim.x * U;
  for (size_t base = (size_t)blockIdx.x * tile; base < n; base += (size_t)gridDim.x * tile) {
    Row r[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
      if (i < n) {
        if (NT == 2) { r[u].k = __builtin_nontemporal_load(&src[i].k); r[u].v = __builtin_nontemporal_load(&src[i].v); }
        else r[u] = src[i];
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
      if (i < n) {
        if (NT >= 1) { __builtin_nontemporal_store(r[u].k, &dst[i].k); __builtin_nontemporal_store(r[u].v, &dst[i].v); }
        else dst[i] = r[u];
      }
    }
  }
}
template <int U>
__global__ void read_kernel(const Row* __restrict__ src, u64* out, size_t n) {
  const size_t tile = (size_t)blockDim.x * U;
  u64 acc = 0;
  for (size_t base = (size_t)blockIdx.x * tile; base < n; base += (size_t)gridDim.x * tile) {
    Row r[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
      r[u] = i < n ? src[i] : Row{0, 0};
    }
#pragma unroll
    for (int u = 0; u < U; u++) acc += r[u].k ^ r[u].v;
  }
  if (acc == 0x1234567) out[0] = acc;
}
template <int U>
__global__ void write_kernel(Row* __restrict__ dst, size_t n) {
  const size_t tile = (size_t)blockDim.x * U;
  for (size_t base = (size_t)blockIdx.x * tile; base < n; base += (size_t)gridDim.x * tile) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      size_t i = base + (size_t)u * blockDim.x + threadIdx.x;
      if (i < n) dst[i] = Row{i, base};
    }
  }
}
template <typename F>
static float time_ms(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int i = 0; i < reps; i++) {
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  return best;
}
int main() {
  const size_t n = 1ull << 28;  // 4 GiB of rows
  Row *src, *dst; u64* out;
  CK(hipMalloc(&src, n * 16)); CK(hipMalloc(&dst, n * 16)); CK(hipMalloc(&out, 8));
  CK(hipMemset(src, 1, n * 16)); CK(hipMemset(dst, 2, n * 16));
  const double gb = n * 16 / 1e9;
  float ms = time_ms([&] { CK(hipMemcpyAsync(dst, src, n * 16, hipMemcpyDeviceToDevice, 0)); });
  printf("hipMemcpy D2D                          %.3f ms  %.2f TB/s (rd+wr)\n", ms, 2 * gb / ms);
#define RUN_COPY(U, NT, THREADS, GRIDMUL) { int grid = 256 * GRIDMUL; \
    ms = time_ms([&] { hipLaunchKernelGGL((copy_kernel<U, NT>), dim3(grid), dim3(THREADS), 0, 0, src, dst, n); }); \
    printf("copy U=%d nt=%d threads=%4d grid=256x%-3d  %.3f ms  %.2f TB/s (rd+wr)\n", U, NT, THREADS, GRIDMUL, ms, 2 * gb / ms); }
  RUN_COPY(1, 0, 256, 8) RUN_COPY(1, 0, 256, 32) RUN_COPY(4, 0, 256, 8) RUN_COPY(4, 0, 256, 16) RUN_COPY(4, 0, 512, 4) RUN_COPY(4, 0, 512, 8)
  RUN_COPY(8, 0, 256, 8) RUN_COPY(8, 0, 512, 4) RUN_COPY(4, 0, 1024, 2) RUN_COPY(4, 0, 1024, 4) RUN_COPY(2, 0, 1024, 4)
  RUN_COPY(4, 1, 256, 8) RUN_COPY(4, 1, 512, 4) RUN_COPY(4, 1, 512, 8) RUN_COPY(8, 1, 256, 8) RUN_COPY(4, 2, 512, 4) RUN_COPY(4, 2, 256, 8) RUN_COPY(8, 2, 256, 8)
  {
    const size_t full = (n + 1023) / 1024;  // one tile per block, no grid-stride loop
    ms = time_ms([&] { hipLaunchKernelGGL((copy_kernel<4, 0>), dim3((unsigned)full), dim3(256), 0, 0, src, dst, n); });
    printf("copy U=4 nt=0 threads= 256 grid=n/1024     %.3f ms  %.2f TB/s (rd+wr)\n", ms, 2 * gb / ms);
    ms = time_ms([&] { hipLaunchKernelGGL((copy_kernel<4, 1>), dim3((unsigned)full), dim3(256), 0, 0, src, dst, n); });
    printf("copy U=4 nt=1 threads= 256 grid=n/1024     %.3f ms  %.2f TB/s (rd+wr)\n", ms, 2 * gb / ms);
  }
#define RUN_READ(U, THREADS, GRIDMUL) { int grid = 256 * GRIDMUL; \
    ms = time_ms([&] { hipLaunchKernelGGL((read_kernel<U>), dim3(grid), dim3(THREADS), 0, 0, src, out, n); }); \
    printf("read U=%d threads=%4d grid=256x%-3d        %.3f ms  %.2f TB/s\n", U, THREADS, GRIDMUL, ms, gb / ms); }
  RUN_READ(4, 256, 8) RUN_READ(8, 256, 8) RUN_READ(4, 512, 8) RUN_READ(8, 512, 4) RUN_READ(5, 1024, 4) RUN_READ(8, 1024, 2)
#define RUN_WRITE(U, THREADS, GRIDMUL) { int grid = 256 * GRIDMUL; \
    ms = time_ms([&] { hipLaunchKernelGGL((write_kernel<U>), dim3(grid), dim3(THREADS), 0, 0, dst, n); }); \
    printf("write U=%d threads=%4d grid=256x%-3d       %.3f ms  %.2f TB/s\n", U, THREADS, GRIDMUL, ms, gb / ms); }
  RUN_WRITE(4, 256, 8) RUN_WRITE(4, 512, 8)
  return 0;
}
