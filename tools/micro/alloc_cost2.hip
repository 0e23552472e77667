// Follow-up: transparent huge pages for fresh result memory; H2D from pageable vs pinned.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipFree(0);
  { FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r"); char b[128] = {0}; if (f) { fgets(b, 127, f); fclose(f); } printf("THP enabled: %s", b); }
  for (size_t mb : {128, 1536}) {
    const size_t bytes = mb << 20;
    void* d; hipMalloc(&d, bytes); hipMemset(d, 1, bytes); hipDeviceSynchronize();
    void* p = aligned_alloc(2u << 20, bytes);
    int mr = madvise(p, bytes, MADV_HUGEPAGE);
    double t0 = now(); hipMemcpy(p, d, bytes, hipMemcpyDeviceToHost); double t_thp = now() - t0;
    t0 = now(); hipMemcpy(p, d, bytes, hipMemcpyDeviceToHost); double t_thp2 = now() - t0;
    t0 = now(); free(p); double t_free = now() - t0;
    void* q = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
    double t_pop = 0;
    { t0 = now(); void* q2 = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0); t_pop = now() - t0; munmap(q2, bytes); }
    // H2D: pageable (touched) source, single hipMemcpy
    memset(q, 3, bytes);
    t0 = now(); hipMemcpy(d, q, bytes, hipMemcpyHostToDevice); double t_h2d_page = now() - t0;
    t0 = now(); hipMemcpy(d, q, bytes, hipMemcpyHostToDevice); double t_h2d_page2 = now() - t0;
    void* h; hipHostMalloc(&h, bytes, hipHostMallocDefault); memset(h, 3, bytes);
    t0 = now(); hipMemcpy(d, h, bytes, hipMemcpyHostToDevice); double t_h2d_pin = now() - t0;
    printf("%5zu MiB: D2H->fresh THP(madvise rc %d) %.2f ms, again %.2f ms, free %.2f | mmap POPULATE %.2f ms | H2D pageable %.2f / %.2f ms, pinned %.2f ms\n",
           mb, mr, t_thp, t_thp2, t_free, t_pop, t_h2d_page, t_h2d_page2, t_h2d_pin);
    hipHostFree(h); munmap(q, bytes); hipFree(d);
  }
  return 0;
}
