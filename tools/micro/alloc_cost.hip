// What does a first call pay for?  hipMalloc / hipHostMalloc / fresh pageable memory, per size.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipFree(0);
  for (size_t mb : {32, 128, 512, 1536}) {
    const size_t bytes = mb << 20;
    void* d; double t0 = now(); hipMalloc(&d, bytes); double t_dev = now() - t0;
    void* h; t0 = now(); hipHostMalloc(&h, bytes, hipHostMallocDefault); double t_pin = now() - t0;
    t0 = now(); hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost); double t_d2h_pin = now() - t0;
    t0 = now(); hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost); double t_d2h_pin2 = now() - t0;
    char* p = (char*)malloc(bytes);
    t0 = now(); hipMemcpy(p, d, bytes, hipMemcpyDeviceToHost); double t_d2h_page = now() - t0;
    t0 = now(); hipMemcpy(p, d, bytes, hipMemcpyDeviceToHost); double t_d2h_page2 = now() - t0;
    char* q = (char*)malloc(bytes);
    t0 = now();
    { std::vector<std::thread> th; const int T = 8; for (int t = 0; t < T; t++) th.emplace_back([=] { memset(q + bytes / T * t, 1, bytes / T); }); for (auto& x : th) x.join(); }
    double t_touch8 = now() - t0;
    t0 = now(); hipHostRegister(q, bytes, hipHostRegisterDefault); double t_reg = now() - t0;
    t0 = now(); hipMemcpy(q, d, bytes, hipMemcpyDeviceToHost); double t_d2h_reg = now() - t0;
    hipHostUnregister(q);
    t0 = now(); hipHostFree(h); double t_pinfree = now() - t0;
    printf("%5zu MiB: hipMalloc %.2f ms | hipHostMalloc %.2f ms (free %.2f) | D2H->pinned %.2f / %.2f ms | D2H->fresh pageable %.2f, again %.2f ms | touch(8 thr) %.2f ms | hostRegister(touched) %.2f ms, D2H->registered %.2f ms\n",
           mb, t_dev, t_pin, t_pinfree, t_d2h_pin, t_d2h_pin2, t_d2h_page, t_d2h_page2, t_touch8, t_reg, t_d2h_reg);
    hipFree(d); free(p); free(q);
  }
  return 0;
}
