// Round-4 placement experiment: round 3 found that how fast a buffer can be written is a property of
// the PHYSICAL memory the driver hands out at that moment (ranges of tens of GB fill at 4.4 or at 5.6 TB/s).  If so, a buffer
// put together from individually measured 1 GiB physical chunks (hipMemCreate) -- the fast ones kept, the slow ones given
// back -- should write fast every time.  This program creates N chunks, measures the fill rate of each (mapped one by one
// into a reserved range), then maps the best K and the worst K into two 'buffers' and measures fill and the pass-A-shaped
// scatter of place_bw.hip on both.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d: %s\n", hipGetErrorString(e), __LINE__, #x); exit(1); } } while (0)

__global__ __launch_bounds__(512) void fill_kernel(uint4* __restrict__ dst, size_t n16) {
  const uint4 v = make_uint4(1, 2, 3, 4);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(v.x, &dst[i].x), __builtin_nontemporal_store(v.y, &dst[i].y),
        __builtin_nontemporal_store(v.z, &dst[i].z), __builtin_nontemporal_store(v.w, &dst[i].w);
}
__global__ __launch_bounds__(512) void scatter_kernel(uint4* __restrict__ dst, unsigned WA, unsigned CA, unsigned lines_per_slab) {
  const unsigned w = blockIdx.x, t = threadIdx.x, sub = t & 7, ln = t >> 3;
  const uint4 v = make_uint4(w, t, 3, 4);
  const unsigned steps = lines_per_slab * 256 / 64;
  for (unsigned s = 0; s < steps; s++) {
    const unsigned d = (s * 64 + ln) & 255;
    const unsigned line = (s * 64 + ln) >> 8;
    const size_t row = ((size_t)d * WA + w) * CA + (size_t)line * 8 + sub;
    __builtin_nontemporal_store(v.x, &dst[row].x), __builtin_nontemporal_store(v.y, &dst[row].y),
        __builtin_nontemporal_store(v.z, &dst[row].z), __builtin_nontemporal_store(v.w, &dst[row].w);
  }
}

static hipEvent_t e0, e1;
static double fill_rate(void* p, size_t bytes) {  // TB/s of the faster of two timed fills (after one untimed)
  float best = 1e30f, ms;
  fill_kernel<<<4096, 512>>>((uint4*)p, bytes / 16);
  for (int r = 0; r < 2; r++) {
    CK(hipEventRecord(e0));
    fill_kernel<<<4096, 512>>>((uint4*)p, bytes / 16);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  return bytes / best * 1e-9;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 40, K = argc > 2 ? atoi(argv[2]) : 6;
  const size_t chunk = (argc > 3 ? (size_t)atoi(argv[3]) : 1024) << 20;
  CK(hipSetDevice(0));
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<hipMemGenericAllocationHandle_t> h(N);
  std::vector<double> rate(N);
  void* va = nullptr;
  CK(hipMemAddressReserve(&va, chunk, 0, nullptr, 0));
  float ms_create = 0;
  for (int i = 0; i < N; i++) {
    CK(hipEventRecord(e0));
    CK(hipMemCreate(&h[i], chunk, &prop, 0));
    CK(hipMemMap(va, chunk, 0, h[i], 0));
    CK(hipMemSetAccess(va, chunk, &acc, 1));
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms_create += ms;
    rate[i] = fill_rate(va, chunk);
    CK(hipDeviceSynchronize());
    CK(hipMemUnmap(va, chunk));
  }
  printf("%d chunks of %zu MiB: create + map %.2f ms each; fill rates in creation order (TB/s):\n", N, chunk >> 20, ms_create / N);
  for (int i = 0; i < N; i++) printf("%.2f%s", rate[i], (i % 10 == 9 || i == N - 1) ? "\n" : " ");
  std::vector<int> idx(N);
  for (int i = 0; i < N; i++) idx[i] = i;
  std::sort(idx.begin(), idx.end(), [&](int a, int b) { return rate[a] > rate[b]; });
  const unsigned WA = 2048, CA = 720, lines = 64;
  const size_t bytes = (size_t)256 * WA * CA * 16;  // 5.6 GiB
  const size_t need = (bytes + chunk - 1) / chunk;
  if ((size_t)K < need || 2 * K > N) { printf("K too small / N too small for the composed buffers\n"); return 0; }
  const double sb = 256.0 * WA * lines * 128;
  for (int which = 0; which < 2; which++) {
    void* p = nullptr;
    CK(hipMemAddressReserve(&p, (size_t)K * chunk, 0, nullptr, 0));
    for (int j = 0; j < K; j++) CK(hipMemMap((char*)p + (size_t)j * chunk, chunk, 0, h[which == 0 ? idx[j] : idx[N - 1 - j]], 0));
    CK(hipMemSetAccess(p, (size_t)K * chunk, &acc, 1));
    const double f = fill_rate(p, bytes);
    float s1;
    scatter_kernel<<<WA, 512>>>((uint4*)p, WA, CA, lines);
    CK(hipEventRecord(e0)); scatter_kernel<<<WA, 512>>>((uint4*)p, WA, CA, lines); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&s1, e0, e1));
    printf("buffer of the %s %d chunks (mean chunk rate %.2f): fill %.2f TB/s | scatter %.2f TB/s\n", which == 0 ? "BEST" : "WORST", K,
           [&] { double m = 0; for (int j = 0; j < K; j++) m += rate[which == 0 ? idx[j] : idx[N - 1 - j]]; return m / K; }(), f, sb / s1 * 1e-9);
    CK(hipDeviceSynchronize());
    CK(hipMemUnmap(p, (size_t)K * chunk));
    CK(hipMemAddressFree(p, (size_t)K * chunk));
  }
  // the same with hipMalloc, for this process's luck
  for (int i = 0; i < 3; i++) {
    void* p;
    CK(hipMalloc(&p, bytes));
    printf("hipMalloc %d: fill %.2f TB/s\n", i, fill_rate(p, bytes));
  }
  return 0;
}
