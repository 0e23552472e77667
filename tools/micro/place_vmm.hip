// Round-3 placement experiment: does a buffer backed by ONE physically
// created allocation (hipMemCreate + hipMemMap, the virtual-memory-management API) write as fast every time, where
// hipMalloc hands out buffers that fill at 4.4 or at 5.6 TB/s?  Three kinds of 5.6 GiB buffers, all held at once:
//   M  hipMalloc
//   V  hipMemCreate of the whole size, one handle, mapped into a reserved range
//   C  hipMemCreate in 1 GiB chunks (one handle each), mapped back to back into one reserved range
// Every buffer is filled three times (nontemporal 16-byte stores, 4096 x 512 threads); the last two fills are timed.
// Then the pass-A-shaped scatter of place_bw.hip on each.  Prints one line per buffer.
#include <hip/hip_runtime.h>
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

struct Buf {
  char kind;
  void* p;
  std::vector<hipMemGenericAllocationHandle_t> handles;
};

int main(int argc, char** argv) {
  const int per_kind = argc > 1 ? atoi(argv[1]) : 6;
  const unsigned WA = 2048, CA = 720, lines = 64;
  const size_t rows = (size_t)256 * WA * CA;
  size_t bytes = rows * 16;
  int dev = 0;
  CK(hipSetDevice(dev));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gmin = 0, grec = 0;
  CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
  printf("allocation granularity: minimum %zu, recommended %zu bytes\n", gmin, grec);
  const size_t chunk = 1ull << 30;
  const size_t gran = grec ? grec : (2u << 20);
  const size_t vbytes = (bytes + chunk - 1) / chunk * chunk;  // multiple of the chunk (and of the granularity)
  (void)gran;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<Buf> bufs;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // interleave the kinds so that no kind gets "the early memory"
  for (int i = 0; i < per_kind; i++) {
    for (char kind : {'M', 'V', 'C'}) {
      Buf b;
      b.kind = kind;
      hipEventRecord(e0);
      if (kind == 'M') {
        CK(hipMalloc(&b.p, bytes));
      } else {
        CK(hipMemAddressReserve(&b.p, vbytes, 0, nullptr, 0));
        const size_t step = kind == 'V' ? vbytes : chunk;
        for (size_t off = 0; off < vbytes; off += step) {
          hipMemGenericAllocationHandle_t h;
          CK(hipMemCreate(&h, step, &prop, 0));
          CK(hipMemMap((char*)b.p + off, step, 0, h, 0));
          b.handles.push_back(h);
        }
        CK(hipMemSetAccess(b.p, vbytes, &acc, 1));
      }
      bufs.push_back(b);
    }
  }
  printf("%zu buffers of %.2f GiB held\n", bufs.size(), bytes / 1073741824.0);
  const double sb = 256.0 * WA * lines * 128;
  for (int rep = 0; rep < 2; rep++)
    for (size_t i = 0; i < bufs.size(); i++) {
      float f1, f2, s1;
      uint4* d = (uint4*)bufs[i].p;
      fill_kernel<<<4096, 512>>>(d, bytes / 16);
      CK(hipEventRecord(e0)); fill_kernel<<<4096, 512>>>(d, bytes / 16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&f1, e0, e1));
      CK(hipEventRecord(e0)); fill_kernel<<<4096, 512>>>(d, bytes / 16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&f2, e0, e1));
      scatter_kernel<<<WA, 512>>>(d, WA, CA, lines);
      CK(hipEventRecord(e0)); scatter_kernel<<<WA, 512>>>(d, WA, CA, lines); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&s1, e0, e1));
      printf("rep %d buf %2zu %c %p : fill %.2f %.2f TB/s | scatter %.2f TB/s\n", rep, i, bufs[i].kind, bufs[i].p, bytes / f1 * 1e-9,
             bytes / f2 * 1e-9, sb / s1 * 1e-9);
      fflush(stdout);
    }
  return 0;
}
