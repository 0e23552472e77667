// Issue cost of 64-bit against 32-bit integer compares on gfx950 (the run ranking and the LDS probe walks compare
// 64-bit payloads / keys).  Measured (MI355X, 4 waves per SIMD): v_cmp_lt_u64 + v_addc 10.9 cycles per pair and wave,
// v_cmp_lt_u32 + v_addc 10.0, v_addc alone 5.3 (at an assumed 2.4 GHz): the 64-bit compare issues at the 32-bit rate.
// hipcc -O3 --offload-arch=gfx950 tools/micro/cmp_rate.hip -o /tmp/cmp_rate && /tmp/cmp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u32;

template <int KIND>
__global__ __launch_bounds__(256) void k(const u64* __restrict__ in, u32* __restrict__ out, int iters) {
  u64 x[8];
#pragma unroll
  for (int i = 0; i < 8; i++) x[i] = in[threadIdx.x * 8 + i];
  u64 v = in[2048 + threadIdx.x];
  u32 cnt = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (KIND == 0) {
        asm volatile("v_cmp_lt_u64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(cnt) : "v"(x[i]), "v"(v) : "vcc");
      } else if (KIND == 1) {
        u32 xl = (u32)x[i], vl = (u32)v;
        asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(cnt) : "v"(xl), "v"(vl) : "vcc");
      } else if (KIND == 2) {  // 64-bit compare from 32-bit halves: hi <, hi ==, lo <
        u32 xl = (u32)x[i], vl = (u32)v, xh = (u32)(x[i] >> 32), vh = (u32)(v >> 32);
        asm volatile(
            "v_cmp_lt_u32 s[20:21], %3, %4\n\tv_cmp_eq_u32 s[22:23], %3, %4\n\tv_cmp_lt_u32 vcc, %1, %2\n\t"
            "s_and_b64 vcc, vcc, s[22:23]\n\ts_or_b64 vcc, vcc, s[20:21]\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc"
            : "+v"(cnt) : "v"(xl), "v"(vl), "v"(xh), "v"(vh) : "vcc", "s20", "s21", "s22", "s23");
      } else {  // the addc alone
        asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(cnt) : : "vcc");
      }
    }
    v += 0x9E3779B97F4A7C15ull;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = cnt;
}

int main() {
  u64* in; u32* out;
  hipMalloc(&in, 4096 * 8); hipMalloc(&out, 256 * 4 * 256 * 16);
  hipMemset(in, 0x5a, 4096 * 8);
  const int iters = 20000, grid = 256 * 4;  // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char* names[] = {"v_cmp_lt_u64 + addc", "v_cmp_lt_u32 + addc", "3 x cmp_u32 + 2 salu + addc", "addc alone"};
  for (int kind = 0; kind < 4; kind++) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(a);
      if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, in, out, iters);
      if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, in, out, iters);
      if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, in, out, iters);
      if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, in, out, iters);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      // per SIMD: 4 waves x iters x 8 groups; cycles at 2.4 GHz
      const double groups_per_simd = 4.0 * iters * 8;
      if (rep) printf("%-32s %.3f ms  -> %.2f cycles per group per wave (4 waves/SIMD, 2.4 GHz assumed)\n", names[kind], ms, ms * 1e-3 * 2.4e9 / groups_per_simd);
    }
  }
  return 0;
}
