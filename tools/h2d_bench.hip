// h2d_bench.hip -- what the host link gives to a pinned H2D stream (tools; not part of the library).
// Usage: h2d_bench [MiB per copy]   -> GB/s for 1, 2 and 4 concurrent copy streams, and with D2H traffic beside it.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
// occupies every CU with persistent workgroups for about `us` microseconds (the classify kernel's launch shape)
__global__ __launch_bounds__(512) void busy_kernel(unsigned long long ticks, unsigned *out)
{
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
  unsigned x = threadIdx.x;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) x = x * 1664525u + 1013904223u;
  if (x == 12345u) out[0] = x;
}

int main(int argc, char **argv)
{
  const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 600;
  const size_t bytes = mib << 20;
  const int NS = 4;
  char *h[NS], *d[NS], *hd, *dd;
  hipStream_t s[NS], sd;
  for (int i = 0; i < NS; ++i) {
    CK(hipHostMalloc((void **)&h[i], bytes, hipHostMallocDefault));
    CK(hipMalloc((void **)&d[i], bytes));
    CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    for (size_t j = 0; j < bytes; j += 4096) h[i][j] = (char)j;
  }
  CK(hipHostMalloc((void **)&hd, 64 << 20, hipHostMallocDefault));
  CK(hipMalloc((void **)&dd, 64 << 20));
  CK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
  for (int with_d2h = 0; with_d2h < 2; ++with_d2h)
    for (int ns : {1, 2, 4}) {
      const int reps = 6;
      // the same total bytes per repetition, split over ns streams
      const size_t per = bytes / ns;
      for (int i = 0; i < ns; ++i) CK(hipMemcpyAsync(d[i], h[i], per, hipMemcpyHostToDevice, s[i]));
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < reps; ++r) {
        for (int i = 0; i < ns; ++i) CK(hipMemcpyAsync(d[i], h[i], per, hipMemcpyHostToDevice, s[i]));
        if (with_d2h) CK(hipMemcpyAsync(hd, dd, 64 << 20, hipMemcpyDeviceToHost, sd));
      }
      CK(hipDeviceSynchronize());
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      printf("{\"h2d_streams\": %d, \"d2h_beside\": %d, \"MiB_per_rep\": %zu, \"GBps\": %.2f}\n", ns, with_d2h, mib, reps * (double)(per * ns) / dt / 1e9);
    }
  // does a copy overlap a kernel that fills every CU (4 x 512-thread workgroups per CU, as the classify kernel does)?
  unsigned *dout;
  CK(hipMalloc((void **)&dout, 64));
  hipStream_t sk;
  CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
  for (int wg_per_cu : {4, 3}) {
    hipEvent_t c0, c1;
    CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(busy_kernel, dim3(256 * wg_per_cu), dim3(512), 0, sk, 3000000ull /* 30 ms */, dout);
    CK(hipEventRecord(c0, s[0]));
    CK(hipMemcpyAsync(d[0], h[0], bytes, hipMemcpyHostToDevice, s[0]));
    CK(hipEventRecord(c1, s[0]));
    CK(hipEventSynchronize(c1));
    const double t_copy = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    CK(hipDeviceSynchronize());
    const double t_all = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    float ms = 0;
    CK(hipEventElapsedTime(&ms, c0, c1));
    printf("{\"busy_kernel_wg_per_cu\": %d, \"busy_ms\": 30, \"copy_MiB\": %zu, \"copy_done_after_ms\": %.2f, \"copy_event_ms\": %.2f, \"all_done_after_ms\": %.2f}\n",
           wg_per_cu, mib, t_copy * 1e3, ms, t_all * 1e3);
  }
  return 0;
}
