// h2d_bench.hip -- what the host link gives to a pinned H2D stream (tools; not part of the library).
// Usage: h2d_bench [MiB per copy]   -> GB/s for 1, 2 and 4 concurrent copy streams, and with D2H traffic beside it.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
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
  return 0;
}
