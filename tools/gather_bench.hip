// gather_bench.hip -- ceiling of independent random 16-byte lookups in a table far larger than the caches
// (tools; not part of the library).  This is the access pattern of the position table on a GENCODE-scale index
// (DESIGN.md 7.4): one 16-byte bucket per k-mer at a hashed address.  Reports requests/s for table sizes,
// loads in flight per lane, waves per SIMD and cache policies, to tell whether ~50 G requests/s is the part's
// limit for this pattern or an artefact of how the classify kernel issues its probes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

// POLICY 0 plain, 1 nontemporal, 2 sc1 (bypass L1);  W = bytes per lookup (4, 8, 16)
template <int U, int POLICY, int W>
__global__ __launch_bounds__(256) void gather_kernel(const uint8_t *__restrict__ tab, uint64_t mask16, uint32_t iters, uint32_t *__restrict__ out)
{
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  uint64_t x = mix(tid + 1);
  for (uint32_t it = 0; it < iters; ++it) {
    uint64_t a[U];
#pragma unroll
    for (int j = 0; j < U; ++j) { x = mix(x + j + 1); a[j] = (x & mask16) << 4; }
    if (W == 16) {
      u32x4 v[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const u32x4 *p = reinterpret_cast<const u32x4 *>(tab + a[j]);
        if (POLICY == 1) v[j] = __builtin_nontemporal_load(p);
        // (load and wait in ONE asm statement: the compiler cannot track a load that is still in flight when the asm ends)
        else if (POLICY == 2) { asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v[j]) : "v"(p) : "memory"); }
        else v[j] = *p;
      }
#pragma unroll
      for (int j = 0; j < U; ++j) acc += v[j].x ^ v[j].w;
    } else if (W == 8) {
      u32x2 v[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const u32x2 *p = reinterpret_cast<const u32x2 *>(tab + a[j]);
        v[j] = POLICY == 1 ? __builtin_nontemporal_load(p) : *p;
      }
#pragma unroll
      for (int j = 0; j < U; ++j) acc += v[j].x ^ v[j].y;
    } else {
      uint32_t v[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(tab + a[j]);
        v[j] = POLICY == 1 ? __builtin_nontemporal_load(p) : *p;
      }
#pragma unroll
      for (int j = 0; j < U; ++j) acc += v[j];
    }
  }
  if (acc == 0x12345678u) out[tid & 1023] = acc;   // keep the loads alive
}

template <int U, int POLICY, int W>
static int run(const uint8_t *tab, uint64_t bytes, int wg_per_cu, uint32_t *out, const char *pname)
{
  const uint32_t iters = 2000 / U;
  const unsigned grid = 256u * wg_per_cu;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((gather_kernel<U, POLICY, W>), dim3(grid), dim3(256), 0, 0, tab, bytes / 16 - 1, 8u, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((gather_kernel<U, POLICY, W>), dim3(grid), dim3(256), 0, 0, tab, bytes / 16 - 1, iters, out);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double req = (double)grid * 256 * iters * U;
  printf("{\"table_GiB\": %.3f, \"bytes_per_lookup\": %d, \"in_flight_per_lane\": %d, \"waves_per_simd\": %d, \"policy\": \"%s\", \"G_lookups_per_s\": %.1f, \"ms\": %.2f}\n",
         bytes / 1073741824.0, W, U, wg_per_cu, pname, req / (ms * 1e-3) / 1e9, ms);
  fflush(stdout);
  return 0;
}

int main()
{
  uint32_t *out;
  CK(hipMalloc((void **)&out, 4096));
  for (uint64_t gib : {8ull, 1ull}) {
    const uint64_t bytes = gib << 30;
    uint8_t *tab;
    CK(hipMalloc((void **)&tab, bytes));
    CK(hipMemset(tab, 1, bytes));
    for (int wg : {4, 8}) {   // 256-thread workgroups per CU = waves per SIMD
      run<1, 0, 16>(tab, bytes, wg, out, "plain");
      run<5, 0, 16>(tab, bytes, wg, out, "plain");
      run<10, 0, 16>(tab, bytes, wg, out, "plain");
      run<5, 1, 16>(tab, bytes, wg, out, "nt");
      run<10, 1, 16>(tab, bytes, wg, out, "nt");
      run<5, 0, 8>(tab, bytes, wg, out, "plain");
      run<5, 0, 4>(tab, bytes, wg, out, "plain");
      run<10, 1, 4>(tab, bytes, wg, out, "nt");
    }
    CK(hipFree(tab));
  }
  // a table that fits the Infinity Cache / the L2s
  for (uint64_t mib : {128ull, 16ull, 8ull, 4ull, 2ull, 1ull}) {
    const uint64_t bytes = mib << 20;
    uint8_t *tab;
    CK(hipMalloc((void **)&tab, bytes));
    CK(hipMemset(tab, 1, bytes));
    run<5, 0, 16>(tab, bytes, 8, out, "plain");
    run<10, 0, 16>(tab, bytes, 8, out, "plain");
    run<5, 0, 8>(tab, bytes, 8, out, "plain");
    run<5, 0, 4>(tab, bytes, 8, out, "plain");
    run<10, 0, 4>(tab, bytes, 6, out, "plain");
    CK(hipFree(tab));
  }
  return 0;
}
