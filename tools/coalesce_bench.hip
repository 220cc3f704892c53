// coalesce_bench.hip -- what a wave-level load costs by access shape, in units of "random 16-byte lookups" (tools; not part
// of the library).  Every wave-iteration draws a random base in a 2 GiB array and loads one shape from there; the time per
// shape divided by the time per independent random lookup says how many memory-side requests the shape is worth.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const uint32_t *__restrict__ a, uint64_t n_dw, uint32_t iters, uint32_t *__restrict__ out)
{
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t wid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t acc = 0;
  constexpr int F = 8;   // independent loads in flight per wave (the shapes are compared by throughput, not by latency)
  for (uint32_t it = 0; it < iters; it += F) {
    uint64_t bs[F], rs[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
      rs[f] = mix(wid * 1000003ull + it + f);
      bs[f] = __builtin_amdgcn_readfirstlane((uint32_t)(rs[f] & (n_dw / 2 - 1)));   // wave-uniform random dword index
    }
    uint32_t v0[F], v1[F], v2[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
      const uint64_t base = bs[f], r = rs[f];
      v0[f] = v1[f] = v2[f] = 0;
      if (SHAPE == 0) {          // 64 independent random 16-byte buckets
        const uint64_t i = mix(r + lane) & (n_dw / 8 - 1);
        const u32x4 v = reinterpret_cast<const u32x4 *>(a)[i];
        v0[f] = v.x; v1[f] = v.w;
      } else if (SHAPE == 1) {   // 64 lanes x dword, contiguous, unaligned start
        v0[f] = a[base + lane];
      } else if (SHAPE == 2) {   // three dword loads per lane at (x + lane) >> 4, + 0 / 1 / 2
        const uint64_t i = (base + lane) >> 4;
        v0[f] = a[i]; v1[f] = a[i + 1]; v2[f] = a[i + 2];
      } else if (SHAPE == 3) {   // 34 lanes x 16 bytes, contiguous, 4-byte aligned start
        if (lane < 34) { const u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(a + base + 4 * lane); v0[f] = v.x; v1[f] = v.w; }
      } else if (SHAPE == 4) {   // 34 lanes x 16 bytes, contiguous, 16-byte aligned start
        if (lane < 34) { const u32x4 v = *reinterpret_cast<const u32x4 *>(a + (base & ~3ull) + 4 * lane); v0[f] = v.x; v1[f] = v.w; }
      } else if (SHAPE == 5) {   // 16 lanes x 16 bytes = 256 bytes, 16-byte aligned start
        if (lane < 16) { const u32x4 v = *reinterpret_cast<const u32x4 *>(a + (base & ~3ull) + 4 * lane); v0[f] = v.x; v1[f] = v.w; }
      } else if (SHAPE == 6) {   // 64 lanes x dword contiguous, 256-byte aligned start
        v0[f] = a[(base & ~63ull) + lane];
      } else if (SHAPE == 7) {   // 11 lanes x dword contiguous (the packed bases of one mate)
        if (lane < 11) v0[f] = a[base + lane];
      } else if (SHAPE == 8) {   // 64 lanes x 16 bytes contiguous (1 KiB), 16-byte aligned
        const u32x4 v = *reinterpret_cast<const u32x4 *>(a + (base & ~3ull) + 4 * lane); v0[f] = v.x; v1[f] = v.w;
      } else if (SHAPE >= 10 && SHAPE <= 14) {   // runs of R adjacent lanes share a random 128-byte line, each lane one of its eight 16-byte buckets (the minimiser table's probes)
        constexpr int R = SHAPE == 10 ? 2 : (SHAPE == 11 ? 3 : (SHAPE == 12 ? 4 : (SHAPE == 13 ? 5 : 8)));
        const uint64_t line = mix(r + lane / R) & (n_dw / 32 - 1);
        const uint64_t i = line * 8 + (mix(r ^ (lane * 77u)) & 7);
        const u32x4 v = reinterpret_cast<const u32x4 *>(a)[i];
        v0[f] = v.x; v1[f] = v.w;
      } else if (SHAPE == 9) {   // 16 independent random 16-byte buckets (the sample)
        if (lane < 16) { const uint64_t i = mix(r + lane) & (n_dw / 8 - 1); const u32x4 v = reinterpret_cast<const u32x4 *>(a)[i]; v0[f] = v.x; v1[f] = v.w; }
      }
    }
#pragma unroll
    for (int f = 0; f < F; ++f) acc += v0[f] ^ v1[f] ^ v2[f];
  }
  if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

template <int SHAPE>
static int run(const uint32_t *a, uint64_t n_dw, uint32_t *out, const char *what, double *t_rand)
{
  const unsigned grid = 256u * 8u;
  const uint32_t iters = (SHAPE == 0 || SHAPE >= 10) ? 320 : 4000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(256), 0, 0, a, n_dw, 20u, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(256), 0, 0, a, n_dw, iters, out);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double wave_iters = (double)grid * 4 * iters;
  const double ps = ms * 1e9 / wave_iters;       // picoseconds of the whole GPU per wave-iteration
  if (SHAPE == 0) *t_rand = ps / 64.0;
  printf("{\"shape\": %d, \"what\": \"%s\", \"ps_per_wave_load\": %.1f, \"worth_random_lookups\": %.2f}\n", SHAPE, what, ps, ps / *t_rand);
  fflush(stdout);
  return 0;
}

int main()
{
  const uint64_t bytes = 2ull << 30, n_dw = bytes / 4;
  uint32_t *a, *out;
  CK(hipMalloc((void **)&a, bytes));
  CK(hipMemset(a, 1, bytes));
  CK(hipMalloc((void **)&out, 4096));
  double t = 1;
  run<0>(a, n_dw, out, "64 random 16-B buckets", &t);
  run<9>(a, n_dw, out, "16 random 16-B buckets (16 lanes)", &t);
  run<1>(a, n_dw, out, "64 lanes x dword contiguous, unaligned", &t);
  run<6>(a, n_dw, out, "64 lanes x dword contiguous, 256-B aligned", &t);
  run<2>(a, n_dw, out, "3 x dword per lane under 64 consecutive 2-bit positions", &t);
  run<3>(a, n_dw, out, "34 lanes x 16 B contiguous, 4-B aligned", &t);
  run<4>(a, n_dw, out, "34 lanes x 16 B contiguous, 16-B aligned", &t);
  run<5>(a, n_dw, out, "16 lanes x 16 B contiguous, 16-B aligned", &t);
  run<7>(a, n_dw, out, "11 lanes x dword contiguous", &t);
  run<8>(a, n_dw, out, "64 lanes x 16 B contiguous (1 KiB)", &t);
  run<10>(a, n_dw, out, "64 lanes x 16 B, runs of 2 adjacent lanes in one random line (32 lines)", &t);
  run<11>(a, n_dw, out, "64 lanes x 16 B, runs of 3 adjacent lanes in one random line (22 lines)", &t);
  run<12>(a, n_dw, out, "64 lanes x 16 B, runs of 4 adjacent lanes in one random line (16 lines)", &t);
  run<13>(a, n_dw, out, "64 lanes x 16 B, runs of 5 adjacent lanes in one random line (13 lines)", &t);
  run<14>(a, n_dw, out, "64 lanes x 16 B, runs of 8 adjacent lanes in one random line (8 lines)", &t);
  return 0;
}
